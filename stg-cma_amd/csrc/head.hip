// Small kernels of the AVQA question-answering head (AVQA/model/Swin_AVQAModel_V1.py:37-59 QstEncoder, :1768-1903 forward):
// everything here works on a few hundred rows of 1536 channels -- launch-bound bookkeeping around the GEMMs, written as plain
// VALU kernels with fp32 arithmetic on bf16 storage.
#include <math.h>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

inline unsigned grid_for(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------ unary / binary
__global__ void unary_fwd_kernel(int op, const bf16_t* x, bf16_t* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = bf2f(x[i]);
        y[i] = f2bf(op == 0 ? fmaxf(v, 0.f) : tanhf(v));
    }
}
// derivative from the INPUT x: relu' = [x > 0], tanh' = 1 - tanh(x)^2 in fp32 (1 - y^2 from the bf16-rounded OUTPUT loses every
// significant digit where tanh saturates)
__global__ void unary_bwd_kernel(int op, const bf16_t* x, const bf16_t* dy, bf16_t* dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = bf2f(x[i]), g = bf2f(dy[i]);
        const float t = tanhf(v);
        dx[i] = f2bf(op == 0 ? (v > 0.f ? g : 0.f) : g * (1.0f - t * t));
    }
}
// The same two on 16-byte pieces, two pieces per lane and trip (round 5b: the scalar forms above moved 2 bytes per lane and load -- and the backward
// paid a tanhf per element even for ReLU; the AVS decoder's ReLUs over 8 M x 32 / 2 M x 256 maps cost 2.8 ms per step).  16-byte aligned pointers;
// the caller handles the last n % 8 elements with the scalar kernels.
__global__ void __launch_bounds__(256) unary_fwd_v8_kernel(int op, const bf16_t* x, bf16_t* y, int64_t n8) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += 2 * stride) {
        const bool two = i + stride < n8;
        const u16x8 a = reinterpret_cast<const u16x8*>(x)[i];
        const u16x8 b = two ? reinterpret_cast<const u16x8*>(x)[i + stride] : a;
        u16x8 oa, ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float va = bf2f(a.v[j]), vb = bf2f(b.v[j]);
            oa.v[j] = f2bf(op == 0 ? fmaxf(va, 0.f) : tanhf(va));
            ob.v[j] = f2bf(op == 0 ? fmaxf(vb, 0.f) : tanhf(vb));
        }
        reinterpret_cast<u16x8*>(y)[i] = oa;
        if (two) reinterpret_cast<u16x8*>(y)[i + stride] = ob;
    }
}
__global__ void __launch_bounds__(256) unary_bwd_v8_kernel(int op, const bf16_t* x, const bf16_t* dy, bf16_t* dx, int64_t n8) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += 2 * stride) {
        const bool two = i + stride < n8;
        const int64_t i1 = two ? i + stride : i;
        const u16x8 xa = reinterpret_cast<const u16x8*>(x)[i], ga = reinterpret_cast<const u16x8*>(dy)[i];
        const u16x8 xb = reinterpret_cast<const u16x8*>(x)[i1], gb = reinterpret_cast<const u16x8*>(dy)[i1];
        u16x8 oa, ob;
        if (op == 0) {                                       // kernel-uniform
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                oa.v[j] = bf2f(xa.v[j]) > 0.f ? ga.v[j] : (bf16_t)0;          // g or +0: the bf16 bits pass through
                ob.v[j] = bf2f(xb.v[j]) > 0.f ? gb.v[j] : (bf16_t)0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float ta = tanhf(bf2f(xa.v[j])), tb = tanhf(bf2f(xb.v[j]));
                oa.v[j] = f2bf(bf2f(ga.v[j]) * (1.0f - ta * ta));
                ob.v[j] = f2bf(bf2f(gb.v[j]) * (1.0f - tb * tb));
            }
        }
        reinterpret_cast<u16x8*>(dx)[i] = oa;
        if (two) reinterpret_cast<u16x8*>(dx)[i1] = ob;
    }
}
__global__ void mul_kernel(const bf16_t* a, const bf16_t* b, bf16_t* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f2bf(bf2f(a[i]) * bf2f(b[i]));
}

// ------------------------------------------------------------------------------------------------ embedding
__global__ void embed_fwd_kernel(const float* table, const int64_t* idx, bf16_t* out, int64_t n, int V, int E) {
    const int64_t total = n * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / E;
        int64_t ix = idx[r];
        ix = ix < 0 ? 0 : (ix >= V ? V - 1 : ix);
        out[i] = f2bf(table[ix * E + (i - r * E)]);
    }
}
__global__ void embed_bwd_kernel(const bf16_t* dout, const int64_t* idx, float* dtable, int64_t n, int V, int E) {
    const int64_t total = n * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / E;
        int64_t ix = idx[r];
        ix = ix < 0 ? 0 : (ix >= V ? V - 1 : ix);
        atomicAdd(dtable + ix * E + (i - r * E), bf2f(dout[i]));
    }
}

// ------------------------------------------------------------------------------------------------ LSTM cell
// gates fp32 [B, 4H] in torch's (i, f, g, o) order; c fp32; h bf16.
__global__ void lstm_cell_fwd_kernel(const float* gates, const float* c_prev, float* c, bf16_t* h, int64_t B, int H) {
    const int64_t total = B * H;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = id / H;
        const int j = (int)(id - b * H);
        const float* g4 = gates + b * 4 * H;
        const float i = sigmoidf_(g4[j]), f = sigmoidf_(g4[H + j]), g = tanhf(g4[2 * H + j]), o = sigmoidf_(g4[3 * H + j]);
        const float cn = f * c_prev[id] + i * g;
        c[id] = cn;
        h[id] = f2bf(o * tanhf(cn));
    }
}
// dgates (bf16, feeds the dgrad / wgrad GEMMs), dc_prev fp32; dh bf16 or NULL, dc fp32 or NULL (the step's two incoming grads)
__global__ void lstm_cell_bwd_kernel(const float* gates, const float* c_prev, const float* c, const bf16_t* dh, const float* dc_in,
                                     bf16_t* dgates, float* dc_prev, int64_t B, int H) {
    const int64_t total = B * H;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = id / H;
        const int j = (int)(id - b * H);
        const float* g4 = gates + b * 4 * H;
        const float i = sigmoidf_(g4[j]), f = sigmoidf_(g4[H + j]), g = tanhf(g4[2 * H + j]), o = sigmoidf_(g4[3 * H + j]);
        const float tc = tanhf(c[id]);
        const float dhv = dh ? bf2f(dh[id]) : 0.f;
        const float dcv = (dc_in ? dc_in[id] : 0.f) + dhv * o * (1.0f - tc * tc);
        bf16_t* d4 = dgates + b * 4 * H;
        d4[j] = f2bf(dcv * g * i * (1.0f - i));
        d4[H + j] = f2bf(dcv * c_prev[id] * f * (1.0f - f));
        d4[2 * H + j] = f2bf(dcv * i * (1.0f - g * g));
        d4[3 * H + j] = f2bf(dhv * tc * o * (1.0f - o));
        dc_prev[id] = dcv * f;
    }
}

// ------------------------------------------------------------------------------------------------ block reductions
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum<64>(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    return t;
}

// ------------------------------------------------------------------------------------------------ audio-visual grounding
// One block per frame (Swin_AVQAModel_V1.py:1797-1815): V fp32 [n, C] (49 visual tokens), a bf16 [C] (audio feature)
//   vmean = mean_j V_j;  vh_j = V_j / max(|V_j|, eps);  ah = a / max(|a|, eps);  s_j = vh_j . ah;  p = softmax(s);  grd = sum_j p_j vh_j
// saved for the backward: p [n], rnorm [n] = 1 / max(|V_j|, eps), ra = 1 / max(|a|, eps).
constexpr int GMAXN = 64;
__global__ void __launch_bounds__(256) grounding_fwd_kernel(const float* V, const bf16_t* a, bf16_t* vmean, bf16_t* grd, float* p_out,
                                                            float* rnorm_out, float* ra_out, int n, int C) {
    __shared__ float red[4];
    __shared__ float s_p[GMAXN], s_rn[GMAXN];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* Vf = V + (int64_t)f * n * C;
    const bf16_t* af = a + (int64_t)f * C;
    float an = 0.f;
    for (int c = tid; c < C; c += 256) { const float x = bf2f(af[c]); an += x * x; }
    an = block_sum<256>(an, red);
    const float ra = 1.0f / fmaxf(sqrtf(an), 1e-12f);
    for (int j = 0; j < n; ++j) {
        float nn = 0.f, dot = 0.f;
        for (int c = tid; c < C; c += 256) { const float x = Vf[(int64_t)j * C + c]; nn += x * x; dot += x * bf2f(af[c]); }
        nn = block_sum<256>(nn, red);
        dot = block_sum<256>(dot, red);
        const float rn = 1.0f / fmaxf(sqrtf(nn), 1e-12f);
        if (tid == 0) { s_rn[j] = rn; s_p[j] = dot * rn * ra; }
    }
    __syncthreads();
    float mx = -1e30f;
    for (int j = 0; j < n; ++j) mx = fmaxf(mx, s_p[j]);
    float sum = 0.f;
    for (int j = 0; j < n; ++j) sum += __expf(s_p[j] - mx);
    __syncthreads();
    if (tid < n) s_p[tid] = __expf(s_p[tid] - mx) / sum;
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float m = 0.f, g = 0.f;
        for (int j = 0; j < n; ++j) { const float x = Vf[(int64_t)j * C + c]; m += x; g += s_p[j] * s_rn[j] * x; }
        vmean[(int64_t)f * C + c] = f2bf(m / n);
        grd[(int64_t)f * C + c] = f2bf(g);
    }
    if (tid < n) { p_out[(int64_t)f * n + tid] = s_p[tid]; rnorm_out[(int64_t)f * n + tid] = s_rn[tid]; }
    if (tid == 0) ra_out[f] = ra;
}

// dV fp32 [n, C] (may be NULL: the negative clip carries no gradient), da bf16 [C]
__global__ void __launch_bounds__(256) grounding_bwd_kernel(const float* V, const bf16_t* a, const float* p, const float* rnorm,
                                                            const float* ra_in, const bf16_t* dvmean, const bf16_t* dgrd, float* dV,
                                                            bf16_t* da, int n, int C) {
    __shared__ float red[4];
    __shared__ float s_p[GMAXN], s_rn[GMAXN], s_t[GMAXN], s_ds[GMAXN], s_u[GMAXN];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* Vf = V + (int64_t)f * n * C;
    const bf16_t* af = a + (int64_t)f * C;
    const bf16_t* dg = dgrd + (int64_t)f * C;
    const float ra = ra_in[f];
    if (tid < n) { s_p[tid] = p[(int64_t)f * n + tid]; s_rn[tid] = rnorm[(int64_t)f * n + tid]; }
    __syncthreads();
    // t_j = vh_j . dgrd
    for (int j = 0; j < n; ++j) {
        float t = 0.f;
        for (int c = tid; c < C; c += 256) t += Vf[(int64_t)j * C + c] * bf2f(dg[c]);
        t = block_sum<256>(t, red);
        if (tid == 0) s_t[j] = t * s_rn[j];
    }
    __syncthreads();
    float pt = 0.f;
    for (int j = 0; j < n; ++j) pt += s_p[j] * s_t[j];
    __syncthreads();
    if (tid < n) s_ds[tid] = s_p[tid] * (s_t[tid] - pt);                       // ds_j
    __syncthreads();
    // u_j = vh_j . dvh_j with dvh_j = p_j dgrd + ds_j ah   (for the normalisation backward)
    for (int j = 0; j < n; ++j) {
        float u = 0.f;
        for (int c = tid; c < C; c += 256) {
            const float vh = Vf[(int64_t)j * C + c] * s_rn[j];
            u += vh * (s_p[j] * bf2f(dg[c]) + s_ds[j] * bf2f(af[c]) * ra);
        }
        u = block_sum<256>(u, red);
        if (tid == 0) s_u[j] = u;
    }
    __syncthreads();
    // dah_c = sum_j ds_j vh_j[c];  w = ah . dah
    float w = 0.f;
    for (int c = tid; c < C; c += 256) {
        float dah = 0.f;
        for (int j = 0; j < n; ++j) dah += s_ds[j] * s_rn[j] * Vf[(int64_t)j * C + c];
        w += bf2f(af[c]) * ra * dah;
    }
    w = block_sum<256>(w, red);
    for (int c = tid; c < C; c += 256) {
        const float ah = bf2f(af[c]) * ra, dgc = bf2f(dg[c]);
        float dah = 0.f;
        for (int j = 0; j < n; ++j) {
            const float x = Vf[(int64_t)j * C + c];
            dah += s_ds[j] * s_rn[j] * x;
            if (dV) {
                const float vh = x * s_rn[j];
                const float dvh = s_p[j] * dgc + s_ds[j] * ah;
                dV[((int64_t)f * n + j) * C + c] = (dvh - vh * s_u[j]) * s_rn[j] + (dvmean ? bf2f(dvmean[(int64_t)f * C + c]) / n : 0.f);
            }
        }
        da[(int64_t)f * C + c] = f2bf((dah - ah * w) * ra);
    }
}

// ------------------------------------------------------------------------------------------------ single-query multi-head attention
// nn.MultiheadAttention(E, H) core with ONE query per batch element (Swin_AVQAModel_V1.py:1866-1880): q bf16 [B, E] (projected),
// k, v bf16 [T, B, E] (projected; row t*B + b), probabilities saved fp32 [B, H, T] AFTER the optional dropout mask
// drop [B, H, T] (Bernoulli(keep) / keep, fp32) -- o[b, h] = sum_t (p * drop)[t] v[t, b, h].  One block per (b, h).
constexpr int MAXT = 64;
__global__ void __launch_bounds__(128) mha1_fwd_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const float* drop, bf16_t* o,
                                                       float* p_out, int B, int H, int T, int hd, float scale) {
    __shared__ float red[2];
    __shared__ float s_s[MAXT];
    const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
    const int E = H * hd;
    const bf16_t* qb = q + (int64_t)b * E + h * hd;
    for (int t = 0; t < T; ++t) {
        const bf16_t* kb = k + ((int64_t)t * B + b) * E + h * hd;
        float d = 0.f;
        for (int c = tid; c < hd; c += 128) d += bf2f(qb[c]) * bf2f(kb[c]);
        d = block_sum<128>(d, red);
        if (tid == 0) s_s[t] = d * scale;
    }
    __syncthreads();
    float mx = -1e30f;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, s_s[t]);
    float sum = 0.f;
    for (int t = 0; t < T; ++t) sum += __expf(s_s[t] - mx);
    __syncthreads();
    if (tid < T) {
        const float pv = __expf(s_s[tid] - mx) / sum;
        p_out[((int64_t)b * H + h) * T + tid] = pv;
        s_s[tid] = pv * (drop ? drop[((int64_t)b * H + h) * T + tid] : 1.0f);
    }
    __syncthreads();
    for (int c = tid; c < hd; c += 128) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += s_s[t] * bf2f(v[((int64_t)t * B + b) * E + h * hd + c]);
        o[(int64_t)b * E + h * hd + c] = f2bf(acc);
    }
}
__global__ void __launch_bounds__(128) mha1_bwd_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const float* drop, const float* p,
                                                       const bf16_t* dout, bf16_t* dq, bf16_t* dk, bf16_t* dv, int B, int H, int T, int hd,
                                                       float scale) {
    __shared__ float red[2];
    __shared__ float s_p[MAXT], s_dp[MAXT], s_ds[MAXT];
    const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
    const int E = H * hd;
    const bf16_t* qb = q + (int64_t)b * E + h * hd;
    const bf16_t* dob = dout + (int64_t)b * E + h * hd;
    if (tid < T) s_p[tid] = p[((int64_t)b * H + h) * T + tid];
    for (int t = 0; t < T; ++t) {
        const bf16_t* vb = v + ((int64_t)t * B + b) * E + h * hd;
        float d = 0.f;
        for (int c = tid; c < hd; c += 128) d += bf2f(dob[c]) * bf2f(vb[c]);
        d = block_sum<128>(d, red);
        if (tid == 0) s_dp[t] = d * (drop ? drop[((int64_t)b * H + h) * T + t] : 1.0f);       // d(p) through the dropout mask
    }
    __syncthreads();
    float pd = 0.f;
    for (int t = 0; t < T; ++t) pd += s_p[t] * s_dp[t];
    __syncthreads();
    if (tid < T) s_ds[tid] = s_p[tid] * (s_dp[tid] - pd) * scale;
    __syncthreads();
    for (int c = tid; c < hd; c += 128) {
        float accq = 0.f;
        const float qc = bf2f(qb[c]), doc = bf2f(dob[c]);
        for (int t = 0; t < T; ++t) {
            const int64_t off = ((int64_t)t * B + b) * E + h * hd + c;
            accq += s_ds[t] * bf2f(k[off]);
            dk[off] = f2bf(s_ds[t] * qc);
            dv[off] = f2bf(s_p[t] * (drop ? drop[((int64_t)b * H + h) * T + t] : 1.0f) * doc);
        }
        dq[(int64_t)b * E + h * hd + c] = f2bf(accq);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int stg_unary_fwd(int op, const void* x, void* y, int64_t numel, void* stream) {
    STG_CHECK(x && y, -1, "stg_unary_fwd: null pointer");
    STG_CHECK(op == 0 || op == 1, -2, "stg_unary_fwd: op must be 0 (relu) or 1 (tanh)");
    if (numel <= 0) return 0;
    int64_t done = 0;
    if (numel >= 8 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const int64_t n8 = numel >> 3;
        hipLaunchKernelGGL(unary_fwd_v8_kernel, dim3(grid_for((n8 + 1) / 2, 256)), dim3(256), 0, ST, op, (const bf16_t*)x, (bf16_t*)y, n8);
        STG_LAUNCH_CHECK();
        done = n8 << 3;
    }
    if (done < numel) {
        hipLaunchKernelGGL(unary_fwd_kernel, dim3(grid_for(numel - done, 256)), dim3(256), 0, ST, op, (const bf16_t*)x + done, (bf16_t*)y + done, numel - done);
        STG_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int stg_unary_bwd(int op, const void* x, const void* dy, void* dx, int64_t numel, void* stream) {
    STG_CHECK(x && dy && dx, -1, "stg_unary_bwd: null pointer");
    STG_CHECK(op == 0 || op == 1, -2, "stg_unary_bwd: op must be 0 (relu) or 1 (tanh)");
    if (numel <= 0) return 0;
    int64_t done = 0;
    if (numel >= 8 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
        const int64_t n8 = numel >> 3;
        hipLaunchKernelGGL(unary_bwd_v8_kernel, dim3(grid_for((n8 + 1) / 2, 256)), dim3(256), 0, ST, op, (const bf16_t*)x, (const bf16_t*)dy,
                           (bf16_t*)dx, n8);
        STG_LAUNCH_CHECK();
        done = n8 << 3;
    }
    if (done < numel) {
        hipLaunchKernelGGL(unary_bwd_kernel, dim3(grid_for(numel - done, 256)), dim3(256), 0, ST, op, (const bf16_t*)x + done,
                           (const bf16_t*)dy + done, (bf16_t*)dx + done, numel - done);
        STG_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int stg_mul(const void* a, const void* b, void* out, int64_t numel, void* stream) {
    STG_CHECK(a && b && out, -1, "stg_mul: null pointer");
    if (numel <= 0) return 0;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(numel, 256)), dim3(256), 0, ST, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_embed_fwd(const float* table, const int64_t* idx, void* out, int64_t n, int V, int E, void* stream) {
    STG_CHECK(table && idx && out, -1, "stg_embed_fwd: null pointer");
    STG_CHECK(n >= 0 && V > 0 && E > 0, -2, "stg_embed_fwd: bad shape");
    if (n == 0) return 0;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(n * E, 256)), dim3(256), 0, ST, table, idx, (bf16_t*)out, n, V, E);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_embed_bwd(const void* dout, const int64_t* idx, float* dtable, int64_t n, int V, int E, void* stream) {
    STG_CHECK(dout && idx && dtable, -1, "stg_embed_bwd: null pointer");
    STG_CHECK(n >= 0 && V > 0 && E > 0, -2, "stg_embed_bwd: bad shape");
    if (n == 0) return 0;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(n * E, 256)), dim3(256), 0, ST, (const bf16_t*)dout, idx, dtable, n, V, E);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_lstm_cell_fwd(const float* gates, const float* c_prev, float* c, void* h, int64_t B, int H, void* stream) {
    STG_CHECK(gates && c_prev && c && h, -1, "stg_lstm_cell_fwd: null pointer");
    STG_CHECK(B >= 0 && H > 0, -2, "stg_lstm_cell_fwd: bad shape");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid_for(B * H, 256)), dim3(256), 0, ST, gates, c_prev, c, (bf16_t*)h, B, H);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_lstm_cell_bwd(const float* gates, const float* c_prev, const float* c, const void* dh, const float* dc,
                                 void* dgates, float* dc_prev, int64_t B, int H, void* stream) {
    STG_CHECK(gates && c_prev && c && dgates && dc_prev, -1, "stg_lstm_cell_bwd: null pointer");
    STG_CHECK(B >= 0 && H > 0, -2, "stg_lstm_cell_bwd: bad shape");
    if (B == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for(B * H, 256)), dim3(256), 0, ST, gates, c_prev, c, (const bf16_t*)dh, dc,
                       (bf16_t*)dgates, dc_prev, B, H);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_grounding_fwd(const float* V, const void* a, void* vmean, void* grd, float* p, float* rnorm, float* ra,
                                 int64_t F, int n, int C, void* stream) {
    STG_CHECK(V && a && vmean && grd && p && rnorm && ra, -1, "stg_grounding_fwd: null pointer");
    STG_CHECK(F >= 0 && F < (1ll << 31) && n >= 1 && n <= GMAXN && C >= 1, -2, "stg_grounding_fwd: bad shape (n <= 64)");
    if (F == 0) return 0;
    hipLaunchKernelGGL(grounding_fwd_kernel, dim3((unsigned)F), dim3(256), 0, ST, V, (const bf16_t*)a, (bf16_t*)vmean, (bf16_t*)grd, p,
                       rnorm, ra, n, C);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_grounding_bwd(const float* V, const void* a, const float* p, const float* rnorm, const float* ra,
                                 const void* dvmean, const void* dgrd, float* dV, void* da, int64_t F, int n, int C, void* stream) {
    STG_CHECK(V && a && p && rnorm && ra && dgrd && da, -1, "stg_grounding_bwd: null pointer");
    STG_CHECK(F >= 0 && F < (1ll << 31) && n >= 1 && n <= GMAXN && C >= 1, -2, "stg_grounding_bwd: bad shape (n <= 64)");
    if (F == 0) return 0;
    hipLaunchKernelGGL(grounding_bwd_kernel, dim3((unsigned)F), dim3(256), 0, ST, V, (const bf16_t*)a, p, rnorm, ra,
                       (const bf16_t*)dvmean, (const bf16_t*)dgrd, dV, (bf16_t*)da, n, C);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_mha1_fwd(const void* q, const void* k, const void* v, const float* drop, void* o, float* p, int B, int H, int T,
                            int hd, float scale, void* stream) {
    STG_CHECK(q && k && v && o && p, -1, "stg_mha1_fwd: null pointer");
    STG_CHECK(B >= 0 && H >= 1 && T >= 1 && T <= MAXT && hd >= 1, -2, "stg_mha1_fwd: bad shape (T <= 64)");
    if (B == 0) return 0;
    hipLaunchKernelGGL(mha1_fwd_kernel, dim3((unsigned)(B * H)), dim3(128), 0, ST, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       drop, (bf16_t*)o, p, B, H, T, hd, scale);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_mha1_bwd(const void* q, const void* k, const void* v, const float* drop, const float* p, const void* dout,
                            void* dq, void* dk, void* dv, int B, int H, int T, int hd, float scale, void* stream) {
    STG_CHECK(q && k && v && p && dout && dq && dk && dv, -1, "stg_mha1_bwd: null pointer");
    STG_CHECK(B >= 0 && H >= 1 && T >= 1 && T <= MAXT && hd >= 1, -2, "stg_mha1_bwd: bad shape (T <= 64)");
    if (B == 0) return 0;
    hipLaunchKernelGGL(mha1_bwd_kernel, dim3((unsigned)(B * H)), dim3(128), 0, ST, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       drop, p, (const bf16_t*)dout, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, B, H, T, hd, scale);
    STG_LAUNCH_CHECK();
    return 0;
}
