// Small HBM-bound element-wise / re-indexing kernels (see include/stgcma.h).  All use 16-byte accesses and a
// capped grid with a grid-stride loop.
#include "common.h"
#include "../../include/stgcma.h"

namespace {

inline unsigned grid_for(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    uint4 o;
    o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
    return o;
}

// ---- out = h + gate * r
__global__ void gate_fwd_kernel(const bf16_t* h, const bf16_t* r, const float* gate, bf16_t* out, int64_t n8, int64_t numel) {
    const float g = gate[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        unpack8(reinterpret_cast<const uint4*>(h)[i], a);
        unpack8(reinterpret_cast<const uint4*>(r)[i], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += g * b[j];
        reinterpret_cast<uint4*>(out)[i] = pack8(a);
    }
    // tail
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        out[i] = f2bf(bf2f(h[i]) + g * bf2f(r[i]));
    }
}

// ---- dr = gate * dout ; dgate += sum(dout * r)
__global__ void gate_bwd_kernel(const bf16_t* dout, const bf16_t* r, const float* gate, bf16_t* dr, float* dgate,
                                int64_t n8, int64_t numel) {
    const float g = gate[0];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        unpack8(reinterpret_cast<const uint4*>(dout)[i], a);
        unpack8(reinterpret_cast<const uint4*>(r)[i], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc += a[j] * b[j]; a[j] *= g; }
        reinterpret_cast<uint4*>(dr)[i] = pack8(a);
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        const float d = bf2f(dout[i]);
        acc += d * bf2f(r[i]);
        dr[i] = f2bf(g * d);
    }
    acc = wave_sum<64>(acc);
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
        atomicAdd(dgate, t);
    }
}

// ---- out = a + b (+ c)
__global__ void add_kernel(const bf16_t* a, const bf16_t* b, const bf16_t* c, bf16_t* out, int64_t n8, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8];
        unpack8(reinterpret_cast<const uint4*>(a)[i], x);
        unpack8(reinterpret_cast<const uint4*>(b)[i], y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        if (c) {
            unpack8(reinterpret_cast<const uint4*>(c)[i], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += y[j];
        }
        reinterpret_cast<uint4*>(out)[i] = pack8(x);
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        out[i] = f2bf(bf2f(a[i]) + bf2f(b[i]) + (c ? bf2f(c[i]) : 0.f));
    }
}

// ---- out = (a + b + c) * z: the join of the three gradient paths into an adapter's hidden state and the activation backward
// behind it (one rounding instead of two, one launch instead of two)
__global__ void add3_mul_kernel(const bf16_t* a, const bf16_t* b, const bf16_t* c, const bf16_t* z, bf16_t* out, int64_t n8, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8];
        unpack8(reinterpret_cast<const uint4*>(a)[i], x);
        unpack8(reinterpret_cast<const uint4*>(b)[i], y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        unpack8(reinterpret_cast<const uint4*>(c)[i], y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        unpack8(reinterpret_cast<const uint4*>(z)[i], y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= y[j];
        reinterpret_cast<uint4*>(out)[i] = pack8(x);
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        out[i] = f2bf((bf2f(a[i]) + bf2f(b[i]) + bf2f(c[i])) * bf2f(z[i]));
    }
}

// ---- the same three kernels on TWO equally sized problems per launch (blockIdx.y): the two directions of a cross-modal pair
// (Swin_AVE.py:750-760 / :799-808: h_v' = h_v + gate_v r_v and h_a' = h_a + gate_a r_a).  At [62 720, 32] a problem is 4 MB and its
// launch is latency, not bandwidth: the step carried 3 x 96 of them.
struct Ew2 {
    const bf16_t* a[2]; const bf16_t* b[2]; const bf16_t* c[2]; const bf16_t* z[2];
    const float* g[2]; bf16_t* out[2]; float* dg[2];
    int64_t n[2];                    // elements per problem (round 5b: the two may differ -- ViT's 197 video and 49 audio tokens)
};
__global__ void gate_fwd2_kernel(Ew2 p) {
    const int y = blockIdx.y;
    const int64_t numel = p.n[y], n8 = numel >> 3;
    const bf16_t* h = p.a[y]; const bf16_t* r = p.b[y]; bf16_t* out = p.out[y];
    const float g = p.g[y][0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        unpack8(reinterpret_cast<const uint4*>(h)[i], a);
        unpack8(reinterpret_cast<const uint4*>(r)[i], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += g * b[j];
        reinterpret_cast<uint4*>(out)[i] = pack8(a);
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        out[i] = f2bf(bf2f(h[i]) + g * bf2f(r[i]));
    }
}
__global__ void gate_bwd2_kernel(Ew2 p) {
    const int y = blockIdx.y;
    const int64_t numel = p.n[y], n8 = numel >> 3;
    const bf16_t* dout = p.a[y]; const bf16_t* r = p.b[y]; bf16_t* dr = p.out[y];
    const float g = p.g[y][0];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        unpack8(reinterpret_cast<const uint4*>(dout)[i], a);
        unpack8(reinterpret_cast<const uint4*>(r)[i], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc += a[j] * b[j]; a[j] *= g; }
        reinterpret_cast<uint4*>(dr)[i] = pack8(a);
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        const float d = bf2f(dout[i]);
        acc += d * bf2f(r[i]);
        dr[i] = f2bf(g * d);
    }
    acc = wave_sum<64>(acc);
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
        atomicAdd(p.dg[y], t);
    }
}
__global__ void add3_mul2_kernel(Ew2 p) {
    const int y = blockIdx.y;
    const int64_t numel = p.n[y], n8 = numel >> 3;
    const bf16_t* a = p.a[y]; const bf16_t* b = p.b[y]; const bf16_t* c = p.c[y]; const bf16_t* z = p.z[y]; bf16_t* out = p.out[y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], v[8];
        unpack8(reinterpret_cast<const uint4*>(a)[i], x);
        unpack8(reinterpret_cast<const uint4*>(b)[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += v[j];
        if (c) {                                           // block-uniform: c == NULL -> (a + b) z
            unpack8(reinterpret_cast<const uint4*>(c)[i], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += v[j];
        }
        unpack8(reinterpret_cast<const uint4*>(z)[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= v[j];
        reinterpret_cast<uint4*>(out)[i] = pack8(x);
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        out[i] = f2bf((bf2f(a[i]) + bf2f(b[i]) + (c ? bf2f(c[i]) : 0.f)) * bf2f(z[i]));
    }
}

// ---- dz = dh * act'(z)
__global__ void act_bwd_kernel(const bf16_t* dh, const bf16_t* z, bf16_t* dz, int64_t n8, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8];
        unpack8(reinterpret_cast<const uint4*>(dh)[i], x);
        unpack8(reinterpret_cast<const uint4*>(z)[i], y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= y[j];
        reinterpret_cast<uint4*>(dz)[i] = pack8(x);
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        dz[i] = f2bf(bf2f(dh[i]) * bf2f(z[i]));
    }
}

// ---- out = a * mask (fp32 mask)
__global__ void mul_mask_kernel(const bf16_t* a, const float* mask, bf16_t* out, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f2bf(bf2f(a[i]) * mask[i]);
}

// ---- patch gather for kernel==stride convs.  One thread writes 8 output columns (16 bytes).
__global__ void im2col_kernel(const void* x, int x_f32, bf16_t* out, int64_t B, int Cin, int T, int Hin, int Win, int p,
                              int Kpad, int64_t nrows) {
    const int Hp = Hin / p, Wp = Win / p;
    const int K = Cin * p * p;
    const int cpr = Kpad / 8;
    const int64_t total = nrows * cpr;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = id / cpr;
        const int c8 = (int)(id - row * cpr) * 8;
        // row = ((b*T + t)*Hp + hp)*Wp + wp
        const int wp = (int)(row % Wp);
        int64_t r2 = row / Wp;
        const int hp = (int)(r2 % Hp);
        r2 /= Hp;
        const int t = (int)(r2 % T);
        const int64_t b = r2 / T;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = c8 + j;
            float val = 0.f;
            if (col < K) {
                const int c = col / (p * p);
                const int rem = col - c * p * p;
                const int ph = rem / p, pw = rem - ph * p;
                const int64_t off = (((b * Cin + c) * T + t) * Hin + (hp * p + ph)) * (int64_t)Win + (wp * p + pw);
                val = x_f32 ? reinterpret_cast<const float*>(x)[off] : bf2f(reinterpret_cast<const bf16_t*>(x)[off]);
            }
            v[j] = val;
        }
        *reinterpret_cast<uint4*>(out + row * Kpad + c8) = pack8(v);
    }
}

// ---- casts
// every trainable weight matrix of a model in ONE launch: tensor t = blockIdx.y is cast to bf16 at arena + off (row-major,
// leading dimension ld) and, transposed, at arena + offT (leading dimension ldT); the arena is pre-zeroed, so padding columns
// are never written.  Descriptors live in device memory (the table is static for a model: offsets, not pointers, name the
// destinations, so a fresh arena per refresh needs no new table).
__global__ void cast_bf16_multi_kernel(const stg_cast_desc* descs, bf16_t* arena) {
    const stg_cast_desc d = descs[blockIdx.y];
    const int total = d.R * d.C;
    bf16_t* out = arena + d.off;
    bf16_t* outT = arena + d.offT;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / d.C, c = i - r * d.C;
        const bf16_t v = f2bf(d.in[i]);
        out[(int64_t)r * d.ld + c] = v;
        if (d.offT >= 0) outT[(int64_t)c * d.ldT + r] = v;
    }
}

__global__ void add_temporal_kernel(float* x, const float* emb, int64_t total4, int T, int64_t N, int C4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / C4;
        const int c4 = (int)(i - row * C4), t = (int)((row / N) % T);
        float4 v = reinterpret_cast<float4*>(x)[i];
        const float4 e = reinterpret_cast<const float4*>(emb)[(int64_t)t * C4 + c4];
        v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        reinterpret_cast<float4*>(x)[i] = v;
    }
}

// ---- fused multi-tensor Adam (stg_adam_multi)
__global__ void adam_bump_kernel(const stg_adam_desc* descs, int n, const double* hyper) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* st = descs[i].st;
    const double* h = hyper + 8 * descs[i].group;
    const float step = st[0] + 1.0f;
    st[0] = step;
    st[1] = (float)(h[0] / (1.0 - pow(h[1], (double)step)));      // lr / bias_correction1
    st[2] = (float)sqrt(1.0 - pow(h[2], (double)step));           // sqrt(bias_correction2)
}
__global__ void adam_multi_kernel(const stg_adam_desc* descs, const double* hyper) {
    const stg_adam_desc d = descs[blockIdx.y];
    const double* h = hyper + 8 * d.group;
    const float b2 = (float)h[2], eps = (float)h[3], wd = (float)h[4], step_size = d.st[1], bc2s = d.st[2];
    const float w1 = (float)(1.0 - h[1]), w2 = (float)(1.0 - h[2]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = d.p[i];
        float g = d.g[i];
        if (wd != 0.f) g = g + wd * p;
        float m = d.m[i], v = d.v[i];
        m = m + w1 * (g - m);
        v = v * b2 + w2 * g * g;
        const float denom = sqrtf(v) / bc2s + eps;
        d.p[i] = p - step_size * (m / denom);
        d.m[i] = m;
        d.v[i] = v;
    }
}

__global__ void cast_bf16_kernel(const float* in, bf16_t* out, int64_t R, int64_t Cc, int64_t ld) {
    const int64_t total = R * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld, c = i - r * ld;
        out[i] = c < Cc ? f2bf(in[r * Cc + c]) : (bf16_t)0;
    }
}
// out [Cc, ld] = in[R, Cc]^T, columns R..ld-1 zero
__global__ void cast_bf16_t_kernel(const float* in, bf16_t* out, int64_t R, int64_t Cc, int64_t ld) {
    __shared__ float tile[32][33];
    const int64_t bx = (int64_t)blockIdx.x * 32, by = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = by + j, c = bx + tx;
        tile[j][tx] = (r < R && c < Cc) ? in[r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = bx + j, r = by + tx;  // out[c][r]
        if (c < Cc && r < ld) out[c * ld + r] = (r < R) ? f2bf(tile[tx][j]) : (bf16_t)0;
    }
}
__global__ void cast_f32_kernel(const bf16_t* in, float* out, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = bf2f(in[i]);
}

// ---- token mean: in [G, n, C] -> out [G, C].  One thread per (g, 8 columns).
__global__ void meanpool_fwd_kernel(const bf16_t* in, void* out, int out_f32, int64_t ldo, int64_t G, int n, int C) {
    const int c8n = C / 8;
    const int64_t total = G * c8n;
    const float inv = 1.0f / (float)n;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = id / c8n;
        const int c8 = (int)(id - g * c8n) * 8;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(in + (g * n + i) * C + c8), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] *= inv;
        if (out_f32) {
            float* o = reinterpret_cast<float*>(out) + g * ldo + c8;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = s[j];
        } else {
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(out) + g * ldo + c8) = pack8(s);
        }
    }
}
// long groups (n >= 256 tokens, C <= 2048; round 5): ONE WORKGROUP per group -- its 256 threads are C / 8 column pieces x 256 / (C / 8) row lanes,
// a lane sums every R-th row (coalesced: consecutive threads read consecutive 16-byte pieces of a row), the lanes meet in LDS.  The thread-per-
// (group, piece) form above walked 3 136 rows per thread on 2 560 threads: 1.5 ms for the AVS decoder's 160 x 3136 x 128 pooling (84 GB/s).
__global__ void __launch_bounds__(256) meanpool_fwd_blk_kernel(const bf16_t* in, void* out, int out_f32, int64_t ldo, int n, int C) {
    __shared__ float part[256 * 8];
    const int c8n = C / 8;
    const int R = 256 / c8n;                         // row lanes (c8n <= 256, a power of two or not: threads beyond R * c8n idle)
    const int tid = threadIdx.x;
    const int cp = tid % c8n, rl = tid / c8n;
    const int64_t g = blockIdx.x;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < R) {
        const bf16_t* base = in + g * (int64_t)n * C + cp * 8;
        for (int i = rl; i < n; i += R) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(base + (int64_t)i * C), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[tid * 8 + j] = s[j];
    __syncthreads();
    if (tid < c8n) {
        for (int r = 1; r < R; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += part[(r * c8n + tid) * 8 + j];
        const float inv = 1.0f / (float)n;
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] *= inv;
        if (out_f32) {
            float* o = reinterpret_cast<float*>(out) + g * ldo + tid * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = s[j];
        } else {
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(out) + g * ldo + tid * 8) = pack8(s);
        }
    }
}
__global__ void meanpool_bwd_kernel(const bf16_t* dout, int64_t lddo, bf16_t* din, int64_t G, int n, int C) {
    const int c8n = C / 8;
    const int64_t total = G * n * c8n;
    const float inv = 1.0f / (float)n;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = id / c8n;
        const int c8 = (int)(id - row * c8n) * 8;
        const int64_t g = row / n;
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(dout + g * lddo + c8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= inv;
        *reinterpret_cast<uint4*>(din + row * C + c8) = pack8(v);
    }
}

// ---- relative-position bias expansion: out[h, ij] = table[index[ij], h]
__global__ void bias_gather_kernel(const float* table, const int64_t* index, float* out, int L, int H, int nn) {
    const int total = H * nn;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        const int h = id / nn, ij = id - h * nn;
        int64_t ix = index[ij];
        if (ix < 0) ix = 0;
        if (ix >= L) ix = L - 1;
        out[id] = table[ix * H + h];
    }
}
__global__ void bias_scatter_kernel(const float* dbias, const int64_t* index, float* dtable, int L, int H, int nn) {
    const int total = H * nn;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        const int h = id / nn, ij = id - h * nn;
        int64_t ix = index[ij];
        if (ix < 0 || ix >= L) continue;
        atomicAdd(dtable + ix * H + h, dbias[id]);
    }
}

// ---- ViT token assembly: out[bt, 0] = cls + pos[0] + temb[t]; out[bt, 1+i] = patch[bt, i] + pos[1+i] + temb[t]   (fp32)
__global__ void vit_embed_kernel(const bf16_t* patch, const float* cls, const float* pos, const float* temb, float* out,
                                 int64_t BT, int T, int np, int D) {
    const int d4 = D / 4;
    const int64_t total = BT * (np + 1) * d4;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(id % d4) * 4;
        const int64_t row = id / d4;
        const int n = (int)(row % (np + 1));
        const int64_t bt = row / (np + 1);
        const int t = (int)(bt % T);
        const float4 pe = *reinterpret_cast<const float4*>(pos + (int64_t)n * D + c);
        const float4 te = *reinterpret_cast<const float4*>(temb + (int64_t)t * D + c);
        float4 v;
        if (n == 0) {
            v = *reinterpret_cast<const float4*>(cls + c);
        } else {
            const u16x4 q = *reinterpret_cast<const u16x4*>(patch + (bt * np + (n - 1)) * D + c);
            v = make_float4(bf2f(q.v[0]), bf2f(q.v[1]), bf2f(q.v[2]), bf2f(q.v[3]));
        }
        v.x += pe.x + te.x; v.y += pe.y + te.y; v.z += pe.z + te.z; v.w += pe.w + te.w;
        *reinterpret_cast<float4*>(out + row * D + c) = v;
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int stg_vit_embed(const void* patch, const float* cls, const float* pos, const float* temb, float* out, int64_t BT,
                             int T, int np, int D, void* stream) {
    STG_CHECK(patch && cls && pos && temb && out, -1, "stg_vit_embed: null pointer");
    STG_CHECK(BT >= 0 && T > 0 && np > 0 && D > 0 && D % 4 == 0 && BT % T == 0, -2, "stg_vit_embed: bad shape");
    if (BT == 0) return 0;
    hipLaunchKernelGGL(vit_embed_kernel, dim3(grid_for(BT * (np + 1) * (D / 4), 256)), dim3(256), 0, ST, (const bf16_t*)patch,
                       cls, pos, temb, out, BT, T, np, D);
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_gate_fwd(const void* h, const void* r, const float* gate, void* out, int64_t numel, void* stream) {
    STG_CHECK(h && r && gate && out, -1, "stg_gate_fwd: null pointer");
    STG_CHECK((((uintptr_t)h | (uintptr_t)r | (uintptr_t)out) & 15) == 0, -2, "stg_gate_fwd: pointers must be 16-byte aligned");
    if (numel <= 0) return 0;
    const int64_t n8 = numel >> 3;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, ST, (const bf16_t*)h, (const bf16_t*)r, gate,
                       (bf16_t*)out, n8, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_gate_bwd(const void* dout, const void* r, const float* gate, void* dr, float* dgate, int64_t numel,
                            void* stream) {
    STG_CHECK(dout && r && gate && dr && dgate, -1, "stg_gate_bwd: null pointer");
    STG_CHECK((((uintptr_t)dout | (uintptr_t)r | (uintptr_t)dr) & 15) == 0, -2, "stg_gate_bwd: pointers must be 16-byte aligned");
    if (numel <= 0) return 0;
    const int64_t n8 = numel >> 3;
    unsigned gb = grid_for(n8, 256);
    if (gb > 256) gb = 256;      // one memory-side atomic per block on ONE address: 4096 of them serialised into ~20 us
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(gb), dim3(256), 0, ST, (const bf16_t*)dout, (const bf16_t*)r,
                       gate, (bf16_t*)dr, dgate, n8, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_add(const void* a, const void* b, const void* c, void* out, int64_t numel, void* stream) {
    STG_CHECK(a && b && out, -1, "stg_add: null pointer");
    STG_CHECK((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15) == 0, -2, "stg_add: pointers must be 16-byte aligned");
    if (numel <= 0) return 0;
    const int64_t n8 = numel >> 3;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, ST, (const bf16_t*)a, (const bf16_t*)b,
                       (const bf16_t*)c, (bf16_t*)out, n8, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_add3_mul(const void* a, const void* b, const void* c, const void* z, void* out, int64_t numel, void* stream) {
    STG_CHECK(a && b && c && z && out, -1, "stg_add3_mul: null pointer");
    STG_CHECK((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)z | (uintptr_t)out) & 15) == 0, -2, "stg_add3_mul: pointers must be 16-byte aligned");
    if (numel <= 0) return 0;
    const int64_t n8 = numel >> 3;
    hipLaunchKernelGGL(add3_mul_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, ST, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c,
                       (const bf16_t*)z, (bf16_t*)out, n8, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
extern "C" int stg_gate_fwd2n(const void* h0, const void* r0, const float* g0, void* out0, int64_t numel0, const void* h1, const void* r1,
                              const float* g1, void* out1, int64_t numel1, void* stream) {
    STG_CHECK(h0 && r0 && g0 && out0 && h1 && r1 && g1 && out1, -1, "stg_gate_fwd2: null pointer");
    STG_CHECK(al16(h0) && al16(r0) && al16(out0) && al16(h1) && al16(r1) && al16(out1), -2, "stg_gate_fwd2: pointers must be 16-byte aligned");
    if (numel0 <= 0 && numel1 <= 0) return 0;
    Ew2 p = {};
    p.a[0] = (const bf16_t*)h0; p.b[0] = (const bf16_t*)r0; p.g[0] = g0; p.out[0] = (bf16_t*)out0; p.n[0] = numel0 < 0 ? 0 : numel0;
    p.a[1] = (const bf16_t*)h1; p.b[1] = (const bf16_t*)r1; p.g[1] = g1; p.out[1] = (bf16_t*)out1; p.n[1] = numel1 < 0 ? 0 : numel1;
    const int64_t n8 = (p.n[0] > p.n[1] ? p.n[0] : p.n[1]) >> 3;
    hipLaunchKernelGGL(gate_fwd2_kernel, dim3(grid_for(n8 > 0 ? n8 : 1, 256), 2), dim3(256), 0, ST, p);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_gate_fwd2(const void* h0, const void* r0, const float* g0, void* out0, const void* h1, const void* r1, const float* g1,
                             void* out1, int64_t numel, void* stream) {
    return stg_gate_fwd2n(h0, r0, g0, out0, numel, h1, r1, g1, out1, numel, stream);
}
extern "C" int stg_gate_bwd2n(const void* dout0, const void* r0, const float* g0, void* dr0, float* dgate0, int64_t numel0, const void* dout1,
                              const void* r1, const float* g1, void* dr1, float* dgate1, int64_t numel1, void* stream) {
    STG_CHECK(dout0 && r0 && g0 && dr0 && dgate0 && dout1 && r1 && g1 && dr1 && dgate1, -1, "stg_gate_bwd2: null pointer");
    STG_CHECK(al16(dout0) && al16(r0) && al16(dr0) && al16(dout1) && al16(r1) && al16(dr1), -2, "stg_gate_bwd2: pointers must be 16-byte aligned");
    if (numel0 <= 0 && numel1 <= 0) return 0;
    Ew2 p = {};
    p.a[0] = (const bf16_t*)dout0; p.b[0] = (const bf16_t*)r0; p.g[0] = g0; p.out[0] = (bf16_t*)dr0; p.dg[0] = dgate0; p.n[0] = numel0 < 0 ? 0 : numel0;
    p.a[1] = (const bf16_t*)dout1; p.b[1] = (const bf16_t*)r1; p.g[1] = g1; p.out[1] = (bf16_t*)dr1; p.dg[1] = dgate1; p.n[1] = numel1 < 0 ? 0 : numel1;
    const int64_t n8 = (p.n[0] > p.n[1] ? p.n[0] : p.n[1]) >> 3;
    unsigned gb = grid_for(n8 > 0 ? n8 : 1, 256);
    if (gb > 256) gb = 256;      // one memory-side atomic per block and address (see stg_gate_bwd)
    hipLaunchKernelGGL(gate_bwd2_kernel, dim3(gb, 2), dim3(256), 0, ST, p);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_gate_bwd2(const void* dout0, const void* r0, const float* g0, void* dr0, float* dgate0, const void* dout1, const void* r1,
                             const float* g1, void* dr1, float* dgate1, int64_t numel, void* stream) {
    return stg_gate_bwd2n(dout0, r0, g0, dr0, dgate0, numel, dout1, r1, g1, dr1, dgate1, numel, stream);
}
extern "C" int stg_add3_mul2n(const void* a0, const void* b0, const void* c0, const void* z0, void* out0, int64_t numel0, const void* a1,
                              const void* b1, const void* c1, const void* z1, void* out1, int64_t numel1, void* stream) {
    STG_CHECK(a0 && b0 && z0 && out0 && a1 && b1 && z1 && out1 && ((c0 == nullptr) == (c1 == nullptr)), -1, "stg_add3_mul2: null pointer (c0 / c1 may be NULL together)");
    STG_CHECK(al16(a0) && al16(b0) && al16(c0) && al16(z0) && al16(out0) && al16(a1) && al16(b1) && al16(c1) && al16(z1) && al16(out1), -2,
              "stg_add3_mul2: pointers must be 16-byte aligned");
    if (numel0 <= 0 && numel1 <= 0) return 0;
    Ew2 p = {};
    p.a[0] = (const bf16_t*)a0; p.b[0] = (const bf16_t*)b0; p.c[0] = (const bf16_t*)c0; p.z[0] = (const bf16_t*)z0; p.out[0] = (bf16_t*)out0; p.n[0] = numel0 < 0 ? 0 : numel0;
    p.a[1] = (const bf16_t*)a1; p.b[1] = (const bf16_t*)b1; p.c[1] = (const bf16_t*)c1; p.z[1] = (const bf16_t*)z1; p.out[1] = (bf16_t*)out1; p.n[1] = numel1 < 0 ? 0 : numel1;
    const int64_t n8 = (p.n[0] > p.n[1] ? p.n[0] : p.n[1]) >> 3;
    hipLaunchKernelGGL(add3_mul2_kernel, dim3(grid_for(n8 > 0 ? n8 : 1, 256), 2), dim3(256), 0, ST, p);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_add3_mul2(const void* a0, const void* b0, const void* c0, const void* z0, void* out0, const void* a1, const void* b1,
                             const void* c1, const void* z1, void* out1, int64_t numel, void* stream) {
    return stg_add3_mul2n(a0, b0, c0, z0, out0, numel, a1, b1, c1, z1, out1, numel, stream);
}
// Test aid (tests/test_lds_poison_gpu.py, STG_LDS_POISON=1): fill the LDS of every CU with NaN bit patterns (0x7FC07FC0: a NaN as fp32 and as two
// bf16 values).  A kernel that reads an LDS byte nobody wrote -- harmless while the previous tenant of the CU left finite data there, e.g. padded
// rows whose probabilities are zero: 0 x NaN = NaN -- then shows up as a non-finite result (round 4: that is how a 2-rank rehearsal on one
// GPU, where the other PROCESS's kernels leave their data in the LDS, produced NaN losses once in ~8 runs).
__global__ void __launch_bounds__(256) lds_poison_kernel(int words) {
    extern __shared__ uint32_t lds_all[];
    volatile uint32_t* s = lds_all;
    for (int i = threadIdx.x; i < words; i += 256) s[i] = 0x7FC07FC0u;
    __syncthreads();
}
extern "C" int stg_debug_poison_lds(void* stream) {
    static std::atomic<uint64_t> done{0};
    const int bytes = 160 * 1024;
    STG_CHECK(stg_reserve_lds(lds_poison_kernel, bytes, done), -101, "stg_debug_poison_lds: cannot reserve 160 KiB of LDS");
    hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(256), bytes, ST, bytes / 4);      // one workgroup owns a CU's whole LDS: 4 rounds over 256 CUs
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_act_bwd(const void* dh, const void* z, void* dz, int64_t numel, void* stream) {
    STG_CHECK(dh && z && dz, -1, "stg_act_bwd: null pointer");
    STG_CHECK((((uintptr_t)dh | (uintptr_t)z | (uintptr_t)dz) & 15) == 0, -2, "stg_act_bwd: pointers must be 16-byte aligned");
    if (numel <= 0) return 0;
    const int64_t n8 = numel >> 3;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, ST, (const bf16_t*)dh, (const bf16_t*)z,
                       (bf16_t*)dz, n8, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_mul_mask(const void* a, const float* mask, void* out, int64_t numel, void* stream) {
    STG_CHECK(a && mask && out, -1, "stg_mul_mask: null pointer");
    if (numel <= 0) return 0;
    hipLaunchKernelGGL(mul_mask_kernel, dim3(grid_for(numel, 256)), dim3(256), 0, ST, (const bf16_t*)a, mask, (bf16_t*)out, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_im2col_patch(const void* x, int x_dtype, void* out, int64_t B, int Cin, int T, int Hin, int Win, int p,
                                int Kpad, void* stream) {
    STG_CHECK(x && out, -1, "stg_im2col_patch: null pointer");
    STG_CHECK(x_dtype == STG_F32 || x_dtype == STG_BF16, -3, "stg_im2col_patch: unsupported dtype");
    STG_CHECK(B >= 0 && Cin > 0 && T > 0 && p > 0 && Hin >= p && Win >= p, -2, "stg_im2col_patch: bad shape");
    STG_CHECK(Kpad % 8 == 0 && Kpad >= Cin * p * p, -2, "stg_im2col_patch: Kpad must be a multiple of 8 and >= Cin*p*p");
    const int64_t nrows = B * T * (Hin / p) * (Win / p);
    if (nrows == 0) return 0;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(nrows * (Kpad / 8), 256)), dim3(256), 0, ST, x, (int)(x_dtype == STG_F32),
                       (bf16_t*)out, B, Cin, T, Hin, Win, p, Kpad, nrows);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_cast_bf16(const float* in, void* out, int64_t R, int64_t Cc, int transpose, int64_t ld_out, void* stream) {
    STG_CHECK(in && out, -1, "stg_cast_bf16: null pointer");
    if (R <= 0 || Cc <= 0) return 0;
    if (!transpose) {
        STG_CHECK(ld_out >= Cc, -2, "stg_cast_bf16: ld_out < columns");
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(R * ld_out, 256)), dim3(256), 0, ST, in, (bf16_t*)out, R, Cc, ld_out);
    } else {
        STG_CHECK(ld_out >= R, -2, "stg_cast_bf16: ld_out < rows (transpose)");
        const int64_t gx = (Cc + 31) / 32, gy = (ld_out + 31) / 32;
        STG_CHECK(gy < 65536, -2, "stg_cast_bf16: too many rows for transpose");
        hipLaunchKernelGGL(cast_bf16_t_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, ST, in, (bf16_t*)out, R, Cc, ld_out);
    }
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_cast_bf16_multi(const stg_cast_desc* descs, int n, int max_elems, void* arena, void* stream) {
    STG_CHECK(descs && arena, -1, "stg_cast_bf16_multi: null pointer");
    STG_CHECK(n >= 0 && n < 65536 && max_elems >= 0, -2, "stg_cast_bf16_multi: bad count");
    STG_CHECK((((uintptr_t)arena) & 15) == 0, -2, "stg_cast_bf16_multi: arena must be 16-byte aligned");
    if (n == 0 || max_elems == 0) return 0;
    int gx = (max_elems + 1023) / 1024;          // <= 4 elements per thread for the largest tensor
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3(gx, n), dim3(256), 0, ST, descs, (bf16_t*)arena);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_add_temporal(float* x, const float* emb, int64_t B, int T, int64_t N, int C, void* stream) {
    STG_CHECK(x && emb, -1, "stg_add_temporal: null pointer");
    STG_CHECK(B >= 0 && T > 0 && N > 0 && C > 0 && C % 4 == 0, -2, "stg_add_temporal: bad shape (C % 4 == 0)");
    STG_CHECK((((uintptr_t)x | (uintptr_t)emb) & 15) == 0, -2, "stg_add_temporal: pointers must be 16-byte aligned");
    const int64_t total4 = B * T * N * (C / 4);
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(add_temporal_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, ST, x, emb, total4, T, N, C / 4);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_adam_multi(const stg_adam_desc* descs, int n, int64_t max_elems, const double* hyper, int n_groups, void* stream) {
    STG_CHECK(descs && hyper, -1, "stg_adam_multi: null pointer");
    STG_CHECK(n >= 0 && n < 65536 && max_elems >= 0 && n_groups > 0, -2, "stg_adam_multi: bad count");
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_bump_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, descs, n, hyper);
    STG_LAUNCH_CHECK();
    if (max_elems == 0) return 0;
    int64_t gx = (max_elems + 1023) / 1024;      // <= 4 elements per thread for the largest tensor
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)gx, n), dim3(256), 0, ST, descs, hyper);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_cast_f32(const void* in, float* out, int64_t numel, void* stream) {
    STG_CHECK(in && out, -1, "stg_cast_f32: null pointer");
    if (numel <= 0) return 0;
    hipLaunchKernelGGL(cast_f32_kernel, dim3(grid_for(numel, 256)), dim3(256), 0, ST, (const bf16_t*)in, out, numel);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_meanpool_fwd(const void* in, void* out, int out_dtype, int64_t ldo, int64_t G, int n, int C, void* stream) {
    STG_CHECK(in && out, -1, "stg_meanpool_fwd: null pointer");
    STG_CHECK(C % 8 == 0 && n > 0 && ldo % 8 == 0, -2, "stg_meanpool_fwd: C and ldo must be multiples of 8");
    if (G <= 0) return 0;
    if (n >= 256 && C <= 2048 && G < (1ll << 31))
        hipLaunchKernelGGL(meanpool_fwd_blk_kernel, dim3((unsigned)G), dim3(256), 0, ST, (const bf16_t*)in, out, (int)(out_dtype == STG_F32), ldo, n, C);
    else
        hipLaunchKernelGGL(meanpool_fwd_kernel, dim3(grid_for(G * (C / 8), 256)), dim3(256), 0, ST, (const bf16_t*)in, out,
                           (int)(out_dtype == STG_F32), ldo, G, n, C);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_meanpool_bwd(const void* dout, int64_t lddo, void* din, int64_t G, int n, int C, void* stream) {
    STG_CHECK(dout && din, -1, "stg_meanpool_bwd: null pointer");
    STG_CHECK(C % 8 == 0 && n > 0 && lddo % 8 == 0, -2, "stg_meanpool_bwd: C and lddo must be multiples of 8");
    if (G <= 0) return 0;
    hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(grid_for(G * n * (C / 8), 256)), dim3(256), 0, ST, (const bf16_t*)dout, lddo,
                       (bf16_t*)din, G, n, C);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bias_gather(const float* table, const int64_t* index, float* out, int L, int H, int nn, void* stream) {
    STG_CHECK(table && index && out, -1, "stg_bias_gather: null pointer");
    STG_CHECK(L > 0 && H > 0 && nn > 0, -2, "stg_bias_gather: bad shape");
    hipLaunchKernelGGL(bias_gather_kernel, dim3(grid_for((int64_t)H * nn, 256)), dim3(256), 0, ST, table, index, out, L, H, nn);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bias_scatter(const float* dbias, const int64_t* index, float* dtable, int L, int H, int nn, void* stream) {
    STG_CHECK(dbias && index && dtable, -1, "stg_bias_scatter: null pointer");
    STG_CHECK(L > 0 && H > 0 && nn > 0, -2, "stg_bias_scatter: bad shape");
    hipLaunchKernelGGL(bias_scatter_kernel, dim3(grid_for((int64_t)H * nn, 256)), dim3(256), 0, ST, dbias, index, dtable, L, H, nn);
    STG_LAUNCH_CHECK();
    return 0;
}
