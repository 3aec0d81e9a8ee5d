// Adapter up-projection + residual join + the NEXT LayerNorm in one row-complete pass (see include/stgcma.h: stg_up_ln_fwd).
//
//   x[m, :] = res32[m, :] (+ res16[m, :]) + rs[m] * (h[m, :K] . W[:, :K]^T + b)          fp32, written out (the residual stream)
//   y[m, :] = LayerNorm(x[m, :]) * gamma + beta                                            bf16 (+ mean, rstd)
//
// In the block body (Swin_AVE.py:716, 780-787, 810-811) every residual join is an adapter's D_fc2 and is followed by a
// LayerNorm (norm1 of the spatial pass, norm2, norm1 of the next block): as two kernels the fp32 row is written by the GEMM
// epilogue and read straight back by the LayerNorm (2 of its 3 U of traffic).  Here a wave owns 16 COMPLETE rows: the fp32
// residual is loaded directly into the MFMA accumulators (C operand), K <= 64 means one or two v_mfma_f32_16x16x32_bf16 per
// 16 columns, the row statistics are a register reduction + two cross-lane adds, and x, y leave from the same registers.
// HBM-bound by construction: per row 4C (res32) + 2C (res16) + 4C (x) + 2C (y) bytes against 2*C*K FLOP.
//
// Register layout.  The MFMA is issued "swapped" (first operand = W fragment, second = h fragment): lane (m = l & 15,
// g = l >> 4) then holds D[slot 4g + r][row m], r = 0..3.  Which 16 output columns a tile's 16 slots stand for is free (it is
// only the order in which W rows are put into the fragment), so tiles are paired: slot 4g + r of tile 2p + j <-> column
// 32p + 8g + 4j + r.  Lane (m, g) therefore owns the 8 CONSECUTIVE columns 32p + 8g .. + 7 of every pair p: 32-byte fp32 /
// 16-byte bf16 pieces per lane, 128 / 64 contiguous bytes per row and instruction -- and, packed to bf16, exactly the
// A-operand layout of a k-block of 32 (used by the backward kernel below).
#include "common.h"
#include "../../include/stgcma.h"

namespace {

struct UpLnP {
    const bf16_t* h; int64_t ldh;
    const bf16_t* w; int64_t ldw;
    const float* bias;
    const float* res32; int64_t ld32;
    const bf16_t* res16; int64_t ld16;
    const float* row_scale; int64_t rs_outer, rs_inner;
    float* x; int64_t ldx;
    const float* gamma; const float* beta; float eps;
    bf16_t* y; int64_t ldy;
    float* mean; float* rstd;
    int64_t M; int C; int K;
    // second row group (pair launches, round 4): rows >= split_m take h2 (indexed from its own row 0) / w2 / bias2 / row_scale2; workgroups
    // [0, nb1) walk group 1, the rest group 2 (a workgroup stages ONE adapter weight).  nb1 == 0: one group.
    const bf16_t* h2; const bf16_t* w2; const float* bias2; const float* row_scale2; int64_t split_m; int nb1;
};

__device__ __forceinline__ void unpack8(const uint4& q, float* v) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(w[j] << 16);
        v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float* t) {
    return make_uint4(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]), pack_bf2(t[4], t[5]), pack_bf2(t[6], t[7]));
}

// W [C, K] -> LDS in fragment order: entry (t * KS + ks) * 64 + l is lane l's 16 bytes of tile t, k-step ks
template <int NT, int KS>
__device__ __forceinline__ void fill_wfrag(uint4* wfrag, const bf16_t* w, int64_t ldw, int K, int tid, int nthreads) {
    for (int f = tid; f < NT * KS * 64; f += nthreads) {
        const int l = f & 63, tk = f >> 6;
        const int t = tk / KS, ks = tk - t * KS;
        const int slot = l & 15, kq = l >> 4;
        const int col = 32 * (t >> 1) + 8 * (slot >> 2) + 4 * (t & 1) + (slot & 3);
        const int k0 = ks * 32 + kq * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (k0 < K) v = *reinterpret_cast<const uint4*>(w + (int64_t)col * ldw + k0);
        wfrag[f] = v;
    }
}

// NT = 16-column tiles per wave, NH = waves per row (column slices of 16 NT; statistics combined through LDS), C = 16 * NT * NH;
// NW = waves per block (8 / 16 where the W fragments of one block take most of the LDS: at C = 768, K = 96 one 147 KiB copy serves 16 waves).
// Round 5: NH = 4 and K <= 96 (KS = 3) -- the Swin-L widths 192 / 384 / 768 all run as NT = 12 with NH = 1 / 2 / 4.
template <int NT, int NH, int KS, bool R16, int NW = 4>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) upln_fwd_kernel(UpLnP p) {
    extern __shared__ __attribute__((aligned(16))) uint4 wfrag[];
    constexpr int TT = NT * NH, CW = NT * 16, CC = TT * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int half = NH > 1 ? (wave % NH) : 0;
    int bid = blockIdx.x, nbk = gridDim.x;
    if (p.nb1 > 0) {                                                // pair launch: this workgroup's row group (block-uniform)
        if (bid >= p.nb1) {
            const int64_t s0 = p.split_m;
            p.h = p.h2; p.w = p.w2; p.bias = p.bias2; p.row_scale = p.row_scale2;
            p.res32 += s0 * p.ld32; p.x += s0 * p.ldx; p.y += s0 * p.ldy;
            if (R16) p.res16 += s0 * p.ld16;
            if (p.mean) p.mean += s0;
            if (p.rstd) p.rstd += s0;
            p.M -= s0; bid -= p.nb1; nbk -= p.nb1;
        } else {
            p.M = p.split_m; nbk = p.nb1;
        }
    }
    fill_wfrag<TT, KS>(wfrag, p.w, p.ldw, p.K, tid, NW * 64);
    // bias / gamma / beta live in LDS and are re-read per row group through a laundered offset: as loop invariants the
    // compiler hoists all 3 * C / 64 * 4 values per lane out of the row loop (384 VGPRs at C = 512: spills)
    float* prm = reinterpret_cast<float*>(wfrag + TT * KS * 64);
    float2* xch = reinterpret_cast<float2*>(prm + 3 * CC);          // [2 parities][NW waves][16 rows] (NH > 1 only)
    for (int c = tid; c < CC; c += NW * 64) {
        prm[c] = p.bias[c];
        prm[CC + c] = p.gamma[c];
        prm[2 * CC + c] = p.beta[c];
    }
    __syncthreads();

    constexpr int GPB = NW / NH;                                    // row groups per block and trip
    const float invC = 1.0f / (float)CC;
    const int64_t ngroups = (p.M + 15) >> 4;
    int par = 0;
    for (int64_t base = (int64_t)bid * GPB; base < ngroups; base += (int64_t)nbk * GPB, par ^= 1) {
        const int64_t grp = base + wave / NH;                       // may run past the end (NH == 2): clamped rows, no stores
        const int64_t row = grp * 16 + m;
        const bool valid = row < p.M;
        const int64_t rc = valid ? row : p.M - 1;          // clamped: loads stay unconditional, stores are predicated
        int go = 8 * g + half * CW;
        int lo = lane + half * NT * KS * 64;
        asm volatile("" : "+v"(go), "+v"(lo));

        // every load of the row group is issued up front: the fp32 residual lands directly in the accumulators
        bf16x8_t hf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int k0 = ks * 32 + g * 8;
            const bool kok = k0 < p.K;
            k0 = kok ? k0 : 0;
            uint4 q = *reinterpret_cast<const uint4*>(p.h + rc * p.ldh + k0);
            const uint32_t msk = kok ? 0xffffffffu : 0u;
            q.x &= msk; q.y &= msk; q.z &= msk; q.w &= msk;
            hf[ks] = __builtin_bit_cast(bf16x8_t, q);
        }
        f32x4_t acc[NT];
        const float* r32 = p.res32 + rc * p.ld32 + 8 * g + half * CW;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const f32x4_t*>(r32 + 32 * (t >> 1) + 4 * (t & 1));
        uint4 r16[R16 ? NT / 2 : 1];
        if (R16) {
            const bf16_t* q16 = p.res16 + rc * p.ld16 + 8 * g + half * CW;
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) r16[pr] = *reinterpret_cast<const uint4*>(q16 + 32 * pr);
        }
        float rs = 1.0f;
        if (p.row_scale) {                                  // DropPath: scales the branch (h W^T + b), not the residual
            rs = p.row_scale[((uint32_t)rc / (uint32_t)p.rs_outer) * (uint32_t)p.rs_inner + ((uint32_t)rc % (uint32_t)p.rs_inner)];     // 32-bit: the host checks M < 2^31 (a 64-bit division is ~120 instructions per row group)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float v[8];
                unpack8(__builtin_bit_cast(uint4, hf[ks]), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= rs;
                hf[ks] = __builtin_bit_cast(bf16x8_t, pack8(v));
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // keeps the LDS fragment reads from piling up in VGPRs
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag[(t * KS + ks) * 64 + lo]),
                                                                 hf[ks], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        float s = 0.f;
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            if ((pr & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const float4 b0 = *reinterpret_cast<const float4*>(prm + 32 * pr + go);
            const float4 b1 = *reinterpret_cast<const float4*>(prm + 32 * pr + go + 4);
            const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float add[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) add[j] = rs * b[j];
            if (R16) {
                float v[8];
                unpack8(r16[pr], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) add[j] += v[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * pr][j] += add[j];
                acc[2 * pr + 1][j] += add[4 + j];
                s += acc[2 * pr][j] + acc[2 * pr + 1][j];
            }
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        float mu = s * (1.0f / (float)CW);                  // two-pass statistics of this wave's CW columns
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = acc[t][j] - mu; q += d * d; }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (NH == 2) {                                      // combine the two halves (Chan et al.: exact, no cancellation)
            float2* xb = xch + par * (NW * 16);
            if (g == 0) xb[wave * 16 + m] = make_float2(s, q);
            __syncthreads();
            const float2 o = xb[(wave ^ 1) * 16 + m];
            const float dlt = (s - o.x) * (1.0f / (float)CW);
            q = q + o.y + dlt * dlt * (0.5f * (float)CW);
            mu = (s + o.x) * invC;
        } else if (NH > 2) {                                // NH slices: M2 = sum_i (q_i + CW (mu_i - mu)^2), every wave in the same order
            float2* xb = xch + par * (NW * 16);
            if (g == 0) xb[wave * 16 + m] = make_float2(s, q);
            __syncthreads();
            const int w0 = wave - half;
            float2 o[NH];
            float st = 0.f;
#pragma unroll
            for (int i = 0; i < NH; ++i) { o[i] = xb[(w0 + i) * 16 + m]; st += o[i].x; }
            mu = st * invC;
            q = 0.f;
#pragma unroll
            for (int i = 0; i < NH; ++i) { const float dm = o[i].x * (1.0f / (float)CW) - mu; q += o[i].y + dm * dm * (float)CW; }
        }
        const float rstd = rsqrtf(q * invC + p.eps);
        if (valid) {
            float* xo = p.x + row * p.ldx + 8 * g + half * CW;
#pragma unroll
            for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4_t*>(xo + 32 * (t >> 1) + 4 * (t & 1)) = acc[t];
            bf16_t* yo = p.y + row * p.ldy + 8 * g + half * CW;
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {
                if ((pr & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                const float4 g0 = *reinterpret_cast<const float4*>(prm + CC + 32 * pr + go);
                const float4 g1 = *reinterpret_cast<const float4*>(prm + CC + 32 * pr + go + 4);
                const float4 e0 = *reinterpret_cast<const float4*>(prm + 2 * CC + 32 * pr + go);
                const float4 e1 = *reinterpret_cast<const float4*>(prm + 2 * CC + 32 * pr + go + 4);
                const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float be[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                float o[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = (acc[2 * pr][j] - mu) * rstd * ga[j] + be[j];
                    o[4 + j] = (acc[2 * pr + 1][j] - mu) * rstd * ga[4 + j] + be[4 + j];
                }
                *reinterpret_cast<uint4*>(yo + 32 * pr) = pack8(o);
            }
            if (g == 0 && half == 0) {
                if (p.mean) p.mean[row] = mu;
                if (p.rstd) p.rstd[row] = rstd;
            }
        }
    }
}

// grid of a (pair) launch: one group -> up to 2048 workgroups; two groups -> up to 1024 each, group 1 first (sets nb1)
template <typename P>
int64_t plan_grid(P& p, int GPB, int64_t cap = 2048) {
    // cap = workgroups of the launch.  Every workgroup first fills LDS with the adapter's weight matrix in fragment order; the wide variants (NW >= 8:
    // C = 384 / 768 with K / J up to 96, 73-147 KiB of weights, ONE workgroup per CU) ran as 2 048 workgroups of ~61 rows until round 5b -- eight
    // rounds per CU, each paying the 147 KiB prologue (40 % of the bytes it then streams, latency exposed) -- and are now ONE round of 256.
    auto blocks = [&](int64_t rows, int64_t c) { const int64_t n = ((rows + 15) / 16 + GPB - 1) / GPB; return n > c ? c : n; };
    if (p.split_m <= 0) { p.nb1 = 0; return blocks(p.M, cap); }
    // pair launch: the two row groups share the cap in proportion to their rows (ViT-B: 63 040 video rows, 15 680 audio rows -- equal halves left
    // the audio half's workgroups done four times earlier)
    int64_t c1 = (cap * p.split_m + p.M / 2) / p.M;
    c1 = c1 < 1 ? 1 : c1 > cap - 1 ? cap - 1 : c1;
    p.nb1 = (int)blocks(p.split_m, c1);
    return p.nb1 + blocks(p.M - p.split_m, cap - c1);
}

template <int NT, int NH, int KS, int NW = 4>
int launch_upln(const UpLnP& pin, hipStream_t st) {
    constexpr int GPB = NW / NH;
    UpLnP p = pin;
    const int64_t nblk = plan_grid(p, GPB, NW >= 8 ? stg_opt_upln_cap.load(std::memory_order_relaxed) : 2048);
    const size_t lds = (size_t)NT * NH * KS * 64 * 16 + (size_t)3 * NT * NH * 16 * 4 + 2 * NW * 16 * sizeof(float2);
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> d1{0}, d0{0};
        const bool ok = stg_reserve_lds(upln_fwd_kernel<NT, NH, KS, true, NW>, (int)lds, d1) && stg_reserve_lds(upln_fwd_kernel<NT, NH, KS, false, NW>, (int)lds, d0);
        STG_CHECK(ok, -101, "stg_up_ln_fwd: cannot reserve %d bytes of LDS", (int)lds);
    }
    if (p.res16) hipLaunchKernelGGL((upln_fwd_kernel<NT, NH, KS, true, NW>), dim3((unsigned)nblk), dim3(NW * 64), lds, st, p);
    else hipLaunchKernelGGL((upln_fwd_kernel<NT, NH, KS, false, NW>), dim3((unsigned)nblk), dim3(NW * 64), lds, st, p);
    STG_LAUNCH_CHECK();
    return 0;
}

template <int NT, int NH, int NW = 4>
int dispatch_ks(const UpLnP& p, hipStream_t st) {
    return p.K <= 32 ? launch_upln<NT, NH, 1, NW>(p, st) : p.K <= 64 ? launch_upln<NT, NH, 2, NW>(p, st) : launch_upln<NT, NH, 3, NW>(p, st);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward + the adapter dgrad that consumes its result (see include/stgcma.h: stg_ln_bwd_down):
//   dx[m, :]  = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)) (+ add_to[m, :])        bf16, written out
//   dh[m, :J] = rs[m] * (dx[m, :] . Wt[:J, :]^T)                                          bf16  (Wt = D_fc2.weight^T, [J, C])
// Same row-complete layout as the forward kernel: the freshly packed bf16 dx pieces ARE the MFMA operand of k-block p, so the
// down-projection costs the row no second trip through HBM (as a separate GEMM it re-read all of dx to produce J <= 64 columns).
struct LnDownP {
    const bf16_t* dy; int64_t lddy;
    const float* x; int64_t ldx;                 // XH: bf16 x_hat rows (ldx in bf16 elements), gamma == 1, no mean
    const float* gamma; const float* mean; const float* rstd;
    const bf16_t* add_to; int64_t ldadd;
    bf16_t* dx; int64_t lddx;
    const bf16_t* wt; int64_t ldwt;
    const float* row_scale; int64_t rs_outer, rs_inner;
    bf16_t* dh; int64_t lddh;
    int64_t M; int C; int J;
    const bf16_t* wt2; const float* row_scale2; bf16_t* dh2; int64_t split_m; int nb1;      // second row group, as in UpLnP (dh2 indexed from its own row 0)
};

template <int NT, int NH, int NJ, bool ADD, int NW = 4, bool XH = false>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) ln_bwd_down_kernel(LnDownP p) {
    extern __shared__ __attribute__((aligned(16))) uint4 wfrag[];
    constexpr int TT = NT * NH, CW = NT * 16, CC = TT * 16, NP = NT / 2, NPT = TT / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int half = NH > 1 ? (wave % NH) : 0;
    int bid = blockIdx.x, nbk = gridDim.x;
    if (p.nb1 > 0) {                                                // pair launch: this workgroup's row group (block-uniform)
        if (bid >= p.nb1) {
            const int64_t s0 = p.split_m;
            p.wt = p.wt2; p.row_scale = p.row_scale2; p.dh = p.dh2;
            p.dy += s0 * p.lddy; p.dx += s0 * p.lddx; p.rstd += s0;
            p.x = XH ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(p.x) + s0 * p.ldx) : p.x + s0 * p.ldx;
            if (!XH) p.mean += s0;
            if (ADD) p.add_to += s0 * p.ldadd;
            p.M -= s0; bid -= p.nb1; nbk -= p.nb1;
        } else {
            p.M = p.split_m; nbk = p.nb1;
        }
    }
    // Wt [J, C] -> LDS in fragment order: entry (jt * NPT + pg) * 64 + l = lane l's 16 bytes of output tile jt, k-block pg.
    // Output slots are paired like the forward kernel's columns: slot 4g + r of tile 2q + i <-> j = 32q + 8g + 4i + r.
    for (int f = tid; f < NJ * NPT * 64; f += NW * 64) {
        const int l = f & 63, tk = f >> 6;
        const int jt = tk / NPT, pg = tk - jt * NPT;
        const int slot = l & 15, kq = l >> 4;
        const int j = (NJ == 1 || ((NJ & 1) && jt == NJ - 1)) ? 16 * jt + slot                     // an unpaired (last) tile
                                                                : 32 * (jt >> 1) + 8 * (slot >> 2) + 4 * (jt & 1) + (slot & 3);
        wfrag[f] = *reinterpret_cast<const uint4*>(p.wt + (int64_t)j * p.ldwt + 32 * pg + 8 * kq);
    }
    float* gam = reinterpret_cast<float*>(wfrag + NJ * NPT * 64);    // not allocated in the x-hat form (gamma == 1)
    float2* xch = reinterpret_cast<float2*>(gam + (XH ? 0 : CC));    // [NW waves][16 rows]
    float* hx = reinterpret_cast<float*>(xch + NW * 16);             // [NW / NH groups][NJ * 4][64 lanes]  (NH > 1)
    if (!XH) for (int c = tid; c < CC; c += NW * 64) gam[c] = p.gamma[c];
    __syncthreads();

    constexpr int GPB = NW / NH;
    const float invC = 1.0f / (float)CC;
    const int64_t ngroups = (p.M + 15) >> 4;
    for (int64_t base = (int64_t)bid * GPB; base < ngroups; base += (int64_t)nbk * GPB) {
        const int64_t grp = base + wave / NH;
        const int64_t row = grp * 16 + m;
        const bool valid = row < p.M;
        const int64_t rc = valid ? row : p.M - 1;
        int go = 8 * g + half * CW;
        int lo = lane + half * NP * 64;
        asm volatile("" : "+v"(go), "+v"(lo));

        uint4 dyq[NP];
        f32x4_t xh[NT];
        uint4 adq[ADD ? NP : 1];
        const bf16_t* dyp = p.dy + rc * p.lddy + 8 * g + half * CW;
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) dyq[pr] = *reinterpret_cast<const uint4*>(dyp + 32 * pr);
        if (XH) {                                           // the lane's 8 columns of x_hat are ONE 16-byte bf16 piece (the layout of dy)
            const bf16_t* xq = reinterpret_cast<const bf16_t*>(p.x) + rc * p.ldx + 8 * g + half * CW;
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                float xv[8];
                unpack8(*reinterpret_cast<const uint4*>(xq + 32 * pr), xv);
#pragma unroll
                for (int j = 0; j < 4; ++j) { xh[2 * pr][j] = xv[j]; xh[2 * pr + 1][j] = xv[4 + j]; }
            }
        } else {
            const float* xp = p.x + rc * p.ldx + 8 * g + half * CW;
#pragma unroll
            for (int t = 0; t < NT; ++t) xh[t] = *reinterpret_cast<const f32x4_t*>(xp + 32 * (t >> 1) + 4 * (t & 1));
        }
        if (ADD) {
            const bf16_t* ap = p.add_to + rc * p.ldadd + 8 * g + half * CW;
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) adq[pr] = *reinterpret_cast<const uint4*>(ap + 32 * pr);
        }
        const float mu = XH ? 0.f : p.mean[rc], rs = p.rstd[rc];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            if ((pr & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            float ga[8];
            if (XH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ga[j] = 1.0f;
            } else {
                const float4 g0 = *reinterpret_cast<const float4*>(gam + 32 * pr + go);
                const float4 g1 = *reinterpret_cast<const float4*>(gam + 32 * pr + go + 4);
                ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
            }
            float dv[8];
            unpack8(dyq[pr], dv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = XH ? xh[2 * pr][j] : (xh[2 * pr][j] - mu) * rs, a1 = XH ? xh[2 * pr + 1][j] : (xh[2 * pr + 1][j] - mu) * rs;
                xh[2 * pr][j] = a0; xh[2 * pr + 1][j] = a1;
                const float e0 = ga[j] * dv[j], e1 = ga[4 + j] * dv[4 + j];
                s1 += e0 + e1;
                s2 += e0 * a0 + e1 * a1;
            }
        }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (NH == 2) {
            if (g == 0) xch[wave * 16 + m] = make_float2(s1, s2);
            __syncthreads();
            const float2 o = xch[(wave ^ 1) * 16 + m];
            s1 += o.x; s2 += o.y;
        } else if (NH > 2) {                                // every wave of the row group sums the NH slices in the same order
            if (g == 0) xch[wave * 16 + m] = make_float2(s1, s2);
            __syncthreads();
            const int w0 = wave - half;
            s1 = 0.f; s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NH; ++i) { const float2 o = xch[(w0 + i) * 16 + m]; s1 += o.x; s2 += o.y; }
        }
        s1 *= invC; s2 *= invC;
        f32x4_t acc[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bf16_t* dxp = p.dx + row * p.lddx + 8 * g + half * CW;
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            if ((pr & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            float ga[8];
            if (XH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ga[j] = 1.0f;
            } else {
                const float4 g0 = *reinterpret_cast<const float4*>(gam + 32 * pr + go);
                const float4 g1 = *reinterpret_cast<const float4*>(gam + 32 * pr + go + 4);
                ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
            }
            float dv[8], o[8];
            unpack8(dyq[pr], dv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = rs * (ga[j] * dv[j] - s1 - xh[2 * pr][j] * s2);
                o[4 + j] = rs * (ga[4 + j] * dv[4 + j] - s1 - xh[2 * pr + 1][j] * s2);
            }
            if (ADD) {
                float av[8];
                unpack8(adq[pr], av);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += av[j];
            }
            const uint4 dq = pack8(o);
            if (valid) *reinterpret_cast<uint4*>(dxp + 32 * pr) = dq;
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, dq);
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag[(jt * NPT + pr) * 64 + lo]), af,
                                                                 acc[jt], 0, 0, 0);
        }
        if (NH == 2) {                                      // the other half's partial products
            float* hb = hx + (wave >> 1) * (NJ * 4 * 64);
            if (half == 1) {
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hb[(jt * 4 + r) * 64 + lane] = acc[jt][r];
            }
            __syncthreads();
            if (half == 0) {
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[jt][r] += hb[(jt * 4 + r) * 64 + lane];
            }
        } else if (NH > 2) {                                // a chain through ONE buffer per row group (the W^T fragments leave no room for NH - 1):
            float* hb = hx + (wave / NH) * (NJ * 4 * 64);   // slice NH - 1 writes, each lower slice adds its own and passes it on: ((s3 + s2) + s1) + s0
#pragma unroll
            for (int hs = NH - 1; hs >= 0; --hs) {
                if (half == hs) {
                    if (hs < NH - 1) {
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[jt][r] += hb[(jt * 4 + r) * 64 + lane];
                    }
                    if (hs > 0) {
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) hb[(jt * 4 + r) * 64 + lane] = acc[jt][r];
                    }
                }
                if (hs > 0) __syncthreads();
            }
        }
        if (valid && half == 0) {
            float sc = 1.0f;
            if (p.row_scale) sc = p.row_scale[((uint32_t)row / (uint32_t)p.rs_outer) * (uint32_t)p.rs_inner + ((uint32_t)row % (uint32_t)p.rs_inner)];   // 32-bit: the host checks M < 2^31
            bf16_t* hp = p.dh + row * p.lddh;
#pragma unroll
            for (int q = 0; q < NJ / 2; ++q) {
                float o[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[r] = acc[2 * q][r] * sc; o[4 + r] = acc[2 * q + 1][r] * sc; }
                *reinterpret_cast<uint4*>(hp + 32 * q + 8 * g) = pack8(o);
            }
            if (NJ & 1) {                                   // the unpaired last tile: columns 16 (NJ - 1) + 4 g .. + 3
                uint2 v;
                v.x = pack_bf2(acc[NJ - 1][0] * sc, acc[NJ - 1][1] * sc); v.y = pack_bf2(acc[NJ - 1][2] * sc, acc[NJ - 1][3] * sc);
                *reinterpret_cast<uint2*>(hp + 16 * (NJ - 1) + 4 * g) = v;
            }
        }
    }
}

template <int NT, int NH, int NJ, int NW = 4, bool XH = false>
int launch_lnbd(const LnDownP& pin, hipStream_t st) {
    constexpr int GPB = NW / NH;
    LnDownP p = pin;
    const int64_t nblk = plan_grid(p, GPB, NW >= 8 ? stg_opt_upln_cap.load(std::memory_order_relaxed) : 2048);
    const size_t lds = (size_t)NJ * (NT * NH / 2) * 64 * 16 + (XH ? 0 : (size_t)NT * NH * 16 * 4) + NW * 16 * sizeof(float2) +
                       (NH > 1 ? (size_t)(NW / NH) * NJ * 4 * 64 * 4 : 0);
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> d1{0}, d0{0};
        const bool ok = stg_reserve_lds(ln_bwd_down_kernel<NT, NH, NJ, true, NW, XH>, (int)lds, d1) && stg_reserve_lds(ln_bwd_down_kernel<NT, NH, NJ, false, NW, XH>, (int)lds, d0);
        STG_CHECK(ok, -101, "stg_ln_bwd_down: cannot reserve %d bytes of LDS", (int)lds);
    }
    if (p.add_to) hipLaunchKernelGGL((ln_bwd_down_kernel<NT, NH, NJ, true, NW, XH>), dim3((unsigned)nblk), dim3(NW * 64), lds, st, p);
    else hipLaunchKernelGGL((ln_bwd_down_kernel<NT, NH, NJ, false, NW, XH>), dim3((unsigned)nblk), dim3(NW * 64), lds, st, p);
    STG_LAUNCH_CHECK();
    return 0;
}

template <int NT, int NH, bool XH = false>
int dispatch_nj(const LnDownP& p, hipStream_t st) {
    if (p.J == 16) return launch_lnbd<NT, NH, 1, 4, XH>(p, st);
    if (p.J == 32) return launch_lnbd<NT, NH, 2, 4, XH>(p, st);
    return launch_lnbd<NT, NH, 4, 4, XH>(p, st);
}
// round 5: the NT = 12 family (C = 192 NH, NH = 1 / 2 / 4 -> 4 / 8 / 16 waves per block): Swin-L's J = 96, CLIP ViT-B's J = 48
template <int NH, bool XH = false>
int dispatch_nj12(const LnDownP& p, hipStream_t st) {
    // 8 waves per block where a row spans several waves: the kernel needs 150 - 200 VGPRs (16 waves = 128 spill 250+ bytes per lane), and at
    // C = 768, J = 96 the W^T fragments ([J, C] bf16 = 147 KiB) leave room for the partial-product buffers of two row groups only
    constexpr int NW = NH == 1 ? 4 : 8;
    if (p.J == 48) return launch_lnbd<12, NH, 3, NW, XH>(p, st);
    return launch_lnbd<12, NH, 6, NW, XH>(p, st);
}

}  // namespace

extern "C" int stg_up_ln_supported(int C, int K) {
    if (C == 192 || C == 384 || C == 768) return K >= 8 && K <= 96 && K % 8 == 0;        // NT = 12 family: Swin-L (d_h = 96), CLIP ViT-B (d_h = 48)
    return (C == 128 || C == 256 || C == 512) && K >= 8 && K <= 64 && K % 8 == 0;
}

static int up_ln_impl(const char* who, const void* h, int64_t ldh, const void* w, int64_t ldw, const float* bias, const float* res32,
                      int64_t ld32, const void* res16, int64_t ld16, const float* row_scale, int64_t rs_outer, int64_t rs_inner, float* x,
                      int64_t ldx, const float* gamma, const float* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd,
                      int64_t M, int C, int K, const void* h2, const void* w2, const float* bias2, const float* row_scale2, int64_t split_m,
                      void* stream) {
    STG_CHECK(h && w && bias && res32 && x && gamma && beta && y, -1, "%s: null pointer", who);
    STG_CHECK(stg_up_ln_supported(C, K), -2, "%s: unsupported C=%d K=%d (C in {128,256,512}: K <= 64; C in {192,384,768}: K <= 96; K %% 8 == 0)", who, C, K);
    STG_CHECK(M >= 0 && M < (1ll << 31), -2, "%s: bad M (row-scale indices are 32-bit)", who);
    STG_CHECK(ldh % 8 == 0 && ldh >= K && ldw % 8 == 0 && ldw >= K, -2, "%s: ldh / ldw must be multiples of 8 and >= K", who);
    STG_CHECK(ld32 % 4 == 0 && ld32 >= C && ldx % 4 == 0 && ldx >= C && ldy % 8 == 0 && ldy >= C, -2, "%s: bad ld32 / ldx / ldy", who);
    STG_CHECK(res16 == nullptr || (ld16 % 8 == 0 && ld16 >= C), -2, "%s: bad ld16", who);
    STG_CHECK((row_scale == nullptr && row_scale2 == nullptr) || (rs_outer > 0 && rs_inner > 0), -2, "%s: bad row_scale geometry", who);
    STG_CHECK(((uintptr_t)h | (uintptr_t)w | (uintptr_t)res32 | (uintptr_t)res16 | (uintptr_t)x | (uintptr_t)y | (uintptr_t)bias |
               (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0, -2, "%s: operands must be 16-byte aligned", who);
    if (split_m > 0) {
        STG_CHECK(h2 && w2 && bias2, -1, "%s: null pointer in the second row group", who);
        STG_CHECK(split_m < M && split_m % 16 == 0, -2, "%s: split_m must be a multiple of 16 inside (0, M)", who);
        STG_CHECK(((uintptr_t)h2 | (uintptr_t)w2 | (uintptr_t)bias2) % 16 == 0, -2, "%s: operands must be 16-byte aligned", who);
        STG_CHECK((row_scale == nullptr) == (row_scale2 == nullptr), -2, "%s: both row groups or neither carry a row_scale", who);
    }
    if (M == 0) return 0;
    UpLnP p = {};
    p.h = (const bf16_t*)h; p.ldh = ldh; p.w = (const bf16_t*)w; p.ldw = ldw; p.bias = bias;
    p.res32 = res32; p.ld32 = ld32; p.res16 = (const bf16_t*)res16; p.ld16 = ld16;
    p.row_scale = row_scale; p.rs_outer = rs_outer; p.rs_inner = rs_inner;
    p.x = x; p.ldx = ldx; p.gamma = gamma; p.beta = beta; p.eps = eps; p.y = (bf16_t*)y; p.ldy = ldy;
    p.mean = mean; p.rstd = rstd; p.M = M; p.C = C; p.K = K;
    p.h2 = (const bf16_t*)h2; p.w2 = (const bf16_t*)w2; p.bias2 = bias2; p.row_scale2 = row_scale2; p.split_m = split_m > 0 ? split_m : 0;
    hipStream_t st = (hipStream_t)stream;
    switch (C / 16) {
        case 8: return dispatch_ks<8, 1>(p, st);
        case 16: return dispatch_ks<16, 1>(p, st);
        case 32: return dispatch_ks<16, 2>(p, st);
        case 12: return dispatch_ks<12, 1, 4>(p, st);                                                        // C = 192 (Swin-L stage 0)
        case 24: return dispatch_ks<12, 2, 16>(p, st);                                                       // C = 384 (Swin-L stage 1)
        default: return dispatch_ks<12, 4, 16>(p, st);                                                       // C = 768 (Swin-L stage 2, CLIP ViT-B)
    }
}

extern "C" int stg_up_ln_fwd(const void* h, int64_t ldh, const void* w, int64_t ldw, const float* bias, const float* res32,
                             int64_t ld32, const void* res16, int64_t ld16, const float* row_scale, int64_t rs_outer,
                             int64_t rs_inner, float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* y,
                             int64_t ldy, float* mean, float* rstd, int64_t M, int C, int K, void* stream) {
    return up_ln_impl("stg_up_ln_fwd", h, ldh, w, ldw, bias, res32, ld32, res16, ld16, row_scale, rs_outer, rs_inner, x, ldx, gamma, beta, eps, y, ldy,
                      mean, rstd, M, C, K, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int stg_up_ln_fwd_pair(const void* h, const void* h2, int64_t ldh, const void* w, const void* w2, int64_t ldw, const float* bias,
                                  const float* bias2, int64_t split_m, const float* res32, int64_t ld32, const void* res16, int64_t ld16,
                                  const float* row_scale, const float* row_scale2, int64_t rs_outer, int64_t rs_inner, float* x, int64_t ldx,
                                  const float* gamma, const float* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd, int64_t M,
                                  int C, int K, void* stream) {
    STG_CHECK(split_m > 0, -2, "stg_up_ln_fwd_pair: split_m must be positive");
    return up_ln_impl("stg_up_ln_fwd_pair", h, ldh, w, ldw, bias, res32, ld32, res16, ld16, row_scale, rs_outer, rs_inner, x, ldx, gamma, beta, eps, y,
                      ldy, mean, rstd, M, C, K, h2, w2, bias2, row_scale2, split_m, stream);
}

extern "C" int stg_ln_bwd_down_supported(int C, int J) {
    // round 3 left C = 768 to stg_layernorm_bwd + stg_gemm_nt (24 tiles per wave: 256 VGPRs with spills, one 8-wave block per CU, 177 us
    // against 125 us for the pair); round 5 runs it as 12 tiles per wave x 4 waves per row, 16 waves per block
    if (C == 192 || C == 384 || C == 768) return J == 48 || J == 96;
    return (C == 128 || C == 256 || C == 512) && (J == 16 || J == 32 || J == 64);
}

static int ln_bwd_down_impl(const char* who, bool xh, const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* mean,
                            const float* rstd, const void* add_to, int64_t ldadd, void* dx, int64_t lddx, const void* wt, int64_t ldwt,
                            const float* row_scale, int64_t rs_outer, int64_t rs_inner, void* dh, int64_t lddh, int64_t M, int C, int J,
                            const void* wt2, const float* row_scale2, void* dh2, int64_t split_m, void* stream) {
    STG_CHECK(dy && x && rstd && dx && wt && dh && (xh || (gamma && mean)), -1, "%s: null pointer", who);
    STG_CHECK(stg_ln_bwd_down_supported(C, J), -2, "%s: unsupported C=%d J=%d (C in {128,256,512}: J in {16,32,64}; C in {192,384,768}: J in {48,96})", who, C, J);
    STG_CHECK(M >= 0 && M < (1ll << 31), -2, "%s: bad M (row-scale indices are 32-bit)", who);
    STG_CHECK(lddy % 8 == 0 && lddy >= C && ldx % (xh ? 8 : 4) == 0 && ldx >= C && lddx % 8 == 0 && lddx >= C, -2, "%s: bad lddy / ldx / lddx", who);
    STG_CHECK(add_to == nullptr || (ldadd % 8 == 0 && ldadd >= C), -2, "%s: bad ldadd", who);
    STG_CHECK(ldwt % 8 == 0 && ldwt >= C && lddh % 4 == 0 && lddh >= J, -2, "%s: bad ldwt / lddh", who);
    STG_CHECK((row_scale == nullptr && row_scale2 == nullptr) || (rs_outer > 0 && rs_inner > 0), -2, "%s: bad row_scale geometry", who);
    STG_CHECK(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)add_to | (uintptr_t)dx | (uintptr_t)wt | (uintptr_t)(xh ? nullptr : gamma)) % 16 == 0 &&
              (uintptr_t)dh % 8 == 0 && (J == 16 || (uintptr_t)dh % 16 == 0), -2, "%s: operands must be 16-byte aligned", who);
    if (split_m > 0) {
        STG_CHECK(wt2 && dh2, -1, "%s: null pointer in the second row group", who);
        STG_CHECK(split_m < M && split_m % 16 == 0, -2, "%s: split_m must be a multiple of 16 inside (0, M)", who);
        STG_CHECK((uintptr_t)wt2 % 16 == 0 && (uintptr_t)dh2 % 8 == 0 && (J == 16 || (uintptr_t)dh2 % 16 == 0), -2, "%s: operands must be 16-byte aligned", who);
        STG_CHECK((row_scale == nullptr) == (row_scale2 == nullptr), -2, "%s: both row groups or neither carry a row_scale", who);
    }
    if (M == 0) return 0;
    LnDownP p = {};
    p.dy = (const bf16_t*)dy; p.lddy = lddy; p.x = (const float*)x; p.ldx = ldx; p.gamma = gamma; p.mean = mean; p.rstd = rstd;
    p.add_to = (const bf16_t*)add_to; p.ldadd = ldadd; p.dx = (bf16_t*)dx; p.lddx = lddx; p.wt = (const bf16_t*)wt; p.ldwt = ldwt;
    p.row_scale = row_scale; p.rs_outer = rs_outer; p.rs_inner = rs_inner; p.dh = (bf16_t*)dh; p.lddh = lddh;
    p.M = M; p.C = C; p.J = J;
    p.wt2 = (const bf16_t*)wt2; p.row_scale2 = row_scale2; p.dh2 = (bf16_t*)dh2; p.split_m = split_m > 0 ? split_m : 0;
    hipStream_t st = (hipStream_t)stream;
    if (xh) {
        switch (C / 16) {
            case 8: return dispatch_nj<8, 1, true>(p, st);
            case 16: return dispatch_nj<16, 1, true>(p, st);
            case 32: return dispatch_nj<16, 2, true>(p, st);
            case 12: return dispatch_nj12<1, true>(p, st);
            case 24: return dispatch_nj12<2, true>(p, st);
            default: return dispatch_nj12<4, true>(p, st);
        }
    }
    switch (C / 16) {
        case 8: return dispatch_nj<8, 1>(p, st);
        case 16: return dispatch_nj<16, 1>(p, st);
        case 32: return dispatch_nj<16, 2>(p, st);
        case 12: return dispatch_nj12<1>(p, st);
        case 24: return dispatch_nj12<2>(p, st);
        default: return dispatch_nj12<4>(p, st);
    }
}

extern "C" int stg_ln_bwd_down(const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                               const float* rstd, const void* add_to, int64_t ldadd, void* dx, int64_t lddx, const void* wt,
                               int64_t ldwt, const float* row_scale, int64_t rs_outer, int64_t rs_inner, void* dh, int64_t lddh,
                               int64_t M, int C, int J, void* stream) {
    return ln_bwd_down_impl("stg_ln_bwd_down", false, dy, lddy, x, ldx, gamma, mean, rstd, add_to, ldadd, dx, lddx, wt, ldwt, row_scale, rs_outer, rs_inner,
                            dh, lddh, M, C, J, nullptr, nullptr, nullptr, 0, stream);
}

// stg_ln_bwd_down from the NORMALISED row (see stg_layernorm_bwd_xhat): xhat [M, C] bf16 in place of the fp32 residual row, gamma == 1.
extern "C" int stg_ln_bwd_down_xhat(const void* dy, int64_t lddy, const void* xhat, int64_t ldx, const float* rstd, const void* add_to,
                                    int64_t ldadd, void* dx, int64_t lddx, const void* wt, int64_t ldwt, const float* row_scale,
                                    int64_t rs_outer, int64_t rs_inner, void* dh, int64_t lddh, int64_t M, int C, int J, void* stream) {
    return ln_bwd_down_impl("stg_ln_bwd_down_xhat", true, dy, lddy, xhat, ldx, nullptr, nullptr, rstd, add_to, ldadd, dx, lddx, wt, ldwt, row_scale, rs_outer,
                            rs_inner, dh, lddh, M, C, J, nullptr, nullptr, nullptr, 0, stream);
}

// Both modalities' adapters in ONE launch (round 4): rows [0, split_m) with (wt, row_scale, dh), rows [split_m, M) with (wt2, row_scale2, dh2).
// x: fp32 residual rows (xhat == 0; gamma, mean required) or bf16 normalised rows (xhat != 0).  Same arithmetic per row as two launches.
extern "C" int stg_ln_bwd_down_pair(int xhat, const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* mean,
                                    const float* rstd, const void* add_to, int64_t ldadd, void* dx, int64_t lddx, const void* wt, const void* wt2,
                                    int64_t ldwt, const float* row_scale, const float* row_scale2, int64_t rs_outer, int64_t rs_inner, void* dh,
                                    void* dh2, int64_t lddh, int64_t split_m, int64_t M, int C, int J, void* stream) {
    STG_CHECK(split_m > 0, -2, "stg_ln_bwd_down_pair: split_m must be positive");
    return ln_bwd_down_impl("stg_ln_bwd_down_pair", xhat != 0, dy, lddy, x, ldx, gamma, mean, rstd, add_to, ldadd, dx, lddx, wt, ldwt, row_scale, rs_outer,
                            rs_inner, dh, lddh, M, C, J, wt2, row_scale2, dh2, split_m, stream);
}
