// Multi-head self-attention core of the CLIP ViT blocks (nn.MultiheadAttention inside ResidualAttentionBlock.attention,
// CLIP_AVE.py:106-108, spatial calls at :379-383) -- and, with H = 1, K = V and scale 1, the frame-global cross-modal attention
// of wide adapters (softmax(h_v h_a^T) h_a, Swin_AVE.py:801-805; d_h = 96 in every Swin-L stage, 64 in Swin-B stage 3): P frames x H heads, n tokens per frame (197 video / 49 audio for ViT-B/16,
// 257 for ViT-L/14), head dim 64 or 96 (the reference runner builds ViT-B with heads = 8 -> 96, AVE/run_adapt_ave29.py:137), no
// mask, no bias.  Token i of frame p is row p*n + i of the fused qkv buffer.
//
// Why its own family: at head dim 96 the generic gather-mapped kernels (attention.hip) keep per-element bias / mask / map
// bookkeeping next to 96-wide accumulators and spill hundreds of VGPRs to scratch.  These kernels are the plain flash
// schedule with nothing else in it:
//   * a block = one (frame, head) and four consecutive 32-row tiles, one per wave; the K / V tile (forward, dQ) or the Q / dO
//     tile (dK/dV) of the inner loop is staged ONCE per block into LDS by all 256 threads with 16-byte loads and shared by the
//     four waves (register prefetch of the next tile across the barrier);
//   * scores are computed transposed (rows = keys, lane = query) so that the softmax statistics are lane-local and P / dS feed
//     the second MFMA straight from registers as the B operand, the other operand comes k-major out of LDS through
//     ds_read_b64_tr_b16; LDS rows are padded by 16 bytes (208 / 144-byte pitch): conflict-free for both read shapes;
//   * the backward recomputes P from the saved log2-domain LSE; dQ also leaves delta = rowsum(dO * O) for the dK/dV kernel.
// MFMA v_mfma_f32_32x32x16_bf16; lane l = (r = l & 31, hh = l >> 5); accumulator row (reg, hh) = (reg & 3) + 8 (reg >> 2) + 4 hh.
#include <math.h>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

struct MP {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; int64_t ld;
    bf16_t* O; int64_t ldo;
    float* lse;                       // [P, H, n]  log2-domain: max + log2(sum)
    float* delta;                     // [P, H, n]
    int P, H, n, nt;                  // nt = 32-row tiles per frame
    float scale, scale2;
    const bf16_t* dO; int64_t lddo;
    bf16_t* dQ; bf16_t* dK; bf16_t* dV; int64_t lddqkv;
    // pair launch (round 5b): blockIdx.z >= P addresses the SECOND problem set (the other direction of a cross-modal pair, same geometry)
    const bf16_t* Q1; const bf16_t* K1; const bf16_t* V1; bf16_t* O1; float* lse1; float* delta1;
    const bf16_t* dO1; bf16_t* dQ1; bf16_t* dK1; bf16_t* dV1;
    int wh, ww, ws, wshift, nwin;     // window map (ws > 0): problem p = window p % nwin of frame p / nwin of a [wh, ww] token image, cyclic shift wshift
};

__device__ __forceinline__ int pair_select(MP& a, int z) {          // block-uniform: which problem set, and the problem inside it
    if (z < a.P) return z;
    a.Q = a.Q1; a.K = a.K1; a.V = a.V1; a.O = a.O1; a.lse = a.lse1; a.delta = a.delta1;
    a.dO = a.dO1; a.dQ = a.dQ1; a.dK = a.dK1; a.dV = a.dV1;
    return z - a.P;
}

// row of token `tok` of problem p in the token tensors: p * n + tok, or -- window map, round 5: the adapters' WINDOW-level cross-modal attention
// at widths 64 / 96 (Swin-L) -- the token's place in its frame (the arithmetic of attention.hip's map_kind 1)
__device__ __forceinline__ int64_t mrow(const MP& a, int p, int tok) {
    if (a.ws == 0) return (int64_t)p * a.n + tok;
    const int f = p / a.nwin, g = p - f * a.nwin;
    const int nww = a.ww / a.ws;
    const int wi = g / nww, wj = g - wi * nww;
    const int ti = tok / a.ws, tj = tok - ti * a.ws;
    int y = wi * a.ws + ti + a.wshift;
    y = y >= a.wh ? y - a.wh : y;
    int x = wj * a.ws + tj + a.wshift;
    x = x >= a.ww ? x - a.ww : x;
    return (int64_t)f * a.wh * a.ww + y * a.ww + x;
}

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define ACC_ROW(reg, hh) (((reg) & 3) + 8 * ((reg) >> 2) + 4 * (hh))

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t pack8(const float* x) {
    const u32x4_t w = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
    return __builtin_bit_cast(bf16x8_t, w);
}

typedef short s4_t __attribute__((ext_vector_type(4)));
// transposed fragment of a [32 rows][DP pitch] tile: A[i = d][k slot j] = tile[16 s2 + 4 hh + (j & 3) + 8 (j >> 2)][32 dt + d]
template <int DP>
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* s, int dt, int s2, int hh, int d) {
    const int gi = d & 15, c = d >> 4;
    const bf16_t* p = s + (16 * s2 + 4 * hh + (gi >> 2)) * DP + 32 * dt + 16 * c + 4 * (gi & 3);
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)(p + 8 * DP));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
// natural fragment (rows on the lane): tile[r][16 s + 8 hh .. +7]
template <int DP>
__device__ __forceinline__ bf16x8_t nat_frag(const bf16_t* s, int r, int hh, int ks) {
    return *reinterpret_cast<const bf16x8_t*>(s + r * DP + 16 * ks + 8 * hh);
}

// Cooperative staging of NTILE [32][D] tiles (rows row0 .. row0+31 of frame p, clamped to the last token) by 256 threads with 16-byte
// pieces: `stage_plan` fixes, once per thread, which pieces it moves (LDS offset, source column, row inside the tile); `stage_fetch` fills
// registers, `stage_commit` writes them to LDS.  Round 5: plain local arrays passed by reference (as a struct with member functions the
// staging registers lived in SCRATCH: 33 scratch_load / scratch_store per kernel on the critical path between the two barriers of a tile),
// NTILE = 1 where K and V are one tensor (the adapters' cross-modal attention), LDS double-buffered: ONE barrier per tile.
template <int N> using U4Arr = u32x4_t[N];      // a native vector type: arrays of HIP's uint4 struct are not promoted to registers
template <int N> using IArr = int[N];
template <int D, int NTILE, int NW = 4, int TH = 32> struct StageC {      // TH = rows per staged tile (32, or 64 in the two-tiles-per-trip kernels)
    static constexpr int NTHR = NW * 64;
    static constexpr int TOTAL = NTILE * (TH / 8) * D;          // 16-byte pieces: NTILE tiles x TH rows x D / 8
    static constexpr int PER = (TOTAL + NTHR - 1) / NTHR;
    static constexpr int DP = D + 8;
    static constexpr int TILE = TH * DP;                        // elements per LDS tile
};
template <int D, int NTILE, int NW, int TH = 32>
__device__ __forceinline__ void stage_plan(int tid, int h, IArr<StageC<D, NTILE, NW, TH>::PER>& lds_off, IArr<StageC<D, NTILE, NW, TH>::PER>& src_col,
                                           IArr<StageC<D, NTILE, NW, TH>::PER>& row) {
    using C = StageC<D, NTILE, NW, TH>;
#pragma unroll
    for (int i = 0; i < C::PER; ++i) {
        const int id = tid + C::NTHR * i;
        const int which = id / ((TH / 8) * D), rem = id - which * (TH / 8) * D;
        const int r = rem / (D / 8), c = rem - r * (D / 8);
        row[i] = (C::TOTAL % C::NTHR != 0 && id >= C::TOTAL) ? -1 : r + TH * which;        // -1: this thread has no i-th piece; bit log2(TH) = tile
        lds_off[i] = which * C::TILE + r * C::DP + 8 * c;
        src_col[i] = h * D + 8 * c;
    }
}
template <int D, int NTILE, int NW, int TH = 32>
__device__ __forceinline__ void stage_fetch(U4Arr<StageC<D, NTILE, NW, TH>::PER>& v, const IArr<StageC<D, NTILE, NW, TH>::PER>& src_col,
                                            const IArr<StageC<D, NTILE, NW, TH>::PER>& row, const bf16_t* t0, int64_t ld0, const bf16_t* t1, int64_t ld1,
                                            const MP& a, int p, int n, int row0) {
#pragma unroll
    for (int i = 0; i < StageC<D, NTILE, NW, TH>::PER; ++i) {
        if (row[i] < 0) continue;
        int tok = row0 + (row[i] & (TH - 1));
        tok = tok < n ? tok : n - 1;
        const bf16_t* src = (NTILE == 2 && (row[i] & TH)) ? t1 + mrow(a, p, tok) * ld1 : t0 + mrow(a, p, tok) * ld0;
        v[i] = *reinterpret_cast<const u32x4_t*>(src + src_col[i]);
    }
}
template <int D, int NTILE, int NW, int TH = 32>
__device__ __forceinline__ void stage_commit(bf16_t* s, const U4Arr<StageC<D, NTILE, NW, TH>::PER>& v, const IArr<StageC<D, NTILE, NW, TH>::PER>& lds_off,
                                             const IArr<StageC<D, NTILE, NW, TH>::PER>& row) {
#pragma unroll
    for (int i = 0; i < StageC<D, NTILE, NW, TH>::PER; ++i)
        if (row[i] >= 0) *reinterpret_cast<u32x4_t*>(s + lds_off[i]) = v[i];
}

// KV1 (every kernel below): K and V are the same tensor (one staged tile serves the score MFMA and, read transposed, the P.V MFMA)
// ------------------------------------------------------------------------------------------------ forward / dQ, TWO key tiles per trip (round 5)
// The round-4 one-tile kernels (removed in round 6; the one-tile dK / dV kernel remains for the register-limited combinations) ran a tile's phases one after the other inside a wave -- score MFMAs, wait, softmax VALU, transposed reads, P.V MFMAs --
// and met only other WAVES' work to fill the gaps (SQ counters: waves waiting 52-57 % of their cycles, matrix pipe busy 28-30 %,
// profiles/r05_mha_sq_counters.txt).  Here a trip covers 64 keys: the two tiles' MFMA chains are independent, so tile B's score MFMAs execute while
// tile A's softmax issues, the statistics are updated once per 64 keys, and there is one barrier per 64 keys.  Staged tiles are [64][D + 8].
template <int D, bool KV1, int NW>
__global__ void __launch_bounds__(NW * 64, NW <= 4 ? 2 : 1) mha_fwd2_kernel(MP a) {
    constexpr int KS = D / 16, DT = D / 32, DP = D + 8, NTILE = KV1 ? 1 : 2;
    using SC = StageC<D, NTILE, NW, 64>;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];                  // two buffers x NTILE x [64][DP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qb = blockIdx.x, h = blockIdx.y, p = pair_select(a, blockIdx.z);
    const int q0 = 32 * (qb * NW + wave);
    const bool live = q0 < a.n;
    const int q = q0 + r;
    const int qc = q < a.n ? q : a.n - 1;
    bf16x8_t qf[KS];
    {
        const bf16_t* qp = a.Q + mrow(a, p, qc) * a.ld + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = ld_frag(qp + 16 * s);
    }
    f32x16_t o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
    float m = NEG_BIG, l = 0.f;

    int lo[SC::PER], sc_[SC::PER], rw[SC::PER];
    u32x4_t sv[SC::PER];
    stage_plan<D, NTILE, NW, 64>(tid, h, lo, sc_, rw);
    stage_fetch<D, NTILE, NW, 64>(sv, sc_, rw, a.K, a.ld, a.V, a.ld, a, p, a.n, 0);
    stage_commit<D, NTILE, NW, 64>(smem, sv, lo, rw);
    __syncthreads();
    const int np = (a.nt + 1) >> 1;
    for (int kp = 0; kp < np; ++kp) {
        const bf16_t* sK = smem + (kp & 1) * NTILE * SC::TILE;
        const bf16_t* sV = KV1 ? sK : sK + SC::TILE;
        if (kp + 1 < np) stage_fetch<D, NTILE, NW, 64>(sv, sc_, rw, a.K, a.ld, a.V, a.ld, a, p, a.n, 64 * (kp + 1));
        if (live) {
        f32x16_t sA = zero16(), sB = zero16();         // St[key][q] of keys 64 kp .. + 31 / + 32 .. + 63
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            sA = MFMA32(nat_frag<DP>(sK, r, hh, s), qf[s], sA);
            sB = MFMA32(nat_frag<DP>(sK + 32 * DP, r, hh, s), qf[s], sB);
        }
        // round 6: the maximum is taken on the RAW scores (scale > 0) and the exponent is one fma, exp2(s scale2 - m): one VALU slot per score less
        float xA[16], xB[16];
        float mr = NEG_BIG;
        const int kbase = 64 * kp;
        const bool tail = kbase + 64 > a.n;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            xA[reg] = sA[reg];
            xB[reg] = sB[reg];
            if (tail) {
                if (kbase + ACC_ROW(reg, hh) >= a.n) xA[reg] = NEG_BIG;
                if (kbase + 32 + ACC_ROW(reg, hh) >= a.n) xB[reg] = NEG_BIG;
            }
            mr = fmaxf(mr, fmaxf(xA[reg], xB[reg]));
        }
        mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
        const float mx = fmaxf(m, mr * a.scale2);
        const float alpha = __builtin_amdgcn_exp2f(m - mx);
        m = mx;
        const float nmx = -mx;
        float ls = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            xA[reg] = __builtin_amdgcn_exp2f(fmaf(xA[reg], a.scale2, nmx));
            xB[reg] = __builtin_amdgcn_exp2f(fmaf(xB[reg], a.scale2, nmx));
            ls += xA[reg] + xB[reg];
        }
        ls += __shfl_xor(ls, 32, 64);
        l = l * alpha + ls;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {       // rescale only when some query's maximum moved
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) o[dt][reg] *= alpha;
        }
        const bf16x8_t pA0 = pack8(xA), pA1 = pack8(xA + 8), pB0 = pack8(xB), pB1 = pack8(xB + 8);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            o[dt] = MFMA32(tr_frag<DP>(sV, dt, 0, hh, r), pA0, o[dt]);
            o[dt] = MFMA32(tr_frag<DP>(sV, dt, 1, hh, r), pA1, o[dt]);
            o[dt] = MFMA32(tr_frag<DP>(sV + 32 * DP, dt, 0, hh, r), pB0, o[dt]);
            o[dt] = MFMA32(tr_frag<DP>(sV + 32 * DP, dt, 1, hh, r), pB1, o[dt]);
        }
        }
        if (kp + 1 < np) stage_commit<D, NTILE, NW, 64>(smem + ((kp + 1) & 1) * NTILE * SC::TILE, sv, lo, rw);
        __syncthreads();
    }
    {
        const float inv = 1.0f / l;
        bf16_t* op = a.O + mrow(a, p, q) * a.ldo + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store_tile32(op + 32 * dt, o[dt], inv, hh, q < a.n);
        if (q < a.n && a.lse && hh == 0) a.lse[((int64_t)p * a.H + h) * a.n + q] = m + __log2f(l);
    }
}

template <int D, bool KV1, int NW>
__global__ void __launch_bounds__(NW * 64, NW <= 4 ? 2 : 1) mha_dq2_kernel(MP a) {
    constexpr int KS = D / 16, DT = D / 32, DP = D + 8, NTILE = KV1 ? 1 : 2;
    using SC = StageC<D, NTILE, NW, 64>;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qb = blockIdx.x, h = blockIdx.y, p = pair_select(a, blockIdx.z);
    const int q = 32 * (qb * NW + wave) + r;
    const bool live = 32 * (qb * NW + wave) < a.n;
    const int qc = q < a.n ? q : a.n - 1;
    bf16x8_t qf[KS], dof[KS];
    float delta = 0.f;
    {
        const bf16_t* qp = a.Q + mrow(a, p, qc) * a.ld + h * D + 8 * hh;
        const bf16_t* dp = a.dO + mrow(a, p, qc) * a.lddo + h * D + 8 * hh;
        const bf16_t* op = a.O + mrow(a, p, qc) * a.ldo + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = ld_frag(qp + 16 * s);
            dof[s] = ld_frag(dp + 16 * s);
            const bf16x8_t of = ld_frag(op + 16 * s);
#pragma unroll
            for (int j = 0; j < 8; ++j) delta += bf2f((bf16_t)dof[s][j]) * bf2f((bf16_t)of[j]);
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    const int64_t si = ((int64_t)p * a.H + h) * a.n + qc;
    const float lse = a.lse[si];
    if (q < a.n && hh == 0) a.delta[si] = delta;
    f32x16_t dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = zero16();

    int lo[SC::PER], sc_[SC::PER], rw[SC::PER];
    u32x4_t sv[SC::PER];
    stage_plan<D, NTILE, NW, 64>(tid, h, lo, sc_, rw);
    stage_fetch<D, NTILE, NW, 64>(sv, sc_, rw, a.K, a.ld, a.V, a.ld, a, p, a.n, 0);
    stage_commit<D, NTILE, NW, 64>(smem, sv, lo, rw);
    __syncthreads();
    const int np = (a.nt + 1) >> 1;
    for (int kp = 0; kp < np; ++kp) {
        const bf16_t* sK = smem + (kp & 1) * NTILE * SC::TILE;
        const bf16_t* sV = KV1 ? sK : sK + SC::TILE;
        if (kp + 1 < np) stage_fetch<D, NTILE, NW, 64>(sv, sc_, rw, a.K, a.ld, a.V, a.ld, a, p, a.n, 64 * (kp + 1));
        if (live) {
        f32x16_t scA = zero16(), dpA = zero16(), scB = zero16(), dpB = zero16();         // St[key][q], dPt[key][q] of the two key tiles
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8_t kA = nat_frag<DP>(sK, r, hh, s), kB = nat_frag<DP>(sK + 32 * DP, r, hh, s);
            scA = MFMA32(kA, qf[s], scA);
            scB = MFMA32(kB, qf[s], scB);
            dpA = MFMA32(KV1 ? kA : nat_frag<DP>(sV, r, hh, s), dof[s], dpA);
            dpB = MFMA32(KV1 ? kB : nat_frag<DP>(sV + 32 * DP, r, hh, s), dof[s], dpB);
        }
        float dsA[16], dsB[16];
        const int kbase = 64 * kp;
        const bool tail = kbase + 64 > a.n;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            float pa = __builtin_amdgcn_exp2f(scA[reg] * a.scale2 - lse), pb = __builtin_amdgcn_exp2f(scB[reg] * a.scale2 - lse);
            if (tail) {
                if (kbase + ACC_ROW(reg, hh) >= a.n) pa = 0.f;
                if (kbase + 32 + ACC_ROW(reg, hh) >= a.n) pb = 0.f;
            }
            dsA[reg] = pa * (dpA[reg] - delta);
            dsB[reg] = pb * (dpB[reg] - delta);
        }
        const bf16x8_t a0 = pack8(dsA), a1 = pack8(dsA + 8), b0 = pack8(dsB), b1 = pack8(dsB + 8);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dq[dt] = MFMA32(tr_frag<DP>(sK, dt, 0, hh, r), a0, dq[dt]);
            dq[dt] = MFMA32(tr_frag<DP>(sK, dt, 1, hh, r), a1, dq[dt]);
            dq[dt] = MFMA32(tr_frag<DP>(sK + 32 * DP, dt, 0, hh, r), b0, dq[dt]);
            dq[dt] = MFMA32(tr_frag<DP>(sK + 32 * DP, dt, 1, hh, r), b1, dq[dt]);
        }
        }
        if (kp + 1 < np) stage_commit<D, NTILE, NW, 64>(smem + ((kp + 1) & 1) * NTILE * SC::TILE, sv, lo, rw);
        __syncthreads();
    }
    {
        bf16_t* op = a.dQ + mrow(a, p, q) * a.lddqkv + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store_tile32(op + 32 * dt, dq[dt], a.scale, hh, q < a.n);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <int D, bool KV1, int NW>
__global__ void __launch_bounds__(NW * 64, NW <= 4 ? 2 : 1) mha_dkv_kernel(MP a) {
    constexpr int KS = D / 16, DT = D / 32, DP = D + 8, NKV = KV1 ? 1 : 2;
    using SC = StageC<D, 2, NW>;
    // two buffers of {shared Q / dO tiles + lse / delta of the query tile}, then per wave its own K (and V) tile
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];      // dkv_lds_bytes(D, KV1) (dynamic: > 64 KiB at D = 96)
    constexpr int BUF = 2 * 32 * DP + 128;                              // elements per buffer (64 floats of statistics = 128 bf16 slots)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    bf16_t* sK = smem + 2 * BUF + wave * NKV * 32 * DP;
    bf16_t* sV = KV1 ? sK : sK + 32 * DP;
    const int kb = blockIdx.x, h = blockIdx.y, p = pair_select(a, blockIdx.z);
    const int key = 32 * (kb * NW + wave) + r;
    const bool live = 32 * (kb * NW + wave) < a.n;
    const int kc = key < a.n ? key : a.n - 1;
    {   // own K / V tile: natural rows (keys) -> wave-private LDS; read back as B operands (columns = keys)
        const bf16_t* kp = a.K + mrow(a, p, kc) * a.ld + h * D + 8 * hh;
        const bf16_t* vp = a.V + mrow(a, p, kc) * a.ld + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8_t*>(sK + r * DP + 16 * s + 8 * hh) = ld_frag(kp + 16 * s);
            if (!KV1) *reinterpret_cast<bf16x8_t*>(sV + r * DP + 16 * s + 8 * hh) = ld_frag(vp + 16 * s);
        }
    }
    f32x16_t dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[dt] = zero16(); dv[dt] = zero16(); }
    const int64_t sbase = ((int64_t)p * a.H + h) * a.n;

    int lo[SC::PER], sc_[SC::PER], rw[SC::PER];
    u32x4_t sv[SC::PER];
    stage_plan<D, 2, NW>(tid, h, lo, sc_, rw);
    float nl = 0.f, nd = 0.f;
    auto fetch_stats = [&](int q0) {
        if (tid < 32) {
            int qi = q0 + tid;
            qi = qi < a.n ? qi : a.n - 1;
            nl = a.lse[sbase + qi];
            nd = a.delta[sbase + qi];
        }
    };
    auto commit_all = [&](bf16_t* buf) {
        stage_commit<D, 2, NW>(buf, sv, lo, rw);
        float* st = reinterpret_cast<float*>(buf + 2 * 32 * DP);
        if (tid < 32) { st[tid] = nl; st[32 + tid] = nd; }
    };
    stage_fetch<D, 2, NW>(sv, sc_, rw, a.Q, a.ld, a.dO, a.lddo, a, p, a.n, 0);
    fetch_stats(0);
    commit_all(smem);
    __syncthreads();
    for (int qt = 0; qt < a.nt; ++qt) {
        const bf16_t* sQ = smem + (qt & 1) * BUF;
        const bf16_t* sD = sQ + 32 * DP;
        const float* sLse = reinterpret_cast<const float*>(sQ + 2 * 32 * DP);
        const float* sDel = sLse + 32;
        if (qt + 1 < a.nt) {
            stage_fetch<D, 2, NW>(sv, sc_, rw, a.Q, a.ld, a.dO, a.lddo, a, p, a.n, 32 * (qt + 1));
            fetch_stats(32 * (qt + 1));
        }
        if (live) {
        f32x16_t sc = zero16(), dp = zero16();         // S[q][key], dP[q][key]
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8_t kf = nat_frag<DP>(sK, r, hh, s);
            sc = MFMA32(nat_frag<DP>(sQ, r, hh, s), kf, sc);
            dp = MFMA32(nat_frag<DP>(sD, r, hh, s), KV1 ? kf : nat_frag<DP>(sV, r, hh, s), dp);
        }
        float pr[16], ds[16];
        const int qbase = 32 * qt;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 ls = *reinterpret_cast<const float4*>(sLse + 8 * g4 + 4 * hh);
            const float4 de = *reinterpret_cast<const float4*>(sDel + 8 * g4 + 4 * hh);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int reg = 4 * g4 + c;
                const bool okq = qbase + 8 * g4 + 4 * hh + c < a.n;              // padded query rows contribute nothing
                const float pv = okq ? __builtin_amdgcn_exp2f(sc[reg] * a.scale2 - lsv[c]) : 0.f;
                pr[reg] = pv;
                ds[reg] = pv * (dp[reg] - dev[c]);
            }
        }
        const bf16x8_t p0 = pack8(pr), p1 = pack8(pr + 8), d0 = pack8(ds), d1 = pack8(ds + 8);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = MFMA32(tr_frag<DP>(sD, dt, 0, hh, r), p0, dv[dt]);
            dv[dt] = MFMA32(tr_frag<DP>(sD, dt, 1, hh, r), p1, dv[dt]);
            dk[dt] = MFMA32(tr_frag<DP>(sQ, dt, 0, hh, r), d0, dk[dt]);
            dk[dt] = MFMA32(tr_frag<DP>(sQ, dt, 1, hh, r), d1, dk[dt]);
        }
        }
        if (qt + 1 < a.nt) commit_all(smem + ((qt + 1) & 1) * BUF);
        __syncthreads();
    }
    {
        const bool okk = key < a.n;
        bf16_t* kp = a.dK + mrow(a, p, key) * a.lddqkv + h * D;
        if (a.dV == nullptr) {                         // K and V are ONE tensor (cross-modal attention): its gradient is dK + dV
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x16_t c;
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = fmaf(dk[dt][i], a.scale, dv[dt][i]);
                store_tile32(kp + 32 * dt, c, 1.0f, hh, okk);
            }
            return;
        }
        bf16_t* vp = a.dV + mrow(a, p, key) * a.lddqkv + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            store_tile32(kp + 32 * dt, dk[dt], a.scale, hh, okk);
            store_tile32(vp + 32 * dt, dv[dt], 1.0f, hh, okk);
        }
    }
}

// two QUERY tiles per trip (round 5): the staged Q / dO tiles are [64][D + 8], one barrier per 64 queries
template <int D, bool KV1, int NW>
__global__ void __launch_bounds__(NW * 64, NW <= 4 ? 2 : 1) mha_dkv2_kernel(MP a) {
    constexpr int KS = D / 16, DT = D / 32, DP = D + 8, NKV = KV1 ? 1 : 2;
    using SC = StageC<D, 2, NW, 64>;
    // two buffers of {shared Q / dO tiles + lse / delta of the query tile}, then per wave its own K (and V) tile
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];      // dkv_lds_bytes(D, KV1) (dynamic: > 64 KiB at D = 96)
    constexpr int BUF = 2 * 64 * DP + 256;                              // elements per buffer: Q and dO tiles of 64 queries + 128 floats of statistics
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    bf16_t* sK = smem + 2 * BUF + wave * NKV * 32 * DP;
    bf16_t* sV = KV1 ? sK : sK + 32 * DP;
    const int kb = blockIdx.x, h = blockIdx.y, p = pair_select(a, blockIdx.z);
    const int key = 32 * (kb * NW + wave) + r;
    const bool live = 32 * (kb * NW + wave) < a.n;
    const int kc = key < a.n ? key : a.n - 1;
    {   // own K / V tile: natural rows (keys) -> wave-private LDS; read back as B operands (columns = keys)
        const bf16_t* kp = a.K + mrow(a, p, kc) * a.ld + h * D + 8 * hh;
        const bf16_t* vp = a.V + mrow(a, p, kc) * a.ld + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8_t*>(sK + r * DP + 16 * s + 8 * hh) = ld_frag(kp + 16 * s);
            if (!KV1) *reinterpret_cast<bf16x8_t*>(sV + r * DP + 16 * s + 8 * hh) = ld_frag(vp + 16 * s);
        }
    }
    f32x16_t dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[dt] = zero16(); dv[dt] = zero16(); }
    const int64_t sbase = ((int64_t)p * a.H + h) * a.n;

    int lo[SC::PER], sc_[SC::PER], rw[SC::PER];
    u32x4_t sv[SC::PER];
    stage_plan<D, 2, NW, 64>(tid, h, lo, sc_, rw);
    float nl = 0.f, nd = 0.f;
    auto fetch_stats = [&](int q0) {
        if (tid < 64) {
            int qi = q0 + tid;
            qi = qi < a.n ? qi : a.n - 1;
            nl = a.lse[sbase + qi];
            nd = a.delta[sbase + qi];
        }
    };
    auto commit_all = [&](bf16_t* buf) {
        stage_commit<D, 2, NW, 64>(buf, sv, lo, rw);
        float* st = reinterpret_cast<float*>(buf + 2 * 64 * DP);
        if (tid < 64) { st[tid] = nl; st[64 + tid] = nd; }
    };
    stage_fetch<D, 2, NW, 64>(sv, sc_, rw, a.Q, a.ld, a.dO, a.lddo, a, p, a.n, 0);
    fetch_stats(0);
    commit_all(smem);
    __syncthreads();
    const int np = (a.nt + 1) >> 1;
    for (int qp = 0; qp < np; ++qp) {
        const bf16_t* sQ = smem + (qp & 1) * BUF;
        const bf16_t* sD = sQ + 64 * DP;
        const float* sLse = reinterpret_cast<const float*>(sQ + 2 * 64 * DP);
        const float* sDel = sLse + 64;
        if (qp + 1 < np) {
            stage_fetch<D, 2, NW, 64>(sv, sc_, rw, a.Q, a.ld, a.dO, a.lddo, a, p, a.n, 64 * (qp + 1));
            fetch_stats(64 * (qp + 1));
        }
        if (live) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {             // the two query tiles of the trip: independent MFMA chains behind one barrier
            const bf16_t* tQ = sQ + half * 32 * DP;
            const bf16_t* tD = sD + half * 32 * DP;
            f32x16_t sc = zero16(), dp = zero16();         // S[q][key], dP[q][key]
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8_t kf = nat_frag<DP>(sK, r, hh, s);
                sc = MFMA32(nat_frag<DP>(tQ, r, hh, s), kf, sc);
                dp = MFMA32(nat_frag<DP>(tD, r, hh, s), KV1 ? kf : nat_frag<DP>(sV, r, hh, s), dp);
            }
            float pr[16], ds[16];
            const int qbase = 64 * qp + 32 * half;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 ls = *reinterpret_cast<const float4*>(sLse + 32 * half + 8 * g4 + 4 * hh);
                const float4 de = *reinterpret_cast<const float4*>(sDel + 32 * half + 8 * g4 + 4 * hh);
                const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int reg = 4 * g4 + c;
                    const bool okq = qbase + 8 * g4 + 4 * hh + c < a.n;              // padded query rows contribute nothing
                    const float pv = okq ? __builtin_amdgcn_exp2f(sc[reg] * a.scale2 - lsv[c]) : 0.f;
                    pr[reg] = pv;
                    ds[reg] = pv * (dp[reg] - dev[c]);
                }
            }
            const bf16x8_t p0 = pack8(pr), p1 = pack8(pr + 8), d0 = pack8(ds), d1 = pack8(ds + 8);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = MFMA32(tr_frag<DP>(tD, dt, 0, hh, r), p0, dv[dt]);
                dv[dt] = MFMA32(tr_frag<DP>(tD, dt, 1, hh, r), p1, dv[dt]);
                dk[dt] = MFMA32(tr_frag<DP>(tQ, dt, 0, hh, r), d0, dk[dt]);
                dk[dt] = MFMA32(tr_frag<DP>(tQ, dt, 1, hh, r), d1, dk[dt]);
            }
        }
        }
        if (qp + 1 < np) commit_all(smem + ((qp + 1) & 1) * BUF);
        __syncthreads();
    }
    {
        const bool okk = key < a.n;
        bf16_t* kp = a.dK + mrow(a, p, key) * a.lddqkv + h * D;
        if (a.dV == nullptr) {                         // K and V are ONE tensor (cross-modal attention): its gradient is dK + dV
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x16_t c;
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = fmaf(dk[dt][i], a.scale, dv[dt][i]);
                store_tile32(kp + 32 * dt, c, 1.0f, hh, okk);
            }
            return;
        }
        bf16_t* vp = a.dV + mrow(a, p, key) * a.lddqkv + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            store_tile32(kp + 32 * dt, dk[dt], a.scale, hh, okk);
            store_tile32(vp + 32 * dt, dv[dt], 1.0f, hh, okk);
        }
    }
}

// ------------------------------------------------------------------------------------------------ merged pair backward (round 6)
// The BACKWARD of a cross-modal pair (K == V in both directions, direction x: queries X, keys = values Y; direction y the mirror image) as ONE pass
// per modality instead of dQ + (dK + dV) passes per direction.  Both directions share the score matrix S = X Y^T (x normalises its rows, y its
// columns), and the gradient of a modality's rows is
//     G_x[i] = dQ^x[i] + dK^y[i] + dV^y[i] = scale sum_j (dS^x[i, j] + dS^y[j, i]) Y[j] + sum_j P^y[j, i] dO_y[j],
//     dS^x = P^x o (dO_x Y^T - delta_x[i]),     dS^y[j, i] = P^y[j, i] (dO_y[j] . X[i] - delta_y[j]),
// so per 32-key tile a wave of 32 rows i needs THREE score-type products (S, dO_x Y^T, dO_y X^T: the first two share the Y fragments) and TWO
// output-type products (the two dS folded into ONE operand before the product with Y^T; P^y with dO_y^T): 3 KS + 4 DT MFMAs = 30 at D = 96, where
// dQ (2 KS + 2 DT = 18) + dK / dV (2 KS + 4 DT = 24) spend 42 and read 36 KB of LDS fragments instead of 24.  Staging is mha_dkv2_kernel's (the other
// modality's rows Y and dO_y in 64-row tiles + its lse / delta), accumulators are mha_dq2_kernel's.  delta of both directions comes from
// mha_delta_kernel (each pass needs the OTHER direction's, which no block of this launch could have produced in time).
template <int D>
__global__ void __launch_bounds__(256) mha_delta_kernel(MP a) {
    // one 8-lane group per (problem, head, token): delta = sum_d dO[row][h D + d] O[row][h D + d]; blockIdx.y = direction
    const bool second = blockIdx.y != 0;
    const bf16_t* O = second ? a.O1 : a.O;
    const bf16_t* dO = second ? a.dO1 : a.dO;
    float* del = second ? a.delta1 : a.delta;
    const int64_t total = (int64_t)a.P * a.H * a.n;
    const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    if (item >= total) return;                             // whole 8-lane groups leave together (the shuffles below stay inside a group)
    const int tok = (int)(item % a.n);
    const int64_t ph = item / a.n;
    const int h = (int)(ph % a.H), p = (int)(ph / a.H);
    const int64_t row = mrow(a, p, tok);
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < (D / 8 + 7) / 8; ++c) {
        const int ch = l + 8 * c;
        if (ch < D / 8) {
            const bf16x8_t x = ld_frag(dO + row * a.lddo + h * D + 8 * ch), y = ld_frag(O + row * a.ldo + h * D + 8 * ch);
#pragma unroll
            for (int j = 0; j < 8; ++j) d += bf2f((bf16_t)x[j]) * bf2f((bf16_t)y[j]);
        }
    }
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (l == 0) del[item] = d;
}

template <int D, int NW>
__global__ void __launch_bounds__(NW * 64, NW <= 4 ? 2 : 1) mha_bwdm_kernel(MP a) {
    constexpr int KS = D / 16, DT = D / 32, DP = D + 8;
    using SC = StageC<D, 2, NW, 64>;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];      // two buffers of {Y and dO_y tiles of 64 rows + 128 floats of statistics}
    constexpr int BUF = 2 * 64 * DP + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qb = blockIdx.x, h = blockIdx.y;
    const bool second = (int)blockIdx.z >= a.P;                        // block-uniform: which modality's rows this block owns
    const int p = second ? blockIdx.z - a.P : blockIdx.z;
    const bf16_t* X = second ? a.Q1 : a.Q;
    const bf16_t* Y = second ? a.K1 : a.K;
    const bf16_t* dOx = second ? a.dO1 : a.dO;
    const bf16_t* dOy = second ? a.dO : a.dO1;
    const float* lse_x = second ? a.lse1 : a.lse;
    const float* lse_y = second ? a.lse : a.lse1;
    const float* del_x = second ? a.delta1 : a.delta;
    const float* del_y = second ? a.delta : a.delta1;
    bf16_t* G = second ? a.dQ1 : a.dQ;
    const int q = 32 * (qb * NW + wave) + r;
    const bool live = 32 * (qb * NW + wave) < a.n;
    const int qc = q < a.n ? q : a.n - 1;
    bf16x8_t xf[KS], dxf[KS];
    {
        const bf16_t* xp = X + mrow(a, p, qc) * a.ld + h * D + 8 * hh;
        const bf16_t* dp = dOx + mrow(a, p, qc) * a.lddo + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) { xf[s] = ld_frag(xp + 16 * s); dxf[s] = ld_frag(dp + 16 * s); }
    }
    const int64_t sbase = ((int64_t)p * a.H + h) * a.n;
    const float nlse_q = -lse_x[sbase + qc], ndel_q = -del_x[sbase + qc];
    const bool unit_scale = a.scale == 1.0f;               // (block-uniform) the cross-modal pairs run with scale 1
    f32x16_t g[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) g[dt] = zero16();

    int lo[SC::PER], sc_[SC::PER], rw[SC::PER];
    u32x4_t sv[SC::PER];
    stage_plan<D, 2, NW, 64>(tid, h, lo, sc_, rw);
    float nl = 0.f, nd = 0.f;
    auto fetch_stats = [&](int j0) {
        if (tid < 64) {
            int j = j0 + tid;
            j = j < a.n ? j : a.n - 1;
            nl = -lse_y[sbase + j];                          // stored NEGATED: the exponent is one fma, exp2(s scale2 - lse)
            nd = del_y[sbase + j];
        }
    };
    auto commit_all = [&](bf16_t* buf) {
        stage_commit<D, 2, NW, 64>(buf, sv, lo, rw);
        float* st = reinterpret_cast<float*>(buf + 2 * 64 * DP);
        if (tid < 64) { st[tid] = nl; st[64 + tid] = nd; }
    };
    stage_fetch<D, 2, NW, 64>(sv, sc_, rw, Y, a.ld, dOy, a.lddo, a, p, a.n, 0);
    fetch_stats(0);
    commit_all(smem);
    __syncthreads();
    const int np = (a.nt + 1) >> 1;
    for (int kp = 0; kp < np; ++kp) {
        const bf16_t* sY = smem + (kp & 1) * BUF;
        const bf16_t* sD = sY + 64 * DP;
        const float* sLse = reinterpret_cast<const float*>(sY + 2 * 64 * DP);
        const float* sDel = sLse + 64;
        if (kp + 1 < np) {
            stage_fetch<D, 2, NW, 64>(sv, sc_, rw, Y, a.ld, dOy, a.lddo, a, p, a.n, 64 * (kp + 1));
            fetch_stats(64 * (kp + 1));
        }
        if (live) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {             // the two 32-row tiles of the trip, one after the other (register budget)
            const int jbase = 64 * kp + 32 * half;
            if (jbase >= a.n) continue;                     // (wave-uniform) an all-padding tile contributes nothing
            const bf16_t* tY = sY + half * 32 * DP;
            const bf16_t* tD = sD + half * 32 * DP;
            const bool tail = jbase + 32 > a.n;             // (wave-uniform) only the frame's last tile masks its padded rows
            f32x16_t st = zero16(), dpx, dpy = zero16();    // S^T[j][i], (dO_x Y^T)^T[j][i] - delta_x[i] (the MFMA's C operand starts there), (dO_y X^T)[j][i]
#pragma unroll
            for (int i = 0; i < 16; ++i) dpx[i] = ndel_q;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8_t yf = nat_frag<DP>(tY, r, hh, s);
                st = MFMA32(yf, xf[s], st);
                dpx = MFMA32(yf, dxf[s], dpx);
                dpy = MFMA32(nat_frag<DP>(tD, r, hh, s), xf[s], dpy);
            }
            float ds[16], py[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 ls = *reinterpret_cast<const float4*>(sLse + 32 * half + 8 * g4 + 4 * hh);
                const float4 de = *reinterpret_cast<const float4*>(sDel + 32 * half + 8 * g4 + 4 * hh);
                const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int reg = 4 * g4 + c;
                    float px = __builtin_amdgcn_exp2f(fmaf(st[reg], a.scale2, nlse_q));     // P^x[i, j]: direction x normalises over j
                    float pv = __builtin_amdgcn_exp2f(fmaf(st[reg], a.scale2, lsv[c]));     // P^y[j, i]: direction y normalises over i (lsv = -lse_y[j])
                    if (tail && jbase + 8 * g4 + 4 * hh + c >= a.n) { px = 0.f; pv = 0.f; }  // padded rows j contribute nothing
                    py[reg] = pv;
                    const float d = fmaf(px, dpx[reg], pv * (dpy[reg] - dev[c]));
                    ds[reg] = unit_scale ? d : d * a.scale;
                }
            }
            const bf16x8_t d0 = pack8(ds), d1 = pack8(ds + 8), p0 = pack8(py), p1 = pack8(py + 8);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                g[dt] = MFMA32(tr_frag<DP>(tY, dt, 0, hh, r), d0, g[dt]);
                g[dt] = MFMA32(tr_frag<DP>(tY, dt, 1, hh, r), d1, g[dt]);
                g[dt] = MFMA32(tr_frag<DP>(tD, dt, 0, hh, r), p0, g[dt]);
                g[dt] = MFMA32(tr_frag<DP>(tD, dt, 1, hh, r), p1, g[dt]);
            }
        }
        }
        if (kp + 1 < np) commit_all(smem + ((kp + 1) & 1) * BUF);
        __syncthreads();
    }
    {
        bf16_t* op = G + mrow(a, p, q) * a.lddqkv + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store_tile32(op + 32 * dt, g[dt], 1.0f, hh, q < a.n);
    }
}

constexpr int bwdm_lds_bytes(int D) { return 2 * (2 * 64 * (D + 8) + 256) * 2; }
template <int D, int NW>
int launch_bwdm(const dim3& grid, const MP& p, hipStream_t stream) {
    static std::atomic<uint64_t> done{0};
    STG_CHECK(stg_reserve_lds(mha_bwdm_kernel<D, NW>, bwdm_lds_bytes(D), done), -101, "stg_mha_bwd_pair_merged: cannot reserve %d bytes of LDS", bwdm_lds_bytes(D));
    hipLaunchKernelGGL((mha_bwdm_kernel<D, NW>), grid, dim3(NW * 64), bwdm_lds_bytes(D), stream, p);
    return 0;
}

constexpr int dkv_lds_bytes(int D, bool kv1, int nw) { return (2 * (2 * 32 * (D + 8) + 128) + nw * (kv1 ? 1 : 2) * 32 * (D + 8)) * 2; }

template <int D, bool KV1, int NW>
int launch_dkv(const dim3& grid, const MP& p, hipStream_t stream) {
    static std::atomic<uint64_t> done{0};
    const bool attr_set = stg_reserve_lds(mha_dkv_kernel<D, KV1, NW>, dkv_lds_bytes(D, KV1, NW), done);
    STG_CHECK(attr_set, -101, "stg_mha_bwd: cannot reserve %d bytes of LDS", dkv_lds_bytes(D, KV1, NW));
    hipLaunchKernelGGL((mha_dkv_kernel<D, KV1, NW>), grid, dim3(NW * 64), dkv_lds_bytes(D, KV1, NW), stream, p);
    return 0;
}

template <int D, bool KV1, int NW>
int launch_fwd2(const dim3& grid, const MP& p, hipStream_t stream) {
    static std::atomic<uint64_t> done{0};
    constexpr int lds = 2 * (KV1 ? 1 : 2) * 64 * (D + 8) * 2;
    STG_CHECK(stg_reserve_lds(mha_fwd2_kernel<D, KV1, NW>, lds, done), -101, "stg_mha_fwd: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((mha_fwd2_kernel<D, KV1, NW>), grid, dim3(NW * 64), lds, stream, p);
    return 0;
}
template <int D, bool KV1, int NW>
int launch_dq2(const dim3& grid, const MP& p, hipStream_t stream) {
    static std::atomic<uint64_t> done{0};
    constexpr int lds = 2 * (KV1 ? 1 : 2) * 64 * (D + 8) * 2;
    STG_CHECK(stg_reserve_lds(mha_dq2_kernel<D, KV1, NW>, lds, done), -101, "stg_mha_bwd: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((mha_dq2_kernel<D, KV1, NW>), grid, dim3(NW * 64), lds, stream, p);
    return 0;
}

constexpr int dkv2_lds_bytes(int D, bool kv1, int nw) { return (2 * (2 * 64 * (D + 8) + 256) + nw * (kv1 ? 1 : 2) * 32 * (D + 8)) * 2; }
template <int D, bool KV1, int NW>
int launch_dkv2(const dim3& grid, const MP& p, hipStream_t stream) {
    static std::atomic<uint64_t> done{0};
    STG_CHECK(stg_reserve_lds(mha_dkv2_kernel<D, KV1, NW>, dkv2_lds_bytes(D, KV1, NW), done), -101, "stg_mha_bwd: cannot reserve %d bytes of LDS", dkv2_lds_bytes(D, KV1, NW));
    hipLaunchKernelGGL((mha_dkv2_kernel<D, KV1, NW>), grid, dim3(NW * 64), dkv2_lds_bytes(D, KV1, NW), stream, p);
    return 0;
}

// waves per block.  Backward: 8 where a frame has more than 4 query tiles (one staging of a tile then serves 256 rows; ViT-B's 197 tokens take ONE
// block per (frame, head)): measured -11 % at 3136 x 3136 x 96 (5589 -> 4971 us), -9 % at 8 x 197 x 197 x 96 (532 -> 485 us).  Forward: 4 -- with 8 it
// measured +5 % on both shapes (its trip is shorter, the barrier among 8 waves weighs more).
// Round 5b: 2 where a problem has at most two query tiles and K == V (the window-level cross-modal pairs: 49 tokens) -- with 4 waves per block half of
// every block idled and a CU held 4 live waves.
int pick_nw(int nt, bool bwd, bool kv1 = false) {
    if (kv1 && nt <= 2) return 2;
    return bwd && nt > 4 ? 8 : 4;
}

int fill(const stg_mha_args* f, MP& p, const char* who) {
    STG_CHECK(f->Q && f->K && f->V && f->O && f->lse, -1, "%s: null pointer", who);
    STG_CHECK(f->D == 64 || f->D == 96, -2, "%s: head dim must be 64 or 96", who);
    STG_CHECK(f->scale > 0.f, -2, "%s: scale must be positive (the forward takes its running maximum on the raw scores)", who);
    STG_CHECK(f->P >= 0 && f->P < 65536 && f->H >= 1 && f->H < 65536 && f->n >= 1 && f->n <= (1 << 20), -2, "%s: bad shape", who);
    STG_CHECK(f->ld % 8 == 0 && f->ldo % 8 == 0, -2, "%s: leading dimensions must be multiples of 8", who);
    STG_CHECK((((uintptr_t)f->Q | (uintptr_t)f->K | (uintptr_t)f->V | (uintptr_t)f->O) & 15) == 0, -2, "%s: misaligned pointers", who);
    p.Q = (const bf16_t*)f->Q; p.K = (const bf16_t*)f->K; p.V = (const bf16_t*)f->V; p.ld = f->ld;
    p.O = (bf16_t*)f->O; p.ldo = f->ldo; p.lse = f->lse;
    p.P = (int)f->P; p.H = f->H; p.n = f->n; p.nt = (f->n + 31) / 32;
    p.scale = f->scale; p.scale2 = f->scale * LOG2E;
    if (f->win_size != 0) {
        STG_CHECK(f->win_size > 0 && f->win_h > 0 && f->win_w > 0 && f->win_h % f->win_size == 0 && f->win_w % f->win_size == 0 &&
                  f->n == f->win_size * f->win_size && f->win_shift >= 0 && f->win_shift < f->win_size, -2, "%s: bad window map", who);
        p.wh = f->win_h; p.ww = f->win_w; p.ws = f->win_size; p.wshift = f->win_shift;
        p.nwin = (f->win_h / f->win_size) * (f->win_w / f->win_size);
        STG_CHECK(f->P % p.nwin == 0, -2, "%s: P must be a whole number of frames (%d windows each)", who, p.nwin);
    }
    return 0;
}

}  // namespace

extern "C" int stg_mha_supported(int n, int D) { return (D == 64 || D == 96) && n >= 1 ? 1 : 0; }

static int mha_pair_ok(const stg_mha_args* f0, const stg_mha_args* f1, const char* who) {
    STG_CHECK(f0->P == f1->P && f0->H == f1->H && f0->n == f1->n && f0->D == f1->D && f0->scale == f1->scale && f0->ld == f1->ld && f0->ldo == f1->ldo &&
              f0->win_h == f1->win_h && f0->win_w == f1->win_w && f0->win_size == f1->win_size && f0->win_shift == f1->win_shift &&
              (f0->K == f0->V) == (f1->K == f1->V), -2, "%s: the two problems of a pair need one geometry", who);
    return 0;
}

static int mha_fwd_impl(const stg_mha_args* f, const stg_mha_args* f1, void* stream) {
    MP p = {};
    int rc = fill(f, p, "stg_mha_fwd");
    if (rc) return rc;
    if (p.P == 0) return 0;
    if (f1) {
        MP q = {};
        rc = fill(f1, q, "stg_mha_fwd");
        if (rc) return rc;
        p.Q1 = q.Q; p.K1 = q.K; p.V1 = q.V; p.O1 = q.O; p.lse1 = q.lse;
    }
    const bool kv1 = f->K == f->V;
    const int nw = pick_nw(p.nt, false, kv1);
    const dim3 grid((p.nt + nw - 1) / nw, p.H, f1 ? 2 * p.P : p.P);
    hipStream_t st = (hipStream_t)stream;
#define STG_MHA_FWD(DD, KV, NW) { rc = launch_fwd2<DD, KV, NW>(grid, p, st); if (rc) return rc; }
#define STG_MHA_FWD2(DD, KV) { if (nw == 8) STG_MHA_FWD(DD, KV, 8) else STG_MHA_FWD(DD, KV, 4) }
    if (nw == 2) { if (f->D == 64) STG_MHA_FWD(64, true, 2) else STG_MHA_FWD(96, true, 2) }
    else if (f->D == 64) { if (kv1) STG_MHA_FWD2(64, true) else STG_MHA_FWD2(64, false) }
    else { if (kv1) STG_MHA_FWD2(96, true) else STG_MHA_FWD2(96, false) }
#undef STG_MHA_FWD2
#undef STG_MHA_FWD
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_mha_fwd(const stg_mha_args* f, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_mha_fwd: null args");
    return mha_fwd_impl(f, nullptr, stream);
}

// Both directions of a cross-modal pair (same geometry) in ONE launch, grid.z = 2 P: half the launches, and the rounds of a launch fill better
// (320 frames on 256 CUs are 1.25 rounds per direction, 2.5 for the pair).  Falls back to two launches where 2 P exceeds the grid's z limit.
extern "C" int stg_mha_fwd_pair(const stg_mha_args* f0, const stg_mha_args* f1, void* stream) {
    STG_CHECK(f0 != nullptr && f1 != nullptr, -1, "stg_mha_fwd_pair: null args");
    const int rc = mha_pair_ok(f0, f1, "stg_mha_fwd_pair");
    if (rc) return rc;
    if (2 * f0->P >= 65536) { const int r0 = mha_fwd_impl(f0, nullptr, stream); return r0 ? r0 : mha_fwd_impl(f1, nullptr, stream); }
    return mha_fwd_impl(f0, f1, stream);
}

static int mha_bwd_impl(const stg_mha_args* f, const void* dO, void* dQ, void* dK, void* dV, float* delta, const stg_mha_args* f1, const void* dO1,
                        void* dQ1, void* dK1, void* dV1, float* delta1, int64_t lddo, int64_t lddqkv, void* stream) {
    MP p = {};
    int rc = fill(f, p, "stg_mha_bwd");
    if (rc) return rc;
    STG_CHECK(dO && dQ && dK && delta, -1, "stg_mha_bwd: null pointer");
    STG_CHECK(dV != nullptr || f->K == f->V, -2, "stg_mha_bwd: dV == NULL (shared K = V gradient) needs K == V");
    STG_CHECK(lddo % 8 == 0 && lddqkv % 8 == 0, -2, "stg_mha_bwd: bad leading dims");
    STG_CHECK((((uintptr_t)dO) & 15) == 0 && (((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15) == 0, -2,
              "stg_mha_bwd: misaligned pointers");
    if (p.P == 0) return 0;
    p.dO = (const bf16_t*)dO; p.lddo = lddo; p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV; p.lddqkv = lddqkv;
    p.delta = delta;
    if (f1) {
        MP q = {};
        rc = fill(f1, q, "stg_mha_bwd");
        if (rc) return rc;
        STG_CHECK(dO1 && dQ1 && dK1 && delta1 && ((dV1 == nullptr) == (dV == nullptr)), -1, "stg_mha_bwd_pair: null pointer / dV given for one problem only");
        STG_CHECK((((uintptr_t)dO1) & 15) == 0 && (((uintptr_t)dQ1 | (uintptr_t)dK1 | (uintptr_t)dV1) & 15) == 0, -2, "stg_mha_bwd_pair: misaligned pointers");
        p.Q1 = q.Q; p.K1 = q.K; p.V1 = q.V; p.O1 = q.O; p.lse1 = q.lse; p.delta1 = delta1;
        p.dO1 = (const bf16_t*)dO1; p.dQ1 = (bf16_t*)dQ1; p.dK1 = (bf16_t*)dK1; p.dV1 = (bf16_t*)dV1;
    }
    const bool kv1 = f->K == f->V;
    const int nw = pick_nw(p.nt, true, kv1);
    const dim3 grid((p.nt + nw - 1) / nw, p.H, f1 ? 2 * p.P : p.P);
    hipStream_t st = (hipStream_t)stream;
    // dK / dV: two query tiles per trip where its 160 KiB of LDS and 256 registers allow (not at D = 96 with separate K / V and 4 waves, nor with 2 waves)
#define STG_MHA_BWD(DD, KV, NW) { rc = launch_dq2<DD, KV, NW>(grid, p, st); if (rc) return rc; STG_LAUNCH_CHECK(); \
                                  rc = (p.nt >= 2 && dkv2_lds_bytes(DD, KV, NW) <= 160 * 1024 && !(DD == 96 && !KV && NW == 4) && !(DD == 96 && NW == 2)) ? launch_dkv2<DD, KV, NW>(grid, p, st) : launch_dkv<DD, KV, NW>(grid, p, st); }
#define STG_MHA_BWD2(DD, KV) { if (nw == 8) STG_MHA_BWD(DD, KV, 8) else STG_MHA_BWD(DD, KV, 4) }
    if (nw == 2) { if (f->D == 64) STG_MHA_BWD(64, true, 2) else STG_MHA_BWD(96, true, 2) }
    else if (f->D == 64) { if (kv1) STG_MHA_BWD2(64, true) else STG_MHA_BWD2(64, false) }
    else { if (kv1) STG_MHA_BWD2(96, true) else STG_MHA_BWD2(96, false) }
#undef STG_MHA_BWD2
#undef STG_MHA_BWD
    if (rc) return rc;
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_mha_bwd(const stg_mha_args* f, const void* dO, int64_t lddo, void* dQ, void* dK, void* dV, int64_t lddqkv,
                           float* delta, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_mha_bwd: null args");
    return mha_bwd_impl(f, dO, dQ, dK, dV, delta, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lddo, lddqkv, stream);
}

extern "C" int stg_mha_bwd_pair(const stg_mha_args* f0, const void* dO0, void* dQ0, void* dK0, void* dV0, float* delta0, const stg_mha_args* f1,
                                const void* dO1, void* dQ1, void* dK1, void* dV1, float* delta1, int64_t lddo, int64_t lddqkv, void* stream) {
    STG_CHECK(f0 != nullptr && f1 != nullptr, -1, "stg_mha_bwd_pair: null args");
    const int rc = mha_pair_ok(f0, f1, "stg_mha_bwd_pair");
    if (rc) return rc;
    if (2 * f0->P >= 65536) {
        const int r0 = mha_bwd_impl(f0, dO0, dQ0, dK0, dV0, delta0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lddo, lddqkv, stream);
        return r0 ? r0 : mha_bwd_impl(f1, dO1, dQ1, dK1, dV1, delta1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lddo, lddqkv, stream);
    }
    return mha_bwd_impl(f0, dO0, dQ0, dK0, dV0, delta0, f1, dO1, dQ1, dK1, dV1, delta1, lddo, lddqkv, stream);
}

// The merged backward of a cross-modal pair (see mha_bwdm_kernel): direction 0 = (Q = X, K = V = Y), direction 1 = (Q = Y, K = V = X) -- the SAME two
// tensors with their roles swapped (checked).  G0 / G1: the whole gradient of X / of Y (dQ of its own direction + dK + dV of the other).
extern "C" int stg_mha_bwd_pair_merged(const stg_mha_args* f0, const void* dO0, void* G0, float* delta0, const stg_mha_args* f1, const void* dO1,
                                       void* G1, float* delta1, int64_t lddo, int64_t lddg, void* stream) {
    STG_CHECK(f0 && f1 && dO0 && dO1 && G0 && G1 && delta0 && delta1, -1, "stg_mha_bwd_pair_merged: null pointer");
    int rc = mha_pair_ok(f0, f1, "stg_mha_bwd_pair_merged");
    if (rc) return rc;
    STG_CHECK(f0->K == f0->V && f1->K == f1->V && f0->Q == f1->K && f1->Q == f0->K, -7,
              "stg_mha_bwd_pair_merged: needs a cross pair (K == V, and each direction's keys are the other's queries)");
    STG_CHECK(2 * f0->P < 65536, -7, "stg_mha_bwd_pair_merged: too many problems for one launch");
    STG_CHECK(lddo % 8 == 0 && lddg % 8 == 0 && ((((uintptr_t)dO0 | (uintptr_t)dO1 | (uintptr_t)G0 | (uintptr_t)G1)) & 15) == 0, -2, "stg_mha_bwd_pair_merged: misaligned operands");
    MP p = {}, q = {};
    rc = fill(f0, p, "stg_mha_bwd_pair_merged");
    if (rc) return rc;
    rc = fill(f1, q, "stg_mha_bwd_pair_merged");
    if (rc) return rc;
    if (p.P == 0) return 0;
    p.dO = (const bf16_t*)dO0; p.lddo = lddo; p.dQ = (bf16_t*)G0; p.lddqkv = lddg; p.delta = delta0;
    p.Q1 = q.Q; p.K1 = q.K; p.V1 = q.V; p.O1 = q.O; p.lse1 = q.lse; p.delta1 = delta1;
    p.dO1 = (const bf16_t*)dO1; p.dQ1 = (bf16_t*)G1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t items = (int64_t)p.P * p.H * p.n;
    const dim3 gd((unsigned)((items * 8 + 255) / 256), 2);
    if (f0->D == 64) hipLaunchKernelGGL(mha_delta_kernel<64>, gd, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(mha_delta_kernel<96>, gd, dim3(256), 0, st, p);
    STG_LAUNCH_CHECK();
    const int nw = p.nt > 4 ? 8 : 4;
    const dim3 grid((p.nt + nw - 1) / nw, p.H, 2 * p.P);
#define STG_BWDM(DD) { if (nw == 8) rc = launch_bwdm<DD, 8>(grid, p, st); else rc = launch_bwdm<DD, 4>(grid, p, st); }
    if (f0->D == 64) STG_BWDM(64) else STG_BWDM(96)
#undef STG_BWDM
    if (rc) return rc;
    STG_LAUNCH_CHECK();
    return 0;
}
