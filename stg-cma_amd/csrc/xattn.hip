// Long-sequence single-head attention with K == V: the adapters' frame-global cross-modal attention
//   h_v' = softmax(h_v h_a^T) h_a   and   h_a' = softmax(h_a h_v^T) h_v        (Swin_AVE.py:799-808, CLIP_AVE.py:386-398)
// with up to 3136 queries x 3136 keys per frame and head dim 16 / 32 (the adapter bottleneck width of stages 1 / 2).
//
// At these widths the work is neither MFMA- nor HBM-bound: a 32 x 32 score tile costs 3 MFMAs but 1024 exponentials, so
// the kernels are built around the VALU instruction count per score (the generic gather-mapped kernels of attention.hip
// spend ~45 issue slots per score on addressing, bounds and bias/mask plumbing; these spend ~4):
//   * scores stay transposed (key on the accumulator row, query on the lane): max / LSE / delta are per-lane scalars;
//   * softmax runs on exp2 with one FMA per score (scale * log2 e folded), the running max is only raised when a tile
//     exceeds it by more than 2^8 (lazy rescale: the common tile does no accumulator rescaling at all);
//   * the row sum comes out of the MFMA: for D = 16 the unused A-operand rows 16..31 of the P.V product are fed ones, so
//     accumulator row 16 IS sum_k p (for D = 32 one extra MFMA with a ones operand does the same);
//   * K == V, so the K fragment a lane loads for the score MFMA (16 bytes of one key row) is also its piece of the V
//     tile: it is written once to a wave-private LDS tile and read back transposed with ds_read_b64_tr_b16 (the k-strided
//     V^T / K^T / Q^T / dO^T operands), 2 reads per fragment instead of 8 16-bit reads + packing;
//   * two 32-key tiles per loop trip, next trip's fragments prefetched into registers; tails are handled in a separate,
//     masked instantiation of the loop body so the steady state carries no bounds logic;
//   * (round 3) the per-score arithmetic around the exponential rides the MATRIX pipe, which has slack here: the wave's constant
//     operand (its queries, or its keys in the dK + dV kernel) is pre-multiplied by scale * log2 e and split into a bf16 hi + lo
//     pair (two score MFMAs instead of one: the product keeps ~16 mantissa bits, the fp32 multiply it replaces had 24 against bf16
//     operands), and -max / -lse / -delta enter as the MFMA's C operand, so a score leaves the MFMA as the exp2 argument and dP
//     leaves it as dP - delta: backward exp + mul + cvt per score (was fma + exp + sub + mul + cvt).  Applied to the two backward
//     kernels (dQ -9 %, dK + dV -4 %); in the forward kernel the 16 extra live registers of the C operand cost more than the FMA saved
//     (+7 %), so it keeps the fp32 FMA.
// One wave owns 32 queries (fwd, dQ) or 32 keys (dK+dV); waves never synchronise with each other.
//
// MFMA v_mfma_f32_32x32x16_bf16, lane l = (r = l & 31, hh = l >> 5): A[row r][k = 8 hh + j], B[k = 8 hh + j][col r];
// C/D col = r, row = (reg & 3) + 8 (reg >> 2) + 4 hh.  An accumulator used as the next B operand has k slot (hh, j) of
// step s2 = row 16 s2 + 8 (j >> 2) + 4 hh + (j & 3); the transposed reads deliver the other operand in that same order.
#include <math.h>
#include <type_traits>
#include "common.h"
#include "../../include/stgcma.h"
#include "xattn.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float RESCALE_THRESHOLD = 8.0f;       // in log2 units: probabilities stay below 2^8 between rescales

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
typedef short s4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ bf16x8_t pack_frag(const float* x) {
    const u32x4_t w = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
    return __builtin_bit_cast(bf16x8_t, w);
}
__device__ __forceinline__ f32x16_t splat16(float v) {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = v;
    return z;
}
// c * f as a bf16 (hi, lo) fragment pair: hi = rn(c f), lo = rn(c f - hi)
__device__ __forceinline__ void scale_split(const bf16x8_t f, float c, bf16x8_t& hi, bf16x8_t& lo) {
    float v[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf2f((bf16_t)f[j]) * c;
    hi = pack_frag(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = v[j] - bf2f((bf16_t)hi[j]);
    lo = pack_frag(l);
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// transposed fragment of a row-major [32 rows][D] LDS tile: element j of lane (r, hh) = tile[16 s2 + 8 (j >> 2) + 4 hh + (j & 3)][r & 15 (+16 if D = 32 and r >= 16)]
// `base` = this lane's address for (s2 = 0, t = 0): tile + (4 hh + q) * D + col0 + 4 p, lane-in-group i = 4 q + p
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* base, int s2, int D) {
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)(base + (16 * s2) * D));
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)(base + (16 * s2 + 8) * D));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

struct XP {
    const bf16_t* Q; int64_t ldq;
    const bf16_t* KV; int64_t ldk;
    bf16_t* O; int64_t ldo;
    float* lse;
    int64_t outer_q, outer_kv;
    int P, n, n_kv;
    float scale, c2;
    const bf16_t* dO; int64_t lddo;
    bf16_t* dQ; int64_t lddq;
    bf16_t* dKV; int64_t lddk;
    float* delta;
    int tiles, total;
    const float* gate; bf16_t* X; int64_t ldx;      // forward, optional (round 5): X = Q + gate[0] * O written beside O (the pair's gate_fwd)
};

// Key rows inside a problem (forward, dQ): a wave-uniform 64-bit base + a 32-bit byte offset per lane that ADVANCES by a uniform step per
// trip (global_load ... v_off, s[base]) instead of a 64-bit (row index) x (leading dimension) product per trip (four v_mul_lo_u32 + two
// v_mad_u64_u32 in the ISA): forward 955 -> 899 us, dQ 1094 -> 1076 at stage 0.  (The dK / dV kernel did not gain from it -- its d_h = 32
// form lost 20 % -- and keeps the products.)  Only a tail tile clamps its rows (wave-uniform test); stg_xattn_eligible keeps a problem's
// key rows inside 2 GiB.
__device__ __forceinline__ bf16x8_t ld_frag_b(const char* base, uint32_t off) { return *reinterpret_cast<const bf16x8_t*>(base + off); }

struct XP2 { XP a[2]; };          // the two directions of a cross-modal pair in one launch (blockIdx.y); a single call fills a[0] only

// per-wave LDS: two [32][D] tiles + 8 bytes of bf16 ones
template <int D> struct Lds { static constexpr int TILE = 32 * D; static constexpr int PER_WAVE = 2 * TILE + 8; };

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ void __launch_bounds__(256, 2) xattn_fwd_kernel(XP2 pp) {
    const XP a = pp.a[blockIdx.y];
    constexpr int KS = D / 16;
    constexpr int TILE = Lds<D>::TILE;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * Lds<D>::PER_WAVE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.total) return;
    const int p = item / a.tiles, qt = item - p * a.tiles;
    bf16_t* tA = smem + wave * Lds<D>::PER_WAVE;
    bf16_t* tB = tA + TILE;
    bf16_t* ones = tB + TILE;
    if (lane < 4) ones[lane] = (bf16_t)0x3F80;

    int q = qt * 32 + r;
    const bool okq = q < a.n;
    q = okq ? q : a.n - 1;
    const int64_t rowq = (int64_t)p * a.outer_q + q;
    bf16x8_t qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = ld_frag(a.Q + rowq * a.ldq + 8 * hh + 16 * s);

    // this lane's pieces: LDS write slot of its K fragment(s), and its transposed-read base
    const char* kvb = reinterpret_cast<const char*>(a.KV + (int64_t)p * a.outer_kv * a.ldk);   // wave-uniform
    const int wr_off = r * D + 8 * hh;                           // + 16 s
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, c = r >> 4;
    // D = 16: lanes r >= 16 (A rows 16..31) read ones -> accumulator rows 16.. hold the row sum
    const int tr_off = (D == 16) ? (4 * hh + gq) * D + 4 * gp : (4 * hh + gq) * D + 16 * c + 4 * gp;
    const bool ones_lane = (D == 16) && c == 1;
    const bf16_t* trA = ones_lane ? ones : tA + tr_off;
    const bf16_t* trB = ones_lane ? ones : tB + tr_off;
    const int tr_stride = ones_lane ? 0 : D;                    // rows advance only for real tiles

    f32x16_t o = zero16(), lacc = zero16();
    float m2 = 0.f;                                             // running max (log2 domain); the first trip adopts its tile's maximum
    bf16x8_t ones_frag;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones_frag[j] = (short)0x3F80;

    bf16x8_t kfA[KS], kfB[KS];
    uint32_t koff = ((uint32_t)r * (uint32_t)a.ldk + 8 * hh) * 2u;         // this lane's piece of key (next tile) + r
    const uint32_t khalf = (uint32_t)a.ldk * 64u, klast = ((uint32_t)(a.n_kv - 1) * (uint32_t)a.ldk + 8 * hh) * 2u;
    auto load_pair = [&](int k0) {                              // tiles are loaded in order: koff tracks k0; every load in bounds, tails masked by value
        uint32_t oa = koff, ob = koff + khalf;
        if (k0 + 64 > a.n_kv) {                                 // wave-uniform: only a tail tile clamps
            oa = k0 + r < a.n_kv ? oa : klast;
            ob = k0 + 32 + r < a.n_kv ? ob : klast;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kfA[s] = ld_frag_b(kvb, oa + 32 * s);
            kfB[s] = ld_frag_b(kvb, ob + 32 * s);
        }
        koff += 2 * khalf;
    };
    load_pair(0);

    auto body = [&](int k0, auto masked_tag, auto first_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        lds_sync();                                             // previous trip's transposed reads are done
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8_t*>(tA + wr_off + 16 * s) = kfA[s];
            *reinterpret_cast<bf16x8_t*>(tB + wr_off + 16 * s) = kfB[s];
        }
        f32x16_t sA = zero16(), sB = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            sA = MFMA32(kfA[s], qf[s], sA);
            sB = MFMA32(kfB[s], qf[s], sB);
        }
        if (k0 + 64 < a.n_kv) load_pair(k0 + 64);               // wave-uniform
        if (MASKED) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int key = k0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                sA[reg] = key < a.n_kv ? sA[reg] : -INFINITY;
                sB[reg] = key + 32 < a.n_kv ? sB[reg] : -INFINITY;
            }
        }
        // x = c2 s - m2 FIRST (packed FMA: two scores per issue slot), the maximum over x afterwards: taken on the raw MFMA outputs
        // every fmaxf is preceded by a canonicalising v_max_f32 (hipcc cannot know an MFMA result is not a signalling NaN) -- 50
        // v_max per trip instead of 16 v_max3
        float xA[16], xB[16];
#pragma unroll
        for (int reg = 0; reg < 16; reg += 2) {
            const f32x2_t xa = pk_fma((f32x2_t){sA[reg], sA[reg + 1]}, pk_splat(a.c2), pk_splat(-m2));
            const f32x2_t xb = pk_fma((f32x2_t){sB[reg], sB[reg + 1]}, pk_splat(a.c2), pk_splat(-m2));
            xA[reg] = xa.x; xA[reg + 1] = xa.y; xB[reg] = xb.x; xB[reg + 1] = xb.y;
        }
        float mt = fmaxf(xA[0], xB[0]);                         // the tile's maximum RELATIVE to the running one
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) mt = fmaxf(mt, fmaxf(xA[reg], xB[reg]));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        if (FIRST || __builtin_amdgcn_ballot_w64(mt > RESCALE_THRESHOLD) != 0) {
            const float shift = FIRST ? mt : fmaxf(mt, 0.f);     // lazy rescale; the first trip adopts the tile's maximum (o, l are zero)
            if (!FIRST) {
                const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) o[reg] *= alpha;
                if (D == 32) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) lacc[reg] *= alpha;
                }
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) { xA[reg] -= shift; xB[reg] -= shift; }
            m2 += shift;
        }
        float pA[16], pB[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { pA[reg] = __builtin_amdgcn_exp2f(xA[reg]); pB[reg] = __builtin_amdgcn_exp2f(xB[reg]); }
        lds_sync();                                             // tile writes visible to the transposed reads
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8_t fa = pack_frag(pA + 8 * s2), fb = pack_frag(pB + 8 * s2);
            o = MFMA32(tr_frag(trA, s2, tr_stride), fa, o);
            o = MFMA32(tr_frag(trB, s2, tr_stride), fb, o);
            if (D == 32) {
                lacc = MFMA32(ones_frag, fa, lacc);
                lacc = MFMA32(ones_frag, fb, lacc);
            }
        }
    };

    const int full = a.n_kv & ~63;                              // n_kv >= 64 (stg_xattn_eligible): the first trip is a full one
    body(0, std::false_type{}, std::true_type{});
    int k0 = 64;
    for (; k0 < full; k0 += 64) body(k0, std::false_type{}, std::false_type{});
    if (k0 < a.n_kv) body(k0, std::true_type{}, std::false_type{});

    if (okq) {
        const float l = (D == 16) ? o[8] : lacc[0];
        const float inv = 1.0f / l;
        bf16_t* op = a.O + rowq * a.ldo;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            uint2 w;
            w.x = pack_bf2(o[4 * g + 0] * inv, o[4 * g + 1] * inv);
            w.y = pack_bf2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * g + 4 * hh) = w;
            if (a.X) {                                      // kernel-uniform: the gate on the bf16-ROUNDED output, gate_fwd's arithmetic
                const uint2 qv = *reinterpret_cast<const uint2*>(a.Q + rowq * a.ldq + 8 * g + 4 * hh);
                const float gv = a.gate[0];
                uint2 xw;
                xw.x = pack_bf2(fmaf(gv, __uint_as_float(w.x << 16), __uint_as_float(qv.x << 16)), fmaf(gv, __uint_as_float(w.x & 0xffff0000u), __uint_as_float(qv.x & 0xffff0000u)));
                xw.y = pack_bf2(fmaf(gv, __uint_as_float(w.y << 16), __uint_as_float(qv.y << 16)), fmaf(gv, __uint_as_float(w.y & 0xffff0000u), __uint_as_float(qv.y & 0xffff0000u)));
                *reinterpret_cast<uint2*>(a.X + rowq * a.ldx + 8 * g + 4 * hh) = xw;
            }
        }
        if (hh == 0) a.lse[(int64_t)p * a.n + q] = (m2 + __log2f(l)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <int D>
__global__ void __launch_bounds__(256, 2) xattn_dq_kernel(XP2 pp) {
    const XP a = pp.a[blockIdx.y];
    constexpr int KS = D / 16;
    constexpr int TILE = Lds<D>::TILE;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * Lds<D>::PER_WAVE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.total) return;
    const int p = item / a.tiles, qt = item - p * a.tiles;
    bf16_t* tA = smem + wave * Lds<D>::PER_WAVE;
    bf16_t* tB = tA + TILE;

    int q = qt * 32 + r;
    const bool okq = q < a.n;
    q = okq ? q : a.n - 1;
    const int64_t rowq = (int64_t)p * a.outer_q + q;
    bf16x8_t qf[KS], dof[KS];
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = ld_frag(a.Q + rowq * a.ldq + 8 * hh + 16 * s);
        dof[s] = ld_frag(a.dO + rowq * a.lddo + 8 * hh + 16 * s);
        const bf16x8_t of = ld_frag(a.O + rowq * a.ldo + 8 * hh + 16 * s);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += bf2f((bf16_t)dof[s][j]) * bf2f((bf16_t)of[j]);
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lse2 = a.lse[(int64_t)p * a.n + q] * LOG2E;
    if (okq && hh == 0) a.delta[(int64_t)p * a.n + q] = delta;

    const char* kvb = reinterpret_cast<const char*>(a.KV + (int64_t)p * a.outer_kv * a.ldk);   // wave-uniform
    const int wr_off = r * D + 8 * hh;
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, c = r >> 4;
    const int tr_off = (D == 16) ? (4 * hh + gq) * D + 4 * gp : (4 * hh + gq) * D + 16 * c + 4 * gp;   // D = 16: rows 16.. duplicate 0..15 (ignored)
    const bf16_t* trA = tA + tr_off;
    const bf16_t* trB = tB + tr_off;

    bf16x8_t qh[KS], ql[KS];                                    // scale * log2 e folded into the queries (hi + lo)
#pragma unroll
    for (int s = 0; s < KS; ++s) scale_split(qf[s], a.c2, qh[s], ql[s]);
    const f32x16_t cl = splat16(-lse2), cd = splat16(-delta);   // C operands: scores leave the MFMA as c2 s - lse2, dP as dP - delta

    f32x16_t dq = zero16();
    bf16x8_t kfA[KS], kfB[KS];
    uint32_t koff = ((uint32_t)r * (uint32_t)a.ldk + 8 * hh) * 2u;         // this lane's piece of key (next tile) + r
    const uint32_t khalf = (uint32_t)a.ldk * 64u, klast = ((uint32_t)(a.n_kv - 1) * (uint32_t)a.ldk + 8 * hh) * 2u;
    auto load_pair = [&](int k0) {
        uint32_t oa = koff, ob = koff + khalf;
        if (k0 + 64 > a.n_kv) {                                 // wave-uniform: only a tail tile clamps
            oa = k0 + r < a.n_kv ? oa : klast;
            ob = k0 + 32 + r < a.n_kv ? ob : klast;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kfA[s] = ld_frag_b(kvb, oa + 32 * s);
            kfB[s] = ld_frag_b(kvb, ob + 32 * s);
        }
        koff += 2 * khalf;
    };
    load_pair(0);

    auto body = [&](int k0, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        lds_sync();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8_t*>(tA + wr_off + 16 * s) = kfA[s];
            *reinterpret_cast<bf16x8_t*>(tB + wr_off + 16 * s) = kfB[s];
        }
        f32x16_t sA = MFMA32(kfA[0], qh[0], cl), sB = MFMA32(kfB[0], qh[0], cl);
        sA = MFMA32(kfA[0], ql[0], sA); sB = MFMA32(kfB[0], ql[0], sB);
        f32x16_t dpA = MFMA32(kfA[0], dof[0], cd), dpB = MFMA32(kfB[0], dof[0], cd);       // V == K
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            sA = MFMA32(kfA[s], qh[s], sA); sB = MFMA32(kfB[s], qh[s], sB);
            sA = MFMA32(kfA[s], ql[s], sA); sB = MFMA32(kfB[s], ql[s], sB);
            dpA = MFMA32(kfA[s], dof[s], dpA); dpB = MFMA32(kfB[s], dof[s], dpB);
        }
        if (k0 + 64 < a.n_kv) load_pair(k0 + 64);
        float dA[16], dB[16];
#pragma unroll
        for (int reg = 0; reg < 16; reg += 2) {                 // exp2 + one packed multiply per pair of scores
            f32x2_t xa = {sA[reg], sA[reg + 1]}, xb = {sB[reg], sB[reg + 1]};
            if (MASKED) {
                const int key = k0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                xa.x = key < a.n_kv ? xa.x : -INFINITY;          xa.y = key + 1 < a.n_kv ? xa.y : -INFINITY;
                xb.x = key + 32 < a.n_kv ? xb.x : -INFINITY;     xb.y = key + 33 < a.n_kv ? xb.y : -INFINITY;
            }
            const f32x2_t ea = {__builtin_amdgcn_exp2f(xa.x), __builtin_amdgcn_exp2f(xa.y)};
            const f32x2_t eb = {__builtin_amdgcn_exp2f(xb.x), __builtin_amdgcn_exp2f(xb.y)};
            const f32x2_t ra = ea * (f32x2_t){dpA[reg], dpA[reg + 1]};
            const f32x2_t rb = eb * (f32x2_t){dpB[reg], dpB[reg + 1]};
            dA[reg] = ra.x; dA[reg + 1] = ra.y; dB[reg] = rb.x; dB[reg + 1] = rb.y;
        }
        lds_sync();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dq = MFMA32(tr_frag(trA, s2, D), pack_frag(dA + 8 * s2), dq);
            dq = MFMA32(tr_frag(trB, s2, D), pack_frag(dB + 8 * s2), dq);
        }
    };
    const int full = a.n_kv & ~63;
    int k0 = 0;
    for (; k0 < full; k0 += 64) body(k0, std::false_type{});
    if (k0 < a.n_kv) body(k0, std::true_type{});

    if (okq) {
        bf16_t* op = a.dQ + rowq * a.lddq;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            uint2 w;
            w.x = pack_bf2(dq[4 * g + 0] * a.scale, dq[4 * g + 1] * a.scale);
            w.y = pack_bf2(dq[4 * g + 2] * a.scale, dq[4 * g + 3] * a.scale);
            *reinterpret_cast<uint2*>(op + 8 * g + 4 * hh) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK + dV (K == V)
template <int D>
__global__ void __launch_bounds__(256, 2) xattn_dkv_kernel(XP2 pp) {
    const XP a = pp.a[blockIdx.y];
    constexpr int KS = D / 16;
    constexpr int TILE = Lds<D>::TILE;
    // per wave: Q tile + dO tile (one 32-query tile per trip) + lse2[32] + delta[32]
    constexpr int PER_WAVE = 2 * TILE + 128;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * PER_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.total) return;
    const int p = item / a.tiles, kt = item - p * a.tiles;
    bf16_t* tQ = smem + wave * PER_WAVE;
    bf16_t* tD = tQ + TILE;
    float* sStat = reinterpret_cast<float*>(tD + TILE);         // [0..31] lse2, [32..63] delta

    int key = kt * 32 + r;
    const bool okk = key < a.n_kv;
    key = okk ? key : a.n_kv - 1;
    const int64_t rowk = (int64_t)p * a.outer_kv + key;
    bf16x8_t kf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kf[s] = ld_frag(a.KV + rowk * a.ldk + 8 * hh + 16 * s);

    const bf16_t* qp = a.Q + (int64_t)p * a.outer_q * a.ldq + 8 * hh;
    const bf16_t* dop = a.dO + (int64_t)p * a.outer_q * a.lddo + 8 * hh;
    const float* statp = (hh == 0 ? a.lse : a.delta) + (int64_t)p * a.n;
    const int wr_off = r * D + 8 * hh;
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, c = r >> 4;
    const int tr_off = (D == 16) ? (4 * hh + gq) * D + 4 * gp : (4 * hh + gq) * D + 16 * c + 4 * gp;
    const bf16_t* trQ = tQ + tr_off;
    const bf16_t* trD = tD + tr_off;

    bf16x8_t kh[KS], kl[KS];                                    // scale * log2 e folded into the wave's keys (hi + lo); V stays kf
#pragma unroll
    for (int s = 0; s < KS; ++s) scale_split(kf[s], a.c2, kh[s], kl[s]);

    f32x16_t dk = zero16(), dv = zero16();
    bf16x8_t qf[KS], dof[KS];
    float stat;
    auto load_tile = [&](int q0) {
        int qq = q0 + r;
        const bool ok = qq < a.n;
        qq = ok ? qq : a.n - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = ld_frag(qp + (int64_t)qq * a.ldq + 16 * s);
            dof[s] = ld_frag(dop + (int64_t)qq * a.lddo + 16 * s);
        }
        const float v = statp[qq];
        // NEGATED (they are the MFMAs' C operands); padded queries: -lse2 = -inf -> p = 0, delta = 0
        stat = hh == 0 ? (ok ? -v * LOG2E : -INFINITY) : (ok ? -v : 0.f);
    };
    load_tile(0);

    for (int q0 = 0; q0 < a.n; q0 += 32) {
        lds_sync();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8_t*>(tQ + wr_off + 16 * s) = qf[s];
            *reinterpret_cast<bf16x8_t*>(tD + wr_off + 16 * s) = dof[s];
        }
        sStat[lane] = stat;
        lds_sync();
        f32x16_t cl, cd;                                        // -lse2 / -delta of the query on each accumulator row
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4*>(sStat + 8 * g + 4 * hh);
            const float4 d4 = *reinterpret_cast<const float4*>(sStat + 32 + 8 * g + 4 * hh);
            cl[4 * g] = l4.x; cl[4 * g + 1] = l4.y; cl[4 * g + 2] = l4.z; cl[4 * g + 3] = l4.w;
            cd[4 * g] = d4.x; cd[4 * g + 1] = d4.y; cd[4 * g + 2] = d4.z; cd[4 * g + 3] = d4.w;
        }
        f32x16_t sc = MFMA32(qf[0], kh[0], cl);                 // c2 S[q][key] - lse2[q]: query on the accumulator row, key on the lane
        sc = MFMA32(qf[0], kl[0], sc);
        f32x16_t dp = MFMA32(dof[0], kf[0], cd);                // dP[q][key] - delta[q],  V == K
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            sc = MFMA32(qf[s], kh[s], sc);
            sc = MFMA32(qf[s], kl[s], sc);
            dp = MFMA32(dof[s], kf[s], dp);
        }
        if (q0 + 32 < a.n) load_tile(q0 + 32);
        float pr[16], ds[16];
#pragma unroll
        for (int reg = 0; reg < 16; reg += 2) {                 // exp2 + one packed multiply per pair of scores
            const f32x2_t pv = {__builtin_amdgcn_exp2f(sc[reg]), __builtin_amdgcn_exp2f(sc[reg + 1])};
            const f32x2_t dv2 = pv * (f32x2_t){dp[reg], dp[reg + 1]};
            pr[reg] = pv.x; pr[reg + 1] = pv.y; ds[reg] = dv2.x; ds[reg + 1] = dv2.y;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dv = MFMA32(tr_frag(trD, s2, D), pack_frag(pr + 8 * s2), dv);
            dk = MFMA32(tr_frag(trQ, s2, D), pack_frag(ds + 8 * s2), dk);
        }
    }

    if (okk) {
        bf16_t* op = a.dKV + rowk * a.lddk;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            uint2 w;
            w.x = pack_bf2(fmaf(dk[4 * g + 0], a.scale, dv[4 * g + 0]), fmaf(dk[4 * g + 1], a.scale, dv[4 * g + 1]));
            w.y = pack_bf2(fmaf(dk[4 * g + 2], a.scale, dv[4 * g + 2]), fmaf(dk[4 * g + 3], a.scale, dv[4 * g + 3]));
            *reinterpret_cast<uint2*>(op + 8 * g + 4 * hh) = w;
        }
    }
}


// ------------------------------------------------------------------------------------------------ backward of a cross-modal PAIR, merged (round 5)
// The two directions of the adapters' cross-modal attention share ONE score matrix: S[i, j] = h_v[i] . h_a[j], direction v normalises its rows
// (P^v[i, j] = exp(S - lse_v[i])), direction a its columns (P^a[j, i] = exp(S - lse_a[j])).  The four backward passes of rounds 1-4 (dQ and
// dK + dV per direction) each rebuilt S and exponentiated it: four exponentials per score.  Here the gradient of a modality's hidden states is
// ONE pass over the score tiles of its rows -- for the v rows (the a rows mirror it):
//     G_v[i] = scale sum_j dS^v[i, j] a[j]  +  scale sum_j dS^a[j, i] a[j]  +  sum_j P^a[j, i] dO_a[j]
//              (dQ of direction v)            (dK of direction a)             (dV of direction a)
//     dS^v = P^v (dO_v[i] . a[j] - delta_v[i]),   dS^a = P^a (dO_a[j] . v[i] - delta_a[j])
// with ONE exponential per score: E = P^v comes out of the MFMA as exp2's argument (-lse_v[i] is the C operand), and
//     P^a[j, i] = E r[i] c[j],   r[i] = 2^(lse2_v[i] - c0),   c[j] = 2^(c0 - lse2_a[j]),   c0 = the frame's mid-range log-sum-exp.
// c[j] is folded into the OTHER side's operands by a preparation kernel (dOh[j] = c[j] dO_a[j], nd[j] = -c[j] delta_a[j]), r[i] is a per-lane
// scalar: T = E (u + r y) with u = dP^v - delta_v and y = c (dP^a - delta_a) straight from two MFMAs, so per score the wave issues one v_exp, one
// fma, one multiply and two bf16 conversions (E and T) against exp + mul + cvt (dQ) and exp + mul + 2 cvt (dK + dV) per direction before, and
// 8 x KS MFMAs per 32 x 32 tile against 10.  G_v = scale Ga + r Gb with Ga += a^T T and Gb += dOh^T E.
// Range: r, c and r y stay finite while every log-sum-exp of the frame lies within +-60 binary orders of c0; where E underflows (x < -126) the
// true P^a is below 2^(-126 + 120), i.e. nothing.  The preparation kernel measures the spread per frame; a frame beyond 120 binary orders (or
// with a non-finite lse) takes the kernel's SLOW path (wave-uniform): c = r = 1 and a second exponential E2 = exp2(x + lse2_v[i] - lse2_a[j])
// per score, T = E u + E2 y, Gb += dO_a^T E2 -- the arithmetic of the four original passes, still in one.
struct XM {
    const bf16_t* Q; int64_t ldq;                 // own rows (hidden states), own dO
    const bf16_t* dO; int64_t lddo;
    const bf16_t* O; int64_t ldo;                 // own attention output (preparation kernel: delta)
    const float* lse; float* delta;               // own rows
    const bf16_t* KV; int64_t ldk;                // the other modality's rows
    const bf16_t* dOh_other; const float* nd_other;     // c[j] dO[j] (dense [rows, D]) and -c[j] delta[j] of the other modality's rows
    const float* lse_other;                       // slow path only
    bf16_t* dOh_own; float* nd_own;               // what the preparation kernel writes for THIS side's rows (read by the other side's pass)
    bf16_t* G; int64_t ldg;
    int n, n_kv, tiles, total;
    const bf16_t* jX; const bf16_t* jZ; int64_t ldjx, ldjz;     // optional join (round 5): G <- (jX + G) * jZ, the add3_mul that follows (jZ = saved act')
    const float* gate; float* dgate;              // optional gate (round 6b): dO is d(x) of x = q + gate o -- the kernels work on bf16(gate dO), stg_gate_bwd2's
                                                  // rounding, and the preparation kernel adds <d(x), o> to dgate (one atomic per workgroup)
};
struct XM2 { XM a[2]; float* c0; int* ok; float scale, c2; int P; };

template <int D>
__global__ void __launch_bounds__(256) xattn_prep_kernel(XM2 pp) {
    __shared__ float red[2][4];
    const int p = blockIdx.x, part = blockIdx.y, nparts = gridDim.y, tid = threadIdx.x;
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const XM& a = pp.a[y];
        for (int r = tid; r < a.n; r += 256) {
            const float l = a.lse[(int64_t)p * a.n + r] * LOG2E;
            mn = l == l ? fminf(mn, l) : -INFINITY;         // a NaN poisons the range: the frame goes to the fallback
            mx = l == l ? fmaxf(mx, l) : INFINITY;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = mn; red[1][tid >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const bool ok = (mx - mn) <= 120.0f;                    // false for infinities / NaNs too
    const float c0 = ok ? 0.5f * (mx + mn) : 0.f;
    if (part == 0 && tid == 0) { pp.c0[p] = c0; pp.ok[p] = ok ? 1 : 0; }
    float dgs[2] = {0.f, 0.f};
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const XM& a = pp.a[y];
        const bool gated = a.gate != nullptr;               // kernel-uniform
        const float gt = gated ? a.gate[0] : 1.0f;
        const int chunk = (a.n + nparts - 1) / nparts;
        const int r1 = min(a.n, (part + 1) * chunk);
        for (int r = part * chunk + tid; r < r1; r += 256) {
            const int64_t row = (int64_t)p * a.n + r;
            const float c = ok ? __builtin_amdgcn_exp2f(c0 - a.lse[row] * LOG2E) : 1.0f;
            float d = 0.f;
            bf16_t* oh = a.dOh_own + row * D;
#pragma unroll
            for (int s = 0; s < D / 8; ++s) {
                const bf16x8_t df = ld_frag(a.dO + row * a.lddo + 8 * s), of = ld_frag(a.O + row * a.ldo + 8 * s);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = bf2f((bf16_t)df[j]);
                    const float o = bf2f((bf16_t)of[j]);
                    if (gated) { dgs[y] += x * o; x = bf2f(f2bf(gt * x)); }      // stg_gate_bwd2's sum and its bf16 rounding of gate d(x)
                    d += x * o; v[j] = x * c;
                }
                *reinterpret_cast<bf16x8_t*>(oh + 8 * s) = pack_frag(v);
            }
            a.delta[row] = d;
            a.nd_own[row] = -c * d;
        }
    }
    if (pp.a[0].gate != nullptr) {                          // kernel-uniform: one atomic per workgroup and gate
        __shared__ float gred[2][4];
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            float v = dgs[y];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((tid & 63) == 0) gred[y][tid >> 6] = v;
        }
        __syncthreads();
        if (tid < 2) atomicAdd(pp.a[tid].dgate, gred[tid][0] + gred[tid][1] + gred[tid][2] + gred[tid][3]);
    }
}

template <int D>
__global__ void __launch_bounds__(256, 2) xattn_bwdm_kernel(XM2 pp) {
    const XM a = pp.a[blockIdx.y];
    constexpr int KS = D / 16;
    constexpr int TILE = 32 * D;
    constexpr int PER_WAVE = 2 * TILE + 128;                      // other-side rows tile + their prescaled dO tile + 32 floats of -c delta + 32 of lse2
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * PER_WAVE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.total) return;
    const int p = item / a.tiles, qt = item - p * a.tiles;
    const bool fast = pp.ok[p] != 0;                              // wave-uniform
    bf16_t* tK = smem + wave * PER_WAVE;
    bf16_t* tH = tK + TILE;
    float* sNd = reinterpret_cast<float*>(tH + TILE);
    float* sL = sNd + 32;

    int q = qt * 32 + r;
    const bool okq = q < a.n;
    q = okq ? q : a.n - 1;
    const int64_t rowq = (int64_t)p * a.n + q;
    bf16x8_t qf[KS], dof[KS], qh[KS], ql[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = ld_frag(a.Q + rowq * a.ldq + 8 * hh + 16 * s);
        dof[s] = ld_frag(a.dO + rowq * a.lddo + 8 * hh + 16 * s);
        if (a.gate) {                                             // kernel-uniform: bf16(gate d(x)), the operand stg_gate_bwd2 used to write
            const float gt = a.gate[0];
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gt * bf2f((bf16_t)dof[s][j]);
            dof[s] = pack_frag(v);
        }
        scale_split(qf[s], pp.c2, qh[s], ql[s]);
    }
    const float lse2 = a.lse[rowq] * LOG2E;
    const float rsc = fast ? __builtin_amdgcn_exp2f(lse2 - pp.c0[p]) : 1.0f;    // r[i]
    const f32x16_t cl = splat16(-lse2), cd = splat16(-a.delta[rowq]);

    const int64_t kbase = (int64_t)p * a.n_kv;
    const int wr_off = r * D + 8 * hh;
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, c = r >> 4;
    const int tr_off = (D == 16) ? (4 * hh + gq) * D + 4 * gp : (4 * hh + gq) * D + 16 * c + 4 * gp;   // D = 16: rows 16.. duplicate 0..15 (ignored)
    const bf16_t* trK = tK + tr_off;
    const bf16_t* trH = tH + tr_off;

    f32x16_t Ga = zero16(), Gb = zero16();
    bf16x8_t kf[KS], hf[KS];
    float ndv, lov = 0.f;
    auto load_tile = [&](int k0) {
        int k = k0 + r;
        k = k < a.n_kv ? k : a.n_kv - 1;
        const int64_t row = kbase + k;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = ld_frag(a.KV + row * a.ldk + 8 * hh + 16 * s);
            hf[s] = ld_frag(a.dOh_other + row * D + 8 * hh + 16 * s);
        }
        ndv = a.nd_other[row];
        if (!fast) lov = a.lse_other[row] * LOG2E;
    };
    load_tile(0);
    for (int k0 = 0; k0 < a.n_kv; k0 += 32) {
        lds_sync();                                               // the previous trip's transposed reads are done
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8_t*>(tK + wr_off + 16 * s) = kf[s];
            *reinterpret_cast<bf16x8_t*>(tH + wr_off + 16 * s) = hf[s];
        }
        if (hh == 0) { sNd[r] = ndv; if (!fast) sL[r] = lov; }
        f32x16_t x = MFMA32(kf[0], qh[0], cl);                    // c2 S - lse2_own: key on the accumulator row, own row on the lane
        x = MFMA32(kf[0], ql[0], x);
        f32x16_t u = MFMA32(kf[0], dof[0], cd);                   // dP(own direction) - delta_own
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            x = MFMA32(kf[s], qh[s], x);
            x = MFMA32(kf[s], ql[s], x);
            u = MFMA32(kf[s], dof[s], u);
        }
        lds_sync();                                               // tiles and -c delta visible
        f32x16_t y;                                               // c[j] (dP(other direction) - delta_other[j]): C operand = -c[j] delta[j] per key
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 n4 = *reinterpret_cast<const float4*>(sNd + 8 * g + 4 * hh);
            y[4 * g] = n4.x; y[4 * g + 1] = n4.y; y[4 * g + 2] = n4.z; y[4 * g + 3] = n4.w;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) y = MFMA32(hf[s], qf[s], y);
        const bool tail = k0 + 32 > a.n_kv;
        if (k0 + 32 < a.n_kv) load_tile(k0 + 32);
        float e[16], t[16];
        if (tail) {                                               // wave-uniform: padded keys contribute nothing
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) x[reg] = k0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh < a.n_kv ? x[reg] : -INFINITY;
        }
        if (fast) {
#pragma unroll
            for (int reg = 0; reg < 16; reg += 2) {
                const f32x2_t ea = {__builtin_amdgcn_exp2f(x[reg]), __builtin_amdgcn_exp2f(x[reg + 1])};
                const f32x2_t w = pk_fma(pk_splat(rsc), (f32x2_t){y[reg], y[reg + 1]}, (f32x2_t){u[reg], u[reg + 1]});
                const f32x2_t tt = ea * w;
                e[reg] = ea.x; e[reg + 1] = ea.y; t[reg] = tt.x; t[reg + 1] = tt.y;
            }
        } else {                                                  // slow path: the other direction's probabilities by their own exponential
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l4 = *reinterpret_cast<const float4*>(sL + 8 * g + 4 * hh);
                const float lo[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int reg = 4 * g + cc;
                    const float e1 = __builtin_amdgcn_exp2f(x[reg]);
                    const float e2 = __builtin_amdgcn_exp2f(x[reg] + (lse2 - lo[cc]));
                    t[reg] = fmaf(e1, u[reg], e2 * y[reg]);
                    e[reg] = e2;
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            Ga = MFMA32(tr_frag(trK, s2, D), pack_frag(t + 8 * s2), Ga);
            Gb = MFMA32(tr_frag(trH, s2, D), pack_frag(e + 8 * s2), Gb);
        }
    }
    if (okq) {
        bf16_t* op = a.G + rowq * a.ldg;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            uint2 w;
            w.x = pack_bf2(fmaf(Ga[4 * g + 0], pp.scale, rsc * Gb[4 * g + 0]), fmaf(Ga[4 * g + 1], pp.scale, rsc * Gb[4 * g + 1]));
            w.y = pack_bf2(fmaf(Ga[4 * g + 2], pp.scale, rsc * Gb[4 * g + 2]), fmaf(Ga[4 * g + 3], pp.scale, rsc * Gb[4 * g + 3]));
            if (a.jZ) {                                     // kernel-uniform: (dX + G) * act' on the bf16-ROUNDED G, add3_mul's arithmetic
                const uint2 xv = *reinterpret_cast<const uint2*>(a.jX + rowq * a.ldjx + 8 * g + 4 * hh);
                const uint2 zv = *reinterpret_cast<const uint2*>(a.jZ + rowq * a.ldjz + 8 * g + 4 * hh);
                w.x = pack_bf2((__uint_as_float(xv.x << 16) + __uint_as_float(w.x << 16)) * __uint_as_float(zv.x << 16),
                               (__uint_as_float(xv.x & 0xffff0000u) + __uint_as_float(w.x & 0xffff0000u)) * __uint_as_float(zv.x & 0xffff0000u));
                w.y = pack_bf2((__uint_as_float(xv.y << 16) + __uint_as_float(w.y << 16)) * __uint_as_float(zv.y << 16),
                               (__uint_as_float(xv.y & 0xffff0000u) + __uint_as_float(w.y & 0xffff0000u)) * __uint_as_float(zv.y & 0xffff0000u));
            }
            *reinterpret_cast<uint2*>(op + 8 * g + 4 * hh) = w;
        }
    }
}

XP make(const stg_attn_args* f) {
    XP p = {};
    p.Q = (const bf16_t*)f->Q; p.ldq = f->ldq;
    p.KV = (const bf16_t*)f->K; p.ldk = f->ldk;
    p.O = (bf16_t*)f->O; p.ldo = f->ldo;
    p.lse = f->lse;
    p.outer_q = f->outer_q; p.outer_kv = f->outer_kv;
    p.P = (int)f->P; p.n = f->n; p.n_kv = f->n_kv;
    p.scale = f->scale; p.c2 = f->scale * LOG2E;
    return p;
}

}  // namespace

extern "C" int stg_xattn_pair_bwd_supported(const stg_attn_args* f0, const stg_attn_args* f1);

// Eligibility of a generic attention description for these kernels (see xattn.h)
bool stg_xattn_eligible(const stg_attn_args* f, bool need_lse) {
    return f->H == 1 && (f->D == 16 || f->D == 32) && f->map_kind == 0 && !f->map_q && !f->map_kv && !f->bias && !f->mask &&
           f->K == f->V && f->ldk == f->ldv && f->G == 1 && f->n >= 64 && f->n_kv >= 64 && f->scale > 0.f &&
           f->outer_q >= f->n && f->outer_kv >= f->n_kv && f->P * (int64_t)((f->n + 31) / 32) < (1ll << 30) &&
           f->P * (int64_t)((f->n_kv + 31) / 32) < (1ll << 30) && (!need_lse || f->lse) &&
           f->ldq % 8 == 0 && f->ldk % 8 == 0 && f->ldo % 4 == 0 &&
           (((uintptr_t)f->Q | (uintptr_t)f->K) & 15) == 0 && ((uintptr_t)f->O & 7) == 0 &&
           (int64_t)f->n_kv * f->ldk < (1ll << 30);              // 32-bit byte offsets of the key rows inside a problem
}

static int xattn_fwd_launch(const stg_attn_args* f0, const stg_attn_args* f1, void* stream, const float* gate0 = nullptr,
                            const float* gate1 = nullptr, void* x0 = nullptr, void* x1 = nullptr, int64_t ldx = 0) {
    XP2 pp = {};
    pp.a[0] = make(f0);
    const int ny = f1 ? 2 : 1;
    if (f1) pp.a[1] = make(f1);
    if (x0) {
        pp.a[0].gate = gate0; pp.a[0].X = (bf16_t*)x0; pp.a[0].ldx = ldx;
        pp.a[1].gate = gate1; pp.a[1].X = (bf16_t*)x1; pp.a[1].ldx = ldx;
    }
    if (pp.a[0].P == 0) return 0;
    for (int y = 0; y < ny; ++y) {
        XP& p = pp.a[y];
        STG_CHECK(p.lse != nullptr, -1, "stg_attn_fwd (cross-modal path): lse is required");
        p.tiles = (p.n + 31) / 32;
        p.total = p.P * p.tiles;
    }
    // the pair's grid covers the LARGER direction (with n_v != n_a, direction 1's extra query tiles never ran); a workgroup
    // whose items lie beyond its direction's total returns at once
    const int tmax = ny == 2 && pp.a[1].total > pp.a[0].total ? pp.a[1].total : pp.a[0].total;
    const dim3 grid((tmax + 3) / 4, ny), block(256);
    if (f0->D == 16) hipLaunchKernelGGL(xattn_fwd_kernel<16>, grid, block, 0, (hipStream_t)stream, pp);
    else hipLaunchKernelGGL(xattn_fwd_kernel<32>, grid, block, 0, (hipStream_t)stream, pp);
    STG_LAUNCH_CHECK();
    return 0;
}

int stg_xattn_fwd(const stg_attn_args* f, void* stream) { return xattn_fwd_launch(f, nullptr, stream); }

// same geometry (P, n, n_kv, D, scale): the pair shares one grid
bool stg_xattn_pairable(const stg_attn_args* f0, const stg_attn_args* f1) {
    return f0->P == f1->P && f0->n == f1->n && f0->n_kv == f1->n_kv && f0->D == f1->D;
}
int stg_xattn_fwd2(const stg_attn_args* f0, const stg_attn_args* f1, void* stream) { return xattn_fwd_launch(f0, f1, stream); }

// both directions of a frame-global cross-modal pair AND its gates (x = q + gate o) in one launch
extern "C" int stg_xattn_fwd2_gate(const stg_attn_args* f0, const stg_attn_args* f1, const float* gate0, const float* gate1, void* x0, void* x1,
                                   int64_t ldx, void* stream) {
    STG_CHECK(f0 && f1 && gate0 && gate1 && x0 && x1, -1, "stg_xattn_fwd2_gate: null pointer");
    STG_CHECK(stg_xattn_pair_bwd_supported(f0, f1), -2, "stg_xattn_fwd2_gate: not a frame-global cross-modal pair these kernels take");
    STG_CHECK(ldx % 4 == 0 && ldx >= f0->D && (((uintptr_t)x0 | (uintptr_t)x1) & 7) == 0, -2, "stg_xattn_fwd2_gate: bad x operands");
    return xattn_fwd_launch(f0, f1, stream, gate0, gate1, x0, x1, ldx);
}

static int xattn_bwd_launch(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* stream) {
    XP2 pp = {};
    const stg_attn_bwd_args* bs[2] = {b0, b1};
    const int ny = b1 ? 2 : 1;
    for (int y = 0; y < ny; ++y) {
        const stg_attn_bwd_args* b = bs[y];
        XP& p = pp.a[y];
        p = make(&b->f);
        if (p.P == 0) return 0;
        STG_CHECK(b->dO && b->dQ && b->dK && b->delta && p.lse && p.O, -1, "stg_attn_bwd (cross-modal path): null pointer");
        STG_CHECK(b->lddo % 8 == 0 && b->lddq % 4 == 0 && b->lddk % 4 == 0 && ((uintptr_t)b->dO & 15) == 0 &&
                  (((uintptr_t)b->dQ | (uintptr_t)b->dK) & 7) == 0 && p.ldo % 8 == 0 && ((uintptr_t)p.O & 15) == 0, -2,
                  "stg_attn_bwd (cross-modal path): misaligned operands");
        p.dO = (const bf16_t*)b->dO; p.lddo = b->lddo;
        p.dQ = (bf16_t*)b->dQ; p.lddq = b->lddq;
        p.dKV = (bf16_t*)b->dK; p.lddk = b->lddk;
        p.delta = b->delta;
    }
    const dim3 block(256);
    auto tmax = [&]() { return ny == 2 && pp.a[1].total > pp.a[0].total ? pp.a[1].total : pp.a[0].total; };
    for (int y = 0; y < ny; ++y) { pp.a[y].tiles = (pp.a[y].n + 31) / 32; pp.a[y].total = pp.a[y].P * pp.a[y].tiles; }
    if (b0->f.D == 16) hipLaunchKernelGGL(xattn_dq_kernel<16>, dim3((tmax() + 3) / 4, ny), block, 0, (hipStream_t)stream, pp);
    else hipLaunchKernelGGL(xattn_dq_kernel<32>, dim3((tmax() + 3) / 4, ny), block, 0, (hipStream_t)stream, pp);
    STG_LAUNCH_CHECK();
    for (int y = 0; y < ny; ++y) { pp.a[y].tiles = (pp.a[y].n_kv + 31) / 32; pp.a[y].total = pp.a[y].P * pp.a[y].tiles; }
    if (b0->f.D == 16) hipLaunchKernelGGL(xattn_dkv_kernel<16>, dim3((tmax() + 3) / 4, ny), block, 0, (hipStream_t)stream, pp);
    else hipLaunchKernelGGL(xattn_dkv_kernel<32>, dim3((tmax() + 3) / 4, ny), block, 0, (hipStream_t)stream, pp);
    STG_LAUNCH_CHECK();
    return 0;
}

int stg_xattn_bwd(const stg_attn_bwd_args* b, void* stream) { return xattn_bwd_launch(b, nullptr, stream); }
int stg_xattn_bwd2(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* stream) { return xattn_bwd_launch(b0, b1, stream); }

// ---- the merged pair backward (see the comment above xattn_bwdm_kernel)
extern "C" int64_t stg_xattn_pair_bwd_ws_bytes(int64_t P, int n0, int n1, int D) {
    const int64_t rows = P * ((int64_t)n0 + n1);
    return rows * D * 2 + rows * 4 * 2 + P * 8 + 256;            // dOh (bf16 [rows, D]) + nd + delta (fp32 [rows]) + c0, ok per frame (+ alignment)
}

extern "C" int stg_xattn_pair_bwd_supported(const stg_attn_args* f0, const stg_attn_args* f1) {
    // direction 0: queries = modality v, keys = modality a; direction 1 the mirror image (same tensors, roles swapped); dense frames
    return stg_opt_xattn.load(std::memory_order_relaxed) != 0 && stg_xattn_eligible(f0, true) && stg_xattn_eligible(f1, true) &&
           f0->P == f1->P && f0->D == f1->D && f0->n == f1->n_kv && f0->n_kv == f1->n && f0->outer_q == f0->n && f0->outer_kv == f0->n_kv &&
           f1->outer_q == f1->n && f1->outer_kv == f1->n_kv && f0->Q == f1->K && f0->K == f1->Q && f0->scale == f1->scale && f0->O && f1->O &&
           f0->P < (1 << 20) ? 1 : 0;
}

static int xattn_pair_bwd_impl(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, const void* jx0,
                               const void* jx1, int64_t ldjx, const void* jz0, const void* jz1, int64_t ldjz, void* ws, int64_t ws_bytes, void* stream,
                               const float* gate0 = nullptr, const float* gate1 = nullptr, float* dgate0 = nullptr, float* dgate1 = nullptr);

extern "C" int stg_xattn_pair_bwd(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, void* ws,
                                  int64_t ws_bytes, void* stream) {
    return xattn_pair_bwd_impl(b0, b1, g0, g1, ldg, nullptr, nullptr, 0, nullptr, nullptr, 0, ws, ws_bytes, stream);
}

/* stg_xattn_pair_bwd followed by the join of the adapters' backward in the same launch: g <- (jx + G) * jz per modality (jx = the gradient of
 * the gated hidden state, jz = the saved activation derivative of D_fc1: what stg_add3_mul2 did in its own pass). */
extern "C" int stg_xattn_pair_bwd_join(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, const void* jx0,
                                       const void* jx1, int64_t ldjx, const void* jz0, const void* jz1, int64_t ldjz, void* ws, int64_t ws_bytes,
                                       void* stream) {
    STG_CHECK(jx0 && jx1 && jz0 && jz1 && ldjx % 4 == 0 && ldjz % 4 == 0 && (((uintptr_t)jx0 | (uintptr_t)jx1 | (uintptr_t)jz0 | (uintptr_t)jz1) & 7) == 0, -2,
              "stg_xattn_pair_bwd_join: bad join operands");
    return xattn_pair_bwd_impl(b0, b1, g0, g1, ldg, jx0, jx1, ldjx, jz0, jz1, ldjz, ws, ws_bytes, stream);
}

/* ABI 220: ... with the pair's GATES inside: b0->dO / b1->dO are d(x) of x = q + gate o (what the forward's stg_xattn_fwd2_gate wrote), the kernels work on
 * bf16(gate d(x)) -- stg_gate_bwd2's rounding, so G is bit-identical to stg_gate_bwd2 + stg_xattn_pair_bwd(_join) -- and dgate_y[0] += <d(x_y), o_y>.
 * jx / jz: the optional join as in stg_xattn_pair_bwd_join (all four NULL: none). */
extern "C" int stg_xattn_pair_bwd_gate(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, const void* jx0,
                                       const void* jx1, int64_t ldjx, const void* jz0, const void* jz1, int64_t ldjz, const float* gate0, const float* gate1,
                                       float* dgate0, float* dgate1, void* ws, int64_t ws_bytes, void* stream) {
    STG_CHECK(gate0 && gate1 && dgate0 && dgate1, -1, "stg_xattn_pair_bwd_gate: null gate pointer");
    const bool join = jx0 || jx1 || jz0 || jz1;
    STG_CHECK(!join || (jx0 && jx1 && jz0 && jz1 && ldjx % 4 == 0 && ldjz % 4 == 0 && (((uintptr_t)jx0 | (uintptr_t)jx1 | (uintptr_t)jz0 | (uintptr_t)jz1) & 7) == 0), -2,
              "stg_xattn_pair_bwd_gate: bad join operands");
    return xattn_pair_bwd_impl(b0, b1, g0, g1, ldg, jx0, jx1, ldjx, jz0, jz1, ldjz, ws, ws_bytes, stream, gate0, gate1, dgate0, dgate1);
}

static int xattn_pair_bwd_impl(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, const void* jx0,
                               const void* jx1, int64_t ldjx, const void* jz0, const void* jz1, int64_t ldjz, void* ws, int64_t ws_bytes, void* stream,
                               const float* gate0, const float* gate1, float* dgate0, float* dgate1) {
    STG_CHECK(b0 && b1 && g0 && g1 && ws, -1, "stg_xattn_pair_bwd: null pointer");
    STG_CHECK(stg_xattn_pair_bwd_supported(&b0->f, &b1->f), -2, "stg_xattn_pair_bwd: not a frame-global cross-modal pair these kernels take");
    const stg_attn_bwd_args* bs[2] = {b0, b1};
    void* gs[2] = {g0, g1};
    const int D = b0->f.D;
    const int64_t P = b0->f.P;
    if (P == 0) return 0;
    STG_CHECK(ws_bytes >= stg_xattn_pair_bwd_ws_bytes(P, b0->f.n, b1->f.n, D) && ((uintptr_t)ws & 15) == 0, -2, "stg_xattn_pair_bwd: workspace too small / misaligned");
    STG_CHECK(ldg % 4 == 0 && ldg >= D && (((uintptr_t)g0 | (uintptr_t)g1) & 7) == 0, -2, "stg_xattn_pair_bwd: bad G operands");
    XM2 pp = {};
    char* w = (char*)ws;
    bf16_t* dOh[2]; float* nd[2]; float* dl[2];
    for (int y = 0; y < 2; ++y) { dOh[y] = (bf16_t*)w; w += P * bs[y]->f.n * D * 2; }
    for (int y = 0; y < 2; ++y) { nd[y] = (float*)w; w += P * bs[y]->f.n * 4; }
    for (int y = 0; y < 2; ++y) { dl[y] = (float*)w; w += P * bs[y]->f.n * 4; }
    pp.c0 = (float*)w; w += P * 4;
    pp.ok = (int*)w;
    pp.P = (int)P; pp.scale = b0->f.scale; pp.c2 = b0->f.scale * LOG2E;
    for (int y = 0; y < 2; ++y) {
        const stg_attn_bwd_args* b = bs[y];
        STG_CHECK(b->dO && b->f.lse && b->lddo % 8 == 0 && ((uintptr_t)b->dO & 15) == 0 && b->f.ldo % 8 == 0 && ((uintptr_t)b->f.O & 15) == 0, -2,
                  "stg_xattn_pair_bwd: misaligned operands");
        XM& a = pp.a[y];
        a.Q = (const bf16_t*)b->f.Q; a.ldq = b->f.ldq; a.dO = (const bf16_t*)b->dO; a.lddo = b->lddo; a.O = (const bf16_t*)b->f.O; a.ldo = b->f.ldo;
        a.lse = b->f.lse; a.delta = dl[y]; a.KV = (const bf16_t*)b->f.K; a.ldk = b->f.ldk;
        a.dOh_other = dOh[1 - y]; a.nd_other = nd[1 - y]; a.dOh_own = dOh[y]; a.nd_own = nd[y]; a.lse_other = bs[1 - y]->f.lse;
        a.G = (bf16_t*)gs[y]; a.ldg = ldg; a.n = b->f.n; a.n_kv = b->f.n_kv;
        a.jX = (const bf16_t*)(y ? jx1 : jx0); a.jZ = (const bf16_t*)(y ? jz1 : jz0); a.ldjx = ldjx; a.ldjz = ldjz;
        a.gate = y ? gate1 : gate0; a.dgate = y ? dgate1 : dgate0;
        a.tiles = (a.n + 31) / 32; a.total = (int)(P * a.tiles);
    }
    hipStream_t st = (hipStream_t)stream;
    // preparation grid: one row per thread and part (8 parts of 25 rows at stage 2's 196-token frames were 2 560 workgroups of a tenth of a wave's work --
    // and, with the gates inside, 2 560 same-address atomics)
    const int nmax = pp.a[0].n > pp.a[1].n ? pp.a[0].n : pp.a[1].n;
    const int parts = nmax >= 2048 ? 8 : (nmax + 255) / 256;
    const int tmax = pp.a[0].total > pp.a[1].total ? pp.a[0].total : pp.a[1].total;
    if (D == 16) {
        hipLaunchKernelGGL(xattn_prep_kernel<16>, dim3((unsigned)P, parts), dim3(256), 0, st, pp);
        hipLaunchKernelGGL(xattn_bwdm_kernel<16>, dim3((tmax + 3) / 4, 2), dim3(256), 0, st, pp);
    } else {
        hipLaunchKernelGGL(xattn_prep_kernel<32>, dim3((unsigned)P, parts), dim3(256), 0, st, pp);
        hipLaunchKernelGGL(xattn_bwdm_kernel<32>, dim3((tmax + 3) / 4, 2), dim3(256), 0, st, pp);
    }
    STG_LAUNCH_CHECK();
    return 0;
}
