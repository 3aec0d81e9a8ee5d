// Temporal attention of the STG-CMA Swin block (WindowAttention.forward temporal branch, Swin_AVE.py:244-255, called from
// the block at :705-716): every spatial token attends over its own T frames (T = 10 AVE / AVQA, 5 AVS), head dim 32, additive
// temporal_position_bias_table(_audio) term, no mask.  The '(b t) n c -> (b n) t c' rearranges (:705,711) are addressing:
// frame t of token n of clip b of modality slab m lives at row ((m*B + b)*T + t)*N + n of the fused token tensor.
//
// Why its own kernel family (next to attention.hip): the sequences are tiny, so the op is pure HBM streaming -- what matters
// is that every byte of q/k/v is read once and that the backward is ONE pass.  `per = 32 / T` sequences of adjacent tokens
// n0 .. n0+per-1 are packed into one 32-row MFMA tile (v_mfma_f32_32x32x16_bf16; off-diagonal blocks and padding rows are
// switched off by a -1e30 additive table built from the bias), one wave owns one (token group, head), the four waves of a
// workgroup own four adjacent heads of the SAME rows (so the workgroup consumes whole 256-byte row segments), and a wave
// walks token groups of one (modality, head) in a grid-stride loop.  The 32 x 32 score block lives in registers: the softmax
// is single-pass, and the backward recomputes it from q, k alone -- it needs neither the forward output nor an LSE, and
// produces dQ, dK, dV and the bias-table gradient in the same kernel (query-on-lane pass for dQ / dbias, key-on-lane pass for
// dK / dV from the same operand registers).  dbias is accumulated in registers over the whole loop and leaves with T*T
// atomics per wave.
//
// Lane l = (r = l & 31, hh = l >> 5); accumulator row (reg, hh) = (reg & 3) + 8 * (reg >> 2) + 4 * hh.
#include <math.h>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr int TD = 32;             // head dim
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

struct TP {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; int64_t ld;
    bf16_t* O; int64_t ldo;
    const float* bm; const float* bmT;      // [nm*H][32][32]: query-major / key-major additive term (log2 domain)
    int nm, B, T, N, H, per, gpb, ngroups;  // gpb = token groups per (m, b) = ceil(N / per); ngroups = B * gpb
    float scale, scale2;
    const bf16_t* dO; int64_t lddo;
    bf16_t* dQ; bf16_t* dK; bf16_t* dV; int64_t lddqkv;
    float* dbias;                           // [nm*H][T*T] or NULL
};

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
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef short s4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pick(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// ------------------------------------------------------------------------------------------------ coalesced kernels (round 2)
// The round-1 kernels (removed in round 6) fetched their operand fragments and stored their results ROW-PER-LANE (a lane moves 16 bytes of its own token row):
// 64 separate requests per wave-instruction, ~64 clocks each in the CU's address path, 8 / 14 such instructions per group.  Here every
// global access is coalesced -- four lanes cover the 64 bytes of a head's row: the operand tiles of the NEXT group arrive by LDS-DMA
// (two instructions per 32-row tile) in a second set of LDS tiles while the current group is computed, fragments are read from LDS, and
// the results leave through a 32 x 32 LDS transposition.  The backward is ONE pass, query on the lane: P and dS feed dQ from the
// accumulators and come back transposed (ds_read_b64_tr_b16) from two LDS tiles as the B operands of dV^T = dO^T P, dK^T = Q^T dS --
// no second orientation, no lse / delta broadcast, half the exponentials, no key-major table.
// Tiles [32][32] bf16 hold SOURCE chunk c of row R at chunk c ^ ((R >> 2) & 3) (operand tiles: conflict-free row fragments and
// transposed reads); P / dS / output tiles hold 8-byte piece u of row q at u ^ ((q >> 2) & 7).
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * TD + ((chunk ^ ((row >> 2) & 3)) << 3); }
__device__ __forceinline__ bf16x8_t tr_frag_sw(const bf16_t* s, int s2, int hh, int d) {       // tr_frag on an sw_off tile
    const int gi = d & 15, c = d >> 4;
    const int row = 16 * s2 + 4 * hh + (gi >> 2), u = 4 * c + (gi & 3);
    const bf16_t* p0 = s + sw_off(row, u >> 1) + ((u & 1) << 2);
    const bf16_t* p1 = s + sw_off(row + 8, u >> 1) + ((u & 1) << 2);
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p0);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p1);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
__device__ __forceinline__ bf16x8_t tr_frag_pc(const bf16_t* s, int s2, int hh, int d) {       // ... on a piece-swizzled tile
    const int gi = d & 15, c = d >> 4;
    const int row = 16 * s2 + 4 * hh + (gi >> 2), u = 4 * c + (gi & 3);
    const bf16_t* p0 = s + row * 32 + ((u ^ ((row >> 2) & 7)) << 2);
    const bf16_t* p1 = s + (row + 8) * 32 + ((u ^ (((row + 8) >> 2) & 7)) << 2);
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p0);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p1);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
__device__ __forceinline__ void put_pieces(bf16_t* T, const bf16x8_t& lo8, const bf16x8_t& hi8, int r, int hh) {   // packed regs 0..7, 8..15
    const int swz = (r >> 2) & 7;
    const u32x4_t w0 = __builtin_bit_cast(u32x4_t, lo8), w1 = __builtin_bit_cast(u32x4_t, hi8);
    *reinterpret_cast<uint2*>(T + r * 32 + (((0 + hh) ^ swz) << 2)) = make_uint2(w0[0], w0[1]);
    *reinterpret_cast<uint2*>(T + r * 32 + (((2 + hh) ^ swz) << 2)) = make_uint2(w0[2], w0[3]);
    *reinterpret_cast<uint2*>(T + r * 32 + (((4 + hh) ^ swz) << 2)) = make_uint2(w1[0], w1[1]);
    *reinterpret_cast<uint2*>(T + r * 32 + (((6 + hh) ^ swz) << 2)) = make_uint2(w1[2], w1[3]);
}
__device__ __forceinline__ void put_acc(bf16_t* T, const f32x16_t& acc, float sc, int r, int hh) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = acc[i] * sc;
    put_pieces(T, pack8(x), pack8(x + 8), r, hh);
}
// 16 token rows x 64 bytes per store instruction; rowoff[j]: element offset of tile row (lane >> 2) + 16 j
__device__ __forceinline__ void flush_rows(const bf16_t* T, bf16_t* dst, const int64_t (&rowoff)[2], int nvalid, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tl = (lane >> 2) + 16 * j, cq = lane & 3, sw = (tl >> 2) & 7;
        const uint2 lo = *reinterpret_cast<const uint2*>(T + tl * 32 + (((2 * cq) ^ sw) << 2));
        const uint2 hi = *reinterpret_cast<const uint2*>(T + tl * 32 + (((2 * cq + 1) ^ sw) << 2));
        if (tl < nvalid) *reinterpret_cast<uint4*>(dst + rowoff[j] + cq * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}
#define TATTN_DMA(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                                             (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__global__ void __launch_bounds__(256, 2) tattn_fwd1_kernel(TP a) {
    constexpr int SET = 3 * 32 * TD;                  // Q, K, V tiles
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 2 * SET];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y * 4 + wave, m = blockIdx.z;
    if (h >= a.H) return;
    const int nv = a.per * a.T;
    bf16_t* base = smem + wave * 2 * SET;
    const float* bmq = a.bm + ((int64_t)(m * a.H + h) * 32 + r) * 32;
    float4 add[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) add[g4] = *reinterpret_cast<const float4*>(bmq + 8 * g4 + 4 * hh);
    // staging lanes: tile row R = (lane >> 2) + 16 j <-> (sequence s, frame t), source chunk (lane & 3) ^ ((R >> 2) & 3)
    const int cs = (lane & 3) ^ ((lane >> 4) & 3);
    int ss[2], tt[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = (lane >> 2) + 16 * j;
        ss[j] = R / a.T; tt[j] = R - ss[j] * a.T;
        if (R >= nv) { ss[j] = 0; tt[j] = 0; }
    }
    int64_t nrow[2] = {0, 0}, crow[2];                // token row of the staging lane's two tile rows: next / current group
    int ncnt = 0;
    auto issue = [&](int g, int set) {
        const int b = g / a.gpb, jg = g - b * a.gpb;
        const int n0 = jg * a.per;
        const int cnt = min(a.per, a.N - n0);
        ncnt = cnt;
        bf16_t* d = base + set * SET;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t row = ((int64_t)(m * a.B + b) * a.T + tt[j]) * a.N + n0 + (ss[j] < cnt ? ss[j] : cnt - 1);
            nrow[j] = row;
            const int64_t off = row * a.ld + h * TD + cs * 8;
            TATTN_DMA(a.Q + off, d + j * 512);
            TATTN_DMA(a.K + off, d + 32 * TD + j * 512);
            TATTN_DMA(a.V + off, d + 64 * TD + j * 512);
        }
    };
    int set = 0;
    if ((int)blockIdx.x < a.ngroups) issue(blockIdx.x, 0);
    for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x, set ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this group's tiles have landed (and the last group's stores are out)
        lds_fence();
        crow[0] = nrow[0]; crow[1] = nrow[1];
        const int nvalid = ncnt * a.T;
        if (g + (int)gridDim.x < a.ngroups) issue(g + gridDim.x, set ^ 1);
        const bf16_t* sQ = base + set * SET;
        const bf16_t* sK = sQ + 32 * TD;
        const bf16_t* sV = sK + 32 * TD;
        f32x16_t st = zero16();                       // St[key][q]
        st = MFMA32(ld_frag(sK + sw_off(r, hh)), ld_frag(sQ + sw_off(r, hh)), st);
        st = MFMA32(ld_frag(sK + sw_off(r, hh + 2)), ld_frag(sQ + sw_off(r, hh + 2)), st);
        float x[16];
        float mx = NEG_BIG;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = st[reg] * a.scale2 + pick(add[reg >> 2], reg & 3);
            mx = fmaxf(mx, x[reg]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = __builtin_amdgcn_exp2f(x[reg] - mx);
            l += x[reg];
        }
        l += __shfl_xor(l, 32, 64);
        f32x16_t o = zero16();                        // O^T[d][q]
        o = MFMA32(tr_frag_sw(sV, 0, hh, r), pack8(x), o);
        o = MFMA32(tr_frag_sw(sV, 1, hh, r), pack8(x + 8), o);
        lds_fence();                                  // every read of the Q tile is done: O leaves through it
        bf16_t* T = base + set * SET;
        put_acc(T, o, 1.0f / l, r, hh);
        lds_fence();
        const int64_t ro[2] = {crow[0] * a.ldo + h * TD, crow[1] * a.ldo + h * TD};
        flush_rows(T, a.O, ro, nvalid, lane);
    }
}

__global__ void __launch_bounds__(256, 2) tattn_bwd1_kernel(TP a) {
    constexpr int SET = 4 * 32 * TD;                  // K, Q, dO, V tiles
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 2 * SET];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y * 4 + wave, m = blockIdx.z;
    if (h >= a.H) return;
    const int nv = a.per * a.T;
    bf16_t* base = smem + wave * 2 * SET;
    const int64_t tb = ((int64_t)(m * a.H + h) * 32 + r) * 32;
    float4 add[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) add[g4] = *reinterpret_cast<const float4*>(a.bm + tb + 8 * g4 + 4 * hh);      // lane = query, float4 along keys
    f32x16_t dbacc = zero16();                        // sum of dS^T[key][q] over this wave's groups
    const int cs = (lane & 3) ^ ((lane >> 4) & 3);
    int ss[2], tt[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = (lane >> 2) + 16 * j;
        ss[j] = R / a.T; tt[j] = R - ss[j] * a.T;
        if (R >= nv) { ss[j] = 0; tt[j] = 0; }
    }
    int64_t nrow[2] = {0, 0}, crow[2];
    int ncnt = 0;
    auto issue = [&](int g, int set) {
        const int b = g / a.gpb, jg = g - b * a.gpb;
        const int n0 = jg * a.per;
        const int cnt = min(a.per, a.N - n0);
        ncnt = cnt;
        bf16_t* d = base + set * SET;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t row = ((int64_t)(m * a.B + b) * a.T + tt[j]) * a.N + n0 + (ss[j] < cnt ? ss[j] : cnt - 1);
            nrow[j] = row;
            const int64_t off = row * a.ld + h * TD + cs * 8;
            TATTN_DMA(a.K + off, d + j * 512);
            TATTN_DMA(a.Q + off, d + 32 * TD + j * 512);
            TATTN_DMA(a.dO + row * a.lddo + h * TD + cs * 8, d + 64 * TD + j * 512);
            TATTN_DMA(a.V + off, d + 96 * TD + j * 512);
        }
    };
    int set = 0;
    if ((int)blockIdx.x < a.ngroups) issue(blockIdx.x, 0);
    for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x, set ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_fence();
        crow[0] = nrow[0]; crow[1] = nrow[1];
        const int nvalid = ncnt * a.T;
        if (g + (int)gridDim.x < a.ngroups) issue(g + gridDim.x, set ^ 1);
        bf16_t* sK = base + set * SET;
        bf16_t* sQ = sK + 32 * TD;
        bf16_t* sD = sQ + 32 * TD;
        bf16_t* sV = sD + 32 * TD;
        f32x16_t st = zero16(), dpt = zero16();       // St[key][q], dPt[key][q]
        st = MFMA32(ld_frag(sK + sw_off(r, hh)), ld_frag(sQ + sw_off(r, hh)), st);
        dpt = MFMA32(ld_frag(sV + sw_off(r, hh)), ld_frag(sD + sw_off(r, hh)), dpt);
        st = MFMA32(ld_frag(sK + sw_off(r, hh + 2)), ld_frag(sQ + sw_off(r, hh + 2)), st);
        dpt = MFMA32(ld_frag(sV + sw_off(r, hh + 2)), ld_frag(sD + sw_off(r, hh + 2)), dpt);
        float x[16], pr[16];
        float mx = NEG_BIG;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = st[reg] * a.scale2 + pick(add[reg >> 2], reg & 3);
            mx = fmaxf(mx, x[reg]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = __builtin_amdgcn_exp2f(x[reg] - mx);
            l += x[reg];
        }
        l += __shfl_xor(l, 32, 64);
        const float inv = (r < nvalid) ? 1.0f / l : 0.f;          // padded / absent sequences contribute nothing
        float delta = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            pr[reg] = x[reg] * inv;
            delta += pr[reg] * dpt[reg];
        }
        delta += __shfl_xor(delta, 32, 64);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = pr[reg] * (dpt[reg] - delta);                  // dS^T[key][q]
            dbacc[reg] += x[reg];
        }
        const bf16x8_t ps0 = pack8(x), ps1 = pack8(x + 8), pp0 = pack8(pr), pp1 = pack8(pr + 8);
        f32x16_t dq = zero16();
        dq = MFMA32(tr_frag_sw(sK, 0, hh, r), ps0, dq);
        dq = MFMA32(tr_frag_sw(sK, 1, hh, r), ps1, dq);
        lds_fence();                                  // the K and V tiles are dead: they take dS and P (T[q][key], piece-swizzled)
        put_pieces(sK, ps0, ps1, r, hh);
        put_pieces(sV, pp0, pp1, r, hh);
        lds_fence();
        f32x16_t dv = zero16(), dk = zero16();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dv = MFMA32(tr_frag_sw(sD, s2, hh, r), tr_frag_pc(sV, s2, hh, r), dv);
            dk = MFMA32(tr_frag_sw(sQ, s2, hh, r), tr_frag_pc(sK, s2, hh, r), dk);
        }
        lds_fence();                                  // all four tiles are dead: dQ, dK, dV leave through three of them
        put_acc(sK, dq, a.scale, r, hh);
        put_acc(sQ, dk, a.scale, r, hh);
        put_acc(sD, dv, 1.0f, r, hh);
        lds_fence();
        const int64_t ro[2] = {crow[0] * a.lddqkv + h * TD, crow[1] * a.lddqkv + h * TD};
        flush_rows(sK, a.dQ, ro, nvalid, lane);
        flush_rows(sQ, a.dK, ro, nvalid, lane);
        flush_rows(sD, a.dV, ro, nvalid, lane);
    }

    if (a.dbias) {
        // fold the `per` diagonal T x T blocks of the accumulated dS^T[key][q] and add them to dbias[m][h][tq][tk]
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_fence();
        float* tile = reinterpret_cast<float*>(base);   // 32 x 32 fp32 = two tiles of set 0
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) tile[ACC_ROW(reg, hh) * 32 + r] = dbacc[reg];
        lds_fence();
        const int TT = a.T * a.T;
        for (int idx = lane; idx < TT; idx += 64) {
            const int tq = idx / a.T, tk = idx - tq * a.T;
            float acc = 0.f;
            for (int sq = 0; sq < a.per; ++sq) acc += tile[(sq * a.T + tk) * 32 + sq * a.T + tq];
            atomicAdd(a.dbias + (int64_t)(m * a.H + h) * TT + idx, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------ wide heads, no bias (ViT)
// The CLIP ViT blocks run the same temporal attention (ResidualAttentionBlock, CLIP_AVE.py:369-377: 'n (b t) d -> t (b n) d') through
// nn.MultiheadAttention: no additive bias, head dim 96 (ViT-B built with 8 heads) or 64 (ViT-L).  Same packing and one-pass backward;
// the block-diagonal structure is a 16-bit mask computed once per lane (same sequence <=> key / T == query / T), operand tiles sit
// in 16-byte-padded LDS rows, and the backward runs one wave per SIMD (its 96-wide dK / dV accumulators live in AGPRs).
template <int DP>
__device__ __forceinline__ bf16x8_t tr_frag_p(const bf16_t* s, int dt, int s2, int hh, int d) {
    const int gi = d & 15, c = d >> 4;
    const bf16_t* p = s + (16 * s2 + 4 * hh + (gi >> 2)) * DP + 32 * dt + 16 * c + 4 * (gi & 3);
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)(p + 8 * DP));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
template <int D>
__device__ __forceinline__ void park_p(bf16_t* s, int r, int hh, const bf16x8_t* f) {
#pragma unroll
    for (int k = 0; k < D / 16; ++k) *reinterpret_cast<bf16x8_t*>(s + r * (D + 8) + 16 * k + 8 * hh) = f[k];
}
// bit reg of the lane's mask: accumulator row (reg, hh) and the lane's own index r belong to the same packed sequence
__device__ __forceinline__ unsigned same_seq_mask(int r, int hh, int T, int nv) {
    unsigned m = 0;
    const int sr = r / T;
    for (int reg = 0; reg < 16; ++reg) {
        const int o = ACC_ROW(reg, hh);
        if (r < nv && o < nv && o / T == sr) m |= 1u << reg;
    }
    return m;
}

template <int D>
__global__ void __launch_bounds__(256, 2) tattn_nb_fwd_kernel(TP a) {
    constexpr int KS = D / 16, DT = D / 32, DP = D + 8;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 32 * DP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y * 4 + wave, m = blockIdx.z;
    if (h >= a.H) return;
    const int nv = a.per * a.T;
    int s = r / a.T, t = r - s * a.T;
    if (r >= nv) { s = 0; t = 0; }
    bf16_t* sV = smem + wave * 32 * DP;
    const unsigned okm = same_seq_mask(r, hh, a.T, nv);
    for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x) {
        const int b = g / a.gpb, j = g - b * a.gpb;
        const int n0 = j * a.per;
        const int cnt = min(a.per, a.N - n0);
        const int64_t row = ((int64_t)(m * a.B + b) * a.T + t) * a.N + n0 + (s < cnt ? s : cnt - 1);
        const int64_t off = row * a.ld + h * D + 8 * hh;
        bf16x8_t q[KS], k[KS], v[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) { q[i] = ld_frag(a.Q + off + 16 * i); k[i] = ld_frag(a.K + off + 16 * i); v[i] = ld_frag(a.V + off + 16 * i); }
        park_p<D>(sV, r, hh, v);
        f32x16_t st = zero16();
#pragma unroll
        for (int i = 0; i < KS; ++i) st = MFMA32(k[i], q[i], st);
        lds_fence();
        float x[16];
        float mx = NEG_BIG;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = (okm >> reg) & 1u ? st[reg] * a.scale2 : NEG_BIG;
            mx = fmaxf(mx, x[reg]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = __builtin_amdgcn_exp2f(x[reg] - mx);
            l += x[reg];
        }
        l += __shfl_xor(l, 32, 64);
        const bf16x8_t p0 = pack8(x), p1 = pack8(x + 8);
        f32x16_t o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            o[dt] = MFMA32(tr_frag_p<DP>(sV, dt, 0, hh, r), p0, zero16());
            o[dt] = MFMA32(tr_frag_p<DP>(sV, dt, 1, hh, r), p1, o[dt]);
        }
        lds_fence();
        {
            const float inv = 1.0f / l;
            bf16_t* op = a.O + row * a.ldo + h * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) store_tile32(op + 32 * dt, o[dt], inv, hh, r < cnt * a.T);
        }
    }
}

template <int D>
__global__ void __launch_bounds__(256, 2) tattn_nb_bwd_kernel(TP a) {
    constexpr int KS = D / 16, DT = D / 32, DP = D + 8;
    constexpr int PER_WAVE = 3 * 32 * DP + 128;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y * 4 + wave, m = blockIdx.z;
    if (h >= a.H) return;
    const int nv = a.per * a.T;
    int s = r / a.T, t = r - s * a.T;
    if (r >= nv) { s = 0; t = 0; }
    bf16_t* sK = smem + wave * PER_WAVE;
    bf16_t* sQ = sK + 32 * DP;
    bf16_t* sD = sQ + 32 * DP;
    float* sLse = reinterpret_cast<float*>(sD + 32 * DP);
    float* sDel = sLse + 32;
    const unsigned okm = same_seq_mask(r, hh, a.T, nv);
    for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x) {
        const int b = g / a.gpb, j = g - b * a.gpb;
        const int n0 = j * a.per;
        const int cnt = min(a.per, a.N - n0);
        const int nvalid = cnt * a.T;
        const int64_t row = ((int64_t)(m * a.B + b) * a.T + t) * a.N + n0 + (s < cnt ? s : cnt - 1);
        const int64_t off = row * a.ld + h * D + 8 * hh;
        const bf16_t* dp_ = a.dO + row * a.lddo + h * D + 8 * hh;
        bf16x8_t q[KS], k[KS], v[KS], d[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            q[i] = ld_frag(a.Q + off + 16 * i); k[i] = ld_frag(a.K + off + 16 * i); v[i] = ld_frag(a.V + off + 16 * i); d[i] = ld_frag(dp_ + 16 * i);
        }
        park_p<D>(sK, r, hh, k);
        park_p<D>(sQ, r, hh, q);
        park_p<D>(sD, r, hh, d);
        {   // phase A: query on the lane
            f32x16_t st = zero16(), dpt = zero16();
#pragma unroll
            for (int i = 0; i < KS; ++i) { st = MFMA32(k[i], q[i], st); dpt = MFMA32(v[i], d[i], dpt); }
            float x[16];
            float mx = NEG_BIG;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                x[reg] = (okm >> reg) & 1u ? st[reg] * a.scale2 : NEG_BIG;
                mx = fmaxf(mx, x[reg]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) { x[reg] = __builtin_amdgcn_exp2f(x[reg] - mx); l += x[reg]; }
            l += __shfl_xor(l, 32, 64);
            const float inv = (r < nvalid) ? 1.0f / l : 0.f;
            float delta = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) { x[reg] *= inv; delta += x[reg] * dpt[reg]; }
            delta += __shfl_xor(delta, 32, 64);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) x[reg] *= dpt[reg] - delta;
            if (hh == 0) { sLse[r] = mx + __log2f(l); sDel[r] = delta; }
            lds_fence();
            const bf16x8_t d0 = pack8(x), d1 = pack8(x + 8);
            bf16_t* op = a.dQ + row * a.lddqkv + h * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {                       // the transposed LDS reads need EXEC all ones: only the stores are masked
                f32x16_t dq = MFMA32(tr_frag_p<DP>(sK, dt, 0, hh, r), d0, zero16());
                dq = MFMA32(tr_frag_p<DP>(sK, dt, 1, hh, r), d1, dq);
                store_tile32(op + 32 * dt, dq, a.scale, hh, r < nvalid);
            }
        }
        {   // phase B: key on the lane
            f32x16_t sc = zero16(), dp = zero16();
#pragma unroll
            for (int i = 0; i < KS; ++i) { sc = MFMA32(q[i], k[i], sc); dp = MFMA32(d[i], v[i], dp); }
            float pr[16], ds[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 ls = *reinterpret_cast<const float4*>(sLse + 8 * g4 + 4 * hh);
                const float4 de = *reinterpret_cast<const float4*>(sDel + 8 * g4 + 4 * hh);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int reg = 4 * g4 + c;
                    const bool ok = ((okm >> reg) & 1u) && 8 * g4 + 4 * hh + c < nvalid;
                    const float pv = ok ? __builtin_amdgcn_exp2f(sc[reg] * a.scale2 - pick(ls, c)) : 0.f;
                    pr[reg] = pv;
                    ds[reg] = pv * (dp[reg] - pick(de, c));
                }
            }
            const bf16x8_t p0 = pack8(pr), p1 = pack8(pr + 8), e0 = pack8(ds), e1 = pack8(ds + 8);
            bf16_t* kp = a.dK + row * a.lddqkv + h * D;
            bf16_t* vp = a.dV + row * a.lddqkv + h * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x16_t dv = MFMA32(tr_frag_p<DP>(sD, dt, 0, hh, r), p0, zero16());
                dv = MFMA32(tr_frag_p<DP>(sD, dt, 1, hh, r), p1, dv);
                f32x16_t dk = MFMA32(tr_frag_p<DP>(sQ, dt, 0, hh, r), e0, zero16());
                dk = MFMA32(tr_frag_p<DP>(sQ, dt, 1, hh, r), e1, dk);
                store_tile32(kp + 32 * dt, dk, a.scale, hh, r < nvalid);
                store_tile32(vp + 32 * dt, dv, 1.0f, hh, r < nvalid);
            }
        }
        lds_fence();
    }
}

// ------------------------------------------------------------------------------------------------ additive table
// bm[mh][q][k] = log2(e) * bias[mh][q % T][k % T] when q and k are frames of the same packed sequence, else -1e30 (other
// sequences of the tile, padding rows);  bmT[mh][k][q] = bm[mh][q][k].
__global__ void tattn_table_kernel(const float* bias, float* bm, float* bmT, int nmH, int T, int per) {
    const int total = nmH * 1024;
    const int nv = per * T;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        const int k = id & 31, q = (id >> 5) & 31, mh = id >> 10;
        float v = NEG_BIG;
        if (q < nv && k < nv && q / T == k / T) v = bias[((int64_t)mh * T + q % T) * T + k % T] * LOG2E;
        bm[(int64_t)mh * 1024 + q * 32 + k] = v;
        bmT[(int64_t)mh * 1024 + k * 32 + q] = v;
    }
}

int fill(const stg_tattn_args* f, TP& p, const char* who) {
    STG_CHECK(f->Q && f->K && f->V, -1, "%s: null pointer", who);
    STG_CHECK(f->D == 32 || f->D == 64 || f->D == 96, -2, "%s: head dim must be 32, 64 or 96", who);
    if (f->D == TD) STG_CHECK(f->bm && f->bmT, -1, "%s: head dim 32 needs the bm / bmT workspaces", who);
    else STG_CHECK(f->bias == nullptr, -2, "%s: head dims 64 / 96 run without an additive bias (nn.MultiheadAttention, CLIP_AVE.py:106-108)", who);
    STG_CHECK(f->T >= 1 && f->T <= 32, -2, "%s: T must be in [1, 32]", who);
    STG_CHECK(f->nm >= 1 && f->nm <= 65535 && f->B >= 0 && f->N >= 1 && f->H >= 1, -2, "%s: bad shape", who);
    STG_CHECK(f->ld % 8 == 0 && (((uintptr_t)f->Q | (uintptr_t)f->K | (uintptr_t)f->V) & 15) == 0, -2, "%s: misaligned qkv", who);
    STG_CHECK((int64_t)f->nm * f->B * f->T * f->N < (1ll << 40), -2, "%s: too many rows", who);
    p.Q = (const bf16_t*)f->Q; p.K = (const bf16_t*)f->K; p.V = (const bf16_t*)f->V; p.ld = f->ld;
    p.O = (bf16_t*)f->O; p.ldo = f->ldo; p.bm = f->bm; p.bmT = f->bmT;
    p.nm = f->nm; p.B = (int)f->B; p.T = f->T; p.N = f->N; p.H = f->H;
    p.per = 32 / f->T;
    p.gpb = (f->N + p.per - 1) / p.per;
    STG_CHECK((int64_t)f->B * p.gpb < (1ll << 31), -2, "%s: too many token groups", who);
    p.ngroups = (int)(f->B * p.gpb);
    p.scale = f->scale; p.scale2 = f->scale * LOG2E;
    return 0;
}

constexpr int nb_lds_bytes(int D) { return 4 * (3 * 32 * (D + 8) + 128) * 2; }

dim3 grid_for(const TP& p) {
    const int hg = (p.H + 3) / 4;
    int gx = 2048 / (hg * p.nm);                // ~8 workgroups per CU over the whole grid: bounds the dbias atomics
    if (gx < 1) gx = 1;
    if (gx > p.ngroups) gx = p.ngroups;
    return dim3(gx, hg, p.nm);
}

}  // namespace

extern "C" int stg_tattn_fwd(const stg_tattn_args* f, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_tattn_fwd: null args");
    TP p = {};
    int rc = fill(f, p, "stg_tattn_fwd");
    if (rc) return rc;
    STG_CHECK(f->O && f->ldo % 8 == 0 && (((uintptr_t)f->O) & 15) == 0, -2, "stg_tattn_fwd: bad O (16-byte stores)");
    if (p.ngroups == 0) return 0;
    if (f->D != TD) {
        if (f->D == 64) hipLaunchKernelGGL(tattn_nb_fwd_kernel<64>, grid_for(p), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(tattn_nb_fwd_kernel<96>, grid_for(p), dim3(256), 0, (hipStream_t)stream, p);
        STG_LAUNCH_CHECK();
        return 0;
    }
    STG_CHECK(f->bias != nullptr, -1, "stg_tattn_fwd: head dim 32 needs the bias");
    const int total = p.nm * p.H * 1024;
    hipLaunchKernelGGL(tattn_table_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, f->bias, f->bm, f->bmT, p.nm * p.H, p.T, p.per);
    STG_LAUNCH_CHECK();
    hipLaunchKernelGGL(tattn_fwd1_kernel, grid_for(p), dim3(256), 0, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_tattn_bwd(const stg_tattn_args* f, const void* dO, int64_t lddo, void* dQ, void* dK, void* dV,
                             int64_t lddqkv, float* dbias, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_tattn_bwd: null args");
    TP p = {};
    int rc = fill(f, p, "stg_tattn_bwd");
    if (rc) return rc;
    STG_CHECK(dO && dQ && dK && dV, -1, "stg_tattn_bwd: null pointer");
    STG_CHECK(lddo % 8 == 0 && lddqkv % 8 == 0, -2, "stg_tattn_bwd: bad leading dims");
    STG_CHECK((((uintptr_t)dO) & 15) == 0 && (((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15) == 0, -2,
              "stg_tattn_bwd: misaligned pointers");
    if (p.ngroups == 0) return 0;
    p.dO = (const bf16_t*)dO; p.lddo = lddo; p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV; p.lddqkv = lddqkv;
    p.dbias = dbias;
    if (f->D != TD) {
        STG_CHECK(dbias == nullptr, -2, "stg_tattn_bwd: no bias gradient without a bias");
        const int lds = f->D == 64 ? nb_lds_bytes(64) : nb_lds_bytes(96);
        static std::atomic<uint64_t> d64{0}, d96{0};
        const bool attr_set = stg_reserve_lds(tattn_nb_bwd_kernel<64>, nb_lds_bytes(64), d64) && stg_reserve_lds(tattn_nb_bwd_kernel<96>, nb_lds_bytes(96), d96);
        STG_CHECK(attr_set, -101, "stg_tattn_bwd: cannot reserve %d bytes of LDS", lds);
        if (f->D == 64) hipLaunchKernelGGL(tattn_nb_bwd_kernel<64>, grid_for(p), dim3(256), lds, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(tattn_nb_bwd_kernel<96>, grid_for(p), dim3(256), lds, (hipStream_t)stream, p);
        STG_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(tattn_bwd1_kernel, grid_for(p), dim3(256), 0, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}
