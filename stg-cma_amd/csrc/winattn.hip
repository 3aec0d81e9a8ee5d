// Whole-window attention for Swin W-MSA / SW-MSA (window <= 64 tokens, head dim 32): ONE WAVE computes one (window, head)
// problem end to end -- forward, and a single backward kernel that produces dQ, dK and dV together.
//
// Why a second attention family next to attention.hip: for 49-token windows the generic flash kernels spend their time in
// VALU bookkeeping (two query-tile waves x two key tiles per problem, online-softmax rescaling, per-element bias and mask
// loads, and a backward split into two kernels that each recompute S and dP).  Here the 64 x 64 padded score block lives in
// registers at once (single-pass softmax), bias + shift mask + key padding come from ONE pre-added, padded table read with
// 16-byte loads, every operand load of the problem is issued up front, and the backward computes both score orientations
// in the same wave (query-on-lane for dQ, key-on-lane for dK/dV) from the SAME operand registers: an A-operand fragment
// with rows = tokens is bit-identical to a B-operand fragment with columns = tokens.
//
// MFMA v_mfma_f32_32x32x16_bf16, lane l = (r = l & 31, hh = l >> 5); accumulator row (reg, hh) = (reg&3) + 8*(reg>>2) + 4*hh.
// Addressing: token i of window w of frame f lives at row f*outer + tok(w, i) with the cyclic shift + partition computed
// arithmetically (Swin_AVE.py:727-740; the inverse scatter :765-776 is the same map).
#include <math.h>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr int WD = 32;             // head dim
constexpr float NEG_BIG = -1.0e30f;

struct WinP {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; int64_t ld;   // fused qkv rows: Q/K/V at base + row*ld + h*32
    bf16_t* O; int64_t ldo;
    float* lse;                    // [P, H, 64]
    const float* bm;               // [Gt, H, 64(q), 64(key)]  bias + mask, key >= n -> -1e30
    const float* bmT;              // [Gt, H, 64(key), 64(q)]
    int Gt;                        // tables per image: nW (shifted blocks) or 1
    int64_t outer;                 // rows per image
    int Himg, Wimg, ws, shift, nww, G, n;
    uint32_t ws_magic, nww_magic;  // ceil(2^20 / ws), ceil(2^20 / nww): x / d == (x * magic) >> 20 for x * d < 2^20
    int P, H;
    float scale, scale2;           // scale2 = scale * log2(e): the tables are pre-multiplied by log2(e), softmax runs on exp2
    // backward
    const bf16_t* dO; int64_t lddo;
    bf16_t* dQ; bf16_t* dK; bf16_t* dV; int64_t lddqkv;
    int total;
    // the adapters' window-level cross-modal pair with its gate (round 6b, stg_winattn_pair_*): forward X = Q + gate[0] * O beside O;
    // backward dO is d(X), every output is scaled by gate[0] (the pass is linear in dO) and dgate += sum_q delta[q] = <d(X), O>
    const float* gate; bf16_t* X; int64_t ldx; float* dgate;
};
struct WinP2 { WinP a[2]; };

__device__ __forceinline__ void unpack8f(const uint4& q, float* v) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8u(const float* t) {
    return make_uint4(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]), pack_bf2(t[4], t[5]), pack_bf2(t[6], t[7]));
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
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8_t pack8(const float* x) {
    const u32x4_t w = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
    return __builtin_bit_cast(bf16x8_t, w);
}
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// row of token i (clamped to the last valid token) of window g of image pg
__device__ __forceinline__ int64_t tok_row(const WinP& a, int pg, int g, int i) {
    i = i < a.n ? i : a.n - 1;
    const int wi = (int)(((uint32_t)g * a.nww_magic) >> 20), wj = g - wi * a.nww;
    const int ti = (int)(((uint32_t)i * a.ws_magic) >> 20), tj = i - ti * a.ws;
    int h = wi * a.ws + ti + a.shift;
    h = h >= a.Himg ? h - a.Himg : h;
    int w = wj * a.ws + tj + a.shift;
    w = w >= a.Wimg ? w - a.Wimg : w;
    return (int64_t)pg * a.outer + h * a.Wimg + w;
}

typedef short s4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ backward, one pass (round 2)
// Two things the two-orientation kernel above pays for, removed:
//  (1) It recomputes S, dP and the exponentials for the key-on-lane products (dK, dV): 16 of its 56 MFMAs, half of its 128
//      exponentials per lane, a second additive table (bmT) and the lse / delta broadcast.  Here each (key tile, q tile) pair is
//      evaluated ONCE, query on the lane; P and dS go to dQ straight from the accumulators (B operand, as above) and, as two 32 x 32
//      bf16 tiles T[q][key] through the wave's own LDS, come back TRANSPOSED (ds_read_b64_tr_b16) as the B operands of
//          dV^T[d][key] += dO^T[d][q] P[q][key],        dK^T[d][key] += Q^T[d][q] dS[q][key].
//  (2) Its operand fragments and its stores are ROW-PER-LANE accesses (a lane reads / writes 16 bytes of its own token row): 64
//      separate requests per wave-instruction, ~64 clocks each in the CU's address path -- 30 such instructions per problem, with 8
//      waves sharing the path, were more than half of the kernel's time.  Here every global access is coalesced (4 lanes cover the
//      64 bytes of a head's row): Q, K, V, dO, O are read once, 16 rows per instruction, into swizzled LDS tiles (O only into the
//      delta dot product), the MFMA operand fragments are read from those tiles, and dQ / dK / dV leave through a 32 x 32 LDS
//      transposition.
// LDS tiles [64 tokens][32] bf16: 16-byte chunk c of row R sits at chunk c ^ ((R >> 2) & 3) -- conflict-free for the row-fragment
// ds_read_b128 (its 16-lane groups mix four values of R >> 2), for the staging writes and for the transposed reads (the four rows of
// a block share R >> 2).  P / dS / output tiles [32][32]: 8-byte piece u of row q at u ^ ((q >> 2) & 7).
// Padded queries carry lse = +1e30 (P = dS = 0), padded keys the table's -1e30; padded rows of the tiles duplicate token n - 1.
// Score register `reg` of key tile kt holds keys 32 kt + (reg & 3) + 8 (reg >> 2) + 4 hh.  With NKEY > 0 (a compile-time key count: 49 for the 7 x 7
// windows of every Swin stage) a register whose keys are padding for BOTH half-waves is skipped at compile time: its table entry is -1e30, so its
// probability is exactly 0 either way (bit-identical results), and 7 of the 32 score registers of a query tile cost no VALU / exp.
// Round 6: the window-level CROSS-MODAL pair of the adapters (softmax(h_v h_a^T) h_a per window, Swin_AVE.py:750-760) on these kernels without its
// dummy table and at width 16 (Swin-B stage 0, the last backbone site on the generic attention kernels):
//   NOTAB  no bias, no shift mask: the additive term is 0 (-1e30 on the padding keys), synthesised from the key index -- no 16 KiB table fetch;
//   D16    head dim 16: rows of 32 bytes in global memory, staged into the SAME [64][32] LDS tiles with the upper 16 columns read from a zero
//          line (the swizzled source chunk cs >= 2 of every row), so every fragment, MFMA and transposition below is unchanged -- the products' second
//          k-step multiplies zeros -- and only the lower two 16-byte chunks of an output row are stored.  (Zero-padded COPIES of the operands were
//          measured in round 5b: the attention family -1.25 ms per step, the copies +0.65.)
__device__ __attribute__((aligned(16))) const uint32_t win_zero16[4] = {0u, 0u, 0u, 0u};
template <int NKEY>
__device__ __forceinline__ float4 notab_add(int kt, int g4, int hh) {       // keys 32 kt + 8 g4 + 4 hh + 0..3
    const int k0 = 32 * kt + 8 * g4 + 4 * hh;
    return make_float4(k0 >= NKEY ? NEG_BIG : 0.f, k0 + 1 >= NKEY ? NEG_BIG : 0.f, k0 + 2 >= NKEY ? NEG_BIG : 0.f, k0 + 3 >= NKEY ? NEG_BIG : 0.f);
}

template <int NKEY>
__device__ __forceinline__ constexpr bool key_reg_live(int kt, int reg) { return NKEY <= 0 || 32 * kt + (reg & 3) + 8 * (reg >> 2) < NKEY; }

__device__ __forceinline__ int sw_off(int row, int chunk) { return row * WD + ((chunk ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ bf16x8_t tr_frag64(const bf16_t* s, int kt, int s2, int hh, int d) {     // tr_frag on a swizzled [64][32] tile
    const int gi = d & 15, c = d >> 4;
    const int row = 32 * kt + 16 * s2 + 4 * hh + (gi >> 2), u = 4 * c + (gi & 3);
    const bf16_t* p0 = s + sw_off(row, u >> 1) + ((u & 1) << 2);
    const bf16_t* p1 = s + sw_off(row + 8, u >> 1) + ((u & 1) << 2);
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p0);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p1);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
__device__ __forceinline__ bf16x8_t tr_frag32(const bf16_t* s, int s2, int hh, int d) {              // on a [32][32] piece-swizzled tile
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
// accumulator tile (lane = token r, rows = head dims) -> [32 tokens][32 dims] bf16 piece-swizzled LDS tile
__device__ __forceinline__ void put_tile32(bf16_t* T, const f32x16_t& acc, float sc, int r, int hh) {
    const int swz = (r >> 2) & 7;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<uint2*>(T + r * 32 + (((2 * g4 + hh) ^ swz) << 2)) =
            make_uint2(pack_bf2(acc[4 * g4] * sc, acc[4 * g4 + 1] * sc), pack_bf2(acc[4 * g4 + 2] * sc, acc[4 * g4 + 3] * sc));
}
// ... and out of it, 16 token rows x 64 bytes per store instruction; rowoff[j]: element offset of token 32 t + (lane >> 2) + 16 j
template <bool D16 = false>
__device__ __forceinline__ void flush_tile32(const bf16_t* T, bf16_t* dst, const int64_t (&rowoff)[2], int t, int n, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tl = (lane >> 2) + 16 * j, cq = lane & 3, sw = (tl >> 2) & 7;
        const uint2 lo = *reinterpret_cast<const uint2*>(T + tl * 32 + (((2 * cq) ^ sw) << 2));
        const uint2 hi = *reinterpret_cast<const uint2*>(T + tl * 32 + (((2 * cq + 1) ^ sw) << 2));
        if (32 * t + tl < n && (!D16 || cq < 2)) *reinterpret_cast<uint4*>(dst + rowoff[j] + cq * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}


// flush_tile32 of an O tile plus the gated residual X = Q + g O (stg_gate_fwd2's arithmetic on the bf16-rounded O): Q's 16 bytes come back from L2
template <bool D16 = false>
__device__ __forceinline__ void flush_tile32_gate(const bf16_t* T, bf16_t* dst, const int64_t (&rowoff)[2], const bf16_t* Q, const int64_t (&rowq)[2], bf16_t* X,
                                                   const int64_t (&rowx)[2], float g, int t, int n, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tl = (lane >> 2) + 16 * j, cq = lane & 3, sw = (tl >> 2) & 7;
        const uint2 lo = *reinterpret_cast<const uint2*>(T + tl * 32 + (((2 * cq) ^ sw) << 2));
        const uint2 hi = *reinterpret_cast<const uint2*>(T + tl * 32 + (((2 * cq + 1) ^ sw) << 2));
        if (32 * t + tl < n && (!D16 || cq < 2)) {
            const uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
            *reinterpret_cast<uint4*>(dst + rowoff[j] + cq * 8) = o;
            const uint4 q = *reinterpret_cast<const uint4*>(Q + rowq[j] + cq * 8);
            float qv[8], ov[8];
            unpack8f(q, qv); unpack8f(o, ov);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] = qv[e] + g * ov[e];
            *reinterpret_cast<uint4*>(X + rowx[j] + cq * 8) = pack8u(qv);
        }
    }
}

// ---- round 6: the additive table shared through LDS (template flag LT, used with NKEY = 49) --------------------------------------------------
// The table is 16 KiB per (window type, head) -- more bytes through a CU's vector-memory path than the q / k / v / o tiles of a problem
// (12.5 KiB), every wave fetching its own copy (L2 hits; a no-table probe ran the forward 10-22 % faster).  In the LT form a workgroup is
// FOUR FRAMES of one (window position g, head h): the four problems share one table, which the workgroup stages ONCE by LDS-DMA (key groups
// below NKEY only: 13 x 1 KiB) and reads with conflict-free ds_read_b128 (a lane's float4s are consecutive in x = query).  A wave still moves
// 64-byte pieces of its head per token row; the heads of one (frame quad, window) are CONSECUTIVE workgroups of ONE XCD (blockIdx -> XCD is
// round-robin), so the other half of every 128-byte line is wanted by the neighbour in that XCD's L2 at the same time.
//     unit u = (frame quad fq, window g), g fastest; XCD x takes units x, x + 8, ...;  block b: x = b & 7, idx = b >> 3, h = idx % H,
//     u = (idx / H) * 8 + x;  wave w of the block: frame 4 fq + w.
// Same arithmetic on the same table values as the per-wave form: bit-identical results.
template <int NKEY> struct LtTab { static constexpr int GROUPS = (NKEY + 3) / 4; static constexpr int FLOATS = GROUPS > 0 ? GROUPS * 256 : 4; };

struct LtItem { int p, pg, g, h; bool block_live, wave_live; };
__device__ __forceinline__ LtItem lt_item(const WinP& a, int wave) {
    LtItem it;
    const int F = a.P / a.G, nq = (F + 3) >> 2, U = nq * a.G;
    const int b = blockIdx.x, x = b & 7, idx = b >> 3;
    it.h = idx % a.H;
    const int u = (idx / a.H) * 8 + x;
    it.block_live = u < U;
    const int fq = u / a.G;
    it.g = u - fq * a.G;
    it.pg = 4 * fq + wave;
    it.wave_live = it.block_live && it.pg < F;
    it.p = it.pg * a.G + it.g;
    return it;
}
template <int NKEY>
__device__ __forceinline__ void lt_stage_table(const WinP& a, float* stab, int g, int h, int wave, int lane) {
    const float* src = a.bm + ((int64_t)(g % a.Gt) * a.H + h) * 4096 + lane * 4;
#pragma unroll
    for (int i = 0; i < (LtTab<NKEY>::GROUPS + 3) / 4; ++i) {
        const int grp = 4 * i + wave;                     // wave-uniform
        if (grp < LtTab<NKEY>::GROUPS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + grp * 256),
                                             (__attribute__((address_space(3))) void*)(stab + grp * 256), 16, 0, 0);
    }
}
// float4 of keys 32 kt + 8 g4 + 4 hh + 0..3 for query x: group 8 kt + 2 g4 + hh; a group past the staged ones is all padding keys (-1e30)
template <int NKEY>
__device__ __forceinline__ float4 lt_read(const float* stab, int kt, int g4, int hh, int x) {
    const int grp0 = 8 * kt + 2 * g4;                     // compile-time after unrolling
    if (grp0 + 1 < LtTab<NKEY>::GROUPS) return *reinterpret_cast<const float4*>(stab + ((grp0 + hh) * 64 + x) * 4);
    const float4 v = *reinterpret_cast<const float4*>(stab + (grp0 * 64 + x) * 4);
    return hh ? make_float4(NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG) : v;
}

template <int NKEY, bool LT, bool NOTAB = false, bool D16 = false>
__device__ __forceinline__ void winattn_bwd1_body(const WinP& a) {
    constexpr int HW = D16 ? 16 : WD;                      // head width in global memory
    // per wave: K, Q, dO tiles ([64][32] bf16), one more tile (V while the fragments are fetched, then the P and dS tiles of the current
    // pair, then the output transpositions), delta[64]
    constexpr int PER_WAVE = 4 * 64 * WD + 128;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * PER_WAVE];
    __shared__ __attribute__((aligned(16))) float stab[LT ? LtTab<NKEY>::FLOATS : 4];
    __shared__ float sgate[2];                             // the gated pair form: the workgroup's dgate sum and its ticket counter
    if (a.dgate) {                                         // kernel-uniform; in front of every early exit
        if (threadIdx.x == 0) { sgate[0] = 0.f; reinterpret_cast<int*>(sgate)[1] = 0; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const bf16_t *Qp = a.Q, *Kp = a.K, *Vp = a.V, *Op = a.O, *dOp = a.dO;
    const float* lsep = a.lse;
    bf16_t* dQp = a.dQ;
    bf16_t* dKp = a.dK;
    int h, p, pg, g;
    bool live = true;
    if (LT) {
        const LtItem it = lt_item(a, wave);
        if (!it.block_live) return;                        // whole workgroup
        h = it.h; p = it.p; pg = it.pg; g = it.g; live = it.wave_live;
        lt_stage_table<NKEY>(a, stab, g, h, wave, lane);
    } else {
        const int item = blockIdx.x * 4 + wave;
        if (item >= a.total) return;
        h = item % a.H;
        p = item / a.H;
        pg = p / a.G; g = p - pg * a.G;
    }
    if (LT && !live) {                                     // a frame past the last one: the wave only helped to stage the table
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        return;
    }
    bf16_t* sK = smem + wave * PER_WAVE;
    bf16_t* sQ = sK + 64 * WD;
    bf16_t* sD = sQ + 64 * WD;
    bf16_t* sP = sD + 64 * WD;
    bf16_t* sS = sP + 32 * 32;
    float* sDel = reinterpret_cast<float*>(sP + 64 * WD);

    // ---- coalesced staging by LDS-DMA: instruction i of a tile fills rows 16 i .. + 15 (1 KiB), lane -> row 16 i + (lane >> 2), LDS
    // chunk lane & 3, which holds SOURCE chunk (lane & 3) ^ ((row >> 2) & 3) (the swizzle lives on the source address; (row >> 2) & 3
    // = (lane >> 4) & 3 for every i).  No registers, nothing for the compiler to serialise: 16 DMAs + 6 loads in flight at once.
    const int cs = (lane & 3) ^ ((lane >> 4) & 3);
    int64_t trow[4];
    uint4 vo[4];
    const bool zc = D16 && cs >= 2;                        // D16: the upper two chunks of every LDS row come from the zero line
    const bf16_t* zl = reinterpret_cast<const bf16_t*>(win_zero16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        trow[i] = tok_row(a, pg, g, (lane >> 2) + 16 * i);
        const int64_t off = trow[i] * a.ld + h * HW + cs * 8;
        const bf16_t* sk = zc ? zl : Kp + off;
        const bf16_t* sq = zc ? zl : Qp + off;
        const bf16_t* sv = zc ? zl : Vp + off;
        const bf16_t* sd = zc ? zl : dOp + trow[i] * a.lddo + h * HW + cs * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sk, (__attribute__((address_space(3))) void*)(sK + i * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq, (__attribute__((address_space(3))) void*)(sQ + i * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sv, (__attribute__((address_space(3))) void*)(sP + i * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sd, (__attribute__((address_space(3))) void*)(sD + i * 512), 16, 0, 0);
        vo[i] = zc ? make_uint4(0u, 0u, 0u, 0u) : *reinterpret_cast<const uint4*>(Op + trow[i] * a.ldo + h * HW + cs * 8);
    }
    float lse_q[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = 32 * t + r;
        lse_q[t] = q < a.n ? lsep[((int64_t)p * a.H + h) * 64 + q] * 1.4426950408889634f : 1.0e30f;
    }
    const int64_t tb = ((int64_t)(g % a.Gt) * a.H + h) * 64 * 64;
    // the additive table of a (key tile, q tile) pair is fetched one pair ahead (L2 hits, but ~1 us each when waited for in place);
    // the fences around the P / dS tiles keep the compiler from sinking the loads back to their use
    float4 addc[4], addn[4];
    auto load_add = [&](float4 (&dst)[4], int kt, int qt) {
        const float* bmq = a.bm + tb + 4 * (32 * qt + (32 * qt + r < a.n ? r : 0));       // (padded queries: lane 0's address; see winattn_fwd1_kernel)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (NKEY > 0 && 32 * kt + 8 * g4 >= NKEY) { dst[g4] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
            if (NOTAB) dst[g4] = notab_add<NKEY>(kt, g4, hh);
            else if (LT) dst[g4] = lt_read<NKEY>(stab, kt, g4, hh, 32 * qt + r);
            else dst[g4] = *reinterpret_cast<const float4*>(bmq + 256 * ((kt * 4 + g4) * 2 + hh));
        }
    };
    if (!LT) load_add(addc, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (LT) { __syncthreads(); load_add(addc, 0, 0); }     // the table's pieces were staged by all four waves
    lds_fence();
    // delta[q] = sum_d dO[q][d] O[q][d]: the lane's own 16-byte piece of dO (back from LDS) times the same piece of O, summed over the
    // four lanes of the row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 vd = *reinterpret_cast<const uint4*>(sD + i * 512 + lane * 8);
        const uint32_t dw[4] = {vd.x, vd.y, vd.z, vd.w}, ow[4] = {vo[i].x, vo[i].y, vo[i].z, vo[i].w};
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            d += __uint_as_float(dw[j] << 16) * __uint_as_float(ow[j] << 16) + __uint_as_float(dw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if ((lane & 3) == 0) sDel[(lane >> 2) + 16 * i] = d;
    }
    lds_fence();
    bf16x8_t vf[2][2];
    float delta[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        vf[t][0] = ld_frag(sP + sw_off(32 * t + r, hh)); vf[t][1] = ld_frag(sP + sw_off(32 * t + r, hh + 2));
        delta[t] = sDel[32 * t + r];
    }
    const float gsc = a.gate ? a.gate[0] : 1.0f;           // kernel-uniform: the pair's gate (dO is d(X) = d(Q + gate O))
    if (a.dgate) {                                         // dgate += <d(X), O> over the window's real queries
        // ONE global atomic per workgroup: memory-side atomics on one address serialise at ~5 ns each (20 000 per-wave atomics of a stage-0 launch were
        // 0.1 ms, more than the gate kernel this replaces).  The waves meet through LDS: each adds its sum, the last one to take a ticket sends the total
        // (a wave's two DS operations execute in order, so every sum is in before its ticket is counted).
        float dg = hh == 0 ? (r < a.n ? delta[0] : 0.f) + (32 + r < a.n ? delta[1] : 0.f) : 0.f;
        dg = wave_sum<64>(dg);
        if (lane == 0) {
            const int nlive = a.total - (int)blockIdx.x * 4 < 4 ? a.total - (int)blockIdx.x * 4 : 4;
            atomicAdd(&sgate[0], dg);
            const int ticket = atomicAdd(reinterpret_cast<int*>(&sgate[1]), 1);
            if (ticket == nlive - 1) atomicAdd(a.dgate, atomicAdd(&sgate[0], 0.f));
        }
    }
    lds_fence();                                           // the V tile becomes the P / dS tiles
    const int swz = (r >> 2) & 7;

    f32x16_t dq[2] = {zero16(), zero16()};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        f32x16_t dv = zero16(), dk = zero16();
        const bf16x8_t kf0 = ld_frag(sK + sw_off(32 * kt + r, hh)), kf1 = ld_frag(sK + sw_off(32 * kt + r, hh + 2));
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (kt * 2 + qt < 3) load_add(addn, (kt * 2 + qt + 1) >> 1, (kt * 2 + qt + 1) & 1);
            const bf16x8_t qf0 = ld_frag(sQ + sw_off(32 * qt + r, hh)), qf1 = ld_frag(sQ + sw_off(32 * qt + r, hh + 2));
            const bf16x8_t df0 = ld_frag(sD + sw_off(32 * qt + r, hh)), df1 = ld_frag(sD + sw_off(32 * qt + r, hh + 2));
            f32x16_t st = zero16(), dpt = zero16();
            st = MFMA32(kf0, qf0, st);
            dpt = MFMA32(vf[kt][0], df0, dpt);
            st = MFMA32(kf1, qf1, st);
            dpt = MFMA32(vf[kt][1], df1, dpt);
            float pr[16], ds[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float4 ad = addc[reg >> 2];
                const float av = (reg & 3) == 0 ? ad.x : (reg & 3) == 1 ? ad.y : (reg & 3) == 2 ? ad.z : ad.w;
                if (key_reg_live<NKEY>(kt, reg)) {
                    pr[reg] = __builtin_amdgcn_exp2f(st[reg] * a.scale2 + (av - lse_q[qt]));  // padded keys: av = -1e30; padded queries: lse = +1e30
                    ds[reg] = pr[reg] * (dpt[reg] - delta[qt]);
                } else {
                    pr[reg] = 0.f; ds[reg] = 0.f;
                }
            }
            bf16x8_t pp[2], ps[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { pp[s2] = pack8(pr + 8 * s2); ps[s2] = pack8(ds + 8 * s2); }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) dq[qt] = MFMA32(tr_frag64(sK, kt, s2, hh, r), ps[s2], dq[qt]);
            // T[q = r][keys 8 g4 + 4 hh .. + 3], g4 = 2 s2 + half: 8-byte piece u = 2 g4 + hh, stored at u ^ swz
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4_t wp = __builtin_bit_cast(u32x4_t, pp[s2]), ws = __builtin_bit_cast(u32x4_t, ps[s2]);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int u = (2 * (2 * s2 + half) + hh) ^ swz;
                    *reinterpret_cast<uint2*>(sP + r * 32 + u * 4) = make_uint2(wp[2 * half], wp[2 * half + 1]);
                    *reinterpret_cast<uint2*>(sS + r * 32 + u * 4) = make_uint2(ws[2 * half], ws[2 * half + 1]);
                }
            }
            lds_fence();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                dv = MFMA32(tr_frag64(sD, qt, s2, hh, r), tr_frag32(sP, s2, hh, r), dv);
                dk = MFMA32(tr_frag64(sQ, qt, s2, hh, r), tr_frag32(sS, s2, hh, r), dk);
            }
            lds_fence();                                   // the tiles are rewritten by the next pair
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) addc[g4] = addn[g4];
        }
        const int64_t ro[2] = {trow[2 * kt] * a.lddqkv + h * HW, trow[2 * kt + 1] * a.lddqkv + h * HW};
        if (a.dV == nullptr) {                             // K == V (the adapters' cross-modal attention): one gradient, dK + dV
#pragma unroll
            for (int i = 0; i < 16; ++i) dk[i] = fmaf(dk[i], a.scale, dv[i]);
            put_tile32(sP, dk, gsc, r, hh);
            lds_fence();
            flush_tile32<D16>(sP, dKp, ro, kt, a.n, lane);
        } else {
            put_tile32(sP, dk, a.scale * gsc, r, hh);
            put_tile32(sS, dv, gsc, r, hh);
            lds_fence();
            flush_tile32<D16>(sP, dKp, ro, kt, a.n, lane);
            flush_tile32<D16>(sS, a.dV, ro, kt, a.n, lane);
        }
        lds_fence();
    }
    put_tile32(sP, dq[0], a.scale * gsc, r, hh);
    put_tile32(sS, dq[1], a.scale * gsc, r, hh);
    lds_fence();
    {
        const int64_t r0[2] = {trow[0] * a.lddqkv + h * HW, trow[1] * a.lddqkv + h * HW};
        const int64_t r1[2] = {trow[2] * a.lddqkv + h * HW, trow[3] * a.lddqkv + h * HW};
        flush_tile32<D16>(sP, dQp, r0, 0, a.n, lane);
        flush_tile32<D16>(sS, dQp, r1, 1, a.n, lane);
    }
}

template <int NKEY, bool LT, bool NOTAB = false, bool D16 = false>
__global__ void __launch_bounds__(256, 2) winattn_bwd1_kernel(WinP a) { winattn_bwd1_body<NKEY, LT, NOTAB, D16>(a); }
// both directions of a cross-modal pair in one launch (blockIdx.y = direction; table-free form)
template <bool D16>
__global__ void __launch_bounds__(256, 2) winattn_bwd1_pair_kernel(WinP2 pp) { winattn_bwd1_body<49, false, true, D16>(pp.a[blockIdx.y]); }

// ------------------------------------------------------------------------------------------------ forward, coalesced (round 2)
// winattn_fwd_kernel with every global access coalesced, like winattn_bwd1_kernel: Q, K, V by LDS-DMA into swizzled tiles, operand
// fragments from LDS, O through a 32 x 32 LDS transposition (the Q / K tiles are dead once the scores exist).
template <int NKEY, bool NOTAB = false, bool D16 = false>
__device__ __forceinline__ void winattn_fwd1_body(const WinP& a) {
    constexpr int HW = D16 ? 16 : WD;                      // head width in global memory
    constexpr int PER_WAVE = 3 * 64 * WD;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * PER_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    int item = blockIdx.x * 4 + wave;
    const bf16_t *Qp = a.Q, *Kp = a.K, *Vp = a.V;
    bf16_t* Op = a.O;
    float* lsep = a.lse;
    if (item >= a.total) return;
    const int h = item % a.H;
    const int p = item / a.H;
    const int pg = p / a.G, g = p - pg * a.G;
    bf16_t* sQ = smem + wave * PER_WAVE;
    bf16_t* sK = sQ + 64 * WD;
    bf16_t* sV = sK + 64 * WD;

    const int cs = (lane & 3) ^ ((lane >> 4) & 3);
    int64_t trow[4];
    const bool zc = D16 && cs >= 2;                        // D16: the upper two chunks of every LDS row come from the zero line
    const bf16_t* zl = reinterpret_cast<const bf16_t*>(win_zero16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        trow[i] = tok_row(a, pg, g, (lane >> 2) + 16 * i);
        const int64_t off = trow[i] * a.ld + h * HW + cs * 8;
        const bf16_t* sq = zc ? zl : Qp + off;
        const bf16_t* sk = zc ? zl : Kp + off;
        const bf16_t* sv = zc ? zl : Vp + off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq, (__attribute__((address_space(3))) void*)(sQ + i * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sk, (__attribute__((address_space(3))) void*)(sK + i * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sv, (__attribute__((address_space(3))) void*)(sV + i * 512), 16, 0, 0);
    }
    // The additive table is 16 KiB per (window, head) -- more bytes through the CU's vector-memory path than the q / k / v / o tiles (12.5 KiB), all
    // L2 hits; with the loads removed (wrong results: a probe) the forward ran 10-22 % faster.  Round 5b: key groups entirely past NKEY are not
    // fetched and the lanes of padded queries read lane 0's address (-1..-2 %).  Tried and not kept: one wave taking the same window position of 2 / 4
    // consecutive frames with ONE table fetch (0 .. -8 % at stage 0, +2 .. +5 % at the others: the frames of a wave then run one after the other).
    const float* bmq[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
        bmq[t] = a.bm + ((int64_t)(g % a.Gt) * a.H + h) * 4096 + 4 * (32 * t + (32 * t + r < a.n ? r : 0));       // tiled: see win_table_kernel
    // (round 6: the LDS-shared table of winattn_bwd1_kernel<.., LT> was measured here too: bit-identical, 10-18 % SLOWER -- 13 KiB more LDS per
    // workgroup takes the forward from 3 to 2 workgroups per CU)
    float4 add[2][2][4];             // [q tile][key tile][row group g4]: keys 32*kt + 8*g4 + 4*hh + 0..3
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (NKEY > 0 && 32 * kt + 8 * g4 >= NKEY) { add[qt][kt][g4] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }      // every key of the group is padding
                if (NOTAB) { add[qt][kt][g4] = notab_add<NKEY>(kt, g4, hh); continue; }
                add[qt][kt][g4] = *reinterpret_cast<const float4*>(bmq[qt] + 256 * ((kt * 4 + g4) * 2 + hh));      // (a half-padded group needs its -1e30s)
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_fence();

    bf16x8_t qf[2][2], kf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[t][s] = ld_frag(sQ + sw_off(32 * t + r, hh + 2 * s));
            kf[t][s] = ld_frag(sK + sw_off(32 * t + r, hh + 2 * s));
        }
    f32x16_t st[2][2];               // [q tile][key tile]: St[key][q]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[qt][kt] = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) st[qt][kt] = MFMA32(kf[kt][s], qf[qt][s], st[qt][kt]);
        }
    lds_fence();                     // the Q and K tiles are free: O leaves through them
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float x[2][16];
        float m = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (!key_reg_live<NKEY>(kt, reg)) continue;
                const float4 ad = add[qt][kt][reg >> 2];
                const float av = (reg & 3) == 0 ? ad.x : (reg & 3) == 1 ? ad.y : (reg & 3) == 2 ? ad.z : ad.w;
                x[kt][reg] = st[qt][kt][reg] * a.scale2 + av;
                m = fmaxf(m, x[kt][reg]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (!key_reg_live<NKEY>(kt, reg)) { x[kt][reg] = 0.f; continue; }
                x[kt][reg] = __builtin_amdgcn_exp2f(x[kt][reg] - m);
                l += x[kt][reg];
            }
        l += __shfl_xor(l, 32, 64);
        f32x16_t o = zero16();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) o = MFMA32(tr_frag64(sV, kt, s2, hh, r), pack8(x[kt] + 8 * s2), o);
        bf16_t* T = qt == 0 ? sQ : sK;
        put_tile32(T, o, 1.0f / l, r, hh);
        const int q = 32 * qt + r;
        if (q < a.n && lsep && hh == 0) lsep[((int64_t)p * a.H + h) * 64 + q] = (m + __log2f(l)) * 0.6931471805599453f;
    }
    lds_fence();
    const int64_t r0[2] = {trow[0] * a.ldo + h * HW, trow[1] * a.ldo + h * HW};
    const int64_t r1[2] = {trow[2] * a.ldo + h * HW, trow[3] * a.ldo + h * HW};
    if (a.X) {                                             // kernel-uniform: the pair's gated residual beside O
        const float gsc = a.gate[0];
        const int64_t q0[2] = {trow[0] * a.ld + h * HW, trow[1] * a.ld + h * HW}, q1[2] = {trow[2] * a.ld + h * HW, trow[3] * a.ld + h * HW};
        const int64_t x0[2] = {trow[0] * a.ldx + h * HW, trow[1] * a.ldx + h * HW}, x1[2] = {trow[2] * a.ldx + h * HW, trow[3] * a.ldx + h * HW};
        flush_tile32_gate<D16>(sQ, Op, r0, Qp, q0, a.X, x0, gsc, 0, a.n, lane);
        flush_tile32_gate<D16>(sK, Op, r1, Qp, q1, a.X, x1, gsc, 1, a.n, lane);
        return;
    }
    flush_tile32<D16>(sQ, Op, r0, 0, a.n, lane);
    flush_tile32<D16>(sK, Op, r1, 1, a.n, lane);
}
template <int NKEY, bool NOTAB = false, bool D16 = false>
__global__ void __launch_bounds__(256, 2) winattn_fwd1_kernel(WinP a) { winattn_fwd1_body<NKEY, NOTAB, D16>(a); }
template <bool D16>
__global__ void __launch_bounds__(256, 2) winattn_fwd1_pair_kernel(WinP2 pp) { winattn_fwd1_body<49, true, D16>(pp.a[blockIdx.y]); }

// ------------------------------------------------------------------------------------------------ bias + mask table
// Additive score term v(q, k) = log2(e) * (table[index[q*n + k]][h] + mask[g][q][k])   (k >= n: -1e30; q >= n: 0), padded to
// 64 x 64 per (window type g, head h) and stored TILED for the kernels' access pattern -- a lane owns one index x (its
// query in `bm`, its key in `bmT`) and reads float4s along the other index y = 32 kt + 8 g4 + 4 hh + e:
//     offset(x, y) = (((kt * 4 + g4) * 2 + hh) * 64 + x) * 4 + e
// so one wave-instruction reads two 512-byte runs instead of 64 pieces of 64 different rows.
//     bm : x = q, y = k          bmT: x = k, y = q
__device__ __forceinline__ int tiled_off(int x, int y) { return ((y >> 2) * 64 + x) * 4 + (y & 3); }   // (y >> 2) == (kt*4 + g4)*2 + hh

__global__ void win_table_kernel(const float* table, const int64_t* index, const float* mask, float* bm, float* bmT, int L, int H,
                                 int n, int Gt) {
    const int total = Gt * H * 64 * 64;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        const int k = id & 63, q = (id >> 6) & 63;
        const int h = (id >> 12) % H, g = (id >> 12) / H;
        float v = 0.f;
        if (k >= n) v = NEG_BIG;
        else if (q < n) {
            int64_t ix = index[q * n + k];
            ix = ix < 0 ? 0 : (ix >= L ? L - 1 : ix);
            v = (table[ix * H + h] + (mask ? mask[((int64_t)g * n + q) * n + k] : 0.f)) * 1.4426950408889634f;
        }
        const int64_t base = ((int64_t)g * H + h) * 4096;
        bm[base + tiled_off(q, k)] = v;
        bmT[base + tiled_off(k, q)] = v;
    }
}

int fill(const stg_winattn_args* f, WinP& p, const char* who) {
    STG_CHECK(f->Q && f->K && f->V, -1, "%s: null pointer", who);
    STG_CHECK((f->bm == nullptr) == (f->bmT == nullptr), -1, "%s: bm and bmT go together (both NULL: no bias / mask, round 6)", who);
    STG_CHECK(f->D == WD || (f->D == 16 && f->bm == nullptr), -2, "%s: head dim must be 32 (or 16 without a table)", who);
    STG_CHECK(f->bm != nullptr || f->ws * f->ws == 49, -2, "%s: the table-free form is built for 7 x 7 windows", who);
    STG_CHECK(f->ws > 0 && f->n == f->ws * f->ws && f->n <= 64, -2, "%s: window must hold <= 64 tokens", who);
    STG_CHECK(f->Himg % f->ws == 0 && f->Wimg % f->ws == 0 && f->shift >= 0 && f->shift < f->ws, -2, "%s: bad window geometry", who);
    STG_CHECK(f->G == (f->Himg / f->ws) * (f->Wimg / f->ws) && (f->Gt == 1 || f->Gt == f->G), -2, "%s: bad G / Gt", who);
    STG_CHECK(f->P >= 0 && f->P % f->G == 0 && f->H > 0 && f->P * (int64_t)f->H < (1ll << 30), -2, "%s: bad problem count", who);
    STG_CHECK(f->outer >= (int64_t)f->Himg * f->Wimg, -2, "%s: outer too small", who);
    STG_CHECK(f->ld % 8 == 0 && (((uintptr_t)f->Q | (uintptr_t)f->K | (uintptr_t)f->V) & 15) == 0, -2, "%s: misaligned qkv", who);
    p.Q = (const bf16_t*)f->Q; p.K = (const bf16_t*)f->K; p.V = (const bf16_t*)f->V; p.ld = f->ld;
    p.O = (bf16_t*)f->O; p.ldo = f->ldo; p.lse = f->lse; p.bm = f->bm; p.bmT = f->bmT; p.Gt = f->Gt; p.outer = f->outer;
    p.Himg = f->Himg; p.Wimg = f->Wimg; p.ws = f->ws; p.shift = f->shift; p.nww = f->Wimg / f->ws; p.G = f->G; p.n = f->n;
    STG_CHECK(f->G < 4096 && (int64_t)f->G * p.nww < (1 << 20) && 64 * f->ws < (1 << 20), -2, "%s: window grid too large", who);
    p.ws_magic = ((1u << 20) + f->ws - 1) / f->ws; p.nww_magic = ((1u << 20) + p.nww - 1) / p.nww;
    p.P = (int)f->P; p.H = f->H; p.scale = f->scale; p.scale2 = f->scale * 1.4426950408889634f; p.total = (int)(f->P * f->H);
    return 0;
}

// grid of the LT form: 8 XCD lanes x ceil(units / 8) x H workgroups (units = frame quads x windows)
unsigned lt_grid(const WinP& p) {
    const int64_t F = p.P / p.G, U = ((F + 3) / 4) * p.G;
    return (unsigned)(((U + 7) / 8) * 8 * p.H);
}

}  // namespace

extern "C" int stg_winattn_table(const float* table, const int64_t* index, const float* mask, float* bm, float* bmT, int L, int H,
                                 int n, int Gt, void* stream) {
    STG_CHECK(table && index && bm && bmT, -1, "stg_winattn_table: null pointer");
    STG_CHECK(L > 0 && H > 0 && n > 0 && n <= 64 && Gt > 0, -2, "stg_winattn_table: bad shape");
    const int total = Gt * H * 64 * 64;
    int blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(win_table_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, index, mask, bm, bmT, L, H, n, Gt);
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_winattn_fwd(const stg_winattn_args* f, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_winattn_fwd: null args");
    WinP p = {};
    int rc = fill(f, p, "stg_winattn_fwd");
    if (rc) return rc;
    STG_CHECK(f->O && f->ldo % 8 == 0 && (((uintptr_t)f->O) & 15) == 0, -2, "stg_winattn_fwd: bad O (16-byte stores)");
    if (p.total == 0) return 0;
    if (!f->bm && f->D == 16) hipLaunchKernelGGL((winattn_fwd1_kernel<49, true, true>), dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    else if (!f->bm) hipLaunchKernelGGL((winattn_fwd1_kernel<49, true, false>), dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    else if (p.n == 49) hipLaunchKernelGGL(winattn_fwd1_kernel<49>, dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(winattn_fwd1_kernel<0>, dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_winattn_bwd(const stg_winattn_args* f, const void* dO, int64_t lddo, void* dQ, void* dK, void* dV,
                               int64_t lddqkv, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_winattn_bwd: null args");
    WinP p = {};
    int rc = fill(f, p, "stg_winattn_bwd");
    if (rc) return rc;
    // LT (the additive table staged once per workgroup of four same-(window, head) frames): measured on one box, bit-identical, stage 2 (16 heads)
    // 255.8 -> 242.5 us, stage 1 (8) 477 -> 461, stage 0 (4 heads) 905 -> 947: taken from 8 heads up
    const bool lt = p.n == 49 && p.H >= 8 && f->bm != nullptr;
    STG_CHECK(f->O && f->lse && dO && dQ && dK, -1, "stg_winattn_bwd: null pointer (dV == NULL -> dK receives dK + dV)");
    STG_CHECK(f->ldo % 8 == 0 && lddo % 8 == 0 && lddqkv % 8 == 0, -2, "stg_winattn_bwd: bad leading dims");
    STG_CHECK((((uintptr_t)f->O | (uintptr_t)dO) & 15) == 0 && (((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15) == 0, -2,
              "stg_winattn_bwd: misaligned pointers");
    if (p.total == 0) return 0;
    p.dO = (const bf16_t*)dO; p.lddo = lddo; p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV; p.lddqkv = lddqkv;
    if (!f->bm && f->D == 16) hipLaunchKernelGGL((winattn_bwd1_kernel<49, false, true, true>), dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    else if (!f->bm) hipLaunchKernelGGL((winattn_bwd1_kernel<49, false, true, false>), dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    else if (lt) hipLaunchKernelGGL((winattn_bwd1_kernel<49, true>), dim3(lt_grid(p)), dim3(256), 0, (hipStream_t)stream, p);
    else if (p.n == 49) hipLaunchKernelGGL((winattn_bwd1_kernel<49, false>), dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((winattn_bwd1_kernel<0, false>), dim3((p.total + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}

// ---- round 6b: the adapters' window-level cross-modal PAIR (Swin_AVE.py:750-760) as ONE launch per pass, its gates included -----------------------
// h_v' = h_v + gate_v softmax(h_v h_a^T) h_a and the mirror image: a0 / a1 are the two directions (table-free: bm == bmT == NULL, H == 1, K == V == the
// other modality's rows, same window geometry).  Before: two forward launches + stg_gate_fwd2; stg_gate_bwd2 + two backward launches -- at Swin-B's stage 2
// the gate kernels (5.8 / 14.6 us) were as long as the attention launches beside them (6.3 / 11.2 us).
static int pair_fill(const stg_winattn_args* f0, const stg_winattn_args* f1, WinP2& pp, const char* who) {
    STG_CHECK(f0 && f1, -1, "%s: null args", who);
    int rc = fill(f0, pp.a[0], who);
    if (rc) return rc;
    rc = fill(f1, pp.a[1], who);
    if (rc) return rc;
    STG_CHECK(!f0->bm && !f1->bm && f0->H == 1 && f1->H == 1 && f0->K == f0->V && f1->K == f1->V, -2, "%s: a table-free, one-head pair with K == V per direction", who);
    STG_CHECK(f0->D == f1->D && f0->P == f1->P && f0->G == f1->G && f0->n == f1->n && f0->Himg == f1->Himg && f0->Wimg == f1->Wimg && f0->ws == f1->ws &&
              f0->shift == f1->shift && f0->outer == f1->outer && f0->scale == f1->scale, -2, "%s: the two directions must share one geometry", who);
    return 0;
}

extern "C" int stg_winattn_pair_fwd(const stg_winattn_args* f0, const stg_winattn_args* f1, const float* gate0, const float* gate1, void* x0, void* x1,
                                    int64_t ldx, void* stream) {
    WinP2 pp = {};
    int rc = pair_fill(f0, f1, pp, "stg_winattn_pair_fwd");
    if (rc) return rc;
    STG_CHECK(gate0 && gate1 && x0 && x1 && f0->O && f1->O && f0->lse && f1->lse, -1, "stg_winattn_pair_fwd: null pointer");
    STG_CHECK(f0->ldo % 8 == 0 && f1->ldo % 8 == 0 && ldx % 8 == 0 && ldx >= f0->D && (((uintptr_t)f0->O | (uintptr_t)f1->O | (uintptr_t)x0 | (uintptr_t)x1) & 15) == 0, -2,
              "stg_winattn_pair_fwd: bad O / x operands (16-byte stores)");
    pp.a[0].gate = gate0; pp.a[0].X = (bf16_t*)x0; pp.a[0].ldx = ldx;
    pp.a[1].gate = gate1; pp.a[1].X = (bf16_t*)x1; pp.a[1].ldx = ldx;
    if (pp.a[0].total == 0) return 0;
    const dim3 grid((pp.a[0].total + 3) / 4, 2);
    if (f0->D == 16) hipLaunchKernelGGL(winattn_fwd1_pair_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, pp);
    else hipLaunchKernelGGL(winattn_fwd1_pair_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, pp);
    STG_LAUNCH_CHECK();
    return 0;
}

// dX0 / dX1: the gradients of the gated outputs x0 / x1 [rows, >= D] (leading dimension lddx); dQ_y / dK_y: direction y's query / key-and-value gradients
// (dK receives dK + dV), all with leading dimension lddqkv; dgate_y (fp32, accumulated atomically) += <dX_y, O_y>.
extern "C" int stg_winattn_pair_bwd(const stg_winattn_args* f0, const stg_winattn_args* f1, const void* dX0, const void* dX1, int64_t lddx, const float* gate0,
                                    const float* gate1, float* dgate0, float* dgate1, void* dQ0, void* dK0, void* dQ1, void* dK1, int64_t lddqkv, void* stream) {
    WinP2 pp = {};
    int rc = pair_fill(f0, f1, pp, "stg_winattn_pair_bwd");
    if (rc) return rc;
    STG_CHECK(dX0 && dX1 && gate0 && gate1 && dgate0 && dgate1 && dQ0 && dK0 && dQ1 && dK1 && f0->O && f1->O && f0->lse && f1->lse, -1, "stg_winattn_pair_bwd: null pointer");
    STG_CHECK(f0->ldo % 8 == 0 && f1->ldo % 8 == 0 && lddx % 8 == 0 && lddqkv % 8 == 0, -2, "stg_winattn_pair_bwd: bad leading dims");
    STG_CHECK((((uintptr_t)f0->O | (uintptr_t)f1->O | (uintptr_t)dX0 | (uintptr_t)dX1 | (uintptr_t)dQ0 | (uintptr_t)dK0 | (uintptr_t)dQ1 | (uintptr_t)dK1) & 15) == 0, -2,
              "stg_winattn_pair_bwd: misaligned pointers");
    const void* dX[2] = {dX0, dX1}; const float* gt[2] = {gate0, gate1}; float* dg[2] = {dgate0, dgate1}; void* dQ[2] = {dQ0, dQ1}; void* dK[2] = {dK0, dK1};
    for (int y = 0; y < 2; ++y) {
        WinP& p = pp.a[y];
        p.dO = (const bf16_t*)dX[y]; p.lddo = lddx; p.dQ = (bf16_t*)dQ[y]; p.dK = (bf16_t*)dK[y]; p.dV = nullptr; p.lddqkv = lddqkv;
        p.gate = gt[y]; p.dgate = dg[y];
    }
    if (pp.a[0].total == 0) return 0;
    const dim3 grid((pp.a[0].total + 3) / 4, 2);
    if (f0->D == 16) hipLaunchKernelGGL(winattn_bwd1_pair_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, pp);
    else hipLaunchKernelGGL(winattn_bwd1_pair_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, pp);
    STG_LAUNCH_CHECK();
    return 0;
}
