// Gather-mapped flash attention for the STG-CMA hot path: forward, dQ and dK/dV (see include/stgcma.h).
//
// One kernel family serves every attention in the model -- (shifted-)window W-MSA with relative-position bias and
// shift mask, temporal attention over T frames, ViT multi-head attention, and the single-head unscaled cross-modal
// attention of the adapters (window-level and frame-global, N up to 3136) -- because on this path they differ only
// in (a) how a (problem, token) pair maps to a row of the token tensor and (b) head dim / bias / mask.
//
// Work decomposition: ONE WAVE per (problem, head, 32-row tile); waves never synchronise with each other
// (LDS regions are wave-private, ordered by wavefront-scope fences), so a launch is just a flat list of waves.
// Short sequences (n <= 16: temporal attention, T = 10 / 5) are PACKED: floor(32 / n) problems share one 32-row tile with
// a block-diagonal validity mask, so a wave's MFMA tile carries 3 (T = 10) or 6 (T = 5) sequences instead of one.
//
// Addressing is arithmetic wherever the model's maps are (map_kind 1: cyclic-shift + window partition, 2: temporal
// regrouping) -- a table lookup would put a dependent HBM/L2 round trip in front of every operand load.  Operand loads of
// tile t+1 are issued into registers before tile t is computed (these kernels are latency-, not FLOP-bound).
//
// MFMA: v_mfma_f32_32x32x16_bf16.  Lane l = (r = l & 31, hh = l >> 5):
//   A operand: A[row r][k = 8*hh + j],  B operand: B[k = 8*hh + j][col r],  j = 0..7
//   C/D:       col = r, row = (reg & 3) + 8 * (reg >> 2) + 4 * hh,  reg = 0..15
// Scores are produced TRANSPOSED (St[key][q] = K . Q^T) so that the query sits on the lane: the row softmax is a
// reduction over registers + one lane^32 exchange, the running max / sum / LSE are per-lane scalars, and the
// probabilities are already the B operand of the next product (O^T[d][q] = V^T . P^T), whose k slots are the
// accumulator rows: k slot (hh, j) of step s  <->  key kappa(s, hh, j) = 16 s + 8 (j >> 2) + 4 hh + (j & 3).
// The other operand of those products (V^T, K^T, Q^T, dO^T) is k-strided in memory; its 32 x D tile is staged
// row-major in LDS with 16-byte stores and gathered with 16-bit reads (32 lanes read 64 contiguous bytes).
#include <math.h>
#include <stdlib.h>
#include "common.h"
#include "../../include/stgcma.h"
#include "xattn.h"

namespace {

struct AttnP {
    const bf16_t* Q; int64_t ldq;
    const bf16_t* K; int64_t ldk;
    const bf16_t* V; int64_t ldv;
    bf16_t* O; int64_t ldo;
    float* lse;
    const int32_t* map_q; const int32_t* map_kv;
    int64_t outer_q, outer_kv;
    int G;
    int map_kind, ma, mb, mc, md;     // 1: window (Himg, Wimg, ws, shift)   2: temporal (N tokens per frame)
    int P; int H; int n; int n_kv;
    int pack;                          // problems per 32-row tile (1 unless n == n_kv <= 16)
    float scale;
    const float* bias; int bias_div; int bias_mod;
    const float* mask;
    // backward
    const bf16_t* dO; int64_t lddo;
    bf16_t* dQ; int64_t lddq;
    bf16_t* dK; int64_t lddk;
    bf16_t* dV; int64_t lddv;
    float* delta;
    float* dbias;
    int pchunk;                        // tile-groups per wave in the dK/dV kernel (dbias reduction)
    int total_items;
};

struct AttnP2 { AttnP a[2]; };     // the two directions of a cross-modal pair in one launch (blockIdx.y); a single call fills a[0] only

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// token offset of (group g, token i) inside its outer block
__device__ __forceinline__ int tok_off(const AttnP& a, bool kv, int g, int i) {
    if (a.map_kind == 1) {
        const int nww = a.mb / a.mc;
        const int wi = g / nww, wj = g - wi * nww;
        const int ti = i / a.mc, tj = i - ti * a.mc;
        int h = wi * a.mc + ti + a.md;
        h = h >= a.ma ? h - a.ma : h;
        int w = wj * a.mc + tj + a.md;
        w = w >= a.mb ? w - a.mb : w;
        return h * a.mb + w;
    }
    if (a.map_kind == 2) return i * a.ma + g;
    const int32_t* m = kv ? a.map_kv : a.map_q;
    const int idx = g * (kv ? a.n_kv : a.n) + i;
    return m ? m[idx] : idx;
}

// tile row x (0..31) of the tile that starts at token t0 of problem(-group) p0 -> (problem, token, valid); invalid rows are
// clamped to an in-bounds (problem, token) so that every load stays unconditional
template <bool PK>
__device__ __forceinline__ bool locate(const AttnP& a, int p0, int t0, int x, int nt, int& p, int& i) {
    bool ok;
    if (PK) {
        const int sub = x / nt;
        i = x - sub * nt;
        p = p0 + sub;
        ok = sub < a.pack && p < a.P;
        if (!ok) { p = p0; i = 0; }
    } else {
        p = p0;
        i = t0 + x;
        ok = i < nt;
        if (!ok) i = nt - 1;
    }
    return ok;
}

__device__ __forceinline__ int64_t row_of(const AttnP& a, bool kv, int p, int i) {
    const unsigned pg = (unsigned)p / (unsigned)a.G;
    const int g = p - (int)pg * a.G;
    return (int64_t)pg * (kv ? a.outer_kv : a.outer_q) + tok_off(a, kv, g, i);
}

// ---- 32 x D tile staging through registers (so the loads of the NEXT tile can be in flight during compute)
template <int D>
struct TileRegs {
    static constexpr int CPR = D / 8;
    static constexpr int NV = (32 * CPR) / 64;     // 16-byte chunks per lane (32*CPR is a multiple of 64 for every D)
    uint4 v[NV];
    template <bool PK>
    __device__ __forceinline__ void load(const AttnP& a, const bf16_t* src, int64_t ld, bool kv, int nt, int p0, int h, int t0,
                                         int lane) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = lane + 64 * i;
            const int tr = idx / CPR, ch = idx - tr * CPR;
            int p, t;
            locate<PK>(a, p0, t0, tr, nt, p, t);
            v[i] = *reinterpret_cast<const uint4*>(src + row_of(a, kv, p, t) * ld + h * D + ch * 8);
        }
    }
    template <bool PK>
    __device__ __forceinline__ void store(const AttnP& a, bf16_t* s, int nt, int p0, int t0, int lane) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = lane + 64 * i;
            const int tr = idx / CPR, ch = idx - tr * CPR;
            int p, t;
            const bool ok = locate<PK>(a, p0, t0, tr, nt, p, t);
            *reinterpret_cast<uint4*>(s + tr * D + ch * 8) = ok ? v[i] : make_uint4(0, 0, 0, 0);
        }
    }
};

// A-operand fragment of the TRANSPOSED tile: A[i = d][k slot j] = tile[kappa(s2, hh, j)][d]; rows d >= D are zero
template <int D>
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* s, int s2, int hh, int d) {
    bf16x8_t f;
    const bool ok = d < D;
    const int dd = ok ? d : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int key = 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3);
        const short v = (short)s[key * D + dd];
        f[j] = ok ? v : (short)0;
    }
    return f;
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t pack_frag(const float* x) {
    const u32x4_t w = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
    return __builtin_bit_cast(bf16x8_t, w);
}

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define ACC_ROW(reg, hh) (((reg) & 3) + 8 * ((reg) >> 2) + 4 * (hh))

__device__ __forceinline__ void decode_item(const AttnP& a, int item, int tiles, int& t, int& h, int& p0) {
    t = item % tiles;
    const int rest = item / tiles;
    h = rest % a.H;
    p0 = (rest / a.H) * a.pack;
}

// additive score terms (bias + mask) and validity of the 16 accumulator rows a lane holds, for "other side" rows of a tile
// fixed = this lane's own token (query for fwd/dq, key for dkv); the bias/mask element is [q token][k token].
struct RowTerms {
    float add[16];
    unsigned okmask;
};

// ------------------------------------------------------------------------------------------------ forward
template <int D, bool PK>
__global__ void __launch_bounds__(256, (D > 64 && PK ? 1 : 2)) attn_fwd_kernel(AttnP2 pp) {
    const AttnP a = pp.a[blockIdx.y];
    constexpr int KS = D / 16;
    constexpr int DB = (D + 31) / 32;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 32 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.total_items) return;
    const int QT = PK ? 1 : (a.n + 31) >> 5;
    int qt, h, p0;
    decode_item(a, item, QT, qt, h, p0);
    bf16_t* sV = smem + wave * 32 * D;

    int pq, iq;
    const bool okq = locate<PK>(a, p0, qt * 32, r, a.n, pq, iq);
    const int64_t rowq = row_of(a, false, pq, iq);
    bf16x8_t qf[KS];
    {
        const bf16_t* qp = a.Q + rowq * a.ldq + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = ld_frag(qp + 16 * s);
    }
    const float* bias_q = nullptr;
    if (a.bias) bias_q = a.bias + ((int64_t)(((pq / a.bias_div) % a.bias_mod) * a.H + h) * a.n + iq) * a.n_kv;
    const float* mask_q = a.mask ? a.mask + ((int64_t)(pq % a.G) * a.n + iq) * a.n_kv : nullptr;

    f32x16_t o[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b) o[b] = zero16();
    float m = -INFINITY, l = 0.f;

    // prefetch tile 0
    TileRegs<D> vt;
    bf16x8_t kf[KS];
    {
        vt.template load<PK>(a, a.V, a.ldv, true, a.n_kv, p0, h, 0, lane);
        int pk, jk;
        locate<PK>(a, p0, 0, r, a.n_kv, pk, jk);
        const bf16_t* kp = a.K + row_of(a, true, pk, jk) * a.ldk + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = ld_frag(kp + 16 * s);
    }

    for (int kv0 = 0; kv0 < a.n_kv; kv0 += 32) {
        lds_fence();
        vt.template store<PK>(a, sV, a.n_kv, p0, kv0, lane);
        f32x16_t st = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) st = MFMA32(kf[s], qf[s], st);
        if (kv0 + 32 < a.n_kv) {                       // wave-uniform: next tile's operands go in flight now
            vt.template load<PK>(a, a.V, a.ldv, true, a.n_kv, p0, h, kv0 + 32, lane);
            int pk, jk;
            locate<PK>(a, p0, kv0 + 32, r, a.n_kv, pk, jk);
            const bf16_t* kp = a.K + row_of(a, true, pk, jk) * a.ldk + h * D + 8 * hh;
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[s] = ld_frag(kp + 16 * s);
        }
        float x[16], add[16];
        unsigned okm = 0;
        int jcol[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int pk, jk;
            const bool ok = locate<PK>(a, p0, kv0, ACC_ROW(reg, hh), a.n_kv, pk, jk) && pk == pq;
            okm |= ok ? (1u << reg) : 0u;
            jcol[reg] = jk;
            add[reg] = 0.f;
        }
        if (bias_q) {                                  // wave-uniform branches around blocks of independent loads
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) add[reg] += bias_q[jcol[reg]];
        }
        if (mask_q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) add[reg] += mask_q[jcol[reg]];
        }
        float mt = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            float v = st[reg] * a.scale + add[reg];
            v = ((okm >> reg) & 1u) ? v : -INFINITY;
            x[reg] = v;
            mt = fmaxf(mt, v);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m, mt);
        const float msafe = mn == -INFINITY ? 0.f : mn;   // a padded query row of a packed tile sees no valid key
        const float alpha = __expf(m - msafe);
        float ps = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = __expf(x[reg] - msafe);
            ps += x[reg];
        }
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) o[b][reg] *= alpha;
        bf16x8_t pf[2];
        pf[0] = pack_frag(x);
        pf[1] = pack_frag(x + 8);
        lds_fence();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < DB; ++b) o[b] = MFMA32(tr_frag<D>(sV, s2, hh, 32 * b + r), pf[s2], o[b]);
    }

    if (okq) {
        const float inv = 1.0f / l;
        bf16_t* op = a.O + rowq * a.ldo + h * D;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * b + 8 * g + 4 * hh;
                if (d < D) {
                    uint2 w;
                    w.x = pack_bf2(o[b][4 * g + 0] * inv, o[b][4 * g + 1] * inv);
                    w.y = pack_bf2(o[b][4 * g + 2] * inv, o[b][4 * g + 3] * inv);
                    *reinterpret_cast<uint2*>(op + d) = w;
                }
            }
        if (a.lse && hh == 0) a.lse[((int64_t)pq * a.H + h) * a.n + iq] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <int D, bool PK>
__global__ void __launch_bounds__(256, (D > 64 && PK ? 1 : 2)) attn_bwd_dq_kernel(AttnP2 pp) {
    const AttnP a = pp.a[blockIdx.y];
    constexpr int KS = D / 16;
    constexpr int DB = (D + 31) / 32;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 32 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.total_items) return;
    const int QT = PK ? 1 : (a.n + 31) >> 5;
    int qt, h, p0;
    decode_item(a, item, QT, qt, h, p0);
    bf16_t* sK = smem + wave * 32 * D;

    int pq, iq;
    const bool okq = locate<PK>(a, p0, qt * 32, r, a.n, pq, iq);
    const int64_t rowq = row_of(a, false, pq, iq);
    bf16x8_t qf[KS], dof[KS];
    float delta = 0.f;
    {
        const bf16_t* qp = a.Q + rowq * a.ldq + h * D + 8 * hh;
        const bf16_t* dp = a.dO + rowq * a.lddo + h * D + 8 * hh;
        const bf16_t* op = a.O + rowq * a.ldo + h * D + 8 * hh;
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
    const int64_t stat_idx = ((int64_t)pq * a.H + h) * a.n + iq;
    if (okq && hh == 0) a.delta[stat_idx] = delta;
    const float lse_q = a.lse[stat_idx];
    const float* bias_q = nullptr;
    if (a.bias) bias_q = a.bias + ((int64_t)(((pq / a.bias_div) % a.bias_mod) * a.H + h) * a.n + iq) * a.n_kv;
    const float* mask_q = a.mask ? a.mask + ((int64_t)(pq % a.G) * a.n + iq) * a.n_kv : nullptr;

    f32x16_t dq[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b) dq[b] = zero16();

    TileRegs<D> kt;
    bf16x8_t vf[KS];
    {
        kt.template load<PK>(a, a.K, a.ldk, true, a.n_kv, p0, h, 0, lane);
        int pk, jk;
        locate<PK>(a, p0, 0, r, a.n_kv, pk, jk);
        const bf16_t* vp = a.V + row_of(a, true, pk, jk) * a.ldv + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) vf[s] = ld_frag(vp + 16 * s);
    }

    for (int kv0 = 0; kv0 < a.n_kv; kv0 += 32) {
        lds_fence();
        kt.template store<PK>(a, sK, a.n_kv, p0, kv0, lane);
        f32x16_t dpt = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) dpt = MFMA32(vf[s], dof[s], dpt);
        if (kv0 + 32 < a.n_kv) {
            kt.template load<PK>(a, a.K, a.ldk, true, a.n_kv, p0, h, kv0 + 32, lane);
            int pk, jk;
            locate<PK>(a, p0, kv0 + 32, r, a.n_kv, pk, jk);
            const bf16_t* vp = a.V + row_of(a, true, pk, jk) * a.ldv + h * D + 8 * hh;
#pragma unroll
            for (int s = 0; s < KS; ++s) vf[s] = ld_frag(vp + 16 * s);
        }
        lds_fence();
        f32x16_t st = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) st = MFMA32(ld_frag(sK + r * D + 16 * s + 8 * hh), qf[s], st);
        float ds[16], add[16];
        unsigned okm = 0;
        int jcol[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int pk, jk;
            const bool ok = locate<PK>(a, p0, kv0, ACC_ROW(reg, hh), a.n_kv, pk, jk) && pk == pq;
            okm |= ok ? (1u << reg) : 0u;
            jcol[reg] = jk;
            add[reg] = 0.f;
        }
        if (bias_q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) add[reg] += bias_q[jcol[reg]];
        }
        if (mask_q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) add[reg] += mask_q[jcol[reg]];
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float v = st[reg] * a.scale + add[reg];
            const float pr = ((okm >> reg) & 1u) ? __expf(v - lse_q) : 0.f;
            ds[reg] = pr * (dpt[reg] - delta);
        }
        bf16x8_t dsf[2];
        dsf[0] = pack_frag(ds);
        dsf[1] = pack_frag(ds + 8);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < DB; ++b) dq[b] = MFMA32(tr_frag<D>(sK, s2, hh, 32 * b + r), dsf[s2], dq[b]);
    }
    if (okq) {
        bf16_t* op = a.dQ + rowq * a.lddq + h * D;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * b + 8 * g + 4 * hh;
                if (d < D) {
                    uint2 w;
                    w.x = pack_bf2(dq[b][4 * g + 0] * a.scale, dq[b][4 * g + 1] * a.scale);
                    w.y = pack_bf2(dq[b][4 * g + 2] * a.scale, dq[b][4 * g + 3] * a.scale);
                    *reinterpret_cast<uint2*>(op + d) = w;
                }
            }
    }
}

// dbias[bg][h][i][j] += dS accumulated in the dK/dV kernel.  Entry [i][j] sits on the lane of key j (sub-problem s) and the
// register of query i of the SAME sub-problem (block diagonal when packed).  The bias group is taken per lane from the
// lane's own problem of tile-group `gi` (host guarantees n <= 32 and that a flush never mixes bias groups on one lane).
template <bool PK>
__device__ __forceinline__ void flush_dbias(const AttnP& a, f32x16_t& dbacc, int gi, int kv0, int r, int hh, int h) {
    const int subk = PK ? r / a.n_kv : 0;
    const int jk = PK ? r - subk * a.n_kv : kv0 + r;
    const int pk = gi * a.pack + subk;
    const bool lane_ok = PK ? (subk < a.pack) : (jk < a.n_kv);
    if (lane_ok) {
        const int pkc = pk < a.P ? pk : a.P - 1;
        const int bg = (pkc / a.bias_div) % a.bias_mod;
        float* db = a.dbias + ((int64_t)(bg * a.H + h) * a.n) * a.n_kv + jk;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int x = ACC_ROW(reg, hh);
            int subq, iq;
            if (PK) { subq = x / a.n; iq = x - subq * a.n; } else { subq = 0; iq = x; }
            if (subq == subk && iq < a.n) atomicAdd(db + iq * a.n_kv, dbacc[reg]);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) dbacc[reg] = 0.f;
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV (+ dbias)
// The wave owns 32 keys (key on the lane): S[q][key] = Q . K^T and dP[q][key] = dO . V^T come out with the query in
// registers; P and dS are then the B operands of dV^T[d][key] = dO^T . P and dK^T[d][key] = Q^T . dS.
template <int D, bool PK>
// D = 96 / 128 (ViT heads): at two waves per SIMD the accumulators spill 300+ VGPRs to scratch; one wave per SIMD (AGPRs) measured
// 130 -> 81 ms over the ViT-B step, the other variants are faster at two
__global__ void __launch_bounds__(256, (D >= 48 && !PK ? 1 : 2)) attn_bwd_dkv_kernel(AttnP2 pp) {
    const AttnP a = pp.a[blockIdx.y];
    constexpr int KS = D / 16;
    constexpr int DB = (D + 31) / 32;
    constexpr bool PREFETCH = D <= 64;                 // register budget: two extra tiles in flight only for small D
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 2 * 32 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.total_items) return;
    const int KT = PK ? 1 : (a.n_kv + 31) >> 5;
    const int kt = item % KT;
    const int rest = item / KT;
    const int h = rest % a.H;
    const int gc = rest / a.H;                          // chunk of tile-groups
    bf16_t* sQ = smem + wave * 2 * 32 * D;
    bf16_t* sDO = sQ + 32 * D;
    const int kv0 = kt * 32;
    f32x16_t dbacc = zero16();
    const int ngroups = (a.P + a.pack - 1) / a.pack;
    const int gbeg = gc * a.pchunk;
    const int gend = gbeg + a.pchunk < ngroups ? gbeg + a.pchunk : ngroups;

    // a chunk whose problems span two bias groups (only at a group boundary) flushes after every tile-group instead of once
    const bool straddle = a.dbias != nullptr &&
        ((gbeg * a.pack) / a.bias_div) != (((gend * a.pack < a.P ? gend * a.pack : a.P) - 1) / a.bias_div);
    for (int gi = gbeg; gi < gend; ++gi) {
        if (straddle && gi > gbeg) flush_dbias<PK>(a, dbacc, gi - 1, kv0, r, hh, h);
        const int p0 = gi * a.pack;
        int pk, jk;
        const bool okk = locate<PK>(a, p0, kv0, r, a.n_kv, pk, jk);
        const int64_t rowk = row_of(a, true, pk, jk);
        bf16x8_t kf[KS], vf[KS];
        {
            const bf16_t* kp = a.K + rowk * a.ldk + h * D + 8 * hh;
            const bf16_t* vp = a.V + rowk * a.ldv + h * D + 8 * hh;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                kf[s] = ld_frag(kp + 16 * s);
                vf[s] = ld_frag(vp + 16 * s);
            }
        }
        const float* bias_k = nullptr;
        if (a.bias) bias_k = a.bias + ((int64_t)(((pk / a.bias_div) % a.bias_mod) * a.H + h) * a.n) * a.n_kv + jk;
        const float* mask_k = a.mask ? a.mask + ((int64_t)(pk % a.G) * a.n) * a.n_kv + jk : nullptr;

        f32x16_t dk[DB], dv[DB];
#pragma unroll
        for (int b = 0; b < DB; ++b) { dk[b] = zero16(); dv[b] = zero16(); }

        TileRegs<D> tq, tdo;
        tq.template load<PK>(a, a.Q, a.ldq, false, a.n, p0, h, 0, lane);
        tdo.template load<PK>(a, a.dO, a.lddo, false, a.n, p0, h, 0, lane);
        const int qend = PK ? 32 : a.n;
        for (int q0 = 0; q0 < qend; q0 += 32) {
            lds_fence();
            tq.template store<PK>(a, sQ, a.n, p0, q0, lane);
            tdo.template store<PK>(a, sDO, a.n, p0, q0, lane);
            if (PREFETCH && q0 + 32 < qend) {
                tq.template load<PK>(a, a.Q, a.ldq, false, a.n, p0, h, q0 + 32, lane);
                tdo.template load<PK>(a, a.dO, a.lddo, false, a.n, p0, h, q0 + 32, lane);
            }
            // per accumulator row (= query): statistics + additive terms, independent loads from clamped addresses
            float lse_r[16], del_r[16], add[16];
            unsigned okm = 0;
            int qoff[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int pqr, iqr;
                const bool ok = locate<PK>(a, p0, q0, ACC_ROW(reg, hh), a.n, pqr, iqr) && pqr == pk && okk;
                okm |= ok ? (1u << reg) : 0u;
                const int64_t si = ((int64_t)pqr * a.H + h) * a.n + iqr;
                lse_r[reg] = a.lse[si];
                del_r[reg] = a.delta[si];
                qoff[reg] = iqr * a.n_kv;
                add[reg] = 0.f;
            }
            if (bias_k) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) add[reg] += bias_k[qoff[reg]];
            }
            if (mask_k) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) add[reg] += mask_k[qoff[reg]];
            }
            lds_fence();
            f32x16_t sc = zero16(), dp = zero16();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                sc = MFMA32(ld_frag(sQ + r * D + 16 * s + 8 * hh), kf[s], sc);
                dp = MFMA32(ld_frag(sDO + r * D + 16 * s + 8 * hh), vf[s], dp);
            }
            float pr[16], ds[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v = sc[reg] * a.scale + add[reg];
                const float pv = ((okm >> reg) & 1u) ? __expf(v - lse_r[reg]) : 0.f;
                const float dsv = pv * (dp[reg] - del_r[reg]);
                pr[reg] = pv;
                ds[reg] = dsv;
                dbacc[reg] += dsv;
            }
            bf16x8_t pf[2], dsf[2];
            pf[0] = pack_frag(pr); pf[1] = pack_frag(pr + 8);
            dsf[0] = pack_frag(ds); dsf[1] = pack_frag(ds + 8);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    dv[b] = MFMA32(tr_frag<D>(sDO, s2, hh, 32 * b + r), pf[s2], dv[b]);
                    dk[b] = MFMA32(tr_frag<D>(sQ, s2, hh, 32 * b + r), dsf[s2], dk[b]);
                }
            if (!PREFETCH && q0 + 32 < qend) {
                tq.template load<PK>(a, a.Q, a.ldq, false, a.n, p0, h, q0 + 32, lane);
                tdo.template load<PK>(a, a.dO, a.lddo, false, a.n, p0, h, q0 + 32, lane);
            }
        }
        if (okk) {
            bf16_t* kp = a.dK + rowk * a.lddk + h * D;
            bf16_t* vp = a.dV ? a.dV + rowk * a.lddv + h * D : nullptr;
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * b + 8 * g + 4 * hh;
                    if (d < D) {
                        float kk[4], vv[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) { kk[t] = dk[b][4 * g + t] * a.scale; vv[t] = dv[b][4 * g + t]; }
                        if (vp) {
                            uint2 w; w.x = pack_bf2(vv[0], vv[1]); w.y = pack_bf2(vv[2], vv[3]);
                            *reinterpret_cast<uint2*>(vp + d) = w;
                        } else {  // K and V are the same tensor (cross-modal adapter): one fused gradient
#pragma unroll
                            for (int t = 0; t < 4; ++t) kk[t] += vv[t];
                        }
                        uint2 w; w.x = pack_bf2(kk[0], kk[1]); w.y = pack_bf2(kk[2], kk[3]);
                        *reinterpret_cast<uint2*>(kp + d) = w;
                    }
                }
        }
    }
    if (a.dbias) flush_dbias<PK>(a, dbacc, gend - 1, kv0, r, hh, h);
}

int fill(const stg_attn_args* f, AttnP& p, const char* who) {
    STG_CHECK(f->Q && f->K && f->V, -1, "%s: null Q/K/V", who);
    STG_CHECK(f->P >= 0 && f->H > 0 && f->n > 0 && f->n_kv > 0 && f->G > 0, -2, "%s: bad shape", who);
    STG_CHECK(f->P < (1ll << 30) && f->P * f->H < (1ll << 30), -2, "%s: too many problems", who);
    STG_CHECK(f->D == 16 || f->D == 32 || f->D == 48 || f->D == 64 || f->D == 96 || f->D == 128, -2,
              "%s: unsupported head dim %d", who, f->D);
    STG_CHECK(f->ldq % 8 == 0 && f->ldk % 8 == 0 && f->ldv % 8 == 0, -2, "%s: leading dims must be multiples of 8", who);
    STG_CHECK((((uintptr_t)f->Q | (uintptr_t)f->K | (uintptr_t)f->V) & 15) == 0, -2, "%s: Q/K/V must be 16-byte aligned", who);
    STG_CHECK(f->map_kind >= 0 && f->map_kind <= 2, -2, "%s: bad map_kind", who);
    if (f->map_kind == 0) {
        STG_CHECK(f->map_q != nullptr || f->outer_q >= (int64_t)f->G * f->n, -2, "%s: outer_q too small for identity map", who);
        STG_CHECK(f->map_kv != nullptr || f->outer_kv >= (int64_t)f->G * f->n_kv, -2, "%s: outer_kv too small for identity map", who);
    } else if (f->map_kind == 1) {       // window: (Himg, Wimg, ws, shift); G windows of ws*ws tokens tile the image
        STG_CHECK(f->map_c > 0 && f->map_a % f->map_c == 0 && f->map_b % f->map_c == 0 && f->map_d >= 0 && f->map_d < f->map_c,
                  -2, "%s: bad window map", who);
        STG_CHECK(f->n == f->map_c * f->map_c && f->n_kv == f->n && f->G == (f->map_a / f->map_c) * (f->map_b / f->map_c), -2,
                  "%s: window map does not match n / G", who);
        STG_CHECK(f->outer_q >= (int64_t)f->map_a * f->map_b && f->outer_kv >= (int64_t)f->map_a * f->map_b, -2,
                  "%s: outer too small for window map", who);
    } else {                              // temporal: map_a = tokens per frame (= G), n = T frames
        STG_CHECK(f->map_a == f->G && f->n_kv == f->n, -2, "%s: temporal map needs map_a == G and n_kv == n", who);
        STG_CHECK(f->outer_q >= (int64_t)f->n * f->map_a && f->outer_kv >= (int64_t)f->n * f->map_a, -2,
                  "%s: outer too small for temporal map", who);
    }
    if (f->bias) STG_CHECK(f->bias_div > 0 && f->bias_mod > 0 && f->bias_div < (1ll << 31), -2, "%s: bad bias grouping", who);
    p.Q = (const bf16_t*)f->Q; p.ldq = f->ldq; p.K = (const bf16_t*)f->K; p.ldk = f->ldk;
    p.V = (const bf16_t*)f->V; p.ldv = f->ldv; p.O = (bf16_t*)f->O; p.ldo = f->ldo; p.lse = f->lse;
    p.map_q = f->map_q; p.map_kv = f->map_kv; p.outer_q = f->outer_q; p.outer_kv = f->outer_kv; p.G = f->G;
    p.map_kind = f->map_kind; p.ma = f->map_a; p.mb = f->map_b; p.mc = f->map_c; p.md = f->map_d;
    p.P = (int)f->P; p.H = f->H; p.n = f->n; p.n_kv = f->n_kv; p.scale = f->scale;
    p.pack = (f->n == f->n_kv && f->n <= 16) ? 32 / f->n : 1;
    p.bias = f->bias; p.bias_div = f->bias ? (int)f->bias_div : 1; p.bias_mod = f->bias ? f->bias_mod : 1; p.mask = f->mask;
    return 0;
}

// p1 != nullptr: a second problem of the same head dim and packing in the same launch (grid.y = 2; the shorter one's surplus workgroups leave at once)
template <template <int, bool> class Launcher>
int dispatch_d(int D, const AttnP& p, hipStream_t st, const AttnP* p1 = nullptr) {
    const bool pk = p.pack > 1;
    AttnP2 pp;
    pp.a[0] = p; pp.a[1] = p1 ? *p1 : p;
    const int ny = p1 ? 2 : 1;
    switch (D) {
        case 16: return pk ? Launcher<16, true>::run(pp, ny, st) : Launcher<16, false>::run(pp, ny, st);
        case 32: return pk ? Launcher<32, true>::run(pp, ny, st) : Launcher<32, false>::run(pp, ny, st);
        case 48: return pk ? Launcher<48, true>::run(pp, ny, st) : Launcher<48, false>::run(pp, ny, st);
        case 64: return pk ? Launcher<64, true>::run(pp, ny, st) : Launcher<64, false>::run(pp, ny, st);
        case 96: return pk ? Launcher<96, true>::run(pp, ny, st) : Launcher<96, false>::run(pp, ny, st);
        case 128: return pk ? Launcher<128, true>::run(pp, ny, st) : Launcher<128, false>::run(pp, ny, st);
    }
    return -2;
}

inline int grid_of(int64_t items, unsigned& g) {
    const int64_t blocks = (items + 3) / 4;
    if (blocks <= 0 || items >= (1ll << 31)) return -2;
    g = (unsigned)blocks;
    return 0;
}

template <int D, bool PK> struct FwdL {
    static int run(const AttnP2& pp, int ny, hipStream_t st) {
        unsigned g, g1 = 0;
        STG_CHECK(grid_of(pp.a[0].total_items, g) == 0 && (ny == 1 || grid_of(pp.a[1].total_items, g1) == 0), -2, "attention: grid out of range");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_fwd_kernel<D, PK>), dim3(g > g1 ? g : g1, ny), dim3(256), 0, st, pp);
        STG_LAUNCH_CHECK();
        return 0;
    }
};
template <int D, bool PK> struct DqL {
    static int run(const AttnP2& pp, int ny, hipStream_t st) {
        unsigned g, g1 = 0;
        STG_CHECK(grid_of(pp.a[0].total_items, g) == 0 && (ny == 1 || grid_of(pp.a[1].total_items, g1) == 0), -2, "attention: grid out of range");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_bwd_dq_kernel<D, PK>), dim3(g > g1 ? g : g1, ny), dim3(256), 0, st, pp);
        STG_LAUNCH_CHECK();
        return 0;
    }
};
template <int D, bool PK> struct DkvL {
    static int run(const AttnP2& pp, int ny, hipStream_t st) {
        unsigned g, g1 = 0;
        STG_CHECK(grid_of(pp.a[0].total_items, g) == 0 && (ny == 1 || grid_of(pp.a[1].total_items, g1) == 0), -2, "attention: grid out of range");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_bwd_dkv_kernel<D, PK>), dim3(g > g1 ? g : g1, ny), dim3(256), 0, st, pp);
        STG_LAUNCH_CHECK();
        return 0;
    }
};

inline int64_t tiles_q(const AttnP& p) { return p.pack > 1 ? 1 : (p.n + 31) / 32; }
inline int64_t tiles_kv(const AttnP& p) { return p.pack > 1 ? 1 : (p.n_kv + 31) / 32; }
inline int64_t groups(const AttnP& p) { return ((int64_t)p.P + p.pack - 1) / p.pack; }

}  // namespace

static bool xattn_on() {          // option "xattn" = 0: keep the frame-global cross-modal attention on the generic kernels (A/B knob)
    return stg_opt_xattn.load(std::memory_order_relaxed) != 0;
}

extern "C" int stg_attn_fwd(const stg_attn_args* f, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_attn_fwd: null args");
    AttnP p = {};
    int rc = fill(f, p, "stg_attn_fwd");
    if (rc) return rc;
    STG_CHECK(f->O != nullptr && f->ldo % 4 == 0 && (((uintptr_t)f->O) & 7) == 0, -2, "stg_attn_fwd: bad O");
    if (f->P == 0) return 0;
    if (xattn_on() && stg_xattn_eligible(f, true)) return stg_xattn_fwd(f, stream);      // frame-global cross-modal attention
    const int64_t items = groups(p) * f->H * tiles_q(p);
    STG_CHECK(items < (1ll << 31), -2, "stg_attn_fwd: too many work items");
    p.total_items = (int)items;
    return dispatch_d<FwdL>(f->D, p, (hipStream_t)stream);
}

extern "C" int stg_attn_bwd(const stg_attn_bwd_args* b, void* stream) {
    STG_CHECK(b != nullptr, -1, "stg_attn_bwd: null args");
    const stg_attn_args* f = &b->f;
    AttnP p = {};
    int rc = fill(f, p, "stg_attn_bwd");
    if (rc) return rc;
    STG_CHECK(f->O && f->lse && b->dO && b->dQ && b->dK && b->delta, -1, "stg_attn_bwd: null O/lse/dO/dQ/dK/delta");
    STG_CHECK(f->ldo % 8 == 0 && b->lddo % 8 == 0 && b->lddq % 4 == 0 && b->lddk % 4 == 0 && (b->dV == nullptr || b->lddv % 4 == 0),
              -2, "stg_attn_bwd: bad leading dims");
    STG_CHECK((((uintptr_t)f->O | (uintptr_t)b->dO) & 15) == 0, -2, "stg_attn_bwd: O/dO must be 16-byte aligned");
    STG_CHECK((((uintptr_t)b->dQ | (uintptr_t)b->dK | (uintptr_t)b->dV) & 7) == 0, -2, "stg_attn_bwd: dQ/dK/dV must be 8-byte aligned");
    if (f->P == 0) return 0;
    if (xattn_on() && b->dV == nullptr && b->dbias == nullptr && stg_xattn_eligible(f, true)) return stg_xattn_bwd(b, stream);
    p.dO = (const bf16_t*)b->dO; p.lddo = b->lddo;
    p.dQ = (bf16_t*)b->dQ; p.lddq = b->lddq; p.dK = (bf16_t*)b->dK; p.lddk = b->lddk;
    p.dV = (bf16_t*)b->dV; p.lddv = b->lddv; p.delta = b->delta; p.dbias = b->dbias;
    p.pchunk = 1;
    int64_t items = groups(p) * f->H * tiles_q(p);
    STG_CHECK(items < (1ll << 31), -2, "stg_attn_bwd: too many work items");
    p.total_items = (int)items;
    rc = dispatch_d<DqL>(f->D, p, (hipStream_t)stream);
    if (rc) return rc;
    int64_t nchunks = groups(p);
    if (b->dbias) {
        STG_CHECK(f->bias != nullptr, -2, "stg_attn_bwd: dbias without bias");
        STG_CHECK(f->n <= 32 && f->n_kv <= 32, -2, "stg_attn_bwd: dbias needs n <= 32 (temporal attention), got %d", f->n);
        const int64_t ch = 32;             // tile-groups per wave: 32x fewer dbias atomics
        p.pchunk = (int)ch;
        nchunks = (groups(p) + ch - 1) / ch;
    }
    items = nchunks * f->H * tiles_kv(p);
    STG_CHECK(items < (1ll << 31), -2, "stg_attn_bwd: too many work items");
    p.total_items = (int)items;
    return dispatch_d<DkvL>(f->D, p, (hipStream_t)stream);
}

// Two attention problems of one cross-modal pair (h_v <- h_a and h_a <- h_v, Swin_AVE.py:799-808).  When both take the frame-global
// kernels of xattn.hip with one geometry they share a launch (grid.y = 2); otherwise this is two calls.
extern "C" int stg_attn_fwd2(const stg_attn_args* f0, const stg_attn_args* f1, void* stream) {
    STG_CHECK(f0 != nullptr && f1 != nullptr, -1, "stg_attn_fwd2: null args");
    if (xattn_on() && f0->P > 0 && stg_xattn_eligible(f0, true) && stg_xattn_eligible(f1, true) && stg_xattn_pairable(f0, f1)) {
        AttnP p = {};
        int rc = fill(f0, p, "stg_attn_fwd2");
        if (rc) return rc;
        rc = fill(f1, p, "stg_attn_fwd2");
        if (rc) return rc;
        STG_CHECK(f0->O && f1->O && f0->ldo % 4 == 0 && f1->ldo % 4 == 0 && ((((uintptr_t)f0->O) | ((uintptr_t)f1->O)) & 7) == 0, -2, "stg_attn_fwd2: bad O");
        return stg_xattn_fwd2(f0, f1, stream);
    }
    // otherwise the generic kernels: one launch for both directions where they share head dim and packing (ViT-B's pair: 197 video and 49 audio
    // tokens at d = 48 -- two launches of ~17 us each were mostly ramp)
    if (f0->P > 0 && f1->P > 0 && f0->D == f1->D && !(xattn_on() && (stg_xattn_eligible(f0, true) || stg_xattn_eligible(f1, true)))) {
        AttnP p0 = {}, p1 = {};
        int rc = fill(f0, p0, "stg_attn_fwd2");
        if (rc) return rc;
        rc = fill(f1, p1, "stg_attn_fwd2");
        if (rc) return rc;
        STG_CHECK(f0->O && f1->O && f0->ldo % 4 == 0 && f1->ldo % 4 == 0 && ((((uintptr_t)f0->O) | ((uintptr_t)f1->O)) & 7) == 0, -2, "stg_attn_fwd2: bad O");
        const int64_t i0 = groups(p0) * f0->H * tiles_q(p0), i1 = groups(p1) * f1->H * tiles_q(p1);
        if ((p0.pack > 1) == (p1.pack > 1) && i0 < (1ll << 31) && i1 < (1ll << 31)) {
            p0.total_items = (int)i0; p1.total_items = (int)i1;
            return dispatch_d<FwdL>(f0->D, p0, (hipStream_t)stream, &p1);
        }
    }
    const int rc = stg_attn_fwd(f0, stream);
    return rc ? rc : stg_attn_fwd(f1, stream);
}

extern "C" int stg_attn_bwd2(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* stream) {
    STG_CHECK(b0 != nullptr && b1 != nullptr, -1, "stg_attn_bwd2: null args");
    const bool x0 = b0->dV == nullptr && b0->dbias == nullptr && stg_xattn_eligible(&b0->f, true);
    const bool x1 = b1->dV == nullptr && b1->dbias == nullptr && stg_xattn_eligible(&b1->f, true);
    if (xattn_on() && b0->f.P > 0 && x0 && x1 && stg_xattn_pairable(&b0->f, &b1->f)) {
        AttnP p = {};
        int rc = fill(&b0->f, p, "stg_attn_bwd2");
        if (rc) return rc;
        rc = fill(&b1->f, p, "stg_attn_bwd2");
        if (rc) return rc;
        return stg_xattn_bwd2(b0, b1, stream);
    }
    if (b0->f.P > 0 && b1->f.P > 0 && b0->f.D == b1->f.D && !b0->dbias && !b1->dbias &&
        !(xattn_on() && ((b0->dV == nullptr && stg_xattn_eligible(&b0->f, true)) || (b1->dV == nullptr && stg_xattn_eligible(&b1->f, true))))) {
        AttnP p[2] = {};
        const stg_attn_bwd_args* bb[2] = {b0, b1};
        for (int i = 0; i < 2; ++i) {                                 // the checks of stg_attn_bwd, per problem
            const stg_attn_bwd_args* b = bb[i];
            const stg_attn_args* f = &b->f;
            int rc = fill(f, p[i], "stg_attn_bwd2");
            if (rc) return rc;
            STG_CHECK(f->O && f->lse && b->dO && b->dQ && b->dK && b->delta, -1, "stg_attn_bwd2: null O/lse/dO/dQ/dK/delta");
            STG_CHECK(f->ldo % 8 == 0 && b->lddo % 8 == 0 && b->lddq % 4 == 0 && b->lddk % 4 == 0 && (b->dV == nullptr || b->lddv % 4 == 0),
                      -2, "stg_attn_bwd2: bad leading dims");
            STG_CHECK((((uintptr_t)f->O | (uintptr_t)b->dO) & 15) == 0, -2, "stg_attn_bwd2: O/dO must be 16-byte aligned");
            STG_CHECK((((uintptr_t)b->dQ | (uintptr_t)b->dK | (uintptr_t)b->dV) & 7) == 0, -2, "stg_attn_bwd2: dQ/dK/dV must be 8-byte aligned");
            p[i].dO = (const bf16_t*)b->dO; p[i].lddo = b->lddo;
            p[i].dQ = (bf16_t*)b->dQ; p[i].lddq = b->lddq; p[i].dK = (bf16_t*)b->dK; p[i].lddk = b->lddk;
            p[i].dV = (bf16_t*)b->dV; p[i].lddv = b->lddv; p[i].delta = b->delta; p[i].dbias = nullptr;
            p[i].pchunk = 1;
        }
        const int64_t q0 = groups(p[0]) * p[0].H * tiles_q(p[0]), q1 = groups(p[1]) * p[1].H * tiles_q(p[1]);
        const int64_t k0 = groups(p[0]) * p[0].H * tiles_kv(p[0]), k1 = groups(p[1]) * p[1].H * tiles_kv(p[1]);
        if ((p[0].pack > 1) == (p[1].pack > 1) && q0 < (1ll << 31) && q1 < (1ll << 31) && k0 < (1ll << 31) && k1 < (1ll << 31)) {
            p[0].total_items = (int)q0; p[1].total_items = (int)q1;
            int rc = dispatch_d<DqL>(b0->f.D, p[0], (hipStream_t)stream, &p[1]);          // dQ (+ delta) of both directions ...
            if (rc) return rc;
            p[0].total_items = (int)k0; p[1].total_items = (int)k1;
            return dispatch_d<DkvL>(b0->f.D, p[0], (hipStream_t)stream, &p[1]);          // ... then dK / dV of both
        }
    }
    const int rc = stg_attn_bwd(b0, stream);
    return rc ? rc : stg_attn_bwd(b1, stream);
}
