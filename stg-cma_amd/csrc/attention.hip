// Gather-mapped flash attention for the STG-CMA hot path: forward, dQ and dK/dV (see include/stgcma.h).
//
// One kernel family serves every attention in the model -- (shifted-)window W-MSA with relative-position bias and
// shift mask, temporal attention over T frames, ViT multi-head attention, and the single-head unscaled cross-modal
// attention of the adapters (window-level and frame-global, N up to 3136) -- because on this path they differ only
// in (a) how a (problem, token) pair maps to a row of the token tensor and (b) head dim / bias / mask.
//
// Work decomposition: ONE WAVE per (problem, head, 32-row tile); waves never synchronise with each other
// (LDS regions are wave-private, ordered by wavefront-scope fences), so a launch is just a flat list of waves.
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
#include "common.h"
#include "../../include/stgcma.h"

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
    int64_t P; int H; int n; int n_kv;
    float scale;
    const float* bias; int64_t bias_div; int bias_mod;
    const float* mask;
    // backward
    const bf16_t* dO; int64_t lddo;
    bf16_t* dQ; int64_t lddq;
    bf16_t* dK; int64_t lddk;
    bf16_t* dV; int64_t lddv;
    float* delta;
    float* dbias;
    int pchunk;            // problems per wave in the dK/dV kernel (dbias reduction)
    int64_t total_items;
};

__device__ __forceinline__ int64_t tok_row(const int32_t* map, int64_t outer, int G, int n, int64_t p, int i) {
    const int64_t pg = p / G;
    const int idx = (int)(p - pg * G) * n + i;
    return pg * outer + (map ? (int64_t)map[idx] : (int64_t)idx);
}

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// stage a 32 x D tile (rows t0..t0+31 of `src` for problem p, head h) into wave-private LDS, zero-filling rows >= nt
template <int D>
__device__ __forceinline__ void stage_tile(bf16_t* s, const bf16_t* src, int64_t ld, const int32_t* map, int64_t outer,
                                           int G, int nt, int64_t p, int h, int t0, int lane) {
    constexpr int CPR = D / 8;                    // 32 * CPR is a multiple of 64 for every supported D
    uint4 v[(32 * CPR) / 64];
    // loads are unconditional from clamped rows (a `cond ? load : 0` serialises every load behind a branch + vmcnt(0));
    // rows beyond nt are zeroed by a select after the loads have been issued back to back
#pragma unroll
    for (int i = 0; i < (32 * CPR) / 64; ++i) {
        const int idx = lane + 64 * i;
        const int tr = idx / CPR, ch = idx - tr * CPR;
        const int t = t0 + tr;
        const int64_t row = tok_row(map, outer, G, nt, p, t < nt ? t : nt - 1);
        v[i] = *reinterpret_cast<const uint4*>(src + row * ld + h * D + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < (32 * CPR) / 64; ++i) {
        const int idx = lane + 64 * i;
        const int tr = idx / CPR, ch = idx - tr * CPR;
        if (t0 + tr >= nt) v[i] = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(s + tr * D + ch * 8) = v[i];
    }
}

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

__device__ __forceinline__ bf16x8_t pack_frag(const float* x) {
    bf16x8_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (short)f2bf(x[j]);
    return f;
}

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ void __launch_bounds__(256) attn_fwd_kernel(AttnP a) {
    constexpr int KS = D / 16;
    constexpr int DB = (D + 31) / 32;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 32 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= a.total_items) return;
    const int QT = (a.n + 31) >> 5;
    const int qt = (int)(item % QT);
    const int h = (int)((item / QT) % a.H);
    const int64_t p = item / ((int64_t)QT * a.H);
    bf16_t* sV = smem + wave * 32 * D;

    const int q = qt * 32 + r;
    const int qc = q < a.n ? q : a.n - 1;
    const int64_t rowq = tok_row(a.map_q, a.outer_q, a.G, a.n, p, qc);
    bf16x8_t qf[KS];
    {
        const bf16_t* qp = a.Q + rowq * a.ldq + h * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = ld_frag(qp + 16 * s);
    }
    const float* bias_q = nullptr;
    if (a.bias) {
        const int64_t bg = (p / a.bias_div) % a.bias_mod;
        bias_q = a.bias + ((bg * a.H + h) * a.n + qc) * (int64_t)a.n_kv;
    }
    const float* mask_q = a.mask ? a.mask + ((p % a.G) * a.n + qc) * (int64_t)a.n_kv : nullptr;

    f32x16_t o[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b) o[b] = zero16();
    float m = -INFINITY, l = 0.f;

    for (int kv0 = 0; kv0 < a.n_kv; kv0 += 32) {
        lds_fence();
        stage_tile<D>(sV, a.V, a.ldv, a.map_kv, a.outer_kv, a.G, a.n_kv, p, h, kv0, lane);
        const int key = kv0 + r;
        const int kc = key < a.n_kv ? key : a.n_kv - 1;
        const int64_t rowk = tok_row(a.map_kv, a.outer_kv, a.G, a.n_kv, p, kc);
        const bf16_t* kp = a.K + rowk * a.ldk + h * D + 8 * hh;
        f32x16_t st = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) st = MFMA32(ld_frag(kp + 16 * s), qf[s], st);
        float x[16];
        float mt = -INFINITY;
        float add[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) add[reg] = 0.f;
        if (bias_q) {                                  // wave-uniform branches around blocks of independent loads
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int kr = kv0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                add[reg] += bias_q[kr < a.n_kv ? kr : a.n_kv - 1];
            }
        }
        if (mask_q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int kr = kv0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                add[reg] += mask_q[kr < a.n_kv ? kr : a.n_kv - 1];
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int kr = kv0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            float v = st[reg] * a.scale + add[reg];
            v = kr < a.n_kv ? v : -INFINITY;
            x[reg] = v;
            mt = fmaxf(mt, v);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m, mt);
        const float alpha = __expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            x[reg] = __expf(x[reg] - mn);
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

    if (q < a.n) {
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
        if (a.lse && hh == 0) a.lse[(p * a.H + h) * a.n + q] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <int D>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(AttnP a) {
    constexpr int KS = D / 16;
    constexpr int DB = (D + 31) / 32;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 32 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= a.total_items) return;
    const int QT = (a.n + 31) >> 5;
    const int qt = (int)(item % QT);
    const int h = (int)((item / QT) % a.H);
    const int64_t p = item / ((int64_t)QT * a.H);
    bf16_t* sK = smem + wave * 32 * D;

    const int q = qt * 32 + r;
    const int qc = q < a.n ? q : a.n - 1;
    const int64_t rowq = tok_row(a.map_q, a.outer_q, a.G, a.n, p, qc);
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
    const int64_t stat_idx = (p * a.H + h) * a.n + qc;
    if (q < a.n && hh == 0) a.delta[stat_idx] = delta;
    const float lse_q = a.lse[stat_idx];
    const float* bias_q = nullptr;
    if (a.bias) {
        const int64_t bg = (p / a.bias_div) % a.bias_mod;
        bias_q = a.bias + ((bg * a.H + h) * a.n + qc) * (int64_t)a.n_kv;
    }
    const float* mask_q = a.mask ? a.mask + ((p % a.G) * a.n + qc) * (int64_t)a.n_kv : nullptr;

    f32x16_t dq[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b) dq[b] = zero16();

    for (int kv0 = 0; kv0 < a.n_kv; kv0 += 32) {
        lds_fence();
        stage_tile<D>(sK, a.K, a.ldk, a.map_kv, a.outer_kv, a.G, a.n_kv, p, h, kv0, lane);
        const int key = kv0 + r;
        const int kc = key < a.n_kv ? key : a.n_kv - 1;
        const int64_t rowk = tok_row(a.map_kv, a.outer_kv, a.G, a.n_kv, p, kc);
        const bf16_t* kp = a.K + rowk * a.ldk + h * D + 8 * hh;
        const bf16_t* vp = a.V + rowk * a.ldv + h * D + 8 * hh;
        f32x16_t st = zero16(), dpt = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            st = MFMA32(ld_frag(kp + 16 * s), qf[s], st);
            dpt = MFMA32(ld_frag(vp + 16 * s), dof[s], dpt);
        }
        float ds[16];
        float add[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) add[reg] = 0.f;
        if (bias_q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int kr = kv0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                add[reg] += bias_q[kr < a.n_kv ? kr : a.n_kv - 1];
            }
        }
        if (mask_q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int kr = kv0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                add[reg] += mask_q[kr < a.n_kv ? kr : a.n_kv - 1];
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int kr = kv0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            const float v = st[reg] * a.scale + add[reg];
            const float pr = kr < a.n_kv ? __expf(v - lse_q) : 0.f;
            ds[reg] = pr * (dpt[reg] - delta);
        }
        bf16x8_t dsf[2];
        dsf[0] = pack_frag(ds);
        dsf[1] = pack_frag(ds + 8);
        lds_fence();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < DB; ++b) dq[b] = MFMA32(tr_frag<D>(sK, s2, hh, 32 * b + r), dsf[s2], dq[b]);
    }
    if (q < a.n) {
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

// ------------------------------------------------------------------------------------------------ backward: dK, dV (+ dbias)
// The wave owns 32 keys (key on the lane): S[q][key] = Q . K^T and dP[q][key] = dO . V^T come out with the query in
// registers; P and dS are then the B operands of dV^T[d][key] = dO^T . P and dK^T[d][key] = Q^T . dS.
template <int D>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(AttnP a) {
    constexpr int KS = D / 16;
    constexpr int DB = (D + 31) / 32;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * 2 * 32 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= a.total_items) return;
    const int KT = (a.n_kv + 31) >> 5;
    const int kt = (int)(item % KT);
    const int h = (int)((item / KT) % a.H);
    const int64_t pc = item / ((int64_t)KT * a.H);
    bf16_t* sQ = smem + wave * 2 * 32 * D;
    bf16_t* sDO = sQ + 32 * D;

    const int kv0 = kt * 32;
    const int key = kv0 + r;
    const int kc = key < a.n_kv ? key : a.n_kv - 1;
    f32x16_t dbacc = zero16();

    const int64_t pbeg = pc * a.pchunk;
    int64_t pend = pbeg + a.pchunk;
    if (pend > a.P) pend = a.P;
    for (int64_t p = pbeg; p < pend; ++p) {
        const int64_t rowk = tok_row(a.map_kv, a.outer_kv, a.G, a.n_kv, p, kc);
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
        if (a.bias) {
            const int64_t bg = (p / a.bias_div) % a.bias_mod;
            bias_k = a.bias + ((bg * a.H + h) * a.n) * (int64_t)a.n_kv + kc;
        }
        const float* mask_k = a.mask ? a.mask + ((p % a.G) * a.n) * (int64_t)a.n_kv + kc : nullptr;
        const float* lse_p = a.lse + (p * a.H + h) * a.n;
        const float* del_p = a.delta + (p * a.H + h) * a.n;

        f32x16_t dk[DB], dv[DB];
#pragma unroll
        for (int b = 0; b < DB; ++b) { dk[b] = zero16(); dv[b] = zero16(); }

        for (int q0 = 0; q0 < a.n; q0 += 32) {
            lds_fence();
            stage_tile<D>(sQ, a.Q, a.ldq, a.map_q, a.outer_q, a.G, a.n, p, h, q0, lane);
            stage_tile<D>(sDO, a.dO, a.lddo, a.map_q, a.outer_q, a.G, a.n, p, h, q0, lane);
            lds_fence();
            f32x16_t sc = zero16(), dp = zero16();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                sc = MFMA32(ld_frag(sQ + r * D + 16 * s + 8 * hh), kf[s], sc);
                dp = MFMA32(ld_frag(sDO + r * D + 16 * s + 8 * hh), vf[s], dp);
            }
            float pr[16], ds[16], add[16], lse_r[16], del_r[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {               // independent loads, clamped rows, issued back to back
                const int qr = q0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                const int qc = qr < a.n ? qr : a.n - 1;
                lse_r[reg] = lse_p[qc];
                del_r[reg] = del_p[qc];
                add[reg] = 0.f;
            }
            if (bias_k) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int qr = q0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    add[reg] += bias_k[(int64_t)(qr < a.n ? qr : a.n - 1) * a.n_kv];
                }
            }
            if (mask_k) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int qr = q0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    add[reg] += mask_k[(int64_t)(qr < a.n ? qr : a.n - 1) * a.n_kv];
                }
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int qr = q0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                const bool ok = qr < a.n && key < a.n_kv;
                const float v = sc[reg] * a.scale + add[reg];
                const float pv = ok ? __expf(v - lse_r[reg]) : 0.f;
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
        }
        if (key < a.n_kv) {
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
    if (a.dbias && key < a.n_kv) {
        // host guarantees n <= 32 (one query tile) and a single bias group per problem chunk
        const int64_t bg = (pbeg / a.bias_div) % a.bias_mod;
        float* db = a.dbias + ((bg * a.H + h) * a.n) * (int64_t)a.n_kv + key;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int qr = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (qr < a.n) atomicAdd(db + (int64_t)qr * a.n_kv, dbacc[reg]);
        }
    }
}

int fill(const stg_attn_args* f, AttnP& p, const char* who) {
    STG_CHECK(f->Q && f->K && f->V, -1, "%s: null Q/K/V", who);
    STG_CHECK(f->P >= 0 && f->H > 0 && f->n > 0 && f->n_kv > 0 && f->G > 0, -2, "%s: bad shape", who);
    STG_CHECK(f->D == 16 || f->D == 32 || f->D == 48 || f->D == 64 || f->D == 96 || f->D == 128, -2,
              "%s: unsupported head dim %d", who, f->D);
    STG_CHECK(f->ldq % 4 == 0 && f->ldk % 8 == 0 && f->ldv % 8 == 0, -2, "%s: leading dims must be multiples of 8", who);
    STG_CHECK(f->ldq % 8 == 0, -2, "%s: ldq must be a multiple of 8", who);
    STG_CHECK((((uintptr_t)f->Q | (uintptr_t)f->K | (uintptr_t)f->V) & 15) == 0, -2, "%s: Q/K/V must be 16-byte aligned", who);
    STG_CHECK(f->map_q != nullptr || f->outer_q >= (int64_t)f->G * f->n, -2, "%s: outer_q too small for identity map", who);
    STG_CHECK(f->map_kv != nullptr || f->outer_kv >= (int64_t)f->G * f->n_kv, -2, "%s: outer_kv too small for identity map", who);
    if (f->bias) STG_CHECK(f->bias_div > 0 && f->bias_mod > 0, -2, "%s: bad bias grouping", who);
    p.Q = (const bf16_t*)f->Q; p.ldq = f->ldq; p.K = (const bf16_t*)f->K; p.ldk = f->ldk;
    p.V = (const bf16_t*)f->V; p.ldv = f->ldv; p.O = (bf16_t*)f->O; p.ldo = f->ldo; p.lse = f->lse;
    p.map_q = f->map_q; p.map_kv = f->map_kv; p.outer_q = f->outer_q; p.outer_kv = f->outer_kv; p.G = f->G;
    p.P = f->P; p.H = f->H; p.n = f->n; p.n_kv = f->n_kv; p.scale = f->scale;
    p.bias = f->bias; p.bias_div = f->bias ? f->bias_div : 1; p.bias_mod = f->bias ? f->bias_mod : 1; p.mask = f->mask;
    return 0;
}

template <template <int> class Launcher>
int dispatch_d(int D, const AttnP& p, hipStream_t st) {
    switch (D) {
        case 16: return Launcher<16>::run(p, st);
        case 32: return Launcher<32>::run(p, st);
        case 48: return Launcher<48>::run(p, st);
        case 64: return Launcher<64>::run(p, st);
        case 96: return Launcher<96>::run(p, st);
        case 128: return Launcher<128>::run(p, st);
    }
    return -2;
}

inline int grid_of(int64_t items, unsigned& g) {
    const int64_t blocks = (items + 3) / 4;
    if (blocks <= 0 || blocks >= (1ll << 31)) return -2;
    g = (unsigned)blocks;
    return 0;
}

template <int D> struct FwdL {
    static int run(const AttnP& p, hipStream_t st) {
        unsigned g;
        STG_CHECK(grid_of(p.total_items, g) == 0, -2, "attention: grid out of range");
        hipLaunchKernelGGL(attn_fwd_kernel<D>, dim3(g), dim3(256), 0, st, p);
        STG_LAUNCH_CHECK();
        return 0;
    }
};
template <int D> struct DqL {
    static int run(const AttnP& p, hipStream_t st) {
        unsigned g;
        STG_CHECK(grid_of(p.total_items, g) == 0, -2, "attention: grid out of range");
        hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, dim3(g), dim3(256), 0, st, p);
        STG_LAUNCH_CHECK();
        return 0;
    }
};
template <int D> struct DkvL {
    static int run(const AttnP& p, hipStream_t st) {
        unsigned g;
        STG_CHECK(grid_of(p.total_items, g) == 0, -2, "attention: grid out of range");
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<D>, dim3(g), dim3(256), 0, st, p);
        STG_LAUNCH_CHECK();
        return 0;
    }
};

}  // namespace

extern "C" int stg_attn_fwd(const stg_attn_args* f, void* stream) {
    STG_CHECK(f != nullptr, -1, "stg_attn_fwd: null args");
    AttnP p = {};
    int rc = fill(f, p, "stg_attn_fwd");
    if (rc) return rc;
    STG_CHECK(f->O != nullptr && f->ldo % 4 == 0 && (((uintptr_t)f->O) & 7) == 0, -2, "stg_attn_fwd: bad O");
    if (f->P == 0) return 0;
    p.total_items = f->P * f->H * (int64_t)((f->n + 31) / 32);
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
    p.dO = (const bf16_t*)b->dO; p.lddo = b->lddo;
    p.dQ = (bf16_t*)b->dQ; p.lddq = b->lddq; p.dK = (bf16_t*)b->dK; p.lddk = b->lddk;
    p.dV = (bf16_t*)b->dV; p.lddv = b->lddv; p.delta = b->delta; p.dbias = b->dbias;
    p.pchunk = 1;
    p.total_items = f->P * f->H * (int64_t)((f->n + 31) / 32);
    rc = dispatch_d<DqL>(f->D, p, (hipStream_t)stream);
    if (rc) return rc;
    int64_t nchunks = f->P;
    if (b->dbias) {
        STG_CHECK(f->bias != nullptr, -2, "stg_attn_bwd: dbias without bias");
        STG_CHECK(f->n <= 32, -2, "stg_attn_bwd: dbias needs n <= 32 (temporal attention), got %d", f->n);
        // problems of one chunk must share a bias group: chunk size divides bias_div
        int64_t ch = 64;
        while (ch > 1 && (f->bias_div % ch) != 0) ch >>= 1;
        p.pchunk = (int)ch;
        nchunks = (f->P + ch - 1) / ch;
    }
    p.total_items = nchunks * f->H * (int64_t)((f->n_kv + 31) / 32);
    return dispatch_d<DkvL>(f->D, p, (hipStream_t)stream);
}
