// bf16 MFMA GEMM with fused epilogues:  C[M,N] = epi(A[M,K] . W[N,K]^T)   (see include/stgcma.h: stg_gemm_nt)
//
// Shape of the work: M = token rows of the fused audio+video tensor (up to 2.0 M at B=32),
// N,K in {16..4096}.  Both operands are K-contiguous, which is exactly the per-lane layout of
// v_mfma_f32_16x16x32_bf16 (lane l holds row l&15, k = 8*(l>>4)..+7), so tiles go global -> VGPR -> LDS (16-byte
// chunks, XOR-swizzled) -> ds_read_b128 fragments with no transposes anywhere.
//
// The MFMA is issued "swapped": first operand = W fragment, second = A fragment, so the accumulator holds
// D[n][m] with m on the lane and 4 consecutive n in registers => the epilogue reads/writes 8-byte bf16x4 runs
// of a C row (bias, saved pre-activation, activation-gradient source, residuals are all row-major [M,N]).
//
// Block tile 128x128x64, 256 threads = 2x2 waves of 64x64, LDS double buffered (64 KiB => 2 blocks / CU),
// global prefetch of tile k+1 into registers while tile k is in the MFMA loop (one barrier per k-tile).
// Block ids are remapped so that the N-tiles sharing one A row-panel run on the same XCD (its L2 holds the panel).
#include <type_traits>
#include "common.h"
#include "../../include/stgcma.h"

// Timing-only ablations exist in the diagnostics build alone (make diag -> libstgcma_hip_diag.so, -DSTG_GEMM_DIAG): the product
// kernels carry no run-time debug branches.
#ifdef STG_GEMM_DIAG
#define DIAG_ON(p, v) ((p).dbg == (v))
#else
#define DIAG_ON(p, v) false
#endif

#ifdef STG_GEMM_DIAG
// clock stamps around the 8-phase main loop (option gemm_dbg = 4): per workgroup {s_memtime, s_memrealtime} before and after, written
// to a buffer of their own that no other code reads (MI355X_MICROARCH.md, DVFS give-back item 6); read with stg_diag_read_stamps
__device__ unsigned long long stg_diag_stamps[4 * 8192];
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int CPR = BK / 8;  // 16-byte chunks per LDS row

struct GemmParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    void* C; int64_t ldc; int c_f32;
    const float* bias;
    float alpha;
    int act;
    bf16_t* dact; int64_t ldp;                // EV_*8 variants: uint8 codes, ldp / ldd in bytes
    const bf16_t* dact_src; int64_t ldd;
    const float* row_scale; int64_t rs_outer; int64_t rs_inner;
    const void* res1; int64_t ldr1; int res1_f32;
    const void* res2; int64_t ldr2; int res2_f32;
    int64_t M; int N; int K;
    int nbm, nbn;
    int ntl;           // multi-tile 8-phase kernel: consecutive column tiles per workgroup (divides nbn)
    int64_t split_m; const bf16_t* W2; const float* bias2;           // two row groups (see stgcma.h): rows >= split_m use W2 / bias2
    int vec_ok;
    int epi_variant;   // EV_* (row-layout epilogue), or -1: element-wise fallback
    int conv_H, conv_W, conv_d, conv_C; const bf16_t* conv_zero;     // implicit 3x3 convolution (conv_H > 0), see stgcma.h
    int batch; int64_t a_bstride, w_bstride, c_bstride;              // batched mode (blockIdx.y = problem), see stgcma.h
#ifdef STG_GEMM_DIAG
    int dbg;   // diagnostics build only (libstgcma_hip_diag.so, tools/): 1 = no in-loop tile loads, 2 = no MFMA work, 3 = no epilogue
#endif
};

__device__ __forceinline__ float ld_res1(const void* res, int f32, int64_t off) {
    return f32 ? reinterpret_cast<const float*>(res)[off] : bf2f(reinterpret_cast<const bf16_t*>(res)[off]);
}

__device__ __forceinline__ int swz(int row, int c) { return (c ^ (row & 7)); }

struct AccTile { f32x4_t v[4][4]; };   // passed BY VALUE: a by-reference accumulator array ends up mirrored in scratch

// Element-wise fallback epilogue (N % 8 != 0 or unaligned operands: the 29-column classifier head): lane owns row m and
// 4 consecutive columns n per (ni, mi) accumulator tile; scalar, bounds-checked, every option a run-time test.
__device__ __forceinline__ void gemm_epilogue_elems(const GemmParams& p, const AccTile accs, int64_t m0, int n0, int wm, int wn,
                                                    int lrow, int lk) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 16 + lrow;
        if (m >= p.M) continue;
        float rs = 1.0f;
        if (p.row_scale) rs = p.row_scale[(m / p.rs_outer) * p.rs_inner + (m % p.rs_inner)];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + 4 * lk;
            for (int r = 0; r < 4; ++r) {
                const int nn = n + r;
                if (nn >= p.N) break;
                float v = accs.v[ni][mi][r] * p.alpha;
                if (p.bias) v += p.bias[nn];
                float dv = 1.0f;
                act_both(p.act, v, v, dv);
                if (p.dact && p.act) p.dact[m * p.ldp + nn] = f2bf(dv);
                if (p.dact_src) v *= bf2f(p.dact_src[m * p.ldd + nn]);
                v *= rs;
                if (p.res1) v += ld_res1(p.res1, p.res1_f32, m * p.ldr1 + nn);
                if (p.res2) v += ld_res1(p.res2, p.res2_f32, m * p.ldr2 + nn);
                if (p.c_f32) reinterpret_cast<float*>(p.C)[m * p.ldc + nn] = v;
                else reinterpret_cast<bf16_t*>(p.C)[m * p.ldc + nn] = f2bf(v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Row-layout epilogue (every 8-aligned call, i.e. everything but the 29-column head).
// In the accumulator layout a wave-wide store touches 16 rows x 32 bytes: measured, the bare C store of a K = 512 GEMM
// cost as much as its whole MFMA loop, and the residual-add variants ran at ~2.6 TB/s.  So each wave transposes its
// 64 x 64 fp32 sub-tile through a private 8 KiB LDS region (32 rows at a time, 16-byte chunks XOR-swizzled by row so both
// the accumulator-layout writes and the row-layout reads are conflict-free) and then owns, per instruction, 8 rows x 64
// columns with 8 consecutive columns per lane: every global access of the epilogue (C, saved derivative, derivative
// source, residuals) is a 16- or 32-byte piece of a 128/256-byte contiguous row segment.
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

struct Row8 { uint4 lo, hi; };     // 8 values: bf16 in lo, or fp32 in lo (0..3) + hi (4..7)

__device__ __forceinline__ Row8 ld_row8(const void* base, int f32, int64_t off) {
    Row8 r;
    if (f32) {
        const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(base) + off);
        r.lo = q[0]; r.hi = q[1];
    } else {
        r.lo = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(base) + off);
        r.hi = r.lo;
    }
    return r;
}
__device__ __forceinline__ void row8_to_f32(const Row8& r, int f32, float* v) {
    if (f32) {
        v[0] = __uint_as_float(r.lo.x); v[1] = __uint_as_float(r.lo.y); v[2] = __uint_as_float(r.lo.z); v[3] = __uint_as_float(r.lo.w);
        v[4] = __uint_as_float(r.hi.x); v[5] = __uint_as_float(r.hi.y); v[6] = __uint_as_float(r.hi.z); v[7] = __uint_as_float(r.hi.w);
    } else {
        const uint32_t w[4] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = __uint_as_float(w[j] << 16);
            v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ uint4 pack_row8(const float* t) {
    return make_uint4(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]), pack_bf2(t[4], t[5]), pack_bf2(t[6], t[7]));
}

// The option set is a compile-time VARIANT picked on the host (epi_variant): with every option a run-time test, the
// straight-line code of one 8-row step carried ~10 wave-uniform branches and the register allocation of all paths at once
// (spills at the 128-VGPR budget of 4 waves / SIMD) -- measured 555 us vs 369 us for the plain store of a
// 125440 x 2048 x 512 GEMM.  The variants are the signatures that carry the training step (tools/step_gemm_shapes.py);
// anything else takes EV_GENERIC, which keeps every test at run time.  Partial tiles are predicated, not branched.
enum { EV_GENERIC = 0, EV_PLAIN, EV_GELU, EV_QGELU, EV_DSRC, EV_R16, EV_BRQ, EV_GELU8, EV_QGELU8, EV_DSRC8 };   // *8: 8-bit saved derivative

// saved-derivative byte code (STG_U8_LIN): code = round((d + 0.14) * 200), d = code * 0.005 - 0.14
__device__ __forceinline__ uint2 pack_d8(const float* d) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lo = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d[j], 200.f, 28.f), j, lo);          // v_cvt_pk_u8_f32: round to nearest, saturate to 0..255
        hi = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d[4 + j], 200.f, 28.f), j, hi);
    }
    return make_uint2(lo, hi);
}
__device__ __forceinline__ void unpack_d8(const uint2 q, float* v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = fmaf((float)((q.x >> (8 * j)) & 0xffu), 0.005f, -0.14f);
        v[4 + j] = fmaf((float)((q.y >> (8 * j)) & 0xffu), 0.005f, -0.14f);
    }
}

// FULL: the wave's 64 x 64 piece lies entirely inside C (decided once per wave by the caller): no bounds tests at all.  With the tests the
// compiler wraps every 8-row step in an exec-mask branch (s_and_saveexec / s_cbranch_execz), sinks the step's LDS reads into it, and --
// because its wait-count bookkeeping is merged conservatively at every join -- puts `s_waitcnt vmcnt(0)` in front of every step's first
// use of the bias registers: each wave then waits for its PREVIOUS global store to complete before it issues the next one (16 steps x a
// full store latency per wave: that, not the memory system, was the 8.9 us "un-overlapped epilogue" of a 256 x 256 tile; a pure-store
// kernel writes the same tile in 5 us with every CU storing at once, tools/probe/store_probe.hip).
template <int V, bool FULL>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmParams& p, const AccTile accs, int64_t m0, int n0, int wm, int wn,
                                                   int lane, float* stg, int ngap = 0) {
    constexpr bool G = V == EV_GENERIC;
    const bool alpha_on = G ? p.alpha != 1.0f : false;
    constexpr bool D8 = V == EV_GELU8 || V == EV_QGELU8 || V == EV_DSRC8;     // 8-bit saved derivative (byte leading dimensions)
    const int act = G ? p.act : ((V == EV_GELU || V == EV_GELU8) ? (int)STG_ACT_GELU : (V == EV_QGELU || V == EV_QGELU8) ? (int)STG_ACT_QUICKGELU : (int)STG_ACT_NONE);
    const bool dsrc_on = G ? p.dact_src != nullptr : (V == EV_DSRC || V == EV_DSRC8);
    const bool rs_on = G ? p.row_scale != nullptr : false;
    const bool r1_on = G ? p.res1 != nullptr : (V == EV_R16 || V == EV_BRQ);
    const int r1_f32 = G ? p.res1_f32 : 0;
    const bool r2_on = G ? p.res2 != nullptr : V == EV_BRQ;
    const int r2_f32 = G ? p.res2_f32 : 1;
    const bool c_f32 = G ? p.c_f32 != 0 : V == EV_BRQ;

    const int lrow = lane & 15, lk = lane >> 4;
    const int rr = lane >> 3, cc = lane & 7;
    int n = n0 + wn * 64 + cc * 8 + ((cc & 4) ? ngap : 0);     // ngap: the tile's right 32 columns sit ngap further right (8-phase kernel)
    const bool col_ok = FULL || n < p.N;
    if (!col_ok) n = 0;                                // clamped: loads stay unconditional, stores are predicated
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = 0.f;
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w;
        bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    // Row operands (saved derivative, residuals) are loaded for ALL steps of a group BEFORE the group's first store: vmcnt counts
    // loads and stores together, in order, so a wait for a load that was issued behind a store is also a wait for that store's
    // completion (one full store latency per 8-row step: see FULL above).  Group = the whole 64 x 64 piece (8 steps) for the byte-wide
    // derivative, one 32-row half for the bf16 operands, a single step where fp32 residuals would not fit the 128-VGPR kernels.
    constexpr int LG = (V == EV_DSRC8) ? 8 : (V == EV_DSRC || V == EV_R16) ? 4 : 1;     // steps per load group (register budget)
    Row8 qdv[2][4], q1v[2][4], q2v[2][4];
    uint2 qd8v[2][4];
    auto row_of = [&](int half, int it) -> int64_t {
        int64_t m = m0 + wm * 64 + half * 32 + it * 8 + rr;
        if (!FULL) m = m < p.M ? m : p.M - 1;
        return m;
    };
    auto load_step = [&](int half, int it) {
        const int64_t m = row_of(half, it);
        if (D8 && dsrc_on) qd8v[half][it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(p.dact_src) + m * p.ldd + n);
        else if (dsrc_on) qdv[half][it] = ld_row8(p.dact_src, 0, m * p.ldd + n);
        if (r1_on) q1v[half][it] = ld_row8(p.res1, r1_f32, m * p.ldr1 + n);
        if (r2_on) q2v[half][it] = ld_row8(p.res2, r2_f32, m * p.ldr2 + n);
    };
    if (LG == 8) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int it = 0; it < 4; ++it) load_step(hh, it);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (LG == 4) {
#pragma unroll
            for (int it = 0; it < 4; ++it) load_step(half, it);
        }
        if (half) lds_wave_sync();                     // the previous half's reads are done before it is overwritten
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4_t*>(stg + (mi2 * 16 + lrow) * 64 + (((ni * 4 + lk) ^ lrow) << 2)) = accs.v[ni][half * 2 + mi2];
        lds_wave_sync();                               // LDS-only fences: the row operands' global loads may be hoisted above
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + rr;
            const int64_t m = row_of(half, it);
            const bool ok = FULL || (m0 + wm * 64 + half * 32 + row < p.M && col_ok);
            if (LG == 1) load_step(half, it);
            const Row8& qd = qdv[half][it];
            const Row8& q1 = q1v[half][it];
            const Row8& q2 = q2v[half][it];
            const uint2 qd8 = qd8v[half][it];
            const float* src = stg + row * 64;
            const int c0 = (2 * cc) ^ (row & 15);
            const f32x4_t u0 = *reinterpret_cast<const f32x4_t*>(src + (c0 << 2));
            const f32x4_t u1 = *reinterpret_cast<const f32x4_t*>(src + ((c0 ^ 1) << 2));
            float t[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
            if (alpha_on) {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] *= p.alpha;
            }
            if (!G || p.bias) {                        // variants always add (zeros without a bias): no branch
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] += bias[j];
            }
            if (act == STG_ACT_GELU) {
                float d[8];
                if (!G && !c_f32) {                    // bf16 destinations: the polynomial form (common.h)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {   // packed fp32 pairs
                        f32x2_t y2, d2;
                        gelu_pw_both((f32x2_t){t[j], t[j + 1]}, y2, d2);
                        t[j] = y2.x; t[j + 1] = y2.y; d[j] = d2.x; d[j + 1] = d2.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) gelu_fast(t[j], t[j], d[j]);
                }
                if (D8) { if (ok) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(p.dact) + m * p.ldp + n) = pack_d8(d); }
                else if (p.dact && ok) *reinterpret_cast<uint4*>(p.dact + m * p.ldp + n) = pack_row8(d);
            } else if (act == STG_ACT_QUICKGELU) {
                float d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) quick_gelu_fast(t[j], t[j], d[j]);
                if (D8) { if (ok) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(p.dact) + m * p.ldp + n) = pack_d8(d); }
                else if (p.dact && ok) *reinterpret_cast<uint4*>(p.dact + m * p.ldp + n) = pack_row8(d);
            }
            float v[8];
            if (dsrc_on) {
                if (D8) unpack_d8(qd8, v);
                else row8_to_f32(qd, 0, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] *= v[j];
            }
            if (rs_on) {
                const float rs = p.row_scale[(m / p.rs_outer) * p.rs_inner + (m % p.rs_inner)];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] *= rs;
            }
            if (r1_on) {
                row8_to_f32(q1, r1_f32, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] += v[j];
            }
            if (r2_on) {
                row8_to_f32(q2, r2_f32, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] += v[j];
            }
            if (ok) {
                if (c_f32) {
                    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + m * p.ldc + n);
                    dst[0] = make_float4(t[0], t[1], t[2], t[3]);
                    dst[1] = make_float4(t[4], t[5], t[6], t[7]);
                } else {
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n) = pack_row8(t);
                }
            }
        }
    }
}

// smem: the block's tile buffer (>= 32 KiB), free once every wave is past the main loop's final barrier
__device__ __forceinline__ void gemm_epilogue_dispatch(const GemmParams& p, const AccTile accs, int64_t m0, int n0, int wm,
                                                       int wn, int lane, float* stg, int ngap = 0) {
    if (DIAG_ON(p, 3) && accs.v[0][0][0] != 12345.678f) return;
    // wave-uniform: does the wave's 64 x 64 piece (its right half ngap further right) lie inside C?
    const bool full = m0 + wm * 64 + 64 <= p.M && n0 + wn * 64 + 64 + ngap <= p.N;
#define STG_EPI_CASE(V) case V: if (full) gemm_epilogue_rows<V, true>(p, accs, m0, n0, wm, wn, lane, stg, ngap); \
                                else gemm_epilogue_rows<V, false>(p, accs, m0, n0, wm, wn, lane, stg, ngap); break;
    switch (p.epi_variant) {
        STG_EPI_CASE(EV_PLAIN)
        STG_EPI_CASE(EV_GELU)
        STG_EPI_CASE(EV_QGELU)
        STG_EPI_CASE(EV_DSRC)
        STG_EPI_CASE(EV_R16)
        STG_EPI_CASE(EV_BRQ)
        STG_EPI_CASE(EV_GELU8)
        STG_EPI_CASE(EV_QGELU8)
        STG_EPI_CASE(EV_DSRC8)
        STG_EPI_CASE(EV_GENERIC)
        default: gemm_epilogue_elems(p, accs, m0, n0, wm, wn, lane & 15, lane >> 4); break;     // unaligned / N % 8 != 0
    }
#undef STG_EPI_CASE
}

__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (BM + BN) * BK];

    // XCD-aware bijective remap: consecutive logical tiles (same A panel) land on one XCD.
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.nbn, bn = bid % p.nbn;
    const int64_t m0 = (int64_t)bm * BM;
    const int n0 = bn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, lk = lane >> 4;

    constexpr int ITERS = BM * CPR / 256;  // 4 chunks of A and 4 of W per thread per k-tile
    uint4 ra[ITERS], rb[ITERS];

    // Loads are UNCONDITIONAL from clamped (always in-bounds) addresses; out-of-range rows of A / W only feed output
    // rows / columns that are never stored, and the K tail (K % 64 != 0) is zeroed by a value select AFTER the load.
    // (A `cond ? load : 0` makes hipcc branch around every load and wait vmcnt(0) each: 8 dependent HBM round trips
    // per k-tile -- measured 10x slower.)
    const bf16_t* pa[ITERS];
    const bf16_t* pw[ITERS];
    const int kc = (tid % CPR) * 8;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int row = (tid + i * 256) / CPR;
        int64_t gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        pa[i] = p.A + gm * p.lda;
        pw[i] = p.W + (int64_t)gn * p.ldw;
    }
    const bool kfull = (p.K % BK) == 0;

    auto gload = [&](int kt) {
        int gk = kt * BK + kc;
        const bool kok = kfull || gk < p.K;
        gk = kok ? gk : 0;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            ra[i] = *reinterpret_cast<const uint4*>(pa[i] + gk);
            rb[i] = *reinterpret_cast<const uint4*>(pw[i] + gk);
        }
        if (!kfull) {
            const unsigned msk = kok ? 0xffffffffu : 0u;
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                ra[i].x &= msk; ra[i].y &= msk; ra[i].z &= msk; ra[i].w &= msk;
                rb[i].x &= msk; rb[i].y &= msk; rb[i].z &= msk; rb[i].w &= msk;
            }
        }
    };
    auto swrite = [&](int stage) {
        bf16_t* sA = smem + stage * (BM + BN) * BK;
        bf16_t* sW = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int id = tid + i * 256;
            const int row = id / CPR, c = id % CPR;
            *reinterpret_cast<uint4*>(sA + row * BK + swz(row, c) * 8) = ra[i];
            *reinterpret_cast<uint4*>(sW + row * BK + swz(row, c) * 8) = rb[i];
        }
    };

    AccTile accs;  // [n tile][m tile]
    auto& acc = accs.v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    swrite(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const bf16_t* sA = smem + stage * (BM + BN) * BK;
        const bf16_t* sW = sA + BM * BK;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t af[4], wf[4];
            const int c = 4 * s + lk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ar = wm * 64 + i * 16 + lrow;
                const int wr = wn * 64 + i * 16 + lrow;
                af[i] = *reinterpret_cast<const bf16x8_t*>(sA + ar * BK + swz(ar, c) * 8);
                wf[i] = *reinterpret_cast<const bf16x8_t*>(sW + wr * BK + swz(wr, c) * 8);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        if (kt + 1 < nk) swrite(stage ^ 1);
        __syncthreads();
    }

    gemm_epilogue_dispatch(p, accs, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem) + (wm * 2 + wn) * 2048);
}

// ------------------------------------------------------------------------------------------------
// Fast path (K % 64 == 0): tiles go HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR staging, no ds_write --
// the register-staged variant above is LDS-WRITE bound: 32 KB of ds_write_b128 per k-tile at ~79 B/clk/CU).
// One wave-instruction writes 1 KiB = 8 rows x 128 B linearly (LDS dest = wave-uniform base + lane*16), so the XOR
// swizzle lives on the per-lane SOURCE address: LDS position `pos` of row r receives global chunk pos ^ (r & 7), and the
// fragment reads apply the same involution.  Rows beyond M / N are clamped (their products are never stored).
// 2-stage pipeline: issue tile t+1's DMA, run tile t's MFMAs, then vmcnt(0) + barrier.
// CONV: implicit 3x3 convolution (its own instantiation: the plain kernel sits exactly at its 128-VGPR budget)
// KTAIL: K % 64 != 0 (K % 8 == 0): the chunks of the last k-tile beyond K come from the zero line (Swin-L adapters, K = 96)
template <int NST, bool CONV = false, bool BATCH = false, bool KTAIL = false>   // NST = 2: double-buffered LDS (64 KiB, 2 blocks / CU);  NST = 1: single buffer (32 KiB, up to 4 blocks / CU)
__global__ void __launch_bounds__(256, NST == 1 ? 4 : 2) gemm_nt_glds_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[NST * (BM + BN) * BK];
    if (BATCH) {                                             // blockIdx.y = problem: same shape, strided operands
        p.A += (int64_t)blockIdx.y * p.a_bstride;
        p.W += (int64_t)blockIdx.y * p.w_bstride;
        p.C = p.c_f32 ? (void*)(reinterpret_cast<float*>(p.C) + (int64_t)blockIdx.y * p.c_bstride)
                      : (void*)(reinterpret_cast<bf16_t*>(p.C) + (int64_t)blockIdx.y * p.c_bstride);
    }
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.nbn, bn = bid % p.nbn;
    const int64_t m0 = (int64_t)bm * BM;
    const int n0 = bn * BN;
    if (!CONV && !BATCH && p.split_m > 0 && m0 >= p.split_m) { p.W = p.W2; p.bias = p.bias2; }   // second row group (block-uniform)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, lk = lane >> 4;

    // this lane's 4 + 4 DMA pieces per k-tile: LDS chunk q = (wave*4 + j)*64 + lane  ->  row q>>3, position q&7
    const bf16_t* pa[4];
    const bf16_t* pw[4];
    int py[CONV ? 4 : 1], px[CONV ? 4 : 1];          // implicit convolution: the pixel (y, x) of this lane's four A rows
    int pc[(KTAIL || CONV) ? 4 : 1];                 // k-tail / few-channel convolution: this lane's source chunk (0..7) of each piece
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave * 4 + j) * 64 + lane;
        const int row = q >> 3, c = (q & 7) ^ (row & 7);
        int64_t gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        pa[j] = p.A + gm * p.lda + c * 8;
        pw[j] = p.W + (int64_t)gn * p.ldw + c * 8;
        if (KTAIL || CONV) pc[j] = c;
        if (CONV) {
            const int rem = (int)(gm % ((int64_t)p.conv_H * p.conv_W));
            py[j] = rem / p.conv_W;
            px[j] = rem - py[j] * p.conv_W;
        }
    }
    auto stage = [&](int buf, int kt) {
        bf16_t* sA = smem + buf * (BM + BN) * BK;
        bf16_t* sW = sA + BM * BK;
        if (CONV) {
            // k-tile kt of the im2col image = channels c0 .. c0 + 63 of tap (kh, kw): the DMA source of a row is the same
            // channel run of the neighbouring pixel (a constant element offset from the row's own pointer) or the zero line
            const int k0 = kt * BK;
            const int tap = k0 / p.conv_C, c0 = k0 - tap * p.conv_C;
            const int dy = (tap / 3 - 1) * p.conv_d, dx = (tap % 3 - 1) * p.conv_d;
            const int64_t shift = ((int64_t)dy * p.conv_W + dx) * p.lda + c0;
            if (p.conv_C < BK) {
                // few channels (conv_C = 8 / 16 / 32: a k-tile spans several taps, and K = 9 conv_C ends inside the last one): the
                // tap is a property of the lane's 16-byte piece; pieces at or beyond K read the zero line on both operands
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ke = k0 + pc[j] * 8;
                    const int tj = ke / p.conv_C, cj = ke - tj * p.conv_C;
                    const int dyj = (tj / 3 - 1) * p.conv_d, dxj = (tj % 3 - 1) * p.conv_d;
                    const int yy = py[j] + dyj, xx = px[j] + dxj;
                    const bool kin = ke < p.K;
                    const bool in = kin && yy >= 0 && yy < p.conv_H && xx >= 0 && xx < p.conv_W;
                    const bf16_t* src = in ? pa[j] - pc[j] * 8 + ((int64_t)dyj * p.conv_W + dxj) * p.lda + cj : p.conv_zero;
                    const bf16_t* sw = kin ? pw[j] + kt * BK : p.conv_zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(sA + (wave * 4 + j) * 512), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sw,
                                                     (__attribute__((address_space(3))) void*)(sW + (wave * 4 + j) * 512), 16, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yy = py[j] + dy, xx = px[j] + dx;
                const bool in = yy >= 0 && yy < p.conv_H && xx >= 0 && xx < p.conv_W;
                const bf16_t* src = in ? pa[j] + shift : p.conv_zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(sA + (wave * 4 + j) * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pw[j] + kt * BK),
                                                 (__attribute__((address_space(3))) void*)(sW + (wave * 4 + j) * 512), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool kin = !KTAIL || kt * BK + pc[KTAIL ? j : 0] * 8 < p.K;
            const bf16_t* sa = kin ? pa[j] + kt * BK : p.conv_zero;
            const bf16_t* sw = kin ? pw[j] + kt * BK : p.conv_zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                             (__attribute__((address_space(3))) void*)(sA + (wave * 4 + j) * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sw,
                                             (__attribute__((address_space(3))) void*)(sW + (wave * 4 + j) * 512), 16, 0, 0);
        }
    };

    AccTile accs;
    auto& acc = accs.v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (KTAIL || CONV) ? (p.K + BK - 1) / BK : p.K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = NST == 2 ? (kt & 1) : 0;
        if (NST == 2 && kt + 1 < nk && !DIAG_ON(p, 1)) stage(cur ^ 1, kt + 1);
        const bf16_t* sA = smem + cur * (BM + BN) * BK;
        const bf16_t* sW = sA + BM * BK;
        if (!DIAG_ON(p, 2))
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t af[4], wf[4];
            const int c = 4 * s + lk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ar = wm * 64 + i * 16 + lrow;
                const int wr = wn * 64 + i * 16 + lrow;
                af[i] = *reinterpret_cast<const bf16x8_t*>(sA + ar * BK + swz(ar, c) * 8);
                wf[i] = *reinterpret_cast<const bf16x8_t*>(sW + wr * BK + swz(wr, c) * 8);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        __syncthreads();
        if (NST == 1 && kt + 1 < nk) {      // single buffer: refill after every wave is done reading, then wait for the DMA
            stage(0, kt + 1);
            __syncthreads();
        }
    }
    gemm_epilogue_dispatch(p, accs, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem) + (wm * 2 + wn) * 2048);
}

// ------------------------------------------------------------------------------------------------
// Block tile of the 8-phase kernels below.  (The simple-loop 256 x 256 kernel of round 1, gemm_nt_big_kernel -- one barrier per k-tile, a vmcnt(0) in
// front of it -- was their bit-identical predecessor and A/B arm until round 6.)
constexpr int GBM = 256, GBN = 256;

// 16 bytes of zeros for the NX forms of the 8-phase kernels below (a zero-filled k-tile: every lane's LDS-DMA reads these)
__device__ __attribute__((aligned(16))) const uint32_t gemm_zero16[4] = {0u, 0u, 0u, 0u};

// ------------------------------------------------------------------------------------------------
// 8-phase variant of the 256 x 256 x 64 kernel (K % 64 == 0, N % 256 == 0, row-layout epilogue only).
// Same block tile and the same MFMA / k order as the removed simple-loop kernel (results were bit-identical), different pipeline: the
// simple loop stalled every k-tile on the vmcnt(0) in front of its barrier while the next tile's DMA is in flight and
// leaves the MFMA pipe idle while fragments are read.  Here
//   * a k-tile is staged as FOUR 16 KiB half-tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255) into 2 x 4 LDS slots, one
//     half-tile per phase, two half-tiles always in flight across the barriers (counted vmcnt, raw s_barrier);
//   * a wave's 128 x 64 output is interleaved over the halves -- rows {wr*64.. of half 0} + {wr*64.. of half 1}, columns
//     {wc*32.. of half 0} + {wc*32.. of half 1} -- so phase p of a k-tile computes one 64 x 32 x 64 quadrant from
//     (A0,W0) (A0,W1) (A1,W1) (A1,W0): 16 MFMAs behind 12 / 4 / 8 / 4 fragment reads, and each slot is dead early
//     (A0 after phase 0, W1 after phase 1, A1 after phase 2, W0 after phase 3);
//   * waves 4-7 run one barrier behind waves 0-3: one group reads fragments / issues DMA while the other runs its MFMA cluster.
// Schedule of k-tile t (slot parity t & 1), phase p: reads as above; DMA issue p0: A1(t+1)  p1: W0(t+1)  p2: A0(t+2)  p3: W1(t+2);
// in p3, before its first barrier, s_waitcnt vmcnt(4): everything but the two half-tiles just issued has landed; in p2 vmcnt(8): A0(t+1),
// which p3 pre-reads, has landed (round-4 fix).
// RAW: A1(t+1), W0(t+1) are retired by that wait >= 1 barrier before any wave reads them (phase 0 / 2 of t+1, the late group
// included); A0(t+2), W1(t+2) by the wait of tile t+1.  WAR: a slot is re-staged >= 3 barrier intervals after its last ds_read
// (late group included): A0 p0 -> p2, W1 p1 -> p3, A1 p2 -> p0', W0 p3 -> p1'.
// NX (round 6; Swin-L's widths 192 / 384 / 576 / 1152): N % 64 == 0 instead of N % 256 == 0 and K % 64 == 0 instead of K % 128 == 0.
//   * the last column tile may hold 64 / 128 / 192 valid columns: W rows beyond N are clamped to row N - 1 (their products are never
//     stored); when its W1 half (columns 128..255) is entirely out of range, the two phases that multiply by W1 skip their MFMA clusters
//     (a workgroup-uniform branch; barriers and DMAs unchanged, so the counted waits stay exact) -- N = 1152 costs 4.5 tiles, not 5;
//   * an odd number of k-tiles (K = 192, 576) is rounded up with a ZERO k-tile: its DMAs read gemm_zero16 (all lanes the same 16 bytes).
// NX = false compiles to the round-5 kernel (every NX test is a compile-time constant).
template <bool NX>
__global__ void __launch_bounds__(512, 1) gemm_nt_8ph_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem8[];            // 8 slots x 128 rows x 64 bf16 = 128 KiB
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.nbn, bn = bid % p.nbn;
    const int64_t m0 = (int64_t)bm * GBM;
    const int n0 = bn * GBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lrow = lane & 15, lk = lane >> 4;
    constexpr int SLOT = 128 * BK;                       // bf16 elements per half-tile slot
    enum { HA0 = 0, HA1 = 1, HW0 = 2, HW1 = 3 };

    // DMA pieces of a half-tile: chunk q = j * 512 + tid (j = 0, 1) -> row q >> 3 (0..127), position q & 7 <- source chunk (q & 7) ^ (row & 7)
    uint32_t oa[2][2], ow[2][2];                         // [half][j] byte offsets from the block's A / W base rows
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = j * 512 + tid;
            const int row = hf * 128 + (q >> 3), c = (q & 7) ^ ((q >> 3) & 7);
            const int64_t rm = p.M - 1 - m0;
            const int ra = row < rm ? row : (int)rm;     // rows beyond M are clamped (their products are never stored)
            const int rn = p.N - 1 - n0;
            const int rw = NX ? (row < rn ? row : rn) : row;                     // NX = false: N % 256 == 0, always in range
            oa[hf][j] = (uint32_t)(((int64_t)ra * p.lda + c * 8) * 2);
            ow[hf][j] = (uint32_t)(((int64_t)rw * p.ldw + c * 8) * 2);
        }
    const char* baseA = reinterpret_cast<const char*>(p.A + m0 * p.lda);
    const char* baseW = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
    const int nkr = p.K / BK;                            // k-tiles that exist
    const int nk = NX ? (nkr + 1) & ~1 : nkr;            // k-tiles the ring walks (even)
    const bool w1_live = NX ? p.N - n0 > 128 : true;     // does the tile's W1 half hold any column of C?
    const char* zsrc = reinterpret_cast<const char*>(gemm_zero16);
    auto issue = [&](int which, int kt) {                // which: HA0 / HA1 / HW0 / HW1 of k-tile kt (clamped: a dummy re-load past the end)
        if (DIAG_ON(p, 1)) return;                       // diagnostics build: no LDS-DMA at all (tools/gemm_loop_bound.py)
        kt = kt < nk ? kt : nk - 1;
        bf16_t* dst = smem8 + ((kt & 1) * 4 + which) * SLOT + wave * 512;
        const bool isw = which >= HW0;
        const int hf = which & 1;
        const char* g = (isw ? baseW : baseA) + (size_t)kt * (BK * 2);
        const bool z = NX && kt >= nkr;                  // the zero k-tile (wave-uniform): scalar base = the zero line, lane offsets masked to 0
        if (NX) g = z ? zsrc : g;
        const uint32_t om = z ? 0u : 0xffffffffu;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t off = isw ? ow[hf][j] : oa[hf][j];
            if (NX) off &= om;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off),
                                             (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0, 0);
        }
    };

    f32x4_t acc[2][2][2][4];                             // [mh][nh][ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][b][c][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment offsets inside a slot (bf16 elements): A rows wr*64 + mi*16 + lrow, W rows wc*32 + ni*16 + lrow; chunk 4*s + lk
    int offA[4][2], offW[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { const int r = wr * 64 + mi * 16 + lrow; offA[mi][s2] = r * BK + swz(r, 4 * s2 + lk) * 8; }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) { const int r = wc * 32 + ni * 16 + lrow; offW[ni][s2] = r * BK + swz(r, 4 * s2 + lk) * 8; }
    }

    // prologue: all of tile 0 + the two early half-tiles of tile 1; the last two stay in flight
    issue(HA0, 0); issue(HW0, 0); issue(HW1, 0); issue(HA1, 0); issue(HA0, 1); issue(HW1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // Fragment reads per phase are balanced against the 16-MFMA cluster of the other wave group (8 ds_read_b128 of a group's
    // four waves = 256 LDS clocks = one cluster):  p0: W0 + the k-step-1 half of A0 (8)   p1: W1 (4)   p2: A1 (8)
    // p3: W0 again + the k-step-0 half of the NEXT tile's A0 (8; second A register set, that half-tile landed a k-tile ago).
    bf16x8_t af[2][4][2], wf[2][2];                      // A sub-tile sets [tile parity][mi][s]; W sub-tile [ni][s]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) af[0][mi][0] = *reinterpret_cast<const bf16x8_t*>(smem8 + HA0 * SLOT + offA[mi][0]);
    if (wr == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run one barrier behind
#ifdef STG_GEMM_DIAG
    if (p.dbg == 4 && tid == 0 && blockIdx.x < 8192) {
        stg_diag_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
        stg_diag_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    for (int kt2 = 0; kt2 < nk; kt2 += 2) {
#pragma unroll
        for (int cur = 0; cur < 2; ++cur) {
            const int kt = kt2 + cur;
            const bf16_t* sl = smem8 + cur * 4 * SLOT;           // nk is even: parity of kt == cur
            const bf16_t* sn = smem8 + (cur ^ 1) * 4 * SLOT;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int mh = ph >> 1, nh = (ph == 1 || ph == 2) ? 1 : 0;
                if (ph != 2) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
                            wf[ni][s2] = *reinterpret_cast<const bf16x8_t*>(sl + (HW0 + nh) * SLOT + offW[ni][s2]);
                }
                if (ph == 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) af[cur][mi][1] = *reinterpret_cast<const bf16x8_t*>(sl + HA0 * SLOT + offA[mi][1]);
                } else if (ph == 2) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) af[cur][mi][s2] = *reinterpret_cast<const bf16x8_t*>(sl + HA1 * SLOT + offA[mi][s2]);
                } else if (ph == 3) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) af[cur ^ 1][mi][0] = *reinterpret_cast<const bf16x8_t*>(sn + HA0 * SLOT + offA[mi][0]);
                }
                if (ph == 0) issue(HA1, kt + 1);
                if (ph == 1) issue(HW0, kt + 1);
                // round 4 FIX: phase 3 reads the next tile's A0 (k-step 0) BEFORE its own counted wait, and the rows the other wave group
                // staged are only covered one barrier after that group's wait -- the read relied on "issued a k-tile ago", which a slow DMA
                // breaks (1 launch in ~4 000 at 31 360 x 1024 x 4096: one k-tile of rows 32-63 of the partial panel stale).  A0(kt + 1) is the
                // ninth-youngest DMA instruction here (behind W1(kt+1), A1(kt+1), W0(kt+1), A0(kt+2)): vmcnt(8) retires it one phase early, and
                // this phase's two barriers carry it to both groups' phase-3 reads.
                if (ph == 2) { issue(HA0, kt + 2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                if (ph == 3) { issue(HW1, kt + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_setprio(1);
                if (!DIAG_ON(p, 2) && (!NX || nh == 0 || w1_live))
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi)
                            acc[mh][nh][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][s2], af[cur][mi][s2], acc[mh][nh][ni][mi], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
#ifdef STG_GEMM_DIAG
    if (p.dbg == 4 && tid == 0 && blockIdx.x < 8192) {
        stg_diag_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        stg_diag_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (wr == 0) __builtin_amdgcn_s_barrier();           // pairs with the late group's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dummy tail DMAs land before the staging regions reuse the slots
    __builtin_amdgcn_s_barrier();

    // epilogue: per row half one 64 x 64 piece whose right 32 columns sit 96 further right, through the wave's private 8 KiB region
    float* stg = reinterpret_cast<float*>(smem8) + wave * 2048;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        AccTile t;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) t.v[nh * 2 + ni][mi] = acc[mh][nh][ni][mi];
        if (mh) lds_wave_sync();
        gemm_epilogue_dispatch(p, t, m0 + mh * 128 + wr * 64, n0 + wc * 32, 0, 0, lane, stg, 96);
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-tile form of the 8-phase kernel (round 3): one workgroup computes `ntl` CONSECUTIVE COLUMN TILES of one 256-row panel and the
// half-tile pipeline keeps running across the tile boundary.  In the one-tile kernel a K = 512 tile costs 12.9 us of main loop plus
// ~4.8 us in which the matrix pipe has nothing to do before the epilogue even starts (workgroup turnover + the first DMAs' trip to L2 /
// HBM + the drain of the dummy tail loads: 271 us for 15.3 tiles per CU with the epilogue compiled out) -- here the schedule's own
// look-ahead (A1, W0 of k-tile kt + 1 and A0, W1 of kt + 2) simply addresses the NEXT tile's first k-tiles once kt runs past K, so a
// tile's epilogue runs while the next tile's six prologue half-tiles land, and the next main loop starts on operands that are already
// in LDS.  Same MFMA / k order: bit-identical to the one-tile kernel.
//   * LDS: the ring's 8 slots (128 KiB) + 32 KiB.  After the last k-tile the slots parity-1 A1 and W0 (slots 5, 6) are the last ones read
//     and the only ones the look-ahead has not re-staged: they and the extra 32 KiB are the eight 8 KiB epilogue staging regions.
//   * the epilogue variant is a template parameter (one variant per instantiation: the register allocator sees one epilogue, and the
//     main loop's per-lane offsets stay live across it without spills);
//   * before the next main loop: s_waitcnt vmcnt(NST) with NST = the global stores the epilogue issued behind the DMAs (in-order
//     completion: everything older than the NST youngest operations has landed), then the workgroup barrier.
// NX: see gemm_nt_8ph_kernel.  Only the LAST column tile of C can be partial: it takes a second set of clamped W-row offsets.
template <int V, bool NX>
__global__ void __launch_bounds__(512, 1) gemm_nt_8phm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem8[];            // 8 slots x 16 KiB + 32 KiB of epilogue staging
    const int ntl = p.ntl;
    const int gpr = p.nbn / ntl;                         // tile groups per row panel
    const int ngrp = p.nbm * gpr;
    int gid = blockIdx.x;
    {
        const int q = ngrp >> 3, r = ngrp & 7;
        const int xcd = gid & 7, idx = gid >> 3;
        gid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = gid / gpr, bn0 = (gid % gpr) * ntl;
    const int64_t m0 = (int64_t)bm * GBM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lrow = lane & 15, lk = lane >> 4;
    constexpr int SLOT = 128 * BK;
    enum { HA0 = 0, HA1 = 1, HW0 = 2, HW1 = 3 };

    uint32_t oa[2][2], ow[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = j * 512 + tid;
            const int row = hf * 128 + (q >> 3), c = (q & 7) ^ ((q >> 3) & 7);
            const int64_t rm = p.M - 1 - m0;
            const int ra = row < rm ? row : (int)rm;
            oa[hf][j] = (uint32_t)(((int64_t)ra * p.lda + c * 8) * 2);
            ow[hf][j] = (uint32_t)(((int64_t)row * p.ldw + c * 8) * 2);
        }
    const char* baseA = reinterpret_cast<const char*>(p.A + m0 * p.lda);
    const int nkr = p.K / BK;
    const int nk = NX ? (nkr + 1) & ~1 : nkr;
    const char* zsrc = reinterpret_cast<const char*>(gemm_zero16);

    int offA[4][2], offW[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { const int r = wr * 64 + mi * 16 + lrow; offA[mi][s2] = r * BK + swz(r, 4 * s2 + lk) * 8; }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) { const int r = wc * 32 + ni * 16 + lrow; offW[ni][s2] = r * BK + swz(r, 4 * s2 + lk) * 8; }
    }
    // staging regions: waves 0-3 in slots 5 and 6 (contiguous 32 KiB), waves 4-7 behind the ring
    float* stg = reinterpret_cast<float*>(smem8 + (wave < 4 ? 5 * SLOT : 8 * SLOT)) + (wave & 3) * 2048;

#pragma unroll 1
    for (int t = 0; t < ntl; ++t) {
        const int n0 = (bn0 + t) * GBN;
        // the row base is laundered per tile: with a loop-invariant m0 hipcc hoists the epilogue's row addresses out of the tile loop
        // and keeps them in VGPRs across the main loop (32 .. 70 spilled registers; 5 .. 7 this way)
        int64_t mt = m0;
        asm volatile("" : "+s"(mt));
        const char* baseW = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
        const bool has_next = t + 1 < ntl;
        const char* baseWn = reinterpret_cast<const char*>(p.W + (int64_t)(n0 + (has_next ? GBN : 0)) * p.ldw);
        // NX: valid 64-row groups of this tile's / the next tile's W rows (N % 64 == 0; 4 = a full tile).  DMA piece (hf, j) covers W rows
        // 64 (2 hf + j) .. + 63: a group beyond N re-reads rows 0-63 of the tile (always valid; its products are never stored)
        const int nvg_cur = NX ? ((p.N - n0) >> 6 < 4 ? (p.N - n0) >> 6 : 4) : 4;
        const int nvg_next = NX ? ((p.N - n0 - GBN) >> 6 < 4 ? (p.N - n0 - GBN) >> 6 : 4) : 4;
        const bool w1_live = nvg_cur > 2;
        // which: HA0 / HA1 / HW0 / HW1 of k-tile kt; past the end of K: the next tile's k-tile kt - nk (same A panel, next W tile), or --
        // behind the group's last tile -- a dummy re-load of the last k-tile (keeps the counted waits exact)
        auto issue = [&](int which, int kt) {
            const bool isw = which >= HW0;
            const int hf = which & 1;
            const int par = kt & 1;                      // nk is even: the slot parity continues across the tile boundary
            const char* g = isw ? baseW : baseA;
            int nvg = nvg_cur;
            if (kt >= nk) {
                if (has_next) { kt -= nk; g = isw ? baseWn : baseA; nvg = nvg_next; } else kt = nk - 1;
            }
            bf16_t* dst = smem8 + (par * 4 + which) * SLOT + wave * 512;
            g += (size_t)kt * (BK * 2);
            const bool z = NX && kt >= nkr;              // the zero k-tile (wave-uniform): scalar base = the zero line, lane offsets masked to 0
            if (NX) g = z ? zsrc : g;
            const uint32_t om = z ? 0u : 0xffffffffu;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t off = isw ? ((NX && 2 * hf + j >= nvg) ? ow[0][0] : ow[hf][j]) : oa[hf][j];
                if (NX) off &= om;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0, 0);
            }
        };
        f32x4_t acc[2][2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int d = 0; d < 4; ++d) acc[a][b][c][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (t == 0) {                                    // the group's first tile: explicit prologue, the last two half-tiles stay in flight
            issue(HA0, 0); issue(HW0, 0); issue(HW1, 0); issue(HA1, 0); issue(HA0, 1); issue(HW1, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        bf16x8_t af[2][4][2], wf[2][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[0][mi][0] = *reinterpret_cast<const bf16x8_t*>(smem8 + HA0 * SLOT + offA[mi][0]);
        if (wr == 1) __builtin_amdgcn_s_barrier();       // waves 4-7 run one barrier behind

        for (int kt2 = 0; kt2 < nk; kt2 += 2) {
#pragma unroll
            for (int cur = 0; cur < 2; ++cur) {
                const int kt = kt2 + cur;
                const bf16_t* sl = smem8 + cur * 4 * SLOT;
                const bf16_t* sn = smem8 + (cur ^ 1) * 4 * SLOT;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const int mh = ph >> 1, nh = (ph == 1 || ph == 2) ? 1 : 0;
                    if (ph != 2) {
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2)
                                wf[ni][s2] = *reinterpret_cast<const bf16x8_t*>(sl + (HW0 + nh) * SLOT + offW[ni][s2]);
                    }
                    if (ph == 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) af[cur][mi][1] = *reinterpret_cast<const bf16x8_t*>(sl + HA0 * SLOT + offA[mi][1]);
                    } else if (ph == 2) {
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) af[cur][mi][s2] = *reinterpret_cast<const bf16x8_t*>(sl + HA1 * SLOT + offA[mi][s2]);
                    } else if (ph == 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        // the next k-tile's A0, k-step 0 -- past the tile's last k-tile these registers are re-read at the top of the next tile
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) af[cur ^ 1][mi][0] = *reinterpret_cast<const bf16x8_t*>(sn + HA0 * SLOT + offA[mi][0]);
                    }
                    if (ph == 0) issue(HA1, kt + 1);
                    if (ph == 1) issue(HW0, kt + 1);
                    if (ph == 2) { issue(HA0, kt + 2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }      // see gemm_nt_8ph_kernel (round-4 fix)
                    if (ph == 3) { issue(HW1, kt + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_setprio(1);
                    if (!NX || nh == 0 || w1_live)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int mi = 0; mi < 4; ++mi)
                                acc[mh][nh][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][s2], af[cur][mi][s2], acc[mh][nh][ni][mi], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();       // pairs with the late group's last barrier: every wave is past its last ring read
        __builtin_amdgcn_s_barrier();

        // epilogue of tile t through the wave's 8 KiB staging region (slots 5 / 6 and the extra 32 KiB: nothing in flight writes them)
        const bool full = mt + GBM <= p.M && (!NX || n0 + GBN <= p.N);      // wave-uniform (NX = false: N % 256 == 0, only the last row panel can be partial)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            AccTile tl;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) tl.v[nh * 2 + ni][mi] = acc[mh][nh][ni][mi];
            if (mh) lds_wave_sync();
            if (full) gemm_epilogue_rows<V, true>(p, tl, mt + mh * 128 + wr * 64, n0 + wc * 32, 0, 0, lane, stg, 96);
            else gemm_epilogue_rows<V, false>(p, tl, mt + mh * 128 + wr * 64, n0 + wc * 32, 0, 0, lane, stg, 96);
        }
        if (has_next) {
            // the next tile's four k-tile-0 half-tiles (and its first two of k-tile 1) were issued before this epilogue's stores:
            // all but the wave's youngest NST operations done = every DMA landed (vmcnt counts in issue order)
            // every variant stores >= 16 times per wave and tile (8 steps x 2 calls); the byte-derivative variants always 32
            constexpr int NST = (V == EV_GELU8 || V == EV_QGELU8) ? 32 : 16;
            if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dummy tail DMAs must land before the workgroup's LDS is handed on
}

// ------------------------------------------------------------------------------------------------
// fp8 variant (ab_dtype == STG_FP8_MX): A and W are OCP e4m3 bytes with one E8M0 scale per 32-wide k-block of a row (stg_quant_fp8_mx).
// Same 128 x 128 block tile, 2 x 2 waves of 64 x 64 and the same LDS geometry as gemm_nt_glds_kernel -- a k-tile is 128 BYTES per row,
// now 128 k-elements -- but ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 output tile and k-tile instead of two bf16 16x16x32:
// per MFMA clock twice the math of the bf16 form at the same LDS and DMA bytes per k-tile.  Operand map (probed on the device,
// tools/probe/mx_probe.hip): lane 16 g + i carries row i, bytes k = 16 g .. +15 and 64 + 16 g .. +15 -- the 16-byte chunks g and
// 4 + g of the row's 128-byte k-tile, i.e. the very chunks the bf16 kernel reads for its two k-steps -- and the scale of k-block g.
// Fragments are 8 VGPRs each (64 for the wave tile), so the kernel runs 2 workgroups per CU with double-buffered LDS.
typedef int i32x8_t __attribute__((ext_vector_type(8)));

template <int OW, int OA>
__device__ __forceinline__ f32x4_t mfma_mx(const i32x8_t& w, const i32x8_t& a, const f32x4_t c, int sw, int sa) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, c, 0, 0, OW, sw, OA, sa);     // cbsz = blgp = 0: e4m3 x e4m3
}

struct Fp8Scales { const uint32_t* SA; const uint32_t* SW; int KB; int nrbA; int nrbW; };

__global__ void __launch_bounds__(256, 2) gemm_nt_fp8_kernel(GemmParams p, Fp8Scales e) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * (BM + BN) * 128];
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.nbn, bn = bid % p.nbn;
    const int64_t m0 = (int64_t)bm * BM;
    const int n0 = bn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, lk = lane >> 4;
    const uint8_t* A8 = reinterpret_cast<const uint8_t*>(p.A);
    const uint8_t* W8 = reinterpret_cast<const uint8_t*>(p.W);

    const uint8_t* pa[4];
    const uint8_t* pw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave * 4 + j) * 64 + lane;
        const int row = q >> 3, c = (q & 7) ^ (row & 7);
        int64_t gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        pa[j] = A8 + gm * p.lda + c * 16;
        pw[j] = W8 + (int64_t)gn * p.ldw + c * 16;
    }
    auto stage = [&](int buf, int kt) {
        uint8_t* sA = smem + buf * (BM + BN) * 128;
        uint8_t* sW = sA + BM * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[j] + kt * 128),
                                             (__attribute__((address_space(3))) void*)(sA + (wave * 4 + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pw[j] + kt * 128),
                                             (__attribute__((address_space(3))) void*)(sW + (wave * 4 + j) * 1024), 16, 0, 0);
        }
    };
    // scale dwords of this lane: (64-row group, k-block 4 kt + lk, row lrow of each 16-row tile)
    int64_t rbA = m0 / 64 + wm;
    rbA = rbA < e.nrbA ? rbA : e.nrbA - 1;
    int rbW = n0 / 64 + wn;
    rbW = rbW < e.nrbW ? rbW : e.nrbW - 1;
    const uint32_t* psa = e.SA + (rbA * e.KB + lk) * 16 + lrow;
    const uint32_t* psw = e.SW + ((int64_t)rbW * e.KB + lk) * 16 + lrow;

    AccTile accs;
    auto& acc = accs.v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = e.KB / 4;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const int sa = (int)psa[kt * 64], sw = (int)psw[kt * 64];
        const uint8_t* sA = smem + cur * (BM + BN) * 128;
        const uint8_t* sW = sA + BM * 128;
        i32x8_t af[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ar = wm * 64 + i * 16 + lrow;
            const int wr = wn * 64 + i * 16 + lrow;
            const uint4 alo = *reinterpret_cast<const uint4*>(sA + ar * 128 + swz(ar, lk) * 16);
            const uint4 ahi = *reinterpret_cast<const uint4*>(sA + ar * 128 + swz(ar, 4 + lk) * 16);
            const uint4 wlo = *reinterpret_cast<const uint4*>(sW + wr * 128 + swz(wr, lk) * 16);
            const uint4 whi = *reinterpret_cast<const uint4*>(sW + wr * 128 + swz(wr, 4 + lk) * 16);
            af[i] = (i32x8_t){(int)alo.x, (int)alo.y, (int)alo.z, (int)alo.w, (int)ahi.x, (int)ahi.y, (int)ahi.z, (int)ahi.w};
            wf[i] = (i32x8_t){(int)wlo.x, (int)wlo.y, (int)wlo.z, (int)wlo.w, (int)whi.x, (int)whi.y, (int)whi.z, (int)whi.w};
        }
#define STG_MX_ROW(NI)                                                      \
        acc[NI][0] = mfma_mx<NI, 0>(wf[NI], af[0], acc[NI][0], sw, sa);     \
        acc[NI][1] = mfma_mx<NI, 1>(wf[NI], af[1], acc[NI][1], sw, sa);     \
        acc[NI][2] = mfma_mx<NI, 2>(wf[NI], af[2], acc[NI][2], sw, sa);     \
        acc[NI][3] = mfma_mx<NI, 3>(wf[NI], af[3], acc[NI][3], sw, sa);
        STG_MX_ROW(0) STG_MX_ROW(1) STG_MX_ROW(2) STG_MX_ROW(3)
#undef STG_MX_ROW
        __syncthreads();      // drains vmcnt (tile kt + 1 has landed) and every wave is done reading tile kt
    }
    gemm_epilogue_dispatch(p, accs, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem) + (wm * 2 + wn) * 2048);
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[N1,N2] += sum_m dY[m,N1] X[m,N2]   (reduction over the huge token dimension, tiny output)
// MFMA k index = token row, so both operands are needed k-major per lane; tiles are staged row-major in LDS
// (coalesced 16-byte global loads) and fragments are gathered with 16-bit LDS reads (the op is HBM-bound:
// ~9 KB of HBM per 8 MFMAs).  Output tile 64 (n1) x 128 (n2), split over M across blocks, fp32 atomics.
constexpr int T1 = 64, T2 = 128, TK = 64;

struct WgradParams {
    const bf16_t* dY; int64_t lddy;
    const bf16_t* X; int64_t ldx;
    float* dW; int64_t lddw;
    float* db;
    int64_t M; int N1, N2;
    int nt1, nt2, msplit;
    int64_t rows_per_split;
    int vec_y, vec_x;
    const float* row_scale; int64_t rs_outer, rs_inner;
};

__global__ void __launch_bounds__(256, 2) wgrad_tn_generic_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t sY[TK * T1];
    __shared__ __attribute__((aligned(16))) bf16_t sX[TK * T2];
    const int tile = blockIdx.x;
    const int t1 = tile / p.nt2, t2 = tile % p.nt2;
    const int n1_0 = t1 * T1, n2_0 = t2 * T2;
    const int64_t mbeg = (int64_t)blockIdx.y * p.rows_per_split;
    int64_t mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;

    f32x4_t acc[4][2];
    f32x4_t accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = (p.db != nullptr) && (t2 == 0) && (wave == 0);
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;

    for (int64_t mb = mbeg; mb < mend; mb += TK) {
        // stage dY[mb..mb+64, n1_0..+64] and X[mb..mb+64, n2_0..+128]
        for (int id = tid; id < TK * T1 / 8; id += 256) {
            const int row = id / (T1 / 8), c = id % (T1 / 8);
            const int64_t gm = mb + row;
            const int gn = n1_0 + c * 8;
            u16x8 v;
            if (gm < mend && p.vec_y && gn + 7 < p.N1) {
                v = *reinterpret_cast<const u16x8*>(p.dY + gm * p.lddy + gn);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v.v[j] = (gm < mend && gn + j < p.N1) ? p.dY[gm * p.lddy + gn + j] : 0;
            }
            if (p.row_scale && gm < mend) {
                const float rs = p.row_scale[(gm / p.rs_outer) * p.rs_inner + (gm % p.rs_inner)];
#pragma unroll
                for (int j = 0; j < 8; ++j) v.v[j] = f2bf(bf2f(v.v[j]) * rs);
            }
            *reinterpret_cast<u16x8*>(sY + row * T1 + c * 8) = v;
        }
        for (int id = tid; id < TK * T2 / 8; id += 256) {
            const int row = id / (T2 / 8), c = id % (T2 / 8);
            const int64_t gm = mb + row;
            const int gn = n2_0 + c * 8;
            u16x8 v;
            if (gm < mend && p.vec_x && gn + 7 < p.N2) {
                v = *reinterpret_cast<const u16x8*>(p.X + gm * p.ldx + gn);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v.v[j] = (gm < mend && gn + j < p.N2) ? p.X[gm * p.ldx + gn + j] : 0;
            }
            *reinterpret_cast<u16x8*>(sX + row * T2 + c * 8) = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t pf[4], qf[2];
            const int kr = 32 * s + 8 * lg;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[i][j] = (short)sY[(kr + j) * T1 + i * 16 + li];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[q][j] = (short)sX[(kr + j) * T2 + (wave * 2 + q) * 16 + li];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[q], acc[i][q], 0, 0, 0);
                if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], ones, accb[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[row = n1 (4*lg + r)][col = n2 (li)]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n1 = n1_0 + i * 16 + 4 * lg + r;
            if (n1 >= p.N1) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int n2 = n2_0 + (wave * 2 + q) * 16 + li;
                if (n2 < p.N2) atomicAdd(p.dW + (int64_t)n1 * p.lddw + n2, acc[i][q][r]);
            }
            if (do_bias && li == 0) atomicAdd(p.db + n1, accb[i][r]);
        }
    }
}


// Fast path (both leading dims multiples of 8, 16-byte aligned): A = the NARROW operand (<= 64 columns per block: adapter
// hidden width / head classes), B = the wide one; unconditional clamped loads (rows beyond the split are zeroed by a select),
// only the NT1 valid 16-column tiles of A are ever read / multiplied, output optionally transposed so that the narrow
// operand can be either dY (D_fc1: dW[d_h, C]) or X (D_fc2: dW[C, d_h] = (X^T dY)^T).
struct WgradFast {
    const bf16_t* A; int64_t lda; int NA;      // narrow
    const bf16_t* B; int64_t ldb; int NB;      // wide
    float* dW; int64_t lddw; int transpose_out;
    float* db; int bias_on;                    // 1: column sums of A, 2: column sums of B
    const float* row_scale; int64_t rs_outer, rs_inner; int scale_on;   // 1: scale A rows, 2: scale B rows
    int64_t M; int64_t rows_per_split;
    int nta;                                   // column tiles of A (each 16*NT1 wide)
};

template <int NT1>
__global__ void __launch_bounds__(256, 2) wgrad_tn_fast_kernel(WgradFast p) {
    constexpr int TA = 16 * NT1;
    __shared__ __attribute__((aligned(16))) bf16_t sA[TK * TA];
    __shared__ __attribute__((aligned(16))) bf16_t sB[TK * T2];
    const int ta = blockIdx.x % p.nta, tb = blockIdx.x / p.nta;
    const int a0 = ta * TA, b0 = tb * T2;
    const int64_t mbeg = (int64_t)blockIdx.y * p.rows_per_split;
    int64_t mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;

    f32x4_t acc[NT1][2];
    f32x4_t accba[NT1], accbb[2];
#pragma unroll
    for (int i = 0; i < NT1; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NT1; ++i) accba[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    accbb[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    accbb[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool bias_a = p.db != nullptr && p.bias_on == 1 && tb == 0 && wave == 0;
    const bool bias_b = p.db != nullptr && p.bias_on == 2 && ta == 0;
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;

    constexpr int CA = TA / 8;                       // 16-byte chunks per A row
    constexpr int ITA = (TK * CA + 255) / 256;
    constexpr int ITB = TK * (T2 / 8) / 256;         // 4
    const int na8 = (p.NA + 7) & ~7, nb8 = (p.NB + 7) & ~7;

    uint4 va[ITA], vb[ITB];
    auto gload = [&](int64_t mb) {                   // unconditional clamped loads: issued early, consumed one chunk later
#pragma unroll
        for (int i = 0; i < ITA; ++i) {
            const int id = tid + i * 256;
            const int row = (id / CA) % TK, c = id % CA;
            int64_t gm = mb + row;
            gm = gm < mend ? gm : mend - 1;
            int gn = a0 + c * 8;
            gn = gn < na8 ? gn : 0;
            va[i] = *reinterpret_cast<const uint4*>(p.A + gm * p.lda + gn);
        }
#pragma unroll
        for (int i = 0; i < ITB; ++i) {
            const int id = tid + i * 256;
            const int row = id / (T2 / 8), c = id % (T2 / 8);
            int64_t gm = mb + row;
            gm = gm < mend ? gm : mend - 1;
            int gn = b0 + c * 8;
            gn = gn < nb8 ? gn : 0;
            vb[i] = *reinterpret_cast<const uint4*>(p.B + gm * p.ldb + gn);
        }
    };
    auto scale8 = [&](uint4& v, int64_t gm) {
        const float rs = p.row_scale[(gm / p.rs_outer) * p.rs_inner + (gm % p.rs_inner)];
        v.x = pack_bf2(__uint_as_float(v.x << 16) * rs, __uint_as_float(v.x & 0xffff0000u) * rs);
        v.y = pack_bf2(__uint_as_float(v.y << 16) * rs, __uint_as_float(v.y & 0xffff0000u) * rs);
        v.z = pack_bf2(__uint_as_float(v.z << 16) * rs, __uint_as_float(v.z & 0xffff0000u) * rs);
        v.w = pack_bf2(__uint_as_float(v.w << 16) * rs, __uint_as_float(v.w & 0xffff0000u) * rs);
    };
    gload(mbeg);
    for (int64_t mb = mbeg; mb < mend; mb += TK) {
#pragma unroll
        for (int i = 0; i < ITA; ++i) {
            const int id = tid + i * 256;
            if (ITA * 256 > TK * CA && id >= TK * CA) continue;
            const int row = id / CA, c = id % CA;
            const int64_t gm = mb + row;
            if (gm >= mend) va[i] = make_uint4(0, 0, 0, 0);
            else if (p.row_scale && p.scale_on == 1) scale8(va[i], gm);
            *reinterpret_cast<uint4*>(sA + row * TA + c * 8) = va[i];
        }
#pragma unroll
        for (int i = 0; i < ITB; ++i) {
            const int id = tid + i * 256;
            const int row = id / (T2 / 8), c = id % (T2 / 8);
            const int64_t gm = mb + row;
            if (gm >= mend) vb[i] = make_uint4(0, 0, 0, 0);
            else if (p.row_scale && p.scale_on == 2) scale8(vb[i], gm);
            *reinterpret_cast<uint4*>(sB + row * T2 + c * 8) = vb[i];
        }
        if (mb + TK < mend) gload(mb + TK);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t pf[NT1], qf[2];
            const int kr = 32 * s + 8 * lg;
#pragma unroll
            for (int i = 0; i < NT1; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[i][j] = (short)sA[(kr + j) * TA + i * 16 + li];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[q][j] = (short)sB[(kr + j) * T2 + (wave * 2 + q) * 16 + li];
#pragma unroll
            for (int i = 0; i < NT1; ++i) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[q], acc[i][q], 0, 0, 0);
                if (bias_a) accba[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], ones, accba[i], 0, 0, 0);
            }
            if (bias_b) {
#pragma unroll
                for (int q = 0; q < 2; ++q) accbb[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, qf[q], accbb[q], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[row = A column (4*lg + r)][col = B column (li)]
#pragma unroll
    for (int i = 0; i < NT1; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int na = a0 + i * 16 + 4 * lg + r;
            if (na >= p.NA) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int nb = b0 + (wave * 2 + q) * 16 + li;
                if (nb < p.NB) {
                    float* dst = p.transpose_out ? p.dW + (int64_t)nb * p.lddw + na : p.dW + (int64_t)na * p.lddw + nb;
                    atomicAdd(dst, acc[i][q][r]);
                }
            }
            if (bias_a && li == 0) atomicAdd(p.db + na, accba[i][r]);
        }
    }
    if (bias_b && lg == 0) {            // every row of accb[q] holds the column sums; take row 0 (lg == 0, r == 0)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nb = b0 + (wave * 2 + q) * 16 + li;
            if (nb < p.NB) atomicAdd(p.db + nb, accbb[q][0]);
        }
    }
}

std::atomic<uint64_t> lds_8ph_done[2], lds_8phm_done[8];

}  // namespace

int stg_gemm_skinny_down(const void* A, int64_t lda, const void* W, const void* W2, int64_t ldw, const float* bias, const float* bias2, void* Cout, int64_t ldc,
                         void* dact, int64_t ldp, int64_t M, int64_t split_m, int N, int K, void* stream);      // skinny.hip

extern "C" int stg_gemm_nt(stg_gemm_args* a, void* stream) {
    STG_CHECK(a != nullptr, -1, "stg_gemm_nt: null args");
    STG_CHECK(a->A && a->W && a->C, -1, "stg_gemm_nt: null A/W/C");
    STG_CHECK(a->M >= 0 && a->N > 0 && a->K > 0, -2, "stg_gemm_nt: bad shape M=%lld N=%d K=%d", (long long)a->M, a->N, a->K);
    STG_CHECK(a->K % 8 == 0, -2, "stg_gemm_nt: K=%d must be a multiple of 8", a->K);
    STG_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, -2, "stg_gemm_nt: A/W must be 16-byte aligned");
    const bool fp8 = a->ab_dtype == STG_FP8_MX;
    STG_CHECK(fp8 || a->ab_dtype == 0 || a->ab_dtype == STG_BF16, -3, "stg_gemm_nt: unsupported ab_dtype %d", a->ab_dtype);
    if (fp8) {
        const int64_t kp = ((int64_t)a->K + 127) / 128 * 128;
        STG_CHECK(a->a_scale && a->w_scale && (((uintptr_t)a->a_scale | (uintptr_t)a->w_scale) & 3) == 0, -1, "stg_gemm_nt: fp8 operands need their scale tables");
        STG_CHECK(a->lda % 16 == 0 && a->ldw % 16 == 0 && a->lda >= kp && a->ldw >= kp && a->ldc >= a->N, -2,
                  "stg_gemm_nt: fp8 leading dimensions are bytes, multiples of 16 and >= K rounded up to 128");
        STG_CHECK(a->conv_H <= 0 && a->batch <= 1, -3, "stg_gemm_nt: fp8 operands do not combine with the implicit convolution / batched mode");
    } else {
        STG_CHECK(a->lda % 8 == 0 && a->ldw % 8 == 0, -2, "stg_gemm_nt: lda/ldw must be multiples of 8");
        STG_CHECK((a->conv_H > 0 || a->lda >= a->K) && a->ldw >= a->K && a->ldc >= a->N, -2, "stg_gemm_nt: leading dimension too small");
    }
    STG_CHECK(a->c_dtype == STG_BF16 || a->c_dtype == STG_F32, -3, "stg_gemm_nt: unsupported c_dtype %d", a->c_dtype);
    STG_CHECK(a->act >= 0 && a->act <= 2, -3, "stg_gemm_nt: bad act");
    STG_CHECK(!a->dact || a->act != 0, -3, "stg_gemm_nt: dact output needs an activation");
    if (a->row_scale) STG_CHECK(a->rs_outer > 0 && a->rs_inner > 0, -2, "stg_gemm_nt: bad row_scale params");
    if (a->res1) STG_CHECK(a->res1_dtype == STG_BF16 || a->res1_dtype == STG_F32, -3, "stg_gemm_nt: bad res1 dtype");
    if (a->res2) STG_CHECK(a->res2_dtype == STG_BF16 || a->res2_dtype == STG_F32, -3, "stg_gemm_nt: bad res2 dtype");
    if (a->M == 0) return 0;
    GemmParams p;
    p.A = (const bf16_t*)a->A; p.lda = a->lda;
    p.W = (const bf16_t*)a->W; p.ldw = a->ldw;
    p.C = a->C; p.ldc = a->ldc; p.c_f32 = (a->c_dtype == STG_F32);
    p.bias = a->bias; p.alpha = a->alpha; p.act = a->act;
    p.dact = (bf16_t*)a->dact; p.ldp = a->ldp;
    p.dact_src = (const bf16_t*)a->dact_src; p.ldd = a->ldd;
    p.row_scale = a->row_scale; p.rs_outer = a->rs_outer; p.rs_inner = a->rs_inner;
    p.res1 = a->res1; p.ldr1 = a->ldr1; p.res1_f32 = (a->res1_dtype == STG_F32);
    p.res2 = a->res2; p.ldr2 = a->ldr2; p.res2_f32 = (a->res2_dtype == STG_F32);
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.conv_H = a->conv_H; p.conv_W = a->conv_W; p.conv_d = a->conv_d; p.conv_C = a->conv_C; p.conv_zero = (const bf16_t*)a->conv_zero;
    const bool conv = a->conv_H > 0;
    p.batch = a->batch > 1 ? a->batch : 1; p.a_bstride = a->a_bstride; p.w_bstride = a->w_bstride; p.c_bstride = a->c_bstride;
    p.split_m = a->split_m > 0 ? a->split_m : 0; p.W2 = (const bf16_t*)a->W2; p.bias2 = a->bias2;
    const bool split = p.split_m > 0;
    if (split) {
        STG_CHECK(a->W2 && (a->bias == nullptr) == (a->bias2 == nullptr) && a->split_m % BM == 0 && a->split_m < a->M, -2,
                  "stg_gemm_nt: split mode needs W2, bias2 like bias, split_m %% 128 == 0 and split_m < M");
        STG_CHECK(a->conv_H <= 0 && a->batch <= 1 && a->ab_dtype != STG_FP8_MX && ((uintptr_t)a->W2 & 15) == 0 && ((uintptr_t)a->bias2 & 15) == 0, -3,
                  "stg_gemm_nt: split mode takes bf16 operands, no convolution / batch, 16-byte aligned W2 / bias2");
        STG_CHECK(a->K % BK == 0 || (a->conv_zero && ((uintptr_t)a->conv_zero & 15) == 0), -2, "stg_gemm_nt: split mode with K %% 64 != 0 needs the zero line (conv_zero)");
    }
    if (p.batch > 1) {
        STG_CHECK(!conv && !a->dact && !a->dact_src && !a->res1 && !a->res2 && !a->row_scale, -3, "stg_gemm_nt: batched mode takes bias / alpha / activation only");
        STG_CHECK(a->K % BK == 0 && a->a_bstride % 8 == 0 && a->w_bstride % 8 == 0 && a->c_bstride % 8 == 0 && p.batch <= 65535, -2,
                  "stg_gemm_nt: batched mode needs K %% 64 == 0 and strides that keep 16-byte alignment");
    }
    if (conv) {
        STG_CHECK(a->conv_W > 0 && a->conv_d >= 1 && a->conv_C > 0 && (a->conv_C % BK == 0 || a->conv_C == 8 || a->conv_C == 16 || a->conv_C == 32) &&
                  a->K == 9 * a->conv_C, -2, "stg_gemm_nt: implicit convolution needs conv_C %% 64 == 0 (or 8 / 16 / 32) and K == 9 * conv_C");
        STG_CHECK(a->lda >= a->conv_C && a->M % ((int64_t)a->conv_H * a->conv_W) == 0, -2, "stg_gemm_nt: implicit convolution: bad lda / M");
        STG_CHECK(a->conv_zero && ((uintptr_t)a->conv_zero & 15) == 0, -2, "stg_gemm_nt: implicit convolution needs a 16-byte aligned zero line");
    }
#ifdef STG_GEMM_DIAG
    p.dbg = stg_opt_gemm_dbg.load(std::memory_order_relaxed);
#endif
    const int64_t nbm = (a->M + BM - 1) / BM;
    const int64_t nbn = (a->N + BN - 1) / BN;
    STG_CHECK(nbm * nbn < (1ll << 31), -2, "stg_gemm_nt: grid too large");
    p.nbm = (int)nbm; p.nbn = (int)nbn;
    auto al = [](const void* ptr, int64_t ld, int bytes_per, int need) {
        return ptr == nullptr || ((((uintptr_t)ptr) % need) == 0 && (ld * bytes_per) % need == 0);
    };
    p.vec_ok = 0;
    const bool d8 = a->dact_dtype == STG_U8_LIN;
    STG_CHECK(d8 || a->dact_dtype == 0 || a->dact_dtype == STG_BF16, -3, "stg_gemm_nt: unsupported dact_dtype %d", a->dact_dtype);
    const bool vec8 = a->N % 8 == 0 && al(a->C, a->ldc, p.c_f32 ? 4 : 2, 16) && al(a->bias, 4, 4, 16) &&
                      al(a->dact, a->ldp, d8 ? 1 : 2, d8 ? 8 : 16) && al(a->dact_src, a->ldd, d8 ? 1 : 2, d8 ? 8 : 16) &&
                      al(a->res1, a->ldr1, p.res1_f32 ? 4 : 2, 16) && al(a->res2, a->ldr2, p.res2_f32 ? 4 : 2, 16);
    p.epi_variant = -1;
    if (vec8) {
        p.epi_variant = EV_GENERIC;
        const bool r1b = a->res1 && !p.res1_f32, r2q = a->res2 && p.res2_f32;
        if (a->alpha == 1.0f && !a->row_scale) {
            if (!p.c_f32 && !a->dact_src && !a->res1 && !a->res2)
                p.epi_variant = a->act == STG_ACT_NONE ? EV_PLAIN : (a->act == STG_ACT_GELU ? EV_GELU : EV_QGELU);
            else if (!p.c_f32 && !a->act && a->dact_src && !a->res1 && !a->res2) p.epi_variant = EV_DSRC;
            else if (!p.c_f32 && !a->act && !a->dact_src && r1b && !a->res2) p.epi_variant = EV_R16;
            else if (p.c_f32 && !a->act && !a->dact_src && r1b && r2q) p.epi_variant = EV_BRQ;
        }
        if (d8 && (a->dact || a->dact_src)) {              // the 8-bit derivative exists in its dedicated variants only
            const int v0 = p.epi_variant;
            const bool out8 = a->dact && !a->dact_src && (a->alpha == 1.0f && !a->row_scale && !p.c_f32 && !a->res1 && !a->res2) && a->act != STG_ACT_NONE;
            const bool in8 = a->dact_src && !a->dact && (a->alpha == 1.0f && !a->row_scale && !p.c_f32 && !a->res1 && !a->res2) && a->act == STG_ACT_NONE;
            (void)v0;
            STG_CHECK(out8 || in8, -3, "stg_gemm_nt: the 8-bit derivative needs a plain GELU / QuickGELU (+ bias) output or a plain derivative-source epilogue");
            p.epi_variant = out8 ? (a->act == STG_ACT_GELU ? EV_GELU8 : EV_QGELU8) : EV_DSRC8;
        }
    }
    STG_CHECK(!(d8 && (a->dact || a->dact_src)) || vec8, -2, "stg_gemm_nt: the 8-bit derivative needs the row-layout epilogue (N %% 8 == 0, aligned operands)");
    if (fp8) {
        STG_CHECK(p.epi_variant >= 0, -2, "stg_gemm_nt: fp8 operands need the row-layout epilogue (N %% 8 == 0, 16-byte aligned outputs)");
        Fp8Scales e;
        e.SA = (const uint32_t*)a->a_scale; e.SW = (const uint32_t*)a->w_scale;
        e.KB = (int)(((int64_t)a->K + 127) / 128 * 4);
        e.nrbA = (int)((a->M + 63) / 64); e.nrbW = (a->N + 63) / 64;
        a->kernel_chosen = STG_GEMM_KERNEL_FP8;
        hipLaunchKernelGGL(gemm_nt_fp8_kernel, dim3((unsigned)(nbm * nbn)), dim3(256), 0, (hipStream_t)stream, p, e);
        STG_LAUNCH_CHECK();
        return 0;
    }
    // the long-K shapes (K >= 1024, whole 256-column tiles): the epilogue is a small share of a tile
    const bool big = !split && !conv && p.batch == 1 && a->K % BK == 0 && a->M >= GBM && a->N >= GBN && a->N % GBN == 0 && a->K >= 1024;
    // round 6: the adapters' down-projection ([rows, C] -> [rows, d_h <= 64] + bias + GELU + bf16 derivative) as a row stream (skinny.hip): bit-identical,
    // 1.6-1.8 x the byte floor on the 128 x 128 kernel -> ~1.2 x.  Option gemm_nx = 0 (the round-5 routing, A/B) keeps it off too.
    if (p.epi_variant == EV_GELU && a->dact && !d8 && !conv && p.batch == 1 && a->M >= 8192 && a->N <= 64 && a->N % 16 == 0 && a->K >= 128 &&
        (int64_t)a->N * a->K <= 32768 && a->ldp % 4 == 0 && a->ldc % 4 == 0 && stg_opt_gemm_nx.load(std::memory_order_relaxed) != 0) {
        const int rc = stg_gemm_skinny_down(a->A, a->lda, a->W, split ? a->W2 : nullptr, a->ldw, a->bias, split ? a->bias2 : nullptr, a->C, a->ldc, a->dact, a->ldp,
                                            a->M, split ? a->split_m : 0, a->N, a->K, stream);
        if (rc <= 0) { if (rc == 0) a->kernel_chosen = STG_GEMM_KERNEL_SKINNY; return rc; }
    }
    const int ph8_mode = stg_opt_gemm_8ph.load(std::memory_order_relaxed);   // 0 off, 1 = in place of the large-tile kernel (default), 2 = every legal shape
    // round 6, the NX forms (template flag of both 8-phase kernels): N % 64 == 0 with N >= 192 and K % 64 == 0 with K >= 192 -- Swin-L's widths
    // (C = 192 / 384: qkv N = 576 / 1152, K = 192 / 576, the N = 192 / 384 projections) kept 45-65 ms of its step on the 128 x 128 kernel
    const int nx_mode = stg_opt_gemm_nx.load(std::memory_order_relaxed);     // 0: the round-5 shape rules (A/B)
    const bool nx = nx_mode != 0 && (a->N % GBN != 0 || a->K % (2 * BK) != 0);
    const bool ph8_ok = !split && !conv && p.batch == 1 && a->M >= GBM && p.epi_variant >= 0 &&
                        (nx ? (a->K % BK == 0 && a->K >= 3 * BK && a->N % 64 == 0 && a->N >= 192) : (a->K % (2 * BK) == 0 && a->N % GBN == 0));
    // mode 1: the long-K shapes (K >= 1024), and K >= 512 with a wide [M, >= 1536] output behind a plain / activation epilogue
    // (measured +4..5 % on 125440 x 1536 x 512, +2..4 % on x 2048 x 512 with GELU + derivative; the derivative-source epilogue
    // of the fc2 dgrad and the N = 512 shapes are faster on the 128 x 128 kernel)
    // (r2, measured inside the step: K = 512 -> N = 512 projections 101 -> 88 us with bias, 96 -> 91 us plain.  The derivative-source
    // epilogue of the fc2 dgrad stays on the 128 x 128 kernel: after round 3's epilogue fix the class reads 334-360 us there and the three
    // routings tried in tools/gemm_route_ab.py are within 1 % of each other, profiles/r03_gemm_route_ab.txt)
    // (r2, stage-1 shapes, microbenchmark: 501 760 x 768 x 256 + bias 437 -> 388 us, x 256 x 768 343 -> 297, x 1024 x 256 with GELU +
    // derivative 814 -> 763, x 256 x 256 156 -> 149: the 128 x 128 kernel's main loop is L2-bandwidth-bound there)
    // round 5 (option gemm_d8m, default on): the fc2 dgrad's byte-derivative-source epilogue joins the wide-output classes when the MULTI-tile
    // walk will take it (its derivative bytes are ordinary global loads in the epilogue: waiting for them also waits for the next tile's older
    // prologue DMAs, which is safe -- the counted wait behind the epilogue only needs >= NST younger operations)
    // (round 5b: option value 2, the default, extends this to K <= 1024 -- Swin-L's and ViT-B's fc2 dgrad, 3072 x 768, ran on the 128 x 128 kernel
    // at 0.81 PFLOP/s, 13 ms of the Swin-L step; on the one-tile 8-phase kernel the step's GEMM time drops 122.1 -> 119.6 ms)
    const int d8opt = stg_opt_gemm_d8m.load(std::memory_order_relaxed);
    const bool d8m = p.epi_variant == EV_DSRC8 && d8opt != 0 && a->K <= (d8opt >= 2 ? 1024 : 512) && a->N >= (nx ? 768 : 1024);
    const bool plainish = p.epi_variant == EV_PLAIN || p.epi_variant == EV_GELU || p.epi_variant == EV_QGELU || p.epi_variant == EV_GELU8 || p.epi_variant == EV_QGELU8 || d8m;
    const bool ph8_wide = a->K >= 256 && a->N >= 256 && a->M >= 8192 && plainish;
    // NX shapes, measured per class of the Swin-L step (profiles/r06_gemm_nx_ab.txt): the K = 192 / 384 classes are store-bound (output 3-4 x the
    // input, 3.0-3.1 TB/s) and run 0-18 % SLOWER here than on the 128 x 128 kernel, whose four workgroups per CU overlap one tile's stores with the
    // others' main loops; the NX forms win where the main loop is long and the output narrow (N = 192, K = 768: fc2 and the fc1 dgrad of stage 0,
    // 967 -> 893 us).  Default (1): those; option value 2: every legal NX shape (tests, A/B)
    const bool ph8_nx = nx && a->M >= 8192 && (nx_mode >= 2 || (plainish && a->N < GBN && a->K >= 768));
    if (ph8_ok && ((ph8_mode == 1 && (big || ph8_wide || ph8_nx)) || ph8_mode == 2 || (ph8_mode == 3 && big))) {      // 3 = long-K shapes only (A/B knob)
        const int64_t gbm = (a->M + GBM - 1) / GBM, gbn = (a->N + GBN - 1) / GBN;
        p.nbm = (int)gbm; p.nbn = (int)gbn;
        // multi-tile form (option gemm_8phm, default on): ntl >= 3 consecutive column tiles per workgroup where the tile is short
        // (K <= 512: turnover + epilogue are a third of it; at K = 768 -- ViT-B, Swin-L stage 2 -- the whole-model A/B is neutral to -0.8 %) and the group count still fills the chip twice.  Measured per class of the
        // step, one process, interleaved (tools/gemm_route_ab.py ... gemm_8phm): qkv 125440 x 1536 x 512 247 -> 217 us (ntl = 3), fc1 with
        // GELU + byte derivative x 2048 x 512 426 -> 396 (ntl = 4), stage-1 qkv 327 -> 312; PAIRS of tiles (N = 512) lose 6 .. 25 %:
        // half as many, twice as long workgroups end on a longer tail, and the next tile's first counted wait also waits for the
        // epilogue's own stores (vmcnt is one in-order counter), which a fresh workgroup does not.
        int ntl = 1;
        int m8 = stg_opt_gemm_8phm.load(std::memory_order_relaxed);
        const bool half_ok = m8 != -1;                     // -1: the round-3 rule alone (A/B knob)
        if (m8 == -1) m8 = 1;
        if (m8 > 0 && gbn >= 3 && (a->K <= 512 || m8 >= 2)) {
            for (int c = (int)(gbn < 8 ? gbn : 8); c >= (m8 >= 2 ? 2 : 3); --c)
                if (gbn % c == 0 && (m8 >= 2 ? c <= m8 : true) && gbm * (gbn / c) >= 2 * 256) { ntl = c; break; }
            // round 4: the default step form runs HALF batches (two micro-batch chains): at 62 720 rows no walk of >= 3 tiles leaves two
            // workgroups per CU, and the classes fell back to the one-tile kernel.  Second pass: the shortest walk that still gives every
            // CU a workgroup and a half (qkv: 3 tiles, 490 workgroups; fc1: 4 tiles, 490)
            if (ntl == 1 && half_ok && m8 == 1) {
                for (int c = 3; c <= (int)(gbn < 8 ? gbn : 8); ++c)
                    if (gbn % c == 0 && gbm * (gbn / c) >= 256 + 128) { ntl = c; break; }
            }
        }
        if (ntl > 1) {
            p.ntl = ntl;
            const int lds = 8 * 128 * BK * 2 + 8 * 4096;
            const unsigned grid = (unsigned)(gbm * (gbn / ntl));
#define STG_8PHM(V, slot) case V: if (nx) { STG_CHECK(stg_reserve_lds(gemm_nt_8phm_kernel<V, true>, lds, lds_8phm_done[4 + slot]), -101, "stg_gemm_nt: cannot reserve 160 KiB of LDS"); \
                                            hipLaunchKernelGGL((gemm_nt_8phm_kernel<V, true>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p); } \
                                  else { STG_CHECK(stg_reserve_lds(gemm_nt_8phm_kernel<V, false>, lds, lds_8phm_done[slot]), -101, "stg_gemm_nt: cannot reserve 160 KiB of LDS"); \
                                         hipLaunchKernelGGL((gemm_nt_8phm_kernel<V, false>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p); } \
                                  launched = true; break;
            bool launched = false;
            switch (p.epi_variant) {
                STG_8PHM(EV_PLAIN, 0) STG_8PHM(EV_GELU8, 1) STG_8PHM(EV_QGELU8, 2) STG_8PHM(EV_DSRC8, 3)
                default: break;
            }
#undef STG_8PHM
            if (launched) {
                a->kernel_chosen = STG_GEMM_KERNEL_8PHM;
                STG_LAUNCH_CHECK();
                return 0;
            }
        }
        a->kernel_chosen = STG_GEMM_KERNEL_8PH;
        if (nx) {
            STG_CHECK(stg_reserve_lds(gemm_nt_8ph_kernel<true>, 8 * 128 * BK * 2, lds_8ph_done[1]), -101, "stg_gemm_nt: cannot reserve 128 KiB of LDS");
            hipLaunchKernelGGL(gemm_nt_8ph_kernel<true>, dim3((unsigned)(gbm * gbn)), dim3(512), 8 * 128 * BK * 2, (hipStream_t)stream, p);
        } else {
            STG_CHECK(stg_reserve_lds(gemm_nt_8ph_kernel<false>, 8 * 128 * BK * 2, lds_8ph_done[0]), -101, "stg_gemm_nt: cannot reserve 128 KiB of LDS");
            hipLaunchKernelGGL(gemm_nt_8ph_kernel<false>, dim3((unsigned)(gbm * gbn)), dim3(512), 8 * 128 * BK * 2, (hipStream_t)stream, p);
        }
        STG_LAUNCH_CHECK();
        return 0;
    }
    if (p.batch > 1) {
        a->kernel_chosen = STG_GEMM_KERNEL_GLDS_BATCH;
        hipLaunchKernelGGL((gemm_nt_glds_kernel<1, false, true>), dim3((unsigned)(nbm * nbn), (unsigned)p.batch), dim3(256), 0, (hipStream_t)stream, p);
    } else if (conv) {
        a->kernel_chosen = STG_GEMM_KERNEL_GLDS_CONV;
        hipLaunchKernelGGL((gemm_nt_glds_kernel<1, true>), dim3((unsigned)(nbm * nbn)), dim3(256), 0, (hipStream_t)stream, p);
    } else if (a->K % BK == 0) {
        a->kernel_chosen = STG_GEMM_KERNEL_GLDS;
        hipLaunchKernelGGL(gemm_nt_glds_kernel<1>, dim3((unsigned)(nbm * nbn)), dim3(256), 0, (hipStream_t)stream, p);
    } else if (a->conv_zero && ((uintptr_t)a->conv_zero & 15) == 0) {  // K % 64 != 0 (K = 96, 48, 16 ...): LDS-DMA kernel with a zero-filled k tail
        a->kernel_chosen = STG_GEMM_KERNEL_GLDS_KTAIL;
        hipLaunchKernelGGL((gemm_nt_glds_kernel<1, false, false, true>), dim3((unsigned)(nbm * nbn)), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        a->kernel_chosen = STG_GEMM_KERNEL_REG;
        hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)(nbm * nbn)), dim3(256), 0, (hipStream_t)stream, p);
    }
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_wgrad_tn(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                            float* db, int64_t M, int N1, int N2, const float* row_scale, int64_t rs_outer,
                            int64_t rs_inner, void* stream) {
    STG_CHECK(dY && X && dW, -1, "stg_wgrad_tn: null pointer");
    STG_CHECK(M >= 0 && N1 > 0 && N2 > 0, -2, "stg_wgrad_tn: bad shape");
    STG_CHECK(lddy >= N1 && ldx >= N2 && lddw >= N2, -2, "stg_wgrad_tn: leading dimension too small");
    if (row_scale) STG_CHECK(rs_outer > 0 && rs_inner > 0, -2, "stg_wgrad_tn: bad row_scale params");
    if (M == 0) return 0;
    const int64_t kchunks = (M + TK - 1) / TK;
    const bool fast = (lddy % 8 == 0) && (ldx % 8 == 0) && (((uintptr_t)dY & 15) == 0) && (((uintptr_t)X & 15) == 0) &&
                      lddy >= ((N1 + 7) & ~7) && ldx >= ((N2 + 7) & ~7);
    if (fast) {
        WgradFast p;
        const bool y_narrow = N1 <= N2;
        p.A = (const bf16_t*)(y_narrow ? dY : X); p.lda = y_narrow ? lddy : ldx; p.NA = y_narrow ? N1 : N2;
        p.B = (const bf16_t*)(y_narrow ? X : dY); p.ldb = y_narrow ? ldx : lddy; p.NB = y_narrow ? N2 : N1;
        p.dW = dW; p.lddw = lddw; p.transpose_out = y_narrow ? 0 : 1;
        p.db = db; p.bias_on = y_narrow ? 1 : 2;
        p.row_scale = row_scale; p.rs_outer = row_scale ? rs_outer : 1; p.rs_inner = row_scale ? rs_inner : 1;
        p.scale_on = y_narrow ? 1 : 2;
        p.M = M;
        const int nt1 = p.NA <= 16 ? 1 : (p.NA <= 32 ? 2 : 4);
        const int TA = 16 * nt1;
        p.nta = (p.NA + TA - 1) / TA;
        const int ntb = (p.NB + T2 - 1) / T2;
        const int ntiles = p.nta * ntb;
        int64_t msplit = 640 / ntiles;        // ~2.5 blocks per CU: every block then adds its tile atomically once, and
        if (msplit < 1) msplit = 1;           // 2048 blocks hammering the same few KB of dW were atomic-contention bound
        if (msplit > kchunks) msplit = kchunks;
        if (msplit > 65535) msplit = 65535;
        const int64_t cps = (kchunks + msplit - 1) / msplit;
        msplit = (kchunks + cps - 1) / cps;
        p.rows_per_split = cps * TK;
        const dim3 grid(ntiles, (unsigned)msplit);
        if (nt1 == 1) hipLaunchKernelGGL(wgrad_tn_fast_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (nt1 == 2) hipLaunchKernelGGL(wgrad_tn_fast_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(wgrad_tn_fast_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
        STG_LAUNCH_CHECK();
        return 0;
    }
    WgradParams p;
    p.dY = (const bf16_t*)dY; p.lddy = lddy; p.X = (const bf16_t*)X; p.ldx = ldx;
    p.dW = dW; p.lddw = lddw; p.db = db; p.M = M; p.N1 = N1; p.N2 = N2;
    p.nt1 = (N1 + T1 - 1) / T1; p.nt2 = (N2 + T2 - 1) / T2;
    const int ntiles = p.nt1 * p.nt2;
    int64_t msplit = 2048 / ntiles;
    if (msplit < 1) msplit = 1;
    if (msplit > kchunks) msplit = kchunks;
    if (msplit > 65535) msplit = 65535;
    int64_t cps = (kchunks + msplit - 1) / msplit;  // 64-row chunks per split
    msplit = (kchunks + cps - 1) / cps;
    p.msplit = (int)msplit; p.rows_per_split = cps * TK;
    p.row_scale = row_scale; p.rs_outer = row_scale ? rs_outer : 1; p.rs_inner = row_scale ? rs_inner : 1;
    p.vec_y = (lddy % 8 == 0) && (((uintptr_t)dY & 15) == 0);
    p.vec_x = (ldx % 8 == 0) && (((uintptr_t)X & 15) == 0);
    hipLaunchKernelGGL(wgrad_tn_generic_kernel, dim3(ntiles, (unsigned)msplit), dim3(256), 0, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}

#ifdef STG_GEMM_DIAG
extern "C" int stg_diag_read_stamps(unsigned long long* host, int n_workgroups) {
    if (!host || n_workgroups <= 0 || n_workgroups > 8192) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(stg_diag_stamps), sizeof(unsigned long long) * 4 * n_workgroups) == hipSuccess ? 0 : -100;
}
#endif
