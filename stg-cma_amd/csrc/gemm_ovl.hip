// bf16 MFMA GEMM  C = epi(A . W^T)  for the SHORT-K shapes (K = 256 / 512), with the epilogue of a tile running INSIDE the main loop of
// the next tile (round 4).  What it replaces: the qkv / proj / fc1 Linears of the Swin block and their data gradients,
// AVE/model/Swin_AVE.py:111-127 (Mlp.fc1 + GELU), :231-243 (attn.qkv), :274-276 (attn.proj).
//
// Why.  In the 8-phase kernels (gemm.hip) a 256 x 256 tile at K = 512 is 12.9 us of main loop followed by 7 us (plain) .. 14 us (bias +
// GELU + byte derivative) of epilogue in which the matrix pipe idles: one workgroup owns the CU, so nothing else can run there, and the
// fc1 class sits at the SUM of its MFMA and its VALU / store time (398 us for 263 GFLOP).  The multi-tile form hides the DMA latency of
// the next tile behind the epilogue, not the epilogue behind MFMAs.
//
// How.  Block tile 256 x 128 x 64, 8 waves, a wave owns 128 x 32 of the output = 64 accumulator registers -- TWO sets of them.  A workgroup
// walks `ntl` consecutive column tiles of one 256-row panel; tile t accumulates into set t & 1 while the finished set of tile t - 1 is
// written out in NK slices, one 16-row m-tile per k-tile, placed in the wave's LOAD segment (fragment reads + DMA issue), i.e. while the
// other wave group's MFMA cluster runs on the same SIMD.  The epilogue is DIRECT (no LDS transposition): the W rows of a 32-column pair are
// read permuted (slot 4 g + r of tile j <-> column 8 g + 4 j + r, the upln.hip trick), so a lane owns 8 consecutive columns of its row and a
// slice is one 16-byte store of C (+ one 8-byte store of the derivative code) per lane.  16 rows x 64 bytes per instruction is the slower
// store shape on this chip ("direct epilogue"), but it no longer sits on the critical path.
//
// Pipeline (the 8-phase kernel's, cut to the narrower tile): a k-tile is three 16 KiB half-tiles A0 (rows 0-127), A1 (128-255), W (128
// columns) in a ring of 3 k-tile slots (144 KiB); per k-tile two phases -- (A0, W) and (A1, W), 16 MFMAs each behind 12 / 8 fragment reads --
// each bracketed by raw s_barriers, waves 4-7 one barrier behind waves 0-3.  Phase 0 issues A0 and the first half of W of k-tile g + 2, phase
// 1 A1 and the second half; before phase 1's first barrier `s_waitcnt vmcnt(6 + S)`: the six DMAs just issued (+ the S stores of this
// k-tile's epilogue slice, which sit between them in the in-order queue) may stay in flight, everything older -- all of k-tile g + 1 -- has
// landed.  RAW: that wait precedes a barrier which every reader of k-tile g + 1 passes first (the late group's wait precedes the barrier the
// early group crosses before its first read).  WAR: A0 / W of slot (g + 2) % 3 = slot of k-tile g - 1 are re-staged in phase 0 of g (both
// groups read them for the last time in phase 0 of g - 1), A1 in phase 1 (last read: phase 1 of g - 1, which the late group has left when
// the early group enters phase 1 of g).  The look-ahead runs across the tile boundary (same A panel, next W tile) and past the last tile as
// dummy re-loads, so the counted waits stay exact.  Same MFMA / k order as the other kernels: bit-identical results.
#include <atomic>
#include <type_traits>
#include "gemm_ovl.h"
#include "../../include/stgcma.h"

namespace {

constexpr int BK = 64;
constexpr int HT = 128 * BK;          // bf16 elements of a half-tile slot (16 KiB)
constexpr int KTE = 3 * HT;           // A0, A1, W of one k-tile (48 KiB)
constexpr int MAXTL = 8;              // column tiles per workgroup (bias staging: 8 x 128 floats = 4 KiB behind the ring)

__device__ __forceinline__ int swz(int row, int c) { return c ^ (row & 7); }

// saved-derivative byte code (STG_U8_LIN), as in gemm.hip: code = round((d + 0.14) * 200)
__device__ __forceinline__ uint2 pack_d8(const float* d) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lo = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d[j], 200.f, 28.f), j, lo);
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

// One 16-row m-tile of a wave's finished 128 x 32 piece: lane (row lrow, column group lk) holds columns 8 lk .. + 7 in (a0, a1).
template <int V>
__device__ __forceinline__ void epi_mtile(const GemmOvlParams& p, const f32x4_t a0, const f32x4_t a1, int64_t row, int col, const float* bias) {
    float t[8] = {a0[0] + bias[0], a0[1] + bias[1], a0[2] + bias[2], a0[3] + bias[3], a1[0] + bias[4], a1[1] + bias[5], a1[2] + bias[6], a1[3] + bias[7]};
    if (V == OV_GELU8) {
        float d[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {          // packed fp32 pairs, the same operations in the same order as gemm.hip's epilogue
            f32x2_t y2, d2;
            gelu_pw_both((f32x2_t){t[j], t[j + 1]}, y2, d2);
            t[j] = y2.x; t[j + 1] = y2.y; d[j] = d2.x; d[j + 1] = d2.y;
        }
        *reinterpret_cast<uint2*>(p.d8out + row * p.ldp + col) = pack_d8(d);
    }
    if (V == OV_DSRC8) {
        float v[8];
        unpack_d8(*reinterpret_cast<const uint2*>(p.d8src + row * p.ldd + col), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] *= v[j];
    }
    *reinterpret_cast<uint4*>(p.C + row * p.ldc + col) =
        make_uint4(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]), pack_bf2(t[4], t[5]), pack_bf2(t[6], t[7]));
}

template <int V, int NK>
__global__ void __launch_bounds__(512, 1) gemm_nt_ovl_kernel(GemmOvlParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t sm[];           // ring: 3 x 48 KiB; bias: MAXTL x 128 floats
    float* sbias = reinterpret_cast<float*>(sm + 3 * KTE);
    const int ntl = p.ntl;
    const int gpr = p.nbn / ntl;                         // tile groups per row panel
    const int ngrp = p.nbm * gpr;
    int gid = blockIdx.x;
    {
        const int q = ngrp >> 3, r = ngrp & 7;           // XCD-aware bijective remap: the groups of one A panel share an XCD's L2
        const int xcd = gid & 7, idx = gid >> 3;
        gid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = gid / gpr, bn0 = (gid % gpr) * ntl;
    const int64_t m0 = (int64_t)bm * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lrow = lane & 15, lk = lane >> 4;

    for (int i = tid; i < ntl * 128; i += 512) sbias[i] = p.bias ? p.bias[bn0 * 128 + i] : 0.f;      // visible behind the prologue's barrier

    // DMA pieces of a half-tile: chunk q = j * 512 + tid -> row q >> 3 (0..127), position q & 7 <- source chunk (q & 7) ^ (row & 7)
    uint32_t oa[2][2], ow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = j * 512 + tid;
        const int row = q >> 3, c = (q & 7) ^ (row & 7);
        oa[0][j] = (uint32_t)(((int64_t)row * p.lda + c * 8) * 2);
        oa[1][j] = (uint32_t)(((int64_t)(128 + row) * p.lda + c * 8) * 2);
        ow[j] = (uint32_t)(((int64_t)row * p.ldw + c * 8) * 2);
    }
    const char* baseA = reinterpret_cast<const char*>(p.A + m0 * p.lda);
    const char* baseW = reinterpret_cast<const char*>(p.W + (int64_t)bn0 * 128 * p.ldw);
    const int64_t wtile = (int64_t)128 * p.ldw * 2;

    int ti = 0, ki = 0, si = 0;                          // issue cursor: column tile, k-tile, ring slot of the k-tile being staged
    auto issue = [&](int which) {                        // which = 0: A0 + first half of W;  1: A1 + second half of W
        const bool in = ti < ntl;                        // past the group's last tile: a dummy re-load (keeps the counted waits exact)
        const int tt = in ? ti : ntl - 1, kk = in ? ki : NK - 1;
        bf16_t* dst = sm + si * KTE + wave * 512;
        const char* ga = baseA + (size_t)kk * (BK * 2);
        const char* gw = baseW + tt * wtile + (size_t)kk * (BK * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + oa[which][j]),
                                             (__attribute__((address_space(3))) void*)(dst + which * HT + j * 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + ow[which]),
                                         (__attribute__((address_space(3))) void*)(dst + 2 * HT + which * 4096), 16, 0, 0);
    };
    auto advance = [&]() {
        if (++ki == NK) { ki = 0; ++ti; }
        si = si == 2 ? 0 : si + 1;
    };

    // fragment offsets inside a half-tile (bf16 elements): A rows wr*64 + mi*16 + lrow; W rows of the wave's 32 columns in PAIR order --
    // slot lrow of tile j is column 8 (lrow >> 2) + 4 j + (lrow & 3) -- so that lane (lrow, lk) ends up with columns 8 lk .. + 7 of row lrow
    int offA[4][2], offW[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { const int r = wr * 64 + mi * 16 + lrow; offA[mi][s2] = r * BK + swz(r, 4 * s2 + lk) * 8; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int r = wc * 32 + 8 * (lrow >> 2) + 4 * j + (lrow & 3); offW[j][s2] = r * BK + swz(r, 4 * s2 + lk) * 8; }
    }

    f32x4_t acc[2][8][2];                                // [set][m-tile: 0-3 rows of A0, 4-7 rows of A1][column half j]
    float bias[8];                                       // bias of the tile whose epilogue is running (its 8 columns of this lane)
    const int64_t rbase = m0 + wr * 64 + lrow;           // row of m-tile mt: rbase + (mt >> 2) * 128 + (mt & 3) * 16
    const int cbase = wc * 32 + 8 * lk;                  // column inside a tile

    // prologue: k-tiles 0 and 1 of the first tile; the second stays in flight
    issue(0); issue(1); advance();
    issue(0); issue(1); advance();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run one barrier behind

    int sc = 0;                                          // ring slot of the k-tile being computed
    auto tile_body = [&](auto CURC, int t) {
        constexpr int cur = decltype(CURC)::value;
        constexpr int prv = cur ^ 1;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[cur][mt][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const bool has_prev = t > 0;
        int colp = 0;
        if (has_prev) {
            const float4 b0 = *reinterpret_cast<const float4*>(sbias + (t - 1) * 128 + cbase);
            const float4 b1 = *reinterpret_cast<const float4*>(sbias + (t - 1) * 128 + cbase + 4);
            bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
            colp = (bn0 + t - 1) * 128 + cbase;
        }
#pragma unroll 1
        for (int kt = 0; kt < NK; ++kt) {
            const bf16_t* sl = sm + sc * KTE;
            bf16x8_t af[4][2], wf[2][2];
            // ---------------- phase 0: (A0, W)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) wf[j][s2] = *reinterpret_cast<const bf16x8_t*>(sl + 2 * HT + offW[j][s2]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) af[mi][s2] = *reinterpret_cast<const bf16x8_t*>(sl + offA[mi][s2]);
            issue(0);
            if (has_prev) {                              // epilogue slice of the previous tile: m-tile(s) kt (wave-uniform switch)
                constexpr int MPS = 8 / NK;              // m-tiles per slice
#define OVL_MT(mt_) epi_mtile<V>(p, acc[prv][mt_][0], acc[prv][mt_][1], rbase + ((mt_) >> 2) * 128 + ((mt_) & 3) * 16, colp, bias);
#define OVL_SLICE(i) case i: { OVL_MT((i) * MPS) if (MPS == 2) { OVL_MT(((i) * MPS + 1) & 7) } } break;
                switch (kt) {
                    OVL_SLICE(0) OVL_SLICE(1) OVL_SLICE(2) OVL_SLICE(3)
                    default:
                        if (NK == 8) {
                            switch (kt) { OVL_SLICE(4) OVL_SLICE(5) OVL_SLICE(6) OVL_SLICE(7) default: break; }
                        }
                        break;
                }
#undef OVL_SLICE
#undef OVL_MT
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[cur][mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s2], af[mi][s2], acc[cur][mi][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 1: (A1, W)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) af[mi][s2] = *reinterpret_cast<const bf16x8_t*>(sl + HT + offA[mi][s2]);
            issue(1);
            advance();
            // all but the six DMAs of this k-tile (and the stores of its epilogue slice, issued between them) have landed
            constexpr int NST = (V == OV_GELU8 ? 2 : 1) * (8 / NK);
            if (has_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[cur][4 + mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s2], af[mi][s2], acc[cur][4 + mi][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            sc = sc == 2 ? 0 : sc + 1;
        }
    };
#pragma unroll 1
    for (int t = 0; t < ntl; t += 2) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntl) tile_body(std::integral_constant<int, 1>{}, t + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();           // pairs with the late group's last barrier

    // the last tile's epilogue: nothing left to hide it behind
    {
        const int t = ntl - 1;
        const float4 b0 = *reinterpret_cast<const float4*>(sbias + t * 128 + cbase);
        const float4 b1 = *reinterpret_cast<const float4*>(sbias + t * 128 + cbase + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
        const int colp = (bn0 + t) * 128 + cbase;
        if (t & 1) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) epi_mtile<V>(p, acc[1][mt][0], acc[1][mt][1], rbase + (mt >> 2) * 128 + (mt & 3) * 16, colp, bias);
        } else {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) epi_mtile<V>(p, acc[0][mt][0], acc[0][mt][1], rbase + (mt >> 2) * 128 + (mt & 3) * 16, colp, bias);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dummy tail DMAs must land before the workgroup's LDS is handed on
}

constexpr int OVL_LDS = 3 * KTE * 2 + MAXTL * 128 * 4;
std::atomic<uint64_t> ovl_done[6];

template <int V, int NK>
int launch(const GemmOvlParams& p, unsigned grid, hipStream_t st, int slot) {
    STG_CHECK(stg_reserve_lds(gemm_nt_ovl_kernel<V, NK>, OVL_LDS, ovl_done[slot]), -101, "stg_gemm_nt: cannot reserve %d bytes of LDS", OVL_LDS);
    hipLaunchKernelGGL((gemm_nt_ovl_kernel<V, NK>), dim3(grid), dim3(512), OVL_LDS, st, p);
    STG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool stg_gemm_ovl_supported(int variant, int64_t M, int N, int K) {
    return (variant == OV_PLAIN || variant == OV_GELU8 || variant == OV_DSRC8) && M >= 256 && M % 256 == 0 && N % 128 == 0 && K == 512;
}

int stg_gemm_ovl_launch(int variant, GemmOvlParams p, void* stream) {
    STG_CHECK(stg_gemm_ovl_supported(variant, p.M, p.N, p.K), -2, "stg_gemm_ovl: unsupported shape");
    p.nbm = (int)(p.M / 256);
    p.nbn = p.N / 128;
    // column tiles per workgroup: the longest walk (fewest un-overlapped last epilogues) that still leaves >= 2 workgroups per CU
    int ntl = 1;
    for (int c = MAXTL; c >= 1; --c)
        if (p.nbn % c == 0 && ((int64_t)p.nbm * (p.nbn / c) >= 2 * 256 || c == 1)) { ntl = c; break; }
    if (p.ntl > 0 && p.ntl <= MAXTL && p.nbn % p.ntl == 0) ntl = p.ntl;      // explicit (tools/)
    p.ntl = ntl;
    const unsigned grid = (unsigned)(p.nbm * (p.nbn / ntl));
    hipStream_t st = (hipStream_t)stream;
    if (variant == OV_PLAIN) return launch<OV_PLAIN, 8>(p, grid, st, 0);
    if (variant == OV_GELU8) return launch<OV_GELU8, 8>(p, grid, st, 1);
    return launch<OV_DSRC8, 8>(p, grid, st, 2);
}
