// The adapters' DOWN-projection (Adapter.D_fc1 + GELU, Swin_AVE.py:15-24: [rows, C] -> [rows, d_h], d_h = 16 / 32 / 64) as a row stream.
//
// Why (round 6): on the 128 x 128 LDS-DMA GEMM kernel these classes are 128-column tiles with 16-64 live columns, a block barrier per 64-wide k-tile
// and a row-layout epilogue through LDS -- 1.6-1.8 x their byte floor (3.5 TB/s: `N32 K512 bap` 45 launches x 37 us per Swin-B step).  The op is a pure
// read stream of A with a [d_h, C] weight that fits LDS as MFMA fragments (<= 64 KB), exactly the situation of upln.hip's join kernels:
//   * a wave owns 16 rows at a time; its A fragments come STRAIGHT from global memory in the MFMA's own layout (lane (m, g) reads A[m][32 ks + 8 g .. + 7]:
//     a row's 64 bytes per k-step are four adjacent lanes) -- all KS loads of the next row tile are issued before the current tile's MFMAs;
//   * W lives in LDS in fragment order (one ds_read_b128 per MFMA), no barrier after the prologue;
//   * v_mfma_f32_16x16x32_bf16 "swapped" like gemm.hip (first operand = W rows n, second = A rows m): a lane ends up with 4 consecutive n of its row m,
//     bias + GELU + derivative in registers (the polynomial forms of the GEMM epilogue: same values bit for bit), 8-byte stores;
//   * the video | audio pair (two row groups with their own weight, stg_gemm_nt's split mode) is one launch: a workgroup serves one group.
// Same k order as the GEMM kernels (k-steps ascending into one accumulator): results are bit-identical to stg_gemm_nt's.
#include "common.h"
#include "../../include/stgcma.h"

namespace {

struct SkP {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; const bf16_t* W2; int64_t ldw;
    const float* bias; const float* bias2;
    bf16_t* C; int64_t ldc;
    bf16_t* dact; int64_t ldp;
    int64_t M, split_m;
    int N, K;
    int nb1;                           // split mode: workgroups [0, nb1) serve rows [0, split_m), the rest rows [split_m, M)
};

template <int NT, int KS>
__global__ void __launch_bounds__(256, 2) skinny_down_kernel(SkP p) {
    extern __shared__ __attribute__((aligned(16))) uint4 wfrag[];              // [NT][KS][64] fragments, then NT * 16 bias floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    int bid = blockIdx.x, nbk = gridDim.x;
    int64_t row0 = 0;
    if (p.nb1 > 0) {                                                           // block-uniform: this workgroup's row group
        if (bid >= p.nb1) { p.W = p.W2; p.bias = p.bias2; row0 = p.split_m; bid -= p.nb1; nbk -= p.nb1; }
        else { p.M = p.split_m; nbk = p.nb1; }
    }
    for (int f = tid; f < NT * KS * 64; f += 256) {
        const int l = f & 63, ks = (f >> 6) % KS, t = (f >> 6) / KS;
        const int n = 16 * t + (l & 15), k0 = 32 * ks + 8 * (l >> 4);
        wfrag[f] = *reinterpret_cast<const uint4*>(p.W + (int64_t)n * p.ldw + k0);
    }
    float* sb = reinterpret_cast<float*>(wfrag + NT * KS * 64);
    for (int c = tid; c < NT * 16; c += 256) sb[c] = p.bias ? p.bias[c] : 0.f;
    __syncthreads();
    float bv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[t][r] = sb[16 * t + 4 * g + r];

    const int64_t ntile = (p.M - row0 + 15) >> 4;
    const int64_t stride = (int64_t)nbk * 4;
    int64_t tile = (int64_t)bid * 4 + wave;
    auto load = [&](int64_t tl, bf16x8_t (&af)[KS]) {
        int64_t row = row0 + tl * 16 + m;
        row = row < p.M ? row : p.M - 1;                                        // clamped: loads stay unconditional, stores are predicated
        const bf16_t* ap = p.A + row * p.lda + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const bf16x8_t*>(ap + 32 * ks);
    };
    bf16x8_t cur[KS], nxt[KS];
    if (tile < ntile) load(tile, cur);
    for (; tile < ntile; tile += stride) {
        const bool more = tile + stride < ntile;
        if (more) load(tile + stride, nxt);
        f32x4_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        int lo = lane;                                      // laundered: as loop invariants the compiler hoists all NT * KS fragments into registers (128 VGPRs at 32 x 512)
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if ((ks & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // keeps the LDS fragment reads from piling up in VGPRs
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag[(t * KS + ks) * 64 + lo]), cur[ks], acc[t], 0, 0, 0);
        }
        const int64_t row = row0 + tile * 16 + m;
        if (row < p.M) {
            bf16_t* cp = p.C + row * p.ldc + 4 * g;
            bf16_t* dp = p.dact + row * p.ldp + 4 * g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x2_t y0, d0, y1, d1;
                gelu_pw_both((f32x2_t){acc[t][0] + bv[t][0], acc[t][1] + bv[t][1]}, y0, d0);
                gelu_pw_both((f32x2_t){acc[t][2] + bv[t][2], acc[t][3] + bv[t][3]}, y1, d1);
                *reinterpret_cast<uint2*>(cp + 16 * t) = make_uint2(pack_bf2(y0.x, y0.y), pack_bf2(y1.x, y1.y));
                *reinterpret_cast<uint2*>(dp + 16 * t) = make_uint2(pack_bf2(d0.x, d0.y), pack_bf2(d1.x, d1.y));
            }
        }
        if (more) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) cur[ks] = nxt[ks];
        }
    }
}

template <int NT, int KS>
int launch(const SkP& p, unsigned grid, hipStream_t st) {
    static std::atomic<uint64_t> done{0};
    constexpr int lds = NT * KS * 64 * 16 + NT * 16 * 4;
    STG_CHECK(stg_reserve_lds((skinny_down_kernel<NT, KS>), lds, done), -101, "stg_gemm_nt (skinny): cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL((skinny_down_kernel<NT, KS>), dim3(grid), dim3(256), lds, st, p);
    return 0;
}

}  // namespace

// Called by stg_gemm_nt (gemm.hip) for: bf16 operands, N in {16, 32, 64}, K in {128, 256, 512, 1024} with N * K <= 32 K elements (the weight as
// fragments fits 64 KiB of LDS), bias + GELU with a bf16 saved derivative, bf16 output, no residual / row scale, M >= 8192.  Returns 1 when the
// shape is not one of the built instantiations (the caller falls back to the tile kernels), 0 after a launch, < 0 on error.
int stg_gemm_skinny_down(const void* A, int64_t lda, const void* W, const void* W2, int64_t ldw, const float* bias, const float* bias2, void* Cout, int64_t ldc,
                         void* dact, int64_t ldp, int64_t M, int64_t split_m, int N, int K, void* stream) {
    SkP p = {};
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.W2 = (const bf16_t*)W2; p.ldw = ldw; p.bias = bias; p.bias2 = bias2;
    p.C = (bf16_t*)Cout; p.ldc = ldc; p.dact = (bf16_t*)dact; p.ldp = ldp; p.M = M; p.split_m = split_m; p.N = N; p.K = K;
    // every workgroup first fills LDS with the weight's fragments (N K x 2 bytes from L2: 32 KB at 32 x 512): at one row tile per wave that prologue is
    // half of a workgroup's traffic and nothing is prefetched -- at least four row tiles per wave, at most two rounds of two workgroups per CU
    const int64_t tiles = (M + 15) / 16;
    int64_t blocks = (tiles + 15) / 16;
    blocks = blocks > 1024 ? 1024 : blocks;
    if (split_m > 0) {
        int64_t b1 = (blocks * split_m + M / 2) / M;
        b1 = b1 < 1 ? 1 : (b1 > blocks - 1 ? blocks - 1 : b1);
        if (blocks < 2) blocks = 2, b1 = 1;
        p.nb1 = (int)b1;
    }
    hipStream_t st = (hipStream_t)stream;
    const int key = N * 10000 + K;
    int rc = 1;
    switch (key) {
        case 16 * 10000 + 128: rc = launch<1, 4>(p, (unsigned)blocks, st); break;
        case 16 * 10000 + 256: rc = launch<1, 8>(p, (unsigned)blocks, st); break;
        case 32 * 10000 + 256: rc = launch<2, 8>(p, (unsigned)blocks, st); break;
        case 32 * 10000 + 512: rc = launch<2, 16>(p, (unsigned)blocks, st); break;
        case 16 * 10000 + 512: rc = launch<1, 16>(p, (unsigned)blocks, st); break;
        case 32 * 10000 + 128: rc = launch<2, 4>(p, (unsigned)blocks, st); break;
        case 64 * 10000 + 256: rc = launch<4, 8>(p, (unsigned)blocks, st); break;
        case 64 * 10000 + 512: rc = launch<4, 16>(p, (unsigned)blocks, st); break;
        default: return 1;
    }
    if (rc) return rc;
    STG_LAUNCH_CHECK();
    return 0;
}
