// Weight gradient of the trainable adapter / head Linears (autograd of Swin_AVE.py:15-16 D_fc1 / D_fc2) for the shapes that
// matter: a huge token dimension M (31 K .. 1 M rows), one NARROW operand (adapter hidden width, <= 96 columns) and one wide
// operand (the channel dimension):   D[na, nb] = sum_m A[m, na] * B[m, nb].
//
// The op is a pure HBM stream (~1 KB of operands per 8 MFMAs).  What the atomic kernel in gemm.hip pays for is (i) 16-bit
// LDS gathers to build k-major fragments and (ii) one memory-side fp32 atomic per output element per block -- each a
// 64-byte fabric transaction, more bytes than the operands themselves.  Here
//   * a WAVE owns 32-row chunks of the full narrow operand and a 128-column slab of the wide one, staged row-major into
//     wave-private LDS (no block barrier in the loop) and read back k-major with ds_read_b64_tr_b16;
//   * the four waves of a block take interleaved chunks of the block's row range and fold their accumulators through LDS, so
//     a block leaves ONE partial tile, written lane-major (256-byte coalesced stores) to a caller-owned workspace;
//   * a second kernel sums the partial tiles over the row splits and adds the result into dW / db (plain read-modify-write).
//
// MFMA v_mfma_f32_16x16x32_bf16, lane = (li = lane & 15, kg = lane >> 4).  k slot j of k-group kg is chunk row
// rho(kg, j) = 4 kg + (j & 3) + 16 (j >> 2) for BOTH operands, so one ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane
// group) serves rows 0..15 and a second rows 16..31: the eight rows touched by a 32-lane half are consecutive, and the
// 32-byte column chunks of a row are XOR-swizzled by the row so that they fall into eight different bank groups.
#include <math.h>
#include <type_traits>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr int WK = 32;        // rows per chunk
constexpr int WNB = 128;      // wide-operand columns per block
constexpr int NTB = WNB / 16;

struct Wg2 {
    const bf16_t* A; int64_t lda; int NA;      // narrow
    const bf16_t* B; int64_t ldb; int NB;      // wide
    float* ws;                                  // [S][ncg][nacc * 256] raw accumulators
    int bias_on;                                // 0: none, 1: column sums of A, 2: column sums of B
    const float* row_scale; int64_t rs_outer, rs_inner; int scale_on;   // 1: scale A rows, 2: scale B rows
    int64_t M; int64_t rows_per_block; int ncg;
};

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef short s4_t __attribute__((ext_vector_type(4)));
// [32 rows][CH 32-byte chunks]: chunk i of row r sits at slot i ^ ((r >> SH) & (CH - 1)), SH = log2(8 / CH)
template <int CH>
__device__ __forceinline__ int chunk_slot(int r, int i) {
    constexpr int SH = CH == 8 ? 0 : CH == 4 ? 1 : CH == 2 ? 2 : 3;
    return r * CH + (i ^ ((r >> SH) & (CH - 1)));
}
template <int CH>
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* s, int i, int li, int kg) {
    const int r0 = 4 * kg + (li >> 2), sub = 4 * (li & 3);
    const bf16_t* p0 = s + chunk_slot<CH>(r0, i) * 16 + sub;
    const bf16_t* p1 = s + chunk_slot<CH>(r0 + 16, i) * 16 + sub;
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p0);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p1);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__device__ __forceinline__ void scale8(uint4& v, float rs) {
    v.x = pack_bf2(__uint_as_float(v.x << 16) * rs, __uint_as_float(v.x & 0xffff0000u) * rs);
    v.y = pack_bf2(__uint_as_float(v.y << 16) * rs, __uint_as_float(v.y & 0xffff0000u) * rs);
    v.z = pack_bf2(__uint_as_float(v.z << 16) * rs, __uint_as_float(v.z & 0xffff0000u) * rs);
    v.w = pack_bf2(__uint_as_float(v.w << 16) * rs, __uint_as_float(v.w & 0xffff0000u) * rs);
}

// Up to WMAX problems of one plan (same NT1, column groups and row splits) per launch, blockIdx.z = problem: the twelve adapter
// weight gradients of a Swin block are 64 MB streams of 25 us each on their own -- launch ramp and tail, not bandwidth.
constexpr int WMAX = 16;
struct WgMulti { Wg2 p[WMAX]; };

// accumulators per lane: NT1 * 8 product tiles + 8 bias tiles (bias_on == 1 uses the first NT1), each f32x4
// Round 6: the chunk loop rewritten around its instruction count.  Rounds 2-5 computed, per 16-byte piece and chunk, a 64-bit row x leading-dimension
// product, a clamp, an exec-masked "row beyond the split" test and -- with a DropPath row scale -- a 64-bit division (~4 400 instructions per
// 32-row chunk around 27-51 MFMAs: the kernel ran at 2.3-2.4 TB/s at 48 x 768 however the launch was split).  Now a lane's pieces have constant
// 32-bit byte offsets from a wave-uniform chunk base, the LDS slots are constants, only the LAST chunk of a row split takes the clamped / zero-filled
// path (wave-uniform branch), and the row scale (template flag RS) is fetched once per row and chunk by lane row & 31 -- 32-bit index arithmetic,
// prefetched with the chunk -- and handed to the pieces' lanes by ds_bpermute.
template <int NT1, bool RS>
__global__ void __launch_bounds__(256, (NT1 > 2 ? 1 : 2)) wgrad_ws_kernel(WgMulti pm) {   // 48+ wide: 128+ accumulator registers, one wave per SIMD -- with TWO chunks in flight per wave (PF)
    // The column groups of one row split share the narrow operand's rows (re-read once per group: +37 % of the bytes at 48 x 768, +56 % at
    // 96 x 768).  blockIdx -> (cg, sp, problem) goes through a chunked XCD map (XCD x takes virtual ids [x T / 8, (x + 1) T / 8), cg fastest), so the
    // groups of a split are consecutive workgroups of ONE XCD and the re-reads hit its L2.
    const int T = gridDim.x * gridDim.y * gridDim.z;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int v = L;
    if ((T & 7) == 0) v = (L & 7) * (T >> 3) + (L >> 3);
    const int cg = v % gridDim.x, sp = (v / gridDim.x) % gridDim.y, pz = v / (gridDim.x * gridDim.y);
    const Wg2 p = pm.p[pz];
    constexpr int TA = 16 * NT1;
    constexpr int NACC = NT1 * NTB + NTB;
    constexpr int CHA = NT1 > 4 ? 8 : (NT1 == 3 ? 4 : NT1);        // A-tile row pitch in 32-byte chunks (a power of two)
    constexpr int OPER = WK * WNB + WK * 16 * CHA;                 // bf16 slots of operand LDS per wave
    constexpr int FOLD = NACC * 256 * 2;                           // bf16 slots one wave's accumulators take (fp32)
    constexpr int SMEM = 4 * OPER > FOLD ? 4 * OPER : FOLD;
    __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int b0 = cg * WNB;
    const int64_t mbeg = (int64_t)sp * p.rows_per_block;
    int64_t mend = mbeg + p.rows_per_block;
    if (mend > p.M) mend = p.M;
    bf16_t* sB = smem + wave * OPER;
    bf16_t* sA = sB + WK * WNB;

    f32x4_t acc[NT1][NTB], accb[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        accb[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT1; ++i) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const bool bias_a = p.bias_on == 1 && cg == 0;
    const bool bias_b = p.bias_on == 2;
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;

    const int na8 = (p.NA + 7) & ~7, nb8 = (p.NB + 7) & ~7;
    constexpr int PA = 2 * NT1;                                    // 16-byte pieces per A row
    // piece i of a lane: B: row (lane >> 4) + 4 i, 16-byte column piece lane & 15;  A: id = lane + 64 i -> row id / PA, piece id % PA
    int colB, rowA[NT1], colA[NT1];
    uint32_t offB[8], offA[NT1];
    int slotB[8], slotA[NT1];
    {
        int gn = b0 + (lane & 15) * 8;
        colB = gn < nb8 ? gn : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (lane >> 4) + 4 * i, pc = lane & 15;
            offB[i] = (uint32_t)(((int64_t)row * p.ldb + colB) * 2);
            slotB[i] = chunk_slot<8>(row, pc >> 1) * 16 + (pc & 1) * 8;
        }
#pragma unroll
        for (int i = 0; i < NT1; ++i) {
            const int id = lane + 64 * i;
            const int row = id / PA, pc = id % PA;
            rowA[i] = row;
            colA[i] = pc * 8 < na8 ? pc * 8 : 0;
            offA[i] = (uint32_t)(((int64_t)row * p.lda + colA[i]) * 2);
            slotA[i] = chunk_slot<CHA>(row, pc >> 1) * 16 + (pc & 1) * 8;
        }
    }
    const uint32_t rso = (uint32_t)p.rs_outer, rsi = (uint32_t)p.rs_inner;      // row-scale index arithmetic in 32 bits (host: M < 2^31)
    // PF register sets of a chunk's pieces: with one wave per SIMD (NT1 > 2: 512 registers) two chunks are in flight per wave -- the kernel is a
    // latency-bound stream (a chunk is 11-14 KB per wave and ~600 clocks of LDS + MFMA work), and a second resident workgroup does not fit
    // (NT1 = 3 at 256 registers: 34 spilled)
    constexpr int PF = NT1 > 2 ? 2 : 1;
    uint4 va[PF][NT1], vb[PF][8];
    float rsl[PF];                                                 // RS: the scale of row (lane & 31) of the chunk in flight
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_) rsl[s_] = 1.0f;
    auto gload = [&](auto SET, int64_t mb) {
        constexpr int st = decltype(SET)::value;
        const char* gB = reinterpret_cast<const char*>(p.B + mb * p.ldb);
        const char* gA = reinterpret_cast<const char*>(p.A + mb * p.lda);
        if (mb + WK <= mend) {                                     // wave-uniform: a whole chunk
#pragma unroll
            for (int i = 0; i < 8; ++i) vb[st][i] = *reinterpret_cast<const uint4*>(gB + offB[i]);
#pragma unroll
            for (int i = 0; i < NT1; ++i) va[st][i] = *reinterpret_cast<const uint4*>(gA + offA[i]);
        } else {                                                   // the split's last chunk: rows clamped here, zero-filled at commit
            const int last = (int)(mend - 1 - mb);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int row = (lane >> 4) + 4 * i;
                row = row < last ? row : last;
                vb[st][i] = *reinterpret_cast<const uint4*>(gB + ((int64_t)row * p.ldb + colB) * 2);
            }
#pragma unroll
            for (int i = 0; i < NT1; ++i) {
                const int row = rowA[i] < last ? rowA[i] : last;
                va[st][i] = *reinterpret_cast<const uint4*>(gA + ((int64_t)row * p.lda + colA[i]) * 2);
            }
        }
        if (RS && p.row_scale) {                                   // (a launch mixes problems with and without a scale: block-uniform)
            int64_t gm = mb + (lane & 31);
            gm = gm < mend ? gm : mend - 1;
            const uint32_t g32 = (uint32_t)gm;
            rsl[st] = p.row_scale[(g32 / rso) * rsi + (g32 % rsi)];
        }
    };
    const int64_t nchunks = (mend - mbeg + WK - 1) / WK;
    // one chunk: commit register set SET to LDS (scaled / zero-filled), refill the set with chunk c + 4 PF, multiply
    auto chunk = [&](auto SET, int64_t c) {
        constexpr int st = decltype(SET)::value;
        const int64_t mb = mbeg + c * WK;
        const bool tail = mb + WK > mend;                          // wave-uniform
        if (RS && p.row_scale) {
            if (p.scale_on == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) scale8(vb[st][i], __shfl(rsl[st], (lane >> 4) + 4 * i, 64));
            } else {
#pragma unroll
                for (int i = 0; i < NT1; ++i) scale8(va[st][i], __shfl(rsl[st], rowA[i], 64));
            }
        }
        if (tail) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (mb + (lane >> 4) + 4 * i >= mend) vb[st][i] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NT1; ++i)
                if (mb + rowA[i] >= mend) va[st][i] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(sB + slotB[i]) = vb[st][i];
#pragma unroll
        for (int i = 0; i < NT1; ++i) *reinterpret_cast<uint4*>(sA + slotA[i]) = va[st][i];
        if (c + 4 * PF < nchunks) gload(SET, mb + 4 * PF * WK);
        lds_fence();
        bf16x8_t af[NT1];
#pragma unroll
        for (int i = 0; i < NT1; ++i) {
            af[i] = tr_frag<CHA>(sA, i, li, kg);
            if (bias_a) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            const bf16x8_t bf = tr_frag<8>(sB, j, li, kg);
#pragma unroll
            for (int i = 0; i < NT1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
            if (bias_b) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf, accb[j], 0, 0, 0);
        }
        lds_fence();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, PF - 1>;
    int64_t c = wave;
    if (c < nchunks) gload(I0{}, mbeg + c * WK);
    if (PF == 2 && c + 4 < nchunks) gload(I1{}, mbeg + (c + 4) * WK);
    for (; c < nchunks; c += 4 * PF) {
        chunk(I0{}, c);
        if (PF == 2 && c + 4 < nchunks) chunk(I1{}, c + 4);
    }

    // fold the four waves' accumulators (3 -> 2 -> 1 -> 0), then wave 0 stores the block's partial tile lane-major
    float* fold = reinterpret_cast<float*>(smem);
    for (int w = 3; w >= 1; --w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
#pragma unroll
                for (int i = 0; i < NT1; ++i) *reinterpret_cast<f32x4_t*>(fold + ((i * NTB + j) * 64 + lane) * 4) = acc[i][j];
                *reinterpret_cast<f32x4_t*>(fold + ((NT1 * NTB + j) * 64 + lane) * 4) = accb[j];
            }
        }
        __syncthreads();
        if (wave == w - 1) {
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
#pragma unroll
                for (int i = 0; i < NT1; ++i) acc[i][j] += *reinterpret_cast<const f32x4_t*>(fold + ((i * NTB + j) * 64 + lane) * 4);
                accb[j] += *reinterpret_cast<const f32x4_t*>(fold + ((NT1 * NTB + j) * 64 + lane) * 4);
            }
        }
    }
    if (wave == 0) {
        float* out = p.ws + ((int64_t)sp * p.ncg + cg) * (NACC * 256);
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
#pragma unroll
            for (int i = 0; i < NT1; ++i) *reinterpret_cast<f32x4_t*>(out + ((i * NTB + j) * 64 + lane) * 4) = acc[i][j];
            *reinterpret_cast<f32x4_t*>(out + ((NT1 * NTB + j) * 64 + lane) * 4) = accb[j];
        }
    }
}

struct Wr2 {
    const float* ws; int S, ncg, nt1;
    float* dW; int64_t lddw; int transpose_out;
    float* db; int bias_on;
    int NA, NB;
};
struct WrMulti { Wr2 p[WMAX]; };

// Sum of the S partial tiles.  A block owns 16 consecutive float4 "quads" of the raw (lane-major) tile of column group cg;
// its 16 row-split groups stride over S with independent 16-byte loads, fold through LDS, and 16 threads scatter the 64 sums:
// raw element e = (tile * 64 + lane) * 4 + r  <->  D[16 ia + 4 (lane >> 4) + r][16 j + (lane & 15)].
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WrMulti pm) {
    const Wr2 p = pm.p[blockIdx.z];
    __shared__ float4 part[16][16];
    const int nacc = p.nt1 * NTB + NTB;
    const int per_cg = nacc * 256;
    const int q = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int e4 = blockIdx.x * 16 + q;                        // quad index: per_cg / 4 is a multiple of 16
    const int cg = blockIdx.y;
    const float4* src = reinterpret_cast<const float4*>(p.ws + (int64_t)cg * per_cg) + e4;
    const int64_t stride4 = (int64_t)p.ncg * per_cg / 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = sg; s < p.S; s += 16) {
        const float4 v = src[s * stride4];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    part[sg][q] = a;
    __syncthreads();
    if (sg != 0) return;
#pragma unroll
    for (int g = 1; g < 16; ++g) {
        const float4 v = part[g][q];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const int lane = e4 & 63, tile = e4 >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const float vals[4] = {a.x, a.y, a.z, a.w};
    if (tile < p.nt1 * NTB) {
        const int ia = tile / NTB, j = tile % NTB;
        const int nb = cg * WNB + 16 * j + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int na = 16 * ia + 4 * lg + r;
            if (na < p.NA && nb < p.NB) {
                float* dst = p.transpose_out ? p.dW + (int64_t)nb * p.lddw + na : p.dW + (int64_t)na * p.lddw + nb;
                *dst += vals[r];
            }
        }
    } else if (p.db) {
        const int j = tile - p.nt1 * NTB;
        if (p.bias_on == 1) {                      // accb[ia] = A^T . ones: every column holds the sums; take column 0
            if (cg == 0 && j < p.nt1 && li == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int na = 16 * j + 4 * lg + r;
                    if (na < p.NA) p.db[na] += vals[r];
                }
            }
        } else if (p.bias_on == 2) {               // accb[j] = ones^T . B: every row holds the sums; take row 0
            const int nb = cg * WNB + 16 * j + li;
            if (lg == 0 && nb < p.NB) p.db[nb] += vals[0];
        }
    }
}

struct Plan { bool ok; int nt1, ncg, S; int64_t rows_per_block, ws_floats; bool y_narrow; };

Plan plan_for(int64_t M, int N1, int N2, int64_t lddy, int64_t ldx, const void* dY, const void* X, int nprob = 1, int mode = -1) {
    if (mode < 0) mode = stg_opt_wgrad_plan.load(std::memory_order_relaxed);      // option wgrad_plan (A/B knob; default 2)
    Plan pl = {};
    const bool y_narrow = N1 <= N2;
    const int NA = y_narrow ? N1 : N2, NB = y_narrow ? N2 : N1;
    const bool aligned = (lddy % 8 == 0) && (ldx % 8 == 0) && (((uintptr_t)dY & 15) == 0) && (((uintptr_t)X & 15) == 0) &&
                         lddy >= ((N1 + 7) & ~7) && ldx >= ((N2 + 7) & ~7);
    if (!aligned || NA > 96 || M < 4096 || M >= (1ll << 31)) return pl;      // (the kernel's row-scale index arithmetic is 32-bit)
    pl.ok = true; pl.y_narrow = y_narrow;
    pl.nt1 = (NA + 15) / 16;                       // 1 .. 6 column tiles of the narrow operand
    pl.ncg = (NB + WNB - 1) / WNB;
    // Row splits S (round 5b).  The workgroups of a launch (ncg x S x nprob, all the same size) run in ROUNDS of `slots` (one per CU for nt1 > 2: 350+
    // VGPRs; two otherwise), and each pays a fixed price on top of its rows -- the first chunk's exposed HBM trip and the partial tile it leaves in
    // the workspace (57 KiB at nt1 = 6), priced here as OVH rows.  Rounds 1-5a took S = 512 / ncg whatever nprob was: twelve Swin-L problems of
    // 62 720 x (96, 768) ran as 6 120 workgroups of 768 rows (24 chunks: 6 per wave) at 1.9 TB/s, the fixed price about half of each.  Now: the S that
    // minimises rounds x (rows per workgroup + OVH), at least 4 chunks per wave.  Same-box A/Bs of the REPLAYED step (option wgrad_plan 0 / 1 / 2 =
    // old rule / chooser for nt1 > 2 / chooser for every width): Swin-L 133.4 -> 137.6 clips/s (0 -> 1), ViT-B 477.7 -> 481.1, Swin-B 286.8 (0) /
    // 286.7 (1) / 288.4 (2).  NOTE for narrow adapters (nt1 <= 2) the kernel ALONE, as rocprofv3 times it in the one-stream eager chain, is slower
    // with the chooser (twelve 62 720 x (32, 512) problems: 222 us as 5 904 workgroups, 304 us as 480) while the two-stream step is faster: few long
    // workgroups leave CUs to the other chain's kernels.  `value` is the two-stream step, so 2 is the default; profiles/r05b_wgrad_split_sweep.txt
    // is the stand-alone sweep.
    const int64_t slots = 256 * (pl.nt1 > 2 ? 1 : 2);
    const int64_t smax = (M + 4 * 4 * WK - 1) / (4 * 4 * WK);
    const int64_t per_s = (int64_t)pl.ncg * nprob;
    const int64_t OVH = 512;
    int64_t S = 512 / pl.ncg;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    if ((pl.nt1 > 2 && mode != 0) || mode == 2) {
        double best = 1e300;
        for (int64_t c = 1; c <= smax && c <= 1024; ++c) {
            const int64_t rounds = (per_s * c + slots - 1) / slots;
            const int64_t rows = ((M + c - 1) / c + WK - 1) / WK * WK;
            const double cost = (double)rounds * (double)(rows + OVH);
            if (cost < best * 0.98) { best = cost; S = c; }           // a later (finer) split must win by 2 %: fewer partial tiles to fold otherwise
        }
    }
    int64_t rpb = (M + S - 1) / S;
    rpb = (rpb + WK - 1) / WK * WK;
    S = (M + rpb - 1) / rpb;
    pl.S = (int)S; pl.rows_per_block = rpb;
    pl.ws_floats = S * pl.ncg * (int64_t)(pl.nt1 * NTB + NTB) * 256;
    return pl;
}

}  // namespace

extern "C" int64_t stg_wgrad_ws_floats(int64_t M, int N1, int N2) {
    if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
    int64_t need = 0;                               // per problem, whatever the number of problems its launch carries (the split depends on it)
    for (int n = 1; n <= WMAX; ++n)
        for (int mode = 0; mode < 3; ++mode) {          // ... and whichever way option wgrad_plan stands when the launch comes
            const Plan pl = plan_for(M, N1, N2, (N1 + 7) & ~7, (N2 + 7) & ~7, nullptr, nullptr, n, mode);
            if (!pl.ok) return 0;
            need = pl.ws_floats > need ? pl.ws_floats : need;
        }
    return need;
}

namespace {

void fill_problem(const Plan& pl, const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                  int64_t M, int N1, int N2, const float* row_scale, int64_t rs_outer, int64_t rs_inner, float* ws, Wg2& p, Wr2& r) {
    const bool yn = pl.y_narrow;
    p.A = (const bf16_t*)(yn ? dY : X); p.lda = yn ? lddy : ldx; p.NA = yn ? N1 : N2;
    p.B = (const bf16_t*)(yn ? X : dY); p.ldb = yn ? ldx : lddy; p.NB = yn ? N2 : N1;
    p.ws = ws;
    p.bias_on = db ? (yn ? 1 : 2) : 0;
    p.row_scale = row_scale; p.rs_outer = row_scale ? rs_outer : 1; p.rs_inner = row_scale ? rs_inner : 1;
    p.scale_on = yn ? 1 : 2;
    p.M = M; p.rows_per_block = pl.rows_per_block; p.ncg = pl.ncg;
    r.ws = ws; r.S = pl.S; r.ncg = pl.ncg; r.nt1 = pl.nt1;
    r.dW = dW; r.lddw = lddw; r.transpose_out = yn ? 0 : 1;
    r.db = db; r.bias_on = p.bias_on; r.NA = p.NA; r.NB = p.NB;
}

int launch_multi(const Plan& pl, const WgMulti& pm, const WrMulti& rm, int n, hipStream_t st) {
    const dim3 grid(pl.ncg, pl.S, n);
    bool rs = false;                                      // any problem with a row scale -> the RS instantiation (a problem without one reads its dummy 1.0)
    for (int i = 0; i < n; ++i) rs = rs || pm.p[i].row_scale != nullptr;
#define STG_WG(N) case N: if (rs) hipLaunchKernelGGL((wgrad_ws_kernel<N, true>), grid, dim3(256), 0, st, pm); \
                          else hipLaunchKernelGGL((wgrad_ws_kernel<N, false>), grid, dim3(256), 0, st, pm); break;
    switch (pl.nt1) { STG_WG(1) STG_WG(2) STG_WG(3) STG_WG(4) STG_WG(5) default: STG_WG(6) }
#undef STG_WG
    STG_LAUNCH_CHECK();
    const int per_cg = (pl.nt1 * NTB + NTB) * 256;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(per_cg / 64, pl.ncg, n), dim3(256), 0, st, rm);
    STG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int stg_wgrad_tn_ws(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                               int64_t M, int N1, int N2, const float* row_scale, int64_t rs_outer, int64_t rs_inner,
                               float* ws, int64_t ws_floats, void* stream) {
    STG_CHECK(dY && X && dW, -1, "stg_wgrad_tn_ws: null pointer");
    STG_CHECK(M >= 0 && N1 > 0 && N2 > 0, -2, "stg_wgrad_tn_ws: bad shape");
    STG_CHECK(lddy >= N1 && ldx >= N2 && lddw >= N2, -2, "stg_wgrad_tn_ws: leading dimension too small");
    if (row_scale) STG_CHECK(rs_outer > 0 && rs_inner > 0, -2, "stg_wgrad_tn_ws: bad row_scale params");
    if (M == 0) return 0;
    const Plan pl = plan_for(M, N1, N2, lddy, ldx, dY, X);
    if (!pl.ok || ws == nullptr || ws_floats < pl.ws_floats)          // shapes outside this path: the atomic kernels
        return stg_wgrad_tn(dY, lddy, X, ldx, dW, lddw, db, M, N1, N2, row_scale, rs_outer, rs_inner, stream);
    STG_CHECK((((uintptr_t)ws) & 15) == 0, -2, "stg_wgrad_tn_ws: workspace must be 16-byte aligned");
    WgMulti pm; WrMulti rm;
    fill_problem(pl, dY, lddy, X, ldx, dW, lddw, db, M, N1, N2, row_scale, rs_outer, rs_inner, ws, pm.p[0], rm.p[0]);
    return launch_multi(pl, pm, rm, 1, (hipStream_t)stream);
}

extern "C" int stg_wgrad_tn_ws_multi(const stg_wgrad_desc* d, int n, float* ws, int64_t ws_floats, void* stream) {
    STG_CHECK(d && n >= 1 && n <= WMAX, -1, "stg_wgrad_tn_ws_multi: 1..%d problems", WMAX);
    STG_CHECK(ws && (((uintptr_t)ws) & 15) == 0, -2, "stg_wgrad_tn_ws_multi: needs a 16-byte aligned workspace");
    WgMulti pm; WrMulti rm;
    Plan pl0 = {}, plm = {};
    for (int i = 0; i < n; ++i) {
        const stg_wgrad_desc& q = d[i];
        STG_CHECK(q.dY && q.X && q.dW && q.M > 0 && q.N1 > 0 && q.N2 > 0 && q.lddy >= q.N1 && q.ldx >= q.N2 && q.lddw >= q.N2, -2,
                  "stg_wgrad_tn_ws_multi: bad problem %d", i);
        if (q.row_scale) STG_CHECK(q.rs_outer > 0 && q.rs_inner > 0, -2, "stg_wgrad_tn_ws_multi: bad row_scale params");
        const Plan pl = plan_for(q.M, q.N1, q.N2, q.lddy, q.ldx, q.dY, q.X, n);
        STG_CHECK(pl.ok, -7, "stg_wgrad_tn_ws_multi: problem %d is not eligible for the workspace path", i);
        if (i == 0) {
            pl0 = plm = pl;
        }
        STG_CHECK(pl.nt1 == pl0.nt1 && pl.ncg == pl0.ncg && pl.S == pl0.S && pl.rows_per_block == pl0.rows_per_block, -7,
                  "stg_wgrad_tn_ws_multi: problem %d has a different launch plan than problem 0", i);
        for (int j = 0; j < i; ++j) STG_CHECK(d[j].dW != q.dW, -2, "stg_wgrad_tn_ws_multi: problems %d and %d share dW", j, i);
        Plan pli = pl;                                    // this problem's own operand orientation, the launch's shared row split
        pli.S = plm.S; pli.rows_per_block = plm.rows_per_block;
        fill_problem(pli, q.dY, q.lddy, q.X, q.ldx, q.dW, q.lddw, q.db, q.M, q.N1, q.N2, q.row_scale, q.rs_outer, q.rs_inner,
                     ws + (int64_t)i * pl.ws_floats, pm.p[i], rm.p[i]);
    }
    STG_CHECK(ws_floats >= (int64_t)n * pl0.ws_floats, -2, "stg_wgrad_tn_ws_multi: workspace too small");
    return launch_multi(plm, pm, rm, n, (hipStream_t)stream);
}
