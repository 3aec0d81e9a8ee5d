// Kernels of the AVS dense decoder (AVS/model/Swin_AVSModel_Base.py:14-130 Classifier_Module / ResidualConvUnit /
// FeatureFusionBlock / Interpolate, :1474-1506 ctor, :1838-1894 forward; AVS/model/TPAVI.py:57-61 BatchNorm3d).
// Feature maps are kept channels-last as token rows [F*H*W, C] bf16 -- the layout of the backbone's taps -- so a 3x3 (dilated)
// convolution is an im2col gather into [rows, 9*C] followed by the MFMA GEMM of gemm.hip (stg_gemm_nt), its data gradient the same
// gather applied to dY with the flipped kernel matrix, its weight gradient stg_wgrad_tn on (dY, im2col(X)).
#include <math.h>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

inline unsigned grid_for(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// BatchNorm 16-byte forms: a thread loads its 8 channels' constants once, so give it >= 4 rows and at most 8 workgroups per CU
inline unsigned bn_grid(int64_t R, int RL) {
    int64_t b = (R + 4 * RL - 1) / (4 * RL);
    return (unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}
inline unsigned rows_grid(int64_t rows) { return (unsigned)(rows < 8 ? 8 : rows > (1 << 20) ? (1 << 20) : (rows + 7) / 8 * 8); }

// ------------------------------------------------------------------------------------------------ im2col 3x3 (stride 1, padding = dilation)
// out[(f, h, w), (kh, kw, c)] = x[f, h + (kh-1) d, w + (kw-1) d, c] (zero outside); one thread per 16-byte piece (8 channels)
__global__ void im2col3x3_kernel(const bf16_t* x, int64_t ldx, bf16_t* out, int64_t F, int H, int W, int C, int d) {
    const int c8 = C >> 3;
    const int64_t total = F * H * W * 9 * c8;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int pc = (int)(id % c8);
        int64_t r = id / c8;
        const int tap = (int)(r % 9);
        r /= 9;                                                   // pixel row (f, h, w)
        const int w = (int)(r % W);
        const int h = (int)((r / W) % H);
        const int64_t f = r / ((int64_t)W * H);
        const int hh = h + (tap / 3 - 1) * d, ww = w + (tap % 3 - 1) * d;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = *reinterpret_cast<const uint4*>(x + ((f * H + hh) * W + ww) * ldx + 8 * pc);
        *reinterpret_cast<uint4*>(out + r * (int64_t)(9 * C) + tap * C + 8 * pc) = v;
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 convolution weight gradient
// dW[o, (kh, kw, i)] = sum_m dY[m, o] * X[pixel(m) shifted by tap (kh, kw), i]   (zero outside the frame)
// A "tn" GEMM whose reduction index is the token row m (both operands are m-major in memory) and whose X operand is the
// im2col image that is never formed: one block owns a 128 (o) x 128 (columns of ONE tap) tile of dW for a slice of the rows,
// streams 64-row tiles of dY and of the tap-shifted X through LDS by global_load_lds (zero line for padded taps and for rows
// outside the slice), reads both as k-major MFMA fragments with ds_read_b64_tr_b16, and leaves its fp32 partial tile in a
// workspace [splits, O, 9 I]; the caller sums over the splits.  (The path it replaces wrote the 9x im2col image and then ran
// the generic atomic wgrad: 38 + 13 ms per AVS step.)
struct CwP {
    const bf16_t* dy; int64_t lddy;
    const bf16_t* x; int64_t ldx;
    const bf16_t* zero;
    float* ws;                       // [splits][O][9 * I]
    float* dbws;                     // [splits][O] column sums of dY (the bias gradient), or NULL
    int64_t M; int H, W, d, O, I; int taps;   // taps = 9 (3x3 convolution) or 1 (plain dW = dY^T X: no shift, H = W = 1)
    int splits; int64_t rows_per_split;   // multiple of 64
    int nto, ntc;                    // tiles over O and over 9 * I
    int64_t ws_bstride;              // batched (blockIdx.y = problem): rows b*M .. b*M + M - 1 of both operands, ws + b*ws_bstride
};

typedef short cw_s4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int cw_swz(int row) { return (row & 7) << 1; }          // 16-byte chunk XOR (16 chunks per 256-byte row)

// k-major fragment for v_mfma_f32_16x16x32_bf16 from a row-major [64 m][128 c] LDS tile: lane (i = l & 15, kq = l >> 4) gets
// tile[32 s + 8 kq + j][16 t + i], j = 0..7, by two transposing reads (4 rows x 16 columns per 16-lane group each).
// The reads are INLINE ASM: behind an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every __builtin_amdgcn_ds_read_tr16_b64 (the
// builtin carries no memory operand, so the wait-count pass must assume it reads what the DMA writes) -- which waits for the prefetch that was just
// issued and serialises the pipeline (measured: the round-5a kernel ran at the DMA's latency).  cw_wait8 is the matching `s_waitcnt lgkmcnt(0)`,
// tied to the fragment registers so that no use can be scheduled above it.
__device__ __forceinline__ uint32_t cw_lds_addr(const bf16_t* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ bf16x8_t cw_frag(const bf16_t* tile, int t, int s, int lane) {
    const int gi = lane & 15, kq = lane >> 4;
    const int r0 = 32 * s + 8 * kq + (gi >> 2), r1 = r0 + 4;
    const int ch = 2 * t + ((gi & 3) >> 1), sub = (gi & 1) * 4;
    const uint32_t a0 = cw_lds_addr(tile + r0 * 128 + ((ch ^ cw_swz(r0)) << 3) + sub);
    const uint32_t a1 = cw_lds_addr(tile + r1 * 128 + ((ch ^ cw_swz(r1)) << 3) + sub);
    cw_s4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
__device__ __forceinline__ void cw_wait8(bf16x8_t (&a)[4], bf16x8_t (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}

// Round 5: (1) NBUF LDS buffers of ROWS rows: the DMA of row block mb + (NBUF - 1) ROWS is issued right after the barrier that publishes block mb and
// lands while block mb is multiplied; the wait in front of a barrier is COUNTED (only the oldest block must have landed).  Measured on the decoder's
// shapes: <64, 2> (one block ahead, 64 KiB, two workgroups per CU) 529-584 TFLOP/s, <32, 4> (three ahead, twice the barriers) 485-554 -- the
// kernel is not bound by the DMA's latency once the compiler's hidden vmcnt(0) is gone (cw_frag); <64, 2> is what runs.  (2) A 128-column tile may
// span SEVERAL taps: I % 8 == 0 suffices (each 16-byte piece knows its own tap: round 5a I % 64, round 5b any multiple of 8), which takes the ASPP convolutions over the 64- and 320-channel maps
// off the im2col + atomic-wgrad path; columns >= taps * I are zero-filled and never stored.  (3) Row splits chosen to fill the launch's last round
// of workgroups (cw_ws_floats): 470-520 -> 600-680 TFLOP/s together with (1).
template <int ROWS, int NBUF>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(CwP p) {
    constexpr int J = ROWS / 16;                                              // 16-byte pieces per thread and tile: ROWS x 16 chunks / 256 threads
    constexpr int BUF = 2 * ROWS * 128;                                       // bf16 elements per buffer: dY tile + X tile, [ROWS][128] each
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];             // NBUF buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave >> 1, wc = wave & 1;
    p.dy += (int64_t)blockIdx.y * p.M * p.lddy;
    p.x += (int64_t)blockIdx.y * p.M * p.ldx;
    p.ws += (int64_t)blockIdx.y * p.ws_bstride;
    // workgroups are dealt round-robin to the 8 XCDs: ALL tiles of a row split go to one XCD (split = 8 * s_hi + xcd), so the dY tile the ntc column
    // tiles share and the X rows the nine taps share are fetched into ONE L2 instead of up to eight (the grid covers ceil(splits / 8) * 8 splits)
    const int ntile = p.nto * p.ntc;
    const int kq = blockIdx.x >> 3;
    const int tile = kq % ntile, split = (kq / ntile) * 8 + (blockIdx.x & 7);
    if (split >= p.splits) return;
    const int to = tile / p.ntc, tc = tile - to * p.ntc;
    const int o0 = to * 128, col0 = tc * 128;
    const int ncol = p.taps * p.I;
    const int64_t r0 = (int64_t)split * p.rows_per_split;
    int64_t r1 = r0 + p.rows_per_split;
    r1 = r1 < p.M ? r1 : p.M;

    // DMA pieces: chunk q = j * 256 + tid (j < J) -> tile row q >> 4, position q & 15 <- source chunk (q & 15) ^ swz(row); the source chunk is the
    // SAME for all of a thread's pieces (row = 16 j + (tid >> 4): row & 7 does not depend on j), hence one column, one tap, one pixel shift per thread
    const int prow0 = tid >> 4;
    const int pch = (tid & 15) ^ cw_swz(prow0);
    int py[J], px[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t m = r0 + prow0 + 16 * j;
        const int rem = (int)(m % ((int64_t)p.H * p.W));
        py[j] = rem / p.W;
        px[j] = rem - py[j] * p.W;
    }
    const int col = col0 + pch * 8;
    const int tap = col / p.I;
    const int pi0 = col < ncol ? col - tap * p.I : -1;                        // -1: a column past the last tap (zero-filled)
    const int pdy = p.taps == 9 ? (tap / 3 - 1) * p.d : 0;
    const int pdx = p.taps == 9 ? (tap % 3 - 1) * p.d : 0;
    const int64_t pshift = (int64_t)pdy * p.W + pdx;
    const bool ocol = o0 + pch * 8 < p.O;                                     // O % 128 != 0: zero columns
    const bf16_t* ybase = p.dy + o0 + pch * 8;
    const bf16_t* xbase = p.x + pshift * p.ldx + (pi0 >= 0 ? pi0 : 0);
    const int adv_y = ROWS / p.W, adv_x = ROWS - adv_y * p.W;

    f32x4_t acc[4][4];               // [col tile][o tile]: D[col][o], lane o = l & 15, registers = 4 consecutive columns
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // bias gradient: column sums of dY ride on the MFMA pipe (ones as the other operand) in the waves that see each dY tile first
    const bool do_db = p.dbws != nullptr && tc == 0 && wc == 0;
    f32x4_t accb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) accb[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;

    auto issue = [&](int64_t mb, int buf) {                  // row block mb -> buffer buf; advances the pieces' pixel coordinates by ROWS rows.
        bf16_t* sY = smem + buf * BUF;                       // ALWAYS 2 J DMA instructions (rows past the split read the zero line): the counted wait
        bf16_t* sX = sY + ROWS * 128;                        // below relies on it
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int64_t m = mb + prow0 + 16 * j;
            const bool live = m < r1;
            const int yy = py[j] + pdy, xx = px[j] + pdx;
            const bool in = live && pi0 >= 0 && (p.taps == 1 || (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W));
            const bf16_t* sy = (live && ocol) ? ybase + m * p.lddy : p.zero;
            const bf16_t* sx = in ? xbase + m * p.ldx : p.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sy,
                                             (__attribute__((address_space(3))) void*)(sY + (j * 256 + wave * 64) * 8), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx,
                                             (__attribute__((address_space(3))) void*)(sX + (j * 256 + wave * 64) * 8), 16, 0, 0);
            px[j] += adv_x; py[j] += adv_y;                  // this row's pixel ROWS rows further on
            if (px[j] >= p.W) { px[j] -= p.W; py[j] += 1; }
            if (py[j] >= p.H) { py[j] -= p.H; if (py[j] >= p.H) py[j] -= p.H; }
        }
    };
    // NBUF - 1 row blocks in flight: the DMA of block mb + (NBUF - 1) ROWS is issued when block mb is about to be multiplied, and the wait in front of
    // a block's barrier is COUNTED -- only the oldest block must have landed (the 2-buffer form of round 5a waited for its one prefetch at every
    // barrier: with ~2 us to HBM against ~0.3 us of MFMA work per block the kernel ran at the DMA's latency, 0.5 PFLOP/s)
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) issue(r0 + (int64_t)s * ROWS, s);
    int cur = 0, nxt = NBUF - 1;
    for (int64_t mb = r0; mb < r1; mb += ROWS) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * 2 * J) : "memory");     // this wave's pieces of block mb have landed ...
        __builtin_amdgcn_s_barrier();                       // ... and everyone's; every wave is done reading buffer `nxt` (block mb - ROWS)
        issue(mb + (int64_t)(NBUF - 1) * ROWS, nxt);
        const bf16_t* sY = smem + cur * BUF;
        const bf16_t* sX = sY + ROWS * 128;
#pragma unroll
        for (int s2 = 0; s2 < ROWS / 32; ++s2) {
            bf16x8_t fy[4], fx[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fy[t] = cw_frag(sY, wo * 4 + t, s2, lane);
                fx[t] = cw_frag(sX, wc * 4 + t, s2, lane);
            }
            cw_wait8(fy, fx);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot)
                    acc[ct][ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[ct], fy[ot], acc[ct][ot], 0, 0, 0);
            if (do_db) {
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) accb[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fy[ot], accb[ot], 0, 0, 0);
            }
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dummy tail DMAs land before the workgroup's LDS is handed on
    // partial tile -> workspace: lane (o = l & 15, g = l >> 4) holds columns 4 g .. 4 g + 3 of each 16-column tile
    const int o_l = lane & 15, g = lane >> 4;
    if (do_db && g == 0) {                               // every "column" of accb is the same sum: lane (o, 0) register 0
        float* dbp = p.dbws + (int64_t)split * p.O + o0 + wo * 64;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
            if (o0 + wo * 64 + ot * 16 + o_l < p.O) dbp[ot * 16 + o_l] = accb[ot][0];
    }
    const int64_t ldws = p.taps * (int64_t)p.I;
    float* wsp = p.ws + ((int64_t)split * p.O + o0 + wo * 64) * ldws + col0 + wc * 64;
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (o0 + wo * 64 + ot * 16 + o_l < p.O && col0 + wc * 64 + ct * 16 + 4 * g < ncol)
                *reinterpret_cast<f32x4_t*>(wsp + (int64_t)(ot * 16 + o_l) * ldws + ct * 16 + 4 * g) = acc[ct][ot];
}

// out[b, i] (+)= sum_s ws[b, s, i]: the row-slice partial tiles of conv_wgrad_kernel folded (fp32, 16-byte accesses)
__global__ void sum_splits_kernel(const float* ws, float* out, int S, int64_t n4, int64_t batch, int accumulate) {
    const int64_t total = batch * n4;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = id / n4, i = id - b * n4;
        const float4* src = reinterpret_cast<const float4*>(ws) + b * S * n4 + i;
        float4 a = accumulate ? reinterpret_cast<const float4*>(out)[id] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sidx = 0; sidx < S; ++sidx) {
            const float4 v = src[(int64_t)sidx * n4];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(out)[id] = a;
    }
}

// ------------------------------------------------------------------------------------------------ bilinear x2 (F.interpolate)
__device__ __forceinline__ void src_index(int o, int in_size, int out_size, int align, int& i0, int& i1, float& w1) {
    float s;
    if (align) s = out_size > 1 ? o * (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    else { s = (o + 0.5f) * (float)in_size / (float)out_size - 0.5f; s = s < 0.f ? 0.f : s; }
    i0 = (int)s;
    i0 = i0 > in_size - 1 ? in_size - 1 : i0;
    i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
    w1 = s - (float)i0;
}
// one workgroup per output row (f, oh) -- the row's source rows and weight are wave-uniform and the per-item index math is 32-bit (the round-4 form
// paid four 64-bit divisions per 16-byte piece: 2.0 TB/s on the decoder's 896 x 896 x 128 map)
// workgroups are dealt round-robin to the 8 XCDs: give each XCD a CONTIGUOUS band of rows, so the 2 (forward) / 7 (adjoint) neighbouring rows that
// share source rows meet in one L2 instead of each XCD fetching its own copy
// (the grid is a multiple of 8 workgroups -- rows_grid -- so b -> (b % 8) * grid / 8 + b / 8 is a bijection of [0, grid); rows past the end are skipped)
__device__ __forceinline__ unsigned xcd_row(unsigned b, unsigned grid) { return (b & 7) * (grid >> 3) + (b >> 3); }
__global__ void __launch_bounds__(256) bilinear_up2_fwd_kernel(const bf16_t* x, bf16_t* y, int64_t F, int H, int W, int C, int align) {
    const int OH = 2 * H, OW = 2 * W, c8 = C >> 3;
    const unsigned per_row = (unsigned)OW * (unsigned)c8;
    for (int64_t pass0 = 0; pass0 < F * OH; pass0 += gridDim.x) {
        const int64_t row = pass0 + xcd_row(blockIdx.x, gridDim.x);
        if (row >= F * OH) continue;                                       // (uniform per workgroup)
        const int64_t f = row / OH;
        const int oh = (int)(row - f * OH);
        int h0, h1; float lh;
        src_index(oh, H, OH, align, h0, h1, lh);
        const bf16_t* r0 = x + (f * H + h0) * (int64_t)W * C;
        const bf16_t* r1 = x + (f * H + h1) * (int64_t)W * C;
        bf16_t* yo = y + row * (int64_t)OW * C;
        for (unsigned i = threadIdx.x; i < per_row; i += 256) {
            const unsigned ow = i / (unsigned)c8, pc = i - ow * (unsigned)c8;
            int w0, w1i; float lw;
            src_index((int)ow, W, OW, align, w0, w1i, lw);
            const u16x8 a = *reinterpret_cast<const u16x8*>(r0 + (unsigned)w0 * (unsigned)C + 8 * pc), b = *reinterpret_cast<const u16x8*>(r0 + (unsigned)w1i * (unsigned)C + 8 * pc);
            const u16x8 c = *reinterpret_cast<const u16x8*>(r1 + (unsigned)w0 * (unsigned)C + 8 * pc), e = *reinterpret_cast<const u16x8*>(r1 + (unsigned)w1i * (unsigned)C + 8 * pc);
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float top = bf2f(a.v[j]) * (1.f - lw) + bf2f(b.v[j]) * lw;
                const float bot = bf2f(c.v[j]) * (1.f - lw) + bf2f(e.v[j]) * lw;
                o.v[j] = f2bf(top * (1.f - lh) + bot * lh);
            }
            *reinterpret_cast<u16x8*>(yo + (size_t)i * 8) = o;
        }
    }
}
// gather form of the adjoint: input pixel (h, w) collects from the (at most 6 x 6) outputs whose two source rows / columns
// include it -- no atomics.  Candidates: outputs 2h-3 .. 2h+3 (covers both align modes for an exact x2 resize).
__global__ void __launch_bounds__(256) bilinear_up2_bwd_kernel(const bf16_t* dy, bf16_t* dx, int64_t F, int H, int W, int C, int align) {
    const int OH = 2 * H, OW = 2 * W, c8 = C >> 3;
    const unsigned per_row = (unsigned)W * (unsigned)c8;
    for (int64_t pass0 = 0; pass0 < F * H; pass0 += gridDim.x) {             // one workgroup per input row (f, h): row weights are wave-uniform
        const int64_t row = pass0 + xcd_row(blockIdx.x, gridDim.x);
        if (row >= F * H) continue;
        const int64_t f = row / H;
        const int h = (int)(row - f * H);
        float whs[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int oh = 2 * h - 3 + k;
            whs[k] = 0.f;
            if (oh >= 0 && oh < OH) {
                int h0, h1; float lh;
                src_index(oh, H, OH, align, h0, h1, lh);
                whs[k] = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
            }
        }
        const bf16_t* fbase = dy + f * (int64_t)OH * OW * C;
        bf16_t* xo = dx + row * (int64_t)W * C;
        for (unsigned i = threadIdx.x; i < per_row; i += 256) {
            const unsigned w = i / (unsigned)c8, pc = i - w * (unsigned)c8;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            float wws[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int ow = 2 * (int)w - 3 + k;
                wws[k] = 0.f;
                if (ow >= 0 && ow < OW) {
                    int t0, t1; float lw;
                    src_index(ow, W, OW, align, t0, t1, lw);
                    wws[k] = (t0 == (int)w ? 1.f - lw : 0.f) + (t1 == (int)w ? lw : 0.f);
                }
            }
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const float wh = whs[kh];
                if (wh == 0.f) continue;
                const bf16_t* rbase = fbase + (int64_t)(2 * h - 3 + kh) * OW * C + 8 * pc;
#pragma unroll
                for (int kw = 0; kw < 7; ++kw) {
                    const float ww = wws[kw];
                    if (ww == 0.f) continue;
                    const u16x8 g = *reinterpret_cast<const u16x8*>(rbase + (int64_t)(2 * (int)w - 3 + kw) * C);
                    const float wt = wh * ww;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += wt * bf2f(g.v[j]);
                }
            }
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = f2bf(acc[j]);
            *reinterpret_cast<u16x8*>(xo + (size_t)i * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm over rows
// Column sums of x (and of x*x, or of dy and dy*xhat) over R rows of a [R, C] bf16 tensor: blocks of 64 channel-threads x 4 row
// lanes stride over the rows, fold in LDS, one fp32 atomic per (block, channel) into out[2][C] (zeroed by the caller).
__global__ void __launch_bounds__(256) colsum2_kernel(const bf16_t* a, const bf16_t* b, const float* mean, const float* rstd,
                                                      float* out, int64_t R, int C, int mode) {
    __shared__ float s0[4][64], s1[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float t0 = 0.f, t1 = 0.f;
    if (c < C) {
        const float mu = mode == 1 ? mean[c] : 0.f, rs = mode == 1 ? rstd[c] : 0.f;
        for (int64_t r = (int64_t)blockIdx.y * 4 + rl; r < R; r += (int64_t)gridDim.y * 4) {
            const float x = bf2f(a[r * C + c]);
            if (mode == 0) { t0 += x; t1 += x * x; }                           // first pass: sum x (sum x^2 unused by the caller)
            else if (mode == 2) { const float dlt = x - mean[c]; t0 += dlt; t1 += dlt * dlt; }   // second pass: centred sums
            else { const float g = bf2f(b[r * C + c]); t0 += g; t1 += g * (x - mu) * rs; }     // backward: sum dy, sum dy * xhat
        }
    }
    s0[rl][cl] = t0; s1[rl][cl] = t1;
    __syncthreads();
    if (rl == 0 && c < C) {
        atomicAdd(out + c, s0[0][cl] + s0[1][cl] + s0[2][cl] + s0[3][cl]);
        atomicAdd(out + C + c, s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl]);
    }
}
// the same for C % 8 == 0 (round 5): a thread owns 8 consecutive channels (16-byte loads; the scalar form above moved 2 bytes per lane and load:
// 0.64 TB/s on the decoder's 501 760 x 256 maps), a block = C / 8 channel pieces (<= 256) x 256 / (C / 8) row lanes
__global__ void __launch_bounds__(1024) colsum2_v8_kernel(const bf16_t* a, const bf16_t* b, const float* mean, const float* rstd,
                                                          float* out, int64_t R, int C, int mode) {
    __shared__ float red[1024 * 16];
    const int c8n = C >> 3, RL = 1024 / c8n;
    const int cp = threadIdx.x % c8n, rl = threadIdx.x / c8n;
    float t0[8], t1[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { t0[j] = 0.f; t1[j] = 0.f; mu[j] = mode ? mean[8 * cp + j] : 0.f; rs[j] = mode == 1 ? rstd[8 * cp + j] : 0.f; }
    {
        const int64_t chunk = (R + gridDim.y - 1) / gridDim.y;
        const int64_t r_end = (blockIdx.y + 1) * chunk < R ? (blockIdx.y + 1) * chunk : R;
        int64_t r = blockIdx.y * chunk + rl;
        for (; r + 3 * RL < r_end; r += 4 * RL) {                                       // 4 independent 16-byte loads (x2 in mode 1) per trip
            u16x8 xa[4], gb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xa[u] = *reinterpret_cast<const u16x8*>(a + (r + u * RL) * C + 8 * cp);
            if (mode == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) gb[u] = *reinterpret_cast<const u16x8*>(b + (r + u * RL) * C + 8 * cp);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (mode == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float g = bf2f(gb[u].v[j]); t0[j] += g; t1[j] += g * (bf2f(xa[u].v[j]) - mu[j]) * rs[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = bf2f(xa[u].v[j]) - mu[j]; t0[j] += d; t1[j] += d * d; }
                }
            }
        }
        for (; r < r_end; r += RL) {
            const u16x8 xa = *reinterpret_cast<const u16x8*>(a + r * C + 8 * cp);
            if (mode == 1) {
                const u16x8 gb = *reinterpret_cast<const u16x8*>(b + r * C + 8 * cp);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float g = bf2f(gb.v[j]); t0[j] += g; t1[j] += g * (bf2f(xa.v[j]) - mu[j]) * rs[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = bf2f(xa.v[j]) - mu[j]; t0[j] += d; t1[j] += d * d; }
            }
        }
    }
    // red[row lane][channel piece][which sum][j]; then one thread per output value folds the row lanes and issues the block's ONE atomic for it
    // (all [2, C] sums live in 2 C / 32 cache lines: the atomics of a launch serialise there, 0.2 us per workgroup measured -- hence few, fat workgroups)
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = t0[j]; red[threadIdx.x * 16 + 8 + j] = t1[j]; }
    __syncthreads();
    for (int v = threadIdx.x; v < 2 * C; v += 1024) {
        const int which = v >= C, c = which ? v - C : v;
        const int idx = (c >> 3) * 16 + which * 8 + (c & 7);
        float acc = 0.f;
        for (int r2 = 0; r2 < RL; ++r2) acc += red[r2 * c8n * 16 + idx];
        atomicAdd(out + v, acc);
    }
}
// LayerNorm parameter gradients over MANY rows (TPAVI's trainable norm_layer sees up to 500 K rows; stg_layernorm_bwd's per-row
// atomics are meant for the few rows of a classifier head): dgamma[c] += sum_r dy[r,c] (x[r,c] - mean[r]) rstd[r], dbeta[c] += sum_r dy[r,c]
__global__ void __launch_bounds__(256) ln_param_grad_kernel(const bf16_t* dy, const bf16_t* x, const float* mean, const float* rstd,
                                                            float* dgamma, float* dbeta, int64_t R, int C) {
    __shared__ float s0[4][64], s1[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float t0 = 0.f, t1 = 0.f;
    if (c < C) {
        for (int64_t r = (int64_t)blockIdx.y * 4 + rl; r < R; r += (int64_t)gridDim.y * 4) {
            const float g = bf2f(dy[r * C + c]);
            t0 += g;
            t1 += g * (bf2f(x[r * C + c]) - mean[r]) * rstd[r];
        }
    }
    s0[rl][cl] = t0; s1[rl][cl] = t1;
    __syncthreads();
    if (rl == 0 && c < C) {
        atomicAdd(dbeta + c, s0[0][cl] + s0[1][cl] + s0[2][cl] + s0[3][cl]);
        atomicAdd(dgamma + c, s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl]);
    }
}
// y = (x - mean) * rstd * gamma + beta
__global__ void bn_apply_kernel(const bf16_t* x, const float* mean, const float* rstd, const float* gamma, const float* beta, bf16_t* y,
                                int64_t R, int C) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        y[i] = f2bf((bf2f(x[i]) - mean[c]) * rstd[c] * gamma[c] + beta[c]);
    }
}
// training: dx = gamma * rstd * (dy - sum_dy / R - xhat * sum_dy_xhat / R);  eval (sums == NULL): dx = gamma * rstd * dy
__global__ void bn_bwd_kernel(const bf16_t* x, const bf16_t* dy, const float* mean, const float* rstd, const float* gamma,
                              const float* sums, bf16_t* dx, int64_t R, int C) {
    const int64_t total = R * C;
    const float inv = 1.0f / (float)R;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float g = bf2f(dy[i]);
        float v = g;
        if (sums) v = g - sums[c] * inv - (bf2f(x[i]) - mean[c]) * rstd[c] * sums[C + c] * inv;
        dx[i] = f2bf(gamma[c] * rstd[c] * v);
    }
}
// the same two for C % 8 == 0 and 256 % (C / 8) == 0 (round 5): 16-byte accesses, a thread keeps ONE channel piece (its per-channel constants in
// registers) and walks rows -- the scalar forms above paid a 64-bit modulo and 2-byte accesses per element
__global__ void __launch_bounds__(256) bn_apply_v8_kernel(const bf16_t* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                          bf16_t* y, int64_t R, int C) {
    const int c8n = C >> 3, RL = 256 / c8n;
    const int cp = threadIdx.x % c8n, rl = threadIdx.x / c8n;
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = mean[8 * cp + j]; rs[j] = rstd[8 * cp + j]; ga[j] = gamma[8 * cp + j]; be[j] = beta[8 * cp + j]; }
    for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < R; r += (int64_t)gridDim.x * RL) {
        const u16x8 xa = *reinterpret_cast<const u16x8*>(x + r * C + 8 * cp);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = f2bf((bf2f(xa.v[j]) - mu[j]) * rs[j] * ga[j] + be[j]);   // the scalar form's expression, term for term
        *reinterpret_cast<u16x8*>(y + r * C + 8 * cp) = o;
    }
}
__global__ void __launch_bounds__(256) bn_bwd_v8_kernel(const bf16_t* x, const bf16_t* dy, const float* mean, const float* rstd, const float* gamma,
                                                        const float* sums, bf16_t* dx, int64_t R, int C) {
    const int c8n = C >> 3, RL = 256 / c8n;
    const int cp = threadIdx.x % c8n, rl = threadIdx.x / c8n;
    const float inv = 1.0f / (float)R;
    float mu[8], rs[8], gm[8], s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * cp + j;
        mu[j] = mean[c]; rs[j] = rstd[c]; gm[j] = gamma[c];
        s0[j] = sums ? sums[c] : 0.f; s1[j] = sums ? sums[C + c] : 0.f;
    }
    for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < R; r += (int64_t)gridDim.x * RL) {
        const u16x8 ga = *reinterpret_cast<const u16x8*>(dy + r * C + 8 * cp);
        u16x8 xa = ga;
        if (sums) xa = *reinterpret_cast<const u16x8*>(x + r * C + 8 * cp);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = bf2f(ga.v[j]);
            float v = g;
            if (sums) v = g - s0[j] * inv - (bf2f(xa.v[j]) - mu[j]) * rs[j] * s1[j] * inv;          // the scalar form's expression, term for term
            o.v[j] = f2bf(gm[j] * rs[j] * v);
        }
        *reinterpret_cast<u16x8*>(dx + r * C + 8 * cp) = o;
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int stg_im2col3x3(const void* x, int64_t ldx, void* out, int64_t F, int H, int W, int C, int dilation, void* stream) {
    STG_CHECK(x && out, -1, "stg_im2col3x3: null pointer");
    STG_CHECK(F >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && dilation >= 1 && ldx >= C && ldx % 8 == 0, -2, "stg_im2col3x3: bad shape (C % 8 == 0)");
    STG_CHECK((((uintptr_t)x | (uintptr_t)out) & 15) == 0, -2, "stg_im2col3x3: pointers must be 16-byte aligned");
    if (F == 0) return 0;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(F * H * W * 9 * (C / 8), 256)), dim3(256), 0, ST, (const bf16_t*)x, ldx, (bf16_t*)out,
                       F, H, W, C, dilation);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bilinear_up2_fwd(const void* x, void* y, int64_t F, int H, int W, int C, int align_corners, void* stream) {
    STG_CHECK(x && y, -1, "stg_bilinear_up2_fwd: null pointer");
    STG_CHECK(F >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, -2, "stg_bilinear_up2_fwd: bad shape (C % 8 == 0)");
    if (F == 0) return 0;
    hipLaunchKernelGGL(bilinear_up2_fwd_kernel, dim3(rows_grid(F * 2 * H)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y,
                       F, H, W, C, align_corners ? 1 : 0);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bilinear_up2_bwd(const void* dy, void* dx, int64_t F, int H, int W, int C, int align_corners, void* stream) {
    STG_CHECK(dy && dx, -1, "stg_bilinear_up2_bwd: null pointer");
    STG_CHECK(F >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, -2, "stg_bilinear_up2_bwd: bad shape (C % 8 == 0)");
    if (F == 0) return 0;
    hipLaunchKernelGGL(bilinear_up2_bwd_kernel, dim3(rows_grid(F * H)), dim3(256), 0, ST, (const bf16_t*)dy, (bf16_t*)dx,
                       F, H, W, C, align_corners ? 1 : 0);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bn_colsum(const void* a, const void* b, const float* mean, const float* rstd, float* out, int64_t R, int C,
                             int mode, void* stream) {
    STG_CHECK(a && out, -1, "stg_bn_colsum: null pointer");
    STG_CHECK(mode == 0 || (mode == 2 && mean) || (mode == 1 && b && mean && rstd), -1, "stg_bn_colsum: mode 1 needs dy, mean, rstd; mode 2 needs mean");
    STG_CHECK(R >= 0 && C > 0, -2, "stg_bn_colsum: bad shape");
    if (R == 0) return 0;
    int gy = (int)((R + 255) / 256);
    if (gy > 512) gy = 512;
    const bool v8 = C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    if (v8) {
        const int RL = 1024 / (C / 8);
        int g2 = (int)((R + RL * 8 - 1) / (RL * 8));                    // >= 8 rows per row lane, at most one workgroup per CU
        if (g2 > 256) g2 = 256;
        if (g2 < 1) g2 = 1;
        hipLaunchKernelGGL(colsum2_v8_kernel, dim3(1, g2), dim3(1024), 0, ST, (const bf16_t*)a, (const bf16_t*)b, mean, rstd, out, R, C, mode);
    } else
        hipLaunchKernelGGL(colsum2_kernel, dim3((C + 63) / 64, gy), dim3(256), 0, ST, (const bf16_t*)a, (const bf16_t*)b, mean, rstd, out, R, C, mode);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_ln_param_grad(const void* dy, const void* x, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                 int64_t R, int C, void* stream) {
    STG_CHECK(dy && x && mean && rstd && dgamma && dbeta, -1, "stg_ln_param_grad: null pointer");
    STG_CHECK(R >= 0 && C > 0, -2, "stg_ln_param_grad: bad shape");
    if (R == 0) return 0;
    int gy = (int)((R + 255) / 256);
    if (gy > 512) gy = 512;
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((C + 63) / 64, gy), dim3(256), 0, ST, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd,
                       dgamma, dbeta, R, C);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bn_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y,
                            int64_t R, int C, void* stream) {
    STG_CHECK(x && mean && rstd && gamma && beta && y, -1, "stg_bn_apply: null pointer");
    STG_CHECK(R >= 0 && C > 0, -2, "stg_bn_apply: bad shape");
    if (R == 0) return 0;
    if (C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const int RL = 256 / (C / 8);
        hipLaunchKernelGGL(bn_apply_v8_kernel, dim3(bn_grid(R, RL)), dim3(256), 0, ST, (const bf16_t*)x, mean, rstd, gamma, beta, (bf16_t*)y, R, C);
    } else
        hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(R * C, 256)), dim3(256), 0, ST, (const bf16_t*)x, mean, rstd, gamma, beta, (bf16_t*)y, R, C);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bn_bwd(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma, const float* sums,
                          void* dx, int64_t R, int C, void* stream) {
    STG_CHECK(x && dy && mean && rstd && gamma && dx, -1, "stg_bn_bwd: null pointer");
    STG_CHECK(R >= 0 && C > 0, -2, "stg_bn_bwd: bad shape");
    if (R == 0) return 0;
    if (C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
        const int RL = 256 / (C / 8);
        hipLaunchKernelGGL(bn_bwd_v8_kernel, dim3(bn_grid(R, RL)), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)dy, mean,
                           rstd, gamma, sums, (bf16_t*)dx, R, C);
    } else
        hipLaunchKernelGGL(bn_bwd_kernel, dim3(grid_for(R * C, 256)), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, gamma, sums,
                           (bf16_t*)dx, R, C);
    STG_LAUNCH_CHECK();
    return 0;
}

static int64_t cw_ws_floats(int64_t M, int O, int I, int taps, int* splits_out) {
    if (M <= 0 || O <= 0 || I <= 0 || O % 8 != 0 || I % 8 != 0) return -1;       // O and taps * I are padded to 128 with zero columns inside the kernel
    const int tiles = ((O + 127) / 128) * ((taps * I + 127) / 128);
    // Row splits.  Every workgroup does the same work and two fit a CU (64 KiB of LDS each), so a launch runs in ROUNDS of 64 workgroups per XCD; a
    // split's tiles all go to one XCD (conv_wgrad_kernel), split s to XCD s % 8.  Choose splits = 8 k so that k * tiles workgroups per XCD fill
    // their last round: "~4 per CU" (1024 / tiles, rounds 1-5a) left e.g. 36 tiles x 29 splits = 2.04 rounds -- a third round at 2 % occupancy,
    // a third of the kernel's time (SQ_WAVES / duration, profiles/r05b_conv_wgrad_pmc.txt).  Ties go to the fewer splits (less workspace to fold).
    const int64_t max_splits = (M + 2047) / 2048;                      // >= 2048 rows per workgroup
    int64_t splits = max_splits < 8 ? (max_splits < 1 ? 1 : max_splits) : 8;
    if (max_splits >= 8) {
        double best = -1.0;
        for (int64_t k = 1; 8 * k <= max_splits && (k == 1 || k * tiles <= 256); ++k) {
            const int64_t b = k * tiles;
            const double fill = (double)b / (double)((b + 63) / 64 * 64);
            if (fill > best + 0.03) { best = fill; splits = 8 * k; }
        }
    }
    if (splits_out) *splits_out = (int)splits;
    return splits * O * taps * (int64_t)I;
}

static int cw_launch(const char* who, const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* zero_line, float* ws,
                     int64_t ws_floats, int64_t M, int H, int W, int O, int I, int dilation, int taps, void* stream, int batch = 1,
                     float* dbws = nullptr) {
    STG_CHECK(dy && x && zero_line && ws, -1, "%s: null pointer", who);
    STG_CHECK(M > 0 && H > 0 && W > 0 && dilation >= 1 && O % 8 == 0 && I % 8 == 0 && O > 0 && I > 0, -2,
              "%s: needs O %% 8 == 0 and I %% 8 == 0", who);
    STG_CHECK(lddy % 8 == 0 && lddy >= O && ldx % 8 == 0 && ldx >= I, -2, "%s: bad leading dimensions", who);
    // the kernel moves a piece's pixel 64 rows per block and wraps its row coordinate at most twice: needs 64 / W < 2 H
    STG_CHECK(taps == 1 || (int64_t)H * W > 32, -2, "%s: a 3x3 weight gradient needs maps of more than 32 pixels per frame (H * W = %d)", who, H * W);
    STG_CHECK((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)zero_line | (uintptr_t)ws) & 15) == 0, -2, "%s: pointers must be 16-byte aligned", who);
    int splits = 0;
    const int64_t need = cw_ws_floats(M, O, I, taps, &splits);
    STG_CHECK(need > 0 && ws_floats >= need * batch, -2, "%s: workspace too small (%lld < %lld floats)", who, (long long)ws_floats, (long long)need * batch);
    STG_CHECK(batch >= 1 && batch <= 65535, -2, "%s: bad batch", who);
    CwP p = {};
    p.ws_bstride = need;
    p.dbws = batch == 1 ? dbws : nullptr;
    p.dy = (const bf16_t*)dy; p.lddy = lddy; p.x = (const bf16_t*)x; p.ldx = ldx; p.zero = (const bf16_t*)zero_line; p.ws = ws;
    p.M = M; p.H = H; p.W = W; p.d = dilation; p.O = O; p.I = I; p.taps = taps;
    p.splits = splits;
    p.rows_per_split = ((M + splits - 1) / splits + 63) / 64 * 64;
    p.nto = (O + 127) / 128; p.ntc = (taps * I + 127) / 128;
    static std::atomic<uint64_t> lds_done{0};
    constexpr int CW_LDS = 2 * 2 * 64 * 128 * 2;                       // 2 buffers x (dY tile + X tile) x [64][128] bf16 = 64 KiB: two workgroups per CU
    const dim3 grid((unsigned)(p.nto * p.ntc * ((splits + 7) / 8 * 8)), (unsigned)batch);
    STG_CHECK(stg_reserve_lds(conv_wgrad_kernel<64, 2>, CW_LDS, lds_done), -101, "%s: cannot reserve %d bytes of LDS", who, CW_LDS);
    hipLaunchKernelGGL((conv_wgrad_kernel<64, 2>), grid, dim3(256), CW_LDS, ST, p);
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t stg_conv3x3_wgrad_ws_floats(int64_t M, int O, int I, int* splits_out) { return cw_ws_floats(M, O, I, 9, splits_out); }

extern "C" int stg_conv3x3_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* zero_line, float* ws,
                                 int64_t ws_floats, float* db_ws, int64_t F, int H, int W, int O, int I, int dilation, void* stream) {
    STG_CHECK(F > 0, -2, "stg_conv3x3_wgrad: bad F");
    return cw_launch("stg_conv3x3_wgrad", dy, lddy, x, ldx, zero_line, ws, ws_floats, F * H * W, H, W, O, I, dilation, 9, stream, 1, db_ws);
}

extern "C" int64_t stg_wgrad_wide_ws_floats(int64_t M, int N1, int N2, int* splits_out) { return cw_ws_floats(M, N1, N2, 1, splits_out); }

extern "C" int stg_wgrad_wide(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* zero_line, float* ws, int64_t ws_floats,
                              float* db_ws, int64_t M, int N1, int N2, void* stream) {
    return cw_launch("stg_wgrad_wide", dy, lddy, x, ldx, zero_line, ws, ws_floats, M, 1, 1, N1, N2, 1, 1, stream, 1, db_ws);
}

extern "C" int stg_wgrad_wide_batched(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* zero_line, float* ws,
                                      int64_t ws_floats, int64_t M, int N1, int N2, int batch, void* stream) {
    return cw_launch("stg_wgrad_wide_batched", dy, lddy, x, ldx, zero_line, ws, ws_floats, M, 1, 1, N1, N2, 1, 1, stream, batch);
}

extern "C" int stg_sum_splits(const float* ws, float* out, int splits, int64_t n, int64_t batch, int accumulate, void* stream) {
    STG_CHECK(ws && out, -1, "stg_sum_splits: null pointer");
    STG_CHECK(splits >= 1 && n >= 0 && n % 4 == 0 && batch >= 1 && (((uintptr_t)ws | (uintptr_t)out) & 15) == 0, -2,
              "stg_sum_splits: needs n %% 4 == 0 and 16-byte aligned pointers");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sum_splits_kernel, dim3(grid_for(batch * (n / 4), 256)), dim3(256), 0, ST, ws, out, splits, n / 4, batch, accumulate ? 1 : 0);
    STG_LAUNCH_CHECK();
    return 0;
}
