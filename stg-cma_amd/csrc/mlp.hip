// Swin MLP as ONE kernel for the narrow stages (C = 128): out = fc2(GELU(fc1(y))) without the [rows, 4C] hidden tensor ever leaving
// the CU (reference: Mlp.forward, AVE/model/Swin_AVE.py:111-127, called at :790-794 on norm2(x) of both modalities).
//
// Why: at C = 128 / 256 the two GEMMs are pure streams of the hidden tensor -- fc1 writes 4C values (+ the saved derivative) per
// row, fc2 reads them back: 9 + 5 row-widths of traffic around 2 + 2 of compute, and the fc1 epilogue is VALU-bound on GELU for
// 10^9 elements.  Here a row tile's hidden values live in accumulator registers between the two products.
//
// Formulation (v_mfma_f32_16x16x32_bf16, "swapped" like gemm.hip: the token m sits on the lane).  A wave owns 32 rows (two 16-row
// tiles) and, per 128-wide chunk j of the hidden dimension,
//     S^T[h, m]  = sum_c W1[jH + h, c] * y[m, c]                       A = W1 chunk rows (LDS), B = y fragments (registers, loaded
//                                                                       once per row tile: lane (m, g) holds y[m, 32 ks + 8 g .. + 7])
//     H^T        = bf16(GELU(S^T + b1))                                in registers: an accumulator tile pair (rows 32 q + 4 g + r and
//                                                                       32 q + 16 + 4 g + r) IS the B operand of k-step q once the A
//                                                                       operand uses the same k order -- W2 is stored with its hidden
//                                                                       index permuted accordingly (stg_mlp_w2_perm), no LDS round trip
//     O^T[c, m] += sum_h W2[c, jH + h] * H^T[h, m]                      A = permuted W2 chunk rows (LDS), B = H^T (registers)
// The weights (4 x 64 KiB of chunks at C = 128) stream through a double-buffered 128 KiB LDS ring by LDS-DMA, shared by the eight
// waves of the workgroup; workgroups are persistent over row tiles so the ring never drains.  One barrier per chunk.
// Per chunk and SIMD: 2 waves x 128 MFMAs (4096 clocks) beside ~3.4 k clocks of GELU VALU in the other wave and 2048 LDS clocks.
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr int HC = 128;                  // hidden units per chunk
constexpr int MLP_WAVES = 8;
constexpr int MLP_ROWS = MLP_WAVES * 32; // rows per workgroup tile

struct MlpP {
    const bf16_t* Y; int64_t ldy;
    const bf16_t* W1; const float* b1;   // [4C, C], [4C]
    const bf16_t* W2p; const float* b2;  // [C, 4C] hidden index permuted (stg_mlp_w2_perm), [C]
    bf16_t* Out; int64_t ldo;
    int64_t rows; int ntiles;
};

__device__ __forceinline__ bf16x8_t ld16(const void* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// DIAG (diagnostics build only, -DSTG_GEMM_DIAG + option "gemm_dbg"): timing ablations -- 1: identity instead of GELU, 2: no second
// product, 3: no first product, 4: no barrier / DMA (weights of chunk 0 reused), 5: no output stores
template <int C, int DIAG = 0>
__global__ void __launch_bounds__(512, 2) mlp_fwd_kernel(MlpP p) {
    static_assert(C == 128, "one LDS row = one 256-byte weight row");
    constexpr int NCH = 4 * C / HC;                       // chunks of the hidden dimension
    constexpr int KS1 = C / 32;                           // k-steps of the first product
    constexpr int CT = C / 16;                            // output-channel tiles
    constexpr int WBYTES = HC * C * 2;                    // one chunk of W1 (= one chunk of W2): 32 KiB
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];      // 2 x (W1 chunk + W2 chunk) | 8 x 2 KiB output staging | b1, b2
    uint8_t* stage_out = smem + 4 * WBYTES;
    float* sb1 = reinterpret_cast<float*>(stage_out + MLP_WAVES * 2048);      // the biases live in LDS: a global load in front of every
    float* sb2 = sb1 + 4 * C;                                                  // GELU group cost its full L2 latency, four times per chunk
    for (int i = threadIdx.x; i < 4 * C; i += 512) sb1[i] = p.b1[i];
    if (threadIdx.x < C) sb2[threadIdx.x] = p.b2[threadIdx.x];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;

    // DMA pieces of a chunk: 32 KiB = 32 wave-instructions of 1 KiB (4 rows x 256 B); this wave issues pieces wave*4 .. +3 of each
    // matrix.  LDS position (row, pos) <- source chunk pos ^ (row & 15) (the XOR swizzle lives on the source address).
    uint32_t off1[4], off2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave * 4 + j) * 64 + lane;
        const int row = q >> 4, sc = (q & 15) ^ (row & 15);
        off1[j] = (uint32_t)((row * C + sc * 8) * 2);                    // W1 chunk: rows h, C columns
        off2[j] = (uint32_t)((row * (4 * C) + sc * 8) * 2);              // W2p chunk: rows c, pitch 4C, 128 columns of the chunk
    }
    auto issue = [&](int buf, int ch) {
        uint8_t* d1 = smem + buf * 2 * WBYTES;
        uint8_t* d2 = d1 + WBYTES;
        const char* s1 = reinterpret_cast<const char*>(p.W1) + (size_t)ch * HC * C * 2;
        const char* s2 = reinterpret_cast<const char*>(p.W2p) + (size_t)ch * HC * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s1 + off1[j]),
                                             (__attribute__((address_space(3))) void*)(d1 + (wave * 4 + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s2 + off2[j]),
                                             (__attribute__((address_space(3))) void*)(d2 + (wave * 4 + j) * 1024), 16, 0, 0);
        }
    };
    // fragment (tile t of 16 rows, 16-byte chunk position kc) of an LDS matrix with 256-byte rows
    auto frag = [&](const uint8_t* base, int t, int kc) {
        const int row = t * 16 + li;
        return ld16(base + row * 256 + ((kc ^ (row & 15)) << 4));
    };
    auto load_y = [&](int64_t tile, bf16x8_t (&yf)[2][KS1]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            int64_t m = tile * MLP_ROWS + wave * 32 + mt * 16 + li;
            m = m < p.rows ? m : p.rows - 1;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) yf[mt][ks] = ld16(p.Y + m * p.ldy + ks * 32 + g * 8);
        }
    };

    int64_t tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    bf16x8_t yf[2][KS1];
    load_y(tile, yf);
    issue(0, 0);
    int gch = 0;                                          // running chunk counter: buffer = gch & 1
    for (; tile < p.ntiles; tile += gridDim.x) {
        f32x4_t O[CT][2];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { O[ct][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; O[ct][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
        const int64_t tnext = tile + gridDim.x;
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch, ++gch) {
            const int buf = gch & 1;
            if (DIAG != 4 || gch == 0) __syncthreads();    // chunk gch has landed (its DMA was issued a chunk ago: vmcnt drained here)
                                                          // and every wave is done reading the other buffer
            const bool more = ch + 1 < NCH || tnext < p.ntiles;
            if (more && DIAG != 4) issue(buf ^ 1, (ch + 1) % NCH);
            const uint8_t* w1 = smem + (DIAG == 4 ? 0 : buf) * 2 * WBYTES;
            const uint8_t* w2 = w1 + WBYTES;
            // ---- S^T = W1c . y^T.  The A fragments of h-tile ht + 1 are read while the 8 MFMAs of tile ht issue: with two waves per
            // SIMD a ds_read_b128 consumed right away costs its whole latency (measured: 3.7 x the MFMA time)
            f32x4_t S[HC / 16][2];
            bf16x8_t af[2][KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) af[0][ks] = frag(w1, 0, ks * 4 + g);
#pragma unroll
            for (int ht = 0; ht < (DIAG == 3 ? 0 : HC / 16); ++ht) {
                if (ht + 1 < HC / 16) {
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) af[(ht + 1) & 1][ks] = frag(w1, ht + 1, ks * 4 + g);
                }
                // fence the scheduler: the next tile's reads stay ABOVE this tile's MFMAs (hipcc otherwise sinks every read to just in
                // front of its first use: read, two MFMAs, wait)
                __builtin_amdgcn_sched_barrier(0);
                S[ht][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; S[ht][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    S[ht][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ht & 1][ks], yf[0][ks], S[ht][0], 0, 0, 0);
                    S[ht][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ht & 1][ks], yf[1][ks], S[ht][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the y fragments are dead after the last chunk's first product: the next tile's rows load behind GELU, the second product
            // and the epilogue
            if (ch == NCH - 1 && tnext < p.ntiles) load_y(tnext, yf);
            // ---- H^T = bf16(GELU(S^T + b1)), packed straight into the B operands of the second product
            bf16x8_t hb[HC / 32][2];
#pragma unroll
            for (int q = 0; q < HC / 32; ++q) {
                const float4 ba = *reinterpret_cast<const float4*>(sb1 + ch * HC + 32 * q + 4 * g);
                const float4 bb = *reinterpret_cast<const float4*>(sb1 + ch * HC + 32 * q + 16 + 4 * g);
                const float bv[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float x[8];
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {     // packed fp32 pairs (common.h)
                        const f32x2_t a2 = {S[2 * q][mt][r] + bv[r], S[2 * q][mt][r + 1] + bv[r + 1]};
                        const f32x2_t b2 = {S[2 * q + 1][mt][r] + bv[4 + r], S[2 * q + 1][mt][r + 1] + bv[4 + r + 1]};
                        const f32x2_t ya = DIAG == 1 ? a2 : gelu_pw(a2), yb = DIAG == 1 ? b2 : gelu_pw(b2);
                        x[r] = ya.x; x[r + 1] = ya.y; x[4 + r] = yb.x; x[4 + r + 1] = yb.y;
                    }
                    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                    const u32x4_t w = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
                    hb[q][mt] = __builtin_bit_cast(bf16x8_t, w);
                }
            }
            // ---- O^T += W2c . H^T (same one-tile-ahead fragment reads)
            bf16x8_t wf[2][HC / 32];
#pragma unroll
            for (int q = 0; q < HC / 32; ++q) wf[0][q] = frag(w2, 0, q * 4 + g);
#pragma unroll
            for (int ct = 0; ct < (DIAG == 2 ? 0 : CT); ++ct) {
                if (ct + 1 < CT) {
#pragma unroll
                    for (int q = 0; q < HC / 32; ++q) wf[(ct + 1) & 1][q] = frag(w2, ct + 1, q * 4 + g);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < HC / 32; ++q) {
                    O[ct][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct & 1][q], hb[q][0], O[ct][0], 0, 0, 0);
                    O[ct][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct & 1][q], hb[q][1], O[ct][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: + b2, bf16, transposed through the wave's private 2 KiB staging area (8 rows at a time) so that every store
        // is a full 256-byte row
        uint8_t* st = stage_out + wave * 2048;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                if ((li >> 3) == ps) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const float4 b = *reinterpret_cast<const float4*>(sb2 + ct * 16 + 4 * g);
                        const uint2 v = make_uint2(pack_bf2(O[ct][mt][0] + b.x, O[ct][mt][1] + b.y), pack_bf2(O[ct][mt][2] + b.z, O[ct][mt][3] + b.w));
                        const int cc = (ct * 2 + (g >> 1)) ^ (li & 7);           // 16-byte chunk of row li & 7, XOR-swizzled by the row
                        *reinterpret_cast<uint2*>(st + (li & 7) * 256 + cc * 16 + (g & 1) * 8) = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = it * 4 + g;                                  // 0..7
                    const uint4 v = *reinterpret_cast<const uint4*>(st + row * 256 + ((li ^ row) << 4));
                    const int64_t m = tile * MLP_ROWS + wave * 32 + mt * 16 + ps * 8 + row;
                    if (m < p.rows && (DIAG != 5 || v.x == 0x12345678u)) *reinterpret_cast<uint4*>(p.Out + m * p.ldo + li * 8) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// dy = ((dm . W2) * GELU'(y . W1^T + b1)) . W1 in one kernel, the pre-activation RECOMPUTED from y (nothing of the hidden tensor was
// saved): per 128-wide hidden chunk
//     S^T[h, m] = W1c . y^T          G^T[h, m] = W2^T c . dm^T          (A = rows h of W1 / of fc2.weight^T from LDS, B = y / dm fragments)
//     dZ^T      = bf16(G^T * GELU'(S^T + b1))                           packed, per pair of 16-row h tiles, into the B operand of
//     dY^T[c, m] += W1c^T . dZ^T                                         A = the SAME LDS tile of W1 read transposed (ds_read_b64_tr_b16:
//                                                                        4 rows h x 16 columns c per 16-lane group = exactly the four h of
//                                                                        a k-slot quad, so the accumulator order needs no permuted copy)
// Registers: dY 64 + one tile pair of S and G (32) + dZ 32 + y / dm fragments 64 + a three-deep ring of A fragments.
struct MlpB {
    const bf16_t* Y; int64_t ldy;
    const bf16_t* dM; int64_t ldm;
    const bf16_t* W1; const float* b1;   // [4C, C], [4C]
    const bf16_t* W2T;                    // [4C, C] = fc2.weight^T
    bf16_t* dY; int64_t ldo;
    int64_t rows; int ntiles;
};

typedef short s4v_t __attribute__((ext_vector_type(4)));

// MT = 16-row tiles per wave.  MT = 2 (the forward's shape) needs dY 64 + S / G 32 + dZ 32 + y / dm 64 + fragment rings and spills 100+
// registers; MT = 1 halves every one of them (~150 VGPRs) at twice the LDS fragment bytes per MFMA (LDS and MFMA then tie).
template <int C, int MT>
__global__ void __launch_bounds__(512, 2) mlp_bwd_kernel(MlpB p) {
    constexpr int ROWS = MLP_WAVES * 16 * MT;
    static_assert(C == 128, "one LDS row = one 256-byte weight row");
    constexpr int NCH = 4 * C / HC;
    constexpr int KS1 = C / 32;
    constexpr int CT = C / 16;
    constexpr int WBYTES = HC * C * 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];      // 2 x (W1 chunk + W2^T chunk) | 8 x 2 KiB output staging | b1
    uint8_t* stage_out = smem + 4 * WBYTES;
    float* sb1 = reinterpret_cast<float*>(stage_out + MLP_WAVES * 2048);
    for (int i = threadIdx.x; i < 4 * C; i += 512) sb1[i] = p.b1[i];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;

    uint32_t off1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (wave * 4 + j) * 64 + lane;
        const int row = q >> 4, sc = (q & 15) ^ (row & 15);
        off1[j] = (uint32_t)((row * C + sc * 8) * 2);                    // both matrices: rows h, C columns
    }
    auto issue = [&](int buf, int ch) {
        uint8_t* d1 = smem + buf * 2 * WBYTES;
        uint8_t* d2 = d1 + WBYTES;
        const char* s1 = reinterpret_cast<const char*>(p.W1) + (size_t)ch * HC * C * 2;
        const char* s2 = reinterpret_cast<const char*>(p.W2T) + (size_t)ch * HC * C * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s1 + off1[j]),
                                             (__attribute__((address_space(3))) void*)(d1 + (wave * 4 + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s2 + off1[j]),
                                             (__attribute__((address_space(3))) void*)(d2 + (wave * 4 + j) * 1024), 16, 0, 0);
        }
    };
    auto frag = [&](const uint8_t* base, int t, int kc) {
        const int row = t * 16 + li;
        return ld16(base + row * 256 + ((kc ^ (row & 15)) << 4));
    };
    // transposed A fragment of the W1 tile: rows c = 16 ct + (lane & 15), k slots = h 32 q + 4 g + {0..3} and 32 q + 16 + 4 g + {0..3}
    auto trfrag = [&](const uint8_t* base, int q, int ct) {
        const int qq = li >> 2, pp = li & 3;
        const int r0 = 32 * q + 4 * g + qq, r1 = r0 + 16;
        const int ch0 = 2 * ct + (pp >> 1);
        const uint8_t* a0 = base + r0 * 256 + ((ch0 ^ (r0 & 15)) << 4) + (pp & 1) * 8;
        const uint8_t* a1 = base + r1 * 256 + ((ch0 ^ (r1 & 15)) << 4) + (pp & 1) * 8;
        const s4v_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v_t*)a0);
        const s4v_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v_t*)a1);
        bf16x8_t f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };
    auto load_rows = [&](int64_t tile, bf16x8_t (&yf)[MT][KS1], bf16x8_t (&df)[MT][KS1]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int64_t m = tile * ROWS + wave * (16 * MT) + mt * 16 + li;
            m = m < p.rows ? m : p.rows - 1;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                yf[mt][ks] = ld16(p.Y + m * p.ldy + ks * 32 + g * 8);
                df[mt][ks] = ld16(p.dM + m * p.ldm + ks * 32 + g * 8);
            }
        }
    };

    int64_t tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    bf16x8_t yf[MT][KS1], df[MT][KS1];
    load_rows(tile, yf, df);
    issue(0, 0);
    int gch = 0;
    for (; tile < p.ntiles; tile += gridDim.x) {
        f32x4_t D[CT][MT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) D[ct][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int64_t tnext = tile + gridDim.x;
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch, ++gch) {
            const int buf = gch & 1;
            __syncthreads();
            const bool more = ch + 1 < NCH || tnext < p.ntiles;
            if (more) issue(buf ^ 1, (ch + 1) % NCH);
            const uint8_t* w1 = smem + buf * 2 * WBYTES;
            const uint8_t* w2t = w1 + WBYTES;
            bf16x8_t dz[HC / 32][MT];
            // ---- per pair q of h tiles: S and G (32 steps of 4 MFMAs, A fragments two steps ahead), then dZ
            bf16x8_t A1[3], A2[3];
            A1[0] = frag(w1, 0, g); A2[0] = frag(w2t, 0, g);
            A1[1] = frag(w1, 0, 4 + g); A2[1] = frag(w2t, 0, 4 + g);
            f32x4_t S[2][MT], G[2][MT];
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                const int i = (st >> 2) & 1, ks = st & 3, q = st >> 3;
                if (st + 2 < 32) {
                    const int s2 = st + 2;
                    A1[s2 % 3] = frag(w1, s2 >> 2, (s2 & 3) * 4 + g);
                    A2[s2 % 3] = frag(w2t, s2 >> 2, (s2 & 3) * 4 + g);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (ks == 0) { S[i][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; G[i][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
                    S[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[st % 3], yf[mt][ks], S[i][mt], 0, 0, 0);
                    G[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[st % 3], df[mt][ks], G[i][mt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if ((st & 7) == 7) {
                    const float4 ba = *reinterpret_cast<const float4*>(sb1 + ch * HC + 32 * q + 4 * g);
                    const float4 bb = *reinterpret_cast<const float4*>(sb1 + ch * HC + 32 * q + 16 + 4 * g);
                    const float bv[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        float x[8];
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {          // packed fp32 pairs: only the derivative is needed here
                            const f32x2_t d0 = gelu_pw_grad((f32x2_t){S[0][mt][r] + bv[r], S[0][mt][r + 1] + bv[r + 1]});
                            const f32x2_t d1 = gelu_pw_grad((f32x2_t){S[1][mt][r] + bv[4 + r], S[1][mt][r + 1] + bv[4 + r + 1]});
                            x[r] = G[0][mt][r] * d0.x; x[r + 1] = G[0][mt][r + 1] * d0.y;
                            x[4 + r] = G[1][mt][r] * d1.x; x[4 + r + 1] = G[1][mt][r + 1] * d1.y;
                        }
                        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                        const u32x4_t w = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
                        dz[q][mt] = __builtin_bit_cast(bf16x8_t, w);
                        __builtin_amdgcn_sched_barrier(0);       // one row tile's 8 derivative evaluations at a time: hipcc otherwise
                    }                                            // interleaves all 16 and spills
                }
            }
            if (ch == NCH - 1 && tnext < p.ntiles) load_rows(tnext, yf, df);        // the next tile's rows load behind the third product
            // ---- dY^T += W1c^T . dZ^T: transposed fragments of the W1 tile, four steps ahead
            bf16x8_t T[4];
#pragma unroll
            for (int s0 = 0; s0 < 3; ++s0) T[s0] = trfrag(w1, s0 & 3, s0 >> 2);
#pragma unroll
            for (int st = 0; st < CT * 4; ++st) {
                const int ct = st >> 2, q = st & 3;
                if (st + 3 < CT * 4) T[(st + 3) % 4] = trfrag(w1, (st + 3) & 3, (st + 3) >> 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) D[ct][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T[st % 4], dz[q][mt], D[ct][mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: bf16 rows through the wave's 2 KiB staging area, 8 rows at a time, full 256-byte row stores
        uint8_t* st8 = stage_out + wave * 2048;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                if ((li >> 3) == ps) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const uint2 v = make_uint2(pack_bf2(D[ct][mt][0], D[ct][mt][1]), pack_bf2(D[ct][mt][2], D[ct][mt][3]));
                        const int cc = (ct * 2 + (g >> 1)) ^ (li & 7);
                        *reinterpret_cast<uint2*>(st8 + (li & 7) * 256 + cc * 16 + (g & 1) * 8) = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = it * 4 + g;
                    const uint4 v = *reinterpret_cast<const uint4*>(st8 + row * 256 + ((li ^ row) << 4));
                    const int64_t m = tile * ROWS + wave * (16 * MT) + mt * 16 + ps * 8 + row;
                    if (m < p.rows) *reinterpret_cast<uint4*>(p.dY + m * p.ldo + li * 8) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            }
        }
    }
}

std::atomic<uint64_t> mlp_bwd_lds_done{0};
std::atomic<uint64_t> mlp_fwd_lds_done{0};

}  // namespace

// hidden-index permutation of fc2's weight for the fused kernel: position h' = 32 q + 8 g + j holds original column
// 32 q + 4 g + j (j < 4) or 32 q + 16 + 4 g + (j - 4) (j >= 4)
extern "C" int stg_mlp_w2_perm(int hidden, int* perm) {
    STG_CHECK(perm && hidden > 0 && hidden % 32 == 0, -2, "stg_mlp_w2_perm: hidden must be a multiple of 32");
    for (int hp = 0; hp < hidden; ++hp) {
        const int q = hp >> 5, gg = (hp >> 3) & 3, j = hp & 7;
        perm[hp] = 32 * q + (j < 4 ? 4 * gg + j : 16 + 4 * gg + (j - 4));
    }
    return 0;
}

extern "C" int stg_mlp_fused_supported(int C) { return C == 128 ? 1 : 0; }

extern "C" int stg_mlp_fwd(const void* Y, int64_t ldy, const void* W1, const float* b1, const void* W2p, const float* b2,
                           void* Out, int64_t ldo, int64_t rows, int C, void* stream) {
    STG_CHECK(Y && W1 && b1 && W2p && b2 && Out, -1, "stg_mlp_fwd: null pointer");
    STG_CHECK(C == 128, -3, "stg_mlp_fwd: C = %d is not built (128)", C);
    STG_CHECK(rows >= 0 && ldy >= C && ldo >= C && ldy % 8 == 0 && ldo % 8 == 0, -2, "stg_mlp_fwd: bad shape / leading dimensions");
    STG_CHECK((((uintptr_t)Y | (uintptr_t)W1 | (uintptr_t)W2p | (uintptr_t)Out | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0, -2, "stg_mlp_fwd: misaligned pointers");
    if (rows == 0) return 0;
    MlpP p;
    p.Y = (const bf16_t*)Y; p.ldy = ldy; p.W1 = (const bf16_t*)W1; p.b1 = b1; p.W2p = (const bf16_t*)W2p; p.b2 = b2;
    p.Out = (bf16_t*)Out; p.ldo = ldo; p.rows = rows;
    const int64_t nt = (rows + MLP_ROWS - 1) / MLP_ROWS;
    STG_CHECK(nt < (1ll << 31), -2, "stg_mlp_fwd: too many rows");
    p.ntiles = (int)nt;
    const int lds = 4 * HC * 128 * 2 + MLP_WAVES * 2048 + (4 * 128 + 128) * 4;      // 128 KiB ring + 16 KiB staging + biases
    STG_CHECK(stg_reserve_lds(mlp_fwd_kernel<128>, lds, mlp_fwd_lds_done), -101, "stg_mlp_fwd: cannot reserve %d bytes of LDS", lds);
    int grid = nt < 256 ? (int)nt : 256;                               // persistent: one workgroup per CU
#ifdef STG_GEMM_DIAG
    {
        static std::atomic<uint64_t> dd[6];
        const int dg = stg_opt_gemm_dbg.load(std::memory_order_relaxed);
#define STG_MLP_DIAG(D) if (dg == D) { STG_CHECK(stg_reserve_lds(mlp_fwd_kernel<128, D>, lds, dd[D]), -101, "lds"); \
        hipLaunchKernelGGL((mlp_fwd_kernel<128, D>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p); STG_LAUNCH_CHECK(); return 0; }
        STG_MLP_DIAG(1) STG_MLP_DIAG(2) STG_MLP_DIAG(3) STG_MLP_DIAG(4) STG_MLP_DIAG(5)
#undef STG_MLP_DIAG
    }
#endif
    hipLaunchKernelGGL(mlp_fwd_kernel<128>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_mlp_bwd(const void* Y, int64_t ldy, const void* dM, int64_t ldm, const void* W1, const float* b1, const void* W2T,
                           void* dY, int64_t ldo, int64_t rows, int C, void* stream) {
    STG_CHECK(Y && dM && W1 && b1 && W2T && dY, -1, "stg_mlp_bwd: null pointer");
    STG_CHECK(C == 128, -3, "stg_mlp_bwd: C = %d is not built (128)", C);
    STG_CHECK(rows >= 0 && ldy >= C && ldm >= C && ldo >= C && ldy % 8 == 0 && ldm % 8 == 0 && ldo % 8 == 0, -2, "stg_mlp_bwd: bad shape / leading dimensions");
    STG_CHECK((((uintptr_t)Y | (uintptr_t)dM | (uintptr_t)W1 | (uintptr_t)W2T | (uintptr_t)dY | (uintptr_t)b1) & 15) == 0, -2, "stg_mlp_bwd: misaligned pointers");
    if (rows == 0) return 0;
    MlpB p;
    p.Y = (const bf16_t*)Y; p.ldy = ldy; p.dM = (const bf16_t*)dM; p.ldm = ldm; p.W1 = (const bf16_t*)W1; p.b1 = b1; p.W2T = (const bf16_t*)W2T;
    p.dY = (bf16_t*)dY; p.ldo = ldo; p.rows = rows;
    constexpr int BMT = 1;
    const int64_t nt = (rows + MLP_WAVES * 16 * BMT - 1) / (MLP_WAVES * 16 * BMT);
    STG_CHECK(nt < (1ll << 31), -2, "stg_mlp_bwd: too many rows");
    p.ntiles = (int)nt;
    const int lds = 4 * HC * 128 * 2 + MLP_WAVES * 2048 + 4 * 128 * 4;
    STG_CHECK(stg_reserve_lds(mlp_bwd_kernel<128, BMT>, lds, mlp_bwd_lds_done), -101, "stg_mlp_bwd: cannot reserve %d bytes of LDS", lds);
    const int grid = nt < 256 ? (int)nt : 256;
    hipLaunchKernelGGL((mlp_bwd_kernel<128, BMT>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}
