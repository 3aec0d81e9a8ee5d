// Block-scaled e4m3 (OCP "MX") quantisation for the fp8 frozen-weight GEMM path (include/stgcma.h: stg_quant_fp8_mx; BASELINE
// config 5 -- no reference counterpart: AVQA/model/Swin_AVQAModel_V1.py runs its qkv / proj / fc1 / fc2 / reduction Linears in
// fp32 / fp16 autocast).  HBM-bound byte work: one 16-byte bf16 load and one 8-byte store per thread, the four threads of a
// 32-wide k-block agree on the block's exponent with two shuffles.
//
// Scale table layout (probed on the device, tools/probe/mx_probe.hip): in v_mfma_scale_f32_16x16x128_f8f6f4 lane 16 g + i holds
// row i's bytes k = 16 g .. 16 g + 15 and k = 64 + 16 g .. 64 + 16 g + 15, and supplies the E8M0 scale of row i's k-block g (of 32)
// in the byte OPSEL picks.  A 64-row wave tile is four 16-row MFMA tiles, so the table stores, per (64-row group, k-block), 16
// dwords: dword i = the four scales of rows i, 16 + i, 32 + i, 48 + i -- one coalesced dword load per lane and k-tile.
#include "common.h"
#include "../../include/stgcma.h"

namespace {

struct QP {
    const bf16_t* X; int64_t ldx; int64_t rows; int K;
    uint8_t* Q; int64_t ldq; uint8_t* S; int KB; int64_t rows_pad; int cpr;
};

__global__ void __launch_bounds__(256) quant_fp8_mx_kernel(QP p) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / p.cpr;
    const int c = (int)(idx - row * p.cpr);                       // 8-element chunk of the padded row
    if (row >= p.rows_pad) return;                                 // whole 4-lane groups leave together (cpr % 16 == 0)
    const bool live = row < p.rows && c * 8 < p.K;
    float v[8];
    uint4 raw = make_uint4(0u, 0u, 0u, 0u);
    if (live) raw = *reinterpret_cast<const uint4*>(p.X + row * p.ldx + c * 8);
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(w[j] << 16);
        v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        amax = fmaxf(amax, fmaxf(fabsf(v[2 * j]), fabsf(v[2 * j + 1])));
    }
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    // e = ceil(log2(amax / 448)) + 127 from the float's own exponent field: x * 2^(127 - e) then lies in [-448, 448]
    int e = 127;
    if (amax > 0.f && amax < 3.0e38f) {
        const uint32_t b = __float_as_uint(amax * (1.0f / 448.0f));
        e = (int)((b >> 23) & 0xffu) + ((b & 0x7fffffu) ? 1 : 0);
        e = e < 1 ? 1 : (e > 254 ? 254 : e);
    }
    if (row < p.rows) {
        const float inv = __uint_as_float((uint32_t)(254 - e) << 23);       // 2^(127 - e)
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
        *reinterpret_cast<uint2*>(p.Q + row * p.ldq + c * 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
    }
    if ((c & 3) == 0) {
        const int b = c >> 2;
        p.S[((row >> 6) * p.KB + b) * 64 + (row & 15) * 4 + ((row & 63) >> 4)] = (uint8_t)(row < p.rows ? e : 127);
    }
}

}  // namespace

extern "C" int64_t stg_quant_fp8_scale_bytes(int64_t rows, int K) {
    if (rows < 0 || K <= 0) return 0;
    const int64_t rp = (rows + 63) / 64 * 64;
    const int64_t kb = ((int64_t)K + 127) / 128 * 4;
    return rp * kb;
}

extern "C" int stg_quant_fp8_mx(const void* X, int64_t ldx, int64_t rows, int K, void* Q, int64_t ldq, void* S, void* stream) {
    STG_CHECK(X && Q && S, -1, "stg_quant_fp8_mx: null pointer");
    STG_CHECK(rows >= 0 && K > 0 && K % 8 == 0, -2, "stg_quant_fp8_mx: bad shape rows=%lld K=%d (K %% 8 == 0)", (long long)rows, K);
    const int64_t kp = ((int64_t)K + 127) / 128 * 128;
    STG_CHECK(ldq == kp, -2, "stg_quant_fp8_mx: ldq must be K rounded up to 128 (%lld), got %lld", (long long)kp, (long long)ldq);
    STG_CHECK(ldx >= K && ldx % 8 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Q & 15) == 0 && ((uintptr_t)S & 3) == 0, -2,
              "stg_quant_fp8_mx: misaligned operands");
    if (rows == 0) return 0;
    QP p;
    p.X = (const bf16_t*)X; p.ldx = ldx; p.rows = rows; p.K = K; p.Q = (uint8_t*)Q; p.ldq = ldq; p.S = (uint8_t*)S;
    p.KB = (int)(kp / 32); p.rows_pad = (rows + 63) / 64 * 64; p.cpr = (int)(kp / 8);
    const int64_t total = p.rows_pad * p.cpr;
    const int64_t nblk = (total + 255) / 256;
    STG_CHECK(nblk < (1ll << 31), -2, "stg_quant_fp8_mx: grid too large");
    hipLaunchKernelGGL(quant_fp8_mx_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}
