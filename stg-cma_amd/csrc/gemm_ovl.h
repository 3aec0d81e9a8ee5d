// Internal interface of csrc/gemm_ovl.hip (the overlapped-epilogue GEMM, round 4): called by stg_gemm_nt's dispatch in gemm.hip.
#pragma once
#include <stdint.h>
#include "common.h"

struct GemmOvlParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    bf16_t* C; int64_t ldc;
    const float* bias;                    // may be null
    uint8_t* d8out; int64_t ldp;          // OV_GELU8 / OV_QGELU8: saved derivative, byte code (STG_U8_LIN), byte leading dimension
    const uint8_t* d8src; int64_t ldd;    // OV_DSRC8: derivative source
    int64_t M; int N; int K;
    int nbm, nbn, ntl;                    // 256-row panels, 128-column tiles, consecutive column tiles per workgroup (divides nbn)
};
enum { OV_PLAIN = 0, OV_GELU8 = 1, OV_DSRC8 = 2 };

// true when (variant, shape) is served by the kernel: M % 256 == 0, N % 128 == 0, K == 512
bool stg_gemm_ovl_supported(int variant, int64_t M, int N, int K);
// picks ntl, launches; returns 0 or a negative error code (stg_last_error set)
int stg_gemm_ovl_launch(int variant, GemmOvlParams p, void* stream);
