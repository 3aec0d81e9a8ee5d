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
__global__ void bilinear_up2_fwd_kernel(const bf16_t* x, bf16_t* y, int64_t F, int H, int W, int C, int align) {
    const int OH = 2 * H, OW = 2 * W, c8 = C >> 3;
    const int64_t total = F * OH * OW * c8;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int pc = (int)(id % c8);
        int64_t r = id / c8;
        const int ow = (int)(r % OW);
        const int oh = (int)((r / OW) % OH);
        const int64_t f = r / ((int64_t)OW * OH);
        int h0, h1, w0, w1i; float lh, lw;
        src_index(oh, H, OH, align, h0, h1, lh);
        src_index(ow, W, OW, align, w0, w1i, lw);
        const bf16_t* base = x + f * (int64_t)H * W * C + 8 * pc;
        const u16x8 a = *reinterpret_cast<const u16x8*>(base + ((int64_t)h0 * W + w0) * C), b = *reinterpret_cast<const u16x8*>(base + ((int64_t)h0 * W + w1i) * C);
        const u16x8 c = *reinterpret_cast<const u16x8*>(base + ((int64_t)h1 * W + w0) * C), e = *reinterpret_cast<const u16x8*>(base + ((int64_t)h1 * W + w1i) * C);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float top = bf2f(a.v[j]) * (1.f - lw) + bf2f(b.v[j]) * lw;
            const float bot = bf2f(c.v[j]) * (1.f - lw) + bf2f(e.v[j]) * lw;
            o.v[j] = f2bf(top * (1.f - lh) + bot * lh);
        }
        *reinterpret_cast<u16x8*>(y + r * C + 8 * pc) = o;
    }
}
// gather form of the adjoint: input pixel (h, w) collects from the (at most 6 x 6) outputs whose two source rows / columns
// include it -- no atomics.  Candidates: outputs 2h-3 .. 2h+3 (covers both align modes for an exact x2 resize).
__global__ void bilinear_up2_bwd_kernel(const bf16_t* dy, bf16_t* dx, int64_t F, int H, int W, int C, int align) {
    const int OH = 2 * H, OW = 2 * W, c8 = C >> 3;
    const int64_t total = F * H * W * c8;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int pc = (int)(id % c8);
        int64_t r = id / c8;
        const int w = (int)(r % W);
        const int h = (int)((r / W) % H);
        const int64_t f = r / ((int64_t)W * H);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const bf16_t* base = dy + f * (int64_t)OH * OW * C + 8 * pc;
        for (int oh = 2 * h - 3; oh <= 2 * h + 3; ++oh) {
            if (oh < 0 || oh >= OH) continue;
            int h0, h1, t0, t1; float lh, lw;
            src_index(oh, H, OH, align, h0, h1, lh);
            const float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
            if (wh == 0.f) continue;
            for (int ow = 2 * w - 3; ow <= 2 * w + 3; ++ow) {
                if (ow < 0 || ow >= OW) continue;
                src_index(ow, W, OW, align, t0, t1, lw);
                const float ww = (t0 == w ? 1.f - lw : 0.f) + (t1 == w ? lw : 0.f);
                if (ww == 0.f) continue;
                const u16x8 g = *reinterpret_cast<const u16x8*>(base + ((int64_t)oh * OW + ow) * C);
                const float wt = wh * ww;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += wt * bf2f(g.v[j]);
            }
        }
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = f2bf(acc[j]);
        *reinterpret_cast<u16x8*>(dx + r * C + 8 * pc) = o;
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
    hipLaunchKernelGGL(bilinear_up2_fwd_kernel, dim3(grid_for(F * 4 * H * W * (C / 8), 256)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y,
                       F, H, W, C, align_corners ? 1 : 0);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bilinear_up2_bwd(const void* dy, void* dx, int64_t F, int H, int W, int C, int align_corners, void* stream) {
    STG_CHECK(dy && dx, -1, "stg_bilinear_up2_bwd: null pointer");
    STG_CHECK(F >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, -2, "stg_bilinear_up2_bwd: bad shape (C % 8 == 0)");
    if (F == 0) return 0;
    hipLaunchKernelGGL(bilinear_up2_bwd_kernel, dim3(grid_for(F * H * W * (C / 8), 256)), dim3(256), 0, ST, (const bf16_t*)dy, (bf16_t*)dx,
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
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(R * C, 256)), dim3(256), 0, ST, (const bf16_t*)x, mean, rstd, gamma, beta, (bf16_t*)y, R, C);
    STG_LAUNCH_CHECK();
    return 0;
}
extern "C" int stg_bn_bwd(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma, const float* sums,
                          void* dx, int64_t R, int C, void* stream) {
    STG_CHECK(x && dy && mean && rstd && gamma && dx, -1, "stg_bn_bwd: null pointer");
    STG_CHECK(R >= 0 && C > 0, -2, "stg_bn_bwd: bad shape");
    if (R == 0) return 0;
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(grid_for(R * C, 256)), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, gamma, sums,
                       (bf16_t*)dx, R, C);
    STG_LAUNCH_CHECK();
    return 0;
}
