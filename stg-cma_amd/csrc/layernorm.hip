// LayerNorm forward / backward over the last dim (see include/stgcma.h).  HBM-bound: one pass over x, 16-byte
// (bf16x8) or 2x16-byte (fp32) loads per lane, the row lives in registers between the statistics and the
// normalisation, reductions are wave shuffles over LPR lanes (LPR lanes per row; 64/LPR rows per wave).
// gather4 folds PatchMerging's 2x2 strided gather + concat into the addressing (Swin_AVE.py:967-972).
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr int MAXCH = 8;  // 16-byte chunks per lane  => C <= 64 * 8 * 8 = 4096

struct LnParams {
    const void* x; int64_t ldx; int x_f32;
    const float* gamma; const float* beta; float eps;
    void* y; int64_t ldy; int y_f32;
    float* mean; float* rstd;
    int64_t rows; int C;   // C = logical row width (4*Csrc when gather4)
    int gather4; int Csrc; int H, W;
    // backward
    const bf16_t* dy; int64_t lddy;
    const bf16_t* add_to; int64_t ldadd;
    bf16_t* dx; int64_t lddx;
    float* dgamma; float* dbeta;
    int xhat;          // backward: `x` holds the NORMALISED row (bf16 x_hat, saved by the forward in place of y), gamma == 1: no mean
};

// source element offset (in elements) of logical (row, col8) chunk start
__device__ __forceinline__ int64_t src_off(const LnParams& p, int64_t row, int col, int64_t ld) {
    if (!p.gather4) return row * ld + col;
    const int seg = col / p.Csrc, cc = col - seg * p.Csrc;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int64_t f = row / (H2 * W2);
    const int rem = (int)(row - f * (H2 * W2));
    const int i = rem / W2, j = rem - i * W2;
    // segments: 0:(2i,2j) 1:(2i+1,2j) 2:(2i,2j+1) 3:(2i+1,2j+1)
    const int si = 2 * i + (seg & 1), sj = 2 * j + (seg >> 1);
    return (f * (int64_t)(p.H * p.W) + (int64_t)si * p.W + sj) * ld + cc;
}

__device__ __forceinline__ void load8(const LnParams& p, const void* base, int f32, int64_t off, float* v) {
    if (f32) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const u16x8 a = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16_t*>(base) + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf2f(a.v[j]);
    }
}
__device__ __forceinline__ void store8(bf16_t* ptr, const float* v) {
    uint4 o;
    o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(ptr) = o;
}

template <int LPR>
__global__ void __launch_bounds__(256) ln_fwd_kernel(LnParams p) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub;
    const bool valid = row < p.rows;
    const int nch = p.C / 8;
    float v[MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = sl + c * LPR;
        if (valid && ch < nch) {
            load8(p, p.x, p.x_f32, src_off(p, row, ch * 8, p.ldx), v[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    s = wave_sum<LPR>(s);
    const float mu = s / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = sl + c * LPR;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mu; q += d * d; }
        }
    }
    q = wave_sum<LPR>(q);
    const float rs = rsqrtf(q / (float)p.C + p.eps);
    if (!valid) return;
    if (sl == 0) {
        if (p.mean) p.mean[row] = mu;
        if (p.rstd) p.rstd[row] = rs;
    }
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = sl + c * LPR;
        if (ch < nch) {
            float o[8];
            const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + ch * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + ch * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(p.beta + ch * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(p.beta + ch * 8 + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mu) * rs * g[j] + b[j];
            if (p.y_f32) {
                float* yp = reinterpret_cast<float*>(p.y) + row * p.ldy + ch * 8;
                *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(yp + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
                store8(reinterpret_cast<bf16_t*>(p.y) + row * p.ldy + ch * 8, o);
            }
        }
    }
}

// Shape-fitted forward (round 5b): NC chunks per lane with EVERY lane of a row busy (C = 8 * LPR * NC), RU row sets per wave -- NC * RU = 3-4 independent
// 32-byte loads per lane and 24-32 data registers instead of the generic kernel's 64 (MAXCH = 8 whatever C is: at C <= 768 a lane had one or two loads in
// flight and the kernel ran at 3.4-4.3 TB/s where the 1 024 / 1 536-wide rows, two or three chunks per lane, reached 5.2-5.6).
template <int LPR, int NC, int RU>
__global__ void __launch_bounds__(256) ln_fwd_fit_kernel(LnParams p) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    float v[RU][NC][8];
    int64_t row[RU];
    bool valid[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        row[u] = (((int64_t)blockIdx.x * 4 + wave) * RU + u) * RPW + sub;
        valid[u] = row[u] < p.rows;
        const int64_t rc = valid[u] ? row[u] : p.rows - 1;               // (a clamped row keeps every load unconditional; nothing of it is stored)
#pragma unroll
        for (int c = 0; c < NC; ++c) load8(p, p.x, p.x_f32, src_off(p, rc, (sl + c * LPR) * 8, p.ldx), v[u][c]);
    }
    float g[NC][8], b[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = sl + c * LPR;
        const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(p.gamma + ch * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.beta + ch * 8), b1 = *reinterpret_cast<const float4*>(p.beta + ch * 8 + 4);
        g[c][0] = g0.x; g[c][1] = g0.y; g[c][2] = g0.z; g[c][3] = g0.w; g[c][4] = g1.x; g[c][5] = g1.y; g[c][6] = g1.z; g[c][7] = g1.w;
        b[c][0] = b0.x; b[c][1] = b0.y; b[c][2] = b0.z; b[c][3] = b0.w; b[c][4] = b1.x; b[c][5] = b1.y; b[c][6] = b1.z; b[c][7] = b1.w;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[u][c][j];
        s = wave_sum<LPR>(s);
        const float mu = s / (float)p.C;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[u][c][j] - mu; q += d * d; }
        q = wave_sum<LPR>(q);
        const float rs = rsqrtf(q / (float)p.C + p.eps);
        if (!valid[u]) continue;
        if (sl == 0) {
            if (p.mean) p.mean[row[u]] = mu;
            if (p.rstd) p.rstd[row[u]] = rs;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = sl + c * LPR;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[u][c][j] - mu) * rs * g[c][j] + b[c][j];
            if (p.y_f32) {
                float* yp = reinterpret_cast<float*>(p.y) + row[u] * p.ldy + ch * 8;
                *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(yp + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
                store8(reinterpret_cast<bf16_t*>(p.y) + row[u] * p.ldy + ch * 8, o);
            }
        }
    }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))  [+ add_to]
template <int LPR>
__global__ void __launch_bounds__(256) ln_bwd_kernel(LnParams p) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub;
    const bool valid = row < p.rows;
    const int nch = p.C / 8;
    float xh[MAXCH][8], gd[MAXCH][8];
    const float mu = (valid && !p.xhat) ? p.mean[row] : 0.f;
    const float rs = valid ? p.rstd[row] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = sl + c * LPR;
        if (valid && ch < nch) {
            float xv[8], dv[8];
            load8(p, p.x, p.x_f32, src_off(p, row, ch * 8, p.ldx), xv);
            load8(p, p.dy, 0, row * p.lddy + ch * 8, dv);
            float g[8];
            if (p.xhat) {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = 1.0f;
            } else {
                const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + ch * 8);
                const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + ch * 8 + 4);
                g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[c][j] = p.xhat ? xv[j] : (xv[j] - mu) * rs;
                gd[c][j] = g[j] * dv[j];
                s1 += gd[c][j];
                s2 += gd[c][j] * xh[c][j];
            }
            if (p.dgamma) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    atomicAdd(p.dgamma + ch * 8 + j, dv[j] * xh[c][j]);
                    atomicAdd(p.dbeta + ch * 8 + j, dv[j]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { xh[c][j] = 0.f; gd[c][j] = 0.f; }
        }
    }
    s1 = wave_sum<LPR>(s1) / (float)p.C;
    s2 = wave_sum<LPR>(s2) / (float)p.C;
    if (!valid) return;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = sl + c * LPR;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rs * (gd[c][j] - s1 - xh[c][j] * s2);
            // dx addressed like x (scatter back through the 2x2 gather when gather4)
            const int64_t off = src_off(p, row, ch * 8, p.lddx);
            if (p.add_to) {
                float av[8];
                load8(p, p.add_to, 0, src_off(p, row, ch * 8, p.ldadd), av);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += av[j];
            }
            store8(p.dx + off, o);
        }
    }
}

// The backward in the same shape-fitted form (frozen LayerNorms: no parameter gradients): NC chunks of x (or x_hat) and of dy per lane,
// 16 NC data registers instead of 128.
template <int LPR, int NC>
__global__ void __launch_bounds__(256) ln_bwd_fit_kernel(LnParams p) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub;
    const bool valid = row < p.rows;
    const int64_t rc = valid ? row : p.rows - 1;
    float xh[NC][8], gd[NC][8];
    const float mu = p.xhat ? 0.f : p.mean[rc];
    const float rs = p.rstd[rc];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = sl + c * LPR;
        float xv[8], dv[8];
        load8(p, p.x, p.x_f32, src_off(p, rc, ch * 8, p.ldx), xv);
        load8(p, p.dy, 0, rc * p.lddy + ch * 8, dv);
        float g[8];
        if (p.xhat) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = 1.0f;
        } else {
            const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + ch * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + ch * 8 + 4);
            g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xh[c][j] = p.xhat ? xv[j] : (xv[j] - mu) * rs;
            gd[c][j] = g[j] * dv[j];
            s1 += gd[c][j];
            s2 += gd[c][j] * xh[c][j];
        }
    }
    s1 = wave_sum<LPR>(s1) / (float)p.C;
    s2 = wave_sum<LPR>(s2) / (float)p.C;
    if (!valid) return;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = sl + c * LPR;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (gd[c][j] - s1 - xh[c][j] * s2);
        if (p.add_to) {
            float av[8];
            load8(p, p.add_to, 0, src_off(p, row, ch * 8, p.ldadd), av);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += av[j];
        }
        store8(p.dx + src_off(p, row, ch * 8, p.lddx), o);           // dx addressed like x (scatter back through the 2x2 gather when gather4)
    }
}

template <int LPR>
int launch_ln(bool bwd, const LnParams& p, hipStream_t st) {
    constexpr int RPW = 64 / LPR;
    const int64_t rows_per_block = 4 * RPW;
    const int64_t nblk = (p.rows + rows_per_block - 1) / rows_per_block;
    if (bwd) hipLaunchKernelGGL(ln_bwd_kernel<LPR>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(ln_fwd_kernel<LPR>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    STG_LAUNCH_CHECK();
    return 0;
}

template <int LPR, int NC>
int launch_ln_fit(bool bwd, const LnParams& p, hipStream_t st) {
    constexpr int RU = NC == 2 ? 2 : 1;
    if (bwd) {
        const int64_t rows_per_block = 4 * (64 / LPR);
        hipLaunchKernelGGL((ln_bwd_fit_kernel<LPR, NC>), dim3((unsigned)((p.rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st, p);
    } else {
        const int64_t rows_per_block = 4 * (64 / LPR) * RU;
        hipLaunchKernelGGL((ln_fwd_fit_kernel<LPR, NC, RU>), dim3((unsigned)((p.rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st, p);
    }
    STG_LAUNCH_CHECK();
    return 0;
}

int dispatch_ln(bool bwd, const LnParams& p, hipStream_t st) {
    const int nch = p.C / 8;
    if (!(bwd && p.dgamma)) {   // (the fitted backward: frozen LayerNorms only)
        // the widths of the four backbones: C = 8 * LPR * NC exactly
        switch (nch) {
            case 16: return launch_ln_fit<8, 2>(bwd, p, st);       // 128
            case 24: return launch_ln_fit<8, 3>(bwd, p, st);       // 192
            case 32: return launch_ln_fit<16, 2>(bwd, p, st);      // 256
            case 48: return launch_ln_fit<16, 3>(bwd, p, st);      // 384
            case 64: return launch_ln_fit<32, 2>(bwd, p, st);      // 512
            case 96: return launch_ln_fit<32, 3>(bwd, p, st);      // 768
            case 128: return launch_ln_fit<64, 2>(bwd, p, st);     // 1024
            case 192: return launch_ln_fit<64, 3>(bwd, p, st);     // 1536
            case 256: return launch_ln_fit<64, 4>(bwd, p, st);     // 2048
            default: break;
        }
    }
    if (nch <= 16) return launch_ln<16>(bwd, p, st);
    if (nch <= 32) return launch_ln<32>(bwd, p, st);
    return launch_ln<64>(bwd, p, st);
}

int check_common(const char* who, int64_t rows, int C, int gather4, int H, int W, int64_t ldx, int x_dtype) {
    STG_CHECK(rows >= 0 && C > 0, -2, "%s: bad shape", who);
    STG_CHECK(C % 8 == 0 && C <= 64 * 8 * MAXCH, -2, "%s: C=%d must be a multiple of 8 and <= %d", who, C, 64 * 8 * MAXCH);
    STG_CHECK(x_dtype == STG_BF16 || x_dtype == STG_F32, -3, "%s: unsupported x dtype", who);
    STG_CHECK(ldx % 8 == 0, -2, "%s: ldx must be a multiple of 8", who);
    if (gather4) {
        STG_CHECK(C % 32 == 0, -2, "%s: gather4 needs C %% 32 == 0", who);
        STG_CHECK(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, -2, "%s: x size (%d*%d) are not even", who, H, W);
        STG_CHECK(rows % ((H / 2) * (W / 2)) == 0, -2, "%s: rows not a multiple of H/2*W/2", who);
    }
    return 0;
}

}  // namespace

extern "C" int stg_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, float eps,
                                 void* y, int y_dtype, int64_t ldy, float* mean, float* rstd, int64_t rows, int C, int gather4,
                                 int H, int W, void* stream) {
    STG_CHECK(x && gamma && beta && y, -1, "stg_layernorm_fwd: null pointer");
    int rc = check_common("stg_layernorm_fwd", rows, C, gather4, H, W, ldx, x_dtype);
    if (rc) return rc;
    STG_CHECK(ldy % 8 == 0 && ldy >= C, -2, "stg_layernorm_fwd: bad ldy");
    if (rows == 0) return 0;
    LnParams p = {};
    p.x = x; p.ldx = ldx; p.x_f32 = (x_dtype == STG_F32);
    p.gamma = gamma; p.beta = beta; p.eps = eps;
    STG_CHECK(y_dtype == STG_BF16 || y_dtype == STG_F32, -3, "stg_layernorm_fwd: unsupported y dtype");
    p.y = y; p.ldy = ldy; p.y_f32 = (y_dtype == STG_F32); p.mean = mean; p.rstd = rstd;
    p.rows = rows; p.C = C; p.gather4 = gather4; p.Csrc = gather4 ? C / 4 : C; p.H = H; p.W = W;
    return dispatch_ln(false, p, (hipStream_t)stream);
}

extern "C" int stg_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int x_dtype, int64_t ldx,
                                 const float* gamma, const float* mean, const float* rstd, const void* add_to,
                                 int64_t ldadd, void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int C,
                                 int gather4, int H, int W, void* stream) {
    STG_CHECK(dy && x && gamma && mean && rstd && dx, -1, "stg_layernorm_bwd: null pointer");
    int rc = check_common("stg_layernorm_bwd", rows, C, gather4, H, W, ldx, x_dtype);
    if (rc) return rc;
    STG_CHECK(lddy % 8 == 0 && lddx % 8 == 0 && (add_to == nullptr || ldadd % 8 == 0), -2, "stg_layernorm_bwd: bad ld");
    STG_CHECK((dgamma == nullptr) == (dbeta == nullptr), -1, "stg_layernorm_bwd: dgamma/dbeta must come together");
    if (rows == 0) return 0;
    LnParams p = {};
    p.x = x; p.ldx = ldx; p.x_f32 = (x_dtype == STG_F32);
    p.gamma = gamma; p.mean = (float*)mean; p.rstd = (float*)rstd;
    p.rows = rows; p.C = C; p.gather4 = gather4; p.Csrc = gather4 ? C / 4 : C; p.H = H; p.W = W;
    p.dy = (const bf16_t*)dy; p.lddy = lddy; p.add_to = (const bf16_t*)add_to; p.ldadd = ldadd;
    p.dx = (bf16_t*)dx; p.lddx = lddx; p.dgamma = dgamma; p.dbeta = dbeta;
    return dispatch_ln(true, p, (hipStream_t)stream);
}

// The same backward from the NORMALISED row: x_hat [rows, C] bf16 is what the forward saved (it feeds the frozen GEMM whose weight has
// gamma / beta folded in, so the affine never ran), gamma == 1, no mean -- the fp32 residual row is not re-read (4C -> 2C bytes per row).
extern "C" int stg_layernorm_bwd_xhat(const void* dy, int64_t lddy, const void* xhat, int64_t ldx, const float* rstd, const void* add_to,
                                      int64_t ldadd, void* dx, int64_t lddx, int64_t rows, int C, void* stream) {
    STG_CHECK(dy && xhat && rstd && dx, -1, "stg_layernorm_bwd_xhat: null pointer");
    int rc = check_common("stg_layernorm_bwd_xhat", rows, C, 0, 0, 0, ldx, STG_BF16);
    if (rc) return rc;
    STG_CHECK(lddy % 8 == 0 && lddx % 8 == 0 && (add_to == nullptr || ldadd % 8 == 0), -2, "stg_layernorm_bwd_xhat: bad ld");
    if (rows == 0) return 0;
    LnParams p = {};
    p.x = xhat; p.ldx = ldx; p.x_f32 = 0; p.xhat = 1;
    p.rstd = (float*)rstd;
    p.rows = rows; p.C = C; p.gather4 = 0; p.Csrc = C;
    p.dy = (const bf16_t*)dy; p.lddy = lddy; p.add_to = (const bf16_t*)add_to; p.ldadd = ldadd;
    p.dx = (bf16_t*)dx; p.lddx = lddx;
    return dispatch_ln(true, p, (hipStream_t)stream);
}
