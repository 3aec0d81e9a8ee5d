/*
 * libstgcma_hip.so -- C ABI of the MI355X (gfx950) kernels behind the STG-CMA hot path.
 *
 * The reference (kaiw7/STG-CMA) has no native layer: its "operator API" is the nn.Module surface of
 * AVE/model/Swin_AVE.py, AVE/model/CLIP_AVE.py, AVQA/model/Swin_AVQAModel_V1.py, AVS/model/Swin_AVSModel*.py,
 * which bottoms out in chains of ATen ops.  Each entry point below replaces one such chain; the chain it
 * replaces is cited as reference file:line (relative to the reference root).  The Python modules in
 * stg-cma_amd/model/ bind these through ctypes (stg-cma_amd/_lib.py); INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (stg_last_error() holds the message, thread-local);
 *   - the caller owns all memory (device pointers from the PyTorch allocator), the library never allocates or
 *     frees => re-entrant across threads / devices / streams.  Process state is limited to (a) the per-device
 *     "dynamic LDS size reserved" bits of the large-LDS kernels and (b) the dispatch options of stg_set_option,
 *     which default to the product configuration and are never read from the environment;
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued, never synchronised;
 *   - activations are bf16 (uint16 storage) row-major with explicit leading dimensions (in elements);
 *     statistics, biases, trainable-parameter gradients and logits are fp32;
 *   - "rows" are tokens of the fused audio+video tensor X[2, B*T, N, C] (modality 0 = video, 1 = audio).
 *
 * Not exported, deliberately: a collective.  The data-parallel exchange of the path -- the average of the trainable gradients,
 * reference AVE/traintest_adapt_ave29.py:32-35 (nn.DataParallel's gather / reduce) -- is ONE all-reduce of the flat fp32 gradient
 * arena per step plus one bucket for the task heads (stg-cma_amd/ddp.py), issued through torch.distributed: the process group owns the
 * RCCL communicator, its bootstrap and its stream ordering against the autograd engine, and a `stg_allreduce_bucket` entry would
 * have to duplicate all three for a single ncclAllReduce call.  SURVEY.md section 8(b) lists that symbol in its minimum export set;
 * this library leaves it out on purpose.
 */
#ifndef STGCMA_H
#define STGCMA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define STG_VERSION 220

enum stg_act { STG_ACT_NONE_ = 0, STG_ACT_GELU_ = 1, STG_ACT_QUICKGELU_ = 2 };
enum stg_dtype { STG_F32 = 0, STG_BF16 = 1, STG_FP8_MX = 2, STG_U8_LIN = 3 };

int stg_version(void);
const char* stg_last_error(void);
/* Dispatch options for A/B measurements (tools/, tests) -- eight names since ABI 219: "gemm_8ph" (8-phase GEMM kernels: 0 off / 1 auto / 2 every legal
 * shape / 3 long-K only), "gemm_8phm" (their multi-tile form: 0 off / 1 auto / n >= 2: at most n column tiles per workgroup, any K), "gemm_nx" (their
 * N % 64 == 0 / K % 64 == 0 forms: 0 off / 1 the classes measured faster / 2 every legal shape), "gemm_d8m" (fc2 dgrad's byte-derivative epilogue on
 * them: 0 / 1 K <= 512 / 2 K <= 1024), "xattn" (0: frame-global cross-modal attention on the generic kernels), "wgrad_plan" (row splits of the
 * workspace weight-gradient kernels: 0 round-1 rule / 1 / 2 per-launch chooser), "upln_cap" (workgroups of the wide join kernels), "gemm_dbg"
 * (diagnostics build only).  Returns -2 for an unknown name.  The product never calls it. */
int stg_set_option(const char* name, int value);

/* ---------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epi( A[M,K] . W[N,K]^T )      (bf16 MFMA 16x16x32, fp32 accumulate)
 * replaces F.linear / addmm / mm at: Swin_AVE.py:15-16,20-22 (adapters), :119-126 (Mlp), :238,274 (qkv/proj),
 * :973-979 (PatchMerging.reduction), :1319-1322 (mlp_head); CLIP_AVE.py:56-60,106-108 (c_fc/c_proj/in_proj/out_proj).
 * The same entry serves dgrad (W := pre-transposed weight) -- frozen weights never get a wgrad.
 *   t = acc*alpha + bias[n]
 *   if dact:     dact[m,n] = bf16(act'(t))                   (saved for the activation's backward; needs act != NONE)
 *   t = act(t)
 *   if dact_src: t *= dact_src[m,n]                          (backward through GELU/QuickGELU: the saved derivative)
 *   if row_scale: t *= row_scale[(m / rs_outer) * rs_inner + (m % rs_inner)]   (DropPath mask, Swin_AVE.py:709,715)
 *   if res1: t += res1[m,n];  if res2: t += res2[m,n]        (residual adds, Swin_AVE.py:780-787,810-811)
 *   C[m,n] = (c_dtype == STG_BF16) ? bf16(t) : t
 * Requirements: K % 8 == 0, lda/ldw % 8 == 0, 16-byte aligned A/W.
 */
typedef struct {
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    void* C; int64_t ldc; int c_dtype;
    const float* bias;
    float alpha;
    int act;
    void* dact; int64_t ldp;
    const void* dact_src; int64_t ldd;
    const float* row_scale; int64_t rs_outer; int64_t rs_inner;
    const void* res1; int64_t ldr1; int res1_dtype;   /* STG_BF16 (branch tensors) or STG_F32 (the residual stream) */
    const void* res2; int64_t ldr2; int res2_dtype;
    int64_t M; int N; int K;
    /* Implicit 3x3 convolution (conv_H > 0; AVS decoder, Swin_AVSModel_Base.py:14-130): A is the channels-last feature map
     * [F*conv_H*conv_W, conv_C] (lda >= conv_C) and the GEMM runs over its im2col image without ever forming it: logical
     * A'[m, tap*conv_C + c] = A[pixel(m) shifted by ((tap/3 - 1)*conv_d, (tap%3 - 1)*conv_d), c], zero outside the frame
     * (padding == dilation), K = 9*conv_C, W = [N, (kh, kw, c)].  conv_C % 64 == 0 or conv_C in {8, 16, 32}, M % (conv_H*conv_W) == 0;
     * conv_zero: >= 16 zero bytes, 16-byte aligned (the source of padded taps; also, when given with K % 64 != 0 and K > 64, of the
     * k tail of the LDS-DMA kernel, which then replaces the register-staged one). */
    int conv_H; int conv_W; int conv_d; int conv_C;
    const void* conv_zero;
    /* Batched mode (batch > 1; TPAVI's per-clip 128 x 128 mixing matrices, TPAVI.py:113-139): problem b = 0..batch-1 uses
     * A + b*a_bstride, W + b*w_bstride, C + b*c_bstride (elements) with the same M, N, K; bias is shared; no dact / dact_src /
     * residuals / row_scale / convolution in this mode. */
    int batch; int64_t a_bstride; int64_t w_bstride; int64_t c_bstride;
    /* Operand dtype (BASELINE config 5: the fp8 frozen-weight path of the Swin-L AVQA model; no reference counterpart --
     * AVQA/model/Swin_AVQAModel_V1.py qkv / proj / fc1 / fc2 / reduction Linears run in fp32 / fp16 autocast there).
     * ab_dtype == STG_BF16 (or 0): A and W are bf16 as above.  ab_dtype == STG_FP8_MX: A and W are OCP e4m3 bytes, row-major with
     * leading dimensions lda / ldw in BYTES (multiples of 128, rows zero-padded to K rounded up to 128), each 32-wide k-block of a
     * row scaled by an E8M0 exponent (value = 2^(e - 127) * e4m3): a_scale / w_scale are the packed scale tables stg_quant_fp8_mx
     * writes.  The MFMA is v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulate); every epilogue option above applies unchanged.
     * Not combinable with conv_H > 0 or batch > 1. */
    int ab_dtype;
    const void* a_scale; const void* w_scale;
    /* dact_dtype: storage of the saved activation derivative, both as output (`dact`) and as input (`dact_src`): STG_BF16 (or 0), or
     * STG_U8_LIN -- one byte per element, code = round((act'(t) + 0.14) * 200) (GELU' and QuickGELU' lie in [-0.13, 1.13]; step
     * 0.005, |error| <= 0.0025, the size of a bf16 rounding at 1).  Halves the bytes of the widest tensor of the step, the
     * [rows, 4C] MLP hidden derivative (Swin_AVE.py:119-126 fc1 -> GELU; its backward reads it once).  ldp / ldd are then in bytes
     * (multiples of 8).  Needs the row-layout epilogue, alpha == 1, no row_scale, bf16 C, and -- as input -- no activation / residual. */
    int dact_dtype;
    /* out: which kernel the dispatch chose (STG_GEMM_KERNEL_*), for profilers that attribute time per kernel */
    int kernel_chosen;
    /* Two row groups in one launch (split_m > 0; round 3): rows m >= split_m take W2 / bias2 (same ldw, same N, K) instead of W / bias;
     * everything else -- A, C, the saved derivative, residuals, row_scale -- is addressed by the global row as usual.  This is the
     * video | audio layout of the fused token tensor: the two modalities' adapters (S_Adapter / S_Adapter_Audio ..., Swin_AVE.py:747-748,
     * 796-797) are separate Linears over the two halves of the rows, and their narrow GEMMs (N = d_h or K = d_h, 490 workgroups each at
     * stage 2) fill the chip only together.  split_m % 128 == 0; bf16 operands, no convolution / batch; the 128 x 128 LDS-DMA kernels. */
    int64_t split_m; const void* W2; const float* bias2;
} stg_gemm_args;
enum { STG_GEMM_KERNEL_REG = 0, STG_GEMM_KERNEL_GLDS = 1, STG_GEMM_KERNEL_BIG = 2 /* retired in ABI 219: never reported */, STG_GEMM_KERNEL_8PH = 3, STG_GEMM_KERNEL_GLDS_CONV = 4,
       STG_GEMM_KERNEL_GLDS_BATCH = 5, STG_GEMM_KERNEL_GLDS_KTAIL = 6, STG_GEMM_KERNEL_FP8 = 7, STG_GEMM_KERNEL_8PHM = 8,
       STG_GEMM_KERNEL_SKINNY = 9 /* ABI 219: the adapters' down-projection as a row stream (csrc/skinny.hip) */ };
int stg_gemm_nt(stg_gemm_args* args, void* stream);

/* Block-scaled e4m3 quantisation of a bf16 matrix (the producer side of ab_dtype == STG_FP8_MX): for every row r and 32-wide
 * k-block b,  e = ceil(log2(max|x| / 448)) + 127 (127 for an all-zero block),  q = e4m3_rne(x * 2^(127 - e)).
 *   Q   [rows, ldq] bytes, ldq = K rounded up to a multiple of 128, the pad bytes zero;
 *   S   packed E8M0 table, rows rounded up to 64: byte ((r / 64) * KB + b) * 64 + (r % 16) * 4 + (r % 64) / 16 with KB = ldq / 32
 *       -- the dword a lane of the scaled MFMA needs (one byte per 16-row tile of a 64-row wave tile, picked by OPSEL);
 *       pad rows / blocks hold 127.  stg_quant_fp8_scale_bytes(rows, K) returns its size. */
int64_t stg_quant_fp8_scale_bytes(int64_t rows, int K);
int stg_quant_fp8_mx(const void* X, int64_t ldx, int64_t rows, int K, void* Q, int64_t ldq, void* S, void* stream);

/* Swin MLP in one kernel, narrow stages (stg_mlp_fused_supported(C): C = 128): Out = fc2(GELU(fc1(Y))) with the [rows, 4C] hidden
 * tensor kept in accumulator registers between the two products (replaces Mlp.forward, Swin_AVE.py:111-127 as called at :790-794).
 *   Y [rows, C] bf16, W1 [4C, C] bf16, b1 [4C] fp32, W2p [C, 4C] bf16 = fc2.weight with its hidden index permuted by stg_mlp_w2_perm
 *   (position h' holds original column perm[h']: the accumulator layout of the first product then IS the operand layout of the
 *   second), b2 [C] fp32, Out [rows, C] bf16.  Same arithmetic as two stg_gemm_nt calls (bf16 operands, fp32 accumulate, bf16 hidden). */
int stg_mlp_fused_supported(int C);
int stg_mlp_w2_perm(int hidden, int* perm);
int stg_mlp_fwd(const void* Y, int64_t ldy, const void* W1, const float* b1, const void* W2p, const float* b2,
                void* Out, int64_t ldo, int64_t rows, int C, void* stream);
/* Its backward, also one kernel: dY = ((dM . W2) * GELU'(Y . W1^T + b1)) . W1 with the pre-activation recomputed from Y (the forward
 * saves nothing of the hidden tensor; the weights are frozen, so there is no weight gradient -- Swin_AVE.py:119-126 under the
 * freeze filter of traintest_adapt_ave29.py:52-61).  W2T [4C, C] bf16 = fc2.weight^T; W1 is read once per chunk and used both
 * row-wise (first product) and transposed (third product, ds_read_b64_tr_b16). */
int stg_mlp_bwd(const void* Y, int64_t ldy, const void* dM, int64_t ldm, const void* W1, const float* b1, const void* W2T,
                void* dY, int64_t ldo, int64_t rows, int C, void* stream);

/* Weight gradient of a trainable nn.Linear y = x W^T + b   (autograd of Swin_AVE.py:15-16 D_fc1/D_fc2, :1319-1322 head)
 *   dW[N1,N2] (+)= sum_m dY[m,N1] * X[m,N2]      fp32, atomically accumulated (dW must be zeroed or hold a prior grad)
 *   db[N1]    (+)= sum_m dY[m,N1]                (optional)
 */
int stg_wgrad_tn(const void* dY, int64_t lddy, const void* X, int64_t ldx,
                 float* dW, int64_t lddw, float* db, int64_t M, int N1, int N2,
                 const float* row_scale, int64_t rs_outer, int64_t rs_inner,   /* optional DropPath scale on dY rows */
                 void* stream);

/* Same contract, without memory-side atomics, for the shapes of the adapter path (one operand <= 96 columns wide, M >= 4096,
 * 16-byte aligned operands): row splits leave partial tiles in a caller-owned fp32 workspace `ws`, and a second kernel sums
 * them into dW / db with a plain read-modify-write (dW / db must not be written concurrently by another stream).
 * stg_wgrad_ws_floats returns the workspace size in floats this path needs for (M, N1, N2), 0 when the shape is not eligible;
 * with ws == NULL, a too-small workspace or an ineligible shape the call is forwarded to stg_wgrad_tn. */
int64_t stg_wgrad_ws_floats(int64_t M, int N1, int N2);
int stg_wgrad_tn_ws(const void* dY, int64_t lddy, const void* X, int64_t ldx,
                    float* dW, int64_t lddw, float* db, int64_t M, int N1, int N2,
                    const float* row_scale, int64_t rs_outer, int64_t rs_inner,
                    float* ws, int64_t ws_floats, void* stream);
/* n <= 16 such weight gradients in ONE pair of launches (the 8 / 12 adapter Linears of a Swin block, Swin_AVE.py:747-811: 64 MB
 * streams that are launch-ramp bound on their own).  Every problem must be eligible for the workspace path and share one launch
 * plan -- same M, same narrow width class (N rounded up to 16) and wide width -- else -7 is returned and nothing is launched
 * (the caller then issues them one by one); the dW buffers must be distinct.  ws: n * stg_wgrad_ws_floats(M, N1, N2) floats. */
typedef struct {
    const void* dY; int64_t lddy; const void* X; int64_t ldx;
    float* dW; int64_t lddw; float* db;
    int64_t M; int N1; int N2;
    const float* row_scale; int64_t rs_outer; int64_t rs_inner;
} stg_wgrad_desc;
int stg_wgrad_tn_ws_multi(const stg_wgrad_desc* problems, int n, float* ws, int64_t ws_floats, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim, eps inside rsqrt (nn.LayerNorm; Swin_AVE.py:341,352 norm1/norm2, :960,976 PatchMerging.norm,
 * :1099,1117-1119 PatchEmbed3D.norm, :1312 final norm; CLIP_AVE.py:33-39).
 * gather4 != 0 folds PatchMerging's 2x2 strided gather + cat (Swin_AVE.py:967-972): logical row r of width 4*C is
 * [x(2i,2j), x(2i+1,2j), x(2i,2j+1), x(2i+1,2j+1)] of frame r / (H/2*W/2); C is then the width of one source row.
 * x_dtype / y_dtype: STG_BF16 or STG_F32 (the residual stream is fp32, branch inputs bf16); mean/rstd fp32 per logical
 * row (may be NULL in inference).
 */
int stg_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, float eps,
                      void* y, int y_dtype, int64_t ldy, float* mean, float* rstd,
                      int64_t rows, int C, int gather4, int H, int W, void* stream);
/* dx (bf16) = LN backward wrt input; if add_to != NULL, dx = add_to + LN_bwd (fuses the residual-branch join).
 * dgamma/dbeta (fp32, atomically accumulated) optional -- only CLIP ln_post is trainable (traintest_adapt_ave29.py:52). */
int stg_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int x_dtype, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, const void* add_to, int64_t ldadd,
                      void* dx, int64_t lddx, float* dgamma, float* dbeta,
                      int64_t rows, int C, int gather4, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Adapter up-projection + residual join + the LayerNorm that follows it, one row-complete pass (csrc/upln.hip):
 *   x[m,:] = res32[m,:] (+ res16[m,:]) + rs[m] * (h[m,:K] . W[:, :K]^T + bias)     fp32 residual stream, written out
 *   y[m,:] = LayerNorm(x[m,:]) * gamma + beta                                       bf16; mean / rstd fp32 per row (may be NULL)
 * replaces the chains `x = x + drop_path(T_Adapter(...))` -> norm1 (Swin_AVE.py:705-716,718), `x = shortcut + attn +
 * D_fc2(h)` -> norm2 (:780-787,790) and `x = x + xn + D_fc2(h)` -> the next block's norm1 (:810-811,703/718), which as a
 * GEMM epilogue + stg_layernorm_fwd write the fp32 row and read it straight back.
 * rs[m] = row_scale[(m / rs_outer) * rs_inner + m % rs_inner] (DropPath; NULL = 1) is applied to the bf16 h row.
 * stg_up_ln_supported(C, K): C in {128, 256, 512}, K % 8 == 0, K <= 64; everything else takes the two-kernel path.
 */
int stg_up_ln_supported(int C, int K);
int stg_up_ln_fwd(const void* h, int64_t ldh, const void* w, int64_t ldw, const float* bias, const float* res32, int64_t ld32,
                  const void* res16, int64_t ld16, const float* row_scale, int64_t rs_outer, int64_t rs_inner,
                  float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* y, int64_t ldy,
                  float* mean, float* rstd, int64_t M, int C, int K, void* stream);

/* LayerNorm backward + the adapter down-projection (D_fc2 dgrad) that consumes its result, one row-complete pass (csrc/upln.hip):
 *   dx[m,:]  = rstd (gamma dy - mean(gamma dy) - xhat mean(gamma dy xhat)) (+ add_to[m,:])      bf16, written out
 *   dh[m,:J] = rs[m] * (dx[m,:] . wt[:J,:]^T)                                                    bf16, wt = D_fc2.weight^T [J, C]
 * replaces stg_layernorm_bwd + the stg_gemm_nt that re-reads all of dx for J <= 64 output columns (backward of
 * Swin_AVE.py:716 / :780-787 / :810-811: the gradient of every residual join enters the adapter through D_fc2^T).
 * x is the fp32 residual-stream row the forward normalised; frozen norms only (no dgamma / dbeta).
 * stg_ln_bwd_down_supported(C, J): C in {128, 256, 512}, J in {16, 32, 64}. */
int stg_ln_bwd_down_supported(int C, int J);
int stg_ln_bwd_down(const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                    const float* rstd, const void* add_to, int64_t ldadd, void* dx, int64_t lddx, const void* wt, int64_t ldwt,
                    const float* row_scale, int64_t rs_outer, int64_t rs_inner, void* dh, int64_t lddh,
                    int64_t M, int C, int J, void* stream);

/* Both modalities in ONE launch (round 4).  The fused [video rows | audio rows] token tensor passes every join twice -- once per modality, each
 * with its own adapter (Swin_AVE.py:716 T_Adapter / T_Adapter_audio, :780-787 S_Adapter2 / _audio, :810-811 S_Adapter / _audio) -- and the two
 * launches differ only in (h, w, bias, row_scale) resp. (wt, row_scale, dh).  Rows [0, split_m) take the first set, rows [split_m, M) the second
 * (h2 / dh2 / row_scale2 are indexed from THEIR row 0); split_m % 16 == 0.  Per row the arithmetic is that of the single launches (bit-identical);
 * a workgroup serves one row group (it stages one adapter weight).  xhat != 0: x holds bf16 normalised rows (stg_ln_bwd_down_xhat's form). */
int stg_up_ln_fwd_pair(const void* h, const void* h2, int64_t ldh, const void* w, const void* w2, int64_t ldw, const float* bias,
                       const float* bias2, int64_t split_m, const float* res32, int64_t ld32, const void* res16, int64_t ld16,
                       const float* row_scale, const float* row_scale2, int64_t rs_outer, int64_t rs_inner, float* x, int64_t ldx,
                       const float* gamma, const float* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd, int64_t M,
                       int C, int K, void* stream);
int stg_ln_bwd_down_pair(int xhat, const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* mean,
                         const float* rstd, const void* add_to, int64_t ldadd, void* dx, int64_t lddx, const void* wt, const void* wt2,
                         int64_t ldwt, const float* row_scale, const float* row_scale2, int64_t rs_outer, int64_t rs_inner, void* dh,
                         void* dh2, int64_t lddh, int64_t split_m, int64_t M, int C, int J, void* stream);

/* The two LayerNorm backwards from the NORMALISED row (round 3).  A frozen LayerNorm in front of a frozen Linear (Swin_AVE.py:703 / :718
 * norm1 -> attn.qkv, :790 norm2 -> mlp.fc1) never needs its affine: y W^T + b = x_hat (W gamma)^T + (b + W beta), so the caller folds
 * gamma / beta into the frozen weight once, the forward writes x_hat (bf16) where it wrote y, and the backward reads that bf16 row
 * instead of the fp32 residual row + mean (per row 2C bytes instead of 4C; gamma == 1 in the formula above, dy is the gradient wrt x_hat). */
int stg_layernorm_bwd_xhat(const void* dy, int64_t lddy, const void* xhat, int64_t ldx, const float* rstd, const void* add_to,
                           int64_t ldadd, void* dx, int64_t lddx, int64_t rows, int C, void* stream);
int stg_ln_bwd_down_xhat(const void* dy, int64_t lddy, const void* xhat, int64_t ldx, const float* rstd, const void* add_to,
                         int64_t ldadd, void* dx, int64_t lddx, const void* wt, int64_t ldwt, const float* row_scale,
                         int64_t rs_outer, int64_t rs_inner, void* dh, int64_t lddh, int64_t M, int C, int J, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Generic gather-mapped multi-head attention (flash-style, MFMA 32x32x16, scores never hit HBM).
 * One "problem" p in [0, P) attends n query tokens to n_kv key tokens; token i of problem p lives at row
 *     row(p,i) = (p / G) * outer + map[(p % G) * n + i]            (map == NULL: identity (p % G) * n + i)
 * so that torch.roll + window_partition / window_reverse (Swin_AVE.py:130-159,727-740,765-776), the temporal
 * '(b t) n c -> (b n) t c' rearranges (:705,711) and the per-frame global grouping (:801-805) are pure addressing.
 *   S = scale * Q K^T + bias[(p / bias_div) % bias_mod][h] + mask[p % G]      (bias: fp32 [*,H,n,n_kv], mask: fp32 [G,n,n_kv])
 *   O = softmax(S) V ;  lse[p,h,i] = log sum exp (fp32, for backward)
 * replaces WindowAttention.forward spatial (:256-276) and temporal (:244-255) branches, the cross-modal
 * softmax(h_v h_a^T) h_a pairs (:753-757, :801-805; CLIP_AVE.py:386-398,415-424) with H=1, scale=1, K=V,
 * and nn.MultiheadAttention's core (CLIP_AVE.py:106-108).
 * D (head dim) in {16,32,48,64,96,128}.  Q/K/V/O rows are at base + row*ld + h*D.
 */
typedef struct {
    const void* Q; int64_t ldq;
    const void* K; int64_t ldk;
    const void* V; int64_t ldv;
    void* O; int64_t ldo;
    float* lse;                       /* [P, H, n] or NULL */
    const int32_t* map_q; const int32_t* map_kv;
    int64_t outer_q, outer_kv;
    int G;
    /* map_kind 0: tables above (NULL = identity).  Arithmetic maps (no lookup in front of the operand loads), same map for
     * queries and keys:  1 = cyclic shift + window partition, map_a/b = image H/W in tokens, map_c = window size,
     * map_d = shift (G = windows per image, n = ws*ws);  2 = temporal regrouping, map_a = tokens per frame (= G), n = T. */
    int map_kind; int map_a, map_b, map_c, map_d;
    int64_t P; int H; int n; int n_kv; int D;
    float scale;
    const float* bias; int64_t bias_div; int bias_mod;
    const float* mask;
} stg_attn_args;
int stg_attn_fwd(const stg_attn_args* a, void* stream);

/* Backward: dQ/dK/dV written (not accumulated) with the same addressing as Q/K/V (each row belongs to exactly one problem).
 * delta[p,h,i] = sum_d dO*O is computed by stg_attn_bwd_prep into a caller workspace.
 * dbias (fp32 [bias_mod,H,n,n_kv], atomically accumulated) optional: temporal_position_bias_table(_audio) is trainable. */
typedef struct {
    stg_attn_args f;                  /* forward description (O = forward output, lse = forward lse) */
    const void* dO; int64_t lddo;
    void* dQ; int64_t lddq;
    void* dK; int64_t lddk;
    void* dV; int64_t lddv;
    float* delta;                     /* workspace [P, H, n] */
    float* dbias;
} stg_attn_bwd_args;
int stg_attn_bwd(const stg_attn_bwd_args* a, void* stream);
/* The two directions of one cross-modal pair (Swin_AVE.py:799-808: h_v <- h_a and h_a <- h_v) in one call: where both take the
 * frame-global kernels with one geometry they share a launch (grid.y = 2), otherwise the call equals two stg_attn_fwd / _bwd. */
int stg_attn_fwd2(const stg_attn_args* f0, const stg_attn_args* f1, void* stream);
int stg_attn_bwd2(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* stream);
/* Backward of a frame-global cross-modal PAIR in one pass per modality (round 5; replaces the two stg_attn_bwd calls and the torch.add of
 * their results behind AVE/model/Swin_AVE.py:796-811's autograd): b0 = direction v (Q = h_v, K = V = h_a, O = r_v, lse_v, dO = d r_v), b1 = the
 * mirror image (Q = h_a, K = V = h_v, ...); g0 / g1 [rows, D] bf16 (leading dimension ldg) receive the COMPLETE gradients of h_v / h_a through
 * both directions (dQ of the own direction + dK + dV of the other) -- the b->dQ / dK / delta fields are not used.  One exponential per score
 * instead of two while the frame's log-sum-exps span <= 120 binary orders, the plain two-exponential arithmetic otherwise (decided per frame on
 * the device, no host synchronisation).  ws: stg_xattn_pair_bwd_ws_bytes(P, n0, n1, D) bytes of 16-byte-aligned scratch.  _supported: both
 * directions eligible for the frame-global kernels (H = 1, D = 16 / 32, K == V, dense frames of >= 64 rows) and mirror images of each other. */
int64_t stg_xattn_pair_bwd_ws_bytes(int64_t P, int n0, int n1, int D);
int stg_xattn_pair_bwd_supported(const stg_attn_args* f0, const stg_attn_args* f1);
int stg_xattn_pair_bwd(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, void* ws, int64_t ws_bytes,
                       void* stream);
/* The same with the join that follows it in the adapters' backward: g <- (jx + G) * jz per modality (jx = gradient of the gated hidden state,
 * jz = saved activation derivative of D_fc1: stg_add3_mul2's arithmetic on the bf16-rounded G, no separate pass). */
int stg_xattn_pair_bwd_join(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, const void* jx0,
                            const void* jx1, int64_t ldjx, const void* jz0, const void* jz1, int64_t ldjz, void* ws, int64_t ws_bytes, void* stream);
/* ABI 220: the same with the pair's GATES inside: b0->dO / b1->dO are d(x) of x = q + gate o (stg_xattn_fwd2_gate's x), the kernels work on
 * bf16(gate d(x)) -- stg_gate_bwd2's rounding: G is bit-identical to stg_gate_bwd2 followed by stg_xattn_pair_bwd(_join) -- and
 * dgate_y[0] += <d(x_y), o_y> (fp32 atomics, one per workgroup of the preparation kernel).  jx / jz: the optional join (all four NULL: none). */
int stg_xattn_pair_bwd_gate(const stg_attn_bwd_args* b0, const stg_attn_bwd_args* b1, void* g0, void* g1, int64_t ldg, const void* jx0,
                            const void* jx1, int64_t ldjx, const void* jz0, const void* jz1, int64_t ldjz, const float* gate0, const float* gate1,
                            float* dgate0, float* dgate1, void* ws, int64_t ws_bytes, void* stream);
/* Forward of such a pair with its gates: O and lse per direction as stg_attn_fwd2 writes them, and x = q + gate[0] * o (stg_gate_fwd2's
 * arithmetic on the bf16-rounded o) into x0 / x1 [rows, D] -- h_v' = h_v + gate_v softmax(h_v h_a^T) h_a, Swin_AVE.py:799-808. */
int stg_xattn_fwd2_gate(const stg_attn_args* f0, const stg_attn_args* f1, const float* gate0, const float* gate1, void* x0, void* x1,
                        int64_t ldx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-window attention: WindowAttention.forward's spatial branch (Swin_AVE.py:256-276) with roll + window_partition /
 * window_reverse + roll (:727-740, :765-776) as addressing, specialised for windows of n = ws*ws <= 64 tokens and head
 * dim 32 (every Swin-T/S/B/L stage).  One wavefront owns one (window, head): the padded 64 x 64 score block stays in
 * registers, and the backward produces dQ, dK and dV in ONE kernel (no S / dP recompute split, no delta workspace).
 *
 * stg_winattn_table builds the additive term once per block (the bias table is frozen, traintest_adapt_ave29.py:52-61):
 *   bm [g][h][q][k] = log2(e) * (table[index[q*n + k]][h] + mask[g][q][k])   (Swin_AVE.py:262-273; the kernels run the
 *   softmax on exp2), padded to 64 x 64 with k >= n -> -1e30 (padding keys drop out of the softmax) and q >= n -> 0;
 *   bmT[g][h][k][q] = bm[g][h][q][k].
 *   table fp32 [L, H], index int64 [n*n], mask fp32 [Gt, n, n] or NULL (then Gt = 1).
 * Image pg = p / G occupies rows [pg*outer, pg*outer + Himg*Wimg); Q, K, V share one leading dimension (the fused qkv
 * buffer), Q/K/V/O/dO/dQ/dK/dV of head h at base + row*ld + h*D.  lse: fp32 [P, H, 64] (entries >= n unused).
 * ABI 219: bm == bmT == NULL = no bias and no mask (the adapters' window-level cross-modal pair, Swin_AVE.py:750-760: the additive
 * term is synthesised from the key index, nothing is fetched; 7 x 7 windows), and with it D == 16 (Swin-B stage 0's d_h: 32-byte
 * rows in memory, staged beside a zero line into the 32-wide tiles the kernels work on).
 */
typedef struct {
    const void* Q; const void* K; const void* V; int64_t ld;
    void* O; int64_t ldo;
    float* lse;
    const float* bm; const float* bmT;
    int Gt;                           /* tables per image: G (shifted blocks) or 1 */
    int64_t outer;
    int Himg, Wimg, ws, shift;        /* image size in tokens, window size, cyclic shift */
    int G, n;                         /* windows per image, tokens per window (= ws*ws) */
    int64_t P; int H; int D;
    float scale;
} stg_winattn_args;
int stg_winattn_table(const float* table, const int64_t* index, const float* mask, float* bm, float* bmT,
                      int L, int H, int n, int Gt, void* stream);
int stg_winattn_fwd(const stg_winattn_args* a, void* stream);
/* dV == NULL: K and V are the same tensor (the adapters' window-level cross-modal attention softmax(h hother^T) hother,
 * AVE/model/Swin_AVE.py:750-760, run with H = 1 and no table: bm == bmT == NULL) and dK receives dK + dV. */
int stg_winattn_bwd(const stg_winattn_args* a, const void* dO, int64_t lddo, void* dQ, void* dK, void* dV,
                    int64_t lddqkv, void* stream);
/* ABI 220: the adapters' window-level cross-modal PAIR (Swin_AVE.py:750-760: h_v' = h_v + gate_v softmax(h_v h_a^T) h_a and its mirror image) as ONE
 * launch per pass with the gates inside.  a0 / a1: the two directions, table-free (bm == bmT == NULL), H == 1, K == V == the other modality's rows, one
 * window geometry.  Forward: O and lse per direction as stg_winattn_fwd writes them, and x_y = Q_y + gate_y[0] * O_y (stg_gate_fwd2's arithmetic on the
 * bf16-rounded O) into x0 / x1 [rows, >= D].  Backward: dX_y = d(x_y); dQ_y / dK_y (dK receives dK + dV) are the gradients through the attention, scaled
 * by gate_y[0] inside the kernel (the pass is linear in dO), and dgate_y[0] += <dX_y, O_y> (fp32 atomics, one per wavefront) -- stg_gate_bwd2 is not
 * needed.  The gradient through the residual term (dX_y itself) is the caller's. */
int stg_winattn_pair_fwd(const stg_winattn_args* a0, const stg_winattn_args* a1, const float* gate0, const float* gate1, void* x0, void* x1,
                         int64_t ldx, void* stream);
int stg_winattn_pair_bwd(const stg_winattn_args* a0, const stg_winattn_args* a1, const void* dX0, const void* dX1, int64_t lddx, const float* gate0,
                         const float* gate1, float* dgate0, float* dgate1, void* dQ0, void* dK0, void* dQ1, void* dK1, int64_t lddqkv, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Temporal attention: WindowAttention.forward's temporal branch (Swin_AVE.py:244-255; block call sites :705-716) with the
 * '(b t) n c -> (b n) t c' rearranges as addressing.  The fused token tensor holds nm modality slabs of B clips x T frames x
 * N tokens; frame t of token n of clip b of slab m is row ((m*B + b)*T + t)*N + n, and every (m, b, n) is one sequence of T
 * frames.  S = scale * Q K^T + bias[m][h] (fp32 [nm, H, T, T], temporal_position_bias_table(_audio) gathered by
 * t_relative_coords(_a), :246-253), O = softmax(S) V.  T <= 32, head dim 32 -- or, with bias == NULL (and bm / bmT unused), head dim
 * 64 / 96: the temporal nn.MultiheadAttention of the CLIP ViT blocks (CLIP_AVE.py:369-377).  Q, K, V share one leading dimension (the fused
 * qkv buffer); Q/K/V/O/dO/dQ/dK/dV of head h at base + row*ld + h*D.
 * bm / bmT: caller-owned fp32 workspaces of nm*H*1024 floats each, FILLED by stg_tattn_fwd (block-diagonal additive tables)
 * and read again by stg_tattn_bwd, which must be handed the same, unmodified buffers.
 * The backward recomputes the softmax from Q and K (it needs neither O nor an LSE) and writes dQ, dK, dV in one kernel;
 * dbias (fp32 [nm, H, T, T], atomically accumulated, optional) is the gradient of the trainable bias term.
 */
typedef struct {
    const void* Q; const void* K; const void* V; int64_t ld;
    void* O; int64_t ldo;
    const float* bias;
    float* bm; float* bmT;
    int nm; int64_t B; int T; int N; int H; int D;
    float scale;
} stg_tattn_args;
int stg_tattn_fwd(const stg_tattn_args* a, void* stream);
int stg_tattn_bwd(const stg_tattn_args* a, const void* dO, int64_t lddo, void* dQ, void* dK, void* dV, int64_t lddqkv,
                  float* dbias, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head self-attention core of the CLIP ViT blocks: nn.MultiheadAttention inside ResidualAttentionBlock.attention
 * (CLIP_AVE.py:106-108; spatial call sites :379-383 and the audio twin), no mask, no bias, dropout 0.  P frames x H heads,
 * n tokens per frame at rows p*n .. p*n + n-1, head dim D in {64, 96}; Q, K, V are column slices of the fused in_proj output
 * (one leading dimension), head h at columns h*D.  O = softmax(scale * Q K^T) V; lse (fp32 [P, H, n]) is saved in the
 * log2 domain for the backward.  stg_mha_bwd runs two kernels (dQ, which also fills the caller's `delta` workspace
 * fp32 [P, H, n] = rowsum(dO * O), then dK/dV) and writes dQ, dK, dV with the addressing of Q, K, V.
 * The same kernels serve the frame-global cross-modal attention of wide adapters (Swin_AVE.py:801-805: H = 1, scale 1, K and V
 * the SAME tensor = the other modality's hidden states; d_h = 96 in Swin-L, 64 in Swin-B stage 3): pass K == V and, in the
 * backward, dV == NULL -- dK then receives the gradient of the shared tensor (dK + dV).
 */
typedef struct {
    const void* Q; const void* K; const void* V; int64_t ld;
    void* O; int64_t ldo;
    float* lse;
    int64_t P; int H; int n; int D;
    float scale;
    /* window map (ABI 216; all zero = rows p*n + i): win_size > 0 makes problem p the window p % nW of frame p / nW of a [win_h, win_w] token
     * image (nW = (win_h / win_size) * (win_w / win_size), n = win_size^2, cyclic shift win_shift) -- the geometry of stg_attn's
     * map_kind 1 -- for the adapters' WINDOW-level cross-modal attention at widths 64 / 96 (Swin_AVE.py:750-760 with d_h = 96 in Swin-L). */
    int win_h, win_w, win_size, win_shift;
} stg_mha_args;
int stg_mha_supported(int n, int D);
int stg_mha_fwd(const stg_mha_args* a, void* stream);
int stg_mha_bwd(const stg_mha_args* a, const void* dO, int64_t lddo, void* dQ, void* dK, void* dV, int64_t lddqkv,
                float* delta, void* stream);
/* Two problems of ONE geometry (P, H, n, D, scale, window map, leading dimensions, K == V or not) in one launch each -- the two directions of a
 * cross-modal pair (Swin_AVE.py:750-760, :799-808) -- ABI 218.  Same results as two calls. */
int stg_mha_fwd_pair(const stg_mha_args* a0, const stg_mha_args* a1, void* stream);
int stg_mha_bwd_pair(const stg_mha_args* a0, const void* dO0, void* dQ0, void* dK0, void* dV0, float* delta0, const stg_mha_args* a1,
                     const void* dO1, void* dQ1, void* dK1, void* dV1, float* delta1, int64_t lddo, int64_t lddqkv, void* stream);
/* ABI 219: the backward of a cross-modal PAIR as one pass per modality (Swin_AVE.py:796-811 / :750-760 with d_h 64 / 96): direction 0 = (Q = X, K = V = Y),
 * direction 1 = (Q = Y, K = V = X) -- the same two tensors with their roles swapped (checked: -7 otherwise).  Both directions share S = X Y^T, so
 * G0 = d loss / d X = dQ of direction 0 + dK + dV of direction 1 (and G1 for Y) costs three score-type and two output-type products per tile pair
 * instead of the four + six of stg_mha_bwd_pair, which it replaces together with the caller's dQ + dKV addition.  delta0 / delta1: fp32 [P, H, n]
 * workspaces (filled here).  G0 / G1 bf16 with leading dimension lddg. */
int stg_mha_bwd_pair_merged(const stg_mha_args* a0, const void* dO0, void* G0, float* delta0, const stg_mha_args* a1, const void* dO1,
                            void* G1, float* delta1, int64_t lddo, int64_t lddg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The cross-modal PAIR of the CLIP-ViT blocks on small frames (ABI 219; csrc/xsmall.hip): per frame p, video rows Xv[p*nv .. +nv) and audio rows
 * Xa[p*na .. +na) of width D; Ov = softmax(scale Xv Xa^T) Xa, Oa = softmax(scale Xa Xv^T) Xv (CLIP_AVE.py:386-398; nv <= 256, na <= 64,
 * D in {32, 48, 64}: ViT-B/16's 197 + 49 tokens at adapter width 48).  One workgroup per frame, both modalities' rows in LDS; forward = both
 * directions in one launch (lse_v / lse_a: fp32 [P, nv] / [P, na], log2 domain), backward = both whole gradients in one launch:
 * Gv = d loss / d Xv (dQ of the video direction + dK + dV of the audio direction), Ga likewise.  Replaces the generic stg_attn_fwd2 / _bwd2 pair
 * launches (and the caller's dQ + dKV addition) for these shapes.
 */
typedef struct {
    const void* Xv; const void* Xa; int64_t ldv, lda;
    void* Ov; void* Oa; int64_t ldov, ldoa;
    float* lse_v; float* lse_a;
    int P, nv, na, D;
    float scale;
} stg_xsmall_args;
int stg_xsmall_supported(int nv, int na, int D);
int stg_xsmall_fwd(const stg_xsmall_args* a, void* stream);
int stg_xsmall_bwd(const stg_xsmall_args* a, const void* dOv, const void* dOa, int64_t lddov, int64_t lddoa, void* Gv, void* Ga,
                   int64_t ldgv, int64_t ldga, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Small element-wise / layout kernels
 */
/* out = h + gate[0] * r  (Swin_AVE.py:759-760,807-808); all bf16 [rows, d] contiguous-ld tensors, gate fp32 scalar on device */
int stg_gate_fwd(const void* h, const void* r, const float* gate, void* out, int64_t numel, void* stream);
/* dh = dout (alias allowed by caller), dr = gate*dout, dgate += sum(dout*r) */
int stg_gate_bwd(const void* dout, const void* r, const float* gate, void* dr, float* dgate, int64_t numel, void* stream);

/* Conv3d/Conv2d with kernel == stride as a gather (im2col is a pure re-indexing then; Swin_AVE.py:1097,1115 PatchEmbed3D.proj,
 * CLIP_AVE.py:1091-1094 conv1).  x: [B, Cin, T, Hin, Win] (fp32 or bf16, contiguous) -> out bf16 [B*T*(Hin/p)*(Win/p), Kpad],
 * column = c*p*p + ph*p + pw, zero-padded to Kpad. */
int stg_im2col_patch(const void* x, int x_dtype, void* out, int64_t B, int Cin, int T, int Hin, int Win, int p, int Kpad,
                     void* stream);

/* fp32 -> bf16 cast for weight shadows / gradient hand-over: in [R,Cc] contiguous -> out [R, ld_out] (or, transposed,
 * out [Cc, ld_out]); columns beyond the data are zero-filled up to ld_out (GEMM needs K % 8 == 0). */
int stg_cast_bf16(const float* in, void* out, int64_t R, int64_t Cc, int transpose, int64_t ld_out, void* stream);
/* The same cast for MANY small matrices in one launch (the trainable adapter / head weights change every optimizer step,
 * ~260 matrices for Swin-B, each needed in both orientations by the forward and the dgrad GEMMs).  `descs` is an array of n
 * descriptors IN DEVICE MEMORY; matrix t (fp32 [R, C] contiguous at `in`) is written as bf16 [R, ld] at arena + off and, when
 * offT >= 0, transposed as [C, ldT] at arena + offT (offsets in bf16 elements).  Padding columns are not written: zero the
 * arena first.  max_elems = the largest R*C (sizes the grid). */
typedef struct { const float* in; int64_t off; int64_t offT; int R, C, ld, ldT; } stg_cast_desc;
int stg_cast_bf16_multi(const stg_cast_desc* descs, int n, int max_elems, void* arena, void* stream);
/* torch.optim.Adam's update (the reference's optimizer, AVE/traintest_adapt_ave29.py:63-69: betas (0.95, 0.999), weight_decay 5e-7 as
 * L2 folded into the gradient, no amsgrad, not maximize) for MANY tensors in one launch, and capturable in a HIP graph: nothing of a
 * step lives on the host.  `descs`: n descriptors IN DEVICE MEMORY (p / m / v updated in place, g read; all fp32, n elements).
 * `st` of a descriptor: that tensor's own {step, lr / bias_correction1, sqrt(bias_correction2), -} (device, fp32) -- torch counts
 * steps per parameter (a parameter that receives its first gradient later starts later).  `hyper`: device DOUBLE [n_groups][8] =
 * {lr, beta1, beta2, eps, weight_decay, -, -, -} (double: torch evaluates 1 - beta and the corrections from Python floats; beta2 =
 * 0.999 rounded to fp32 first would move 1 - beta2 by 1.3e-5).  The call first advances `step` of every LISTED tensor by one and
 * refreshes its step size and correction (one small launch), then applies, per element and in torch's order of operations,
 *     g' = g + wd p;  m += (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g' g';  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
 * max_elems = the largest tensor (sizes the grid).  ATen runs this as ~13 multi-tensor launches, and its graph-capturable form as
 * ~1 000 single-element launches per step. */
typedef struct { float* p; const float* g; float* m; float* v; float* st; int64_t n; int group; int pad_; } stg_adam_desc;
int stg_adam_multi(const stg_adam_desc* descs, int n, int64_t max_elems, const double* hyper, int n_groups, void* stream);
/* x[((b T + t) N + n), :] += emb[t, :]  -- the absolute temporal embedding of t_relative=False, '(b t) n c -> (b n) t c' + embedding
 * (AVE/model/Swin_AVE.py:1483-1487, :1569-1576): x fp32 [B*T*N, C] in place, emb fp32 [T, C].  Its gradient is a token mean
 * (stg_meanpool_fwd over the N rows of every frame) summed over the clips. */
int stg_add_temporal(float* x, const float* emb, int64_t B, int T, int64_t N, int C, void* stream);
/* Audio front end (AVE/dataloader.py:204-272 `_wav2fbank`): Kaldi-compatible log-mel filterbank features of S
 * waveform segments -- torchaudio.compliance.kaldi.fbank(htk_compat=True, use_energy=False, window_type='hanning', dither=0,
 * num_mel_bins, frame_shift; other options at their defaults: frame DC removal, pre-emphasis, power spectrum of the frame zero-padded
 * to `padded` samples, log with the fp32-epsilon floor) -- then (x - norm_mean) / (2 norm_std) and zero padding / cropping to
 * target_frames rows.  wave fp32 [S, n_samples] (row stride wave_stride), window fp32 [size], melw fp32 [num_mel_bins, padded/2 + 1]
 * (the caller builds both once: stg-cma_amd/audio.py), out fp32 [S, target_frames, num_mel_bins] = the models' spectrogram input. */
int stg_fbank(const float* wave, int64_t n_samples, int64_t wave_stride, int S, int shift, int size, int padded,
              const float* window, const float* melw, int num_mel_bins, float preemphasis, float norm_mean, float norm_std,
              int target_frames, float* out, void* stream);
/* Video front end (video half): the tensor part of AVE/dataloader.py:346-394 (_aug_frame_train behind the PIL
 * RandAugment) in one launch -- ToTensor, tensor_normalize (:470-485), random_resized_crop (transforms/video_transforms.py:529-561:
 * crop box + bilinear resize, align_corners = False), horizontal_flip (:152-186), RandomErasing in 'pixel' mode with one box per clip
 * (transforms/random_erasing.py:118-152) -- from decoded frames to the models' video input.
 * frames u8 [B, T, H, W, 3] (host-decoded, RandAugment applied); params int32 [B, 9] = crop top, left, height, width, flip (0 / 1), erase
 * top, left, height, width in OUTPUT coordinates (height 0: no erase) -- the caller's random draws, the host points of the reference
 * (Python / NumPy generators); noise fp32 [B, T, 3, S, S], read inside the erase box only (may be NULL when no clip erases);
 * mean3 / std3 HOST pointers to 3 floats; out fp32 [B, 3, T, S, S] ('b c t h w', what model(a, v, mode) takes). */
int stg_video_aug(const void* frames, int B, int T, int H, int W, const int32_t* params, const float* noise, const float* mean3,
                  const float* std3, float* out, int S, void* stream);
/* bf16 -> fp32 */
int stg_cast_f32(const void* in, float* out, int64_t numel, void* stream);

/* token mean over n rows: in bf16 [G, n, C] -> out (bf16 or fp32) [G, C] written at out + g*ldo  (AdaptiveAvgPool1d, Swin_AVE.py:1588-1594) */
int stg_meanpool_fwd(const void* in, void* out, int out_dtype, int64_t ldo, int64_t G, int n, int C, void* stream);
int stg_meanpool_bwd(const void* dout, int64_t lddo, void* din, int64_t G, int n, int C, void* stream);

/* out = a + b (+ c) (bf16), joins gradient branches; c may be NULL */
int stg_add(const void* a, const void* b, const void* c, void* out, int64_t numel, void* stream);
/* dz = dh * dact  (bf16), backward of the adapter activation (Swin_AVE.py:21 GELU; CLIP QuickGELU) given the derivative
 * act'(pre-activation) that stg_gemm_nt saved in its `dact` output */
int stg_act_bwd(const void* dh, const void* dact, void* dz, int64_t numel, void* stream);
/* out = (a + b + c) * z (bf16): the three gradient paths into an adapter hidden state (own path + the two cross-modal
 * directions, backward of Swin_AVE.py:750-760 / :799-808) joined and taken through the activation derivative in one pass. */
int stg_add3_mul(const void* a, const void* b, const void* c, const void* z, void* out, int64_t numel, void* stream);
/* stg_gate_fwd / stg_gate_bwd / stg_add3_mul on TWO equally sized problems in one launch: the two directions of a cross-modal pair
 * (Swin_AVE.py:750-760, :799-808).  A problem is a few MB there and its launch is latency: 3 x 96 launches per Swin-B step -> 3 x 48. */
int stg_gate_fwd2(const void* h0, const void* r0, const float* g0, void* out0, const void* h1, const void* r1, const float* g1,
                  void* out1, int64_t numel, void* stream);
int stg_gate_bwd2(const void* dout0, const void* r0, const float* g0, void* dr0, float* dgate0, const void* dout1, const void* r1,
                  const float* g1, void* dr1, float* dgate1, int64_t numel, void* stream);
/* out = (a + b + c) z on two equally sized problems; c0 == c1 == NULL: (a + b) z (the join behind stg_xattn_pair_bwd) */
int stg_add3_mul2(const void* a0, const void* b0, const void* c0, const void* z0, void* out0, const void* a1, const void* b1,
                  const void* c1, const void* z1, void* out1, int64_t numel, void* stream);
/* The same three with a size per problem (ABI 217): ViT's pair has 197 video and 49 audio tokens per frame (CLIP_AVE.py:379-401). */
int stg_gate_fwd2n(const void* h0, const void* r0, const float* g0, void* out0, int64_t numel0, const void* h1, const void* r1,
                   const float* g1, void* out1, int64_t numel1, void* stream);
int stg_gate_bwd2n(const void* dout0, const void* r0, const float* g0, void* dr0, float* dgate0, int64_t numel0, const void* dout1,
                   const void* r1, const float* g1, void* dr1, float* dgate1, int64_t numel1, void* stream);
int stg_add3_mul2n(const void* a0, const void* b0, const void* c0, const void* z0, void* out0, int64_t numel0, const void* a1,
                   const void* b1, const void* c1, const void* z1, void* out1, int64_t numel1, void* stream);
/* Test aid: fill the LDS of every CU with NaN bit patterns, so that a kernel reading an LDS byte nobody wrote produces a non-finite result
 * (stgcma._lib wraps every launch with it under STG_LDS_POISON=1; tests/test_lds_poison_gpu.py). */
int stg_debug_poison_lds(void* stream);
/* out = a * mask  (bf16 * fp32 mask), Dropout in mlp_head (Swin_AVE.py:1320) */
int stg_mul_mask(const void* a, const float* mask, void* out, int64_t numel, void* stream);
/* bias gather: out[g,h,i,j] = table[index[i*nj+j] , h]  (Swin_AVE.py:246-253,257-261) ; table fp32 [L,H] */
int stg_bias_gather(const float* table, const int64_t* index, float* out, int L, int H, int nn, void* stream);
/* dtable[index[ij], h] += dbias[h, ij] */
int stg_bias_scatter(const float* dbias, const int64_t* index, float* dtable, int L, int H, int nn, void* stream);
/* ViT token assembly (CLIP_AVE.py:1091-1103): out[bt,0] = class_embedding + pos[0] + temb[t], out[bt,1+i] = patch[bt,i] +
 * pos[1+i] + temb[t]; patch bf16 [BT*np, D] (conv-as-GEMM output), pos fp32 [np+1, D], temb fp32 [T, D], out fp32. */
int stg_vit_embed(const void* patch, const float* cls, const float* pos, const float* temb, float* out, int64_t BT, int T,
                  int np, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * AVQA question-answering head (AVQA/model/Swin_AVQAModel_V1.py:37-59 QstEncoder, :1768-1903 forward): the small kernels
 * around its GEMMs (which run on stg_gemm_nt / stg_wgrad_tn).  bf16 storage, fp32 arithmetic.
 */
/* y = relu(x) (op 0) / tanh(x) (op 1)  (F.relu :1783,1821-1823; nn.Tanh :42,50,1817,1886,1891); the backward takes the INPUT x */
int stg_unary_fwd(int op, const void* x, void* y, int64_t numel, void* stream);
int stg_unary_bwd(int op, const void* x, const void* dy, void* dx, int64_t numel, void* stream);
/* out = a * b (torch.mul(feat, qst_feature), :1890) */
int stg_mul(const void* a, const void* b, void* out, int64_t numel, void* stream);
/* nn.Embedding (:41,48): out bf16 [n, E] = table fp32 [V, E][idx];  dtable[idx] += dout (fp32 atomics) */
int stg_embed_fwd(const float* table, const int64_t* idx, void* out, int64_t n, int V, int E, void* stream);
int stg_embed_bwd(const void* dout, const int64_t* idx, float* dtable, int64_t n, int V, int E, void* stream);
/* nn.LSTM cell (:44,52): gates fp32 [B, 4H] in (i, f, g, o) order = x W_ih^T + b_ih + h W_hh^T + b_hh;  c' = f c + i g,
 * h' = o tanh(c').  c fp32, h bf16.  Backward: dh (bf16) / dc (fp32) may be NULL; dgates bf16 [B, 4H], dc_prev fp32. */
int stg_lstm_cell_fwd(const float* gates, const float* c_prev, float* c, void* h, int64_t B, int H, void* stream);
int stg_lstm_cell_bwd(const float* gates, const float* c_prev, const float* c, const void* dh, const float* dc,
                      void* dgates, float* dc_prev, int64_t B, int H, void* stream);
/* Audio-visual grounding of one frame (:1797-1815, :1829-1843): V fp32 [F, n, C] visual tokens (n <= 64), a bf16 [F, C];
 * vmean = mean_j V_j (AdaptiveAvgPool2d), grd = sum_j softmax_j(V^_j . a^) V^_j with ^ = F.normalize(dim = C).
 * p, rnorm fp32 [F, n] and ra fp32 [F] are saved for the backward.  dV (fp32 [F, n, C], written) may be NULL (negative clip),
 * dvmean may be NULL. */
int stg_grounding_fwd(const float* V, const void* a, void* vmean, void* grd, float* p, float* rnorm, float* ra,
                      int64_t F, int n, int C, void* stream);
int stg_grounding_bwd(const float* V, const void* a, const float* p, const float* rnorm, const float* ra,
                      const void* dvmean, const void* dgrd, float* dV, void* da, int64_t F, int n, int C, void* stream);
/* nn.MultiheadAttention core with ONE query per batch element (:1866-1880): q bf16 [B, H*hd], k / v bf16 [T, B, H*hd] (already
 * projected), drop fp32 [B, H, T] = Bernoulli(keep) / keep or NULL; o = sum_t (softmax_t(scale q.k_t) * drop)_t v_t; p fp32
 * [B, H, T] (pre-dropout) saved.  T <= 64. */
int stg_mha1_fwd(const void* q, const void* k, const void* v, const float* drop, void* o, float* p, int B, int H, int T,
                 int hd, float scale, void* stream);
int stg_mha1_bwd(const void* q, const void* k, const void* v, const float* drop, const float* p, const void* dout,
                 void* dq, void* dk, void* dv, int B, int H, int T, int hd, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * AVS dense decoder (AVS/model/Swin_AVSModel_Base.py:14-130, :1474-1506, :1838-1894; AVS/model/TPAVI.py).  Feature maps are
 * channels-last token rows [F*H*W, C] bf16.
 */
/* im2col of a 3x3 convolution, stride 1, padding = dilation (nn.Conv2d(k=3, padding=d, dilation=d), :18-20,56-61,:1497-1503):
 * out bf16 [F*H*W, 9*C], column (kh*3 + kw)*C + c = x[f, h + (kh-1) d, w + (kw-1) d, c] or 0.  The convolution is then
 * stg_gemm_nt(out, Wm) with Wm[o, (kh, kw, c)] = W[o, c, kh, kw]; its data gradient the same gather on dY with
 * Wd[c, (kh, kw, o)] = W[o, c, 2-kh, 2-kw]; its weight gradient stg_wgrad_tn(dY, out).  C % 8 == 0. */
int stg_im2col3x3(const void* x, int64_t ldx, void* out, int64_t F, int H, int W, int C, int dilation, void* stream);
/* Weight gradient of the same convolution without the im2col image (autograd of nn.Conv2d(I, O, 3, padding = dilation) at
 * Swin_AVSModel_Base.py:27,37-38,117): ws[s, o, (kh, kw, i)] = sum over the s-th slice of the rows m of dy[m, o] * x[pixel(m) shifted
 * by tap (kh, kw), i]; dW = sum_s ws[s] (fp32; the caller reduces and owns the workspace).  O % 8 == 0 and I % 8 == 0 (tiles of 128 x 128,
 * zero-padded inside the kernel).  A convolution with FEW output channels is cheaper with the operands swapped -- stg_conv3x3_wgrad(x, dy) gives
 * T[i, (kh, kw, o)] with dW[o, (kh, kw, i)] = T[i, (2 - kh, 2 - kw, o)] (the shift moves to the small operand; kernels.conv3x3_wgrad does this
 * below 64 output channels).
 * stg_conv3x3_wgrad_ws_floats returns the workspace size in floats (and the number of slices), -1 when the shape is unsupported. */
int64_t stg_conv3x3_wgrad_ws_floats(int64_t M, int O, int I, int* splits_out);
int stg_conv3x3_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* zero_line, float* ws, int64_t ws_floats,
                      float* db_ws /* [splits, O] partial column sums of dy (bias gradient), or NULL */,
                      int64_t F, int H, int W, int O, int I, int dilation, void* stream);
/* The same tn-GEMM without taps: ws[s] = partial dY^T X of the s-th row slice for a trainable Linear / 1x1 convolution whose
 * two widths are both multiples of 128 (the AVS decoder's tap Linears and TPAVI 1x1 convolutions, Swin_AVSModel_Base.py,
 * TPAVI.py:40-75), in place of the atomic fallback of stg_wgrad_tn; dW = sum_s ws[s] (caller), db = column sums of dY. */
int64_t stg_wgrad_wide_ws_floats(int64_t M, int N1, int N2, int* splits_out);
int stg_wgrad_wide(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* zero_line, float* ws, int64_t ws_floats,
                   float* db_ws /* [splits, N1] or NULL */, int64_t M, int N1, int N2, void* stream);
/* batch of independent dY_b^T X_b over consecutive groups of M rows (problem b: rows b*M .. b*M + M - 1 of both operands):
 * ws[b, s, N1, N2] partial tiles; stg_wgrad_wide_ws_floats(M, N1, N2, &splits) floats per problem. */
/* out[b, :] (+)= sum_s ws[b, s, :] (fp32; n floats per slice, n % 4 == 0): folds the partial tiles of the three entries above. */
int stg_sum_splits(const float* ws, float* out, int splits, int64_t n, int64_t batch, int accumulate, void* stream);
int stg_wgrad_wide_batched(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* zero_line, float* ws, int64_t ws_floats,
                           int64_t M, int N1, int N2, int batch, void* stream);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=...) (:108-110 align_corners=True, :1500 False) and its adjoint */
int stg_bilinear_up2_fwd(const void* x, void* y, int64_t F, int H, int W, int C, int align_corners, void* stream);
int stg_bilinear_up2_bwd(const void* dy, void* dx, int64_t F, int H, int W, int C, int align_corners, void* stream);
/* BatchNorm over the rows of [R, C] (nn.BatchNorm3d of TPAVI's W_z, TPAVI.py:57-61).  stg_bn_colsum accumulates (atomically, into
 * a zeroed out[2][C]) mode 0: sum x, sum x^2;  mode 2: sum (x - mean), sum (x - mean)^2 (the variance is taken in this second,
 * centred pass: E[x^2] - mean^2 cancels catastrophically on feature maps whose mean dwarfs their spread);  mode 1: sum dy,
 * sum dy * xhat (a = x, b = dy).  stg_bn_apply: y = (x - mean) rstd
 * gamma + beta.  stg_bn_bwd: dx = gamma rstd (dy - sums[0]/R - xhat sums[1]/R), or gamma rstd dy when sums == NULL (eval). */
/* LayerNorm weight / bias gradients over many rows of bf16 [R, C] (TPAVI's trainable norm_layer, TPAVI.py:30,149: up to 500 K rows):
 * dgamma[c] += sum_r dy (x - mean[r]) rstd[r], dbeta[c] += sum_r dy; block-folded, one atomic per (block, channel). */
int stg_ln_param_grad(const void* dy, const void* x, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                      int64_t R, int C, void* stream);
int stg_bn_colsum(const void* a, const void* b, const float* mean, const float* rstd, float* out, int64_t R, int C, int mode, void* stream);
int stg_bn_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y, int64_t R, int C, void* stream);
int stg_bn_bwd(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma, const float* sums, void* dx,
               int64_t R, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
