// Shared device helpers for the gfx950 (CDNA4) kernels of libstgcma_hip.so.
// Wave = 64 lanes everywhere; bf16 is carried as raw uint16_t in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#define STG_WAVE 64

typedef uint16_t bf16_t;

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator

struct __attribute__((aligned(16))) u16x8 { uint16_t v[8]; };
struct __attribute__((aligned(8))) u16x4 { uint16_t v[4]; };

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: a plain vector cast lowers to ONE v_cvt_pk_bf16_f32 per pair on
// gfx950 (integer-arithmetic rounding costs ~6 VALU per element and dominated the attention kernels' issue slots)
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_hw_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}

__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    float s = 1.0f / (1.0f + __expf(-1.702f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

enum { STG_ACT_NONE = 0, STG_ACT_GELU = 1, STG_ACT_QUICKGELU = 2 };

// GELU and its derivative together, ~16 VALU instead of erff + expf (the GEMM epilogues were VALU-bound on them):
// Phi(x) = 1 - phi(x) (b1 t + .. + b5 t^5), t = 1 / (1 + 0.2316419 |x|)  (Abramowitz-Stegun 7.1.26 in erfc form,
// |error| < 5e-7 on x Phi(x) and on its derivative -- fp32 round-off level, checked against scipy erf over [-12, 12]);
// the Gaussian e^{-x^2/2} it needs is the derivative's x phi(x) term as well.
__device__ __forceinline__ void gelu_fast(float x, float& y, float& dy) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.231641888f, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
    const float q = 0.5f * poly * t * e;
    const float phi = x >= 0.f ? 1.0f - q : q;
    dy = fmaf(x * e, 0.3989422804014327f, phi);
    y = x * phi;
}
// GELU for bf16 destinations WITHOUT transcendentals (round 3; the logistic form it replaces, Phi ~ sigmoid(a1 x + a3 x^3 + a5 x^5),
// cost 7 VALU + v_exp + v_rcp per value, and the two quarter-rate instructions were 8 of its 15 issue slots):
//     Phi(x)   ~ 0.5 + xc P(xc^2),   gelu'(x) = Phi(x) + x phi(x) ~ 0.5 + xc Q(xc^2),   xc = clamp(x, -4.5, 4.5)
// P (9 terms) and Q (8 terms) are weighted minimax fits (Lawson iteration, tools/gelu_fit.py) of the odd parts on [0, 4.5]:
// |x Phi(x) error| <= 5.2e-5 over all x -- the level of the form it replaces, 1/35 of a bf16 half-ulp at 1 -- and |gelu' error|
// <= 8.8e-4 (a fifth of the 8-bit derivative code's step; bf16 resolves 3.9e-3 at 1).  Beyond the clamp Phi is 1 - 3.4e-6 / 3.4e-6.
// All multiply-adds: on PAIRS of values v_pk_mul_f32 / v_pk_fma_f32 carry two fp32 lanes per issue slot, so value + derivative cost
// 1 (clamp) + 9.5 packed = 10.5 slots per value against 15, the value alone 6.5 against 12.  The fc1 epilogues and the fused MLP
// kernels are VALU-bound on this function (10^9 values per stage-0 block).  The scalar and the packed forms run the same operations
// in the same order: bit-identical.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t pk_splat(float v) { return (f32x2_t){v, v}; }
__device__ __forceinline__ float gp_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f32x2_t gp_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float gp_k(float, float v) { return v; }
__device__ __forceinline__ f32x2_t gp_k(f32x2_t, float v) { return (f32x2_t){v, v}; }
__device__ __forceinline__ float gp_clamp(float x) { return __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f); }
__device__ __forceinline__ f32x2_t gp_clamp(f32x2_t x) { return (f32x2_t){__builtin_amdgcn_fmed3f(x.x, -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x.y, -4.5f, 4.5f)}; }
template <typename T>
__device__ __forceinline__ T gelu_pw_phi(T xc, T s) {         // Phi(x) from the clamped argument and its square
    T p = gp_fma(s, gp_k(s, 2.759560758e-11f), gp_k(s, -3.063619494e-09f));
    p = gp_fma(p, s, gp_k(s, 1.496385803e-07f));
    p = gp_fma(p, s, gp_k(s, -4.260212331e-06f));
    p = gp_fma(p, s, gp_k(s, 7.918093045e-05f));
    p = gp_fma(p, s, gp_k(s, -1.022331839e-03f));
    p = gp_fma(p, s, gp_k(s, 9.528348002e-03f));
    p = gp_fma(p, s, gp_k(s, -6.588462659e-02f));
    p = gp_fma(p, s, gp_k(s, 3.986567907e-01f));
    return gp_fma(xc, p, gp_k(s, 0.5f));
}
template <typename T>
__device__ __forceinline__ T gelu_pw_dphi(T xc, T s) {        // gelu'(x)
    T q = gp_fma(s, gp_k(s, -7.272945256e-09f), gp_k(s, 6.534896567e-07f));
    q = gp_fma(q, s, gp_k(s, -2.478124810e-05f));
    q = gp_fma(q, s, gp_k(s, 5.185177887e-04f));
    q = gp_fma(q, s, gp_k(s, -6.576145592e-03f));
    q = gp_fma(q, s, gp_k(s, 5.220721148e-02f));
    q = gp_fma(q, s, gp_k(s, -2.566600037e-01f));
    q = gp_fma(q, s, gp_k(s, 7.945217645e-01f));
    return gp_fma(xc, q, gp_k(s, 0.5f));
}
template <typename T>
__device__ __forceinline__ T gelu_pw(T x) {
    const T xc = gp_clamp(x);
    return x * gelu_pw_phi(xc, xc * xc);
}
template <typename T>
__device__ __forceinline__ T gelu_pw_grad(T x) {
    const T xc = gp_clamp(x);
    return gelu_pw_dphi(xc, xc * xc);
}
template <typename T>
__device__ __forceinline__ void gelu_pw_both(T x, T& y, T& dy) {
    const T xc = gp_clamp(x);
    const T s = xc * xc;
    dy = gelu_pw_dphi(xc, s);
    y = x * gelu_pw_phi(xc, s);
}
__device__ __forceinline__ void quick_gelu_fast(float x, float& y, float& dy) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.4554669595930157f));   // sigmoid(1.702 x)
    dy = s * fmaf(1.702f * x, 1.0f - s, 1.0f);
    y = x * s;
}
__device__ __forceinline__ void act_both(int act, float x, float& y, float& dy) {
    if (act == STG_ACT_GELU) gelu_fast(x, y, dy);
    else if (act == STG_ACT_QUICKGELU) quick_gelu_fast(x, y, dy);
    else { y = x; dy = 1.0f; }
}

__device__ __forceinline__ float act_apply(int act, float x) {
    if (act == STG_ACT_GELU) return gelu_erf(x);
    if (act == STG_ACT_QUICKGELU) return quick_gelu(x);
    return x;
}
__device__ __forceinline__ float act_grad(int act, float x) {
    if (act == STG_ACT_GELU) return gelu_erf_grad(x);
    if (act == STG_ACT_QUICKGELU) return quick_gelu_grad(x);
    return 1.0f;
}

template <int W>
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int W>
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Store of one 32-token tile of an O^T / dQ^T / dK^T / dV^T accumulator: lane (token r, half hh) holds head dims
// 8*g4 + 4*hh + e in acc[4*g4 + e], i.e. four 8-byte pieces of the token's 64-byte row.  Two v_permlane32_swap per pair of
// pieces (lanes r and r + 32 trade their middle pieces) leave each lane with 8 CONSECUTIVE dims: two 16-byte stores per
// tile instead of four 8-byte ones (the store tail of these kernels is issue-bound).  Must run with every lane active;
// only the store itself is predicated.
__device__ __forceinline__ void store_tile32(bf16_t* rowp, const f32x16_t& acc, float sc, int hh, bool ok) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t a0 = pack_bf2(acc[8 * i + 0] * sc, acc[8 * i + 1] * sc), a1 = pack_bf2(acc[8 * i + 2] * sc, acc[8 * i + 3] * sc);
        const uint32_t b0 = pack_bf2(acc[8 * i + 4] * sc, acc[8 * i + 5] * sc), b1 = pack_bf2(acc[8 * i + 6] * sc, acc[8 * i + 7] * sc);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);     // -> {[A.lo, B.lo], [A.hi, B.hi]} by half
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        if (ok) *reinterpret_cast<uint4*>(rowp + 16 * i + 8 * hh) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

// host-side error plumbing (api.cpp)
void stg_set_error(const char* fmt, ...);

// Dispatch options (A/B knobs of tools/ and of the tests): process-wide integers defined in api.cpp and written only by
// stg_set_option -- nothing on a launch path reads the environment.
// EIGHT names since round 6 (sixteen knobs were sixteen configurations the suite ran a handful of times): the retired ones selected kernels
// that no longer ship (round-1 window / temporal attention, one-tile mha forward / dQ, the simple-loop 256 x 256 GEMM) or A/B arms that were
// settled (gemm_epi, gemm_ktail, ln_fit, mha_nw).
extern std::atomic<int> stg_opt_gemm_8ph;     // 8-phase kernel: 0 off, 1 auto (default), 2 every legal shape, 3 long-K shapes only
extern std::atomic<int> stg_opt_gemm_8phm;    // multi-tile 8-phase kernel: 0 off, 1 auto (default), n >= 2: at most n column tiles per workgroup
extern std::atomic<int> stg_opt_gemm_nx;      // 8-phase kernels for N % 64 == 0 / K % 64 == 0 shapes (NX forms, round 6): 0 off, 1 (default) the classes measured faster (N < 256, K >= 768), 2 every legal shape
extern std::atomic<int> stg_opt_gemm_d8m;     // the byte-derivative-source epilogue (fc2 dgrad) on the 8-phase kernels: 0 off, 1 K <= 512, 2 (default) K <= 1024
extern std::atomic<int> stg_opt_gemm_dbg;     // diagnostics build only (-DSTG_GEMM_DIAG)
extern std::atomic<int> stg_opt_xattn;        // 0: frame-global cross-modal attention on the generic attention kernels
extern std::atomic<int> stg_opt_wgrad_plan;   // wgrad_ws row splits: 0 = the round-1 rule (512 / column groups); 1 = per-launch chooser for nt1 > 2 only; 2 = for every width (default)
extern std::atomic<int> stg_opt_upln_cap;     // workgroups per launch of the wide (NW >= 8) join kernels of upln.hip: 256 = one round (default), 2048 = rounds 1-5a

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE setting: `done` (one static per kernel instantiation) remembers the
// devices that have it, so a model on cuda:1 (nn.DataParallel replicas, one process driving several GPUs) gets it too.
template <typename Kern>
inline bool stg_reserve_lds(Kern kern, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}
#define STG_CHECK(cond, code, ...)                \
    do {                                          \
        if (!(cond)) {                            \
            stg_set_error(__VA_ARGS__);           \
            return (code);                        \
        }                                         \
    } while (0)
#define STG_LAUNCH_CHECK()                                                   \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            stg_set_error("launch failed: %s", hipGetErrorString(e__));      \
            return -100;                                                     \
        }                                                                    \
    } while (0)
