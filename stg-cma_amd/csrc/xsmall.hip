// The cross-modal PAIR of the CLIP-ViT blocks (CLIP_AVE.py:386-398: h_v' = softmax(h_v h_a^T) h_a, h_a' = softmax(h_a h_v^T) h_v per frame; 197 video and
// 49 audio tokens, adapter width 48 for ViT-B/16, 64 for ViT-L) as ONE workgroup per frame with both modalities' rows resident in LDS -- forward (both
// directions) in one launch, backward (both modalities' whole gradients) in one launch.
//
// Why its own family (round 6): the frames are far too small for the flash kernels' one-wave-per-query-tile grids -- the generic kernels spent 25 us on
// the forward and 27 + 45 us on the backward of 2.4 GFLOP (0.016 of the MFMA peak: 640 long waves for the 49-query direction, each walking 197 keys
// alone), 2.3 ms of a 64 ms step in 96 launches.  Here a frame's 246 rows x D are loaded once (24-31 KB), the nine (direction, 32-query tile) tasks of
// the forward are dealt to the four waves by cost, and the backward is the merged pass of mha.hip's mha_bwdm_kernel (the two directions share
// S = X_v X_a^T; G_x = dQ of x's own direction + dK + dV of the other: three score-type and two output-type products per tile pair) on LDS-resident
// operands, delta computed in the kernel.
//
// MFMA v_mfma_f32_32x32x16_bf16, scores transposed (rows = keys, lane = query) as in mha.hip; LDS rows are D + 8 elements (16-byte-multiple pitch,
// conflict-free for the row fragments at 112 / 144 bytes); D = 48 computes its second 32-column output tile on 16 real + 16 don't-care columns (the
// transposed reads run into the next row; those accumulator rows are never stored).
#include <math.h>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

struct XsP {
    const bf16_t* Xv; const bf16_t* Xa; int64_t ldv, lda;
    bf16_t* Ov; bf16_t* Oa; int64_t ldov, ldoa;
    float* lse_v; float* lse_a;       // [P, nv], [P, na]: log2 domain (max + log2 sum)
    int P, nv, na, nvt, nat;          // tokens and 32-row tiles per frame
    float scale, scale2;
    // backward
    const bf16_t* dOv; const bf16_t* dOa; int64_t lddov, lddoa;
    bf16_t* Gv; bf16_t* Ga; int64_t ldgv, ldga;
};

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define ACC_ROW(reg, hh) (((reg) & 3) + 8 * ((reg) >> 2) + 4 * (hh))

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t pack8(const float* x) {
    const u32x4_t w = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
    return __builtin_bit_cast(bf16x8_t, w);
}
typedef short s4_t __attribute__((ext_vector_type(4)));
// transposed fragment of a [32 rows][DP] tile: A[i = d][k slot j] = tile[16 s2 + 4 hh + (j & 3) + 8 (j >> 2)][32 dt + d]
template <int DP>
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* s, int dt, int s2, int hh, int d) {
    const int gi = d & 15, c = d >> 4;
    const bf16_t* p = s + (16 * s2 + 4 * hh + (gi >> 2)) * DP + 32 * dt + 16 * c + 4 * (gi & 3);
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)p);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)(p + 8 * DP));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
// natural fragment (rows on the lane): tile[r][16 s + 8 hh .. + 7]
template <int DP>
__device__ __forceinline__ bf16x8_t nat_frag(const bf16_t* s, int r, int hh, int ks) {
    return *reinterpret_cast<const bf16x8_t*>(s + r * DP + 16 * ks + 8 * hh);
}
// store_tile32 (common.h) for a tile of which only the first `cols` (16 or 32) columns exist
__device__ __forceinline__ void store_tile_cols(bf16_t* rowp, const f32x16_t& acc, float sc, int hh, bool ok, int cols) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t a0 = pack_bf2(acc[8 * i + 0] * sc, acc[8 * i + 1] * sc), a1 = pack_bf2(acc[8 * i + 2] * sc, acc[8 * i + 3] * sc);
        const uint32_t b0 = pack_bf2(acc[8 * i + 4] * sc, acc[8 * i + 5] * sc), b1 = pack_bf2(acc[8 * i + 6] * sc, acc[8 * i + 7] * sc);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        if (ok && 16 * i < cols) *reinterpret_cast<uint4*>(rowp + 16 * i + 8 * hh) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

// rows [0, n) of a frame's [n, D] slab -> LDS rows of pitch DP; rows [n, rows_pad) are zero-filled (their scores are masked, their products vanish)
template <int D>
__device__ __forceinline__ void load_rows(bf16_t* s, const bf16_t* g, int64_t ld, int n, int rows_pad, int tid) {
    constexpr int DP = D + 8, CH = D / 8;
    for (int id = tid; id < rows_pad * CH; id += 256) {
        const int row = id / CH, c = id - row * CH;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < n) v = *reinterpret_cast<const uint4*>(g + (int64_t)row * ld + 8 * c);
        *reinterpret_cast<uint4*>(s + row * DP + 8 * c) = v;
    }
}

// which (direction, query tile) tasks wave w runs: the audio-query tiles (each walks ALL video key tiles) go to waves 0 .. nat - 1, the video-query
// tiles (each walks the nat audio key tiles) round-robin over the other waves.  task id: 0 .. nvt - 1 = video tile, nvt .. nvt + nat - 1 = audio tile.
__device__ __forceinline__ bool wave_task(int wave, int k, int nvt, int nat, int& task) {
    if (wave < nat) {
        if (k > 0) return false;
        task = nvt + wave;
        return true;
    }
    const int nvw = 4 - nat;
    const int t = (wave - nat) + k * nvw;
    if (t >= nvt) return false;
    task = t;
    return true;
}

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ void __launch_bounds__(256) xsmall_fwd_kernel(XsP a) {
    constexpr int KS = D / 16, DT = (D + 31) / 32, DP = D + 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];          // sV [32 nvt + 1][DP], sA [32 nat + 1][DP]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int p = blockIdx.x;
    bf16_t* sV = smem;
    bf16_t* sA = sV + (32 * a.nvt + 1) * DP;
    load_rows<D>(sV, a.Xv + (int64_t)p * a.nv * a.ldv, a.ldv, a.nv, 32 * a.nvt + 1, tid);
    load_rows<D>(sA, a.Xa + (int64_t)p * a.na * a.lda, a.lda, a.na, 32 * a.nat + 1, tid);
    __syncthreads();
    for (int k = 0; k < 8; ++k) {
        int task;
        if (!wave_task(wave, k, a.nvt, a.nat, task)) break;
        const bool qa = task >= a.nvt;                                     // queries = audio rows, keys = values = video rows
        const int qt = qa ? task - a.nvt : task;
        const bf16_t* sQ = qa ? sA : sV;
        const bf16_t* sK = qa ? sV : sA;
        const int nq = qa ? a.na : a.nv, nk = qa ? a.nv : a.na, nkt = qa ? a.nvt : a.nat;
        const int q = 32 * qt + r;
        bf16x8_t qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = nat_frag<DP>(sQ + 32 * qt * DP, r, hh, s);
        f32x16_t o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
        float m = NEG_BIG, l = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            const bf16_t* tK = sK + 32 * kt * DP;
            f32x16_t sc = zero16();                                        // S^T[key][q]
#pragma unroll
            for (int s = 0; s < KS; ++s) sc = MFMA32(nat_frag<DP>(tK, r, hh, s), qf[s], sc);
            float x[16];
            float mr = NEG_BIG;
            const int kbase = 32 * kt;
            const bool tail = kbase + 32 > nk;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                x[reg] = sc[reg];
                if (tail && kbase + ACC_ROW(reg, hh) >= nk) x[reg] = NEG_BIG;
                mr = fmaxf(mr, x[reg]);
            }
            mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
            const float mx = fmaxf(m, mr * a.scale2);
            const float alpha = __builtin_amdgcn_exp2f(m - mx);
            m = mx;
            const float nmx = -mx;
            float ls = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                x[reg] = __builtin_amdgcn_exp2f(fmaf(x[reg], a.scale2, nmx));
                ls += x[reg];
            }
            ls += __shfl_xor(ls, 32, 64);
            l = l * alpha + ls;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) o[dt][reg] *= alpha;
            const bf16x8_t p0 = pack8(x), p1 = pack8(x + 8);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[dt] = MFMA32(tr_frag<DP>(tK, dt, 0, hh, r), p0, o[dt]);
                o[dt] = MFMA32(tr_frag<DP>(tK, dt, 1, hh, r), p1, o[dt]);
            }
        }
        const float inv = 1.0f / l;
        bf16_t* O = qa ? a.Oa : a.Ov;
        const int64_t ldo = qa ? a.ldoa : a.ldov;
        bf16_t* op = O + ((int64_t)p * nq + (q < nq ? q : nq - 1)) * ldo;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store_tile_cols(op + 32 * dt, o[dt], inv, hh, q < nq, D - 32 * dt);
        float* lse = qa ? a.lse_a : a.lse_v;
        if (q < nq && hh == 0) lse[(int64_t)p * nq + q] = m + __log2f(l);
    }
}

// ------------------------------------------------------------------------------------------------ backward (merged: see mha.hip mha_bwdm_kernel)
template <int D>
__global__ void __launch_bounds__(256) xsmall_bwd_kernel(XsP a) {
    constexpr int KS = D / 16, DT = (D + 31) / 32, DP = D + 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];          // sV, sA, sdV, sdA tiles + statistics
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int p = blockIdx.x;
    const int rv = 32 * a.nvt + 1, ra = 32 * a.nat + 1;
    bf16_t* sV = smem;
    bf16_t* sA = sV + rv * DP;
    bf16_t* sdV = sA + ra * DP;
    bf16_t* sdA = sdV + rv * DP;
    float* nlv = reinterpret_cast<float*>(sdA + ra * DP + 8);             // -lse_v [32 nvt], delta_v, -lse_a [32 nat], delta_a
    float* dlv = nlv + 32 * a.nvt;
    float* nla = dlv + 32 * a.nvt;
    float* dla = nla + 32 * a.nat;
    load_rows<D>(sV, a.Xv + (int64_t)p * a.nv * a.ldv, a.ldv, a.nv, rv, tid);
    load_rows<D>(sA, a.Xa + (int64_t)p * a.na * a.lda, a.lda, a.na, ra, tid);
    load_rows<D>(sdV, a.dOv + (int64_t)p * a.nv * a.lddov, a.lddov, a.nv, rv, tid);
    load_rows<D>(sdA, a.dOa + (int64_t)p * a.na * a.lddoa, a.lddoa, a.na, ra, tid);
    // statistics of both directions: -lse, and delta = rowsum(dO o O) (O from global memory: read once, here)
    for (int row = tid; row < 32 * (a.nvt + a.nat); row += 256) {
        const bool isa = row >= 32 * a.nvt;
        const int i = isa ? row - 32 * a.nvt : row;
        const int n = isa ? a.na : a.nv;
        float nl = 0.f, dl = 0.f;
        if (i < n) {
            const bf16_t* o = isa ? a.Oa + ((int64_t)p * n + i) * a.ldoa : a.Ov + ((int64_t)p * n + i) * a.ldov;
            const bf16_t* d = isa ? a.dOa + ((int64_t)p * n + i) * a.lddoa : a.dOv + ((int64_t)p * n + i) * a.lddov;
            nl = -(isa ? a.lse_a : a.lse_v)[(int64_t)p * n + i];
#pragma unroll
            for (int c = 0; c < D / 8; ++c) {
                const bf16x8_t x = *reinterpret_cast<const bf16x8_t*>(o + 8 * c), y = *reinterpret_cast<const bf16x8_t*>(d + 8 * c);
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += bf2f((bf16_t)x[j]) * bf2f((bf16_t)y[j]);
            }
        }
        (isa ? nla : nlv)[i] = nl;
        (isa ? dla : dlv)[i] = dl;
    }
    __syncthreads();
    const bool unit_scale = a.scale == 1.0f;
    for (int k = 0; k < 8; ++k) {
        int task;
        if (!wave_task(wave, k, a.nvt, a.nat, task)) break;
        const bool xa = task >= a.nvt;                                     // this pass owns audio rows (X = audio, Y = video)
        const int qt = xa ? task - a.nvt : task;
        const bf16_t* sX = (xa ? sA : sV) + 32 * qt * DP;
        const bf16_t* sdX = (xa ? sdA : sdV) + 32 * qt * DP;
        const bf16_t* sY = xa ? sV : sA;
        const bf16_t* sdY = xa ? sdV : sdA;
        const float* nly = xa ? nlv : nla;
        const float* dly = xa ? dlv : dla;
        const int nx = xa ? a.na : a.nv, ny = xa ? a.nv : a.na, nyt = xa ? a.nvt : a.nat;
        const int q = 32 * qt + r;
        bf16x8_t xf[KS], dxf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { xf[s] = nat_frag<DP>(sX, r, hh, s); dxf[s] = nat_frag<DP>(sdX, r, hh, s); }
        const float nlse_q = (xa ? nla : nlv)[q], ndel_q = -(xa ? dla : dlv)[q];
        f32x16_t g[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) g[dt] = zero16();
        for (int kt = 0; kt < nyt; ++kt) {
            const int jbase = 32 * kt;
            const bool tail = jbase + 32 > ny;
            const bf16_t* tY = sY + jbase * DP;
            const bf16_t* tD = sdY + jbase * DP;
            f32x16_t st = zero16(), dpx, dpy = zero16();                   // S^T[j][i], (dO_x Y^T)^T[j][i] - delta_x[i], (dO_y X^T)[j][i]
#pragma unroll
            for (int i = 0; i < 16; ++i) dpx[i] = ndel_q;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8_t yf = nat_frag<DP>(tY, r, hh, s);
                st = MFMA32(yf, xf[s], st);
                dpx = MFMA32(yf, dxf[s], dpx);
                dpy = MFMA32(nat_frag<DP>(tD, r, hh, s), xf[s], dpy);
            }
            float ds[16], py[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 ls = *reinterpret_cast<const float4*>(nly + jbase + 8 * g4 + 4 * hh);
                const float4 de = *reinterpret_cast<const float4*>(dly + jbase + 8 * g4 + 4 * hh);
                const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int reg = 4 * g4 + c;
                    float px = __builtin_amdgcn_exp2f(fmaf(st[reg], a.scale2, nlse_q));     // P^x[i, j]
                    float pv = __builtin_amdgcn_exp2f(fmaf(st[reg], a.scale2, lsv[c]));     // P^y[j, i]
                    if (tail && jbase + 8 * g4 + 4 * hh + c >= ny) { px = 0.f; pv = 0.f; }
                    py[reg] = pv;
                    const float d = fmaf(px, dpx[reg], pv * (dpy[reg] - dev[c]));
                    ds[reg] = unit_scale ? d : d * a.scale;
                }
            }
            const bf16x8_t d0 = pack8(ds), d1 = pack8(ds + 8), p0 = pack8(py), p1 = pack8(py + 8);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                g[dt] = MFMA32(tr_frag<DP>(tY, dt, 0, hh, r), d0, g[dt]);
                g[dt] = MFMA32(tr_frag<DP>(tY, dt, 1, hh, r), d1, g[dt]);
                g[dt] = MFMA32(tr_frag<DP>(tD, dt, 0, hh, r), p0, g[dt]);
                g[dt] = MFMA32(tr_frag<DP>(tD, dt, 1, hh, r), p1, g[dt]);
            }
        }
        bf16_t* G = xa ? a.Ga : a.Gv;
        const int64_t ldg = xa ? a.ldga : a.ldgv;
        bf16_t* op = G + ((int64_t)p * nx + (q < nx ? q : nx - 1)) * ldg;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store_tile_cols(op + 32 * dt, g[dt], 1.0f, hh, q < nx, D - 32 * dt);
    }
}

int fwd_lds(int nvt, int nat, int D) { return ((32 * nvt + 1) + (32 * nat + 1)) * (D + 8) * 2; }
int bwd_lds(int nvt, int nat, int D) { return 2 * fwd_lds(nvt, nat, D) + 16 + 2 * 32 * (nvt + nat) * 4; }

int fill(const stg_xsmall_args* f, XsP& p, const char* who) {
    STG_CHECK(f != nullptr && f->Xv && f->Xa && f->Ov && f->Oa && f->lse_v && f->lse_a, -1, "%s: null pointer", who);
    STG_CHECK(stg_xsmall_supported(f->nv, f->na, f->D), -2, "%s: unsupported geometry nv=%d na=%d D=%d (nv <= 256, na <= 64, D in {32, 48, 64})", who, f->nv, f->na, f->D);
    STG_CHECK(f->P >= 0 && f->P < (1 << 30) && f->scale > 0.f, -2, "%s: bad P / scale", who);
    STG_CHECK(f->ldv % 8 == 0 && f->lda % 8 == 0 && f->ldov % 8 == 0 && f->ldoa % 8 == 0 && f->ldv >= f->D && f->lda >= f->D && f->ldov >= f->D && f->ldoa >= f->D, -2,
              "%s: leading dimensions must be multiples of 8 and >= D", who);
    STG_CHECK((((uintptr_t)f->Xv | (uintptr_t)f->Xa | (uintptr_t)f->Ov | (uintptr_t)f->Oa) & 15) == 0, -2, "%s: misaligned pointers", who);
    p.Xv = (const bf16_t*)f->Xv; p.Xa = (const bf16_t*)f->Xa; p.ldv = f->ldv; p.lda = f->lda;
    p.Ov = (bf16_t*)f->Ov; p.Oa = (bf16_t*)f->Oa; p.ldov = f->ldov; p.ldoa = f->ldoa;
    p.lse_v = f->lse_v; p.lse_a = f->lse_a;
    p.P = f->P; p.nv = f->nv; p.na = f->na; p.nvt = (f->nv + 31) / 32; p.nat = (f->na + 31) / 32;
    p.scale = f->scale; p.scale2 = f->scale * LOG2E;
    return 0;
}

}  // namespace

extern "C" int stg_xsmall_supported(int nv, int na, int D) {
    return nv >= 1 && na >= 1 && nv <= 256 && na <= 64 && (D == 32 || D == 48 || D == 64) ? 1 : 0;
}

extern "C" int stg_xsmall_fwd(const stg_xsmall_args* f, void* stream) {
    XsP p = {};
    const int rc = fill(f, p, "stg_xsmall_fwd");
    if (rc) return rc;
    if (p.P == 0) return 0;
    const int lds = fwd_lds(p.nvt, p.nat, f->D);
    static std::atomic<uint64_t> d32{0}, d48{0}, d64{0};
    hipStream_t st = (hipStream_t)stream;
#define STG_XS(DD, flag) { STG_CHECK(stg_reserve_lds(xsmall_fwd_kernel<DD>, fwd_lds(8, 2, DD), flag), -101, "stg_xsmall_fwd: cannot reserve LDS"); \
                           hipLaunchKernelGGL(xsmall_fwd_kernel<DD>, dim3(p.P), dim3(256), lds, st, p); }
    if (f->D == 32) STG_XS(32, d32) else if (f->D == 48) STG_XS(48, d48) else STG_XS(64, d64)
#undef STG_XS
    STG_LAUNCH_CHECK();
    return 0;
}

extern "C" int stg_xsmall_bwd(const stg_xsmall_args* f, const void* dOv, const void* dOa, int64_t lddov, int64_t lddoa, void* Gv, void* Ga,
                              int64_t ldgv, int64_t ldga, void* stream) {
    XsP p = {};
    const int rc = fill(f, p, "stg_xsmall_bwd");
    if (rc) return rc;
    STG_CHECK(dOv && dOa && Gv && Ga, -1, "stg_xsmall_bwd: null pointer");
    STG_CHECK(lddov % 8 == 0 && lddoa % 8 == 0 && ldgv % 8 == 0 && ldga % 8 == 0 && lddov >= f->D && lddoa >= f->D && ldgv >= f->D && ldga >= f->D, -2,
              "stg_xsmall_bwd: leading dimensions must be multiples of 8 and >= D");
    STG_CHECK((((uintptr_t)dOv | (uintptr_t)dOa | (uintptr_t)Gv | (uintptr_t)Ga) & 15) == 0, -2, "stg_xsmall_bwd: misaligned pointers");
    if (p.P == 0) return 0;
    p.dOv = (const bf16_t*)dOv; p.dOa = (const bf16_t*)dOa; p.lddov = lddov; p.lddoa = lddoa;
    p.Gv = (bf16_t*)Gv; p.Ga = (bf16_t*)Ga; p.ldgv = ldgv; p.ldga = ldga;
    const int lds = bwd_lds(p.nvt, p.nat, f->D);
    static std::atomic<uint64_t> d32{0}, d48{0}, d64{0};
    hipStream_t st = (hipStream_t)stream;
#define STG_XS(DD, flag) { STG_CHECK(stg_reserve_lds(xsmall_bwd_kernel<DD>, bwd_lds(8, 2, DD), flag), -101, "stg_xsmall_bwd: cannot reserve LDS"); \
                           hipLaunchKernelGGL(xsmall_bwd_kernel<DD>, dim3(p.P), dim3(256), lds, st, p); }
    if (f->D == 32) STG_XS(32, d32) else if (f->D == 48) STG_XS(48, d48) else STG_XS(64, d64)
#undef STG_XS
    STG_LAUNCH_CHECK();
    return 0;
}
