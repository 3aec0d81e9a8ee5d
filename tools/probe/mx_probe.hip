// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, E8M0 block scales) on gfx950: which (row, k) each lane's 32 operand
// bytes hold and which 32-wide k-block a lane's scale byte applies to.  Exact small-magnitude data, checked against a host fp64
// reference; prints PASS / FAIL per layout hypothesis.  Build: hipcc --offload-arch=gfx950 -O2 mx_probe.hip -o mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void mx_kernel(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
// opsel variant: scale taken from byte 1 of the scale registers
__global__ void mx_kernel_sel1(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 1, sa[l], 1, sb[l]);
    c[l] = acc;
}

static float e4m3_decode(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? std::ldexp((float)m / 8.f, -6) : std::ldexp(1.f + (float)m / 8.f, e - 7);
    return s ? -f : f;
}

int main() {
    srand(1234);
    uint8_t A[16][128], B[128][16];   // A[row][k], B[k][col] as e4m3 codes
    int SA[16][4], SB[16][4];         // E8M0 exponents per (row / col, k-block of 32)
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) {
        // codes with exponent field 5..9 (|value| in [0.25, 7.5]) and random sign / mantissa, some zeros
        auto code = [&]() -> uint8_t { if (rand() % 7 == 0) return 0; return (uint8_t)(((rand() & 1) << 7) | ((5 + rand() % 5) << 3) | (rand() & 7)); };
        A[i][k] = code(); B[k][i] = code();
    }
    for (int i = 0; i < 16; ++i) for (int kb = 0; kb < 4; ++kb) { SA[i][kb] = 124 + rand() % 7; SB[i][kb] = 124 + rand() % 7; }

    auto run = [&](int hyp, bool use_scales, bool sel1) -> double {
        std::vector<v8i> ha(64), hb(64); std::vector<int> hsa(64), hsb(64);
        for (int l = 0; l < 64; ++l) {
            uint8_t ba[32], bb[32];
            for (int j = 0; j < 32; ++j) {
                int k;
                if (hyp == 1) k = 32 * (l >> 4) + j;
                else if (hyp == 2) k = 16 * (l >> 4) + (j & 15) + 64 * (j >> 4);
                else k = 8 * (l >> 4) + (j & 7) + 32 * (j >> 3);
                ba[j] = A[l & 15][k]; bb[j] = B[k][l & 15];
            }
            for (int w = 0; w < 8; ++w) {
                ha[l][w] = (int)(ba[4 * w] | (ba[4 * w + 1] << 8) | (ba[4 * w + 2] << 16) | ((uint32_t)ba[4 * w + 3] << 24));
                hb[l][w] = (int)(bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | ((uint32_t)bb[4 * w + 3] << 24));
            }
            const int ea = use_scales ? SA[l & 15][l >> 4] : 127, eb = use_scales ? SB[l & 15][l >> 4] : 127;
            hsa[l] = sel1 ? ((ea << 8) | 0x33) : (ea | 0x5500);       // garbage in the unselected byte
            hsb[l] = sel1 ? ((eb << 8) | 0x44) : (eb | 0x6600);
        }
        v8i *da, *db; v4f* dc; int *dsa, *dsb;
        hipMalloc(&da, 64 * sizeof(v8i)); hipMalloc(&db, 64 * sizeof(v8i)); hipMalloc(&dc, 64 * sizeof(v4f));
        hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
        hipMemcpy(da, ha.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
        if (sel1) hipLaunchKernelGGL(mx_kernel_sel1, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        else hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        std::vector<v4f> hc(64);
        hipMemcpy(hc.data(), dc, 64 * sizeof(v4f), hipMemcpyDeviceToHost);
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dsa); hipFree(dsb);
        double worst = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;       // C/D map of the 16x16 shapes: col = lane & 15, row = 4 * (lane >> 4) + reg
            double ref = 0;
            for (int kb = 0; kb < 4; ++kb) {
                double part = 0;
                for (int k = 32 * kb; k < 32 * kb + 32; ++k) part += (double)e4m3_decode(A[row][k]) * (double)e4m3_decode(B[k][col]);
                ref += part * (use_scales ? std::ldexp(1.0, SA[row][kb] - 127 + SB[col][kb] - 127) : 1.0);
            }
            worst = std::fmax(worst, std::fabs(ref - (double)hc[l][r]));
        }
        return worst;
    };
    // ---- which (row, k-block) does lane L's A-scale byte act on?  All scales 1 except ONE lane's A scale = 2^3; the outputs that
    // change, and by how much (7 x one k-block's partial sum), identify the association.
    {
        auto raw = [&](int hot_lane, int hot_exp, bool on_b, float out[16][16]) {
            std::vector<v8i> ha(64), hb(64); std::vector<int> hsa(64, 127), hsb(64, 127);
            for (int l = 0; l < 64; ++l) for (int w = 0; w < 8; ++w) {
                uint32_t xa = 0, xb = 0;
                for (int q = 0; q < 4; ++q) { const int k = 32 * (l >> 4) + 4 * w + q; xa |= (uint32_t)A[l & 15][k] << (8 * q); xb |= (uint32_t)B[k][l & 15] << (8 * q); }
                ha[l][w] = (int)xa; hb[l][w] = (int)xb;
            }
            if (hot_lane >= 0) (on_b ? hsb : hsa)[hot_lane] = hot_exp;
            v8i *da, *db; v4f* dc; int *dsa, *dsb;
            hipMalloc(&da, 64 * sizeof(v8i)); hipMalloc(&db, 64 * sizeof(v8i)); hipMalloc(&dc, 64 * sizeof(v4f)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
            hipMemcpy(da, ha.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice);
            hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
            std::vector<v4f> hc(64);
            hipMemcpy(hc.data(), dc, 64 * sizeof(v4f), hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) out[4 * (l >> 4) + r][l & 15] = hc[l][r];
            hipFree(da); hipFree(db); hipFree(dc); hipFree(dsa); hipFree(dsb);
        };
        float base[16][16], hot[16][16];
        raw(-1, 127, false, base);
        for (int on_b = 0; on_b < 2; ++on_b)
            for (int L : {0, 16, 32, 48, 5, 21}) {
                raw(L, 130, on_b, hot);
                printf("%s-scale of lane %2d = 2^3 -> changed outputs:", on_b ? "B" : "A", L);
                int nch = 0;
                for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (hot[i][j] != base[i][j]) {
                    // which k-block explains the delta?
                    int kbm = -1;
                    for (int kb = 0; kb < 4; ++kb) {
                        double part = 0;
                        for (int k = 32 * kb; k < 32 * kb + 32; ++k) part += (double)e4m3_decode(A[i][k]) * (double)e4m3_decode(B[k][j]);
                        if (std::fabs((double)hot[i][j] - (double)base[i][j] - 7.0 * part) < 1e-2 * (1 + std::fabs(part))) kbm = kb;
                    }
                    if (nch < 6) printf(" (r%d,c%d,kb%d)", i, j, kbm);
                    ++nch;
                }
                printf("  [%d changed]\n", nch);
                if (!on_b) {
                    const int i = L & 15, j = 3;
                    printf("    row %d col %d: delta/7 = %.4f; partials per 32-k block:", i, j, ((double)hot[i][j] - base[i][j]) / 7.0);
                    for (int kb = 0; kb < 4; ++kb) {
                        double part = 0;
                        for (int k = 32 * kb; k < 32 * kb + 32; ++k) part += (double)e4m3_decode(A[i][k]) * (double)e4m3_decode(B[k][j]);
                        printf(" %.4f", part);
                    }
                    printf("; per 16-k block:");
                    for (int kb = 0; kb < 8; ++kb) {
                        double part = 0;
                        for (int k = 16 * kb; k < 16 * kb + 16; ++k) part += (double)e4m3_decode(A[i][k]) * (double)e4m3_decode(B[k][j]);
                        printf(" %.4f", part);
                    }
                    printf("; base %.4f\n", base[i][j]);
                }
            }
    }
    for (int hyp = 1; hyp <= 3; ++hyp) {
        const double e = run(hyp, false, false);
        printf("layout hypothesis %d (unit scales): max abs err %.6g -> %s\n", hyp, e, e < 1e-2 ? "PASS (any consistent k permutation does)" : "FAIL");
    }
    for (int hyp = 1; hyp <= 3; ++hyp) {
        const double es = run(hyp, true, false), es1 = run(hyp, true, true);
        printf("hypothesis %d with per-lane E8M0 scales (lane 16 g + i: row i, logical k-block g): opsel 0 max abs err %.6g -> %s; opsel 1 %.6g -> %s\n",
               hyp, es, es < 2e-2 ? "PASS" : "FAIL", es1, es1 < 2e-2 ? "PASS" : "FAIL");
    }
    return 0;
}
