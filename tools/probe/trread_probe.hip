// Probe of ds_read_b64_tr_b16 lane mapping for the xattn kernels (dev aid):  hipcc --offload-arch=gfx950 -o trread_probe trread_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
// tile[32 keys][16 d] of shorts, value = key*100 + d.  Lane (r = l&31, hh = l>>5), c = r>>4, i = l&15 = 4q+p.
// Want: lane gets, for (s2, t), elements e=0..3 = tile[16*s2 + 8*t + 4*hh + e][d = r & 15]
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short tile[32 * 16];
    const int l = threadIdx.x;
    for (int x = l; x < 512; x += 64) tile[x] = (short)((x / 16) * 100 + (x % 16));
    __syncthreads();
    const int hh = l >> 5, i = l & 15, q = i >> 2, p = i & 3;
    for (int s2 = 0; s2 < 2; ++s2)
        for (int t = 0; t < 2; ++t) {
            const short* addr = tile + (16 * s2 + 8 * t + 4 * hh + q) * 16 + 4 * p;
            s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
            for (int e = 0; e < 4; ++e) out[((s2 * 2 + t) * 64 + l) * 4 + e] = v[e];
        }
}
int main() {
    short* d; hipMalloc(&d, 4 * 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[4 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s2 = 0; s2 < 2; ++s2) for (int t = 0; t < 2; ++t) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        const int hh = l >> 5, r = l & 31;
        const int want = (16 * s2 + 8 * t + 4 * hh + e) * 100 + (r & 15);
        const int got = h[((s2 * 2 + t) * 64 + l) * 4 + e];
        if (want != got) { if (bad < 10) printf("s2=%d t=%d lane=%d e=%d want %d got %d\n", s2, t, l, e, want, got); ++bad; }
    }
    printf("trread probe: %d mismatches\n", bad);
    return bad != 0;
}
