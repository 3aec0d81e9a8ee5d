// Probe: what bounds the store phase of a 256 x 256 bf16 GEMM tile on gfx950 (the un-overlapped epilogue of gemm_nt_8ph_kernel:
// 128 KiB per tile in ~8.9 us = ~7 B/clk/CU)?  One 512-thread workgroup per tile (one per CU at a time, like the GEMM: the kernel
// declares 128 KiB of LDS), 128 wave-stores of 16 B per lane per tile, in several shapes; optional busy work between stores.
//   shape 0: 8 rows x 128 B per wave-instruction (what the row-layout epilogue does: a wave owns a 64-column strip)
//   shape 1: 2 rows x 512 B      shape 2: 4 rows x 256 B      shape 3: 1 KiB contiguous (tile-major output: the ideal)
//   shape 4: like 0 with non-temporal stores                  shape 5: like 1 with non-temporal stores
// Output matrix: M x N bf16, N = 2048 (row pitch 4096 B), tiles in the GEMM's XCD-chunked order.  Prints us per launch, TB/s and
// B/clk/CU (at 2.1 GHz) per shape.  Build: hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int SHAPE>
__global__ void __launch_bounds__(512, 1) store_kernel(uint4* out, int nbm, int nbn, int64_t pitch16, int spin) {
    extern __shared__ uint8_t smem[];
    const int nblk = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) smem[0] = 1;                                      // keep the LDS allocation alive
    uint4 v = make_uint4(tid, bid, 0x3f803f80u, 0x3f803f80u);
    uint4* base = out + (int64_t)bm * 256 * pitch16 + bn * 32;       // tile origin in 16-byte units (256 columns = 32 units)
    float busy = (float)tid;
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {                                // 16 stores per wave = 128 per tile
        int row, c16;
        if (SHAPE == 0 || SHAPE == 4) {                              // wave strip: columns wave&3 (x 64) + 128 * (..), rows (wave>>2)*128 + ...
            row = (wave >> 2) * 128 + it * 8 + (lane >> 3);
            c16 = (wave & 3) * 8 + (lane & 7);
        } else if (SHAPE == 1 || SHAPE == 5) {
            row = wave * 32 + it * 2 + (lane >> 5);
            c16 = lane & 31;
        } else if (SHAPE == 2) {
            row = wave * 32 + (it >> 1) * 4 + (lane >> 4);
            c16 = (it & 1) * 16 + (lane & 15);
        } else {
            row = 0; c16 = 0;
        }
        uint4* p = (SHAPE == 3) ? out + ((int64_t)bid * 128 + wave * 16 + it) * 64 + lane : base + (int64_t)row * pitch16 + c16;
        v.z += it;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        if (SHAPE == 4 || SHAPE == 5) __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p)); else *p = v;
        for (int s = 0; s < spin; ++s) busy = busy * 1.0001f + 0.5f;  // optional VALU work between stores
    }
    if (busy == 12345.678f) smem[1] = 2;
}

// nwg workgroups (<= 256: one per CU), each writing `per` tiles back to back in shape 0: is the store rate a per-CU limit?
__global__ void __launch_bounds__(512, 1) store_some_kernel(uint4* out, int per, int64_t pitch16) {
    extern __shared__ uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) smem[0] = 1;
    uint4 v = make_uint4(tid, blockIdx.x, 0x3f803f80u, 0x3f803f80u);
#pragma unroll 1
    for (int t = 0; t < per; ++t) {
        const int bid = blockIdx.x * per + t;
        uint4* base = out + (int64_t)(bid >> 3) * 256 * pitch16 + (bid & 7) * 32;
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            const int row = (wave >> 2) * 128 + it * 8 + (lane >> 3), c16 = (wave & 3) * 8 + (lane & 7);
            v.z += it;
            base[(int64_t)row * pitch16 + c16] = v;
        }
    }
}

template <int SHAPE>
static float run(uint4* out, int nbm, int nbn, int64_t pitch16, int spin, int reps) {
    hipFuncSetAttribute((const void*)store_kernel<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<SHAPE>, dim3(nbm * nbn), dim3(512), 128 * 1024, 0, out, nbm, nbn, pitch16, spin);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<SHAPE>, dim3(nbm * nbn), dim3(512), 128 * 1024, 0, out, nbm, nbn, pitch16, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const int M = 125440, N = 2048;
    const int nbm = M / 256, nbn = N / 256;
    uint4* out;
    hipMalloc(&out, (size_t)M * N * 2);
    const int64_t pitch16 = N * 2 / 16;
    const double bytes = (double)M * N * 2;
    for (int spin = 0; spin <= 200; spin += 200) {
        float t[6] = {run<0>(out, nbm, nbn, pitch16, spin, 20), run<1>(out, nbm, nbn, pitch16, spin, 20), run<2>(out, nbm, nbn, pitch16, spin, 20),
                      run<3>(out, nbm, nbn, pitch16, spin, 20), run<4>(out, nbm, nbn, pitch16, spin, 20), run<5>(out, nbm, nbn, pitch16, spin, 20)};
        const char* nm[6] = {"8 rows x 128 B", "2 rows x 512 B", "4 rows x 256 B", "1 KiB contiguous", "8 x 128 B nt", "2 x 512 B nt"};
        for (int s = 0; s < 6; ++s)
            printf("spin %3d  %-18s %8.1f us  %6.2f TB/s  %5.1f B/clk/CU  (%.1f us per tile per CU)\n", spin, nm[s], t[s], bytes / t[s] * 1e-6,
                   bytes / t[s] * 1e-6 * 1e12 / 256 / 2.1e9, t[s] / (nbm * nbn / 256.0));
    }
    hipFuncSetAttribute((const void*)store_some_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int nwg = 16; nwg <= 256; nwg *= 2) {
        const int per = 3920 / 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(store_some_kernel, dim3(nwg), dim3(512), 128 * 1024, 0, out, per, pitch16);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_some_kernel, dim3(nwg), dim3(512), 128 * 1024, 0, out, per, pitch16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 10 * 1e3, by = (double)nwg * per * 131072;
        printf("persistent, %3d workgroups x %d tiles: %7.1f us  %5.2f TB/s  %5.1f B/clk/CU  (%.2f us per tile per CU)\n", nwg, per, us, by / us * 1e-6,
               by / us * 1e-6 * 1e12 / nwg / 2.1e9, us / per);
    }
    return hipGetLastError() != hipSuccess;
}
