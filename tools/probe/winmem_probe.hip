// Memory-pattern probe for the window-attention kernels (round 4): the loads and stores of winattn_fwd1_kernel at the step's stage-2
// geometry (640 images of 14 x 14 tokens, C = 512 = 16 heads x 32, fused qkv rows of 1536 bf16, shifted 7 x 7 windows), WITHOUT the
// arithmetic, in several lane -> address maps that move the same bytes:
//   HP = 1: a wave loads its own head's 64-byte row pieces (16 rows per LDS-DMA instruction)           -- the shipped kernels
//   HP = 2: two waves share two heads: 128-byte pieces, 8 rows per instruction
//   HP = 4: the four waves of a workgroup share its four heads: 256-byte pieces, 4 rows per instruction
//   HP = 4, WPW = 2/4: the workgroup walks 2 / 4 windows (more bytes in flight per workgroup)
// and with / without the 16 KiB additive-table read per wave.  Build: hipcc -O3 --offload-arch=gfx950 winmem_probe.hip -o winmem_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int IMG = 14, WS = 7, SHIFT = 3, NW = 4, NTOK = 49, H = 16, C = 512, LD = 1536;

__device__ __forceinline__ int64_t tok_row(int pg, int g, int i) {
    i = i < NTOK ? i : NTOK - 1;
    const int wi = g / 2, wj = g % 2, ti = i / WS, tj = i % WS;
    int h = wi * WS + ti + SHIFT; h = h >= IMG ? h - IMG : h;
    int w = wj * WS + tj + SHIFT; w = w >= IMG ? w - IMG : w;
    return (int64_t)pg * (IMG * IMG) + h * IMG + w;
}

// one workgroup (4 waves) = one window x 4 consecutive heads; LDS per workgroup: 3 tensors x 64 rows x 256 B = 48 KiB
template <int HP, bool TABLE, bool LOADS, bool STORES>
__global__ void __launch_bounds__(256, 2) probe_kernel(const uint16_t* qkv, uint16_t* out, const float* table, int total_wg, float* sink) {
    __shared__ __attribute__((aligned(16))) uint16_t smem[3 * 64 * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wgi = blockIdx.x;
    if (wgi >= total_wg) return;
    const int hq = wgi % (H / 4), p = wgi / (H / 4);
    const int pg = p / NW, g = p % NW;
    float4 tb[16];
    if (TABLE) {
        const float* t = table + ((int64_t)(g * H + hq * 4 + wave)) * 4096 + 4 * (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) tb[i] = *reinterpret_cast<const float4*>(t + 256 * (i * 2 + (lane >> 5)) % 4096);
    }
    // loads: per tensor 64 (padded) rows x 256 B for the workgroup = 1024 16-byte chunks = 4 instructions per wave
    constexpr int LPR = 4 * HP;                  // lanes per row piece
    constexpr int RPI = 64 / LPR;                // rows per instruction
    if (LOADS) {
#pragma unroll
        for (int ten = 0; ten < 3; ++ten)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // HP = 1: wave w owns head w, instruction i covers rows 16 i .. 16 i + 15
                // HP = 2: waves (2 k, 2 k + 1) own heads 2 k, 2 k + 1; wave parity picks the instruction's row block
                // HP = 4: all waves own the 4 heads; (wave, i) picks the row block
                int row, colb;
                if (HP == 1) { row = 16 * i + (lane >> 2); colb = wave * 64 + (lane & 3) * 16; }
                else if (HP == 2) { row = 8 * (2 * i + (wave & 1)) + (lane >> 3); colb = (wave >> 1) * 128 + (lane & 7) * 16; }
                else { row = 4 * (4 * i + wave) + (lane >> 4); colb = (lane & 15) * 16; }
                const int64_t off = tok_row(pg, g, row) * LD + ten * C + hq * 128;      // bf16 elements
                const char* src = reinterpret_cast<const char*>(qkv + off) + colb;
                uint16_t* dst = smem + ten * 64 * 128 + (HP == 1 ? (wave * 4 + i) * 512 : HP == 2 ? ((wave >> 1) * 8 + 2 * i + (wave & 1)) * 512 : (4 * i + wave) * 512);
                if (row < NTOK || HP == 1)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (TABLE) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += tb[i].x + tb[i].y + tb[i].z + tb[i].w;
        if (s == 123.456f) sink[0] = s;
    }
    if (STORES) {
        // O: 49 rows x 256 B per workgroup, same maps
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row, colb;
            if (HP == 1) { row = 16 * i + (lane >> 2); colb = wave * 64 + (lane & 3) * 16; }
            else if (HP == 2) { row = 8 * (2 * i + (wave & 1)) + (lane >> 3); colb = (wave >> 1) * 128 + (lane & 7) * 16; }
            else { row = 4 * (4 * i + wave) + (lane >> 4); colb = (lane & 15) * 16; }
            if (row < NTOK) {
                const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(smem) + (lane + 64 * (wave * 4 + i)) * 16);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out + tok_row(pg, g, row) * C + hq * 128) + colb) = v;
            }
        }
    }
}

template <int HP, bool TABLE, bool LOADS, bool STORES>
float run(const uint16_t* qkv, uint16_t* out, const float* table, int total_wg, float* sink, int rounds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < rounds + 1; ++r) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((probe_kernel<HP, TABLE, LOADS, STORES>), dim3(total_wg), dim3(256), 0, 0, qkv, out, table, total_wg, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (r) ts.push_back(ms * 100.f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <int HP>
__global__ void __launch_bounds__(256, 2) probe_bwd_kernel(const uint16_t* qkv, const uint16_t* o2, uint16_t* dqkv, int total_wg) {
    __shared__ __attribute__((aligned(16))) uint16_t smem[3 * 64 * 128];       // re-used per tensor group (no arithmetic: contents irrelevant)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wgi = blockIdx.x;
    if (wgi >= total_wg) return;
    const int hq = wgi % (H / 4), p = wgi / (H / 4);
    const int pg = p / NW, g = p % NW;
#pragma unroll
    for (int ten = 0; ten < 5; ++ten)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row, colb;
            if (HP == 1) { row = 16 * i + (lane >> 2); colb = wave * 64 + (lane & 3) * 16; }
            else if (HP == 2) { row = 8 * (2 * i + (wave & 1)) + (lane >> 3); colb = (wave >> 1) * 128 + (lane & 7) * 16; }
            else { row = 4 * (4 * i + wave) + (lane >> 4); colb = (lane & 15) * 16; }
            const int64_t r = tok_row(pg, g, row);
            const char* src = ten < 3 ? reinterpret_cast<const char*>(qkv + r * LD + ten * C + hq * 128) + colb
                                      : reinterpret_cast<const char*>(o2 + ((int64_t)(ten - 3) * gridDim.x / (H / 4) / NW * IMG * IMG + r) * C + hq * 128) + colb;
            uint16_t* dst = smem + (ten % 3) * 64 * 128 + (4 * i + wave) * 512;
            if (row < NTOK || HP == 1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ten = 0; ten < 3; ++ten)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row, colb;
            if (HP == 1) { row = 16 * i + (lane >> 2); colb = wave * 64 + (lane & 3) * 16; }
            else if (HP == 2) { row = 8 * (2 * i + (wave & 1)) + (lane >> 3); colb = (wave >> 1) * 128 + (lane & 7) * 16; }
            else { row = 4 * (4 * i + wave) + (lane >> 4); colb = (lane & 15) * 16; }
            if (row < NTOK) {
                const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(smem) + (lane + 64 * (wave * 4 + i)) * 16 + ten * 16384);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dqkv + tok_row(pg, g, row) * LD + ten * C + hq * 128) + colb) = v;
            }
        }
}

template <int HP>
float run_bwd(const uint16_t* qkv, const uint16_t* o2, uint16_t* dqkv, int total_wg, int rounds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < rounds + 1; ++r) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((probe_bwd_kernel<HP>), dim3(total_wg), dim3(256), 0, 0, qkv, o2, dqkv, total_wg);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (r) ts.push_back(ms * 100.f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const int images = argc > 1 ? atoi(argv[1]) : 640;
    const int64_t rows = (int64_t)images * IMG * IMG;
    uint16_t *qkv, *out; float *table, *sink;
    CK(hipMalloc(&qkv, rows * LD * 2)); CK(hipMalloc(&out, rows * C * 2)); CK(hipMalloc(&table, NW * H * 4096 * 4)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(qkv, 1, rows * LD * 2)); CK(hipMemset(out, 0, rows * C * 2)); CK(hipMemset(table, 0, NW * H * 4096 * 4));
    const int total_wg = images * NW * (H / 4);
    const double bl = (double)rows * LD * 2, bs = (double)rows * C * 2;
    printf("images %d rows %lld: loads %.0f MB stores %.0f MB, %d workgroups\n", images, (long long)rows, bl / 1e6, bs / 1e6, total_wg);
#define RUN(HP, T, L, S, what, bytes) { float us = run<HP, T, L, S>(qkv, out, table, total_wg, sink, 5); printf("  %-58s %7.1f us  %5.2f TB/s\n", what, us, (bytes) / us / 1e6); }
    RUN(1, true, true, true, "HP=1 (64 B pieces) table + loads + stores", bl + bs)
    RUN(1, false, true, true, "HP=1 loads + stores", bl + bs)
    RUN(2, false, true, true, "HP=2 (128 B pieces) loads + stores", bl + bs)
    RUN(4, false, true, true, "HP=4 (256 B pieces) loads + stores", bl + bs)
    RUN(4, true, true, true, "HP=4 table + loads + stores", bl + bs)
    RUN(1, false, true, false, "HP=1 loads only", bl)
    RUN(2, false, true, false, "HP=2 loads only", bl)
    RUN(4, false, true, false, "HP=4 loads only", bl)
    RUN(1, false, false, true, "HP=1 stores only", bs)
    RUN(2, false, false, true, "HP=2 stores only", bs)
    RUN(4, false, false, true, "HP=4 stores only", bs)
    {
        uint16_t *o2, *dqkv;
        CK(hipMalloc(&o2, 2 * rows * C * 2)); CK(hipMalloc(&dqkv, rows * LD * 2));
        CK(hipMemset(o2, 1, 2 * rows * C * 2)); CK(hipMemset(dqkv, 0, rows * LD * 2));
        const double bb = (double)rows * C * 2 * 8;
        float us = run_bwd<1>(qkv, o2, dqkv, total_wg, 5); printf("  %-58s %7.1f us  %5.2f TB/s\n", "backward pattern HP=1: 5 loads + 3 stores", us, bb / us / 1e6);
        us = run_bwd<2>(qkv, o2, dqkv, total_wg, 5); printf("  %-58s %7.1f us  %5.2f TB/s\n", "backward pattern HP=2", us, bb / us / 1e6);
        us = run_bwd<4>(qkv, o2, dqkv, total_wg, 5); printf("  %-58s %7.1f us  %5.2f TB/s\n", "backward pattern HP=4", us, bb / us / 1e6);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
