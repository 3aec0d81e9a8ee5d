"""Is the 8-phase main loop waiting for HBM?  Same tile count per CU, A either streaming from HBM (larger than the 256 MB MALL) or resident
across back-to-back launches (smaller than it): TFLOP/s per shape, shipped library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
for (M, N, Kd) in ((131072, 512, 2048), (32768, 512, 2048), (16384, 512, 2048), (131072, 512, 512), (32768, 512, 512), (65536, 2048, 8192), (8192, 2048, 8192), (65536, 1024, 4096), (16384, 1024, 4096)):
    A = torch.randn(M, Kd, device="cuda").bfloat16(); W = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    best = 1e9
    for rnd in range(3):
        for _ in range(3): K.gemm_nt(A, W, None, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): K.gemm_nt(A, W, None, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    tiles = ((M + 255) // 256) * (N // 256)
    print(f"M={M:7d} N={N:5d} K={Kd:5d}  A {M*Kd*2/1e6:7.1f} MB  {tiles/256:5.2f} tiles/CU  {best:8.1f} us  {best/(tiles/256):7.2f} us per tile round  {2.0*M*N*Kd/best/1e6:7.1f} TFLOP/s  [{K.LAST_GEMM_KERNEL}]", flush=True)
