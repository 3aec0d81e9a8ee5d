"""GEMM micro-benchmark over the shapes of the Swin-B AVE step (B=32): TFLOP/s and GB/s per shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K

def bench(M, N, K_, reps=5, **kw):
    A = torch.randn(M, K_, device="cuda").bfloat16(); W = (torch.randn(N, K_, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(2): K.gemm_nt(A, W, b, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): K.gemm_nt(A, W, b, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * N * K_
    by = 2.0 * (M * K_ + N * K_ + M * N)
    print(f"M={M:8d} N={N:5d} K={K_:5d}  {ms:8.3f} ms  {fl/ms/1e9:8.1f} TF/s  {by/ms/1e6:8.1f} GB/s", flush=True)

R0 = 2 * 320 * 3136
shapes = [(R0, 384, 128), (R0, 128, 128), (R0, 512, 128), (R0, 128, 512), (R0 // 2, 16, 128), (R0 // 2, 128, 16),
          (R0 // 4, 768, 256), (R0 // 4, 256, 256), (R0 // 4, 1024, 256), (R0 // 4, 256, 1024),
          (R0 // 16, 1536, 512), (R0 // 16, 512, 512), (R0 // 16, 2048, 512), (R0 // 16, 512, 2048), (R0 // 32, 32, 512),
          (R0 // 64, 3072, 1024), (R0 // 64, 1024, 1024), (R0 // 64, 4096, 1024), (R0 // 64, 1024, 4096),
          (8192, 8192, 8192), (4096, 4096, 4096)]
for s in shapes:
    bench(*s)
