import os, sys
sys.path.insert(0, os.getcwd())
import torch, stgcma
from stgcma import kernels as K
for (M, N1, N2) in ((125440, 256, 512), (501760, 256, 256), (2007040, 256, 128), (31360, 256, 1024)):
    dY = torch.randn(M, N1, device="cuda").bfloat16(); X = torch.randn(M, N2, device="cuda").bfloat16()
    dW = torch.zeros(N1, N2, device="cuda"); db = torch.zeros(N1, device="cuda")
    K.wgrad_tn(dY, X, dW, db); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.wgrad_tn(dY, X, dW, db)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"wide wgrad M{M} {N1}x{N2}: {us:8.1f} us  {2.0*M*N1*N2/us/1e6:6.1f} TFLOP/s  {M*(N1+N2)*2/us/1e3:6.0f} GB/s", flush=True)
