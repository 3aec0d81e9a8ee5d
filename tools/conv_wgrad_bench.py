"""Times stg_conv3x3_wgrad at the AVS decoder's shapes (B = 8 clips x T = 5 frames): python tools/conv_wgrad_bench.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa: F401
from stgcma import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1          # index into SHAPES (PMC passes time one shape)
dev = "cuda:0"
SHAPES = [(40, 112, 112, 256, 256, 1), (40, 56, 56, 256, 256, 1), (40, 28, 28, 256, 256, 1), (40, 14, 14, 256, 256, 1),
          (40, 112, 112, 256, 128, 6), (40, 224, 224, 128, 256, 1), (40, 448, 448, 32, 128, 1)]
for si, (F, H, W, O, I, d) in enumerate(SHAPES):
    if only >= 0 and si != only:
        continue
    M = F * H * W
    x = torch.randn(M, I, device=dev).bfloat16()
    dy = torch.randn(M, O, device=dev).bfloat16()
    K.conv3x3_wgrad(dy, x, F, H, W, d, want_db=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.conv3x3_wgrad(dy, x, F, H, W, d, want_db=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * M * O * 9 * I
    print(f"F{F} {H}x{W} O{O} I{I} d{d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  alg {(M * (O + I) * 2) / us / 1e3:6.0f} GB/s", flush=True)
