"""Times stg_wgrad_tn_ws_multi on the adapters' weight-gradient shapes (n problems per launch):  python tools/wgrad_multi_bench.py [reps]
(profiles/r05b_wgrad_split_sweep.txt was taken with a temporary override of the row split in csrc/wgrad.hip: S = -1 the round-5a rule, 0 the chooser.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa: F401
from stgcma import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
# (M rows, narrow J, wide C, problems): Swin-B stage 2 / 0 / 3, Swin-L stage 2, ViT-B
SHAPES = [(62720, 32, 512, 12), (62720, 32, 512, 8), (1003520, 16, 128, 12), (15680, 64, 1024, 12), (62720, 96, 768, 12), (62720, 96, 768, 8),
          (63040, 48, 768, 12)]
for M, J, C, n in SHAPES:
    probs = []
    for i in range(n):
        narrow_is_dy = i % 2 == 0
        a = torch.randn(M, J, device=dev).bfloat16()
        b = torch.randn(M, C, device=dev).bfloat16()
        dy, x = (a, b) if narrow_is_dy else (b, a)
        dW = torch.zeros(dy.shape[1], x.shape[1], device=dev)
        db = torch.zeros(dy.shape[1], device=dev)
        probs.append((dy, x, dW, db, None, 1, 1))
    K.wgrad_tn_multi(probs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.wgrad_tn_multi(probs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    by = n * M * (J + C) * 2
    print(f"M{M} J{J} C{C} n{n}: {us:8.1f} us  {by / us / 1e6:5.2f} TB/s", flush=True)
