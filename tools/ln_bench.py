"""LayerNorm forward (fp32 rows -> bf16) at the backbones' shapes:  python tools/ln_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa: F401
from stgcma import kernels as K

dev = "cuda:0"
for rows, C in ((2007040, 128), (2007040, 192), (501760, 256), (501760, 384), (125440, 512), (125440, 768), (31360, 1024), (31360, 1536)):
    x = torch.randn(rows, C, device=dev)
    g = torch.rand(C, device=dev) + 0.5
    b = torch.randn(C, device=dev)
    y, _, _ = K.layernorm_fwd(x, g, b, want_stats=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.layernorm_fwd(x, g, b, want_stats=False, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"rows {rows:8d} C {C:5d}: {us:7.1f} us  {rows * C * 6 / us / 1e6:5.2f} TB/s", flush=True)
