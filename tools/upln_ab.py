"""upln_fwd / ln_bwd_down (x-hat form) at the stage-2 shape, for same-box A/B runs of two builds of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma  # noqa
from stgcma import kernels as k
dev = "cuda"; BF16 = torch.bfloat16
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, C, K in [(125440, 512, 32), (501760, 256, 32), (2007040, 128, 16)]:
    h = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(C, K, device=dev) * 0.1).to(BF16); b = torch.randn(C, device=dev)
    r32 = torch.randn(M, C, device=dev); r16 = torch.randn(M, C, device=dev).to(BF16)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev); x = torch.empty(M, C, device=dev)
    for use16 in (True, False):
        us = t(lambda: k.up_ln_fwd(h, w, b, r32, ga, be, res16=r16 if use16 else None, out=x))
        byt = M * C * (4 + 4 + 2 + (2 if use16 else 0)) + M * K * 2
        print(f"upln_fwd M={M} C={C} K={K} res16={use16}: {us:8.1f} us ({byt / us / 1e6:5.2f} TB/s)", flush=True)
    dy = torch.randn(M, C, device=dev).to(BF16); ad = torch.randn(M, C, device=dev).to(BF16)
    xh = torch.randn(M, C, device=dev).to(BF16); rstd = torch.rand(M, device=dev) + 0.5
    wt = (torch.randn(K, C, device=dev) * 0.1).to(BF16)
    us = t(lambda: k.ln_bwd_down_xhat(dy, xh, rstd, wt, add_to=ad))
    byt = M * C * (2 + 2 + 2 + 2) + M * K * 2
    print(f"ln_bwd_down_xhat M={M} C={C} J={K}: {us:8.1f} us ({byt / us / 1e6:5.2f} TB/s)", flush=True)
