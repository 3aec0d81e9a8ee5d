"""Timing ablations of the fused MLP kernel (diagnostics build):  STGCMA_LIB=stg-cma_amd/libstgcma_hip_diag.so python tools/mlp_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K, _lib
dev = "cuda"
rows, C_ = 2007040, 128
g = torch.Generator(device=dev).manual_seed(0)
Y = torch.randn(rows, C_, generator=g, device=dev).bfloat16()
W1 = (torch.randn(4 * C_, C_, generator=g, device=dev) / C_ ** 0.5).bfloat16()
W2p = (torch.randn(C_, 4 * C_, generator=g, device=dev) / (4 * C_) ** 0.5).bfloat16()
b1 = torch.randn(4 * C_, generator=g, device=dev) * 0.1
b2 = torch.randn(C_, generator=g, device=dev) * 0.1
out = torch.empty_like(Y)
def t(fn, iters=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
names = {0: "full", 1: "identity instead of GELU", 2: "no second product", 3: "no first product", 4: "no barrier / DMA", 5: "no output stores"}
for d in range(6):
    _lib.check(_lib.lib().stg_set_option(b"gemm_dbg", d), "opt")
    print(f"diag {d} ({names[d]}): {t(lambda: K.mlp_fwd(Y, W1, b1, W2p, b2, out=out)):8.1f} us")
