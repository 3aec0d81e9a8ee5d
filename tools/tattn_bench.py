"""Temporal attention kernels (head dim 32, T = 10) at the four Swin-B stage shapes, B = 32 clips x 2 modalities:
python tools/tattn_bench.py   (STG_TATTN_KERNELS=0/1 selects the round-1 / coalesced kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
dev = "cuda"
def t(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, T = int(os.environ.get("CLIPS", 32)), 10
for res, heads in ((56, 4), (28, 8), (14, 16), (7, 32)):
    C, N = heads * 32, res * res
    rows = 2 * B * T * N
    qkv = torch.randn(rows, 3 * C, device=dev).bfloat16()
    dO = torch.randn(rows, C, device=dev).bfloat16()
    bias = (torch.randn(2, heads, T * T, device=dev) * 0.5).contiguous()
    tg = K.TGeom(2, B, T, N, heads, 32 ** -0.5, bias)
    O = K.tattn_fwd(tg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:])
    d = torch.empty_like(qkv)
    db = torch.zeros(2, heads, T * T, device=dev)
    f = t(lambda: K.tattn_fwd(tg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]))
    b = t(lambda: K.tattn_bwd(tg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], dO, dQ=d[:, :C], dK=d[:, C:2 * C], dV=d[:, 2 * C:], dbias=db))
    U = rows * C * 2
    print(f"res {res:2d} heads {heads:2d}: fwd {f:7.1f} us ({4*U/f/1e6:5.2f} TB/s)  bwd {b:7.1f} us ({7*U/b/1e6:5.2f} TB/s)")
