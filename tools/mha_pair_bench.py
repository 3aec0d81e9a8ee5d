"""Backward of a cross-modal pair on mha.hip: two-direction path (mha_bwd_pair: dQ + dK/dV per direction) vs the merged pass (mha_bwd_pair_merged):
python tools/mha_pair_bench.py     (Swin-L stage 0 frame-global 320 x 3136 x 96, stage 1, and the 49-token window pairs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma  # noqa
from stgcma import kernels as K


def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, geo, rows in (("frame 3136 x 96 (P 320)", K.MhaGeom(320, 1, 3136, 96, 1.0), 320 * 3136), ("frame 784 x 96 (P 320)", K.MhaGeom(320, 1, 784, 96, 1.0), 320 * 784),
                        ("frame 196 x 96 (P 320)", K.MhaGeom(320, 1, 196, 96, 1.0), 320 * 196),
                        ("windows 49 x 96 (320 frames of 56 x 56)", K.MhaGeom(320 * 64, 1, 49, 96, 1.0, window=(56, 56, 7, 3)), 320 * 3136),
                        ("frame 49 x 64 (Swin-B stage 3, P 320)", K.MhaGeom(320, 1, 49, 64, 1.0), 320 * 49)):
    D = geo.D
    X, Y = (torch.randn(rows, D, device="cuda") * 0.3).bfloat16(), (torch.randn(rows, D, device="cuda") * 0.3).bfloat16()
    dX, dY = torch.randn(rows, D, device="cuda").bfloat16(), torch.randn(rows, D, device="cuda").bfloat16()
    (r0, l0), (r1, l1) = K.mha_fwd_pair(geo, (X, Y, Y), (Y, X, X))
    a0, a1, b0, b1 = (torch.empty_like(X) for _ in range(4))
    two = t(lambda: K.mha_bwd_pair(geo, (X, Y, Y, r0, l0, dX, a0, a1, None), (Y, X, X, r1, l1, dY, b0, b1, None)))
    mer = t(lambda: K.mha_bwd_pair_merged(geo, (X, Y, r0, l0, dX), (Y, X, r1, l1, dY)))
    print(f"{name:44s} two-direction {two:9.1f} us   merged {mer:9.1f} us   ({mer / two:.2f} x)", flush=True)
