"""Time vs problem size for the HBM-bound stage-2 kernels (C = 512): is the distance from the chip's copy rate a RATE (slope) or a fixed cost per
launch (intercept: ramp-up, tail, turnover)?  Rows sweep 1/8 x .. 4 x the step's 125 440; a least-squares line through (bytes, us) per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa
from stgcma import kernels as K, ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def timeit(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fit(pts):
    xs = torch.tensor([p[0] for p in pts], dtype=torch.float64)
    ys = torch.tensor([p[1] for p in pts], dtype=torch.float64)
    A = torch.stack([xs, torch.ones_like(xs)], 1)
    sol = torch.linalg.lstsq(A, ys[:, None]).solution[:, 0]
    return float(sol[0]), float(sol[1])


C, Kd = 512, 32
imgs = [80, 160, 320, 640, 1280, 2560]
res = {}
for images in imgs:
    M = images * 196
    h = torch.randn(M, Kd, device=dev).to(BF16)
    w = (torch.randn(C, Kd, device=dev) * 0.1).to(BF16)
    b = torch.randn(C, device=dev)
    r32 = torch.randn(M, C, device=dev)
    r16 = torch.randn(M, C, device=dev).to(BF16)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    x = torch.empty(M, C, device=dev)
    t = timeit(lambda: K.up_ln_fwd(h, w, b, r32, ga, be, res16=r16, out=x))
    res.setdefault("upln_fwd", []).append((M * C * (4 + 4 + 2 + 2) + M * Kd * 2, t))
    dy = torch.randn(M, C, device=dev).to(BF16)
    ad = torch.randn(M, C, device=dev).to(BF16)
    mean, rstd = r32.mean(1), (r32.var(1, unbiased=False) + 1e-5).rsqrt()
    wt = (torch.randn(Kd, C, device=dev) * 0.1).to(BF16)
    t = timeit(lambda: K.ln_bwd_down(dy, r32, ga, mean, rstd, wt, add_to=ad))
    res.setdefault("ln_bwd_down(y-form)", []).append((M * C * (2 + 4 + 2 + 2) + M * Kd * 2, t))
    del r32, r16, x, dy, ad
    heads = 16
    qkv = torch.randn(M, 3 * C, device=dev).to(BF16)
    dO = torch.randn(M, C, device=dev).to(BF16)
    table = torch.randn(169, heads, device=dev)
    co = torch.stack(torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")).flatten(1)
    rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0) + 6
    index = (rel[:, :, 0] * 13 + rel[:, :, 1]).reshape(-1).to(dev)
    bm, bmT = K.winattn_table(table, index, None, 49)
    wg = K.WinGeom(images, heads, 14, 14, 7, 0, 32 ** -0.5, bm, bmT)
    O, lse = K.winattn_fwd(wg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:])
    d = torch.empty_like(qkv)
    U = M * C * 2
    t = timeit(lambda: K.winattn_fwd(wg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out=O))
    res.setdefault("winattn_fwd", []).append((4 * U, t))
    t = timeit(lambda: K.winattn_bwd(wg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], O, lse, dO, dQ=d[:, :C], dK=d[:, C:2 * C], dV=d[:, 2 * C:]))
    res.setdefault("winattn_bwd", []).append((8 * U, t))
    # plain copy of the same bytes for the chip's rate on this box
    src = torch.empty(2 * U, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    t = timeit(lambda: dst.copy_(src))
    res.setdefault("aten_copy", []).append((4 * U, t))
    del qkv, dO, d, O, src, dst
    torch.cuda.empty_cache()

for name, pts in res.items():
    slope, icpt = fit(pts)
    line = "  ".join(f"{b / 1e6:7.0f}MB:{t:7.1f}us({b / t / 1e6:4.2f})" for b, t in pts)
    print(f"{name:22s} asymptotic {1 / slope / 1e6:5.2f} TB/s, fixed {icpt:6.1f} us | {line}", flush=True)
