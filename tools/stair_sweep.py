"""Fine sweep of the row count around the stage-2 size for the C = 512 join + LayerNorm kernel: a staircase (tail quantisation) or a line?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa
from stgcma import kernels as K
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
C, Kd = 512, 32


def timeit(fn, n=20):
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


Mmax = 64 * 768 * 4
h = torch.randn(Mmax, Kd, device=dev).to(BF16)
w = (torch.randn(C, Kd, device=dev) * 0.1).to(BF16)
b = torch.randn(C, device=dev)
r32 = torch.randn(Mmax, C, device=dev)
r16 = torch.randn(Mmax, C, device=dev).to(BF16)
ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
x = torch.empty(Mmax, C, device=dev)
for wg in list(range(768, 768 * 4 + 1, 96)):
    M = wg * 64
    t = timeit(lambda: K.up_ln_fwd(h[:M], w, b, r32[:M], ga, be, res16=r16[:M], out=x[:M]))
    by = M * C * 12 + M * Kd * 2
    print(f"rows {M:7d} ({wg / 768:5.3f} x 768 workgroup-slots of 64 rows): {t:7.1f} us  {by / t / 1e6:5.2f} TB/s  {t / (wg / 768):6.1f} us per 768", flush=True)
