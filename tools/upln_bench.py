"""GPU micro-bench: fused up-projection + residual + LayerNorm vs the GEMM-epilogue + LayerNorm pair (stage shapes, B = 32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa
from stgcma import kernels as k

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


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


for M, C, K in [(1003520, 128, 16), (250880, 256, 32), (62720, 512, 32), (62720, 512, 64), (63040, 768, 48), (15680, 768, 48)]:
    h = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(C, K, device=dev) * 0.1).to(BF16)
    b = torch.randn(C, device=dev)
    r32 = torch.randn(M, C, device=dev)
    r16 = torch.randn(M, C, device=dev).to(BF16)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    x = torch.empty(M, C, device=dev)
    for use16 in (True, False):
        def fused():
            k.up_ln_fwd(h, w, b, r32, ga, be, res16=r16 if use16 else None, out=x)

        def pair():
            if use16:
                k.gemm_nt(h, w, b, out=x, res1=r16, res2=r32)
            else:
                k.gemm_nt(h, w, b, out=x, res1=r32)
            k.layernorm_fwd(x, ga, be)
        tf, tp = timeit(fused), timeit(pair)
        byt_f = M * C * (4 + 4 + 2 + (2 if use16 else 0)) + M * K * 2
        print(f"M={M} C={C} K={K} res16={use16}: fused {tf:8.1f} us ({byt_f / tf / 1e6:6.2f} TB/s)   pair {tp:8.1f} us", flush=True)

print("--- LayerNorm backward + down-projection")
for M, C, J in [(1003520, 128, 16), (250880, 256, 32), (62720, 512, 32), (62720, 512, 64), (63040, 768, 48), (15680, 768, 48)]:
    x = torch.randn(M, C, device=dev)
    dy = torch.randn(M, C, device=dev).to(BF16)
    ad = torch.randn(M, C, device=dev).to(BF16)
    ga = torch.ones(C, device=dev)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()
    wt = (torch.randn(J, C, device=dev) * 0.1).to(BF16)

    def fused():
        k.ln_bwd_down(dy, x, ga, mean, rstd, wt, add_to=ad)

    def pair():
        dx = k.layernorm_bwd(dy, x, ga, mean, rstd, add_to=ad)
        k.gemm_nt(dx, wt)
    tf, tp = timeit(fused), timeit(pair)
    byt = M * C * (2 + 4 + 2 + 2) + M * J * 2
    print(f"M={M} C={C} J={J}: fused {tf:8.1f} us ({byt / tf / 1e6:6.2f} TB/s)   pair {tp:8.1f} us", flush=True)
