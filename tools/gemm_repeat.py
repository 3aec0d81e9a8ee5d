"""Is a GEMM launch deterministic?  The same gemm_nt call thousands of times back to back (other kernels in between, as in the step), every
output compared with the first one on the device (no host synchronisation inside the loop).  usage: gemm_repeat.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
REPS = int(os.environ.get("REPS", 6000))
SHAPES = [tuple(int(x) for x in s_.split("x")) for s_ in os.environ.get("SHAPES", "7840x512x2048,7680x512x2048,31360x1024x4096,30720x1024x4096,125440x512x2048").split(",")]
for o_ in os.environ.get("OPTS", "").split(","):
    if o_:
        stgcma.configure(**{"lib_" + o_.split("=")[0]: int(o_.split("=")[1])})
for M, N, Kd in SHAPES:
    torch.manual_seed(0)
    A = (torch.randn(M, Kd, device=dev) * 0.5).to(BF16)
    W = (torch.randn(N, Kd, device=dev) * 0.05).to(BF16)
    b = torch.randn(N, device=dev) * 0.1
    other = torch.randn(3920, 32, device=dev).to(BF16)
    ref = K.gemm_nt(A, W, b).clone()
    bad = torch.zeros((), device=dev, dtype=torch.int64)
    nonfin = torch.zeros((), device=dev, dtype=torch.int64)
    for r in range(REPS):
        if r % 3 == 0:
            other = other * 1.0001           # a small kernel in between, like the step's element-wise launches
        out = K.gemm_nt(A, W, b)
        bad += (out.view(torch.int16) != ref.view(torch.int16)).any()
        nonfin += (~torch.isfinite(out)).any()
    torch.cuda.synchronize()
    kern = K.gemm_nt.__wrapped__ if hasattr(K.gemm_nt, "__wrapped__") else None
    print(f"[{os.environ.get('OPTS', '')}] M={M} N={N} K={Kd}: {int(bad)} of {REPS} launches differ from the first, {int(nonfin)} contain non-finite values", flush=True)
