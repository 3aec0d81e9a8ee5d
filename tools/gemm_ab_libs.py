"""Two builds of the library on one box, per GEMM class: run once per build (STGCMA_LIB=...), SAVE=path stores the outputs' checksums, CHECK=path compares bit for bit.
usage: [STGCMA_LIB=alt.so] [SAVE=f | CHECK=f] python tools/gemm_ab_libs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
shapes = [(125440, 512, 2048, ""), (125440, 512, 1536, ""), (125440, 512, 512, "b"), (62720, 512, 2048, "b"), (31360, 1024, 4096, ""), (31360, 1024, 3072, ""), (31360, 3072, 1024, "b"),
          (31360, 1024, 1024, "b"), (501760, 256, 1024, ""), (501760, 256, 768, ""), (65536, 2048, 8192, ""), (31520, 1024, 4096, "b"), (7840, 512, 2048, "")]
saved = {}
check = torch.load(os.environ["CHECK"]) if os.environ.get("CHECK") else None
for (M, N, Kd, epi) in shapes:
    torch.manual_seed(1)
    A = (torch.randn(M, Kd, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda") * 0.1 if "b" in epi else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    best = 1e9
    for rnd in range(4):
        for _ in range(3): K.gemm_nt(A, W, b, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): K.gemm_nt(A, W, b, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    key = f"{M}x{N}x{Kd}{epi}"
    saved[key] = out.cpu()
    same = "" if check is None else f"  identical to the saved build: {torch.equal(saved[key].view(torch.int16), check[key].view(torch.int16))}"
    print(f"{key:24s} {best:8.1f} us  {2.0*M*N*Kd/best/1e6:7.1f} TFLOP/s  [{K.LAST_GEMM_KERNEL}]{same}", flush=True)
if os.environ.get("SAVE"):
    torch.save(saved, os.environ["SAVE"])
