"""Library GEMM (torch.mm -> hipBLASLt / rocBLAS) against stg_gemm_nt on the PLAIN classes of the step (no epilogue beyond a bias):
which, if any, would a library call beat?  Interleaved timing, 512 MiB written between launches (cold caches, as in the step)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa: F401
from stgcma import kernels as K

torch.manual_seed(0)
junk = torch.empty(512 * 2**20 // 4, device="cuda")
for M, N, Kd in ((125440, 512, 1536), (125440, 512, 2048), (125440, 1536, 512), (125440, 512, 512), (125440, 2048, 512), (62720, 512, 2048),
                 (501760, 256, 1024), (31360, 1024, 4096)):
    A = (torch.randn(M, Kd, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    Wt = W.t().contiguous()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for name, fn in (("stg", lambda: K.gemm_nt(A, W, out=out)), ("mm_nt", lambda: torch.mm(A, W.t(), out=out)), ("mm_nn", lambda: torch.mm(A, Wt, out=out))):
        ts = []
        for i in range(12):
            junk.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[name] = ts[len(ts) // 2]
    fl = 2.0 * M * N * Kd
    print(f"{M} x {N} x {Kd}: " + "  ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:6.0f} TFLOP/s)" for k, v in res.items()), flush=True)
