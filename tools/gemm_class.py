"""One GEMM class of the step in isolation, for rocprofv3 passes:  python tools/gemm_class.py M N K {plain|bias|gelu8|d8} [launches]
(the class = what bench.py's roofline_classes call kernel x N x K x epilogue; routing knobs through the STG_GEMM_* environment)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa: F401
from stgcma import kernels as K
from stgcma._lib import ACT_GELU

M, N, Kd = (int(x) for x in sys.argv[1:4])
epi = sys.argv[4] if len(sys.argv) > 4 else "plain"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
torch.manual_seed(0)
A = (torch.randn(M, Kd, device="cuda") * 0.5).bfloat16()
W = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
b = torch.randn(N, device="cuda") * 0.1
D8 = torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8) if epi == "d8" else None
junk = torch.empty(512 * 2**20 // 4, device="cuda")              # 512 MiB written between launches: the step never re-runs a GEMM on warm caches
for i in range(reps):
    junk.fill_(float(i))
    if epi == "d8":
        K.gemm_nt(A, W, dact_src=D8)
    elif epi == "gelu8":
        K.gemm_nt(A, W, b, act=ACT_GELU, want_dact="u8")
    elif epi == "bias":
        K.gemm_nt(A, W, b)
    else:
        K.gemm_nt(A, W)
torch.cuda.synchronize()
print("kernel:", K.LAST_GEMM_KERNEL)
