"""Where do the rare non-reproducible outputs of a GEMM launch lie?  Accumulates (out != first output) element-wise over many launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
REPS = int(os.environ.get("REPS", 40000))
M, N, Kd = [int(x) for x in os.environ.get("SHAPE", "31360x1024x4096").split("x")]
torch.manual_seed(0)
A = (torch.randn(M, Kd, device=dev) * 0.5).to(BF16)
W = (torch.randn(N, Kd, device=dev) * 0.05).to(BF16)
b = torch.randn(N, device=dev) * 0.1
ref = K.gemm_nt(A, W, b).clone()
mask = torch.zeros(M, N, device=dev, dtype=torch.bool)
cnt = torch.zeros((), device=dev, dtype=torch.int64)
maxd = torch.zeros((), device=dev)
for r in range(REPS):
    out = K.gemm_nt(A, W, b)
    d = out.view(torch.int16) != ref.view(torch.int16)
    mask |= d
    cnt += d.any()
    maxd = torch.maximum(maxd, (out.float() - ref.float()).abs().max())
torch.cuda.synchronize()
idx = mask.nonzero()
print(f"M={M} N={N} K={Kd}: {int(cnt)} of {REPS} launches differ; {idx.shape[0]} distinct elements ever differed; max |diff| {float(maxd):.4g}")
if idx.shape[0]:
    rows, cols = idx[:, 0], idx[:, 1]
    print("row panels (256):", sorted(set((rows // 256).tolist()))[:20], "rows in panel:", sorted(set((rows % 256).tolist()))[:40])
    print("col tiles (256):", sorted(set((cols // 256).tolist())), "cols in tile:", sorted(set((cols % 256).tolist()))[:40])
    print("first elements:", idx[:12].tolist())
