import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
M, N, Kd = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, Kd, device="cuda").bfloat16(); W = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5): K.gemm_nt(A, W, None, out=out)
torch.cuda.synchronize()
