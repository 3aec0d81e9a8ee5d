import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as k
torch.manual_seed(0)
M, N1, N2 = 333, 128, 16
dY = torch.randn(M, N1).bfloat16(); X = torch.randn(M, N2).bfloat16()
dW = torch.zeros(N1, N2, device="cuda"); db = torch.zeros(N1, device="cuda")
k.wgrad_tn(dY.cuda(), X.cuda(), dW, db)
ref = dY.float().sum(0)
err = (db.cpu() - ref).abs()
print("bad idx", torch.nonzero(err > 0.05).flatten().tolist())
print("ratio", (db.cpu() / ref)[:20])
ones = torch.ones(M, N1).bfloat16()
db2 = torch.zeros(N1, device="cuda"); dW2 = torch.zeros(N1, N2, device="cuda")
k.wgrad_tn(ones.cuda(), X.cuda(), dW2, db2); print("ones:", db2.cpu()[:40])
