import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K, ops
kind = sys.argv[1]; B, T = 32, 10; BT = B * T
C, res, H = 128, 56, 4; N = res * res; R = 2 * BT * N; hd = 32
qkv = torch.randn(R, 3 * C, device="cuda").bfloat16()
if kind == "window":
    bias = torch.randn(1, H, 49, 49, device="cuda"); nW = 64
    ge = ops.geom(torch.device("cuda"), res, res, 7, 3, T)
    g = K.AttnGeom(2 * BT * nW, H, 49, hd, G=nW, outer=N, window=(res, res, 7, 3), scale=hd ** -0.5, bias=bias, bias_div=2 * BT * nW, bias_mod=1, mask=ge["mask"])
else:
    tb = torch.randn(2, H, T, T, device="cuda")
    g = K.AttnGeom(2 * B * N, H, T, hd, G=N, outer=T * N, temporal=N, scale=hd ** -0.5, bias=tb, bias_div=B * N, bias_mod=2)
for _ in range(3): O, lse = K.attn_fwd(g, qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:])
torch.cuda.synchronize()
