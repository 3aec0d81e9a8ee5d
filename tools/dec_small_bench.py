"""Times the AVS decoder's small kernels (BatchNorm passes, bilinear x2) at the step's shapes; run under rocprofv3 --kernel-trace --stats."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stgcma
from stgcma import kernels as K

dev = "cuda:0"
torch.manual_seed(0)
for R, C in ((501760, 256), (125440, 256)):
    x = torch.randn(R, C, device=dev).bfloat16()
    dy = torch.randn(R, C, device=dev).bfloat16()
    g = torch.rand(C, device=dev) + 0.5
    b = torch.randn(C, device=dev)
    for _ in range(5):
        mean = (K.bn_colsum(x)[0] / R).contiguous()
        s2 = K.bn_colsum(x, mean=mean)
        rstd = torch.rsqrt(s2[1] / R + 1e-5).contiguous()
        y = K.bn_apply(x, mean, rstd, g, b)
        s = K.bn_colsum(x, dy, mean, rstd)
        dx = K.bn_bwd(x, dy, mean, rstd, g, s)
for F, H, W, C in ((40, 224, 224, 128), (40, 112, 112, 256)):
    x = torch.randn(F * H * W, C, device=dev).bfloat16()
    for _ in range(5):
        y = K.bilinear_up2_fwd(x, F, H, W, True)
        dx = K.bilinear_up2_bwd(y, F, H, W, True)
torch.cuda.synchronize()
print("done")
