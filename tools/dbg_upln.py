"""GPU debug: fused join + LayerNorm vs the two-kernel pair, element by element."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa
from stgcma import kernels as k

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
torch.manual_seed(0)
for M, C, K_, use16, use_rs in [(490, 128, 16, True, False), (490, 128, 16, False, True), (245, 256, 32, True, False), (980, 512, 64, True, False),
                                (62720, 512, 32, True, False), (1960, 256, 32, False, True), (490, 256, 64, True, False), (490, 128, 32, True, False)]:
    h = torch.randn(M, K_, device=dev).to(BF16)
    w = (torch.randn(C, K_, device=dev) * 0.3).to(BF16)
    b = torch.randn(C, device=dev)
    r32 = torch.randn(M, C, device=dev) * 3
    r16 = torch.randn(M, C, device=dev).to(BF16) if use16 else None
    ga, be = torch.randn(C, device=dev), torch.randn(C, device=dev)
    T, N = 5, 49
    rs = None
    if use_rs:
        rs = (torch.rand(-(-M // (T * N)) * N, device=dev) < 0.8).float() / 0.8
    kw = dict(row_scale=rs, rs_outer=T * N, rs_inner=N)
    x1, y1, m1, s1 = k.up_ln_fwd(h, w, b, r32, ga, be, res16=r16, **kw)
    if use16:
        x2 = k.gemm_nt(h, w, b, out_dtype=F32, res1=r16, res2=r32, **kw)
    else:
        x2 = k.gemm_nt(h, w, b, out_dtype=F32, res1=r32, **kw)
    y2, m2, s2 = k.layernorm_fwd(x2, ga, be)
    y3, m3, s3 = k.layernorm_fwd(x1, ga, be)
    dx = (x1 - x2).abs()
    print(f"M={M} C={C} K={K_} r16={use16} rs={use_rs}: |dx|max={float(dx.max()):.3e} at {int(dx.argmax()) // C},{int(dx.argmax()) % C}  "
          f"y(fused) vs LN(x_fused): mismatches {int((y1 != y3).sum())} / {y1.numel()}, max {float((y1.float() - y3.float()).abs().max()):.3e}; "
          f"mean diff {float((m1 - m3).abs().max()):.2e} rstd rel {float(((s1 - s3) / s3).abs().max()):.2e}", flush=True)
