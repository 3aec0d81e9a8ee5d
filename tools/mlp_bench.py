"""Fused MLP kernel (stg_mlp_fwd) against the two-GEMM path at the Swin-B stage-0 shape:  python tools/mlp_bench.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
dev = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2007040
C_ = 128
g = torch.Generator(device=dev).manual_seed(0)
Y = torch.randn(rows, C_, generator=g, device=dev).bfloat16()
W1 = (torch.randn(4 * C_, C_, generator=g, device=dev) / C_ ** 0.5).bfloat16()
W2 = (torch.randn(C_, 4 * C_, generator=g, device=dev) / (4 * C_) ** 0.5).bfloat16()
b1 = torch.randn(4 * C_, generator=g, device=dev) * 0.1
b2 = torch.randn(C_, generator=g, device=dev) * 0.1
W2p = W2[:, K.mlp_w2_perm(4 * C_, dev)].contiguous()
def t(fn, iters=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def two():
    H, Z = K.gemm_nt(Y, W1, b1, act=K.ACT_GELU, want_dact="u8")
    return K.gemm_nt(H, W2, b2)
ref = two()
out = K.mlp_fwd(Y, W1, b1, W2p, b2)
d = (out.float() - ref.float())
print(f"max |fused - two-GEMM| = {float(d.abs().max()):.4e} (scale {float(ref.float().abs().max()):.3f}), relL2 {float(d.norm() / ref.float().norm()):.3e}")
t2, t1 = t(two), t(lambda: K.mlp_fwd(Y, W1, b1, W2p, b2))
fl = 2.0 * rows * C_ * 4 * C_ * 2
print(f"rows {rows}: two GEMMs {t2:8.1f} us   fused {t1:8.1f} us ({fl / t1 / 1e6:6.1f} TF/s, {(rows * C_ * 4) / t1 / 1e3:5.0f} GB/s in+out)")
dM = torch.randn(rows, C_, generator=g, device=dev).bfloat16()
W2T, W1T = W2.t().contiguous(), W1.t().contiguous()
H, Z = K.gemm_nt(Y, W1, b1, act=K.ACT_GELU, want_dact="u8")
def two_b():
    return K.gemm_nt(K.gemm_nt(dM, W2T, dact_src=Z), W1T)
rb = two_b()
ob = K.mlp_bwd(Y, dM, W1, b1, W2T)
db = ob.float() - rb.float()
print(f"bwd: max |fused - two-GEMM| = {float(db.abs().max()):.4e} (scale {float(rb.float().abs().max()):.3f}), relL2 {float(db.norm() / rb.float().norm()):.3e}")
t2, t1 = t(two_b), t(lambda: K.mlp_bwd(Y, dM, W1, b1, W2T))
print(f"bwd rows {rows}: two GEMMs {t2:8.1f} us   fused {t1:8.1f} us ({1.5 * fl / t1 / 1e6:6.1f} TF/s)")
