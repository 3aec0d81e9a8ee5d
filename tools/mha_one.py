"""The mha.hip kernels on one shape, for rocprofv3 passes / quick timings:  python tools/mha_one.py {xl|vit|vita} [reps]
xl = Swin-L stage-0 frame-global cross-modal attention (P 320, H 1, n 3136, D 96, K == V); vit = ViT-B/16 video self-attention
(P 320, H 8, n 197, D 96); vita = its audio half (n 49)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma  # noqa: F401
from stgcma import kernels as K

kind = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
if kind == "xl":
    P, H, n, D, sc, kv1 = 320, 1, 3136, 96, 1.0, True
else:
    P, H, n, D, sc, kv1 = 320, 8, (197 if kind == "vit" else 49), 96, 96 ** -0.5, False
g = K.MhaGeom(P, H, n, D, sc)
if kv1:
    q = (torch.randn(P * n, D, device="cuda") * 0.3).bfloat16()
    kk = (torch.randn(P * n, D, device="cuda") * 0.3).bfloat16()
    Q, Kk, V = q, kk, kk
else:
    qkv = (torch.randn(P * n, 3 * H * D, device="cuda") * 0.5).bfloat16()
    Q, Kk, V = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
dO = torch.randn(P * n, H * D, device="cuda").bfloat16()
dQ, dK = torch.empty(P * n, H * D, device="cuda", dtype=torch.bfloat16), torch.empty(P * n, H * D, device="cuda", dtype=torch.bfloat16)
dV = None if kv1 else torch.empty_like(dK)
O, lse = K.mha_fwd(g, Q, Kk, V)
K.mha_bwd(g, Q, Kk, V, O, lse, dO, dQ=dQ, dK=dK, dV=dV)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    O, lse = K.mha_fwd(g, Q, Kk, V)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    K.mha_bwd(g, Q, Kk, V, O, lse, dO, dQ=dQ, dK=dK, dV=dV)
torch.cuda.synchronize()
t2 = time.perf_counter()
fl = 4.0 * P * H * n * n * D
print(f"{kind}: fwd {(t1 - t0) / reps * 1e6:.1f} us ({fl / ((t1 - t0) / reps) / 1e12:.0f} TFLOP/s), bwd {(t2 - t1) / reps * 1e6:.1f} us ({2.5 * fl / ((t2 - t1) / reps) / 1e12:.0f} TFLOP/s)")
