"""Reproducibility stress of more kernels: fused MLP forward / backward (stage 0), temporal attention forward / backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K
dev = torch.device("cuda:0"); BF16, F32 = torch.bfloat16, torch.float32
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
bf = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF16)


def stress(name, fn, reps=REPS):
    ref = [t.clone() for t in fn()]
    bad = torch.zeros((), device=dev, dtype=torch.int64)
    junk = torch.randn(4096, 32, device=dev)
    for r in range(reps):
        if r % 3 == 0:
            junk = junk * 1.0001
        for t, q in zip(fn(), ref):
            bad += (t.view(torch.int16) != q.view(torch.int16)).any()
    torch.cuda.synchronize()
    print(f"{name:50s} {int(bad):5d} of {reps} launches differ", flush=True)


torch.manual_seed(0)
for M in (125440, 62720 + 96):                       # a multiple of the kernel's row chunk and a ragged count
    Y, dM = bf(M, 128, sc=0.5), bf(M, 128, sc=0.5)
    w1, b1 = bf(512, 128, sc=0.05), torch.randn(512, device=dev) * 0.1
    w2, b2 = bf(128, 512, sc=0.05), torch.randn(128, device=dev) * 0.1
    w2p = w2[:, K.mlp_w2_perm(512, dev)].contiguous()
    w2t = w2.t().contiguous()
    stress(f"mlp_fwd C=128 M={M}", lambda: (K.mlp_fwd(Y, w1, b1, w2p, b2),))
    stress(f"mlp_bwd C=128 M={M}", lambda: (K.mlp_bwd(Y, dM, w1, b1, w2t),), reps=REPS // 2)
for B in (32, 2):
    nm, T, N, H = 2, 10, 196, 16
    rows = nm * B * T * N
    qkv, dO = bf(rows, 3 * H * 32), bf(rows, H * 32)
    tb = torch.randn(nm, H, T * T, device=dev) * 0.3
    tg = K.TGeom(nm, B, T, N, H, 32 ** -0.5, tb)
    C = H * 32
    Q, Kk, V = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    dq = torch.empty_like(qkv)
    stress(f"tattn_fwd B={B}", lambda: (K.tattn_fwd(tg, Q, Kk, V),), reps=REPS // 2)
    stress(f"tattn_bwd B={B}", lambda: (K.tattn_bwd(tg, Q, Kk, V, dO, dQ=dq[:, :C], dK=dq[:, C:2 * C], dV=dq[:, 2 * C:]) and dq,), reps=REPS // 2)
