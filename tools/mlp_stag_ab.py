"""Fused MLP forward (C = 128): the lockstep kernel against the staggered one (option mlp_stagger), same process, interleaved; outputs
must be bit-identical (same operations, same order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
dev = "cuda"
C_ = 128
for rows in (2007040, 1003520, 300):
    g = torch.Generator(device=dev).manual_seed(0)
    Y = torch.randn(rows, C_, generator=g, device=dev).bfloat16()
    W1 = (torch.randn(4 * C_, C_, generator=g, device=dev) / C_ ** 0.5).bfloat16()
    W2 = (torch.randn(C_, 4 * C_, generator=g, device=dev) / (4 * C_) ** 0.5).bfloat16()
    b1 = torch.randn(4 * C_, generator=g, device=dev) * 0.1
    b2 = torch.randn(C_, generator=g, device=dev) * 0.1
    W2p = W2[:, K.mlp_w2_perm(4 * C_, dev)].contiguous()

    def t(fn, iters=8):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    stgcma.configure(lib_mlp_stagger=0)
    ref = K.mlp_fwd(Y, W1, b1, W2p, b2)
    stgcma.configure(lib_mlp_stagger=1)
    same = []
    for rep in range(6):                       # repeats: a race in the ring would come and go
        out = K.mlp_fwd(Y, W1, b1, W2p, b2)
        torch.cuda.synchronize()
        same.append(bool(torch.equal(out.view(torch.int16), ref.view(torch.int16))))
    ts = {0: [], 1: []}
    for r in range(4):
        for m in (0, 1):
            stgcma.configure(lib_mlp_stagger=m)
            ts[m].append(t(lambda: K.mlp_fwd(Y, W1, b1, W2p, b2)))
    print(f"rows {rows}: lockstep {sorted(ts[0])[1]:8.1f} us   staggered {sorted(ts[1])[1]:8.1f} us   bit-identical over 6 runs: {same}", flush=True)
stgcma.configure(lib_mlp_stagger=0)
