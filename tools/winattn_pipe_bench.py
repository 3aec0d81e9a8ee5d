"""Pipelined persistent window-attention forward (csrc/winattn.hip winattn_fwd2_kernel, option winattn_pipe) against winattn_fwd1: outputs
and lse compared bit for bit, then interleaved timing in one process, at the step's stage shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as k, ops
import oracle.swin as OS          # relative_position_index only (a tool, not the product path)

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
ROUNDS = int(os.environ.get("ROUNDS", 5))
MODES = [int(x) for x in os.environ.get("MODES", "0,2,4,8,16,32").split(",")]
BWD = int(os.environ.get("BWD", "0"))
OPT = "lib_winattn_pipe_bwd" if BWD else "lib_winattn_pipe"


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (name, images, heads, Himg, shift): images = 2 modalities x B 32 x T 8 (stage 2: 14 x 14 tokens, 16 heads, C = 512)
cases = [("stage2 shifted", 640, 16, 14, 3), ("stage2 plain", 640, 16, 14, 0), ("stage2 half batch shifted", 320, 16, 14, 3),
         ("stage1 shifted", 640, 8, 28, 3), ("stage0 shifted", 640, 4, 56, 3), ("odd sizes", 37, 12, 14, 3), ("odd plain", 5, 4, 28, 0)]
sel = os.environ.get("CASES")
for name, images, heads, Himg, shift in cases:
    if sel and not any(s in name for s in sel.split(",")):
        continue
    ws = 7
    n, N, C = ws * ws, Himg * Himg, heads * 32
    torch.manual_seed(0)
    qkv = torch.randn(images * N, 3 * C, device=dev).to(BF16)
    dO = torch.randn(images * N, C, device=dev).to(BF16)
    table = (torch.randn((2 * ws - 1) ** 2, heads) * 0.5).to(dev)
    index = OS.relative_position_index(ws).reshape(-1).to(dev)
    mask = ops.shift_mask(Himg, Himg, ws, shift).to(dev) if shift > 0 else None
    bm, bmT = k.winattn_table(table, index, mask, n)
    wg = k.WinGeom(images, heads, Himg, Himg, ws, shift, 32 ** -0.5, bm, bmT)
    Q, K_, V = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    stgcma.configure(lib_winattn_pipe=0, lib_winattn_pipe_bwd=0) if BWD else stgcma.configure(lib_winattn_pipe=0)
    O0, lse0 = k.winattn_fwd(wg, Q, K_, V)
    dq = torch.empty_like(qkv)

    def run():
        if BWD:
            k.winattn_bwd(wg, Q, K_, V, O0, lse0, dO, dQ=dq[:, :C], dK=dq[:, C:2 * C], dV=dq[:, 2 * C:])
            return (dq,)
        return k.winattn_fwd(wg, Q, K_, V)

    ref = [t.clone() for t in run()]
    torch.cuda.synchronize()
    ts = {m: [] for m in MODES}
    same = {}
    for r in range(ROUNDS):
        for m in MODES:
            stgcma.configure(**{OPT: m})
            if r == 0:
                if BWD:
                    dq.fill_(float("nan"))
                out = run()
                torch.cuda.synchronize()
                same[m] = all(torch.equal(x.view(torch.int16) if x.dtype == BF16 else x[..., :n], y.view(torch.int16) if y.dtype == BF16 else y[..., :n]) for x, y in zip(out, ref))
                if not same[m]:
                    d = (out[0].float() - ref[0].float()).abs()
                    same[m] = f"MISMATCH max {float(d.max()):.3e} frac {float((d > 0).float().mean()):.3e} nan {int(torch.isnan(out[0].float()).sum())}"
            ts[m].append(timeit(run))
    byt = images * N * C * 2 * (8 if BWD else 4)
    print(f"{name}: images={images} heads={heads} {Himg}x{Himg} shift={shift}  ({byt / 1e6:.0f} MB algorithmic)", flush=True)
    for m in MODES:
        t = sorted(ts[m])
        print(f"    pipe {m:3d}  median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  {byt / t[len(t) // 2] / 1e6:6.2f} TB/s  identical {same[m]}", flush=True)
stgcma.configure(**{OPT: 0})
