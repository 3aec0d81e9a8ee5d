"""Window attention forward + backward at the step's shapes, timed with whatever library is in place; with SAVE=path the outputs are stored,
with CHECK=path compared bit for bit (A/B of two builds on one box: run twice, swapping stg-cma_amd/libstgcma_hip.so in between)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as k, ops
import oracle.swin as OS          # relative_position_index only (a tool, not the product path)

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
ROUNDS = int(os.environ.get("ROUNDS", 5))
for kv in os.environ.get("OPTS", "").split(","):
    if kv:
        name, v = kv.split("=")
        stgcma.configure(**{"lib_" + name: int(v)})


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


cases = [("stage2 shifted", 640, 16, 14, 3), ("stage2 plain", 640, 16, 14, 0), ("stage2 half batch shifted", 320, 16, 14, 3),
         ("stage1 shifted", 640, 8, 28, 3), ("stage0 shifted", 640, 4, 56, 3), ("odd sizes", 37, 12, 14, 3), ("audio-like 1 head", 33, 1, 14, 0)]
saved = {}
check = torch.load(os.environ["CHECK"]) if os.environ.get("CHECK") else None
for name, images, heads, Himg, shift in cases:
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
    O, lse = k.winattn_fwd(wg, Q, K_, V)
    dq = torch.full_like(qkv, float("nan"))
    k.winattn_bwd(wg, Q, K_, V, O, lse, dO, dQ=dq[:, :C], dK=dq[:, C:2 * C], dV=dq[:, 2 * C:])
    torch.cuda.synchronize()
    small = images <= 40 or "half" in name
    if small:
        saved[name] = (O.cpu(), lse[..., :n].cpu(), dq.cpu())
    else:
        saved[name] = (O[:50000].cpu(), lse[:2000, :, :n].cpu(), dq[:50000].cpu(), float(O.float().sum()), float(dq.float().sum()))
    verdict = ""
    if check is not None:
        ok = all((torch.equal(x.view(torch.int16), y.view(torch.int16)) if torch.is_tensor(x) and x.dtype == BF16 else (torch.equal(x, y) if torch.is_tensor(x) else x == y))
                 for x, y in zip(saved[name], check[name]))
        verdict = f"  identical to the saved build: {ok}"
    tf, tb = [], []
    for r in range(ROUNDS):
        tf.append(timeit(lambda: k.winattn_fwd(wg, Q, K_, V)))
        tb.append(timeit(lambda: k.winattn_bwd(wg, Q, K_, V, O, lse, dO, dQ=dq[:, :C], dK=dq[:, C:2 * C], dV=dq[:, 2 * C:])))
    bf = images * N * C * 2
    tf.sort(); tb.sort()
    print(f"{name:28s} fwd {tf[len(tf) // 2]:7.1f} us {4 * bf / tf[len(tf) // 2] / 1e6:5.2f} TB/s   bwd {tb[len(tb) // 2]:7.1f} us {8 * bf / tb[len(tb) // 2] / 1e6:5.2f} TB/s{verdict}", flush=True)
if os.environ.get("SAVE"):
    torch.save(saved, os.environ["SAVE"])
