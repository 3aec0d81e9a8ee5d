"""Where does an intermittent non-finite value first appear?  ops.block_forward / block_backward are wrapped to record ASYNCHRONOUS finiteness
flags (device scalars; no host synchronisation inside the step, so the timing of the step is barely disturbed); after each step the
flags are read and the first bad boundary is reported.  usage: python tools/nan_where.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import ops, recipe
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 700
flags = []
_bf, _bb = ops.block_forward, ops.block_backward


def fin(t):
    return torch.isfinite(t).all() if torch.is_tensor(t) and t.is_floating_point() and t.numel() else None


def block_forward(X, spec, P, *a, **kw):
    fi = fin(X)
    r = _bf(X, spec, P, *a, **kw)
    flags.append((f"fwd C={spec.C if hasattr(spec, 'C') else X.shape[1]} shift={spec.shift} t_attn={spec.t_attn} rows={X.shape[0]}", fi, fin(r[0]), None))
    return r


def block_backward(S, spec, P, need, prefix, dX3, *a, **kw):
    fi = fin(dX3)
    r = _bb(S, spec, P, need, prefix, dX3, *a, **kw)
    gbad = [(n, fin(g)) for n, g in (r[1] or {}).items()]
    flags.append((f"bwd {prefix} shift={spec.shift} t_attn={spec.t_attn} rows={dX3.shape[0]}", fi, fin(r[0]) if r[0] is not None else None, gbad))
    return r


ops.block_forward, ops.block_backward = block_forward, block_backward
dev = torch.device("cuda:0")
model = bench.build_model(torch, dev, "swin_b")
opt = recipe.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
a, v, labels = bench.synth_batch(torch, B, dev, 0, "swin_b")
loss_fn = torch.nn.CrossEntropyLoss()
for it in range(STEPS):
    flags.clear()
    out = model(a, v, "fusion")
    loss = loss_fn(out, labels)
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    first = None
    for k, (what, fi, fo, gb) in enumerate(flags):
        bad_g = [n for n, f in (gb or []) if f is not None and not bool(f)]
        if (fi is not None and not bool(fi)) or (fo is not None and not bool(fo)) or bad_g:
            first = (k, what, None if fi is None else bool(fi), None if fo is None else bool(fo), bad_g[:8])
            break
    if first or not bool(torch.isfinite(out).all()):
        print(f"B={B} step {it}: first bad boundary {first}; logits finite {bool(torch.isfinite(out).all())}; boundaries {len(flags)}", flush=True)
        sys.exit(1)
    opt.step()
print(f"B={B}: {STEPS} steps finite", flush=True)
