"""Find the kernel call that first produces a non-finite value: every function of stgcma.kernels is wrapped -- tensors among the arguments are
checked before the call (so an already-bad input is reported as such), tensors among arguments and results after it.  Small batch, many steps.
usage: python tools/nan_trace.py [B] [steps]"""
import os, sys, types, inspect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K, recipe
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400


def tensors(obj, out):
    if torch.is_tensor(obj):
        if obj.is_cuda and obj.is_floating_point() and obj.numel():
            out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            tensors(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            tensors(o, out)
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("stgcma"):
        for o in vars(obj).values():
            if torch.is_tensor(o):
                tensors(o, out)
    return out


class Found(Exception):
    pass


def bad_of(ts):
    res = []
    for i, t in enumerate(ts):
        tv = t if t.is_contiguous() else t
        if not bool(torch.isfinite(tv).all()):
            res.append((i, tuple(t.shape), str(t.dtype), int((~torch.isfinite(tv)).sum())))
    return res


def wrap(name, fn):
    def w(*a, **kw):
        ins = tensors((a, kw), [])
        # outputs passed in (out=...) may legitimately hold garbage before the call: only complain about them afterwards
        pre = {id(t): bool(torch.isfinite(t).all()) for t in ins}
        r = fn(*a, **kw)
        outs = tensors((a, kw, r), [])
        b = bad_of(outs)
        if b:
            newly = [x for x, t in zip(b, [outs[i] for i, *_ in b]) if pre.get(id(t), True) or id(t) not in pre]
            inputs_bad = [tuple(t.shape) for t in ins if not pre[id(t)]]
            raise Found(f"{name}: non-finite after the call: {b[:6]}; tensors already non-finite BEFORE the call (inputs or uninitialised outputs): {inputs_bad[:6]}; "
                        f"arg shapes {[tuple(t.shape) for t in ins][:10]}")
        return r
    return w


for n, f in list(vars(K).items()):
    if isinstance(f, types.FunctionType) and not n.startswith("_") and f.__module__ == K.__name__ and n not in ("family_profile_start", "family_profile_stop",
                                                                                                                 "family_profile_reset", "gemm_profile_start", "gemm_profile_stop", "gemm_profile_reset", "gemm_profile_sequence"):
        setattr(K, n, wrap(n, f))

dev = torch.device("cuda:0")
model = bench.build_model(torch, dev, "swin_b")
opt = recipe.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
a, v, labels = bench.synth_batch(torch, B, dev, 0, "swin_b")
loss_fn = torch.nn.CrossEntropyLoss()
try:
    for it in range(STEPS):
        out = model(a, v, "fusion")
        loss = loss_fn(out, labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
    print(f"B={B}: {STEPS} steps, no kernel produced a non-finite value", flush=True)
except Found as e:
    print(f"B={B} step {it}: {e}", flush=True)
    sys.exit(1)
