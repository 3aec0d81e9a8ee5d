"""Which kernel call first produces a non-finite value -- WITHOUT synchronising inside the step (the synchronising tracer, tools/nan_trace.py, makes
the intermittent fault disappear): every function of stgcma.kernels is wrapped to append asynchronous finiteness flags (device scalars) of its
tensor arguments before the call and of its arguments + results after it; the flags are read once per step.  usage: nan_kernel.py [B] [steps]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K, recipe
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 500
log = []


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
    return out


ROWS = {7840, 3920}      # stage 2 of the B = 2 step (both modalities / one): where tools/nan_where.py saw the fault


def interesting(t):
    return t.dim() >= 1 and t.shape[0] in ROWS


def wrap(name, fn):
    def w(*a, **kw):
        r = fn(*a, **kw)
        outs = [t for t in tensors((r, [v for k, v in kw.items() if "out" in k or k in ("dQ", "dK", "dV", "dx_out", "outs")]), []) if interesting(t)]
        if outs:
            ins = [t for t in tensors((a,), []) if interesting(t)]
            log.append((name, [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in ins],
                        [(tuple(t.shape), str(t.dtype).replace("torch.", ""), torch.isfinite(t).all()) for t in outs],
                        {k: (v if isinstance(v, (int, float, bool, str, type(None))) else type(v).__name__) for k, v in kw.items()}))
        return r
    return w


skip = {"family_profile_start", "family_profile_stop", "family_profile_reset", "gemm_profile_start", "gemm_profile_stop", "gemm_profile_reset", "gemm_profile_sequence"}
for n, f in list(vars(K).items()):
    if isinstance(f, types.FunctionType) and not n.startswith("_") and f.__module__ == K.__name__ and n not in skip and not n.endswith("_supported"):
        setattr(K, n, wrap(n, f))

dev = torch.device("cuda:0")
model = bench.build_model(torch, dev, "swin_b")
opt = recipe.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
a, v, labels = bench.synth_batch(torch, B, dev, 0, "swin_b")
loss_fn = torch.nn.CrossEntropyLoss()
for it in range(STEPS):
    log.clear()
    out = model(a, v, "fusion")
    loss = loss_fn(out, labels)
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    for k, (name, ins, post, kw) in enumerate(log):
        bad_post = [(s_, d) for s_, d, f in post if not bool(f)]
        if bad_post:
            print(f"B={B} step {it} call #{k} of {len(log)}: {name} kwargs {kw}\n   non-finite outputs: {bad_post[:6]}\n   stage-2 inputs: {ins[:8]}\n"
                  f"   previous calls: {[(n_, [s_ for s_, *_ in p_][:2]) for n_, _, p_, _ in log[max(0, k - 8):k]]}", flush=True)
            sys.exit(1)
    opt.step()
print(f"B={B}: {STEPS} steps finite", flush=True)
