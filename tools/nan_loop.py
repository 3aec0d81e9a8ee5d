"""Hunt for an intermittent non-finite step at small batch: one process, many eager steps of the bench workload at batch B, finiteness of the
loss and of every gradient checked after each step.  usage: python tools/nan_loop.py [B] [steps] [mb]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import recipe
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
model = bench.build_model(torch, dev, "swin_b")
params = [p for p in model.parameters() if p.requires_grad]
opt = recipe.FusedAdam(params, lr=1e-4)
a, v, labels = bench.synth_batch(torch, B, dev, 0, "swin_b")
loss_fn = torch.nn.CrossEntropyLoss()
bad = None
for it in range(STEPS):
    out = model(a, v, "fusion")
    loss = loss_fn(out, labels)
    opt.zero_grad()
    loss.backward()
    ok_out = bool(torch.isfinite(out).all())
    ng = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    if not ok_out or ng:
        bad = (it, ok_out, float(loss.detach()), len(ng), ng[:6])
        break
    opt.step()
print(f"B={B}: {'first non-finite step ' + repr(bad) if bad else f'{STEPS} steps finite'}", flush=True)
sys.exit(1 if bad else 0)
