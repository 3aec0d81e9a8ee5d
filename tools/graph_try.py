"""Capture one whole training step (forward + loss + backward + Adam) of the bench model in a HIP graph and replay it:
host time per step and step time, eager vs graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, stgcma
from stgcma import recipe
dev = torch.device('cuda:0')
B = int(os.environ.get("B", 32))
m = bench.build_model(torch, dev, 'swin_b')
adapt, head = recipe.apply_freeze(m)
opt = torch.optim.Adam([{"params": adapt, "lr": 1e-4}, {"params": head, "lr": 1e-5}], weight_decay=5e-7, betas=(0.95, 0.999), capturable=True)
loss_fn = torch.nn.CrossEntropyLoss()
a, v, labels = bench.synth_batch(torch, B, dev, 0, 'swin_b')
def step():
    loss = loss_fn(m(a, v, "fusion"), labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss
def timeit(fn, n=5):
    hs, ts = [], []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
    return sorted(hs)[n // 2], sorted(ts)[n // 2]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager: host %.1f ms, step %.1f ms" % timeit(step), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
l0 = float(static_loss)
print("graph captured; loss", l0, flush=True)
print("graph: host %.1f ms, step %.1f ms" % timeit(g.replay), flush=True)
print("loss after replays", float(static_loss))
