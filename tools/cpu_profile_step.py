"""Where the HOST time of a training step goes (cProfile over a few steps of the bench model)."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
import bench
dev = torch.device("cuda:0")
m = bench.build_model(torch, dev)
B = int(os.environ.get("B", 32))
a, v, labels = bench.synth_batch(torch, B, dev, 0)
loss_fn = torch.nn.CrossEntropyLoss()
params = [p for p in m.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-4)
def step():
    loss = loss_fn(m(a, v, "fusion"), labels); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/3:.1f} ms/step, with sync {1e3*(t2-t0)/3:.1f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
