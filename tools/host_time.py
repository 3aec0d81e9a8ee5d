import sys, os, time
sys.path.insert(0, '/root/repo')
import torch, bench, stgcma
from stgcma import recipe
dev = torch.device('cuda:0')
m = bench.build_model(torch, dev, 'swin_b')
opt = recipe.build_optimizer(m, lr=1e-4, head_lr=0.1)
loss_fn = torch.nn.CrossEntropyLoss()
a, v, labels = bench.synth_batch(torch, 32, dev, 0, 'swin_b')
for _ in range(3):
    recipe.train_step(m, opt, loss_fn, a, v, labels, 'fusion')
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(5):
    t0 = time.perf_counter()
    recipe.train_step(m, opt, loss_fn, a, v, labels, 'fusion')
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
print("host enqueue ms:", [round(x, 1) for x in hs], " step ms:", [round(x, 1) for x in ts], "cpus:", os.cpu_count())
