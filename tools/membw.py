"""Calibration: attainable HBM fill / copy / read bandwidth on this GPU through plain ATen kernels."""
import torch
dev = "cuda"
def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (64, 512, 2048):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.empty_like(a)
    ms = t(lambda: a.fill_(1.0)); print(f"fill  {mb:5d} MB: {ms*1e3:8.1f} us  {mb*1.048576/ms:7.1f} GB/s")
    ms = t(lambda: b.copy_(a)); print(f"copy  {mb:5d} MB: {ms*1e3:8.1f} us  {2*mb*1.048576/ms:7.1f} GB/s (r+w)")
    ms = t(lambda: a.sum()); print(f"sum   {mb:5d} MB: {ms*1e3:8.1f} us  {mb*1.048576/ms:7.1f} GB/s")
