"""Which Python lines issue device-to-device copies (hipMemcpyDtoD / aten::copy_) during one training step?  Patches Tensor.copy_, clone and
contiguous, counts calls that actually copy CUDA -> CUDA, by the nearest stack frame inside stg-cma_amd/."""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
model = bench.build_model(torch, dev)
from stgcma import recipe
opt = recipe.build_optimizer(model, lr=1e-4, head_lr=0.1)
a, v, labels = bench.synth_batch(torch, 2, dev, 0)
loss_fn = torch.nn.CrossEntropyLoss()


def step():
    loss = loss_fn(model(a, v, "fusion"), labels.reshape(-1, labels.shape[-1]))
    opt.zero_grad()
    loss.backward()
    opt.step()


step(); step()
torch.cuda.synchronize()
sites = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "stgcma" in fr.filename or "stg-cma_amd" in fr.filename or "bench.py" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


o_copy, o_clone, o_contig, o_to = torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous, torch.Tensor.to


def copy_(self, src, *a_, **k):
    if self.is_cuda and isinstance(src, torch.Tensor) and src.is_cuda:
        sites["copy_ " + site()] += 1
    return o_copy(self, src, *a_, **k)


def clone(self, *a_, **k):
    if self.is_cuda:
        sites["clone " + site()] += 1
    return o_clone(self, *a_, **k)


def contiguous(self, *a_, **k):
    if self.is_cuda and not self.is_contiguous():
        sites["contiguous " + site()] += 1
    return o_contig(self, *a_, **k)


torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = copy_, clone, contiguous
step()
torch.cuda.synchronize()
torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = o_copy, o_clone, o_contig
for s, n in sites.most_common(25):
    print(n, s)
