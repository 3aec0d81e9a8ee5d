"""-m gpu: every kernel launch preceded by an LDS full of NaN bit patterns (STG_LDS_POISON=1, stg_debug_poison_lds): a kernel that reads an LDS
byte it never wrote -- harmless while the CU's previous tenant left finite data, e.g. padded key rows whose probabilities are exactly zero:
0 x NaN = NaN -- produces a non-finite loss or gradient.  Found in round 4 by a 2-rank rehearsal on ONE GPU, where the other process's kernels
leave their data in the LDS (a NaN loss once in ~8 runs).  Runs in a child process: the poisoning proxy is installed when the library loads."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os
sys.path.insert(0, os.environ["STG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["STG_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["STG_ROOT"], "tests", "golden"))
import torch
if os.environ.get("STG_NAN_EMPTY"):
    import tools.nan_fill_empty  # noqa: F401   every torch.empty starts as NaN
import stgcma
from stgcma import recipe
import bench
dev = torch.device("cuda:0")
bad = []
for workload, B in (("swin_b", 2), ("swin_b", 3), ("vit_b", 2), ("swin_l", 1)):
    torch.manual_seed(0)
    model = bench.build_model(torch, dev, workload)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = recipe.FusedAdam(params, lr=1e-4)
    a, v, _ = bench.synth_batch(torch, B, dev, 0, workload)
    for it in range(2):
        out = model(a, v, "fusion")
        loss = out.float().square().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        ng = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        if not bool(torch.isfinite(out).all()) or ng:
            bad.append((workload, B, it, bool(torch.isfinite(out).all()), len(ng), ng[:3]))
    del model, opt
    torch.cuda.empty_cache()
from stgcma import _lib
print("BAD", bad, "poisoned launches", _lib._PoisonedLds.launches)
sys.exit(1 if bad or (os.environ.get("STG_LDS_POISON") and _lib._PoisonedLds.launches < 1000) else 0)
'''


def test_model_steps_survive_poisoned_lds(stg, gpu):
    env = dict(os.environ, STG_LDS_POISON="1", STG_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "poisoned launches" in r.stdout


def test_model_steps_survive_nan_filled_allocations(stg, gpu):
    """The same steps with every torch.empty / empty_like / new_empty NaN-filled (tools/nan_fill_empty.py): a kernel that reads a global-memory
    byte nobody wrote would produce a non-finite output or gradient."""
    env = dict(os.environ, STG_NAN_EMPTY="1", STG_ROOT=ROOT)
    env.pop("STG_LDS_POISON", None)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
