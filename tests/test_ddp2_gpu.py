"""-m gpu: the REAL data-parallel path under a 2-rank process group (AVE/traintest_adapt_ave29.py:32-35 is what
it replaces).  Two fresh child processes (started, not exec'ed; both on cuda:0, gloo -- RCCL refuses two ranks on one device) run
the HIP forward / backward of a small Swin AVE model, an AVS model and an AVQA model with ddp.attach; the parent then checks that
every trainable .grad (a) is identical on the two ranks and (b) equals the mean of the two ranks' own single-clip gradients."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
def test_two_ranks_hip_backward_gradients_are_averaged(gpu, tmp_path):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "helpers", "ddp2_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=800)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
    for tag in ("swin_ave", "swin_avs", "swin_avqa"):
        r0 = torch.load(tmp_path / f"{tag}_r0.pt")
        r1 = torch.load(tmp_path / f"{tag}_r1.pt")
        assert r0["synced"].keys() == r1["synced"].keys() and len(r0["synced"]) > 0
        worst = 0.0
        for n in r0["synced"]:
            assert torch.equal(r0["synced"][n], r1["synced"][n]), f"{tag}: ranks disagree on {n} after the exchange"
            mean = 0.5 * (r0["local"][n] + r1["local"][n])
            # the two backward passes of a rank are separate launches (fp32 atomics in the gate / bias-table gradients): equal up to
            # summation order
            err = float((r0["synced"][n] - mean).norm()) / max(float(mean.norm()), 1e-12)
            worst = max(worst, err)
            assert err <= 1e-3 or float((r0["synced"][n] - mean).abs().max()) <= 1e-7, f"{tag}: {n} is not the rank mean (rel {err:.2e})"
        # the ranks saw different clips, so their own gradients differ: the exchange did something
        assert any(not torch.equal(r0["local"][n], r1["local"][n]) for n in r0["local"]), tag
        print(f"{tag}: {len(r0['synced'])} trainable tensors identical across ranks; worst rel deviation from the rank mean {worst:.2e}")


@pytest.mark.timeout(900)
def test_two_ranks_replayed_step_forms_equal_the_eager_ddp_step(gpu, tmp_path):
    """bench.py's N > 1 step forms under a real 2-rank group (gloo, both ranks on this GPU): recipe.capture_train_step_ddp's two graphs and
    recipe.capture_train_step_mb(sync=...)'s micro-batch graphs, K replays each, against 1 + K eager DDP steps from the same start
    (tests/helpers/ddp2_step_worker.py).  After every form the trainable parameters are BIT-IDENTICAL across the two ranks (they applied the same
    averaged gradients), and equal to the eager form's: the two-graph form to fp32 atomics' summation order (gate / bias-table gradients), the
    micro-batch form to the re-association of two half-batch gradient sums."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "helpers", "ddp2_step_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=800)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
    r0, r1 = torch.load(tmp_path / "steps_r0.pt"), torch.load(tmp_path / "steps_r1.pt")
    names = list(r0["eager"])
    assert len(names) > 50
    for form in ("eager", "ddp2g", "mb"):
        for n in names:
            assert torch.equal(r0[form][n], r1[form][n]), f"{form}: the ranks' parameters differ after the steps ({n})"
    moved = sum(float((r0["eager"][n] - r0["start"][n]).abs().max()) > 0 for n in names)
    assert moved >= 0.9 * len(names), f"only {moved} of {len(names)} tensors moved in the eager form"
    # Adam moves every element by ~lr per step whatever the gradient's size, so an element whose gradient is summation-order noise can land a
    # whole step away: compare element-wise, relative to the eager form's own movement, at the 99th percentile
    upd = torch.cat([(r0["eager"][n] - r0["start"][n]).reshape(-1).abs() for n in names])
    for form, tol in (("ddp2g", 0.05), ("mb", 0.25)):
        dev = torch.cat([(r0[form][n] - r0["eager"][n]).reshape(-1).abs() for n in names])
        rel = dev / upd.clamp_min(1e-12)
        q99 = float(torch.quantile(rel[upd > 0][:: max(1, int((upd > 0).sum()) // 1_000_000)], 0.99))
        same = sum(torch.equal(r0[form][n], r0["eager"][n]) for n in names)
        print(f"{form}: 99th percentile of |parameter - eager form's| / |eager form's movement| = {q99:.2e}; {same} of {len(names)} tensors bit-identical")
        assert q99 <= tol, (form, q99)
    assert abs(r0["ddp2g_loss"] - r1["ddp2g_loss"]) > 0            # different clips per rank: the exchange, not the data, made the parameters equal
    # round 6: the micro-batch form with a task-head bucket (AVQA: backbone arena + ONE bucket of the head's summed gradients)
    qn = list(r0["avqa_eager"])
    assert any(n.startswith("avqatask_") for n in qn) and any(not n.startswith("avqatask_") for n in qn)
    for form in ("avqa_eager", "avqa_mb"):
        for n in qn:
            assert torch.equal(r0[form][n], r1[form][n]), f"{form}: the ranks' parameters differ after the steps ({n})"
    upd = torch.cat([(r0["avqa_eager"][n] - r0["avqa_start"][n]).reshape(-1).abs() for n in qn])
    dev = torch.cat([(r0["avqa_mb"][n] - r0["avqa_eager"][n]).reshape(-1).abs() for n in qn])
    rel = dev / upd.clamp_min(1e-12)
    q99 = float(torch.quantile(rel[upd > 0][:: max(1, int((upd > 0).sum()) // 1_000_000)], 0.99))
    print(f"avqa mb: 99th percentile of |parameter - eager form's| / |eager form's movement| = {q99:.2e}")
    assert q99 <= 0.3, q99
    head_moved = sum(float((r0["avqa_mb"][n] - r0["avqa_start"][n]).abs().max()) > 0 for n in qn if n.startswith("avqatask_"))
    assert head_moved >= 0.8 * sum(n.startswith("avqatask_") for n in qn), "the task head's parameters did not move in the micro-batch form"
