"""-m gpu: the micro-batched training step (recipe.capture_train_step_mb: two HIP graphs on two streams + a join graph) against the plain
full-batch step of the reference loop (AVE/traintest_adapt_ave29.py:149-164: forward, CrossEntropyLoss on float targets, zero_grad, backward,
optimizer.step) -- same loss, same gradients, same parameters after Adam -- and the launcher-free `bench.py --gpus 2`."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny(gpu, seed=0):
    from stgcma import recipe
    from stgcma.model import Swin_AVE as S
    torch.manual_seed(seed)
    m = S.SwinTransformer2D_Adapter_New(label_dim=29, patch_size=[1, 4, 4], pretrained=None, ftmode="fusion", embed_dim=32, depths=[2, 2, 2, 2],
                                        num_heads=[1, 2, 4, 8], num_frames=2, adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625], window_size=7)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "D_fc2" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "gate_" in n:
                p.fill_(0.3)
    m = m.to(gpu).eval()                       # eval: no DropPath / Dropout draw, so the two step forms see the same function
    opt = recipe.build_optimizer(m, lr=1e-3, head_lr=1.0)
    return m, opt


def test_microbatched_step_equals_full_batch_step(stg, gpu):
    from stgcma import recipe
    B, T = 4, 2
    g = torch.Generator().manual_seed(11)
    a = (torch.randn(B, T, 224, 224, generator=g) * 0.5).to(gpu)
    v = torch.randn(B, 3, T, 224, 224, generator=g).to(gpu)
    y = torch.softmax(torch.randn(B, T, 29, generator=g) * 2, -1).to(gpu)
    loss_fn = torch.nn.CrossEntropyLoss()

    # the reference loop's step
    m0, o0 = _tiny(gpu)
    loss0 = recipe.train_step(m0, o0, loss_fn, a, v, y, "fusion")
    names = [n for n, p in m0.named_parameters() if p.requires_grad]
    g0 = {n: p.grad.detach().clone() for n, p in m0.named_parameters() if p.requires_grad}
    p0 = {n: p.detach().clone() for n, p in m0.named_parameters() if p.requires_grad}

    # the same step as two concurrent micro-batches (warm-up steps run on a scratch copy of the state: the capture needs lazily built
    # tables, and its eager warm-up is a real optimizer step)
    m1, o1 = _tiny(gpu)
    sd = {k: t.detach().clone() for k, t in m1.state_dict().items()}

    def fwd_loss(a_, v_, y_):
        return loss_fn(m1(a_, v_, "fusion"), y_.reshape(-1, y_.shape[-1]))

    replay, static_loss, how = recipe.capture_train_step_mb(fwd_loss, (a, v, y), o1, splits=2, warmup=1, require_overlap=False)
    assert "2 micro-batch graphs" in how
    m1.load_state_dict(sd)                     # undo the warm-up step: same start as the reference run
    for st in o1.state.values():               # fresh Adam state (moments and step counters are views of the optimizer's flat buffers)
        st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    o1._st.zero_()
    replay()
    torch.cuda.synchronize()
    assert abs(float(static_loss) - float(loss0)) <= 2e-3 * max(1.0, abs(float(loss0)))
    d1 = dict(m1.named_parameters())
    num = den = 0.0
    for n in names:
        assert d1[n].grad is not None, n
        num += float((d1[n].grad - g0[n]).double().pow(2).sum())
        den += float(g0[n].double().pow(2).sum())
    assert (num / den) ** 0.5 <= 2e-3, (num / den) ** 0.5          # fp32 re-association + bf16 tile paths at half the rows
    # Adam's first step moves every element by ~lr whatever the gradient's size: compare the UPDATES where the gradient is not noise
    worst = 0.0
    for n in names:
        big = g0[n].abs() > 1e-3 * g0[n].abs().max()
        if big.any():
            worst = max(worst, float((d1[n].detach() - p0[n])[big].abs().max()))
    assert worst <= 2.5e-4, worst                                   # lr = 1e-3: updates of +-1e-3 agree to a quarter of a step at worst
    # Further replays keep training on FRESH weights: the loss a replay reports must be the loss of the parameters it started from.  (Each
    # micro-batch graph re-casts the bf16 shadows of the trainable weights into its own arena at its start; a graph that read the other's
    # arena while that one rewrites it -- or last step's shadows -- would report another loss.)
    for g in o1.param_groups:
        g["lr"] = 2e-2                                    # make the weights move visibly between the steps
    o1._push_hyper()
    for it in range(3):
        with torch.no_grad():
            want = float(fwd_loss(a, v, y))               # eager, full batch, the parameters as they are now
        replay()
        torch.cuda.synchronize()
        got = float(static_loss.detach())
        assert abs(got - want) <= 2e-3 * max(1.0, abs(want)), (it, got, want)
    assert torch.isfinite(static_loss).all()


def test_bench_launches_its_own_ranks(stg, gpu):
    """`python bench.py --gpus 2` typed WITHOUT a launcher: bench.py starts the two ranks itself (children of a parent that never touches
    the GPU) and rank 0 prints one JSON line with n_gpus = 2.  gloo, both ranks on this box's one GPU (RCCL refuses two ranks per device)."""
    env = dict(os.environ, STG_DDP_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import gc
    gc.collect()
    torch.cuda.empty_cache()                   # the children share this process's GPU: hand the cached blocks of earlier tests back first
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--batch", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/self_launch_failure.txt", "w") as f:
            f.write(r.stdout[-4000:] + "\n==== stderr\n" + r.stderr[-12000:])
    assert r.returncode == 0, [ln for ln in r.stderr.splitlines() if "Error" in ln or "error" in ln][-6:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["config"]["global_batch"] == 4
    # which step form the timed region used is decided per run from two timed steps of each (all ranks together): two ranks time-slicing ONE
    # GPU usually lose on the replayed form, real one-GPU-per-rank runs are expected not to -- either way the line must say what it measured
    assert "graphs" in d["config"]["step"] or ("eager" in d["config"]["step"] and d["config"]["step_forms_measured"]), d["config"]["step"]


def test_bench_watchdog_prints_the_eager_line_when_the_replayed_form_never_returns(stg, gpu):
    """N > 1 safety net: RCCL has never run under the replayed step form on this pool's hardware, so bench.py times the eager form before it
    captures anything and arms a watchdog around pass 2; when that fires (here: forced after 1 s) rank 0 prints ONE complete line from the
    eager measurement, says so in config.step, and every rank exits with code 4 (a code gpurun itself never returns): a hang is a failure, the line keeps the number."""
    env = dict(os.environ, STG_DDP_BACKEND="gloo", OMP_NUM_THREADS="4", STG_BENCH_WATCHDOG_S="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--batch", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) <= 6144 and r.stdout.rstrip().splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "WATCHDOG" in d["config"]["step"] and d["roofline"] is not None


def test_bench_n1_capture_failure_falls_back_to_the_eager_measurement(stg, gpu):
    """N = 1: when the micro-batch capture raises, the complete eager measurement is the line (it used to die on a None loss)."""
    env = dict(os.environ, STG_BENCH_FAIL_CAPTURE="1", OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "2", "--batch", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.rstrip().splitlines()[-1]
    assert len(last) <= 6144
    d = json.loads(last)
    assert "capture failed" in d["config"]["step"] and d["value"] > 0 and d["final_loss"] == d["final_loss"]
    assert d["roofline"]["frac"] > 0 and d["detail"]
