"""Child process of tests/test_ddp2_gpu.py::test_two_ranks_replayed_step_forms_equal_the_eager_ddp_step: one data-parallel rank (cuda:0, 2-rank
gloo group) that runs the SAME three optimizer steps of a small Swin AVE model in the three forms bench.py's N > 1 path can take:
  eager   forward, loss, backward (ddp.GradSync averages the arena inside backward), FusedAdam step            -- the reference form
  ddp2g   recipe.capture_train_step_ddp: graph(forward + backward, GradSync deferred) -> eager all-reduce -> graph(scatter + Adam)
  mb      recipe.capture_train_step_mb(sync=...): two micro-batch graphs on two streams -> join graph -> eager all-reduce -> Adam graph
Each form starts from the same parameters and optimizer state (one eager warm-up step is part of every form's capture, so the reference
runs 1 + K eager steps).  Saves the trainable parameters after the steps.  usage: ddp2_step_worker.py <rank> <world> <port> <out_dir>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")

import torch  # noqa: E402
import stgcma  # noqa: F401,E402
from stgcma import ddp, recipe  # noqa: E402
from stgcma.model import Swin_AVE  # noqa: E402

gpu = torch.device("cuda", 0)
torch.cuda.set_device(0)
r, _, w = ddp.init_from_env("gloo")
assert (r, w) == (rank, world)
B, T, K = 2, 2, 2

g = torch.Generator().manual_seed(200 + rank)                 # every rank its own clips
a = (torch.randn(B, T, 224, 224, generator=g) * 0.5).to(gpu)
v = torch.randn(B, 3, T, 224, 224, generator=g).to(gpu)
y = torch.softmax(torch.randn(B, T, 29, generator=g) * 2, -1).to(gpu)
loss_fn = torch.nn.CrossEntropyLoss()

torch.manual_seed(0)                                          # same parameters on every rank
m = Swin_AVE.SwinTransformer2D_Adapter_New(label_dim=29, num_frames=T, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], ftmode="fusion",
                                           adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
with torch.no_grad():
    for n, p in m.named_parameters():
        if "D_fc2" in n or "gate_" in n:
            p.normal_(0.0, 0.05)
m = m.to(gpu).eval()                                          # eval: no DropPath / Dropout draws, every form sees the same function
recipe.apply_freeze(m)
ddp.broadcast_parameters(m)
sync = ddp.attach(m)
state0 = {k: t.detach().clone() for k, t in m.state_dict().items()}
names = [n for n, p in m.named_parameters() if p.requires_grad]
d = dict(m.named_parameters())


def fresh():
    m.load_state_dict(state0)                                 # bumps the parameters' version: the bf16 shadows are re-cast
    for p in m.parameters():
        p.grad = None
    return recipe.build_optimizer(m, lr=1e-3, head_lr=1e-2, capturable=False)


def fwd_bwd(opt):
    loss = loss_fn(m(a, v, "fusion"), y.reshape(-1, y.shape[-1]))
    opt.zero_grad()
    loss.backward()
    return loss


def snap():
    torch.cuda.synchronize()
    return {n: d[n].detach().float().cpu().clone() for n in names}


out = {}
opt = fresh()
for _ in range(1 + K):
    fwd_bwd(opt)
    opt.step()
out["eager"] = snap()

opt = fresh()
replay, static_loss, how = recipe.capture_train_step_ddp(lambda: fwd_bwd(opt), opt, sync, warmup=1)
assert how == "two_graphs", how
for _ in range(K):
    replay()
out["ddp2g"] = snap()
out["ddp2g_loss"] = float(static_loss.detach())
replay.release()
del replay

opt = fresh()


def fwd_loss(a_, v_, y_):
    return loss_fn(m(a_, v_, "fusion"), y_.reshape(-1, y_.shape[-1]))


replay, static_loss, how = recipe.capture_train_step_mb(fwd_loss, (a, v, y), opt, splits=2, sync=sync, warmup=1, require_overlap=False)
assert "2 micro-batch graphs" in how and "all-reduce" in how, how
for _ in range(K):
    replay()
out["mb"] = snap()
out["mb_loss"] = float(static_loss.detach())
replay.release()

# after release the GradSync averages inside backward again: one more eager step must still run
opt = fresh()
fwd_bwd(opt)
opt.step()
torch.cuda.synchronize()
out["start"] = {n: state0[n].detach().float().cpu().clone() for n in names}

# ---- round 6: the micro-batch form with a TASK-HEAD bucket (the AVQA model: its head's gradients do not pass through the backbone's arena)
del replay, opt, sync, m, d
torch.cuda.empty_cache()
from stgcma.model import Swin_AVQAModel_V1  # noqa: E402
vt = v.permute(0, 2, 1, 3, 4).contiguous()
vn = vt.flip(1)
q = torch.randint(0, 93, (B, 14), generator=g).to(gpu)
ans = torch.randint(0, 42, (B,), generator=g).to(gpu)
torch.manual_seed(0)
m = Swin_AVQAModel_V1.SwinTransformer2D_Adapter_AVQA(pretrained=None, num_frames=T, embed_dim=192, depths=[2, 2, 2, 2], num_heads=[6, 12, 24, 48],
                                                     ftmode="fusion", adapter_mlp_ratio=[0.25, 0.125, 0.125, 0.0625])
with torch.no_grad():
    for n, p in m.named_parameters():
        if "D_fc2" in n or "gate_" in n:
            p.normal_(0.0, 0.05)
m = m.to(gpu).eval()
recipe.apply_freeze(m)
ddp.broadcast_parameters(m)
sync = ddp.attach(m)
assert sync.extra, "the AVQA head's parameters must travel in the task-head bucket"
state0 = {k: t.detach().clone() for k, t in m.state_dict().items()}
names = [n for n, p in m.named_parameters() if p.requires_grad]
d = dict(m.named_parameters())
ce = torch.nn.CrossEntropyLoss()


def q_loss(a_, vt_, vn_, q_, ans_):
    out_qa, mp, mn = m(a_, vt_, vn_, q_, "fusion")
    return ce(out_qa, ans_) + 0.5 * (mp.float().square().mean() + mn.float().square().mean())


opt = fresh()
for _ in range(1 + K):
    loss = q_loss(a, vt, vn, q, ans)
    opt.zero_grad()
    loss.backward()
    opt.step()
out["avqa_eager"] = snap()
out["avqa_start"] = {n: state0[n].detach().float().cpu().clone() for n in names}
del loss                                                     # the eager step's autograd graph (AccumulateGrad nodes bound to this stream) must not survive into the capture
import gc  # noqa: E402
gc.collect()
print(f"rank {rank}: avqa eager steps done", flush=True)
opt = fresh()
replay, static_loss, how = recipe.capture_train_step_mb(q_loss, (a, vt, vn, q, ans), opt, splits=2, sync=sync, warmup=1, require_overlap=False)
assert "2 micro-batch graphs" in how and "all-reduce" in how, how
for _ in range(K):
    replay()
out["avqa_mb"] = snap()
replay.release()
print(f"rank {rank}: avqa micro-batch steps done", flush=True)
torch.save(out, os.path.join(out_dir, f"steps_r{rank}.pt"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print(f"rank {rank} done", flush=True)
