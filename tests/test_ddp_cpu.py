"""CPU, world_size 2, gloo: the data-parallel path.  The HIP model itself cannot run here (no CPU fallback), so the
collective logic (ddp.GradSync on the flat gradient arena) is exercised with real processes, and the sharding math --
average of per-rank mean-loss gradients == gradient of the mean loss over the global batch -- is checked with the oracle."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import ROOT, build_state, load_case


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import stgcma  # noqa: F401
    from stgcma import ddp, ops
    import oracle.swin as OS
    from params import seeded_tensor
    r, lr, w = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.set_num_threads(2)
    # ---- the collective: flat arena averaged in place
    sync = ddp.GradSync()
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    sync.allreduce_(flat)
    assert torch.allclose(flat, torch.arange(10, dtype=torch.float32) * 1.5)
    assert sync.calls == 1 and sync.last_numel == 10
    # ---- sharded clips: each rank runs the oracle on ITS clip and fills an arena with its trainable gradients
    z, cfg, shapes, names = load_case("swin_tiny_fusion")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for n in names:
        P[n].requires_grad_(True)
    T = cfg["num_frames"]
    a = seeded_tensor((2, T, 224, 224), 11, 0.5)
    v = seeded_tensor((2, 3, T, 224, 224), 12)
    tgt = torch.softmax(seeded_tensor((2 * T, 29), 13, 2.0), -1)
    sl = slice(rank, rank + 1)
    logits = OS.swin_forward(P, a[sl], v[sl], cfg, "fusion")
    OS.soft_target_cross_entropy(logits, tgt[rank * T:(rank + 1) * T]).backward()
    need = {n: True for n in names}
    arena = ops.GradArena(names, P, need, torch.device("cpu"))
    for n in names:
        arena.view(n, P[n]).copy_(P[n].grad)
    sync.allreduce_(arena.flat)
    torch.save({n: arena.view(n, P[n]).clone() for n in names}, os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_average_equals_global_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(tmp_path / "g0.pt")
    g1 = torch.load(tmp_path / "g1.pt")
    for n in g0:
        assert torch.equal(g0[n], g1[n]), f"ranks disagree on {n} after the all-reduce"
    # single-process reference: both clips in one batch
    import oracle.swin as OS
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("swin_tiny_fusion")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for n in names:
        P[n].requires_grad_(True)
    T = cfg["num_frames"]
    a = seeded_tensor((2, T, 224, 224), 11, 0.5)
    v = seeded_tensor((2, 3, T, 224, 224), 12)
    tgt = torch.softmax(seeded_tensor((2 * T, 29), 13, 2.0), -1)
    OS.soft_target_cross_entropy(OS.swin_forward(P, a, v, cfg, "fusion"), tgt).backward()
    for n in names:
        ref = P[n].grad
        scale = max(1e-6, float(ref.abs().max()))
        assert float((g0[n] - ref).abs().max()) <= 1e-4 * scale + 1e-7, n


def test_attach_sets_plan_hook(stg, monkeypatch):
    """ddp.attach wires the GradSync into the model's launch plan (what SwinModelFn.backward consults)."""
    from stgcma import ddp
    from stgcma.model import Swin_AVE as S

    class FakeSync:
        def __init__(self, group=None):
            self.group = group
    monkeypatch.setattr(ddp, "GradSync", FakeSync)
    m = S.SwinTransformer2D_Adapter_New(label_dim=5, embed_dim=32, depths=[2, 2], num_heads=[1, 2], num_frames=2,
                                        ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25])
    s = ddp.attach(m)
    assert m._plan().ddp is s
