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


def _task_worker(rank, world, port, out_dir, kind):
    """One rank of the AVS / AVQA data-parallel check: the mirror nn.Module supplies the parameters and ddp.attach's hooks, the
    ORACLE supplies the arithmetic (the HIP path cannot run on CPU), each rank sees ITS clip.  Adapter gradients travel through the
    flat arena like on the GPU (ops.GradArena + GradSync.allreduce_), task-head gradients through the end-of-backward bucket."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import stgcma  # noqa: F401
    from stgcma import ddp, ops
    ddp.init_from_env("gloo")
    torch.set_num_threads(2)
    m, P, names, run = _task_setup(kind)
    sync = ddp.attach(m)
    inside = set(m._flat_tensors()[0])
    head = [n for n in names if n not in inside]
    assert head and all(n.startswith(("avstask_", "avqatask_")) for n in head)
    assert len(sync.extra) == len([n for n, _ in m.named_parameters() if n not in inside])
    run(P, slice(rank, rank + 1)).backward()                      # hooks fire; the end-of-backward callback averages the head bucket
    n_head = sum(P[n].numel() for n in head)
    assert sync.last_numel == n_head, (sync.last_numel, n_head)
    back = [n for n in names if n in inside]
    need = {n: True for n in back}
    arena = ops.GradArena(back, P, need, torch.device("cpu"))
    for n in back:
        arena.view(n, P[n]).copy_(P[n].grad)
    sync.allreduce_(arena.flat, arena.n_real)
    assert sync.last_numel == sum(P[n].numel() for n in back)
    out = {n: arena.view(n, P[n]).clone() for n in back}
    out.update({n: P[n].grad.clone() for n in head})
    torch.save(out, os.path.join(out_dir, f"{kind}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _task_setup(kind):
    """(mirror model on CPU, {name: tensor} over its parameters + the oracle's index buffers, trainable names, run(P, clips) -> loss)."""
    import stgcma  # noqa: F401
    from stgcma import recipe
    import oracle.swin as OS
    from params import float_shapes, seeded_state, seeded_tensor
    T = 2
    cfg = dict(embed_dim=192 if kind == "avqa" else 32, depths=[2, 2, 2, 2], num_heads=[6, 12, 24, 48] if kind == "avqa" else [1, 2, 4, 8],
               num_frames=T, adapter_mlp_ratio=[0.25, 0.125, 0.125, 0.0625], img_size=224, window_size=7)
    if kind == "avs":
        from stgcma.model import Swin_AVSModel_Base as M
        m = M.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=T, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                                 num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"],
                                                 channel=256, opt=None, config=None, vis_dim=[64, 128, 320, 512],
                                                 tpavi_stages=[0, 1, 2, 3], tpavi_vv_flag=False, tpavi_va_flag=True).eval()
    else:
        from stgcma.model import Swin_AVQAModel_V1 as M
        m = M.SwinTransformer2D_Adapter_AVQA(pretrained=None, num_frames=T, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                             num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"]).eval()
    sd = m.state_dict()
    sd.update(seeded_state(float_shapes(sd), 9100))
    m.load_state_dict(sd)
    with torch.no_grad():                            # BatchNorm running statistics: a seeded variance must stay positive
        gg = torch.Generator().manual_seed(9150)
        for n, b in m.named_buffers():
            if n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=gg) + 0.5)
            elif n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=gg) * 0.1)
    recipe.apply_freeze(m)
    P = dict(m.named_parameters())
    P.update({k: v for k, v in m.named_buffers() if not k.endswith("attn_mask")})
    for k in list(P):
        if k.endswith("attn.relative_position_bias_table"):
            pre = k[: -len("relative_position_bias_table")]
            P[pre + "relative_position_index"] = OS.relative_position_index(7)
            P[pre + "t_relative_coords"] = OS.temporal_relative_index(T)
            P[pre + "t_relative_coords_a"] = OS.temporal_relative_index(T)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    a = seeded_tensor((2, T, 224, 224), 31, 0.5)
    v = seeded_tensor((2, T, 3, 224, 224), 32)
    if kind == "avs":
        import oracle.avs_decoder as OD
        up = seeded_tensor((2 * T, 1, 224, 224), 33, 1e-2)

        def run(P, sl):                              # eval-mode BatchNorm: no cross-sample statistic, so the mean over clips shards
            pred, _, _ = OD.avs_forward(P, a[sl], v[sl], cfg, bn_training=False)
            return (pred * up[sl.start * T:sl.stop * T]).sum() / (sl.stop - sl.start)
    else:
        import oracle.avqa_head as OH
        vn = seeded_tensor((2, T, 3, 224, 224), 34)
        q = torch.randint(0, 93, (2, 14), generator=torch.Generator().manual_seed(35))
        g1, g2, g3 = seeded_tensor((2, 42), 36), seeded_tensor((2 * T, 2), 37), seeded_tensor((2 * T, 2), 38)

        def run(P, sl):
            qa, mp, mn = OH.avqa_forward(P, a[sl], v[sl], vn[sl], q[sl], cfg)
            fr = slice(sl.start * T, sl.stop * T)
            return ((qa * g1[sl]).sum() + (mp * g2[fr]).sum() + (mn * g3[fr]).sum()) / (sl.stop - sl.start)
    return m, P, names, run


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["avs", "avqa"])
def test_two_rank_task_models_average_every_trainable_gradient(tmp_path, kind):
    """BASELINE configs 4 / 5 under data parallelism: after one backward per rank EVERY trainable tensor -- adapters through the
    arena, `avstask_*` / `avqatask_*` through ddp.attach's end-of-backward bucket -- equals the gradient of the mean loss over the
    global batch (the task heads used to be left un-reduced)."""
    port = _free_port()
    mp.spawn(_task_worker, args=(2, port, str(tmp_path), kind), nprocs=2, join=True)
    g0 = torch.load(tmp_path / f"{kind}0.pt")
    g1 = torch.load(tmp_path / f"{kind}1.pt")
    assert any(n.startswith(("avstask_", "avqatask_")) for n in g0)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), f"ranks disagree on {n} after the all-reduce"
    m, P, names, run = _task_setup(kind)
    assert set(names) == set(g0)
    run(P, slice(0, 2)).backward()
    worst = 0.0
    for n in names:
        ref = P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])
        scale = max(1e-6, float(ref.abs().max()))
        worst = max(worst, float((g0[n] - ref).abs().max()) / scale)
        assert float((g0[n] - ref).abs().max()) <= 2e-3 * scale + 1e-7, n   # fp32 summation order (B = 1 twice vs B = 2); an un-reduced gradient is off by O(1)
    print(f"{kind}: worst relative deviation from the global-batch gradient {worst:.2e}")


def test_attach_wires_arena_and_head_bucket(stg):
    """ddp.attach: the GradSync sits in the launch plan (what SwinModelFn / SwinBackboneFn.backward consult) and watches exactly the
    parameters that do not travel through the arena."""
    import torch.distributed as dist
    from stgcma import ddp
    from stgcma.model import Swin_AVE as S, Swin_AVSModel_Base as A
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        m = S.SwinTransformer2D_Adapter_New(label_dim=5, embed_dim=32, depths=[2, 2], num_heads=[1, 2], num_frames=2,
                                            ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25])
        s = ddp.attach(m)
        assert m._plan().ddp is s and s.extra == []
        a = A.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=2, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8],
                                                 ftmode="fusion", adapter_mlp_ratio=[0.25] * 4)
        s = ddp.attach(a)
        watched = {id(p) for p in s.extra}
        for n, p in a.named_parameters():
            assert (id(p) in watched) == n.startswith("avstask_"), n
        # direct call (outside a backward pass): accounted immediately, padding not counted
        flat = torch.ones(12)
        s.allreduce_(flat, n_real=10)
        assert s.last_numel == 10 and s.calls == 1
    finally:
        dist.destroy_process_group()
