"""-m gpu: FusedAdam (stg_adam_multi) against torch.optim.Adam -- the reference's optimizer (AVE/traintest_adapt_ave29.py:63-69:
Adam, betas (0.95, 0.999), weight_decay 5e-7, two parameter groups, a learning rate rewritten every iteration :139-144)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(gpu, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1,), (3,), (7, 5), (1025,), (64, 128), (300, 333), (8, 1, 3), (2, 2)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(gpu)) for s in shapes]


def _grads(ps, seed, skip=()):
    g = torch.Generator().manual_seed(seed)
    for i, p in enumerate(ps):
        p.grad = None if i in skip else (torch.randn(p.shape, generator=g) * (10.0 ** ((i % 3) - 1))).to(p.device)


def test_fused_adam_matches_torch_adam(stg, gpu):
    from stgcma import recipe
    ref_p, my_p = _params(gpu, 0), _params(gpu, 0)
    mk = lambda ps: [{"params": ps[:5], "lr": 1e-2}, {"params": ps[5:], "lr": 1e-3}]
    ref = torch.optim.Adam(mk(ref_p), weight_decay=5e-3, betas=(0.95, 0.999))
    mine = recipe.FusedAdam(mk(my_p), weight_decay=5e-3, betas=(0.95, 0.999))
    for it in range(12):
        skip = (6,) if it < 2 else ()                       # a parameter without a gradient is left alone (and joins later)
        _grads(ref_p, 100 + it, skip); _grads(my_p, 100 + it, skip)
        for opt in (ref, mine):                             # the loop's per-iteration learning rate
            opt.param_groups[0]["lr"] = 1e-2 * (1 + it) / 4
            opt.param_groups[1]["lr"] = 1e-3 * (12 - it) / 12
        v0 = my_p[0]._version
        ref.step(); mine.step()
        assert my_p[0]._version > v0                        # the bf16 weight shadows key on the version counter
        if True:
            for a, b in zip(ref_p, my_p):
                assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (it, tuple(a.shape), float((a - b).abs().max()))
    # torch's per-parameter counter starts when the parameter first receives a gradient
    assert float(ref.state[ref_p[6]]["step"]) == float(mine.state[my_p[6]]["step"]) == 10
    st_r, st_m = ref.state[ref_p[1]], mine.state[my_p[1]]
    assert float(st_r["step"]) == float(st_m["step"]) == 12
    assert torch.allclose(st_r["exp_avg"], st_m["exp_avg"], rtol=1e-5, atol=1e-8)
    assert torch.allclose(st_r["exp_avg_sq"], st_m["exp_avg_sq"], rtol=1e-5, atol=1e-10)


def test_fused_adam_state_dict_interchanges_with_torch(stg, gpu):
    from stgcma import recipe
    a_p, b_p, c_p = _params(gpu, 1), _params(gpu, 1), _params(gpu, 1)
    a = torch.optim.Adam([{"params": a_p}], lr=3e-3, weight_decay=1e-4, betas=(0.95, 0.999))
    b = recipe.FusedAdam([{"params": b_p}], lr=3e-3, weight_decay=1e-4, betas=(0.95, 0.999))
    for it in range(3):
        _grads(a_p, it); a.step()
    with torch.no_grad():
        for x, y, w in zip(a_p, b_p, c_p):
            y.copy_(x); w.copy_(x)
    b.load_state_dict(copy.deepcopy(a.state_dict()))        # torch -> fused
    c = torch.optim.Adam([{"params": c_p}], lr=3e-3, weight_decay=1e-4, betas=(0.95, 0.999))
    c.load_state_dict(copy.deepcopy(b.state_dict()))        # fused -> torch
    for it in range(3, 6):
        _grads(a_p, it); _grads(b_p, it); _grads(c_p, it)
        a.step(); b.step(); c.step()
    for x, y, w in zip(a_p, b_p, c_p):
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-7) and torch.allclose(x, w, rtol=2e-6, atol=2e-7)


def test_fused_adam_load_resets_parameters_the_checkpoint_does_not_cover(stg, gpu):
    """torch.optim.Adam.load_state_dict leaves a parameter that the loaded state does not cover WITHOUT state (it restarts from zero
    moments and step 0); FusedAdam, whose moments live in flat buffers, must zero them."""
    from stgcma import recipe
    a_p, b_p = _params(gpu, 3), _params(gpu, 3)
    a = torch.optim.Adam([{"params": a_p}], lr=3e-3, betas=(0.95, 0.999))
    b = recipe.FusedAdam([{"params": b_p}], lr=3e-3, betas=(0.95, 0.999))
    skip = (2, 5)                                           # these two never see a gradient before the checkpoint is taken
    _grads(a_p, 0, skip=skip); a.step()
    ckpt = copy.deepcopy(a.state_dict())
    assert all(i not in ckpt["state"] for i in skip)
    for it in range(1, 4):                                  # both optimizers run on (FusedAdam on every tensor), then load the old checkpoint
        _grads(a_p, it); _grads(b_p, it)
        a.step(); b.step()
    with torch.no_grad():
        for x, y in zip(a_p, b_p):
            y.copy_(x)
    a.load_state_dict(copy.deepcopy(ckpt)); b.load_state_dict(copy.deepcopy(ckpt))
    for it in range(4, 7):
        _grads(a_p, it); _grads(b_p, it)
        a.step(); b.step()
    for i, (x, y) in enumerate(zip(a_p, b_p)):
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-7), f"tensor {i} ({'not ' if i in skip else ''}covered by the checkpoint)"


def test_capture_train_step_bumps_versions_and_rejects_zero_warmup(stg, gpu):
    """replay() must bump the trainable parameters' version counters (FusedAdam.step's own bump is host code a replay does not run):
    ops.shadow keys its bf16 weight shadows on the version, so an eager forward after replays would otherwise read stale shadows
    (ADVICE r2).  warmup < 1 is rejected: lazily built tables / shadows would stay out of the graph's steady state."""
    from stgcma import recipe
    ps = _params(gpu, 4)
    opt = recipe.FusedAdam([{"params": ps}], lr=1e-2, betas=(0.95, 0.999))
    target = [torch.randn_like(p) for p in ps]

    def step():
        loss = sum(((p - t) ** 2).sum() for p, t in zip(ps, target))
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss

    with pytest.raises(ValueError):
        recipe.capture_train_step(step, warmup=0)
    replay, static_loss = recipe.capture_train_step(step, warmup=2)
    v0 = [p._version for p in ps]
    before = [p.detach().clone() for p in ps]
    l0 = None
    for _ in range(3):
        replay()
        l0 = float(static_loss.detach()) if l0 is None else l0
    torch.cuda.synchronize()
    assert all(p._version > v for p, v in zip(ps, v0)), "replay() left the version counters untouched"
    assert any(not torch.equal(p.detach(), b) for p, b in zip(ps, before)) and float(static_loss.detach()) < l0


def test_fused_adam_in_a_hip_graph(stg, gpu):
    """The whole update replays from a captured graph: step counters and bias corrections live on the device."""
    from stgcma import recipe
    ref_p, my_p = _params(gpu, 2), _params(gpu, 2)
    ref = torch.optim.Adam([{"params": ref_p}], lr=1e-2, betas=(0.95, 0.999))
    mine = recipe.FusedAdam([{"params": my_p}], lr=1e-2, betas=(0.95, 0.999))
    _grads(ref_p, 7); _grads(my_p, 7)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        mine.step()
    torch.cuda.current_stream().wait_stream(s)
    ref.step()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mine.step()
    ref.step()                                              # capture does not execute: replay = step 2
    for it in range(3):
        g.replay()
        if it:
            ref.step()
    torch.cuda.synchronize()
    for a, b in zip(ref_p, my_p):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), float((a - b).abs().max())


def test_fused_adam_rejects_cpu_parameters(stg):
    from stgcma import recipe
    with pytest.raises(RuntimeError):
        recipe.FusedAdam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3)
