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
