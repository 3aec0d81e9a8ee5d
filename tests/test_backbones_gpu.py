"""-m gpu: the AVS / AVQA backbone mirrors (stg-cma_amd/model/Swin_AVSModel.py, Swin_AVQAModel_V1.py) against golden
vectors generated from the reference's own AVS / AVQA model classes.  Same metrics / tolerances as test_model_gpu.py."""
import os

import numpy as np
import pytest
import torch

from golden_util import build_state, load_case

pytestmark = pytest.mark.gpu
_report = []


def _cmp(got, ref, what, max_rel=2e-2, l2_rel=1e-2):
    got = got.detach().float().cpu().reshape(-1)
    ref = torch.as_tensor(np.asarray(ref)).float().reshape(-1)
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    scale = max(float(ref.abs().max()), 1e-6)
    e_max = float((got - ref).abs().max()) / scale
    e_l2 = float((got - ref).norm() / max(float(ref.norm()), 1e-12))
    _report.append(f"{what}: max/scale={e_max:.3e} relL2={e_l2:.3e} scale={scale:.3g}")
    assert e_max <= max_rel and e_l2 <= l2_rel, f"{what}: max/scale={e_max:.3e} relL2={e_l2:.3e} (scale {scale:.3g})"


@pytest.fixture(scope="module", autouse=True)
def _dump_report():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write("\n".join(_report) + "\n")


def _build(cls, cfg, shapes, gpu):
    from stgcma import recipe
    m = cls(pretrained=None, num_frames=cfg["num_frames"], embed_dim=cfg["embed_dim"], depths=cfg["depths"],
            num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"],
            **({"t_relative": cfg["t_relative"]} if "t_relative" in cfg else {})).eval()
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    sd = m.state_dict()
    for k in sd:
        if sd[k].is_floating_point() and not k.endswith("attn_mask") and k in P:
            sd[k] = P[k]
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu)
    names = []
    backbone = ("patch_embed.", "patch_embed_audio.", "layers.", "norm.", "temporal_embedding")     # the goldens of this file are backbone-only
    for n, p in m.named_parameters():
        p.requires_grad = recipe.is_trainable(n) and n.startswith(backbone)
        if p.requires_grad:
            names.append(n)
    return m, names


def _check_grads(m, names, ref, tag):
    dct = dict(m.named_parameters())
    off = 0
    # a gate gradient is ONE number, the sum of rows x d_h signed bf16-rounded products: its noise floor is a share of the LARGEST gate
    # gradient of the model, not of its own (possibly small) value -- measured deviations 0-8 % of the value, up to 1.5 % of that maximum
    gmax, o2 = 0.0, 0
    for n in names:
        k = dct[n].numel()
        if "gate_" in n:
            gmax = max(gmax, abs(float(ref[o2])))
        o2 += k
    for n in names:
        k = dct[n].numel()
        r = ref[off:off + k]
        off += k
        assert dct[n].grad is not None, n
        if np.abs(r).max() > 0:
            if "gate_" in n:
                assert abs(float(dct[n].grad) - float(r[0])) <= 8e-2 * abs(float(r[0])) + max(0.5, 1.5e-2 * gmax), f"{tag} grad[{n}]"
            elif "temporal_position_bias_table" in n:
                # (2T-1) x heads numbers, each the sum of B*N*T^2 signed bf16-rounded terms (N = 3136 in stage 0): a
                # cancellation noise floor like the gates', not a fraction of the tensor's own (small) value
                _cmp(dct[n].grad, r, f"{tag} grad[{n}]", max_rel=1.5e-1, l2_rel=1.2e-1)
            else:
                # small tensors (16 .. 512 values) behind 8 bf16 blocks: one realisation of the rounding noise moves a single
                # tensor by +-40 % of its deviation (see NOISE in test_model_gpu.py); 6e-2 leaves ~20 % headroom
                _cmp(dct[n].grad, r, f"{tag} grad[{n}]", max_rel=8e-2, l2_rel=6e-2)
    assert off == ref.size


@pytest.mark.parametrize("case", ["avs_tiny_backbone", "avs_tiny_backbone_tabs"])      # _tabs: t_relative=False (absolute temporal embeddings), B = 2
def test_avs_backbone_matches_reference(stg, gpu, case):
    """a19: taps before every downsample + norm'd last stage + norm(a); gradients arrive through all five outputs."""
    from stgcma.model import Swin_AVSModel
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(case)
    m, mine = _build(Swin_AVSModel.SwinTransformer2D_Adapter_AVS, cfg, shapes, gpu)
    assert mine == names
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, T, 3, 224, 224), cfg["seed"] + 2).to(gpu)
    ms, f_a = m.forward_features(a, v)
    assert [tuple(t.shape) for t in ms] == [(B * T, 3136, 32), (B * T, 784, 64), (B * T, 196, 128), (B * T, 49, 256)]
    for i in range(3):
        _cmp(ms[i][:, ::7], z[f"tap{i}"], f"avs tap{i}")
    # last-stage features: 8 bf16 blocks deep on deliberately "hot" unit-gain parameters -> bounds as for the logits of
    # test_model_gpu.py's hot fixture (rel-L2 2e-2, max 3e-2 of the tensor's scale)
    _cmp(ms[3], z["tap3"], "avs tap3", max_rel=3e-2, l2_rel=2e-2)
    _cmp(f_a, z["f_a"], "avs f_a", max_rel=3e-2, l2_rel=2e-2)
    loss = sum((t * seeded_tensor(t.shape, cfg["seed"] + 10 + i).to(gpu)).sum() for i, t in enumerate(ms)) + \
        (f_a * seeded_tensor(f_a.shape, cfg["seed"] + 20).to(gpu)).sum()
    loss.backward()
    _check_grads(m, names, z["grads"], "avs")


@pytest.mark.parametrize("case", ["avqa_tiny_backbone", "avqa_tiny_backbone_tabs"])    # _tabs: t_relative=False, B = 2
def test_avqa_backbone_matches_reference(stg, gpu, case):
    """a18: the negative clip rides through every block as the plain frozen Swin block (forward-only)."""
    from stgcma.model import Swin_AVQAModel_V1
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(case)
    m, mine = _build(Swin_AVQAModel_V1.SwinTransformer2D_Adapter_AVQA, cfg, shapes, gpu)
    assert mine == names
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, T, 3, 224, 224), cfg["seed"] + 2).to(gpu)
    vn = seeded_tensor((B, T, 3, 224, 224), cfg["seed"] + 3).to(gpu)
    f_v, f_a, f_n = m.forward_features(a, v, vn)
    assert not f_n.requires_grad
    _cmp(f_v, z["f_v"], "avqa f_v", max_rel=3e-2, l2_rel=2e-2)
    _cmp(f_a, z["f_a"], "avqa f_a", max_rel=3e-2, l2_rel=2e-2)
    _cmp(f_n, z["f_nega"], "avqa f_nega", max_rel=3e-2, l2_rel=2e-2)
    ((f_v * seeded_tensor(f_v.shape, cfg["seed"] + 10).to(gpu)).sum() +
     (f_a * seeded_tensor(f_a.shape, cfg["seed"] + 11).to(gpu)).sum()).backward()
    _check_grads(m, names, z["grads"], "avqa")


def test_backbones_train_mode(stg, gpu):
    """DropPath active on all three streams: finite, stochastic, gradients for every trainable tensor."""
    from stgcma.model import Swin_AVQAModel_V1
    from stgcma import recipe
    m = Swin_AVQAModel_V1.SwinTransformer2D_Adapter_AVQA(pretrained=None, num_frames=2, embed_dim=32, depths=[2, 2, 2, 2],
                                                 num_heads=[1, 2, 4, 8], ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.125])
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "D_fc2" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "gate_" in n:
                p.fill_(0.3)
    m = m.to(gpu).train()
    for n, p in m.named_parameters():
        p.requires_grad = recipe.is_trainable(n) and not n.startswith("avqatask_")     # forward_features: the backbone only
    a = torch.randn(1, 2, 224, 224, device=gpu); v = torch.randn(1, 2, 3, 224, 224, device=gpu); vn = torch.randn(1, 2, 3, 224, 224, device=gpu)
    o1 = m.forward_features(a, v, vn); o2 = m.forward_features(a, v, vn)
    assert all(torch.isfinite(t).all() for t in o1)
    assert float((o1[2] - o2[2]).abs().max()) > 0                      # drop_path on the negative stream
    (o1[0].sum() + o1[1].sum()).backward()
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
