"""-m gpu: the HIP-backed CLIP-ViT + STG-CMA modules (stg-cma_amd/model/CLIP_AVE.py) against golden vectors generated
from the reference (BASELINE.json configs 1 and 2 shapes).  Same metrics / tolerances as test_model_gpu.py."""
import os

import numpy as np
import pytest
import torch

from golden_util import build_state, load_case

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
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


def _load_into(module, P, prefix=""):
    sd = module.state_dict()
    for k in sd:
        if sd[k].is_floating_point():
            sd[k] = P[prefix + k]
    module.load_state_dict(sd, strict=True)


def _apply_freeze(module):
    from stgcma import recipe
    names = []
    for n, p in module.named_parameters():
        p.requires_grad = recipe.is_trainable(n)
        if p.requires_grad:
            names.append(n)
    return names


@pytest.mark.parametrize("tag", ["vit_block_cfg1", "vit_block_small"])
def test_vit_block_matches_reference(stg, gpu, tag):
    """Config 1: one ViT-B/16 ResidualAttentionBlock (heads=8 => head dim 96) with cross-modal adapters, 196 + 196 tokens."""
    from stgcma.model import CLIP_AVE as Cm
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="vit_block")
    blk = Cm.ResidualAttentionBlock(cfg["d"], cfg["heads"], None, 0.5, 1, cfg["T"], 0.0, mode=cfg["mode"]).eval()
    _load_into(blk, P, "blk.")
    blk = blk.to(gpu)
    assert _apply_freeze(blk) == names
    BT, d, nv, na, st = cfg["B"] * cfg["T"], cfg["d"], cfg["nv"], cfg["na"], cfg["stride"]
    v = seeded_tensor((nv, BT, d), cfg["seed"] + 1).permute(1, 0, 2).reshape(-1, d)
    a = seeded_tensor((na, BT, d), cfg["seed"] + 2).permute(1, 0, 2).reshape(-1, d)
    gv = seeded_tensor((nv, BT, d), cfg["seed"] + 3).permute(1, 0, 2).reshape(-1, d)
    ga = seeded_tensor((na, BT, d), cfg["seed"] + 4).permute(1, 0, 2).reshape(-1, d)
    X = torch.cat([v, a]).to(gpu).requires_grad_(True)
    out = blk(X, n_tok=(nv, na))
    ov = out[:BT * nv].view(BT, nv, d).permute(1, 0, 2)
    oa = out[BT * nv:].view(BT, na, d).permute(1, 0, 2)
    _cmp(ov[::st], z["out_v"], f"{tag} out_v")
    _cmp(oa[::st], z["out_a"], f"{tag} out_a")
    out.backward(torch.cat([gv, ga]).to(gpu))
    dv = X.grad[:BT * nv].view(BT, nv, d).permute(1, 0, 2)
    da = X.grad[BT * nv:].view(BT, na, d).permute(1, 0, 2)
    _cmp(dv[::st], z["din_v"], f"{tag} din_v", max_rel=1.3e-2, l2_rel=1.2e-2)          # 1.5 x measured (8.3e-3 / 7.7e-3)
    _cmp(da[::st], z["din_a"], f"{tag} din_a", max_rel=1.3e-2, l2_rel=1.2e-2)
    dct = dict(blk.named_parameters())
    off = 0
    for n in names:
        k = dct[n].numel()
        ref = z["grads"][off:off + k]
        off += k
        if "gate_" in n:
            # a scalar that sums ~BT*(nv+na)*d_h*2 signed bf16-rounded products: its noise floor is
            # ~2^-8 * sqrt(#terms) * rms|term| (heavy cancellation), not a fraction of its own value
            terms = BT * (nv + na) * dct["S_Adapter.D_fc1.weight"].shape[0] * 2
            tol = 6e-2 * abs(float(ref[0])) + 7e-3 * terms ** 0.5        # (observed up to 5.5e-3 sqrt(terms) on gate_a of cfg1, |ref| = 4.4)
            err = abs(float(dct[n].grad) - float(ref[0]))
            _report.append(f"{tag} grad[{n}]: err {err:.3g} (ref {float(ref[0]):.3g}, tol {tol:.3g})")
            assert err <= tol, f"{tag} grad[{n}]: {float(dct[n].grad)} vs {float(ref[0])} (tol {tol})"
        elif np.abs(ref).max() > 0:
            _cmp(dct[n].grad, ref, f"{tag} grad[{n}]", max_rel=1.7e-2, l2_rel=1.4e-2)     # 1.5 x measured (1.1e-2 / 8.9e-3)


def test_vit_tiny_fusion_model_matches_reference(stg, gpu):
    """MM_CLIP_AVE (2 layers of ViT-B/16 width, 197 video + 49 audio tokens): logits, loss, every trainable gradient
    including temporal_embedding(_audio) and ln_post."""
    from stgcma.model import CLIP_AVE as Cm
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("vit_tiny_fusion")
    P = build_state(shapes, cfg["seed"], kind="vit")
    m = Cm.MM_CLIP_AVE(label_dim=29, layers=cfg["layers"], num_video_frames=cfg["T"], embed_dim=cfg["d"], patch_size=16,
                       heads=cfg["heads"], pretrained=None, ftmode="fusion").eval()
    _load_into(m, P)
    m = m.to(gpu)
    assert _apply_freeze(m) == names
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n_train == int(z["n_params"][1])
    B, T = cfg["B"], cfg["T"]
    a = seeded_tensor((B, T, 102, 128), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2).to(gpu)
    logits = m(a, v, "fusion")
    assert logits.dtype == F32 and tuple(logits.shape) == (B * T, 29)
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1).to(gpu)
    loss = torch.nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    _cmp(logits, z["logits"], "vit_tiny logits")
    err = float((logits.detach().cpu() - torch.as_tensor(z["logits"])).abs().max())
    _report.append(f"vit_tiny logits max abs err {err:.3e} (|logits| max {float(np.abs(z['logits']).max()):.3g})")
    assert abs(float(loss.detach()) - float(z["loss"][0])) <= 1e-2
    dct = dict(m.named_parameters())
    off = 0
    for n in names:
        k = dct[n].numel()
        ref = z["grads"][off:off + k]
        off += k
        g = dct[n].grad
        assert g is not None, n
        if "gate_" in n:
            terms = B * T * (197 + 49) * 48 * 2
            tol = 8e-2 * abs(float(ref[0])) + 5e-3 * terms ** 0.5
            assert abs(float(g) - float(ref[0])) <= tol, f"vit_tiny grad[{n}]: {float(g)} vs {float(ref[0])} (tol {tol})"
        elif np.abs(ref).max() > 0:
            _cmp(g, ref, f"vit_tiny grad[{n}]", max_rel=2.3e-2, l2_rel=2.1e-2)       # 1.5 x measured (1.48e-2 / 1.38e-2, profiles/r03_parity_report.txt)


def test_vit_b_full_depth_model_matches_reference(stg, gpu):
    """BASELINE config 2 at FULL depth (12 layers, ViT-B/16 width, heads 8), reference-initialisation scale, fixture from the reference's
    MM_CLIP_AVE: north_star's ABSOLUTE bound on the logits (<= 1e-2 max-abs), per-tensor gradient norms and the strided gradient sample."""
    from stgcma.model import CLIP_AVE as Cm
    from params import seeded_tensor, refinit_state
    z, cfg, shapes, names = load_case("vit_b12_fusion_refinit")
    P = build_state(shapes, cfg["seed"], kind="vit", state_fn=refinit_state)
    m = Cm.MM_CLIP_AVE(label_dim=29, layers=cfg["layers"], num_video_frames=cfg["T"], embed_dim=cfg["d"], patch_size=16,
                       heads=cfg["heads"], pretrained=None, ftmode="fusion").eval()
    _load_into(m, P)
    m = m.to(gpu)
    assert _apply_freeze(m) == names
    B, T = cfg["B"], cfg["T"]
    a = seeded_tensor((B, T, 102, 128), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2).to(gpu)
    logits = m(a, v, "fusion")
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1).to(gpu)
    loss = torch.nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    err = float((logits.detach().cpu() - torch.as_tensor(z["logits"])).abs().max())
    scale = float(np.abs(z["logits"]).max())
    _report.append(f"vit_b12_refinit logits max abs err {err:.3e} (|logits| max {scale:.3g})")
    assert err <= 1e-2, f"logits deviate {err:.3e} max-abs (bound 1e-2; scale {scale:.3g})"
    assert abs(float(loss.detach()) - float(z["loss"][0])) <= 5e-3
    dct = dict(m.named_parameters())
    ref_norms = np.asarray(z["grad_norms"])
    worst = 0.0
    for n, rn in zip(names, ref_norms):
        assert dct[n].grad is not None and torch.isfinite(dct[n].grad).all(), n
        if rn > 1e-3 and "gate_" not in n:
            worst = max(worst, abs(float(dct[n].grad.norm()) - float(rn)) / float(rn))
    flat = torch.cat([dct[n].grad.reshape(-1).float().cpu() for n in names])[::97]
    ref = torch.as_tensor(z["grads_sample"])
    e_l2 = float((flat - ref).norm() / ref.norm())
    _report.append(f"vit_b12_refinit gradients: strided-sample relL2={e_l2:.3e} worst per-tensor norm deviation={worst:.3e}")
    assert worst <= 6.7e-3 and e_l2 <= 9.1e-3, (worst, e_l2)      # 1.5 x measured (4.4e-3 / 6.0e-3; logits 3.6e-3 max-abs, round 4)


def test_vit_train_mode_and_no_cpu_fallback(stg, gpu):
    from stgcma.model import CLIP_AVE as Cm
    m = Cm.MM_CLIP_AVE(label_dim=29, layers=2, num_video_frames=2, embed_dim=256, patch_size=16, heads=4, ftmode="fusion")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 102, 128), torch.zeros(1, 3, 2, 224, 224), "fusion")
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "D_fc2" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "gate_" in n:
                p.fill_(0.3)
    m = m.to(gpu).train()
    _apply_freeze(m)
    a = torch.randn(2, 2, 102, 128, device=gpu); v = torch.randn(2, 3, 2, 224, 224, device=gpu)
    l1 = m(a, v, "fusion"); l2 = m(a, v, "fusion")
    assert torch.isfinite(l1).all() and float((l1 - l2).abs().max()) > 0          # DropPath / Dropout active
    l1.sum().backward()
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
