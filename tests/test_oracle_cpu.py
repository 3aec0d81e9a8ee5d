"""CPU: pin the oracle (oracle/*.py, fp32 restatement) to the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Tolerance 1e-3 (north-star's fp32 bound); observed deviations are ~1e-5."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLD, build_state, load_case

import oracle.swin as OS


def _close(got, ref, tol=1e-3, what=""):
    got = got.detach().float(); ref = torch.as_tensor(ref).float()
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.3g})"


def _grads(P, names):
    return torch.cat([(P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])).reshape(-1) for n in names])


@pytest.mark.parametrize("tag", ["swin_block_even", "swin_block_odd", "swin_block_s0", "swin_block_s3",
                                 "swin_block_nofusion", "swin_block_video", "swin_block_audio", "swin_block_wide64",
                                 "swin_block_wide96"])
def test_swin_block_matches_reference(tag):
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="swin_block", T=cfg["T"], res=cfg["res"])
    for n in names:
        P["blk." + n].requires_grad_(True)
    # structural known-answers from the reference's integer buffers
    ws, shift = OS.block_geometry(cfg["res"], cfg["res"], 7, cfg["shift"])
    assert torch.equal(OS.relative_position_index(ws), torch.as_tensor(z["rel_index"]))
    m = OS.shift_attn_mask(cfg["res"], cfg["res"], ws, shift)
    if m is None:
        assert z["attn_mask"].size == 0
    else:
        assert torch.equal(m, torch.as_tensor(z["attn_mask"]))
    from params import seeded_tensor
    BT, N, C = cfg["B"] * cfg["T"], cfg["res"] ** 2, cfg["dim"]
    v = seeded_tensor((BT, N, C), cfg["seed"] + 1).requires_grad_(True)
    a = seeded_tensor((BT, N, C), cfg["seed"] + 2).requires_grad_(True)
    gv, ga = seeded_tensor((BT, N, C), cfg["seed"] + 3), seeded_tensor((BT, N, C), cfg["seed"] + 4)
    kw = dict(H=cfg["res"], W=cfg["res"], T=cfg["T"], heads=cfg["heads"], window_size=7, shift_size=cfg["shift"],
              t_attn=cfg["t_attn"], mode=cfg["mode"])
    if cfg["mode"] in ("fusion_adapt", "multimodal_adapt_no_fusion"):
        ov, oa = OS.swin_block(P, "blk", (v, a), **kw)
        ((ov * gv).sum() + (oa * ga).sum()).backward()
        _close(ov, z["out_v"], what="out_v"); _close(oa, z["out_a"], what="out_a")
        _close(v.grad, z["din_v"], what="din_v"); _close(a.grad, z["din_a"], what="din_a")
    else:
        x = v if cfg["mode"] == "video_adapt" else a
        o = OS.swin_block(P, "blk", x, **kw)
        (o * gv).sum().backward()
        _close(o, z["out"], what="out"); _close(x.grad, z["din"], what="din")
    _close(_grads(P, ["blk." + n for n in names]), z["grads"], what="param grads")


@pytest.mark.parametrize("tag", ["swin_tiny_fusion", "swin_tiny_multimodal", "swin_tiny_videoonly"])
def test_swin_tiny_model_matches_reference(tag):
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for n in names:
        P[n].requires_grad_(True)
    from params import seeded_tensor
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2)
    logits = OS.swin_forward(P, a, v, cfg, cfg["mode"])
    tgt = torch.softmax(seeded_tensor((B * T, cfg["label_dim"]), cfg["seed"] + 3, 2.0), -1)
    loss = OS.soft_target_cross_entropy(logits, tgt)
    loss.backward()
    _close(logits, z["logits"], what="logits")
    _close(loss.reshape(1), z["loss"], tol=1e-4, what="loss")
    _close(_grads(P, names), z["grads"], what="grads")


def test_swin_b_full_model_matches_reference():
    """The headline configuration (Swin-B + STG-CMA, ftmode='fusion', AVE shape, B=1): logits, loss, per-tensor grad norms."""
    z, cfg, shapes, names = load_case("swin_b_fusion")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for n in names:
        P[n].requires_grad_(True)
    from params import seeded_tensor
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2)
    logits = OS.swin_forward(P, a, v, cfg, "fusion")
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1)
    loss = OS.soft_target_cross_entropy(logits, tgt)
    loss.backward()
    _close(logits, z["logits"], what="logits")
    _close(loss.reshape(1), z["loss"], tol=1e-4, what="loss")
    norms = torch.stack([P[n].grad.norm() if P[n].grad is not None else torch.zeros(()) for n in names])
    _close(norms, z["grad_norms"], what="grad norms")
    _close(_grads(P, names)[::97], z["grads_sample"], what="grad sample")
    # known answers quoted by the reference's launcher comment (AVE/run_swin_adapt_ave29.sh:55 "5.6M")
    assert list(z["n_params"]) == [92345613, 5599957, 1063965]


def test_swin_b_refinit_model_matches_reference():
    """Same model at the reference's own initialisation scale (the fixture behind the GPU test's absolute 1e-2 logit bound)."""
    from params import refinit_state, seeded_tensor
    z, cfg, shapes, names = load_case("swin_b_fusion_refinit")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"], state_fn=refinit_state)
    for n in names:
        P[n].requires_grad_(True)
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2)
    logits = OS.swin_forward(P, a, v, cfg, "fusion")
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1)
    OS.soft_target_cross_entropy(logits, tgt).backward()
    assert float((logits.detach() - torch.as_tensor(z["logits"])).abs().max()) <= 1e-4
    norms = torch.stack([P[n].grad.norm() if P[n].grad is not None else torch.zeros(()) for n in names])
    _close(norms, z["grad_norms"], what="grad norms")


def test_structure_contract():
    with open(os.path.join(GOLD, "structure.json")) as f:
        st = json.load(f)
    assert st["swin_b_fusion"]["n_trainable"] == 5599957 and st["swin_b_fusion"]["n_head"] == 1063965
    assert st["swin_l_fusion"]["n_trainable"] == 19026857
    assert st["vit_b_fusion"]["n_trainable"] == 6185909
    keys = [k for k, _, _ in st["swin_b_fusion"]["keys"]]
    assert "layers.0.blocks.0.attn.temporal_position_bias_table_audio" in keys
    assert "layers.0.blocks.1.attn_mask" in keys and "layers.0.blocks.0.attn_mask" not in keys
    assert "layers.2.blocks.17.S_Adapter2_Audio.D_fc2.weight" in keys


def test_cosine_scheduler_matches_reference():
    z = np.load(os.path.join(GOLD, "cosine_scheduler.npz"))
    np.testing.assert_allclose(OS.cosine_scheduler(5e-5, 2e-6, 20, 3339, warmup_epochs=2).numpy(), z["t1"], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(OS.cosine_scheduler(5e-6, 2e-6, 20, 3339, warmup_epochs=2).numpy(), z["t2"], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(OS.cosine_scheduler(1e-4, 2e-6, 3, 7, warmup_epochs=1).numpy(), z["t3"], rtol=1e-12, atol=1e-18)


# ----------------------------------------------------------------------------------------------- ViT (CLIP) path
import oracle.vit as OV  # noqa: E402


@pytest.mark.parametrize("tag", ["vit_block_cfg1", "vit_block_small"])
def test_vit_block_matches_reference(tag):
    """BASELINE.json config 1: single ViT-B/16 block + cross-modal adapters, 196 audio + 196 video tokens, fp32 CPU."""
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="vit_block")
    for n in names:
        P["blk." + n].requires_grad_(True)
    BT, d, nv, na, st = cfg["B"] * cfg["T"], cfg["d"], cfg["nv"], cfg["na"], cfg["stride"]
    v = seeded_tensor((nv, BT, d), cfg["seed"] + 1).permute(1, 0, 2).contiguous().requires_grad_(True)
    a = seeded_tensor((na, BT, d), cfg["seed"] + 2).permute(1, 0, 2).contiguous().requires_grad_(True)
    gv = seeded_tensor((nv, BT, d), cfg["seed"] + 3).permute(1, 0, 2)
    ga = seeded_tensor((na, BT, d), cfg["seed"] + 4).permute(1, 0, 2)
    ov, oa = OV.vit_block(P, "blk", (v, a), T=cfg["T"], heads=cfg["heads"], mode=cfg["mode"])
    ((ov * gv).sum() + (oa * ga).sum()).backward()
    _close(ov.permute(1, 0, 2)[::st], z["out_v"], what="out_v")
    _close(oa.permute(1, 0, 2)[::st], z["out_a"], what="out_a")
    _close(v.grad.permute(1, 0, 2)[::st], z["din_v"], what="din_v")
    _close(a.grad.permute(1, 0, 2)[::st], z["din_a"], what="din_a")
    stats = torch.stack([ov.sum(), ov.abs().sum(), oa.sum(), oa.abs().sum()])
    _close(stats / stats.abs().max(), torch.as_tensor(z["stats"]) / float(np.abs(z["stats"]).max()), tol=1e-4, what="stats")
    _close(_grads(P, ["blk." + n for n in names]), z["grads"], what="param grads")


def test_vit_tiny_model_matches_reference():
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("vit_tiny_fusion")
    P = build_state(shapes, cfg["seed"], kind="vit")
    for n in names:
        P[n].requires_grad_(True)
    B, T = cfg["B"], cfg["T"]
    a = seeded_tensor((B, T, 102, 128), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2)
    logits = OV.vit_forward(P, a, v, cfg, "fusion")
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1)
    loss = OS.soft_target_cross_entropy(logits, tgt)
    loss.backward()
    _close(logits, z["logits"], what="logits")
    _close(loss.reshape(1), z["loss"], tol=1e-4, what="loss")
    _close(_grads(P, names), z["grads"], what="grads")
    assert list(z["n_params"][1:]) == [sum(P[n].numel() for n in names), sum(P[n].numel() for n in names if n.startswith("mlp_head"))]


# ----------------------------------------------------------------------------------------------- AVS / AVQA backbones
def test_avs_backbone_matches_reference():
    """multi-scale video taps (before each downsample, last one through norm) + norm(a), and the gradients of
    every trainable backbone tensor for seeded upstream gradients on all five outputs."""
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("avs_tiny_backbone")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for n in names:
        P[n].requires_grad_(True)
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, T, 3, 224, 224), cfg["seed"] + 2)
    out = OS.swin_backbone(P, a, v, cfg)
    taps, f_a = out["taps"], out["f_a"]
    for i in range(3):
        _close(taps[i][:, ::7], z[f"tap{i}"], what=f"tap{i}")
    _close(taps[3], z["tap3"], what="tap3")
    _close(f_a, z["f_a"], what="f_a")
    loss = sum((t * seeded_tensor(t.shape, cfg["seed"] + 10 + i)).sum() for i, t in enumerate(taps)) + \
        (f_a * seeded_tensor(f_a.shape, cfg["seed"] + 20)).sum()
    loss.backward()
    g, ref = _grads(P, names), torch.as_tensor(z["grads"])
    assert float((g - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))


def test_avqa_backbone_matches_reference():
    """(v, a, v_nega) through every block and downsample; the negative stream is the frozen Swin block."""
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("avqa_tiny_backbone")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for n in names:
        P[n].requires_grad_(True)
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, T, 3, 224, 224), cfg["seed"] + 2)
    vn = seeded_tensor((B, T, 3, 224, 224), cfg["seed"] + 3)
    out = OS.swin_backbone(P, a, v, cfg, v_nega=vn)
    for k in ("f_v", "f_a", "f_nega"):
        _close(out[k], z[k], what=k)
    ((out["f_v"] * seeded_tensor(out["f_v"].shape, cfg["seed"] + 10)).sum() +
     (out["f_a"] * seeded_tensor(out["f_a"].shape, cfg["seed"] + 11)).sum() + out["f_nega"].sum()).backward()
    g, ref = _grads(P, names), torch.as_tensor(z["grads"])
    assert float((g - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))


def test_avqa_full_model_matches_reference():
    """backbone + QA head (question LSTM, grounding on the positive / negative clip, single-query attentions, fusion
    MLPs) of the reference's SwinTransformer2D_Adapter_AVQA, outputs and every trainable gradient."""
    import oracle.avqa_head as OH
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("avqa_full_tiny")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for n in names:
        P[n].requires_grad_(True)
    B, T, seed = cfg["B"], cfg["num_frames"], cfg["seed"]
    a = seeded_tensor((B, T, 224, 224), seed + 1, 0.5)
    v = seeded_tensor((B, T, 3, 224, 224), seed + 2)
    vn = seeded_tensor((B, T, 3, 224, 224), seed + 3)
    question = torch.as_tensor(z["question"])
    assert torch.equal(question, torch.randint(0, 93, (B, 14), generator=torch.Generator().manual_seed(seed + 4)))
    out_qa, mp, mn = OH.avqa_forward(P, a, v, vn, question, cfg)
    _close(out_qa, z["out_qa"], what="out_qa")
    _close(mp, z["out_match_posi"], what="out_match_posi")
    _close(mn, z["out_match_nega"], what="out_match_nega")
    ((out_qa * seeded_tensor(out_qa.shape, seed + 5)).sum() + (mp * seeded_tensor(mp.shape, seed + 6)).sum() +
     (mn * seeded_tensor(mn.shape, seed + 7)).sum()).backward()
    norms = torch.stack([(P[n].grad if P[n].grad is not None else torch.zeros(())).norm() for n in names])
    ref_norms = torch.as_tensor(z["grad_norms"])
    assert float(((norms - ref_norms).abs() / ref_norms.clamp_min(1e-6)).max()) <= 2e-3, "per-tensor gradient norms"
    g, ref = _grads(P, names)[::197], torch.as_tensor(z["grads_sample"])
    assert float((g - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))
