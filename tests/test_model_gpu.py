"""-m gpu: the HIP-backed nn.Modules (stg-cma_amd/model) against the golden vectors generated from the reference
(tests/golden/make_golden.py) and against the fp32 oracle on the same seeded inputs.

Tolerances (bf16 activations, fp32 accumulate): logits |err| <= 1e-2 (the north-star's bf16 bound); tensors are compared
by max-abs error relative to the reference tensor's max-abs value (<= 2e-2) and by relative L2 error (<= 1e-2).
"""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLD, build_state, load_case

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
_report = []


# NOISE: every bound below holds ONE realisation of the bf16 rounding noise.  Swapping a kernel pair for an fp32-equivalent
# fused kernel (csrc/upln.hip: x equal to 4e-6, 579 of 32 M bf16 outputs one ulp apart, tools/dbg_upln.py) moved single small
# tensors by up to +-40 % of their deviation while the mean over the 393 compared tensors stayed put (-2 %; 184 closer, 96
# farther).  Scalars / 16-element biases and the tiny models' logits therefore carry ~20 % headroom over the first realisation.
# Per-tensor gradient bounds of the block fixtures = 1.5 x the worst deviation measured on MI355X (profiles/r03_parity_report.txt:
# per case the worst trainable tensor and the worst scalar gate; a bf16 pipeline against the fp32 reference), never below 1.5e-2.
# (max-abs / tensor max, relative L2) for tensors, one number for the scalar gates.
_BLOCK_GRAD_MEASURED = {
    "swin_block_audio": ((9.1e-3, 8.6e-3), 0.0), "swin_block_video": ((8.3e-3, 8.0e-3), 0.0),
    "swin_block_even": ((1.59e-2, 1.18e-2), 9.5e-3), "swin_block_odd": ((9.7e-3, 8.5e-3), 1.01e-2),
    "swin_block_s0": ((9.3e-3, 8.2e-3), 3.6e-3), "swin_block_s3": ((3.41e-2, 2.54e-2), 4.73e-2),   # gate: 2.4e-2 .. 4.7e-2 over builds (a sum with heavy cancellation)
    "swin_block_nofusion": ((1.34e-2, 1.13e-2), 0.0), "swin_block_wide64": ((2.19e-2, 1.66e-2), 1.16e-2),
    "swin_block_wide96": ((8.5e-3, 8.5e-3), 2.0e-2),
}


def _block_grad_bounds(tag, scalar, ref_abs=None):
    (mx, l2), gate = _BLOCK_GRAD_MEASURED[tag]
    if scalar:
        # a gate's gradient is ONE number summing ~10^5 .. 10^6 signed bf16-rounded products: its error is an absolute floor (measured
        # 0.16 .. 0.84 over the fixtures and two builds of the LayerNorm path, whatever |ref| = 3.4 .. 124 is), not a fraction of its value
        tol = (1.2e-2 * ref_abs + 0.5) / max(ref_abs, 1e-6)
        return tol, tol
    return max(1.5 * mx, 1.5e-2), max(1.5 * l2, 1.5e-2)


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
        if sd[k].is_floating_point() and not k.endswith("attn_mask"):
            sd[k] = P[prefix + k]
    module.load_state_dict(sd, strict=True)


def _apply_freeze(module):
    from params import is_trainable
    names = []
    for n, p in module.named_parameters():
        p.requires_grad = is_trainable(n)
        if p.requires_grad:
            names.append(n)
    return names


def _flat_grads(module, names):
    d = dict(module.named_parameters())
    return torch.cat([(d[n].grad if d[n].grad is not None else torch.zeros_like(d[n])).reshape(-1).float().cpu() for n in names])


@pytest.mark.parametrize("tag", ["swin_block_video", "swin_block_audio"])
def test_single_stream_block_matches_reference(stg, gpu, tag):
    """'video_adapt' / 'audio_adapt' blocks: adapter parallel to the MLP (Swin_AVE.py:394-488)."""
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="swin_block", T=cfg["T"], res=cfg["res"])
    blk = S.SwinTransformerBlock(dim=cfg["dim"], input_resolution=(cfg["res"], cfg["res"]), num_frames=cfg["T"],
                                 num_heads=cfg["heads"], window_size=7, shift_size=cfg["shift"], t_attn=cfg["t_attn"],
                                 adapter_mlp_ratio=cfg["ratio"], mode=cfg["mode"]).eval()
    _load_into(blk, P, "blk.")
    blk = blk.to(gpu)
    assert _apply_freeze(blk) == names
    BT, N, C = cfg["B"] * cfg["T"], cfg["res"] ** 2, cfg["dim"]
    x = seeded_tensor((BT, N, C), cfg["seed"] + (1 if cfg["mode"] == "video_adapt" else 2))
    gx = seeded_tensor((BT, N, C), cfg["seed"] + 3)
    X = x.reshape(-1, C).to(gpu).requires_grad_(True)          # fp32 residual-stream input
    out = blk(X)
    assert out.dtype == torch.float32
    _cmp(out, z["out"], f"{tag} out")
    out.backward(gx.reshape(-1, C).to(gpu))
    _cmp(X.grad, z["din"], f"{tag} din", max_rel=1.5e-2, l2_rel=1.2e-2)                # 1.5 x measured (9.9e-3 / 6.8e-3)
    d = dict(blk.named_parameters())
    off = 0
    for n in names:
        k = d[n].numel()
        ref = z["grads"][off:off + k]
        off += k
        if np.abs(ref).max() > 0:
            mr, lr = _block_grad_bounds(tag, k == 1, float(np.abs(ref).max()))
            _cmp(d[n].grad, ref, f"{tag} grad[{n}]", max_rel=mr, l2_rel=lr)
        else:
            assert d[n].grad is None or float(d[n].grad.abs().max()) == 0


@pytest.mark.parametrize("tag,mode", [("swin_tiny_multimodal", "multimodal"), ("swin_tiny_videoonly", "videoonly"),
                                      ("swin_tiny_fusion_tabs", "fusion")])      # _tabs: t_relative=False, absolute temporal embeddings (B = 2)
def test_swin_tiny_other_modes_match_reference(stg, gpu, tag, mode):
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    m = _build_model(S, cfg, P, gpu)
    assert _apply_freeze(m) == names
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2).to(gpu)
    logits = m(a, v, mode)
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1).to(gpu)
    loss = torch.nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    _cmp(logits, z["logits"], f"{tag} logits", l2_rel=1.2e-2)                                              # NOISE
    err = float((logits.detach().cpu() - torch.as_tensor(z["logits"])).abs().max())
    _report.append(f"{tag} logits max abs err {err:.3e}")
    assert abs(float(loss) - float(z["loss"][0])) <= 1e-2
    _cmp(_flat_grads(m, names), z["grads"], f"{tag} grads", max_rel=1.5e-2, l2_rel=1.7e-2)   # 1.5 x measured (9.3e-3 / 1.13e-2)
    if "t_relative" in cfg:                                    # the embeddings' own gradients, not hidden behind the adapters'
        d = dict(m.named_parameters())
        off = 0
        for n in names:
            k = d[n].numel()
            if n.startswith("temporal_embedding"):
                _cmp(d[n].grad, z["grads"][off:off + k], f"{tag} grad[{n}]", max_rel=3e-2, l2_rel=3.6e-2)   # 1.5 x measured (2.0e-2 / 2.4e-2)
            off += k


@pytest.mark.parametrize("tag", ["swin_block_even", "swin_block_odd", "swin_block_s0", "swin_block_s3", "swin_block_nofusion",
                                 "swin_block_wide64", "swin_block_wide96"])
def test_fusion_block_matches_reference(stg, gpu, tag):
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="swin_block", T=cfg["T"], res=cfg["res"])
    blk = S.SwinTransformerBlock(dim=cfg["dim"], input_resolution=(cfg["res"], cfg["res"]), num_frames=cfg["T"],
                                 num_heads=cfg["heads"], window_size=7, shift_size=cfg["shift"], t_attn=cfg["t_attn"],
                                 adapter_mlp_ratio=cfg["ratio"], mode=cfg["mode"]).eval()
    _load_into(blk, P, "blk.")
    # structural: the block's own integer buffers equal the reference's
    assert torch.equal(blk.attn.relative_position_index, torch.as_tensor(z["rel_index"]))
    if blk.attn_mask is not None:
        assert torch.equal(blk.attn_mask, torch.as_tensor(z["attn_mask"]))
    blk = blk.to(gpu)
    mine = _apply_freeze(blk)
    assert mine == names
    BT, N, C = cfg["B"] * cfg["T"], cfg["res"] ** 2, cfg["dim"]
    v = seeded_tensor((BT, N, C), cfg["seed"] + 1); a = seeded_tensor((BT, N, C), cfg["seed"] + 2)
    gv = seeded_tensor((BT, N, C), cfg["seed"] + 3); ga = seeded_tensor((BT, N, C), cfg["seed"] + 4)
    X = torch.cat([v.reshape(-1, C), a.reshape(-1, C)]).to(BF16).to(gpu).requires_grad_(True)
    out = blk(X)
    _cmp(out[:BT * N], z["out_v"], f"{tag} out_v")
    _cmp(out[BT * N:], z["out_a"], f"{tag} out_a")
    dO = torch.cat([gv.reshape(-1, C), ga.reshape(-1, C)]).to(BF16).to(gpu)
    out.backward(dO)
    _cmp(X.grad[:BT * N], z["din_v"], f"{tag} din_v", max_rel=1.6e-2, l2_rel=1.2e-2)   # 1.5 x measured (1.06e-2 / 7.8e-3)
    _cmp(X.grad[BT * N:], z["din_a"], f"{tag} din_a", max_rel=1.6e-2, l2_rel=1.2e-2)
    _cmp(_flat_grads(blk, names), z["grads"], f"{tag} param grads", max_rel=1.3e-2, l2_rel=1.3e-2)   # 1.5 x measured (8.5e-3 / 8.6e-3)
    # per-tensor view of the same gradients (a small tensor must not hide behind a large one)
    d = dict(blk.named_parameters())
    off = 0
    for n in names:
        k = d[n].numel()
        ref = z["grads"][off:off + k]
        off += k
        if np.abs(ref).max() > 0:
            mr, lr = _block_grad_bounds(tag, k == 1, float(np.abs(ref).max()))
            _cmp(d[n].grad, ref, f"{tag} grad[{n}]", max_rel=mr, l2_rel=lr)


def _build_model(S, cfg, P, gpu, train=False):
    m = S.SwinTransformer2D_Adapter_New(label_dim=cfg["label_dim"], patch_size=[1, 4, 4], num_frames=cfg["num_frames"],
                                        embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                                        window_size=7, pretrained=None, ftmode=cfg["mode"],
                                        adapter_mlp_ratio=cfg["adapter_mlp_ratio"], t_relative=cfg.get("t_relative", True))
    _load_into(m, P)
    m = m.to(gpu)
    m.train(train)
    return m


def test_swin_tiny_fusion_model_matches_reference(stg, gpu):
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("swin_tiny_fusion")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    m = _build_model(S, cfg, P, gpu)
    assert _apply_freeze(m) == names
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2).to(gpu)
    logits = m(a, v, "fusion")
    assert logits.dtype == F32 and tuple(logits.shape) == (B * T, 29)
    err = float((logits.cpu() - torch.as_tensor(z["logits"])).abs().max())
    _report.append(f"swin_tiny logits max abs err {err:.3e} (|logits| max {float(np.abs(z['logits']).max()):.3g})")
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1).to(gpu)
    loss = torch.nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    _cmp(logits, z["logits"], "swin_tiny logits")
    assert abs(float(loss) - float(z["loss"][0])) <= 1e-2
    _cmp(_flat_grads(m, names), z["grads"], "swin_tiny grads", max_rel=1.5e-2, l2_rel=1.5e-2)     # 1.5 x measured (6.3e-3 / 8.5e-3)
    assert err <= 1e-2 * max(1.0, float(np.abs(z["logits"]).max())), f"logit deviation {err}"


def test_swin_b_fusion_full_model_matches_reference(stg, gpu):
    """Headline configuration, B=1: logits within 1e-2 of the reference's fp32 CPU path; adapter-gradient norms agree."""
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("swin_b_fusion")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    m = _build_model(S, cfg, P, gpu)
    del P
    assert _apply_freeze(m) == names
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert [sum(p.numel() for p in m.parameters()), n_train] == list(z["n_params"][:2])
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2).to(gpu)
    logits = m(a, v, "fusion")
    err = float((logits.cpu() - torch.as_tensor(z["logits"])).abs().max())
    _report.append(f"swin_b logits max abs err {err:.3e} (|logits| max {float(np.abs(z['logits']).max()):.3g})")
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1).to(gpu)
    loss = torch.nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    d = dict(m.named_parameters())
    norms = torch.stack([d[n].grad.float().norm().cpu() for n in names])
    # bounds = 1.5 x measured (norms 1.0e-3 / 2.2e-3, sample 2.0e-2 / 1.3e-2: profiles/r03_parity_report.txt)
    _cmp(norms, z["grad_norms"], "swin_b grad norms", max_rel=5e-3, l2_rel=5e-3)
    _cmp(_flat_grads(m, names)[::97], z["grads_sample"], "swin_b grad sample", max_rel=3e-2, l2_rel=2e-2)
    assert abs(float(loss) - float(z["loss"][0])) <= 1e-2
    # This fixture uses deliberately "hot" parameters (every Linear at gain ~1, |logits| up to 2.1) so that a wrong kernel
    # cannot hide; through 24 blocks that puts the bf16-operand floor (2^-9 per GEMM input) at ~1.3 % of the logit scale.
    # The north-star's absolute 1e-2 bound is asserted on the reference-initialised model below.
    assert err <= 1.5e-2 * float(np.abs(z["logits"]).max()), f"logit deviation {err}"


@pytest.mark.parametrize("case", ["swin_b_fusion_refinit", "swin_l_fusion_refinit"])
def test_swin_refinit_logits_within_1e2_abs(stg, gpu, case):
    """Swin-B and Swin-L + STG-CMA, full depth, at the reference's own initialisation scale (trunc_normal .02 Linears, unit
    LayerNorms, Swin_AVE.py:1353-1361) with the zero-initialised D_fc2 / gates de-zeroed: max-abs logit deviation <= 1e-2
    (BASELINE.json north_star) against the reference's fp32 CPU logits.  Swin-L (embed_dim 192, heads 6..48, adapters
    [.5, .25, .125, .0625]: AVE/run_adapt_ave29.py:167-181) is the backbone geometry of the AVQA model of BASELINE config 5."""
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor, refinit_state
    z, cfg, shapes, names = load_case(case)
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"], state_fn=refinit_state)
    m = _build_model(S, cfg, P, gpu)
    del P
    _apply_freeze(m)
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2).to(gpu)
    logits = m(a, v, "fusion")
    ref = torch.as_tensor(z["logits"])
    err = float((logits.detach().cpu() - ref).abs().max())
    _report.append(f"{case} logits max abs err {err:.3e} (|logits| max {float(ref.abs().max()):.3g})")
    assert err <= 1e-2, f"logit deviation {err}"
    tgt = torch.softmax(seeded_tensor((B * T, 29), cfg["seed"] + 3, 2.0), -1).to(gpu)
    loss = torch.nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    assert abs(float(loss) - float(z["loss"][0])) <= 1e-2
    d = dict(m.named_parameters())
    norms = torch.stack([d[n].grad.float().norm().cpu() for n in names])
    _cmp(norms, z["grad_norms"], f"{case} grad norms", max_rel=5e-2, l2_rel=3e-2)


def test_train_mode_droppath_and_dropout_run(stg, gpu):
    """DropPath (temporal residual) and head Dropout are active in train mode: outputs differ between calls, stay finite,
    and every trainable tensor receives a finite gradient."""
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor
    z, cfg, shapes, names = load_case("swin_tiny_fusion")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    m = _build_model(S, cfg, P, gpu, train=True)
    _apply_freeze(m)
    B, T = 2, cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), 1, 0.5).to(gpu); v = seeded_tensor((B, 3, T, 224, 224), 2).to(gpu)
    torch.manual_seed(0)
    l1 = m(a, v, "fusion")
    l2 = m(a, v, "fusion")
    assert torch.isfinite(l1).all() and torch.isfinite(l2).all()
    assert float((l1 - l2).abs().max()) > 0
    l1.sum().backward()
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize("tag", ["swin_block_even", "swin_block_video"])
def test_droppath_mask_semantics(stg, gpu, tag, monkeypatch):
    """DropPath of the reference blocks: on the temporal residual the mask is drawn on dim 0 of the '(b n) t c' layout -- ONE
    Bernoulli(keep) / keep value per (clip, token) row, constant over the T frames and the channels (Swin_AVE.py:705-716); on the
    parallel FFN adapter of 'video_adapt' one value per frame of the '(b t) n c' layout (:438-440).  The block is run in train mode
    with KNOWN masks (ops.drop_scale patched) and compared, forward and backward, with the oracle given the same masks."""
    import oracle.swin as OS
    from stgcma import ops
    from stgcma.model import Swin_AVE as S
    from params import seeded_tensor
    z, cfg, shapes, names = load_case(tag)
    P = build_state(shapes, cfg["seed"], kind="swin_block", T=cfg["T"], res=cfg["res"])
    p_drop = 0.3
    blk = S.SwinTransformerBlock(dim=cfg["dim"], input_resolution=(cfg["res"], cfg["res"]), num_frames=cfg["T"],
                                 num_heads=cfg["heads"], window_size=7, shift_size=cfg["shift"], t_attn=cfg["t_attn"],
                                 adapter_mlp_ratio=cfg["ratio"], mode=cfg["mode"], drop_path=p_drop).train()
    _load_into(blk, P, "blk.")
    blk = blk.to(gpu)
    _apply_freeze(blk)
    B, T, N, C = cfg["B"], cfg["T"], cfg["res"] ** 2, cfg["dim"]
    BT = B * T
    two = cfg["mode"] == "fusion_adapt"
    drawn = []

    def fake_drop_scale(p, n_rows, device, training, pool=None):
        assert training and abs(p - p_drop) < 1e-12
        g = torch.Generator().manual_seed(900 + len(drawn))
        m = (torch.rand(n_rows, generator=g) < (1 - p)).float() / (1 - p)
        drawn.append(m)
        return m.to(device)
    monkeypatch.setattr(ops, "drop_scale", fake_drop_scale)
    xs = [seeded_tensor((BT, N, C), cfg["seed"] + 1 + i) for i in range(2 if two else 1)]
    gs = [seeded_tensor((BT, N, C), cfg["seed"] + 3 + i) for i in range(2 if two else 1)]
    X = torch.cat([x.reshape(-1, C) for x in xs]).to(gpu).requires_grad_(True)
    out = blk(X)
    out.backward(torch.cat([g.reshape(-1, C) for g in gs]).to(gpu))
    # the masks the block asked for: one per modality with B * N entries (temporal residual), then -- parallel mode -- B * T entries
    want = [B * N] * (2 if two else 1) + ([BT] if not two else [])
    assert [m.numel() for m in drawn] == want, [m.numel() for m in drawn]
    assert all(bool(((m == 0) | ((m - 1.0 / (1 - p_drop)).abs() < 1e-6)).all()) for m in drawn)
    assert all(0 < float((m == 0).float().mean()) < 1 for m in drawn), "the seeded masks must drop some rows and keep others"
    dps = {"t_v": drawn[0], "t_a": drawn[1]} if two else {"t_v": drawn[0], "ffn": drawn[1]}
    for n in P:
        if P[n].is_floating_point():
            P[n] = P[n].clone().requires_grad_(True)
    xr = [x.clone().requires_grad_(True) for x in xs]
    ref = OS.swin_block(P, "blk", tuple(xr) if two else xr[0], H=cfg["res"], W=cfg["res"], T=T, heads=cfg["heads"],
                        shift_size=cfg["shift"], t_attn=cfg["t_attn"], mode=cfg["mode"], dp_scale=dps)
    ref = ref if two else (ref,)
    sum((r * g).sum() for r, g in zip(ref, gs)).backward()
    R = BT * N
    for i, r in enumerate(ref):
        _cmp(out[i * R:(i + 1) * R], r.detach().reshape(-1, C), f"{tag} droppath out[{i}]")
        _cmp(X.grad[i * R:(i + 1) * R], xr[i].grad.reshape(-1, C), f"{tag} droppath din[{i}]", max_rel=3e-2, l2_rel=2e-2)
    d = dict(blk.named_parameters())
    for n in ("T_Adapter.D_fc2.weight", "T_Adapter.D_fc1.weight", "S_Adapter.D_fc2.weight"):
        _cmp(d[n].grad, P["blk." + n].grad, f"{tag} droppath grad[{n}]", max_rel=6e-2, l2_rel=4e-2)
    # a dropped (clip, token) row passes the temporal branch unchanged for EVERY frame: with everything dropped the block equals
    # its own eval-mode spatial + FFN part applied to the input -- checked through the oracle above; here the direct invariant:
    kept = drawn[0].view(B, N) != 0
    assert kept.any() and (~kept).any()


def test_product_path_has_no_cpu_fallback(stg):
    from stgcma.model import Swin_AVE as S
    m = S.SwinTransformer2D_Adapter_New(label_dim=29, embed_dim=32, depths=[2, 2], num_heads=[1, 2], num_frames=2,
                                        ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25])
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 224, 224), torch.zeros(1, 3, 2, 224, 224), "fusion")


def test_use_checkpoint_flag_is_accepted(stg, gpu):
    """use_checkpoint=True (Swin_AVE.py:1049-1050) trades memory for recompute in the reference and leaves results unchanged; here it
    is accepted and a no-op: same logits as without it."""
    from stgcma.model import Swin_AVE
    kw = dict(pretrained=None, num_frames=2, embed_dim=32, depths=[2, 2], num_heads=[1, 2], ftmode="fusion", label_dim=5,
              adapter_mlp_ratio=[0.5, 0.5])
    torch.manual_seed(0)
    m0 = Swin_AVE.SwinTransformer2D_Adapter_New(**kw).to(gpu).eval()
    m1 = Swin_AVE.SwinTransformer2D_Adapter_New(use_checkpoint=True, **kw).to(gpu).eval()
    m1.load_state_dict(m0.state_dict())
    a = torch.randn(2, 2, 224, 224, device=gpu) * 0.5
    v = torch.randn(2, 3, 2, 224, 224, device=gpu)
    with torch.no_grad():
        assert torch.equal(m0(a, v, "fusion"), m1(a, v, "fusion"))
