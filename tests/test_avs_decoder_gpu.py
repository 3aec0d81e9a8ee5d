"""-m gpu: the AVS dense decoder on the HIP path.
  * building blocks (ops_dec: ASPP, FeatureFusionBlock, TPAVI train / eval, output_conv) against the goldens produced by the
    REFERENCE modules at small sizes (tests/golden/avs_decoder_modules.npz);
  * the full model -- backbone + decoder -- against the golden of the reference's SwinTransformer2D_Adapter_AVS_Base
    (avs_full_tiny.npz); a train-mode loop with the AVS loss."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLD, build_state, load_case

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _rel(got, ref):
    got = got.detach().float().cpu().reshape(-1); ref = torch.as_tensor(np.asarray(ref)).float().reshape(-1)
    assert got.shape == ref.shape, (tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6), float((got - ref).norm() / max(float(ref.norm()), 1e-12))


def _chk(got, ref, what, max_rel=3e-2, l2_rel=2e-2):
    e_max, e_l2 = _rel(got, ref)
    assert e_max <= max_rel and e_l2 <= l2_rel, f"{what}: max/scale={e_max:.3e} relL2={e_l2:.3e}"


def _rows(x_nchw, gpu, grad=True):
    """NCHW fp32 (numpy) -> channels-last bf16 rows [F*H*W, C] on the GPU."""
    t = torch.as_tensor(np.asarray(x_nchw)).permute(0, 2, 3, 1).contiguous()
    return t.reshape(-1, t.shape[-1]).to(BF16).to(gpu).requires_grad_(grad)


def _nchw(rows, F_, H, W):
    return rows.detach().float().cpu().view(F_, H, W, -1).permute(0, 3, 1, 2)


def _load(mod, z, key, seed_off, gpu):
    from params import seeded_state
    shapes = [(k, tuple(s)) for k, s in json.loads(str(z[key]))]
    sd = mod.state_dict()
    sd.update(seeded_state(shapes, int(z["seed"][0]) + seed_off))
    mod.load_state_dict(sd, strict=True)
    return mod.to(gpu), [k for k, _ in shapes if "running" not in k]


def _pgrads(mod, keys):
    d = dict(mod.named_parameters())
    return torch.cat([(d[k].grad if d[k].grad is not None else torch.zeros_like(d[k])).reshape(-1).float().cpu() for k in keys])


def test_aspp_ffb_output_conv_match_reference_modules(stg, gpu):
    from stgcma import ops_dec as D
    from stgcma.model import Swin_AVSModel as M
    z = np.load(os.path.join(GOLD, "avs_decoder_modules.npz"))
    # ASPP: four dilated 3x3 convolutions (dilations up to 18 on a 14 x 14 map: mostly padding)
    aspp, keys = _load(M.Classifier_Module([3, 6, 12, 18], [3, 6, 12, 18], 24, 16), z, "aspp_shapes", 1, gpu)
    x = _rows(z["aspp_x"], gpu)
    y = D.aspp(x, aspp, (3, 14, 14))
    y.backward(_rows(z["aspp_gy"], gpu, False))
    _chk(_nchw(y, 3, 14, 14), z["aspp_y"], "aspp y"); _chk(_nchw(x.grad, 3, 14, 14), z["aspp_dx"], "aspp dx")
    _chk(_pgrads(aspp, keys), z["aspp_grads"], "aspp grads")
    # FeatureFusionBlock, two inputs and one
    ffb, keys = _load(M.FeatureFusionBlock(16), z, "ffb_shapes", 2, gpu)
    x0, x1 = _rows(z["ffb_x0"], gpu), _rows(z["ffb_x1"], gpu)
    y, seen = D.feature_fusion(ffb, x0, x1, (2, 7, 7))
    y.backward(_rows(z["ffb_gy"], gpu, False))
    # gradients that pass ReLU gates: with ~0.5 % bf16 noise on the pre-activations a few of the 1568 gates of these tiny maps sit
    # on the other side of zero than in the fp32 reference, each a full-size element error (see test_avqa_head_gpu.py)
    GATED = dict(max_rel=1.5e-1, l2_rel=6e-2)
    _chk(_nchw(y, 2, 14, 14), z["ffb_y"], "ffb y"); _chk(_nchw(x0.grad, 2, 7, 7), z["ffb_dx0"], "ffb dx0", **GATED)
    _chk(_nchw(x1.grad, 2, 7, 7), z["ffb_dx1"], "ffb dx1", **GATED)
    _chk(_pgrads(ffb, keys), z["ffb_grads"], "ffb grads", **GATED)
    assert torch.equal(seen, torch.relu(x1.detach()))
    ffb.zero_grad()
    b0 = _rows(z["ffb_x0"], gpu)
    y1, _ = D.feature_fusion(ffb, b0, None, (2, 7, 7))
    y1.backward(_rows(z["ffb_gy"], gpu, False))
    _chk(_nchw(y1, 2, 14, 14), z["ffb1_y"], "ffb1 y"); _chk(_nchw(b0.grad, 2, 7, 7), z["ffb1_dx0"], "ffb1 dx0", **GATED)
    # output_conv stack: conv3x3 -> bilinear x2 (align_corners=False) -> conv3x3 -> ReLU -> conv1x1
    oc = torch.nn.Sequential(torch.nn.Conv2d(16, 24, 3, 1, 1), M.Interpolate(2, "bilinear"), torch.nn.Conv2d(24, 8, 3, 1, 1),
                             torch.nn.ReLU(True), torch.nn.Conv2d(8, 1, 1, 1, 0))
    oc, keys = _load(oc, z, "oc_shapes", 4, gpu)
    x = _rows(z["oc_x"], gpu)
    y = D.output_conv(oc, x, (2, 6, 6))
    assert y.dtype == F32
    y.backward(torch.as_tensor(z["oc_gy"]).permute(0, 2, 3, 1).reshape(-1, 1).to(gpu))
    _chk(_nchw(y, 2, 12, 12), z["oc_y"], "oc y"); _chk(_nchw(x.grad, 2, 6, 6), z["oc_dx"], "oc dx", **GATED)
    _chk(_pgrads(oc, keys), z["oc_grads"], "oc grads", **GATED)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_tpavi_matches_reference_module(stg, gpu, mode):
    from stgcma import ops_dec as D
    from stgcma.model import Swin_AVSModel as M
    z = np.load(os.path.join(GOLD, "avs_decoder_modules.npz"))
    tp, keys = _load(M.TPAVIModule(in_channels=32, mode='dot'), z, "tpavi_shapes", 3, gpu)
    with torch.no_grad():
        tp.W_z[1].running_mean.copy_(torch.as_tensor(z[f"tpavi_{mode}_rm0"])); tp.W_z[1].running_var.copy_(torch.as_tensor(z[f"tpavi_{mode}_rv0"]))
    B, C, T, H, W = z["tpavi_x"].shape
    x5 = torch.as_tensor(z["tpavi_x"])                                   # [B, C, T, H, W] -> rows (b t h w)
    x = x5.permute(0, 2, 3, 4, 1).reshape(-1, C).to(BF16).to(gpu).requires_grad_(True)
    au = torch.as_tensor(z["tpavi_audio"]).reshape(B * T, 128).to(BF16).to(gpu).requires_grad_(True)
    zz, at = D.tpavi(tp, x, au, B, T, H * W, mode == "train")
    gz = torch.as_tensor(z[f"tpavi_{mode}_gz"]).permute(0, 2, 3, 4, 1).reshape(-1, C)
    ga = torch.as_tensor(z[f"tpavi_{mode}_ga"]).reshape(B * T, C)
    ((zz.float() * gz.to(gpu)).sum() + (at.float() * ga.to(gpu)).sum()).backward()
    _chk(zz.detach().float().cpu().view(B, T, H, W, C).permute(0, 4, 1, 2, 3), z[f"tpavi_{mode}_z"], "z")
    _chk(at.detach().float().cpu().view(B, T, C), z[f"tpavi_{mode}_a"], "audio_temp")
    _chk(x.grad.float().cpu().view(B, T, H, W, C).permute(0, 4, 1, 2, 3), z[f"tpavi_{mode}_dx"], "dx")
    _chk(au.grad.float().cpu().view(B, T, 128), z[f"tpavi_{mode}_da"], "daudio", max_rel=5e-2, l2_rel=4e-2)
    _chk(_pgrads(tp, keys), z[f"tpavi_{mode}_grads"], "param grads", max_rel=5e-2, l2_rel=4e-2)
    _chk(tp.W_z[1].running_mean, z[f"tpavi_{mode}_rm1"], "running_mean"); _chk(tp.W_z[1].running_var, z[f"tpavi_{mode}_rv1"], "running_var")


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_tpavi_visual_self_attention_matches_reference_module(stg, gpu, mode):
    """tpavi_vv_flag's form: TPAVIModule without audio (TPAVI.py:96-98), golden `avs_tpavi_vv` from the reference module."""
    from stgcma import ops_dec as D
    from stgcma.model import Swin_AVSModel as M
    z = np.load(os.path.join(GOLD, "avs_tpavi_vv.npz"))
    tp, keys = _load(M.TPAVIModule(in_channels=32, mode='dot'), z, "tpavi_shapes", 3, gpu)
    with torch.no_grad():
        tp.W_z[1].running_mean.copy_(torch.as_tensor(z[f"tpavi_{mode}_rm0"])); tp.W_z[1].running_var.copy_(torch.as_tensor(z[f"tpavi_{mode}_rv0"]))
    B, C, T, H, W = z["tpavi_x"].shape
    x = torch.as_tensor(z["tpavi_x"]).permute(0, 2, 3, 4, 1).reshape(-1, C).to(BF16).to(gpu).requires_grad_(True)
    zz, at = D.tpavi(tp, x, None, B, T, H * W, mode == "train")
    assert at is None
    gz = torch.as_tensor(z[f"tpavi_{mode}_gz"]).permute(0, 2, 3, 4, 1).reshape(-1, C)
    (zz.float() * gz.to(gpu)).sum().backward()
    _chk(zz.detach().float().cpu().view(B, T, H, W, C).permute(0, 4, 1, 2, 3), z[f"tpavi_{mode}_z"], "z")
    _chk(x.grad.float().cpu().view(B, T, H, W, C).permute(0, 4, 1, 2, 3), z[f"tpavi_{mode}_dx"], "dx")
    g = torch.cat([(dict(tp.named_parameters())[k].grad if dict(tp.named_parameters())[k].grad is not None
                    else torch.zeros_like(dict(tp.named_parameters())[k])).reshape(-1).float().cpu() for k in keys])
    _chk(g, z[f"tpavi_{mode}_grads"], "param grads", max_rel=5e-2, l2_rel=4e-2)
    _chk(tp.W_z[1].running_mean, z[f"tpavi_{mode}_rm1"], "running_mean"); _chk(tp.W_z[1].running_var, z[f"tpavi_{mode}_rv1"], "running_var")


def _build_full(gpu, vv=False, want_state=False, case="avs_full_tiny", state_fn=None):
    from stgcma import recipe
    from stgcma.model import Swin_AVSModel
    z, cfg, shapes, names = load_case(case)
    m = Swin_AVSModel.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                                    num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"],
                                                    drop_path_rate=0.0, tpavi_vv_flag=vv).train()      # the golden: train-mode BatchNorm, no DropPath
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"], state_fn=state_fn)
    gg = torch.Generator().manual_seed(cfg["seed"] + 50)
    for k, sh in shapes:                                                  # the running statistics make_golden.py drew after seeding
        if k.endswith("running_var"):
            P[k] = torch.rand(sh, generator=gg) + 0.5
        elif k.endswith("running_mean"):
            P[k] = torch.randn(sh, generator=gg) * 0.1
        elif "W_z.1.weight" in k:
            P[k] = P[k] * 0.1                                             # TPAVI's BatchNorm scale kept small (reference init: 0)
    sd = m.state_dict()
    assert [k for k in sd if sd[k].is_floating_point() and not k.endswith("attn_mask")] == [k for k, _ in shapes]
    for k in sd:
        if sd[k].is_floating_point() and not k.endswith("attn_mask"):
            sd[k] = P[k]
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu)
    mine = []
    for n, p in m.named_parameters():
        p.requires_grad = recipe.is_trainable(n)
        if p.requires_grad:
            mine.append(n)
    assert mine == names
    if want_state:
        return m, z, cfg, names, P
    return m, z, cfg, names


def test_avs_full_model_with_visual_self_attention_matches_oracle(stg, gpu):
    """tpavi_vv_flag=True AND tpavi_va_flag=True (Swin_AVSModel_Base.py:1873-1886: every TPAVI block runs its visual self-attention form
    and its audio-visual form, the results are averaged).  No runner uses it, so there is no whole-model golden: the HIP model is
    compared with the oracle, whose vv form is pinned on the reference module (`avs_tpavi_vv`) and whose decoder is pinned on the
    reference model (`avs_full_tiny`).  Train-mode BatchNorm without DropPath, like that fixture."""
    import oracle.avs_decoder as OD
    from params import seeded_tensor
    m, z, cfg, names, P = _build_full(gpu, vv=True, want_state=True)
    B, seed = cfg["B"], cfg["seed"]
    a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5)
    v = seeded_tensor((B, 5, 3, 224, 224), seed + 2)
    with torch.no_grad():
        ref_pred, ref_maps, ref_af = OD.avs_forward(P, a, v, dict(cfg, num_frames=5), bn_training=True, tpavi_vv=True, tpavi_va=True)
        base_pred, _, _ = OD.avs_forward(P, a, v, dict(cfg, num_frames=5), bn_training=True)
    assert float((ref_pred - base_pred).abs().max()) > 1e-3 * float(base_pred.abs().max()), "the vv form changed nothing: fixture does not test it"
    pred, fmaps, afeas = m(a.to(gpu), v.to(gpu), "fusion")
    errs = {"pred": _rel(pred, ref_pred.numpy())}
    for i in range(4):
        errs[f"fmap{i}"] = _rel(fmaps[i], ref_maps[i].numpy())
        errs[f"afea{i}"] = _rel(afeas[i], ref_af[i].numpy())
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write("avs_full_tiny vv+va (vs oracle) " + " ".join(f"{k}: max/scale={a_:.3e} relL2={b_:.3e}" for k, (a_, b_) in errs.items()) + "\n")
    for k, (e_max, e_l2) in errs.items():
        # relative L2 as for the va-only fixture below; the maximum is taken over the WHOLE maps here (16 M elements at stage 0, not that
        # fixture's every-8th-channel sample), so its bound is wider: measured pred 2.2e-2 / 1.8e-2, maps <= 3.3e-2 / 0.9e-2
        lim = (3.5e-2, 3e-2) if k == "pred" else (5e-2, 1.5e-2)
        assert e_max <= lim[0] and e_l2 <= lim[1], f"{k}: max/scale={e_max:.3e} relL2={e_l2:.3e}"
    pred.float().square().mean().backward()
    d = dict(m.named_parameters())
    for n in names:
        assert d[n].grad is None or torch.isfinite(d[n].grad).all(), n
    assert d["avstask_tpavi_b1.phi.weight"].grad is not None and float(d["avstask_tpavi_b1.phi.weight"].grad.abs().max()) > 0


def test_avs_full_depth_model_forward_matches_reference(stg, gpu):
    """BASELINE config 4's model at FULL depth (Swin-B, depths [2, 2, 18, 2], T = 5): outputs against the fixture the reference's
    SwinTransformer2D_Adapter_AVS_Base produced (avs_full_b18, B = 1).  24 blocks of seeded unit-gain parameters in front of the decoder:
    bounds relative to each output's scale, first measured in round 4."""
    from params import seeded_tensor
    m, z, cfg, names = _build_full(gpu, case="avs_full_b18")
    B, seed = cfg["B"], cfg["seed"]
    a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 5, 3, 224, 224), seed + 2).to(gpu)
    with torch.no_grad():
        pred, fmaps, afeas = m(a, v, "fusion")
    errs = {"pred": _rel(pred, z["pred"])}
    for i in range(4):
        errs[f"fmap{i}"] = _rel(fmaps[i][:, ::8], z[f"fmap{i}"])
        errs[f"afea{i}"] = _rel(afeas[i], z[f"afea{i}"])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write("avs_full_b18 " + " ".join(f"{k}: max/scale={a_:.3e} relL2={b_:.3e}" for k, (a_, b_) in errs.items()) + "\n")
    # 1.5 x measured (round 4): pred 1.7e-2 / 1.7e-2, maps and audio features <= 1.5e-2 / 1.1e-2; the stage-3 map has one element at 6.8e-2 of
    # its scale (49 positions behind 24 blocks) with the same relative L2 as the others
    for k, (e_max, e_l2) in errs.items():
        lim = (2.6e-2, 2.6e-2) if k == "pred" else ((1.05e-1, 1.65e-2) if k == "fmap3" else (2.3e-2, 1.65e-2))
        assert e_max <= lim[0] and e_l2 <= lim[1], f"{k}: max/scale={e_max:.3e} relL2={e_l2:.3e}"


def test_avs_full_depth_refinit_matches_reference(stg, gpu):
    """BASELINE config 4's model at full depth AND at the reference's initialisation scale (fixture avs_full_b18_refinit, round 5):
    pred is O(0.4), so north_star's ABSOLUTE bound applies -- max |pred - reference| <= 1e-2 -- next to the relative ones; the gradients are checked
    in aggregate as for the other AVS fixtures (one-clip batch-statistics BatchNorm: DESIGN.md section 3)."""
    from params import seeded_tensor, refinit_state
    m, z, cfg, names = _build_full(gpu, case="avs_full_b18_refinit", state_fn=refinit_state)
    B, seed = cfg["B"], cfg["seed"]
    a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 5, 3, 224, 224), seed + 2).to(gpu)
    pred, fmaps, afeas = m(a, v, "fusion")
    e_abs = float((pred.detach().float().cpu() - torch.as_tensor(np.asarray(z["pred"]))).abs().max())
    errs = {"pred": _rel(pred, z["pred"])}
    for i in range(4):
        errs[f"fmap{i}"] = _rel(fmaps[i][:, ::8], z[f"fmap{i}"])
        errs[f"afea{i}"] = _rel(afeas[i], z[f"afea{i}"])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write(f"avs_full_b18_refinit pred max-abs={e_abs:.3e} (scale {float(np.abs(np.asarray(z['pred'])).max()):.3g}) " +
                " ".join(f"{k}: max/scale={a_:.3e} relL2={b_:.3e}" for k, (a_, b_) in errs.items()) + "\n")
    assert e_abs <= 1e-2, f"pred: max-abs deviation {e_abs:.3e} from the reference (BASELINE: <= 1e-2)"
    for k, (e_max, e_l2) in errs.items():
        assert e_max <= 4e-2 and e_l2 <= 2.5e-2, f"{k}: max/scale={e_max:.3e} relL2={e_l2:.3e}"
    loss = (pred * seeded_tensor(pred.shape, seed + 3, 1e-2).to(gpu)).sum()
    for i, (fm, af) in enumerate(zip(fmaps, afeas)):
        loss = loss + (fm * seeded_tensor(fm.shape, seed + 10 + i, 1e-2).to(gpu)).sum() + (af * seeded_tensor(af.shape, seed + 20 + i, 1e-1).to(gpu)).sum()
    loss.backward()
    d = dict(m.named_parameters())
    ref_norms = np.asarray(z["grad_norms"])
    ratios = []
    for n, rn in zip(names, ref_norms):
        if d[n].grad is None:
            assert rn == 0, n
            continue
        assert torch.isfinite(d[n].grad).all(), n
        if rn > 1e-3 and "gate_" not in n and "temporal_position_bias_table" not in n:
            ratios.append((float(d[n].grad.norm()) / float(rn), n))
    rr = np.array([r for r, _ in ratios])
    med, share = float(np.median(rr)), float(((rr > 0.75) & (rr < 1.25)).mean())
    flat = torch.cat([(d[n].grad if d[n].grad is not None else torch.zeros_like(d[n])).reshape(-1).float().cpu() for n in names])[::97]
    ref = torch.as_tensor(z["grads_sample"])
    cos = float(torch.dot(flat, ref) / (flat.norm() * ref.norm()))
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write(f"avs_full_b18_refinit grads: cos={cos:.4f} norm ratio median={med:.3f} share within 25%={share:.3f} min={min(ratios)} max={max(ratios)}\n")
    # at the reference's initialisation scale the whole-model gradient IS well conditioned (measured, round 5: cosine 0.9984, median norm ratio 0.999,
    # 99.8 % of the tensors within 25 %; the outliers are biases in front of a batch-statistics BatchNorm, whose true gradient is zero)
    assert cos >= 0.99 and 0.97 <= med <= 1.03 and share >= 0.95, f"gradient sample cosine {cos:.4f}, median norm ratio {med:.3f}, share within 25 % {share:.3f}"


def test_avs_full_model_matches_reference(stg, gpu):
    from params import seeded_tensor
    m, z, cfg, names = _build_full(gpu)
    B, seed = cfg["B"], cfg["seed"]
    a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5).to(gpu)
    v = seeded_tensor((B, 5, 3, 224, 224), seed + 2).to(gpu)
    pred, fmaps, afeas = m(a, v, "fusion")
    assert pred.dtype == F32 and tuple(pred.shape) == (B * 5, 1, 224, 224)
    errs = {"pred": _rel(pred, z["pred"])}
    for i in range(4):
        errs[f"fmap{i}"] = _rel(fmaps[i][:, ::8], z[f"fmap{i}"])
        errs[f"afea{i}"] = _rel(afeas[i], z[f"afea{i}"])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write("avs_full_tiny " + " ".join(f"{k}: max/scale={a:.3e} relL2={b:.3e}" for k, (a, b) in errs.items()) + "\n")
    # pred sits behind 8 backbone blocks + ~25 bf16 convolutions of a seeded ("hot", unit-gain) decoder with ReLU gates in between;
    # the feature maps (after ASPP + TPAVI) and the audio features are earlier in the chain
    for k, (e_max, e_l2) in errs.items():
        lim = (3.5e-2, 3e-2) if k == "pred" else (2.5e-2, 1.5e-2)      # ~1.5 x measured (pred 2.2 % / 1.9 %, maps <= 1.5 % / 0.9 %)
        assert e_max <= lim[0] and e_l2 <= lim[1], f"{k}: max/scale={e_max:.3e} relL2={e_l2:.3e}"
    loss = (pred * seeded_tensor(pred.shape, seed + 3, 1e-2).to(gpu)).sum()
    for i, (fm, af) in enumerate(zip(fmaps, afeas)):
        loss = loss + (fm * seeded_tensor(fm.shape, seed + 10 + i, 1e-2).to(gpu)).sum() + (af * seeded_tensor(af.shape, seed + 20 + i, 1e-1).to(gpu)).sum()
    loss.backward()
    d = dict(m.named_parameters())
    ref_norms = np.asarray(z["grad_norms"])
    # Gradients.  Forward parity above is tight, and the backward of every building block is pinned at module level (tests above,
    # 1-6 %).  The WHOLE-model gradient of this configuration, though, is not a well-conditioned quantity: with batch-statistics
    # BatchNorm over a single clip, the gradients of TPAVI's inner branch (W_z.0, g, theta, phi) and of everything upstream of it
    # swing several-fold in the fp32 oracle itself when the taps change by 0.7 % (tools/dbg_avs.py: g.weight of stage 3: 16 on the
    # oracle's taps, 144 on the HIP taps; reference vs oracle on IDENTICAL fp32 inputs already differ by 2e-3 from summation order),
    # and the decoder's ReLU gates add sqrt(flipped fraction) on top (test_avqa_head_gpu.py).  So direction and size are checked in
    # aggregate: cosine of a strided sample, median norm ratio, share of tensors within 25 %.
    ratios = []
    for n, rn in zip(names, ref_norms):
        if d[n].grad is None:                            # path4.resConfUnit1: unused with a single input (:1887), no gradient in the reference either
            assert rn == 0, n
            continue
        assert torch.isfinite(d[n].grad).all(), n
        if rn > 1e-3 and "gate_" not in n and "temporal_position_bias_table" not in n:
            ratios.append((float(d[n].grad.norm()) / float(rn), n))
    lo, hi = min(ratios), max(ratios)
    rr = np.array([r for r, _ in ratios])
    med, share = float(np.median(rr)), float(((rr > 0.75) & (rr < 1.25)).mean())
    flat = torch.cat([(d[n].grad if d[n].grad is not None else torch.zeros_like(d[n])).reshape(-1).float().cpu() for n in names])[::97]
    ref = torch.as_tensor(z["grads_sample"])
    cos = float(torch.dot(flat, ref) / (flat.norm() * ref.norm()))
    e_l2 = float((flat - ref).norm() / ref.norm())
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write(f"avs_full_tiny grads: cos={cos:.4f} relL2={e_l2:.3e} norm ratio median={med:.3f} share within 25%={share:.3f} min={lo} max={hi}\n")
    # train-mode BatchNorm over ONE clip makes the TPAVI branch a chaotic amplifier (the fp32 oracle's own gradient
    # moves several-fold under a 0.7 % change of the taps): the sample cosine read 0.90 with the LayerNorms on fp32 rows and 0.76 with
    # them on bf16 x_hat rows -- same median norm ratio (1.00), every backbone / decoder building block pinned tightly elsewhere (the
    # eval-BatchNorm fixture below splits at the taps).  This aggregate only guards against a broken sign / scale.
    assert cos >= 0.65 and 0.9 <= med <= 1.1 and share >= 0.6, f"gradient sample cosine {cos:.4f}, median norm ratio {med:.3f}, share within 25 % {share:.3f}"


def _relcos(a, b):
    a = a.detach().float().cpu().reshape(-1); b = b.detach().float().cpu().reshape(-1)
    return float((a - b).norm() / b.norm()), float(a.norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))


def test_avs_full_model_eval_batchnorm_gradients(stg, gpu):
    """Whole-model gradient parity of the AVS model on the eval-BatchNorm fixture (ADVICE r1 / VERDICT r1 item 8), SPLIT at the
    backbone / decoder boundary (the four video taps + the audio feature) with the fp32 oracle run beside the HIP path -- the oracle
    reproduces the reference's gradients of this fixture to 2e-3 (tests/test_oracle_cpu.py), so it stands in for the reference on
    either side of the cut:
      1. the taps themselves (forward)                                        tight: <= 1.5e-2 relative L2
      2. the HIP BACKBONE driven by the oracle's d(taps)                      tight: median per-tensor relL2 <= 3e-2 (measured 1.0e-2)
      3. the HIP DECODER on the oracle's taps: pred tight (<= 2e-2), d(taps) and its parameter gradients in AGGREGATE only
      4. end to end against the golden                                        aggregate only
    Why 3 / 4 are aggregate bounds (round-2 finding, tools/avs_grad_conditioning.py reproduces it on the CPU in fp32): the decoder is a
    ReLU network (ResidualConvUnit, output_conv), and a ReLU network's gradient is DISCONTINUOUS in its activations -- a forward
    perturbation of relative size eps flips ~eps of the masks, and the gradient moves by ~sqrt(eps).  Rounding ONE decoder tensor of
    the fp32 oracle to bf16 (eps = 2^-9) moves the oracle's own d(taps) by 4-8 % and single parameter gradients by up to 35 %; all
    decoder activations rounded: 8-11 % on d(taps), 92 % on the worst tensor (TPAVI block 1's W_z, a sum over 15 680 positions per clip
    that mostly cancels).  The backbone (GELU, LayerNorm: smooth) does not have this: 1 % (check 2).  The reference trains this decoder
    under fp16 autocast (AVS/traintest_adapt_avs.py), whose gradients scatter the same way at half the size."""
    from stgcma import recipe
    from stgcma.model import Swin_AVSModel_Base
    from stgcma.ops_dec import avs_decoder_forward
    from params import seeded_tensor
    import oracle.avs_decoder as OD
    from oracle.swin import swin_backbone
    z, cfg, shapes, names = load_case("avs_full_tiny_evalbn")
    m = Swin_AVSModel_Base.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                                              num_heads=cfg["num_heads"], ftmode="fusion",
                                                              adapter_mlp_ratio=cfg["adapter_mlp_ratio"]).eval()
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for k, _ in shapes:
        if "W_z.1.weight" in k:
            P[k] = P[k] * 0.1
    for i, k in enumerate(json.loads(str(z["stat_names_json"]))):
        P[k] = torch.as_tensor(np.asarray(z[f"stat{i}"]))
    sd = m.state_dict()
    for k in sd:
        if sd[k].is_floating_point() and not k.endswith("attn_mask"):
            sd[k] = P[k]
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu)
    recipe.apply_freeze(m)
    assert [n for n, p in m.named_parameters() if p.requires_grad] == names
    B, seed = cfg["B"], cfg["seed"]
    a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5)
    v = seeded_tensor((B, 5, 3, 224, 224), seed + 2)
    w = seeded_tensor(tuple(np.asarray(z["pred"]).shape), seed + 3, 1e-2)
    d = dict(m.named_parameters())
    rep = []

    def zero():
        for p_ in m.parameters():
            p_.grad = None

    # ---- the oracle, cut at the taps
    Po = {k: t.clone() for k, t in P.items()}
    for n in names:
        Po[n].requires_grad_(True)
    out = swin_backbone(Po, a, v, cfg)
    taps_o = [t.detach().clone().requires_grad_(True) for t in out["taps"]] + [out["f_a"].detach().clone().requires_grad_(True)]
    pred_o, _, _ = OD.avs_decoder(Po, taps_o[:4], taps_o[4], B, 5, bn_training=False)
    assert _rel(pred_o, z["pred"])[1] <= 1e-4                         # the oracle IS the reference here
    (pred_o * w).sum().backward()
    dtaps_o = [t.grad.clone() for t in taps_o]
    dec_o = {n: Po[n].grad.clone() for n in names if Po[n].grad is not None and float(Po[n].grad.norm()) > 0}
    for n in names:
        Po[n].grad = None
    torch.autograd.backward(out["taps"] + [out["f_a"]], dtaps_o)
    bb_o = {n: Po[n].grad.clone() for n in names if Po[n].grad is not None and not n.startswith("avstask_") and float(Po[n].grad.norm()) > 1e-12}

    # ---- 1. taps, 2. the backbone alone
    ms, a_feat = m.forward_features(a.to(gpu), v.to(gpu))
    hip_taps = list(ms) + [a_feat]
    e_taps = [_relcos(t, to.reshape(t.shape))[0] for t, to in zip(hip_taps, taps_o)]
    rep.append("taps relL2 " + " ".join(f"{e:.2e}" for e in e_taps))
    assert max(e_taps) <= 1.5e-2, rep[-1]
    zero()
    torch.autograd.backward(hip_taps, [g.reshape(t.shape).to(gpu) for g, t in zip(dtaps_o, hip_taps)])
    rows = sorted(((_relcos(d[n].grad, g)[0], n) for n, g in bb_o.items()), reverse=True)
    assert all(d[n].grad is not None and torch.isfinite(d[n].grad).all() for n in bb_o)
    med_bb = float(np.median([r for r, _ in rows]))
    plain = [(r, n) for r, n in rows if "gate_" not in n and "temporal_position_bias_table" not in n]
    rep.append(f"backbone alone (oracle d(taps) in): median per-tensor relL2 {med_bb:.3e} over {len(rows)} tensors, worst {rows[0][0]:.3e} ({rows[0][1]}), "
               f"worst outside gates / bias tables {plain[0][0]:.3e} ({plain[0][1]})")
    assert med_bb <= 3e-2 and plain[0][0] <= 8e-2 and rows[0][0] <= 0.3, rep[-1]

    # ---- 3. the decoder alone
    zero()
    in2 = [to.detach().reshape(t.shape).to(gpu).requires_grad_(True) for t, to in zip(hip_taps, taps_o)]
    pred2, _, _ = avs_decoder_forward(m, in2[:4], in2[4], B, 5, False)
    e_pred2 = _relcos(pred2, pred_o)[0]
    (pred2 * w.to(gpu)).sum().backward()
    dt = [_relcos(t.grad, g.reshape(t.shape)) for t, g in zip(in2, dtaps_o)]
    decs = sorted(((_relcos(d[n].grad, g)[0], n) for n, g in dec_o.items() if d[n].grad is not None), reverse=True)
    med_dec = float(np.median([r for r, _ in decs]))
    rep.append(f"decoder alone (oracle taps in): pred relL2 {e_pred2:.3e}; d(taps) relL2 " + " ".join(f"{x[0]:.3f}" for x in dt) + "; norm ratio " +
               " ".join(f"{x[1]:.3f}" for x in dt) + "; cos " + " ".join(f"{x[2]:.4f}" for x in dt) +
               f"; parameter gradients median relL2 {med_dec:.3e}, worst {decs[0][0]:.3e} ({decs[0][1]})")
    assert e_pred2 <= 2e-2, rep[-1]
    assert min(x[2] for x in dt) >= 0.93 and all(0.85 <= x[1] <= 1.15 for x in dt) and med_dec <= 0.35, rep[-1]

    # ---- 4. end to end against the golden
    zero()
    pred, _, _ = m(a.to(gpu), v.to(gpu), "fusion")
    e_max, e_l2 = _rel(pred, z["pred"])
    (pred * w.to(gpu)).sum().backward()
    flat = torch.cat([(d[n].grad if d[n].grad is not None else torch.zeros_like(d[n])).reshape(-1).float().cpu() for n in names])[::97]
    ref = torch.as_tensor(z["grads_sample"])
    cos = float(torch.dot(flat, ref) / (flat.norm() * ref.norm()))
    ref_norms = np.asarray(z["grad_norms"])
    ratios = [float(d[n].grad.norm()) / float(rn) for n, rn in zip(names, ref_norms) if d[n].grad is not None and rn > 1e-3 * ref_norms.max()]
    for n, rn in zip(names, ref_norms):
        if d[n].grad is None:
            assert rn == 0, n
        else:
            assert torch.isfinite(d[n].grad).all(), n
    med_ratio = float(np.median(ratios))
    rep.append(f"end to end: pred max/scale={e_max:.3e} relL2={e_l2:.3e}; gradient sample cosine {cos:.4f}, per-tensor norm ratio median {med_ratio:.3f} "
               f"(min {min(ratios):.3f}, max {max(ratios):.3f})")
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/model_parity_report.txt", "a") as f:
        f.write("avs_full_tiny_evalbn (split at the taps)\n" + "".join("    " + r + "\n" for r in rep))
    assert e_max <= 4.5e-2 and e_l2 <= 3.5e-2, rep[-1]
    assert cos >= 0.9 and 0.7 <= med_ratio <= 1.4, rep[-1]


def test_avs_train_mode_loop(stg, gpu):
    """train(): DropPath in the backbone, BatchNorm on batch statistics (running statistics move), the AVS loss on the first frame
    of each clip (AVS/loss.py:7-26) goes down on a repeated batch."""
    from stgcma import recipe
    from stgcma.model import Swin_AVSModel
    torch.manual_seed(0)
    m = Swin_AVSModel.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=128, depths=[2, 2, 2, 2],
                                                    num_heads=[4, 8, 16, 32], ftmode="fusion",
                                                    adapter_mlp_ratio=[0.25, 0.25, 0.125, 0.125]).to(gpu).train()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "W_z.1.weight" in n:
                p.fill_(0.5)                                              # the reference zero-initialises TPAVI's BatchNorm scale
    opt = recipe.build_optimizer(m, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(1, 5, 224, 224, generator=g) * 0.5).to(gpu)
    v = torch.randn(1, 5, 3, 224, 224, generator=g).to(gpu)
    mask = (torch.rand(1, 1, 224, 224, generator=g) < 0.3).float().to(gpu)
    rm0 = m.avstask_tpavi_b1.W_z[1].running_mean.clone()
    losses = []
    for _ in range(5):
        pred, fmaps, afeas = m(a, v, "fusion")
        first = torch.sigmoid(pred)[::5]                                  # index_select of every 5th prediction (:17-20)
        loss = torch.nn.BCELoss()(first, mask)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert not torch.equal(m.avstask_tpavi_b1.W_z[1].running_mean, rm0)
    assert int(m.avstask_tpavi_b1.W_z[1].num_batches_tracked) == 5


def test_avs_whole_step_graph_equals_eager_steps(stg, gpu):
    """bench.py's N = 1 form for the AVS model (BatchNorm over the batch: no micro-batches): ONE HIP graph of forward + loss + backward + Adam
    (recipe.capture_train_step_ddp(..., collective_in_graph=True) without a GradSync).  Replayed steps must walk the same losses as eager
    steps from the same state, and move the BatchNorm running statistics the same way (DropPath off: its masks are random)."""
    from stgcma import recipe
    from stgcma.model import Swin_AVSModel

    def build():
        torch.manual_seed(0)
        m = Swin_AVSModel.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=128, depths=[2, 2, 2, 2],
                                                        num_heads=[4, 8, 16, 32], ftmode="fusion", drop_path_rate=0.0,
                                                        adapter_mlp_ratio=[0.25, 0.25, 0.125, 0.125]).to(gpu).train()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "W_z.1.weight" in n:
                    p.fill_(0.5)
        recipe.apply_freeze(m)
        return m, recipe.build_optimizer(m, lr=2e-5, capturable=True)      # small steps: the comparison is of trajectories, keep them tame

    g = torch.Generator().manual_seed(1)
    a = (torch.randn(2, 5, 224, 224, generator=g) * 0.5).to(gpu)
    v = torch.randn(2, 5, 3, 224, 224, generator=g).to(gpu)
    mask = (torch.rand(2, 1, 224, 224, generator=g) < 0.3).float().to(gpu)
    bce = torch.nn.BCELoss()

    def make_fwd_bwd(m, opt):
        def fwd_bwd():
            pred, _, _ = m(a, v, "fusion")
            loss = bce(torch.sigmoid(pred)[::5], mask)
            opt.zero_grad()
            loss.backward()
            return loss
        return fwd_bwd

    m_e, opt_e = build()
    fb = make_fwd_bwd(m_e, opt_e)
    eager = []
    for _ in range(4):
        loss = fb(); opt_e.step()
        eager.append(float(loss))
    m_r, opt_r = build()
    replay, static_loss, how = recipe.capture_train_step_ddp(make_fwd_bwd(m_r, opt_r), opt_r, None, warmup=1, collective_in_graph=True)
    assert how == "one_graph"
    replayed = []
    for _ in range(3):
        replay()
        replayed.append(float(static_loss))
    assert all(np.isfinite(replayed)), replayed
    for k in range(3):
        assert abs(replayed[k] - eager[k + 1]) <= 5e-3 * max(1.0, abs(eager[k + 1])), (eager, replayed)
    bn_e, bn_r = m_e.avstask_tpavi_b1.W_z[1], m_r.avstask_tpavi_b1.W_z[1]
    assert int(bn_r.num_batches_tracked) == int(bn_e.num_batches_tracked) == 4
    assert torch.allclose(bn_r.running_mean, bn_e.running_mean, rtol=2e-2, atol=2e-3)
    # (the parameters themselves are not compared: Adam moves a parameter by ~lr per step whatever its gradient's size, so the ones whose
    # gradients are noise -- zero-initialised gates -- differ by sign between two runs of the SAME form; the losses above depend on all of them)
