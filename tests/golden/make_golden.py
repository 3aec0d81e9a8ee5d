"""Generate the golden vectors in tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference with a
timm shim) on seeded parameters and inputs.  Runs only in the build container (the reference never travels); the .npz
files and this script are committed.  Usage:  python tests/golden/make_golden.py [case ...]

Each fixture stores: `shapes_json` (state_dict float keys + shapes = the naming contract), seeds, outputs, and gradients
of the trainable tensors (the reference's name filter, traintest_adapt_ave29.py:38-61) for a seeded upstream gradient.
Parameters / inputs are regenerated from the seeds by tests/golden/params.py.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import params as GP  # noqa: E402

REF = "/root/reference"


def install_shims():
    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    class DropPath(nn.Module):  # timm 0.4.5 semantics
        def __init__(self, p=0.):
            super().__init__()
            self.drop_prob = p

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep).div_(keep)
            return x * m

    timm, tm, tl = types.ModuleType("timm"), types.ModuleType("timm.models"), types.ModuleType("timm.models.layers")
    tl.DropPath, tl.to_2tuple, tl.trunc_normal_ = DropPath, to_2tuple, trunc_normal_
    timm.models, tm.layers = tm, tl
    ipdb = types.ModuleType("ipdb")
    ipdb.set_trace = lambda *a, **k: None
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl, "clip": types.ModuleType("clip"),
                        "loratorch": types.ModuleType("loratorch"), "ipdb": ipdb, "torchvision": tv,
                        "torchvision.models": tvm})
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def seed_module(mod, seed, state_fn=None):
    sd = mod.state_dict()
    shapes = GP.float_shapes(sd)
    new = (state_fn or GP.seeded_state)(shapes, seed)
    sd.update(new)
    mod.load_state_dict(sd, strict=True)
    return shapes


def apply_freeze(mod):
    names = []
    for n, p in mod.named_parameters():
        p.requires_grad = GP.is_trainable(n)
        if p.requires_grad:
            names.append(n)
    return names


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


def flat_grads(mod, names):
    d = dict(mod.named_parameters())
    return torch.cat([(d[n].grad if d[n].grad is not None else torch.zeros_like(d[n])).reshape(-1) for n in names])


# --------------------------------------------------------------------------------------------------- Swin blocks
def swin_block_case(S, tag, *, dim, res, T, B, heads, shift, t_attn, ratio, mode, seed):
    blk = S.SwinTransformerBlock(dim=dim, input_resolution=(res, res), num_frames=T, num_heads=heads, window_size=7,
                                 shift_size=shift, t_attn=t_attn, adapter_mlp_ratio=ratio, mode=mode).eval()
    shapes = seed_module(blk, seed)
    names = apply_freeze(blk)
    BT, N = B * T, res * res
    v = GP.seeded_tensor((BT, N, dim), seed + 1).requires_grad_(True)
    a = GP.seeded_tensor((BT, N, dim), seed + 2).requires_grad_(True)
    gv, ga = GP.seeded_tensor((BT, N, dim), seed + 3), GP.seeded_tensor((BT, N, dim), seed + 4)
    if mode in ("fusion_adapt", "multimodal_adapt_no_fusion"):
        ov, oa = blk((v, a))
        ((ov * gv).sum() + (oa * ga).sum()).backward()
        extra = dict(out_v=ov, out_a=oa, din_v=v.grad, din_a=a.grad)
    else:
        x = v if mode == "video_adapt" else a
        o = blk(x)
        (o * gv).sum().backward()
        extra = dict(out=o, din=x.grad)
    # structural known-answers: the reference's integer buffers
    bufs = {k: val for k, val in blk.state_dict().items() if not val.is_floating_point()}
    mask = blk.attn_mask if blk.attn_mask is not None else torch.zeros(0)
    save(tag, shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(dim=dim, res=res, T=T, B=B, heads=heads, shift=shift,
         t_attn=t_attn, ratio=ratio, mode=mode, seed=seed)), grad_names_json=json.dumps(names), grads=flat_grads(blk, names),
         attn_mask=mask, rel_index=bufs["attn.relative_position_index"], **extra)


# --------------------------------------------------------------------------------------------------- Swin models
def swin_model_case(S, tag, *, cfg, B, mode, seed, store_all_grads=True, state_fn=None):
    m = S.SwinTransformer2D_Adapter_New(label_dim=cfg["label_dim"], patch_size=[1, 4, 4], num_frames=cfg["num_frames"],
                                        embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                                        window_size=7, pretrained=None, ftmode=mode,
                                        adapter_mlp_ratio=cfg["adapter_mlp_ratio"],
                                        **({"t_relative": cfg["t_relative"]} if "t_relative" in cfg else {})).eval()
    shapes = seed_module(m, seed, state_fn)
    names = apply_freeze(m)
    T = cfg["num_frames"]
    a = GP.seeded_tensor((B, T, 224, 224), seed + 1, 0.5)
    v = GP.seeded_tensor((B, 3, T, 224, 224), seed + 2)
    logits = m(a, v, mode)
    tgt = torch.softmax(GP.seeded_tensor((B * T, cfg["label_dim"]), seed + 3, 2.0), -1)
    loss = nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    g = flat_grads(m, names)
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    n_head = sum(p.numel() for n, p in m.named_parameters() if n in GP.MLP_HEAD)
    arrs = dict(shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(cfg, B=B, mode=mode, seed=seed)),
                grad_names_json=json.dumps(names), logits=logits, loss=loss.reshape(1),
                n_params=np.array([sum(p.numel() for p in m.parameters()), n_train, n_head]))
    if store_all_grads:
        arrs["grads"] = g
    else:  # big model: per-tensor L2 norms + a strided sample
        d = dict(m.named_parameters())
        arrs["grad_norms"] = torch.stack([d[n].grad.norm() for n in names])
        arrs["grads_sample"] = g[::97].clone()
    save(tag, **arrs)


# --------------------------------------------------------------------------------------------------- AVS / AVQA backbones
BACKBONE_PREFIXES = ("patch_embed.", "patch_embed_audio.", "layers.", "norm.", "temporal_embedding")


def _seed_backbone(m, seed):
    """Seed only the backbone tensors (the decoder / QA-head parameters of the reference classes are not on this path)."""
    sd = m.state_dict()
    shapes = [(k, sh) for k, sh in GP.float_shapes(sd) if k.startswith(BACKBONE_PREFIXES)]
    sd.update(GP.seeded_state(shapes, seed))
    m.load_state_dict(sd, strict=True)
    names = []
    for n, p in m.named_parameters():
        p.requires_grad = GP.is_trainable(n) and n.startswith(BACKBONE_PREFIXES)
        if p.requires_grad:
            names.append(n)
    return shapes, names


def _trel(cfg):
    return {"t_relative": cfg["t_relative"]} if "t_relative" in cfg else {}


def _add_temporal(m, xv, xa, B, T):
    """t_relative=False: the absolute temporal embeddings behind the patch embeddings, on the reference model's own parameters
    (Swin_AVSModel.py:1800-1806, Swin_AVQAModel_V1.py:1752-1758: '(b t) n c -> (b n) t c', + embedding, back)."""
    if m.t_relative:
        return xv, xa
    from einops import rearrange
    out = []
    for x, e in ((xv, m.temporal_embedding), (xa, m.temporal_embedding_audio)):
        x = rearrange(x, '(b t) n c -> (b n) t c', b=B, t=T) + e
        out.append(rearrange(x, '(b n) t c -> (b t) n c', b=B, t=T))
    return out


def avs_backbone_case(A, tag, *, cfg, B, seed):
    """Backbone part of SwinTransformer2D_Adapter_AVS.forward[fusion] (AVS/model/Swin_AVSModel.py:1793-1822), run through the
    reference's own modules: patch embeds, BasicLayers returning (x, x_before_downsample), final norm."""
    from einops import rearrange
    m = A.SwinTransformer2D_Adapter_AVS(pretrained=None, num_frames=cfg["num_frames"], embed_dim=cfg["embed_dim"],
                                        depths=cfg["depths"], num_heads=cfg["num_heads"], ftmode="fusion",
                                        adapter_mlp_ratio=cfg["adapter_mlp_ratio"], **_trel(cfg)).eval()
    shapes, names = _seed_backbone(m, seed)
    T = cfg["num_frames"]
    a = GP.seeded_tensor((B, T, 224, 224), seed + 1, 0.5)
    v = GP.seeded_tensor((B, T, 3, 224, 224), seed + 2)
    xv, _, _ = m.patch_embed(rearrange(v, 'b t c h w -> b c t h w'))
    xa, _, _ = m.patch_embed_audio(a.unsqueeze(1))
    xv, xa = _add_temporal(m, xv, xa, B, T)
    x = (m.pos_drop(xv), m.pos_drop(xa))
    taps = []
    for idx, layer in enumerate(m.layers):
        x, before = layer(x)
        taps.append(before[0] if idx != len(m.layers) - 1 else m.norm(before[0]))
    f_a = m.norm(x[1])
    gs = [GP.seeded_tensor(t.shape, seed + 10 + i) for i, t in enumerate(taps)]
    ga = GP.seeded_tensor(f_a.shape, seed + 20)
    (sum((t * g).sum() for t, g in zip(taps, gs)) + (f_a * ga).sum()).backward()
    save(tag, shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(cfg, B=B, seed=seed)), grad_names_json=json.dumps(names),
         grads=flat_grads(m, names), f_a=f_a, tap3=taps[3], **{f"tap{i}": taps[i][:, ::7] for i in range(3)})


def avqa_backbone_case(Q, tag, *, cfg, B, seed):
    """Backbone part of SwinTransformer2D_Adapter_AVQA.forward[fusion] (AVQA/model/Swin_AVQAModel_V1.py:1742-1766): the
    (v, a, v_nega) triple through every BasicLayer, then the final norm of each stream."""
    from einops import rearrange
    m = Q.SwinTransformer2D_Adapter_AVQA(pretrained=None, num_frames=cfg["num_frames"], embed_dim=cfg["embed_dim"],
                                         depths=cfg["depths"], num_heads=cfg["num_heads"], ftmode="fusion",
                                         adapter_mlp_ratio=cfg["adapter_mlp_ratio"], **_trel(cfg)).eval()
    shapes, names = _seed_backbone(m, seed)
    T = cfg["num_frames"]
    a = GP.seeded_tensor((B, T, 224, 224), seed + 1, 0.5)
    v = GP.seeded_tensor((B, T, 3, 224, 224), seed + 2)
    vn = GP.seeded_tensor((B, T, 3, 224, 224), seed + 3)
    xv, _, _ = m.patch_embed(rearrange(v, 'b t c h w -> b c t h w'))
    xa, _, _ = m.patch_embed_audio(a.unsqueeze(1))
    xn, _, _ = m.patch_embed(rearrange(vn, 'b t c h w -> b c t h w'))
    xv, xa = _add_temporal(m, xv, xa, B, T)              # the negative clip gets none (Swin_AVQAModel_V1.py:1752-1758)
    xv, xa, xn = m.pos_drop(xv), m.pos_drop(xa), m.pos_drop(xn)
    for layer in m.layers:
        xv, xa, xn = layer((xv, xa, xn))
    f_v, f_a, f_n = m.norm(xv), m.norm(xa), m.norm(xn)
    gv, ga = GP.seeded_tensor(f_v.shape, seed + 10), GP.seeded_tensor(f_a.shape, seed + 11)
    ((f_v * gv).sum() + (f_a * ga).sum() + f_n.sum()).backward()
    save(tag, shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(cfg, B=B, seed=seed)), grad_names_json=json.dumps(names),
         grads=flat_grads(m, names), f_v=f_v, f_a=f_a, f_nega=f_n)


def avqa_full_case(Q, tag, *, cfg, B, seed, state_fn=None):
    """The whole SwinTransformer2D_Adapter_AVQA.forward[fusion] (AVQA/model/Swin_AVQAModel_V1.py:1740-1903): backbone + QA head
    (question LSTM, grounding on the positive and the negative clip, two single-query attentions, fusion MLPs), eval mode,
    every trainable tensor of the AVQA loop's name filter (traintest_adapt_avqa.py:72: adapters + `avqatask_`) with a gradient.
    The head's widths are hard-coded to 1536, so embed_dim must be 192."""
    m = Q.SwinTransformer2D_Adapter_AVQA(pretrained=None, grounding_pretrained=None, num_frames=cfg["num_frames"],
                                         embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"], ftmode="fusion",
                                         adapter_mlp_ratio=cfg["adapter_mlp_ratio"]).eval()
    shapes = seed_module(m, seed, state_fn)
    names = apply_freeze(m)
    T = cfg["num_frames"]
    a = GP.seeded_tensor((B, T, 224, 224), seed + 1, 0.5)
    v = GP.seeded_tensor((B, T, 3, 224, 224), seed + 2)
    vn = GP.seeded_tensor((B, T, 3, 224, 224), seed + 3)
    question = torch.randint(0, 93, (B, 14), generator=torch.Generator().manual_seed(seed + 4))
    out_qa, mp, mn = m(a, v, vn, question, "fusion")
    g1, g2, g3 = GP.seeded_tensor(out_qa.shape, seed + 5), GP.seeded_tensor(mp.shape, seed + 6), GP.seeded_tensor(mn.shape, seed + 7)
    ((out_qa * g1).sum() + (mp * g2).sum() + (mn * g3).sum()).backward()
    d = dict(m.named_parameters())
    g = flat_grads(m, names)
    save(tag, shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(cfg, B=B, seed=seed)), grad_names_json=json.dumps(names),
         out_qa=out_qa, out_match_posi=mp, out_match_nega=mn, question=question,
         grad_norms=torch.stack([(d[n].grad if d[n].grad is not None else torch.zeros(())).norm() for n in names]),
         grads_sample=g[::197].clone())


# --------------------------------------------------------------------------------------------------- AVS dense decoder
def _seed_all(mod, seed):
    sd = mod.state_dict()
    shapes = GP.float_shapes(sd)
    sd.update(GP.seeded_state(shapes, seed))
    mod.load_state_dict(sd, strict=True)
    return shapes


def avs_modules_case(A, tag, seed):
    """The decoder's building blocks of the REFERENCE at small sizes (they are size-agnostic nn.Modules): Classifier_Module (ASPP,
    Swin_AVSModel_Base.py:14-29), FeatureFusionBlock (:81-112, incl. the in-place ReLU of ResidualConvUnit :56-75), TPAVIModule
    (TPAVI.py, mode='dot', with audio; train-mode BatchNorm and eval-mode), and the output_conv stack (:1497-1503)."""
    import sys as _s
    TP = _s.modules["AVS.model.TPAVI"]
    arrs = {}
    g = torch.Generator().manual_seed(seed)
    # ASPP
    aspp = A.Classifier_Module([3, 6, 12, 18], [3, 6, 12, 18], 24, 16)
    arrs["aspp_shapes"] = json.dumps(_seed_all(aspp, seed + 1))
    x = torch.randn(3, 16, 14, 14, generator=g).requires_grad_(True)
    y = aspp(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    arrs.update(aspp_x=x.detach(), aspp_y=y, aspp_gy=gy, aspp_dx=x.grad, aspp_grads=torch.cat([p.grad.reshape(-1) for p in aspp.parameters()]))
    # FeatureFusionBlock with two inputs (in-place ReLU semantics of the RCUs included)
    ffb = A.FeatureFusionBlock(16)
    arrs["ffb_shapes"] = json.dumps(_seed_all(ffb, seed + 2))
    x0 = torch.randn(2, 16, 7, 7, generator=g); x1 = torch.randn(2, 16, 7, 7, generator=g)
    a0, a1 = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
    y = ffb(a0 * 1.0, a1 * 1.0)                       # * 1.0: non-leaf copies, so the module's in-place ops are legal
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    arrs.update(ffb_x0=x0, ffb_x1=x1, ffb_y=y, ffb_gy=gy, ffb_dx0=a0.grad, ffb_dx1=a1.grad,
                ffb_grads=torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ffb.parameters()]))
    ffb1 = A.FeatureFusionBlock(16)
    ffb1.load_state_dict(ffb.state_dict())
    b0 = x0.clone().requires_grad_(True)
    y1 = ffb1(b0 * 1.0)                               # single-input form (path4, :1887)
    y1.backward(gy)
    arrs.update(ffb1_y=y1, ffb1_dx0=b0.grad)
    # TPAVI, train-mode BatchNorm then eval-mode
    tp = TP.TPAVIModule(in_channels=32, mode='dot')
    arrs["tpavi_shapes"] = json.dumps(_seed_all(tp, seed + 3))
    with torch.no_grad():
        tp.W_z[1].running_mean.copy_(torch.randn(32, generator=g) * 0.1); tp.W_z[1].running_var.copy_(torch.rand(32, generator=g) + 0.5)
    xt = torch.randn(2, 32, 3, 5, 5, generator=g); au = torch.randn(2, 3, 128, generator=g)
    for mode in ("train", "eval"):
        tp.train(mode == "train")
        rm0, rv0 = tp.W_z[1].running_mean.clone(), tp.W_z[1].running_var.clone()
        xr, ar = xt.clone().requires_grad_(True), au.clone().requires_grad_(True)
        for p in tp.parameters():
            p.grad = None
        z_, at_ = tp(xr, ar)
        gz = torch.randn(z_.shape, generator=g); ga = torch.randn(at_.shape, generator=g)
        ((z_ * gz).sum() + (at_ * ga).sum()).backward()
        arrs.update({f"tpavi_{mode}_z": z_, f"tpavi_{mode}_a": at_, f"tpavi_{mode}_gz": gz, f"tpavi_{mode}_ga": ga,
                     f"tpavi_{mode}_dx": xr.grad, f"tpavi_{mode}_da": ar.grad, f"tpavi_{mode}_rm0": rm0, f"tpavi_{mode}_rv0": rv0,
                     f"tpavi_{mode}_rm1": tp.W_z[1].running_mean.clone(), f"tpavi_{mode}_rv1": tp.W_z[1].running_var.clone(),
                     f"tpavi_{mode}_grads": torch.cat([p.grad.reshape(-1) for p in tp.parameters()])})
    arrs.update(tpavi_x=xt, tpavi_audio=au)
    # output_conv stack
    oc = nn.Sequential(nn.Conv2d(16, 24, kernel_size=3, stride=1, padding=1), A.Interpolate(scale_factor=2, mode="bilinear"),
                       nn.Conv2d(24, 8, kernel_size=3, stride=1, padding=1), nn.ReLU(True), nn.Conv2d(8, 1, kernel_size=1, stride=1, padding=0))
    arrs["oc_shapes"] = json.dumps(_seed_all(oc, seed + 4))
    xo = torch.randn(2, 16, 6, 6, generator=g).requires_grad_(True)
    yo = oc(xo)
    go = torch.randn(yo.shape, generator=g)
    yo.backward(go)
    arrs.update(oc_x=xo.detach(), oc_y=yo, oc_gy=go, oc_dx=xo.grad, oc_grads=torch.cat([p.grad.reshape(-1) for p in oc.parameters()]))
    save(tag, seed=np.array([seed]), **arrs)


def avs_tpavi_vv_case(A, tag, seed):
    """TPAVIModule WITHOUT audio (TPAVI.py:96-98: `audio = x`): the visual self-attention form behind tpavi_vv_flag=True
    (Swin_AVSModel_Base.py:1532-1538, :1876-1879), mode 'dot', train-mode BatchNorm then eval-mode.  align_channel gets no gradient."""
    import sys as _s
    TP = _s.modules["AVS.model.TPAVI"]
    arrs = {}
    g = torch.Generator().manual_seed(seed)
    tp = TP.TPAVIModule(in_channels=32, mode='dot')
    arrs["tpavi_shapes"] = json.dumps(_seed_all(tp, seed + 3))
    with torch.no_grad():
        tp.W_z[1].running_mean.copy_(torch.randn(32, generator=g) * 0.1); tp.W_z[1].running_var.copy_(torch.rand(32, generator=g) + 0.5)
    xt = torch.randn(2, 32, 3, 5, 5, generator=g)
    for mode in ("train", "eval"):
        tp.train(mode == "train")
        rm0, rv0 = tp.W_z[1].running_mean.clone(), tp.W_z[1].running_var.clone()
        xr = xt.clone().requires_grad_(True)
        for p in tp.parameters():
            p.grad = None
        z_, at_ = tp(xr)
        assert at_ == 0
        gz = torch.randn(z_.shape, generator=g)
        (z_ * gz).sum().backward()
        arrs.update({f"tpavi_{mode}_z": z_, f"tpavi_{mode}_gz": gz, f"tpavi_{mode}_dx": xr.grad, f"tpavi_{mode}_rm0": rm0, f"tpavi_{mode}_rv0": rv0,
                     f"tpavi_{mode}_rm1": tp.W_z[1].running_mean.clone(), f"tpavi_{mode}_rv1": tp.W_z[1].running_var.clone(),
                     f"tpavi_{mode}_grads": torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in tp.parameters()])})
    arrs.update(tpavi_x=xt)
    save(tag, seed=np.array([seed]), **arrs)


def avs_full_case(AB, tag, *, cfg, B, seed, state_fn=None):
    """The whole SwinTransformer2D_Adapter_AVS_Base.forward[fusion] (AVS/model/Swin_AVSModel_Base.py:1790-1894): backbone + dense
    decoder, train mode without DropPath (BatchNorm on batch statistics), seeded upstream gradients on pred, the returned feature
    maps and the audio features so that every output is pinned."""
    m = AB.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                              num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"],
                                              channel=256, opt=None, config=None, vis_dim=[64, 128, 320, 512], tpavi_stages=[0, 1, 2, 3],
                                              tpavi_vv_flag=False, tpavi_va_flag=True, drop_path_rate=0.0).train()
    # train mode with drop_path_rate = 0: deterministic, and TPAVI's BatchNorm runs on BATCH statistics -- with seeded running
    # statistics (eval) single channels of W_z's output reach the hundreds and the LayerNorm behind it becomes a difference of large
    # numbers, which pins nothing at bf16
    shapes = seed_module(m, seed, state_fn)
    with torch.no_grad():                              # BatchNorm running statistics are buffers, not covered by float seeding rules
        gg = torch.Generator().manual_seed(seed + 50)
        for n, b in m.named_buffers():
            if n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=gg) + 0.5)
            elif n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=gg) * 0.1)
        # TPAVI's BatchNorm scale: the reference zero-initialises it (TPAVI.py:62-63) so the non-local branch starts as a small
        # perturbation of the residual; with a unit-size seeded scale the branch dominates and the fp32 model's own gradients turn
        # chaotic (sub-1 % structured changes of the taps move single gradient norms tenfold), which pins nothing
        for n, p in m.named_parameters():
            if "W_z.1.weight" in n:
                p.mul_(0.1)
    names = apply_freeze(m)
    a = GP.seeded_tensor((B, 5, 224, 224), seed + 1, 0.5)
    v = GP.seeded_tensor((B, 5, 3, 224, 224), seed + 2)
    pred, fmaps, afeas = m(a, v, "fusion")
    up = GP.seeded_tensor(pred.shape, seed + 3, 1e-2)
    loss = (pred * up).sum()
    for i, (fm, af) in enumerate(zip(fmaps, afeas)):
        loss = loss + (fm * GP.seeded_tensor(fm.shape, seed + 10 + i, 1e-2)).sum() + (af * GP.seeded_tensor(af.shape, seed + 20 + i, 1e-1)).sum()
    loss.backward()
    d = dict(m.named_parameters())
    g = flat_grads(m, names)
    save(tag, shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(cfg, B=B, seed=seed)), grad_names_json=json.dumps(names),
         pred=pred, **{f"fmap{i}": fm[:, ::8] for i, fm in enumerate(fmaps)}, **{f"afea{i}": af for i, af in enumerate(afeas)},
         fmap_stats=torch.stack([torch.stack([fm.sum(), fm.abs().sum()]) for fm in fmaps]),
         grad_norms=torch.stack([(d[n].grad if d[n].grad is not None else torch.zeros(())).norm() for n in names]),
         grads_sample=g[::97].clone())


def avs_full_evalbn_case(AB, tag, *, cfg, B, seed):
    """The same model in EVAL mode with CALIBRATED BatchNorm statistics: one train-mode forward with momentum 1 sets every running
    mean / variance to the batch statistics of these very inputs, then eval() -- BatchNorm is a fixed per-channel affine map with
    O(1) outputs, so the whole-model gradients are well conditioned (the train-mode fixture above is not: batch statistics over a
    single clip).  The calibrated statistics travel in the fixture.  Upstream gradient on `pred` only."""
    m = AB.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                              num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"],
                                              channel=256, opt=None, config=None, vis_dim=[64, 128, 320, 512], tpavi_stages=[0, 1, 2, 3],
                                              tpavi_vv_flag=False, tpavi_va_flag=True, drop_path_rate=0.0).train()
    shapes = seed_module(m, seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "W_z.1.weight" in n:
                p.mul_(0.1)      # like avs_full_tiny: the reference zero-initialises it (TPAVI.py:62-63)
    a = GP.seeded_tensor((B, 5, 224, 224), seed + 1, 0.5)
    v = GP.seeded_tensor((B, 5, 3, 224, 224), seed + 2)
    bns = [mod for mod in m.modules() if isinstance(mod, nn.modules.batchnorm._BatchNorm)]
    for bn in bns:
        bn.momentum = 1.0
    with torch.no_grad():
        m(a, v, "fusion")
    m.eval()
    stats = {n: b.clone() for n, b in m.named_buffers() if n.endswith(("running_mean", "running_var"))}
    names = apply_freeze(m)
    pred, fmaps, afeas = m(a, v, "fusion")
    (pred * GP.seeded_tensor(pred.shape, seed + 3, 1e-2)).sum().backward()
    d = dict(m.named_parameters())
    g = flat_grads(m, names)
    save(tag, shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(cfg, B=B, seed=seed)), grad_names_json=json.dumps(names),
         pred=pred, stat_names_json=json.dumps(list(stats)), **{f"stat{i}": t for i, t in enumerate(stats.values())},
         grad_norms=torch.stack([(d[n].grad if d[n].grad is not None else torch.zeros(())).norm() for n in names]),
         grads_sample=g[::97].clone())


# --------------------------------------------------------------------------------------------------- ViT (CLIP) path
def vit_block_case(Cm, tag, *, d, heads, T, B, nv, na, seed, mode="fusion_adapt"):
    blk = Cm.ResidualAttentionBlock(d, heads, None, 0.5, 1, T, 0.0, mode=mode).eval()
    shapes = seed_module(blk, seed)
    names = apply_freeze(blk)
    BT = B * T
    v = GP.seeded_tensor((nv, BT, d), seed + 1).requires_grad_(True)
    a = GP.seeded_tensor((na, BT, d), seed + 2).requires_grad_(True)
    gv, ga = GP.seeded_tensor((nv, BT, d), seed + 3), GP.seeded_tensor((na, BT, d), seed + 4)
    ov, oa = blk((v, a))
    ((ov * gv).sum() + (oa * ga).sum()).backward()
    st = max(1, nv // 25)
    save(tag, shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(d=d, heads=heads, T=T, B=B, nv=nv, na=na, seed=seed,
         mode=mode, stride=st)), grad_names_json=json.dumps(names), grads=flat_grads(blk, names),
         out_v=ov[::st], out_a=oa[::st], din_v=v.grad[::st], din_a=a.grad[::st],
         stats=torch.stack([ov.sum(), ov.abs().sum(), oa.sum(), oa.abs().sum()]))


def vit_model_case(Cm, tag, *, layers, heads, d, B, T, seed, mode="fusion", state_fn=None, store_all_grads=True):
    m = Cm.MM_CLIP_AVE(label_dim=29, layers=layers, num_video_frames=T, embed_dim=d, patch_size=16, heads=heads,
                       pretrained=None, ftmode=mode).eval()
    shapes = seed_module(m, seed, state_fn)
    names = apply_freeze(m)
    a = GP.seeded_tensor((B, T, 102, 128), seed + 1, 0.5)
    v = GP.seeded_tensor((B, 3, T, 224, 224), seed + 2)
    logits = m(a, v, mode)
    tgt = torch.softmax(GP.seeded_tensor((B * T, 29), seed + 3, 2.0), -1)
    loss = nn.CrossEntropyLoss()(logits, tgt)
    loss.backward()
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    n_head = sum(p.numel() for n, p in m.named_parameters() if n in GP.MLP_HEAD)
    arrs = dict(shapes_json=json.dumps(shapes), cfg_json=json.dumps(dict(layers=layers, heads=heads, d=d, B=B, T=T, seed=seed, mode=mode)),
                grad_names_json=json.dumps(names), logits=logits, loss=loss.reshape(1),
                n_params=np.array([sum(p.numel() for p in m.parameters()), n_train, n_head]))
    g = flat_grads(m, names)
    if store_all_grads:
        arrs["grads"] = g
    else:                                                  # full depth: per-tensor L2 norms + a strided sample (like the big Swin fixtures)
        dd = dict(m.named_parameters())
        arrs["grad_norms"] = torch.stack([dd[n].grad.norm() for n in names])
        arrs["grads_sample"] = g[::97].clone()
    save(tag, **arrs)


def structure_case(S, Cm):
    """Naming / counting contract of the full-size models (no forward)."""
    out = {}
    for tag, ctor in {
        "swin_b_fusion": lambda: S.SwinTransformer2D_Adapter_New(label_dim=29, patch_size=[1, 4, 4], num_frames=10, embed_dim=128,
                                                                  depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7,
                                                                  pretrained=None, ftmode="fusion",
                                                                  adapter_mlp_ratio=[.125, .125, .0625, .0625]),
        "swin_l_fusion": lambda: S.SwinTransformer2D_Adapter_New(label_dim=29, patch_size=[1, 4, 4], img_size=224, num_frames=10,
                                                                  embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48],
                                                                  window_size=7, pretrained=None, ftmode="fusion",
                                                                  adapter_mlp_ratio=[.5, .25, .125, .0625]),
        "vit_b_fusion": lambda: Cm.MM_CLIP_AVE(label_dim=29, layers=12, num_video_frames=10, embed_dim=768, patch_size=16,
                                                heads=8, pretrained=None, ftmode="fusion"),
    }.items():
        m = ctor()
        sd = m.state_dict()
        keys = [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
        n_total = sum(p.numel() for p in m.parameters())
        n_train = sum(p.numel() for n, p in m.named_parameters() if GP.is_trainable(n))
        n_head = sum(p.numel() for n, p in m.named_parameters() if n in GP.MLP_HEAD)
        out[tag] = dict(keys=keys, n_total=n_total, n_trainable=n_train, n_head=n_head)
        print(tag, n_total, n_train, n_head)
    with open(os.path.join(HERE, "structure.json"), "w") as f:
        json.dump(out, f)
    print("wrote structure.json")


def pretrained_ingest_case(S, tag, cfg, seed, patch_size):
    """Swin checkpoint ingestion of the REFERENCE constructor (Swin_AVE.py:1363-1415): patch-embed inflation / patch_size[0],
    audio patch embedding = channel mean, strict=False load; D_fc2 zeroing afterwards (:1422-1468)."""
    import contextlib
    import io
    import tempfile
    torch.manual_seed(seed)
    probe = S.SwinTransformer2D_Adapter_New(patch_size=patch_size, window_size=7, pretrained=None, ftmode="fusion", **cfg)
    ck = GP.swin2d_checkpoint(probe.state_dict(), seed + 1)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "swin2d.pth")
        torch.save(ck, path)
        buf = io.StringIO()
        torch.manual_seed(seed + 2)
        with contextlib.redirect_stdout(buf):
            m = S.SwinTransformer2D_Adapter_New(patch_size=patch_size, window_size=7, pretrained=path, ftmode="fusion", **cfg)
    lines = buf.getvalue().splitlines()
    missing = [ln for ln in lines if ln.startswith("Missing keys: ")][0][len("Missing keys: "):]
    unexpected = [ln for ln in lines if ln.startswith("Unexpected keys: ")][0][len("Unexpected keys: "):]
    sd = m.state_dict()
    loaded = [k for k in sd if k in ck["model"] or k.startswith("patch_embed_audio.")]
    keep = {k.replace(".", "__"): sd[k] for k in sd if k.startswith("patch_embed") }
    stats = {k: [float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in loaded if sd[k].is_floating_point()}
    zeroed = [k for k in sd if "D_fc2" in k]
    assert all(float(sd[k].abs().max()) == 0 for k in zeroed)
    save(tag, cfg_json=json.dumps(dict(cfg, patch_size=patch_size, seed=seed)), missing=missing, unexpected=unexpected,
         stats_json=json.dumps(stats), **keep)


def clip_ingest_case(Cm, tag, *, layers, embed_dim, patch, res, audio_length, seed):
    """OpenAI-CLIP checkpoint ingestion of the REFERENCE constructor (CLIP_AVE.py:811-853) with `clip.load` stubbed to hand
    back a synthetic visual state_dict (the `clip` package is not installed; its only role is to produce that dict)."""
    import contextlib
    import io
    sd_clip = GP.clip_visual_state(embed_dim, res // patch, layers, patch, seed)

    class _Vis:
        def state_dict(self):
            return {k: v.clone() for k, v in sd_clip.items()}

    class _Clip:
        visual = _Vis()
    sys.modules["clip"].load = lambda name, device="cpu", download_root=None: (_Clip(), None)
    Cm.clip = sys.modules["clip"]
    buf = io.StringIO()
    torch.manual_seed(seed + 1)
    with contextlib.redirect_stdout(buf):
        m = Cm.MM_CLIP_AVE(label_dim=5, input_resolution=res, audio_length=audio_length, num_video_frames=2, patch_size=patch,
                           embed_dim=embed_dim, layers=layers, heads=4, pretrained="/nonexistent/clip", ftmode="fusion")
    lines = buf.getvalue().splitlines()
    missing = [ln for ln in lines if ln.startswith("Missing keys: ")][0][len("Missing keys: "):]
    unexpected = [ln for ln in lines if ln.startswith("Unexpected keys: ")][0][len("Unexpected keys: "):]
    sd = m.state_dict()
    loaded = [k for k in sd if k in sd_clip or k in ("conv1_audio.weight", "positional_embedding_audio")]
    stats = {k: [float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in loaded}
    save(tag, cfg_json=json.dumps(dict(layers=layers, embed_dim=embed_dim, patch=patch, res=res, audio_length=audio_length,
                                       seed=seed)), missing=missing, unexpected=unexpected, stats_json=json.dumps(stats),
         conv1_audio=sd["conv1_audio.weight"], pos_audio=sd["positional_embedding_audio"])


def avqa_ingest_case(Q, tag, seed):
    """SwinTransformer2D_Adapter_AVQA(pretrained=<swin ckpt>, grounding_pretrained=<grounding ckpt>) of the REFERENCE
    (Swin_AVQAModel_V1.py:1500-1590) on synthetic checkpoints."""
    import contextlib
    import io
    import tempfile
    cfg = dict(num_frames=2, embed_dim=32, depths=[2, 2], num_heads=[1, 2], adapter_mlp_ratio=[0.5, 0.25])
    torch.manual_seed(seed)
    probe = Q.SwinTransformer2D_Adapter_AVQA(pretrained=None, grounding_pretrained=None, ftmode="fusion", **cfg)
    ck = GP.swin2d_checkpoint({k: v for k, v in probe.state_dict().items() if not k.startswith("avqatask_")}, seed + 1)
    gk = GP.grounding_checkpoint(seed + 2)
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "swin2d.pth"), os.path.join(d, "grounding.pt")
        torch.save(ck, p1); torch.save(gk, p2)
        buf = io.StringIO()
        torch.manual_seed(seed + 3)
        with contextlib.redirect_stdout(buf):
            m = Q.SwinTransformer2D_Adapter_AVQA(pretrained=p1, grounding_pretrained=p2, ftmode="fusion", **cfg)
    lines = buf.getvalue().splitlines()
    missing = [ln for ln in lines if ln.startswith("Missing keys: ")][0][len("Missing keys: "):]
    unexpected = [ln for ln in lines if ln.startswith("Unexpected keys: ")][0][len("Unexpected keys: "):]
    sd = m.state_dict()
    loaded = [k for k in sd if k in ck["model"] or k.startswith("patch_embed_audio.") or k.replace("avqatask_", "module.") in gk]
    stats = {k: [float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in loaded if sd[k].is_floating_point()}
    save(tag, cfg_json=json.dumps(dict(cfg, seed=seed)), missing=missing, unexpected=unexpected, stats_json=json.dumps(stats))


def video_aug_case(tag, seeds):
    """SURVEY f4, video half: the tensor part of the reference's training-time frame pipeline, AVE/dataloader.py:346-394 (_aug_frame_train
    behind the PIL RandAugment): ToTensor -> tensor_normalize (:470-485) -> spatial_sampling (:396-468: random_resized_crop, scale [0.08, 1],
    ratio [3/4, 4/3], bilinear to 224 x 224, then horizontal_flip) -> RandomErasing(0.25, 'pixel', cube) (transforms/random_erasing.py).
    The reference's own functions are called (dataloader.py / video_transforms.py import with empty stand-ins for the ABSENT torchvision / cv2 /
    torchaudio / h5py modules: none of the functions used here touches them; ToTensor, a torchvision class, is the one restated line).  The
    random draws they make (crop box, flip, erase box, erase noise) are recorded and stored: the device kernel takes them as arguments."""
    import random
    for name in ("cv2", "torchaudio", "h5py"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = sys.modules["torchvision"]
    tvt, tvf = types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    sys.modules.update({"torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    DL = load(os.path.join(REF, "AVE/dataloader.py"), "ref_ave_dataloader")
    VT, RE = DL.video_transforms, DL.random_erasing
    ds = object.__new__(DL.AudiosetDataset)
    out = {}
    for ci, (seed, T, H, W) in enumerate(seeds):
        g = torch.Generator().manual_seed(seed)
        # smooth synthetic frames (a random low-resolution field, upsampled) + noise: interpolation weights matter on such data
        low = torch.rand((T, 3, H // 16 + 1, W // 16 + 1), generator=g)
        frames = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=False).clamp(0, 1)
        frames = (frames * 235 + torch.rand((T, 3, H, W), generator=g) * 20).round().clamp(0, 255).to(torch.uint8)     # T C H W, what ToPILImage / RandAugment hand on
        frames_hwc = frames.permute(0, 2, 3, 1).contiguous()
        buffer = frames.float().div(255)                               # transforms.ToTensor()(img): uint8 HWC -> float CHW in [0, 1]
        buffer = buffer.permute(0, 2, 3, 1)                            # dataloader.py:359-360
        buffer = ds.tensor_normalize(buffer, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
        buffer = buffer.permute(3, 0, 1, 2)                            # C T H W (:366)
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        rec = {}
        crop0, flip0, pix0 = VT._get_param_spatial_crop, VT.horizontal_flip, RE._get_pixels

        def crop_rec(*a, **k):
            r = crop0(*a, **k)
            rec["crop"] = r
            return r

        def flip_rec(prob, images, boxes=None):
            o, b = flip0(prob, images, boxes)
            rec["flip"] = int(not torch.equal(o, images))
            return o, b

        def pix_rec(per_pixel, rand_color, patch_size, dtype=torch.float32, device="cpu"):
            t = pix0(per_pixel, rand_color, patch_size, dtype=dtype, device=device)
            rec.setdefault("noise", []).append(t.clone())
            return t
        VT._get_param_spatial_crop, VT.horizontal_flip, RE._get_pixels = crop_rec, flip_rec, pix_rec
        try:
            buffer = ds.spatial_sampling(buffer, spatial_idx=-1, min_scale=256, max_scale=320, crop_size=224, random_horizontal_flip=True,
                                         inverse_uniform_sampling=False, aspect_ratio=[0.75, 1.3333], scale=[0.08, 1.0], motion_shift=False)
            before = buffer.clone()
            erase = RE.RandomErasing(0.25, mode="pixel", max_count=1, num_splits=1, device="cpu")
            buffer = buffer.permute(1, 0, 2, 3)                        # T C H W (:389)
            buffer = erase(buffer)
            buffer = (buffer if buffer is not None else before.permute(1, 0, 2, 3)).permute(1, 0, 2, 3)
        finally:
            VT._get_param_spatial_crop, VT.horizontal_flip, RE._get_pixels = crop0, flip0, pix0
        # RandomErasing works in place (its __call__ returns None): `before` shares no storage, the erased tensor is the permuted view's base
        erased = before.permute(1, 0, 2, 3).clone()
        noise = torch.zeros((T, 3, 224, 224))
        box = (0, 0, 0, 0)
        if rec.get("noise"):
            # redo the erase on a copy to learn the box: the positions where the in-place result differs from `before`
            random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        final = buffer
        diff = (final != before).any(dim=0).any(dim=0)                 # [224, 224]
        if rec.get("noise"):
            ys, xs = torch.nonzero(diff.any(dim=1)).flatten(), torch.nonzero(diff.any(dim=0)).flatten()
            eh, ew = rec["noise"][0].shape[1:]
            top, left = int(ys.min()), int(xs.min())
            assert int(ys.max()) - top + 1 == eh and int(xs.max()) - left + 1 == ew
            for t in range(T):
                noise[t, :, top:top + eh, left:left + ew] = rec["noise"][t]
            box = (top, left, eh, ew)
        i, j, h, w = rec["crop"]
        out[f"frames{ci}"] = frames_hwc.numpy()
        out[f"params{ci}"] = np.asarray([i, j, h, w, rec["flip"], *box], dtype=np.int32)
        out[f"noise_box{ci}"] = noise[:, :, box[0]:box[0] + box[2], box[1]:box[1] + box[3]].numpy()
        out[f"out{ci}"] = final.contiguous().numpy()
        print(f"   case {ci}: seed {seed} frames {T}x{H}x{W} crop {rec['crop']} flip {rec['flip']} erase box {box}")
    out["mean"] = np.asarray([0.485, 0.456, 0.406], np.float32)
    out["std"] = np.asarray([0.229, 0.224, 0.225], np.float32)
    out["ncases"] = np.asarray([len(seeds)])
    save(tag, **out)


def scheduler_case():
    sch = load(os.path.join(REF, "utilities/scheduler.py"), "ref_sched")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        t1 = sch.cosine_scheduler(5e-5, 2e-6, 20, 3339, warmup_epochs=2, start_warmup_value=0, warmup_steps=-1)
        t2 = sch.cosine_scheduler(5e-5 * 0.1, 2e-6, 20, 3339, warmup_epochs=2, start_warmup_value=0, warmup_steps=-1)
        t3 = sch.cosine_scheduler(1e-4, 2e-6, 3, 7, warmup_epochs=1)
    save("cosine_scheduler", t1=t1, t2=t2, t3=t3)


SWIN_TINY = dict(label_dim=29, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], num_frames=2,
                 adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
AVS_TINY = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], num_frames=3, adapter_mlp_ratio=[0.5, 0.5, 0.25, 0.25])
AVQA_TINY = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], num_frames=2, adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.125])
AVS_FULL_TINY = dict(embed_dim=128, depths=[2, 2, 2, 2], num_heads=[4, 8, 16, 32], num_frames=5, adapter_mlp_ratio=[0.25, 0.25, 0.125, 0.125])
# BASELINE config 4's model at FULL depth (AVS/run_adapt_avs.py:146-160: Swin-B, depths [2, 2, 18, 2], adapter ratios [.25, .25, .125, .125], T = 5)
AVS_FULL_B = dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], num_frames=5, adapter_mlp_ratio=[0.25, 0.25, 0.125, 0.125])
AVQA_FULL_TINY = dict(embed_dim=192, depths=[2, 2, 2, 2], num_heads=[6, 12, 24, 48], num_frames=2, adapter_mlp_ratio=[0.25, 0.125, 0.125, 0.0625])
# BASELINE config 5's model beyond depths [2, 2, 2, 2]: Swin-L widths, the AVQA runner's adapter ratios (AVQA/run_adapt_avqa.py:288-301), six
# stage-2 blocks (three temporal + three shifted), B = 1, reference-initialised backbone (GP.avqa_deep_state)
AVQA_FULL_D6 = dict(embed_dim=192, depths=[2, 2, 6, 2], num_heads=[6, 12, 24, 48], num_frames=2, adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
SWIN_B = dict(label_dim=29, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], num_frames=10,
              adapter_mlp_ratio=[0.125, 0.125, 0.0625, 0.0625])
# AVE/run_adapt_ave29.py:167-181 (MM-Swin-AVE-Large) = the backbone geometry of AVQA/run_adapt_avqa.py:288-301 (BASELINE config 5)
SWIN_L = dict(label_dim=29, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], num_frames=10,
              adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])


def main(argv):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shims()
    S = load(os.path.join(REF, "AVE/model/Swin_AVE.py"), "ref_swin_ave")
    Cm = load(os.path.join(REF, "AVE/model/CLIP_AVE.py"), "ref_clip_ave")
    lazy = {}

    def ref_avs():
        if "avs" not in lazy:
            lazy["avs"] = load(os.path.join(REF, "AVS/model/Swin_AVSModel.py"), "ref_swin_avs")
        return lazy["avs"]

    def ref_avs_base():
        if "avsb" not in lazy:
            lazy["avsb"] = load(os.path.join(REF, "AVS/model/Swin_AVSModel_Base.py"), "ref_swin_avs_base")
        return lazy["avsb"]

    def ref_avqa():
        if "avqa" not in lazy:
            lazy["avqa"] = load(os.path.join(REF, "AVQA/model/Swin_AVQAModel_V1.py"), "ref_swin_avqa")
        return lazy["avqa"]

    def ref_avqa512():
        if "avqa512" not in lazy:
            lazy["avqa512"] = load(os.path.join(REF, "AVQA/model/Swin_AVQAModel.py"), "ref_swin_avqa512")
        return lazy["avqa512"]
    cases = {
        "swin_block_even": lambda: swin_block_case(S, "swin_block_even", dim=128, res=14, T=5, B=1, heads=4, shift=0, t_attn=True,
                                                   ratio=0.125, mode="fusion_adapt", seed=100),
        "swin_block_odd": lambda: swin_block_case(S, "swin_block_odd", dim=128, res=14, T=5, B=1, heads=4, shift=3, t_attn=False,
                                                  ratio=0.125, mode="fusion_adapt", seed=110),
        "swin_block_s0": lambda: swin_block_case(S, "swin_block_s0", dim=32, res=56, T=1, B=1, heads=1, shift=3, t_attn=False,
                                                 ratio=0.5, mode="fusion_adapt", seed=120),
        "swin_block_s3": lambda: swin_block_case(S, "swin_block_s3", dim=256, res=7, T=2, B=2, heads=8, shift=3, t_attn=True,
                                                 ratio=0.0625, mode="fusion_adapt", seed=130),
        "swin_block_nofusion": lambda: swin_block_case(S, "swin_block_nofusion", dim=64, res=14, T=2, B=2, heads=2, shift=3,
                                                       t_attn=True, ratio=0.25, mode="multimodal_adapt_no_fusion", seed=140),
        "swin_block_video": lambda: swin_block_case(S, "swin_block_video", dim=64, res=14, T=2, B=2, heads=2, shift=0, t_attn=True,
                                                    ratio=0.25, mode="video_adapt", seed=150),
        "swin_block_audio": lambda: swin_block_case(S, "swin_block_audio", dim=64, res=14, T=2, B=2, heads=2, shift=3, t_attn=False,
                                                    ratio=0.25, mode="audio_adapt", seed=160),
        # wide adapters (Swin-L d_h = 96 in every stage, Swin-B stage 3 / AVS ratios d_h = 64): the frame-global cross-modal pair
        # then runs on the flash kernels of mha.hip; 196 tokens = 6 full 32-row tiles + a 4-row tail
        "swin_block_wide64": lambda: swin_block_case(S, "swin_block_wide64", dim=128, res=14, T=2, B=1, heads=4, shift=3, t_attn=True,
                                                     ratio=0.5, mode="fusion_adapt", seed=170),
        "swin_block_wide96": lambda: swin_block_case(S, "swin_block_wide96", dim=192, res=14, T=2, B=2, heads=6, shift=0, t_attn=False,
                                                     ratio=0.5, mode="fusion_adapt", seed=180),
        "swin_tiny_fusion": lambda: swin_model_case(S, "swin_tiny_fusion", cfg=SWIN_TINY, B=1, mode="fusion", seed=200),
        "swin_tiny_multimodal": lambda: swin_model_case(S, "swin_tiny_multimodal", cfg=SWIN_TINY, B=1, mode="multimodal", seed=210),
        "swin_tiny_videoonly": lambda: swin_model_case(S, "swin_tiny_videoonly", cfg=SWIN_TINY, B=1, mode="videoonly", seed=220),
        # t_relative=False (Swin_AVE.py:1207-1212, :1569-1576): trainable absolute temporal embeddings added behind the patch embedding
        "swin_tiny_fusion_tabs": lambda: swin_model_case(S, "swin_tiny_fusion_tabs", cfg=dict(SWIN_TINY, t_relative=False), B=2, mode="fusion", seed=230),
        "swin_b_fusion": lambda: swin_model_case(S, "swin_b_fusion", cfg=SWIN_B, B=1, mode="fusion", seed=300, store_all_grads=False),
        "swin_b_fusion_refinit": lambda: swin_model_case(S, "swin_b_fusion_refinit", cfg=SWIN_B, B=1, mode="fusion", seed=310,
                                                         store_all_grads=False, state_fn=GP.refinit_state),
        "vit_block_cfg1": lambda: vit_block_case(Cm, "vit_block_cfg1", d=768, heads=8, T=10, B=1, nv=196, na=196, seed=400),
        "vit_block_small": lambda: vit_block_case(Cm, "vit_block_small", d=256, heads=4, T=2, B=2, nv=50, na=13, seed=410),
        "vit_tiny_fusion": lambda: vit_model_case(Cm, "vit_tiny_fusion", layers=2, heads=8, d=768, B=1, T=2, seed=500),
        # BASELINE config 2 at FULL depth (12 layers of ViT-B/16 width, heads = 8 as the runner builds it), reference-initialisation scale, two frames
        "vit_b12_fusion_refinit": lambda: vit_model_case(Cm, "vit_b12_fusion_refinit", layers=12, heads=8, d=768, B=1, T=2, seed=520,
                                                         state_fn=GP.refinit_state, store_all_grads=False),
        "avs_tiny_backbone": lambda: avs_backbone_case(ref_avs(), "avs_tiny_backbone", cfg=AVS_TINY, B=1, seed=600),
        "avqa_tiny_backbone": lambda: avqa_backbone_case(ref_avqa(), "avqa_tiny_backbone", cfg=AVQA_TINY, B=1, seed=610),
        # t_relative=False on the AVS / AVQA classes (no runner uses it; constructor completeness): B = 2 so that the sum over clips is exercised
        "avs_tiny_backbone_tabs": lambda: avs_backbone_case(ref_avs(), "avs_tiny_backbone_tabs", cfg=dict(AVS_TINY, t_relative=False, num_frames=2), B=2, seed=650),
        "avqa_tiny_backbone_tabs": lambda: avqa_backbone_case(ref_avqa(), "avqa_tiny_backbone_tabs", cfg=dict(AVQA_TINY, t_relative=False), B=2, seed=660),
        "swin_pretrained_ingest": lambda: pretrained_ingest_case(S, "swin_pretrained_ingest", SWIN_TINY, 700, [1, 4, 4]),
        "swin_pretrained_ingest_pd2": lambda: pretrained_ingest_case(S, "swin_pretrained_ingest_pd2", SWIN_TINY, 710, [2, 4, 4]),
        # audio grid 7 x 5 inside the 14 x 14 image grid (centre crop both ways), and 7 x 19 (time axis bilinearly stretched)
        "clip_pretrained_ingest": lambda: clip_ingest_case(Cm, "clip_pretrained_ingest", layers=12, embed_dim=64, patch=16, res=224,
                                                           audio_length=1024, seed=720),
        "clip_pretrained_ingest_long": lambda: clip_ingest_case(Cm, "clip_pretrained_ingest_long", layers=12, embed_dim=64, patch=16,
                                                                res=224, audio_length=3200, seed=730),
        "avqa_full_tiny": lambda: avqa_full_case(ref_avqa(), "avqa_full_tiny", cfg=AVQA_FULL_TINY, B=2, seed=620),
        # the 512-d head variant that AVQA/test.py:8 imports (yb_fc_v / yb_fc_a projections, fc_a1)
        "avqa_full_d6": lambda: avqa_full_case(ref_avqa(), "avqa_full_d6", cfg=AVQA_FULL_D6, B=1, seed=640, state_fn=GP.avqa_deep_state),
        "avqa512_full_tiny": lambda: avqa_full_case(ref_avqa512(), "avqa512_full_tiny", cfg=AVQA_FULL_TINY, B=2, seed=630),
        # full-depth Swin-L at the reference's initialisation scale: the fixture the fp8 frozen-weight path is measured on
        "swin_l_fusion_refinit": lambda: swin_model_case(S, "swin_l_fusion_refinit", cfg=SWIN_L, B=1, mode="fusion", seed=320,
                                                         store_all_grads=False, state_fn=GP.refinit_state),
        "avs_decoder_modules": lambda: avs_modules_case(ref_avs_base(), "avs_decoder_modules", 800),
        "avs_tpavi_vv": lambda: avs_tpavi_vv_case(ref_avs_base(), "avs_tpavi_vv", 860),
        "avs_full_tiny": lambda: avs_full_case(ref_avs_base(), "avs_full_tiny", cfg=AVS_FULL_TINY, B=1, seed=820),
        "avs_full_b18": lambda: avs_full_case(ref_avs_base(), "avs_full_b18", cfg=AVS_FULL_B, B=1, seed=830),
        # the same model at the REFERENCE's initialisation scale (GP.refinit_state; decoder convolutions at 0.02 ~ their kaiming-uniform
        # default): BASELINE's absolute 1e-2 bound on pred is meaningful here, as for the Swin-B / Swin-L / ViT-B refinit fixtures
        "avs_full_b18_refinit": lambda: avs_full_case(ref_avs_base(), "avs_full_b18_refinit", cfg=AVS_FULL_B, B=1, seed=850,
                                                      state_fn=GP.refinit_state),
        "avs_full_tiny_evalbn": lambda: avs_full_evalbn_case(ref_avs_base(), "avs_full_tiny_evalbn", cfg=AVS_FULL_TINY, B=1, seed=840),
        "avqa_pretrained_ingest": lambda: avqa_ingest_case(ref_avqa(), "avqa_pretrained_ingest", 740),
        # (seed, frames, height, width): seeds picked so that the cases cover erase / no erase and flip / no flip
        "video_aug": lambda: video_aug_case("video_aug", [(2, 2, 240, 320), (4, 2, 360, 270), (3, 1, 240, 320), (5, 2, 180, 320)]),
        "structure": lambda: structure_case(S, Cm),
        "cosine_scheduler": scheduler_case,
    }
    todo = argv or list(cases)
    for c in todo:
        print("==", c)
        cases[c]()


if __name__ == "__main__":
    main(sys.argv[1:])
