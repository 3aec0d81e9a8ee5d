"""Developer aid (not a test): layer-by-layer comparison of the HIP path against the fp32 oracle for a tiny Swin model.
Usage on the GPU box:  python tests/debug_trace.py swin_tiny_videoonly"""
import sys

import torch

sys.path.insert(0, "tests")
from golden_util import build_state, load_case  # noqa: E402
from params import seeded_tensor  # noqa: E402

import oracle.swin as OS  # noqa: E402
import stgcma  # noqa: E402,F401
from stgcma import ops  # noqa: E402
from stgcma.model import Swin_AVE as S  # noqa: E402


def rel(got, ref):
    got = got.detach().float().cpu().reshape(-1)
    ref = ref.detach().float().reshape(-1)
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def main(tag):
    z, cfg, shapes, names = load_case(tag)
    mode = cfg["mode"]
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    m = S.SwinTransformer2D_Adapter_New(label_dim=29, patch_size=[1, 4, 4], num_frames=cfg["num_frames"],
                                        embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                                        window_size=7, pretrained=None, ftmode=mode, adapter_mlp_ratio=cfg["adapter_mlp_ratio"])
    sd = m.state_dict()
    for k in sd:
        if sd[k].is_floating_point() and not k.endswith("attn_mask"):
            sd[k] = P[k]
    m.load_state_dict(sd)
    m = m.cuda().eval()
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2)
    plan = m._plan()
    Pg = dict(m.named_parameters())
    Pg.update(dict(m.named_buffers()))
    mods = plan.mods
    # ---- oracle trace
    xs = []
    if 0 in mods:
        xs.append(OS.patch_embed(P, "patch_embed", v))
    if 1 in mods:
        xs.append(OS.patch_embed(P, "patch_embed_audio", a.unsqueeze(1)))
    Rm = B * T * plan.n_patches
    X = torch.empty((len(mods) * Rm, plan.embed_dim), dtype=torch.float32, device="cuda")
    for i, mm in enumerate(mods):
        pe = "patch_embed_audio" if mm else "patch_embed"
        ops.patch_embed_into((a.unsqueeze(1) if mm else v).cuda(), Pg[pe + ".proj.weight"], Pg[pe + ".proj.bias"],
                             Pg[pe + ".norm.weight"], Pg[pe + ".norm.bias"], X[i * Rm:(i + 1) * Rm])
    ref = torch.cat([x.reshape(-1, x.shape[-1]) for x in xs])
    print("patch_embed", rel(X, ref))
    xo = tuple(xs) if len(xs) == 2 else xs[0]
    block_mode = {"videoonly": "video_adapt", "audioonly": "audio_adapt", "multimodal": "multimodal_adapt_no_fusion",
                  "fusion": "fusion_adapt"}[mode]
    res = 56
    with torch.no_grad():
        for s, st in enumerate(plan.stages):
            H = W = res // (2 ** s)
            for i, (spec, pre) in enumerate(st["blocks"]):
                Pb = {n: Pg[pre + n] for n in st["names"][pre]}
                X, _ = ops.block_forward(X, spec, Pb, False, False)
                xo = OS.swin_block(P, pre[:-1], xo, H=H, W=W, T=T, heads=cfg["num_heads"][s], window_size=7,
                                   shift_size=0 if i % 2 == 0 else 3, t_attn=(i % 2 == 0), mode=block_mode)
                ref = torch.cat([x.reshape(-1, x.shape[-1]) for x in (xo if isinstance(xo, tuple) else (xo,))])
                print(pre, rel(X, ref))
            if st["merge"] is not None:
                Hh, Ww, pre = st["merge"]
                Pm = {n: Pg[pre + n] for n in ("norm.weight", "norm.bias", "reduction.weight")}
                X, _ = ops.merge_forward(X, Hh, Ww, Pm, False)
                if isinstance(xo, tuple):
                    xo = tuple(OS.patch_merging(P, pre[:-1], t, Hh, Ww) for t in xo)
                else:
                    xo = OS.patch_merging(P, pre[:-1], xo, Hh, Ww)
                ref = torch.cat([x.reshape(-1, x.shape[-1]) for x in (xo if isinstance(xo, tuple) else (xo,))])
                print(pre, rel(X, ref))
        logits = m(a.cuda(), v.cuda(), mode)
        print("logits", rel(logits, torch.as_tensor(z["logits"])))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "swin_tiny_videoonly")
