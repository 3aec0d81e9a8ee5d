"""Diagnostic (test infrastructure; runs the CPU oracle beside the HIP path): where does the whole-model gradient deviation of the
eval-BatchNorm AVS fixture come from?  Splits the model at the backbone / decoder boundary (the four taps + the audio feature):
  1. taps: HIP vs oracle (forward)
  2. d(taps) of the HIP decoder fed with the ORACLE's taps vs the oracle's d(taps)   -> the decoder alone
  3. d(taps) of the HIP model end to end vs the oracle's                              -> decoder on the HIP backbone's taps
  4. backbone gradients of the HIP backbone driven by the ORACLE's d(taps)            -> the backbone alone
Usage: python tools/avs_evalbn_diag.py   (on the GPU box)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import stgcma                                                  # noqa: E402  (tests/conftest-style alias of stg-cma_amd)
from golden_util import build_state, load_case                 # noqa: E402
from params import seeded_tensor                               # noqa: E402
import oracle.avs_decoder as OD                                # noqa: E402
from oracle.swin import swin_backbone                          # noqa: E402


def rel(a, b):
    a = a.detach().float().cpu().reshape(-1); b = b.detach().float().cpu().reshape(-1)
    return float((a - b).norm() / b.norm()), float(a.norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))


def main():
    from stgcma import recipe
    from stgcma.model import Swin_AVSModel_Base
    from stgcma.ops_dec import avs_decoder_forward
    gpu = torch.device("cuda:0")
    z, cfg, shapes, names = load_case("avs_full_tiny_evalbn")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
    for k, _ in shapes:
        if "W_z.1.weight" in k:
            P[k] = P[k] * 0.1
    for i, k in enumerate(json.loads(str(z["stat_names_json"]))):
        P[k] = torch.as_tensor(np.asarray(z[f"stat{i}"]))
    m = Swin_AVSModel_Base.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                                              num_heads=cfg["num_heads"], ftmode="fusion",
                                                              adapter_mlp_ratio=cfg["adapter_mlp_ratio"]).eval()
    sd = m.state_dict()
    for k in sd:
        if sd[k].is_floating_point() and not k.endswith("attn_mask"):
            sd[k] = P[k]
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu)
    recipe.apply_freeze(m)
    B, seed = cfg["B"], cfg["seed"]
    a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5)
    v = seeded_tensor((B, 5, 3, 224, 224), seed + 2)
    w = seeded_tensor(tuple(np.asarray(z["pred"]).shape), seed + 3, 1e-2)

    # ---- oracle, split at the taps
    Po = {k: t.clone() for k, t in P.items()}
    for n in names:
        Po[n].requires_grad_(True)
    torch.set_num_threads(16)
    out = swin_backbone(Po, a, v, cfg)
    taps_o = [t.detach().clone().requires_grad_(True) for t in out["taps"]]
    fa_o = out["f_a"].detach().clone().requires_grad_(True)
    pred_o, _, _ = OD.avs_decoder(Po, taps_o, fa_o, B, 5, bn_training=False)
    (pred_o * w).sum().backward()
    dtaps_o = [t.grad.clone() for t in taps_o] + [fa_o.grad.clone()]
    dec_grads_o = {n: Po[n].grad.clone() for n in names if Po[n].grad is not None}
    for n in names:
        Po[n].grad = None
    torch.autograd.backward(out["taps"] + [out["f_a"]], dtaps_o)
    bb_grads_o = {n: Po[n].grad.clone() for n in names if Po[n].grad is not None and not n.startswith("avstask_")}
    print("oracle: pred vs golden relL2 %.2e" % rel(pred_o, torch.as_tensor(z["pred"]))[0], flush=True)

    # ---- 1. HIP taps
    ms, a_feat = m.forward_features(a.to(gpu), v.to(gpu))
    for i, (t, to) in enumerate(zip(list(ms) + [a_feat], [t.detach() for t in out["taps"]] + [out["f_a"].detach()])):
        print("1. tap %d forward: relL2 %.3e (norm ratio %.4f, cos %.5f)" % ((i,) + rel(t, to.reshape(t.shape))))

    # ---- 2. HIP decoder on the oracle's taps
    ms2 = [t.detach().reshape(m_.shape).to(gpu).requires_grad_(True) for t, m_ in zip(out["taps"], ms)]
    fa2 = out["f_a"].detach().reshape(a_feat.shape).to(gpu).requires_grad_(True)
    for p_ in m.parameters():
        p_.grad = None
    pred2, _, _ = avs_decoder_forward(m, ms2, fa2, B, 5, False)
    print("2. decoder on oracle taps: pred relL2 %.3e" % rel(pred2, pred_o)[0])
    (pred2 * w.to(gpu)).sum().backward()
    for i, (t, to) in enumerate(zip(ms2 + [fa2], dtaps_o)):
        print("2. d(tap %d): relL2 %.3e (norm ratio %.4f, cos %.5f)" % ((i,) + rel(t.grad, to.reshape(t.shape))))
    d = dict(m.named_parameters())
    worst = sorted(((rel(d[n].grad, g)[0], rel(d[n].grad, g)[1], n) for n, g in dec_grads_o.items() if d[n].grad is not None and float(g.norm()) > 0), reverse=True)
    print("2. decoder parameter gradients: median relL2 %.3e; worst:" % float(np.median([x[0] for x in worst])))
    for x in worst[:8]:
        print("      relL2 %.3e norm ratio %.4f %s" % x)

    # ---- 3. end to end on the HIP backbone's taps
    for p_ in m.parameters():
        p_.grad = None
    ms3 = [t.detach().requires_grad_(True) for t in ms]
    fa3 = a_feat.detach().requires_grad_(True)
    pred3, _, _ = avs_decoder_forward(m, ms3, fa3, B, 5, False)
    print("3. decoder on HIP taps: pred relL2 %.3e" % rel(pred3, pred_o)[0])
    (pred3 * w.to(gpu)).sum().backward()
    for i, (t, to) in enumerate(zip(ms3 + [fa3], dtaps_o)):
        print("3. d(tap %d): relL2 %.3e (norm ratio %.4f, cos %.5f)" % ((i,) + rel(t.grad, to.reshape(t.shape))))

    # ---- 4. the HIP backbone driven by the oracle's d(taps)
    for p_ in m.parameters():
        p_.grad = None
    torch.autograd.backward(list(ms) + [a_feat], [g.reshape(t.shape).to(gpu) for g, t in zip(dtaps_o, list(ms) + [a_feat])])
    rows = sorted(((rel(d[n].grad, g)[0], rel(d[n].grad, g)[1], n) for n, g in bb_grads_o.items() if d[n].grad is not None and float(g.norm()) > 1e-12), reverse=True)
    print("4. backbone gradients from the oracle's d(taps): median relL2 %.3e, median norm ratio %.4f; worst:" %
          (float(np.median([x[0] for x in rows])), float(np.median([x[1] for x in rows]))))
    for x in rows[:8]:
        print("      relL2 %.3e norm ratio %.4f %s" % x)
    by_layer = {}
    for r_, nr, n in rows:
        by_layer.setdefault(n.split(".blocks.")[0] if ".blocks." in n else n.split(".")[0], []).append((r_, nr))
    for k in sorted(by_layer):
        print("      %-12s n=%3d median relL2 %.3e median norm ratio %.4f" % (k, len(by_layer[k]), float(np.median([x[0] for x in by_layer[k]])),
                                                                             float(np.median([x[1] for x in by_layer[k]]))))


if __name__ == "__main__":
    main()
