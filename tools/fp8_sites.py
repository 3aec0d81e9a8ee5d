"""Per-site sensitivity of the fp8 (block-scaled e4m3) frozen-weight path on the full-depth reference-initialised fixtures
(VERDICT r2 item 4): max-abs logit deviation from the REFERENCE's fp32 logits (tests/golden/<case>.npz) with only some of the frozen
GEMMs on e4m3.  GPU box:  python tools/fp8_sites.py [case ...]  -> gpurun_out/fp8_sites.txt (copied to profiles/r03_fp8_sites.txt).
Bound to meet: BASELINE.json's 1e-2."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import stgcma  # noqa: F401,E402
from stgcma import fp8, recipe  # noqa: E402
from stgcma.model import Swin_AVE as S  # noqa: E402
from golden_util import build_state, load_case  # noqa: E402
from params import refinit_state, seeded_tensor  # noqa: E402

CASES = [("bf16", None, None), ("all sites, fwd + bwd", True, None), ("all sites, bwd only (*.b)", fp8.BACKWARD_ONLY, None),
         ("all sites, fwd only (*.f)", [s + ".f" for s in fp8.SITES], None)]
CASES += [(f"only {s}.f", [s + ".f"], None) for s in fp8.SITES]
CASES += [(f"all sites fwd, stage {i} only", [s + ".f" for s in fp8.SITES], [i]) for i in range(4)]
CASES += [("fc1.f + fc2.f, stage 3 only", ["fc1.f", "fc2.f"], [3]), ("fc1.f, stage 2 only", ["fc1.f"], [2]), ("qkv.f, stage 2 only", ["qkv.f"], [2])]


def main(cases):
    gpu = torch.device("cuda", 0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out = open(os.path.join(ROOT, "gpurun_out", "fp8_sites.txt"), "w")

    def say(line):
        print(line, flush=True)
        out.write(line + "\n")
        out.flush()

    for case in cases:
        z, cfg, shapes, names = load_case(case)
        P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"], state_fn=refinit_state)
        m = S.SwinTransformer2D_Adapter_New(label_dim=cfg["label_dim"], patch_size=[1, 4, 4], num_frames=cfg["num_frames"], embed_dim=cfg["embed_dim"],
                                            depths=cfg["depths"], num_heads=cfg["num_heads"], window_size=7, pretrained=None, ftmode=cfg["mode"],
                                            adapter_mlp_ratio=cfg["adapter_mlp_ratio"]).eval()
        sd = m.state_dict()
        for k in sd:
            if sd[k].is_floating_point() and not k.endswith("attn_mask"):
                sd[k] = P[k]
        m.load_state_dict(sd, strict=True)
        del P
        m = m.to(gpu)
        recipe.apply_freeze(m)
        B, T = cfg["B"], cfg["num_frames"]
        a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5).to(gpu)
        v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2).to(gpu)
        tgt = torch.softmax(seeded_tensor((B * T, cfg["label_dim"]), cfg["seed"] + 3, 2.0), -1).to(gpu)
        ref = torch.as_tensor(z["logits"])
        ref_norms = np.asarray(z["grad_norms"])
        live = ref_norms > 1e-3 * ref_norms.max()
        say(f"# {case}: depths {cfg['depths']}, embed_dim {cfg['embed_dim']}, B {B}, max |reference logit| {float(ref.abs().max()):.3g}; bound 1e-2 max-abs")
        say(f"{'frozen GEMMs on e4m3':44s} {'max-abs logit dev':>18s} {'<= 1e-2':>8s} {'worst grad-norm dev':>20s}")
        for tag, sites, stages in CASES:
            fp8.enable(m, sites is not None, sites=sites, stages=stages)
            m.zero_grad(set_to_none=True)
            logits = m(a, v, "fusion")
            torch.nn.CrossEntropyLoss()(logits, tgt).backward()
            err = float((logits.detach().cpu() - ref).abs().max())
            d = dict(m.named_parameters())
            norms = np.array([float(d[n].grad.float().norm()) for n in names])
            gdev = float(np.abs(norms[live] / ref_norms[live] - 1).max())
            say(f"{tag:44s} {err:18.3e} {'yes' if err <= 1e-2 else 'NO':>8s} {gdev:20.3e}")
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(sys.argv[1:] or ["swin_l_fusion_refinit"])
