"""BASELINE config 5 asks for an "fp8 MFMA weight path" under the north-star's <= 1e-2 max-abs logit bound.  Before building it:
what does e4m3 do to the logits of the reference's own arithmetic?  The fp32 oracle of the headline model (Swin-B + STG-CMA at the
reference's initialisation scale, the fixture behind the GPU tests' absolute logit bound) is run with the frozen backbone
Linears (qkv, proj, fc1, fc2, PatchMerging.reduction -- the 92.5 % of the FLOPs such a path would move to fp8) quantised the way
such a path would: weights e4m3 with one scale per output channel, and -- because an fp8 MFMA needs BOTH operands in fp8 -- the
activation rows entering those Linears e4m3 with one scale per row.  Everything else stays fp32, so the measured deviation is
the floor of the number format, not of any kernel.  DESIGN.md section 8 quotes the numbers this test prints."""
import pytest
import torch
import torch.nn.functional as F

import oracle.swin as OS
from golden_util import build_state, load_case

E4M3_MAX = 448.0
FROZEN = (".attn.qkv", ".attn.proj", ".mlp.fc1", ".mlp.fc2")


def _q_rows(t):
    """e4m3 round trip with one scale per row of the last dim (amax -> 448)."""
    s = t.abs().amax(-1, keepdim=True).clamp(min=1e-12) / E4M3_MAX
    return (t / s).to(torch.float8_e4m3fn).float() * s


@pytest.mark.timeout(600)
def test_e4m3_noise_floor_against_the_logit_bound(monkeypatch):
    from params import refinit_state, seeded_tensor
    z, cfg, shapes, names = load_case("swin_b_fusion_refinit")
    P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"], state_fn=refinit_state)
    B, T = cfg["B"], cfg["num_frames"]
    a = seeded_tensor((B, T, 224, 224), cfg["seed"] + 1, 0.5)
    v = seeded_tensor((B, 3, T, 224, 224), cfg["seed"] + 2)
    ref = torch.as_tensor(z["logits"])
    mode = {"w": False, "a": False}
    real_lin = OS._lin

    def lin(Pd, name, x):
        if name.endswith(FROZEN):
            w = _q_rows(Pd[name + ".weight"]) if mode["w"] else Pd[name + ".weight"]
            return F.linear(_q_rows(x) if mode["a"] else x, w, Pd.get(name + ".bias"))
        return real_lin(Pd, name, x)
    monkeypatch.setattr(OS, "_lin", lin)
    dev = {}
    with torch.no_grad():
        for tag, (mw, ma) in {"fp32": (False, False), "e4m3 weights": (True, False), "e4m3 weights + activations": (True, True)}.items():
            mode["w"], mode["a"] = mw, ma
            logits = OS.swin_forward(P, a, v, cfg, "fusion")
            dev[tag] = float((logits - ref).abs().max())
    scale = float(ref.abs().max())
    print(f"\nfp8 feasibility (Swin-B refinit, logit scale {scale:.3f}): " + ", ".join(f"{k}: {d:.3e}" for k, d in dev.items()))
    assert dev["fp32"] <= 1e-4                                     # the harness itself is exact
    # the bf16 HIP path sits at 2.3e-3 on this fixture (tests/test_model_gpu.py); the number format alone must leave room under 1e-2
    # measured: 2.8e-2 (weights only) and 2.5e-2 (both operands) against the 1e-2 bound, on a logit scale of 0.47.  The format's
    # floor alone is 2.5 x the bound, so an e4m3 path cannot be a drop-in for this model under the north-star's tolerance; if a
    # better quantisation scheme ever brings the floor under the bound this assertion fails and DESIGN.md section 8 is due a rewrite.
    assert dev["e4m3 weights"] > 1e-2 and dev["e4m3 weights + activations"] > 1e-2
