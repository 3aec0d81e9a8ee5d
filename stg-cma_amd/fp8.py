"""The fp8 frozen-weight path (BASELINE config 5: "Swin-L + STG-CMA, AVQA shape, fp8 MFMA weight path").

What it is.  The FROZEN Linears of the Swin backbone -- qkv, proj, fc1, fc2 of every block and the PatchMerging reductions (the
reference's AVQA/model/Swin_AVQAModel_V1.py:214-300 WindowAttention, :163-180 Mlp, :1029-1075 PatchMerging; frozen by the loop's name
filter, AVQA/traintest_adapt_avqa.py:72) -- run on v_mfma_scale_f32_16x16x128_f8f6f4: weights stored once as OCP e4m3 with one E8M0
scale per 32-wide k-block (the MX block format the instruction dequantises in hardware), activations quantised the same way on the
way into each GEMM, fp32 accumulation, the bf16 path's epilogues and outputs.  Adapters, gates, heads and everything that trains
stay bf16 / fp32.  `sites` selects WHICH frozen GEMMs take the path: tags '<site>.f' (forward) and '<site>.b' (data gradient) with
site in qkv / proj / fc1 / fc2 / merge, optionally restricted to some stages.

What it measures (MI355X; profiles/r02_fp8_parity_report.txt, profiles/r03_fp8_sites.txt).  e4m3 keeps 3 mantissa bits: one GEMM
with both operands in e4m3 deviates ~3.4 % (relative L2) from its bf16 result whatever K is (tests/test_fp8_gpu.py).  The reference
has no fp8 path, so the bound is BASELINE.json's <= 1e-2 max-abs logit deviation against the reference's fp32 logits on the
full-depth reference-initialised fixtures:
    every site, both directions:   Swin-L 4.7e-2, Swin-B 3.7e-2   (bf16 path: 4.5e-3 / 2.9e-3)  -> FAILS the bound by 4-5 x
    data-gradient GEMMs only ('*.b'): logits are the bf16 path's (the forward is untouched); per-tensor gradient norms within ~3 %
and it is SLOWER than bf16 as built: Swin-L AVE 100.3 vs 110.8 clips/s, AVQA 86.0 vs 94.4 (the e4m3 GEMM runs at the tuned bf16
kernel's rate and the step pays the stand-alone quantisation passes).  So the path is OPT-IN and no bench line produced with it is
the headline metric: `stgcma.fp8.enable(model)` or STG_FP8=1 in the environment before the model is built.
"""
import os

SITES = ("qkv", "proj", "fc1", "fc2", "merge")
ALL_TAGS = frozenset(s + d for s in SITES for d in (".f", ".b"))
BACKWARD_ONLY = frozenset(s + ".b" for s in SITES)


def _norm(sites):
    if sites is None or sites is True:
        return True
    tags = set()
    for s in sites:
        if s in SITES:
            tags |= {s + ".f", s + ".b"}
        elif s in ALL_TAGS:
            tags.add(s)
        else:
            raise ValueError(f"stgcma.fp8: unknown site '{s}' (sites: {SITES}, optionally suffixed .f / .b)")
    return frozenset(tags)


def enable(model, on=True, sites=None, stages=None):
    """Switch frozen backbone Linears of a Swin mirror (Swin_AVE / Swin_AVQAModel* / Swin_AVSModel*) to block-scaled e4m3.
    sites: None = every site in both directions, or an iterable of 'qkv' / 'proj' / 'fc1' / 'fc2' / 'merge' (both directions) and
    'qkv.f' / 'qkv.b' ... (one direction); stages: None = all, or the stage indices whose blocks (and downsample) take the path."""
    layers = getattr(model, "layers", None)
    if layers is None or not hasattr(model, "_plan"):
        raise TypeError("stgcma.fp8.enable: expects one of the Swin mirrors (SwinTransformer2D_Adapter_*)")
    sel = _norm(sites) if on else False
    plan = model._plan()
    merge_on = sel is True or (sel and any(t.startswith("merge") for t in sel))
    for si, layer in enumerate(layers):
        here = sel if (stages is None or si in stages) else False
        for blk in layer.blocks:
            blk._spec.fp8 = here
    plan.fp8 = sel if (merge_on and stages is None) else False
    model._fp8 = bool(on)
    return model


def enabled(model):
    return bool(getattr(model, "_fp8", False))


def env_default():
    from . import config
    return bool(config.opt("fp8"))
