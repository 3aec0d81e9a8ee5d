"""The fp8 frozen-weight path (BASELINE config 5: "Swin-L + STG-CMA, AVQA shape, fp8 MFMA weight path").

What it is.  The FROZEN Linears of the Swin backbone -- qkv, proj, fc1, fc2 of every block and the PatchMerging reductions (the
reference's AVQA/model/Swin_AVQAModel_V1.py:214-300 WindowAttention, :163-180 Mlp, :1029-1075 PatchMerging; frozen by the loop's name
filter, AVQA/traintest_adapt_avqa.py:72) -- run on v_mfma_scale_f32_16x16x128_f8f6f4: weights stored once as OCP e4m3 with one E8M0
scale per 32-wide k-block (the MX block format the instruction dequantises in hardware), activations quantised the same way on the
way into each GEMM, fp32 accumulation, the bf16 path's epilogues and outputs.  Forward and the data-gradient GEMMs both; adapters,
gates, heads and everything that trains stay bf16 / fp32.

What it costs.  e4m3 keeps 3 mantissa bits: one GEMM deviates ~3.4 % (relative L2) from its bf16 result (tests/test_fp8_gpu.py).
The reference has no fp8 path, so the bound is BASELINE.json's <= 1e-2 max-abs logit deviation, measured on the full-depth Swin-L
fixture at the reference's initialisation scale (tests/test_fp8_model_gpu.py; DESIGN.md section 8 holds the numbers).  It is OPT-IN:
`stgcma.fp8.enable(model)` or STG_FP8=1 in the environment before the model is built.
"""
import os


def enable(model, on=True):
    """Switch the frozen backbone Linears of a Swin mirror (Swin_AVE / Swin_AVQAModel* / Swin_AVSModel*) to block-scaled e4m3."""
    layers = getattr(model, "layers", None)
    if layers is None or not hasattr(model, "_plan"):
        raise TypeError("stgcma.fp8.enable: expects one of the Swin mirrors (SwinTransformer2D_Adapter_*)")
    for layer in layers:
        for blk in layer.blocks:
            blk._spec.fp8 = bool(on)
    model._plan().fp8 = bool(on)
    model._fp8 = bool(on)
    return model


def enabled(model):
    return bool(getattr(model, "_fp8", False))


def env_default():
    return os.environ.get("STG_FP8", "0") not in ("", "0")
