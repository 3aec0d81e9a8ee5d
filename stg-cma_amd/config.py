"""Every switch of the Python side in ONE place: `stgcma.configure(name=value, ...)` sets them, `stgcma.options()`
lists them, and the environment is read exactly once -- when the package is imported -- through the same table (STG_* variables, the
tools/ A/B scripts' interface).  All of them are A/B knobs: the defaults are the product path, nothing here selects a fallback that
leaves the HIP library.  The C side has its own explicit options (`stg_set_option`, include/stgcma.h); `configure(lib_<option>=int)`
forwards to it.

Modules keep the value as a module attribute (`ops.USE_UPLN`, ...), which the hot path reads as a plain global; configure() rewrites
the attribute on every module that holds a copy.  Change options BEFORE building a model: cached shadows / plans are not rebuilt."""
import os
import sys

import torch

_PKG = __name__.rsplit(".", 1)[0]


def _b(raw):
    return str(raw).strip().lower() not in ("0", "", "false", "off", "no")


def _dact(raw):                     # saved GELU derivative of the MLP hidden: one byte per element, or bf16
    return "u8" if str(raw).lower() in ("u8", "1", "true") else True


def _res(raw):                      # residual-stream dtype: fp32 (the reference's autocast loop keeps it in fp32) or bf16 (A/B only)
    if isinstance(raw, torch.dtype):
        return raw
    return torch.bfloat16 if str(raw).lower() == "bf16" else torch.float32


# name: (environment variable, parser, default, [(module, attribute)], what it does)
SPEC = {
    "upln":        ("STG_UPLN", _b, True, [("ops", "USE_UPLN")], "adapter up-projection + residual join + LayerNorm in one kernel (csrc/upln.hip)"),
    "mlp_dact":    ("STG_MLP_DACT", _dact, "u8", [("ops", "MLP_DACT")], "storage of the MLP hidden's saved GELU derivative: 'u8' (one byte) or 'bf16'"),
    "residual":    ("STG_RESIDUAL", _res, torch.float32, [("ops", "RESIDUAL_DTYPE"), ("ops_vit", "RESIDUAL_DTYPE")],
                    "residual-stream dtype; 'bf16' is an A/B knob, never the headline (the reference keeps fp32)"),
    "xhat":        ("STG_XHAT", _b, True, [("ops", "USE_XHAT")], "LayerNorm in x-hat form (frozen affines folded into the frozen GEMM weights)"),
    "mlp_fused":   ("STG_MLP_FUSED", _b, True, [("ops", "MLP_FUSED")], "one-kernel fc1 -> GELU -> fc2 where built (C = 128)"),
    "winattn":     ("STG_WINATTN", _b, True, [("ops", "USE_WINATTN")], "dedicated window-attention kernels (0: generic attention kernels)"),
    "tattn":       ("STG_TATTN", _b, True, [("ops", "USE_TATTN")], "dedicated temporal-attention kernels"),
    "mha_x":       ("STG_MHA_X", _b, True, [("ops", "USE_MHA_X")], "flash kernels for wide frame-global cross-modal attention"),
    "xwin":        ("STG_XWIN", _b, True, [("ops", "USE_XWIN")], "window-level cross-modal attention on the whole-window kernels"),
    "mha_pair":    ("STG_MHA_PAIR", _b, True, [("ops", "MHA_PAIR")], "both directions of a cross-modal pair per mha launch"),
    "xsmall":      ("STG_XSMALL", _b, True, [("ops", "XSMALL")], "ViT cross-modal pair on small frames: one workgroup per frame, forward / merged backward in one launch each (xsmall.hip)"),
    "mha_merged":  ("STG_MHA_MERGED", _b, True, [("ops", "MHA_MERGED")], "wide frame-global cross-modal pair: backward as one merged pass per modality (mha.hip)"),
    "mha_win":     ("STG_MHA_WIN", _b, True, [("ops", "USE_MHA_WIN")], "flash kernels (window map) for wide window-level cross-modal attention"),
    "xattn_merged": ("STG_XATTN_MERGED", _b, True, [("ops", "XATTN_MERGED")], "frame-global cross-modal pair: backward as one merged pass per modality (one exponential per score)"),
    "pair_ew":     ("STG_PAIR_EW", _b, True, [("ops", "PAIR_EW")], "both directions of a cross-modal pair per element-wise launch"),
    "xattn_gate":  ("STG_XATTN_GATE", _b, True, [("ops", "XATTN_GATE")], "the frame-global cross-modal pair's gates inside its merged backward (0 = gate kernel first)"),
    "xwin_pair":   ("STG_XWIN_PAIR", _b, True, [("ops", "XWIN_PAIR")], "the window-level cross-modal pair as one launch per pass with its gates (0 = two launches + gate kernels)"),
    "join_pair":   ("STG_JOIN_PAIR", _b, True, [("ops", "JOIN_PAIR")], "both modalities' residual joins (+ LayerNorm) and LayerNorm-backward + adapter dgrad in one launch each"),
    "gemm_split":  ("STG_GEMM_SPLIT", _b, True, [("ops", "GEMM_SPLIT")], "video | audio adapter GEMMs as one launch with two row groups"),
    "wgrad_ws":    ("STG_WGRAD_WS", _b, True, [("kernels", "USE_WGRAD_WS")], "workspace (atomic-free) weight-gradient kernels"),
    "wgrad_multi": ("STG_WGRAD_MULTI", _b, True, [("kernels", "USE_WGRAD_MULTI")], "adapter weight gradients of a block in one launch pair"),
    "mha":         ("STG_MHA", _b, True, [("ops_vit", "USE_MHA")], "ViT spatial attention on the flash kernels"),
    "tattn_vit":   ("STG_TATTN_VIT", _b, True, [("ops_vit", "USE_TATTN_VIT")], "ViT temporal attention on the packed kernels"),
    "fp8":         ("STG_FP8", _b, False, [], "frozen backbone Linears on block-scaled e4m3 (opt-in; misses the 1e-2 logit bound, DESIGN section 8)"),
}
# options of the C library (stg_set_option): environment variable -> option name
LIB_SPEC = {"STG_GEMM_8PH": "gemm_8ph", "STG_GEMM_8PHM": "gemm_8phm", "STG_GEMM_NX": "gemm_nx", "STG_GEMM_D8M": "gemm_d8m", "STG_GEMM_DBG": "gemm_dbg",
            "STG_XATTN": "xattn", "STG_WGRAD_PLAN": "wgrad_plan", "STG_UPLN_CAP": "upln_cap"}

_values = {k: v[2] for k, v in SPEC.items()}
_lib_values = {}
_from_env = []
for _k, (_env, _parse, _d, _t, _doc) in SPEC.items():          # the ONE read of the environment
    if _env in os.environ:
        _values[_k] = _parse(os.environ[_env])
        _from_env.append(_env)
for _env, _o in LIB_SPEC.items():
    if _env in os.environ:
        _lib_values[_o] = int(os.environ[_env])
        _from_env.append(_env)


def opt(name):
    return _values[name]


def lib_options():
    """C-library options taken from the environment / configure(lib_...): applied by _lib.lib() when the library is loaded."""
    return dict(_lib_values)


def configure(**kw):
    """Set options by name (see SPEC; `lib_<option>=int` for the C library's).  Returns the resulting option table."""
    for k, v in kw.items():
        if k.startswith("lib_"):
            _lib_values[k[4:]] = int(v)
            from . import _lib
            if _lib._lib is not None:
                _lib.check(_lib._lib.stg_set_option(k[4:].encode(), int(v)), f"stg_set_option({k[4:]})")
            continue
        if k not in SPEC:
            raise KeyError(f"stgcma.configure: unknown option {k!r}; known: {sorted(SPEC)} and lib_<option>")
        _values[k] = SPEC[k][1](v)
        for mod, attr in SPEC[k][3]:
            m = sys.modules.get(f"{_PKG}.{mod}")
            if m is not None:
                setattr(m, attr, _values[k])
    return options()


def options():
    """{name: value} of every option (dtypes / flags as strings), plus which came from the environment and the C library's overrides."""
    o = {k: (str(v).replace("torch.", "") if not isinstance(v, (bool, int, str)) else v) for k, v in _values.items()}
    if _lib_values:
        o["lib"] = dict(_lib_values)
    if _from_env:
        o["from_env"] = list(_from_env)
    return o
