"""ctypes binding of libstgcma_hip.so (C ABI declared in include/stgcma.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  `lib()` raises if the shared object was not
built (run `python __graft_entry__.py` or `make -C stg-cma_amd/csrc`), and every wrapper raises RuntimeError with
stg_last_error() when a kernel entry point reports a failure.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STGCMA_LIB") or os.path.join(_HERE, "libstgcma_hip.so")   # env: an alternative build (kernel A/B runs)

STG_F32, STG_BF16, STG_FP8_MX, STG_U8_LIN = 0, 1, 2, 3
(GEMM_KERNEL_REG, GEMM_KERNEL_GLDS, GEMM_KERNEL_BIG, GEMM_KERNEL_8PH, GEMM_KERNEL_GLDS_CONV, GEMM_KERNEL_GLDS_BATCH, GEMM_KERNEL_GLDS_KTAIL,
 GEMM_KERNEL_FP8, GEMM_KERNEL_8PHM, GEMM_KERNEL_SKINNY) = range(10)
ACT_NONE, ACT_GELU, ACT_QUICKGELU = 0, 1, 2

c_i64 = C.c_int64
c_vp = C.c_void_p
c_fp = C.c_void_p  # float* passed as raw address


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", c_vp), ("lda", c_i64),
        ("W", c_vp), ("ldw", c_i64),
        ("C", c_vp), ("ldc", c_i64), ("c_dtype", C.c_int),
        ("bias", c_vp),
        ("alpha", C.c_float),
        ("act", C.c_int),
        ("dact", c_vp), ("ldp", c_i64),
        ("dact_src", c_vp), ("ldd", c_i64),
        ("row_scale", c_vp), ("rs_outer", c_i64), ("rs_inner", c_i64),
        ("res1", c_vp), ("ldr1", c_i64), ("res1_dtype", C.c_int),
        ("res2", c_vp), ("ldr2", c_i64), ("res2_dtype", C.c_int),
        ("M", c_i64), ("N", C.c_int), ("K", C.c_int),
        ("conv_H", C.c_int), ("conv_W", C.c_int), ("conv_d", C.c_int), ("conv_C", C.c_int),
        ("conv_zero", c_vp),
        ("batch", C.c_int), ("a_bstride", c_i64), ("w_bstride", c_i64), ("c_bstride", c_i64),
        ("ab_dtype", C.c_int), ("a_scale", c_vp), ("w_scale", c_vp),
        ("dact_dtype", C.c_int),
        ("kernel_chosen", C.c_int),
        ("split_m", c_i64), ("W2", c_vp), ("bias2", c_vp),
    ]


class WgradDesc(C.Structure):
    _fields_ = [("dY", c_vp), ("lddy", c_i64), ("X", c_vp), ("ldx", c_i64), ("dW", c_vp), ("lddw", c_i64), ("db", c_vp),
                ("M", c_i64), ("N1", C.c_int), ("N2", C.c_int), ("row_scale", c_vp), ("rs_outer", c_i64), ("rs_inner", c_i64)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", c_vp), ("ldq", c_i64),
        ("K", c_vp), ("ldk", c_i64),
        ("V", c_vp), ("ldv", c_i64),
        ("O", c_vp), ("ldo", c_i64),
        ("lse", c_vp),
        ("map_q", c_vp), ("map_kv", c_vp),
        ("outer_q", c_i64), ("outer_kv", c_i64),
        ("G", C.c_int),
        ("map_kind", C.c_int), ("map_a", C.c_int), ("map_b", C.c_int), ("map_c", C.c_int), ("map_d", C.c_int),
        ("P", c_i64), ("H", C.c_int), ("n", C.c_int), ("n_kv", C.c_int), ("D", C.c_int),
        ("scale", C.c_float),
        ("bias", c_vp), ("bias_div", c_i64), ("bias_mod", C.c_int),
        ("mask", c_vp),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("f", AttnArgs),
        ("dO", c_vp), ("lddo", c_i64),
        ("dQ", c_vp), ("lddq", c_i64),
        ("dK", c_vp), ("lddk", c_i64),
        ("dV", c_vp), ("lddv", c_i64),
        ("delta", c_vp),
        ("dbias", c_vp),
    ]


class XsmallArgs(C.Structure):
    _fields_ = [("Xv", c_vp), ("Xa", c_vp), ("ldv", c_i64), ("lda", c_i64), ("Ov", c_vp), ("Oa", c_vp), ("ldov", c_i64), ("ldoa", c_i64),
                ("lse_v", c_vp), ("lse_a", c_vp), ("P", C.c_int), ("nv", C.c_int), ("na", C.c_int), ("D", C.c_int), ("scale", C.c_float)]


class WinAttnArgs(C.Structure):
    _fields_ = [
        ("Q", c_vp), ("K", c_vp), ("V", c_vp), ("ld", c_i64),
        ("O", c_vp), ("ldo", c_i64),
        ("lse", c_vp),
        ("bm", c_vp), ("bmT", c_vp),
        ("Gt", C.c_int),
        ("outer", c_i64),
        ("Himg", C.c_int), ("Wimg", C.c_int), ("ws", C.c_int), ("shift", C.c_int),
        ("G", C.c_int), ("n", C.c_int),
        ("P", c_i64), ("H", C.c_int), ("D", C.c_int),
        ("scale", C.c_float),
    ]


class MhaArgs(C.Structure):
    _fields_ = [
        ("Q", c_vp), ("K", c_vp), ("V", c_vp), ("ld", c_i64),
        ("O", c_vp), ("ldo", c_i64),
        ("lse", c_vp),
        ("P", c_i64), ("H", C.c_int), ("n", C.c_int), ("D", C.c_int),
        ("scale", C.c_float),
        ("win_h", C.c_int), ("win_w", C.c_int), ("win_size", C.c_int), ("win_shift", C.c_int),
    ]


class CastDesc(C.Structure):
    _fields_ = [("in_", c_vp), ("off", c_i64), ("offT", c_i64), ("R", C.c_int), ("C", C.c_int), ("ld", C.c_int), ("ldT", C.c_int)]


class AdamDesc(C.Structure):
    _fields_ = [("p", c_vp), ("g", c_vp), ("m", c_vp), ("v", c_vp), ("st", c_vp), ("n", c_i64), ("group", C.c_int), ("pad_", C.c_int)]


class TAttnArgs(C.Structure):
    _fields_ = [
        ("Q", c_vp), ("K", c_vp), ("V", c_vp), ("ld", c_i64),
        ("O", c_vp), ("ldo", c_i64),
        ("bias", c_vp),
        ("bm", c_vp), ("bmT", c_vp),
        ("nm", C.c_int), ("B", c_i64), ("T", C.c_int), ("N", C.c_int), ("H", C.c_int), ("D", C.c_int),
        ("scale", C.c_float),
    ]


# name -> (restype, argtypes); every symbol include/stgcma.h declares
SIGNATURES = {
    "stg_version": (C.c_int, []),
    "stg_last_error": (C.c_char_p, []),
    "stg_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "stg_gemm_nt": (C.c_int, [C.POINTER(GemmArgs), c_vp]),
    "stg_quant_fp8_scale_bytes": (c_i64, [c_i64, C.c_int]),
    "stg_quant_fp8_mx": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, c_i64, c_vp, c_vp]),
    "stg_wgrad_tn": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, c_vp, c_i64, c_i64, c_vp]),
    "stg_wgrad_ws_floats": (c_i64, [c_i64, C.c_int, C.c_int]),
    "stg_wgrad_tn_ws": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, c_vp, c_i64, c_i64,
                                  c_vp, c_i64, c_vp]),
    "stg_mlp_fused_supported": (C.c_int, [C.c_int]),
    "stg_mlp_w2_perm": (C.c_int, [C.c_int, c_vp]),
    "stg_mlp_fwd": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, C.c_int, c_vp]),
    "stg_mlp_bwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, C.c_int, c_vp]),
    "stg_wgrad_tn_ws_multi": (C.c_int, [C.POINTER(WgradDesc), C.c_int, c_vp, c_i64, c_vp]),
    "stg_layernorm_fwd": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, c_vp, C.c_float, c_vp, C.c_int, c_i64, c_vp, c_vp,
                                    c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_layernorm_bwd": (C.c_int, [c_vp, c_i64, c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64,
                                    c_vp, c_i64, c_vp, c_vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_up_ln_supported": (C.c_int, [C.c_int, C.c_int]),
    "stg_up_ln_fwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64,
                                c_vp, c_vp, C.c_float, c_vp, c_i64, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_up_ln_fwd_pair": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64,
                                     c_vp, c_i64, c_vp, c_vp, C.c_float, c_vp, c_i64, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_ln_bwd_down_pair": (C.c_int, [C.c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp,
                                       c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_ln_bwd_down_supported": (C.c_int, [C.c_int, C.c_int]),
    "stg_ln_bwd_down": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                  c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_layernorm_bwd_xhat": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, C.c_int, c_vp]),
    "stg_ln_bwd_down_xhat": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                       c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_attn_fwd": (C.c_int, [C.POINTER(AttnArgs), c_vp]),
    "stg_attn_bwd": (C.c_int, [C.POINTER(AttnBwdArgs), c_vp]),
    "stg_attn_fwd2": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(AttnArgs), c_vp]),
    "stg_attn_bwd2": (C.c_int, [C.POINTER(AttnBwdArgs), C.POINTER(AttnBwdArgs), c_vp]),
    "stg_xattn_pair_bwd_ws_bytes": (c_i64, [c_i64, C.c_int, C.c_int, C.c_int]),
    "stg_xattn_pair_bwd_supported": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(AttnArgs)]),
    "stg_xattn_pair_bwd": (C.c_int, [C.POINTER(AttnBwdArgs), C.POINTER(AttnBwdArgs), c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "stg_xattn_pair_bwd_join": (C.c_int, [C.POINTER(AttnBwdArgs), C.POINTER(AttnBwdArgs), c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "stg_xattn_pair_bwd_gate": (C.c_int, [C.POINTER(AttnBwdArgs), C.POINTER(AttnBwdArgs), c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                          c_vp, c_i64, c_vp]),
    "stg_xattn_fwd2_gate": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(AttnArgs), c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_winattn_table": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_winattn_fwd": (C.c_int, [C.POINTER(WinAttnArgs), c_vp]),
    "stg_winattn_bwd": (C.c_int, [C.POINTER(WinAttnArgs), c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_winattn_pair_fwd": (C.c_int, [C.POINTER(WinAttnArgs), C.POINTER(WinAttnArgs), c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_winattn_pair_bwd": (C.c_int, [C.POINTER(WinAttnArgs), C.POINTER(WinAttnArgs), c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_tattn_fwd": (C.c_int, [C.POINTER(TAttnArgs), c_vp]),
    "stg_tattn_bwd": (C.c_int, [C.POINTER(TAttnArgs), c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "stg_mha_supported": (C.c_int, [C.c_int, C.c_int]),
    "stg_mha_fwd": (C.c_int, [C.POINTER(MhaArgs), c_vp]),
    "stg_mha_bwd": (C.c_int, [C.POINTER(MhaArgs), c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "stg_mha_fwd_pair": (C.c_int, [C.POINTER(MhaArgs), C.POINTER(MhaArgs), c_vp]),
    "stg_mha_bwd_pair": (C.c_int, [C.POINTER(MhaArgs), c_vp, c_vp, c_vp, c_vp, c_vp, C.POINTER(MhaArgs), c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "stg_xsmall_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "stg_xsmall_fwd": (C.c_int, [C.POINTER(XsmallArgs), c_vp]),
    "stg_xsmall_bwd": (C.c_int, [C.POINTER(XsmallArgs), c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "stg_mha_bwd_pair_merged": (C.c_int, [C.POINTER(MhaArgs), c_vp, c_vp, c_vp, C.POINTER(MhaArgs), c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "stg_gate_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_gate_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_im2col_patch": (C.c_int, [c_vp, C.c_int, c_vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_cast_bf16": (C.c_int, [c_vp, c_vp, c_i64, c_i64, C.c_int, c_i64, c_vp]),
    "stg_cast_bf16_multi": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, c_vp]),
    "stg_adam_multi": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, C.c_int, c_vp]),
    "stg_add_temporal": (C.c_int, [c_vp, c_vp, c_i64, C.c_int, c_i64, C.c_int, c_vp]),
    "stg_fbank": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp, C.c_int, C.c_float, C.c_float, C.c_float,
                            C.c_int, c_vp, c_vp]),
    "stg_video_aug": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, c_vp]),
    "stg_cast_f32": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "stg_meanpool_fwd": (C.c_int, [c_vp, c_vp, C.c_int, c_i64, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_meanpool_bwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_add": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_act_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_add3_mul": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_gate_fwd2": (C.c_int, [c_vp] * 8 + [c_i64, c_vp]),
    "stg_gate_bwd2": (C.c_int, [c_vp] * 10 + [c_i64, c_vp]),
    "stg_add3_mul2": (C.c_int, [c_vp] * 10 + [c_i64, c_vp]),
    "stg_gate_fwd2n": (C.c_int, [c_vp] * 4 + [c_i64] + [c_vp] * 4 + [c_i64, c_vp]),
    "stg_gate_bwd2n": (C.c_int, [c_vp] * 5 + [c_i64] + [c_vp] * 5 + [c_i64, c_vp]),
    "stg_add3_mul2n": (C.c_int, [c_vp] * 5 + [c_i64] + [c_vp] * 5 + [c_i64, c_vp]),
    "stg_debug_poison_lds": (C.c_int, [c_vp]),
    "stg_mul_mask": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_bias_gather": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_bias_scatter": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_unary_fwd": (C.c_int, [C.c_int, c_vp, c_vp, c_i64, c_vp]),
    "stg_unary_bwd": (C.c_int, [C.c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_mul": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "stg_embed_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_embed_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_lstm_cell_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, c_vp]),
    "stg_lstm_cell_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, c_vp]),
    "stg_grounding_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_grounding_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_mha1_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_vp]),
    "stg_mha1_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_vp]),
    "stg_im2col3x3": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_conv3x3_wgrad_ws_floats": (c_i64, [c_i64, C.c_int, C.c_int, c_vp]),
    "stg_wgrad_wide_ws_floats": (c_i64, [c_i64, C.c_int, C.c_int, c_vp]),
    "stg_wgrad_wide": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_sum_splits": (C.c_int, [c_vp, c_vp, C.c_int, c_i64, c_i64, C.c_int, c_vp]),
    "stg_wgrad_wide_batched": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_conv3x3_wgrad": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_bilinear_up2_fwd": (C.c_int, [c_vp, c_vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_bilinear_up2_bwd": (C.c_int, [c_vp, c_vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "stg_ln_param_grad": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, c_vp]),
    "stg_bn_colsum": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp]),
    "stg_bn_apply": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, c_vp]),
    "stg_bn_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, c_vp]),
    "stg_vit_embed": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, C.c_int, c_vp]),
}

ABI_VERSION = 220
_lib = None


class StgLibraryMissing(RuntimeError):
    pass


class _PoisonedLds:
    """Test aid (STG_LDS_POISON=1): every kernel launch through the C ABI is preceded by stg_debug_poison_lds on the same stream, so each kernel
    starts on LDS full of NaN patterns and a read of an LDS byte it never wrote becomes a non-finite result."""

    launches = 0

    def __init__(self, handle):
        self._h = handle
        self._poison = handle.stg_debug_poison_lds

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        sig = SIGNATURES.get(name)
        if (name in ("stg_debug_poison_lds", "stg_mlp_w2_perm", "stg_conv3x3_wgrad_ws_floats", "stg_wgrad_wide_ws_floats")      # last pointer is not a stream
                or sig is None or not sig[1] or sig[1][-1] is not c_vp or name.endswith("_supported")):
            return fn
        poison = self._poison

        def call(*a):
            rc = poison(a[-1])                 # the launch functions take the stream last
            if rc != 0:
                raise RuntimeError(f"stg_debug_poison_lds failed ({rc})")
            _PoisonedLds.launches += 1
            return fn(*a)
        return call


def lib():
    """Load (once) and return the ctypes handle; fail loudly when the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StgLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            f"(or `make -C stg-cma_amd/csrc`). There is no CPU fallback for the product path.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if handle.stg_version() != ABI_VERSION:
        raise RuntimeError(f"libstgcma_hip.so version {handle.stg_version()} != binding version {ABI_VERSION}")
    if os.environ.get("STG_LDS_POISON"):
        handle = _PoisonedLds(handle)
    _lib = handle
    # A/B knobs of tools/ (never set in production): forwarded ONCE from the environment to the library's explicit options
    from . import config                     # (config.py reads the environment once; configure(lib_<option>=...) lands here too)
    for opt, val in config.lib_options().items():
        check(handle.stg_set_option(opt.encode(), int(val)), f"stg_set_option({opt})")
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().stg_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
