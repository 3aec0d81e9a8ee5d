"""Raw (non-autograd) wrappers: torch tensors in, one libstgcma_hip.so launch each, on torch's current HIP stream.

Every wrapper validates shapes/dtypes/devices on the host before the launch -- a kernel must never see an operand whose
extent differs from what its grid assumes.  PyTorch supplies device memory and the stream only.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU, STG_BF16, STG_F32  # noqa: F401

BF16 = torch.bfloat16
F32 = torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_ptr_dev = -1      # device index of the last tensor whose pointer was taken (_p): every wrapper takes its pointers before _stream()


def _stream():
    """hipStream_t of torch's current stream on the device of the launch's tensors (the raw getter is ~20x cheaper than building a
    torch.cuda.Stream object per launch: 1 500 launches per step).  A HIP launch goes to the CURRENT device, so the tensors must
    live there: the module entry points enter torch.cuda.device(input.device) (nn.DataParallel replicas and autograd's backward
    threads do the same), and a call that reaches a kernel from another device fails here instead of launching on the wrong
    device's stream with foreign pointers."""
    dev = torch.cuda.current_device()
    if _ptr_dev >= 0 and _ptr_dev != dev:
        raise RuntimeError(f"stg-cma_amd: operands live on cuda:{_ptr_dev} but the current device is cuda:{dev}; wrap the call in "
                           f"torch.cuda.device(tensor.device) (the model / block modules do)")
    if _raw_stream is not None:
        return _raw_stream(dev)
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    global _ptr_dev
    if t is None:
        return None
    _ptr_dev = t.device.index if t.is_cuda else -1
    return t.data_ptr()


def _chk2d(t, name, dtype, cols=None, rows=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"{name}: expected a 2-D row-major tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    if cols is not None and t.shape[1] != cols:
        raise RuntimeError(f"{name}: expected {cols} columns, got {t.shape[1]}")
    if rows is not None and t.shape[0] != rows:
        raise RuntimeError(f"{name}: expected {rows} rows, got {t.shape[0]}")
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        raise RuntimeError(f"{name}: overlapping rows")


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def _dt(t):
    if t.dtype == BF16:
        return STG_BF16
    if t.dtype == F32:
        return STG_F32
    raise RuntimeError(f"unsupported dtype {t.dtype} (bf16 / fp32 only)")


def _chk1d(t, name, dtype, n):
    if not t.is_cuda or t.dtype != dtype or t.dim() != 1 or t.shape[0] != n or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected contiguous {dtype} GPU vector of length {n}")


_zero_lines = {}


def _zero_line(device):
    z = _zero_lines.get(device)
    if z is None:
        z = _zero_lines[device] = torch.zeros(64, dtype=BF16, device=device)
    return z


# ------------------------------------------------------------------------------------------------ per-family accounting (bench.py roofline_families)
# The non-GEMM half of the step, accounted like the GEMM classes (gemm_profile_*): a wrapper decorated with @_family(name, cost) reports, per
# launch, its ALGORITHMIC bytes (every operand / output once, from the shapes the wrapper already checks) and FLOPs; every stride-th launch of
# a (family, key) class is bracketed by HIP events on the launch stream.  Off (one global test per call) unless bench.py turns it on.
import functools as _functools

_fam_prof = None


def family_profile_start(stride=5):
    global _fam_prof
    _fam_prof = {"stride": int(stride), "classes": {}}


def family_profile_reset():
    if _fam_prof is not None:
        _fam_prof["classes"] = {}


def family_profile_stop():
    """[{family, key, launches, bytes, flops, sampled, sampled_ms, sampled_bytes, sampled_flops}] since the last reset."""
    global _fam_prof
    prof, _fam_prof = _fam_prof, None
    if prof is None:
        return []
    torch.cuda.synchronize()
    out = []
    for (name, key), pc in prof["classes"].items():
        rec = pc["rec"]
        out.append({"family": name, "key": key, "launches": pc["launches"], "bytes": pc["bytes"], "flops": pc["flops"], "sampled": len(rec),
                    "sampled_ms": sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec), "sampled_bytes": sum(r[2] for r in rec),
                    "sampled_flops": sum(r[3] for r in rec)})
    return out


def _family(name, cost):
    """cost(*args, **kwargs) -> (key, algorithmic bytes, FLOPs) of the call (computed from shapes only, before the launch)."""
    def deco(fn):
        @_functools.wraps(fn)
        def wrapped(*a, **kw):
            prof = _fam_prof
            if prof is None:
                return fn(*a, **kw)
            key, nbytes, flops = cost(*a, **kw)
            ck = (name, key)
            pc = prof["classes"].get(ck)
            if pc is None:
                pc = prof["classes"][ck] = {"launches": 0, "bytes": 0.0, "flops": 0.0, "rec": []}
            pc["launches"] += 1
            pc["bytes"] += nbytes
            pc["flops"] += flops
            if pc["launches"] % prof["stride"]:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            pc["rec"].append((e0, e1, nbytes, flops))
            return r
        return wrapped
    return deco


def _nb(*ts):
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


def _family_io(name, flops=None):
    """_family for the kernels whose algorithmic bytes are simply every tensor argument once + every returned tensor once (the AVS decoder's
    im2col / bilinear / BatchNorm / TPAVI kernels, casts, the AVQA head's small kernels): key = shape of the first tensor argument;
    flops(*args, **kwargs) optional.  The cost is computed AFTER the call (the outputs' sizes are part of it)."""
    def deco(fn):
        @_functools.wraps(fn)
        def wrapped(*a, **kw):
            prof = _fam_prof
            if prof is None:
                return fn(*a, **kw)
            ts = [t for t in list(a) + list(kw.values()) if isinstance(t, torch.Tensor)]
            ck = (name, tuple(ts[0].shape) if ts else ())
            pc = prof["classes"].get(ck)
            if pc is None:
                pc = prof["classes"][ck] = {"launches": 0, "bytes": 0.0, "flops": 0.0, "rec": []}
            pc["launches"] += 1
            sampled = pc["launches"] % prof["stride"] == 0
            if sampled:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            r = fn(*a, **kw)
            if sampled:
                e1.record()
            outs = [t for t in (r if isinstance(r, (tuple, list)) else (r,)) if isinstance(t, torch.Tensor)]
            seen, nbytes = set(), 0.0
            for t in ts + outs:
                if t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    nbytes += t.numel() * t.element_size()
            fl = float(flops(*a, **kw)) if flops is not None else 0.0
            pc["bytes"] += nbytes
            pc["flops"] += fl
            if sampled:
                pc["rec"].append((e0, e1, nbytes, fl))
            return r
        return wrapped
    return deco


def conv3x3_gemm_supported(Cin):
    return Cin % 64 == 0 or Cin in (8, 16, 32)


class Fp8:
    """A block-scaled e4m3 matrix (stg_quant_fp8_mx): q uint8 [rows, Kp] (Kp = K rounded up to 128, pad bytes zero), s uint8 packed
    E8M0 scale table, logical shape (rows, K)."""
    __slots__ = ("q", "s", "rows", "K")

    def __init__(self, q, s, rows, K):
        self.q, self.s, self.rows, self.K = q, s, rows, K

    @property
    def shape(self):
        return (self.rows, self.K)

    @property
    def device(self):
        return self.q.device


def quant_fp8(x):
    """bf16 [rows, K] (K % 8 == 0, row-major, 16-byte aligned rows) -> Fp8: per 32-wide k-block of a row one E8M0 exponent
    e = ceil(log2(max|x| / 448)) + 127 and e4m3 round-to-nearest-even of x * 2^(127 - e)."""
    _chk2d(x, "x", BF16)
    rows, Kd = x.shape
    if Kd % 8 != 0 or _ld(x) % 8 != 0:
        raise RuntimeError("quant_fp8: K and the leading dimension must be multiples of 8")
    Kp = (Kd + 127) // 128 * 128
    q = torch.empty((rows, Kp), dtype=torch.uint8, device=x.device)
    nb = int(_lib.lib().stg_quant_fp8_scale_bytes(rows, Kd))
    sc = torch.empty((max(nb, 4),), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().stg_quant_fp8_mx(_p(x), _ld(x), rows, Kd, _p(q), Kp, _p(sc), _stream()), "stg_quant_fp8_mx")
    return Fp8(q, sc, rows, Kd)


def gemm_nt(A, W, bias=None, *, out=None, out_dtype=BF16, alpha=1.0, act=ACT_NONE, want_dact=False, dact_src=None,
            row_scale=None, rs_outer=1, rs_inner=1, res1=None, res2=None, conv=None, batch=None, split=None):
    """C = epi(A @ W.T); see stg_gemm_nt in include/stgcma.h.  Returns C or (C, dact) with dact = bf16(act'(pre-activation)),
    the tensor a later call takes as dact_src (or act_bwd as its second argument).  A and W are bf16 tensors, or both Fp8
    (quant_fp8): the block-scaled e4m3 MFMA path."""
    if batch is not None:
        return _gemm_nt_batched(A, W, bias, out, out_dtype, alpha, act, int(batch))
    fp8 = isinstance(A, Fp8)
    if fp8 != isinstance(W, Fp8):
        raise RuntimeError("gemm_nt: A and W must both be bf16 tensors or both Fp8")
    M, K = A.shape
    N = W.shape[0]
    dev = A.device
    if fp8:
        if W.K != K or conv is not None:
            raise RuntimeError("gemm_nt(fp8): K mismatch / no implicit convolution on fp8 operands")
    else:
        _chk2d(A, "A", BF16)
    if conv is not None:                                  # implicit 3x3 convolution: A is the [F*H*W, Cin] feature map, K = 9 * Cin
        Hc, Wc, dc = conv
        Cin = K
        K = 9 * Cin
        if not conv3x3_gemm_supported(Cin) or M % (Hc * Wc) != 0 or dc < 1:
            raise RuntimeError("gemm_nt(conv=...): needs Cin % 64 == 0 (or 8 / 16 / 32) and rows = F * H * W")
    if not fp8:
        _chk2d(W, "W", BF16, cols=K)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=dev)
    _chk2d(out, "out", out.dtype, cols=N, rows=M)
    if out.dtype not in (BF16, F32):
        raise RuntimeError("out dtype must be bf16 or fp32")
    a = _lib.GemmArgs()
    if fp8:
        a.A, a.lda, a.a_scale = _p(A.q), A.q.shape[1], _p(A.s)
        a.W, a.ldw, a.w_scale = _p(W.q), W.q.shape[1], _p(W.s)
        a.ab_dtype = _lib.STG_FP8_MX
    else:
        a.A, a.lda = _p(A), _ld(A)
        a.W, a.ldw = _p(W), _ld(W)
    a.C, a.ldc, a.c_dtype = _p(out), _ld(out), (STG_BF16 if out.dtype == BF16 else STG_F32)
    if bias is not None:
        _chk1d(bias, "bias", F32, N)
    a.bias = _p(bias)
    a.alpha = float(alpha)
    a.act = int(act)
    pre = None
    if want_dact:
        if act == ACT_NONE:
            raise RuntimeError("gemm_nt: want_dact needs an activation")
        d8 = want_dact == "u8"                              # 8-bit linear code of the derivative (STG_U8_LIN), else bf16
        pre = torch.empty((M, N), dtype=torch.uint8 if d8 else BF16, device=dev)
        a.dact, a.ldp = _p(pre), _ld(pre)
        a.dact_dtype = _lib.STG_U8_LIN if d8 else STG_BF16
    if dact_src is not None:
        d8 = dact_src.dtype == torch.uint8
        _chk2d(dact_src, "dact_src", torch.uint8 if d8 else BF16, cols=N, rows=M)
        if want_dact and d8 != (want_dact == "u8"):
            raise RuntimeError("gemm_nt: dact and dact_src must share one storage type")
        a.dact_src, a.ldd = _p(dact_src), _ld(dact_src)
        a.dact_dtype = _lib.STG_U8_LIN if d8 else STG_BF16
    if row_scale is not None:
        if not row_scale.is_cuda or row_scale.dtype != F32 or not row_scale.is_contiguous():
            raise RuntimeError("row_scale: expected contiguous fp32 GPU tensor")
        need = ((M - 1) // rs_outer) * rs_inner + rs_inner if M > 0 else 0
        if row_scale.numel() < need:
            raise RuntimeError(f"row_scale: needs >= {need} entries, got {row_scale.numel()}")
        a.row_scale, a.rs_outer, a.rs_inner = _p(row_scale), int(rs_outer), int(rs_inner)
    if res1 is not None:
        _chk2d(res1, "res1", res1.dtype, cols=N, rows=M)
        a.res1, a.ldr1, a.res1_dtype = _p(res1), _ld(res1), _dt(res1)
    if res2 is not None:
        _chk2d(res2, "res2", res2.dtype, cols=N, rows=M)
        a.res2, a.ldr2, a.res2_dtype = _p(res2), _ld(res2), _dt(res2)
    a.M, a.N, a.K = M, N, K
    if split is not None:                                 # two row groups: rows >= split_m take (W2, bias2) -- the video | audio adapters
        split_m, W2, bias2 = split
        if fp8 or conv is not None:
            raise RuntimeError("gemm_nt(split=...): bf16 operands, no implicit convolution")
        _chk2d(W2, "W2", BF16, cols=K, rows=N)
        if _ld(W2) != _ld(W) or (bias is None) != (bias2 is None) or split_m % 128 != 0 or not 0 < split_m < M:
            raise RuntimeError("gemm_nt(split=...): W2 must match W's layout, bias2 come with bias, 0 < split_m < M, split_m % 128 == 0")
        if bias2 is not None:
            _chk1d(bias2, "bias2", F32, N)
        a.split_m, a.W2, a.bias2 = int(split_m), _p(W2), _p(bias2)
        a.conv_zero = _p(_zero_line(dev))
    if not fp8 and K % 64 != 0:                           # k tail of the LDS-DMA kernel reads zeros from here (csrc/gemm.hip KTAIL)
        a.conv_zero = _p(_zero_line(dev))
    if conv is not None:
        zl = _zero_line(dev)
        a.conv_H, a.conv_W, a.conv_d, a.conv_C, a.conv_zero = int(Hc), int(Wc), int(dc), int(Cin), _p(zl)
    prof = _gemm_prof
    if prof is not None and M > 0:
        # per-class accounting for bench.py: class = (kernel the C dispatch chose, N, K, epilogue signature).  The kernel of a call
        # signature is learnt from stg_gemm_nt's `kernel_chosen` on its first launch (no host-side copy of the dispatch rules).
        epi = ("b" if bias is not None else "") + ("a" if act else "") + (("p8" if want_dact == "u8" else "p") if want_dact else "") + \
              ("" if dact_src is None else ("d8" if dact_src.dtype == torch.uint8 else "d")) + \
              ("" if res1 is None else ("r" if res1.dtype == BF16 else "q")) + ("" if res2 is None else ("R" if res2.dtype == BF16 else "Q")) + \
              ("s" if row_scale is not None else "") + ("A" if alpha != 1.0 else "") + ("" if out.dtype == BF16 else "F") + \
              ("" if conv is None else f"c{int(dc)}")      # implicit 3x3 convolution (dilation dc): A is read as [M, Cin], K = 9 Cin
        sig = (M, N, K, epi, fp8)
        kid = prof["kid"].get(sig)
        if kid is not None:
            ck = (kid, N, K, epi)
            pc = prof["classes"].get(ck)
            if pc is None:
                pc = prof["classes"][ck] = {"launches": 0, "flops": 0.0, "bytes": 0.0, "rec": []}
            esz = 1.0 if fp8 else 2.0
            nbytes = esz * M * (K if conv is None else K // 9) + esz * N * K + M * N * out.element_size() + (pre.element_size() * M * N if want_dact else 0.0) + \
                (dact_src.element_size() * M * N if dact_src is not None else 0.0) + (M * N * res1.element_size() if res1 is not None else 0.0) + \
                (M * N * res2.element_size() if res2 is not None else 0.0)          # algorithmic HBM bytes: every operand / output once
            pc["launches"] += 1
            pc["flops"] += 2.0 * M * N * K
            pc["bytes"] += nbytes
            if prof["seq"] is not None:                    # launch-ordered class log (bench.py --gemm-seq; tools/ledger.py joins it with a trace)
                prof["seq"].append((kid, M, N, K, epi, nbytes))
            if pc["launches"] % prof["stride"] == 0:       # HIP events around every stride-th launch of the class, on the launch stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(_lib.lib().stg_gemm_nt(C.byref(a), _stream()), "stg_gemm_nt")
                e1.record()
                pc["rec"].append((e0, e1, 2.0 * M * N * K, nbytes))
                return (out, pre) if want_dact else out
        _lib.check(_lib.lib().stg_gemm_nt(C.byref(a), _stream()), "stg_gemm_nt")
        if kid is None:
            prof["kid"][sig] = int(a.kernel_chosen)
        return (out, pre) if want_dact else out
    _lib.check(_lib.lib().stg_gemm_nt(C.byref(a), _stream()), "stg_gemm_nt")
    global LAST_GEMM_KERNEL
    LAST_GEMM_KERNEL = GEMM_KERNEL_NAMES.get(int(a.kernel_chosen), str(int(a.kernel_chosen)))     # which kernel the C dispatch chose (tests)
    return (out, pre) if want_dact else out


@_family_io("dec_tpavi_bmm", flops=lambda A, W, *a, **kw: 2.0 * A.shape[0] * W.shape[1] * A.shape[1])
def _gemm_nt_batched(A, W, bias, out, out_dtype, alpha, act, nb):
    """nb independent problems of one shape: A [nb * M, K] (consecutive row groups), W [nb, N, K] -> C [nb * M, N]."""
    _chk2d(A, "A", BF16)
    R, Kd = A.shape
    if W.dim() != 3 or W.shape[0] != nb or W.shape[2] != Kd or W.dtype != BF16 or not W.is_contiguous() or R % nb != 0 or Kd % 64 != 0:
        raise RuntimeError("gemm_nt(batch=nb): needs A [nb*M, K], contiguous bf16 W [nb, N, K], K % 64 == 0")
    M, N = R // nb, W.shape[1]
    if out is None:
        out = torch.empty((R, N), dtype=out_dtype, device=A.device)
    _chk2d(out, "out", out.dtype, cols=N, rows=R)
    a = _lib.GemmArgs()
    a.A, a.lda = _p(A), _ld(A)
    a.W, a.ldw = _p(W), Kd
    a.C, a.ldc, a.c_dtype = _p(out), _ld(out), (STG_BF16 if out.dtype == BF16 else STG_F32)
    if bias is not None:
        _chk1d(bias, "bias", F32, N)
    a.bias, a.alpha, a.act = _p(bias), float(alpha), int(act)
    a.M, a.N, a.K = M, N, Kd
    a.batch, a.a_bstride, a.w_bstride, a.c_bstride = nb, M * _ld(A), N * Kd, M * _ld(out)
    _lib.check(_lib.lib().stg_gemm_nt(C.byref(a), _stream()), "stg_gemm_nt (batched)")
    return out


def _sum_splits(ws, out=None, accumulate=False):
    """ws [S, ...] or [nb, S, ...] (batched when out is 1 dim smaller than ws by the split axis): fold the split axis in fp32."""
    batched = out is not None and ws.dim() == out.dim() + 1 and ws.dim() >= 3 and ws.shape[0] == out.shape[0] and ws.shape[2:] == out.shape[1:]
    if batched:
        nb, S = ws.shape[0], ws.shape[1]
    else:
        nb, S = 1, ws.shape[0]
    n = ws.numel() // (nb * S)
    if out is None:
        out = torch.empty(ws.shape[1:], dtype=F32, device=ws.device)
    if not out.is_contiguous() or out.dtype != F32 or out.numel() != nb * n:
        raise RuntimeError("_sum_splits: out must be a contiguous fp32 tensor of one slice")
    _lib.check(_lib.lib().stg_sum_splits(_p(ws), _p(out), int(S), n, nb, 1 if accumulate else 0, _stream()), "stg_sum_splits")
    return out


@_family_io("dec_tpavi_bmm", flops=lambda A, B, rows: 2.0 * A.shape[0] * A.shape[1] * B.shape[1])
def bmm_tn(A, B, rows):
    """[nb, N1, N2] fp32 = A_b^T B_b over consecutive groups of `rows` rows of A [nb*rows, N1] and B [nb*rows, N2] (bf16);
    N1, N2 multiples of 128."""
    _chk2d(A, "A", BF16)
    _chk2d(B, "B", BF16, rows=A.shape[0])
    R = A.shape[0]
    if rows <= 0 or R % rows != 0:
        raise RuntimeError("bmm_tn: rows must divide the row count")
    nb, N1, N2 = R // rows, A.shape[1], B.shape[1]
    splits = C.c_int(0)
    n = _lib.lib().stg_wgrad_wide_ws_floats(rows, N1, N2, C.byref(splits))
    if n <= 0:
        raise RuntimeError("bmm_tn: widths must be multiples of 128")
    ws = torch.empty((nb, splits.value, N1, N2), dtype=F32, device=A.device)
    _lib.check(_lib.lib().stg_wgrad_wide_batched(_p(A), _ld(A), _p(B), _ld(B), _p(_zero_line(A.device)), _p(ws), ws.numel(), rows, N1, N2,
                                                 nb, _stream()), "stg_wgrad_wide_batched")
    out = torch.empty((nb, N1, N2), dtype=F32, device=A.device)
    return _sum_splits(ws, out)


_gemm_prof = None
from . import config as _cfg
USE_WGRAD_WS = _cfg.opt("wgrad_ws")          # 0 = atomic wgrad kernels only (A/B knob)
USE_WGRAD_MULTI = _cfg.opt("wgrad_multi")    # 0 = one launch pair per adapter Linear (A/B knob)


LAST_GEMM_KERNEL = None
GEMM_KERNEL_NAMES = {_lib.GEMM_KERNEL_REG: "gemm_nt_kernel", _lib.GEMM_KERNEL_GLDS: "gemm_nt_glds_kernel<1, false, false, false>",
                     _lib.GEMM_KERNEL_BIG: "gemm_nt_big_kernel", _lib.GEMM_KERNEL_8PH: "gemm_nt_8ph_kernel", _lib.GEMM_KERNEL_8PHM: "gemm_nt_8phm_kernel",
                     _lib.GEMM_KERNEL_GLDS_CONV: "gemm_nt_glds_kernel<1, true, false, false>",
                     _lib.GEMM_KERNEL_GLDS_BATCH: "gemm_nt_glds_kernel<1, false, true, false>",
                     _lib.GEMM_KERNEL_GLDS_KTAIL: "gemm_nt_glds_kernel<1, false, false, true>", _lib.GEMM_KERNEL_FP8: "gemm_nt_fp8_kernel",
                     _lib.GEMM_KERNEL_SKINNY: "skinny_down_kernel"}


def gemm_profile_start(stride=7, log_sequence=False):
    """Start per-class accounting of stg_gemm_nt launches (class = kernel chosen by the C dispatch x N x K x epilogue signature):
    every `stride`-th launch of a class is bracketed by HIP events on the launch stream (events around every launch cost ~10 % of
    the step).  Call before the warm-up steps (the kernel of each call signature is learnt on its first launch) and
    gemm_profile_reset() at the start of the timed region."""
    global _gemm_prof
    _gemm_prof = {"stride": int(stride), "kid": {}, "classes": {}, "seq": [] if log_sequence else None}


def gemm_profile_reset():
    if _gemm_prof is not None:
        _gemm_prof["classes"] = {}
        if _gemm_prof["seq"] is not None:
            _gemm_prof["seq"] = []


def gemm_profile_sequence():
    """[(kernel name, M, N, K, epilogue signature, algorithmic bytes)] of every stg_gemm_nt launch since the last reset, in launch order
    (gemm_profile_start(log_sequence=True))."""
    seq = _gemm_prof["seq"] if _gemm_prof is not None else None
    return [(GEMM_KERNEL_NAMES.get(k, str(k)), M, N, Kd, epi, nb) for k, M, N, Kd, epi, nb in (seq or [])]


def gemm_profile_stop():
    """[{kernel, N, K, epi, launches, flops, bytes, sampled, sampled_ms, sampled_flops, sampled_bytes}] since the last reset."""
    global _gemm_prof
    prof, _gemm_prof = _gemm_prof, None
    torch.cuda.synchronize()
    out = []
    for (kid, N, Kd, epi), pc in prof["classes"].items():
        rec = pc["rec"]
        out.append({"kernel": GEMM_KERNEL_NAMES.get(kid, str(kid)), "N": N, "K": Kd, "epi": epi, "launches": pc["launches"],
                    "flops": pc["flops"], "bytes": pc["bytes"], "sampled": len(rec),
                    "sampled_ms": sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec), "sampled_flops": sum(r[2] for r in rec),
                    "sampled_bytes": sum(r[3] for r in rec)})
    return out


@_family("wgrad", lambda dY, X, dW, db=None, **kw: ((dY.shape[1], X.shape[1]), _nb(dY, X) + 4.0 * dW.numel(), 2.0 * dY.shape[0] * dY.shape[1] * X.shape[1]))
def wgrad_tn(dY, X, dW, db=None, *, n1=None, row_scale=None, rs_outer=1, rs_inner=1):
    """dW[N1,N2] += (s * dY).T @ X ; db[N1] += (s * dY).sum(0)   (fp32 accumulate into existing buffers).
    n1: use only the first n1 columns of dY (dY may be zero-padded to a multiple of 8 columns)."""
    M = dY.shape[0]
    N1 = dY.shape[1] if n1 is None else int(n1)
    N2 = X.shape[1]
    _chk2d(dY, "dY", BF16)
    if N1 > dY.shape[1]:
        raise RuntimeError("wgrad_tn: n1 exceeds dY columns")
    _chk2d(X, "X", BF16, rows=M)
    _chk2d(dW, "dW", F32, cols=N2, rows=N1)
    if db is not None:
        _chk1d(db, "db", F32, N1)
    if row_scale is not None:
        if not row_scale.is_cuda or row_scale.dtype != F32 or not row_scale.is_contiguous():
            raise RuntimeError("row_scale: expected contiguous fp32 GPU tensor")
        need = ((M - 1) // rs_outer) * rs_inner + rs_inner if M > 0 else 0
        if row_scale.numel() < need:
            raise RuntimeError(f"row_scale: needs >= {need} entries, got {row_scale.numel()}")
    L = _lib.lib()
    if USE_WGRAD_WS and row_scale is None and N1 % 128 == 0 and N2 % 128 == 0 and N1 == dY.shape[1] and M >= 4096:
        # both operands wide (the AVS decoder's Linears / 1x1 convolutions): tn-GEMM with partial tiles instead of the atomic fallback
        splits = C.c_int(0)
        n = L.stg_wgrad_wide_ws_floats(M, N1, N2, C.byref(splits))
        ws = torch.empty((splits.value, N1, N2), dtype=F32, device=dY.device)
        dbw = torch.empty((splits.value, N1), dtype=F32, device=dY.device) if db is not None else None
        _lib.check(L.stg_wgrad_wide(_p(dY), _ld(dY), _p(X), _ld(X), _p(_zero_line(dY.device)), _p(ws), n, _p(dbw), M, N1, N2, _stream()),
                   "stg_wgrad_wide")
        if dW.is_contiguous():
            _sum_splits(ws, dW, accumulate=True)
        else:
            dW.add_(_sum_splits(ws))
        if db is not None:
            db.add_(dbw.sum(0))                           # [splits, N1] -> [N1]: a few hundred floats
        return
    nws = _wgrad_ws_floats(M, N1, N2) if USE_WGRAD_WS else 0
    if nws > 0:                                          # partial tiles + reduce (no memory-side atomics)
        ws = torch.empty((nws,), dtype=F32, device=dY.device)
        _lib.check(L.stg_wgrad_tn_ws(_p(dY), _ld(dY), _p(X), _ld(X), _p(dW), _ld(dW), _p(db), M, N1, N2,
                                     _p(row_scale), int(rs_outer), int(rs_inner), _p(ws), nws, _stream()), "stg_wgrad_tn_ws")
        return
    _lib.check(L.stg_wgrad_tn(_p(dY), _ld(dY), _p(X), _ld(X), _p(dW), _ld(dW), _p(db), M, N1, N2,
                              _p(row_scale), int(rs_outer), int(rs_inner), _stream()), "stg_wgrad_tn")


_mlp_perm_cache = {}
_wgrad_ws_cache = {}


def _wgrad_ws_floats(M, N1, N2):
    """stg_wgrad_ws_floats evaluates every launch plan of (M, N1, N2) on the host (WMAX x 3 plans, each a loop of up to 1 024 candidates) and it
    sat on the eager launch path: its value depends on the shape alone, so it is cached here.  Note that the row split of a problem
    depends on how many problems its launch carries and on option wgrad_plan: a single and a multi launch sum the same gradient in a different
    order -- equal to fp32 rounding, not bit-identical."""
    key = (int(M), int(N1), int(N2))
    v = _wgrad_ws_cache.get(key)
    if v is None:
        v = _wgrad_ws_cache[key] = int(_lib.lib().stg_wgrad_ws_floats(*key))
    return v


def mlp_fused_supported(C_):
    return bool(_lib.lib().stg_mlp_fused_supported(int(C_)))


def mlp_w2_perm(hidden, device):
    """Index tensor `perm` with W2p = W2[:, perm] (stg_mlp_w2_perm)."""
    key = (int(hidden), str(device))
    t = _mlp_perm_cache.get(key)
    if t is None:
        buf = (C.c_int * hidden)()
        _lib.check(_lib.lib().stg_mlp_w2_perm(hidden, buf), "stg_mlp_w2_perm")
        t = torch.tensor(list(buf), dtype=torch.long, device=device)
        _mlp_perm_cache[key] = t
    return t


@_family("mlp_fused_fwd", lambda Y, W1, b1, W2p, b2, out=None: (Y.shape[1], 2.0 * _nb(Y) + _nb(W1, W2p), 16.0 * Y.shape[0] * Y.shape[1] * Y.shape[1]))
def mlp_fwd(Y, W1, b1, W2p, b2, out=None):
    """out = fc2(GELU(fc1(Y))) in one kernel (stg_mlp_fwd).  Y [rows, C] bf16, W1 [4C, C] bf16, W2p [C, 4C] bf16 = W2[:, mlp_w2_perm]."""
    rows, C_ = Y.shape
    _chk2d(Y, "Y", BF16)
    _chk2d(W1, "W1", BF16, cols=C_, rows=4 * C_)
    _chk2d(W2p, "W2p", BF16, cols=4 * C_, rows=C_)
    _chk1d(b1, "b1", F32, 4 * C_)
    _chk1d(b2, "b2", F32, C_)
    if not W1.is_contiguous() or not W2p.is_contiguous():
        raise RuntimeError("mlp_fwd: weights must be contiguous")
    if out is None:
        out = torch.empty((rows, C_), dtype=BF16, device=Y.device)
    _chk2d(out, "out", BF16, cols=C_, rows=rows)
    _lib.check(_lib.lib().stg_mlp_fwd(_p(Y), _ld(Y), _p(W1), _p(b1), _p(W2p), _p(b2), _p(out), _ld(out), rows, C_, _stream()), "stg_mlp_fwd")
    return out


@_family("mlp_fused_bwd", lambda Y, dM, W1, b1, W2T, out=None: (Y.shape[1], 3.0 * _nb(Y) + _nb(W1, W2T), 24.0 * Y.shape[0] * Y.shape[1] * Y.shape[1]))
def mlp_bwd(Y, dM, W1, b1, W2T, out=None):
    """dY = ((dM @ W2) * GELU'(Y @ W1.T + b1)) @ W1 in one kernel (stg_mlp_bwd).  W2T [4C, C] bf16 = fc2.weight.T."""
    rows, C_ = Y.shape
    _chk2d(Y, "Y", BF16)
    _chk2d(dM, "dM", BF16, cols=C_, rows=rows)
    _chk2d(W1, "W1", BF16, cols=C_, rows=4 * C_)
    _chk2d(W2T, "W2T", BF16, cols=C_, rows=4 * C_)
    _chk1d(b1, "b1", F32, 4 * C_)
    if not W1.is_contiguous() or not W2T.is_contiguous():
        raise RuntimeError("mlp_bwd: weights must be contiguous")
    if out is None:
        out = torch.empty((rows, C_), dtype=BF16, device=Y.device)
    _chk2d(out, "out", BF16, cols=C_, rows=rows)
    _lib.check(_lib.lib().stg_mlp_bwd(_p(Y), _ld(Y), _p(dM), _ld(dM), _p(W1), _p(b1), _p(W2T), _p(out), _ld(out), rows, C_, _stream()),
               "stg_mlp_bwd")
    return out


@_family("wgrad", lambda problems: (("multi", len(problems), problems[0][0].shape[1], problems[0][1].shape[1]) if problems else ("multi", 0, 0, 0),
                                    sum(_nb(q[0], q[1]) + 4.0 * q[2].numel() for q in problems), sum(2.0 * q[0].shape[0] * q[0].shape[1] * q[1].shape[1] for q in problems)))
def wgrad_tn_multi(problems):
    """Several wgrad_tn calls at once: problems = [(dY, X, dW, db, row_scale, rs_outer, rs_inner)].  Problems that share a launch
    plan of the workspace path (same M, narrow-width class, wide width) go out as ONE pair of launches (stg_wgrad_tn_ws_multi);
    the rest one by one.  Same results as the individual calls (same kernels, same row splits)."""
    L = _lib.lib()
    groups, single = {}, []
    for pr in problems:
        dY, X, dW, db, rs, ro, ri = pr
        M, N1, N2 = dY.shape[0], dY.shape[1], X.shape[1]
        nws = _wgrad_ws_floats(M, N1, N2) if (USE_WGRAD_WS and USE_WGRAD_MULTI) else 0
        if nws > 0 and dW.is_contiguous() and dY.dtype == BF16 and X.dtype == BF16 and X.shape[0] == M and dY.stride(1) == 1 and X.stride(1) == 1 \
                and tuple(dW.shape) == (N1, N2):
            groups.setdefault((M, (min(N1, N2) + 15) // 16, max(N1, N2), nws), []).append(pr)
        else:
            single.append(pr)
    for (M, _, _, nws), prs in groups.items():
        if len(prs) == 1:
            single.append(prs[0])
            continue
        for i0 in range(0, len(prs), 16):
            chunk = prs[i0:i0 + 16]
            n = len(chunk)
            descs = (_lib.WgradDesc * n)()
            for d, (dY, X, dW, db, rs, ro, ri) in zip(descs, chunk):
                d.dY, d.lddy, d.X, d.ldx = _p(dY), _ld(dY), _p(X), _ld(X)
                d.dW, d.lddw, d.db = _p(dW), _ld(dW), _p(db)
                d.M, d.N1, d.N2 = M, dY.shape[1], X.shape[1]
                d.row_scale, d.rs_outer, d.rs_inner = _p(rs), int(ro), int(ri)
            ws = torch.empty((n * nws,), dtype=F32, device=chunk[0][0].device)
            rc = L.stg_wgrad_tn_ws_multi(descs, n, _p(ws), n * nws, _stream())
            if rc == -7:                                  # not one plan after all (alignment of a view): one by one
                single.extend(chunk)
            else:
                _lib.check(rc, "stg_wgrad_tn_ws_multi")
    for dY, X, dW, db, rs, ro, ri in single:
        wgrad_tn(dY, X, dW, db, row_scale=rs, rs_outer=ro, rs_inner=ri)


@_family("ln_fwd", lambda x, gamma, beta, eps=1e-5, *, gather4=None, want_stats=True, out=None, out_dtype=BF16:
         (gamma.numel(), _nb(x) + x.numel() * (2.0 if out is None and out_dtype == BF16 else (out.element_size() if out is not None else 4.0)), 0.0))
def layernorm_fwd(x, gamma, beta, eps=1e-5, *, gather4=None, want_stats=True, out=None, out_dtype=BF16):
    """x: [rows, C] bf16/fp32 -> y bf16 (+ mean, rstd).  gather4=(H, W): PatchMerging gather, x is [F*H*W, C] -> y [F*H*W/4, 4C]."""
    if x.dtype not in (BF16, F32):
        raise RuntimeError("layernorm: x must be bf16 or fp32")
    _chk2d(x, "x", x.dtype)
    R, Cs = x.shape
    if gather4 is not None:
        H, W = gather4
        if R % (H * W) != 0:
            raise RuntimeError("layernorm gather4: rows not a multiple of H*W")
        rows, Cl = R // 4, 4 * Cs
        g4 = 1
    else:
        H = W = 0
        rows, Cl = R, Cs
        g4 = 0
    _chk1d(gamma, "gamma", F32, Cl)
    _chk1d(beta, "beta", F32, Cl)
    y = torch.empty((rows, Cl), dtype=out_dtype, device=x.device) if out is None else out
    _chk2d(y, "y", y.dtype, cols=Cl, rows=rows)
    mean = torch.empty((rows,), dtype=F32, device=x.device) if want_stats else None
    rstd = torch.empty((rows,), dtype=F32, device=x.device) if want_stats else None
    _lib.check(_lib.lib().stg_layernorm_fwd(_p(x), STG_BF16 if x.dtype == BF16 else STG_F32, _ld(x), _p(gamma), _p(beta),
                                            float(eps), _p(y), _dt(y), _ld(y), _p(mean), _p(rstd), rows, Cl, g4, H, W, _stream()),
               "stg_layernorm_fwd")
    return y, mean, rstd


@_family("ln_bwd", lambda dy, x, gamma, mean, rstd, *, add_to=None, **kw: (gamma.numel(), _nb(dy, x, add_to) + 2.0 * x.numel(), 0.0))
def layernorm_bwd(dy, x, gamma, mean, rstd, *, add_to=None, gather4=None, dgamma=None, dbeta=None):
    """Returns dx (bf16, shaped like x).  add_to (shaped like x) is added to the result."""
    _chk2d(x, "x", x.dtype)
    R, Cs = x.shape
    if gather4 is not None:
        H, W = gather4
        rows, Cl, g4 = R // 4, 4 * Cs, 1
    else:
        H = W = 0
        rows, Cl, g4 = R, Cs, 0
    _chk2d(dy, "dy", BF16, cols=Cl, rows=rows)
    _chk1d(gamma, "gamma", F32, Cl)
    _chk1d(mean, "mean", F32, rows)
    _chk1d(rstd, "rstd", F32, rows)
    if add_to is not None:
        _chk2d(add_to, "add_to", BF16, cols=Cs, rows=R)
    if dgamma is not None:
        _chk1d(dgamma, "dgamma", F32, Cl)
        _chk1d(dbeta, "dbeta", F32, Cl)
    dx = torch.empty((R, Cs), dtype=BF16, device=x.device)
    _lib.check(_lib.lib().stg_layernorm_bwd(_p(dy), _ld(dy), _p(x), STG_BF16 if x.dtype == BF16 else STG_F32, _ld(x),
                                            _p(gamma), _p(mean), _p(rstd), _p(add_to), _ld(add_to) if add_to is not None else 0,
                                            _p(dx), _ld(dx), _p(dgamma), _p(dbeta), rows, Cl, g4, H, W, _stream()),
               "stg_layernorm_bwd")
    return dx


def up_ln_supported(C_, K_):
    return bool(_lib.lib().stg_up_ln_supported(int(C_), int(K_)))


@_family("upln_fwd", lambda h, w, bias, res32, gamma, beta, *, res16=None, **kw:
         ((w.shape[0], h.shape[1]), _nb(h, w, res32, res16) + 6.0 * res32.numel(), 2.0 * h.shape[0] * h.shape[1] * w.shape[0]))
def up_ln_fwd(h, w, bias, res32, gamma, beta, *, res16=None, row_scale=None, rs_outer=1, rs_inner=1, out=None, eps=1e-5,
              want_stats=True, y_out=None, mean_out=None, rstd_out=None):
    """x = res32 (+ res16) + rs * (h w^T + bias) (fp32) and y = LayerNorm(x) * gamma + beta (bf16) in one pass.
    h [M, K] bf16, w [C, >= K] bf16 (shadow of D_fc2.weight), res32 [M, C] fp32, res16 [M, C] bf16 or None.
    Returns (x, y, mean, rstd)."""
    _chk2d(h, "h", BF16)
    M, K_ = h.shape
    Cc = w.shape[0]
    _chk2d(w, "w", BF16, rows=Cc)
    if w.shape[1] < K_:
        raise RuntimeError("up_ln: w has fewer columns than h")
    _chk1d(bias, "bias", F32, Cc)
    _chk1d(gamma, "gamma", F32, Cc)
    _chk1d(beta, "beta", F32, Cc)
    _chk2d(res32, "res32", F32, cols=Cc, rows=M)
    if res16 is not None:
        _chk2d(res16, "res16", BF16, cols=Cc, rows=M)
    x = torch.empty((M, Cc), dtype=F32, device=h.device) if out is None else out
    _chk2d(x, "x", F32, cols=Cc, rows=M)
    if x.data_ptr() == res32.data_ptr():
        raise RuntimeError("up_ln: x must not alias res32")
    if row_scale is not None:
        if row_scale.dtype != F32 or not row_scale.is_cuda or not row_scale.is_contiguous():
            raise RuntimeError("up_ln: row_scale must be a contiguous fp32 GPU vector")
        if M > 0 and ((M - 1) // rs_outer) * rs_inner + rs_inner > row_scale.numel():
            raise RuntimeError("up_ln: row_scale too short for (M, rs_outer, rs_inner)")
    y = torch.empty((M, Cc), dtype=BF16, device=h.device) if y_out is None else y_out
    _chk2d(y, "y", BF16, cols=Cc, rows=M)
    mean, rstd = mean_out, rstd_out
    if mean is None and want_stats:
        mean = torch.empty((M,), dtype=F32, device=h.device)
        rstd = torch.empty((M,), dtype=F32, device=h.device)
    if mean is not None:
        _chk1d(mean, "mean", F32, M)
        _chk1d(rstd, "rstd", F32, M)
    _lib.check(_lib.lib().stg_up_ln_fwd(_p(h), _ld(h), _p(w), _ld(w), _p(bias), _p(res32), _ld(res32), _p(res16),
                                        _ld(res16) if res16 is not None else 0, _p(row_scale), int(rs_outer), int(rs_inner),
                                        _p(x), _ld(x), _p(gamma), _p(beta), float(eps), _p(y), _ld(y), _p(mean), _p(rstd),
                                        M, Cc, K_, _stream()), "stg_up_ln_fwd")
    return x, y, mean, rstd


@_family("upln_fwd", lambda h, h2, w, w2, bias, bias2, res32, gamma, beta, *, res16=None, **kw:
         ((w.shape[0], h.shape[1], "pair"), _nb(h, h2, w, w2, res32, res16) + 6.0 * res32.numel(), 2.0 * res32.shape[0] * h.shape[1] * w.shape[0]))
def up_ln_fwd_pair(h, h2, w, w2, bias, bias2, res32, gamma, beta, *, res16=None, row_scale=None, row_scale2=None, rs_outer=1, rs_inner=1,
                   out, y_out, mean_out=None, rstd_out=None, eps=1e-5):
    """up_ln_fwd for both modalities in ONE launch: rows [0, h.shape[0]) with (h, w, bias, row_scale), the rest with (h2, w2, bias2,
    row_scale2); res32 / res16 / out / y_out / mean_out / rstd_out span all rows.  Same arithmetic per row as two up_ln_fwd calls."""
    _chk2d(h, "h", BF16)
    S, K_ = h.shape
    _chk2d(h2, "h2", BF16, cols=K_)
    M = S + h2.shape[0]
    Cc = w.shape[0]
    _chk2d(w, "w", BF16, rows=Cc)
    _chk2d(w2, "w2", BF16, rows=Cc)
    if w.shape[1] < K_ or w2.shape[1] < K_ or _ld(w) != _ld(w2) or _ld(h) != _ld(h2):
        raise RuntimeError("up_ln_pair: the two row groups must share K and leading dimensions")
    if S % 16 or S == 0 or h2.shape[0] == 0:
        raise RuntimeError("up_ln_pair: the first row group must hold a positive multiple of 16 rows, the second at least one")
    for name, t in (("bias", bias), ("bias2", bias2), ("gamma", gamma), ("beta", beta)):
        _chk1d(t, name, F32, Cc)
    _chk2d(res32, "res32", F32, cols=Cc, rows=M)
    if res16 is not None:
        _chk2d(res16, "res16", BF16, cols=Cc, rows=M)
    _chk2d(out, "x", F32, cols=Cc, rows=M)
    _chk2d(y_out, "y", BF16, cols=Cc, rows=M)
    if out.data_ptr() == res32.data_ptr():
        raise RuntimeError("up_ln_pair: x must not alias res32")
    if (row_scale is None) != (row_scale2 is None):
        raise RuntimeError("up_ln_pair: both row groups or neither carry a row_scale")
    for rs_, rows_ in ((row_scale, S), (row_scale2, M - S)):
        if rs_ is not None:
            if rs_.dtype != F32 or not rs_.is_cuda or not rs_.is_contiguous():
                raise RuntimeError("up_ln_pair: row_scale must be a contiguous fp32 GPU vector")
            if ((rows_ - 1) // rs_outer) * rs_inner + rs_inner > rs_.numel():
                raise RuntimeError("up_ln_pair: row_scale too short for (rows, rs_outer, rs_inner)")
    if mean_out is not None:
        _chk1d(mean_out, "mean", F32, M)
        _chk1d(rstd_out, "rstd", F32, M)
    _lib.check(_lib.lib().stg_up_ln_fwd_pair(_p(h), _p(h2), _ld(h), _p(w), _p(w2), _ld(w), _p(bias), _p(bias2), S, _p(res32), _ld(res32),
                                             _p(res16), _ld(res16) if res16 is not None else 0, _p(row_scale), _p(row_scale2), int(rs_outer),
                                             int(rs_inner), _p(out), _ld(out), _p(gamma), _p(beta), float(eps), _p(y_out), _ld(y_out),
                                             _p(mean_out), _p(rstd_out), M, Cc, K_, _stream()), "stg_up_ln_fwd_pair")
    return out, y_out, mean_out, rstd_out


@_family("ln_bwd_down", lambda dy, x, gamma, mean, rstd, wt, wt2, split, *, add_to=None, **kw:
         ((x.shape[1], wt.shape[0], "pair"), _nb(dy, x, add_to, wt, wt2) + 2.0 * x.numel() + 2.0 * x.shape[0] * wt.shape[0], 2.0 * x.shape[0] * x.shape[1] * wt.shape[0]))
def ln_bwd_down_pair(dy, x, gamma, mean, rstd, wt, wt2, split, *, add_to=None, row_scale=None, row_scale2=None, rs_outer=1, rs_inner=1, dx_out=None):
    """ln_bwd_down (x fp32 + gamma + mean) or ln_bwd_down_xhat (x bf16 normalised rows, gamma = mean = None) for both modalities in ONE
    launch: rows [0, split) project onto wt, the rest onto wt2.  Returns (dx [M, C], dh [M, J]): dh[:split] / dh[split:] are the two
    single-launch results."""
    xh = mean is None
    _chk2d(x, "x", BF16 if xh else F32)
    M, Cc = x.shape
    _chk2d(dy, "dy", BF16, cols=Cc, rows=M)
    J = wt.shape[0]
    _chk2d(wt, "wt", BF16, rows=J)
    _chk2d(wt2, "wt2", BF16, rows=J)
    if wt.shape[1] < Cc or wt2.shape[1] < Cc or _ld(wt) != _ld(wt2):
        raise RuntimeError("ln_bwd_down_pair: wt / wt2 must be [J, >= C] with one leading dimension")
    if split % 16 or not 0 < split < M:
        raise RuntimeError("ln_bwd_down_pair: split must be a multiple of 16 inside (0, M)")
    if not xh:
        _chk1d(gamma, "gamma", F32, Cc)
        _chk1d(mean, "mean", F32, M)
    _chk1d(rstd, "rstd", F32, M)
    if add_to is not None:
        _chk2d(add_to, "add_to", BF16, cols=Cc, rows=M)
    if (row_scale is None) != (row_scale2 is None):
        raise RuntimeError("ln_bwd_down_pair: both row groups or neither carry a row_scale")
    for rs_, rows_ in ((row_scale, split), (row_scale2, M - split)):
        if rs_ is not None:
            if rs_.dtype != F32 or not rs_.is_cuda or not rs_.is_contiguous():
                raise RuntimeError("ln_bwd_down_pair: row_scale must be a contiguous fp32 GPU vector")
            if ((rows_ - 1) // rs_outer) * rs_inner + rs_inner > rs_.numel():
                raise RuntimeError("ln_bwd_down_pair: row_scale too short for (rows, rs_outer, rs_inner)")
    dx = torch.empty((M, Cc), dtype=BF16, device=x.device) if dx_out is None else dx_out
    _chk2d(dx, "dx", BF16, cols=Cc, rows=M)
    dh = torch.empty((M, J), dtype=BF16, device=x.device)
    _lib.check(_lib.lib().stg_ln_bwd_down_pair(1 if xh else 0, _p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), _p(mean), _p(rstd), _p(add_to),
                                               _ld(add_to) if add_to is not None else 0, _p(dx), _ld(dx), _p(wt), _p(wt2), _ld(wt),
                                               _p(row_scale), _p(row_scale2), int(rs_outer), int(rs_inner), _p(dh), _p(dh[split:]), _ld(dh),
                                               int(split), M, Cc, J, _stream()), "stg_ln_bwd_down_pair")
    return dx, dh


def ln_bwd_down_supported(C_, J_):
    return bool(_lib.lib().stg_ln_bwd_down_supported(int(C_), int(J_)))


@_family("ln_bwd_down", lambda dy, x, gamma, mean, rstd, wt, *, add_to=None, **kw:
         ((x.shape[1], wt.shape[0]), _nb(dy, x, add_to, wt) + 2.0 * x.numel() + 2.0 * x.shape[0] * wt.shape[0], 2.0 * x.shape[0] * x.shape[1] * wt.shape[0]))
def ln_bwd_down(dy, x, gamma, mean, rstd, wt, *, add_to=None, row_scale=None, rs_outer=1, rs_inner=1, dx_out=None):
    """dx = LayerNorm backward (+ add_to), bf16, and dh = rs * (dx wt^T), bf16, in one pass.  x [M, C] fp32 (the normalised
    residual row), dy / add_to [M, C] bf16, wt [J, C] bf16 (transposed shadow of D_fc2.weight).  Returns (dx, dh)."""
    _chk2d(x, "x", F32)
    M, Cc = x.shape
    _chk2d(dy, "dy", BF16, cols=Cc, rows=M)
    J = wt.shape[0]
    _chk2d(wt, "wt", BF16, rows=J)
    if wt.shape[1] < Cc:
        raise RuntimeError("ln_bwd_down: wt has fewer columns than x")
    _chk1d(gamma, "gamma", F32, Cc)
    _chk1d(mean, "mean", F32, M)
    _chk1d(rstd, "rstd", F32, M)
    if add_to is not None:
        _chk2d(add_to, "add_to", BF16, cols=Cc, rows=M)
    if row_scale is not None:
        if row_scale.dtype != F32 or not row_scale.is_cuda or not row_scale.is_contiguous():
            raise RuntimeError("ln_bwd_down: row_scale must be a contiguous fp32 GPU vector")
        if M > 0 and ((M - 1) // rs_outer) * rs_inner + rs_inner > row_scale.numel():
            raise RuntimeError("ln_bwd_down: row_scale too short for (M, rs_outer, rs_inner)")
    dx = torch.empty((M, Cc), dtype=BF16, device=x.device) if dx_out is None else dx_out
    _chk2d(dx, "dx", BF16, cols=Cc, rows=M)
    dh = torch.empty((M, J), dtype=BF16, device=x.device)
    _lib.check(_lib.lib().stg_ln_bwd_down(_p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), _p(mean), _p(rstd), _p(add_to),
                                          _ld(add_to) if add_to is not None else 0, _p(dx), _ld(dx), _p(wt), _ld(wt),
                                          _p(row_scale), int(rs_outer), int(rs_inner), _p(dh), _ld(dh), M, Cc, J, _stream()),
               "stg_ln_bwd_down")
    return dx, dh


@_family("ln_bwd_down", lambda dy, xhat, rstd, wt, *, add_to=None, **kw:
         ((xhat.shape[1], wt.shape[0]), _nb(dy, xhat, add_to, wt) + 2.0 * xhat.numel() + 2.0 * xhat.shape[0] * wt.shape[0],
          2.0 * xhat.shape[0] * xhat.shape[1] * wt.shape[0]))
def ln_bwd_down_xhat(dy, xhat, rstd, wt, *, add_to=None, row_scale=None, rs_outer=1, rs_inner=1, dx_out=None):
    """ln_bwd_down from the NORMALISED row: xhat [M, C] bf16 (what the forward wrote in place of y; gamma / beta live in the frozen
    GEMM weight behind it), rstd [M].  Returns (dx, dh)."""
    _chk2d(xhat, "xhat", BF16)
    M, Cc = xhat.shape
    _chk2d(dy, "dy", BF16, cols=Cc, rows=M)
    J = wt.shape[0]
    _chk2d(wt, "wt", BF16, rows=J)
    if wt.shape[1] < Cc:
        raise RuntimeError("ln_bwd_down_xhat: wt has fewer columns than xhat")
    _chk1d(rstd, "rstd", F32, M)
    if add_to is not None:
        _chk2d(add_to, "add_to", BF16, cols=Cc, rows=M)
    if row_scale is not None:
        if row_scale.dtype != F32 or not row_scale.is_cuda or not row_scale.is_contiguous():
            raise RuntimeError("ln_bwd_down_xhat: row_scale must be a contiguous fp32 GPU vector")
        if M > 0 and ((M - 1) // rs_outer) * rs_inner + rs_inner > row_scale.numel():
            raise RuntimeError("ln_bwd_down_xhat: row_scale too short for (M, rs_outer, rs_inner)")
    dx = torch.empty((M, Cc), dtype=BF16, device=xhat.device) if dx_out is None else dx_out
    _chk2d(dx, "dx", BF16, cols=Cc, rows=M)
    dh = torch.empty((M, J), dtype=BF16, device=xhat.device)
    _lib.check(_lib.lib().stg_ln_bwd_down_xhat(_p(dy), _ld(dy), _p(xhat), _ld(xhat), _p(rstd), _p(add_to),
                                               _ld(add_to) if add_to is not None else 0, _p(dx), _ld(dx), _p(wt), _ld(wt),
                                               _p(row_scale), int(rs_outer), int(rs_inner), _p(dh), _ld(dh), M, Cc, J, _stream()),
               "stg_ln_bwd_down_xhat")
    return dx, dh


@_family("ln_bwd", lambda dy, xhat, rstd, add_to=None: (xhat.shape[1], _nb(dy, xhat, add_to) + 2.0 * xhat.numel(), 0.0))
def layernorm_bwd_xhat(dy, xhat, rstd, add_to=None):
    """LayerNorm backward wrt the input from the NORMALISED row (gamma == 1): dx bf16 = rstd (dy - mean(dy) - xhat mean(dy xhat)) (+ add_to)."""
    _chk2d(xhat, "xhat", BF16)
    M, Cc = xhat.shape
    _chk2d(dy, "dy", BF16, cols=Cc, rows=M)
    _chk1d(rstd, "rstd", F32, M)
    if add_to is not None:
        _chk2d(add_to, "add_to", BF16, cols=Cc, rows=M)
    dx = torch.empty((M, Cc), dtype=BF16, device=xhat.device)
    _lib.check(_lib.lib().stg_layernorm_bwd_xhat(_p(dy), _ld(dy), _p(xhat), _ld(xhat), _p(rstd), _p(add_to),
                                                 _ld(add_to) if add_to is not None else 0, _p(dx), _ld(dx), M, Cc, _stream()),
               "stg_layernorm_bwd_xhat")
    return dx


def _chk_flat(t, name, dtype=BF16):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected contiguous {dtype} GPU tensor")


def gate_fwd(h, r, gate):
    _chk_flat(h, "h"); _chk_flat(r, "r"); _chk_flat(gate, "gate", F32)
    if h.shape != r.shape or gate.numel() != 1:
        raise RuntimeError("gate_fwd: shape mismatch")
    out = torch.empty_like(h)
    _lib.check(_lib.lib().stg_gate_fwd(_p(h), _p(r), _p(gate), _p(out), h.numel(), _stream()), "stg_gate_fwd")
    return out


@_family("elementwise", lambda h0, *a: ("gate_fwd2", 6.0 * _nb(h0), 0.0))
def gate_fwd2(h0, r0, g0, h1, r1, g1):
    """gate_fwd on two problems (sizes may differ), one launch: returns (out0, out1)."""
    for t in (h0, r0, h1, r1):
        _chk_flat(t, "gate_fwd2 operand")
    _chk_flat(g0, "gate", F32); _chk_flat(g1, "gate", F32)
    if h0.shape != r0.shape or h1.shape != r1.shape or g0.numel() != 1 or g1.numel() != 1:
        raise RuntimeError("gate_fwd2: shape mismatch")
    o0, o1 = torch.empty_like(h0), torch.empty_like(h1)
    _lib.check(_lib.lib().stg_gate_fwd2n(_p(h0), _p(r0), _p(g0), _p(o0), h0.numel(), _p(h1), _p(r1), _p(g1), _p(o1), h1.numel(), _stream()), "stg_gate_fwd2n")
    return o0, o1


@_family("elementwise", lambda d0, *a: ("gate_bwd2", 6.0 * _nb(d0), 0.0))
def gate_bwd2(d0, r0, g0, dg0, d1, r1, g1, dg1):
    """gate_bwd on two problems (sizes may differ), one launch: returns (dr0, dr1); dgate0 / dgate1 accumulate."""
    for t in (d0, r0, d1, r1):
        _chk_flat(t, "gate_bwd2 operand")
    for t in (g0, g1, dg0, dg1):
        _chk_flat(t, "gate", F32)
    if d0.shape != r0.shape or d1.shape != r1.shape or any(t.numel() != 1 for t in (g0, g1, dg0, dg1)):
        raise RuntimeError("gate_bwd2: shape mismatch")
    o0, o1 = torch.empty_like(d0), torch.empty_like(d1)
    _lib.check(_lib.lib().stg_gate_bwd2n(_p(d0), _p(r0), _p(g0), _p(o0), _p(dg0), d0.numel(), _p(d1), _p(r1), _p(g1), _p(o1), _p(dg1), d1.numel(),
                                         _stream()), "stg_gate_bwd2n")
    return o0, o1


@_family("elementwise", lambda a0, *a, **kw: ("add3_mul2", 10.0 * _nb(a0), 0.0))
def add3_mul2(a0, b0, c0, z0, a1, b1, c1, z1, outs=None):
    """add3_mul on two problems (sizes may differ), one launch: returns (out0, out1).  c0 = c1 = None: (a + b) * z."""
    if (c0 is None) != (c1 is None):
        raise RuntimeError("add3_mul2: c0 and c1 are given together or not at all")
    if outs is None:
        outs = (torch.empty_like(a0), torch.empty_like(a1))
    for ref, ts in ((a0, (a0, b0, c0, z0, outs[0])), (a1, (a1, b1, c1, z1, outs[1]))):
        for t in ts:
            if t is None:
                continue
            _chk_flat(t, "add3_mul2 operand")
            if t.shape != ref.shape:
                raise RuntimeError("add3_mul2: shape mismatch")
    _lib.check(_lib.lib().stg_add3_mul2n(_p(a0), _p(b0), _p(c0), _p(z0), _p(outs[0]), a0.numel(), _p(a1), _p(b1), _p(c1), _p(z1), _p(outs[1]),
                                         a1.numel(), _stream()), "stg_add3_mul2n")
    return outs[0], outs[1]


def gate_bwd(dout, r, gate, dgate):
    _chk_flat(dout, "dout"); _chk_flat(r, "r"); _chk_flat(gate, "gate", F32); _chk_flat(dgate, "dgate", F32)
    if dout.shape != r.shape or gate.numel() != 1 or dgate.numel() != 1:
        raise RuntimeError("gate_bwd: shape mismatch")
    dr = torch.empty_like(dout)
    _lib.check(_lib.lib().stg_gate_bwd(_p(dout), _p(r), _p(gate), _p(dr), _p(dgate), dout.numel(), _stream()), "stg_gate_bwd")
    return dr


@_family("elementwise", lambda a, b, c=None, out=None: ("add", (3.0 if c is None else 4.0) * _nb(a), 0.0))
def add(a, b, c=None, out=None):
    """out = a + b (+ c), bf16; `out` may alias an input (element-wise)."""
    _chk_flat(a, "a"); _chk_flat(b, "b")
    if a.shape != b.shape:
        raise RuntimeError("add: shape mismatch")
    if c is not None:
        _chk_flat(c, "c")
        if c.shape != a.shape:
            raise RuntimeError("add: shape mismatch")
    if out is None:
        out = torch.empty_like(a)
    else:
        _chk_flat(out, "out")
        if out.shape != a.shape:
            raise RuntimeError("add: shape mismatch")
    _lib.check(_lib.lib().stg_add(_p(a), _p(b), _p(c), _p(out), a.numel(), _stream()), "stg_add")
    return out


def add3_mul(a, b, c, z, out=None):
    """(a + b + c) * z, bf16, one pass."""
    for t, n in ((a, "a"), (b, "b"), (c, "c"), (z, "z")):
        _chk_flat(t, n)
        if t.shape != a.shape:
            raise RuntimeError("add3_mul: shape mismatch")
    if out is None:
        out = torch.empty_like(a)
    else:
        _chk_flat(out, "out")
        if out.shape != a.shape:
            raise RuntimeError("add3_mul: shape mismatch")
    _lib.check(_lib.lib().stg_add3_mul(_p(a), _p(b), _p(c), _p(z), _p(out), a.numel(), _stream()), "stg_add3_mul")
    return out


@_family("elementwise", lambda dh, z, out=None: ("act_bwd", 2.0 * _nb(dh) + _nb(z), 0.0))
def act_bwd(dh, z, out=None):
    """dz = dh * z, z = the activation derivative saved by gemm_nt(want_dact=True)."""
    _chk_flat(dh, "dh"); _chk_flat(z, "z")
    if dh.shape != z.shape:
        raise RuntimeError("act_bwd: shape mismatch")
    if out is None:
        dz = torch.empty_like(dh)
    else:
        _chk_flat(out, "out")
        if out.shape != dh.shape:
            raise RuntimeError("act_bwd: shape mismatch")
        dz = out
    _lib.check(_lib.lib().stg_act_bwd(_p(dh), _p(z), _p(dz), dh.numel(), _stream()), "stg_act_bwd")
    return dz


@_family_io("elementwise")
def mul_mask(a, mask):
    _chk_flat(a, "a"); _chk_flat(mask, "mask", F32)
    if a.numel() != mask.numel():
        raise RuntimeError("mul_mask: shape mismatch")
    out = torch.empty_like(a)
    _lib.check(_lib.lib().stg_mul_mask(_p(a), _p(mask), _p(out), a.numel(), _stream()), "stg_mul_mask")
    return out


@_family("mha_bwd", lambda g, *a, **kw: tuple(x if i == 0 else 2 * x for i, x in enumerate(_attn_cost(g, 5, 3, 5))))
def mha_bwd_pair_merged(g, p0, p1):
    """The backward of a cross-modal pair as one pass per modality (stg_mha_bwd_pair_merged): p0 = (X, Y, O0, lse0, dO0) is direction 0 (queries X,
    keys = values Y), p1 = (Y, X, O1, lse1, dO1) its mirror image -- the same two tensors.  Returns (G_X, G_Y) = the whole gradients of X and Y
    (dQ of a tensor's own direction + dK + dV of the other).  Accounted like the two-direction mha_bwd_pair it replaces (same algorithmic work)."""
    (X, Y, O0, l0, d0), (Y1, X1, O1, l1, d1) = p0, p1
    if X.data_ptr() != X1.data_ptr() or Y.data_ptr() != Y1.data_ptr():
        raise RuntimeError("mha_bwd_pair_merged: the two directions must be the same two tensors with their roles swapped")
    for t, name in ((d0, "dO0"), (d1, "dO1")):
        _chk2d(t, name, BF16)
    if _ld(d0) != _ld(d1):
        raise RuntimeError("mha_bwd_pair_merged: dO0 and dO1 must share one leading dimension")
    G0 = torch.empty((X.shape[0], g.H * g.D), dtype=BF16, device=X.device)
    G1 = torch.empty((Y.shape[0], g.H * g.D), dtype=BF16, device=X.device)
    dl0 = torch.empty((g.P, g.H, g.n), dtype=F32, device=X.device)
    dl1 = torch.empty((g.P, g.H, g.n), dtype=F32, device=X.device)
    a0, a1 = _mha_fill(g, X, Y, Y, O0, l0), _mha_fill(g, Y, X, X, O1, l1)
    _lib.check(_lib.lib().stg_mha_bwd_pair_merged(C.byref(a0), _p(d0), _p(G0), _p(dl0), C.byref(a1), _p(d1), _p(G1), _p(dl1), _ld(d0), _ld(G0), _stream()),
               "stg_mha_bwd_pair_merged")
    return G0, G1


# ------------------------------------------------------------------------------------------------ the ViT blocks' cross-modal pair on small frames (xsmall.hip)
def xsmall_supported(nv, na, D):
    return bool(_lib.lib().stg_xsmall_supported(int(nv), int(na), int(D)))


class XsGeom:
    """P frames of nv video + na audio rows of width D (CLIP_AVE.py:386-398: 197 + 49 tokens, adapter width 48 for ViT-B/16)."""

    def __init__(self, P, nv, na, D, scale=1.0):
        self.P, self.nv, self.na, self.D, self.scale = int(P), int(nv), int(na), int(D), float(scale)
        if not xsmall_supported(self.nv, self.na, self.D):
            raise RuntimeError(f"xsmall: unsupported geometry nv={nv} na={na} D={D}")
        self.n, self.H = self.nv, 1                           # (for the family accounting)


def _xs_fill(g, Xv, Xa, Ov, Oa, lv, la):
    for t, name, rows in ((Xv, "Xv", g.P * g.nv), (Xa, "Xa", g.P * g.na), (Ov, "Ov", g.P * g.nv), (Oa, "Oa", g.P * g.na)):
        _chk2d(t, name, BF16)
        if t.shape[0] < rows or t.shape[1] < g.D:
            raise RuntimeError(f"xsmall {name}: needs >= {rows} rows x {g.D} columns, got {tuple(t.shape)}")
    for t, n in ((lv, g.P * g.nv), (la, g.P * g.na)):
        if t.dtype != F32 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
            raise RuntimeError("xsmall: lse must be contiguous fp32 [P, n] GPU tensors")
    a = _lib.XsmallArgs()
    a.Xv, a.Xa, a.ldv, a.lda = _p(Xv), _p(Xa), _ld(Xv), _ld(Xa)
    a.Ov, a.Oa, a.ldov, a.ldoa = _p(Ov), _p(Oa), _ld(Ov), _ld(Oa)
    a.lse_v, a.lse_a = _p(lv), _p(la)
    a.P, a.nv, a.na, a.D, a.scale = g.P, g.nv, g.na, g.D, g.scale
    return a


def _xs_cost(g, nin, nout, nmm):
    by = 2.0 * g.P * (g.nv + g.na) * g.D * (nin + nout)
    return (g.D, g.nv, g.na), by, 2.0 * nmm * g.P * g.nv * g.na * g.D


@_family("xattn_fwd", lambda g, *a, **kw: _xs_cost(g, 1, 1, 4))
def xsmall_fwd(g, Xv, Xa):
    """Both directions of the pair, one launch: ((Ov, lse_v), (Oa, lse_a)) with O = softmax(scale X Xother^T) Xother per frame (lse: log2 domain)."""
    Ov = torch.empty((Xv.shape[0], g.D), dtype=BF16, device=Xv.device)
    Oa = torch.empty((Xa.shape[0], g.D), dtype=BF16, device=Xa.device)
    lv = torch.empty((g.P, g.nv), dtype=F32, device=Xv.device)
    la = torch.empty((g.P, g.na), dtype=F32, device=Xv.device)
    a = _xs_fill(g, Xv, Xa, Ov, Oa, lv, la)
    _lib.check(_lib.lib().stg_xsmall_fwd(C.byref(a), _stream()), "stg_xsmall_fwd")
    return (Ov, lv), (Oa, la)


@_family("xattn_bwd", lambda g, *a, **kw: _xs_cost(g, 3, 1, 10))
def xsmall_bwd(g, Xv, Xa, Ov, Oa, lv, la, dOv, dOa):
    """The pair's backward, one launch: (Gv, Ga) = the whole gradients of Xv and Xa (dQ of a tensor's own direction + dK + dV of the other)."""
    for t, name in ((dOv, "dOv"), (dOa, "dOa")):
        _chk2d(t, name, BF16)
    Gv = torch.empty((Xv.shape[0], g.D), dtype=BF16, device=Xv.device)
    Ga = torch.empty((Xa.shape[0], g.D), dtype=BF16, device=Xa.device)
    a = _xs_fill(g, Xv, Xa, Ov, Oa, lv, la)
    _lib.check(_lib.lib().stg_xsmall_bwd(C.byref(a), _p(dOv), _p(dOa), _ld(dOv), _ld(dOa), _p(Gv), _p(Ga), _ld(Gv), _ld(Ga), _stream()), "stg_xsmall_bwd")
    return Gv, Ga


@_family_io("patch_embed")
def im2col_patch(x, p, Kpad):
    """x: [B, Cin, T, H, W] fp32/bf16 contiguous -> [B*T*(H/p)*(W/p), Kpad] bf16."""
    if x.dim() != 5 or not x.is_contiguous() or not x.is_cuda or x.dtype not in (BF16, F32):
        raise RuntimeError("im2col_patch: expected contiguous 5-D fp32/bf16 GPU tensor")
    B, Cin, T, H, W = x.shape
    out = torch.empty((B * T * (H // p) * (W // p), Kpad), dtype=BF16, device=x.device)
    _lib.check(_lib.lib().stg_im2col_patch(_p(x), STG_BF16 if x.dtype == BF16 else STG_F32, _p(out), B, Cin, T, H, W, p, Kpad,
                                           _stream()), "stg_im2col_patch")
    return out


@_family_io("cast")
def cast_bf16(w, transpose=False, pad_to=8):
    """fp32 [R, C] -> bf16 [R, C'] or (transpose) [C, R'], trailing dim zero-padded to a multiple of `pad_to`."""
    _chk_flat(w, "w", F32)
    w2 = w.reshape(w.shape[0], -1) if w.dim() > 1 else w.reshape(1, -1)
    R, Cc = w2.shape
    inner = R if transpose else Cc
    ld = (inner + pad_to - 1) // pad_to * pad_to
    out = torch.empty((Cc, ld) if transpose else (R, ld), dtype=BF16, device=w.device)
    _lib.check(_lib.lib().stg_cast_bf16(_p(w2), _p(out), R, Cc, 1 if transpose else 0, ld, _stream()), "stg_cast_bf16")
    return out


def cast_desc_table(entries, device):
    """entries: [(src data_ptr, off, offT, R, C, ld, ldT)] -> uint8 GPU tensor holding the stg_cast_desc array."""
    arr = (_lib.CastDesc * len(entries))()
    for i, (ptr, off, offT, R, Cc, ld, ldT) in enumerate(entries):
        arr[i].in_, arr[i].off, arr[i].offT, arr[i].R, arr[i].C, arr[i].ld, arr[i].ldT = ptr, off, offT, R, Cc, ld, ldT
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


@_family_io("cast")
def cast_bf16_multi(desc, n, max_elems, arena):
    """One launch casting n fp32 matrices (device-resident descriptor table from cast_desc_table) into the zeroed bf16 arena."""
    if desc.dtype != torch.uint8 or not desc.is_cuda or desc.numel() != n * C.sizeof(_lib.CastDesc):
        raise RuntimeError("cast_bf16_multi: bad descriptor table")
    _chk_flat(arena, "arena", BF16)
    _lib.check(_lib.lib().stg_cast_bf16_multi(_p(desc), int(n), int(max_elems), _p(arena), _stream()), "stg_cast_bf16_multi")


@_family_io("elementwise")
def add_temporal(x, emb, B, T, N):
    """x fp32 [B*T*N, C] (a contiguous row slice) += emb fp32 [T, C] broadcast over clips and tokens, in place."""
    _chk_flat(x, "x", F32)
    Cc = x.shape[-1]
    if x.shape[0] != B * T * N or emb.dtype != F32 or not emb.is_contiguous() or emb.numel() != T * Cc:
        raise RuntimeError("add_temporal: shape mismatch")
    _lib.check(_lib.lib().stg_add_temporal(_p(x), _p(emb), B, T, N, Cc, _stream()), "stg_add_temporal")
    return x


def adam_desc_table(entries, device, host=None):
    """entries: [(p, g, m, v, per-tensor state data_ptrs, numel, group)] -> (uint8 GPU tensor holding the stg_adam_desc array, its
    pinned host source).  `host`: a pinned uint8 buffer to stage through (nothing is allocated on the host then -- a pinned
    allocation is not allowed while a HIP graph is being captured); it must stay alive and unchanged as long as a captured graph
    replays the copy."""
    arr = (_lib.AdamDesc * len(entries))()
    for i, (pp, gp, mp, vp, sp, n, grp) in enumerate(entries):
        arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].st, arr[i].n, arr[i].group = pp, gp, mp, vp, sp, n, grp
    nbytes = C.sizeof(arr)
    if host is None:
        host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    if host.numel() < nbytes or not host.is_pinned():
        raise RuntimeError("adam_desc_table: the staging buffer must be pinned and large enough")
    C.memmove(host.data_ptr(), C.addressof(arr), nbytes)
    dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dev.copy_(host[:nbytes], non_blocking=True)
    return dev, host


@_family_io("optimizer")
def adam_multi(desc, n, max_elems, hyper):
    """One optimizer step of torch.optim.Adam's arithmetic over n fp32 tensors (device-resident descriptor table from adam_desc_table);
    hyper: fp64 [groups, 8] = lr, beta1, beta2, eps, weight_decay, -, -, -; each tensor's own {step, lr / bc1, sqrt(bc2)} triple (the
    descriptor's `st`) is advanced by the call."""
    if desc.dtype != torch.uint8 or not desc.is_cuda or desc.numel() != n * C.sizeof(_lib.AdamDesc):
        raise RuntimeError("adam_multi: bad descriptor table")
    if hyper.dtype != torch.float64 or hyper.dim() != 2 or hyper.shape[1] != 8 or not hyper.is_contiguous() or not hyper.is_cuda:
        raise RuntimeError("adam_multi: hyper must be a contiguous fp64 [groups, 8] device tensor")
    _lib.check(_lib.lib().stg_adam_multi(_p(desc), int(n), int(max_elems), _p(hyper), int(hyper.shape[0]), _stream()), "stg_adam_multi")


@_family_io("cast")
def cast_f32(x):
    _chk_flat(x, "x")
    out = torch.empty(x.shape, dtype=F32, device=x.device)
    _lib.check(_lib.lib().stg_cast_f32(_p(x), _p(out), x.numel(), _stream()), "stg_cast_f32")
    return out


@_family_io("head_small")
def meanpool_fwd(x, G, n, out=None, out_dtype=BF16):
    """x: [G*n, C] bf16 contiguous -> [G, C]; `out` may be a column slice of a wider row-major buffer."""
    _chk_flat(x, "x")
    Cc = x.shape[-1]
    if x.numel() != G * n * Cc:
        raise RuntimeError("meanpool_fwd: shape mismatch")
    if out is None:
        out = torch.empty((G, Cc), dtype=out_dtype, device=x.device)
    _chk2d(out, "out", out.dtype, cols=Cc, rows=G)
    _lib.check(_lib.lib().stg_meanpool_fwd(_p(x), _p(out), STG_BF16 if out.dtype == BF16 else STG_F32, _ld(out), G, n, Cc,
                                           _stream()), "stg_meanpool_fwd")
    return out


@_family_io("head_small")
def meanpool_bwd(dout, G, n, out=None):
    _chk2d(dout, "dout", BF16, rows=G)
    Cc = dout.shape[1]
    din = torch.empty((G * n, Cc), dtype=BF16, device=dout.device) if out is None else out
    _chk_flat(din, "din")
    if din.numel() != G * n * Cc:
        raise RuntimeError("meanpool_bwd: bad out")
    _lib.check(_lib.lib().stg_meanpool_bwd(_p(dout), _ld(dout), _p(din), G, n, Cc, _stream()), "stg_meanpool_bwd")
    return din


@_family_io("elementwise")
def bias_gather(table, index, out=None):
    """table fp32 [L, H], index int64 [nn] -> fp32 [H, nn]."""
    _chk_flat(table, "table", F32); _chk_flat(index, "index", torch.int64)
    L, H = table.shape
    nn = index.numel()
    if out is None:
        out = torch.empty((H, nn), dtype=F32, device=table.device)
    _chk_flat(out, "out", F32)
    if out.numel() != H * nn:
        raise RuntimeError("bias_gather: bad out")
    _lib.check(_lib.lib().stg_bias_gather(_p(table), _p(index), _p(out), L, H, nn, _stream()), "stg_bias_gather")
    return out


@_family_io("elementwise")
def bias_scatter(dbias, index, dtable):
    _chk_flat(dbias, "dbias", F32); _chk_flat(index, "index", torch.int64); _chk_flat(dtable, "dtable", F32)
    L, H = dtable.shape
    nn = index.numel()
    if dbias.numel() != H * nn:
        raise RuntimeError("bias_scatter: shape mismatch")
    _lib.check(_lib.lib().stg_bias_scatter(_p(dbias), _p(index), _p(dtable), L, H, nn, _stream()), "stg_bias_scatter")


class AttnGeom:
    """Addressing of one attention call: row(p, i) = (p // G) * outer + map[(p % G) * n + i] (map None = identity)."""

    def __init__(self, P, H, n, D, G=1, outer=None, map_q=None, n_kv=None, outer_kv=None, map_kv=None, scale=1.0,
                 bias=None, bias_div=1, bias_mod=1, mask=None, window=None, temporal=None):
        """window=(Himg, Wimg, ws, shift) / temporal=N select the arithmetic maps (no table); else map_q/map_kv tables."""
        self.kind, self.mp = 0, (0, 0, 0, 0)
        if window is not None:
            self.kind, self.mp = 1, tuple(int(x) for x in window)
        elif temporal is not None:
            self.kind, self.mp = 2, (int(temporal), 0, 0, 0)
        self.P, self.H, self.n, self.D, self.G = int(P), int(H), int(n), int(D), int(G)
        self.n_kv = int(n if n_kv is None else n_kv)
        self.outer = int(G * n if outer is None else outer)
        self.outer_kv = int((self.outer if n_kv is None else G * self.n_kv) if outer_kv is None else outer_kv)
        self.map_q = map_q
        self.map_kv = map_q if (map_kv is None and n_kv is None) else map_kv
        self.scale = float(scale)
        self.bias, self.bias_div, self.bias_mod, self.mask = bias, int(bias_div), int(bias_mod), mask


def _attn_check_rows(t, name, g, n_tok, outer, amap, dev):
    _chk2d(t, name, BF16)
    if t.shape[1] < g.H * g.D:
        raise RuntimeError(f"{name}: needs >= H*D = {g.H * g.D} columns, got {t.shape[1]}")
    if g.P % g.G != 0:
        raise RuntimeError(f"{name}: P={g.P} must be a multiple of G={g.G}")
    need_rows = (g.P // g.G) * outer
    if amap is not None:
        if amap.dtype != torch.int32 or not amap.is_cuda or not amap.is_contiguous() or amap.numel() != g.G * n_tok:
            raise RuntimeError(f"{name}: map must be a contiguous int32 GPU tensor of {g.G * n_tok} entries")
    elif outer < g.G * n_tok:
        raise RuntimeError(f"{name}: outer too small")
    if t.shape[0] < need_rows:
        raise RuntimeError(f"{name}: needs >= {need_rows} rows, got {t.shape[0]}")


def _attn_fill(a, g, Q, K, V, O, lse):
    a.Q, a.ldq = _p(Q), _ld(Q)
    a.K, a.ldk = _p(K), _ld(K)
    a.V, a.ldv = _p(V), _ld(V)
    a.O, a.ldo = _p(O), _ld(O)
    a.lse = _p(lse)
    a.map_q, a.map_kv = _p(g.map_q), _p(g.map_kv)
    a.outer_q, a.outer_kv, a.G = g.outer, g.outer_kv, g.G
    a.map_kind, (a.map_a, a.map_b, a.map_c, a.map_d) = g.kind, g.mp
    a.P, a.H, a.n, a.n_kv, a.D = g.P, g.H, g.n, g.n_kv, g.D
    a.scale = g.scale
    if g.bias is not None:
        if g.bias.dtype != F32 or not g.bias.is_contiguous() or g.bias.numel() != g.bias_mod * g.H * g.n * g.n_kv:
            raise RuntimeError("attention: bias must be contiguous fp32 [bias_mod, H, n, n_kv]")
        a.bias, a.bias_div, a.bias_mod = _p(g.bias), g.bias_div, g.bias_mod
    if g.mask is not None:
        if g.mask.dtype != F32 or not g.mask.is_contiguous() or g.mask.numel() != g.G * g.n * g.n_kv:
            raise RuntimeError("attention: mask must be contiguous fp32 [G, n, n_kv]")
        a.mask = _p(g.mask)


def _attn_cost(g, nin, nout, nmm):
    """Generic / frame-global / ViT attention: nin input + nout output tensors of P * n (or n_kv) rows x H * D bf16; nmm n x n_kv x D products."""
    P, H, n, D = g.P, g.H, g.n, g.D
    nkv = getattr(g, "n_kv", n)
    return (H * D, n), 2.0 * P * H * D * (nin * 0.5 * (n + nkv) + nout * n), 2.0 * nmm * P * H * n * nkv * D


@_family("attn_fwd", lambda g, *a, **kw: _attn_cost(g, 3, 1, 2))
def attn_fwd(g, Q, K, V, out=None, want_lse=True):
    """Q/K/V: 2-D bf16 (possibly column-slice views of a fused qkv buffer).  Returns (O, lse)."""
    dev = Q.device
    _attn_check_rows(Q, "Q", g, g.n, g.outer, g.map_q, dev)
    _attn_check_rows(K, "K", g, g.n_kv, g.outer_kv, g.map_kv, dev)
    _attn_check_rows(V, "V", g, g.n_kv, g.outer_kv, g.map_kv, dev)
    if out is None:
        out = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=dev)
    _attn_check_rows(out, "O", g, g.n, g.outer, g.map_q, dev)
    lse = torch.empty((g.P, g.H, g.n), dtype=F32, device=dev) if want_lse else None
    a = _lib.AttnArgs()
    _attn_fill(a, g, Q, K, V, out, lse)
    _lib.check(_lib.lib().stg_attn_fwd(C.byref(a), _stream()), "stg_attn_fwd")
    return out, lse


@_family("xattn_fwd", lambda g0, Q0, K0, V0, g1, *a: tuple(x if i == 0 else 2 * x for i, x in enumerate(_attn_cost(g0, 2, 1, 2))))
def attn_fwd2(g0, Q0, K0, V0, g1, Q1, K1, V1):
    """Two attn_fwd problems (the two directions of a cross-modal pair) in one call; the frame-global kernels share a launch.
    Returns ((O0, lse0), (O1, lse1))."""
    outs, args = [], []
    for g, Q, K_, V in ((g0, Q0, K0, V0), (g1, Q1, K1, V1)):
        dev = Q.device
        _attn_check_rows(Q, "Q", g, g.n, g.outer, g.map_q, dev)
        _attn_check_rows(K_, "K", g, g.n_kv, g.outer_kv, g.map_kv, dev)
        _attn_check_rows(V, "V", g, g.n_kv, g.outer_kv, g.map_kv, dev)
        out = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=dev)
        lse = torch.empty((g.P, g.H, g.n), dtype=F32, device=dev)
        a = _lib.AttnArgs()
        _attn_fill(a, g, Q, K_, V, out, lse)
        outs.append((out, lse)); args.append(a)
    _lib.check(_lib.lib().stg_attn_fwd2(C.byref(args[0]), C.byref(args[1]), _stream()), "stg_attn_fwd2")
    return outs[0], outs[1]


@_family("xattn_bwd", lambda p0, p1: tuple(x if i == 0 else 2 * x for i, x in enumerate(_attn_cost(p0[0], 4, 2, 5))))
def attn_bwd2(p0, p1):
    """Two shared-K/V attn_bwd problems in one call: p = (g, Q, KV, O, lse, dO).  Returns ((dQ0, dKV0), (dQ1, dKV1))."""
    res, args, keep = [], [], []
    for g, Q, KV, O, lse, dO in (p0, p1):
        dev = Q.device
        for t, name, nt, outer, mp in ((Q, "Q", g.n, g.outer, g.map_q), (KV, "K", g.n_kv, g.outer_kv, g.map_kv),
                                       (O, "O", g.n, g.outer, g.map_q), (dO, "dO", g.n, g.outer, g.map_q)):
            _attn_check_rows(t, name, g, nt, outer, mp, dev)
        if lse.dtype != F32 or lse.numel() != g.P * g.H * g.n:
            raise RuntimeError("attn_bwd2: bad lse")
        dQ = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=dev)
        dK = torch.empty((KV.shape[0], g.H * g.D), dtype=BF16, device=dev)
        delta = torch.empty((g.P, g.H, g.n), dtype=F32, device=dev)
        b = _lib.AttnBwdArgs()
        _attn_fill(b.f, g, Q, KV, KV, O, lse)
        b.dO, b.lddo = _p(dO), _ld(dO)
        b.dQ, b.lddq = _p(dQ), _ld(dQ)
        b.dK, b.lddk = _p(dK), _ld(dK)
        b.delta = _p(delta)
        res.append((dQ, dK)); args.append(b); keep.append(delta)
    _lib.check(_lib.lib().stg_attn_bwd2(C.byref(args[0]), C.byref(args[1]), _stream()), "stg_attn_bwd2")
    return res[0], res[1]


def _xpair_args(p0, p1):
    args = []
    for g, Q, KV, O, lse, dO in (p0, p1):
        dev = Q.device
        for t, name, nt, outer, mp in ((Q, "Q", g.n, g.outer, g.map_q), (KV, "K", g.n_kv, g.outer_kv, g.map_kv),
                                       (O, "O", g.n, g.outer, g.map_q), (dO, "dO", g.n, g.outer, g.map_q)):
            _attn_check_rows(t, name, g, nt, outer, mp, dev)
        if lse.dtype != F32 or lse.numel() != g.P * g.H * g.n:
            raise RuntimeError("xattn_pair_bwd: bad lse")
        b = _lib.AttnBwdArgs()
        _attn_fill(b.f, g, Q, KV, KV, O, lse)
        b.dO, b.lddo = _p(dO), _ld(dO)
        args.append(b)
    return args


def xattn_pair_bwd_supported(p0, p1):
    """p = (g, Q, KV, O, lse, dO) per direction: can the pair's backward run as one merged pass per modality (stg_xattn_pair_bwd)?"""
    a0, a1 = _xpair_args(p0, p1)
    return bool(_lib.lib().stg_xattn_pair_bwd_supported(C.byref(a0.f), C.byref(a1.f)))


@_family("xattn_bwd", lambda p0, p1, join=None, outs=None, gates=None: (("pair",) + tuple(_attn_cost(p0[0], 4, 2, 5)[0]), 2.0 * _attn_cost(p0[0], 5 if join is None else 7, 1, 5)[1],
                                                                         2.0 * _attn_cost(p0[0], 4, 2, 5)[2]))
def xattn_pair_bwd(p0, p1, join=None, outs=None, gates=None):
    """Backward of a frame-global cross-modal pair, one pass per modality: p0 = (g, h_v, h_a, r_v, lse_v, d r_v), p1 = the mirror image.
    Returns (G_v, G_a): the complete gradients of h_v / h_a through both directions (what attn_bwd2's dQ_0 + dKV_1 and dQ_1 + dKV_0 sum to).
    join = (dX_v, Z_v, dX_a, Z_a): returns ((dX_v + G_v) * Z_v, (dX_a + G_a) * Z_a) instead -- the join of the adapters' backward (add3_mul2)
    inside the same launch; outs = destination tensors.  gates = (gate_v, gate_a, dgate_v, dgate_a) (round 6b): the last element of p0 / p1 is then
    d(h') of h' = h + gate r, the kernels apply the gate themselves (bit-identical to gate_bwd2 first) and accumulate dgate += <d(h'), r>."""
    a0, a1 = _xpair_args(p0, p1)
    g0, g1 = p0[0], p1[0]
    dev = p0[1].device
    if outs is None:
        outs = (torch.empty((p0[1].shape[0], g0.D), dtype=BF16, device=dev), torch.empty((p1[1].shape[0], g1.D), dtype=BF16, device=dev))
    G0, G1 = outs
    for t, ref in ((G0, p0[1]), (G1, p1[1])):
        _chk2d(t, "G", BF16, cols=g0.D, rows=ref.shape[0])
    if _ld(G0) != _ld(G1):
        raise RuntimeError("xattn_pair_bwd: the two outputs must share one leading dimension")
    nb = int(_lib.lib().stg_xattn_pair_bwd_ws_bytes(g0.P, g0.n, g1.n, g0.D))
    ws = torch.empty((nb + 15) // 16 * 4, dtype=F32, device=dev)
    if join is None and gates is None:
        _lib.check(_lib.lib().stg_xattn_pair_bwd(C.byref(a0), C.byref(a1), _p(G0), _p(G1), _ld(G0), _p(ws), ws.numel() * 4, _stream()),
                   "stg_xattn_pair_bwd")
        return G0, G1
    if join is None:
        _lib.check(_lib.lib().stg_xattn_pair_bwd_gate(C.byref(a0), C.byref(a1), _p(G0), _p(G1), _ld(G0), None, None, 0, None, None, 0, _p(gates[0]), _p(gates[1]),
                                                      _p(gates[2]), _p(gates[3]), _p(ws), ws.numel() * 4, _stream()), "stg_xattn_pair_bwd_gate")
        return G0, G1
    dx0, z0, dx1, z1 = join
    for t, ref in ((dx0, p0[1]), (z0, p0[1]), (dx1, p1[1]), (z1, p1[1])):
        _chk2d(t, "join operand", BF16, cols=g0.D, rows=ref.shape[0])
    if _ld(dx0) != _ld(dx1) or _ld(z0) != _ld(z1):
        raise RuntimeError("xattn_pair_bwd: the join operands of the two modalities must share leading dimensions")
    if gates is not None:
        _lib.check(_lib.lib().stg_xattn_pair_bwd_gate(C.byref(a0), C.byref(a1), _p(G0), _p(G1), _ld(G0), _p(dx0), _p(dx1), _ld(dx0), _p(z0), _p(z1), _ld(z0),
                                                      _p(gates[0]), _p(gates[1]), _p(gates[2]), _p(gates[3]), _p(ws), ws.numel() * 4, _stream()), "stg_xattn_pair_bwd_gate")
        return G0, G1
    _lib.check(_lib.lib().stg_xattn_pair_bwd_join(C.byref(a0), C.byref(a1), _p(G0), _p(G1), _ld(G0), _p(dx0), _p(dx1), _ld(dx0), _p(z0), _p(z1),
                                                  _ld(z0), _p(ws), ws.numel() * 4, _stream()), "stg_xattn_pair_bwd_join")
    return G0, G1


@_family("xattn_fwd", lambda g0, Q0, K0, g1, Q1, K1, gate0, gate1: tuple(x if i == 0 else 2 * x for i, x in enumerate(_attn_cost(g0, 2, 2, 2))))
def xattn_fwd2_gate(g0, Q0, K0, g1, Q1, K1, gate0, gate1):
    """Forward of a frame-global cross-modal pair (K == V) with its gates in one launch: returns ((O0, lse0, X0), (O1, lse1, X1)) with
    X = Q + gate * O.  Needs xattn_pair_bwd_supported's geometry (check with xattn_pair_fwd_supported)."""
    outs, args = [], []
    for g, Q, K_, gate in ((g0, Q0, K0, gate0), (g1, Q1, K1, gate1)):
        dev = Q.device
        _attn_check_rows(Q, "Q", g, g.n, g.outer, g.map_q, dev)
        _attn_check_rows(K_, "K", g, g.n_kv, g.outer_kv, g.map_kv, dev)
        if gate.dtype != F32 or not gate.is_cuda or gate.numel() != 1:
            raise RuntimeError("xattn_fwd2_gate: a gate must be a one-element fp32 GPU tensor")
        out = torch.empty((Q.shape[0], g.D), dtype=BF16, device=dev)
        x = torch.empty((Q.shape[0], g.D), dtype=BF16, device=dev)
        lse = torch.empty((g.P, g.H, g.n), dtype=F32, device=dev)
        a = _lib.AttnArgs()
        _attn_fill(a, g, Q, K_, K_, out, lse)
        outs.append((out, lse, x)); args.append(a)
    _lib.check(_lib.lib().stg_xattn_fwd2_gate(C.byref(args[0]), C.byref(args[1]), _p(gate0), _p(gate1), _p(outs[0][2]), _p(outs[1][2]),
                                              _ld(outs[0][2]), _stream()), "stg_xattn_fwd2_gate")
    return outs[0], outs[1]


def xattn_pair_fwd_supported(g0, Q0, K0, g1, Q1, K1):
    """Can the pair's forward (and backward) run on the frame-global kernels as one mirror-image pair (stg_xattn_fwd2_gate / _pair_bwd)?"""
    args = []
    for g, Q, K_ in ((g0, Q0, K0), (g1, Q1, K1)):
        a = _lib.AttnArgs()
        _attn_fill(a, g, Q, K_, K_, Q, Q)             # O / lse only have to be non-null for the geometry check (never dereferenced)
        args.append(a)
    return bool(_lib.lib().stg_xattn_pair_bwd_supported(C.byref(args[0]), C.byref(args[1])))


@_family("attn_bwd", lambda g, *a, **kw: _attn_cost(g, 5, 3, 5))
def attn_bwd(g, Q, K, V, O, lse, dO, *, dQ=None, dK=None, dV=None, shared_kv=False, dbias=None):
    """Returns (dQ, dK, dV).  shared_kv: K and V are the same tensor -> dV is None and dK holds dK + dV."""
    dev = Q.device
    for t, name, nt, outer, mp in ((Q, "Q", g.n, g.outer, g.map_q), (K, "K", g.n_kv, g.outer_kv, g.map_kv),
                                   (V, "V", g.n_kv, g.outer_kv, g.map_kv), (O, "O", g.n, g.outer, g.map_q),
                                   (dO, "dO", g.n, g.outer, g.map_q)):
        _attn_check_rows(t, name, g, nt, outer, mp, dev)
    if lse.dtype != F32 or lse.numel() != g.P * g.H * g.n:
        raise RuntimeError("attn_bwd: bad lse")
    if dQ is None:
        dQ = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=dev)
    if dK is None:
        dK = torch.empty((K.shape[0], g.H * g.D), dtype=BF16, device=dev)
    if dV is None and not shared_kv:
        dV = torch.empty((V.shape[0], g.H * g.D), dtype=BF16, device=dev)
    _attn_check_rows(dQ, "dQ", g, g.n, g.outer, g.map_q, dev)
    _attn_check_rows(dK, "dK", g, g.n_kv, g.outer_kv, g.map_kv, dev)
    if dV is not None:
        _attn_check_rows(dV, "dV", g, g.n_kv, g.outer_kv, g.map_kv, dev)
    delta = torch.empty((g.P, g.H, g.n), dtype=F32, device=dev)
    b = _lib.AttnBwdArgs()
    _attn_fill(b.f, g, Q, K, V, O, lse)
    b.dO, b.lddo = _p(dO), _ld(dO)
    b.dQ, b.lddq = _p(dQ), _ld(dQ)
    b.dK, b.lddk = _p(dK), _ld(dK)
    if dV is not None:
        b.dV, b.lddv = _p(dV), _ld(dV)
    b.delta = _p(delta)
    if dbias is not None:
        if dbias.dtype != F32 or not dbias.is_contiguous() or g.bias is None or dbias.numel() != g.bias.numel():
            raise RuntimeError("attn_bwd: dbias must match bias")
        b.dbias = _p(dbias)
    _lib.check(_lib.lib().stg_attn_bwd(C.byref(b), _stream()), "stg_attn_bwd")
    return dQ, dK, dV


class WinGeom:
    """One W-MSA / SW-MSA call over the fused qkv buffer: P = images * nW windows of n = ws*ws <= 64 tokens, head dim 32.
    bm / bmT come from winattn_table (bias + shift mask, padded to 64 x 64, times log2 e)."""

    def __init__(self, images, H, Himg, Wimg, ws, shift, scale, bm, bmT, D=32):
        self.Himg, self.Wimg, self.ws, self.shift = int(Himg), int(Wimg), int(ws), int(shift)
        self.G = (self.Himg // self.ws) * (self.Wimg // self.ws)
        self.n = self.ws * self.ws
        self.P, self.H, self.scale, self.D = int(images) * self.G, int(H), float(scale), int(D)
        self.outer = self.Himg * self.Wimg
        self.bm, self.bmT = bm, bmT
        if bm is None or bmT is None:                        # no bias, no mask (round 6): the window-level cross-modal pair; widths 32 and 16
            if bm is not None or bmT is not None or self.n != 49 or self.D not in (16, 32):
                raise RuntimeError("winattn: the table-free form takes bm = bmT = None, 7 x 7 windows and head dim 16 or 32")
            self.Gt = 1
            if self.Himg % self.ws or self.Wimg % self.ws or not 0 <= self.shift < self.ws:
                raise RuntimeError("winattn: unsupported window geometry")
            return
        if self.D != 32:
            raise RuntimeError("winattn: head dim 16 only without a table")
        for t in (bm, bmT):
            if t.dtype != F32 or not t.is_cuda or not t.is_contiguous() or t.dim() != 4 or tuple(t.shape[1:]) != (self.H, 64, 64) \
                    or t.shape[0] not in (1, self.G):
                raise RuntimeError("winattn: bm / bmT must be contiguous fp32 [1 or nW, H, 64, 64] GPU tensors")
        if bm.shape != bmT.shape:
            raise RuntimeError("winattn: bm and bmT differ in shape")
        self.Gt = int(bm.shape[0])
        if self.n > 64 or self.Himg % self.ws or self.Wimg % self.ws or not 0 <= self.shift < self.ws:
            raise RuntimeError("winattn: unsupported window geometry")


def winattn_supported(n, hd, table=True):
    """table=False: the table-free form (no bias / mask): 7 x 7 windows, head dim 32 or 16."""
    if not table:
        return n == 49 and hd in (16, 32)
    return n <= 64 and hd == 32


def winattn_table(table, index, mask, n):
    """table fp32 [L, H], index int64 [n*n], mask fp32 [nW, n, n] or None -> (bm, bmT) fp32 [Gt, H, 64, 64]."""
    _chk_flat(table, "table", F32); _chk_flat(index, "index", torch.int64)
    L, H = table.shape
    if index.numel() != n * n or n > 64:
        raise RuntimeError("winattn_table: index must hold n*n entries, n <= 64")
    Gt = 1
    if mask is not None:
        _chk_flat(mask, "mask", F32)
        Gt = mask.shape[0]
        if mask.numel() != Gt * n * n:
            raise RuntimeError("winattn_table: mask must be [nW, n, n]")
    bm = torch.empty((Gt, H, 64, 64), dtype=F32, device=table.device)
    bmT = torch.empty_like(bm)
    _lib.check(_lib.lib().stg_winattn_table(_p(table), _p(index), _p(mask), _p(bm), _p(bmT), L, H, n, Gt, _stream()),
               "stg_winattn_table")
    return bm, bmT


def _win_fill(g, Q, K, V, O, lse):
    for t, name in ((Q, "Q"), (K, "K"), (V, "V"), (O, "O")):
        _chk2d(t, name, BF16)
        if t.shape[1] < g.H * g.D or t.shape[0] < (g.P // g.G) * g.outer:
            raise RuntimeError(f"winattn {name}: needs >= {(g.P // g.G) * g.outer} rows x {g.H * g.D} columns, got {tuple(t.shape)}")
    if not (_ld(Q) == _ld(K) == _ld(V)):
        raise RuntimeError("winattn: Q, K, V must share one leading dimension (slices of the fused qkv buffer)")
    a = _lib.WinAttnArgs()
    a.Q, a.K, a.V, a.ld = _p(Q), _p(K), _p(V), _ld(Q)
    a.O, a.ldo = _p(O), _ld(O)
    a.lse = _p(lse)
    a.bm, a.bmT, a.Gt = (_p(g.bm) if g.bm is not None else None), (_p(g.bmT) if g.bmT is not None else None), g.Gt
    a.outer = g.outer
    a.Himg, a.Wimg, a.ws, a.shift, a.G, a.n = g.Himg, g.Wimg, g.ws, g.shift, g.G, g.n
    a.P, a.H, a.D, a.scale = g.P, g.H, g.D, g.scale
    return a


@_family("winattn_fwd", lambda g, Q, *a, **kw: (g.H * g.D, 8.0 * (g.P // g.G) * g.outer * g.H * g.D, 4.0 * g.P * g.H * g.n * g.n * g.D))
def winattn_fwd(g, Q, K, V, out=None, want_lse=True):
    """Returns (O bf16 [rows, H*32], lse fp32 [P, H, 64] or None)."""
    if out is None:
        out = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=Q.device)
    lse = torch.empty((g.P, g.H, 64), dtype=F32, device=Q.device) if want_lse else None
    a = _win_fill(g, Q, K, V, out, lse)
    _lib.check(_lib.lib().stg_winattn_fwd(C.byref(a), _stream()), "stg_winattn_fwd")
    return out, lse


@_family("winattn_bwd", lambda g, Q, *a, **kw: (g.H * g.D, 16.0 * (g.P // g.G) * g.outer * g.H * g.D, 10.0 * g.P * g.H * g.n * g.n * g.D))
def winattn_bwd(g, Q, K, V, O, lse, dO, *, dQ, dK, dV):
    """dQ / dK / dV: column slices of one bf16 buffer (same leading dimension), written in place."""
    if lse is None or lse.dtype != F32 or lse.numel() != g.P * g.H * 64:
        raise RuntimeError("winattn_bwd: bad lse")
    rows = (g.P // g.G) * g.outer
    for t, name in ((dO, "dO"), (dQ, "dQ"), (dK, "dK")) + (((dV, "dV"),) if dV is not None else ()):      # dV None: K is V, dK <- dK + dV
        _chk2d(t, name, BF16)
        if t.shape[1] < g.H * g.D or t.shape[0] < rows:
            raise RuntimeError(f"winattn_bwd {name}: needs >= {rows} rows x {g.H * g.D} columns")
    if not (_ld(dQ) == _ld(dK) == (_ld(dV) if dV is not None else _ld(dK))):
        raise RuntimeError("winattn_bwd: dQ, dK, dV must share one leading dimension")
    a = _win_fill(g, Q, K, V, O, lse)
    _lib.check(_lib.lib().stg_winattn_bwd(C.byref(a), _p(dO), _ld(dO), _p(dQ), _p(dK), _p(dV), _ld(dQ), _stream()),
               "stg_winattn_bwd")
    return dQ, dK, dV


@_family("winattn_fwd", lambda g, hv, *a, **kw: ("pair", 2 * 10.0 * (g.P // g.G) * g.outer * g.D, 2 * 4.0 * g.P * g.n * g.n * g.D))
def winattn_pair_fwd(g, hv, ha, gate_v, gate_a):
    """The window-level cross-modal pair with its gates in ONE launch (round 6b, stg_winattn_pair_fwd): g is the table-free one-head geometry
    (ops._xwin_geom).  Returns ((r_v, lse_v, h_v'), (r_a, lse_a, h_a')) with h' = h + gate * r."""
    for t in (hv, ha):
        _chk2d(t, "winattn_pair_fwd operand", BF16)
    if hv.shape != ha.shape or hv.shape[1] != g.D or g.H != 1 or g.bm is not None:
        raise RuntimeError("winattn_pair_fwd: a table-free one-head pair over two [rows, D] tensors")
    outs = []
    args = []
    for q, kv in ((hv, ha), (ha, hv)):
        o = torch.empty((q.shape[0], g.D), dtype=BF16, device=q.device)
        lse = torch.empty((g.P, 1, 64), dtype=F32, device=q.device)
        x = torch.empty_like(o)
        args.append(_win_fill(g, q, kv, kv, o, lse))
        outs.append((o, lse, x))
    _lib.check(_lib.lib().stg_winattn_pair_fwd(C.byref(args[0]), C.byref(args[1]), _p(gate_v), _p(gate_a), _p(outs[0][2]), _p(outs[1][2]), _ld(outs[0][2]), _stream()),
               "stg_winattn_pair_fwd")
    return outs[0], outs[1]


@_family("winattn_bwd", lambda g, hv, *a, **kw: ("pair", 2 * 14.0 * (g.P // g.G) * g.outer * g.D, 2 * 10.0 * g.P * g.n * g.n * g.D))
def winattn_pair_bwd(g, hv, ha, rv, ra, lse_v, lse_a, dxv, dxa, gate_v, gate_a, dgate_v, dgate_a):
    """Backward of winattn_pair_fwd through the attention term: returns (dq_v, dkv_a, dq_a, dkv_v) -- direction v's query gradient (rows of h_v) and
    key-and-value gradient (rows of h_a), then direction a's -- already scaled by the gates; dgate_* accumulate <dx, r>."""
    for t in (hv, ha, rv, ra, dxv, dxa):
        _chk2d(t, "winattn_pair_bwd operand", BF16)
    if _ld(dxv) != _ld(dxa):
        raise RuntimeError("winattn_pair_bwd: dx_v and dx_a must share one leading dimension")
    dq_v, dkv_a, dq_a, dkv_v = (torch.empty((hv.shape[0], g.D), dtype=BF16, device=hv.device) for _ in range(4))
    a0 = _win_fill(g, hv, ha, ha, rv, lse_v)
    a1 = _win_fill(g, ha, hv, hv, ra, lse_a)
    _lib.check(_lib.lib().stg_winattn_pair_bwd(C.byref(a0), C.byref(a1), _p(dxv), _p(dxa), _ld(dxv), _p(gate_v), _p(gate_a), _p(dgate_v), _p(dgate_a),
                                               _p(dq_v), _p(dkv_a), _p(dq_a), _p(dkv_v), _ld(dq_v), _stream()), "stg_winattn_pair_bwd")
    return dq_v, dkv_a, dq_a, dkv_v


class TGeom:
    """One temporal-attention call over the fused qkv buffer: nm modality slabs x B clips x T frames x N tokens, H heads of
    dim 32; row(m, b, t, n) = ((m*B + b)*T + t)*N + n.  bias: fp32 [nm, H, T*T].  The additive tables the kernels read
    (bm / bmT) are workspaces filled by tattn_fwd and kept for tattn_bwd."""

    def __init__(self, nm, B, T, N, H, scale, bias, D=32):
        self.nm, self.B, self.T, self.N, self.H, self.scale, self.D = int(nm), int(B), int(T), int(N), int(H), float(scale), int(D)
        if not 1 <= self.T <= 32:
            raise RuntimeError("tattn: T must be in [1, 32]")
        self.rows = self.nm * self.B * self.T * self.N
        self.bias = bias
        if self.D == 32:
            if bias is None or bias.dtype != F32 or not bias.is_cuda or not bias.is_contiguous() or bias.numel() != self.nm * self.H * self.T * self.T:
                raise RuntimeError("tattn: bias must be a contiguous fp32 [nm, H, T*T] GPU tensor")
            self.bm = torch.empty((self.nm * self.H, 32, 32), dtype=F32, device=bias.device)
            self.bmT = torch.empty_like(self.bm)
        elif self.D in (64, 96):
            if bias is not None:
                raise RuntimeError("tattn: head dims 64 / 96 (ViT) run without a bias")
            self.bm = self.bmT = None
        else:
            raise RuntimeError("tattn: head dim must be 32, 64 or 96")


def tattn_supported(T, hd):
    return T <= 32 and hd in (32, 64, 96)


def _tattn_fill(g, Q, K, V, O):
    for t, name in ((Q, "Q"), (K, "K"), (V, "V")) + (((O, "O"),) if O is not None else ()):
        _chk2d(t, name, BF16)
        if t.shape[1] < g.H * g.D or t.shape[0] < g.rows:
            raise RuntimeError(f"tattn {name}: needs >= {g.rows} rows x {g.H * g.D} columns, got {tuple(t.shape)}")
    if not (_ld(Q) == _ld(K) == _ld(V)):
        raise RuntimeError("tattn: Q, K, V must share one leading dimension (slices of the fused qkv buffer)")
    a = _lib.TAttnArgs()
    a.Q, a.K, a.V, a.ld = _p(Q), _p(K), _p(V), _ld(Q)
    if O is not None:
        a.O, a.ldo = _p(O), _ld(O)
    a.bias, a.bm, a.bmT = _p(g.bias), _p(g.bm), _p(g.bmT)
    a.nm, a.B, a.T, a.N, a.H, a.D, a.scale = g.nm, g.B, g.T, g.N, g.H, g.D, g.scale
    return a


@_family("tattn_fwd", lambda g, Q, *a, **kw: (g.H * g.D, 8.0 * g.rows * g.H * g.D, 4.0 * g.rows * g.H * g.T * g.D))
def tattn_fwd(g, Q, K, V, out=None):
    """Returns O bf16 [rows, H*32]; fills g.bm / g.bmT."""
    if out is None:
        out = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=Q.device)
    a = _tattn_fill(g, Q, K, V, out)
    _lib.check(_lib.lib().stg_tattn_fwd(C.byref(a), _stream()), "stg_tattn_fwd")
    return out


@_family("tattn_bwd", lambda g, Q, *a, **kw: (g.H * g.D, 14.0 * g.rows * g.H * g.D, 10.0 * g.rows * g.H * g.T * g.D))
def tattn_bwd(g, Q, K, V, dO, *, dQ, dK, dV, dbias=None):
    """dQ / dK / dV: column slices of one bf16 buffer, written in place; dbias fp32 [nm, H, T*T] accumulated (optional).
    g must be the TGeom the forward ran with (its tables are reused)."""
    for t, name in ((dO, "dO"), (dQ, "dQ"), (dK, "dK"), (dV, "dV")):
        _chk2d(t, name, BF16)
        if t.shape[1] < g.H * g.D or t.shape[0] < g.rows:
            raise RuntimeError(f"tattn_bwd {name}: needs >= {g.rows} rows x {g.H * g.D} columns")
    if not (_ld(dQ) == _ld(dK) == _ld(dV)):
        raise RuntimeError("tattn_bwd: dQ, dK, dV must share one leading dimension")
    if dbias is not None and (dbias.dtype != F32 or not dbias.is_cuda or not dbias.is_contiguous()
                              or dbias.numel() != g.nm * g.H * g.T * g.T):
        raise RuntimeError("tattn_bwd: dbias must be a contiguous fp32 [nm, H, T*T] GPU tensor")
    a = _tattn_fill(g, Q, K, V, None)
    _lib.check(_lib.lib().stg_tattn_bwd(C.byref(a), _p(dO), _ld(dO), _p(dQ), _p(dK), _p(dV), _ld(dQ), _p(dbias), _stream()),
               "stg_tattn_bwd")
    return dQ, dK, dV


class MhaGeom:
    """ViT multi-head self-attention over the fused in_proj output: P frames x H heads, n tokens per frame at rows p*n + i,
    head dim D in {64, 96}.  window = (Himg, Wimg, ws, shift): P = frames * nW window problems of n = ws^2 tokens, each token at its
    place in the frame's [Himg, Wimg] image (cyclic shift included) -- the window-level cross-modal attention of wide adapters."""

    def __init__(self, P, H, n, D, scale, window=None):
        self.P, self.H, self.n, self.D, self.scale = int(P), int(H), int(n), int(D), float(scale)
        if not mha_supported(self.n, self.D):
            raise RuntimeError(f"mha: unsupported geometry n={n} D={D}")
        self.rows = self.P * self.n
        self.window = (0, 0, 0, 0)
        if window is not None:
            Hi, Wi, ws, shift = (int(x) for x in window)
            nW = (Hi // ws) * (Wi // ws)
            if ws < 1 or Hi % ws or Wi % ws or self.n != ws * ws or self.P % nW or not 0 <= shift < ws:
                raise RuntimeError(f"mha: bad window geometry {window} for P={P}, n={n}")
            self.window = (Hi, Wi, ws, shift)
            self.n_kv, self.outer = self.n, Hi * Wi        # (what _attn_cost reads)


def mha_supported(n, D):
    return D in (64, 96) and n >= 1


def _mha_fill(g, Q, K, V, O, lse):
    for t, name in ((Q, "Q"), (K, "K"), (V, "V"), (O, "O")):
        _chk2d(t, name, BF16)
        if t.shape[1] < g.H * g.D or t.shape[0] < g.rows:
            raise RuntimeError(f"mha {name}: needs >= {g.rows} rows x {g.H * g.D} columns, got {tuple(t.shape)}")
    if not (_ld(Q) == _ld(K) == _ld(V)):
        raise RuntimeError("mha: Q, K, V must share one leading dimension (slices of the fused in_proj output)")
    if lse.dtype != F32 or not lse.is_cuda or not lse.is_contiguous() or lse.numel() != g.P * g.H * g.n:
        raise RuntimeError("mha: lse must be a contiguous fp32 [P, H, n] GPU tensor")
    a = _lib.MhaArgs()
    a.Q, a.K, a.V, a.ld = _p(Q), _p(K), _p(V), _ld(Q)
    a.O, a.ldo = _p(O), _ld(O)
    a.lse = _p(lse)
    a.P, a.H, a.n, a.D, a.scale = g.P, g.H, g.n, g.D, g.scale
    a.win_h, a.win_w, a.win_size, a.win_shift = g.window
    return a


@_family("mha_fwd", lambda g, *a, **kw: _attn_cost(g, 3, 1, 2))
def mha_fwd(g, Q, K, V, out=None):
    """Returns (O bf16 [rows, H*D], lse fp32 [P, H, n] in the log2 domain)."""
    if out is None:
        out = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=Q.device)
    lse = torch.empty((g.P, g.H, g.n), dtype=F32, device=Q.device)
    a = _mha_fill(g, Q, K, V, out, lse)
    _lib.check(_lib.lib().stg_mha_fwd(C.byref(a), _stream()), "stg_mha_fwd")
    return out, lse


@_family("mha_bwd", lambda g, *a, **kw: _attn_cost(g, 5, 3, 5))
def mha_bwd(g, Q, K, V, O, lse, dO, *, dQ, dK, dV):
    """dV=None: K and V are one tensor (cross-modal attention) and dK receives its whole gradient dK + dV."""
    if dV is None and K.data_ptr() != V.data_ptr():
        raise RuntimeError("mha_bwd: dV=None needs K and V to be the same tensor")
    for t, name in ((dO, "dO"), (dQ, "dQ"), (dK, "dK")) + (((dV, "dV"),) if dV is not None else ()):
        _chk2d(t, name, BF16)
        if t.shape[1] < g.H * g.D or t.shape[0] < g.rows:
            raise RuntimeError(f"mha_bwd {name}: needs >= {g.rows} rows x {g.H * g.D} columns")
    if not (_ld(dQ) == _ld(dK) == (_ld(dV) if dV is not None else _ld(dK))):
        raise RuntimeError("mha_bwd: dQ, dK, dV must share one leading dimension")
    a = _mha_fill(g, Q, K, V, O, lse)
    delta = torch.empty((g.P, g.H, g.n), dtype=F32, device=Q.device)
    _lib.check(_lib.lib().stg_mha_bwd(C.byref(a), _p(dO), _ld(dO), _p(dQ), _p(dK), _p(dV), _ld(dQ), _p(delta), _stream()),
               "stg_mha_bwd")
    return dQ, dK, dV


@_family("mha_fwd", lambda g, *a, **kw: tuple(x if i == 0 else 2 * x for i, x in enumerate(_attn_cost(g, 3, 1, 2))))
def mha_fwd_pair(g, qkv0, qkv1):
    """mha_fwd on two problems of one geometry (the two directions of a cross-modal pair), one launch: ((O0, lse0), (O1, lse1))."""
    outs, args = [], []
    for Q, K_, V in (qkv0, qkv1):
        out = torch.empty((Q.shape[0], g.H * g.D), dtype=BF16, device=Q.device)
        lse = torch.empty((g.P, g.H, g.n), dtype=F32, device=Q.device)
        args.append(_mha_fill(g, Q, K_, V, out, lse)); outs.append((out, lse))
    _lib.check(_lib.lib().stg_mha_fwd_pair(C.byref(args[0]), C.byref(args[1]), _stream()), "stg_mha_fwd_pair")
    return outs[0], outs[1]


@_family("mha_bwd", lambda g, *a, **kw: tuple(x if i == 0 else 2 * x for i, x in enumerate(_attn_cost(g, 5, 3, 5))))
def mha_bwd_pair(g, p0, p1):
    """mha_bwd on two problems of one geometry, one launch per kernel: p = (Q, K, V, O, lse, dO, dQ, dK, dV) with dV None where K is V."""
    args, ptrs, keep = [], [], []
    lddo = lddq = None
    for Q, K_, V, O, lse, dO, dQ, dK, dV in (p0, p1):
        if dV is None and K_.data_ptr() != V.data_ptr():
            raise RuntimeError("mha_bwd_pair: dV=None needs K and V to be the same tensor")
        for t, name in ((dO, "dO"), (dQ, "dQ"), (dK, "dK")) + (((dV, "dV"),) if dV is not None else ()):
            _chk2d(t, name, BF16)
            if t.shape[1] < g.H * g.D or t.shape[0] < g.rows:
                raise RuntimeError(f"mha_bwd_pair {name}: needs >= {g.rows} rows x {g.H * g.D} columns")
        if not (_ld(dQ) == _ld(dK) == (_ld(dV) if dV is not None else _ld(dK))):
            raise RuntimeError("mha_bwd_pair: dQ, dK, dV must share one leading dimension")
        if lddo is None:
            lddo, lddq = _ld(dO), _ld(dQ)
        elif (lddo, lddq) != (_ld(dO), _ld(dQ)):
            raise RuntimeError("mha_bwd_pair: the two problems must share their leading dimensions")
        delta = torch.empty((g.P, g.H, g.n), dtype=F32, device=Q.device)
        args.append(_mha_fill(g, Q, K_, V, O, lse)); ptrs.append((_p(dO), _p(dQ), _p(dK), _p(dV), _p(delta))); keep.append(delta)
    _lib.check(_lib.lib().stg_mha_bwd_pair(C.byref(args[0]), *ptrs[0], C.byref(args[1]), *ptrs[1], lddo, lddq, _stream()), "stg_mha_bwd_pair")


@_family_io("patch_embed")
def vit_embed(patch, cls, pos, temb, BT, T):
    """ViT token assembly -> fp32 [BT*(np+1), D]; see stg_vit_embed."""
    _chk_flat(patch, "patch"); _chk_flat(cls, "cls", F32); _chk_flat(pos, "pos", F32); _chk_flat(temb, "temb", F32)
    D = cls.numel()
    n = pos.shape[0]
    np_ = n - 1
    if patch.numel() != BT * np_ * D or pos.numel() != n * D or temb.numel() != T * D:
        raise RuntimeError("vit_embed: shape mismatch")
    out = torch.empty((BT * n, D), dtype=F32, device=patch.device)
    _lib.check(_lib.lib().stg_vit_embed(_p(patch), _p(cls), _p(pos), _p(temb), _p(out), BT, T, np_, D, _stream()), "stg_vit_embed")
    return out


# ------------------------------------------------------------------------------------------------ AVQA head kernels (head.hip)
RELU, TANH = 0, 1


@_family_io("head_small")
def unary_fwd(op, x):
    _chk_flat(x, "x")
    y = torch.empty_like(x)
    _lib.check(_lib.lib().stg_unary_fwd(int(op), _p(x), _p(y), x.numel(), _stream()), "stg_unary_fwd")
    return y


@_family_io("head_small")
def unary_bwd(op, y, dy):
    _chk_flat(y, "y"); _chk_flat(dy, "dy")
    if y.numel() != dy.numel():
        raise RuntimeError("unary_bwd: size mismatch")
    dx = torch.empty_like(y)
    _lib.check(_lib.lib().stg_unary_bwd(int(op), _p(y), _p(dy), _p(dx), y.numel(), _stream()), "stg_unary_bwd")
    return dx


@_family_io("head_small")
def mul(a, b):
    _chk_flat(a, "a"); _chk_flat(b, "b")
    if a.numel() != b.numel():
        raise RuntimeError("mul: size mismatch")
    out = torch.empty_like(a)
    _lib.check(_lib.lib().stg_mul(_p(a), _p(b), _p(out), a.numel(), _stream()), "stg_mul")
    return out


@_family_io("head_small")
def embed_fwd(table, idx):
    _chk_flat(table, "table", F32); _chk_flat(idx, "idx", torch.int64)
    V, E = table.shape
    out = torch.empty((idx.numel(), E), dtype=BF16, device=table.device)
    _lib.check(_lib.lib().stg_embed_fwd(_p(table), _p(idx), _p(out), idx.numel(), V, E, _stream()), "stg_embed_fwd")
    return out


@_family_io("head_small")
def embed_bwd(dout, idx, dtable):
    _chk_flat(dout, "dout"); _chk_flat(idx, "idx", torch.int64); _chk_flat(dtable, "dtable", F32)
    V, E = dtable.shape
    if dout.numel() != idx.numel() * E:
        raise RuntimeError("embed_bwd: size mismatch")
    _lib.check(_lib.lib().stg_embed_bwd(_p(dout), _p(idx), _p(dtable), idx.numel(), V, E, _stream()), "stg_embed_bwd")


@_family_io("head_small")
def lstm_cell_fwd(gates, c_prev):
    _chk_flat(gates, "gates", F32); _chk_flat(c_prev, "c_prev", F32)
    B, H = c_prev.shape
    if tuple(gates.shape) != (B, 4 * H):
        raise RuntimeError("lstm_cell_fwd: gates must be [B, 4H]")
    c = torch.empty_like(c_prev)
    h = torch.empty((B, H), dtype=BF16, device=gates.device)
    _lib.check(_lib.lib().stg_lstm_cell_fwd(_p(gates), _p(c_prev), _p(c), _p(h), B, H, _stream()), "stg_lstm_cell_fwd")
    return h, c


@_family_io("head_small")
def lstm_cell_bwd(gates, c_prev, c, dh, dc):
    B, H = c_prev.shape
    for t, n, dt in ((gates, "gates", F32), (c_prev, "c_prev", F32), (c, "c", F32)):
        _chk_flat(t, n, dt)
    if dh is not None:
        _chk_flat(dh, "dh")
    if dc is not None:
        _chk_flat(dc, "dc", F32)
    dgates = torch.empty((B, 4 * H), dtype=BF16, device=gates.device)
    dc_prev = torch.empty_like(c_prev)
    _lib.check(_lib.lib().stg_lstm_cell_bwd(_p(gates), _p(c_prev), _p(c), _p(dh), _p(dc), _p(dgates), _p(dc_prev), B, H, _stream()),
               "stg_lstm_cell_bwd")
    return dgates, dc_prev


@_family_io("head_small")
def grounding_fwd(V, a):
    """V fp32 [F, n, C], a bf16 [F, C] -> (vmean bf16 [F, C], grd bf16 [F, C], saved = (p, rnorm, ra))."""
    _chk_flat(V, "V", F32); _chk_flat(a, "a")
    Fr, n, Cc = V.shape
    if tuple(a.shape) != (Fr, Cc) or n > 64:
        raise RuntimeError("grounding_fwd: a must be [F, C] and n <= 64")
    vmean = torch.empty((Fr, Cc), dtype=BF16, device=V.device)
    grd = torch.empty_like(vmean)
    p = torch.empty((Fr, n), dtype=F32, device=V.device)
    rn = torch.empty_like(p)
    ra = torch.empty((Fr,), dtype=F32, device=V.device)
    _lib.check(_lib.lib().stg_grounding_fwd(_p(V), _p(a), _p(vmean), _p(grd), _p(p), _p(rn), _p(ra), Fr, n, Cc, _stream()),
               "stg_grounding_fwd")
    return vmean, grd, (p, rn, ra)


@_family_io("head_small")
def grounding_bwd(V, a, saved, dvmean, dgrd, want_dV=True):
    Fr, n, Cc = V.shape
    p, rn, ra = saved
    _chk_flat(dgrd, "dgrd")
    if dvmean is not None:
        _chk_flat(dvmean, "dvmean")
    dV = torch.empty_like(V) if want_dV else None
    da = torch.empty_like(a)
    _lib.check(_lib.lib().stg_grounding_bwd(_p(V), _p(a), _p(p), _p(rn), _p(ra), _p(dvmean), _p(dgrd), _p(dV), _p(da), Fr, n, Cc,
                                            _stream()), "stg_grounding_bwd")
    return dV, da


@_family_io("head_small")
def mha1_fwd(q, k, v, drop, H):
    """q bf16 [B, E], k / v bf16 [T*B, E] (row t*B + b), drop fp32 [B, H, T] or None -> (o bf16 [B, E], p fp32 [B, H, T])."""
    _chk_flat(q, "q"); _chk_flat(k, "k"); _chk_flat(v, "v")
    B, E = q.shape
    T = k.shape[0] // max(B, 1)
    if k.shape != v.shape or k.shape[0] != T * B or k.shape[1] != E or E % H or T > 64:
        raise RuntimeError("mha1_fwd: bad shapes")
    if drop is not None:
        _chk_flat(drop, "drop", F32)
        if drop.numel() != B * H * T:
            raise RuntimeError("mha1_fwd: drop must be [B, H, T]")
    hd = E // H
    o = torch.empty_like(q)
    p = torch.empty((B, H, T), dtype=F32, device=q.device)
    _lib.check(_lib.lib().stg_mha1_fwd(_p(q), _p(k), _p(v), _p(drop), _p(o), _p(p), B, H, T, hd, hd ** -0.5, _stream()), "stg_mha1_fwd")
    return o, p


@_family_io("head_small")
def mha1_bwd(q, k, v, drop, p, dout, H):
    _chk_flat(dout, "dout")
    B, E = q.shape
    T = k.shape[0] // max(B, 1)
    hd = E // H
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _lib.check(_lib.lib().stg_mha1_bwd(_p(q), _p(k), _p(v), _p(drop), _p(p), _p(dout), _p(dq), _p(dk), _p(dv), B, H, T, hd, hd ** -0.5,
                                       _stream()), "stg_mha1_bwd")
    return dq, dk, dv


# ------------------------------------------------------------------------------------------------ AVS decoder kernels (dec.hip)
@_family_io("dec_im2col3x3")
def im2col3x3(x, F_, H, W, dilation):
    """x bf16 [F*H*W, C] (rows may be column slices of a wider buffer) -> bf16 [F*H*W, 9*C]."""
    _chk2d(x, "x", BF16, rows=F_ * H * W)
    Cc = x.shape[1]
    out = torch.empty((F_ * H * W, 9 * Cc), dtype=BF16, device=x.device)
    _lib.check(_lib.lib().stg_im2col3x3(_p(x), _ld(x), _p(out), F_, H, W, Cc, int(dilation), _stream()), "stg_im2col3x3")
    return out


def conv3x3_wgrad_supported(O, I, H=None, W=None):
    """conv_wgrad_kernel advances a piece's pixel 64 rows per block and wraps its row coordinate at most twice: maps of <= 32 pixels per
    frame (H * W <= 32) are refused by the C entry and take the im2col path."""
    return O % 8 == 0 and I % 8 == 0 and (H is None or H * W > 32)


@_family_io("dec_conv_wgrad", flops=lambda dy, x, F_, H, W, dilation, want_db=False: 18.0 * dy.shape[0] * dy.shape[1] * x.shape[1])
def conv3x3_wgrad(dy, x, F_, H, W, dilation, want_db=False):
    """dW [O, 9 * I] fp32 (columns ordered (kh, kw, i)) of a 3x3 convolution with padding = dilation, from dy [F*H*W, O] and
    x [F*H*W, I] (bf16, channels-last rows), without the im2col image; with want_db also db [O] = column sums of dy.
    Few output channels (O < 64 <= I): the operands are SWAPPED -- the kernel's 128-wide "output channel" tile carries x's channels (no zero
    columns), the taps shift the narrow dy (64 bytes per pixel and tap instead of x's rows re-read once per tap), and
    dW[o, (kh, kw, i)] = T[i, (2 - kh, 2 - kw, o)]: the 448 x 448 output convolution of the AVS decoder (128 -> 32) 3.1 -> 1.2 ms."""
    M = F_ * H * W
    _chk2d(dy, "dy", BF16, rows=M)
    _chk2d(x, "x", BF16, rows=M)
    O, I = dy.shape[1], x.shape[1]
    if O < 64 <= I and conv3x3_wgrad_supported(I, O, H, W):
        T = _conv3x3_wgrad_raw(x, dy, F_, H, W, dilation, False)                       # [I, (kh', kw', o)]
        dW = T.view(I, 3, 3, O).flip(1, 2).permute(3, 1, 2, 0).reshape(O, 9 * I).contiguous()
        if not want_db:
            return dW
        sums = torch.zeros((2, O), dtype=F32, device=dy.device)                        # db = column sums of dy (stg_bn_colsum, mode 0)
        if dy.stride(0) != O:                             # stg_bn_colsum takes dense [M, O] rows (no leading dimension): ADVICE r5
            dy = dy.contiguous()
        _lib.check(_lib.lib().stg_bn_colsum(_p(dy), None, None, None, _p(sums), M, O, 0, _stream()), "stg_bn_colsum")
        return dW, sums[0].contiguous()
    return _conv3x3_wgrad_raw(dy, x, F_, H, W, dilation, want_db)


def _conv3x3_wgrad_raw(dy, x, F_, H, W, dilation, want_db):
    M = F_ * H * W
    O, I = dy.shape[1], x.shape[1]
    splits = C.c_int(0)
    n = _lib.lib().stg_conv3x3_wgrad_ws_floats(M, O, I, C.byref(splits))
    if n <= 0:
        raise RuntimeError("conv3x3_wgrad: unsupported shape (O % 8 == 0 and I % 8 == 0)")
    ws = torch.empty((splits.value, O, 9 * I), dtype=F32, device=x.device)
    dbw = torch.empty((splits.value, O), dtype=F32, device=x.device) if want_db else None
    _lib.check(_lib.lib().stg_conv3x3_wgrad(_p(dy), _ld(dy), _p(x), _ld(x), _p(_zero_line(x.device)), _p(ws), ws.numel(), _p(dbw), F_, H, W, O, I,
                                            int(dilation), _stream()), "stg_conv3x3_wgrad")
    dW = _sum_splits(ws)
    if want_db:
        return dW, dbw.sum(0)                             # [splits, O] -> [O]: a few hundred floats
    return dW


@_family_io("dec_bilinear")
def bilinear_up2_fwd(x, F_, H, W, align_corners):
    _chk_flat(x, "x")
    Cc = x.shape[-1]
    if x.numel() != F_ * H * W * Cc:
        raise RuntimeError("bilinear_up2_fwd: shape mismatch")
    y = torch.empty((F_ * 4 * H * W, Cc), dtype=BF16, device=x.device)
    _lib.check(_lib.lib().stg_bilinear_up2_fwd(_p(x), _p(y), F_, H, W, Cc, int(bool(align_corners)), _stream()), "stg_bilinear_up2_fwd")
    return y


@_family_io("dec_bilinear")
def bilinear_up2_bwd(dy, F_, H, W, align_corners):
    _chk_flat(dy, "dy")
    Cc = dy.shape[-1]
    if dy.numel() != F_ * 4 * H * W * Cc:
        raise RuntimeError("bilinear_up2_bwd: shape mismatch")
    dx = torch.empty((F_ * H * W, Cc), dtype=BF16, device=dy.device)
    _lib.check(_lib.lib().stg_bilinear_up2_bwd(_p(dy), _p(dx), F_, H, W, Cc, int(bool(align_corners)), _stream()), "stg_bilinear_up2_bwd")
    return dx


@_family_io("dec_batchnorm")
def bn_colsum(a, b=None, mean=None, rstd=None):
    """[2, C] fp32: (sum a, sum a^2) when b and mean are None; (sum (a - mean), sum (a - mean)^2) when only mean is given;
    (sum b, sum b * (a - mean) * rstd) when b is given."""
    _chk_flat(a, "a")
    R, Cc = a.shape
    out = torch.zeros((2, Cc), dtype=F32, device=a.device)
    mode = 0
    if b is not None:
        _chk_flat(b, "b"); _chk_flat(mean, "mean", F32); _chk_flat(rstd, "rstd", F32)
        mode = 1
    elif mean is not None:
        _chk_flat(mean, "mean", F32)
        mode = 2
    _lib.check(_lib.lib().stg_bn_colsum(_p(a), _p(b), _p(mean), _p(rstd), _p(out), R, Cc, mode, _stream()), "stg_bn_colsum")
    return out


@_family_io("dec_batchnorm")
def bn_apply(x, mean, rstd, gamma, beta):
    _chk_flat(x, "x")
    R, Cc = x.shape
    for t, n in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (beta, "beta")):
        _chk1d(t, n, F32, Cc)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().stg_bn_apply(_p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), R, Cc, _stream()), "stg_bn_apply")
    return y


@_family_io("dec_batchnorm")
def bn_bwd(x, dy, mean, rstd, gamma, sums):
    _chk_flat(x, "x"); _chk_flat(dy, "dy")
    R, Cc = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().stg_bn_bwd(_p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(sums), _p(dx), R, Cc, _stream()), "stg_bn_bwd")
    return dx


@_family_io("ln_bwd")
def ln_param_grad(dy, x, mean, rstd, dgamma, dbeta):
    """dgamma / dbeta (fp32 [C], accumulated) of a LayerNorm over the rows of bf16 [R, C] -- block-folded column sums."""
    _chk_flat(dy, "dy"); _chk_flat(x, "x")
    R, Cc = x.shape
    _chk1d(mean, "mean", F32, R); _chk1d(rstd, "rstd", F32, R); _chk1d(dgamma, "dgamma", F32, Cc); _chk1d(dbeta, "dbeta", F32, Cc)
    _lib.check(_lib.lib().stg_ln_param_grad(_p(dy), _p(x), _p(mean), _p(rstd), _p(dgamma), _p(dbeta), R, Cc, _stream()), "stg_ln_param_grad")
