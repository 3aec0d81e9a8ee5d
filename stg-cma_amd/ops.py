"""Forward / backward orchestration of the Swin + STG-CMA hot path as explicit sequences of libstgcma_hip.so launches
(kernels.py), wrapped in torch.autograd.Functions.  No ATen compute kernel runs on the hot path: residual joins,
activation gradients, DropPath scaling, window (un)partitioning and the temporal rearranges are fused into GEMM epilogues,
LayerNorm-backward `add_to`, or attention addressing.

Data layout.  One fused token tensor X[M*BT*N, C]: rows of modality 0 (video) first, then modality 1 (audio), each in the
reference's '(b t) n c' order, so every frozen (shared) weight runs ONE GEMM over both modalities (Swin_AVE.py:743-745
calls self.attn twice with the same weights) while per-modality adapters address row slices.  The residual stream X is
fp32 (what the reference's autocast loop effectively keeps: LayerNorm outputs / residual adds promote to fp32), every
branch tensor (LN output, qkv, attention output, MLP hidden, adapter hidden) is bf16, and the gradient stream dX is bf16.

Backward computes dgrad through the frozen backbone and wgrad only for tensors whose gradient autograd asks for
(adapters, gates, temporal bias tables, head) -- what autograd does for the reference under its freeze filter
(traintest_adapt_ave29.py:38-61).  Asking for a backbone weight gradient raises NotImplementedError.
"""
import weakref

import torch

from . import kernels as K
from .kernels import ACT_GELU, ACT_NONE, BF16, F32

from . import config as _cfg       # every switch below: stgcma.configure(...) / the STG_* environment, read once (config.py)
USE_UPLN = _cfg.opt("upln")             # fused up-projection + residual + LayerNorm (csrc/upln.hip)
# the MLP hidden's saved GELU derivative ([rows, 4C], the widest tensor of the step) as one byte per element (STG_U8_LIN) instead of bf16
MLP_DACT = _cfg.opt("mlp_dact")
RESIDUAL_DTYPE = _cfg.opt("residual")   # fp32 default; bf16 = A/B knob

# ------------------------------------------------------------------------------------------------ weight shadows
_shadow_cache = {}   # id(parameter) -> (weakref to it, {transpose: ((data_ptr, version), bf16 tensor)})


def shadow(p, transpose=False):
    """bf16 copy (optionally transposed, trailing dim zero-padded to a multiple of 8) of an fp32 parameter, cached until the
    parameter's storage or version changes (optimizer steps bump the version; frozen weights are cast once)."""
    t = p.detach()
    if not t.is_cuda:
        raise RuntimeError("stg-cma_amd: parameters must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != F32:
        raise RuntimeError(f"stg-cma_amd: parameters are expected in fp32 (got {t.dtype}); bf16 shadows are made internally")
    # keyed by the parameter OBJECT (weakly): a freed model's storage may be recycled for a new parameter of the same shape
    # and version, so the address alone does not identify the values
    ent = _shadow_cache.get(id(p))
    if ent is None or ent[0]() is not p:
        key = id(p)
        ent = (weakref.ref(p, lambda _r, k=key: _shadow_cache.pop(k, None)), {})
        _shadow_cache[key] = ent
    slot = ent[1]
    tag = (t.data_ptr(), p._version)
    hit = slot.get(bool(transpose))
    if hit is not None and hit[0] == tag:
        return hit[1]
    s = K.cast_bf16(t.contiguous(), transpose=transpose)
    slot[bool(transpose)] = (tag, s)
    return s


class ShadowSet:
    """bf16 shadows, both orientations, of ALL trainable weight matrices of a model from ONE launch.  The adapter and head
    weights change at every optimizer step: one cast launch per matrix and orientation was ~520 launches (2.6 ms of GPU time and
    several ms of host time) per Swin-B step.  refresh() re-casts everything into a fresh arena when any parameter's storage or
    version changed and registers the views in the shadow cache, so shadow() hits; older arenas stay alive while referenced."""

    def __init__(self, params):
        self.params = list(params)
        self.meta, off = [], 0
        for p in self.params:
            R = p.shape[0]
            Cc = p.numel() // R
            ld, ldT = (Cc + 7) // 8 * 8, (R + 7) // 8 * 8
            self.meta.append((R, Cc, ld, ldT, off, off + R * ld))
            off += R * ld + Cc * ldT
        self.total = off
        self.max_elems = max((m[0] * m[1] for m in self.meta), default=0)
        self.ptrs = self.tags = self.desc = None

    def refresh(self):
        if not self.params:
            return
        tags = tuple((p.data_ptr(), p._version) for p in self.params)
        if tags == self.tags and all(id(p) in _shadow_cache for p in self.params):
            return
        for p in self.params:
            if not p.is_cuda or p.dtype != F32 or not p.is_contiguous():
                raise RuntimeError("stg-cma_amd: trainable weights must be contiguous fp32 GPU tensors")
        dev = self.params[0].device
        ptrs = tuple(t[0] for t in tags)
        if ptrs != self.ptrs:
            self.desc = K.cast_desc_table([(ptr, m[4], m[5], m[0], m[1], m[2], m[3]) for ptr, m in zip(ptrs, self.meta)], dev)
            self.ptrs = ptrs
        arena = torch.zeros(self.total, dtype=BF16, device=dev)
        K.cast_bf16_multi(self.desc, len(self.params), self.max_elems, arena)
        for p, tag, (R, Cc, ld, ldT, off, offT) in zip(self.params, tags, self.meta):
            key = id(p)
            ent = _shadow_cache.get(key)
            if ent is None or ent[0]() is not p:
                ent = (weakref.ref(p, lambda _r, k=key: _shadow_cache.pop(k, None)), {})
                _shadow_cache[key] = ent
            ent[1][False] = (tag, arena[off:off + R * ld].view(R, ld))
            ent[1][True] = (tag, arena[offT:offT + Cc * ldT].view(Cc, ldT))
        self.tags = tags


def refresh_shadows(holder, names, P, need):
    """Batch-refresh the shadows of the trainable Linear weights (adapters, head) of a model call; `holder` (the model's
    plan) keeps the ShadowSet."""
    sel = tuple(n for n in names if need.get(n, False) and P[n].dim() == 2 and n.endswith(".weight")
                and ("D_fc" in n or "mlp_head" in n))
    ss = getattr(holder, "_shadowset", None)
    if ss is None or ss[0] != sel or any(a is not b for a, b in zip(ss[1].params, (P[n] for n in sel))):
        ss = (sel, ShadowSet([P[n] for n in sel]))
        holder._shadowset = ss
    ss[1].refresh()


USE_XHAT = _cfg.opt("xhat")     # 0 = LayerNorm affine applied by the LN kernels, fp32 rows re-read in backward (A/B knob)
_unit_cache = {}


def unit_affine(C, device):
    """(ones[C], zeros[C]) fp32: the affine of a LayerNorm whose gamma / beta were folded into the frozen Linear behind it."""
    key = (int(C), str(device))
    u = _unit_cache.get(key)
    if u is None:
        u = _unit_cache[key] = (torch.ones(C, dtype=F32, device=device), torch.zeros(C, dtype=F32, device=device))
    return u


class _Folded:
    """A frozen Linear behind a frozen LayerNorm with the norm's affine folded in:  LN(x) W^T + b = x_hat (W gamma)^T + (b + W beta).
    w: bf16 [N, K] shadow of W * gamma[None, :], wt: its transpose (the dgrad operand, made on first use), b: fp32 folded bias."""

    def __init__(self, wp, gp, bp, biasp):
        W = wp.detach().float()
        self._wf = (W * gp.detach().float()[None, :]).contiguous()
        self.w = K.cast_bf16(self._wf)
        self._wt = None
        b = torch.mv(W, bp.detach().float())
        self.b = (b + biasp.detach().float()).contiguous() if biasp is not None else b.contiguous()

    @property
    def wt(self):
        if self._wt is None:
            self._wt = K.cast_bf16(self._wf, transpose=True)
            self._wf = None
        return self._wt


def folded(wp, gp, bp, biasp):
    """Cached _Folded of (weight, LayerNorm gamma, LayerNorm beta, bias) parameters, rebuilt when any of them changes."""
    for t in (wp, gp, bp):
        if not t.is_cuda or t.dtype != F32:
            raise RuntimeError("stg-cma_amd: parameters are expected as fp32 GPU tensors")
    shadow(wp)                                               # makes sure the weak-ref'd cache entry of wp exists
    slot = _shadow_cache[id(wp)][1]
    tag = tuple((t.data_ptr(), t._version) for t in (wp, gp, bp) + ((biasp,) if biasp is not None else ()))
    key = ("fold", id(gp))
    hit = slot.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    f = _Folded(wp, gp, bp, biasp)
    slot[key] = (tag, f)
    return f


def spec_xhat(spec):
    """Does this block run its LayerNorms in x_hat form (stg_*_xhat, include/stgcma.h)?  Both norms feed only frozen Linears (qkv,
    fc1) in the two-stream modes; the single-stream modes hand norm2(x) to a TRAINABLE adapter (Swin_AVE.py:438-440) and keep y."""
    return USE_XHAT and not spec.parallel and not getattr(spec, "fp8", False) and RESIDUAL_DTYPE == F32


def f32c(p):
    t = p.detach()
    if t.dtype != F32 or not t.is_cuda:
        raise RuntimeError("stg-cma_amd: expected an fp32 GPU parameter")
    return t.contiguous()


def clear_shadow_cache():
    _shadow_cache.clear()
    _wintab_cache.clear()


def shadow_fp8(p, transpose=False):
    """Block-scaled e4m3 copy (K.Fp8) of a FROZEN fp32 weight, in the orientation given, cached next to its bf16 shadow (same
    invalidation: parameter storage / version)."""
    w = shadow(p, transpose)
    slot = _shadow_cache[id(p)][1]
    key = ("fp8", bool(transpose))
    hit = slot.get(key)
    tag = slot[bool(transpose)][0]
    if hit is not None and hit[0] == tag:
        return hit[1]
    f = K.quant_fp8(w)
    slot[key] = (tag, f)
    return f


def shadow_mlp_w2(p):
    """fc2.weight as the fused MLP kernel wants it: bf16, hidden index permuted (K.mlp_w2_perm), cached like the plain shadow."""
    w = shadow(p)
    slot = _shadow_cache[id(p)][1]
    tag = slot[False][0]
    hit = slot.get("mlp_w2p")
    if hit is not None and hit[0] == tag:
        return hit[1]
    wp = w[:, K.mlp_w2_perm(w.shape[1], w.device)].contiguous()
    slot["mlp_w2p"] = (tag, wp)
    return wp


# fc1 -> GELU -> fc2 (and its backward, pre-activation recomputed) as ONE kernel where csrc/mlp.hip is built (C = 128: Swin-B stage 0):
# the [rows, 4C] hidden tensor and its saved derivative never reach HBM
MLP_FUSED = _cfg.opt("mlp_fused")


def _f8(sel, site, bwd=False):
    """Is the frozen Linear `site` ('qkv' / 'proj' / 'fc1' / 'fc2' / 'merge'; bwd: its data-gradient GEMM) on the e4m3 path?
    sel: False / True (every site, both directions) / a set of 'site.f' and 'site.b' tags (stgcma.fp8.enable(model, sites=...))."""
    if not sel or sel is True:
        return bool(sel)
    return (site + (".b" if bwd else ".f")) in sel


def frozen_gemm(A, wp, bias=None, *, t=False, fp8=False, **kw):
    """A . W^T (t=True: A . W, the dgrad) for a FROZEN Linear weight `wp`.  bf16 MFMA on the weight's bf16 shadow, or -- fp8 (BASELINE
    config 5, opt-in: stgcma.fp8) -- the block-scaled e4m3 MFMA: the weight's cached e4m3 shadow and A quantised per 32-wide k-block
    on the way in (stg_quant_fp8_mx); accumulation, epilogue and output are what they are on the bf16 path."""
    if fp8:
        return K.gemm_nt(K.quant_fp8(A), shadow_fp8(wp, t), bias, **kw)
    return K.gemm_nt(A, shadow(wp, t), bias, **kw)


# ------------------------------------------------------------------------------------------------ geometry caches
_geom_cache = {}


def window_token_map(H, W, ws, shift):
    """[nW*ws*ws] int32: natural token index of every token of every shifted window (roll(-shift) + window_partition,
    Swin_AVE.py:727-740; window_reverse + roll(+shift) is the inverse scatter through the same map)."""
    wi = torch.arange(H // ws).view(-1, 1, 1, 1)
    wj = torch.arange(W // ws).view(1, -1, 1, 1)
    ti = torch.arange(ws).view(1, 1, -1, 1)
    tj = torch.arange(ws).view(1, 1, 1, -1)
    h = (wi * ws + ti + shift) % H
    w = (wj * ws + tj + shift) % W
    m = (h * W + w).reshape(-1).to(torch.int32)
    assert int(m.min()) >= 0 and int(m.max()) < H * W and m.unique().numel() == H * W
    return m


def shift_mask(H, W, ws, shift):
    """[nW, ws*ws, ws*ws] fp32 0 / -100 (Swin_AVE.py:368-389)."""
    region = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            region[hs, wsl] = cnt
            cnt += 1
    ids = region.reshape(-1)[window_token_map(H, W, ws, 0).long()].view(-1, ws * ws)
    diff = ids[:, None, :] - ids[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).contiguous()


def temporal_token_map(N, T):
    """[N*T] int32: sequence of token n over frames, map[n*T + t] = t*N + n ('(b t) n c -> (b n) t c', Swin_AVE.py:705)."""
    n = torch.arange(N).view(-1, 1)
    t = torch.arange(T).view(1, -1)
    return (t * N + n).reshape(-1).to(torch.int32)


def geom(device, H, W, ws, shift, T):
    key = (str(device), H, W, ws, shift, T)
    g = _geom_cache.get(key)
    if g is None:
        g = {
            "wmap": window_token_map(H, W, ws, shift).to(device),
            "mask": shift_mask(H, W, ws, shift).to(device) if shift > 0 else None,
            "tmap": temporal_token_map(H * W, T).to(device),
        }
        _geom_cache[key] = g
    return g


_wintab_cache = {}   # id(bias-table parameter) -> (weakref, tag, (bm, bmT))
USE_WINATTN = _cfg.opt("winattn")     # 0 = route W-MSA through the generic attention kernels (A/B knob)
USE_TATTN = _cfg.opt("tattn")         # 0 = route temporal attention through the generic kernels
USE_MHA_X = _cfg.opt("mha_x")         # 0 = wide frame-global cross-modal attention through the generic kernels


def win_tables(tab_p, index, mask, n):
    """Padded bias + shift-mask tables of one block for the whole-window attention kernels, rebuilt only when the (frozen)
    relative_position_bias_table changes (Swin_AVE.py:262-273)."""
    t = tab_p.detach()
    tag = (t.data_ptr(), tab_p._version, None if mask is None else mask.data_ptr())
    ent = _wintab_cache.get(id(tab_p))
    if ent is not None and ent[0]() is tab_p and ent[1] == tag:
        return ent[2]
    tabs = K.winattn_table(f32c(tab_p), index.reshape(-1), mask, n)
    key = id(tab_p)
    _wintab_cache[key] = (weakref.ref(tab_p, lambda _r, k=key: _wintab_cache.pop(k, None)), tag, tabs)
    return tabs


def _check_frozen(need, names, who):
    """need: {param name: autograd wants its gradient}."""
    for n in names:
        if need.get(n, False):
            raise NotImplementedError(
                f"{who}: parameter '{n}' has requires_grad=True, but this build computes weight gradients only for the "
                f"adapter / gate / temporal-bias / head tensors (the reference's freeze_base=True recipe, "
                f"traintest_adapt_ave29.py:52-61). Freeze the backbone before calling forward.")


def drop_scale(p, n_rows, device, training, pool=None):
    """timm DropPath mask over n_rows leading entries, scaled by 1/keep; None in eval or when p == 0."""
    if not training or p == 0.:
        return None
    keep = 1.0 - p
    if pool is not None:
        return pool.take(n_rows, keep)
    return torch.empty(n_rows, dtype=F32, device=device).bernoulli_(keep).div_(keep)


_keep_cache = {}


class DropPool:
    """Every DropPath / Dropout mask of one forward pass from ONE uniform draw: a bernoulli_ launch per mask (23 per step
    for Swin-B) costs ~0.4 ms of host time each on ROCm.  requests: [(entries, keep probability)] in the order the forward
    will take them; same distribution as timm's DropPath / nn.Dropout (Bernoulli(keep) / keep)."""

    def __init__(self, requests, device):
        self.requests, self.i, self.off = requests, 0, 0
        key = (tuple(requests), str(device))
        kv = _keep_cache.get(key)
        if kv is None:
            keep = torch.cat([torch.full((n,), float(k)) for n, k in requests]).to(device)
            kv = (keep, 1.0 / keep)
            _keep_cache[key] = kv
        self.mask = (torch.rand(kv[0].shape, device=device) < kv[0]).to(F32).mul_(kv[1])

    def take(self, n, keep):
        rn, rk = self.requests[self.i]
        if rn != n or abs(rk - keep) > 1e-12:
            raise RuntimeError(f"DropPool: request {self.i} is ({rn}, {rk}), forward asked for ({n}, {keep})")
        out = self.mask[self.off:self.off + n]
        self.i += 1
        self.off += n
        return out


def block_drop_requests(spec, B):
    """The (entries, keep) masks block_forward draws, in order."""
    if spec.drop_path == 0.:
        return []
    keep = 1.0 - spec.drop_path
    req = [(B * spec.N, keep) for _ in spec.mods] if spec.t_attn else []
    if spec.parallel:
        req.append((B * spec.T, keep))
    return req


# ------------------------------------------------------------------------------------------------ Swin block
class BlockSpec:
    """Static description of one SwinTransformerBlock (Swin_AVE.py:317-391).

    mods: modality ids present in the fused tensor ((0, 1) two-stream, (0,) video only, (1,) audio only);
    fuse: gated cross-modal attention inside the S-adapters ('fusion_adapt');
    parallel: FFN adapter parallel to the MLP on norm2(x), scaled 0.5 with DropPath ('video_adapt' / 'audio_adapt'),
              otherwise serial on the MLP output."""

    def __init__(self, C, H, W, T, heads, ws, shift, t_attn, mode="fusion_adapt", drop_path=0.):
        self.C, self.H, self.W, self.T, self.heads, self.ws, self.shift, self.t_attn = C, H, W, T, heads, ws, shift, t_attn
        self.N = H * W
        self.nW = (H // ws) * (W // ws)
        self.hd = C // heads
        self.mode = mode
        self.drop_path = drop_path
        self.mods = {"fusion_adapt": (0, 1), "multimodal_adapt_no_fusion": (0, 1), "video_adapt": (0,), "audio_adapt": (1,)}[mode]
        self.fuse = mode == "fusion_adapt"
        self.parallel = mode in ("video_adapt", "audio_adapt")
        if self.hd not in (16, 32, 48, 64, 96, 128):
            raise NotImplementedError(f"attention head dim {self.hd} unsupported (need 16/32/48/64/96/128)")


FusionBlockSpec = BlockSpec
_SFX = ("", "_Audio")


def block_param_names(spec):
    """Parameter names (relative to the block module) the block functions read, in a fixed order."""
    names = ["norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
             "attn.relative_position_bias_table", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
             "mlp.fc2.weight", "mlp.fc2.bias", "gate_v", "gate_a"]
    ad = []
    for m in spec.mods:
        ad += ["S_Adapter" + _SFX[m], "S_Adapter2" + _SFX[m]]
    if spec.t_attn:
        names += ["attn.temporal_position_bias_table", "attn.temporal_position_bias_table_audio"]
        ad += ["T_Adapter" + _SFX[m] for m in spec.mods]
    for a in ad:
        names += [f"{a}.D_fc1.weight", f"{a}.D_fc1.bias", f"{a}.D_fc2.weight", f"{a}.D_fc2.bias"]
    return names


def block_buffer_names(spec):
    return ["attn.relative_position_index"] + (["attn.t_relative_coords", "attn.t_relative_coords_a"] if spec.t_attn else [])


def fusion_param_names(t_attn):
    return block_param_names(BlockSpec(32, 7, 7, 1, 1, 7, 0, t_attn))


FROZEN_ONLY = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
               "attn.relative_position_bias_table", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
               "mlp.fc2.weight", "mlp.fc2.bias")


class _Adapter:
    """Shadows + parameter handles of one Adapter (D_fc1 -> GELU -> D_fc2)."""

    def __init__(self, P, name):
        self.name = name
        self.w1p, self.b1p = P[name + ".D_fc1.weight"], P[name + ".D_fc1.bias"]
        self.w2p, self.b2p = P[name + ".D_fc2.weight"], P[name + ".D_fc2.bias"]
        self.dh = self.w1p.shape[0]
        if self.dh not in (16, 32, 48, 64, 96, 128):
            raise NotImplementedError(f"adapter hidden width {self.dh} unsupported (need 16/32/48/64/96/128)")

    w1 = property(lambda s: shadow(s.w1p))
    w1t = property(lambda s: shadow(s.w1p, True))
    w2 = property(lambda s: shadow(s.w2p))
    w2t = property(lambda s: shadow(s.w2p, True))
    b1 = property(lambda s: f32c(s.b1p))
    b2 = property(lambda s: f32c(s.b2p))


def _zeros_like_f32(p):
    return torch.zeros(p.shape, dtype=F32, device=p.device)


class _Grads:
    """fp32 gradient buffers for the trainable tensors of a call, keyed by parameter name."""

    def __init__(self, P, need, prefix="", arena=None):
        self.P, self.need, self.prefix, self.arena = P, need, prefix, arena
        self.g = {}
        self.pending = []        # deferred weight-gradient problems (operands stay referenced until flush)

    def flush(self):
        """Issue the deferred adapter weight gradients of a block: one launch pair per launch plan (K.wgrad_tn_multi)."""
        if self.pending:
            K.wgrad_tn_multi(self.pending)
            self.pending = []

    def buf(self, name):
        if not self.need.get(self.prefix + name, False):
            return None
        if name not in self.g:
            if self.arena is not None:
                self.g[name] = self.arena.view(self.prefix + name, self.P[name])
            else:
                self.g[name] = _zeros_like_f32(self.P[name])
        return self.g[name]


class GradArena:
    """One flat zero-initialised fp32 buffer holding every trainable gradient of a backward pass: a single memset instead
    of one per tensor, and the whole payload of the data-parallel exchange (ONE RCCL all-reduce over xGMI, ddp.py)."""

    def __init__(self, names, P, need, device):
        self.off = {}
        total = 0
        self.n_real = 0
        for n in names:
            if need.get(n, False):
                self.off[n] = total
                total += (P[n].numel() + 3) // 4 * 4          # keep every view 16-byte aligned
                self.n_real += P[n].numel()
        self.flat = torch.zeros(total, dtype=F32, device=device)

    def view(self, name, like):
        o = self.off[name]
        return self.flat[o:o + like.numel()].view(like.shape)


def _adapter_wgrad(G, name, dZ, X, dY2, H2, *, rs=None, rs_outer=1, rs_inner=1):
    """Accumulate dW/db of D_fc1 (dZ^T X) and D_fc2 ((s*dY2)^T H2) when they train."""
    w1, b1 = G.buf(name + ".D_fc1.weight"), G.buf(name + ".D_fc1.bias")
    if w1 is not None or b1 is not None:
        if w1 is None or b1 is None:
            raise NotImplementedError("adapter weight and bias must be frozen/trained together")
        G.pending.append((dZ, X, w1, b1, None, 1, 1))
    w2, b2 = G.buf(name + ".D_fc2.weight"), G.buf(name + ".D_fc2.bias")
    if w2 is not None or b2 is not None:
        if w2 is None or b2 is None:
            raise NotImplementedError("adapter weight and bias must be frozen/trained together")
        G.pending.append((dY2, H2, w2, b2, rs, rs_outer, rs_inner))


def _xattn_geom(spec, BT, dh, window, g):
    if window:
        n = spec.ws * spec.ws
        return K.AttnGeom(BT * spec.nW, 1, n, dh, G=spec.nW, outer=spec.N, n_kv=n, outer_kv=spec.N, scale=1.0,
                          window=(spec.H, spec.W, spec.ws, spec.shift))
    return K.AttnGeom(BT, 1, spec.N, dh, G=1, outer=spec.N, n_kv=spec.N, outer_kv=spec.N, scale=1.0)


USE_XWIN = _cfg.opt("xwin")     # 0 = window-level cross-modal attention on the generic kernels (A/B knob)
USE_MHA_WIN = _cfg.opt("mha_win")   # 0 = WIDE (d_h = 64 / 96) window-level cross-modal attention on the generic kernels (A/B knob)


def _xwin_geom(spec, BT, dev, dh=32):
    """The window-level cross-modal pair on the whole-window kernels of winattn.hip: one head of width 32 or 16, K = V = the other
    modality's hidden states, scale 1, no bias and no shift mask (round 6: the table-free form -- the kernels synthesise the -1e30 of the
    padding keys; rounds 2-5 fetched an all-zero 16 KiB table per window)."""
    return K.WinGeom(BT, 1, spec.H, spec.W, spec.ws, spec.shift, 1.0, None, None, D=dh)


PAIR_EW = _cfg.opt("pair_ew")      # 0 = one element-wise launch per direction of a cross-modal pair (A/B knob)
XATTN_GATE = _cfg.opt("xattn_gate")  # 0 = stg_gate_bwd2 in front of the frame-global pair's merged backward (rounds 4-6a; A/B knob)
XWIN_PAIR = _cfg.opt("xwin_pair")  # 0 = the window-level cross-modal pair as two launches per pass + the gate kernels (rounds 2-6a; A/B knob)
XATTN_MERGED = _cfg.opt("xattn_merged")   # 0 = the frame-global cross-modal pair's backward as four passes (dQ, dK + dV per direction: rounds 1-4)


def _gate2(hv, rv, gate_v, ha, ra, gate_a):
    if PAIR_EW:
        return K.gate_fwd2(hv, rv, gate_v, ha, ra, gate_a)
    return K.gate_fwd(hv, rv, gate_v), K.gate_fwd(ha, ra, gate_a)


def _gate_bwd2(dhv2, rv, gate_v, dgate_v, dha2, ra, gate_a, dgate_a):
    if PAIR_EW:
        return K.gate_bwd2(dhv2, rv, gate_v, dgate_v, dha2, ra, gate_a, dgate_a)
    return K.gate_bwd(dhv2, rv, gate_v, dgate_v), K.gate_bwd(dha2, ra, gate_a, dgate_a)


def _join3(dhv2, dq_v, dkv_v, dha2, dq_a, dkv_a, zs, outs):
    """The three gradient paths into each adapter hidden state joined (and, with zs, multiplied by the saved activation derivative)."""
    if zs is None:
        return K.add(dhv2, dq_v, dkv_v), K.add(dha2, dq_a, dkv_a)
    if PAIR_EW:
        return K.add3_mul2(dhv2, dq_v, dkv_v, zs[0], dha2, dq_a, dkv_a, zs[1], outs=outs)
    return K.add3_mul(dhv2, dq_v, dkv_v, zs[0], out=None if outs is None else outs[0]), \
        K.add3_mul(dha2, dq_a, dkv_a, zs[1], out=None if outs is None else outs[1])


MHA_PAIR = _cfg.opt("mha_pair")     # 0 = one mha launch per direction of a cross-modal pair (A/B knob)
XSMALL = _cfg.opt("xsmall")         # 0 = the ViT blocks' cross-modal pair on the generic attention kernels (rounds 1-5)
MHA_MERGED = _cfg.opt("mha_merged")  # 0 = the wide frame-global pair's backward as dQ + dK/dV passes per direction (rounds 1-5)


def _mha_pair_fwd(mg, hv, ha):
    if MHA_PAIR and hv.stride(0) == ha.stride(0):
        return K.mha_fwd_pair(mg, (hv, ha, ha), (ha, hv, hv))        # both directions, one launch
    return K.mha_fwd(mg, hv, ha, ha), K.mha_fwd(mg, ha, hv, hv)


def _cross_modal_fwd(spec, BT, hv, ha, gate_v, gate_a, window, g, save, geoms=None):
    """h' = h + gate * softmax(h hother^T) hother, both directions (Swin_AVE.py:750-760 / :799-808).
    geoms = (video-queries geometry, audio-queries geometry) when the two token counts differ (ViT)."""
    if geoms is None and not window and USE_MHA_X and K.mha_supported(spec.N, hv.shape[1]):
        # wide adapters (d_h = 64 / 96): the frame-global pair runs on the flash kernels of mha.hip (H = 1, K = V, scale 1)
        mg = K.MhaGeom(BT, 1, spec.N, hv.shape[1], 1.0)
        (rv, lse_v), (ra, lse_a) = _mha_pair_fwd(mg, hv, ha)
        return _gate2(hv, rv, gate_v, ha, ra, gate_a) + ((rv, ra, lse_v, lse_a, mg),)
    if geoms is None and window and USE_MHA_WIN and K.mha_supported(spec.ws * spec.ws, hv.shape[1]) and BT * spec.nW < 65536:
        # wide adapters, window level (Swin-L: d_h = 96): the same flash kernels with the window map (one 64-key trip per window)
        mg = K.MhaGeom(BT * spec.nW, 1, spec.ws * spec.ws, hv.shape[1], 1.0, window=(spec.H, spec.W, spec.ws, spec.shift))
        (rv, lse_v), (ra, lse_a) = _mha_pair_fwd(mg, hv, ha)
        return _gate2(hv, rv, gate_v, ha, ra, gate_a) + ((rv, ra, lse_v, lse_a, mg),)
    if geoms is None and window and USE_WINATTN and USE_XWIN and K.winattn_supported(spec.ws * spec.ws, hv.shape[1], table=False):
        wg = _xwin_geom(spec, BT, hv.device, hv.shape[1])      # d_h = 32, and since round 6 d_h = 16 (Swin-B stage 0: was on the generic kernels)
        if XWIN_PAIR and PAIR_EW and hv.shape == ha.shape and hv.stride(0) == ha.stride(0):
            # round 6b: both directions AND the gates in one launch (three launches before; the gate kernel was as long as an attention launch at stage 2)
            (rv, lse_v, hv2), (ra, lse_a, ha2) = K.winattn_pair_fwd(wg, hv, ha, gate_v, gate_a)
            return hv2, ha2, (rv, ra, lse_v, lse_a, wg)
        rv, lse_v = K.winattn_fwd(wg, hv, ha, ha, want_lse=True)
        ra, lse_a = K.winattn_fwd(wg, ha, hv, hv, want_lse=True)
        return _gate2(hv, rv, gate_v, ha, ra, gate_a) + ((rv, ra, lse_v, lse_a, wg),)
    ag_v, ag_a = geoms if geoms is not None else (_xattn_geom(spec, BT, hv.shape[1], window, g),) * 2
    if geoms is not None and XSMALL and ag_v.H == 1 and ag_v.G == 1 and ag_v.scale == ag_a.scale and ag_v.outer == ag_v.n and ag_a.outer == ag_a.n \
            and K.xsmall_supported(ag_v.n, ag_a.n, hv.shape[1]) and hv.shape[0] == ag_v.P * ag_v.n and ha.shape[0] == ag_a.P * ag_a.n:
        # round 6: the ViT blocks' pair (197 video + 49 audio tokens, width 48): one workgroup per frame, both directions in one launch (xsmall.hip)
        xg = K.XsGeom(ag_v.P, ag_v.n, ag_a.n, hv.shape[1], ag_v.scale)
        (rv, lse_v), (ra, lse_a) = K.xsmall_fwd(xg, hv, ha)
        hv2, ha2 = _gate2(hv, rv, gate_v, ha, ra, gate_a)
        return hv2, ha2, (rv, ra, lse_v, lse_a, xg)
    if XATTN_MERGED and PAIR_EW and not window and hv.stride(0) == ha.stride(0) and K.xattn_pair_fwd_supported(ag_v, hv, ha, ag_a, ha, hv):
        # both directions and the gates in one launch (the frame-global kernels write x = q + gate o beside o)
        (rv, lse_v, hv2), (ra, lse_a, ha2) = K.xattn_fwd2_gate(ag_v, hv, ha, ag_a, ha, hv, gate_v, gate_a)
        return hv2, ha2, (rv, ra, lse_v, lse_a)
    if PAIR_EW:
        (rv, lse_v), (ra, lse_a) = K.attn_fwd2(ag_v, hv, ha, ha, ag_a, ha, hv, hv)
    else:
        rv, lse_v = K.attn_fwd(ag_v, hv, ha, ha, want_lse=save)
        ra, lse_a = K.attn_fwd(ag_a, ha, hv, hv, want_lse=save)
    hv2, ha2 = _gate2(hv, rv, gate_v, ha, ra, gate_a)
    return hv2, ha2, (rv, ra, lse_v, lse_a)


def _cross_modal_bwd(spec, BT, hv, ha, gate_v, gate_a, window, g, saved, dhv2, dha2, dgate_v, dgate_a, geoms=None, zs=None, outs=None):
    """Returns (dhv, dha) = gradients wrt the pre-fusion hidden states; with zs = (Z_v, Z_a), the saved activation derivatives of
    the adapters' D_fc1, the gradients wrt the D_fc1 pre-activations instead (the join and the activation backward in one pass)."""
    mg = saved[4] if len(saved) == 5 else None
    rv, ra, lse_v, lse_a = saved[:4]
    if isinstance(mg, K.XsGeom):                                                      # the ViT pair on small frames: one merged backward launch
        if dgate_v is None:
            dgate_v = torch.zeros(1, dtype=F32, device=hv.device)
        if dgate_a is None:
            dgate_a = torch.zeros(1, dtype=F32, device=hv.device)
        drv, dra = _gate_bwd2(dhv2, rv, gate_v, dgate_v, dha2, ra, gate_a, dgate_a)
        G_v, G_a = K.xsmall_bwd(mg, hv, ha, rv, ra, lse_v, lse_a, drv, dra)
        if zs is None:
            return K.add(dhv2, G_v), K.add(dha2, G_a)
        return K.add3_mul2(dhv2, G_v, None, zs[0], dha2, G_a, None, zs[1], outs=outs)
    if isinstance(mg, K.WinGeom):                                                     # whole-window kernels, dK <- dK + dV
        if dgate_v is None:
            dgate_v = torch.zeros(1, dtype=F32, device=hv.device)
        if dgate_a is None:
            dgate_a = torch.zeros(1, dtype=F32, device=hv.device)
        if XWIN_PAIR and PAIR_EW and hv.shape == ha.shape and hv.stride(0) == ha.stride(0) and dhv2.stride(0) == dha2.stride(0) and dhv2.stride(0) % 8 == 0:
            # round 6b: one launch for both directions; the kernel scales by the gate and accumulates dgate = <dh', r> from the delta it computes anyway
            dq_v, dkv_a, dq_a, dkv_v = K.winattn_pair_bwd(mg, hv, ha, rv, ra, lse_v, lse_a, dhv2, dha2, gate_v, gate_a, dgate_v, dgate_a)
            return _join3(dhv2, dq_v, dkv_v, dha2, dq_a, dkv_a, zs, outs)
        drv, dra = _gate_bwd2(dhv2, rv, gate_v, dgate_v, dha2, ra, gate_a, dgate_a)
        dq_v, dkv_a, dq_a, dkv_v = (torch.empty_like(hv) for _ in range(4))
        K.winattn_bwd(mg, hv, ha, ha, rv, lse_v, drv, dQ=dq_v, dK=dkv_a, dV=None)      # direction a -> v
        K.winattn_bwd(mg, ha, hv, hv, ra, lse_a, dra, dQ=dq_a, dK=dkv_v, dV=None)      # direction v -> a
        return _join3(dhv2, dq_v, dkv_v, dha2, dq_a, dkv_a, zs, outs)
    if mg is None:
        ag_v, ag_a = geoms if geoms is not None else (_xattn_geom(spec, BT, hv.shape[1], window, g),) * 2
    if dgate_v is None:
        dgate_v = torch.zeros(1, dtype=F32, device=hv.device)
    if dgate_a is None:
        dgate_a = torch.zeros(1, dtype=F32, device=hv.device)
    if mg is None and XATTN_MERGED and XATTN_GATE and PAIR_EW:
        # round 6b: the frame-global pair's gates inside its merged backward (the preparation kernel rounds gate * d(h') as stg_gate_bwd2 did and sums dgate)
        pv, pa = (ag_v, hv, ha, rv, lse_v, dhv2), (ag_a, ha, hv, ra, lse_a, dha2)
        if K.xattn_pair_bwd_supported(pv, pa):
            gates = (gate_v, gate_a, dgate_v, dgate_a)
            if zs is not None and dhv2.stride(0) == dha2.stride(0) and zs[0].stride(0) == zs[1].stride(0) and (outs is None or outs[0].stride(0) == outs[1].stride(0)):
                return K.xattn_pair_bwd(pv, pa, join=(dhv2, zs[0], dha2, zs[1]), outs=outs, gates=gates)
            G_v, G_a = K.xattn_pair_bwd(pv, pa, gates=gates)
            if zs is None:
                return K.add(dhv2, G_v), K.add(dha2, G_a)
            if dhv2.shape == dha2.shape:
                return K.add3_mul2(dhv2, G_v, None, zs[0], dha2, G_a, None, zs[1], outs=outs)
            return K.add3_mul(dhv2, G_v, torch.zeros_like(G_v), zs[0], out=None if outs is None else outs[0]), \
                K.add3_mul(dha2, G_a, torch.zeros_like(G_a), zs[1], out=None if outs is None else outs[1])
    drv, dra = _gate_bwd2(dhv2, rv, gate_v, dgate_v, dha2, ra, gate_a, dgate_a)
    if mg is not None and MHA_MERGED and mg.window[2] == 0 and hv.stride(0) == ha.stride(0) and rv.stride(0) == ra.stride(0) and drv.stride(0) == dra.stride(0) \
            and 2 * mg.P < 65536:
        # round 6: frame-global pairs of wide adapters -- one pass per modality over the pair's shared score tiles (mha.hip mha_bwdm_kernel): G = dQ of a
        # tensor's own direction + dK + dV of the other (3 136-token frames at d_h = 96: 8.9 -> 7.1 ms per pair; the 49-token WINDOW pairs measured
        # 13 % slower on it and keep the two-direction path)
        G_v, G_a = K.mha_bwd_pair_merged(mg, (hv, ha, rv, lse_v, drv), (ha, hv, ra, lse_a, dra))
        if zs is None:
            return K.add(dhv2, G_v), K.add(dha2, G_a)
        if PAIR_EW and dhv2.shape == dha2.shape:
            return K.add3_mul2(dhv2, G_v, None, zs[0], dha2, G_a, None, zs[1], outs=outs)
        return K.add3_mul(dhv2, G_v, torch.zeros_like(G_v), zs[0], out=None if outs is None else outs[0]), \
            K.add3_mul(dha2, G_a, torch.zeros_like(G_a), zs[1], out=None if outs is None else outs[1])
    if mg is not None:
        dq_v, dkv_a, dq_a, dkv_v = (torch.empty_like(hv) for _ in range(4))
        if MHA_PAIR and hv.stride(0) == ha.stride(0) and rv.stride(0) == ra.stride(0) and drv.stride(0) == dra.stride(0):
            K.mha_bwd_pair(mg, (hv, ha, ha, rv, lse_v, drv, dq_v, dkv_a, None), (ha, hv, hv, ra, lse_a, dra, dq_a, dkv_v, None))
        else:
            K.mha_bwd(mg, hv, ha, ha, rv, lse_v, drv, dQ=dq_v, dK=dkv_a, dV=None)           # direction a -> v
            K.mha_bwd(mg, ha, hv, hv, ra, lse_a, dra, dQ=dq_a, dK=dkv_v, dV=None)           # direction v -> a
        return _join3(dhv2, dq_v, dkv_v, dha2, dq_a, dkv_a, zs, outs)
    pv, pa = (ag_v, hv, ha, rv, lse_v, drv), (ag_a, ha, hv, ra, lse_a, dra)
    if XATTN_MERGED and K.xattn_pair_bwd_supported(pv, pa):
        # one pass per modality over the pair's shared score tiles (one exponential per score): G = dQ (own direction) + dK + dV (other)
        if zs is not None and dhv2.stride(0) == dha2.stride(0) and zs[0].stride(0) == zs[1].stride(0) and \
                (outs is None or outs[0].stride(0) == outs[1].stride(0)):
            return K.xattn_pair_bwd(pv, pa, join=(dhv2, zs[0], dha2, zs[1]), outs=outs)      # ... and the join (dX + G) * act' in the same launch
        G_v, G_a = K.xattn_pair_bwd(pv, pa)
        if zs is None:
            return K.add(dhv2, G_v), K.add(dha2, G_a)
        if PAIR_EW and dhv2.shape == dha2.shape:
            return K.add3_mul2(dhv2, G_v, None, zs[0], dha2, G_a, None, zs[1], outs=outs)
        z_ = torch.zeros_like(G_v)
        return K.add3_mul(dhv2, G_v, z_, zs[0], out=None if outs is None else outs[0]), \
            K.add3_mul(dha2, G_a, torch.zeros_like(G_a), zs[1], out=None if outs is None else outs[1])
    if PAIR_EW:
        (dq_v, dkv_a), (dq_a, dkv_v) = K.attn_bwd2(pv, pa)
    else:
        dq_v, dkv_a, _ = K.attn_bwd(ag_v, hv, ha, ha, rv, lse_v, drv, shared_kv=True)   # direction a -> v
        dq_a, dkv_v, _ = K.attn_bwd(ag_a, ha, hv, hv, ra, lse_a, dra, shared_kv=True)   # direction v -> a
    return _join3(dhv2, dq_v, dkv_v, dha2, dq_a, dkv_a, zs, outs)


def _slices(spec, R):
    nm = len(spec.mods)
    Rm = R // nm
    return Rm, [slice(i * Rm, (i + 1) * Rm) for i in range(nm)]


def _temporal_geom(spec, B, g, tbias, nm):
    return K.AttnGeom(nm * B * spec.N, spec.heads, spec.T, spec.hd, G=spec.N, outer=spec.T * spec.N, temporal=spec.N,
                      scale=spec.hd ** -0.5, bias=tbias, bias_div=B * spec.N, bias_mod=nm)


def _window_geom(spec, BT, g, sbias, nm):
    P = nm * BT * spec.nW
    return K.AttnGeom(P, spec.heads, spec.ws * spec.ws, spec.hd, G=spec.nW, outer=spec.N, window=(spec.H, spec.W, spec.ws, spec.shift),
                      scale=spec.hd ** -0.5, bias=sbias, bias_div=P, bias_mod=1, mask=g["mask"])


class _LnOut:
    """Destination of a LayerNorm fused behind a residual join (K.up_ln_fwd): y / mean / rstd over ALL rows of the fused
    tensor, filled modality slice by modality slice."""

    def __init__(self, like, gamma, beta):
        R = like.shape[0]
        self.gamma, self.beta = gamma, beta
        self.y = torch.empty(like.shape, dtype=BF16, device=like.device)
        self.mean = torch.empty((R,), dtype=F32, device=like.device)
        self.rstd = torch.empty((R,), dtype=F32, device=like.device)

    def triple(self):
        return self.y, self.mean, self.rstd


def _join(Hh, A, out, rows, res32, res16, ln, **rs):
    """out[rows] = res32[rows] (+ res16[rows]) + rs * (Hh A.w2^T + A.b2)  (an adapter's D_fc2 and the residual join behind it);
    with ln (_LnOut) the LayerNorm that follows is computed in the same pass."""
    if ln is not None:
        K.up_ln_fwd(Hh, A.w2, A.b2, res32[rows], ln.gamma, ln.beta, res16=None if res16 is None else res16[rows], out=out[rows],
                    y_out=ln.y[rows], mean_out=ln.mean[rows], rstd_out=ln.rstd[rows], **rs)
    elif res16 is None:
        K.gemm_nt(Hh, A.w2, A.b2, out=out[rows], res1=res32[rows], **rs)
    else:
        K.gemm_nt(Hh, A.w2, A.b2, out=out[rows], res1=res16[rows], res2=res32[rows], **rs)


JOIN_PAIR = _cfg.opt("join_pair")       # 1: both modalities' joins / LayerNorm-backward + dgrad in one launch (A/B knob)


def _join_all(Hs, ads, out, sl, res32, res16, ln, rss=None, **rsg):
    """_join for every modality; with two modalities behind a fused LayerNorm (whole-tensor operands, 16-aligned halves) ONE launch."""
    nm = len(ads)
    if (JOIN_PAIR and ln is not None and nm == 2 and sl[0].start == 0 and sl[0].stop % 16 == 0 and sl[1].start == sl[0].stop
            and sl[1].stop == res32.shape[0] and Hs[0].shape[1] == Hs[1].shape[1] and Hs[0].stride(0) == Hs[1].stride(0)
            and ads[0].w2.stride(0) == ads[1].w2.stride(0) and (rss is None or (rss[0] is None) == (rss[1] is None))):
        kw = dict(row_scale=rss[0], row_scale2=rss[1], **rsg) if rss is not None and rss[0] is not None else {}
        K.up_ln_fwd_pair(Hs[0], Hs[1], ads[0].w2, ads[1].w2, ads[0].b2, ads[1].b2, res32, ln.gamma, ln.beta, res16=res16, out=out,
                         y_out=ln.y, mean_out=ln.mean, rstd_out=ln.rstd, **kw)
        return
    for i, A in enumerate(ads):
        kw = dict(row_scale=rss[i], **rsg) if rss is not None and rss[i] is not None else {}
        _join(Hs[i], A, out, sl[i], res32, res16, ln, **kw)


def _ln_bwd_join(dY, Xs, gamma, mean, rstd, add_to, sl, w2ts, rss=None, **rsg):
    """LayerNorm backward over all rows; with w2ts (one transposed D_fc2 shadow per modality) the down-projection of the
    adapter that consumes the result is computed in the same pass.  Returns (dX, [dH per modality] or None).
    mean is None: Xs holds the NORMALISED rows (bf16 x_hat, spec_xhat) and gamma == 1."""
    xh = mean is None
    fusable = w2ts is not None and USE_UPLN and (xh or Xs.dtype == F32) and all(K.ln_bwd_down_supported(Xs.shape[1], w.shape[0]) for w in w2ts)
    if not fusable:
        if xh:
            return K.layernorm_bwd_xhat(dY, Xs, rstd, add_to=add_to), None
        return K.layernorm_bwd(dY, Xs, gamma, mean, rstd, add_to=add_to), None
    dX = torch.empty(dY.shape, dtype=BF16, device=dY.device)
    if (JOIN_PAIR and len(w2ts) == 2 and sl[0].start == 0 and sl[0].stop % 16 == 0 and sl[1].start == sl[0].stop and sl[1].stop == dY.shape[0]
            and w2ts[0].shape == w2ts[1].shape and w2ts[0].stride(0) == w2ts[1].stride(0) and (rss is None or (rss[0] is None) == (rss[1] is None))):
        kw = dict(row_scale=rss[0], row_scale2=rss[1], **rsg) if rss is not None and rss[0] is not None else {}
        _, dh = K.ln_bwd_down_pair(dY, Xs, None if xh else gamma, mean, rstd, w2ts[0], w2ts[1], sl[0].stop, add_to=add_to, dx_out=dX, **kw)
        return dX, [dh[sl[0]], dh[sl[1]]]
    dH = []
    for i, w in enumerate(w2ts):
        r = sl[i]
        kw = dict(row_scale=rss[i], **rsg) if rss is not None and rss[i] is not None else {}
        if xh:
            dH.append(K.ln_bwd_down_xhat(dY[r], Xs[r], rstd[r], w, add_to=None if add_to is None else add_to[r], dx_out=dX[r], **kw)[1])
        else:
            dH.append(K.ln_bwd_down(dY[r], Xs[r], gamma, mean[r], rstd[r], w, add_to=None if add_to is None else add_to[r], dx_out=dX[r], **kw)[1])
    return dX, dH


GEMM_SPLIT = _cfg.opt("gemm_split")     # 0 = one adapter GEMM launch per modality (A/B knob)


def _pairable(sl, ads):
    return GEMM_SPLIT and len(ads) == 2 and sl[0].start == 0 and sl[0].stop % 128 == 0 and sl[1].start == sl[0].stop and ads[0].dh == ads[1].dh


def _down_pair(src, sl, ads):
    """[(H, Z)] per modality: H = GELU(src[rows] D_fc1^T + b) and its saved derivative.  The two modalities' adapters are separate
    Linears over the two halves of the rows: ONE launch with two row groups (stg_gemm_nt split mode) where the halves are 128-aligned."""
    if _pairable(sl, ads):
        H, Z = K.gemm_nt(src, ads[0].w1, ads[0].b1, act=ACT_GELU, want_dact=True, split=(sl[0].stop, ads[1].w1, ads[1].b1))
        return [(H[r], Z[r]) for r in sl]
    return [K.gemm_nt(src[sl[i]], A.w1, A.b1, act=ACT_GELU, want_dact=True) for i, A in enumerate(ads)]


def _dz_buffers(like, sl, ads):
    """One [rows, d_h] buffer for both modalities' D_fc1 pre-activation gradients (so that their dgrad is one split launch), or None."""
    if not _pairable(sl, ads):
        return None, None
    buf = torch.empty((sl[1].stop, ads[0].dh), dtype=BF16, device=like.device)
    return buf, [buf[r] for r in sl]


def _up_dgrad_pair(dZ_all, dZs, ads, sl, out, res):
    """out[rows] = (res[rows] +) dZ D_fc1 -- the adapters' input gradient joined with the branch gradient it rides on."""
    if dZ_all is not None:
        K.gemm_nt(dZ_all, ads[0].w1t, out=out, res1=res, split=(sl[0].stop, ads[1].w1t, None))
        return
    for i, A in enumerate(ads):
        K.gemm_nt(dZs[i], A.w1t, out=out[sl[i]], res1=None if res is None else res[sl[i]])


def _ln_fusable(X, ads):
    return USE_UPLN and X.dtype == F32 and all(K.up_ln_supported(X.shape[1], A.dh) for A in ads)


def block_forward(X, spec, P, training, save, pool=None, pre=None, nxt=None):
    """SwinTransformerBlock.forward for every mode (Swin_AVE.py:393-813) on the fused fp32 token tensor.
    P: {name: tensor} (block_param_names + block_buffer_names).  Returns (X_out, saved-state dict or None).
    pre = (Y, mean, rstd): norm1 of X, already computed by the previous block's last residual join.
    nxt = {"gamma", "beta"} of the NEXT block's norm1: when the join kernel supports the shape, nxt["pre"] receives that
    block's `pre`."""
    R, C = X.shape
    assert C == spec.C and X.dtype == RESIDUAL_DTYPE
    Rm, sl = _slices(spec, R)
    nm = len(spec.mods)
    BT = Rm // spec.N
    B = BT // spec.T
    assert B * spec.T * spec.N == Rm, "input feature has wrong size"
    T, N, H = spec.T, spec.N, spec.heads
    g = geom(X.device, spec.H, spec.W, spec.ws, spec.shift, T)
    S = {}
    n1g, n1b = f32c(P["norm1.weight"]), f32c(P["norm1.bias"])
    n2g, n2b = f32c(P["norm2.weight"]), f32c(P["norm2.bias"])
    fp8 = getattr(spec, "fp8", False)
    xh = spec_xhat(spec)
    F1 = F2 = None
    if xh:          # the norms write x_hat; their affine lives in the folded qkv / fc1 weights (include/stgcma.h: stg_*_xhat)
        F1 = folded(P["attn.qkv.weight"], P["norm1.weight"], P["norm1.bias"], P["attn.qkv.bias"])
        F2 = folded(P["mlp.fc1.weight"], P["norm2.weight"], P["norm2.bias"], P["mlp.fc1.bias"])
        n1g, n1b = unit_affine(C, X.device)
        n2g, n2b = n1g, n1b
    wqkv, bqkv = P["attn.qkv.weight"], f32c(P["attn.qkv.bias"])
    wproj, bproj = P["attn.proj.weight"], f32c(P["attn.proj.bias"])
    gate_v, gate_a = f32c(P["gate_v"]), f32c(P["gate_a"])

    # ---------------- temporal attention + T_Adapter (even blocks; :705-716).  DropPath per (b, n) row.
    if spec.t_attn:
        dps = [drop_scale(spec.drop_path, B * N, X.device, training, pool) for _ in spec.mods]
        Y, mean, rstd = pre if pre is not None else K.layernorm_fwd(X, n1g, n1b, want_stats=save)
        pre = None
        QKV = K.gemm_nt(Y, F1.w, F1.b) if xh else frozen_gemm(Y, wqkv, bqkv, fp8=_f8(fp8, "qkv"))
        Yt = Y if xh else None                  # x_hat form: the backward reads this bf16 row instead of the fp32 residual row
        del Y
        tbias = torch.empty((nm, H, T * T), dtype=F32, device=X.device)
        for i, m in enumerate(spec.mods):
            tab = "attn.temporal_position_bias_table" + ("_audio" if m else "")
            K.bias_gather(f32c(P[tab]), P["attn.t_relative_coords" + ("_a" if m else "")], out=tbias[i])
        if USE_TATTN and K.tattn_supported(T, spec.hd):
            tgeo = K.TGeom(nm, B, T, N, H, spec.hd ** -0.5, tbias)
            AO, lse = K.tattn_fwd(tgeo, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:]), tgeo    # backward recomputes: no O / LSE kept
        else:
            AO, lse = K.attn_fwd(_temporal_geom(spec, B, g, tbias, nm), QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], want_lse=save)
        PO = frozen_gemm(AO, wproj, bproj, fp8=_f8(fp8, "proj"))
        if isinstance(lse, K.TGeom):
            AO = None
        X1 = torch.empty_like(X)
        hz = []
        ads = [_Adapter(P, "T_Adapter" + _SFX[m]) for m in spec.mods]
        pre = _LnOut(X, n1g, n1b) if _ln_fusable(X, ads) else None          # norm1 of the spatial pass
        hz = _down_pair(PO, sl, ads)
        _join_all([hz_[0] for hz_ in hz], ads, X1, sl, X, None, pre, dps, rs_outer=T * N, rs_inner=N)
        pre = pre.triple() if pre is not None else None
        if save:
            S["t"] = (Yt if xh else X, None if xh else mean, rstd, QKV, AO, lse, PO, hz, tbias, dps)
        del QKV, AO, PO, Yt
    else:
        X1 = X

    # ---------------- (shifted-)window attention + S_Adapter2 (window-level cross-modal when fusing) (:718-787)
    Y, mean, rstd = pre if pre is not None else K.layernorm_fwd(X1, n1g, n1b, want_stats=save)
    QKV = K.gemm_nt(Y, F1.w, F1.b) if xh else frozen_gemm(Y, wqkv, bqkv, fp8=_f8(fp8, "qkv"))
    Ys = Y if xh else None
    del Y
    if USE_WINATTN and K.winattn_supported(spec.ws * spec.ws, spec.hd):
        bm, bmT = win_tables(P["attn.relative_position_bias_table"], P["attn.relative_position_index"], g["mask"], spec.ws * spec.ws)
        sbias = K.WinGeom(nm * BT, H, spec.H, spec.W, spec.ws, spec.shift, spec.hd ** -0.5, bm, bmT)
        AO, lse = K.winattn_fwd(sbias, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], want_lse=save)
    else:
        sbias = K.bias_gather(f32c(P["attn.relative_position_bias_table"]), P["attn.relative_position_index"].reshape(-1))
        AO, lse = K.attn_fwd(_window_geom(spec, BT, g, sbias, nm), QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], want_lse=save)
    PO = frozen_gemm(AO, wproj, bproj, fp8=_f8(fp8, "proj"))
    ads = [_Adapter(P, "S_Adapter2" + _SFX[m]) for m in spec.mods]
    HZ = _down_pair(PO, sl, ads)
    xs = None
    if spec.fuse:
        Hv2, Ha2, xs = _cross_modal_fwd(spec, BT, HZ[0][0], HZ[1][0], gate_v, gate_a, True, g, save)
        H2 = [Hv2, Ha2]
    else:
        H2 = [hz_[0] for hz_ in HZ]
    X2 = torch.empty_like(X)
    ln2 = _LnOut(X, n2g, n2b) if _ln_fusable(X, ads) else None
    _join_all(H2, ads, X2, sl, X1, PO, ln2)
    if save:
        S["s"] = (Ys if xh else X1, None if xh else mean, rstd, QKV, AO, lse, PO, HZ, H2, xs, sbias)
    del QKV, AO, PO, HZ, H2, Ys

    # ---------------- FFN + S_Adapter (:790-811; parallel variant :438-440)
    Y, mean, rstd = ln2.triple() if ln2 is not None else K.layernorm_fwd(X2, n2g, n2b, want_stats=save)
    del ln2
    fused_mlp = MLP_FUSED and not any(_f8(fp8, s_, b_) for s_ in ("fc1", "fc2") for b_ in (False, True)) and K.mlp_fused_supported(C)
    if fused_mlp:
        M = K.mlp_fwd(Y, F2.w if xh else shadow(P["mlp.fc1.weight"]), F2.b if xh else f32c(P["mlp.fc1.bias"]),
                      shadow_mlp_w2(P["mlp.fc2.weight"]), f32c(P["mlp.fc2.bias"]))
        Zm = ("recompute", Y)                 # backward recomputes GELU' from norm2(x): Y stays alive instead of the 4C-wide derivative
    else:
        if xh:
            Hm, Zm = K.gemm_nt(Y, F2.w, F2.b, act=ACT_GELU, want_dact=MLP_DACT)
        else:
            Hm, Zm = frozen_gemm(Y, P["mlp.fc1.weight"], f32c(P["mlp.fc1.bias"]), fp8=_f8(fp8, "fc1"), act=ACT_GELU, want_dact=MLP_DACT)
        M = frozen_gemm(Hm, P["mlp.fc2.weight"], f32c(P["mlp.fc2.bias"]), fp8=_f8(fp8, "fc2"))
        del Hm
    Yf = Y if xh else None
    ads = [_Adapter(P, "S_Adapter" + _SFX[m]) for m in spec.mods]
    X3 = torch.empty_like(X)
    ln3 = _LnOut(X, nxt["gamma"], nxt["beta"]) if nxt is not None and _ln_fusable(X, ads) else None
    if spec.parallel:
        # x + mlp(xn) + drop_path(0.5 * S_Adapter(xn)): DropPath per frame (dim 0 of the (BT, N, C) tensor)
        dpf = drop_scale(spec.drop_path, BT, X.device, training, pool)
        rs = (0.5 * dpf) if dpf is not None else torch.full((BT,), 0.5, dtype=F32, device=X.device)
        A = ads[0]
        Ha_, Za_ = K.gemm_nt(Y, A.w1, A.b1, act=ACT_GELU, want_dact=True)
        _join(Ha_, A, X3, slice(None), X2, M, ln3, row_scale=rs, rs_outer=N, rs_inner=1)
        if save:
            S["f"] = (X2, mean, rstd, Zm, Y, Ha_, Za_, rs)
    else:
        del Y
        HZ = _down_pair(M, sl, ads)
        xs = None
        if spec.fuse:
            Hv2, Ha2, xs = _cross_modal_fwd(spec, BT, HZ[0][0], HZ[1][0], gate_v, gate_a, False, g, save)
            H2 = [Hv2, Ha2]
        else:
            H2 = [hz_[0] for hz_ in HZ]
        _join_all(H2, ads, X3, sl, X2, M, ln3)
        if save:
            S["f"] = (Yf if xh else X2, None if xh else mean, rstd, Zm, M, HZ, H2, xs)
    if ln3 is not None:
        nxt["pre"] = ln3.triple()
    return X3, (S if save else None)


def _mlp_bwd_fused(P, Y, dM, F2=None):
    if F2 is not None:                                   # x_hat form: Y is the normalised row, fc1 carries norm2's affine
        return K.mlp_bwd(Y, dM, F2.w, F2.b, shadow(P["mlp.fc2.weight"], True))
    return K.mlp_bwd(Y, dM, shadow(P["mlp.fc1.weight"]), f32c(P["mlp.fc1.bias"]), shadow(P["mlp.fc2.weight"], True))


def block_backward(S, spec, P, need, prefix, dX3, arena=None, dH_in=None, prev=None, need_dx0=True):
    """Backward of block_forward.  dX3: bf16 [R, C].  Returns (dX0 bf16, {param name: fp32 grad}, dH_prev).
    dH_in: this block's S_Adapter dgrad (dX3 . D_fc2), already computed by the LayerNorm backward that produced dX3.
    prev: transposed D_fc2 shadows of the PREVIOUS block's S_Adapters; dH_prev is then their dgrad of dX0 (else None).
    need_dx0=False: nothing trainable lies below this block (the first block behind the frozen patch embedding): the qkv
    data-gradient GEMM and the LayerNorm backward of its FIRST attention pass, which only produce dX0, are skipped (dX0 = None)."""
    R, C = dX3.shape
    Rm, sl = _slices(spec, R)
    nm = len(spec.mods)
    BT = Rm // spec.N
    B = BT // spec.T
    T, N, H = spec.T, spec.N, spec.heads
    g = geom(dX3.device, spec.H, spec.W, spec.ws, spec.shift, T)
    G = _Grads(P, need, prefix, arena)
    gate_v, gate_a = f32c(P["gate_v"]), f32c(P["gate_a"])
    dgv, dga = G.buf("gate_v"), G.buf("gate_a")
    n1g = f32c(P["norm1.weight"])
    fp8 = getattr(spec, "fp8", False)
    wqkv, wproj = P["attn.qkv.weight"], P["attn.proj.weight"]
    xh = spec_xhat(spec)
    F1 = folded(P["attn.qkv.weight"], P["norm1.weight"], P["norm1.bias"], P["attn.qkv.bias"]) if xh else None
    F2 = folded(P["mlp.fc1.weight"], P["norm2.weight"], P["norm2.bias"], P["mlp.fc1.bias"]) if xh else None

    # ---------------- FFN + S_Adapter
    ads = [_Adapter(P, "S_Adapter" + _SFX[m]) for m in spec.mods]
    if spec.parallel:
        X2, mean, rstd, Zm, Y, Ha_, Za_, rs = S.pop("f")
        A = ads[0]
        dHa = K.gemm_nt(dX3, A.w2t, row_scale=rs, rs_outer=N, rs_inner=1)
        dZa = K.act_bwd(dHa, Za_)
        _adapter_wgrad(G, A.name, dZa, Y, dX3, Ha_, rs=rs, rs_outer=N, rs_inner=1)
        dYa = K.gemm_nt(dZa, A.w1t)
        if isinstance(Zm, tuple):
            dY = K.add(_mlp_bwd_fused(P, Zm[1], dX3), dYa)
        else:
            dZm = frozen_gemm(dX3, P["mlp.fc2.weight"], t=True, fp8=_f8(fp8, "fc2", True), dact_src=Zm)
            dY = frozen_gemm(dZm, P["mlp.fc1.weight"], t=True, fp8=_f8(fp8, "fc1", True), res1=dYa)
            del dZm
        del Y, Ha_, Za_, dHa, dZa, dYa, Zm
    else:
        X2, mean, rstd, Zm, M, HZ, H2, xs = S.pop("f")
        dH2 = dH_in if dH_in is not None else [K.gemm_nt(dX3[sl[i]], A.w2t) for i, A in enumerate(ads)]
        dZ_all, dZo = _dz_buffers(dX3, sl, ads)
        if spec.fuse:
            dZs = list(_cross_modal_bwd(spec, BT, HZ[0][0], HZ[1][0], gate_v, gate_a, False, g, xs, dH2[0], dH2[1], dgv, dga,
                                        zs=(HZ[0][1], HZ[1][1]), outs=dZo))
        else:
            dZs = [K.act_bwd(dH2[i], HZ[i][1], out=None if dZo is None else dZo[i]) for i in range(len(ads))]
        dM = torch.empty_like(dX3)
        for i, A in enumerate(ads):
            _adapter_wgrad(G, A.name, dZs[i], M[sl[i]], dX3[sl[i]], H2[i])
        _up_dgrad_pair(dZ_all, dZs, ads, sl, dM, dX3)
        del HZ, H2, xs, dH2, dZs, M, dZ_all, dZo
        if isinstance(Zm, tuple):
            dY = _mlp_bwd_fused(P, Zm[1], dM, F2)
        else:
            dZm = frozen_gemm(dM, P["mlp.fc2.weight"], t=True, fp8=_f8(fp8, "fc2", True), dact_src=Zm)
            dY = K.gemm_nt(dZm, F2.wt) if xh else frozen_gemm(dZm, P["mlp.fc1.weight"], t=True, fp8=_f8(fp8, "fc1", True))
            del dZm
        del dM, Zm
    ads = [_Adapter(P, "S_Adapter2" + _SFX[m]) for m in spec.mods]
    dX2, dH2 = _ln_bwd_join(dY, X2, f32c(P["norm2.weight"]), mean, rstd, dX3, sl, [A.w2t for A in ads])
    del dY, X2, dX3

    # ---------------- window attention + S_Adapter2
    X1, mean, rstd, QKV, AO, lse, PO, HZ, H2, xs, sbias = S.pop("s")
    if dH2 is None:
        dH2 = [K.gemm_nt(dX2[sl[i]], A.w2t) for i, A in enumerate(ads)]
    dZ_all, dZo = _dz_buffers(dX2, sl, ads)
    if spec.fuse:
        dZs = list(_cross_modal_bwd(spec, BT, HZ[0][0], HZ[1][0], gate_v, gate_a, True, g, xs, dH2[0], dH2[1], dgv, dga,
                                    zs=(HZ[0][1], HZ[1][1]), outs=dZo))
    else:
        dZs = [K.act_bwd(dH2[i], HZ[i][1], out=None if dZo is None else dZo[i]) for i in range(len(ads))]
    dPO = torch.empty_like(dX2)
    for i, A in enumerate(ads):
        _adapter_wgrad(G, A.name, dZs[i], PO[sl[i]], dX2[sl[i]], H2[i])
    _up_dgrad_pair(dZ_all, dZs, ads, sl, dPO, dX2)
    del HZ, H2, xs, dH2, dZs, PO, dZ_all, dZo
    dAO = frozen_gemm(dPO, wproj, t=True, fp8=_f8(fp8, "proj", True))
    del dPO
    dQKV = torch.empty_like(QKV)
    if isinstance(sbias, K.WinGeom):
        K.winattn_bwd(sbias, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], AO, lse, dAO,
                      dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:])
    else:
        K.attn_bwd(_window_geom(spec, BT, g, sbias, nm), QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], AO, lse, dAO,
                   dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:])
    del QKV, AO, dAO
    dH_prev = None
    if not spec.t_attn and not need_dx0:
        G.flush()
        return None, G.g, None
    dY = K.gemm_nt(dQKV, F1.wt) if xh else frozen_gemm(dQKV, wqkv, t=True, fp8=_f8(fp8, "qkv", True))
    del dQKV
    if spec.t_attn:
        tads = [_Adapter(P, "T_Adapter" + _SFX[m]) for m in spec.mods]
        dX1, dHts = _ln_bwd_join(dY, X1, n1g, mean, rstd, dX2, sl, [A.w2t for A in tads], S["t"][9], rs_outer=T * N, rs_inner=N)
    else:
        dX1, dH_prev = _ln_bwd_join(dY, X1, n1g, mean, rstd, dX2, sl, prev)
    del dY, dX2, X1

    # ---------------- temporal attention + T_Adapter
    if spec.t_attn:
        X0, mean, rstd, QKV, AO, lse, PO, hz, tbias, dps = S.pop("t")
        dPO = torch.empty_like(dX1)
        dZ_all, dZo = _dz_buffers(dX1, sl, tads)
        dZts = []
        for i, m in enumerate(spec.mods):
            A = tads[i]
            Ht, Zt = hz[i]
            dHt = dHts[i] if dHts is not None else K.gemm_nt(dX1[sl[i]], A.w2t, row_scale=dps[i], rs_outer=T * N, rs_inner=N)
            dZt = K.act_bwd(dHt, Zt, out=None if dZo is None else dZo[i])
            _adapter_wgrad(G, A.name, dZt, PO[sl[i]], dX1[sl[i]], Ht, rs=dps[i], rs_outer=T * N, rs_inner=N)
            dZts.append(dZt)
        _up_dgrad_pair(dZ_all, dZts, tads, sl, dPO, None)
        del hz, PO, dZ_all, dZo, dZts
        dAO = frozen_gemm(dPO, wproj, t=True, fp8=_f8(fp8, "proj", True))
        del dPO
        tabs = [G.buf("attn.temporal_position_bias_table" + ("_audio" if m else "")) for m in spec.mods]
        dtb = torch.zeros_like(tbias) if any(t is not None for t in tabs) else None
        dQKV = torch.empty_like(QKV)
        if isinstance(lse, K.TGeom):
            K.tattn_bwd(lse, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], dAO,
                        dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:], dbias=dtb)
        else:
            K.attn_bwd(_temporal_geom(spec, B, g, tbias, nm), QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], AO, lse, dAO,
                       dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:], dbias=dtb)
        for i, m in enumerate(spec.mods):
            if tabs[i] is not None:
                K.bias_scatter(dtb[i], P["attn.t_relative_coords" + ("_a" if m else "")], tabs[i])
        del QKV, AO, dAO
        if not need_dx0:
            G.flush()
            return None, G.g, None
        dY = K.gemm_nt(dQKV, F1.wt) if xh else frozen_gemm(dQKV, wqkv, t=True, fp8=_f8(fp8, "qkv", True))
        del dQKV
        dX0, dH_prev = _ln_bwd_join(dY, X0, n1g, mean, rstd, dX1, sl, prev)
    else:
        dX0 = dX1
    G.flush()
    return dX0, G.g, dH_prev


# ------------------------------------------------------------------------------------------------ patch merging / embedding / heads
def merge_forward(X, H, W, P, save, fp8=False):
    """PatchMerging.forward on every frame of the fused tensor (Swin_AVE.py:958-981): 2x2 gather + LN(4C) + Linear(4C->2C)."""
    Y, mean, rstd = K.layernorm_fwd(X, f32c(P["norm.weight"]), f32c(P["norm.bias"]), gather4=(H, W), want_stats=save)
    out = frozen_gemm(Y, P["reduction.weight"], fp8=_f8(fp8, "merge"), out_dtype=RESIDUAL_DTYPE)
    return out, ((X, mean, rstd) if save else None)


def merge_backward(S, H, W, P, dout, fp8=False):
    X, mean, rstd = S
    dY = frozen_gemm(dout, P["reduction.weight"], t=True, fp8=_f8(fp8, "merge", True))
    return K.layernorm_bwd(dY, X, f32c(P["norm.weight"]), mean, rstd, gather4=(H, W))


def patch_embed_into(x5, proj_w, proj_b, norm_w, norm_b, out_rows):
    """PatchEmbed3D (Swin_AVE.py:1104-1124) for kernel == stride == (1,p,p): im2col gather -> GEMM(+bias) -> LayerNorm,
    written into `out_rows` ([B*T*Hp*Wp, E] slice of the fused fp32 token tensor).  Frozen in the reference recipe and the
    inputs carry no gradient, so there is no backward."""
    E, Cin, pd, ph, pw = proj_w.shape
    if pd != 1 or ph != pw:
        raise NotImplementedError("patch embedding supports patch_size (1, p, p) only")
    Kd = Cin * ph * pw
    Kpad = (Kd + 7) // 8 * 8
    if x5.dtype not in (F32, BF16):
        x5 = x5.float()
    cols = K.im2col_patch(x5.contiguous(), ph, Kpad)
    if norm_w is None:
        K.gemm_nt(cols, shadow(proj_w), f32c(proj_b), out=out_rows)
        return
    Y = K.gemm_nt(cols, shadow(proj_w), f32c(proj_b))
    K.layernorm_fwd(Y, f32c(norm_w), f32c(norm_b), want_stats=False, out=out_rows)


def head_fusion_forward(X, n_tok, P, training, drop_p, save, pool=None):
    """final norm -> token mean -> cat((a, v)) -> Linear -> Dropout -> Linear (Swin_AVE.py:1585-1599), fp32 logits."""
    R, C = X.shape
    Rm = R // 2
    BT = Rm // n_tok
    Y, mean, rstd = K.layernorm_fwd(X, f32c(P["norm.weight"]), f32c(P["norm.bias"]), want_stats=save)
    pooled = torch.empty((BT, 2 * C), dtype=BF16, device=X.device)
    K.meanpool_fwd(Y[Rm:], BT, n_tok, out=pooled[:, :C])      # audio first: torch.cat((a, v)) (:1596)
    K.meanpool_fwd(Y[:Rm], BT, n_tok, out=pooled[:, C:])
    del Y
    h0 = K.gemm_nt(pooled, shadow(P["mlp_head.0.weight"]), f32c(P["mlp_head.0.bias"]))
    mask = None
    if training and drop_p > 0:
        keep = 1.0 - drop_p
        mask = pool.take(h0.numel(), keep).view(h0.shape) if pool is not None else \
            torch.empty(h0.shape, dtype=F32, device=X.device).bernoulli_(keep).div_(keep)
    h0d = K.mul_mask(h0, mask) if mask is not None else h0
    logits = K.gemm_nt(h0d, shadow(P["mlp_head.2.weight"]), f32c(P["mlp_head.2.bias"]), out_dtype=F32)
    return logits, ((X, mean, rstd, pooled, h0d, mask, n_tok) if save else None)


def head_fusion_backward(S, P, need, dlogits, need_dx, arena=None):
    X, mean, rstd, pooled, h0d, mask, n_tok = S
    R, C = X.shape
    Rm = R // 2
    BT = Rm // n_tok
    w0, w2 = P["mlp_head.0.weight"], P["mlp_head.2.weight"]
    L = w2.shape[0]
    G = _Grads(P, need, "", arena)
    dl = K.cast_bf16(dlogits.float().contiguous())                   # [BT, L padded to 8]
    gw2, gb2 = G.buf("mlp_head.2.weight"), G.buf("mlp_head.2.bias")
    if gw2 is not None:
        K.wgrad_tn(dl, h0d, gw2, gb2, n1=L)
    dh0 = K.gemm_nt(dl, shadow(w2, True))                            # [BT, 512]; K = L padded
    if mask is not None:
        dh0 = K.mul_mask(dh0, mask)
    gw0, gb0 = G.buf("mlp_head.0.weight"), G.buf("mlp_head.0.bias")
    if gw0 is not None:
        K.wgrad_tn(dh0, pooled, gw0, gb0)
    dX = None
    if need_dx:
        dpool = K.gemm_nt(dh0, shadow(w0, True))                     # [BT, 2C]
        dY = torch.empty((R, C), dtype=BF16, device=X.device)
        K.meanpool_bwd(dpool[:, :C], BT, n_tok, out=dY[Rm:])
        K.meanpool_bwd(dpool[:, C:], BT, n_tok, out=dY[:Rm])
        dX = K.layernorm_bwd(dY, X, f32c(P["norm.weight"]), mean, rstd)
    return dX, G.g


def head_single_forward(X, n_tok, P, save):
    """final norm -> token mean -> mlp_head = LayerNorm + Linear (Swin_AVE.py:1494-1503, :1324-1325)."""
    R, C = X.shape
    BT = R // n_tok
    Y, mean, rstd = K.layernorm_fwd(X, f32c(P["norm.weight"]), f32c(P["norm.bias"]), want_stats=save)
    pooled = K.meanpool_fwd(Y, BT, n_tok)
    del Y
    Z, m2, r2 = K.layernorm_fwd(pooled, f32c(P["mlp_head.0.weight"]), f32c(P["mlp_head.0.bias"]), want_stats=save)
    logits = K.gemm_nt(Z, shadow(P["mlp_head.1.weight"]), f32c(P["mlp_head.1.bias"]), out_dtype=F32)
    return logits, ((X, mean, rstd, pooled, m2, r2, Z, n_tok) if save else None)


def head_single_backward(S, P, need, dlogits, need_dx, arena=None):
    X, mean, rstd, pooled, m2, r2, Z, n_tok = S
    R, C = X.shape
    BT = R // n_tok
    w1 = P["mlp_head.1.weight"]
    L = w1.shape[0]
    G = _Grads(P, need, "", arena)
    dl = K.cast_bf16(dlogits.float().contiguous())
    gw, gb = G.buf("mlp_head.1.weight"), G.buf("mlp_head.1.bias")
    if gw is not None:
        K.wgrad_tn(dl, Z, gw, gb, n1=L)
    dZ = K.gemm_nt(dl, shadow(w1, True))
    gg, gbeta = G.buf("mlp_head.0.weight"), G.buf("mlp_head.0.bias")
    if (gg is None) != (gbeta is None):
        raise NotImplementedError("mlp_head.0 weight and bias must be frozen/trained together")
    dpool = K.layernorm_bwd(dZ, pooled, f32c(P["mlp_head.0.weight"]), m2, r2, dgamma=gg, dbeta=gbeta)
    dX = None
    if need_dx:
        dY = K.meanpool_bwd(dpool, BT, n_tok)
        dX = K.layernorm_bwd(dY, X, f32c(P["norm.weight"]), mean, rstd)
    return dX, G.g


# ------------------------------------------------------------------------------------------------ autograd wrappers
class SwinBlockFn(torch.autograd.Function):
    """One block as its own autograd node (used when a block is called stand-alone).  autograd wants gradients in the
    dtype of the fp32 residual tensor, so the bf16 gradient stream is cast at the node boundary; the whole-model
    Function below keeps it in bf16 end to end."""

    @staticmethod
    def forward(ctx, X, spec, names, training, grad_on, *params):
        P = dict(zip(names, params))
        need = {n: bool(grad_on and f) for n, f in zip(names, ctx.needs_input_grad[5:])}
        _check_frozen(need, FROZEN_ONLY, "SwinTransformerBlock")
        save = bool(grad_on) and any(ctx.needs_input_grad)
        Xr = X if X.dtype == RESIDUAL_DTYPE else (K.cast_f32(X.contiguous()) if X.dtype == BF16 else X.float())
        out, S = block_forward(Xr.contiguous(), spec, P, training, save)
        ctx.S, ctx.P, ctx.spec, ctx.names, ctx.need, ctx.in_dtype = S, P, spec, names, need, X.dtype
        return out if X.dtype == RESIDUAL_DTYPE else out.to(X.dtype)

    @staticmethod
    def backward(ctx, dout):
        d = dout.contiguous()
        d = K.cast_bf16(d.float().reshape(d.shape[0], -1)) if d.dtype != BF16 else d
        dX0, g, _ = block_backward(ctx.S, ctx.spec, ctx.P, ctx.need, "", d)
        ctx.S = None
        dX0 = dX0 if ctx.in_dtype == BF16 else K.cast_f32(dX0).to(ctx.in_dtype)
        return (dX0, None, None, None, None) + tuple(g.get(n) for n in ctx.names)


SwinFusionBlockFn = SwinBlockFn


def _prev_down(tape):
    """Transposed D_fc2 shadows of the S_Adapters of the block below the one being differentiated (None when a PatchMerging
    or nothing lies below, or its adapter runs parallel to the MLP with a DropPath row scale of its own)."""
    if not USE_UPLN or not tape or tape[-1][0] != "block" or tape[-1][1].parallel:
        return None
    spec, Pb = tape[-1][1], tape[-1][3]
    return [_Adapter(Pb, "S_Adapter" + _SFX[m]).w2t for m in spec.mods]


def _next_norm1(st, j, P):
    """norm1 of the block after block j of a stage (None behind the last one: PatchMerging / the final norm follow); the unit affine
    when that block takes its norms in x_hat form (spec_xhat)."""
    if not USE_UPLN or j + 1 >= len(st["blocks"]):
        return None
    nxt_spec, nxt_pre = st["blocks"][j + 1]
    if spec_xhat(nxt_spec):
        w = P[nxt_pre + "norm1.weight"]
        g, b = unit_affine(w.shape[0], w.device)
        return {"gamma": g, "beta": b}
    return {"gamma": f32c(P[nxt_pre + "norm1.weight"]), "beta": f32c(P[nxt_pre + "norm1.bias"])}


def plain_block_forward(X, spec, P, training, pool=None):
    """The AVQA negative-video stream: the FROZEN Swin block -- window attention and FFN with drop_path on both residuals, no
    temporal attention, no adapters (AVQA/model/Swin_AVQAModel_V1.py:780-860).  Nothing trainable sits on or behind this
    stream (its input is the frozen patch embedding of a negative clip), so it is forward-only.  X: fp32 [BT*N, C]."""
    R, C = X.shape
    N = spec.N
    BT = R // N
    H = spec.heads
    g = geom(X.device, spec.H, spec.W, spec.ws, spec.shift, spec.T)
    dp1 = drop_scale(spec.drop_path, BT, X.device, training, pool)
    dp2 = drop_scale(spec.drop_path, BT, X.device, training, pool)
    Y, _, _ = K.layernorm_fwd(X, f32c(P["norm1.weight"]), f32c(P["norm1.bias"]), want_stats=False)
    fp8 = getattr(spec, "fp8", False)
    QKV = frozen_gemm(Y, P["attn.qkv.weight"], f32c(P["attn.qkv.bias"]), fp8=_f8(fp8, "qkv"))
    if USE_WINATTN and K.winattn_supported(spec.ws * spec.ws, spec.hd):
        bm, bmT = win_tables(P["attn.relative_position_bias_table"], P["attn.relative_position_index"], g["mask"], spec.ws * spec.ws)
        wg = K.WinGeom(BT, H, spec.H, spec.W, spec.ws, spec.shift, spec.hd ** -0.5, bm, bmT)
        AO, _ = K.winattn_fwd(wg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], want_lse=False)
    else:
        sbias = K.bias_gather(f32c(P["attn.relative_position_bias_table"]), P["attn.relative_position_index"].reshape(-1))
        AO, _ = K.attn_fwd(_window_geom(spec, BT, g, sbias, 1), QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], want_lse=False)
    X1 = frozen_gemm(AO, P["attn.proj.weight"], f32c(P["attn.proj.bias"]), fp8=_f8(fp8, "proj"), out_dtype=RESIDUAL_DTYPE, res1=X,
                   row_scale=dp1, rs_outer=N, rs_inner=1)
    del QKV, AO
    Y, _, _ = K.layernorm_fwd(X1, f32c(P["norm2.weight"]), f32c(P["norm2.bias"]), want_stats=False)
    Hm = frozen_gemm(Y, P["mlp.fc1.weight"], f32c(P["mlp.fc1.bias"]), fp8=_f8(fp8, "fc1"), act=ACT_GELU)
    return frozen_gemm(Hm, P["mlp.fc2.weight"], f32c(P["mlp.fc2.bias"]), fp8=_f8(fp8, "fc2"), out_dtype=RESIDUAL_DTYPE, res1=X1,
                     row_scale=dp2, rs_outer=N, rs_inner=1)


class SwinBackboneFn(torch.autograd.Function):
    """The `fusion` backbone of the AVS / AVQA models as ONE autograd node: patch embeds -> stages ->
    final norm, returning features instead of logits.
      outputs (all fp32): f_v [BT*N_last, C_last], f_a, then f_nega when a negative clip is given (AVQA,
      Swin_AVQAModel_V1.py:1742-1766), then -- when `taps` -- the video stream before each downsample (AVS,
      Swin_AVSModel.py:1190-1201, 1813-1821; views of the residual stream: do not modify in place).
    Gradients flow from f_v, f_a and the taps; the negative stream is forward-only (nothing trainable behind it)."""

    @staticmethod
    def forward(ctx, a, v, v_nega, plan, training, grad_on, taps, names, *params):
        P = dict(zip(names, params))
        need = {n: bool(grad_on and f) for n, f in zip(names, ctx.needs_input_grad[8:])}
        save = any(need.values())
        for n in names:
            if need[n] and not plan.trainable_ok(n):
                _check_frozen({n: True}, [n], "Swin backbone")
        refresh_shadows(plan, names, P, need)
        B, T = v.shape[0], v.shape[1]
        N0 = plan.n_patches
        Rm = B * T * N0
        dev = v.device
        X = torch.empty((2 * Rm, plan.embed_dim), dtype=RESIDUAL_DTYPE, device=dev)
        pe = ("patch_embed", "patch_embed_audio")
        for i, x5 in enumerate((v.permute(0, 2, 1, 3, 4), a.unsqueeze(1))):     # 'b t c h w -> b c t h w' (:1793 / :1742)
            patch_embed_into(x5, P[pe[i] + ".proj.weight"], P[pe[i] + ".proj.bias"], P.get(pe[i] + ".norm.weight"),
                             P.get(pe[i] + ".norm.bias"), X[i * Rm:(i + 1) * Rm])
            te = P.get("temporal_embedding_audio" if i else "temporal_embedding")
            if te is not None:       # t_relative=False (Swin_AVSModel_Base.py:1800-1806, Swin_AVQAModel_V1.py:1752-1758): v and a, not v_nega
                K.add_temporal(X[i * Rm:(i + 1) * Rm], f32c(te).reshape(T, -1), B, T, N0)
        Xn = None
        if v_nega is not None:
            Xn = torch.empty((Rm, plan.embed_dim), dtype=RESIDUAL_DTYPE, device=dev)
            patch_embed_into(v_nega.permute(0, 2, 1, 3, 4), P["patch_embed.proj.weight"], P["patch_embed.proj.bias"],
                             P.get("patch_embed.norm.weight"), P.get("patch_embed.norm.bias"), Xn)
        pool = None
        if training:
            req = []
            for st in plan.stages:
                for spec, _ in st["blocks"]:
                    req += block_drop_requests(spec, B)
                    if Xn is not None and spec.drop_path > 0.:
                        req += [(B * T, 1.0 - spec.drop_path)] * 2
            if req:
                pool = DropPool(req, dev)
        tape, tap_out = [], []
        plan_fp8 = getattr(plan, "fp8", False)
        for st in plan.stages:
            carry = None
            for j, (spec, pre) in enumerate(st["blocks"]):
                Pb = {n: P[pre + n] for n in st["names"][pre]}
                nxt = _next_norm1(st, j, P)
                X, S = block_forward(X, spec, Pb, training, save, pool, carry, nxt)
                carry = nxt.get("pre") if nxt is not None else None
                tape.append(("block", spec, pre, Pb, S))
                if Xn is not None:
                    Xn = plain_block_forward(Xn, spec, Pb, training, pool)
            if st["merge"] is not None:
                H, W, pre = st["merge"]
                Pm = {n: P[pre + n] for n in ("norm.weight", "norm.bias", "reduction.weight")}
                if taps:
                    tap_out.append(X[:X.shape[0] // 2])
                X, S = merge_forward(X, H, W, Pm, save, plan_fp8)
                tape.append(("merge", (H, W), pre, Pm, S))
                if Xn is not None:
                    Xn, _ = merge_forward(Xn, H, W, Pm, False, plan_fp8)
        ng, nb = f32c(P["norm.weight"]), f32c(P["norm.bias"])
        Fall, mean, rstd = K.layernorm_fwd(X, ng, nb, want_stats=save, out_dtype=F32)
        half = X.shape[0] // 2
        outs = [Fall[:half], Fall[half:]]
        if Xn is not None:
            Fn, _, _ = K.layernorm_fwd(Xn, ng, nb, want_stats=False, out_dtype=F32)
            outs.append(Fn)
            ctx.mark_non_differentiable(Fn)
        outs += tap_out
        ctx.tape, ctx.final, ctx.P, ctx.need, ctx.names = tape, (X, mean, rstd), P, need, names
        ctx.n_tap, ctx.has_nega = len(tap_out), Xn is not None
        ctx.geom0 = (B, T, N0)
        ctx.ddp = getattr(plan, "ddp", None)
        ctx.fp8 = plan_fp8
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        P, need = ctx.P, ctx.need
        X, mean, rstd = ctx.final
        ctx.final = None
        dev = X.device
        arena = GradArena(ctx.names, P, need, dev)
        grads = {}
        half = X.shape[0] // 2
        dY = torch.zeros(X.shape, dtype=BF16, device=dev) if (douts[0] is None or douts[1] is None) else \
            torch.empty(X.shape, dtype=BF16, device=dev)
        for i in range(2):
            if douts[i] is not None:
                dY[i * half:(i + 1) * half].copy_(douts[i])            # fp32 -> bf16 hand-over of the head's gradient
        G = _Grads(P, need, "", arena)
        dX = K.layernorm_bwd(dY, X, f32c(P["norm.weight"]), mean, rstd, dgamma=G.buf("norm.weight"), dbeta=G.buf("norm.bias"))
        grads.update(G.g)
        del dY, X
        dtaps = list(douts[2 + (1 if ctx.has_nega else 0):])
        tape = ctx.tape
        dH_carry = None
        need_emb = need.get("temporal_embedding", False) or need.get("temporal_embedding_audio", False)
        while tape:
            kind, spec, pre, Pl, S = tape.pop()
            if kind == "block":
                dX, g, dH_carry = block_backward(S, spec, Pl, need, pre, dX, arena, dH_carry, _prev_down(tape), need_dx0=bool(tape) or need_emb)
                for k, val in g.items():
                    grads[pre + k] = val
            else:
                dH_carry = None
                dX = merge_backward(S, spec[0], spec[1], Pl, dX, ctx.fp8)
                dt = dtaps.pop() if dtaps else None
                if dt is not None:                                       # the tap's gradient joins the video rows
                    dv = dX[:dX.shape[0] // 2]
                    K.add(dv, K.cast_bf16(dt.contiguous()), out=dv)
        B, T, N0 = ctx.geom0
        for i, name in enumerate(("temporal_embedding", "temporal_embedding_audio")):     # t_relative=False: as in SwinModelFn.backward
            if need.get(name, False):
                Rm = B * T * N0
                pooled = K.meanpool_fwd(dX[i * Rm:(i + 1) * Rm], B * T, N0, out_dtype=F32)
                g = arena.view(name, P[name])
                g.copy_(pooled.view(B, T, -1).sum(0).mul_(float(N0)).view(g.shape))
                grads[name] = g
        if ctx.ddp is not None:
            ctx.ddp.allreduce_(arena.flat, arena.n_real)
        return (None,) * 8 + tuple(grads.get(n) for n in ctx.names)


class SwinModelFn(torch.autograd.Function):
    """The whole Swin + STG-CMA forward as ONE autograd node: patch embeddings -> stages of blocks / merges -> head.
    Between blocks the residual stream stays fp32 and the gradient stream bf16 with no autograd bookkeeping or dtype
    casts; per-block saved state lives in ctx and is released block by block during backward."""

    @staticmethod
    def forward(ctx, a, v, plan, training, grad_on, names, *params):
        P = dict(zip(names, params))
        need = {n: bool(grad_on and f) for n, f in zip(names, ctx.needs_input_grad[6:])}
        save = any(need.values())
        for n in names:
            if need[n] and not plan.trainable_ok(n):
                _check_frozen({n: True}, [n], "SwinTransformer2D_Adapter_New")
        refresh_shadows(plan, names, P, need)
        mods = plan.mods
        src = v if 0 in mods else a
        B, T = src.shape[0], (src.shape[2] if 0 in mods else src.shape[1])
        N0 = plan.n_patches
        Rm = B * T * N0
        X = torch.empty((len(mods) * Rm, plan.embed_dim), dtype=RESIDUAL_DTYPE, device=src.device)
        for i, m in enumerate(mods):
            pe = "patch_embed_audio" if m else "patch_embed"
            x5 = a.unsqueeze(1) if m else v
            nw = P.get(pe + ".norm.weight")
            patch_embed_into(x5, P[pe + ".proj.weight"], P[pe + ".proj.bias"], nw, P.get(pe + ".norm.bias"), X[i * Rm:(i + 1) * Rm])
            te = P.get("temporal_embedding_audio" if m else "temporal_embedding")
            if te is not None:                       # t_relative=False: absolute temporal embedding behind the patch embedding (:1569-1576)
                K.add_temporal(X[i * Rm:(i + 1) * Rm], f32c(te).reshape(T, -1), B, T, N0)
        pool = None
        if training:
            req = [r for st in plan.stages for spec, _ in st["blocks"] for r in block_drop_requests(spec, B)]
            if len(mods) == 2 and plan.head_drop > 0:
                req.append((B * T * P["mlp_head.0.weight"].shape[0], 1.0 - plan.head_drop))
            if req:
                pool = DropPool(req, src.device)
        tape = []
        plan_fp8 = getattr(plan, "fp8", False)
        for st in plan.stages:
            carry = None
            for j, (spec, pre) in enumerate(st["blocks"]):
                Pb = {n: P[pre + n] for n in st["names"][pre]}
                nxt = _next_norm1(st, j, P)
                X, S = block_forward(X, spec, Pb, training, save, pool, carry, nxt)
                carry = nxt.get("pre") if nxt is not None else None
                tape.append(("block", spec, pre, Pb, S))
            if st["merge"] is not None:
                H, W, pre = st["merge"]
                Pm = {n: P[pre + n] for n in ("norm.weight", "norm.bias", "reduction.weight")}
                X, S = merge_forward(X, H, W, Pm, save, plan_fp8)
                tape.append(("merge", (H, W), pre, Pm, S))
        if len(mods) == 2:
            logits, S = head_fusion_forward(X, plan.n_tok_last, P, training, plan.head_drop, save, pool)
        else:
            logits, S = head_single_forward(X, plan.n_tok_last, P, save)
        ctx.tape, ctx.head, ctx.P, ctx.need, ctx.names, ctx.two = tape, S, P, need, names, len(mods) == 2
        ctx.ddp = getattr(plan, "ddp", None)
        ctx.fp8 = plan_fp8
        ctx.geom0 = (mods, B, T, N0)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        P, need = ctx.P, ctx.need
        grads = {}
        arena = GradArena(ctx.names, P, need, dlogits.device)
        if ctx.two:
            dX, g = head_fusion_backward(ctx.head, P, need, dlogits, True, arena)
        else:
            dX, g = head_single_backward(ctx.head, P, need, dlogits, True, arena)
        grads.update(g)
        ctx.head = None
        tape = ctx.tape
        dH_carry = None
        need_emb = need.get("temporal_embedding", False) or need.get("temporal_embedding_audio", False)
        while tape:
            kind, spec, pre, Pl, S = tape.pop()
            if kind == "block":
                # the bottom block's input gradient feeds only the absolute temporal embeddings (t_relative=False); otherwise nothing
                # trainable lies below it (frozen patch embedding, inputs without gradient)
                dX, g, dH_carry = block_backward(S, spec, Pl, need, pre, dX, arena, dH_carry, _prev_down(tape),
                                                 need_dx0=bool(tape) or need_emb)
                for k, val in g.items():
                    grads[pre + k] = val
            else:
                dH_carry = None
                dX = merge_backward(S, spec[0], spec[1], Pl, dX, ctx.fp8)
        mods, B, T, N0 = ctx.geom0
        for i, m in enumerate(mods):                 # t_relative=False: d emb[t] = sum over clips and tokens of dX (token mean x N0, then clips)
            name = "temporal_embedding_audio" if m else "temporal_embedding"
            if need.get(name, False):
                Rm = B * T * N0
                pooled = K.meanpool_fwd(dX[i * Rm:(i + 1) * Rm], B * T, N0, out_dtype=F32)
                g = arena.view(name, P[name])
                g.copy_(pooled.view(B, T, -1).sum(0).mul_(float(N0)).view(g.shape))
                grads[name] = g
        if ctx.ddp is not None:
            ctx.ddp.allreduce_(arena.flat, arena.n_real)      # one collective for every trainable gradient of the step
        return (None, None, None, None, None, None) + tuple(grads.get(n) for n in ctx.names)
