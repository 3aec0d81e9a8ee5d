"""Forward / backward orchestration of the CLIP-ViT + STG-CMA path (reference AVE/model/CLIP_AVE.py) on the HIP kernels.

Same conventions as ops.py (fp32 residual stream, bf16 branches and gradient stream, fused token tensor with the video
rows first, frozen weights shared by both modalities run as one GEMM), with the ViT specifics:
  * tokens are kept in '(b t) n d' order (the reference is sequence-first 'n (b t) d'; its rearranges, CLIP_AVE.py:369-377,
    become attention addressing: temporal map_kind 2, spatial identity);
  * video and audio have different token counts (197 vs 49), so each attention / adapter call addresses its own row range
    and the cross-modal attention runs with n != n_kv;
  * nn.MultiheadAttention's packed in_proj / out_proj, QuickGELU MLP, LayerNorm computed in fp32 (CLIP_AVE.py:33-43);
  * DropPath on the temporal residual is drawn per TOKEN INDEX (dim 0 of the 'n (b t) d' tensor, CLIP_AVE.py:372,377).
"""
import torch

from . import kernels as K
from .kernels import ACT_GELU, ACT_QUICKGELU, BF16, F32
from .ops import (RESIDUAL_DTYPE, refresh_shadows, GradArena, _Adapter, _adapter_wgrad, _check_frozen, _cross_modal_bwd, _cross_modal_fwd, _Grads,
                  _join, _LnOut, _ln_fusable, drop_scale, f32c, shadow)
from . import ops as _ops

_SFX = ("", "_Audio")
from . import config as _cfg
USE_MHA = _cfg.opt("mha")               # 0 = spatial ViT attention through the generic kernels (A/B knob)
USE_TATTN_VIT = _cfg.opt("tattn_vit")   # 0 = temporal ViT attention through the generic kernels
VIT_FROZEN = ("ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "attn.in_proj_weight", "attn.in_proj_bias",
              "attn.out_proj.weight", "attn.out_proj.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
              "mlp.c_proj.bias")


class VitBlockSpec:
    """ResidualAttentionBlock (CLIP_AVE.py:46-104).  n_tok: tokens per frame for each present modality."""

    def __init__(self, D, heads, T, n_tok, mode="fusion_adapt", drop_path=0.):
        self.D, self.heads, self.T, self.mode, self.drop_path = D, heads, T, mode, drop_path
        self.hd = D // heads
        self.mods = {"fusion_adapt": (0, 1), "multimodal_adapt_no_fusion": (0, 1), "video_adapt": (0,), "audio_adapt": (1,)}[mode]
        self.fuse = mode == "fusion_adapt"
        self.n_tok = tuple(n_tok)
        assert len(self.n_tok) == len(self.mods)
        if self.hd not in (16, 32, 48, 64, 96, 128):
            raise NotImplementedError(f"attention head dim {self.hd} unsupported (need 16/32/48/64/96/128)")


def vit_block_param_names(spec):
    names = list(VIT_FROZEN) + ["gate_v", "gate_a"]
    for m in spec.mods:
        for a in ("T_Adapter", "S_Adapter", "MLP_Adapter"):
            names += [f"{a}{_SFX[m]}.D_fc{i}.{w}" for i in (1, 2) for w in ("weight", "bias")]
    return names


def _ranges(spec, R):
    per = sum(spec.n_tok)
    BT = R // per
    assert BT * per == R and BT % spec.T == 0, "input feature has wrong size"
    sl, off = [], 0
    for n in spec.n_tok:
        sl.append(slice(off, off + BT * n))
        off += BT * n
    return BT, sl


def _mha(spec, BT, B, QKV, sl, temporal, save, out):
    """Self-attention of every modality's row range; temporal: sequences over T frames of one token (map_kind 2)."""
    D = spec.D
    geoms, lses = [], []
    for i, n in enumerate(spec.n_tok):
        q = QKV[sl[i]]
        if temporal and USE_TATTN_VIT and K.tattn_supported(spec.T, spec.hd):
            g = K.TGeom(1, B, spec.T, n, spec.heads, spec.hd ** -0.5, None, D=spec.hd)          # packed sequences, one-pass backward
            K.tattn_fwd(g, q[:, :D], q[:, D:2 * D], q[:, 2 * D:], out=out[sl[i]])
            geoms.append(g)
            lses.append(None)
            continue
        elif temporal:
            g = K.AttnGeom(B * n, spec.heads, spec.T, spec.hd, G=n, outer=spec.T * n, temporal=n, scale=spec.hd ** -0.5)
        elif USE_MHA and K.mha_supported(n, spec.hd):
            g = K.MhaGeom(BT, spec.heads, n, spec.hd, spec.hd ** -0.5)
            _, lse = K.mha_fwd(g, q[:, :D], q[:, D:2 * D], q[:, 2 * D:], out=out[sl[i]])
            geoms.append(g)
            lses.append(lse)
            continue
        else:
            g = K.AttnGeom(BT, spec.heads, n, spec.hd, G=1, outer=n, scale=spec.hd ** -0.5)
        _, lse = K.attn_fwd(g, q[:, :D], q[:, D:2 * D], q[:, 2 * D:], out=out[sl[i]], want_lse=save)
        geoms.append(g)
        lses.append(lse)
    return geoms, lses


def _mha_bwd(spec, QKV, AO, dAO, sl, geoms, lses):
    D = spec.D
    dQKV = torch.empty_like(QKV)
    for i in range(len(spec.n_tok)):
        q, dq = QKV[sl[i]], dQKV[sl[i]]
        if isinstance(geoms[i], K.TGeom):
            K.tattn_bwd(geoms[i], q[:, :D], q[:, D:2 * D], q[:, 2 * D:], dAO[sl[i]], dQ=dq[:, :D], dK=dq[:, D:2 * D], dV=dq[:, 2 * D:])
            continue
        if isinstance(geoms[i], K.MhaGeom):
            K.mha_bwd(geoms[i], q[:, :D], q[:, D:2 * D], q[:, 2 * D:], AO[sl[i]], lses[i], dAO[sl[i]],
                      dQ=dq[:, :D], dK=dq[:, D:2 * D], dV=dq[:, 2 * D:])
            continue
        K.attn_bwd(geoms[i], q[:, :D], q[:, D:2 * D], q[:, 2 * D:], AO[sl[i]], lses[i], dAO[sl[i]],
                   dQ=dq[:, :D], dK=dq[:, D:2 * D], dV=dq[:, 2 * D:])
    return dQKV


def _xgeoms(spec, BT, dh):
    nv, na = spec.n_tok
    return (K.AttnGeom(BT, 1, nv, dh, G=1, outer=nv, n_kv=na, outer_kv=na, scale=1.0),
            K.AttnGeom(BT, 1, na, dh, G=1, outer=na, n_kv=nv, outer_kv=nv, scale=1.0))


def vit_block_forward(X, spec, P, training, save, pre=None, nxt=None):
    """ResidualAttentionBlock.forward (CLIP_AVE.py:361-429).  pre / nxt: as in ops.block_forward -- the LayerNorm behind each of
    the three residual joins (ln_1 of the spatial pass, ln_2, ln_1 of the next block) rides on the join kernel."""
    R, D = X.shape
    assert D == spec.D and X.dtype == RESIDUAL_DTYPE
    BT, sl = _ranges(spec, R)
    B = BT // spec.T
    S = {}
    n1g, n1b = f32c(P["ln_1.weight"]), f32c(P["ln_1.bias"])
    wqkv, bqkv = shadow(P["attn.in_proj_weight"]), f32c(P["attn.in_proj_bias"])
    wout, bout = shadow(P["attn.out_proj.weight"]), f32c(P["attn.out_proj.bias"])
    gate_v, gate_a = f32c(P["gate_v"]), f32c(P["gate_a"])

    # ---- temporal adaptation (CLIP_AVE.py:369-377)
    dps = [drop_scale(spec.drop_path, n, X.device, training) for n in spec.n_tok]
    Y, mean, rstd = pre if pre is not None else K.layernorm_fwd(X, n1g, n1b, want_stats=save)
    QKV = K.gemm_nt(Y, wqkv, bqkv)
    del Y
    AO = torch.empty((R, D), dtype=BF16, device=X.device)
    tg, tlse = _mha(spec, BT, B, QKV, sl, True, save, AO)
    PO = K.gemm_nt(AO, wout, bout)
    X1 = torch.empty_like(X)
    hz = []
    tads = [_Adapter(P, "T_Adapter" + _SFX[m]) for m in spec.mods]
    ln1 = _LnOut(X, n1g, n1b) if _ln_fusable(X, tads) else None
    for i, A in enumerate(tads):
        Ht, Zt = K.gemm_nt(PO[sl[i]], A.w1, A.b1, act=ACT_GELU, want_dact=True)
        _join(Ht, A, X1, sl[i], X, None, ln1, row_scale=dps[i], rs_outer=R + 1, rs_inner=spec.n_tok[i])
        hz.append((Ht, Zt))
    if save:
        S["t"] = (X, mean, rstd, QKV, AO, tg, tlse, PO, hz, dps)
    del QKV, AO, PO

    # ---- spatial adaptation (+ cross-modal fusion of the adapter hidden states) (:379-401)
    Y, mean, rstd = ln1.triple() if ln1 is not None else K.layernorm_fwd(X1, n1g, n1b, want_stats=save)
    del ln1
    QKV = K.gemm_nt(Y, wqkv, bqkv)
    del Y
    AO = torch.empty((R, D), dtype=BF16, device=X.device)
    sg, slse = _mha(spec, BT, B, QKV, sl, False, save, AO)
    PO = K.gemm_nt(AO, wout, bout)
    ads = [_Adapter(P, "S_Adapter" + _SFX[m]) for m in spec.mods]
    HZ = [K.gemm_nt(PO[sl[i]], A.w1, A.b1, act=ACT_GELU, want_dact=True) for i, A in enumerate(ads)]
    xs = None
    if spec.fuse:
        Hv2, Ha2, xs = _cross_modal_fwd(None, BT, HZ[0][0], HZ[1][0], gate_v, gate_a, False, None, save,
                                        geoms=_xgeoms(spec, BT, ads[0].dh))
        H2 = [Hv2, Ha2]
    else:
        H2 = [h[0] for h in HZ]
    X2 = torch.empty_like(X)
    ln2 = _LnOut(X, f32c(P["ln_2.weight"]), f32c(P["ln_2.bias"])) if _ln_fusable(X, ads) else None
    for i, A in enumerate(ads):
        _join(H2[i], A, X2, sl[i], X1, PO, ln2)
    if save:
        S["s"] = (X1, mean, rstd, QKV, AO, sg, slse, PO, HZ, H2, xs)
    del QKV, AO, PO, HZ, H2

    # ---- joint adaptation: QuickGELU MLP, then MLP_Adapter on its output (:403-429)
    Y, mean, rstd = ln2.triple() if ln2 is not None else \
        K.layernorm_fwd(X2, f32c(P["ln_2.weight"]), f32c(P["ln_2.bias"]), want_stats=save)
    del ln2
    Hm, Zm = K.gemm_nt(Y, shadow(P["mlp.c_fc.weight"]), f32c(P["mlp.c_fc.bias"]), act=ACT_QUICKGELU, want_dact=_ops.MLP_DACT)
    del Y
    M = K.gemm_nt(Hm, shadow(P["mlp.c_proj.weight"]), f32c(P["mlp.c_proj.bias"]))
    del Hm
    ads = [_Adapter(P, "MLP_Adapter" + _SFX[m]) for m in spec.mods]
    HZ = [K.gemm_nt(M[sl[i]], A.w1, A.b1, act=ACT_GELU, want_dact=True) for i, A in enumerate(ads)]
    xs = None
    if spec.fuse:
        Hv2, Ha2, xs = _cross_modal_fwd(None, BT, HZ[0][0], HZ[1][0], gate_v, gate_a, False, None, save,
                                        geoms=_xgeoms(spec, BT, ads[0].dh))
        H2 = [Hv2, Ha2]
    else:
        H2 = [h[0] for h in HZ]
    X3 = torch.empty_like(X)
    ln3 = _LnOut(X, nxt["gamma"], nxt["beta"]) if nxt is not None and _ln_fusable(X, ads) else None
    for i, A in enumerate(ads):
        _join(H2[i], A, X3, sl[i], X2, M, ln3)
    if save:
        S["f"] = (X2, mean, rstd, Zm, M, HZ, H2, xs)
    if ln3 is not None:
        nxt["pre"] = ln3.triple()
    return X3, (S if save else None)


def _ln_bwd_down(dY, X, gamma, mean, rstd, add_to, sl, ads, dps=None, n_tok=None):
    """LayerNorm backward + the D_fc2 dgrad of the adapter that consumes its result, one launch per modality (or one for both where the
    halves are 16-aligned and no DropPath row scale is involved): stg_ln_bwd_down.  Returns (dX, [dH per modality] or None)."""
    if ads is None or not (_ops.USE_UPLN and X.dtype == torch.float32 and all(K.ln_bwd_down_supported(X.shape[1], A.dh) for A in ads)):
        return K.layernorm_bwd(dY, X, gamma, mean, rstd, add_to=add_to), None
    if dps is None or all(d is None for d in dps):
        return _ops._ln_bwd_join(dY, X, gamma, mean, rstd, add_to, sl, [A.w2t for A in ads])
    R = dY.shape[0]
    dX = torch.empty(dY.shape, dtype=BF16, device=dY.device)
    dH = []
    for i, A in enumerate(ads):
        r = sl[i]
        kw = dict(row_scale=dps[i], rs_outer=R + 1, rs_inner=n_tok[i]) if dps[i] is not None else {}
        dH.append(K.ln_bwd_down(dY[r], X[r], gamma, mean[r], rstd[r], A.w2t, add_to=None if add_to is None else add_to[r], dx_out=dX[r], **kw)[1])
    return dX, dH


def vit_block_backward(S, spec, P, need, prefix, dX3, arena=None, dH3=None, prev_ads=None):
    """dH3: the MLP_Adapter's D_fc2 dgrad of dX3, when the caller's LayerNorm backward already produced it; prev_ads: the adapters (of the
    block in front of this one) whose D_fc2 dgrad rides on this block's last LayerNorm backward -- then returns (dX0, grads, dH0)."""
    R, D = dX3.shape
    BT, sl = _ranges(spec, R)
    G = _Grads(P, need, prefix, arena)
    gate_v, gate_a = f32c(P["gate_v"]), f32c(P["gate_a"])
    dgv, dga = G.buf("gate_v"), G.buf("gate_a")
    n1g = f32c(P["ln_1.weight"])
    wqkv_t, wout_t = shadow(P["attn.in_proj_weight"], True), shadow(P["attn.out_proj.weight"], True)

    def adapters_bwd(base, HZ, H2, xs, Xin, dOut, dH2=None):
        """Gradient wrt the adapters' input tensor Xin (= res1 of the up-projection), joined with the direct path dOut."""
        ads = [_Adapter(P, base + _SFX[m]) for m in spec.mods]
        if dH2 is None:
            dH2 = [K.gemm_nt(dOut[sl[i]], A.w2t) for i, A in enumerate(ads)]
        if spec.fuse:
            dZs = list(_cross_modal_bwd(None, BT, HZ[0][0], HZ[1][0], gate_v, gate_a, False, None, xs, dH2[0], dH2[1], dgv, dga,
                                        geoms=_xgeoms(spec, BT, ads[0].dh), zs=(HZ[0][1], HZ[1][1])))
        else:
            dZs = [K.act_bwd(dH2[i], HZ[i][1]) for i in range(len(ads))]
        dIn = torch.empty_like(dOut)
        for i, A in enumerate(ads):
            dZ = dZs[i]
            _adapter_wgrad(G, A.name, dZ, Xin[sl[i]], dOut[sl[i]], H2[i])
            K.gemm_nt(dZ, A.w1t, out=dIn[sl[i]], res1=dOut[sl[i]])
        return dIn

    # ---- joint adaptation
    X2, mean, rstd, Zm, M, HZ, H2, xs = S.pop("f")
    dM = adapters_bwd("MLP_Adapter", HZ, H2, xs, M, dX3, dH3)
    del HZ, H2, xs, M
    dZm = K.gemm_nt(dM, shadow(P["mlp.c_proj.weight"], True), dact_src=Zm)
    del dM, Zm
    dY = K.gemm_nt(dZm, shadow(P["mlp.c_fc.weight"], True))
    del dZm
    dX2, dH = _ln_bwd_down(dY, X2, f32c(P["ln_2.weight"]), mean, rstd, dX3, sl, [_Adapter(P, "S_Adapter" + _SFX[m]) for m in spec.mods])
    del dY, X2, dX3

    # ---- spatial adaptation
    X1, mean, rstd, QKV, AO, sg, slse, PO, HZ, H2, xs = S.pop("s")
    dPO = adapters_bwd("S_Adapter", HZ, H2, xs, PO, dX2, dH)
    del HZ, H2, xs, PO
    dAO = K.gemm_nt(dPO, wout_t)
    del dPO
    dQKV = _mha_bwd(spec, QKV, AO, dAO, sl, sg, slse)
    del QKV, AO, dAO
    dY = K.gemm_nt(dQKV, wqkv_t)
    del dQKV
    X0, mean0, rstd0, QKV0, AO0, tg, tlse, PO0, hz, dps = S.pop("t")
    tads = [_Adapter(P, "T_Adapter" + _SFX[m]) for m in spec.mods]
    dX1, dH = _ln_bwd_down(dY, X1, n1g, mean, rstd, dX2, sl, tads, dps, spec.n_tok)
    del dY, dX2, X1

    # ---- temporal adaptation
    mean, rstd, QKV, AO, PO = mean0, rstd0, QKV0, AO0, PO0
    del QKV0, AO0, PO0
    dPO = torch.empty_like(dX1)
    for i, A in enumerate(tads):
        Ht, Zt = hz[i]
        kw = dict(row_scale=dps[i], rs_outer=R + 1, rs_inner=spec.n_tok[i])
        dHt = dH[i] if dH is not None else K.gemm_nt(dX1[sl[i]], A.w2t, **kw)
        dZt = K.act_bwd(dHt, Zt)
        _adapter_wgrad(G, A.name, dZt, PO[sl[i]], dX1[sl[i]], Ht, rs=dps[i], rs_outer=R + 1, rs_inner=spec.n_tok[i])
        K.gemm_nt(dZt, A.w1t, out=dPO[sl[i]])
    del hz, PO
    dAO = K.gemm_nt(dPO, wout_t)
    del dPO
    dQKV = _mha_bwd(spec, QKV, AO, dAO, sl, tg, tlse)
    del QKV, AO, dAO
    dY = K.gemm_nt(dQKV, wqkv_t)
    del dQKV
    dX0, dH0 = _ln_bwd_down(dY, X0, n1g, mean, rstd, dX1, sl, prev_ads)
    G.flush()
    if prev_ads is not None:
        return dX0, G.g, dH0
    return dX0, G.g


class VitBlockFn(torch.autograd.Function):
    """One ResidualAttentionBlock as its own autograd node (stand-alone use; gradient dtype cast at the node boundary)."""

    @staticmethod
    def forward(ctx, X, spec, names, training, grad_on, *params):
        P = dict(zip(names, params))
        need = {n: bool(grad_on and f) for n, f in zip(names, ctx.needs_input_grad[5:])}
        _check_frozen(need, VIT_FROZEN, "ResidualAttentionBlock")
        save = bool(grad_on) and any(ctx.needs_input_grad)
        Xr = X if X.dtype == RESIDUAL_DTYPE else K.cast_f32(X.contiguous())
        out, S = vit_block_forward(Xr.contiguous(), spec, P, training, save)
        ctx.S, ctx.P, ctx.spec, ctx.names, ctx.need, ctx.in_dtype = S, P, spec, names, need, X.dtype
        return out if X.dtype == RESIDUAL_DTYPE else out.to(X.dtype)

    @staticmethod
    def backward(ctx, dout):
        d = dout.contiguous()
        d = K.cast_bf16(d.float()) if d.dtype != BF16 else d
        dX0, g = vit_block_backward(ctx.S, ctx.spec, ctx.P, ctx.need, "", d)
        ctx.S = None
        dX0 = dX0 if ctx.in_dtype == BF16 else K.cast_f32(dX0)
        return (dX0, None, None, None, None) + tuple(g.get(n) for n in ctx.names)


# ------------------------------------------------------------------------------------------------ whole model
def _embed(P, x5, conv, pos, temb, T, save, out):
    """conv1 as im2col-GEMM -> cls/pos/temporal assembly -> ln_pre (CLIP_AVE.py:1091-1105).  Returns (X fp32, saved)."""
    w = P[conv]
    D, Cin, p, _ = w.shape
    B = x5.shape[0]
    Kd = Cin * p * p
    if x5.dtype not in (F32, BF16):
        x5 = x5.float()
    cols = K.im2col_patch(x5.contiguous(), p, (Kd + 7) // 8 * 8)
    tok = K.gemm_nt(cols, shadow(w))
    U = K.vit_embed(tok, f32c(P["class_embedding"]), f32c(P[pos]), f32c(P[temb]).reshape(T, D), B * T, T)
    _, mean, rstd = K.layernorm_fwd(U, f32c(P["ln_pre.weight"]), f32c(P["ln_pre.bias"]), want_stats=save, out=out)
    return (U, mean, rstd) if save else None


_onehot_cache = {}


def _t_onehot(BT, n, T, device):
    """bf16 one-hot [BT*n, 16] of the frame index of every token row: d temporal_embedding = onehot^T . dU (a wgrad)."""
    key = (str(device), BT, n, T)
    oh = _onehot_cache.get(key)
    if oh is None:
        t = (torch.arange(BT * n, device=device) // n) % T
        oh = torch.zeros((BT * n, (T + 7) // 8 * 8), dtype=BF16, device=device)
        oh[torch.arange(BT * n, device=device), t] = 1
        _onehot_cache[key] = oh
    return oh


class VitModelFn(torch.autograd.Function):
    """MM_CLIP_AVE.forward (CLIP_AVE.py:979-1140) as one autograd node."""

    @staticmethod
    def forward(ctx, a, v, plan, training, grad_on, names, *params):
        P = dict(zip(names, params))
        need = {n: bool(grad_on and f) for n, f in zip(names, ctx.needs_input_grad[6:])}
        save = any(need.values())
        for n in names:
            if need[n] and not plan.trainable_ok(n):
                _check_frozen({n: True}, [n], "MM_CLIP_AVE")
        refresh_shadows(plan, names, P, need)
        T = plan.T
        mods = plan.mods
        src = v if 0 in mods else a
        B = src.shape[0]
        n_tok = tuple(P["positional_embedding_audio" if m else "positional_embedding"].shape[0] for m in mods)
        D = P["class_embedding"].numel()
        X = torch.empty((B * T * sum(n_tok), D), dtype=RESIDUAL_DTYPE, device=src.device)
        emb, off = [], 0
        for i, m in enumerate(mods):
            rows = X[off:off + B * T * n_tok[i]]
            off += B * T * n_tok[i]
            if m == 0:
                emb.append(_embed(P, v, "conv1.weight", "positional_embedding", "temporal_embedding", T, save, rows))
            else:
                emb.append(_embed(P, a.unsqueeze(1), "conv1_audio.weight", "positional_embedding_audio",
                                  "temporal_embedding_audio", T, save, rows))
        tape = []
        blocks = list(plan.blocks(n_tok))
        carry = None
        for j, (spec, pre, bnames) in enumerate(blocks):
            Pb = {n: P[pre + n] for n in bnames}
            nxt = None
            if _ops.USE_UPLN and j + 1 < len(blocks):                    # ln_1 of the next block rides on this block's last join
                npre = blocks[j + 1][1]
                nxt = {"gamma": f32c(P[npre + "ln_1.weight"]), "beta": f32c(P[npre + "ln_1.bias"])}
            X, S = vit_block_forward(X, spec, Pb, training, save, carry, nxt)
            carry = nxt.get("pre") if nxt is not None else None
            tape.append((spec, pre, Pb, S))
        # ---- head: ln_post on the class tokens, cat((a, v)), mlp_head (CLIP_AVE.py:1128-1140)
        BT = B * T
        D = X.shape[1]
        R = X.shape[0]
        offs = [0]
        for n in n_tok:
            offs.append(offs[-1] + BT * n)
        two = len(mods) == 2
        pooled = torch.empty((BT, D * (2 if two else 1)), dtype=BF16, device=X.device)
        stats = []
        for i, m in enumerate(mods):
            cls_rows = X[offs[i]:offs[i + 1]].view(BT, n_tok[i] * D)[:, :D]
            dst = pooled[:, (D if m == 0 else 0):(2 * D if m == 0 else D)] if two else pooled
            _, mu, rs = K.layernorm_fwd(cls_rows, f32c(P["ln_post.weight"]), f32c(P["ln_post.bias"]), want_stats=save, out=dst)
            stats.append((mu, rs))
        mask = None
        if two:
            h0 = K.gemm_nt(pooled, shadow(P["mlp_head.0.weight"]), f32c(P["mlp_head.0.bias"]))
            if training and plan.head_drop > 0:
                keep = 1.0 - plan.head_drop
                mask = torch.empty(h0.shape, dtype=F32, device=X.device).bernoulli_(keep).div_(keep)
                h0 = K.mul_mask(h0, mask)
            logits = K.gemm_nt(h0, shadow(P["mlp_head.2.weight"]), f32c(P["mlp_head.2.bias"]), out_dtype=F32)
            head = (pooled, h0, mask, None, None)
        else:
            Z, m2, r2 = K.layernorm_fwd(pooled, f32c(P["mlp_head.0.weight"]), f32c(P["mlp_head.0.bias"]), want_stats=save)
            logits = K.gemm_nt(Z, shadow(P["mlp_head.1.weight"]), f32c(P["mlp_head.1.bias"]), out_dtype=F32)
            head = (pooled, Z, None, m2, r2)
        if save:
            ctx.saved = (tape, emb, X, stats, head, n_tok, offs, BT, T, two)
        ctx.P, ctx.need, ctx.names, ctx.mods, ctx.ddp = P, need, names, mods, getattr(plan, "ddp", None)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        P, need, names, mods = ctx.P, ctx.need, ctx.names, ctx.mods
        tape, emb, X, stats, head, n_tok, offs, BT, T, two = ctx.saved
        ctx.saved = None
        arena = GradArena(names, P, need, dlogits.device)
        G = _Grads(P, need, "", arena)
        D = X.shape[1]
        pooled, hmid, mask, m2, r2 = head
        dl = K.cast_bf16(dlogits.float().contiguous())
        if two:
            w2 = P["mlp_head.2.weight"]
            gw2, gb2 = G.buf("mlp_head.2.weight"), G.buf("mlp_head.2.bias")
            if gw2 is not None:
                K.wgrad_tn(dl, hmid, gw2, gb2, n1=w2.shape[0])
            dh0 = K.gemm_nt(dl, shadow(w2, True))
            if mask is not None:
                dh0 = K.mul_mask(dh0, mask)
            gw0, gb0 = G.buf("mlp_head.0.weight"), G.buf("mlp_head.0.bias")
            if gw0 is not None:
                K.wgrad_tn(dh0, pooled, gw0, gb0)
            dpool = K.gemm_nt(dh0, shadow(P["mlp_head.0.weight"], True))
        else:
            w1 = P["mlp_head.1.weight"]
            gw, gb = G.buf("mlp_head.1.weight"), G.buf("mlp_head.1.bias")
            if gw is not None:
                K.wgrad_tn(dl, hmid, gw, gb, n1=w1.shape[0])
            dZ = K.gemm_nt(dl, shadow(w1, True))
            dpool = K.layernorm_bwd(dZ, pooled, f32c(P["mlp_head.0.weight"]), m2, r2,
                                    dgamma=G.buf("mlp_head.0.weight"), dbeta=G.buf("mlp_head.0.bias"))
        # ln_post backward on the class-token rows only (every other row of dX is zero)
        dX = torch.zeros((X.shape[0], D), dtype=BF16, device=X.device)
        gpw, gpb = G.buf("ln_post.weight"), G.buf("ln_post.bias")
        for i, m in enumerate(mods):
            cls_rows = X[offs[i]:offs[i + 1]].view(BT, n_tok[i] * D)[:, :D]
            dsrc = dpool[:, (D if m == 0 else 0):(2 * D if m == 0 else D)] if two else dpool
            dcls = K.layernorm_bwd(dsrc, cls_rows, f32c(P["ln_post.weight"]), stats[i][0], stats[i][1], dgamma=gpw, dbeta=gpb)
            dX[offs[i]:offs[i + 1]].view(BT, n_tok[i] * D)[:, :D].copy_(dcls)
        grads = dict(G.g)
        dH = None
        while tape:
            spec, pre, Pb, S = tape.pop()
            # the MLP_Adapter dgrad of the block in front rides on this block's last LayerNorm backward (stg_ln_bwd_down)
            prev_ads = [_Adapter(tape[-1][2], "MLP_Adapter" + _SFX[m]) for m in tape[-1][0].mods] if tape else None
            if prev_ads is not None:
                dX, g, dH = vit_block_backward(S, spec, Pb, need, pre, dX, arena, dH3=dH, prev_ads=prev_ads)
            else:
                dX, g = vit_block_backward(S, spec, Pb, need, pre, dX, arena, dH3=dH)
            for k, val in g.items():
                grads[pre + k] = val
        # embeddings: ln_pre backward, then d temporal_embedding(_audio)[t] = sum over (b, n) of dU (the only trainable part)
        for i, m in enumerate(mods):
            tname = "temporal_embedding_audio" if m else "temporal_embedding"
            gt = G.buf(tname)
            if gt is None:
                continue
            U, mu, rs = emb[i]
            dU = K.layernorm_bwd(dX[offs[i]:offs[i + 1]], U, f32c(P["ln_pre.weight"]), mu, rs)
            K.wgrad_tn(_t_onehot(BT, n_tok[i], T, dU.device), dU, gt.view(T, D), None, n1=T)
            grads[tname] = gt
        if ctx.ddp is not None:
            ctx.ddp.allreduce_(arena.flat, arena.n_real)
        return (None, None, None, None, None, None) + tuple(grads.get(n) for n in names)
