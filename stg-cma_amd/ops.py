"""Autograd glue for the Swin + STG-CMA hot path: block-level torch.autograd.Functions whose forward AND backward are
explicit sequences of libstgcma_hip.so launches (kernels.py).  No ATen compute kernel runs on the hot path: residual
joins, activation gradients, DropPath scaling, window (un)partitioning and the temporal rearranges are all fused into GEMM
epilogues, LayerNorm-backward `add_to`, or attention addressing.

Data layout: one fused token tensor X[2*BT*N, C] bf16 -- rows [0, BT*N) are video tokens, rows [BT*N, 2*BT*N) audio tokens,
each in the reference's '(b t) n c' order -- so every frozen (shared) weight runs ONE GEMM over both modalities
(Swin_AVE.py:743-745 calls self.attn twice with the same weights) while per-modality adapters address row slices.

Backward computes dgrad through the frozen backbone and wgrad only for tensors with requires_grad (adapters, gates,
temporal bias tables, head), mirroring what autograd does for the reference under its freeze filter
(traintest_adapt_ave29.py:38-61).
"""
import torch

from . import kernels as K
from .kernels import ACT_GELU, ACT_NONE, BF16, F32

# ------------------------------------------------------------------------------------------------ weight shadows
_shadow_cache = {}


def shadow(p, transpose=False):
    """bf16 copy (optionally transposed, trailing dim zero-padded to a multiple of 8) of an fp32 parameter, cached until the
    parameter's storage or version changes (optimizer steps bump the version; frozen weights are cast once)."""
    t = p.detach()
    if not t.is_cuda:
        raise RuntimeError("stg-cma_amd: parameters must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != F32:
        raise RuntimeError(f"stg-cma_amd: parameters are expected in fp32 (got {t.dtype}); bf16 shadows are made internally")
    key = (t.data_ptr(), bool(transpose), tuple(t.shape))
    hit = _shadow_cache.get(key)
    ver = p._version
    if hit is not None and hit[0] == ver:
        return hit[1]
    s = K.cast_bf16(t.contiguous(), transpose=transpose)
    _shadow_cache[key] = (ver, s)
    return s


def f32c(p):
    t = p.detach()
    if t.dtype != F32 or not t.is_cuda:
        raise RuntimeError("stg-cma_amd: expected an fp32 GPU parameter")
    return t.contiguous()


def clear_shadow_cache():
    _shadow_cache.clear()


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


def _check_frozen(need, names, who):
    """need: {param name: autograd wants its gradient}."""
    for n in names:
        if need.get(n, False):
            raise NotImplementedError(
                f"{who}: parameter '{n}' has requires_grad=True, but this build computes weight gradients only for the "
                f"adapter / gate / temporal-bias / head tensors (the reference's freeze_base=True recipe, "
                f"traintest_adapt_ave29.py:52-61). Freeze the backbone before calling forward.")


# ------------------------------------------------------------------------------------------------ fusion block
class FusionBlockSpec:
    """Static description of one SwinTransformerBlock in 'fusion_adapt' mode (Swin_AVE.py:317-391)."""

    def __init__(self, C, H, W, T, heads, ws, shift, t_attn):
        self.C, self.H, self.W, self.T, self.heads, self.ws, self.shift, self.t_attn = C, H, W, T, heads, ws, shift, t_attn
        self.N = H * W
        self.nW = (H // ws) * (W // ws)
        self.hd = C // heads


# Order of the parameter tensors handed to SwinFusionBlockFn (names relative to the block module)
def fusion_param_names(t_attn):
    names = ["norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
             "attn.relative_position_bias_table", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
             "mlp.fc2.weight", "mlp.fc2.bias", "gate_v", "gate_a"]
    ad = ["S_Adapter", "S_Adapter2", "S_Adapter_Audio", "S_Adapter2_Audio"]
    if t_attn:
        names += ["attn.temporal_position_bias_table", "attn.temporal_position_bias_table_audio"]
        ad += ["T_Adapter", "T_Adapter_Audio"]
    for a in ad:
        names += [f"{a}.D_fc1.weight", f"{a}.D_fc1.bias", f"{a}.D_fc2.weight", f"{a}.D_fc2.bias"]
    return names


FROZEN_ONLY = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
               "attn.relative_position_bias_table", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
               "mlp.fc2.weight", "mlp.fc2.bias")


class _Adapter:
    """Shadows + parameter handles of one Adapter (D_fc1 -> GELU -> D_fc2)."""

    def __init__(self, P, name):
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

    def __init__(self, P, need):
        self.P = P
        self.need = need
        self.g = {}

    def buf(self, name):
        p = self.P[name]
        if not self.need.get(name, False):
            return None
        if name not in self.g:
            self.g[name] = _zeros_like_f32(p)
        return self.g[name]


def _adapter_wgrad(G, name, dZ, X, dY2, H2, *, rs=None, rs_outer=1, rs_inner=1):
    """Accumulate dW/db of D_fc1 (dZ^T X) and D_fc2 (dY2^T H2) when they train."""
    w1, b1 = G.buf(name + ".D_fc1.weight"), G.buf(name + ".D_fc1.bias")
    if w1 is not None or b1 is not None:
        if w1 is None or b1 is None:
            raise NotImplementedError("adapter weight and bias must be frozen/trained together")
        K.wgrad_tn(dZ, X, w1, b1)
    w2, b2 = G.buf(name + ".D_fc2.weight"), G.buf(name + ".D_fc2.bias")
    if w2 is not None or b2 is not None:
        if w2 is None or b2 is None:
            raise NotImplementedError("adapter weight and bias must be frozen/trained together")
        K.wgrad_tn(dY2, H2, w2, b2, row_scale=rs, rs_outer=rs_outer, rs_inner=rs_inner)


def _xattn_geom(spec, BT, dh, window, g):
    if window:
        n = spec.ws * spec.ws
        return K.AttnGeom(BT * spec.nW, 1, n, dh, G=spec.nW, outer=spec.N, map_q=g["wmap"], n_kv=n, outer_kv=spec.N,
                          map_kv=g["wmap"], scale=1.0)
    return K.AttnGeom(BT, 1, spec.N, dh, G=1, outer=spec.N, n_kv=spec.N, outer_kv=spec.N, scale=1.0)


def _cross_modal_fwd(spec, BT, hv, ha, gate_v, gate_a, window, g, save):
    """h' = h + gate * softmax(h hother^T) hother, both directions (Swin_AVE.py:750-760 / :799-808)."""
    ag = _xattn_geom(spec, BT, hv.shape[1], window, g)
    rv, lse_v = K.attn_fwd(ag, hv, ha, ha, want_lse=save)
    ra, lse_a = K.attn_fwd(ag, ha, hv, hv, want_lse=save)
    hv2 = K.gate_fwd(hv, rv, gate_v)
    ha2 = K.gate_fwd(ha, ra, gate_a)
    return hv2, ha2, (rv, ra, lse_v, lse_a)


def _cross_modal_bwd(spec, BT, hv, ha, gate_v, gate_a, window, g, saved, dhv2, dha2, dgate_v, dgate_a):
    """Returns (dhv, dha) = gradients wrt the pre-fusion hidden states."""
    rv, ra, lse_v, lse_a = saved
    ag = _xattn_geom(spec, BT, hv.shape[1], window, g)
    if dgate_v is None:
        dgate_v = torch.zeros(1, dtype=F32, device=hv.device)
    if dgate_a is None:
        dgate_a = torch.zeros(1, dtype=F32, device=hv.device)
    drv = K.gate_bwd(dhv2, rv, gate_v, dgate_v)
    dra = K.gate_bwd(dha2, ra, gate_a, dgate_a)
    dq_v, dkv_a, _ = K.attn_bwd(ag, hv, ha, ha, rv, lse_v, drv, shared_kv=True)   # direction a -> v
    dq_a, dkv_v, _ = K.attn_bwd(ag, ha, hv, hv, ra, lse_a, dra, shared_kv=True)   # direction v -> a
    return K.add(dhv2, dq_v, dkv_v), K.add(dha2, dq_a, dkv_a)


class SwinFusionBlockFn(torch.autograd.Function):
    """SwinTransformerBlock.forward, mode 'fusion_adapt' (Swin_AVE.py:693-813), on the fused token tensor."""

    @staticmethod
    def forward(ctx, X, spec, names, dp_v, dp_a, *params):
        P = dict(zip(names, params))
        need = dict(zip(names, ctx.needs_input_grad[5:]))
        _check_frozen(need, FROZEN_ONLY, "SwinFusionBlock")
        save = any(ctx.needs_input_grad)
        R, C = X.shape
        Rm = R // 2
        BT = Rm // spec.N
        B = BT // spec.T
        T, N, H = spec.T, spec.N, spec.heads
        g = geom(X.device, spec.H, spec.W, spec.ws, spec.shift, T)
        S = {}
        n1g, n1b = f32c(P["norm1.weight"]), f32c(P["norm1.bias"])
        wqkv, bqkv = shadow(P["attn.qkv.weight"]), f32c(P["attn.qkv.bias"])
        wproj, bproj = shadow(P["attn.proj.weight"]), f32c(P["attn.proj.bias"])
        gate_v, gate_a = f32c(P["gate_v"]), f32c(P["gate_a"])
        sl = (slice(0, Rm), slice(Rm, R))
        dps = (dp_v, dp_a)

        # ---------------- temporal attention + T_Adapter (even blocks; :705-716)
        if spec.t_attn:
            Y, mean, rstd = K.layernorm_fwd(X, n1g, n1b, want_stats=save)
            QKV = K.gemm_nt(Y, wqkv, bqkv)
            del Y
            tbias = torch.empty((2, H, T * T), dtype=F32, device=X.device)
            K.bias_gather(f32c(P["attn.temporal_position_bias_table"]), P["_t_index"], out=tbias[0])
            K.bias_gather(f32c(P["attn.temporal_position_bias_table_audio"]), P["_t_index_a"], out=tbias[1])
            tg = K.AttnGeom(2 * B * N, H, T, spec.hd, G=N, outer=T * N, map_q=g["tmap"], scale=spec.hd ** -0.5,
                            bias=tbias, bias_div=B * N, bias_mod=2)
            AO, lse = K.attn_fwd(tg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], want_lse=save)
            PO = K.gemm_nt(AO, wproj, bproj)
            X1 = torch.empty_like(X)
            hz = []
            for m, an in enumerate(("T_Adapter", "T_Adapter_Audio")):
                A = _Adapter(P, an)
                Ht, Zt = K.gemm_nt(PO[sl[m]], A.w1, A.b1, act=ACT_GELU, want_preact=True)
                K.gemm_nt(Ht, A.w2, A.b2, out=X1[sl[m]], res1=X[sl[m]], row_scale=dps[m], rs_outer=T * N, rs_inner=N)
                hz.append((Ht, Zt))
            if save:
                S["t"] = (X, mean, rstd, QKV, AO, lse, PO, hz, tbias)
            del QKV, AO, PO
        else:
            X1 = X

        # ---------------- (shifted-)window attention + window-level cross-modal adapter (:718-787)
        Y, mean, rstd = K.layernorm_fwd(X1, n1g, n1b, want_stats=save)
        QKV = K.gemm_nt(Y, wqkv, bqkv)
        del Y
        nn_ = spec.ws * spec.ws
        sbias = K.bias_gather(f32c(P["attn.relative_position_bias_table"]), P["_rel_index"])
        wg = K.AttnGeom(2 * BT * spec.nW, H, nn_, spec.hd, G=spec.nW, outer=N, map_q=g["wmap"], scale=spec.hd ** -0.5,
                        bias=sbias, bias_div=2 * BT * spec.nW, bias_mod=1, mask=g["mask"])
        AO, lse = K.attn_fwd(wg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], want_lse=save)
        PO = K.gemm_nt(AO, wproj, bproj)
        Av, Aa = _Adapter(P, "S_Adapter2"), _Adapter(P, "S_Adapter2_Audio")
        Hv, Zv = K.gemm_nt(PO[sl[0]], Av.w1, Av.b1, act=ACT_GELU, want_preact=True)
        Ha, Za = K.gemm_nt(PO[sl[1]], Aa.w1, Aa.b1, act=ACT_GELU, want_preact=True)
        Hv2, Ha2, xs = _cross_modal_fwd(spec, BT, Hv, Ha, gate_v, gate_a, True, g, save)
        X2 = torch.empty_like(X)
        K.gemm_nt(Hv2, Av.w2, Av.b2, out=X2[sl[0]], res1=PO[sl[0]], res2=X1[sl[0]])
        K.gemm_nt(Ha2, Aa.w2, Aa.b2, out=X2[sl[1]], res1=PO[sl[1]], res2=X1[sl[1]])
        if save:
            S["s"] = (X1, mean, rstd, QKV, AO, lse, PO, Hv, Zv, Ha, Za, Hv2, Ha2, xs, sbias)
        del QKV, AO, PO

        # ---------------- FFN + frame-global cross-modal adapter on the MLP output (:790-811)
        Y, mean, rstd = K.layernorm_fwd(X2, f32c(P["norm2.weight"]), f32c(P["norm2.bias"]), want_stats=save)
        Hm, Zm = K.gemm_nt(Y, shadow(P["mlp.fc1.weight"]), f32c(P["mlp.fc1.bias"]), act=ACT_GELU, want_preact=True)
        del Y
        M = K.gemm_nt(Hm, shadow(P["mlp.fc2.weight"]), f32c(P["mlp.fc2.bias"]))
        del Hm
        Av, Aa = _Adapter(P, "S_Adapter"), _Adapter(P, "S_Adapter_Audio")
        Hv, Zv = K.gemm_nt(M[sl[0]], Av.w1, Av.b1, act=ACT_GELU, want_preact=True)
        Ha, Za = K.gemm_nt(M[sl[1]], Aa.w1, Aa.b1, act=ACT_GELU, want_preact=True)
        Hv2, Ha2, xs = _cross_modal_fwd(spec, BT, Hv, Ha, gate_v, gate_a, False, g, save)
        X3 = torch.empty_like(X)
        K.gemm_nt(Hv2, Av.w2, Av.b2, out=X3[sl[0]], res1=M[sl[0]], res2=X2[sl[0]])
        K.gemm_nt(Ha2, Aa.w2, Aa.b2, out=X3[sl[1]], res1=M[sl[1]], res2=X2[sl[1]])
        if save:
            S["f"] = (X2, mean, rstd, Zm, M, Hv, Zv, Ha, Za, Hv2, Ha2, xs)
            ctx.S, ctx.P, ctx.spec, ctx.names, ctx.dps, ctx.need = S, P, spec, names, dps, need
            ctx.dims = (R, C, Rm, BT, B)
        return X3

    @staticmethod
    def backward(ctx, dX3):
        S, P, spec, names, dps = ctx.S, ctx.P, ctx.spec, ctx.names, ctx.dps
        R, C, Rm, BT, B = ctx.dims
        T, N, H = spec.T, spec.N, spec.heads
        g = geom(dX3.device, spec.H, spec.W, spec.ws, spec.shift, T)
        dX3 = dX3.contiguous()
        G = _Grads(P, ctx.need)
        sl = (slice(0, Rm), slice(Rm, R))
        gate_v, gate_a = f32c(P["gate_v"]), f32c(P["gate_a"])
        dgv, dga = G.buf("gate_v"), G.buf("gate_a")
        n1g = f32c(P["norm1.weight"])
        wqkv_t, wproj_t = shadow(P["attn.qkv.weight"], True), shadow(P["attn.proj.weight"], True)

        # ---------------- FFN + global cross-modal
        X2, mean, rstd, Zm, M, Hv, Zv, Ha, Za, Hv2, Ha2, xs = S.pop("f")
        Av, Aa = _Adapter(P, "S_Adapter"), _Adapter(P, "S_Adapter_Audio")
        dHv2 = K.gemm_nt(dX3[sl[0]], Av.w2t)
        dHa2 = K.gemm_nt(dX3[sl[1]], Aa.w2t)
        dHv, dHa = _cross_modal_bwd(spec, BT, Hv, Ha, gate_v, gate_a, False, g, xs, dHv2, dHa2, dgv, dga)
        dZv, dZa = K.act_bwd(dHv, Zv, ACT_GELU), K.act_bwd(dHa, Za, ACT_GELU)
        _adapter_wgrad(G, "S_Adapter", dZv, M[sl[0]], dX3[sl[0]], Hv2)
        _adapter_wgrad(G, "S_Adapter_Audio", dZa, M[sl[1]], dX3[sl[1]], Ha2)
        dM = torch.empty_like(dX3)
        K.gemm_nt(dZv, Av.w1t, out=dM[sl[0]], res1=dX3[sl[0]])
        K.gemm_nt(dZa, Aa.w1t, out=dM[sl[1]], res1=dX3[sl[1]])
        del Hv, Zv, Ha, Za, Hv2, Ha2, xs, dHv2, dHa2, dHv, dHa, dZv, dZa, M
        dZm = K.gemm_nt(dM, shadow(P["mlp.fc2.weight"], True), dact_src=Zm, act_bwd=ACT_GELU)
        del dM, Zm
        dY = K.gemm_nt(dZm, shadow(P["mlp.fc1.weight"], True))
        del dZm
        dX2 = K.layernorm_bwd(dY, X2, f32c(P["norm2.weight"]), mean, rstd, add_to=dX3)
        del dY, X2, dX3

        # ---------------- window attention + window-level cross-modal
        X1, mean, rstd, QKV, AO, lse, PO, Hv, Zv, Ha, Za, Hv2, Ha2, xs, sbias = S.pop("s")
        Av, Aa = _Adapter(P, "S_Adapter2"), _Adapter(P, "S_Adapter2_Audio")
        dHv2 = K.gemm_nt(dX2[sl[0]], Av.w2t)
        dHa2 = K.gemm_nt(dX2[sl[1]], Aa.w2t)
        dHv, dHa = _cross_modal_bwd(spec, BT, Hv, Ha, gate_v, gate_a, True, g, xs, dHv2, dHa2, dgv, dga)
        dZv, dZa = K.act_bwd(dHv, Zv, ACT_GELU), K.act_bwd(dHa, Za, ACT_GELU)
        _adapter_wgrad(G, "S_Adapter2", dZv, PO[sl[0]], dX2[sl[0]], Hv2)
        _adapter_wgrad(G, "S_Adapter2_Audio", dZa, PO[sl[1]], dX2[sl[1]], Ha2)
        dPO = torch.empty_like(dX2)
        K.gemm_nt(dZv, Av.w1t, out=dPO[sl[0]], res1=dX2[sl[0]])
        K.gemm_nt(dZa, Aa.w1t, out=dPO[sl[1]], res1=dX2[sl[1]])
        del Hv, Zv, Ha, Za, Hv2, Ha2, xs, dHv2, dHa2, dHv, dHa, dZv, dZa, PO
        dAO = K.gemm_nt(dPO, wproj_t)
        del dPO
        nn_ = spec.ws * spec.ws
        wg = K.AttnGeom(2 * BT * spec.nW, H, nn_, spec.hd, G=spec.nW, outer=N, map_q=g["wmap"], scale=spec.hd ** -0.5,
                        bias=sbias, bias_div=2 * BT * spec.nW, bias_mod=1, mask=g["mask"])
        dQKV = torch.empty_like(QKV)
        K.attn_bwd(wg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], AO, lse, dAO,
                   dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:])
        del QKV, AO, dAO
        dY = K.gemm_nt(dQKV, wqkv_t)
        del dQKV
        dX1 = K.layernorm_bwd(dY, X1, n1g, mean, rstd, add_to=dX2)
        del dY, dX2, X1

        # ---------------- temporal attention + T_Adapter
        if spec.t_attn:
            X0, mean, rstd, QKV, AO, lse, PO, hz, tbias = S.pop("t")
            dPO = torch.empty_like(dX1)
            for m, an in enumerate(("T_Adapter", "T_Adapter_Audio")):
                A = _Adapter(P, an)
                Ht, Zt = hz[m]
                dHt = K.gemm_nt(dX1[sl[m]], A.w2t, row_scale=dps[m], rs_outer=T * N, rs_inner=N)
                dZt = K.act_bwd(dHt, Zt, ACT_GELU)
                _adapter_wgrad(G, an, dZt, PO[sl[m]], dX1[sl[m]], Ht, rs=dps[m], rs_outer=T * N, rs_inner=N)
                K.gemm_nt(dZt, A.w1t, out=dPO[sl[m]])
            del hz, PO
            dAO = K.gemm_nt(dPO, wproj_t)
            del dPO
            tg = K.AttnGeom(2 * B * N, H, T, spec.hd, G=N, outer=T * N, map_q=g["tmap"], scale=spec.hd ** -0.5,
                            bias=tbias, bias_div=B * N, bias_mod=2)
            tv, ta = G.buf("attn.temporal_position_bias_table"), G.buf("attn.temporal_position_bias_table_audio")
            dtb = torch.zeros_like(tbias) if (tv is not None or ta is not None) else None
            dQKV = torch.empty_like(QKV)
            K.attn_bwd(tg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], AO, lse, dAO,
                       dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:], dbias=dtb)
            if tv is not None:
                K.bias_scatter(dtb[0], P["_t_index"], tv)
            if ta is not None:
                K.bias_scatter(dtb[1], P["_t_index_a"], ta)
            del QKV, AO, dAO
            dY = K.gemm_nt(dQKV, wqkv_t)
            del dQKV
            dX0 = K.layernorm_bwd(dY, X0, n1g, mean, rstd, add_to=dX1)
        else:
            dX0 = dX1
        grads = tuple(G.g.get(n) for n in names)
        return (dX0, None, None, None, None) + grads


# ------------------------------------------------------------------------------------------------ patch merging
class PatchMergeFn(torch.autograd.Function):
    """PatchMerging.forward on every frame of both modalities (Swin_AVE.py:958-981): 2x2 gather + LN(4C) + Linear(4C->2C)."""

    @staticmethod
    def forward(ctx, X, H, W, norm_w, norm_b, red_w):
        if any(ctx.needs_input_grad[3:]):
            raise NotImplementedError("PatchMerging: norm / reduction must be frozen (backbone weights get no wgrad in this build)")
        save = ctx.needs_input_grad[0]
        Y, mean, rstd = K.layernorm_fwd(X, f32c(norm_w), f32c(norm_b), gather4=(H, W), want_stats=save)
        out = K.gemm_nt(Y, shadow(red_w))
        if save:
            ctx.saved = (X, mean, rstd, norm_w, red_w, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        X, mean, rstd, norm_w, red_w, H, W = ctx.saved
        dY = K.gemm_nt(dout.contiguous(), shadow(red_w, True))
        dX = K.layernorm_bwd(dY, X, f32c(norm_w), mean, rstd, gather4=(H, W))
        return dX, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ patch embedding (frozen, forward only)
def patch_embed_into(x5, proj_w, proj_b, norm_w, norm_b, out_rows):
    """PatchEmbed3D (Swin_AVE.py:1104-1124) for kernel == stride == (1,p,p): im2col gather -> GEMM(+bias) -> LayerNorm,
    written into `out_rows` ([B*T*Hp*Wp, E] slice of the fused token tensor).  Frozen in the reference recipe; inputs carry
    no gradient, so there is no backward."""
    for p in (proj_w, proj_b, norm_w, norm_b):
        if p is not None and p.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("patch embedding must be frozen (traintest_adapt_ave29.py:52-61)")
    E, Cin, pd, ph, pw = proj_w.shape
    if pd != 1 or ph != pw:
        raise NotImplementedError("patch embedding supports patch_size (1, p, p) only")
    Kd = Cin * ph * pw
    Kpad = (Kd + 7) // 8 * 8
    cols = K.im2col_patch(x5.contiguous(), ph, Kpad)
    if norm_w is None:
        K.gemm_nt(cols, shadow(proj_w), f32c(proj_b), out=out_rows)
        return
    Y = K.gemm_nt(cols, shadow(proj_w), f32c(proj_b))
    K.layernorm_fwd(Y, f32c(norm_w), f32c(norm_b), want_stats=False, out=out_rows)


# ------------------------------------------------------------------------------------------------ head
class FusionHeadFn(torch.autograd.Function):
    """final norm -> token mean -> cat((a, v)) -> Linear -> Dropout -> Linear (Swin_AVE.py:1585-1599), fp32 logits."""

    @staticmethod
    def forward(ctx, X, n_tok, drop_mask, norm_w, norm_b, w0, b0, w2, b2):
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            raise NotImplementedError("final norm must be frozen (it is in the reference recipe: 'norm' matches no trainable substring)")
        save = any(ctx.needs_input_grad)
        ctx.need = ctx.needs_input_grad
        R, C = X.shape
        Rm = R // 2
        BT = Rm // n_tok
        Y, mean, rstd = K.layernorm_fwd(X, f32c(norm_w), f32c(norm_b), want_stats=save)
        pooled = torch.empty((BT, 2 * C), dtype=BF16, device=X.device)
        K.meanpool_fwd(Y[Rm:], BT, n_tok, out=pooled[:, :C])      # audio first: torch.cat((a, v)) (:1596)
        K.meanpool_fwd(Y[:Rm], BT, n_tok, out=pooled[:, C:])
        del Y
        h0 = K.gemm_nt(pooled, shadow(w0), f32c(b0))
        h0d = K.mul_mask(h0, drop_mask) if drop_mask is not None else h0
        logits = K.gemm_nt(h0d, shadow(w2), f32c(b2), out_dtype=F32)
        if save:
            ctx.saved = (X, mean, rstd, pooled, h0d, drop_mask, norm_w, w0, b0, w2, b2, n_tok)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        X, mean, rstd, pooled, h0d, drop_mask, norm_w, w0, b0, w2, b2, n_tok = ctx.saved
        R, C = X.shape
        Rm = R // 2
        BT = Rm // n_tok
        L = w2.shape[0]
        dl = K.cast_bf16(dlogits.float().contiguous())                   # [BT, L padded to 8]
        need = ctx.need
        gw2 = _zeros_like_f32(w2) if need[7] else None
        gb2 = _zeros_like_f32(b2) if need[8] else None
        if gw2 is not None:
            K.wgrad_tn(dl, h0d, gw2, gb2, n1=L)
        dh0 = K.gemm_nt(dl, shadow(w2, True))                            # [BT, 512]; K = L padded
        if drop_mask is not None:
            dh0 = K.mul_mask(dh0, drop_mask)
        gw0 = _zeros_like_f32(w0) if need[5] else None
        gb0 = _zeros_like_f32(b0) if need[6] else None
        if gw0 is not None:
            K.wgrad_tn(dh0, pooled, gw0, gb0)
        dX = None
        if need[0]:
            dpool = K.gemm_nt(dh0, shadow(w0, True))                     # [BT, 2C]
            dY = torch.empty((R, C), dtype=BF16, device=X.device)
            K.meanpool_bwd(dpool[:, :C], BT, n_tok, out=dY[Rm:])
            K.meanpool_bwd(dpool[:, C:], BT, n_tok, out=dY[:Rm])
            dX = K.layernorm_bwd(dY, X, f32c(norm_w), mean, rstd)
        return dX, None, None, None, None, gw0, gb0, gw2, gb2
