"""ORACLE (test infrastructure, not product code): fp32 PyTorch-CPU restatement of the reference's Swin + STG-CMA path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(stg-cma_amd/) never does and has no CPU fallback.

This is a *functional* restatement -- plain functions over a flat {state_dict key: tensor} dict -- of
/root/reference/AVE/model/Swin_AVE.py (citations `Swin_AVE.py:LINE` below refer to that file).  It deliberately does NOT
copy the reference's view/permute/roll choreography: cyclic shift + window partition/reverse and the temporal
rearranges are expressed as index maps (the formulation the HIP kernels use), so agreeing with the golden vectors
generated from the imported reference (tests/golden/make_golden.py) also pins those maps.

Pinned by: tests/golden/*.npz (outputs of the reference itself run in the build container), tests/test_oracle_cpu.py.
The reference ships no tests of its own, so parity is pinned by those fixtures only.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- index maps
def window_token_map(H, W, ws, shift):
    """[nW, ws*ws] natural token index (h*W + w) of every token of every (shifted) window.

    Restates torch.roll(x, (-shift, -shift)) followed by window_partition (Swin_AVE.py:727-740, :130-142): window
    (wi, wj) token (ti, tj) sits at shifted coordinate (wi*ws+ti, wj*ws+tj) = natural ((hs+shift)%H, (ws_+shift)%W).
    window_reverse + roll(+shift) (:765-776) is the inverse scatter through the same map.
    """
    wi = torch.arange(H // ws)[:, None, None, None]
    wj = torch.arange(W // ws)[None, :, None, None]
    ti = torch.arange(ws)[None, None, :, None]
    tj = torch.arange(ws)[None, None, None, :]
    h = (wi * ws + ti + shift) % H
    w = (wj * ws + tj + shift) % W
    return (h * W + w).reshape(-1, ws * ws)


def relative_position_index(ws):
    """Swin_AVE.py:193-202."""
    ch, cw = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ch, cw = ch.reshape(-1), cw.reshape(-1)
    dh = ch[:, None] - ch[None, :] + ws - 1
    dw = cw[:, None] - cw[None, :] + ws - 1
    return dh * (2 * ws - 1) + dw


def temporal_relative_index(T):
    """Swin_AVE.py:217-221."""
    t = torch.arange(T)
    return (t[:, None] - t[None, :] + T - 1).reshape(-1)


def shift_attn_mask(H, W, ws, shift):
    """[nW, ws*ws, ws*ws] with 0 / -100 (Swin_AVE.py:368-389): tokens of different cyclic regions must not attend."""
    if shift == 0:
        return None
    region = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            region[hs, wsl] = cnt
            cnt += 1
    # region ids live in SHIFTED coordinates: plain (unshifted) partition of the id image
    ids = region.reshape(-1)[window_token_map(H, W, ws, 0)]
    diff = ids[:, None, :] - ids[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


# ----------------------------------------------------------------------------------------------- building blocks
def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _ln(P, name, x):
    w = P[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, P[name + ".bias"], 1e-5)


def mha_core(q, k, v, scale, bias=None, mask=None):
    """q,k,v: [B_, heads, n, hd]; (q*scale) @ k^T + bias (+ mask) -> softmax -> @ v (Swin_AVE.py:241-272)."""
    s = (q * scale) @ k.transpose(-2, -1)
    if bias is not None:
        s = s + bias
    if mask is not None:
        s = s + mask
    return torch.softmax(s, dim=-1) @ v


def window_attention(P, pre, x, heads, mask=None):
    """WindowAttention.forward, spatial branch (Swin_AVE.py:231-243,256-276).  x: [B_, n, C] windows, mask [nW, n, n]."""
    B_, n, C = x.shape
    hd = C // heads
    qkv = _lin(P, pre + ".qkv", x).reshape(B_, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    table = P[pre + ".relative_position_bias_table"]
    bias = table[P[pre + ".relative_position_index"].reshape(-1)].reshape(n, n, heads).permute(2, 0, 1)
    m = None
    if mask is not None:
        nW = mask.shape[0]
        m = mask.repeat(B_ // nW, 1, 1)[:, None]          # window index = b_ % nW  (:265-267)
    o = mha_core(qkv[0], qkv[1], qkv[2], hd ** -0.5, bias[None], m)
    return _lin(P, pre + ".proj", o.transpose(1, 2).reshape(B_, n, C))


def temporal_attention(P, pre, x, heads, audio):
    """WindowAttention.forward, temporal branch (Swin_AVE.py:244-255): x: [(b n), T, C]; same qkv/proj weights,
    bias from temporal_position_bias_table (video) or _audio."""
    B_, T, C = x.shape
    hd = C // heads
    qkv = _lin(P, pre + ".qkv", x).reshape(B_, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    table = P[pre + (".temporal_position_bias_table_audio" if audio else ".temporal_position_bias_table")]
    idx = P[pre + (".t_relative_coords_a" if audio else ".t_relative_coords")]
    bias = table[idx].reshape(T, T, heads).permute(2, 0, 1)
    o = mha_core(qkv[0], qkv[1], qkv[2], hd ** -0.5, bias[None])
    return _lin(P, pre + ".proj", o.transpose(1, 2).reshape(B_, T, C))


def adapter(P, pre, x):
    """Adapter / T_Adapter: D_fc2(GELU(D_fc1(x))) without skip (Swin_AVE.py:18-24, :52-58)."""
    return _lin(P, pre + ".D_fc2", F.gelu(_lin(P, pre + ".D_fc1", x)))


def cross_modal(hv, ha, gate_v, gate_a):
    """Gated bidirectional cross-modal attention on adapter hidden states [bt, n, d_h] (Swin_AVE.py:750-760, :799-808):
    single head, no 1/sqrt(d) scale, no mask."""
    s = hv @ ha.transpose(1, 2)
    a2v = torch.softmax(s, dim=-1) @ ha
    v2a = torch.softmax(s.transpose(1, 2), dim=-1) @ hv
    return hv + gate_v * a2v, ha + gate_a * v2a


def mlp(P, pre, x):
    """Mlp (Swin_AVE.py:121-127), dropout p=0."""
    return _lin(P, pre + ".fc2", F.gelu(_lin(P, pre + ".fc1", x)))


# ----------------------------------------------------------------------------------------------- Swin block
def _temporal_branch(P, pre, x, T, heads, audio, adapter_name, dp_scale):
    """Swin_AVE.py:705-716: '(b t) n c -> (b n) t c', T-attention on norm1(x), T_Adapter, x + drop_path(.), back."""
    BT, N, C = x.shape
    B = BT // T
    xt = x.reshape(B, T, N, C).permute(0, 2, 1, 3).reshape(B * N, T, C)
    res = adapter(P, pre + "." + adapter_name, temporal_attention(P, pre + ".attn", _ln(P, pre + ".norm1", xt), heads, audio))
    if dp_scale is not None:                      # DropPath mask per (b, n) row of the '(b n) t c' layout
        res = res * dp_scale.reshape(B * N, 1, 1)
    xt = xt + res
    return xt.reshape(B, N, T, C).permute(0, 2, 1, 3).reshape(BT, N, C)


def _spatial_attention(P, pre, x, H, W, heads, ws, shift, wmap, mask):
    """norm1 -> shifted-window W-MSA, returned in WINDOW layout [BT*nW, ws*ws, C] (Swin_AVE.py:718-745)."""
    BT, N, C = x.shape
    xn = _ln(P, pre + ".norm1", x)
    xw = xn[:, wmap.reshape(-1)].reshape(BT * wmap.shape[0], ws * ws, C)
    return window_attention(P, pre + ".attn", xw, heads, mask)


def _unwindow(xw, BT, N, wmap):
    """window_reverse + reverse roll == scatter through the window map (Swin_AVE.py:765-779)."""
    C = xw.shape[-1]
    out = torch.empty(BT, N, C, dtype=xw.dtype)
    out[:, wmap.reshape(-1)] = xw.reshape(BT, -1, C)
    return out


def block_geometry(H, W, window_size, shift_size):
    """Swin_AVE.py:330-333: no partition / shift when the resolution is not larger than the window."""
    ws, shift = window_size, shift_size
    if min(H, W) <= ws:
        ws, shift = min(H, W), 0
    return ws, shift


def swin_block(P, pre, x, *, H, W, T, heads, window_size=7, shift_size=0, t_attn=False, mode="fusion_adapt", dp_scale=None):
    """SwinTransformerBlock.forward for every mode (Swin_AVE.py:393-813).  x is a tensor (video_adapt / audio_adapt)
    or a (v, a) tuple.  dp_scale: optional dict of DropPath scale tensors (train-mode restatement); None = eval."""
    ws, shift = block_geometry(H, W, window_size, shift_size)
    wmap = window_token_map(H, W, ws, shift)
    mask = shift_attn_mask(H, W, ws, shift)
    dps = dp_scale or {}

    def single(x, audio):
        """video_adapt (:394-440) / audio_adapt (:442-488): adapter PARALLEL to the MLP, scaled 0.5."""
        sfx = "_Audio" if audio else ""
        BT, N, C = x.shape
        if t_attn:
            x = _temporal_branch(P, pre, x, T, heads, audio, "T_Adapter" + sfx, dps.get("t_a" if audio else "t_v"))
        aw = _spatial_attention(P, pre, x, H, W, heads, ws, shift, wmap, mask)
        aw = aw + adapter(P, pre + ".S_Adapter2" + sfx, aw)                     # SAdapter2 has the skip (:36-41)
        x = x + _unwindow(aw, BT, N, wmap)
        xn = _ln(P, pre + ".norm2", x)
        par = 0.5 * adapter(P, pre + ".S_Adapter" + sfx, xn)
        if dps.get("ffn") is not None:
            par = par * dps["ffn"].reshape(BT, 1, 1)
        return x + mlp(P, pre + ".mlp", xn) + par

    def stream_no_fusion(x, audio):
        """multimodal_adapt_no_fusion (:490-590): adapter SERIAL after the MLP, no drop_path on it."""
        sfx = "_Audio" if audio else ""
        BT, N, C = x.shape
        if t_attn:
            x = _temporal_branch(P, pre, x, T, heads, audio, "T_Adapter" + sfx, dps.get("t_a" if audio else "t_v"))
        aw = _spatial_attention(P, pre, x, H, W, heads, ws, shift, wmap, mask)
        aw = aw + adapter(P, pre + ".S_Adapter2" + sfx, aw)
        x = x + _unwindow(aw, BT, N, wmap)
        xn = mlp(P, pre + ".mlp", _ln(P, pre + ".norm2", x))
        return x + xn + adapter(P, pre + ".S_Adapter" + sfx, xn)

    if mode == "video_adapt":
        return single(x, False)
    if mode == "audio_adapt":
        return single(x, True)
    v, a = x
    if mode == "multimodal_adapt_no_fusion":
        return stream_no_fusion(v, False), stream_no_fusion(a, True)
    assert mode == "fusion_adapt", mode

    # ---- fusion_adapt (Swin_AVE.py:693-813)
    BT, N, C = v.shape
    assert N == H * W, "input feature has wrong size"
    if t_attn:
        v = _temporal_branch(P, pre, v, T, heads, False, "T_Adapter", dps.get("t_v"))
        a = _temporal_branch(P, pre, a, T, heads, True, "T_Adapter_Audio", dps.get("t_a"))
    av = _spatial_attention(P, pre, v, H, W, heads, ws, shift, wmap, mask)
    aa = _spatial_attention(P, pre, a, H, W, heads, ws, shift, wmap, mask)
    # window-level cross-modal adapter on the attention output (:747-763)
    hv = F.gelu(_lin(P, pre + ".S_Adapter2.D_fc1", av))
    ha = F.gelu(_lin(P, pre + ".S_Adapter2_Audio.D_fc1", aa))
    hv, ha = cross_modal(hv, ha, P[pre + ".gate_v"], P[pre + ".gate_a"])
    av = av + _lin(P, pre + ".S_Adapter2.D_fc2", hv)
    aa = aa + _lin(P, pre + ".S_Adapter2_Audio.D_fc2", ha)
    v = v + _unwindow(av, BT, N, wmap)
    a = a + _unwindow(aa, BT, N, wmap)
    # FFN, then the frame-global cross-modal adapter on the MLP OUTPUT (:790-811)
    vn = mlp(P, pre + ".mlp", _ln(P, pre + ".norm2", v))
    an = mlp(P, pre + ".mlp", _ln(P, pre + ".norm2", a))
    hv = F.gelu(_lin(P, pre + ".S_Adapter.D_fc1", vn))
    ha = F.gelu(_lin(P, pre + ".S_Adapter_Audio.D_fc1", an))
    hv, ha = cross_modal(hv, ha, P[pre + ".gate_v"], P[pre + ".gate_a"])
    v = v + vn + _lin(P, pre + ".S_Adapter.D_fc2", hv)
    a = a + an + _lin(P, pre + ".S_Adapter_Audio.D_fc2", ha)
    return v, a


# ----------------------------------------------------------------------------------------------- stem / merge / model
def patch_embed(P, pre, x, patch=(1, 4, 4)):
    """PatchEmbed3D (Swin_AVE.py:1104-1124): Conv3d(k=s=patch) + LayerNorm, 'b c d h w -> (b d) (h w) c'.
    The conv with kernel == stride is a patch-row GEMM; written as unfold + matmul here."""
    B, Cin, D, Hh, Ww = x.shape
    pd, ph, pw = patch
    assert pd == 1 and Hh % ph == 0 and Ww % pw == 0
    w = P[pre + ".proj.weight"]                                        # [E, Cin, 1, ph, pw]
    cols = x.reshape(B, Cin, D, Hh // ph, ph, Ww // pw, pw).permute(0, 2, 3, 5, 1, 4, 6).reshape(B * D, (Hh // ph) * (Ww // pw), -1)
    y = cols @ w.reshape(w.shape[0], -1).t() + P[pre + ".proj.bias"]
    if pre + ".norm.weight" in P:
        y = _ln(P, pre + ".norm", y)
    return y


def patch_merging(P, pre, x, H, W):
    """PatchMerging (Swin_AVE.py:958-981)."""
    BT, N, C = x.shape
    assert N == H * W and H % 2 == 0 and W % 2 == 0
    xv = x.reshape(BT, H // 2, 2, W // 2, 2, C)
    # cat order x0(even,even) x1(odd,even) x2(even,odd) x3(odd,odd)  (:967-971)
    cat = torch.cat([xv[:, :, 0, :, 0], xv[:, :, 1, :, 0], xv[:, :, 0, :, 1], xv[:, :, 1, :, 1]], dim=-1)
    cat = cat.reshape(BT, (H // 2) * (W // 2), 4 * C)
    return F.linear(_ln(P, pre + ".norm", cat), P[pre + ".reduction.weight"])


def swin_forward(P, a, v, cfg, mode):
    """SwinTransformer2D_Adapter_New.forward (Swin_AVE.py:1479-1599), eval semantics (DropPath / Dropout identity).

    cfg: dict(embed_dim, depths, num_heads, window_size, num_frames, img_size, label_dim).
    a: [B, T, H, W] spectrogram segments, v: [B, 3, T, H, W] frames.
    """
    depths, heads = cfg["depths"], cfg["num_heads"]
    ws = cfg.get("window_size", 7)
    T = cfg["num_frames"]
    res = cfg.get("img_size", 224) // 4
    block_mode = {"videoonly": "video_adapt", "audioonly": "audio_adapt", "multimodal": "multimodal_adapt_no_fusion",
                  "fusion": "fusion_adapt"}[mode]
    xv = patch_embed(P, "patch_embed", v) if mode != "audioonly" else None
    xa = patch_embed(P, "patch_embed_audio", a.unsqueeze(1)) if mode != "videoonly" else None
    if "temporal_embedding" in P:                     # t_relative=False (:1207-1212): '(b t) n c -> (b n) t c' + embedding (:1483-1487, :1569-1576)
        def add_t(x, e):
            BT, N, C = x.shape
            return (x.view(BT // T, T, N, C) + e.view(1, T, 1, C)).view(BT, N, C)
        xv = add_t(xv, P["temporal_embedding"]) if xv is not None else None
        xa = add_t(xa, P["temporal_embedding_audio"]) if xa is not None else None
    x = xv if mode == "videoonly" else xa if mode == "audioonly" else (xv, xa)
    for s, depth in enumerate(depths):
        H = W = res // (2 ** s)
        for i in range(depth):
            x = swin_block(P, f"layers.{s}.blocks.{i}", x, H=H, W=W, T=T, heads=heads[s], window_size=ws,
                           shift_size=0 if i % 2 == 0 else ws // 2, t_attn=(i % 2 == 0), mode=block_mode)
        if s < len(depths) - 1:                                          # BasicLayer downsample (:1056-1063)
            if isinstance(x, tuple):
                x = tuple(patch_merging(P, f"layers.{s}.downsample", t, H, W) for t in x)
            else:
                x = patch_merging(P, f"layers.{s}.downsample", x, H, W)
    if isinstance(x, tuple):
        xv, xa = x
        pooled = torch.cat((_ln(P, "norm", xa).mean(1), _ln(P, "norm", xv).mean(1)), dim=-1)   # cat((a, v)) (:1596)
        hid = _lin(P, "mlp_head.0", pooled)                                                   # Dropout(.5) identity in eval
        return _lin(P, "mlp_head.2", hid)
    pooled = _ln(P, "norm", x).mean(1)
    return _lin(P, "mlp_head.1", _ln(P, "mlp_head.0", pooled))


def swin_plain_block(P, pre, x, *, H, W, heads, window_size=7, shift_size=0, dp_scale=None):
    """The AVQA block's third ("negative" video) stream: the frozen Swin block, no temporal attention, no adapters,
    drop_path on both residuals (AVQA/model/Swin_AVQAModel_V1.py:780-782, :793, :808-810, :821-823, :836-838, :852-860).
    dp_scale: optional (attn, ffn) per-sample DropPath scale tensors [BT]; None = eval."""
    ws, shift = block_geometry(H, W, window_size, shift_size)
    wmap = window_token_map(H, W, ws, shift)
    mask = shift_attn_mask(H, W, ws, shift)
    BT, N, C = x.shape
    r = _unwindow(_spatial_attention(P, pre, x, H, W, heads, ws, shift, wmap, mask), BT, N, wmap)
    if dp_scale is not None:
        r = r * dp_scale[0].reshape(BT, 1, 1)
    x = x + r
    f = mlp(P, pre + ".mlp", _ln(P, pre + ".norm2", x))
    if dp_scale is not None:
        f = f * dp_scale[1].reshape(BT, 1, 1)
    return x + f


def swin_backbone(P, a, v, cfg, v_nega=None):
    """The `fusion` backbone shared by the AVS and AVQA models, eval semantics: patch embeds -> stages -> final norm.
      AVS  (AVS/model/Swin_AVSModel.py:1790-1822):  BasicLayer returns (x, x_before_downsample) (:1190-1201); the video stream
           before each downsample is tapped, the last tap goes through self.norm (:1813-1821).
      AVQA (AVQA/model/Swin_AVQAModel_V1.py:1742-1766):  a third stream v_nega rides along through every block (:752-872) and
           downsample (:1150-1154); f_v, f_a, visual_nega = norm(v), norm(a), norm(v_nega).
    a: [B, T, H, W]; v, v_nega: [B, T, 3, H, W] (both models rearrange 'b t c h w -> b c t h w', :1793 / :1742,1748).
    Returns dict(f_v, f_a [, f_nega], taps=[v before downsample of stages 0..n-2, norm(v) of the last stage])."""
    depths, heads = cfg["depths"], cfg["num_heads"]
    ws = cfg.get("window_size", 7)
    T = cfg["num_frames"]
    res = cfg.get("img_size", 224) // 4
    xv = patch_embed(P, "patch_embed", v.permute(0, 2, 1, 3, 4))
    xa = patch_embed(P, "patch_embed_audio", a.unsqueeze(1))
    xn = patch_embed(P, "patch_embed", v_nega.permute(0, 2, 1, 3, 4)) if v_nega is not None else None
    if "temporal_embedding" in P:                     # t_relative=False (Swin_AVSModel.py:1800-1806, Swin_AVQAModel_V1.py:1752-1758): v and a only
        def add_t(x, e):
            BT, N, C = x.shape
            return (x.view(BT // T, T, N, C) + e.view(1, T, 1, C)).view(BT, N, C)
        xv, xa = add_t(xv, P["temporal_embedding"]), add_t(xa, P["temporal_embedding_audio"])
    taps = []
    for s, depth in enumerate(depths):
        H = W = res // (2 ** s)
        for i in range(depth):
            kw = dict(H=H, W=W, heads=heads[s], window_size=ws, shift_size=0 if i % 2 == 0 else ws // 2)
            xv, xa = swin_block(P, f"layers.{s}.blocks.{i}", (xv, xa), T=T, t_attn=(i % 2 == 0), mode="fusion_adapt", **kw)
            if xn is not None:
                xn = swin_plain_block(P, f"layers.{s}.blocks.{i}", xn, **kw)
        if s < len(depths) - 1:
            taps.append(xv)
            xv = patch_merging(P, f"layers.{s}.downsample", xv, H, W)
            xa = patch_merging(P, f"layers.{s}.downsample", xa, H, W)
            if xn is not None:
                xn = patch_merging(P, f"layers.{s}.downsample", xn, H, W)
    out = {"f_v": _ln(P, "norm", xv), "f_a": _ln(P, "norm", xa)}
    out["taps"] = taps + [out["f_v"]]
    if xn is not None:
        out["f_nega"] = _ln(P, "norm", xn)
    return out


def soft_target_cross_entropy(logits, target):
    """nn.CrossEntropyLoss with class-probability targets, mean reduction (traintest_adapt_ave29.py:113,159)."""
    return -(target * torch.log_softmax(logits, dim=-1)).sum(-1).mean()


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """utilities/scheduler.py:5-30 restated with vector ops."""
    warmup_iters = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    warm = torch.linspace(start_warmup_value, base_value, warmup_iters, dtype=torch.float64) if warmup_epochs > 0 else \
        torch.zeros(0, dtype=torch.float64)
    n = epochs * niter_per_ep - warmup_iters
    i = torch.arange(n, dtype=torch.float64)
    cos = final_value + 0.5 * (base_value - final_value) * (1 + torch.cos(math.pi * i / n))
    out = torch.cat((warm, cos))
    assert out.numel() == epochs * niter_per_ep
    return out
