"""ORACLE (test infrastructure, not product code): fp32 PyTorch-CPU restatement of the reference's CLIP-ViT + STG-CMA path,
/root/reference/AVE/model/CLIP_AVE.py (citations `CLIP_AVE.py:LINE`).  Functional, over a flat {state_dict key: tensor} dict,
in the '(b t) n d' token order the HIP kernels use (the reference works sequence-first, 'n (b t) d'; the rearranges are
restated as reshapes / index arithmetic).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

Pinned by tests/golden/vit_*.npz (outputs of the reference run in the build container) through tests/test_oracle_cpu.py.
"""
import torch
import torch.nn.functional as F


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _ln(P, name, x):
    """LayerNorm subclass that computes in fp32 (CLIP_AVE.py:33-39)."""
    w = P[name + ".weight"]
    return F.layer_norm(x.float(), (w.shape[0],), w, P[name + ".bias"], 1e-5)


def mha(P, pre, x, heads):
    """nn.MultiheadAttention self-attention core, batch-first here: x [S, L, D] -> [S, L, D] (CLIP_AVE.py:106-108)."""
    S, L, D = x.shape
    hd = D // heads
    qkv = F.linear(x, P[pre + ".in_proj_weight"], P[pre + ".in_proj_bias"]).reshape(S, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    s = (qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1)
    o = (torch.softmax(s, -1) @ qkv[2]).transpose(1, 2).reshape(S, L, D)
    return _lin(P, pre + ".out_proj", o)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)            # CLIP_AVE.py:41-43


def mlp(P, pre, x):
    return _lin(P, pre + ".c_proj", quick_gelu(_lin(P, pre + ".c_fc", x)))


def adapter_hidden(P, pre, x):
    return F.gelu(_lin(P, pre + ".D_fc1", x))


def cross_modal(hv, ha, gate_v, gate_a):
    """[bt, nv, d] x [bt, na, d] gated bidirectional attention (CLIP_AVE.py:386-398, :415-427)."""
    s = hv @ ha.transpose(1, 2)
    a2v = torch.softmax(s, -1) @ ha
    v2a = torch.softmax(s.transpose(1, 2), -1) @ hv
    return hv + gate_v * a2v, ha + gate_a * v2a


def _temporal(P, pre, x, T, heads, adapter_name, dp_scale=None):
    """'n (b t) d -> t (b n) d' temporal attention + T_Adapter, x: [BT, n, D] (CLIP_AVE.py:369-377)."""
    BT, n, D = x.shape
    B = BT // T
    xt = x.reshape(B, T, n, D).permute(0, 2, 1, 3).reshape(B * n, T, D)
    r = mha(P, pre + ".attn", _ln(P, pre + ".ln_1", xt), heads)
    r = _lin(P, pre + "." + adapter_name + ".D_fc2", adapter_hidden(P, pre + "." + adapter_name, r))
    r = r.reshape(B, n, T, D).permute(0, 2, 1, 3).reshape(BT, n, D)
    if dp_scale is not None:                       # DropPath mask along dim 0 of the 'n (b t) d' tensor = per token index
        r = r * dp_scale.reshape(1, n, 1)
    return x + r


def vit_block(P, pre, x, *, T, heads, mode="fusion_adapt"):
    """ResidualAttentionBlock.forward (CLIP_AVE.py:110-429), tokens in '(b t) n d' order.  x: tensor or (v, a)."""
    def single(x, sfx):
        x = _temporal(P, pre, x, T, heads, "T_Adapter" + sfx)
        y = mha(P, pre + ".attn", _ln(P, pre + ".ln_1", x), heads)
        x = x + y + _lin(P, pre + ".S_Adapter" + sfx + ".D_fc2", adapter_hidden(P, pre + ".S_Adapter" + sfx, y))   # skip_connect=True
        xn = mlp(P, pre + ".mlp", _ln(P, pre + ".ln_2", x))
        return x + xn + _lin(P, pre + ".MLP_Adapter" + sfx + ".D_fc2", adapter_hidden(P, pre + ".MLP_Adapter" + sfx, xn))

    if mode == "video_adapt":
        return single(x, "")
    if mode == "audio_adapt":
        return single(x, "_Audio")
    v, a = x
    if mode == "multimodal_adapt_no_fusion":
        return single(v, ""), single(a, "_Audio")
    assert mode == "fusion_adapt"
    gv, ga = P[pre + ".gate_v"], P[pre + ".gate_a"]
    v = _temporal(P, pre, v, T, heads, "T_Adapter")
    a = _temporal(P, pre, a, T, heads, "T_Adapter_Audio")
    vs = mha(P, pre + ".attn", _ln(P, pre + ".ln_1", v), heads)
    as_ = mha(P, pre + ".attn", _ln(P, pre + ".ln_1", a), heads)
    hv, ha = cross_modal(adapter_hidden(P, pre + ".S_Adapter", vs), adapter_hidden(P, pre + ".S_Adapter_Audio", as_), gv, ga)
    v = v + vs + _lin(P, pre + ".S_Adapter.D_fc2", hv)
    a = a + as_ + _lin(P, pre + ".S_Adapter_Audio.D_fc2", ha)
    vn = mlp(P, pre + ".mlp", _ln(P, pre + ".ln_2", v))
    an = mlp(P, pre + ".mlp", _ln(P, pre + ".ln_2", a))
    hv, ha = cross_modal(adapter_hidden(P, pre + ".MLP_Adapter", vn), adapter_hidden(P, pre + ".MLP_Adapter_Audio", an), gv, ga)
    v = v + vn + _lin(P, pre + ".MLP_Adapter.D_fc2", hv)
    a = a + an + _lin(P, pre + ".MLP_Adapter_Audio.D_fc2", ha)
    return v, a


def embed(P, x4, conv, pos, temb, T):
    """conv (k = s = 16, no bias, remainder rows/cols dropped) -> cls + positional + temporal embedding -> ln_pre
    (CLIP_AVE.py:1091-1105 / :1109-1123).  x4: [BT, Cin, H, W] -> [BT, n, D]."""
    w = P[conv]
    D, Cin, p, _ = w.shape
    BT, _, H, W = x4.shape
    Hp, Wp = H // p, W // p
    cols = x4[:, :, :Hp * p, :Wp * p].reshape(BT, Cin, Hp, p, Wp, p).permute(0, 2, 4, 1, 3, 5).reshape(BT, Hp * Wp, -1)
    tok = cols @ w.reshape(D, -1).t()
    tok = torch.cat([P["class_embedding"].expand(BT, 1, D), tok], 1) + P[pos]
    t_idx = torch.arange(BT) % T
    tok = tok + P[temb][0][t_idx][:, None, :]
    return _ln(P, "ln_pre", tok)


def vit_forward(P, a, v, cfg, mode="fusion"):
    """MM_CLIP_AVE.forward (CLIP_AVE.py:979-1140), eval semantics.  a: [B, T, Ha, Wa], v: [B, 3, T, H, W] -> [(B T), label]."""
    T, heads, layers = cfg["T"], cfg["heads"], cfg["layers"]
    B = v.shape[0] if v is not None else a.shape[0]
    xv = xa = None
    if mode != "audioonly":
        xv = embed(P, v.permute(0, 2, 1, 3, 4).reshape(B * T, 3, v.shape[3], v.shape[4]), "conv1.weight",
                   "positional_embedding", "temporal_embedding", T)
    if mode != "videoonly":
        xa = embed(P, a.reshape(B * T, 1, a.shape[2], a.shape[3]), "conv1_audio.weight", "positional_embedding_audio",
                   "temporal_embedding_audio", T)
    bmode = {"videoonly": "video_adapt", "audioonly": "audio_adapt", "multimodal": "multimodal_adapt_no_fusion",
             "fusion": "fusion_adapt"}[mode]
    x = xv if mode == "videoonly" else xa if mode == "audioonly" else (xv, xa)
    for i in range(layers):
        x = vit_block(P, f"transformer.resblocks.{i}", x, T=T, heads=heads, mode=bmode)
    if isinstance(x, tuple):
        cv, ca = _ln(P, "ln_post", x[0])[:, 0], _ln(P, "ln_post", x[1])[:, 0]
        return _lin(P, "mlp_head.2", _lin(P, "mlp_head.0", torch.cat((ca, cv), -1)))           # cat((a, v)); Dropout identity
    c = _ln(P, "ln_post", x)[:, 0]
    return _lin(P, "mlp_head.1", F.layer_norm(c, (c.shape[-1],), P["mlp_head.0.weight"], P["mlp_head.0.bias"], 1e-5))
