"""ORACLE (test infrastructure, not product code): fp32 PyTorch-CPU restatement of the reference's AVQA question-answering
head -- QstEncoder (AVQA/model/Swin_AVQAModel_V1.py:37-59) and lines :1768-1903 of SwinTransformer2D_Adapter_AVQA.forward -- as
plain functions over a flat {state_dict key: tensor} dict, eval semantics (every Dropout / attention dropout off).

Only tests/ may import this; the product path (stg-cma_amd/ops_head.py) never does.  The LSTM and the single-query multi-head
attention are written out from their definitions (torch.nn.LSTM gate order i, f, g, o; nn.MultiheadAttention's packed
in-projection), not through the nn modules.  The 512-d head variant of AVQA/model/Swin_AVQAModel.py (AVQA/test.py:8) is the same code
behind three extra Linears.  Pinned by tests/golden/avqa_full_tiny.npz and avqa512_full_tiny.npz (the reference models run in the
build container, tests/golden/make_golden.py::avqa_full_case) through tests/test_oracle_cpu.py.
"""
import math

import torch
import torch.nn.functional as F

from .swin import swin_backbone


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def question_encoder(P, pre, question):
    """QstEncoder.forward (:46-59): embedding -> tanh -> 1-layer LSTM over the words -> tanh(cat(h, c)) -> fc."""
    x = torch.tanh(P[pre + ".word2vec.weight"][question])            # [B, L, E]
    B, L, _ = x.shape
    Wi, Wh = P[pre + ".lstm.weight_ih_l0"], P[pre + ".lstm.weight_hh_l0"]
    bi, bh = P[pre + ".lstm.bias_ih_l0"], P[pre + ".lstm.bias_hh_l0"]
    Hh = Wh.shape[1]
    h = x.new_zeros(B, Hh)
    c = x.new_zeros(B, Hh)
    for l in range(L):
        g = x[:, l] @ Wi.t() + bi + h @ Wh.t() + bh
        i, f, gg, o = g[:, :Hh], g[:, Hh:2 * Hh], g[:, 2 * Hh:3 * Hh], g[:, 3 * Hh:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
    feat = torch.tanh(torch.cat((h, c), dim=1))                       # num_layers = 1: cat(hidden, cell) (:54-57)
    return _lin(P, pre + ".fc", feat)


def single_query_mha(P, pre, xq, kv, heads=4):
    """nn.MultiheadAttention(E, 4)(xq[None], kv, kv)[0].squeeze(0), xq [B, E], kv [T, B, E] (:1866-1867, 1875-1876)."""
    E = xq.shape[1]
    W, b = P[pre + ".in_proj_weight"], P[pre + ".in_proj_bias"]
    q = F.linear(xq, W[:E], b[:E])
    k = F.linear(kv, W[E:2 * E], b[E:2 * E])
    v = F.linear(kv, W[2 * E:], b[2 * E:])
    T, B, _ = kv.shape
    hd = E // heads
    qh = q.view(B, heads, 1, hd)
    kh = k.view(T, B, heads, hd).permute(1, 2, 0, 3)
    vh = v.view(T, B, heads, hd).permute(1, 2, 0, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    o = (p @ vh).reshape(B, E)
    return F.linear(o, P[pre + ".out_proj.weight"], P[pre + ".out_proj.bias"])


def _ground_and_match(P, fv, audio_feat):
    """Audio-visual grounding of one clip stream + the match head (:1788-1824 / :1828-1853). fv [(B T), 49, C]."""
    vmean = fv.mean(dim=1)                                            # AdaptiveAvgPool2d((1, 1)) over the 7 x 7 map
    vh = F.normalize(fv, dim=2)
    ah = F.normalize(audio_feat, dim=1)
    p = torch.softmax(torch.einsum("fnc,fc->fn", vh, ah), dim=-1)
    grd = torch.einsum("fn,fnc->fc", p, vh)
    vgrd = _lin(P, "avqatask_fc_gl", torch.tanh(torch.cat((vmean, grd), dim=-1)))
    feat = torch.cat((audio_feat, vgrd), dim=-1)
    feat = F.relu(_lin(P, "avqatask_fc1", feat))
    feat = F.relu(_lin(P, "avqatask_fc2", feat))
    feat = F.relu(_lin(P, "avqatask_fc3", feat))
    return vgrd, _lin(P, "avqatask_fc4", feat)


def avqa_head(P, f_v, f_a, f_nega, question, B, T):
    """(:1768-1903) f_*: [(B T), 49, 1536] -> (out_qa [B, 42], out_match_posi [(B T), 2], out_match_nega [(B T), 2])."""
    if "avqatask_yb_fc_v.weight" in P:
        # the 512-d variant, AVQA/model/Swin_AVQAModel.py:1772-1783 (projections of the three streams) and :1798 (fc_a1 + ReLU)
        f_v, f_nega = _lin(P, "avqatask_yb_fc_v", f_v), _lin(P, "avqatask_yb_fc_v", f_nega)
        audio = F.relu(_lin(P, "avqatask_fc_a1", _lin(P, "avqatask_yb_fc_a", f_a).mean(dim=1)))
    else:
        audio = F.relu(f_a.mean(dim=1))                                # Swin_AVQAModel_V1.py:1791, :1800
    C = f_v.shape[-1]
    qst = question_encoder(P, "avqatask_question_encoder", question)
    audio_feat = _lin(P, "avqatask_fc_a2", audio)
    vgrd_posi, out_match_posi = _ground_and_match(P, f_v, audio_feat)
    _, out_match_nega = _ground_and_match(P, f_nega, audio_feat)
    vis = vgrd_posi.view(B, T, C)
    att_v = single_query_mha(P, "avqatask_attn_v", qst, vis.permute(1, 0, 2))
    src = _lin(P, "avqatask_linear12", F.relu(_lin(P, "avqatask_linear11", att_v)))
    att_v = F.layer_norm(att_v + src, (C,), P["avqatask_norm1.weight"], P["avqatask_norm1.bias"])
    aud = audio_feat.view(B, T, C)
    att_a = single_query_mha(P, "avqatask_attn_a", qst, aud.permute(1, 0, 2))
    src = _lin(P, "avqatask_linear22", F.relu(_lin(P, "avqatask_linear21", att_a)))
    att_a = F.layer_norm(att_a + src, (C,), P["avqatask_norm2.weight"], P["avqatask_norm2.bias"])
    feat = torch.cat((att_a + aud.mean(dim=1), att_v + vis.mean(dim=1)), dim=-1)
    feat = _lin(P, "avqatask_fc_fusion", torch.tanh(feat))
    out_qa = _lin(P, "avqatask_fc_ans", torch.tanh(feat * qst))
    return out_qa, out_match_posi, out_match_nega


def avqa_forward(P, a, v, v_nega, question, cfg):
    """SwinTransformer2D_Adapter_AVQA.forward[fusion] (:1740-1903): backbone (oracle.swin.swin_backbone) + head."""
    out = swin_backbone(P, a, v, cfg, v_nega=v_nega)
    return avqa_head(P, out["f_v"], out["f_a"], out["f_nega"], question, v.shape[0], v.shape[1])
