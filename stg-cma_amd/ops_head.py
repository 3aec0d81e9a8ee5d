"""The AVQA question-answering head (AVQA/model/Swin_AVQAModel_V1.py:37-59 QstEncoder, :1768-1903) on libstgcma_hip.so.

Unlike the backbone (one hand-scheduled autograd node), the head is a few hundred rows of 1536 channels: ~20 tiny GEMMs and the
element-wise / reduction steps between them, < 1 % of the step.  It is therefore written the way the reference writes it -- op by
op -- with every op a torch.autograd.Function whose forward AND backward are libstgcma_hip.so launches (GEMMs on stg_gemm_nt /
stg_wgrad_tn, the rest on head.hip); autograd only chains them.  The only ATen kernels involved are data movement
(torch.cat of feature halves, contiguous() of slices) and the uniform draw of the dropout masks.

Activations are bf16, accumulations / LSTM cell state / logits fp32, parameters fp32 masters with bf16 shadows (ops.shadow).
"""
import torch

from . import kernels as K
from .kernels import BF16, F32, RELU, TANH
from .ops import f32c, shadow


def _bf(dy):
    """Incoming gradient -> contiguous bf16 2-D tensor."""
    dy = dy.contiguous()
    if dy.dtype == BF16:
        return dy
    d2 = dy.float().reshape(dy.shape[0], -1)
    return K.cast_bf16(d2)[:, :d2.shape[1]].contiguous() if d2.shape[1] % 8 else K.cast_bf16(d2)


class CastFn(torch.autograd.Function):
    """fp32 [R, C] (C % 8 == 0) -> bf16; the gradient comes back as fp32."""

    @staticmethod
    def forward(ctx, x):
        return K.cast_bf16(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return K.cast_f32(_bf(dy))


class LinearFn(torch.autograd.Function):
    """y = x W^T + b (+ res): stg_gemm_nt forward, stg_gemm_nt dgrad on the transposed shadow, stg_wgrad_tn for dW / db.
    x bf16 [M, K]; W fp32 [N, K] (or a row slice view of a packed weight), b fp32 [N] or None; res fp32 [M, N] or None
    (then the output is fp32: the LSTM gate pre-activations)."""

    @staticmethod
    def forward(ctx, x, W, b, res, out_f32):
        x = x.contiguous()
        M, Kd = x.shape
        N = W.shape[0]
        y = K.gemm_nt(x, shadow(W), f32c(b) if b is not None else None, res1=res,
                      out_dtype=F32 if (out_f32 or res is not None) else BF16)
        ctx.save_for_backward(x)
        ctx.W, ctx.has_b, ctx.has_res = W, b is not None, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        W = ctx.W
        N, Kd = W.shape
        need_x, need_w, need_b, need_r = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        dres = dy if (ctx.has_res and need_r) else None
        d = dy.contiguous()
        if d.dtype != BF16 or N % 8:                  # pad the N columns to a multiple of 8: K of the dgrad GEMM / wgrad operand
            d = K.cast_bf16(d.float().contiguous())
        dx = K.gemm_nt(d, shadow(W, True)) if need_x else None
        dW = db = None
        if need_w:
            dW = torch.zeros((N, Kd), dtype=F32, device=x.device)
            db = torch.zeros((N,), dtype=F32, device=x.device) if (ctx.has_b and need_b) else None
            K.wgrad_tn(d, x, dW, db, n1=N)
        return dx, dW, db, dres, None


def linear(x, W, b=None, res=None, out_f32=False):
    return LinearFn.apply(x, W, b, res, out_f32)


class UnaryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.op = op
        return K.unary_fwd(op, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.unary_bwd(ctx.op, x, _bf(dy).view(x.shape)), None


def relu(x):
    return UnaryFn.apply(x, RELU)


def tanh(x):
    return UnaryFn.apply(x, TANH)


class MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return K.mul(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        d = _bf(dy).view(a.shape)
        return K.mul(d, b.contiguous()), K.mul(d, a.contiguous())


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class MulMaskFn(torch.autograd.Function):
    """nn.Dropout: x * mask, mask = Bernoulli(keep) / keep (fp32)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.mask = mask
        return K.mul_mask(x.contiguous(), mask)

    @staticmethod
    def backward(ctx, dy):
        return K.mul_mask(_bf(dy), ctx.mask), None


def dropout(x, p, training):
    if not training or p == 0.:
        return x
    keep = 1.0 - p
    mask = (torch.rand(x.shape, device=x.device) < keep).to(F32).mul_(1.0 / keep)
    return MulMaskFn.apply(x, mask)


class MeanFn(torch.autograd.Function):
    """x bf16 [G*n, C] -> mean over the n rows of each group: [G, C] (tensor.mean(dim=-2) of a [G, n, C] view)."""

    @staticmethod
    def forward(ctx, x, G, n):
        ctx.G, ctx.n = G, n
        return K.meanpool_fwd(x.contiguous(), G, n)

    @staticmethod
    def backward(ctx, dy):
        return K.meanpool_bwd(_bf(dy), ctx.G, ctx.n), None, None


class EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table):
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return K.embed_fwd(f32c(table), idx.contiguous())

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dt = torch.zeros(ctx.shape, dtype=F32, device=dy.device)
        K.embed_bwd(_bf(dy), idx.contiguous(), dt)
        return None, dt


class LstmCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gates, c_prev):
        h, c = K.lstm_cell_fwd(gates.contiguous(), c_prev.contiguous())
        ctx.save_for_backward(gates, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c = ctx.saved_tensors
        dg, dcp = K.lstm_cell_bwd(gates.contiguous(), c_prev.contiguous(), c, _bf(dh) if dh is not None else None,
                                  dc.contiguous() if dc is not None else None)
        return K.cast_f32(dg), dcp


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        y, mean, rstd = K.layernorm_fwd(x.contiguous(), f32c(gamma), f32c(beta))
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma = gamma
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dg = torch.zeros_like(ctx.gamma, dtype=F32)
        db = torch.zeros_like(dg)
        dyb, xc = _bf(dy), x.contiguous()
        if xc.shape[0] >= 1024 and xc.dtype == BF16:          # many rows (TPAVI's norm_layer): block-folded column sums, not per-row atomics
            dx = K.layernorm_bwd(dyb, xc, f32c(ctx.gamma), mean, rstd)
            K.ln_param_grad(dyb, xc, mean, rstd, dg, db)
        else:
            dx = K.layernorm_bwd(dyb, xc, f32c(ctx.gamma), mean, rstd, dgamma=dg, dbeta=db)
        return dx, dg, db


class GroundingFn(torch.autograd.Function):
    """(vmean, grd) of stg_grounding_fwd.  V fp32 [F, n, C] (gradient only when it requires one), a bf16 [F, C]."""

    @staticmethod
    def forward(ctx, V, a):
        Vc, ac = V.contiguous(), a.contiguous()
        vmean, grd, saved = K.grounding_fwd(Vc, ac)
        ctx.save_for_backward(Vc, ac, *saved)
        return vmean, grd

    @staticmethod
    def backward(ctx, dvmean, dgrd):
        V, a, p, rn, ra = ctx.saved_tensors
        want_dV = ctx.needs_input_grad[0]
        if dgrd is None:
            dgrd = torch.zeros_like(a)
        dV, da = K.grounding_bwd(V, a, (p, rn, ra), _bf(dvmean) if dvmean is not None else None, _bf(dgrd), want_dV)
        return dV, da


class Mha1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, drop, H):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o, p = K.mha1_fwd(q, k, v, drop, H)
        ctx.save_for_backward(q, k, v, p)
        ctx.drop, ctx.H = drop, H
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        dq, dk, dv = K.mha1_bwd(q, k, v, ctx.drop, p, _bf(do), ctx.H)
        return dq, dk, dv, None, None


def mha_single_query(xq, kv, attn, training):
    """nn.MultiheadAttention(E, H, dropout)(xq[None], kv, kv)[0].squeeze(0) for xq [B, E], kv [T*B, E] (row t*B + b)
    (Swin_AVQAModel_V1.py:1866-1867, 1875-1876): packed in-projection rows [0:E] / [E:2E] / [2E:3E], out_proj."""
    E = xq.shape[1]
    H = attn.num_heads
    Wi, bi = attn.in_proj_weight, attn.in_proj_bias
    q = linear(xq, Wi[:E], bi[:E])
    k = linear(kv, Wi[E:2 * E], bi[E:2 * E])
    v = linear(kv, Wi[2 * E:], bi[2 * E:])
    drop = None
    if training and attn.dropout > 0.:
        keep = 1.0 - attn.dropout
        B, T = xq.shape[0], kv.shape[0] // xq.shape[0]
        drop = (torch.rand((B, H, T), device=xq.device) < keep).to(F32).mul_(1.0 / keep)
    o = Mha1Fn.apply(q, k, v, drop, H)
    return linear(o, attn.out_proj.weight, attn.out_proj.bias)


def question_encoder(enc, question):
    """QstEncoder.forward (:46-59): embedding -> tanh -> one-layer LSTM over the words -> tanh(cat(h, c)) -> fc."""
    B, L = question.shape
    lstm = enc.lstm
    if lstm.num_layers != 1 or lstm.bidirectional:
        raise NotImplementedError("question encoder: one unidirectional LSTM layer (the reference's QstEncoder(93, 1536, 1536, 1, 1536))")
    Hh = lstm.hidden_size
    idx = question.transpose(0, 1).contiguous().reshape(-1)                     # word-major: row l*B + b
    x = tanh(EmbedFn.apply(idx, enc.word2vec.weight))                           # [L*B, E]
    xg = linear(x, lstm.weight_ih_l0, lstm.bias_ih_l0, out_f32=True)            # [L*B, 4H] fp32: every step's input part at once
    h = torch.zeros((B, Hh), dtype=BF16, device=question.device)
    c = torch.zeros((B, Hh), dtype=F32, device=question.device)
    for l in range(L):
        gates = linear(h, lstm.weight_hh_l0, lstm.bias_hh_l0, res=xg[l * B:(l + 1) * B])
        h, c = LstmCellFn.apply(gates, c)
    feat = torch.cat((h, CastFn.apply(c)), dim=1)                               # num_layers = 1: [B, 2H]
    return linear(tanh(feat), enc.fc.weight, enc.fc.bias)


def avqa_head_forward(m, f_v, f_a, f_nega, question, B, T, training):
    """Lines :1768-1903 of the reference forward.  f_v, f_a, f_nega: fp32 [(B T), 49, C] (norm of each backbone stream);
    returns (out_qa [B, 42], out_match_posi [(B T), 2], out_match_nega [(B T), 2]), fp32."""
    BT, n, C = f_v.shape
    if getattr(m, "PROJECT_FEATURES", False):
        # the 512-d variant (AVQA/model/Swin_AVQAModel.py:1772-1783, 1798-1801): project the three streams down first
        def proj(f, lin, f32):
            return linear(CastFn.apply(f.reshape(BT * n, C)), lin.weight, lin.bias, out_f32=f32)
        f_v = proj(f_v, m.avqatask_yb_fc_v, True).view(BT, n, -1)
        f_nega = proj(f_nega, m.avqatask_yb_fc_v, True).view(BT, n, -1)
        audio = MeanFn.apply(proj(f_a, m.avqatask_yb_fc_a, False), BT, n)
        audio = relu(linear(audio, m.avqatask_fc_a1.weight, m.avqatask_fc_a1.bias))
        C = f_v.shape[-1]
    else:
        audio = relu(MeanFn.apply(CastFn.apply(f_a.reshape(BT * n, C)), BT, n))   # f_a.mean(dim=1), F.relu  (:1791, :1800)
    qst = question_encoder(m.avqatask_question_encoder, question)              # [B, C]
    audio_feat = linear(audio, m.avqatask_fc_a2.weight, m.avqatask_fc_a2.bias)              # [(B T), C]  (:1801)

    def ground_and_match(fv):
        vmean, grd = GroundingFn.apply(fv, audio_feat)                          # :1797-1815
        gl = tanh(torch.cat((vmean, grd), dim=-1))
        vgrd = linear(gl, m.avqatask_fc_gl.weight, m.avqatask_fc_gl.bias)       # visual_feat_grd_(posi|nega)
        feat = torch.cat((audio_feat, vgrd), dim=-1)
        feat = relu(linear(feat, m.avqatask_fc1.weight, m.avqatask_fc1.bias))
        feat = relu(linear(feat, m.avqatask_fc2.weight, m.avqatask_fc2.bias))
        feat = relu(linear(feat, m.avqatask_fc3.weight, m.avqatask_fc3.bias))
        return vgrd, linear(feat, m.avqatask_fc4.weight, m.avqatask_fc4.bias, out_f32=True)
    vgrd_posi, out_match_posi = ground_and_match(f_v)
    _, out_match_nega = ground_and_match(f_nega)

    def to_time_major(x):                                                       # [(b t), C] -> [(t b), C]  (.view(B, -1, C).permute(1, 0, 2))
        return x.view(B, T, C).transpose(0, 1).contiguous().view(T * B, C)
    # question as the query over the grounded visual features / the audio features (:1861-1880)
    att_v = mha_single_query(qst, to_time_major(vgrd_posi), m.avqatask_attn_v, training)
    src = linear(dropout(relu(linear(att_v, m.avqatask_linear11.weight, m.avqatask_linear11.bias)), m.avqatask_dropout1.p, training),
                 m.avqatask_linear12.weight, m.avqatask_linear12.bias)
    att_v = LayerNormFn.apply(AddFn.apply(att_v, dropout(src, m.avqatask_dropout2.p, training)), m.avqatask_norm1.weight,
                              m.avqatask_norm1.bias)
    att_a = mha_single_query(qst, to_time_major(audio_feat), m.avqatask_attn_a, training)
    src = linear(dropout(relu(linear(att_a, m.avqatask_linear21.weight, m.avqatask_linear21.bias)), m.avqatask_dropout3.p, training),
                 m.avqatask_linear22.weight, m.avqatask_linear22.bias)
    att_a = LayerNormFn.apply(AddFn.apply(att_a, dropout(src, m.avqatask_dropout4.p, training)), m.avqatask_norm2.weight,
                              m.avqatask_norm2.bias)
    feat = torch.cat((AddFn.apply(att_a, MeanFn.apply(audio_feat, B, T)), AddFn.apply(att_v, MeanFn.apply(vgrd_posi, B, T))), dim=-1)
    feat = linear(tanh(feat), m.avqatask_fc_fusion.weight, m.avqatask_fc_fusion.bias)
    combined = tanh(MulFn.apply(feat, qst))                                      # fusion with the question (:1890-1891)
    out_qa = linear(combined, m.avqatask_fc_ans.weight, m.avqatask_fc_ans.bias, out_f32=True)
    return out_qa, out_match_posi, out_match_nega
