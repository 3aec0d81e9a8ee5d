"""The AVS dense decoder (AVS/model/Swin_AVSModel_Base.py:14-130 building blocks, :1474-1506 ctor, :1838-1894 forward;
AVS/model/TPAVI.py) on libstgcma_hip.so.

Layout: every feature map stays channels-last as token rows [F*H*W, C] bf16 -- exactly how the backbone hands over its
multi-scale taps -- so the reference's NHWC -> NCHW rearranges (:1848-1851) disappear, a 3x3 (dilated) convolution is an im2col
gather (dec.hip) + the MFMA GEMM (stg_gemm_nt), its data gradient the same gather on dY with the flipped kernel matrix, its
weight gradient stg_wgrad_tn on (dY, im2col(X)).  TPAVI's (T H W) x (T H W) affinity is never formed: in `dot` mode with the
audio broadcast over the frame, y = theta(x) . M with M = (1/T) sum_t phi(a_t) (x) mean_hw g(x)_t, a 128 x 128 matrix per clip.

Like ops_head, the decoder is written op by op as torch.autograd.Functions whose forward and backward are library launches;
autograd chains them.  ATen only moves data (kernel-matrix re-layouts of the small conv weights, the NCHW copies of the returned
feature maps) and does the per-channel BatchNorm scalar arithmetic on [C]-sized vectors.
"""
import torch

from . import kernels as K
from .kernels import BF16, F32
from .ops import f32c
from .ops_head import AddFn, CastFn, LayerNormFn, MeanFn, _bf, linear, relu


class Conv3x3Fn(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, 3, stride 1, padding = dilation = d, bias) on rows [F*H*W, Cin] -> [F*H*W, Cout] (Cout % 8 == 0)."""

    @staticmethod
    def forward(ctx, x, W, b, F_, H, Wd, d):
        O, I = W.shape[0], W.shape[1]
        Wm = K.cast_bf16(W.detach().permute(0, 2, 3, 1).reshape(O, 9 * I).contiguous())          # [O, (kh, kw, i)]
        if K.conv3x3_gemm_supported(I):            # implicit GEMM: the 9x im2col image is never written
            y = K.gemm_nt(x.contiguous(), Wm, f32c(b) if b is not None else None, conv=(H, Wd, d))
        else:
            y = K.gemm_nt(K.im2col3x3(x.contiguous(), F_, H, Wd, d), Wm, f32c(b) if b is not None else None)
        ctx.save_for_backward(x)
        ctx.W, ctx.has_b, ctx.geom = W, b is not None, (F_, H, Wd, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        W = ctx.W
        O, I = W.shape[0], W.shape[1]
        F_, H, Wd, d = ctx.geom
        dyb = _bf(dy)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            Wf = K.cast_bf16(W.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(I, 9 * O).contiguous())   # [I, (kh, kw, o)], flipped taps
            dx = K.gemm_nt(dyb, Wf, conv=(H, Wd, d)) if K.conv3x3_gemm_supported(O) else K.gemm_nt(K.im2col3x3(dyb, F_, H, Wd, d), Wf)
        if ctx.needs_input_grad[1]:
            want_db = ctx.has_b and ctx.needs_input_grad[2]
            if K.conv3x3_wgrad_supported(O, I, H, Wd):      # tn-GEMM with the tap gather inside: no im2col image, no atomics
                res = K.conv3x3_wgrad(dyb.contiguous(), x.contiguous(), F_, H, Wd, d, want_db=want_db)
                dWm, db = res if want_db else (res, None)
            else:
                dWm = torch.zeros((O, 9 * I), dtype=F32, device=x.device)
                db = torch.zeros((O,), dtype=F32, device=x.device) if want_db else None
                K.wgrad_tn(dyb, K.im2col3x3(x.contiguous(), F_, H, Wd, d), dWm, db)
            dW = dWm.view(O, 3, 3, I).permute(0, 3, 1, 2).contiguous()
        return dx, dW, db, None, None, None, None


def conv3x3(x, conv, geom):
    F_, H, Wd = geom
    d = conv.dilation[0]
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (d, d) or conv.dilation != (d, d) or conv.groups != 1:
        raise NotImplementedError("conv3x3: 3x3, stride 1, padding == dilation convolutions only")
    return Conv3x3Fn.apply(x, conv.weight, conv.bias, F_, H, Wd, d)


class BilinearUp2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, F_, H, Wd, align):
        ctx.geom = (F_, H, Wd, align)
        return K.bilinear_up2_fwd(x.contiguous(), F_, H, Wd, align)

    @staticmethod
    def backward(ctx, dy):
        F_, H, Wd, align = ctx.geom
        return K.bilinear_up2_bwd(_bf(dy), F_, H, Wd, align), None, None, None, None


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm3d over the rows of [R, C] (TPAVI.py:57-61): batch statistics (+ running-statistics update) in training,
    running statistics in eval.  gamma / beta are the module's weight / bias (differentiable inputs)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, training):
        x = x.contiguous()
        R = x.shape[0]
        if training:
            mean = (K.bn_colsum(x)[0] / R).contiguous()
            s2 = K.bn_colsum(x, mean=mean)                        # centred second pass: E[x^2] - mean^2 cancels when |mean| >> spread
            mean = mean + s2[0] / R
            var = (s2[1] / R - (s2[0] / R) ** 2).clamp_min_(0.)
            with torch.no_grad():
                mom = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var * (R / max(R - 1, 1)), alpha=mom)
                bn.num_batches_tracked.add_(1)
        else:
            mean, var = bn.running_mean.float(), bn.running_var.float()
        rstd = torch.rsqrt(var + bn.eps).contiguous()
        mean = mean.contiguous()
        y = K.bn_apply(x, mean, rstd, f32c(gamma), f32c(beta))
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.training = gamma, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dyb = _bf(dy)
        s = K.bn_colsum(x, dyb, mean, rstd)                 # (sum dy, sum dy * xhat) = (dbeta, dgamma)
        dx = K.bn_bwd(x, dyb, mean, rstd, f32c(ctx.gamma), s if ctx.training else None)
        return dx, s[1].clone(), s[0].clone(), None, None


def batchnorm(x, bn, training):
    return BatchNormFn.apply(x, bn.weight, bn.bias, bn, training)


class TpaviMixFn(torch.autograd.Function):
    """y[(b t hw), :] = theta[(b t hw), :] . M_b,  M_b = (1/T) sum_t phi[b, t]^T (x) gbar[b, t]  (TPAVI.py:113-139 in `dot` mode with the
    audio broadcast over the frame: f / N followed by f . g collapses to this 128 x 128 matrix per clip).
    theta bf16 [B*T*HW, Ci], phi / gbar bf16 [B*T, Ci]."""

    @staticmethod
    def forward(ctx, theta, phi, gbar, B, T, HW):
        theta, phi, gbar = theta.contiguous(), phi.contiguous(), gbar.contiguous()
        Ci = theta.shape[1]
        n = T * HW
        batched = Ci % 128 == 0
        if batched:                                   # all clips at once: one tn launch for the M_b, one batched GEMM for y
            Ms = K.bmm_tn(phi, gbar, T).contiguous()                                            # M_b * T = phi_b^T gbar_b
            y = K.gemm_nt(theta, Ms.transpose(1, 2).contiguous().to(BF16), alpha=1.0 / T, batch=B)
        else:
            y = torch.empty_like(theta)
            Ms = torch.zeros((B, Ci, Ci), dtype=F32, device=theta.device)
            for b in range(B):
                K.wgrad_tn(phi[b * T:(b + 1) * T], gbar[b * T:(b + 1) * T], Ms[b])
                K.gemm_nt(theta[b * n:(b + 1) * n], K.cast_bf16(Ms[b], transpose=True), out=y[b * n:(b + 1) * n], alpha=1.0 / T)
        ctx.save_for_backward(theta, phi, gbar, Ms)
        ctx.geom = (B, T, HW, batched)
        return y

    @staticmethod
    def backward(ctx, dy):
        theta, phi, gbar, Ms = ctx.saved_tensors
        B, T, HW, batched = ctx.geom
        Ci = theta.shape[1]
        n = T * HW
        dyb = _bf(dy).contiguous()
        if batched and T * Ci % 8 == 0:
            dtheta = K.gemm_nt(dyb, Ms.to(BF16).contiguous(), alpha=1.0 / T, batch=B)          # dy_b . M_b^T
            dM = K.bmm_tn(theta, dyb, n)                                                         # dM_b * T = theta_b^T dy_b
            # the [T, Ci] per-clip products: rows padded to the GEMM's 16-byte granularity are not needed (T rows per problem)
            dphi = K.gemm_nt(gbar, dM.to(BF16).contiguous(), alpha=1.0 / T, batch=B)             # gbar_b . dM_b^T
            dgbar = K.gemm_nt(phi, dM.transpose(1, 2).contiguous().to(BF16), alpha=1.0 / T, batch=B)
            return dtheta, dphi, dgbar, None, None, None
        dtheta = torch.empty_like(theta)
        dphi, dgbar = torch.empty_like(phi), torch.empty_like(gbar)
        for b in range(B):
            K.gemm_nt(dyb[b * n:(b + 1) * n], K.cast_bf16(Ms[b]), out=dtheta[b * n:(b + 1) * n], alpha=1.0 / T)
            dM = torch.zeros((Ci, Ci), dtype=F32, device=theta.device)
            K.wgrad_tn(theta[b * n:(b + 1) * n], dyb[b * n:(b + 1) * n], dM)                     # dM_b * T = theta_b^T dy_b
            K.gemm_nt(gbar[b * T:(b + 1) * T], K.cast_bf16(dM), out=dphi[b * T:(b + 1) * T], alpha=1.0 / T)
            K.gemm_nt(phi[b * T:(b + 1) * T], K.cast_bf16(dM, transpose=True), out=dgbar[b * T:(b + 1) * T], alpha=1.0 / T)
        return dtheta, dphi, dgbar, None, None, None


def _conv1x1_as_linear(x, conv):
    """1 x 1 (x 1) convolution on channels-last rows = a Linear with the kernel's trailing unit dims dropped."""
    W = conv.weight
    return linear(x, W.view(W.shape[0], W.shape[1]), conv.bias)


def aspp(x, mod, geom):
    """Classifier_Module.forward (:25-29): the sum of four dilated 3x3 convolutions of the same input."""
    out = conv3x3(x, mod.conv2d_list[0], geom)
    for conv in list(mod.conv2d_list)[1:]:
        out = AddFn.apply(out, conv3x3(x, conv, geom))
    return out


def residual_conv_unit(r, rcu, geom):
    """ResidualConvUnit.forward (:63-75) on an input that is ALREADY its in-place-ReLU'd self: nn.ReLU(inplace=True) overwrites x,
    so the reference's `out + x` adds relu(x); the caller passes r = relu(x)."""
    out = conv3x3(r, rcu.conv1, geom)
    out = conv3x3(relu(out), rcu.conv2, geom)
    return AddFn.apply(out, r)


def feature_fusion(ffb, x0, x1, geom):
    """FeatureFusionBlock.forward (:95-112): x0 (+ RCU1(x1)) -> RCU2 -> bilinear x2 (align_corners=True).  Returns (out, relu(x1)):
    the in-place ReLU of RCU1 is visible to the caller of the reference (its feature_map_list holds the same tensors)."""
    r1 = None
    out = x0
    if x1 is not None:
        r1 = relu(x1)
        out = AddFn.apply(out, residual_conv_unit(r1, ffb.resConfUnit1, geom))
    out = residual_conv_unit(relu(out), ffb.resConfUnit2, geom)
    F_, H, Wd = geom
    return BilinearUp2Fn.apply(out, F_, H, Wd, True), r1


def tpavi(mod, x, audio, B, T, HW, training, out_scale=1.0):
    """TPAVIModule.forward (TPAVI.py:81-152), mode 'dot', dimension 3: x bf16 [(b t hw), C], audio bf16 [(b t), 128]
    -> (z [(b t hw), C], audio_temp [(b t), C]).  audio None: the visual self-attention form (`audio = x`, TPAVI.py:96-98;
    tpavi_vv_flag, Swin_AVSModel_Base.py:1532-1538) -> (z, None).  Either way f / N followed by f . g collapses to one 128 x 128
    matrix per clip (TpaviMixFn): M_b = phi_b^T g_b / n over the clip's n = T HW positions; with audio, phi is constant over a frame
    and the sum over its positions is HW gbar.  out_scale: a factor on z, folded into the LayerNorm's affine (the caller's average
    over the vv and va forms, :1885)."""
    if mod.mode != 'dot' or mod.dimension != 3:
        raise NotImplementedError("TPAVI: mode='dot', dimension=3 (what the AVS models build, Swin_AVSModel_Base.py:1495)")
    g_x = _conv1x1_as_linear(x, mod.g)
    theta = _conv1x1_as_linear(x, mod.theta)
    if audio is None:
        a_t = None
        y = TpaviMixFn.apply(theta, _conv1x1_as_linear(x, mod.phi), g_x, B, T * HW, 1)
    else:
        a_t = linear(audio, mod.align_channel.weight, mod.align_channel.bias)                  # [(b t), C]
        phi = _conv1x1_as_linear(a_t, mod.phi)                                                 # audio is constant over the frame
        gbar = MeanFn.apply(g_x, B * T, HW)
        y = TpaviMixFn.apply(theta, phi, gbar, B, T, HW)
    w_y = batchnorm(_conv1x1_as_linear(y, mod.W_z[0]), mod.W_z[1], training)
    lw, lb = mod.norm_layer.weight, mod.norm_layer.bias
    if out_scale != 1.0:
        lw, lb = lw * out_scale, lb * out_scale
    z = LayerNormFn.apply(AddFn.apply(w_y, x), lw, lb)
    return z, a_t


def output_conv(seq, x, geom):
    """avstask_output_conv (:1497-1503): conv3x3 -> bilinear x2 (align_corners=False) -> conv3x3 -> ReLU -> conv1x1 -> fp32."""
    F_, H, Wd = geom
    y = conv3x3(x, seq[0], geom)
    y = BilinearUp2Fn.apply(y, F_, H, Wd, bool(seq[1].align_corners))
    y = relu(conv3x3(y, seq[2], (F_, 2 * H, 2 * Wd)))
    W = seq[4].weight
    return linear(y, W.view(W.shape[0], W.shape[1]), seq[4].bias, out_f32=True)


def avs_decoder_forward(m, ms, a_feat, B, T, training):
    """Lines :1824-1894 of the reference forward.  ms: the four multi-scale video taps fp32 [(B T), N_s, C_s]; a_feat fp32
    [(B T), N_last, C_last].  Returns (pred fp32 [(B T), 1, 4 H0, 4 W0], feature_map_list (NCHW, fp32), a_fea_list)."""
    BT = B * T
    n_a, C_a = a_feat.shape[1], a_feat.shape[2]
    audio = MeanFn.apply(CastFn.apply(a_feat.reshape(BT * n_a, C_a)), BT, n_a)                 # AdaptiveAvgPool1d(1) (:1830-1832)
    audio = linear(audio, m.avstask_audio_linear.weight, m.avstask_audio_linear.bias)          # [(b t), 128]
    lins = (m.avstask_x1_linear, m.avstask_x2_linear, m.avstask_x3_linear, m.avstask_x4_linear)
    convs = (m.avstask_conv1, m.avstask_conv2, m.avstask_conv3, m.avstask_conv4)
    feats, geoms = [], []
    for s in range(4):
        N_s, C_s = ms[s].shape[1], ms[s].shape[2]
        side = int(round(N_s ** 0.5))
        geom = (BT, side, side)
        x = linear(CastFn.apply(ms[s].reshape(BT * N_s, C_s)), lins[s].weight, lins[s].bias)
        feats.append(aspp(x, convs[s], geom))                                                  # [(b t h w), 256]
        geoms.append(geom)
    a_fea_list = [None] * 4
    if len(m.tpavi_stages) > 0:
        if (not m.tpavi_vv_flag) and (not m.tpavi_va_flag):
            raise Exception('tpavi_vv_flag and tpavi_va_flag cannot be False at the same time if len(tpavi_stages)>0')
        sc = 1.0 / (int(bool(m.tpavi_vv_flag)) + int(bool(m.tpavi_va_flag)))                    # conv_feat /= tpavi_count (:1885)
        for i in m.tpavi_stages:
            blk, hw, z = getattr(m, f'avstask_tpavi_b{i + 1}'), geoms[i][1] * geoms[i][2], None
            if m.tpavi_vv_flag:                                                                # the same block, visual self-attention first (:1876-1879)
                z, _ = tpavi(blk, feats[i], None, B, T, hw, training, out_scale=sc)
            if m.tpavi_va_flag:
                z_va, a_t = tpavi(blk, feats[i], audio, B, T, hw, training, out_scale=sc)
                z = z_va if z is None else AddFn.apply(z, z_va)
                a_fea_list[i] = a_t.view(B, T, -1)
            feats[i] = z
    paths = (m.avstask_path1, m.avstask_path2, m.avstask_path3, m.avstask_path4)
    out, _ = feature_fusion(paths[3], feats[3], None, geoms[3])                                # path4(fm[3])            (:1887)
    # path4 has no second input, so its RCU2's in-place ReLU lands on feature_map_list[3] itself
    feats[3] = relu(feats[3])
    for s in (2, 1, 0):                                                                        # path3 / 2 / 1            (:1888-1890)
        out, r1 = feature_fusion(paths[s], out, feats[s], geoms[s])
        feats[s] = r1
    F_, H, Wd = geoms[0]
    pred = output_conv(m.avstask_output_conv, out, (F_, 2 * H, 2 * Wd))
    pred = pred.view(BT, 1, 4 * H, 4 * Wd)
    fmaps = [f.view(BT, g[1], g[2], -1).permute(0, 3, 1, 2).float() for f, g in zip(feats, geoms)]
    a_fea_list = [a.float() if a is not None else None for a in a_fea_list]
    return pred, fmaps, a_fea_list
