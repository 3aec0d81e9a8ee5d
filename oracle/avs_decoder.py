"""ORACLE (test infrastructure, not product code): fp32 PyTorch-CPU restatement of the reference's AVS dense decoder -- the
building blocks of AVS/model/Swin_AVSModel_Base.py:14-130, TPAVIModule.forward in `dot` mode (AVS/model/TPAVI.py:81-152) and
lines :1824-1894 of SwinTransformer2D_Adapter_AVS_Base.forward -- as plain functions over a flat {state_dict key: tensor} dict.

Only tests/ may import this; the product path (stg-cma_amd/ops_dec.py) never does.  Functional, NCHW, no in-place tricks: the
reference's nn.ReLU(inplace=True) inside ResidualConvUnit is written out as what it computes (the residual adds relu(x), and the
caller's tensor becomes relu(x)).  TPAVI keeps the reference's explicit (T H W) x (T H W) affinity matrix here -- the product
collapses it algebraically, so agreeing with the same goldens pins that algebra.
Pinned by tests/golden/avs_decoder_modules.npz and avs_full_tiny.npz (the reference modules / model run in the build container,
tests/golden/make_golden.py::avs_modules_case / avs_full_case) through tests/test_oracle_cpu.py.
"""
import torch
import torch.nn.functional as F

from .swin import swin_backbone


def conv(P, pre, x, dilation=1, padding=None):
    w = P[pre + ".weight"]
    pad = (w.shape[-1] // 2) * dilation if padding is None else padding
    return F.conv2d(x, w, P.get(pre + ".bias"), stride=1, padding=pad, dilation=dilation)


def classifier_module(P, pre, x, dilations=(3, 6, 12, 18)):
    """Classifier_Module.forward (:25-29): sum of the dilated 3x3 convolutions."""
    out = conv(P, f"{pre}.conv2d_list.0", x, dilations[0])
    for i, d in enumerate(dilations[1:], start=1):
        out = out + conv(P, f"{pre}.conv2d_list.{i}", x, d)
    return out


def residual_conv_unit(P, pre, x):
    """ResidualConvUnit.forward (:63-75) with its in-place ReLU spelled out: returns (conv2(relu(conv1(relu(x)))) + relu(x), relu(x))."""
    r = F.relu(x)
    out = conv(P, pre + ".conv2", F.relu(conv(P, pre + ".conv1", r)))
    return out + r, r


def feature_fusion_block(P, pre, x0, x1=None):
    """FeatureFusionBlock.forward (:95-112).  Returns (output, what the caller's x1 (or x0 when alone) holds afterwards)."""
    out, seen = x0, None
    if x1 is not None:
        res, seen = residual_conv_unit(P, pre + ".resConfUnit1", x1)
        out = out + res
        out, _ = residual_conv_unit(P, pre + ".resConfUnit2", out)
    else:
        out, seen = residual_conv_unit(P, pre + ".resConfUnit2", out)
    return F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True), seen


def tpavi(P, pre, x, audio, bn_training=False, bn_stats=None, eps=1e-5):
    """TPAVIModule.forward (TPAVI.py:81-152), mode 'dot'.  x [B, C, T, H, W], audio [B, T, 128] -> (z, audio_temp); audio None: the
    visual self-attention form (`audio = x`, TPAVI.py:96-98; audio_temp is then 0).
    bn_training: batch statistics (the caller may read them back through bn_stats = {} to check the running-stat update)."""
    Bn, C, T, H, W = x.shape
    if audio is None:
        audio_temp, au = 0, x
    else:
        audio_temp = F.linear(audio, P[pre + ".align_channel.weight"], P[pre + ".align_channel.bias"])    # [B, T, C]
        au = audio_temp.permute(0, 2, 1)[:, :, :, None, None].expand(Bn, C, T, H, W)

    def c1(name, t):
        return F.conv3d(t, P[f"{pre}.{name}.weight"], P[f"{pre}.{name}.bias"])
    Ci = P[pre + ".g.weight"].shape[0]
    g_x = c1("g", x).reshape(Bn, Ci, -1).permute(0, 2, 1)
    theta_x = c1("theta", x).reshape(Bn, Ci, -1).permute(0, 2, 1)
    phi_x = c1("phi", au).reshape(Bn, Ci, -1)
    f = theta_x @ phi_x
    y = (f / f.shape[-1]) @ g_x
    y = y.permute(0, 2, 1).reshape(Bn, Ci, T, H, W)
    w_y = c1("W_z.0", y)
    if bn_training:
        mean = w_y.mean(dim=(0, 2, 3, 4))
        var = w_y.var(dim=(0, 2, 3, 4), unbiased=False)
        if bn_stats is not None:
            n = w_y.numel() // C
            bn_stats.update(mean=mean.detach(), var_unbiased=(var * n / (n - 1)).detach())
    else:
        mean, var = P[pre + ".W_z.1.running_mean"], P[pre + ".W_z.1.running_var"]
    sh = (1, C, 1, 1, 1)
    w_y = (w_y - mean.view(sh)) * torch.rsqrt(var.view(sh) + eps) * P[pre + ".W_z.1.weight"].view(sh) + P[pre + ".W_z.1.bias"].view(sh)
    z = (w_y + x).permute(0, 2, 3, 4, 1)
    z = F.layer_norm(z, (C,), P[pre + ".norm_layer.weight"], P[pre + ".norm_layer.bias"])
    return z.permute(0, 4, 1, 2, 3), audio_temp


def output_conv(P, pre, x):
    """avstask_output_conv (:1497-1503)."""
    y = conv(P, pre + ".0", x)
    y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False)
    y = F.relu(conv(P, pre + ".2", y))
    return conv(P, pre + ".4", y)


def avs_decoder(P, taps, f_a, B, T, tpavi_stages=(0, 1, 2, 3), bn_training=False, tpavi_vv=False, tpavi_va=True):
    """(:1824-1894) taps: 4 video token maps [(B T), N_s, C_s] (last one norm'd), f_a [(B T), N, C].  tpavi_vv / tpavi_va: the two
    non-local forms of a stage, averaged when both are on (:1873-1886; the SAME block runs both, vv first)."""
    BT = B * T
    audio = F.linear(f_a.mean(dim=1).view(B, T, -1), P["avstask_audio_linear.weight"], P["avstask_audio_linear.bias"])
    fmaps = []
    for s in range(4):
        side = int(round(taps[s].shape[1] ** 0.5))
        x = F.linear(taps[s].view(BT, side, side, -1), P[f"avstask_x{s + 1}_linear.weight"], P[f"avstask_x{s + 1}_linear.bias"])
        fmaps.append(classifier_module(P, f"avstask_conv{s + 1}", x.permute(0, 3, 1, 2)))
    afeas = [None] * 4
    for i in tpavi_stages:
        _, C, H, W = fmaps[i].shape
        x5 = fmaps[i].reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)
        acc, cnt = 0., 0
        if tpavi_vv:
            z, _ = tpavi(P, f"avstask_tpavi_b{i + 1}", x5, None, bn_training)
            acc, cnt = acc + z, cnt + 1
        if tpavi_va:
            z, a_t = tpavi(P, f"avstask_tpavi_b{i + 1}", x5, audio, bn_training)
            acc, cnt = acc + z, cnt + 1
            afeas[i] = a_t
        fmaps[i] = (acc / cnt).permute(0, 2, 1, 3, 4).reshape(BT, C, H, W)
    out, fmaps[3] = feature_fusion_block(P, "avstask_path4", fmaps[3])
    for s in (2, 1, 0):
        out, fmaps[s] = feature_fusion_block(P, f"avstask_path{s + 1}", out, fmaps[s])
    return output_conv(P, "avstask_output_conv", out), fmaps, afeas


def avs_forward(P, a, v, cfg, bn_training=False, tpavi_vv=False, tpavi_va=True):
    """SwinTransformer2D_Adapter_AVS(_Base).forward[fusion] (:1790-1894): backbone (oracle.swin.swin_backbone) + decoder."""
    out = swin_backbone(P, a, v, cfg)
    return avs_decoder(P, out["taps"], out["f_a"], v.shape[0], v.shape[1], bn_training=bn_training, tpavi_vv=tpavi_vv, tpavi_va=tpavi_va)
