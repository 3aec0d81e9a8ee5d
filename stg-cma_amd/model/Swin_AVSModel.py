"""Mirror of the BACKBONE of the reference's AVS model, `SwinTransformer2D_Adapter_AVS`
(AVS/model/Swin_AVSModel.py:1266-1895), on the HIP path.

What is here: the constructor keywords that shape the backbone, the backbone's parameters under the reference's names
(`patch_embed.*`, `patch_embed_audio.*`, `layers.*`, `norm.*` -- the subset of the reference state_dict that a Swin
checkpoint + adapter fine-tune touches), and `forward_features(a, v)` = lines :1790-1830 of the reference forward: the
multi-scale video features (before each downsample, the last one through `norm`) and the pooled audio feature that feed the
decoder.  The dense decoder (`avstask_*`: 4 Linear taps, 4 ASPP classifiers, 4 TPAVI blocks, the FeatureFusion path and the output
convolutions; ctor :1474-1503, forward :1838-1894, AVS/model/TPAVI.py) is here too under the reference's module names, run by
..ops_dec on libstgcma_hip.so (channels-last rows, im2col + MFMA GEMM convolutions); `forward(a, v, mode)` returns
`(pred, feature_map_list, a_fea_list)` like the reference, with either or both of the non-local forms of a stage
(tpavi_va_flag: audio-visual, what the runners use; tpavi_vv_flag: visual self-attention through the same block, :1532-1538).
"""
import torch
import torch.nn as nn

from .Swin_AVE import SwinTransformer2D_Adapter_New


class Classifier_Module(nn.Module):
    """ASPP classifier (Swin_AVSModel_Base.py:14-29): parameter container, run by ops_dec.aspp."""

    def __init__(self, dilation_series, padding_series, NoLabels, input_channel):
        super().__init__()
        self.conv2d_list = nn.ModuleList()
        for dilation, padding in zip(dilation_series, padding_series):
            self.conv2d_list.append(nn.Conv2d(input_channel, NoLabels, kernel_size=3, stride=1, padding=padding, dilation=dilation, bias=True))
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)


class ResidualConvUnit(nn.Module):
    """(:45-75) two 3x3 convolutions; the reference's in-place ReLU semantics live in ops_dec.residual_conv_unit."""

    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.relu = nn.ReLU(inplace=True)


class FeatureFusionBlock(nn.Module):
    """(:78-112) parameter container, run by ops_dec.feature_fusion."""

    def __init__(self, features):
        super().__init__()
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)


class Interpolate(nn.Module):
    """(:115-130) bilinear x2; no parameters."""

    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners


class TPAVIModule(nn.Module):
    """Parameter container of AVS/model/TPAVI.py:6-79 (mode 'dot', dimension 3, bn_layer); run by ops_dec.tpavi."""

    def __init__(self, in_channels, inter_channels=None, mode='dot', dimension=3, bn_layer=True):
        super().__init__()
        if mode != 'dot' or dimension != 3 or not bn_layer:
            raise NotImplementedError("TPAVIModule: mode='dot', dimension=3, bn_layer=True (what the AVS models build)")
        self.mode, self.dimension, self.in_channels = mode, dimension, in_channels
        self.inter_channels = inter_channels if inter_channels is not None else max(in_channels // 2, 1)
        self.align_channel = nn.Linear(128, in_channels)
        self.norm_layer = nn.LayerNorm(in_channels)
        self.g = nn.Conv3d(in_channels=self.in_channels, out_channels=self.inter_channels, kernel_size=1)
        self.W_z = nn.Sequential(nn.Conv3d(in_channels=self.inter_channels, out_channels=self.in_channels, kernel_size=1),
                                 nn.BatchNorm3d(self.in_channels))
        nn.init.constant_(self.W_z[1].weight, 0)
        nn.init.constant_(self.W_z[1].bias, 0)
        self.theta = nn.Conv3d(in_channels=self.in_channels, out_channels=self.inter_channels, kernel_size=1)
        self.phi = nn.Conv3d(in_channels=self.in_channels, out_channels=self.inter_channels, kernel_size=1)


class SwinTransformer2D_Adapter_AVS(SwinTransformer2D_Adapter_New):
    def __init__(self, pretrained=None, img_size=224, patch_size=[1, 4, 4], num_frames=5, in_chans=3, embed_dim=128,
                 depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, mlp_ratio=4., frozen_stages=-1, qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False,
                 patch_norm=True, t_relative=True, use_checkpoint=False, ftmode='videoonly',
                 adapter_mlp_ratio=[0.25, 0.25, 0.25, 0.25], channel=256, opt=None, config=None, vis_dim=[64, 128, 320, 512],
                 tpavi_stages=[0, 1, 2, 3], tpavi_vv_flag=False, tpavi_va_flag=True, **kwargs):
        super().__init__(label_dim=1, pretrained=pretrained, img_size=img_size, patch_size=patch_size, num_frames=num_frames,
                         in_chans=in_chans, embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window_size,
                         mlp_ratio=mlp_ratio, frozen_stages=frozen_stages, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, ape=ape, patch_norm=patch_norm, t_relative=t_relative,
                         use_checkpoint=use_checkpoint, ftmode=ftmode, adapter_mlp_ratio=adapter_mlp_ratio)
        del self.mlp_head                       # the AVS model has no classification head; the dense decoder replaces it
        self.tpavi_stages, self.tpavi_vv_flag, self.tpavi_va_flag, self.vis_dim = tpavi_stages, tpavi_vv_flag, tpavi_va_flag, vis_dim
        self._build_decoder(channel)

    def _build_decoder(self, channel):
        """Parameter containers of the dense decoder under the reference's names (Swin_AVSModel_Base.py:1474-1503; the Linear taps
        take the stage widths embed_dim * 2^s, which the two reference files hard-code as 128.. / 192..)."""
        dil = [3, 6, 12, 18]
        self.avstask_conv4 = Classifier_Module(dil, dil, channel, self.vis_dim[3])
        self.avstask_conv3 = Classifier_Module(dil, dil, channel, self.vis_dim[2])
        self.avstask_conv2 = Classifier_Module(dil, dil, channel, self.vis_dim[1])
        self.avstask_conv1 = Classifier_Module(dil, dil, channel, self.vis_dim[0])
        self.avstask_path4 = FeatureFusionBlock(channel)
        self.avstask_path3 = FeatureFusionBlock(channel)
        self.avstask_path2 = FeatureFusionBlock(channel)
        self.avstask_path1 = FeatureFusionBlock(channel)
        E = self.embed_dim
        self.avstask_x1_linear = nn.Linear(E, 64)
        self.avstask_x2_linear = nn.Linear(2 * E, 128)
        self.avstask_x3_linear = nn.Linear(4 * E, 320)
        self.avstask_x4_linear = nn.Linear(8 * E, 512)
        self.avstask_audio_linear = nn.Linear(8 * E, 128)
        for i in self.tpavi_stages:
            setattr(self, f"avstask_tpavi_b{i + 1}", TPAVIModule(in_channels=channel, mode='dot'))
        self.avstask_output_conv = nn.Sequential(
            nn.Conv2d(channel, 128, kernel_size=3, stride=1, padding=1),
            Interpolate(scale_factor=2, mode="bilinear"),
            nn.Conv2d(128, 32, kernel_size=3, stride=1, padding=1),
            nn.ReLU(True),
            nn.Conv2d(32, 1, kernel_size=1, stride=1, padding=0),
        )
        for m in (self.avstask_x1_linear, self.avstask_x2_linear, self.avstask_x3_linear, self.avstask_x4_linear,
                  self.avstask_audio_linear):                     # initialize_weights' trunc_normal(.02) on every nn.Linear (:1553-1556)
            nn.init.trunc_normal_(m.weight, std=.02)
            nn.init.constant_(m.bias, 0)
        for i in self.tpavi_stages:
            tp = getattr(self, f"avstask_tpavi_b{i + 1}")
            nn.init.trunc_normal_(tp.align_channel.weight, std=.02)
            nn.init.constant_(tp.align_channel.bias, 0)

    def forward_features(self, a, v):
        """a: [B, T, Ha, Wa] spectrogram segments, v: [B, T, 3, H, W] frames (the reference rearranges 'b t c h w -> b c t h w',
        :1793).  Returns (multi_scale, a_feat): multi_scale[s] = video tokens [(B T), N_s, C_s] before the downsample of stage
        s (fp32), the last one through `norm` (:1813-1821); a_feat = norm(a) [(B T), N_last, C_last] (:1824), whose token
        mean is the reference's pooled audio feature (:1830-1832)."""
        outs = self._backbone(a, v, taps=True)
        f_v, f_a, taps = outs[0], outs[1], list(outs[2:])
        BT = v.shape[0] * v.shape[1]
        ms = [t.view(BT, -1, t.shape[-1]) for t in taps] + [f_v.view(BT, -1, f_v.shape[-1])]
        return ms, f_a.view(BT, -1, f_a.shape[-1])

    def forward(self, a, v, mode):
        """audio [B, T, Ha, Wa], frames [B, T, 3, H, W], mode (Swin_AVSModel_Base.py:1704, 1790-1894) ->
        (pred fp32 [(B T), 1, H, W], feature_map_list (4 x [(B T), 256, h, w]), a_fea_list (4 x [B, T, 256]))."""
        if mode != 'fusion' or mode != self.ftmode:
            raise TypeError('ftmode is not expected !!!')
        ms, a_feat = self.forward_features(a, v)
        if int(round(ms[0].shape[1] ** 0.5)) ** 2 != ms[0].shape[1]:
            raise NotImplementedError("the dense decoder expects square token maps (:1838-1841)")
        from ..ops_dec import avs_decoder_forward
        with torch.cuda.device(a_feat.device):  # the decoder's launches follow the backbone's device (kernels._stream)
            return avs_decoder_forward(self, ms, a_feat, v.shape[0], v.shape[1], self.training)


class SwinTransformer2D_Adapter_AVS_Base(SwinTransformer2D_Adapter_AVS):
    """AVS/model/Swin_AVSModel_Base.py:1266 (Swin-B widths); same code, the Linear taps follow embed_dim."""
