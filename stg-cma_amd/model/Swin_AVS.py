"""Mirror of the BACKBONE of the reference's AVS model, `SwinTransformer2D_Adapter_AVS`
(AVS/model/Swin_AVSModel.py:1266-1895), on the HIP path.

What is here: the constructor keywords that shape the backbone, the backbone's parameters under the reference's names
(`patch_embed.*`, `patch_embed_audio.*`, `layers.*`, `norm.*` -- the subset of the reference state_dict that a Swin
checkpoint + adapter fine-tune touches), and `forward_features(a, v)` = lines :1790-1830 of the reference forward: the
multi-scale video features (before each downsample, the last one through `norm`) and the pooled audio feature that feed the
decoder.  What is not: the dense decoder head (`avstask_*`: 4 ASPP classifiers, 4 TPAVI blocks, FPN path, :1474-1506,
:1838-1894) -- SURVEY section 8f rank 1, next.  `forward` therefore raises; a maintainer swaps the reference's backbone loop
(:1790-1822) for `forward_features`.
"""
import torch
import torch.nn as nn

from .Swin_AVE import SwinTransformer2D_Adapter_New


class SwinTransformer2D_Adapter_AVS(SwinTransformer2D_Adapter_New):
    def __init__(self, pretrained=None, img_size=224, patch_size=[1, 4, 4], num_frames=5, in_chans=3, embed_dim=128,
                 depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, mlp_ratio=4., frozen_stages=-1, qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False,
                 patch_norm=True, t_relative=True, use_checkpoint=False, ftmode='videoonly',
                 adapter_mlp_ratio=[0.25, 0.25, 0.25, 0.25], **kwargs):
        super().__init__(label_dim=1, pretrained=pretrained, img_size=img_size, patch_size=patch_size, num_frames=num_frames,
                         in_chans=in_chans, embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window_size,
                         mlp_ratio=mlp_ratio, frozen_stages=frozen_stages, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, ape=ape, patch_norm=patch_norm, t_relative=t_relative,
                         use_checkpoint=use_checkpoint, ftmode=ftmode, adapter_mlp_ratio=adapter_mlp_ratio)
        del self.mlp_head                       # the AVS model has no classification head; its decoder is out of scope here

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

    def forward(self, *args, **kwargs):
        raise NotImplementedError("the AVS dense decoder (avstask_*: ASPP + TPAVI + FPN, Swin_AVSModel.py:1838-1894) is not part "
                                  "of this build yet (SURVEY.md section 8f); call forward_features(a, v) for the backbone")
