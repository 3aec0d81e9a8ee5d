"""Mirror of the BACKBONE of the reference's AVQA model, `SwinTransformer2D_Adapter_AVQA`
(AVQA/model/Swin_AVQAModel_V1.py:1220-1903), on the HIP path.

The AVQA blocks carry a third stream: the frames of a NEGATIVE clip ride through every block as the plain frozen Swin block
(window attention + FFN, drop_path on both residuals, no temporal attention, no adapters; :752-872) and through every
PatchMerging (:1150-1154).  Nothing trainable sits on or behind that stream, so it runs forward-only next to the
(video, audio) fusion stream, sharing its bf16 weight shadows and window tables.
`forward_features(a, v, v_nega)` = lines :1742-1766 of the reference forward: (f_v, f_a, visual_nega) = norm of each stream,
[(B T), 49, C_last] fp32 -- the inputs of the QA head (question LSTM, grounding, fusion MLPs; `avqatask_*`, :1768-1903), which
is SURVEY section 8f rank 2 and not part of this build: `forward` raises.
"""
import torch
import torch.nn as nn

from .Swin_AVE import SwinTransformer2D_Adapter_New


class SwinTransformer2D_Adapter_AVQA(SwinTransformer2D_Adapter_New):
    def __init__(self, grounding_pretrained=None, pretrained=None, img_size=224, patch_size=[1, 4, 4], num_frames=10, in_chans=3,
                 embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, mlp_ratio=4., frozen_stages=-1,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm,
                 ape=False, patch_norm=True, t_relative=True, use_checkpoint=False, ftmode='videoonly',
                 adapter_mlp_ratio=[0.25, 0.25, 0.25, 0.25], **kwargs):
        super().__init__(label_dim=1, pretrained=pretrained, img_size=img_size, patch_size=patch_size, num_frames=num_frames,
                         in_chans=in_chans, embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window_size,
                         mlp_ratio=mlp_ratio, frozen_stages=frozen_stages, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, ape=ape, patch_norm=patch_norm, t_relative=t_relative,
                         use_checkpoint=use_checkpoint, ftmode=ftmode, adapter_mlp_ratio=adapter_mlp_ratio)
        self.grounding_pretrained = grounding_pretrained
        del self.mlp_head                       # the QA head replaces it in the reference; out of scope here
        del self.avgpool

    def forward_features(self, a, v, v_nega):
        """a: [B, T, Ha, Wa]; v, v_nega: [B, T, 3, H, W].  Returns (f_v, f_a, visual_nega), each [(B T), N_last, C_last] fp32;
        visual_nega carries no gradient."""
        f_v, f_a, f_n = self._backbone(a, v, v_nega=v_nega)[:3]
        BT = v.shape[0] * v.shape[1]
        return tuple(t.view(BT, -1, t.shape[-1]) for t in (f_v, f_a, f_n))

    def forward(self, *args, **kwargs):
        raise NotImplementedError("the AVQA question-answering head (avqatask_*, Swin_AVQAModel_V1.py:1768-1903) is not part of this "
                                  "build yet (SURVEY.md section 8f); call forward_features(a, v, v_nega) for the backbone")
