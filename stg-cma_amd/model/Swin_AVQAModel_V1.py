"""Mirror of the BACKBONE of the reference's AVQA model, `SwinTransformer2D_Adapter_AVQA`
(AVQA/model/Swin_AVQAModel_V1.py:1220-1903), on the HIP path.

The AVQA blocks carry a third stream: the frames of a NEGATIVE clip ride through every block as the plain frozen Swin block
(window attention + FFN, drop_path on both residuals, no temporal attention, no adapters; :752-872) and through every
PatchMerging (:1150-1154).  Nothing trainable sits on or behind that stream, so it runs forward-only next to the
(video, audio) fusion stream, sharing its bf16 weight shadows and window tables.
`forward_features(a, v, v_nega)` = lines :1742-1766 of the reference forward: (f_v, f_a, visual_nega) = norm of each stream,
[(B T), 49, C_last] fp32 -- the inputs of the QA head.

The QA head (`avqatask_*`: question LSTM encoder, audio-visual grounding on the positive and the negative clip, two
single-query multi-head attentions, fusion MLPs; ctor :1420-1473, forward :1768-1903) is here too, under the reference's
module names, run by ..ops_head on libstgcma_hip.so; its widths are the reference's hard-coded 1536 (= Swin-L's last stage),
so `forward(a, v, v_nega, question, mode)` needs embed_dim = 192.  The constructor ingests a Swin image checkpoint together with a
grounding-pretraining checkpoint like the reference's (:1510-1568).  Not carried over: the fp8 weight path of BASELINE.json
config 5.
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
        if pretrained is not None and not isinstance(pretrained, str):
            raise TypeError('pretrained must be a str or None')
        super().__init__(label_dim=1, pretrained=None, img_size=img_size, patch_size=patch_size, num_frames=num_frames,
                         in_chans=in_chans, embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window_size,
                         mlp_ratio=mlp_ratio, frozen_stages=frozen_stages, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, ape=ape, patch_norm=patch_norm, t_relative=t_relative,
                         use_checkpoint=use_checkpoint, ftmode=ftmode, adapter_mlp_ratio=adapter_mlp_ratio)
        del self.mlp_head                       # the QA head replaces it in the reference (:1420-1473)
        del self.avgpool
        self._build_qa_head()
        self.pretrained, self.grounding_pretrained = None, None
        self.initialize_weights(pretrained=pretrained, grounding_pretrained=grounding_pretrained)

    def initialize_weights(self, pretrained=None, grounding_pretrained=None):
        """Swin_AVQAModel_V1.py:1500-1590: trunc_normal(.02) Linears / unit LayerNorms everywhere (the QA head included), then --
        when BOTH a Swin image checkpoint and a grounding-pretraining checkpoint are given (:1510-1512) -- their ingestion: the
        grounding model's fc_a2 / fc_gl / fc1..fc4 become avqatask_* (:1524-1541; fc_a1 and the *_pure copies have no counterpart in
        this model and end up as unexpected keys), patch-embedding inflation and the audio patch embedding = channel mean
        (:1547-1554), strict=False load; finally every adapter up-projection is zeroed."""
        from ._common import trunc_normal_

        def _init_weights(m):
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if not hasattr(self, "avqatask_fc_a2"):                 # the parent constructor's call: the head does not exist yet
            return super().initialize_weights(pretrained=None)
        if pretrained and grounding_pretrained:
            self.pretrained, self.grounding_pretrained = pretrained, grounding_pretrained
        if isinstance(self.pretrained, str):
            self.apply(_init_weights)
            print(f'load model from: {self.pretrained}')
            checkpoint = torch.load(self.pretrained, map_location='cpu')
            state_dict = checkpoint['model']
            print(f'load grounding pretraining model from: {self.grounding_pretrained}')
            grounding_checkpoint = torch.load(self.grounding_pretrained, map_location='cpu')
            tmp = ['module.fc_a1.weight', 'module.fc_a1.bias', 'module.fc_a2.weight', 'module.fc_a2.bias', 'module.fc_gl.weight',
                   'module.fc_gl.bias', 'module.fc1.weight', 'module.fc1.bias', 'module.fc2.weight', 'module.fc2.bias',
                   'module.fc3.weight', 'module.fc3.bias', 'module.fc4.weight', 'module.fc4.bias']
            tmp2 = tmp[:4]
            for k, v in grounding_checkpoint.items():
                if k in tmp:
                    state_dict[k.replace('module.', 'avqatask_')] = v
            for k, v in grounding_checkpoint.items():
                if k in tmp2:
                    parts = str(k).split('.')
                    state_dict[(parts[0] + '.' + parts[1] + '_pure.' + parts[-1]).replace('module.', 'avqatask_')] = v
            pe = state_dict['patch_embed.proj.weight'].unsqueeze(2).repeat(1, 1, self.patch_size[0], 1, 1) / self.patch_size[0]
            state_dict['patch_embed.proj.weight'] = pe
            state_dict['patch_embed_audio.proj.weight'] = torch.mean(pe.unsqueeze(2), dim=1)
            state_dict['patch_embed_audio.proj.bias'] = state_dict['patch_embed.proj.bias']
            state_dict['patch_embed_audio.norm.weight'] = state_dict['patch_embed.norm.weight']
            state_dict['patch_embed_audio.norm.bias'] = state_dict['patch_embed.norm.bias']
            msg = self.load_state_dict(state_dict, strict=False)
            print('Missing keys: {}'.format(msg.missing_keys))
            print('Unexpected keys: {}'.format(msg.unexpected_keys))
            print(f"=> loaded successfully '{self.pretrained}'")
            del checkpoint
        elif self.pretrained is None:
            self.apply(_init_weights)
        else:
            raise TypeError('pretrained must be a str or None')
        from .Swin_AVE import Adapter
        for n, m in self.layers.named_modules():
            if isinstance(m, Adapter):
                nn.init.constant_(m.D_fc2.weight, 0)
                nn.init.constant_(m.D_fc2.bias, 0)

    HEAD_DIM = 1536          # Swin_AVQAModel_V1.py: the head runs at the backbone's last-stage width
    PROJECT_FEATURES = False  # Swin_AVQAModel.py (512-d variant): avqatask_yb_fc_v / _a project the backbone features first

    def _build_qa_head(self):
        """Parameter containers of the reference's QA head under its names (Swin_AVQAModel_V1.py:1420-1473); nn.LSTM /
        nn.MultiheadAttention / nn.Embedding only HOLD the tensors here, the arithmetic is ops_head's."""
        Dh = self.HEAD_DIM
        if self.PROJECT_FEATURES:                         # Swin_AVQAModel.py:1421 (the 512-d variant only)
            self.avqatask_fc_a1 = nn.Linear(128, Dh)
        self.avqatask_fc_a2 = nn.Linear(Dh, Dh)
        self.avqatask_fc_fusion = nn.Linear(Dh + Dh, Dh)
        self.avqatask_linear11 = nn.Linear(Dh, Dh)
        self.avqatask_dropout1 = nn.Dropout(0.1)
        self.avqatask_linear12 = nn.Linear(Dh, Dh)
        self.avqatask_linear21 = nn.Linear(Dh, Dh)
        self.avqatask_dropout2 = nn.Dropout(0.1)
        self.avqatask_linear22 = nn.Linear(Dh, Dh)
        self.avqatask_norm1 = nn.LayerNorm(Dh)
        self.avqatask_norm2 = nn.LayerNorm(Dh)
        self.avqatask_dropout3 = nn.Dropout(0.1)
        self.avqatask_dropout4 = nn.Dropout(0.1)
        self.avqatask_attn_a = nn.MultiheadAttention(Dh, 4, dropout=0.1)
        self.avqatask_attn_v = nn.MultiheadAttention(Dh, 4, dropout=0.1)
        self.avqatask_question_encoder = QstEncoder(93, Dh, Dh, 1, Dh)
        self.avqatask_tanh = nn.Tanh()
        self.avqatask_fc_ans = nn.Linear(Dh, 42)
        self.avqatask_avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.avqatask_fc_gl = nn.Linear(Dh + Dh, Dh)
        self.avqatask_fc1 = nn.Linear(Dh + Dh, 512)
        self.avqatask_fc2 = nn.Linear(512, 256)
        self.avqatask_fc3 = nn.Linear(256, 128)
        self.avqatask_fc4 = nn.Linear(128, 2)
        if self.PROJECT_FEATURES:                         # Swin_AVQAModel.py:1472-1473
            self.avqatask_yb_fc_v = nn.Linear(1536, Dh)
            self.avqatask_yb_fc_a = nn.Linear(1536, 128)

    def forward_features(self, a, v, v_nega):
        """a: [B, T, Ha, Wa]; v, v_nega: [B, T, 3, H, W].  Returns (f_v, f_a, visual_nega), each [(B T), N_last, C_last] fp32;
        visual_nega carries no gradient."""
        f_v, f_a, f_n = self._backbone(a, v, v_nega=v_nega)[:3]
        BT = v.shape[0] * v.shape[1]
        return tuple(t.view(BT, -1, t.shape[-1]) for t in (f_v, f_a, f_n))

    def forward(self, a, v, v_nega, question, mode):
        """audio [B, T, Ha, Wa], visual_posi / visual_nega [B, T, 3, H, W], question int64 [B, L], mode
        (Swin_AVQAModel_V1.py:1654, 1740-1903) -> (out_qa [B, 42], out_match_posi [(B T), 2], out_match_nega [(B T), 2]), fp32."""
        if mode != 'fusion' or mode != self.ftmode:
            raise TypeError('ftmode is not expected !!!')
        f_v, f_a, f_n = self.forward_features(a, v, v_nega)
        if f_v.shape[-1] != 1536 or f_v.shape[1] != 49:
            raise NotImplementedError("the QA head is hard-wired to 49 tokens x 1536 channels (Swin-L at 224 x 224, :1776-1777)")
        from ..ops_head import avqa_head_forward
        with torch.cuda.device(f_v.device):     # the head's launches follow the backbone's device (kernels._stream)
            return avqa_head_forward(self, f_v, f_a, f_n, question, v.shape[0], v.shape[1], self.training)


class QstEncoder(nn.Module):
    """Parameter container of the reference's question encoder (Swin_AVQAModel_V1.py:37-59); run by ops_head.question_encoder."""

    def __init__(self, qst_vocab_size, word_embed_size, embed_size, num_layers, hidden_size):
        super().__init__()
        self.word2vec = nn.Embedding(qst_vocab_size, word_embed_size)
        self.tanh = nn.Tanh()
        self.lstm = nn.LSTM(word_embed_size, hidden_size, num_layers)
        self.fc = nn.Linear(2 * num_layers * hidden_size, embed_size)

    def forward(self, question):
        from ..ops_head import question_encoder
        return question_encoder(self, question)
