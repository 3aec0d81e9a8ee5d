"""MI355X-native drop-in for the reference's AVE/model/CLIP_AVE.py (CLIP ViT-B/16 / L/14 image encoder + STG-CMA adapters):
same class names, constructor keywords, forward(a, v, mode), state_dict keys and parameter names.  Parameter containers and
orchestration only -- the arithmetic runs in libstgcma_hip.so through ..ops_vit; no eager / CPU fallback.

Not carried over: the module-level `clip` / `loratorch` imports (CLIP_AVE.py:7-8; `loratorch` is unused, `clip` is imported
lazily, only when a pretrained directory is given) and the dead ablation code.
"""
from collections import OrderedDict

import contextlib

import torch
import torch.nn as nn

from ..ops_vit import VitBlockFn, VitBlockSpec, VitModelFn, vit_block_param_names
from ._common import DropPath, trunc_normal_


class Adapter(nn.Module):
    """D_fc1 -> GELU -> D_fc2 bottleneck, optionally with skip (CLIP_AVE.py:13-31); d_h = int(D * 0.0625)."""

    def __init__(self, D_features, mlp_ratio=0.0625, act_layer=nn.GELU, skip_connect=True):
        super().__init__()
        self.skip_connect = skip_connect
        D_hidden_features = int(D_features * mlp_ratio)
        self.act = act_layer()
        self.D_fc1 = nn.Linear(D_features, D_hidden_features)
        self.D_fc2 = nn.Linear(D_hidden_features, D_features)


class LayerNorm(nn.LayerNorm):
    """fp32-computing LayerNorm parameter holder (CLIP_AVE.py:33-39)."""


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) (CLIP_AVE.py:41-43): fused into the c_fc GEMM epilogue on the HIP path."""


class ResidualAttentionBlock(nn.Module):
    """CLIP resblock with temporal / spatial / MLP adapters and gated cross-modal fusion (CLIP_AVE.py:46-429).
    forward takes the fused '(b t) n d' token tensor [BT*sum(n_tok), D] (video rows first); set `n_tok` first."""

    def __init__(self, d_model, n_head, attn_mask=None, scale=1., num_tadapter=1, num_frames=8, drop_path=0., mode='videoonly',
                 enable_fusion=False):
        super().__init__()
        if attn_mask is not None:
            raise NotImplementedError("attn_mask is always None in the reference model")
        if num_tadapter != 1:
            raise NotImplementedError("num_tadapter == 2 (t_adapter_in) is unused by the reference forward")
        self.mode, self.num_tadapter, self.enable_fusion = mode, num_tadapter, enable_fusion
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask, self.n_head, self.scale = attn_mask, n_head, scale
        if mode in ('video_adapt', 'multimodal_adapt_no_fusion', 'fusion_adapt'):
            self.MLP_Adapter = Adapter(d_model, skip_connect=False)
            self.S_Adapter = Adapter(d_model)
            self.T_Adapter = Adapter(d_model, skip_connect=False)
        if mode in ('audio_adapt', 'multimodal_adapt_no_fusion', 'fusion_adapt'):
            self.S_Adapter_Audio = Adapter(d_model)
            self.MLP_Adapter_Audio = Adapter(d_model, skip_connect=False)
            self.T_Adapter_Audio = Adapter(d_model, skip_connect=False)
        self.num_frames = num_frames
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.gate_v = nn.Parameter(torch.zeros(1))
        self.gate_a = nn.Parameter(torch.zeros(1))
        self._d, self._dp = d_model, float(drop_path)
        self.n_tok = None

    def spec(self, n_tok):
        return VitBlockSpec(self._d, self.n_head, self.num_frames, n_tok, mode=self.mode, drop_path=self._dp)

    def forward(self, X, n_tok=None):
        n_tok = n_tok or self.n_tok
        if n_tok is None:
            raise RuntimeError("ResidualAttentionBlock: pass n_tok=(tokens per frame of each modality)")
        spec = self.spec(n_tok)
        names = vit_block_param_names(spec)
        sd = dict(self.named_parameters())
        with torch.cuda.device(X.device) if X.is_cuda else contextlib.nullcontext():
            return VitBlockFn.apply(X, spec, tuple(names), self.training, torch.is_grad_enabled(), *[sd[n] for n in names])


class Transformer(nn.Module):
    def __init__(self, num_frames, width, layers, heads, attn_mask=None, num_tadapter=1, scale=1., drop_path=0.1, mode='cascaded',
                 enable_fusion_idx=6):
        super().__init__()
        self.width, self.layers = width, layers
        dpr = [x.item() for x in torch.linspace(0, drop_path, self.layers)]
        self.resblocks = nn.Sequential(*[
            ResidualAttentionBlock(width, heads, attn_mask, scale, num_tadapter, num_frames, dpr[i], mode=mode,
                                   enable_fusion=True if i >= enable_fusion_idx - 1 else False) for i in range(layers)])


class MM_CLIP_AVE(nn.Module):
    """CLIP ViT + STG-CMA for AVE (CLIP_AVE.py:716-1140).  forward(a, v, mode) -> fp32 logits [(B*T), label_dim]."""

    def __init__(self, label_dim, input_resolution=224, audio_length=1024, num_video_frames=10, patch_size=16, embed_dim=768,
                 layers=12, heads=8, drop_path_rate=0.2, num_tadapter=1, adapter_scale=0.5, pretrained=None, ftmode='videoonly'):
        super().__init__()
        self.ftmode, self.input_resolution, self.pretrained, self.embed_dim = ftmode, input_resolution, pretrained, embed_dim
        self.ori_num_patches = (input_resolution // patch_size) ** 2
        self.oringal_hw = int(self.ori_num_patches ** 0.5)
        self.conv1 = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = embed_dim ** -0.5
        self.layers = layers
        self.class_embedding = nn.Parameter(scale * torch.randn(embed_dim))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, embed_dim))
        self.ln_pre = LayerNorm(embed_dim)
        # audio patches: Conv2d(k=16, stride=patch_size) over (128, audio_length/10) drops the remainder (CLIP_AVE.py:737-743,859-865)
        self.f_dim = (128 - 16) // patch_size + 1
        self.t_dim = (int(audio_length * (1 / 10)) - 16) // patch_size + 1
        self.num_patches_a = self.f_dim * self.t_dim
        self.conv1_audio = nn.Conv2d(1, embed_dim, kernel_size=patch_size, stride=patch_size, bias=False)
        self.positional_embedding_audio = nn.Parameter(scale * torch.randn(self.num_patches_a + 1, embed_dim))
        self.num_video_frames = num_video_frames
        self.temporal_embedding = nn.Parameter(torch.zeros(1, num_video_frames, embed_dim))
        self.temporal_embedding_audio = nn.Parameter(torch.zeros(1, num_video_frames, embed_dim))
        bmode = {'videoonly': 'video_adapt', 'audioonly': 'audio_adapt', 'multimodal': 'multimodal_adapt_no_fusion',
                 'fusion': 'fusion_adapt'}.get(ftmode)
        if bmode is None:
            raise TypeError('ftmode is not expected !!!')
        kw = dict(enable_fusion_idx=int(layers * 2 / 3)) if ftmode == 'fusion' else {}
        self.transformer = Transformer(num_video_frames, embed_dim, layers, heads, num_tadapter=num_tadapter, scale=adapter_scale,
                                       drop_path=drop_path_rate, mode=bmode, **kw)
        self.ln_post = LayerNorm(embed_dim)
        if self.ftmode in ('multimodal', 'fusion'):
            self.mlp_head = nn.Sequential(nn.Linear(embed_dim * 2, 512), nn.Dropout(0.5), nn.Linear(512, label_dim))
        else:
            self.mlp_head = nn.Sequential(nn.LayerNorm(embed_dim), nn.Linear(embed_dim, label_dim))
        self.initialize_weights(pretrained=self.pretrained)

    def initialize_weights(self, pretrained=None):
        """trunc_normal(.02) Linears, unit LayerNorms, zeroed adapter D_fc2 (CLIP_AVE.py:788-975).  The reference's
        `clip.load` branch is replaced by load_state_dict of an already-converted checkpoint."""
        def _init_weights(m):
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if pretrained:
            self.pretrained = pretrained
        if isinstance(self.pretrained, str):
            self.apply(_init_weights)
            print(f'load model from: {self.pretrained}')
            try:
                import clip                      # OpenAI CLIP package: the reference imports it at module level (CLIP_AVE.py:7)
            except ImportError as e:
                raise ImportError("pretrained=<dir> loads OpenAI CLIP weights through the `clip` package (CLIP_AVE.py:817-820), "
                                  "which is not installed; install it, or convert the checkpoint once and load_state_dict() it") from e
            name = "ViT-B/16" if self.layers == 12 else "ViT-L/14"
            clip_model, _ = clip.load(name, device="cpu", download_root=self.pretrained)
            pretrain_dict = self.ingest_clip_visual_state(clip_model.visual.state_dict())
            del clip_model
            msg = self.load_state_dict(pretrain_dict, strict=False)
            print('Missing keys: {}'.format(msg.missing_keys))
            print('Unexpected keys: {}'.format(msg.unexpected_keys))
            print(f"=> loaded successfully '{self.pretrained}'")
        elif self.pretrained is None:
            self.apply(_init_weights)
        else:
            raise TypeError('pretrained must be a str or None')
        for m in self.transformer.modules():
            if isinstance(m, Adapter):
                nn.init.constant_(m.D_fc2.weight, 0)
                nn.init.constant_(m.D_fc2.bias, 0)

    def ingest_clip_visual_state(self, pretrain_dict):
        """CLIP image-encoder state_dict -> this model's keys (CLIP_AVE.py:821-850): drop the output projection, audio patch
        conv = channel SUM of the RGB one, audio positional embedding = the class slot + the centre crop (or bilinear resize when
        the audio grid is larger) of the 2-D grid of image positional embeddings."""
        pretrain_dict = dict(pretrain_dict)
        del pretrain_dict['proj']
        pretrain_dict['conv1_audio.weight'] = torch.sum(pretrain_dict['conv1.weight'], dim=1).unsqueeze(1)
        hw = self.oringal_hw
        ori = pretrain_dict['positional_embedding'].unsqueeze(dim=0)
        grid = ori[:, 1:, :].detach().reshape(1, self.ori_num_patches, self.embed_dim).transpose(1, 2).reshape(1, self.embed_dim, hw, hw)
        if self.t_dim <= hw:
            t0 = int(hw / 2) - int(self.t_dim / 2)
            grid = grid[:, :, :, t0:t0 + self.t_dim]
        else:
            grid = torch.nn.functional.interpolate(grid, size=(hw, self.t_dim), mode='bilinear')
        if self.f_dim <= hw:
            f0 = int(hw / 2) - int(self.f_dim / 2)
            grid = grid[:, :, f0:f0 + self.f_dim, :]
        else:
            grid = torch.nn.functional.interpolate(grid, size=(self.f_dim, self.t_dim), mode='bilinear')
        grid = grid.reshape(1, self.embed_dim, self.num_patches_a).transpose(1, 2)
        pretrain_dict['positional_embedding_audio'] = torch.cat([ori[:, :1, :].detach(), grid], dim=1).squeeze(dim=0)
        return pretrain_dict

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'class_embedding', 'positional_embedding', 'temporal_embedding', 'positional_embedding_audio',
                'temporal_embedding_audio'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table', 'temporal_position_bias_table'}

    def _plan(self):
        plan = getattr(self, "_plan_cache", None)
        if plan is not None:
            return plan

        class Plan:
            pass

        plan = Plan()
        plan.T = self.num_video_frames
        blocks = list(self.transformer.resblocks)
        plan.mods = blocks[0].spec((1, 1) if blocks[0].mode in ('fusion_adapt', 'multimodal_adapt_no_fusion') else (1,)).mods
        plan.head_drop = self.mlp_head[1].p if len(plan.mods) == 2 else 0.
        cache = {}

        def make(n_tok):
            if n_tok not in cache:
                out = []
                for i, b in enumerate(blocks):
                    spec = b.spec(n_tok)
                    out.append((spec, f"transformer.resblocks.{i}.", vit_block_param_names(spec)))
                cache[n_tok] = out
            return cache[n_tok]
        plan.blocks = make
        ok = ("Adapter", "gate_", "temporal_embedding", "ln_post", "mlp_head.")
        plan.trainable_ok = lambda n: any(t in n for t in ok)
        self._plan_cache = plan
        return plan

    def forward(self, a, v, mode):
        """a: [B, T, 128-ish, L] log-mel segments, v: [B, 3, T, H, W] frames (CLIP_AVE.py:979-1140)."""
        if mode not in ('audioonly', 'videoonly', 'multimodal', 'fusion') or mode != self.ftmode:
            raise TypeError('ftmode is not expected !!!')
        ref = v if mode != 'audioonly' else a
        if not ref.is_cuda:
            raise RuntimeError("stg-cma_amd runs on MI355X only: move the model and inputs to the GPU (no CPU fallback)")
        names, tensors = [], []
        for n, p in self.named_parameters():
            names.append(n)
            tensors.append(p)
        with torch.cuda.device(ref.device):     # launches go to the current device's stream: make that the tensors' device
            return VitModelFn.apply(a, v, self._plan(), self.training, torch.is_grad_enabled(), tuple(names), *tensors)
