"""MI355X-native drop-in for the reference's AVE/model/Swin_AVE.py: same class names, constructor keywords, forward
signature, state_dict keys and parameter names (the train loop partitions parameters by NAME substring,
AVE/traintest_adapt_ave29.py:38-55), so AVE/run_adapt_ave29.py / traintest_adapt_ave29.py drive it unchanged.

The nn.Modules below are parameter containers + orchestration only.  All arithmetic runs in libstgcma_hip.so through
..ops (block-level autograd Functions over the fused audio+video token tensor); there is no eager / CPU fallback and
forward raises if the extension is missing or the tensors are not on a GPU.

Reference bugs handled deliberately: ftmode='audioonly' dereferences a non-existent
`self.layers_audio` (Swin_AVE.py:1521) -- here it works; `temporal_embedding_audio` is never initialised when
t_relative=False (:1211-1212) -- here it is.
"""
import contextlib

import torch
import torch.nn as nn

from .. import ops
from ..ops import BlockSpec, SwinBlockFn, SwinModelFn, block_buffer_names, block_param_names
from ._common import DropPath, to_2tuple, trunc_normal_

BF16 = torch.bfloat16


class Adapter(nn.Module):
    """Bottleneck D_fc1 -> GELU -> D_fc2 (Swin_AVE.py:10-24); SAdapter2 (:27-41) and T_Adapter (:44-58) share the layout."""

    def __init__(self, D_features, mlp_ratio=0.25, act_layer=nn.GELU):
        super().__init__()
        D_hidden_features = int(D_features * mlp_ratio)
        self.act = act_layer()
        self.D_fc1 = nn.Linear(D_features, D_hidden_features)
        self.D_fc2 = nn.Linear(D_hidden_features, D_features)


class SAdapter2(Adapter):
    pass


class T_Adapter(Adapter):
    pass


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 parameter holder (Swin_AVE.py:111-127)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if drop != 0.:
            raise NotImplementedError("Mlp dropout > 0 is not part of the reference recipe (drop_rate=0)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class WindowAttention(nn.Module):
    """Parameters / index buffers of W-MSA with the extra temporal bias tables (Swin_AVE.py:176-229)."""

    def __init__(self, dim, num_ttokens, window_size, num_heads, use_temporal=True, qkv_bias=True, qk_scale=None,
                 attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("attention / projection dropout > 0 is not part of the reference recipe")
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not supported")
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        ch, cw = torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing="ij")
        ch, cw = ch.reshape(-1), cw.reshape(-1)
        rel = (ch[:, None] - ch[None, :] + Wh - 1) * (2 * Ww - 1) + (cw[:, None] - cw[None, :] + Ww - 1)
        self.register_buffer("relative_position_index", rel)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)
        if use_temporal:
            self.num_ttokens = num_ttokens
            t = torch.arange(num_ttokens)
            coords = (t[:, None] - t[None, :] + num_ttokens - 1).reshape(-1)
            self.temporal_position_bias_table = nn.Parameter(torch.zeros(2 * num_ttokens - 1, num_heads))
            trunc_normal_(self.temporal_position_bias_table, std=.02)
            self.register_buffer("t_relative_coords", coords.clone())
            self.temporal_position_bias_table_audio = nn.Parameter(torch.zeros(2 * num_ttokens - 1, num_heads))
            trunc_normal_(self.temporal_position_bias_table_audio, std=.02)
            self.register_buffer("t_relative_coords_a", coords.clone())

    def extra_repr(self):
        return f'dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}'


class SwinTransformerBlock(nn.Module):
    """Swin block with STG-CMA adapters (Swin_AVE.py:298-813).  forward takes / returns the fused token tensor."""

    def __init__(self, dim, input_resolution, num_frames, num_heads, window_size=7, shift_size=0, mlp_ratio=4., t_attn=False,
                 qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, adapter_mlp_ratio=0.25, mode='video_adapt'):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        self.num_frames, self.mode = num_frames, mode
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        video = mode in ('video_adapt', 'multimodal_adapt_no_fusion', 'fusion_adapt')
        audio = mode in ('audio_adapt', 'multimodal_adapt_no_fusion', 'fusion_adapt')
        self.t_attn = t_attn
        if t_attn:
            if video:
                self.T_Adapter = T_Adapter(D_features=dim, mlp_ratio=adapter_mlp_ratio)
            if audio:
                self.T_Adapter_Audio = T_Adapter(D_features=dim, mlp_ratio=adapter_mlp_ratio)
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, num_ttokens=num_frames, window_size=to_2tuple(self.window_size), num_heads=num_heads,
                                    use_temporal=t_attn, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                                    proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if video:
            self.S_Adapter = Adapter(dim, mlp_ratio=adapter_mlp_ratio)
            self.S_Adapter2 = SAdapter2(dim, mlp_ratio=adapter_mlp_ratio)
        if audio:
            self.S_Adapter_Audio = Adapter(dim, mlp_ratio=adapter_mlp_ratio)
            self.S_Adapter2_Audio = SAdapter2(dim, mlp_ratio=adapter_mlp_ratio)
        self.gate_v = nn.Parameter(torch.zeros(1))
        self.gate_a = nn.Parameter(torch.zeros(1))
        if self.shift_size > 0:
            H, W = self.input_resolution
            attn_mask = ops.shift_mask(H, W, self.window_size, self.shift_size)
        else:
            attn_mask = None
        self.register_buffer("attn_mask", attn_mask)
        H, W = self.input_resolution
        self._spec = BlockSpec(dim, H, W, num_frames, num_heads, self.window_size, self.shift_size, t_attn, mode=mode,
                               drop_path=float(drop_path))

    def tensor_names(self):
        return block_param_names(self._spec) + block_buffer_names(self._spec)

    def forward(self, X):
        """X: fused token tensor [M*BT*N, C] (video rows first, then audio, for the two-stream modes; one modality for
        'video_adapt' / 'audio_adapt').  fp32 (the residual-stream dtype) or bf16."""
        names = self.tensor_names()
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        with torch.cuda.device(X.device) if X.is_cuda else contextlib.nullcontext():
            return SwinBlockFn.apply(X, self._spec, tuple(names), self.training, torch.is_grad_enabled(), *[sd[n] for n in names])

    def extra_repr(self):
        return f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, " \
               f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}"


class PatchMerging(nn.Module):
    """2x2 merge + LayerNorm(4C) + Linear(4C, 2C, bias=False) (Swin_AVE.py:944-981) on every frame of the fused tensor."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def extra_repr(self):
        return f"input_resolution={self.input_resolution}, dim={self.dim}"


class BasicLayer(nn.Module):
    """One Swin stage: even blocks = temporal + unshifted window attention, odd = shifted (Swin_AVE.py:994-1064)."""

    def __init__(self, dim, input_resolution, num_frames, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None,
                 use_checkpoint=False, adapter_mlp_ratio=0.25, mode='video_adapt'):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.use_checkpoint, self.mode = use_checkpoint, mode
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_frames=num_frames, num_heads=num_heads,
                                 window_size=window_size, t_attn=(i % 2 == 0), shift_size=0 if (i % 2 == 0) else window_size // 2,
                                 mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer,
                                 adapter_mlp_ratio=adapter_mlp_ratio, mode=mode)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def extra_repr(self):
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"


class PatchEmbed3D(nn.Module):
    """Conv3d(k = s = patch) + LayerNorm parameter holder (Swin_AVE.py:1078-1124)."""

    def __init__(self, img_size=224, patch_size=(2, 4, 4), in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = patch_size
        img_size = to_2tuple(img_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.patches_resolution = [img_size[0] // patch_size[1], img_size[1] // patch_size[2]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None


class SwinTransformer2D_Adapter_New(nn.Module):
    """Swin-B/L + STG-CMA for AVE (Swin_AVE.py:1129-1599).  forward(a, v, mode) -> logits [(B*T), label_dim] (fp32)."""

    def __init__(self, label_dim, pretrained=None, img_size=224, patch_size=[1, 4, 4], num_frames=10, in_chans=3,
                 embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, mlp_ratio=4.,
                 frozen_stages=-1, qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True, t_relative=True, use_checkpoint=False,
                 ftmode='videoonly', adapter_mlp_ratio=[0.25, 0.25, 0.25, 0.25], **kwargs):
        super().__init__()
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.patch_norm = embed_dim, ape, patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio, self.pretrained, self.num_frames = mlp_ratio, pretrained, num_frames
        self.frozen_stages, self.patch_size, self.t_relative, self.ftmode = frozen_stages, patch_size, t_relative, ftmode
        if drop_rate != 0.:
            raise NotImplementedError("drop_rate > 0 is not part of the reference recipe")
        self.patch_embed = PatchEmbed3D(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                        norm_layer=norm_layer if self.patch_norm else None)
        patches_resolution = self.patch_embed.patches_resolution
        self.patches_resolution = patches_resolution
        self.num_Ttokens = num_frames // patch_size[0]
        self.patch_embed_audio = PatchEmbed3D(img_size=img_size, patch_size=patch_size, in_chans=1, embed_dim=embed_dim,
                                              norm_layer=norm_layer if self.patch_norm else None)
        self.f_dim, self.t_dim = self.patch_embed_audio.patches_resolution
        self.num_patches_audio = self.f_dim * self.t_dim
        self.patches_resolution_audio = [self.f_dim, self.t_dim]
        self.num_Ttokens_audio = num_frames // patch_size[0]
        if not self.t_relative:
            self.temporal_embedding = nn.Parameter(torch.zeros(1, self.num_Ttokens, embed_dim))
            trunc_normal_(self.temporal_embedding, std=.02)
            self.temporal_embedding_audio = nn.Parameter(torch.zeros(1, self.num_Ttokens_audio, embed_dim))
            trunc_normal_(self.temporal_embedding_audio, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        block_mode = {'videoonly': 'video_adapt', 'audioonly': 'audio_adapt', 'multimodal': 'multimodal_adapt_no_fusion',
                      'fusion': 'fusion_adapt'}.get(ftmode)
        if block_mode is None:
            raise TypeError('ftmode is not expected !!!')
        self.layers = nn.ModuleList()
        for i_layer in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i_layer),
                input_resolution=(patches_resolution[0] // (2 ** i_layer), patches_resolution[1] // (2 ** i_layer)),
                num_frames=self.num_Ttokens, depth=depths[i_layer], num_heads=num_heads[i_layer], window_size=window_size,
                mlp_ratio=self.mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i_layer]):sum(depths[:i_layer + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if (i_layer < self.num_layers - 1) else None, use_checkpoint=use_checkpoint,
                adapter_mlp_ratio=adapter_mlp_ratio[i_layer], mode=block_mode))
        self.norm = norm_layer(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        if self.ftmode in ('multimodal', 'fusion'):
            self.mlp_head = nn.Sequential(nn.Linear(self.num_features * 2, 512), nn.Dropout(0.5), nn.Linear(512, label_dim))
        else:
            self.mlp_head = nn.Sequential(nn.LayerNorm(self.num_features), nn.Linear(self.num_features, label_dim))
        self.initialize_weights(pretrained=self.pretrained)
        self._freeze_stages()
        from .. import fp8 as _fp8
        if _fp8.env_default():                  # STG_FP8=1: the opt-in fp8 frozen-weight path without touching the runner
            _fp8.enable(self)

    # ------------------------------------------------------------------ init / checkpoint ingestion
    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for param in self.patch_embed.parameters():
                param.requires_grad = False
        if self.frozen_stages >= 1:
            self.pos_drop.eval()
            for i in range(0, self.frozen_stages):
                m = self.layers[i]
                m.eval()
                for param in m.parameters():
                    param.requires_grad = False

    def initialize_weights(self, pretrained=None):
        """trunc_normal(.02) Linear / unit LayerNorm init, optional Swin checkpoint ingestion with patch-embed inflation and
        audio patch-embed = channel mean (Swin_AVE.py:1353-1419), then zero every adapter D_fc2 (:1422-1468)."""
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
            checkpoint = torch.load(self.pretrained, map_location='cpu')
            state_dict = checkpoint['model']
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
        for n, m in self.layers.named_modules():
            if isinstance(m, Adapter):          # S_Adapter*, S_Adapter2*, T_Adapter*
                nn.init.constant_(m.D_fc2.weight, 0)
                nn.init.constant_(m.D_fc2.bias, 0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed', 'temporal_embedding', 'temporal_embedding_audio'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table', 'temporal_position_bias_table'}

    # ------------------------------------------------------------------ forward
    def _plan(self):
        """Static launch plan of the whole network for ops.SwinModelFn (rebuilt lazily; cheap)."""
        plan = getattr(self, "_plan_cache", None)
        if plan is not None:
            return plan

        class Plan:
            pass

        plan = Plan()
        first = self.layers[0].blocks[0]._spec
        plan.mods = first.mods
        plan.n_patches = self.patch_embed.num_patches
        plan.embed_dim = self.embed_dim
        plan.stages = []
        for s_i, layer in enumerate(self.layers):
            st = {"blocks": [], "names": {}, "merge": None}
            for b_i, blk in enumerate(layer.blocks):
                pre = f"layers.{s_i}.blocks.{b_i}."
                st["blocks"].append((blk._spec, pre))
                st["names"][pre] = blk.tensor_names()
            if layer.downsample is not None:
                H, W = layer.input_resolution
                st["merge"] = (H, W, f"layers.{s_i}.downsample.")
            plan.stages.append(st)
        last = self.layers[-1].input_resolution
        plan.n_tok_last = last[0] * last[1]
        plan.head_drop = self.mlp_head[1].p if (len(plan.mods) == 2 and hasattr(self, 'mlp_head')) else 0.
        train_ok = ("Adapter", "gate_", "temporal_position_bias_table", "mlp_head.", "temporal_embedding")
        plan.trainable_ok = lambda n: any(t in n for t in train_ok)
        self._plan_cache = plan
        return plan

    def _flat_tensors(self):
        names, tensors = [], []
        for n, p in self.named_parameters():
            if n.startswith(("avqatask_", "avstask_")):      # task heads of the AVQA / AVS mirrors: not backbone tensors
                continue
            names.append(n)
            tensors.append(p)
        for n, b in self.named_buffers():
            if not n.endswith("attn_mask"):
                names.append(n)
                tensors.append(b)
        return tuple(names), tensors

    def _backbone(self, a, v, v_nega=None, taps=False):
        """Shared by the AVS / AVQA mirrors (model/Swin_AVSModel.py, model/Swin_AVQAModel_V1.py): ops.SwinBackboneFn on (B, T, 3, H, W)
        clips; returns the flat fp32 feature tensors."""
        from ..ops import SwinBackboneFn
        if self.ftmode != 'fusion':
            raise TypeError('ftmode is not expected !!!')
        if not v.is_cuda:
            raise RuntimeError("stg-cma_amd runs on MI355X only: move the model and inputs to the GPU (no CPU fallback)")
        names, tensors = self._flat_tensors()
        with torch.cuda.device(v.device):       # launches go to the current device's stream: make that the tensors' device
            return SwinBackboneFn.apply(a, v, v_nega, self._plan(), self.training, torch.is_grad_enabled(), bool(taps), names, *tensors)

    def forward(self, a, v, mode):
        """a: [B, T, H, W] spectrogram segments, v: [B, 3, T, H, W] frames -> fp32 logits [(B*T), label_dim]
        (Swin_AVE.py:1479-1599).  Runs entirely on the HIP path; raises when inputs / parameters are not on a GPU."""
        # `mode` picks the reference's forward path (:1479-1599), the blocks were built for `ftmode` (:1220-1309).  The 'multimodal'
        # and 'fusion' paths are the same code over a (v, a) tuple, so either string drives a two-stream model; a single-stream
        # path through two-stream blocks (or the reverse) crashes in the reference, and an unknown string makes it return None:
        # both raise here.
        two = ('multimodal', 'fusion')
        if mode not in ('audioonly', 'videoonly') + two or (mode != self.ftmode and not (mode in two and self.ftmode in two)):
            raise TypeError('ftmode is not expected !!!')
        ref = v if mode != 'audioonly' else a
        if not ref.is_cuda:
            raise RuntimeError("stg-cma_amd runs on MI355X only: move the model and inputs to the GPU (no CPU fallback)")
        # use_checkpoint=True (Swin_AVE.py:1049-1050) is accepted and changes nothing: checkpointing trades memory for recompute and
        # leaves outputs and gradients as they are; this node keeps its activations (80 GB at B = 32 of 288 GB).
        names, tensors = self._all_tensors()
        with torch.cuda.device(ref.device):     # launches go to the current device's stream: make that the tensors' device
            return SwinModelFn.apply(a, v, self._plan(), self.training, torch.is_grad_enabled(), names, *tensors)

    def _all_tensors(self):
        """(names, tensors) of every parameter and buffer the whole-model node takes, in a fixed order.  The NAME walk is cached
        (it cost a named_parameters() traversal of 946 entries per call); the tensors are re-read from the modules' own
        dictionaries on every call, so .to() / load_state_dict / requires_grad_ / DataParallel replicas are always seen."""
        cache = self.__dict__.get("_tensor_walk")
        if cache is not None and cache[2] != id(self):       # a DataParallel replica copies __dict__: its walk must name ITS submodules
            cache = None
        if cache is None:
            mods = dict(self.named_modules())
            cache = []
            for n, _ in self.named_parameters():
                owner, _, leaf = n.rpartition(".")
                cache.append((n, mods[owner], leaf, True))
            for n, _ in self.named_buffers():
                if not n.endswith("attn_mask"):
                    owner, _, leaf = n.rpartition(".")
                    cache.append((n, mods[owner], leaf, False))
            self.__dict__["_tensor_walk"] = cache = (tuple(c[0] for c in cache), cache, id(self))
        names, walk = cache[0], cache[1]
        return names, [(m._parameters if is_p else m._buffers)[leaf] for _, m, leaf, is_p in walk]
