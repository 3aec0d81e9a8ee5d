"""-m gpu, one process: with a GradSync attached, ONE backward of every model class moves every trainable gradient element through
the data-parallel exchange (backbone arena inside the autograd node + end-of-backward bucket of the task heads).  world_size is 1
here, so nothing is communicated; the accounting is what is checked -- SCALE runs on 8 GPUs exercise the same code path."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def pg():
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
        created = True
    yield
    if created:
        dist.destroy_process_group()


def _models(gpu):
    from stgcma.model import CLIP_AVE, Swin_AVE, Swin_AVQAModel_V1, Swin_AVSModel_Base
    T = 2
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(1, T, 224, 224, generator=g) * 0.5).to(gpu)
    v5 = torch.randn(1, 3, T, 224, 224, generator=g).to(gpu)
    vt = v5.permute(0, 2, 1, 3, 4).contiguous()
    yield "swin_ave", Swin_AVE.SwinTransformer2D_Adapter_New(label_dim=29, num_frames=T, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8],
                                                             ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625]), \
        lambda m: m(a, v5, "fusion").float().square().mean()
    a_clip = (torch.randn(1, T, 102, 128, generator=g) * 0.5).to(gpu)
    yield "clip_ave", CLIP_AVE.MM_CLIP_AVE(label_dim=29, num_video_frames=T, layers=2, heads=8, embed_dim=768, ftmode="fusion"), \
        lambda m: m(a_clip, v5, "fusion").float().square().mean()
    yield "swin_avs", Swin_AVSModel_Base.SwinTransformer2D_Adapter_AVS_Base(
        pretrained=None, num_frames=T, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], ftmode="fusion",
        adapter_mlp_ratio=[0.5, 0.5, 0.25, 0.25], channel=256, vis_dim=[64, 128, 320, 512], tpavi_stages=[0, 1, 2, 3],
        tpavi_vv_flag=False, tpavi_va_flag=True), \
        lambda m: m(a, vt, "fusion")[0].float().square().mean()
    q = torch.randint(0, 93, (1, 14), generator=g).to(gpu)
    yield "swin_avqa", Swin_AVQAModel_V1.SwinTransformer2D_Adapter_AVQA(
        pretrained=None, num_frames=T, embed_dim=192, depths=[2, 2, 2, 2], num_heads=[6, 12, 24, 48], ftmode="fusion",
        adapter_mlp_ratio=[0.25, 0.125, 0.125, 0.0625]), \
        lambda m: sum(o.float().square().mean() for o in m(a, vt, vt.flip(1), q, "fusion"))


def test_every_trainable_element_is_exchanged(stg, gpu, pg):
    from stgcma import ddp, recipe
    for tag, m, loss_of in _models(gpu):
        torch.manual_seed(0)
        m = m.to(gpu).train()
        with torch.no_grad():                                   # leave the zero initialisation of D_fc2 / gates
            for n, p in m.named_parameters():
                if "D_fc2" in n or "gate_" in n:
                    p.normal_(0.0, 0.05)
        recipe.apply_freeze(m)
        sync = ddp.attach(m)
        n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
        for step in range(2):                                   # twice: the per-backward arming must re-arm
            m.zero_grad(set_to_none=True)
            loss_of(m).backward()
            torch.cuda.synchronize()
            assert sync.last_numel == n_train, f"{tag} step {step}: exchanged {sync.last_numel} of {n_train} trainable elements"
        missing = [n for n, p in m.named_parameters() if p.requires_grad and p.grad is None]
        assert not missing, f"{tag}: no gradient for {missing[:5]}"


def test_attach_with_a_partly_frozen_task_head_then_unfreeze(stg, gpu, pg):
    """attach() used to hook every task-head parameter and crashed on a frozen one ('cannot register a hook on a tensor that
    doesn't require gradient'); a parameter unfrozen after attach() was never synchronised.  Frozen tensors are now recorded without a
    hook and picked up by rewatch() (run at every arena all-reduce)."""
    from stgcma import ddp, recipe
    tag, m, loss_of = [t for t in _models(gpu) if t[0] == "swin_avqa"][0]
    torch.manual_seed(0)
    m = m.to(gpu).train()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "D_fc2" in n or "gate_" in n:
                p.normal_(0.0, 0.05)
    recipe.apply_freeze(m)
    head = [(n, p) for n, p in m.named_parameters() if n.startswith("avqatask_") and p.requires_grad]
    probe = head[: len(head) // 2]                          # a linear-probe style ablation: half of the task head frozen before attach
    for _, p in probe:
        p.requires_grad_(False)
    sync = ddp.attach(m)                                    # must not raise
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    m.zero_grad(set_to_none=True)
    loss_of(m).backward()
    torch.cuda.synchronize()
    assert sync.last_numel == n_train, f"exchanged {sync.last_numel} of {n_train} trainable elements with half the head frozen"
    assert all(p.grad is None for _, p in probe)
    for _, p in probe:                                      # unfreeze: hooked at the next backward's arena all-reduce (or by hand)
        p.requires_grad_(True)
    sync.rewatch()
    n_all = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n_all > n_train
    m.zero_grad(set_to_none=True)
    loss_of(m).backward()
    torch.cuda.synchronize()
    assert sync.last_numel == n_all, f"exchanged {sync.last_numel} of {n_all} trainable elements after the unfreeze"
