"""Child process of tests/test_ddp2_gpu.py: one data-parallel rank running the HIP forward / backward of small Swin (AVE), AVS and
AVQA models on cuda:0 under a 2-rank gloo group (RCCL refuses two ranks on one device; the collective is torch.distributed either
way, ddp.GradSync).  usage: ddp2_worker.py <rank> <world> <port> <out_dir>
Per model it saves (a) the gradients of its OWN clip without any exchange and (b) the gradients after a backward with ddp.attach."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")

import torch  # noqa: E402
import stgcma  # noqa: F401,E402
from stgcma import ddp, recipe  # noqa: E402
from stgcma.model import Swin_AVE, Swin_AVQAModel_V1, Swin_AVSModel_Base  # noqa: E402

gpu = torch.device("cuda", 0)
torch.cuda.set_device(0)
r, _, w = ddp.init_from_env("gloo")
assert (r, w) == (rank, world)
T = 2


def models():
    g = torch.Generator().manual_seed(100 + rank)            # every rank its own clip
    a = (torch.randn(1, T, 224, 224, generator=g) * 0.5).to(gpu)
    v5 = torch.randn(1, 3, T, 224, 224, generator=g).to(gpu)
    vt = v5.permute(0, 2, 1, 3, 4).contiguous()
    yield "swin_ave", Swin_AVE.SwinTransformer2D_Adapter_New(label_dim=29, num_frames=T, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8],
                                                             ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625]), \
        lambda m: m(a, v5, "fusion").float().square().mean()
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


for tag, m, loss_of in models():
    torch.manual_seed(0)                                      # same parameters on every rank
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "D_fc2" in n or "gate_" in n:
                p.normal_(0.0, 0.05)
    m = m.to(gpu).eval()                                      # eval: no DropPath / Dropout draws, the two passes see the same function
    recipe.apply_freeze(m)
    ddp.broadcast_parameters(m)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    d = dict(m.named_parameters())
    loss_of(m).backward()
    # a few task-head tensors receive no gradient at all (FeatureFusionBlock's resConfUnit1 is unused behind a single input, like in
    # the reference): zeros locally, and zeros are what they contribute to the exchanged bucket
    local = {n: (d[n].grad.detach().float().cpu().clone() if d[n].grad is not None else torch.zeros(d[n].shape)) for n in names}
    assert sum(d[n].grad is None for n in names) <= 8, tag
    m.zero_grad(set_to_none=True)
    sync = ddp.attach(m)
    loss_of(m).backward()
    torch.cuda.synchronize()
    n_train = sum(d[n].numel() for n in names)
    assert sync.last_numel == n_train, f"{tag}: exchanged {sync.last_numel} of {n_train}"
    synced = {n: (d[n].grad.detach().float().cpu().clone() if d[n].grad is not None else torch.zeros(d[n].shape)) for n in names}
    torch.save({"local": local, "synced": synced}, os.path.join(out_dir, f"{tag}_r{rank}.pt"))
    del m, d, sync
    torch.cuda.empty_cache()

torch.distributed.barrier()
torch.distributed.destroy_process_group()
print(f"rank {rank} done", flush=True)
