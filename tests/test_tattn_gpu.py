"""-m gpu: the temporal-attention kernels (stg_tattn_fwd / stg_tattn_bwd: forward, and dQ / dK / dV / dbias in one backward
kernel that recomputes the softmax) against an fp32 PyTorch-CPU statement of the reference's temporal branch
(WindowAttention.forward, Swin_AVE.py:244-255, on the '(b t) n c -> (b n) t c' layout of :705,711)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _close(got, ref, tol=1e-2, what=""):
    got = got.detach().float().cpu(); ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    err = (got - ref).abs(); bound = tol * torch.clamp(ref.abs(), min=1.0)
    bad = err > bound
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} off; first {idx}: got {got[tuple(idx)].item()} "
                             f"ref {ref[tuple(idx)].item()}; max err {err.max().item():.4g}")


def _ref(QKV, bias, nm, B, T, N, H, scale):
    """QKV fp32 [rows, 3*H*32] leaf; bias fp32 [nm, H, T, T] leaf -> O [rows, H*32]."""
    C = H * 32
    x = QKV.view(nm, B, T, N, 3, H, 32).permute(4, 0, 1, 3, 5, 2, 6)      # [3, nm, B, N, H, T, 32]
    q, k, v = x[0], x[1], x[2]
    s = scale * (q @ k.transpose(-1, -2)) + bias[:, None, None]           # [nm, B, N, H, T, T]
    o = torch.softmax(s, -1) @ v                                          # [nm, B, N, H, T, 32]
    return o.permute(0, 1, 4, 2, 3, 5).reshape(nm * B * T * N, C)


def _run(gpu, nm, B, T, N, H, seed=0, mag=1.0, want_dbias=True, pad_cols=0):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(seed)
    C = H * 32
    rows = nm * B * T * N
    QKVb = (torch.randn(rows, 3 * C + pad_cols, generator=g) * mag).to(BF16)
    bias = torch.randn(nm, H, T, T, generator=g) * 0.5
    dOb = torch.randn(rows, C, generator=g).to(BF16)
    Qf = QKVb[:, :3 * C].float().requires_grad_(True)
    bf = bias.clone().requires_grad_(True)
    scale = 32 ** -0.5
    o_ref = _ref(Qf, bf, nm, B, T, N, H, scale)
    o_ref.backward(dOb.float())
    dev = gpu
    QKV = QKVb.to(dev)
    tg = k.TGeom(nm, B, T, N, H, scale, bias.reshape(nm, H, T * T).contiguous().to(dev))
    O = k.tattn_fwd(tg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:3 * C])
    _close(O, o_ref, what="O")
    dQKV = torch.full((rows, 3 * C + pad_cols), float("nan"), dtype=BF16, device=dev)
    dbias = torch.zeros(nm, H, T * T, dtype=F32, device=dev) if want_dbias else None
    k.tattn_bwd(tg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:3 * C], dOb.to(dev),
                dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:3 * C], dbias=dbias)
    gscale = float(Qf.grad.abs().max())
    _close(dQKV[:, :3 * C] / gscale, Qf.grad / gscale, tol=1.5e-2, what="dQKV")
    if want_dbias:
        bs = max(float(bf.grad.abs().max()), 1e-3)       # T == 1: the softmax is constant, the gradient exactly 0
        _close(dbias.view(nm, H, T, T) / bs, bf.grad / bs, tol=1.5e-2, what="dbias")
    if pad_cols:
        assert torch.isnan(dQKV[:, 3 * C:].float()).all(), "wrote outside the qkv columns"


def test_tattn_ave_T10_two_modalities(stg, gpu):
    _run(gpu, nm=2, B=2, T=10, N=49, H=4)            # 49 = 3*16 + 1: a partial last group


def test_tattn_avs_T5(stg, gpu):
    _run(gpu, nm=2, B=1, T=5, N=196, H=8, seed=1)    # 6 sequences per tile, 196 = 6*32 + 4


def test_tattn_heads_not_multiple_of_4(stg, gpu):
    _run(gpu, nm=1, B=2, T=10, N=16, H=6, seed=2)    # Swin-L stage 0: 6 heads


def test_tattn_T16_and_T32_full_tiles(stg, gpu):
    _run(gpu, nm=1, B=1, T=16, N=9, H=4, seed=3)
    _run(gpu, nm=1, B=1, T=32, N=5, H=4, seed=4)


def test_tattn_T1_and_T7(stg, gpu):
    _run(gpu, nm=2, B=1, T=1, N=40, H=4, seed=5)
    _run(gpu, nm=2, B=3, T=7, N=10, H=4, seed=6)


def test_tattn_large_scores_and_padded_buffer(stg, gpu):
    _run(gpu, nm=2, B=1, T=10, N=7, H=4, seed=7, mag=6.0, pad_cols=8)


def test_tattn_many_groups_grid_stride(stg, gpu):
    _run(gpu, nm=2, B=4, T=10, N=3136, H=4, seed=8)  # stage-0 geometry: more groups than the grid has workgroups


def test_tattn_no_dbias(stg, gpu):
    _run(gpu, nm=1, B=1, T=10, N=12, H=4, seed=9, want_dbias=False)


def test_tattn_rejects_bad_geometry(stg, gpu):
    from stgcma import kernels as k
    with pytest.raises(RuntimeError):
        k.TGeom(1, 1, 33, 4, 4, 1.0, torch.zeros(1, 4, 33 * 33, device=gpu))
    tg = k.TGeom(1, 1, 10, 4, 4, 1.0, torch.zeros(1, 4, 100, device=gpu))
    small = torch.zeros(8, 3 * 128, dtype=BF16, device=gpu)
    with pytest.raises(RuntimeError):
        k.tattn_fwd(tg, small[:, :128], small[:, 128:256], small[:, 256:])


def _ref_nb(QKV, B, T, N, H, D, scale):
    x = QKV.view(B, T, N, 3, H, D).permute(3, 0, 2, 4, 1, 5)              # [3, B, N, H, T, D]
    s = scale * (x[0] @ x[1].transpose(-1, -2))
    o = torch.softmax(s, -1) @ x[2]
    return o.permute(0, 3, 1, 2, 4).reshape(B * T * N, H * D)


@pytest.mark.parametrize("D,H,T,N", [(96, 8, 10, 197), (96, 8, 10, 49), (64, 16, 10, 257), (64, 3, 5, 20), (96, 2, 16, 9), (96, 5, 7, 11)])
def test_tattn_wide_heads_no_bias_vit(stg, gpu, D, H, T, N):
    """The temporal nn.MultiheadAttention of the CLIP ViT blocks (CLIP_AVE.py:369-377): head dim 96 / 64, no bias."""
    from stgcma import kernels as k
    B = 2
    g = torch.Generator().manual_seed(D + H + T + N)
    C = H * D
    rows = B * T * N
    QKVb = torch.randn(rows, 3 * C, generator=g).to(BF16)
    dOb = torch.randn(rows, C, generator=g).to(BF16)
    Xf = QKVb.float().requires_grad_(True)
    scale = D ** -0.5
    o_ref = _ref_nb(Xf, B, T, N, H, D, scale)
    o_ref.backward(dOb.float())
    QKV = QKVb.to(gpu)
    tg = k.TGeom(1, B, T, N, H, scale, None, D=D)
    O = k.tattn_fwd(tg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:])
    _close(O, o_ref, what="O")
    dQKV = torch.full((rows, 3 * C), float("nan"), dtype=BF16, device=gpu)
    k.tattn_bwd(tg, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:], dOb.to(gpu), dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:])
    gs = float(Xf.grad.abs().max())
    _close(dQKV / gs, Xf.grad / gs, tol=1.5e-2, what="dQKV")
    with pytest.raises(RuntimeError):
        k.TGeom(1, B, T, N, H, scale, torch.zeros(1, H, T * T, device=gpu), D=D)      # wide heads take no bias
