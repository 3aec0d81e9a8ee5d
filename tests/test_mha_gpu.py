"""-m gpu: the ViT multi-head self-attention kernels (stg_mha_fwd / stg_mha_bwd) against an fp32 PyTorch-CPU statement of
nn.MultiheadAttention's core, softmax(q k^T / sqrt(d)) v per (frame, head) (CLIP_AVE.py:106-108)."""
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


def _run(gpu, P, H, n, D, seed=0, mag=1.0, pad_cols=0):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(seed)
    C = H * D
    rows = P * n
    QKVb = (torch.randn(rows, 3 * C + pad_cols, generator=g) * mag).to(BF16)
    dOb = torch.randn(rows, C, generator=g).to(BF16)
    X = QKVb[:, :3 * C].float().requires_grad_(True)
    x = X.view(P, n, 3, H, D).permute(2, 0, 3, 1, 4)                     # [3, P, H, n, D]
    scale = D ** -0.5
    s = scale * (x[0] @ x[1].transpose(-1, -2))
    o_ref = (torch.softmax(s, -1) @ x[2]).permute(0, 2, 1, 3).reshape(rows, C)
    o_ref.backward(dOb.float())
    QKV = QKVb.to(gpu)
    geo = k.MhaGeom(P, H, n, D, scale)
    O, lse = k.mha_fwd(geo, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:3 * C])
    _close(O / mag, o_ref / mag, what="O")          # P is rounded to bf16 before P V: the error scales with |v|
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634               # the kernels keep the LSE in the log2 domain
    _close(lse, lse_ref.detach(), tol=2e-2, what="lse")
    dQKV = torch.full((rows, 3 * C + pad_cols), float("nan"), dtype=BF16, device=gpu)
    k.mha_bwd(geo, QKV[:, :C], QKV[:, C:2 * C], QKV[:, 2 * C:3 * C], O, lse, dOb.to(gpu),
              dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:3 * C])
    gs = float(X.grad.abs().max())
    _close(dQKV[:, :3 * C] / gs, X.grad / gs, tol=1.5e-2, what="dQKV")
    if pad_cols:
        assert torch.isnan(dQKV[:, 3 * C:].float()).all(), "wrote outside the qkv columns"


def test_mha_vit_b_video_197_d96(stg, gpu):
    _run(gpu, P=4, H=8, n=197, D=96)


def test_mha_vit_b_audio_49_d96(stg, gpu):
    _run(gpu, P=6, H=8, n=49, D=96, seed=1)


def test_mha_vit_l_257_d64(stg, gpu):
    _run(gpu, P=2, H=16, n=257, D=64, seed=2)


@pytest.mark.parametrize("n", [1, 13, 32, 33, 128, 129])
def test_mha_ragged_token_counts(stg, gpu, n):
    _run(gpu, P=3, H=2, n=n, D=96, seed=10 + n)
    _run(gpu, P=2, H=3, n=n, D=64, seed=20 + n)


def test_mha_large_scores_and_padded_buffer(stg, gpu):
    _run(gpu, P=2, H=4, n=50, D=96, seed=3, mag=5.0, pad_cols=8)


def test_mha_rejects_bad_geometry(stg, gpu):
    from stgcma import kernels as k
    with pytest.raises(RuntimeError):
        k.MhaGeom(1, 1, 10, 48, 1.0)
    geo = k.MhaGeom(2, 2, 10, 64, 0.125)
    small = torch.zeros(8, 3 * 128, dtype=BF16, device=gpu)
    with pytest.raises(RuntimeError):
        k.mha_fwd(geo, small[:, :128], small[:, 128:256], small[:, 256:])


@pytest.mark.parametrize("D,n", [(96, 196), (64, 49), (96, 3136)])
def test_mha_cross_modal_shared_kv(stg, gpu, D, n):
    """Frame-global cross-modal attention of a wide adapter: r = softmax(h_q h_kv^T) h_kv, no scale, one head, K and V the SAME
    tensor (Swin_AVE.py:801-805); dK of the kernel = the whole gradient of h_kv."""
    from stgcma import kernels as k
    P = 2 if n > 1000 else 3
    g = torch.Generator().manual_seed(n + D)
    hq_b = (torch.randn(P * n, D, generator=g) * 0.35).to(BF16)
    hk_b = (torch.randn(P * n, D, generator=g) * 0.35).to(BF16)
    dO_b = torch.randn(P * n, D, generator=g).to(BF16)
    hq = hq_b.float().requires_grad_(True)
    hk = hk_b.float().requires_grad_(True)
    s = hq.view(P, n, D) @ hk.view(P, n, D).transpose(1, 2)
    r_ref = (torch.softmax(s, -1) @ hk.view(P, n, D)).reshape(P * n, D)
    r_ref.backward(dO_b.float())
    geo = k.MhaGeom(P, 1, n, D, 1.0)
    q, kv = hq_b.to(gpu), hk_b.to(gpu)
    r, lse = k.mha_fwd(geo, q, kv, kv)
    _close(r, r_ref, what="r")
    dq = torch.full_like(q, float("nan")); dkv = torch.full_like(kv, float("nan"))
    k.mha_bwd(geo, q, kv, kv, r, lse, dO_b.to(gpu), dQ=dq, dK=dkv, dV=None)
    gs = float(max(hq.grad.abs().max(), hk.grad.abs().max()))
    _close(dq / gs, hq.grad / gs, tol=1.5e-2, what="dq")
    _close(dkv / gs, hk.grad / gs, tol=1.5e-2, what="dkv")


def _window_index(F, Hi, Wi, ws, shift):
    """rows of the [F * Hi * Wi] token tensor in window order: [F * nW, ws * ws] (torch.roll by -shift, then window_partition:
    Swin_AVE.py:151-165 / :262-266 -- window (wi, wj) token (ti, tj) sits at ((wi ws + ti + shift) % Hi, (wj ws + tj + shift) % Wi))."""
    f, wi, wj, ti, tj = torch.meshgrid(torch.arange(F), torch.arange(Hi // ws), torch.arange(Wi // ws), torch.arange(ws), torch.arange(ws),
                                       indexing="ij")
    y = (wi * ws + ti + shift) % Hi
    x = (wj * ws + tj + shift) % Wi
    return (f * Hi * Wi + y * Wi + x).reshape(F * (Hi // ws) * (Wi // ws), ws * ws)


@pytest.mark.parametrize("D,F,Hi,Wi,ws,shift", [(96, 3, 14, 14, 7, 0), (96, 2, 28, 14, 7, 3), (64, 2, 7, 7, 7, 0), (64, 1, 12, 8, 4, 2),
                                                (96, 1, 24, 24, 8, 4)])
def test_mha_window_map_cross_modal(stg, gpu, D, F, Hi, Wi, ws, shift):
    """The WINDOW-level cross-modal attention of a wide adapter (Swin-L, d_h = 96) on the flash kernels: problem p = window p % nW of
    frame p / nW, tokens addressed in place (cyclic shift included), K == V.  Reference: gather the windows, attend, scatter back.
    Rows no window writes do not exist (the windows tile the image), so O / dQ / dK are checked whole."""
    from stgcma import kernels as k
    n, N = ws * ws, Hi * Wi
    idx = _window_index(F, Hi, Wi, ws, shift)                                 # [P, n]
    P = idx.shape[0]
    g = torch.Generator().manual_seed(D + F * 7 + Hi + shift)
    hq_b = (torch.randn(F * N, D, generator=g) * 0.4).to(BF16)
    hk_b = (torch.randn(F * N, D, generator=g) * 0.4).to(BF16)
    dO_b = torch.randn(F * N, D, generator=g).to(BF16)
    hq = hq_b.float().requires_grad_(True)
    hk = hk_b.float().requires_grad_(True)
    qw, kw = hq[idx.reshape(-1)].view(P, n, D), hk[idx.reshape(-1)].view(P, n, D)
    s = qw @ kw.transpose(1, 2)
    rw = torch.softmax(s, -1) @ kw
    r_ref = torch.zeros(F * N, D).index_add(0, idx.reshape(-1), rw.reshape(P * n, D))
    r_ref.backward(dO_b.float())
    geo = k.MhaGeom(P, 1, n, D, 1.0, window=(Hi, Wi, ws, shift))
    q, kv = hq_b.to(gpu), hk_b.to(gpu)
    r, lse = k.mha_fwd(geo, q, kv, kv)
    _close(r, r_ref, what="r")
    _close(lse, (torch.logsumexp(s, -1) * 1.4426950408889634).detach().view(P, 1, n), tol=2e-2, what="lse")
    dq = torch.full_like(q, float("nan")); dkv = torch.full_like(kv, float("nan"))
    k.mha_bwd(geo, q, kv, kv, r, lse, dO_b.to(gpu), dQ=dq, dK=dkv, dV=None)
    gs = float(max(hq.grad.abs().max(), hk.grad.abs().max()))
    _close(dq / gs, hq.grad / gs, tol=1.5e-2, what="dq")
    _close(dkv / gs, hk.grad / gs, tol=1.5e-2, what="dkv")


def test_mha_window_map_rejects_bad_geometry(stg, gpu):
    from stgcma import kernels as k
    with pytest.raises(RuntimeError):
        k.MhaGeom(8, 1, 49, 96, 1.0, window=(14, 14, 7, 7))          # shift must be < ws
    with pytest.raises(RuntimeError):
        k.MhaGeom(6, 1, 49, 96, 1.0, window=(14, 14, 7, 0))          # 4 windows per frame: P must be a multiple of 4
    with pytest.raises(RuntimeError):
        k.MhaGeom(4, 1, 36, 96, 1.0, window=(14, 14, 7, 0))          # n != ws^2


@pytest.mark.parametrize("D,n,window", [(96, 196, None), (64, 49, (14, 14, 7, 3)), (96, 3136, None)])
def test_mha_pair_launch_equals_two_launches(stg, gpu, D, n, window):
    """stg_mha_fwd_pair / stg_mha_bwd_pair (both directions of a cross-modal pair, grid.z = 2 P) against two single launches: bit-identical."""
    from stgcma import kernels as k
    if window is None:
        P, rows = (2 if n > 1000 else 5), None
        geo = k.MhaGeom(P, 1, n, D, 1.0)
        rows = P * n
    else:
        Hi, Wi, ws, shift = window
        F = 3
        P = F * (Hi // ws) * (Wi // ws)
        geo = k.MhaGeom(P, 1, n, D, 1.0, window=window)
        rows = F * Hi * Wi
    g = torch.Generator().manual_seed(n + D)
    hv = (torch.randn(rows, D, generator=g) * 0.35).to(BF16).to(gpu)
    ha = (torch.randn(rows, D, generator=g) * 0.35).to(BF16).to(gpu)
    d0 = torch.randn(rows, D, generator=g).to(BF16).to(gpu)
    d1 = torch.randn(rows, D, generator=g).to(BF16).to(gpu)
    rv, lv = k.mha_fwd(geo, hv, ha, ha)
    ra, la = k.mha_fwd(geo, ha, hv, hv)
    (rv2, lv2), (ra2, la2) = k.mha_fwd_pair(geo, (hv, ha, ha), (ha, hv, hv))
    assert torch.equal(rv, rv2) and torch.equal(ra, ra2) and torch.equal(lv, lv2) and torch.equal(la, la2)
    single = [torch.full_like(hv, float("nan")) for _ in range(4)]
    k.mha_bwd(geo, hv, ha, ha, rv, lv, d0, dQ=single[0], dK=single[1], dV=None)
    k.mha_bwd(geo, ha, hv, hv, ra, la, d1, dQ=single[2], dK=single[3], dV=None)
    pair = [torch.full_like(hv, float("nan")) for _ in range(4)]
    k.mha_bwd_pair(geo, (hv, ha, ha, rv, lv, d0, pair[0], pair[1], None), (ha, hv, hv, ra, la, d1, pair[2], pair[3], None))
    for a, b in zip(single, pair):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D,n,window", [(96, 196, None), (64, 49, (14, 14, 7, 3)), (96, 49, (14, 14, 7, 0)), (96, 3136, None), (96, 97, None), (64, 130, None)])
def test_mha_pair_merged_backward(stg, gpu, D, n, window):
    """Round 6: stg_mha_bwd_pair_merged -- the backward of a cross-modal pair as ONE pass per modality (the two directions share S = X Y^T; G_X = dQ of
    X's own direction + dK + dV of the other) -- against the fp32 autograd gradients of the pair and against the two-direction path it replaces
    (mha_bwd_pair: dQ + dKV summed in fp32): frame-global and window-mapped problems, ragged token counts, both head dims."""
    from stgcma import kernels as k
    P = 2 if n > 1000 else 3
    g = torch.Generator().manual_seed(7 * n + D)
    if window is not None:
        Hi, Wi, ws, shift = window
        F = P
        rows = F * Hi * Wi
        idx = _window_index(F, Hi, Wi, ws, shift)                                   # [F * nW, n] rows
        geo = k.MhaGeom(idx.shape[0], 1, n, D, 1.0, window=window)
    else:
        rows = P * n
        idx = torch.arange(rows).view(P, n)
        geo = k.MhaGeom(P, 1, n, D, 1.0)
    xb = (torch.randn(rows, D, generator=g) * 0.35).to(BF16)
    yb = (torch.randn(rows, D, generator=g) * 0.35).to(BF16)
    dxo, dyo = torch.randn(rows, D, generator=g).to(BF16), torch.randn(rows, D, generator=g).to(BF16)
    x, y = xb.float().requires_grad_(True), yb.float().requires_grad_(True)
    S = torch.einsum("pid,pjd->pij", x[idx], y[idx])
    rx = torch.softmax(S, 2) @ y[idx]                                               # direction 0: queries X, keys = values Y
    ry = torch.softmax(S.transpose(1, 2), 2) @ x[idx]                               # direction 1: queries Y, keys = values X
    ((rx * dxo.float()[idx]).sum() + (ry * dyo.float()[idx]).sum()).backward()
    X, Y, dX, dY = xb.to(gpu), yb.to(gpu), dxo.to(gpu), dyo.to(gpu)
    (r0, l0), (r1, l1) = k.mha_fwd_pair(geo, (X, Y, Y), (Y, X, X))
    G0, G1 = k.mha_bwd_pair_merged(geo, (X, Y, r0, l0, dX), (Y, X, r1, l1, dY))
    a0, a1, b0, b1 = (torch.full_like(X, float("nan")) for _ in range(4))
    k.mha_bwd_pair(geo, (X, Y, Y, r0, l0, dX, a0, a1, None), (Y, X, X, r1, l1, dY, b0, b1, None))
    two0, two1 = a0.float() + b1.float(), b0.float() + a1.float()                   # dQ0 + dKV1 -> X ; dQ1 + dKV0 -> Y
    torch.cuda.synchronize()
    for got, two, ref, name in ((G0, two0, x.grad, "G_X"), (G1, two1, y.grad, "G_Y")):
        assert torch.isfinite(got.float()).all(), name
        sc = float(ref.abs().max())
        e_m = float((got.float().cpu() - ref).abs().max()) / sc
        e_t = float((two.cpu() - ref).abs().max()) / sc
        assert e_m <= max(1.5e-2, 2.0 * e_t), f"{name}: merged {e_m:.2e} of scale, two-direction path {e_t:.2e}"
