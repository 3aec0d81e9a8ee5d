"""-m gpu: each HIP kernel, called through the C ABI, against a plain fp32 PyTorch-CPU statement of the same op.

Tolerances: bf16 storage carries 8 significand bits, so an output of magnitude s is expected within ~s*2^-8 of the fp32
result; tests use |err| <= 1e-2 * max(1, |ref|) (the north-star's bf16 tolerance) unless a tighter bound is natural.
"""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def _bf(x):
    return x.to(BF16)


def _close(got, ref, tol=1e-2, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    bound = tol * torch.clamp(ref.abs(), min=1.0)
    bad = err > bound
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} elements off; first at {idx}: got "
                             f"{got[tuple(idx)].item()} ref {ref[tuple(idx)].item()}; max err {err.max().item():.4g}")


def _gelu_grad(z):
    return 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (130, 72, 16), (1000, 384, 128), (257, 29, 512), (4096, 512, 2048),
                                   (300, 16, 48), (64, 2048, 512), (1, 128, 128)])
def test_gemm_plain(stg, gpu, M, N, K):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.1)
    b = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + b
    out = k.gemm_nt(A.to(gpu), W.to(gpu), b.to(gpu))
    _close(out, ref, what=f"gemm {M}x{N}x{K}")
    out32 = k.gemm_nt(A.to(gpu), W.to(gpu), b.to(gpu), out_dtype=F32)
    _close(out32, ref, tol=2e-3, what=f"gemm f32 out {M}x{N}x{K}")


def test_gemm_identity_asymmetric(stg, gpu):
    """A = I with an asymmetric W catches a transposed accumulator mapping."""
    from stgcma import kernels as k
    n = 128
    A = torch.eye(n).to(BF16)
    W = (torch.arange(n * n, dtype=F32).reshape(n, n) % 251 - 125.0).to(BF16)
    out = k.gemm_nt(A.to(gpu), W.to(gpu), out_dtype=F32)
    assert torch.equal(out.cpu(), W.float().t())


def test_gemm_epilogues(stg, gpu):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(5)
    M, N, K = 390, 256, 128
    A = _bf(torch.randn(M, K, generator=g)); W = _bf(torch.randn(N, K, generator=g) * 0.1)
    b = torch.randn(N, generator=g)
    r1 = _bf(torch.randn(M, N, generator=g)); r2 = _bf(torch.randn(M, N, generator=g))
    z = A.float() @ W.float().t() + b
    # GELU + saved derivative (the backward's multiplier)
    out, dact = k.gemm_nt(A.to(gpu), W.to(gpu), b.to(gpu), act=k.ACT_GELU, want_dact=True)
    _close(dact, _gelu_grad(z), what="gelu'")
    _close(out, torch.nn.functional.gelu(z), what="gelu")
    # QuickGELU
    out, dact = k.gemm_nt(A.to(gpu), W.to(gpu), b.to(gpu), act=k.ACT_QUICKGELU, want_dact=True)
    sg = torch.sigmoid(1.702 * z)
    _close(out, z * sg, what="quickgelu")
    _close(dact, sg * (1 + 1.702 * z * (1 - sg)), what="quickgelu'")
    # residuals + alpha
    out = k.gemm_nt(A.to(gpu), W.to(gpu), b.to(gpu), alpha=0.5, res1=r1.to(gpu), res2=r2.to(gpu))
    _close(out, 0.5 * (A.float() @ W.float().t()) + b + r1.float() + r2.float(), what="residuals")
    # activation backward: t * saved derivative
    src = _bf(torch.randn(M, N, generator=g))
    out = k.gemm_nt(A.to(gpu), W.to(gpu), None, dact_src=src.to(gpu))
    _close(out, (A.float() @ W.float().t()) * src.float(), what="dact_src")
    with pytest.raises(RuntimeError):
        k.gemm_nt(A.to(gpu), W.to(gpu), None, want_dact=True)                  # a derivative needs an activation
    # DropPath row scale: rows are (b t) n ; mask per (b, n): idx = (m // (T*n)) * n + m % n
    T, n = 3, 13
    Bc = M // (T * n)
    scale = torch.rand(Bc * n, generator=g)
    Mm = Bc * T * n
    out = k.gemm_nt(A[:Mm].to(gpu), W.to(gpu), b.to(gpu), row_scale=scale.to(gpu), rs_outer=T * n, rs_inner=n,
                    res1=r1[:Mm].to(gpu))
    m = torch.arange(Mm)
    idx = (m // (T * n)) * n + m % n
    _close(out, z[:Mm] * scale[idx][:, None] + r1[:Mm].float(), what="row_scale")



@pytest.mark.parametrize("M,N,K", [(390, 256, 128), (1000, 136, 64), (257, 72, 40), (64, 8, 16), (130, 29, 128), (4099, 384, 512)])
def test_gemm_epilogue_layouts(stg, gpu, M, N, K):
    """Row-layout epilogue (N % 8 == 0) and the element fallback (N = 29): partial row / column tiles, fp32 and bf16
    residuals and outputs, strided views, every option at once."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)); W = _bf(torch.randn(N, K, generator=g) * 0.2)
    b = torch.randn(N, generator=g)
    r1 = _bf(torch.randn(M, N, generator=g)); r2 = torch.randn(M, N, generator=g)
    src = _bf(torch.rand(M, N, generator=g))
    z = A.float() @ W.float().t()
    # fp32 output, bf16 + fp32 residuals (the residual-stream update)
    out = k.gemm_nt(A.to(gpu), W.to(gpu), b.to(gpu), res1=r1.to(gpu), res2=r2.to(gpu), out_dtype=F32)
    assert out.dtype == F32
    _close(out, z + b + r1.float() + r2, tol=4e-3, what="f32 out")
    # into a row slice of a wider fp32 buffer (column offset => strided rows)
    big = torch.full((M, N + 16), 7.0, device=gpu)
    k.gemm_nt(A.to(gpu), W.to(gpu), None, out=big[:, 8:8 + N], res1=r2.to(gpu))
    _close(big[:, 8:8 + N], z + r2, tol=4e-3, what="strided out")
    assert float((big[:, :8] - 7).abs().max()) == 0 and float((big[:, 8 + N:] - 7).abs().max()) == 0
    # gelu + derivative + derivative source + bf16 out
    out, d = k.gemm_nt(A.to(gpu), W.to(gpu), b.to(gpu), act=k.ACT_GELU, want_dact=True, dact_src=src.to(gpu), res1=r1.to(gpu))
    _close(out, torch.nn.functional.gelu(z + b) * src.float() + r1.float(), what="gelu*src+res")
    _close(d, _gelu_grad(z + b), what="gelu' saved")

def test_gemm_strided_views(stg, gpu):
    """Operands / outputs that are column slices of wider buffers (how the fused a/v tensors are addressed)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(9)
    M, N, K = 200, 64, 128
    Abig = _bf(torch.randn(M, 3 * K, generator=g)).to(gpu)
    W = _bf(torch.randn(N, K, generator=g) * 0.1).to(gpu)
    Cbig = torch.zeros(M, 4 * N, dtype=BF16, device=gpu)
    k.gemm_nt(Abig[:, K:2 * K], W, out=Cbig[:, N:2 * N])
    ref = Abig[:, K:2 * K].float().cpu() @ W.float().cpu().t()
    _close(Cbig[:, N:2 * N], ref, what="strided")
    assert float(Cbig[:, :N].abs().max()) == 0 and float(Cbig[:, 2 * N:].abs().max()) == 0


@pytest.mark.parametrize("M,N1,N2", [(64, 16, 128), (1000, 32, 256), (333, 128, 16), (320, 29, 512), (5000, 64, 1024),
                                     (100, 512, 2048),
                                     # the workspace (no-atomics) path: narrow operand <= 32, M >= 4096; either operand narrow,
                                     # ragged row / column tails, several column groups
                                     (4096, 16, 128), (20000, 32, 512), (7777, 512, 32), (5001, 24, 200), (62720, 32, 512),
                                     (9000, 256, 16), (4100, 29, 512), (8000, 48, 768), (6000, 768, 48), (15680, 64, 1024), (4500, 1024, 64), (20000, 96, 768), (9000, 192, 96), (5000, 80, 384)])
def test_wgrad(stg, gpu, M, N1, N2):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + N1 + N2)
    dY = _bf(torch.randn(M, N1, generator=g)); X = _bf(torch.randn(M, N2, generator=g))
    dW = torch.zeros(N1, N2, device=gpu); db = torch.zeros(N1, device=gpu)
    k.wgrad_tn(dY.to(gpu), X.to(gpu), dW, db)
    ref = dY.float().t() @ X.float()
    scale = math.sqrt(M)
    _close(dW / scale, ref / scale, tol=2e-3, what="dW")
    _close(db / scale, dY.float().sum(0) / scale, tol=2e-3, what="db")
    # accumulates
    k.wgrad_tn(dY.to(gpu), X.to(gpu), dW, None)
    _close(dW / scale, 2 * ref / scale, tol=2e-3, what="dW accumulate")


@pytest.mark.parametrize("M,N1,N2,outer,inner", [(6000, 512, 32, 600, 20), (8192, 16, 128, 4096, 1)])
def test_wgrad_row_scale_workspace_path(stg, gpu, M, N1, N2, outer, inner):
    """DropPath row scale on dY rows (T_Adapter D_fc2 wgrad, Swin_AVE.py:709,715) through the workspace kernels, column slices
    of wider buffers as operands."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + N1)
    dYb = _bf(torch.randn(M, N1 + 8, generator=g)); Xb = _bf(torch.randn(M, N2 + 16, generator=g))
    dY, X = dYb[:, :N1], Xb[:, 8:8 + N2]
    n_rs = ((M - 1) // outer) * inner + inner
    rs = (torch.rand(n_rs, generator=g) < 0.8).float() / 0.8
    row = torch.arange(M)
    s = rs[(row // outer) * inner + (row % inner)]
    dW = torch.zeros(N1, N2, device=gpu); db = torch.zeros(N1, device=gpu)
    k.wgrad_tn(dYb.to(gpu)[:, :N1], Xb.to(gpu)[:, 8:8 + N2], dW, db, row_scale=rs.to(gpu), rs_outer=outer, rs_inner=inner)
    ys = _bf(dY.float() * s[:, None]).float()          # the kernels scale the bf16 operand and round it back to bf16
    scale = math.sqrt(M)
    _close(dW / scale, ys.t() @ X.float() / scale, tol=3e-3, what="dW")
    _close(db / scale, ys.sum(0) / scale, tol=3e-3, what="db")


@pytest.mark.parametrize("rows,C", [(7, 128), (1000, 256), (513, 512), (100, 1024), (50, 2048), (33, 768), (20, 3072),
                                    (9, 4096), (64, 48)])
@pytest.mark.parametrize("xdt", [BF16, F32])
def test_layernorm(stg, gpu, rows, C, xdt):
    from stgcma import kernels as k
    if C % 8:
        pytest.skip("C % 8")
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(xdt)
    gamma = torch.randn(C, generator=g) * 0.2 + 1; beta = torch.randn(C, generator=g) * 0.1
    xr = x.float().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(xr, (C,), gamma, beta, 1e-5)
    y, mean, rstd = k.layernorm_fwd(x.to(gpu), gamma.to(gpu), beta.to(gpu))
    _close(y, y_ref, what="ln fwd")
    _close(mean, x.float().mean(1), tol=1e-4, what="mean")
    dy = _bf(torch.randn(rows, C, generator=g))
    add = _bf(torch.randn(rows, C, generator=g))
    y_ref.backward(dy.float())
    dgam = torch.zeros(C, device=gpu); dbet = torch.zeros(C, device=gpu)
    dx = k.layernorm_bwd(dy.to(gpu), x.to(gpu), gamma.to(gpu), mean, rstd, add_to=add.to(gpu), dgamma=dgam, dbeta=dbet)
    _close(dx, xr.grad + add.float(), tol=2e-2, what="ln bwd")
    xh = (x.float() - x.float().mean(1, keepdim=True)) * torch.rsqrt(x.float().var(1, unbiased=False, keepdim=True) + 1e-5)
    _close(dgam / math.sqrt(rows), (dy.float() * xh).sum(0) / math.sqrt(rows), tol=1e-2, what="dgamma")
    _close(dbet / math.sqrt(rows), dy.float().sum(0) / math.sqrt(rows), tol=1e-2, what="dbeta")


def test_layernorm_patch_merge(stg, gpu):
    """gather4 == PatchMerging's x0..x3 strided gather + cat + LayerNorm(4C) (reference Swin_AVE.py:967-976)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(3)
    F_, H, W, C = 3, 6, 8, 32
    x = _bf(torch.randn(F_ * H * W, C, generator=g))
    gamma = torch.randn(4 * C, generator=g) * 0.2 + 1; beta = torch.randn(4 * C, generator=g) * 0.1
    xr = x.float().requires_grad_(True)
    xv = xr.view(F_, H, W, C)
    cat = torch.cat([xv[:, 0::2, 0::2], xv[:, 1::2, 0::2], xv[:, 0::2, 1::2], xv[:, 1::2, 1::2]], -1).reshape(-1, 4 * C)
    y_ref = torch.nn.functional.layer_norm(cat, (4 * C,), gamma, beta, 1e-5)
    y, mean, rstd = k.layernorm_fwd(x.to(gpu), gamma.to(gpu), beta.to(gpu), gather4=(H, W))
    _close(y, y_ref, what="merge ln fwd")
    dy = _bf(torch.randn(F_ * H * W // 4, 4 * C, generator=g))
    y_ref.backward(dy.float())
    dx = k.layernorm_bwd(dy.to(gpu), x.to(gpu), gamma.to(gpu), mean, rstd, gather4=(H, W))
    _close(dx, xr.grad, tol=2e-2, what="merge ln bwd")


def test_elementwise(stg, gpu):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(11)
    for numel_shape in [(1000, 16), (77, 24), (3, 5)]:
        h = _bf(torch.randn(*numel_shape, generator=g)); r = _bf(torch.randn(*numel_shape, generator=g))
        gate = torch.tensor([0.37])
        out = k.gate_fwd(h.to(gpu), r.to(gpu), gate.to(gpu))
        _close(out, h.float() + 0.37 * r.float(), what="gate fwd")
        dgate = torch.zeros(1, device=gpu)
        dr = k.gate_bwd(h.to(gpu), r.to(gpu), gate.to(gpu), dgate)
        _close(dr, 0.37 * h.float(), what="gate bwd dr")
        ref = (h.float() * r.float()).sum()
        assert abs(float(dgate) - float(ref)) <= 1e-3 * max(1.0, abs(float(ref))) + 1e-2
        _close(k.add(h.to(gpu), r.to(gpu)), h.float() + r.float(), what="add")
        mask = (torch.rand(*numel_shape, generator=g) > 0.5).float() * 2
        _close(k.mul_mask(h.to(gpu), mask.to(gpu)), h.float() * mask, what="mul_mask")
    w = torch.randn(48, 100, generator=g)
    c = k.cast_bf16(w.to(gpu))                       # [48, 100 -> padded to 104]
    assert tuple(c.shape) == (48, 104) and float(c[:, 100:].abs().max()) == 0
    _close(c[:, :100], w, what="cast")
    ct = k.cast_bf16(w.to(gpu), transpose=True)       # [100, 48]
    _close(ct, w.t(), what="cast T")
    w2 = torch.randn(29, 512, generator=g)
    ct2 = k.cast_bf16(w2.to(gpu), transpose=True)     # [512, 29 -> 32]
    assert tuple(ct2.shape) == (512, 32) and float(ct2[:, 29:].abs().max()) == 0
    _close(ct2[:, :29], w2.t(), what="cast T pad")
    z = _bf(torch.randn(33, 24, generator=g)); dh = _bf(torch.randn(33, 24, generator=g))
    _close(k.act_bwd(dh.to(gpu), z.to(gpu)), dh.float() * z.float(), what="act_bwd")
    _close(k.add(h.to(gpu), r.to(gpu), h.to(gpu)), 2 * h.float() + r.float(), what="add3")
    _close(k.cast_f32(_bf(w).to(gpu)), _bf(w).float(), tol=0, what="cast f32")
    # meanpool
    G, n, C = 6, 49, 64
    x = _bf(torch.randn(G * n, C, generator=g))
    big = torch.zeros(G, 2 * C, dtype=F32, device=gpu)
    k.meanpool_fwd(x.to(gpu), G, n, out=big[:, C:])
    _close(big[:, C:], x.float().view(G, n, C).mean(1), tol=1e-3, what="meanpool")
    d = _bf(torch.randn(G, C, generator=g))
    _close(k.meanpool_bwd(d.to(gpu), G, n), (d.float() / n)[:, None, :].expand(G, n, C).reshape(G * n, C), what="meanpool bwd")
    for G2, n2, C2 in ((5, 3136, 128), (3, 784, 256), (2, 300, 24), (4, 257, 1024)):      # long groups: the workgroup-per-group kernel (round 5)
        x2 = _bf(torch.randn(G2 * n2, C2, generator=g))
        _close(k.meanpool_fwd(x2.to(gpu), G2, n2), x2.float().view(G2, n2, C2).mean(1), tol=4e-3, what=f"meanpool {G2}x{n2}x{C2}")
        _close(k.meanpool_fwd(x2.to(gpu), G2, n2, out_dtype=F32), x2.float().view(G2, n2, C2).mean(1), tol=1e-4, what=f"meanpool f32 {G2}x{n2}x{C2}")
    # bias gather / scatter
    L, H, nn_ = 13, 4, 49
    table = torch.randn(L, H, generator=g); index = torch.randint(0, L, (nn_,), generator=g)
    out = k.bias_gather(table.to(gpu), index.to(gpu))
    assert torch.equal(out.cpu(), table[index].t().contiguous())
    dtab = torch.zeros(L, H, device=gpu)
    k.bias_scatter(out, index.to(gpu), dtab)
    ref = torch.zeros(L, H).index_add_(0, index, table[index])
    _close(dtab, ref, tol=1e-5, what="bias scatter")


def test_im2col_patch_embed(stg, gpu):
    """Conv3d(k=s=(1,4,4)) == im2col gather + GEMM (reference Swin_AVE.py:1097,1115)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(2)
    for Cin, Kpad in [(3, 48), (1, 16)]:
        B, T, H, W, p, E = 2, 3, 16, 24, 4, 32
        x = torch.randn(B, Cin, T, H, W, generator=g)
        w = torch.randn(E, Cin, 1, p, p, generator=g) * 0.2; b = torch.randn(E, generator=g)
        ref = torch.nn.functional.conv3d(_bf(x).float(), _bf(w).float(), b, stride=(1, p, p))
        ref = ref.permute(0, 2, 3, 4, 1).reshape(-1, E)  # (b t h w) c
        cols = k.im2col_patch(x.to(gpu), p, Kpad)
        wb = _bf(w.reshape(E, -1)).to(gpu)
        out = k.gemm_nt(cols, wb, b.to(gpu))
        _close(out, ref, what=f"patch embed Cin={Cin}")


def test_cast_bf16_multi_equals_single_casts(stg, gpu):
    """ops.ShadowSet: one launch for every trainable weight matrix, both orientations == stg_cast_bf16 per matrix."""
    from stgcma import kernels as k, ops
    g = torch.Generator().manual_seed(3)
    shapes = [(16, 128), (128, 16), (29, 512), (512, 2048), (48, 768), (3, 5)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g).to(gpu)) for s in shapes]
    ss = ops.ShadowSet(params)
    ss.refresh()
    for p in params:
        for tr in (False, True):
            got = ops.shadow(p, tr)
            ref = k.cast_bf16(p.detach(), transpose=tr)
            assert got.shape == ref.shape and torch.equal(got, ref), (tuple(p.shape), tr)
    old = ops.shadow(params[0]).clone()
    with torch.no_grad():
        params[0].add_(1.0)                       # version bump -> the next refresh re-casts into a NEW arena
    ss.refresh()
    assert torch.equal(ops.shadow(params[0]), k.cast_bf16(params[0].detach()))
    assert not torch.equal(ops.shadow(params[0]), old)


@pytest.mark.parametrize("M,N,K,epi", [(4097, 768, 3072, "b"), (300, 256, 1024, ""), (1000, 512, 2048, "bap"), (777, 1024, 1152, "d"),
                                       (256, 256, 128, "b"), (5000, 256, 1536, "r"), (2049, 512, 1024, "bR"),
                                       (3000, 768, 96, "b"), (1000, 192, 96, "r"), (700, 96, 72, ""), (515, 1536, 96, "bR"), (900, 384, 80, "bap")])
def test_gemm_long_k_shapes(stg, gpu, M, N, K, epi):
    """Shapes the host dispatch routes to the 8-phase 256 x 256 kernel (K >= 1024, K % 128 == 0, N % 256 == 0) and its
    neighbours (K % 128 != 0 -> large-tile kernel; K = 128 -> 128 x 128 kernel), with row tails and every epilogue family."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.05)
    b = torch.randn(N, generator=g) if "b" in epi else None
    z = A.float() @ W.float().t() + (b if b is not None else 0.)
    kw = {}
    ref = z
    if "a" in epi:
        kw.update(act=k.ACT_GELU, want_dact=True)
        ref = torch.nn.functional.gelu(z)
    if "d" in epi:
        src = _bf(torch.randn(M, N, generator=g))
        kw["dact_src"] = src.to(gpu)
        ref = z * src.float()
    if "r" in epi:
        r1 = _bf(torch.randn(M, N, generator=g))
        kw["res1"] = r1.to(gpu)
        ref = ref + r1.float()
    if "R" in epi:
        r2 = torch.randn(M, N, generator=g)
        kw.update(res1=_bf(torch.zeros(M, N)).to(gpu), res2=r2.to(gpu), out_dtype=F32)
        ref = ref + r2
    out = k.gemm_nt(A.to(gpu), W.to(gpu), None if b is None else b.to(gpu), **kw)
    if "a" in epi:
        out, dact = out
        _close(dact, _gelu_grad(z), what=f"dact {M}x{N}x{K}")
    _close(out, ref, tol=2e-2 if K >= 2048 else 1e-2, what=f"gemm {epi} {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,epi,kernel", [(98400, 576, 192, "b", "gemm_nt_8phm"), (98500, 1152, 384, "b", "gemm_nt_8phm"), (98304 + 77, 768, 192, "bap8", "gemm_nt_8phm"),
                                              (98304 + 200, 768, 192, "d8", "gemm_nt_8phm"), (8300, 576, 192, "b", "gemm_nt_8ph_"), (8193, 1152, 384, "bap8", "gemm_nt_8ph_"), (8200, 192, 768, "b", "gemm_nt_8ph_"), (8200, 192, 576, "", "gemm_nt_8ph_"),
                                              (8300, 384, 1152, "", "gemm_nt_8ph_"), (8200, 384, 1536, "b", "gemm_nt_8ph_"), (8200, 192, 192, "b", "gemm_nt_8ph_"),
                                              (8211, 384, 384, "", "gemm_nt_8ph_"), (8448, 320, 192, "", "gemm_nt_8ph_"), (8192, 448, 320, "b", "gemm_nt_8ph_"),
                                              (8250, 1344, 448, "bap8", "gemm_nt_8ph")])
def test_gemm_nx_shapes(stg, gpu, M, N, K, epi, kernel):
    """Round 6: the NX forms of the 8-phase kernels -- N % 64 == 0 (a last column tile of 64 / 128 / 192 valid columns, its W1 half skipped
    when empty) and K % 64 == 0 (an odd k-tile count padded with a zero k-tile): Swin-L's widths (192 / 384 / 576 / 1152) and a few others,
    row tails everywhere; against the fp32 product, and the dispatch must really have taken an 8-phase kernel."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.05)
    b = torch.randn(N, generator=g) if "b" in epi else None
    z = A.float() @ W.float().t() + (b if b is not None else 0.)
    kw, ref = {}, z
    if "a" in epi:
        kw.update(act=k.ACT_GELU, want_dact="u8")
        ref = torch.nn.functional.gelu(z)
    if "d8" in epi:
        d8 = torch.randint(0, 256, (M, N), generator=g, dtype=torch.uint8)
        kw["dact_src"] = d8.to(gpu)
        ref = z * (d8.float() * 0.005 - 0.14)
    # poison the allocator's free blocks: a column the kernel fails to write must not look right by accident
    t = torch.full((M, N), float("nan"), dtype=BF16, device=gpu); del t
    stg.configure(lib_gemm_nx=2)        # every legal NX shape (the default routes only the classes measured faster: N < 256, K >= 768)
    try:
        out = k.gemm_nt(A.to(gpu), W.to(gpu), None if b is None else b.to(gpu), **kw)
    finally:
        stg.configure(lib_gemm_nx=1)
    assert k.LAST_GEMM_KERNEL.startswith(kernel), k.LAST_GEMM_KERNEL
    if "a" in epi:
        out, dact = out
        assert float((dact.float().cpu() * 0.005 - 0.14 - _gelu_grad(z)).abs().max()) <= 1.5e-2
    assert torch.isfinite(out.float()).all()
    _close(out, ref, tol=1e-2, what=f"gemm nx {epi} {M}x{N}x{K}")
    # the same call on the round-5 route (128 x 128 kernel): equal up to the bf16 rounding of the output
    stg.configure(lib_gemm_nx=0)
    try:
        o0 = k.gemm_nt(A.to(gpu), W.to(gpu), None if b is None else b.to(gpu), **kw)
        assert not k.LAST_GEMM_KERNEL.startswith("gemm_nt_8ph"), k.LAST_GEMM_KERNEL
    finally:
        stg.configure(lib_gemm_nx=1)
    o0 = o0[0] if isinstance(o0, tuple) else o0
    assert float((o0.float() - out.float()).abs().max()) <= 2.0 ** -7 * float(out.float().abs().max())


@pytest.mark.parametrize("M,N,K,split", [(125440, 32, 512, 62720), (8192, 16, 128, 0), (9999, 32, 256, 4096), (8200, 64, 512, 0), (20011, 16, 256, 128), (8192, 32, 128, 0)])
def test_adapter_down_projection_row_stream(stg, gpu, M, N, K, split):
    """Round 6 (skinny.hip): the adapters' down-projection + bias + GELU + saved bf16 derivative as a row stream with the weight's MFMA fragments in LDS,
    routed from stg_gemm_nt -- BIT-IDENTICAL to the 128 x 128 tile kernel (same k order, same polynomial GELU forms) on whole and ragged row counts, with
    and without the video | audio split (two weights, one launch), and against the fp32 product."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).to(BF16).to(gpu)
    W1, W2 = (torch.randn(N, K, generator=g) * 0.06).to(BF16).to(gpu), (torch.randn(N, K, generator=g) * 0.06).to(BF16).to(gpu)
    b1, b2 = (torch.randn(N, generator=g) * 0.2).to(gpu), (torch.randn(N, generator=g) * 0.2).to(gpu)
    kw = dict(act=k.ACT_GELU, want_dact=True)
    if split:
        kw["split"] = (split, W2, b2)
    t = torch.full((M, N), float("nan"), dtype=BF16, device=gpu); del t          # an unwritten row must not look right by accident
    H, Z = k.gemm_nt(A, W1, b1, **kw)
    assert k.LAST_GEMM_KERNEL == "skinny_down_kernel", k.LAST_GEMM_KERNEL
    stg.configure(lib_gemm_nx=0)
    try:
        H0, Z0 = k.gemm_nt(A, W1, b1, **kw)
        assert k.LAST_GEMM_KERNEL != "skinny_down_kernel"
    finally:
        stg.configure(lib_gemm_nx=1)
    assert torch.equal(H.view(torch.int16), H0.view(torch.int16)) and torch.equal(Z.view(torch.int16), Z0.view(torch.int16)), "not bit-identical to the tile kernel"
    s = split if split else M
    z = torch.cat([A[:s].float() @ W1.float().t() + b1, A[s:].float() @ W2.float().t() + b2]) if split else A.float() @ W1.float().t() + b1
    _close(H, torch.nn.functional.gelu(z).cpu(), what="gelu(down)")
    _close(Z, _gelu_grad(z.cpu()), what="gelu'(down)")


@pytest.mark.parametrize("F_,H,W,Cin,Cout,d", [(3, 14, 14, 64, 64, 1), (2, 28, 28, 256, 256, 3), (1, 7, 7, 128, 32, 1), (2, 14, 14, 256, 256, 18),
                                               (5, 56, 56, 64, 128, 6), (2, 28, 28, 32, 128, 1), (3, 14, 14, 16, 64, 2), (1, 14, 14, 8, 32, 1)])
def test_implicit_conv3x3_equals_im2col_gemm(stg, gpu, F_, H, W, Cin, Cout, d):
    """stg_gemm_nt in implicit-convolution mode (DMA sources gathered per tap, zero line for the padding) against the im2col
    image + the same GEMM: same k order, so bit-identical; and against F.conv2d in fp32."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(F_ + H + Cin + d)
    x = _bf(torch.randn(F_ * H * W, Cin, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    wm = _bf(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin))
    y_imp = k.gemm_nt(x.to(gpu), wm.to(gpu), b.to(gpu), conv=(H, W, d))
    y_col = k.gemm_nt(k.im2col3x3(x.to(gpu), F_, H, W, d), wm.to(gpu), b.to(gpu))
    if Cin % 64 == 0:
        assert torch.equal(y_imp, y_col)
    else:                               # K = 9 Cin is not a multiple of 64: the im2col path runs on another kernel (other k order)
        _close(y_imp, y_col.float(), tol=1e-2, what="implicit vs im2col")
    xn = x.float().view(F_, H, W, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xn, _bf(w).float(), b, padding=d, dilation=d).permute(0, 2, 3, 1).reshape(-1, Cout)
    _close(y_imp, ref, tol=2e-2, what="implicit conv")


@pytest.mark.parametrize("F_,H,W,I,O,d", [(3, 14, 14, 128, 128, 1), (2, 28, 28, 256, 256, 3), (7, 7, 7, 256, 128, 1), (2, 14, 14, 256, 256, 18),
                                          (4, 56, 56, 128, 256, 6), (1, 14, 14, 256, 256, 1), (2, 28, 28, 128, 32, 1), (1, 14, 14, 128, 200, 2),
                                          # round 5: I % 64 == 0 (a 128-column tile spans two taps; 9 I not a multiple of 128: a zero-filled tail) and
                                          # odd row-block counts for the double-buffered ring
                                          (3, 56, 56, 64, 256, 3), (2, 14, 14, 320, 256, 12), (1, 7, 7, 64, 128, 1), (5, 28, 28, 192, 64, 2), (1, 9, 7, 64, 8, 1),
                                          # round 5b: any I % 8 == 0 (a 128-column tile spans up to 16 taps' pieces), and O < 64 <= I = the swapped-operand
                                          # call (the taps move to dy, the result comes back tap-flipped and transposed), with dilation
                                          (2, 14, 14, 32, 128, 1), (1, 9, 7, 8, 64, 2), (2, 12, 10, 24, 72, 1), (2, 28, 28, 128, 32, 3), (1, 14, 14, 256, 16, 6),
                                          (3, 7, 7, 64, 56, 1)])
def test_conv3x3_wgrad(stg, gpu, F_, H, W, I, O, d):
    """The conv weight gradient without the im2col image against autograd of F.conv2d in fp32 (same bf16-rounded operands)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(F_ + H + I + d)
    x = _bf(torch.randn(F_ * H * W, I, generator=g))
    dy = _bf(torch.randn(F_ * H * W, O, generator=g))
    w = torch.zeros(O, I, 3, 3, requires_grad=True)
    xn = x.float().view(F_, H, W, I).permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(xn, w, None, padding=d, dilation=d)
    y.backward(dy.float().view(F_, H, W, O).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(O, 9 * I)
    got, db = k.conv3x3_wgrad(dy.to(gpu), x.to(gpu), F_, H, W, d, want_db=True)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    err = float((got.cpu() - ref).abs().max())
    assert err <= 2e-3 * scale, (err, scale)
    assert db.cpu().allclose(dy.float().sum(0), rtol=1e-4, atol=2e-2)          # bias gradient = column sums of dy (same launch, or stg_bn_colsum when swapped)


@pytest.mark.parametrize("M,N1,N2", [(5000, 128, 128), (20000, 256, 1024), (4096, 256, 128), (31360, 128, 256)])
def test_wgrad_wide(stg, gpu, M, N1, N2):
    """wgrad_tn with both widths multiples of 128 takes the tn-GEMM with partial tiles (no atomics): against fp32 torch."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + N1 + N2)
    dy = _bf(torch.randn(M, N1, generator=g))
    x = _bf(torch.randn(M, N2, generator=g))
    dW = torch.full((N1, N2), 0.5, device=gpu)
    db = torch.full((N1,), -1.0, device=gpu)
    k.wgrad_tn(dy.to(gpu), x.to(gpu), dW, db)
    ref = dy.float().t() @ x.float() + 0.5
    scale = float(ref.abs().max())
    assert float((dW.cpu() - ref).abs().max()) <= 2e-3 * scale
    assert torch.allclose(db.cpu(), dy.float().sum(0) - 1.0, rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("nb,rows,N1,N2", [(3, 5, 128, 128), (4, 245, 128, 128), (2, 15680, 128, 256)])
def test_bmm_tn(stg, gpu, nb, rows, N1, N2):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(nb + rows)
    A = _bf(torch.randn(nb * rows, N1, generator=g))
    B = _bf(torch.randn(nb * rows, N2, generator=g))
    got = k.bmm_tn(A.to(gpu), B.to(gpu), rows).cpu()
    ref = torch.einsum("brn,brm->bnm", A.float().view(nb, rows, N1), B.float().view(nb, rows, N2))
    assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("nb,M,N,K", [(3, 5, 128, 128), (4, 245, 128, 128), (2, 1000, 256, 64)])
def test_gemm_batched(stg, gpu, nb, M, N, K):
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(nb + M + N)
    A = _bf(torch.randn(nb * M, K, generator=g))
    W = _bf(torch.randn(nb, N, K, generator=g) * 0.1)
    got = k.gemm_nt(A.to(gpu), W.to(gpu), alpha=0.2, batch=nb)
    ref = 0.2 * torch.einsum("bmk,bnk->bmn", A.float().view(nb, M, K), W.float()).reshape(nb * M, N)
    _close(got, ref, what="batched gemm")


@pytest.mark.parametrize("M,N,K,act", [(300, 512, 128, "gelu"), (8192, 2048, 512, "gelu"), (1000, 3072, 768, "qgelu")])
def test_gemm_u8_saved_derivative(stg, gpu, M, N, K, act):
    """The 8-bit linear code of the saved activation derivative (STG_U8_LIN: the [rows, 4C] MLP tensor, Swin_AVE.py:119-126): the
    activation output is BIT-identical to the bf16-derivative variant's, the decoded derivative is within half a code step (0.0025)
    of act'(t), and the backward epilogue (dact_src = the codes) equals the product with the decoded values."""
    from stgcma import kernels as Kn
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(BF16).to(gpu)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF16).to(gpu)
    b = torch.randn(N, generator=g).to(gpu)
    a = Kn.ACT_GELU if act == "gelu" else Kn.ACT_QUICKGELU
    h16, d16 = Kn.gemm_nt(A, W, b, act=a, want_dact=True)
    h8, d8 = Kn.gemm_nt(A, W, b, act=a, want_dact="u8")
    assert d8.dtype == torch.uint8 and tuple(d8.shape) == (M, N)
    assert torch.equal(h16, h8)
    t = (A.float() @ W.float().t() + b).double()
    if act == "gelu":
        dref = 0.5 * (1 + torch.erf(t / 2 ** 0.5)) + t * torch.exp(-0.5 * t * t) / (2 * np.pi) ** 0.5
    else:
        sg = torch.sigmoid(1.702 * t)
        dref = sg * (1 + 1.702 * t * (1 - sg))
    dec = d8.double() * 0.005 - 0.14
    assert float((dec - dref).abs().max()) <= 0.0025 + 2e-3          # half a code step + the bf16 operands' effect on t
    assert float((dec - d16.double()).abs().max()) <= 0.0025 + 0.0045   # vs the bf16 derivative (its own rounding: 2^-8 at 1)
    dM = torch.randn(M, K, generator=g).to(BF16).to(gpu)              # backward: dZ = (dM . W2^T) * act'  with W2^T := W (shape [N, K])
    z16 = Kn.gemm_nt(dM, W, dact_src=d16)
    z8 = Kn.gemm_nt(dM, W, dact_src=d8)
    ref8 = (dM.float() @ W.float().t()).double() * dec
    scale = float(ref8.abs().max())
    assert float((z8.double() - ref8).abs().max()) <= 6e-3 * scale    # bf16 rounding of the stored product
    assert float((z8.float() - z16.float()).norm() / z16.float().norm()) <= 6e-3


def test_wgrad_multi_matches_single_calls(stg, gpu):
    """stg_wgrad_tn_ws_multi (the adapter weight gradients of a block in one launch pair) against the same problems issued one by
    one: bit-identical (same kernels, same row splits), incl. the DropPath row scale and both operand orientations; a problem with a
    different launch plan is carried out on its own."""
    from stgcma import kernels as Kn
    g = torch.Generator().manual_seed(11)
    M, C_, dh = 8192, 256, 32
    T, N = 2, 64
    probs, refs = [], []
    for i in range(5):
        narrow_first = i % 2 == 0
        dY = torch.randn(M, dh if narrow_first else C_, generator=g).to(BF16).to(gpu)
        X = torch.randn(M, C_ if narrow_first else dh, generator=g).to(BF16).to(gpu)
        rs = (torch.rand(M // (T * N) * N, generator=g) < 0.7).float().to(gpu) / 0.7 if i == 3 else None
        shp = (dY.shape[1], X.shape[1])
        dW = torch.randn(shp, generator=g).to(gpu)
        db = torch.randn(shp[0], generator=g).to(gpu)
        dW1, db1 = dW.clone(), db.clone()
        Kn.wgrad_tn(dY, X, dW1, db1, row_scale=rs, rs_outer=T * N, rs_inner=N)
        probs.append((dY, X, dW, db, rs, T * N, N))
        refs.append((dW1, db1))
    # an odd one out: other M (another launch plan) -- must still come out right
    dY = torch.randn(4096, dh, generator=g).to(BF16).to(gpu); X = torch.randn(4096, C_, generator=g).to(BF16).to(gpu)
    dW = torch.zeros(dh, C_, device=gpu); db = torch.zeros(dh, device=gpu)
    dW1, db1 = dW.clone(), db.clone()
    Kn.wgrad_tn(dY, X, dW1, db1)
    probs.append((dY, X, dW, db, None, 1, 1)); refs.append((dW1, db1))
    Kn.wgrad_tn_multi(probs)
    for (dY, X, dW, db, *_), (dW1, db1) in zip(probs, refs):
        assert torch.equal(dW, dW1) and torch.equal(db, db1)


@pytest.mark.parametrize("rows", [1, 300, 4096, 70000])
def test_mlp_fused_forward(stg, gpu, rows):
    """stg_mlp_fwd (fc1 -> GELU -> fc2 in one kernel, the hidden tensor never in HBM; Swin_AVE.py:111-127) against (i) the two-GEMM path
    of the same library -- same operands, fp32 accumulation, bf16 hidden: agreement to summation-order / rounding-boundary level --
    and (ii) the fp32 PyTorch statement with erf GELU."""
    from stgcma import kernels as Kn
    C_ = 128
    assert Kn.mlp_fused_supported(C_)
    g = torch.Generator().manual_seed(rows)
    Y = torch.randn(rows, C_, generator=g).to(BF16)
    W1 = (torch.randn(4 * C_, C_, generator=g) / C_ ** 0.5).to(BF16)
    W2 = (torch.randn(C_, 4 * C_, generator=g) / (4 * C_) ** 0.5).to(BF16)
    b1, b2 = torch.randn(4 * C_, generator=g) * 0.2, torch.randn(C_, generator=g) * 0.2
    perm = Kn.mlp_w2_perm(4 * C_, gpu)
    assert sorted(perm.tolist()) == list(range(4 * C_))
    Yg, W1g, W2g, b1g, b2g = (t.to(gpu) for t in (Y, W1, W2, b1, b2))
    out = Kn.mlp_fwd(Yg, W1g, b1g, W2g[:, perm].contiguous(), b2g)
    two = Kn.gemm_nt(Kn.gemm_nt(Yg, W1g, b1g, act=Kn.ACT_GELU), W2g, b2g)
    ref = torch.nn.functional.gelu(Y.float() @ W1.float().t() + b1) @ W2.float().t() + b2
    scale = float(ref.abs().max())
    assert float((out.float() - two.float()).abs().max()) <= 8e-3 * scale           # a bf16 ulp of the output where the hidden rounds differently
    assert float((out.float() - two.float()).norm() / two.float().norm()) <= 2e-4
    assert float((out.float().cpu() - ref).abs().max()) <= 1e-2 * max(1.0, scale)
    assert float((out.float().cpu() - ref).norm() / ref.norm()) <= 4e-3


@pytest.mark.parametrize("rows", [1, 130, 4096, 70000])
def test_mlp_fused_backward(stg, gpu, rows):
    """stg_mlp_bwd (dY of the frozen MLP with the pre-activation recomputed in-kernel) against the library's own two-GEMM backward
    (saved bf16 derivative) and against fp32 autograd of fc2(gelu(fc1(y)))."""
    from stgcma import kernels as Kn
    C_ = 128
    g = torch.Generator().manual_seed(rows + 5)
    Y = torch.randn(rows, C_, generator=g).to(BF16)
    dM = torch.randn(rows, C_, generator=g).to(BF16)
    W1 = (torch.randn(4 * C_, C_, generator=g) / C_ ** 0.5).to(BF16)
    W2 = (torch.randn(C_, 4 * C_, generator=g) / (4 * C_) ** 0.5).to(BF16)
    b1 = torch.randn(4 * C_, generator=g) * 0.2
    Yg, dMg, W1g, W2g, b1g = (t.to(gpu) for t in (Y, dM, W1, W2, b1))
    W2T = W2g.t().contiguous()
    out = Kn.mlp_bwd(Yg, dMg, W1g, b1g, W2T)
    _, Z = Kn.gemm_nt(Yg, W1g, b1g, act=Kn.ACT_GELU, want_dact=True)
    two = Kn.gemm_nt(Kn.gemm_nt(dMg, W2T, dact_src=Z), W1g.t().contiguous())
    y32 = Y.float().requires_grad_(True)
    (torch.nn.functional.gelu(y32 @ W1.float().t() + b1) @ W2.float().t() * dM.float()).sum().backward()
    ref = y32.grad
    scale = float(ref.abs().max())
    assert float((out.float() - two.float()).norm() / two.float().norm()) <= 4e-3         # the two-GEMM path rounds GELU' and dZ to bf16 too
    assert float((out.float().cpu() - ref).abs().max()) <= 1.5e-2 * max(1.0, scale)
    assert float((out.float().cpu() - ref).norm() / ref.norm()) <= 6e-3


@pytest.mark.parametrize("C,J,S,M2", [(512, 32, 64 * 37, 64 * 37), (128, 16, 16 * 5, 16 * 9 + 3), (256, 64, 4096, 1000),
                                      (768, 96, 16 * 123, 16 * 123), (192, 96, 3136, 3136), (384, 96, 784, 800), (768, 48, 16 * 13, 700)])
@pytest.mark.parametrize("with_rs", [False, True])
def test_pair_launches_equal_two_single_launches(stg, gpu, C, J, S, M2, with_rs):
    """stg_up_ln_fwd_pair / stg_ln_bwd_down_pair (both modalities' adapters in one launch, round 4) against the two single launches they
    replace: every output bit for bit (x, y, mean, rstd; dx, dh), fp32-residual and x-hat forms, with and without DropPath row scales."""
    from stgcma import kernels as K
    torch.manual_seed(C + J + with_rs)
    M = S + M2
    bf = lambda *s, sc=1.0: (torch.randn(*s, device=gpu) * sc).to(torch.bfloat16)
    h = bf(M, J)
    w1, w2 = bf(C, J, sc=0.1), bf(C, J, sc=0.1)
    b1, b2 = torch.randn(C, device=gpu) * 0.1, torch.randn(C, device=gpu) * 0.1
    res32, res16 = torch.randn(M, C, device=gpu), bf(M, C)
    gamma, beta = torch.rand(C, device=gpu) + 0.5, torch.randn(C, device=gpu) * 0.1
    inner = 16
    rs1 = (torch.rand((S + inner - 1) // inner * 1, device=gpu) + 0.5) if with_rs else None
    rs2 = (torch.rand((M2 + inner - 1) // inner * 1, device=gpu) + 0.5) if with_rs else None
    rkw = dict(rs_outer=inner, rs_inner=1) if with_rs else {}
    sl = [slice(0, S), slice(S, M)]
    for r16 in (res16, None):
        x0, y0 = torch.empty(M, C, device=gpu), torch.empty(M, C, device=gpu, dtype=torch.bfloat16)
        m0, r0 = torch.empty(M, device=gpu), torch.empty(M, device=gpu)
        for i, (w, b, rs) in enumerate(((w1, b1, rs1), (w2, b2, rs2))):
            K.up_ln_fwd(h[sl[i]], w, b, res32[sl[i]], gamma, beta, res16=None if r16 is None else r16[sl[i]], out=x0[sl[i]], y_out=y0[sl[i]],
                        mean_out=m0[sl[i]], rstd_out=r0[sl[i]], row_scale=rs, **rkw)
        x1, y1 = torch.full_like(x0, float("nan")), torch.full_like(y0, float("nan"))
        m1, r1 = torch.full_like(m0, float("nan")), torch.full_like(r0, float("nan"))
        K.up_ln_fwd_pair(h[sl[0]], h[sl[1]], w1, w2, b1, b2, res32, gamma, beta, res16=r16, out=x1, y_out=y1, mean_out=m1, rstd_out=r1,
                         row_scale=rs1, row_scale2=rs2, **rkw)
        assert torch.equal(x0, x1) and torch.equal(y0.view(torch.int16), y1.view(torch.int16)) and torch.equal(m0, m1) and torch.equal(r0, r1)
    if not K.ln_bwd_down_supported(C, J):
        return
    dy = bf(M, C)
    wt1, wt2 = w1.t().contiguous(), w2.t().contiguous()
    for xh in (True, False):
        X = y0 if xh else x0
        dx0 = torch.empty(M, C, device=gpu, dtype=torch.bfloat16)
        dh0 = []
        for i, (wt, rs) in enumerate(((wt1, rs1), (wt2, rs2))):
            if xh:
                dh0.append(K.ln_bwd_down_xhat(dy[sl[i]], X[sl[i]], r0[sl[i]], wt, add_to=res16[sl[i]], dx_out=dx0[sl[i]], row_scale=rs, **rkw)[1])
            else:
                dh0.append(K.ln_bwd_down(dy[sl[i]], X[sl[i]], gamma, m0[sl[i]], r0[sl[i]], wt, add_to=res16[sl[i]], dx_out=dx0[sl[i]], row_scale=rs, **rkw)[1])
        dx1 = torch.full_like(dx0, float("nan"))
        _, dh1 = K.ln_bwd_down_pair(dy, X, None if xh else gamma, None if xh else m0, r0, wt1, wt2, S, add_to=res16, dx_out=dx1,
                                    row_scale=rs1, row_scale2=rs2, **rkw)
        assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16)), f"dx differs (xhat={xh})"
        assert torch.equal(torch.cat(dh0).view(torch.int16), dh1.view(torch.int16)), f"dh differs (xhat={xh})"


@pytest.mark.parametrize("M,N,Kd,gelu8,kernel", [(31360, 1024, 4096, False, "gemm_nt_8ph_kernel"), (7840, 512, 2048, False, "gemm_nt_8ph_kernel"),
                                                 (62880, 1536, 512, False, "gemm_nt_8phm_kernel"), (125600, 2048, 512, True, "gemm_nt_8phm_kernel"),
                                                 (125600, 2048, 512, "d8", "gemm_nt_8phm_kernel"), (62880, 2048, 512, "d8", "gemm_nt_8phm_kernel"),
                                                 # round 6, the NX forms at partial-panel shapes: 3 tiles with a 64-column last tile + a zero k-tile; 5 tiles with a
                                                 # 128-column last tile; GELU + byte derivative over K = 192; the one-tile kernel with 192 valid columns / 9 k-tiles
                                                 (98464, 576, 192, False, "gemm_nt_8phm_kernel"), (98464, 1152, 384, False, "gemm_nt_8phm_kernel"),
                                                 (98464, 768, 192, True, "gemm_nt_8phm_kernel"), (31520, 192, 576, False, "gemm_nt_8ph_kernel"),
                                                 (31520, 384, 1152, False, "gemm_nt_8ph_kernel")])
def test_gemm_8phase_launches_are_reproducible(stg, gpu, M, N, Kd, gelu8, kernel):
    """Regression test of a race found in round 4: the 8-phase kernel pre-read the next k-tile's A0 fragments one phase BEFORE the counted
    wait that retires their LDS-DMA (it relied on "issued a k-tile ago"); when a DMA was slow, one k-tile of some rows was computed from the
    previous contents of the ring slot -- 1 launch in ~4 000 at 31 360 x 1024 x 4096 (the partial last row panel's timing provokes it), a NaN
    every few hundred training steps at batch 2.  The same launch must give the same bits every time.  Round 5: the MULTI-tile kernel
    (gemm_nt_8phm_kernel, same fix by analogy) gets the same number of launches at partial-panel shapes with 3 (qkv at half batch + 160 rows)
    and 4 (fc1 + GELU + byte derivative, full batch + 160 rows) column tiles per workgroup; the test asserts which kernel the dispatch chose."""
    from stgcma import kernels as K
    from stgcma._lib import ACT_GELU
    torch.manual_seed(0)
    nx = N % 256 != 0 or Kd % 128 != 0
    if nx:
        stg.configure(lib_gemm_nx=2)    # route every legal NX shape to the 8-phase kernels (the default takes only N < 256, K >= 768)
    try:
        _gemm_8phase_repro(gpu, M, N, Kd, gelu8, kernel, K, ACT_GELU)
    finally:
        if nx:
            stg.configure(lib_gemm_nx=1)


def _gemm_8phase_repro(gpu, M, N, Kd, gelu8, kernel, K, ACT_GELU):
    A = (torch.randn(M, Kd, device=gpu) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, Kd, device=gpu) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=gpu) * 0.1
    if gelu8 == "d8":                       # the fc2 dgrad's class (round 5: multi-tile kernel): x byte derivative, an ordinary global load in the epilogue
        d8 = torch.randint(0, 256, (M, N), device=gpu, dtype=torch.uint8)
        run = lambda: (K.gemm_nt(A, W, dact_src=d8),)
        want = K.gemm_nt(A, W).float()      # against the plain product times the decoded derivative: the epilogue is the ONLY difference
        got = run()[0].float()
        dec = d8.float() * 0.005 - 0.14     # the byte code of the saved derivative (STG_U8_LIN)
        assert float((got - want * dec).abs().max()) <= 2e-2 * float(want.abs().max())
    else:
        run = (lambda: K.gemm_nt(A, W, b, act=ACT_GELU, want_dact="u8")) if gelu8 else (lambda: (K.gemm_nt(A, W, b),))
    ref = [t.clone() for t in run()]
    assert K.LAST_GEMM_KERNEL.startswith(kernel), K.LAST_GEMM_KERNEL
    bad = torch.zeros((), device=gpu, dtype=torch.int64)
    for _ in range(12000 if M > 20000 and Kd > 512 else 6000):
        for t, q in zip(run(), ref):
            bad += (t.view(torch.int16) != q.view(torch.int16)).any() if t.dtype == torch.bfloat16 else (t != q).any()
    assert int(bad) == 0, f"{int(bad)} launches differ from the first one"


def test_hot_kernels_are_reproducible(stg, gpu):
    """The same launch must give the same bits every time (no atomics in these kernels): GEMM classes with partial row panels on every kernel
    the dispatch uses, the pair joins, window attention backward -- a few hundred launches each with another kernel in between.  A cheap net
    for the class of bug the 8-phase race was (an operand read that no wait covers)."""
    from stgcma import kernels as K, ops
    from stgcma._lib import ACT_GELU
    import oracle.swin as OS
    torch.manual_seed(1)
    bf = lambda *s, sc=1.0: (torch.randn(*s, device=gpu) * sc).to(torch.bfloat16)

    def same(a, b):
        return bool((a.view(torch.int16) == b.view(torch.int16)).all()) if a.dtype == torch.bfloat16 else torch.equal(a, b)

    def stress(name, fn, reps=400):
        ref = [t.clone() for t in fn()]
        junk = torch.randn(2048, 32, device=gpu)
        bad = torch.zeros((), device=gpu, dtype=torch.int64)
        for r in range(reps):
            if r % 3 == 0:
                junk = junk * 1.0001
            for t, q in zip(fn(), ref):
                bad += (t.view(torch.int16) != q.view(torch.int16)).any() if t.dtype == torch.bfloat16 else (t != q).any()
        assert int(bad) == 0, f"{name}: {int(bad)} of {reps} launches differ from the first"

    M = 7840                                                       # 30 full 256-row panels + 160 rows; 61 full 128-row panels + 32
    A5, A20 = bf(M, 512, sc=0.5), bf(M, 2048, sc=0.5)
    for N, Kd, A in ((1536, 512, A5), (512, 512, A5), (512, 2048, A20)):
        W, b = bf(N, Kd, sc=0.05), torch.randn(N, device=gpu) * 0.1
        stress(f"gemm {M}x{N}x{Kd}", lambda: (K.gemm_nt(A, W, b),))
    W1, b1 = bf(2048, 512, sc=0.05), torch.randn(2048, device=gpu) * 0.1
    stress("gemm fc1 + gelu + d8", lambda: K.gemm_nt(A5, W1, b1, act=ACT_GELU, want_dact="u8"))
    S = M // 2
    h, wa, wb = bf(M, 32), bf(512, 32, sc=0.1), bf(512, 32, sc=0.1)
    ba, bb = torch.randn(512, device=gpu) * 0.1, torch.randn(512, device=gpu) * 0.1
    res32, res16 = torch.randn(M, 512, device=gpu), bf(M, 512)
    gamma, beta = torch.rand(512, device=gpu) + 0.5, torch.randn(512, device=gpu) * 0.1
    x, y = torch.empty(M, 512, device=gpu), torch.empty(M, 512, device=gpu, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=gpu), torch.empty(M, device=gpu)
    stress("up_ln_fwd_pair", lambda: K.up_ln_fwd_pair(h[:S], h[S:], wa, wb, ba, bb, res32, gamma, beta, res16=res16, out=x, y_out=y, mean_out=mean, rstd_out=rstd))
    dy, dx = bf(M, 512), torch.empty(M, 512, device=gpu, dtype=torch.bfloat16)
    stress("ln_bwd_down_pair", lambda: K.ln_bwd_down_pair(dy, y, None, None, rstd, wa.t().contiguous(), wb.t().contiguous(), S, add_to=res16, dx_out=dx))
    images, heads, Himg, ws = 40, 16, 14, 7
    n, N_, C = 49, Himg * Himg, heads * 32
    qkv, dO = bf(images * N_, 3 * C), bf(images * N_, C)
    bm, bmT = K.winattn_table((torch.randn(169, heads) * 0.5).to(gpu), OS.relative_position_index(ws).reshape(-1).to(gpu), ops.shift_mask(Himg, Himg, ws, 3).to(gpu), n)
    wg = K.WinGeom(images, heads, Himg, Himg, ws, 3, 32 ** -0.5, bm, bmT)
    Q, Kk, V = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    O, lse = K.winattn_fwd(wg, Q, Kk, V)
    dq = torch.empty_like(qkv)
    stress("winattn_fwd", lambda: (K.winattn_fwd(wg, Q, Kk, V)[0],))
    stress("winattn_bwd", lambda: (K.winattn_bwd(wg, Q, Kk, V, O, lse, dO, dQ=dq[:, :C], dK=dq[:, C:2 * C], dV=dq[:, 2 * C:]) and dq,))
    # round 5's new kernels: the merged cross-modal backward (+ join), the two-tiles-per-trip mha kernels, the double-buffered conv wgrad
    P_, nv, na, D = 3, 197, 130, 16
    hv, ha, dv, da = bf(P_ * nv, D, sc=0.7), bf(P_ * na, D, sc=0.7), bf(P_ * nv, D), bf(P_ * na, D)
    gv = K.AttnGeom(P_, 1, nv, D, G=1, outer=nv, n_kv=na, outer_kv=na, scale=1.0)
    ga = K.AttnGeom(P_, 1, na, D, G=1, outer=na, n_kv=nv, outer_kv=nv, scale=1.0)
    (Rv, Lv), (Ra, La) = K.attn_fwd2(gv, hv, ha, ha, ga, ha, hv, hv)
    stress("xattn_pair_bwd", lambda: K.xattn_pair_bwd((gv, hv, ha, Rv, Lv, dv), (ga, ha, hv, Ra, La, da)), reps=300)
    mg = K.MhaGeom(6, 8, 197, 96, 96 ** -0.5)
    qkv2, dO2 = bf(6 * 197, 3 * 768, sc=0.5), bf(6 * 197, 768)
    Q2, K2, V2 = qkv2[:, :768], qkv2[:, 768:1536], qkv2[:, 1536:]
    O2, l2 = K.mha_fwd(mg, Q2, K2, V2)
    dqkv2 = torch.empty_like(qkv2)
    stress("mha_fwd", lambda: (K.mha_fwd(mg, Q2, K2, V2)[0],), reps=300)
    stress("mha_bwd", lambda: (K.mha_bwd(mg, Q2, K2, V2, O2, l2, dO2, dQ=dqkv2[:, :768], dK=dqkv2[:, 768:1536], dV=dqkv2[:, 1536:]) and dqkv2,), reps=300)
    xc, dyc = bf(3 * 28 * 28, 64), bf(3 * 28 * 28, 256)
    stress("conv3x3_wgrad", lambda: (K.conv3x3_wgrad(dyc, xc, 3, 28, 28, 3),), reps=200)


@pytest.mark.parametrize("R,C", [(1037, 256), (4099, 128), (33, 64), (517, 24), (260, 2056), (7, 256)])
def test_batchnorm_passes_ragged_rows(stg, gpu, R, C):
    """The decoder's BatchNorm passes (column sums, apply, backward) on row counts that are no multiple of the row lanes / the 4-row
    trips of the 16-byte forms (C % 8 == 0, C / 8 dividing 256) and on widths that take the scalar forms (24, 2056)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(R + C)
    x = _bf(torch.randn(R, C, generator=g) * 1.5 + 0.7)
    dy = _bf(torch.randn(R, C, generator=g))
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xf, df = x.float(), dy.float()
    s = k.bn_colsum(x.to(gpu))
    _close(s[0], xf.sum(0), tol=2e-3, what="sum x")
    mean = (s[0] / R).contiguous()
    s2 = k.bn_colsum(x.to(gpu), mean=mean)
    mc = mean.cpu()
    _close(s2[1], ((xf - mc) ** 2).sum(0), tol=2e-3, what="centred sum of squares")
    rstd = torch.rsqrt(s2[1] / R + 1e-5).contiguous()
    rc = rstd.cpu()
    y = k.bn_apply(x.to(gpu), mean, rstd, gam.to(gpu), bet.to(gpu))
    _close(y, (xf - mc) * rc * gam + bet, what="bn apply")
    sb = k.bn_colsum(x.to(gpu), dy.to(gpu), mean, rstd)
    xhat = (xf - mc) * rc
    _close(sb[0], df.sum(0), tol=2e-3, what="sum dy")
    _close(sb[1] / math.sqrt(R), (df * xhat).sum(0) / math.sqrt(R), tol=4e-3, what="sum dy xhat")
    sbc = sb.cpu()
    dx = k.bn_bwd(x.to(gpu), dy.to(gpu), mean, rstd, gam.to(gpu), sb)
    _close(dx, gam * rc * (df - sbc[0] / R - xhat * sbc[1] / R), what="bn bwd (training)")
    dxe = k.bn_bwd(x.to(gpu), dy.to(gpu), mean, rstd, gam.to(gpu), None)
    _close(dxe, gam * rc * df, what="bn bwd (eval)")


@pytest.mark.parametrize("F,H,W,C,align", [(3, 7, 9, 16, True), (2, 14, 14, 256, False), (11, 5, 3, 8, True), (1, 1, 1, 8, False), (9, 2, 28, 128, False)])
def test_bilinear_up2_rows_spread_over_xcds(stg, gpu, F, H, W, C, align):
    """x2 bilinear resize and its adjoint against torch's interpolate; F * H row counts that are no multiple of 8 exercise the
    XCD-banded row mapping's ragged last band."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(F * 131 + H * 17 + W + C)
    x = _bf(torch.randn(F * H * W, C, generator=g))
    xi = x.float().view(F, H, W, C).permute(0, 3, 1, 2).requires_grad_(True)
    ref = torch.nn.functional.interpolate(xi, scale_factor=2, mode="bilinear", align_corners=align)
    y = k.bilinear_up2_fwd(x.to(gpu), F, H, W, align)
    _close(y, ref.permute(0, 2, 3, 1).reshape(F * 4 * H * W, C), what="bilinear fwd")
    dy = _bf(torch.randn(F * 4 * H * W, C, generator=g))
    ref.backward(dy.float().view(F, 2 * H, 2 * W, C).permute(0, 3, 1, 2))
    dx = k.bilinear_up2_bwd(dy.to(gpu), F, H, W, align)
    _close(dx, xi.grad.permute(0, 2, 3, 1).reshape(F * H * W, C), tol=1.5e-2, what="bilinear bwd")


def test_counted_wait_and_path_switch_kernels_repeat_2000(stg, gpu):
    """>= 2 000 launches, same bits every time, of the kernels with counted waits / a wave-uniform path switch that had only a
    few hundred in-suite repetitions -- xattn_bwdm_kernel<16 / 32> on BOTH of its paths (a frame whose log-sum-exps span > 120 binary orders takes
    the two-exponential path), the mha pair launches (mha_fwd2 / dq2 / dkv2, grid.z = 2 P, K == V, 3 136-token and 49-token window frames),
    conv_wgrad_kernel<64, 2> (inline-asm transposing reads behind the LDS-DMA) -- and of round 6's: the window-attention backward with the
    LDS-shared table (a workgroup barrier in front of wave-private work) and the rewritten wgrad_ws chunk loop (two chunks in flight)."""
    from stgcma import kernels as K, ops
    import oracle.swin as OS
    torch.manual_seed(3)
    bf = lambda *s, sc=1.0: (torch.randn(*s, device=gpu) * sc).to(torch.bfloat16)

    def stress(name, fn, reps=2000):
        ref = [t.clone() for t in fn()]
        bad = torch.zeros((), device=gpu, dtype=torch.int64)
        for r in range(reps):
            for t, q in zip(fn(), ref):
                bad += (t.view(torch.int16) != q.view(torch.int16)).any() if t.dtype == torch.bfloat16 else (t != q).any()
        assert int(bad) == 0, f"{name}: {int(bad)} of {reps} launches differ from the first"
        for t in ref:
            assert torch.isfinite(t.float()).all(), name

    for D, nv, na in ((16, 197, 130), (32, 196, 196)):
        P_ = 3
        hv, ha = (torch.randn(P_ * nv, D, device=gpu) * 0.7), (torch.randn(P_ * na, D, device=gpu) * 0.7)
        hv.view(P_, nv, D)[0, ::7] *= 40.0                          # frame 0: the slow (two-exponential) path; frames 1, 2: the fast path
        hv, ha = hv.to(torch.bfloat16), ha.to(torch.bfloat16)
        dv, da = bf(P_ * nv, D), bf(P_ * na, D)
        gv = K.AttnGeom(P_, 1, nv, D, G=1, outer=nv, n_kv=na, outer_kv=na, scale=1.0)
        ga = K.AttnGeom(P_, 1, na, D, G=1, outer=na, n_kv=nv, outer_kv=nv, scale=1.0)
        (Rv, Lv), (Ra, La) = K.attn_fwd2(gv, hv, ha, ha, ga, ha, hv, hv)
        pv, pa = (gv, hv, ha, Rv, Lv, dv), (ga, ha, hv, Ra, La, da)
        assert K.xattn_pair_bwd_supported(pv, pa)
        stress(f"xattn_pair_bwd D={D}", lambda: K.xattn_pair_bwd(pv, pa))
    for mg, rows in ((K.MhaGeom(2, 1, 3136, 96, 1.0), 2 * 3136), (K.MhaGeom(2 * 4, 1, 49, 96, 1.0, window=(14, 14, 7, 3)), 2 * 196)):
        hq, hk, d0, d1 = bf(rows, 96, sc=0.3), bf(rows, 96, sc=0.3), bf(rows, 96), bf(rows, 96)
        (r0, l0), (r1, l1) = K.mha_fwd_pair(mg, (hq, hk, hk), (hk, hq, hq))
        reps = 2000 if rows < 1000 else 2000
        stress(f"mha_fwd_pair n={mg.n}", lambda: [t for o in K.mha_fwd_pair(mg, (hq, hk, hk), (hk, hq, hq)) for t in o], reps=reps)
        g0, g1, g2, g3 = (torch.empty_like(hq) for _ in range(4))

        def bwd():
            K.mha_bwd_pair(mg, (hq, hk, hk, r0, l0, d0, g0, g1, None), (hk, hq, hq, r1, l1, d1, g2, g3, None))
            return g0, g1, g2, g3
        stress(f"mha_bwd_pair n={mg.n}", bwd, reps=reps)
        if mg.window[2] == 0:                                        # round 6: the merged pass (double-buffered staging + statistics behind one barrier per trip)
            stress(f"mha_bwd_pair_merged n={mg.n}", lambda: K.mha_bwd_pair_merged(mg, (hq, hk, r0, l0, d0), (hk, hq, r1, l1, d1)), reps=reps)
    xg = K.XsGeom(5, 197, 49, 48, 1.0)                                # round 6: the ViT pair on small frames (one barrier, wave-private task lists)
    xv_, xa_, dv_, da_ = bf(5 * 197, 48, sc=0.6), bf(5 * 49, 48, sc=0.6), bf(5 * 197, 48), bf(5 * 49, 48)
    (ov_, lv_), (oa_, la_) = K.xsmall_fwd(xg, xv_, xa_)
    stress("xsmall_fwd", lambda: [t for o in K.xsmall_fwd(xg, xv_, xa_) for t in o])
    stress("xsmall_bwd", lambda: K.xsmall_bwd(xg, xv_, xa_, ov_, oa_, lv_, la_, dv_, da_))
    xc, dyc = bf(3 * 28 * 28, 64), bf(3 * 28 * 28, 256)
    stress("conv3x3_wgrad 64 -> 256", lambda: (K.conv3x3_wgrad(dyc, xc, 3, 28, 28, 3),))
    xc2, dyc2 = bf(2 * 14 * 14, 320), bf(2 * 14 * 14, 256)
    stress("conv3x3_wgrad 320 -> 256", lambda: (K.conv3x3_wgrad(dyc2, xc2, 2, 14, 14, 6),))
    # round 6
    images, heads, Himg, ws = 37, 16, 14, 7                          # 37 frames: the last frame quad of the LT form has one live wave
    n, N_, C = 49, Himg * Himg, heads * 32
    qkv, dO = bf(images * N_, 3 * C), bf(images * N_, C)
    bm, bmT = K.winattn_table((torch.randn(169, heads) * 0.5).to(gpu), OS.relative_position_index(ws).reshape(-1).to(gpu), ops.shift_mask(Himg, Himg, ws, 3).to(gpu), n)
    wg = K.WinGeom(images, heads, Himg, Himg, ws, 3, 32 ** -0.5, bm, bmT)
    Q, Kk, V = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    O, lse = K.winattn_fwd(wg, Q, Kk, V)
    dq = torch.empty_like(qkv)
    stress("winattn_bwd (LDS-shared table)", lambda: (K.winattn_bwd(wg, Q, Kk, V, O, lse, dO, dQ=dq[:, :C], dK=dq[:, C:2 * C], dV=dq[:, 2 * C:]) and dq,))
    for J, Cw in ((48, 768), (96, 768), (32, 512)):
        M = 4096 + 8 * 32 * 9 + 19                                   # whole chunks + a ragged last chunk in every row split
        a, b = bf(M, J), bf(M, Cw)
        rs = torch.rand(M // 64 + 1, device=gpu) + 0.5

        def wg_():
            dW, db = torch.zeros(J, Cw, device=gpu), torch.zeros(J, device=gpu)
            K.wgrad_tn(a, b, dW, db, row_scale=rs, rs_outer=64, rs_inner=1)
            return dW, db
        stress(f"wgrad_ws {J} x {Cw}", wg_, reps=1000)


def test_gated_pair_launches_repeat_2000(stg, gpu):
    """Round 6b: the cross-modal pairs with their gates inside (stg_winattn_pair_fwd / _bwd: an LDS ticket elects the wave that sends the workgroup's
    dgate sum; stg_xattn_pair_bwd_gate: the preparation kernel's workgroup sum) -- 2 000 launches each: every tensor output the same bits every time
    (dgate is a sum of fp32 atomics: its order may differ, its value must not drift beyond fp32 summation noise), a last workgroup of three waves."""
    from stgcma import kernels as K
    torch.manual_seed(5)
    bf = lambda *s, sc=1.0: (torch.randn(*s, device=gpu) * sc).to(torch.bfloat16)
    gate_v, gate_a = torch.tensor([0.37], device=gpu), torch.tensor([-1.21], device=gpu)

    def stress(name, fn, ndg, reps=2000):
        out = fn()
        ref, dg0 = [t.clone() for t in out[:-ndg]], [float(t) for t in out[-ndg:]]
        bad = torch.zeros((), device=gpu, dtype=torch.int64)
        dmax = torch.zeros(ndg, device=gpu)
        for _ in range(reps):
            out = fn()
            for t, q in zip(out[:-ndg], ref):
                bad += (t.view(torch.int16) != q.view(torch.int16)).any() if t.dtype == torch.bfloat16 else (t != q).any()
            dmax = torch.maximum(dmax, torch.stack([(t.reshape(()) - d).abs() for t, d in zip(out[-ndg:], dg0)]))
        assert int(bad) == 0, f"{name}: {int(bad)} of {reps} launches differ from the first"
        for d, m in zip(dg0, dmax.tolist()):
            assert m <= 1e-4 * max(1.0, abs(d)), f"{name}: dgate drifts by {m} around {d}"

    for images, Himg, D in ((3, 21, 32), (6, 14, 16)):
        hv, ha, dxv, dxa = bf(images * Himg * Himg, D, sc=0.7), bf(images * Himg * Himg, D, sc=0.7), bf(images * Himg * Himg, D), bf(images * Himg * Himg, D)
        wg = K.WinGeom(images, 1, Himg, Himg, 7, 3, 1.0, None, None, D=D)
        (rv, lv, xv), (ra, la, xa) = K.winattn_pair_fwd(wg, hv, ha, gate_v, gate_a)
        # (lse is [P, 1, 64] with the entries past the window's 49 tokens never written)
        stress(f"winattn_pair_fwd {Himg} D={D}", lambda: [t if t.dim() == 2 else t[..., :49].contiguous() for o in K.winattn_pair_fwd(wg, hv, ha, gate_v, gate_a) for t in o]
               + [torch.zeros((), device=gpu)], 1)

        def bwd():
            dgv, dga = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
            return list(K.winattn_pair_bwd(wg, hv, ha, rv, ra, lv, la, dxv, dxa, gate_v, gate_a, dgv, dga)) + [dgv, dga]
        stress(f"winattn_pair_bwd {Himg} D={D}", bwd, 2)
    for D, n in ((16, 3136), (32, 196)):
        P_ = 2 if n > 1000 else 5
        hv, ha, dxv, dxa = bf(P_ * n, D, sc=0.7), bf(P_ * n, D, sc=0.7), bf(P_ * n, D), bf(P_ * n, D)
        gv = K.AttnGeom(P_, 1, n, D, G=1, outer=n, n_kv=n, outer_kv=n, scale=1.0)
        (rv, lv, _), (ra, la, _) = K.xattn_fwd2_gate(gv, hv, ha, gv, ha, hv, gate_v, gate_a)

        def xb():
            dgv, dga = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
            return list(K.xattn_pair_bwd((gv, hv, ha, rv, lv, dxv), (gv, ha, hv, ra, la, dxa), gates=(gate_v, gate_a, dgv, dga))) + [dgv, dga]
        stress(f"xattn_pair_bwd gated D={D} n={n}", xb, 2, reps=2000 if n < 1000 else 600)
