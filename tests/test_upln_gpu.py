"""-m gpu: the fused adapter up-projection + residual join + LayerNorm kernels (csrc/upln.hip), through the C ABI, against a
plain fp32 PyTorch-CPU statement of the same chain (the GEMM-epilogue + stg_layernorm pair they replace is the second check)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def _ref_fwd(h, w, b, r32, r16, rs, gamma, beta, eps=1e-5):
    up = h.float() @ w.float().t() + b
    if rs is not None:
        up = up * rs[:, None]
    x = r32 + up + (r16.float() if r16 is not None else 0.)
    y = torch.nn.functional.layer_norm(x, (x.shape[1],), gamma, beta, eps)
    mu = x.mean(1)
    rstd = (x.var(1, unbiased=False) + eps).rsqrt()
    return x, y, mu, rstd


@pytest.mark.parametrize("M,C,K,r16,rs", [(64, 128, 16, True, False), (1000, 128, 16, False, True), (777, 256, 32, True, False),
                                          (4099, 512, 32, True, False), (2048, 512, 32, False, True), (513, 512, 64, True, True),
                                          (15, 256, 24, True, False), (1, 128, 8, False, False), (320, 512, 48, True, False),
                                          (1970, 768, 48, True, False), (490, 768, 48, False, True), (33, 768, 64, True, True)])
def test_up_ln_fwd(stg, gpu, M, C, K, r16, rs):
    from stgcma import kernels as k
    assert k.up_ln_supported(C, K)
    g = torch.Generator().manual_seed(M + C + K)
    h = torch.randn(M, K, generator=g).to(BF16)
    w = (torch.randn(C, K, generator=g) * 0.2).to(BF16)
    b = torch.randn(C, generator=g) * 0.1
    r32 = torch.randn(M, C, generator=g) * 2 + 0.5
    r16t = torch.randn(M, C, generator=g).to(BF16) if r16 else None
    T, N = 5, 7
    rst = None
    if rs:
        B = -(-M // (T * N))
        rst = (torch.rand(B * N, generator=g) < 0.8).float() / 0.8
        rows = torch.arange(M)
        rs_rows = rst[(rows // (T * N)) * N + rows % N]
    gamma = torch.randn(C, generator=g) * 0.3 + 1
    beta = torch.randn(C, generator=g) * 0.1
    hq = h if not rs else (h.float() * rs_rows[:, None]).to(BF16)       # the kernel scales the bf16 h row (documented)
    xr, yr, mur, rsr = _ref_fwd(hq, w, b, r32, r16t, None, gamma, beta)
    if rs:                                                              # bias is scaled in fp32
        xr = xr + (rs_rows[:, None] - 1) * b
        yr = torch.nn.functional.layer_norm(xr, (C,), gamma, beta, 1e-5)
        mur, rsr = xr.mean(1), (xr.var(1, unbiased=False) + 1e-5).rsqrt()
    d = lambda t: None if t is None else t.to(gpu)
    x, y, mu, rstd = k.up_ln_fwd(d(h), d(w), d(b), d(r32), d(gamma), d(beta), res16=d(r16t), row_scale=d(rst), rs_outer=T * N, rs_inner=N)
    torch.cuda.synchronize()
    assert torch.allclose(x.cpu(), xr, rtol=1e-5, atol=2e-5), float((x.cpu() - xr).abs().max())
    assert torch.allclose(mu.cpu(), mur, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rstd.cpu(), rsr, rtol=1e-4, atol=1e-6)
    err = (y.float().cpu() - yr).abs()
    assert float((err / yr.abs().clamp(min=1.0)).max()) <= 1e-2, float(err.max())
    # the two-kernel path it replaces agrees to fp32 round-off (x) / one bf16 ulp (y)
    x2 = k.gemm_nt(d(h), d(w), d(b), out_dtype=F32, res1=d(r16t) if r16 else d(r32), res2=d(r32) if r16 else None,
                   row_scale=d(rst), rs_outer=T * N, rs_inner=N)
    if not rs:
        assert torch.allclose(x, x2, rtol=1e-5, atol=2e-5)


def test_up_ln_unsupported_raises(stg, gpu):
    from stgcma import kernels as k
    assert not k.up_ln_supported(1024, 128)
    assert not k.up_ln_supported(1536, 96)
    h = torch.zeros(16, 96, dtype=BF16, device=gpu)
    w = torch.zeros(1536, 96, dtype=BF16, device=gpu)
    z = torch.zeros(1536, device=gpu)
    with pytest.raises(RuntimeError):
        k.up_ln_fwd(h, w, z, torch.zeros(16, 1536, device=gpu), z, z)


@pytest.mark.parametrize("M,C,J,add,rs", [(64, 128, 16, True, False), (1000, 128, 16, False, True), (777, 256, 32, True, False),
                                          (4099, 512, 32, True, False), (2048, 512, 32, False, True), (513, 512, 64, True, True),
                                          (15, 256, 64, True, False), (1, 128, 16, False, False), (490, 256, 16, True, False)])
def test_ln_bwd_down(stg, gpu, M, C, J, add, rs):
    from stgcma import kernels as k
    assert k.ln_bwd_down_supported(C, J)
    g = torch.Generator().manual_seed(3 * M + C + J)
    x = torch.randn(M, C, generator=g) * 2 + 0.3
    dy = torch.randn(M, C, generator=g).to(BF16)
    gamma = torch.randn(C, generator=g) * 0.3 + 1
    addt = torch.randn(M, C, generator=g).to(BF16) if add else None
    wt = (torch.randn(J, C, generator=g) * 0.1).to(BF16)
    T, N = 5, 7
    rst = rs_rows = None
    if rs:
        rst = (torch.rand(-(-M // (T * N)) * N, generator=g) < 0.8).float() / 0.8
        rows = torch.arange(M)
        rs_rows = rst[(rows // (T * N)) * N + rows % N]
    xr = x.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (C,), gamma, torch.zeros(C), 1e-5)
    y.backward(dy.float())
    dx_ref = xr.grad + (addt.float() if add else 0.)
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    d = lambda t: None if t is None else t.to(gpu)
    dx, dh = k.ln_bwd_down(d(dy), d(x), d(gamma), d(mean), d(rstd), d(wt), add_to=d(addt), row_scale=d(rst), rs_outer=T * N, rs_inner=N)
    torch.cuda.synchronize()
    err = (dx.float().cpu() - dx_ref).abs()
    assert float((err / dx_ref.abs().clamp(min=1.0)).max()) <= 1e-2, float(err.max())
    # the pair it replaces: bit-identical dx is not required (reduction order), dh is checked against the bf16 dx it was made from
    dx2 = k.layernorm_bwd(d(dy), d(x), d(gamma), d(mean), d(rstd), add_to=d(addt))
    assert float((dx.float() - dx2.float()).abs().max()) <= 2e-2 * float(dx_ref.abs().max())
    dh_ref = dx.float().cpu() @ wt.float().t()
    if rs:
        dh_ref = dh_ref * rs_rows[:, None]
    errh = (dh.float().cpu() - dh_ref).abs()
    assert float((errh / dh_ref.abs().clamp(min=1.0)).max()) <= 1e-2, float(errh.max())
