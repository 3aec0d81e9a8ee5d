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
                                          (1970, 768, 48, True, False), (490, 768, 48, False, True), (33, 768, 64, True, True),
                                          # round 5, the NT = 12 family (Swin-L: d_h = 96 at C = 192 / 384 / 768; K = 96 is three k-steps)
                                          (3136, 192, 96, True, False), (1000, 192, 96, False, True), (785, 384, 96, True, True), (64, 384, 48, False, False),
                                          (1961, 768, 96, True, False), (500, 768, 96, False, True), (17, 768, 96, True, True), (1, 192, 96, True, False)])
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
    assert not k.up_ln_supported(512, 96) and not k.up_ln_supported(768, 128) and not k.ln_bwd_down_supported(768, 64)
    h = torch.zeros(16, 96, dtype=BF16, device=gpu)
    w = torch.zeros(1536, 96, dtype=BF16, device=gpu)
    z = torch.zeros(1536, device=gpu)
    with pytest.raises(RuntimeError):
        k.up_ln_fwd(h, w, z, torch.zeros(16, 1536, device=gpu), z, z)


@pytest.mark.parametrize("M,C,J,add,rs", [(64, 128, 16, True, False), (1000, 128, 16, False, True), (777, 256, 32, True, False),
                                          (4099, 512, 32, True, False), (2048, 512, 32, False, True), (513, 512, 64, True, True),
                                          (15, 256, 64, True, False), (1, 128, 16, False, False), (490, 256, 16, True, False),
                                          # round 5, the NT = 12 family: Swin-L (J = 96), CLIP ViT-B (C = 768, J = 48)
                                          (3136, 192, 96, True, False), (1000, 192, 96, False, True), (785, 384, 96, True, True), (64, 384, 48, False, False),
                                          (1961, 768, 96, True, False), (500, 768, 96, False, True), (1970, 768, 48, True, True), (17, 768, 48, False, False),
                                          (1, 192, 96, True, False)])
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


@pytest.mark.parametrize("M,C,J,dys", [(777, 512, 32, 1.0), (3136, 192, 96, 1.0), (785, 384, 96, 1.0), (1961, 768, 96, 1.0), (1970, 768, 48, 1.0), (33, 768, 96, 1.0),
                                       (12544, 192, 48, 1e-3), (3136, 384, 48, 1e-4)])
def test_ln_bwd_down_xhat_equals_the_fp32_row_form(stg, gpu, M, C, J, dys):
    """stg_ln_bwd_down_xhat (the default product path: bf16 normalised rows, gamma == 1) against the same kernel fed the fp32 rows those
    x_hat came from: dx to the bf16 rounding of x_hat, dh against the bf16 dx it was made from."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(M + C + J)
    x = (torch.randn(M, C, generator=g) * 2 + 0.3).to(gpu)
    dy = (torch.randn(M, C, generator=g) * dys).to(BF16).to(gpu)           # dys: gradient magnitudes as small as a deep model's (all bounds relative)
    addt = (torch.randn(M, C, generator=g) * dys).to(BF16).to(gpu)
    wt = (torch.randn(J, C, generator=g) * 0.1).to(BF16).to(gpu)
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    xhat = ((x - mean[:, None]) * rstd[:, None]).to(BF16)
    dx0, dh0 = k.ln_bwd_down(dy, x, torch.ones(C, device=gpu), mean, rstd, wt, add_to=addt)
    dx1, dh1 = k.ln_bwd_down_xhat(dy, xhat, rstd, wt, add_to=addt)
    torch.cuda.synchronize()
    sc = float(dx0.float().abs().max())
    assert float((dx0.float() - dx1.float()).abs().max()) <= 2e-2 * sc
    dh_ref = dx1.float() @ wt.float().t()
    assert float((dh1.float() - dh_ref).abs().max()) <= 1e-2 * float(dh_ref.abs().max())
    assert float((dh1.float() - dh_ref).norm()) <= 4e-3 * float(dh_ref.norm())                  # bf16 rounding of dh: 2^-9 per element
    assert float((dh0.float() - dh1.float()).norm()) <= 2e-2 * float(dh_ref.norm())            # the two forms' dx differ by x_hat's bf16 rounding
