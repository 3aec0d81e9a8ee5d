"""-m gpu: the fp8 (OCP e4m3, E8M0 block scales -- "MX") operand path of stg_gemm_nt (BASELINE config 5; no reference counterpart, so
the kernels are pinned to an emulation written from the format's definition):
  * stg_quant_fp8_mx is BIT-exact against a torch emulation (block exponent from max|x| / 448, round-to-nearest-even e4m3);
  * the scaled-MFMA GEMM equals the fp32 product of the DEQUANTISED operands to fp32 summation-order accuracy, every epilogue;
  * the quantisation error against the unquantised bf16 product is reported (and bounded loosely: e4m3 keeps 3 mantissa bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _emulate_quant(x):
    """x: bf16 CPU [R, K] (K % 8 == 0) -> (q uint8 [R, Kp], e uint8 [R, Kp / 32])."""
    R, K = x.shape
    Kp = (K + 127) // 128 * 128
    xf = torch.zeros((R, Kp), dtype=F32)
    xf[:, :K] = x.float()
    blk = xf.view(R, Kp // 32, 32)
    amax = blk.abs().amax(-1)
    t = (amax * np.float32(1.0 / 448.0)).to(F32)
    bits = t.view(torch.int32)
    e = ((bits >> 23) & 0xff) + ((bits & 0x7fffff) != 0).to(torch.int32)
    e = e.clamp(1, 254)
    e = torch.where(amax > 0, e, torch.full_like(e, 127))
    inv = torch.pow(torch.tensor(2.0, dtype=torch.float64), (127 - e).double()).float()
    q = (blk * inv[..., None]).to(torch.float8_e4m3fn).view(torch.uint8).view(R, Kp)
    return q, e.to(torch.uint8)


def _unpack_scales(s, R, KB):
    r = torch.arange(R).view(-1, 1)
    b = torch.arange(KB).view(1, -1)
    idx = ((r // 64) * KB + b) * 64 + (r % 16) * 4 + (r % 64) // 16
    return s.cpu()[idx.reshape(-1)].view(R, KB)


def _dequant(q, e):
    R, Kp = q.shape
    v = q.view(torch.float8_e4m3fn).float().view(R, Kp // 32, 32)
    return (v * torch.pow(torch.tensor(2.0, dtype=torch.float64), e.double() - 127.0).float()[..., None]).view(R, Kp)


@pytest.mark.parametrize("R,K", [(1, 8), (70, 192), (333, 768), (4096, 2048)])
def test_quant_fp8_bit_exact(stg, gpu, R, K):
    from stgcma import kernels as Kn
    g = torch.Generator().manual_seed(R * 7 + K)
    x = (torch.randn(R, K, generator=g) * torch.exp(torch.randn(R, 1, generator=g) * 2)).to(BF16)     # rows over several binades
    x[0, :8] = 0                                                          # an all-zero chunk inside a live block
    if R > 3:
        x[3] = 0                                                          # all-zero blocks: exponent 127, zero bytes
    f = Kn.quant_fp8(x.to(gpu))
    q_ref, e_ref = _emulate_quant(x)
    assert f.q.shape == q_ref.shape and f.rows == R and f.K == K
    assert torch.equal(f.q.cpu(), q_ref), f"{int((f.q.cpu() != q_ref).sum())} of {q_ref.numel()} bytes differ"
    assert torch.equal(_unpack_scales(f.s, R, q_ref.shape[1] // 32), e_ref)
    # pad rows of the scale table (rows up to the next multiple of 64) hold 127
    Rp, KB = (R + 63) // 64 * 64, q_ref.shape[1] // 32
    assert f.s.numel() == Rp * KB
    if Rp > R:
        full = _unpack_scales(f.s, Rp, KB)
        assert bool((full[R:] == 127).all())


@pytest.mark.parametrize("M,N,K,epi", [(300, 136, 192, "plain"), (512, 256, 768, "bias"), (1000, 2048, 384, "gelu"),
                                         (4100, 512, 1536, "res"), (129, 64, 128, "f32")])
def test_gemm_fp8_equals_dequantised_product(stg, gpu, M, N, K, epi):
    from stgcma import kernels as Kn
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(BF16)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF16)
    fa, fw = Kn.quant_fp8(A.to(gpu)), Kn.quant_fp8(W.to(gpu))
    qa, ea = _emulate_quant(A)
    qw, ew = _emulate_quant(W)
    ref = _dequant(qa, ea).double() @ _dequant(qw, ew).double().t()
    bias = torch.randn(N, generator=g) if epi in ("bias", "gelu", "res") else None
    kw = {}
    if bias is not None:
        ref = ref + bias.double()
    if epi == "gelu":
        kw = dict(act=Kn.ACT_GELU, want_dact=True)
        dref = 0.5 * (1 + torch.erf(ref / 2 ** 0.5)) + ref * torch.exp(-0.5 * ref * ref) / (2 * np.pi) ** 0.5
        ref = torch.nn.functional.gelu(ref)
    res = None
    if epi == "res":
        res = torch.randn(M, N, generator=g)
        ref = ref + res.double()
        kw = dict(res1=res.to(gpu), out_dtype=F32)
    if epi == "f32":
        kw = dict(out_dtype=F32)
    out = Kn.gemm_nt(fa, fw, None if bias is None else bias.to(gpu), **kw)
    dact = None
    if isinstance(out, tuple):
        out, dact = out
    got = out.float().cpu().double()
    scale = float(ref.abs().max())
    tol = 2e-5 if out.dtype == F32 else 6e-3                     # fp32 summation order / the bf16 rounding of the stored output
    err = float((got - ref).abs().max()) / scale
    assert err <= tol, f"{epi}: max err / scale = {err:.3e}"
    if dact is not None:
        assert float((dact.float().cpu().double() - dref).abs().max()) <= 1e-2
    # how far the block-scaled e4m3 operands sit from the unquantised product (information; e4m3 keeps 3 mantissa bits)
    exact = A.double() @ W.double().t()
    if bias is not None:
        exact = exact + bias.double()
    if epi == "gelu":
        exact = torch.nn.functional.gelu(exact)
    if res is not None:
        exact = exact + res.double()
    q_l2 = float((got - exact).norm() / exact.norm())
    print(f"fp8 {M}x{N}x{K} {epi}: kernel vs dequantised product {err:.2e}; vs unquantised product relL2 {q_l2:.3e}")
    assert q_l2 <= 8e-2
