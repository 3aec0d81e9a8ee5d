"""-m gpu: the whole-window attention kernels (stg_winattn_table / _fwd / _bwd) against an fp32 PyTorch-CPU statement of
WindowAttention's spatial branch with roll + window_partition as addressing (Swin_AVE.py:256-276, :727-740, :765-776).
Tolerance 1e-2 (bf16 operands, fp32 accumulate), relative to max(1, |ref|)."""
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


def _case(gpu, *, images, heads, Himg, ws, shift, seed=0, mag=1.0):
    from stgcma import kernels as k, ops
    import oracle.swin as OS
    g = torch.Generator().manual_seed(seed)
    n, N, C = ws * ws, Himg * Himg, heads * 32
    nW = (Himg // ws) ** 2
    rows = images * N
    qkv = (torch.randn(rows, 3 * C, generator=g) * mag).to(BF16)
    dO = torch.randn(rows, C, generator=g).to(BF16)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    index = OS.relative_position_index(ws)
    mask = ops.shift_mask(Himg, Himg, ws, shift) if shift > 0 else None
    scale = 32 ** -0.5

    # ---- fp32 reference on the CPU (autograd)
    x = qkv.float().requires_grad_(True)
    wmap = ops.window_token_map(Himg, Himg, ws, shift).long()
    rq = (torch.arange(images)[:, None] * N + wmap[None, :]).reshape(images * nW, n)            # [P, n] rows
    def split(j):
        return x[:, j * C:(j + 1) * C][rq].view(-1, n, heads, 32).permute(0, 2, 1, 3)
    s = scale * (split(0) @ split(1).transpose(-1, -2)) + table[index.reshape(-1)].view(n, n, heads).permute(2, 0, 1)[None]
    if mask is not None:
        s = s + mask.repeat(images, 1, 1)[:, None]
    o = (torch.softmax(s, -1) @ split(2)).permute(0, 2, 1, 3).reshape(-1, C)
    o_nat = torch.zeros(rows, C).index_add(0, rq.reshape(-1), o)
    lse_ref = torch.logsumexp(s, -1)
    (o_nat * dO.float()).sum().backward()

    # ---- HIP
    d = lambda t: None if t is None else t.to(gpu)
    bm, bmT = k.winattn_table(d(table), d(index.reshape(-1)), d(mask), n)
    untile = lambda t: t.view(t.shape[0], t.shape[1], 16, 64, 4).permute(0, 1, 3, 2, 4).reshape(t.shape[0], t.shape[1], 64, 64)
    plain = untile(bm)                                      # [g][h][q][k]; stored tiled as [(k >> 2)][q][k & 3]
    assert torch.equal(plain.transpose(-1, -2), untile(bmT))
    assert float(plain[..., :n, n:].max()) < -1e29 if n < 64 else True
    want = 1.4426950408889634 * (table[index.reshape(-1)].view(n, n, heads).permute(2, 0, 1)[None] +
                                 (mask[:, None] if mask is not None else 0))
    assert float((plain[..., :n, :n].cpu() - want).abs().max()) < 1e-5
    wg = k.WinGeom(images, heads, Himg, Himg, ws, shift, scale, bm, bmT)
    Q = d(qkv)
    O, lse = k.winattn_fwd(wg, Q[:, :C], Q[:, C:2 * C], Q[:, 2 * C:])
    _close(O.float() / mag, o_nat / mag, what="O")          # |V| ~ mag: the bf16 rounding of P and O scales with it
    _close(lse[..., :n], lse_ref, tol=2e-2, what="lse")
    dQKV = torch.full_like(Q, float("nan"))
    k.winattn_bwd(wg, Q[:, :C], Q[:, C:2 * C], Q[:, 2 * C:], O, lse, d(dO), dQ=dQKV[:, :C], dK=dQKV[:, C:2 * C], dV=dQKV[:, 2 * C:])
    gscale = max(1.0, float(x.grad.abs().max()))
    _close(dQKV[:, :C] / gscale, x.grad[:, :C] / gscale, tol=1.5e-2, what="dQ")
    _close(dQKV[:, C:2 * C] / gscale, x.grad[:, C:2 * C] / gscale, tol=1.5e-2, what="dK")
    _close(dQKV[:, 2 * C:] / gscale, x.grad[:, 2 * C:] / gscale, tol=1.5e-2, what="dV")

    # ---- and against the generic gather-mapped kernels (same inputs, both HIP)
    sb = k.bias_gather(d(table), d(index.reshape(-1)))
    ag = k.AttnGeom(images * nW, heads, n, 32, G=nW, outer=N, window=(Himg, Himg, ws, shift), scale=scale, bias=sb,
                    bias_div=images * nW, bias_mod=1, mask=d(mask))
    O2, lse2 = k.attn_fwd(ag, Q[:, :C], Q[:, C:2 * C], Q[:, 2 * C:])
    _close(O.float() / mag, O2.float() / mag, what="O vs generic")


@pytest.mark.parametrize("shift", [0, 3])
def test_window_7x7_stage1_like(stg, gpu, shift):
    """Swin stage-1 geometry: 56 x 56 tokens, 64 windows of 49 tokens (15 padding keys / queries per tile)."""
    _case(gpu, images=2, heads=4, Himg=56, ws=7, shift=shift, seed=1)


def test_window_single_window_stage4(stg, gpu):
    """Stage 4: the 7 x 7 image is one window (window_size clamps, shift 0; Swin_AVE.py:354-358)."""
    _case(gpu, images=5, heads=32, Himg=7, ws=7, shift=0, seed=2)


def test_window_stage3_shifted_large_scores(stg, gpu):
    """14 x 14, 4 windows, shifted, |q k| large enough that softmax saturates in places."""
    _case(gpu, images=3, heads=16, Himg=14, ws=7, shift=3, seed=3, mag=3.0)


def test_window_full_tile_8x8(stg, gpu):
    """n = 64 exactly: no padding lanes at all."""
    _case(gpu, images=2, heads=2, Himg=16, ws=8, shift=4, seed=4)


def test_window_small_4x4(stg, gpu):
    """n = 16: the second 32-row tile is all padding."""
    _case(gpu, images=3, heads=1, Himg=8, ws=4, shift=1, seed=5)


def test_winattn_rejects_bad_geometry(stg, gpu):
    from stgcma import kernels as k
    bm = torch.zeros(1, 2, 64, 64, device=gpu)
    with pytest.raises(RuntimeError):
        k.WinGeom(1, 2, 14, 14, 5, 0, 1.0, bm, bm)                      # 14 % 5
    with pytest.raises(RuntimeError):
        k.WinGeom(1, 2, 27, 27, 9, 0, 1.0, bm, bm)                      # 81 tokens
    wg = k.WinGeom(1, 2, 14, 14, 7, 0, 1.0, bm, bm)
    q = torch.zeros(14 * 14 - 1, 192, dtype=BF16, device=gpu)
    with pytest.raises(RuntimeError):
        k.winattn_fwd(wg, q[:, :64], q[:, 64:128], q[:, 128:])          # too few rows


def test_window_partial_workgroups(stg, gpu):
    """(window, head) problems not a multiple of the four waves of a workgroup: 9 problems, and a single one."""
    _case(gpu, images=3, heads=3, Himg=7, ws=7, shift=0, seed=6)
    _case(gpu, images=1, heads=1, Himg=7, ws=7, shift=0, seed=7)


def test_window_shared_kv_single_head_sums_dk_dv(stg, gpu):
    """The cross-modal use (H = 1, K = V = the other modality's hidden states, zero bias, scale 1): dV = None makes the backward
    write dK + dV into dK -- the same numbers as the two separate gradients added in fp32, up to one bf16 rounding."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(11)
    images, Himg, ws, shift = 3, 14, 7, 3
    n, N = ws * ws, Himg * Himg
    q = torch.randn(images * N, 32, generator=g).to(BF16).to(gpu)
    kv = torch.randn(images * N, 32, generator=g).to(BF16).to(gpu)
    dO = torch.randn(images * N, 32, generator=g).to(BF16).to(gpu)
    bm, bmT = k.winattn_table(torch.zeros(((2 * ws - 1) ** 2, 1), device=gpu), torch.zeros(n * n, dtype=torch.int64, device=gpu), None, n)
    wg = k.WinGeom(images, 1, Himg, Himg, ws, shift, 1.0, bm, bmT)
    O, lse = k.winattn_fwd(wg, q, kv, kv)
    dQ, dK, dV, dQ2, dKV = (torch.full_like(q, float("nan")) for _ in range(5))
    k.winattn_bwd(wg, q, kv, kv, O, lse, dO, dQ=dQ, dK=dK, dV=dV)
    k.winattn_bwd(wg, q, kv, kv, O, lse, dO, dQ=dQ2, dK=dKV, dV=None)
    assert torch.equal(dQ, dQ2)
    want = dK.float() + dV.float()
    assert torch.isfinite(dKV.float()).all()
    scale = float(want.abs().max())
    assert float((dKV.float() - want).abs().max()) <= 1.2e-2 * scale
    # and against the generic gather-mapped kernels on the same problem
    ag = k.AttnGeom(images * wg.G, 1, n, 32, G=wg.G, outer=N, n_kv=n, outer_kv=N, scale=1.0, window=(Himg, Himg, ws, shift))
    O2, lse2 = k.attn_fwd(ag, q, kv, kv)
    _close(O, O2.float(), what="O vs generic")
    dq_g, dkv_g, _ = k.attn_bwd(ag, q, kv, kv, O2, lse2, dO, shared_kv=True)
    _close(dQ2.float() / scale, dq_g.float() / scale, tol=1.5e-2, what="dQ vs generic")
    _close(dKV.float() / scale, dkv_g.float() / scale, tol=1.5e-2, what="dK + dV vs generic")


def _window_index(F, Hi, Wi, ws, shift):
    """rows of the [F * Hi * Wi] token tensor in window order: [F * nW, ws * ws] (torch.roll by -shift, then window_partition:
    Swin_AVE.py:151-165 / :262-266 -- window (wi, wj) token (ti, tj) sits at ((wi ws + ti + shift) % Hi, (wj ws + tj + shift) % Wi))."""
    f, wi, wj, ti, tj = torch.meshgrid(torch.arange(F), torch.arange(Hi // ws), torch.arange(Wi // ws), torch.arange(ws), torch.arange(ws),
                                       indexing="ij")
    y = (wi * ws + ti + shift) % Hi
    x = (wj * ws + tj + shift) % Wi
    return (f * Hi * Wi + y * Wi + x).reshape(F * (Hi // ws) * (Wi // ws), ws * ws)


@pytest.mark.parametrize("images,Himg,shift", [(3, 14, 3), (5, 28, 0), (2, 56, 3), (1, 7, 0)])
def test_window_cross_modal_table_free_and_16_wide(stg, gpu, images, Himg, shift):
    """Round 6: the window-level cross-modal pair without its dummy table (bm = bmT = None: the kernels synthesise the padding keys' -1e30) and at
    head dim 16 (Swin-B stage 0; 32-byte rows staged beside a zero line).  (i) D = 32 table-free == the all-zero-table run, bit for bit;
    (ii) D = 16 == the same problem zero-padded to 32 columns on the D = 32 kernels, bit for bit (the padded columns multiply zeros), nothing
    written beside the 16 columns; (iii) D = 16 against the fp32 softmax(q kv^T) kv of every window and its autograd gradients."""
    from stgcma import kernels as k
    ws = 7
    n, N = ws * ws, Himg * Himg
    g = torch.Generator().manual_seed(40 + Himg + shift)
    eq = lambda a, b: torch.equal(a.view(torch.int16), b.view(torch.int16))
    # (i)
    q32 = torch.randn(images * N, 32, generator=g).to(BF16).to(gpu)
    kv32 = torch.randn(images * N, 32, generator=g).to(BF16).to(gpu)
    dO32 = torch.randn(images * N, 32, generator=g).to(BF16).to(gpu)
    bm, bmT = k.winattn_table(torch.zeros(((2 * ws - 1) ** 2, 1), device=gpu), torch.zeros(n * n, dtype=torch.int64, device=gpu), None, n)
    wz = k.WinGeom(images, 1, Himg, Himg, ws, shift, 1.0, bm, bmT)
    wf = k.WinGeom(images, 1, Himg, Himg, ws, shift, 1.0, None, None)
    O0, l0 = k.winattn_fwd(wz, q32, kv32, kv32)
    O1, l1 = k.winattn_fwd(wf, q32, kv32, kv32)
    assert eq(O0, O1) and torch.equal(l0[..., :n], l1[..., :n])
    a0, b0, a1, b1 = (torch.full_like(q32, float("nan")) for _ in range(4))
    k.winattn_bwd(wz, q32, kv32, kv32, O0, l0, dO32, dQ=a0, dK=b0, dV=None)
    k.winattn_bwd(wf, q32, kv32, kv32, O1, l1, dO32, dQ=a1, dK=b1, dV=None)
    assert eq(a0, a1) and eq(b0, b1)
    # (ii) + (iii)
    q16, kv16, dO16 = (torch.randn(images * N, 16, generator=g) * 0.7).to(BF16), (torch.randn(images * N, 16, generator=g) * 0.7).to(BF16), torch.randn(images * N, 16, generator=g).to(BF16)
    pad = lambda t: torch.nn.functional.pad(t, (0, 16)).contiguous()
    w16 = k.WinGeom(images, 1, Himg, Himg, ws, shift, 1.0, None, None, D=16)
    wide = torch.full((images * N, 24), float("nan"), dtype=BF16, device=gpu)          # the 16-wide output is a column slice: columns 16.. must stay untouched
    O16, l16 = k.winattn_fwd(w16, q16.to(gpu), kv16.to(gpu), kv16.to(gpu), out=wide[:, :16])
    Op, lp = k.winattn_fwd(wf, pad(q16).to(gpu), pad(kv16).to(gpu), pad(kv16).to(gpu))
    assert eq(O16.contiguous(), Op[:, :16].contiguous()) and torch.equal(l16[..., :n], lp[..., :n])
    assert torch.isnan(wide[:, 16:].float()).all() and float(Op[:, 16:].float().abs().max()) == 0.0
    gq = torch.full((images * N, 24), float("nan"), dtype=BF16, device=gpu)
    gk = torch.full((images * N, 24), float("nan"), dtype=BF16, device=gpu)
    k.winattn_bwd(w16, q16.to(gpu), kv16.to(gpu), kv16.to(gpu), O16, l16, dO16.to(gpu), dQ=gq[:, :16], dK=gk[:, :16], dV=None)
    pq, pk = torch.full((images * N, 32), float("nan"), dtype=BF16, device=gpu), torch.full((images * N, 32), float("nan"), dtype=BF16, device=gpu)
    k.winattn_bwd(wf, pad(q16).to(gpu), pad(kv16).to(gpu), pad(kv16).to(gpu), Op, lp, pad(dO16).to(gpu), dQ=pq, dK=pk, dV=None)
    assert eq(gq[:, :16].contiguous(), pq[:, :16].contiguous()) and eq(gk[:, :16].contiguous(), pk[:, :16].contiguous())
    assert torch.isnan(gq[:, 16:].float()).all() and torch.isnan(gk[:, 16:].float()).all()
    idx = _window_index(images, Himg, Himg, ws, shift)                                 # [images * nW, n] rows in window order
    xq, xk = q16.float().requires_grad_(True), kv16.float().requires_grad_(True)
    S = torch.einsum("wid,wjd->wij", xq[idx], xk[idx])
    R = torch.softmax(S, -1) @ xk[idx]
    out = torch.zeros(images * N, 16).index_put((idx.reshape(-1),), R.reshape(-1, 16))
    _close(O16, out, what="O (16 wide)")
    (out * dO16.float()).sum().backward()
    sc = float(max(xq.grad.abs().max(), xk.grad.abs().max()))
    _close(gq[:, :16].float() / sc, xq.grad / sc, tol=1.5e-2, what="dQ (16 wide)")
    _close(gk[:, :16].float() / sc, xk.grad / sc, tol=1.5e-2, what="dK + dV (16 wide)")


@pytest.mark.parametrize("images,Himg,shift,D", [(3, 14, 3, 32), (5, 28, 0, 32), (2, 56, 3, 16), (1, 7, 0, 16), (9, 14, 0, 16), (3, 21, 3, 32), (2, 21, 0, 16)])   # 21 x 21: 9 windows per image, a last workgroup of 3 / 2 waves
def test_window_cross_modal_pair_one_launch_with_gates(stg, gpu, images, Himg, shift, D):
    """Round 6b (stg_winattn_pair_fwd / _bwd, ABI 220): both directions of the window-level cross-modal pair and its gates in one launch per pass.
    Forward: r, lse and h' = h + gate r are BIT-IDENTICAL to two stg_winattn_fwd launches + stg_gate_fwd2.  Backward: the kernel takes d(h'), scales its
    outputs by the gate in fp32 (the two-launch path rounds gate * d(h') to bf16 first) and accumulates dgate from its own delta: compared with the
    two-launch path at bf16 resolution, and with the fp32 autograd of h + gate softmax(h hother^T) hother per window."""
    from stgcma import kernels as k
    ws = 7
    n, N = ws * ws, Himg * Himg
    g = torch.Generator().manual_seed(7 * Himg + shift + D)
    eq = lambda a, b: torch.equal(a.view(torch.int16), b.view(torch.int16))
    hv, ha = ((torch.randn(images * N, D, generator=g) * 0.7).to(BF16).to(gpu) for _ in range(2))
    dxv, dxa = (torch.randn(images * N, D, generator=g).to(BF16).to(gpu) for _ in range(2))
    gate_v, gate_a = torch.tensor([0.37], device=gpu), torch.tensor([-1.21], device=gpu)
    wg = k.WinGeom(images, 1, Himg, Himg, ws, shift, 1.0, None, None, D=D)
    # forward
    rv0, lv0 = k.winattn_fwd(wg, hv, ha, ha)
    ra0, la0 = k.winattn_fwd(wg, ha, hv, hv)
    xv0, xa0 = k.gate_fwd2(hv, rv0, gate_v, ha, ra0, gate_a)
    (rv1, lv1, xv1), (ra1, la1, xa1) = k.winattn_pair_fwd(wg, hv, ha, gate_v, gate_a)
    assert eq(rv0, rv1) and eq(ra0, ra1) and torch.equal(lv0[..., :n], lv1[..., :n]) and torch.equal(la0[..., :n], la1[..., :n])
    assert eq(xv0, xv1) and eq(xa0, xa1)
    # backward against the two-launch path
    dgv0, dga0 = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    drv, dra = k.gate_bwd2(dxv, rv0, gate_v, dgv0, dxa, ra0, gate_a, dga0)
    q_v0, kv_a0, q_a0, kv_v0 = (torch.empty_like(hv) for _ in range(4))
    k.winattn_bwd(wg, hv, ha, ha, rv0, lv0, drv, dQ=q_v0, dK=kv_a0, dV=None)
    k.winattn_bwd(wg, ha, hv, hv, ra0, la0, dra, dQ=q_a0, dK=kv_v0, dV=None)
    dgv1, dga1 = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    q_v1, kv_a1, q_a1, kv_v1 = k.winattn_pair_bwd(wg, hv, ha, rv1, ra1, lv1, la1, dxv, dxa, gate_v, gate_a, dgv1, dga1)
    for a, b, what in ((q_v0, q_v1, "dq_v"), (kv_a0, kv_a1, "dkv_a"), (q_a0, q_a1, "dq_a"), (kv_v0, kv_v1, "dkv_v")):
        sc = float(a.float().abs().max())
        _close(b.float() / sc, a.float() / sc, tol=1.0e-2, what=what + " vs the two-launch path")
    assert abs(float(dgv1) - float(dgv0)) <= 2e-3 * max(1.0, abs(float(dgv0))) and abs(float(dga1) - float(dga0)) <= 2e-3 * max(1.0, abs(float(dga0)))
    # ... and against fp32 autograd
    idx = _window_index(images, Himg, Himg, ws, shift).to("cpu")
    xv, xa = hv.float().cpu().requires_grad_(True), ha.float().cpu().requires_grad_(True)
    gv, ga = gate_v.cpu().clone().requires_grad_(True), gate_a.cpu().clone().requires_grad_(True)

    def direction(xq, xk):
        R = torch.softmax(torch.einsum("wid,wjd->wij", xq[idx], xk[idx]), -1) @ xk[idx]
        return torch.zeros(images * N, D).index_put((idx.reshape(-1),), R.reshape(-1, D))
    yv, ya = xv + gv * direction(xv, xa), xa + ga * direction(xa, xv)
    _close(xv1, yv.detach(), what="h_v'")
    _close(xa1, ya.detach(), what="h_a'")
    ((yv * dxv.float().cpu()).sum() + (ya * dxa.float().cpu()).sum()).backward()
    tot_v = dxv.float().cpu() + q_v1.float().cpu() + kv_v1.float().cpu()          # d h_v = d(h_v') + through direction v's queries + through direction a's keys / values
    tot_a = dxa.float().cpu() + q_a1.float().cpu() + kv_a1.float().cpu()
    sc = float(max(xv.grad.abs().max(), xa.grad.abs().max()))
    _close(tot_v / sc, xv.grad / sc, tol=1.5e-2, what="d h_v")
    _close(tot_a / sc, xa.grad / sc, tol=1.5e-2, what="d h_a")
    # dgate = <d(h'), r>: exact (fp32 summation order aside) on the kernel's own bf16-rounded r; against autograd the bf16 rounding of r is a random walk
    # over rows x D terms of relative size 2^-9
    for dg, dx, r, ref in ((dgv1, dxv, rv1, gv.grad), (dga1, dxa, ra1, ga.grad)):
        terms = dx.float() * r.float()
        assert abs(float(dg) - float(terms.sum())) <= 2e-4 * float(terms.abs().sum()) ** 0.5 + 1e-4 * abs(float(terms.sum())), (float(dg), float(terms.sum()))
        assert abs(float(dg) - float(ref)) <= 4 * 2.0 ** -9 * float((terms * terms).sum()) ** 0.5 + 1e-3 * abs(float(ref)), (float(dg), float(ref))
