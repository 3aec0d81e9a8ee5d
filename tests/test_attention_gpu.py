"""-m gpu: the gather-mapped flash attention kernels (fwd, dQ, dK/dV, dbias) against an fp32 PyTorch-CPU statement
of softmax(scale*QK^T + bias + mask)V, for every addressing mode the model uses."""
import math

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


def _rows(P, G, n, outer, amap):
    p = torch.arange(P)
    base = (p // G) * outer
    g = p % G
    i = torch.arange(n)
    idx = g[:, None] * n + i[None, :]
    off = amap.long()[idx] if amap is not None else idx
    return base[:, None] + off  # [P, n]


def _ref(Q, K, V, rq, rk, H, D, scale, bias, bias_div, bias_mod, mask, G):
    """fp32 reference with autograd.  Q,K,V: [rows, H*D] fp32 leaf tensors."""
    P = rq.shape[0]
    q = Q[rq].view(P, -1, H, D).permute(0, 2, 1, 3)
    k = K[rk].view(P, -1, H, D).permute(0, 2, 1, 3)
    v = V[rk].view(P, -1, H, D).permute(0, 2, 1, 3)
    s = scale * (q @ k.transpose(-1, -2))
    if bias is not None:
        bg = (torch.arange(P) // bias_div) % bias_mod
        s = s + bias[bg]
    if mask is not None:
        s = s + mask[torch.arange(P) % G][:, None]
    pr = torch.softmax(s, -1)
    o = (pr @ v).permute(0, 2, 1, 3).reshape(P, -1, H * D)
    lse = torch.logsumexp(s, -1)
    return o, lse


def _run_case(gpu, *, P, H, n, D, G=1, n_kv=None, mapped=False, scale=1.0, with_bias=False, bias_mod=1, with_mask=False,
              shared_kv=False, want_dbias=False, seed=0, mag=1.0, window=None, temporal=None):
    from stgcma import kernels as k, ops
    g = torch.Generator().manual_seed(seed)
    nk = n if n_kv is None else n_kv
    assert P % G == 0
    outer_q = G * n + (3 if mapped else 0)
    outer_k = G * nk + (3 if mapped else 0)
    if n_kv is None:
        outer_k = outer_q
    map_q = map_k = None
    if mapped:
        map_q = torch.randperm(outer_q, generator=g)[:G * n].to(torch.int32)
        map_k = map_q if n_kv is None else torch.randperm(outer_k, generator=g)[:G * nk].to(torch.int32)
    if window is not None:                      # arithmetic map, checked against the table form of the same map
        map_q = map_k = ops.window_token_map(*window)
        outer_q = outer_k = window[0] * window[1]
    if temporal is not None:
        map_q = map_k = ops.temporal_token_map(temporal, n)
        outer_q = outer_k = temporal * n
    rows_q = (P // G) * outer_q
    rows_k = (P // G) * outer_k
    C = H * D
    # fused qkv buffer with column slices, like the model
    if shared_kv:
        Qb = (torch.randn(rows_q, C, generator=g) * mag).to(BF16)
        KVb = (torch.randn(rows_k, C, generator=g) * mag).to(BF16)
        Q, K, V = Qb, KVb, KVb
    elif n_kv is None:
        qkv = (torch.randn(rows_q, 3 * C, generator=g) * mag).to(BF16)
        Q, K, V = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    else:
        Q = (torch.randn(rows_q, C, generator=g) * mag).to(BF16)
        K = (torch.randn(rows_k, C, generator=g) * mag).to(BF16)
        V = (torch.randn(rows_k, C, generator=g) * mag).to(BF16)
    bias = (torch.randn(bias_mod, H, n, nk, generator=g)) if with_bias else None
    bias_div = max(1, P // bias_mod)
    mask = None
    if with_mask:
        mask = torch.where(torch.rand(G, n, nk, generator=g) < 0.3, torch.tensor(-100.0), torch.tensor(0.0))
        mask[:, :, 0] = 0.0
    rq = _rows(P, G, n, outer_q, map_q)
    rk = _rows(P, G, nk, outer_k, map_k)

    Qr = Q.float().clone().requires_grad_(True)
    Kr = K.float().clone().requires_grad_(True)
    Vr = Kr if shared_kv else V.float().clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True) if bias is not None else None
    o_ref, lse_ref = _ref(Qr, Kr, Vr, rq, rk, H, D, scale, br, bias_div, bias_mod, mask, G)

    arith = window is not None or temporal is not None
    geom = k.AttnGeom(P, H, n, D, G=G, outer=outer_q, map_q=None if (map_q is None or arith) else map_q.to(gpu),
                      n_kv=n_kv, outer_kv=outer_k if n_kv is not None else None,
                      map_kv=None if (map_k is None or n_kv is None or arith) else map_k.to(gpu), scale=scale,
                      window=window, temporal=temporal,
                      bias=None if bias is None else bias.to(gpu), bias_div=bias_div, bias_mod=bias_mod,
                      mask=None if mask is None else mask.to(gpu))
    Qg, Kg = Q.to(gpu), K.to(gpu)
    if shared_kv:
        Vg = Kg
    elif n_kv is None:
        qg = qkv.to(gpu)
        Qg, Kg, Vg = qg[:, :C], qg[:, C:2 * C], qg[:, 2 * C:]
    else:
        Vg = V.to(gpu)
    O = torch.zeros(rows_q, C, dtype=BF16, device=gpu)
    O, lse = k.attn_fwd(geom, Qg, Kg, Vg, out=O)
    tag = f"P{P} H{H} n{n} nk{nk} D{D} G{G} mapped{mapped}"
    _close(O[rq.to(gpu)], o_ref, what=f"O {tag}")
    _close(lse, lse_ref, tol=2e-2, what=f"lse {tag}")

    dO_full = torch.zeros(rows_q, C)
    dO_full[rq.reshape(-1)] = torch.randn(P * n, C, generator=g)
    dO = dO_full.to(BF16)
    o_ref.backward(dO.float()[rq])
    dbias = torch.zeros_like(bias).to(gpu) if want_dbias else None
    dQ = torch.zeros(rows_q, C, dtype=BF16, device=gpu)
    dK = torch.zeros(rows_k, C, dtype=BF16, device=gpu)
    dV = None if shared_kv else torch.zeros(rows_k, C, dtype=BF16, device=gpu)
    k.attn_bwd(geom, Qg, Kg, Vg, O, lse, dO.to(gpu), dQ=dQ, dK=dK, dV=dV, shared_kv=shared_kv, dbias=dbias)
    gs = max(1.0, float(Qr.grad.abs().max()))
    _close(dQ / gs, Qr.grad / gs, tol=2e-2, what=f"dQ {tag}")
    ks = max(1.0, float(Kr.grad.abs().max()))
    _close(dK / ks, Kr.grad / ks, tol=2e-2, what=f"dK {tag}")
    if not shared_kv:
        vs = max(1.0, float(Vr.grad.abs().max()))
        _close(dV / vs, Vr.grad / vs, tol=2e-2, what=f"dV {tag}")
    if want_dbias:
        bs = max(1.0, float(br.grad.abs().max()))
        _close(dbias / bs, br.grad / bs, tol=2e-2, what=f"dbias {tag}")


def test_window_attention_shapes(stg, gpu):
    # W-MSA: 49-token windows, head dim 32, rel-pos bias, shift mask, roll/partition as a map (Swin_AVE.py:256-276)
    _run_case(gpu, P=8, H=4, n=49, D=32, G=4, mapped=True, scale=32 ** -0.5, with_bias=True, with_mask=True, seed=1)
    _run_case(gpu, P=6, H=2, n=49, D=32, G=1, mapped=False, scale=32 ** -0.5, with_bias=True, seed=2)


def test_arithmetic_maps_and_packing(stg, gpu):
    """map_kind 1 (shift + window partition) and 2 (temporal regrouping) computed in-kernel; short sequences packed 3 / 6
    per tile, including tile-groups that straddle the video/audio bias-table boundary and a ragged last group."""
    _run_case(gpu, P=8, H=4, n=49, D=32, G=4, scale=32 ** -0.5, with_bias=True, with_mask=True, seed=41, window=(14, 14, 7, 3))
    _run_case(gpu, P=18, H=2, n=49, D=32, G=9, scale=32 ** -0.5, with_bias=True, seed=42, window=(21, 21, 7, 0))
    _run_case(gpu, P=4, H=1, n=49, D=16, G=4, n_kv=49, shared_kv=True, seed=43, mag=0.7, window=(14, 14, 7, 3))
    _run_case(gpu, P=20, H=4, n=10, D=32, G=10, scale=32 ** -0.5, with_bias=True, bias_mod=2, want_dbias=True, seed=44, temporal=10)
    _run_case(gpu, P=200, H=2, n=10, D=32, G=50, scale=32 ** -0.5, with_bias=True, bias_mod=2, want_dbias=True, seed=45, temporal=50)
    _run_case(gpu, P=14, H=2, n=5, D=32, G=7, scale=32 ** -0.5, with_bias=True, bias_mod=2, want_dbias=True, seed=46, temporal=7)
    _run_case(gpu, P=7, H=2, n=16, D=64, G=7, scale=0.125, seed=47, temporal=7)


def test_temporal_attention_shapes(stg, gpu):
    # temporal: T=10 tokens, two bias groups (video / audio tables), dbias needed (Swin_AVE.py:244-255)
    _run_case(gpu, P=128, H=4, n=10, D=32, G=32, mapped=True, scale=32 ** -0.5, with_bias=True, bias_mod=2,
              want_dbias=True, seed=3)
    _run_case(gpu, P=6, H=2, n=5, D=32, G=3, mapped=True, scale=32 ** -0.5, with_bias=True, bias_mod=2,
              want_dbias=True, seed=4)


@pytest.mark.parametrize("D", [16, 32, 48, 64, 96, 128])
def test_cross_modal_head_dims(stg, gpu, D):
    # cross-modal adapter attention: single head, no scale, K = V = other modality (Swin_AVE.py:753-757,801-805)
    _run_case(gpu, P=3, H=1, n=49, D=D, G=1, n_kv=49, shared_kv=True, seed=10 + D, mag=0.5)


def test_cross_modal_global_long(stg, gpu):
    # frame-global variant: N = 196 / 784-ish lengths exercise the multi-tile online softmax
    _run_case(gpu, P=2, H=1, n=196, D=32, n_kv=196, shared_kv=True, seed=20, mag=0.7)
    _run_case(gpu, P=1, H=1, n=800, D=16, n_kv=800, shared_kv=True, seed=21, mag=0.7)
    _run_case(gpu, P=2, H=1, n=70, D=64, n_kv=33, shared_kv=True, seed=22, mag=0.5)


def test_cross_modal_global_fast_path(stg, gpu):
    """xattn.hip (H = 1, K == V, D in {16, 32}, n >= 64): stage-1 / stage-2 frame sizes, ragged query and key tails
    (n % 32, n_kv % 64 != 0), n != n_kv, a non-unit scale, and several problems per launch."""
    _run_case(gpu, P=2, H=1, n=3136, D=16, n_kv=3136, shared_kv=True, seed=50, mag=0.7)
    _run_case(gpu, P=2, H=1, n=784, D=32, n_kv=784, shared_kv=True, seed=51, mag=0.7)
    _run_case(gpu, P=3, H=1, n=197, D=32, n_kv=130, shared_kv=True, seed=52, mag=1.0, scale=0.5)
    _run_case(gpu, P=3, H=1, n=130, D=16, n_kv=197, shared_kv=True, seed=53, mag=1.0, scale=0.7)
    _run_case(gpu, P=5, H=1, n=64, D=16, n_kv=64, shared_kv=True, seed=54, mag=1.0)
    _run_case(gpu, P=2, H=1, n=97, D=32, n_kv=65, shared_kv=True, seed=55, mag=0.7)


def test_cross_modal_window_mapped(stg, gpu):
    _run_case(gpu, P=12, H=1, n=49, D=16, G=4, n_kv=49, mapped=True, shared_kv=True, seed=23, mag=0.7)


def test_vit_mha_shapes(stg, gpu):
    # CLIP ViT-B/16 with heads=8 => head dim 96, 197 tokens; temporal T=10 (CLIP_AVE.py:106-108)
    _run_case(gpu, P=2, H=8, n=197, D=96, scale=96 ** -0.5, seed=30)
    _run_case(gpu, P=5, H=8, n=10, D=96, G=5, mapped=True, scale=96 ** -0.5, seed=31)
    _run_case(gpu, P=2, H=4, n=49, D=64, scale=0.125, seed=32)


def test_online_softmax_rescale_spike(stg, gpu):
    """Force the running max to jump in a late KV tile (the rescale branch)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(5)
    n, D = 128, 32
    Q = (torch.randn(n, D, generator=g) * 0.3); KV = (torch.randn(n, D, generator=g) * 0.3)
    KV[100] = Q[7] * 40  # huge score for query 7 at key 100 (4th tile)
    Qb, KVb = Q.to(BF16), KV.to(BF16)
    s = Qb.float() @ KVb.float().t()
    ref = torch.softmax(s, -1) @ KVb.float()
    geom = k.AttnGeom(1, 1, n, D, n_kv=n)
    O, lse = k.attn_fwd(geom, Qb.to(gpu), KVb.to(gpu), KVb.to(gpu))
    _close(O, ref, what="spike O")
    _close(lse.view(-1), torch.logsumexp(s, -1), tol=2e-2, what="spike lse")
    kv = KVb.to(gpu)                                       # K is V (one tensor): the xattn.hip path and its lazy rescale
    O, lse = k.attn_fwd(geom, Qb.to(gpu), kv, kv)
    _close(O, ref, what="spike O (K is V)")
    _close(lse.view(-1), torch.logsumexp(s, -1), tol=2e-2, what="spike lse (K is V)")


def _pair_case(gpu, P, nv, na, D, scale, seed, mag, spike=0.0):
    """The frame-global cross-modal PAIR (Swin_AVE.py:796-811): r_v = softmax(s h_v h_a^T) h_a, r_a = softmax(s h_a h_v^T) h_v; backward from
    (d r_v, d r_a).  Returns fp32 autograd gradients, the four-pass path's (attn_bwd2) and the merged path's (xattn_pair_bwd).  spike > 0: a few
    rows of h_v are scaled up so that the frame's log-sum-exps span more than 120 binary orders (the merged kernel's slow path)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(seed)
    hv = (torch.randn(P * nv, D, generator=g) * mag)
    ha = (torch.randn(P * na, D, generator=g) * mag)
    if spike > 0:
        hv.view(P, nv, D)[0, ::7] *= spike                      # frame 0 only: the other frames stay on the fast path
    hv, ha = hv.to(BF16), ha.to(BF16)
    dv = torch.randn(P * nv, D, generator=g).to(BF16)
    da = torch.randn(P * na, D, generator=g).to(BF16)
    # fp32 reference
    xv, xa = hv.float().requires_grad_(True), ha.float().requires_grad_(True)
    S = scale * torch.einsum("pid,pjd->pij", xv.view(P, nv, D), xa.view(P, na, D))
    rv = torch.softmax(S, 2) @ xa.view(P, na, D)
    ra = torch.softmax(S.transpose(1, 2), 2) @ xv.view(P, nv, D)
    ((rv.reshape(-1, D) * dv.float()).sum() + (ra.reshape(-1, D) * da.float()).sum()).backward()
    # device
    gv = k.AttnGeom(P, 1, nv, D, G=1, outer=nv, n_kv=na, outer_kv=na, scale=scale)
    ga = k.AttnGeom(P, 1, na, D, G=1, outer=na, n_kv=nv, outer_kv=nv, scale=scale)
    d = lambda t: t.to(gpu)
    Hv, Ha, Dv, Da = d(hv), d(ha), d(dv), d(da)
    (Rv, Lv), (Ra, La) = k.attn_fwd2(gv, Hv, Ha, Ha, ga, Ha, Hv, Hv)
    pv, pa = (gv, Hv, Ha, Rv, Lv, Dv), (ga, Ha, Hv, Ra, La, Da)
    (dq_v, dkv_a), (dq_a, dkv_v) = k.attn_bwd2(pv, pa)
    four = (dq_v.float() + dkv_v.float(), dq_a.float() + dkv_a.float())
    assert k.xattn_pair_bwd_supported(pv, pa)
    Gv, Ga = k.xattn_pair_bwd(pv, pa)
    torch.cuda.synchronize()
    return (xv.grad, xa.grad), four, (Gv.float(), Ga.float())


@pytest.mark.parametrize("P,nv,na,D,scale,mag", [(2, 3136, 3136, 16, 1.0, 0.7), (3, 784, 784, 32, 1.0, 0.7), (5, 196, 196, 32, 1.0, 1.0),
                                                 (3, 197, 130, 32, 0.5, 1.0), (3, 130, 197, 16, 0.7, 1.0), (4, 64, 64, 16, 1.0, 1.0), (2, 97, 65, 32, 1.0, 0.7)])
def test_cross_modal_pair_merged_backward(stg, gpu, P, nv, na, D, scale, mag):
    """stg_xattn_pair_bwd (one pass per modality over the pair's shared score tiles, ONE exponential per score) against the fp32 autograd
    gradients of the pair and against the four passes it replaces (dQ + dK + dV summed in fp32): stage-0 / 1 / 2 frame sizes, ragged tails,
    n_v != n_a, non-unit scale."""
    ref, four, got = _pair_case(gpu, P, nv, na, D, scale, seed=60 + nv + D, mag=mag)
    for r, f, m, name in zip(ref, four, got, ("G_v", "G_a")):
        sc = float(r.abs().max())
        e_m = float((m.cpu() - r).abs().max()) / sc
        e_f = float((f.cpu() - r).abs().max()) / sc
        l_m = float((m.cpu() - r).norm() / r.norm())
        l_f = float((f.cpu() - r).norm() / r.norm())
        assert torch.isfinite(m).all(), name
        # the merged pass rounds P and dS to bf16 once per score like each of the four passes: same error level
        assert e_m <= max(2.5e-2, 2.5 * e_f) and l_m <= max(8e-3, 1.5 * l_f), f"{name}: merged max/scale {e_m:.2e} relL2 {l_m:.2e}; four-pass {e_f:.2e} {l_f:.2e}"


def test_cross_modal_pair_merged_backward_slow_path(stg, gpu):
    """Frame 0 carries rows whose scores are ~100x the others': its log-sum-exps span more than 120 binary orders, so the preparation kernel
    sends it down the merged kernel's two-exponential path (no power-of-two factor can over- or underflow there); the other frames stay on the
    one-exponential path.  Both must match the fp32 gradients."""
    from stgcma import kernels as k
    ref, four, got = _pair_case(gpu, 3, 196, 196, 32, 1.0, seed=77, mag=0.7, spike=40.0)
    for r, f, m, name in zip(ref, four, got, ("G_v", "G_a")):
        assert torch.isfinite(m).all(), name
        for fr in range(3):
            rows = slice(fr * 196, (fr + 1) * 196)
            sc = float(r[rows].abs().max())
            e_m = float((m.cpu()[rows] - r[rows]).abs().max()) / sc
            e_f = float((f.cpu()[rows] - r[rows]).abs().max()) / sc
            assert e_m <= max(2e-2, 2.0 * e_f), f"{name} frame {fr}: merged {e_m:.2e}, four-pass {e_f:.2e}"


@pytest.mark.parametrize("P,nv,na,D", [(3, 130, 197, 32), (2, 197, 130, 16), (2, 64, 640, 16)])
def test_cross_modal_pair_fused_gate_unequal_token_counts(stg, gpu, P, nv, na, D):
    """with n_v != n_a the pair launch's grid was sized from direction 0 alone and direction 1's extra query tiles never ran
    (O1 / X1 / lse1 kept torch.empty's bytes).  Both directions of stg_xattn_fwd2_gate against single-direction launches, in both orders of
    (smaller, larger); every output row must be written and finite."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(300 + nv + na)
    bf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF16).to(gpu)
    hv, ha = bf(P * nv, D, sc=0.7), bf(P * na, D, sc=0.7)
    gate_v, gate_a = torch.tensor([0.37], device=gpu), torch.tensor([-1.21], device=gpu)
    gv = k.AttnGeom(P, 1, nv, D, G=1, outer=nv, n_kv=na, outer_kv=na, scale=1.0)
    ga = k.AttnGeom(P, 1, na, D, G=1, outer=na, n_kv=nv, outer_kv=nv, scale=1.0)
    assert k.xattn_pair_fwd_supported(gv, hv, ha, ga, ha, hv)
    rv, lv = k.attn_fwd(gv, hv, ha, ha)
    ra, la = k.attn_fwd(ga, ha, hv, hv)
    # poison the allocator's free blocks so that an unwritten row cannot look right by accident
    for n_ in (nv, na):
        t = torch.full((P * n_, D), float("nan"), dtype=BF16, device=gpu); del t
        t = torch.full((P, 1, n_), float("nan"), dtype=torch.float32, device=gpu); del t
    (rv2, lv2, xv2), (ra2, la2, xa2) = k.xattn_fwd2_gate(gv, hv, ha, ga, ha, hv, gate_v, gate_a)
    torch.cuda.synchronize()
    eq = lambda a, b: torch.equal(a.view(torch.int16), b.view(torch.int16))
    for t in (rv2, ra2, xv2, xa2, lv2, la2):
        assert torch.isfinite(t.float()).all(), "a pair output has unwritten rows"
    assert eq(rv, rv2) and eq(ra, ra2) and torch.equal(lv, lv2) and torch.equal(la, la2)
    for h, r, gt, x in ((hv, rv, gate_v, xv2), (ha, ra, gate_a, xa2)):
        want = h.float() + float(gt) * r.float()
        assert float(((x.float() - want).abs() / want.abs().clamp_min(1e-3)).max()) <= 2.0 ** -7


@pytest.mark.parametrize("P,n,D", [(3, 196, 32), (2, 3136, 16), (2, 130, 16)])
def test_cross_modal_pair_fused_gate_and_join(stg, gpu, P, n, D):
    """Round 5: the frame-global pair's forward writes the gated hidden states itself (stg_xattn_fwd2_gate) and its merged backward applies the
    join (dX + G) * act' itself (stg_xattn_pair_bwd_join): against the launches they replace -- attn_fwd2 + gate_fwd2, xattn_pair_bwd + add3_mul2 --
    bit for bit (the gate's fma against gate_fwd2's multiply-add: at most one bf16 ulp)."""
    from stgcma import kernels as k
    g = torch.Generator().manual_seed(90 + n)
    bf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF16).to(gpu)
    hv, ha = bf(P * n, D, sc=0.7), bf(P * n, D, sc=0.7)
    gate_v, gate_a = torch.tensor([0.37], device=gpu), torch.tensor([-1.21], device=gpu)
    gv = k.AttnGeom(P, 1, n, D, G=1, outer=n, n_kv=n, outer_kv=n, scale=1.0)
    assert k.xattn_pair_fwd_supported(gv, hv, ha, gv, ha, hv)
    (rv, lv), (ra, la) = k.attn_fwd2(gv, hv, ha, ha, gv, ha, hv, hv)
    xv, xa = k.gate_fwd2(hv, rv, gate_v, ha, ra, gate_a)
    (rv2, lv2, xv2), (ra2, la2, xa2) = k.xattn_fwd2_gate(gv, hv, ha, gv, ha, hv, gate_v, gate_a)
    eq = lambda a, b: torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert eq(rv, rv2) and eq(ra, ra2) and torch.equal(lv, lv2) and torch.equal(la, la2)
    for a, b in ((xv, xv2), (xa, xa2)):
        d = (a.float() - b.float()).abs()
        assert float((d / a.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -7, "gated hidden state differs by more than one bf16 ulp"
    dxv, dxa, zv, za = bf(P * n, D), bf(P * n, D), bf(P * n, D, sc=0.5), bf(P * n, D, sc=0.5)
    dg_v, dg_a = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    drv, dra = k.gate_bwd2(dxv, rv, gate_v, dg_v, dxa, ra, gate_a, dg_a)
    pv, pa = (gv, hv, ha, rv, lv, drv), (gv, ha, hv, ra, la, dra)
    Gv, Ga = k.xattn_pair_bwd(pv, pa)
    j0 = k.add3_mul2(dxv, Gv, None, zv, dxa, Ga, None, za)
    buf = torch.full((2 * P * n, D), float("nan"), dtype=BF16, device=gpu)
    j1 = k.xattn_pair_bwd(pv, pa, join=(dxv, zv, dxa, za), outs=(buf[:P * n], buf[P * n:]))
    torch.cuda.synchronize()
    assert eq(j0[0], j1[0]) and eq(j0[1], j1[1]), "fused join differs from add3_mul2 on the merged kernel's output"
    # round 6b (stg_xattn_pair_bwd_gate, ABI 220): the gates inside the merged backward -- the kernels take d(x), round gate * d(x) as gate_bwd2 did
    # (G bit-identical) and sum dgate = <d(x), r> themselves (fp32 atomics: summation order differs)
    pxv, pxa = (gv, hv, ha, rv, lv, dxv), (gv, ha, hv, ra, la, dxa)
    dg2_v, dg2_a = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    G2v, G2a = k.xattn_pair_bwd(pxv, pxa, gates=(gate_v, gate_a, dg2_v, dg2_a))
    assert eq(Gv, G2v) and eq(Ga, G2a), "gated merged backward differs from gate_bwd2 + merged backward"
    dg3_v, dg3_a = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    buf2 = torch.full((2 * P * n, D), float("nan"), dtype=BF16, device=gpu)
    j2 = k.xattn_pair_bwd(pxv, pxa, join=(dxv, zv, dxa, za), outs=(buf2[:P * n], buf2[P * n:]), gates=(gate_v, gate_a, dg3_v, dg3_a))
    torch.cuda.synchronize()
    assert eq(j0[0], j2[0]) and eq(j0[1], j2[1]), "gated merged backward with the join differs"
    for got, want, dx, r in ((dg2_v, dg_v, dxv, rv), (dg2_a, dg_a, dxa, ra), (dg3_v, dg_v, dxv, rv), (dg3_a, dg_a, dxa, ra)):
        terms = dx.float() * r.float()
        assert abs(float(got) - float(want)) <= 2e-4 * float(terms.abs().sum()) ** 0.5 + 1e-4 * abs(float(want)), (float(got), float(want))


@pytest.mark.parametrize("P,nv,na,D,scale", [(5, 197, 49, 48, 1.0), (3, 257, 64, 64, 1.0), (4, 50, 7, 48, 0.5), (2, 33, 33, 32, 1.0), (6, 256, 1, 48, 1.0)])
def test_vit_pair_small_frames(stg, gpu, P, nv, na, D, scale):
    """Round 6 (xsmall.hip): the ViT blocks' cross-modal pair with both modalities' rows of a frame in LDS -- forward (both directions) and the merged
    backward (both whole gradients) in one launch each -- against fp32 autograd and against the generic pair launches it replaces; ViT-B's 197 + 49
    tokens at width 48, ViT-L-like 257 + 64 at 64 (nv = 257 > 256: must be REFUSED), ragged tiles, one audio token."""
    from stgcma import kernels as k
    if nv > 256:
        assert not k.xsmall_supported(nv, na, D)
        with pytest.raises(RuntimeError):
            k.XsGeom(P, nv, na, D, scale)
        return
    g = torch.Generator().manual_seed(11 * nv + na + D)
    xv = (torch.randn(P * nv, D, generator=g) * 0.6).to(BF16)
    xa = (torch.randn(P * na, D, generator=g) * 0.6).to(BF16)
    dv, da = torch.randn(P * nv, D, generator=g).to(BF16), torch.randn(P * na, D, generator=g).to(BF16)
    fv, fa = xv.float().requires_grad_(True), xa.float().requires_grad_(True)
    S = scale * torch.einsum("pid,pjd->pij", fv.view(P, nv, D), fa.view(P, na, D))
    rv = (torch.softmax(S, 2) @ fa.view(P, na, D)).reshape(P * nv, D)
    ra = (torch.softmax(S.transpose(1, 2), 2) @ fv.view(P, nv, D)).reshape(P * na, D)
    ((rv * dv.float()).sum() + (ra * da.float()).sum()).backward()
    xg = k.XsGeom(P, nv, na, D, scale)
    Xv, Xa, dV, dA = xv.to(gpu), xa.to(gpu), dv.to(gpu), da.to(gpu)
    for n_ in (nv, na):                                       # poison the allocator's free blocks: an unwritten row must not look right by accident
        t = torch.full((P * n_, D), float("nan"), dtype=BF16, device=gpu); del t
    (Ov, lv), (Oa, la) = k.xsmall_fwd(xg, Xv, Xa)
    _close(Ov, rv.detach(), what="O_v")
    _close(Oa, ra.detach(), what="O_a")
    _close(lv.view(-1), (torch.logsumexp(S, 2) * 1.4426950408889634).detach().reshape(-1), tol=2e-2, what="lse_v")
    _close(la.view(-1), (torch.logsumexp(S.transpose(1, 2), 2) * 1.4426950408889634).detach().reshape(-1), tol=2e-2, what="lse_a")
    Gv, Ga = k.xsmall_bwd(xg, Xv, Xa, Ov, Oa, lv, la, dV, dA)
    torch.cuda.synchronize()
    sc = float(max(fv.grad.abs().max(), fa.grad.abs().max()))
    assert torch.isfinite(Gv.float()).all() and torch.isfinite(Ga.float()).all()
    _close(Gv.float() / sc, fv.grad / sc, tol=1.5e-2, what="G_v")
    _close(Ga.float() / sc, fa.grad / sc, tol=1.5e-2, what="G_a")
    # the generic pair launches on the same problem
    gv = k.AttnGeom(P, 1, nv, D, G=1, outer=nv, n_kv=na, outer_kv=na, scale=scale)
    ga = k.AttnGeom(P, 1, na, D, G=1, outer=na, n_kv=nv, outer_kv=nv, scale=scale)
    (r0, l0), (r1, l1) = k.attn_fwd2(gv, Xv, Xa, Xa, ga, Xa, Xv, Xv)
    _close(Ov, r0.float(), what="O_v vs generic")
    _close(Oa, r1.float(), what="O_a vs generic")
    (dq_v, dkv_a), (dq_a, dkv_v) = k.attn_bwd2((gv, Xv, Xa, r0, l0, dV), (ga, Xa, Xv, r1, l1, dA))
    _close(Gv.float() / sc, (dq_v.float() + dkv_v.float()) / sc, tol=1.5e-2, what="G_v vs generic")
    _close(Ga.float() / sc, (dq_a.float() + dkv_a.float()) / sc, tol=1.5e-2, what="G_a vs generic")
