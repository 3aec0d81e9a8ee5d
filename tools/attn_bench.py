"""Attention micro-benchmark over the call types of the Swin-B AVE step (B=32, T=10)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K, ops
B, T = int(os.environ.get("B", 32)), 10
BT = B * T
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def run(name, g, Q, Kt, V, shared=False, dbias=False):
    O, lse = K.attn_fwd(g, Q, Kt, V)
    dO = torch.randn_like(O)
    f = timeit(lambda: K.attn_fwd(g, Q, Kt, V))
    db = torch.zeros_like(g.bias) if dbias else None
    b = timeit(lambda: K.attn_bwd(g, Q, Kt, V, O, lse, dO, shared_kv=shared, dbias=db))
    fl = 4.0 * g.P * g.H * g.n * g.n_kv * g.D
    print(f"{name:34s} P={g.P:7d} H={g.H:2d} n={g.n:4d} D={g.D:3d}  fwd {f:7.3f} ms ({fl/f/1e9:6.1f} TF)  bwd {b:7.3f} ms ({2.5*fl/b/1e9:6.1f} TF)", flush=True)
stages = [(128, 56, 4, 16), (256, 28, 8, 32), (512, 14, 16, 32), (1024, 7, 32, 64)]
for s, (C, res, H, dh) in enumerate(stages):
    N = res * res; R = 2 * BT * N; hd = 32
    ge = ops.geom(torch.device("cuda"), res, res, 7, 3 if res > 7 else 0, T)
    nW = (res // 7) ** 2
    qkv = torch.randn(R, 3 * C, device="cuda").bfloat16()
    bias = torch.randn(1, H, 49, 49, device="cuda")
    g = K.AttnGeom(2 * BT * nW, H, 49, hd, G=nW, outer=N, window=(res, res, 7, 3 if res > 7 else 0), scale=hd ** -0.5, bias=bias, bias_div=2 * BT * nW, bias_mod=1, mask=ge["mask"])
    run(f"s{s} window", g, qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:])
    tb = torch.randn(2, H, T, T, device="cuda")
    g = K.AttnGeom(2 * B * N, H, T, hd, G=N, outer=T * N, temporal=N, scale=hd ** -0.5, bias=tb, bias_div=B * N, bias_mod=2)
    run(f"s{s} temporal", g, qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], dbias=True)
    hv = torch.randn(BT * N, dh, device="cuda").bfloat16(); ha = torch.randn(BT * N, dh, device="cuda").bfloat16()
    g = K.AttnGeom(BT * nW, 1, 49, dh, G=nW, outer=N, n_kv=49, outer_kv=N, scale=1.0, window=(res, res, 7, 3 if res > 7 else 0))
    run(f"s{s} xmodal window", g, hv, ha, ha, shared=True)
    g = K.AttnGeom(BT, 1, N, dh, G=1, outer=N, n_kv=N, outer_kv=N, scale=1.0)
    run(f"s{s} xmodal global", g, hv, ha, ha, shared=True)
