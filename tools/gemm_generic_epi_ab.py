import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch, stgcma
from stgcma import kernels as K, _lib
L = _lib.lib()
dev = "cuda"
for (M, N, Kd) in ((62720, 768, 768), (250880, 384, 384), (15680, 1536, 1536), (62720, 768, 3072), (250880, 384, 1536)):
    A = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); rs = torch.rand(M // 196 + 1, device=dev)
    kw = dict(res1=res, row_scale=rs, rs_outer=196, rs_inner=1, out_dtype=torch.float32)
    out = {}
    for mode in (1, 2, 1, 2):
        L.stg_set_option(b"gemm_8ph", mode)
        K.gemm_nt(A, W, b, **kw)
        kern = K.LAST_GEMM_KERNEL
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): K.gemm_nt(A, W, b, **kw)
        e1.record(); torch.cuda.synchronize()
        out.setdefault(mode, []).append((e0.elapsed_time(e1) / 10 * 1e3, kern))
    print(M, N, Kd, {m: [(round(t, 1), k[8:20]) for t, k in v] for m, v in out.items()}, flush=True)
L.stg_set_option(b"gemm_8ph", 1)
