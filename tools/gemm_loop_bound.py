"""What bounds the 8-phase main loop?  Diagnostics build (make -C stg-cma_amd/csrc diag; STGCMA_LIB=stg-cma_amd/libstgcma_hip_diag.so):
launch time of the one-tile 8-phase kernel with gemm_dbg = 0 (as shipped), 1 (no LDS-DMA in the loop: stale LDS contents), 2 (no MFMAs), 3 (no epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K, _lib
L = _lib.lib()
_lib.check(L.stg_set_option(b"gemm_8ph", 2), "opt")
_lib.check(L.stg_set_option(b"gemm_8phm", 0), "opt")
for (M, N, Kd) in ((125440, 512, 2048), (125440, 512, 512), (31360, 1024, 4096), (65536, 2048, 8192)):
    A = torch.randn(M, Kd, device="cuda").bfloat16(); W = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for dbg in (0, 1, 2, 3):
            _lib.check(L.stg_set_option(b"gemm_dbg", dbg), "opt")
            for _ in range(3): K.gemm_nt(A, W, None, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): K.gemm_nt(A, W, None, out=out)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(dbg, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    tiles = ((M + 255) // 256) * (N // 256)
    print(f"M={M} N={N} K={Kd} ({tiles} tiles, {tiles/256:.2f} per CU, {Kd//64} k-tiles): " + "  ".join(f"dbg={d}: {min(v):7.1f} us" for d, v in res.items()) + f"   [{K.LAST_GEMM_KERNEL}]", flush=True)
_lib.check(L.stg_set_option(b"gemm_dbg", 0), "opt")
