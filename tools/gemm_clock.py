"""In-kernel clock of the 8-phase GEMM main loop (diagnostics build):  STGCMA_LIB=stg-cma_amd/libstgcma_hip_diag.so python tools/gemm_clock.py
Stamps s_memtime / s_memrealtime around the main loop of every workgroup (option gemm_dbg = 4) after >= 2 s of back-to-back launches on
random data; clock = d(memtime) / d(memrealtime) x 100 MHz, median over workgroups (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, stgcma
from stgcma import kernels as K, _lib
L = _lib.lib()
L.stg_diag_read_stamps.restype = C.c_int
L.stg_diag_read_stamps.argtypes = [C.c_void_p, C.c_int]
for (M, N, Kd) in ((125440, 2048, 512), (125440, 512, 2048), (8192, 8192, 8192)):
    A = torch.randn(M, Kd, device="cuda").bfloat16(); W = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.stg_set_option(b"gemm_8ph", 2), "opt")
    _lib.check(L.stg_set_option(b"gemm_dbg", 4), "opt")
    t0 = time.time()
    while time.time() - t0 < 2.5:
        for _ in range(20): K.gemm_nt(A, W, None, out=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.gemm_nt(A, W, None, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    nwg = min(8192, ((M + 255) // 256) * (N // 256))
    buf = (C.c_ulonglong * (4 * nwg))()
    assert L.stg_diag_read_stamps(buf, nwg) == 0
    s = np.array(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.float64)
    dcyc, dreal = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
    ok = dreal > 0
    clk = dcyc[ok] / dreal[ok] * 100e6
    nk = Kd // 64
    mfma_cycles = nk * 64 * 16 * 2          # per SIMD and tile: 2 waves x 64 MFMAs of 16 cycles per k-tile
    print(f"M={M} N={N} K={Kd}: {us:8.1f} us/launch {2.0*M*N*Kd/us/1e6:7.1f} TF/s | main loop per tile: median {np.median(dcyc):9.0f} shader cycles "
          f"({np.median(dreal)/100:7.2f} us), in-kernel clock median {np.median(clk)/1e9:.3f} GHz (p10 {np.percentile(clk,10)/1e9:.3f}, p90 {np.percentile(clk,90)/1e9:.3f}); "
          f"MFMA-issue share of the main loop = {mfma_cycles/np.median(dcyc):.3f}")
