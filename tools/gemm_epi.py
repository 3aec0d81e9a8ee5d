"""Epilogue cost ablation of stg_gemm_nt on the hot shapes:  python tools/gemm_epi.py   (env STG_GEMM_DBG=3 skips the epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
dev = "cuda"
def run(M, N, Kd, tag, iters=20, **kw):
    A = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev) if "b" in tag else None
    opts = {}
    if "a" in tag: opts["act"] = K.ACT_GELU
    if "p" in tag: opts["want_dact"] = True
    if "d" in tag: opts["dact_src"] = torch.randn(M, N, device=dev).bfloat16()
    if "r" in tag: opts["res1"] = torch.randn(M, N, device=dev).bfloat16()
    if "R" in tag: opts["res2"] = torch.randn(M, N, device=dev)
    od = torch.float32 if "F" in tag else torch.bfloat16
    out = torch.empty(M, N, device=dev, dtype=od)
    if "p" not in tag: opts["out"] = out
    else: opts["out_dtype"] = od
    for _ in range(3): K.gemm_nt(A, W, bias, **opts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): K.gemm_nt(A, W, bias, **opts)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    by = M * Kd * 2 + M * N * (4 if "F" in tag else 2) * (2 if "p" in tag else 1) + (M * N * 2 if "d" in tag else 0) + \
         (M * N * 2 if "r" in tag else 0) + (M * N * 4 if "R" in tag else 0)
    print(f"M={M} N={N} K={Kd} epi={tag:6s} {ms*1e3:8.1f} us  {2.0*M*N*Kd/ms/1e9:7.1f} TF  {by/ms/1e6:7.0f} GB/s", flush=True)
cases = [(125440, 2048, 512, t) for t in ("", "b", "ba", "bap", "d")] + [(125440, 1536, 512, t) for t in ("", "b")] + \
        [(125440, 512, 512, ""), (125440, 512, 2048, "")] + [(62720, 512, 32, t) for t in ("", "r", "brRF")] + \
        [(2007040, 512, 128, "bap"), (2007040, 384, 128, "b")]
if len(sys.argv) > 1:
    cases = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "")]
for c in cases: run(*c)
