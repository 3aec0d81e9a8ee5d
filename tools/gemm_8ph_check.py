"""Bit-level check + timing of the GEMM kernel selected by STG_GEMM_8PH (0 / 1 / 2): prints one checksum line per shape and
repeat so that two runs with different settings can be diffed (same MFMA / k order => bit-identical outputs; any difference
between repeats of ONE run is a race)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
dev = "cuda"
reps = int(os.environ.get("REPS", 12))
shapes = [(125440, 512, 2048, "b"), (125440, 512, 1536, ""), (125440, 2048, 512, "bap"), (125440, 2048, 512, "d"), (125440, 1536, 512, "b"),
          (125440, 512, 512, "b"), (31360, 1024, 4096, "b"), (31360, 4096, 1024, "bap"), (1000, 256, 64, "b"), (256, 256, 128, ""),
          (300, 512, 192, "b"), (501760, 256, 1024, "b"), (4097, 768, 3072, ""),
          (78720, 2304, 768, "b"), (78720, 3072, 768, "bap"), (78720, 768, 3072, "b"), (31360, 6144, 1536, "bap"), (125440, 3072, 768, "b"),
          (1000, 1536, 512, "b"), (257, 2048, 1024, "bap"), (15680, 4608, 1536, "b")]
def csum(t):
    return int(t.view(torch.int16).to(torch.int64).sum().item()) if t.dtype == torch.bfloat16 else int(t.view(torch.int32).to(torch.int64).sum().item())
for (M, N, Kd, tag) in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + Kd)
    A = torch.randn(M, Kd, device=dev, generator=g).bfloat16(); W = (torch.randn(N, Kd, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev, generator=g) if "b" in tag else None
    opts = {}
    if "a" in tag: opts["act"] = K.ACT_GELU
    if "p" in tag: opts["want_dact"] = True
    if "d" in tag: opts["dact_src"] = torch.randn(M, N, device=dev, generator=g).bfloat16()
    sums = []
    for r in range(reps):
        out = K.gemm_nt(A, W, bias, **opts)
        o = out if not isinstance(out, tuple) else out
        sums.append(tuple(csum(x) for x in (o if isinstance(o, tuple) else (o,))))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(10): K.gemm_nt(A, W, bias, **opts)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    same = all(s == sums[0] for s in sums)
    print(f"M={M} N={N} K={Kd} epi={tag:4s} csum={sums[0]} repeats_identical={same}   # {us:8.1f} us {2.0*M*N*Kd/us/1e6:7.1f} TF", flush=True)
