"""Round 6: the GEMM classes of the Swin-L step whose widths miss the round-5 shape rules of the 8-phase kernels (N % 256 != 0 or K % 128 != 0),
timed under option gemm_nx = 0 (round-5 routing: 128 x 128 LDS-DMA kernel), 1 (NX forms behind the plain / activation epilogues), 2 (every legal
NX shape), interleaved in one process; prints the kernel each option chose.   python tools/gemm_nx_ab.py [half]   (half = micro-batch rows)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma  # noqa: E402
from stgcma import kernels as K, _lib  # noqa: E402
sys.argv = [sys.argv[0], "/dev/null"] + sys.argv[1:]
dev = "cuda"
half = len(sys.argv) > 2 and sys.argv[2] == "half"
R = [2007040, 501760, 125440, 31360]
if half:
    R = [r // 2 for r in R]
# (stage, N, K, epilogue, launches per step) -- Swin-L AVE, from gpurun_out/bench_detail.json of round 5 (C = 192 / 384 / 768 / 1536)
CLASSES = [(0, 768, 192, "bap8", 2), (0, 576, 192, "b", 3), (0, 768, 192, "d8", 2), (0, 192, 768, "b", 2), (0, 192, 768, "", 2), (0, 192, 576, "", 2),
           (0, 192, 192, "b", 3), (0, 192, 192, "", 3), (1, 1152, 384, "b", 3), (1, 384, 1152, "", 3), (1, 384, 1536, "b", 2), (1, 384, 1536, "", 2),
           (1, 384, 384, "b", 3), (1, 384, 384, "", 3), (1, 1536, 384, "bap8", 2), (1, 1536, 384, "d8", 2)]


def make(M, N, Kd, e):
    A = torch.randn(M, Kd, device=dev).bfloat16()
    W = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    kw = {}
    bias = torch.randn(N, device=dev) if "b" in e else None
    if "a" in e:
        kw["act"] = K.ACT_GELU
    if "p8" in e:
        kw["want_dact"] = "u8"
    if "d8" in e:
        kw["dact_src"] = torch.randint(0, 255, (M, N), device=dev, dtype=torch.uint8)
    return A, W, bias, kw


L = _lib.lib()
VALS = [0, 1, 2]
tot = {v: 0.0 for v in VALS}
print(f"{'M':>8s} {'N':>5s} {'K':>5s} {'epi':5s} {'x':>2s} | " + "  ".join(f"nx={v} us (kernel)      " for v in VALS))
for st, N, Kd, epi, n in CLASSES:
    M = R[st]
    A, W, bias, kw = make(M, N, Kd, epi)
    times = {v: [] for v in VALS}
    kern = {}
    outs = {}
    for rnd in range(5):
        for v in VALS:
            L.stg_set_option(b"gemm_nx", v)
            o = K.gemm_nt(A, W, bias, **kw)
            kern[v] = K.LAST_GEMM_KERNEL.replace("gemm_nt_", "").replace("_kernel", "")[:12]
            if rnd == 0:
                outs[v] = (o[0] if isinstance(o, tuple) else o).float().sum().item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                K.gemm_nt(A, W, bias, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 6 * 1e3)
    med = {v: statistics.median(t) for v, t in times.items()}
    for v in VALS:
        tot[v] += med[v] * n
    same = all(abs(outs[v] - outs[0]) <= 1e-3 * abs(outs[0]) + 1.0 for v in VALS)
    print(f"{M:8d} {N:5d} {Kd:5d} {epi:5s} {n:2d} | " + "  ".join(f"{med[v]:8.1f} ({kern[v]:12s})" for v in VALS) + ("" if same else "  CHECKSUMS DIFFER " + str(outs)), flush=True)
    del A, W, bias, kw
L.stg_set_option(b"gemm_nx", 1)
print("per-step totals (ms): " + "   ".join(f"nx={v}: {tot[v] / 1e3:.2f}" for v in VALS))
