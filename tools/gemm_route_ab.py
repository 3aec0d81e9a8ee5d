"""Which GEMM kernel should each class of the training step take?  Times every (M, N, K, epilogue) class of one step -- read from a
bench.py --gemm-seq file -- under the dispatch options gemm_8ph = 0 (128 x 128 LDS-DMA kernel / large-tile kernel), 1 (the shipped rule),
2 (8-phase kernel wherever legal), interleaved in ONE process (5 rounds x 8 launches each, median), and prints the per-class times plus
the step total of each option and of the per-class best.   GPU box:  python tools/gemm_route_ab.py gpurun_out/ledger/<tag>_gemm_seq.json"""
import json, os, sys, statistics, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma  # noqa: E402
from stgcma import kernels as K, _lib  # noqa: E402

dev = "cuda"
seq = json.load(open(sys.argv[1]))["step"]
OPT = (sys.argv[2] if len(sys.argv) > 2 else "gemm_8ph").encode()          # the option to sweep (values 0, 1, 2), e.g. gemm_8phm
VALS = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1,2".split(","))]
cls = collections.Counter((M, N, Kd, epi) for _, M, N, Kd, epi, _ in seq)
L = _lib.lib()


def make(M, N, Kd, epi):
    A = torch.randn(M, Kd, device=dev).bfloat16()
    W = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    kw = {}
    e = epi
    bias = torch.randn(N, device=dev) if "b" in e else None
    if "a" in e:
        kw["act"] = K.ACT_GELU
    if "p8" in e:
        kw["want_dact"] = "u8"
    elif "p" in e:
        kw["want_dact"] = True
    if "d8" in e:
        kw["dact_src"] = torch.randint(0, 255, (M, N), device=dev, dtype=torch.uint8)
    elif "d" in e:
        kw["dact_src"] = torch.randn(M, N, device=dev).bfloat16()
    if "r" in e:
        kw["res1"] = torch.randn(M, N, device=dev).bfloat16()
    if "q" in e:
        kw["res1"] = torch.randn(M, N, device=dev)
    if "R" in e:
        kw["res2"] = torch.randn(M, N, device=dev).bfloat16()
    if "Q" in e:
        kw["res2"] = torch.randn(M, N, device=dev)
    if "s" in e:
        kw["row_scale"] = torch.rand(M, device=dev)
        kw["rs_outer"], kw["rs_inner"] = 1, 1
    if "F" in e:
        kw["out_dtype"] = torch.float32
    return A, W, bias, kw


tot = collections.defaultdict(float)
best_tot = 0.0
print(f"{'M':>8s} {'N':>5s} {'K':>5s} {'epi':6s} {'x/step':>6s} | " + " ".join(f"{OPT.decode()[5:]}={v:<3d}" for v in VALS) + "  best")
for (M, N, Kd, epi), n in sorted(cls.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
    if M * N * Kd < 1e9 or Kd % 64 != 0:
        continue
    A, W, bias, kw = make(M, N, Kd, epi)
    times = {v: [] for v in VALS}
    for rnd in range(5):
        for mode in VALS:
            L.stg_set_option(OPT, mode)
            K.gemm_nt(A, W, bias, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                K.gemm_nt(A, W, bias, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / 8 * 1e3)
    med = {m: statistics.median(v) for m, v in times.items()}
    b = min(med, key=med.get)
    for m in med:
        tot[m] += med[m] * n
    best_tot += med[b] * n
    print(f"{M:8d} {N:5d} {Kd:5d} {epi:6s} {n:6d} | " + " ".join(f"{med[v]:8.1f}" for v in VALS) + f"  {b}" + ("  <-- differs from the shipped rule" if 1 in med and abs(med[b] - med[1]) > 0.02 * med[1] else ""), flush=True)
    del A, W, bias, kw
L.stg_set_option(OPT, 1 if 1 in VALS else VALS[0])
print("step totals (ms): " + "   ".join(f"{OPT.decode()}={v} {tot[v]/1e3:.2f}" for v in VALS) + f"   per-class best {best_tot/1e3:.2f}")
