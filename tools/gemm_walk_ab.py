"""Row walk vs column walk of the multi-tile 8-phase GEMM kernel (options gemm_8phm, gemm_8phm_walk), per GEMM class of the step:
bit-equality against the one-tile kernel (gemm_8phm = 0) and interleaved timing in ONE process (5 rounds x 8 launches, median).
GPU box:  python tools/gemm_walk_ab.py [m8:walk ...]   e.g.  1:0 1:4 1:8 1:16 2:8"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma  # noqa: E402,F401
from stgcma import kernels as K, _lib  # noqa: E402

dev = "cuda"
L = _lib.lib()
CFG = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(1, 0), (1, 4), (1, 8), (1, 16)]
# (M, N, K, epilogue, launches per step)
CLASSES = [(125440, 2048, 512, "bap8", 18), (125440, 1536, 512, "b", 27), (125440, 2048, 512, "d8", 18), (501760, 1024, 256, "bap8", 2),
           (501760, 768, 256, "b", 3), (501760, 1024, 256, "d8", 2), (125440, 512, 512, "b", 27), (125440, 512, 512, "", 27),
           (125440, 512, 2048, "b", 19), (125440, 512, 1536, "", 27), (31360, 4096, 1024, "bap8", 2), (31360, 3072, 1024, "b", 3)]


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


def setopt(m8, walk):
    L.stg_set_option(b"gemm_8phm", m8)
    L.stg_set_option(b"gemm_8phm_walk", walk)


tot = {c: 0.0 for c in CFG}
print(f"{'M':>8s} {'N':>5s} {'K':>5s} {'epi':5s} | " + " ".join(f"{m}:{w:<5d}" for m, w in CFG) + " | bit-equal to the one-tile kernel")
for (M, N, Kd, epi, n) in CLASSES:
    A, W, bias, kw = make(M, N, Kd, epi)
    setopt(0, 0)
    ref = K.gemm_nt(A, W, bias, **kw)
    ref = ref if isinstance(ref, tuple) else (ref,)
    times = {c: [] for c in CFG}
    eq = []
    for c in CFG:
        setopt(*c)
        out = K.gemm_nt(A, W, bias, **kw)
        out = out if isinstance(out, tuple) else (out,)
        eq.append(all(torch.equal(a, b) for a, b in zip(ref, out)))
    for rnd in range(5):
        for c in CFG:
            setopt(*c)
            K.gemm_nt(A, W, bias, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                K.gemm_nt(A, W, bias, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[c].append(e0.elapsed_time(e1) / 8 * 1e3)
    med = {c: statistics.median(v) for c, v in times.items()}
    for c in CFG:
        tot[c] += med[c] * n
    print(f"{M:8d} {N:5d} {Kd:5d} {epi:5s} | " + " ".join(f"{med[c]:7.1f}" for c in CFG) + " | " + " ".join("ok" if e else "DIFF" for e in eq), flush=True)
    del A, W, bias, kw, ref
setopt(1, 0)
print("step totals (ms): " + "   ".join(f"{m}:{w} {tot[(m, w)]/1e3:.2f}" for m, w in CFG))
