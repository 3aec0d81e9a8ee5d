"""Start stagger of the 8-phase GEMM kernels (option gemm_stagger = 256 * phases + q, csrc/gemm.hip start_stagger) per K = 512 class of the
step: interleaved timing in one process, outputs compared bit for bit (the option changes WHEN workgroups start, nothing else)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K
from stgcma._lib import ACT_GELU

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
M = int(os.environ.get("M", 125440))
ROUNDS = int(os.environ.get("ROUNDS", 5))
MODES = [int(x) for x in os.environ.get("MODES", "0,2,4,6,8,514,516,520,2050,2052").split(",")]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = [("fc1 gelu+d8", 2048, 512, "gelu8"), ("qkv plain+bias", 1536, 512, "plain"), ("proj plain+bias", 512, 512, "plain"),
         ("fc2 N=512 K=2048", 512, 2048, "plain"), ("stage1 fc1", 1024, 256, "gelu8"), ("stage1 qkv", 768, 256, "plain")]
sel = os.environ.get("CASES")
for name, N, Kd, kind in cases:
    if sel and not any(s in name for s in sel.split(",")):
        continue
    Mc = M * 4 if Kd == 256 else M
    torch.manual_seed(0)
    A = (torch.randn(Mc, Kd, device=dev) * 0.5).to(BF16)
    W = (torch.randn(N, Kd, device=dev) * 0.05).to(BF16)
    b = torch.randn(N, device=dev) * 0.1

    def run():
        if kind == "gelu8":
            return K.gemm_nt(A, W, b, act=ACT_GELU, want_dact="u8")
        return (K.gemm_nt(A, W, b),)

    stgcma.configure(lib_gemm_stagger=0)
    ref = run()
    torch.cuda.synchronize()
    ts = {m: [] for m in MODES}
    same = {}
    for r in range(ROUNDS):
        for m in MODES:
            stgcma.configure(lib_gemm_stagger=m)
            if r == 0:
                out = run()
                torch.cuda.synchronize()
                same[m] = all(torch.equal(x.view(torch.int16) if x.dtype == BF16 else x, y.view(torch.int16) if y.dtype == BF16 else y) for x, y in zip(out, ref))
            ts[m].append(timeit(run))
    fl = 2.0 * Mc * N * Kd
    print(f"{name}: M={Mc} N={N} K={Kd}", flush=True)
    for m in MODES:
        t = sorted(ts[m])
        print(f"    stagger {m:5d} (phases {(m >> 8) or 4}, q {m & 255:2d})  median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  {fl / t[len(t) // 2] / 1e6:6.0f} TFLOP/s  identical {same[m]}", flush=True)
stgcma.configure(lib_gemm_stagger=0)
