"""The overlapped-epilogue GEMM (csrc/gemm_ovl.hip) against the shipped routing, per class of the step: bit-level comparison of every output,
then interleaved timing in one process (option gemm_ovl: 0 = shipped kernels, 2 = every legal shape, 16 * ntl + 2 = fixed walk length)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stgcma
from stgcma import kernels as K
from stgcma._lib import ACT_GELU, ACT_NONE

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
M = int(os.environ.get("M", 125440))
ROUNDS = int(os.environ.get("ROUNDS", 5))


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
         ("fc2 dgrad x d8", 2048, 512, "dsrc8"), ("stage1 fc1", 1024, 256, "gelu8"), ("stage1 qkv", 768, 256, "plain")]
sel = os.environ.get("CASES")
for name, N, Kd, kind in cases:
    if sel and not any(s in name for s in sel.split(",")):
        continue
    Mc = M * 4 if Kd == 256 else M
    torch.manual_seed(0)
    A = (torch.randn(Mc, Kd, device=dev) * 0.5).to(BF16)
    W = (torch.randn(N, Kd, device=dev) * 0.05).to(BF16)
    b = torch.randn(N, device=dev) * 0.1
    d8 = torch.randint(0, 256, (Mc, N), device=dev, dtype=torch.uint8) if kind == "dsrc8" else None

    def run():
        if kind == "gelu8":
            return K.gemm_nt(A, W, b, act=ACT_GELU, want_dact="u8")
        if kind == "dsrc8":
            return (K.gemm_nt(A, W, None, dact_src=d8),)
        return (K.gemm_nt(A, W, b),)

    stgcma.configure(lib_gemm_ovl=0)
    ref = run()
    torch.cuda.synchronize()
    modes = [("shipped", 0), ("ovl auto-ntl", 2)] + [(f"ovl ntl={c}", 16 * c + 2) for c in (2, 4, 8) if (N // 128) % c == 0]
    ok = {}
    for label, m in modes[1:]:
        stgcma.configure(lib_gemm_ovl=m)
        out = run()
        torch.cuda.synchronize()
        ok[label] = all(torch.equal(x.view(torch.int16) if x.dtype == BF16 else x, y.view(torch.int16) if y.dtype == BF16 else y) for x, y in zip(out, ref))
        if not ok[label]:
            d = (out[0].float() - ref[0].float()).abs()
            ok[label] = f"MISMATCH max {float(d.max()):.3e} frac {float((d > 0).float().mean()):.3e}"
    ts = {label: [] for label, _ in modes}
    for r in range(ROUNDS):
        for label, m in modes:
            stgcma.configure(lib_gemm_ovl=m)
            ts[label].append(timeit(run))
    fl = 2.0 * Mc * N * Kd
    print(f"{name}: M={Mc} N={N} K={Kd}", flush=True)
    for label, _ in modes:
        t = sorted(ts[label])
        print(f"    {label:14s} median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  {fl / t[len(t) // 2] / 1e6:6.0f} TFLOP/s   bit-identical: {ok.get(label, '-')}", flush=True)
stgcma.configure(lib_gemm_ovl=1)
