"""Aggregate per-shape GEMM time over one training step of the bench model (B from env, default 32)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
import bench
dev = torch.device("cuda:0")
m = bench.build_model(torch, dev)
B = int(os.environ.get("B", 32))
a, v, labels = bench.synth_batch(torch, B, dev, 0)
loss_fn = torch.nn.CrossEntropyLoss()
def step():
    loss = loss_fn(m(a, v, "fusion"), labels); loss.backward()
    for p in m.parameters(): p.grad = None
step(); step(); torch.cuda.synchronize()
rec = []
orig = K.gemm_nt
def wrapped(A, W, bias=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(A, W, bias, **kw); e1.record()
    epi = ("b" if bias is not None else "") + ("a" if kw.get("act") else "") + ("d" if kw.get("dact_src") is not None else "") + \
          ("" if kw.get("res1") is None else ("r" if kw["res1"].dtype == torch.bfloat16 else "q")) + \
          ("" if kw.get("res2") is None else ("R" if kw["res2"].dtype == torch.bfloat16 else "Q")) + ("s" if kw.get("row_scale") is not None else "") + \
          ("p" if kw.get("want_dact") else "") + ("A" if kw.get("alpha", 1.0) != 1.0 else "")
    od = kw["out"].dtype if kw.get("out") is not None else kw.get("out_dtype", torch.bfloat16)
    rec.append((A.shape[0], W.shape[0], A.shape[1], epi, str(od)[-4:], e0, e1)); return r
K.gemm_nt = wrapped
import stgcma.ops as ops
step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for M, N, Kd, epi, od, e0, e1 in rec:
    k = (M, N, Kd, epi, od); agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms {tot:.1f} launches {len(rec)}")
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', 40))]:
    M, N, Kd, epi, od = k
    fl = 2.0 * M * N * Kd * c
    by = c * (M * Kd * 2 + M * N * (4 if od == "at32" else 2))
    print(f"M={M:8d} N={N:5d} K={Kd:5d} epi={epi:6s} out={od} x{c:3d}  {ms:7.2f} ms  {fl/ms/1e9:7.1f} TF  ~{by/ms/1e6:7.0f} GB/s(min)")

sig = collections.defaultdict(lambda: [0, 0.0])
for (M, N, Kd, epi, od), (c, ms) in agg.items():
    sig[(epi, od)][0] += c; sig[(epi, od)][1] += ms
print("--- epilogue signatures")
for k, (c, ms) in sorted(sig.items(), key=lambda kv: -kv[1][1]):
    print(f"epi={k[0]:8s} out={k[1]} x{c:4d} {ms:7.2f} ms")
