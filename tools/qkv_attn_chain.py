"""VERDICT r3 item 2, measured the cheap way: what would a C = 512 attention block gain if q, k, v never made the HBM round trip between the
qkv GEMM and the window attention?  The two-kernel chain (stg_gemm_nt N = 1536, K = 512 + stg_winattn_fwd) is run on row chunks small enough
that the [rows, 3C] tensor written by the GEMM is still in the 256 MiB Infinity Cache when the attention kernel reads it (and, for the
smallest chunks, partly in L2): the per-row time of the chain at those sizes is what a fused kernel that keeps q, k, v on chip could at best
save on the HBM side, since it executes the same MFMAs, exponentials and stores.  Also times each kernel alone at every size.
Stage-2 shape: 14 x 14 tokens per frame, 16 heads of 32; full size = 640 frames (B = 32 x 10 frames x 2 modalities)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K, ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
C, heads = 512, 16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
W = (torch.randn(3 * C, C, device=dev) * 0.04).to(BF16)
b = torch.randn(3 * C, device=dev) * 0.1
table = torch.randn(169, heads, device=dev)
co = torch.stack(torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")).flatten(1)
rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0) + 6
index = (rel[:, :, 0] * 13 + rel[:, :, 1]).reshape(-1).to(dev)
bm, bmT = K.winattn_table(table, index, None, 49)
full = 640
# an unrelated 600 MB stream between timed calls would flush the caches; here the chain is timed back to back ON PURPOSE (the best case
# for the chunked form: its x-hat input may be cache-resident too, which a real step would not give it)
print(f"{'frames':>7} {'rows':>8} {'qkv MB':>7} | {'gemm us':>8} {'attn us':>8} {'chain us':>9} | per 125 440 rows: gemm / attn / chain")
for images in (640, 320, 160, 80, 40):
    rows = images * 196
    X = (torch.randn(rows, C, device=dev) * 0.5).to(BF16)
    qkv = torch.empty(rows, 3 * C, dtype=BF16, device=dev)
    O = torch.empty(rows, C, dtype=BF16, device=dev)
    wg = K.WinGeom(images, heads, 14, 14, 7, 0, 32 ** -0.5, bm, bmT)

    def gemm():
        K.gemm_nt(X, W, b, out=qkv)

    def attn():
        K.winattn_fwd(wg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out=O, want_lse=False)

    def chain():
        gemm()
        attn()
    tg, ta, tc = timeit(gemm), timeit(attn), timeit(chain)
    s = full / images
    print(f"{images:7d} {rows:8d} {rows * 3 * C * 2 / 1e6:7.1f} | {tg:8.1f} {ta:8.1f} {tc:9.1f} | {tg * s:7.1f} / {ta * s:7.1f} / {tc * s:7.1f}", flush=True)
    del X, qkv, O
