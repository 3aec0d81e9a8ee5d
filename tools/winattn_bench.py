"""Window attention kernels at the four Swin-B stage shapes (B = 32 clips x 10 frames x 2 modalities), shifted and not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K, ops
dev = "cuda"
def t(fn, iters=None):
    iters = iters or int(os.environ.get('ITERS', 10))
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
images = int(os.environ.get("IMAGES", 640))
stages = [int(x) for x in os.environ.get("STAGES", "56,28,14,7").split(",")]
iters = int(os.environ.get("ITERS", 10))
for res, heads in [rh for rh in ((56, 4), (28, 8), (14, 16), (7, 32)) if rh[0] in stages]:
    C = heads * 32
    for shift in ((0, 3) if res > 7 else (0,)):
        rows = images * res * res
        qkv = torch.randn(rows, 3 * C, device=dev).bfloat16()
        dO = torch.randn(rows, C, device=dev).bfloat16()
        table = torch.randn(169, heads, device=dev)
        co = torch.stack(torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")).flatten(1)     # Swin_AVE.py:199-207
        rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0) + 6
        index = (rel[:, :, 0] * 13 + rel[:, :, 1]).reshape(-1).to(dev)
        mask = ops.shift_mask(res, res, 7, shift).to(dev) if shift else None
        bm, bmT = K.winattn_table(table, index, mask, 49)
        wg = K.WinGeom(images, heads, res, res, 7, shift, 32 ** -0.5, bm, bmT)
        O, lse = K.winattn_fwd(wg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:])
        d = torch.empty_like(qkv)
        f = t(lambda: K.winattn_fwd(wg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out=O))
        b = t(lambda: K.winattn_bwd(wg, qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], O, lse, dO, dQ=d[:, :C], dK=d[:, C:2 * C], dV=d[:, 2 * C:]))
        U = rows * C * 2
        print(f"res {res:2d} heads {heads:2d} shift {shift}: fwd {f:7.1f} us ({4*U/f/1e6:5.0f} TB/s)  bwd {b:7.1f} us ({8*U/b/1e6:5.2f} TB/s)  tables {bm.numel()*8/1e6:.2f} MB")
