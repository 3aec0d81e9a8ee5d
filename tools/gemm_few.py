import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stgcma
from stgcma import kernels as K
def bench(M, N, Kd, lda=None, reps=5):
    lda = lda or Kd
    A = torch.randn(M, lda, device="cuda").bfloat16()[:, :Kd]; W = (torch.randn(N, lda, device="cuda") * 0.05).bfloat16()[:, :Kd]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(2): K.gemm_nt(A, W, None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): K.gemm_nt(A, W, None, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"dbg={os.environ.get('STG_GEMM_DBG','0')} M={M} N={N} K={Kd} lda={lda}: {ms:.3f} ms {2.0*M*N*Kd/ms/1e9:.1f} TF/s", flush=True)
bench(8192, 8192, 8192); bench(8192, 8192, 8192, lda=8192 + 64); bench(125440, 1536, 512); bench(125440, 1536, 512, lda=576); bench(2007040, 384, 128)
