"""stg-cma_amd: MI355X-native (gfx950) implementation of the STG-CMA ViT/Swin + cross-modal-adapter hot path.

Layout
  csrc/         hand-written HIP kernels + the C ABI (include/stgcma.h) -> libstgcma_hip.so
  _lib.py       ctypes binding (fails loudly when the .so is missing; no CPU fallback)
  kernels.py    one-launch wrappers on torch tensors / torch's current HIP stream
  ops.py        torch.autograd.Function glue (forward + hand-written backward kernels)
  model/        nn.Module mirror of the reference's AVE/AVQA/AVS model files (same ctor kwargs, forward, state_dict keys)
  ddp.py        one-process-per-GPU data parallelism: flat-bucket RCCL all-reduce of the trainable gradients
  fp8.py        opt-in block-scaled e4m3 path of the frozen backbone Linears (BASELINE config 5)

The directory name carries a hyphen, so import it as `import stgcma` (top-level alias module) or
`importlib.import_module("stg-cma_amd")`.
"""
__version__ = "0.1.0"

from .config import configure, options  # noqa: E402,F401
