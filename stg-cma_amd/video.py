"""Video front end on the device: the tensor part of the reference's training-time frame pipeline (AVE/dataloader.py:346-394,
`_aug_frame_train` behind the PIL RandAugment) for frames that are already decoded -- ToTensor, tensor_normalize (:470-485), the random
resized crop + bilinear resize and the horizontal flip of `spatial_sampling` (:396-468, transforms/video_transforms.py:483-561, :152-186)
and RandomErasing(0.25, mode='pixel', max_count=1, cube) (transforms/random_erasing.py) -- as ONE launch of stg_video_aug per batch
(video half).

What stays on the host: JPEG decoding (`Image.open`, :304-318) and the PIL RandAugment ('rand-m7-n4-mstd0.5-inc1', :348-356), which work
on PIL images.  The random draws are made HERE on the host with the reference's distributions (`draw_params`), one record per clip, and
handed to the kernel; the erase noise is drawn on the device (torch.randn).  Same distributions, not the same streams as Python's `random`:
the parity fixture (tests/golden/video_aug.npz) therefore records the reference's own draws and feeds them in.

No CPU fallback: CPU tensors raise.  The oracle (oracle/video_aug.py) is not imported here.
"""
import math
import random

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def crop_params(height, width, scale=(0.08, 1.0), ratio=(0.75, 1.3333), rng=random):
    """(top, left, h, w) of `_get_param_spatial_crop` (video_transforms.py:483-526, log-uniform aspect ratio, ten tries, central fallback)."""
    for _ in range(10):
        area = height * width
        target_area = rng.uniform(*scale) * area
        aspect = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return rng.randint(0, height - h), rng.randint(0, width - w), h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def erase_params(size, probability=0.25, min_area=0.02, max_area=1 / 3, min_aspect=0.3, rng=random):
    """(top, left, h, w) of RandomErasing._erase_cube's box in the size x size output (random_erasing.py:118-152, max_count = 1), or
    (0, 0, 0, 0) when this clip is not erased."""
    if rng.random() > probability:
        return 0, 0, 0, 0
    la = (math.log(min_aspect), math.log(1 / min_aspect))
    for _ in range(100):
        target_area = rng.uniform(min_area, max_area) * size * size
        aspect = math.exp(rng.uniform(*la))
        h = int(round(math.sqrt(target_area * aspect)))
        w = int(round(math.sqrt(target_area / aspect)))
        if w < size and h < size:
            return rng.randint(0, size - h), rng.randint(0, size - w), h, w
    return 0, 0, 0, 0


def draw_params(B, height, width, size=224, rng=random, flip_prob=0.5, erase_prob=0.25):
    """One int32 record per clip, in the order the reference draws them: crop box, flip, erase box."""
    rows = []
    for _ in range(B):
        i, j, h, w = crop_params(height, width, rng=rng)
        flip = int(rng.random() < flip_prob)
        rows.append((i, j, h, w, flip) + erase_params(size, erase_prob, rng=rng))
    return torch.tensor(rows, dtype=torch.int32)


def augment(frames, params, noise=None, *, mean=IMAGENET_MEAN, std=IMAGENET_STD, size=224, out=None):
    """frames: uint8 [B, T, H, W, 3] GPU tensor (decoded, RandAugment applied); params: int32 [B, 9] (draw_params; any device);
    noise: fp32 [B, T, 3, size, size] GPU tensor read inside the erase boxes (default: torch.randn when any clip erases).
    Returns fp32 [B, 3, T, size, size] -- the `v` argument of model(a, v, mode)."""
    if not frames.is_cuda:
        raise RuntimeError("stgcma.video.augment runs on MI355X only: move the frames to the GPU (no CPU fallback)")
    if frames.dim() != 5 or frames.shape[-1] != 3 or frames.dtype != torch.uint8 or not frames.is_contiguous():
        raise RuntimeError("augment: expected contiguous uint8 [B, T, H, W, 3] frames")
    B, T, H, W, _ = frames.shape
    params = torch.as_tensor(params, dtype=torch.int32)
    if tuple(params.shape) != (B, 9):
        raise RuntimeError(f"augment: params must be int32 [{B}, 9]")
    pc = params.cpu()
    for i, j, h, w, flip, et, el, eh, ew in pc.tolist():      # a kernel must never see a box that leaves its operand
        if not (0 <= i and 0 <= j and h > 0 and w > 0 and i + h <= H and j + w <= W and flip in (0, 1)):
            raise RuntimeError(f"augment: crop box ({i}, {j}, {h}, {w}) leaves the {H} x {W} frame")
        if eh and not (0 <= et and 0 <= el and eh > 0 and ew > 0 and et + eh <= size and el + ew <= size):
            raise RuntimeError(f"augment: erase box ({et}, {el}, {eh}, {ew}) leaves the {size} x {size} output")
    any_erase = bool((pc[:, 7] > 0).any())
    if any_erase and noise is None:
        noise = torch.randn((B, T, 3, size, size), dtype=torch.float32, device=frames.device)
    if noise is not None and (not noise.is_cuda or noise.dtype != torch.float32 or tuple(noise.shape) != (B, T, 3, size, size) or not noise.is_contiguous()):
        raise RuntimeError(f"augment: noise must be a contiguous fp32 [{B}, {T}, 3, {size}, {size}] GPU tensor")
    pd = params.to(frames.device).contiguous()
    if out is None:
        out = torch.empty((B, 3, T, size, size), dtype=torch.float32, device=frames.device)
    elif tuple(out.shape) != (B, 3, T, size, size) or out.dtype != torch.float32 or not out.is_contiguous() or not out.is_cuda:
        raise RuntimeError("augment: bad `out`")
    import ctypes as C
    m3, s3 = (C.c_float * 3)(*[float(x) for x in mean]), (C.c_float * 3)(*[float(x) for x in std])
    with torch.cuda.device(frames.device):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().stg_video_aug(frames.data_ptr(), B, T, H, W, pd.data_ptr(), noise.data_ptr() if noise is not None else None,
                                            C.cast(m3, C.c_void_p), C.cast(s3, C.c_void_p), out.data_ptr(), int(size), st), "stg_video_aug")
    return out
