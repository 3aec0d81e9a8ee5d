"""-m gpu: stg_video_aug (csrc/video.hip) against the fixture produced by the reference's own frame pipeline with its random draws recorded
(tests/golden/video_aug.npz) and against the oracle on a batch; SURVEY f4, video half (AVE/dataloader.py:346-394)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _noise_full(z, c, T):
    p = z[f"params{c}"]
    n = torch.zeros((T, 3, 224, 224))
    if p[7] > 0:
        n[:, :, p[5]:p[5] + p[7], p[6]:p[6] + p[8]] = torch.as_tensor(z[f"noise_box{c}"])
    return n


def test_video_aug_matches_reference_fixture(stg, gpu):
    from stgcma import video
    z = np.load(os.path.join(GOLD, "video_aug.npz"))
    for c in range(int(z["ncases"][0])):
        fr = torch.as_tensor(z[f"frames{c}"])
        T = fr.shape[0]
        out = video.augment(fr[None].to(gpu), torch.as_tensor(z[f"params{c}"])[None], _noise_full(z, c, T)[None].to(gpu),
                            mean=z["mean"].tolist(), std=z["std"].tolist())
        ref = torch.as_tensor(z[f"out{c}"])
        assert tuple(out.shape) == (1, 3, T, 224, 224)
        # fp32 throughout; the only freedom is the association of the four-tap sum: 1e-5 absolute on values of O(1)
        assert float((out[0].cpu() - ref).abs().max()) <= 1e-5, c


def test_video_aug_batch_matches_oracle_and_rejects_bad_boxes(stg, gpu):
    import random
    import oracle.video_aug as OV
    from stgcma import video
    g = torch.Generator().manual_seed(5)
    B, T, H, W = 5, 3, 200, 260
    fr = torch.randint(0, 256, (B, T, H, W, 3), generator=g, dtype=torch.uint8)
    p = video.draw_params(B, H, W, rng=random.Random(3), erase_prob=0.7)
    noise = torch.randn((B, T, 3, 224, 224), generator=g)
    out = video.augment(fr.to(gpu), p, noise.to(gpu)).cpu()
    for b in range(B):
        i, j, h, w, flip, et, el, eh, ew = p[b].tolist()
        ref = OV.video_aug(fr[b], p[b], noise[b, :, :, et:et + eh, el:el + ew], video.IMAGENET_MEAN, video.IMAGENET_STD)
        assert float((out[b] - ref).abs().max()) <= 1e-5, b
    bad = p.clone()
    bad[0, 2] = H + 1
    with pytest.raises(RuntimeError):
        video.augment(fr.to(gpu), bad, noise.to(gpu))
    with pytest.raises(RuntimeError):
        video.augment(fr, p)                              # CPU frames: no fallback
