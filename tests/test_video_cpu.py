"""SURVEY f4, video half: the oracle (oracle/video_aug.py) against the fixture produced by the REFERENCE's own functions with their random
draws recorded (tests/golden/video_aug.npz, make_golden.py::video_aug_case), and the host-side parameter draws against the reference's
distributions' support."""
import os
import random

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_matches_reference_fixture():
    import oracle.video_aug as OV
    z = np.load(os.path.join(GOLD, "video_aug.npz"))
    n = int(z["ncases"][0])
    kinds = set()
    for c in range(n):
        p = z[f"params{c}"]
        got = OV.video_aug(torch.as_tensor(z[f"frames{c}"]), p, z[f"noise_box{c}"], z["mean"], z["std"])
        ref = torch.as_tensor(z[f"out{c}"])
        assert got.shape == ref.shape and float((got - ref).abs().max()) <= 1e-6, c
        kinds.add((int(p[4]), int(p[7] > 0)))
    assert {(1, 1), (0, 1), (0, 0)} <= kinds          # flip + erase, erase only, neither


def test_parameter_draws_stay_inside_the_frames():
    import stgcma  # noqa: F401
    from stgcma import video
    rng = random.Random(7)
    p = video.draw_params(400, 240, 320, rng=rng)
    assert p.dtype == torch.int32 and tuple(p.shape) == (400, 9)
    i, j, h, w, flip, et, el, eh, ew = p.T.tolist()
    assert all(0 <= a and a + c <= 240 for a, c in zip(i, h)) and all(0 <= a and a + c <= 320 for a, c in zip(j, w))
    assert all(0.07 * 240 * 320 <= a * b <= 240 * 320 for a, b in zip(h, w))           # scale (0.08, 1) up to rounding
    assert 0.35 <= sum(flip) / 400 <= 0.65 and 0.12 <= sum(1 for e in eh if e) / 400 <= 0.40     # p = 0.5, p = 0.25
    assert all((e == 0 and f == 0) or (0 < e < 224 and 0 < f < 224 and a + e <= 224 and b + f <= 224) for e, f, a, b in zip(eh, ew, et, el))
