"""-m gpu: stg_fbank (the device audio front end, AVE/dataloader.py:204-272) against the fp64 oracle (oracle/fbank.py; parity
unpinned, see its header).  Bound: 2e-3 absolute on the normalised log-mel features (values of order 1; fp32 direct DFT)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _waves(S, n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = rng.standard_normal((S, n)) * 0.05
    for s in range(S):
        x[s] += 0.2 * np.sin(2 * np.pi * (200.0 + 700.0 * s) * t) + 0.01 * (s - 1)      # a tone per segment and a DC offset
    return x.astype(np.float32)


@pytest.mark.parametrize("model_type,shape", [("MM-Swin-AVE-Base", (224, 224)), ("MM-CLIP-AVE", (102, 128))])
def test_wav2fbank_matches_oracle(stg, gpu, model_type, shape):
    import oracle.fbank as OF
    from stgcma import audio
    x = _waves(5, 16000, 0)
    got = audio.wav2fbank(torch.as_tensor(x).to(gpu), model_type, melbins=128, target_length=1024)
    assert tuple(got.shape) == (5,) + shape and got.dtype == torch.float32
    for s in range(5):
        ref = OF.wav2fbank(x[s], swin=model_type.startswith("MM-Swin"), melbins=128, target_length=1024)
        err = np.abs(got[s].cpu().numpy() - ref).max()
        assert err <= 2e-3, (s, err)
    frames = 223 if shape[0] == 224 else 98
    assert torch.all(got[:, frames:] == 0)                                      # zero rows behind the last frame


def test_fbank_geometry_edges(stg, gpu):
    import oracle.fbank as OF
    from stgcma import audio
    x = _waves(3, 9000, 1)
    wide = torch.zeros((3, 12000), dtype=torch.float32)
    wide[:, :9000] = torch.as_tensor(x)
    view = wide.to(gpu)[:, :9000]                                               # row stride 12000 > n
    got = audio.fbank(view, num_mel_bins=64, frame_shift=10.0, target_frames=20)  # fewer target frames than the 54 available: cropped
    ref = np.stack([OF.kaldi_fbank(x[s], 16000.0, 64, 10.0)[:20] / (2 * 0.5) for s in range(3)])
    assert tuple(got.shape) == (3, 20, 64) and np.abs(got.cpu().numpy() - ref).max() <= 2e-3
    every = audio.fbank(view, num_mel_bins=64, frame_shift=10.0)                 # default: every frame
    assert every.shape[1] == 1 + (9000 - 400) // 160
    short = audio.fbank(torch.zeros((2, 300), device=gpu), num_mel_bins=64, target_frames=4)    # shorter than one frame: all padding
    assert tuple(short.shape) == (2, 4, 64) and torch.all(short == 0)
    with pytest.raises(RuntimeError):
        audio.fbank(torch.zeros(2, 16000))                                      # CPU tensor: no fallback
