"""Oracle of the audio front end (oracle/fbank.py: torchaudio.compliance.kaldi.fbank as the reference calls it, AVE/dataloader.py:
236-270).  torchaudio is not in this image and the reference's tests hold no fbank fixture, so the restatement is checked against an
independently written construction of the same published algorithm and against known answers (PARITY UNPINNED, see the oracle's header)."""
import numpy as np
import scipy.fft

import oracle.fbank as OF


def _independent_fbank(x, sr, bins, shift_ms):
    """The same chain written differently: explicit loops for the framing and the mel triangles, scipy's FFT."""
    shift, size = int(sr * shift_ms * 0.001), int(sr * 0.025)
    padded = 1
    while padded < size:
        padded *= 2
    m = 1 + (len(x) - size) // shift
    win = np.array([0.5 - 0.5 * np.cos(2 * np.pi * j / (size - 1)) for j in range(size)])
    out = np.zeros((m, bins))
    hz2mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    lo, hi = hz2mel(20.0), hz2mel(sr / 2)
    edges = [lo + i * (hi - lo) / (bins + 1) for i in range(bins + 2)]
    melf = [hz2mel(k * sr / padded) for k in range(padded // 2)]
    W = np.zeros((bins, padded // 2 + 1))
    for b in range(bins):
        for k, mk in enumerate(melf):
            if edges[b] < mk < edges[b + 2]:
                W[b, k] = (mk - edges[b]) / (edges[b + 1] - edges[b]) if mk <= edges[b + 1] else (edges[b + 2] - mk) / (edges[b + 2] - edges[b + 1])
    for i in range(m):
        f = np.array(x[i * shift:i * shift + size], dtype=np.float64)
        f = f - f.mean()
        g = f.copy()
        g[1:] = f[1:] - 0.97 * f[:-1]
        g[0] = f[0] - 0.97 * f[0]
        buf = np.zeros(padded)
        buf[:size] = g * win
        p = np.abs(scipy.fft.rfft(buf)) ** 2
        out[i] = np.log(np.maximum(W @ p, np.finfo(np.float32).eps))
    return out


def test_oracle_matches_independent_construction():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(4000) * 0.1 + 0.05 * np.sin(2 * np.pi * 440 * np.arange(4000) / 16000)
    for bins, shift in ((224, 4.4), (128, 10.0)):
        a = OF.kaldi_fbank(x, 16000.0, bins, shift)
        b = _independent_fbank(x, 16000.0, bins, shift)
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 1e-9


def test_frame_geometry_of_the_reference_configurations():
    assert OF.frame_params(16000.0, 25.0, 4.4) == (70, 400, 512)            # int(16000 * 0.0044) = 70
    assert OF.frame_params(16000.0, 25.0, 10.0) == (160, 400, 512)
    one_second = np.zeros(16000)
    assert OF.kaldi_fbank(one_second + 1e-3, 16000.0, 224, 4.4).shape == (223, 224)     # 1 + (16000 - 400) // 70; padded to 224 rows (:259-263)
    assert OF.kaldi_fbank(one_second + 1e-3, 16000.0, 128, 10.0).shape == (98, 128)
    assert OF.kaldi_fbank(np.zeros(399), 16000.0, 128, 10.0).shape == (0, 128)          # shorter than one frame


def test_known_answers():
    sr, n = 16000.0, 16000
    t = np.arange(n) / sr
    tone = 0.3 * np.sin(2 * np.pi * 1000.0 * t)
    fb = OF.kaldi_fbank(tone, sr, 224, 4.4)
    mel = OF.mel_scale
    lo, hi = float(mel(20.0)), float(mel(8000.0))
    centers = lo + (np.arange(224) + 1.0) * (hi - lo) / 225.0
    want = int(np.argmin(np.abs(centers - float(mel(1000.0)))))
    assert abs(int(fb.mean(0).argmax()) - want) <= 1                            # the tone lands in the bin around 1 kHz
    assert np.abs(OF.kaldi_fbank(tone + 0.25, sr, 224, 4.4) - fb).max() <= 1e-6   # per-frame DC removal: a constant offset changes nothing
    assert np.all(OF.kaldi_fbank(np.zeros(n), sr, 128, 10.0) == np.log(OF.EPS))  # silence sits on the epsilon floor
    w = OF.mel_banks(224, 512, sr)
    assert w.shape == (224, 257) and np.all(w[:, 256] == 0) and w.min() >= 0 and w.max() <= 1.0
    # 224 triangles on 256 FFT bins: below ~600 Hz a triangle is narrower than a bin and some catch none -- those mel bins sit on the
    # epsilon floor whatever the input (a property of the reference's 224-bin / 512-point configuration, reproduced as is)
    empty = (w > 0).sum(1) == 0
    assert 0 < empty.sum() < 40 and not empty[60:].any()
    assert np.all(fb[:, empty] == np.log(OF.EPS))


def test_wav2fbank_normalises_then_pads():
    x = np.random.default_rng(1).standard_normal(16000) * 0.05
    out = OF.wav2fbank(x, swin=True, norm_mean=-4.1426, norm_std=3.2001)
    fb = OF.kaldi_fbank(x, 16000.0, 224, 4.4)
    assert out.shape == (224, 224)
    assert np.abs(out[:223] - (fb + 4.1426) / (2 * 3.2001)).max() <= 1e-12
    assert np.all(out[223] == 0)                                                # the zero row is appended AFTER the normalisation
    clip = OF.wav2fbank(x, swin=False, melbins=128, target_length=1024)
    assert clip.shape == (102, 128) and np.all(clip[98:] == 0)
    long = OF.wav2fbank(np.concatenate([x, x]), swin=True)                      # more frames than target_length: cropped
    assert long.shape == (224, 224)


def test_host_mirror_builds_the_oracles_tables():
    """stg-cma_amd/audio.py builds window / mel-weight tables itself (it may not import the oracle): same numbers."""
    import conftest  # noqa: F401
    import stgcma.audio as A
    for bins, shift in ((224, 4.4), (128, 10.0)):
        sh, size, padded = A._frame_params(16000.0, 25.0, shift)
        assert (sh, size, padded) == OF.frame_params(16000.0, 25.0, shift)
        assert np.abs(A._mel_weights(bins, padded, 16000.0, 20.0, 0.0) - OF.mel_banks(bins, padded, 16000.0)).max() <= 1e-12
