"""ORACLE (test infrastructure, not product code): numpy restatement of the reference's audio front end, AVE/dataloader.py:204-272
(`_wav2fbank`): `torchaudio.compliance.kaldi.fbank(waveform, htk_compat=True, sample_frequency=sr, use_energy=False,
window_type='hanning', num_mel_bins=224 | 128, dither=0.0, frame_shift=4.4 | 10)`, then `(fbank - mean) / (2 std)` and zero-padding /
cropping to `target_length` frames.

PARITY UNPINNED.  The algorithm lives in a third-party dependency that is absent from /root/reference and from this image: torchaudio
(the reference pins no version; its README installs the wheel matching torch 1.x; `compliance/kaldi.py` is stable since 0.8).  This
file restates that module's published algorithm -- Kaldi's `compute-fbank-feats` with the options above and every other option at
its default (frame_length 25 ms, preemphasis 0.97, remove_dc_offset, snip_edges, round_to_power_of_two, low_freq 20 Hz, high_freq =
Nyquist, power spectrum, log with the fp32 epsilon floor, no VTLN warp, no energy column) -- in float64; no output of torchaudio
itself could be generated here, and the reference's tests hold no fbank fixture.  What the tests do pin: the restatement against an
independent scipy construction of the same chain, known answers (a pure tone lands in the mel bin that contains its frequency; a
constant offset changes nothing), and the HIP kernel against this file.

Only tests/ may import this; the product path (stg-cma_amd/audio.py -> stg_fbank) never does.
"""
import math

import numpy as np

EPS = float(np.finfo(np.float32).eps)          # torchaudio: torch.finfo(dtype).eps of the fp32 waveform


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def frame_params(sample_rate, frame_length_ms=25.0, frame_shift_ms=10.0):
    """(window_shift, window_size, padded_window_size) in samples (kaldi.py `_get_waveform_and_window_properties`)."""
    shift = int(sample_rate * frame_shift_ms * 0.001)
    size = int(sample_rate * frame_length_ms * 0.001)
    padded = 1 << (size - 1).bit_length()
    return shift, size, padded


def mel_banks(num_bins, padded_window_size, sample_rate, low_freq=20.0, high_freq=0.0):
    """[num_bins, padded/2 + 1] triangular weights on the mel scale (kaldi.py `get_mel_banks`, vtln_warp = 1; the last column, the
    Nyquist bin, is the zero padding `fbank` appends)."""
    num_fft_bins = padded_window_size // 2
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_rate / padded_window_size
    mel_low, mel_high = float(mel_scale(low_freq)), float(mel_scale(high_freq))
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    mel = mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=np.float64))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([w, np.zeros((num_bins, 1))], axis=1)


def hann_window(size):
    """torch.hann_window(size, periodic=False) -- window_type='hanning'."""
    return 0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(size, dtype=np.float64) / (size - 1))


def kaldi_fbank(wave, sample_rate=16000.0, num_mel_bins=128, frame_shift_ms=10.0, frame_length_ms=25.0, preemphasis=0.97,
                low_freq=20.0, high_freq=0.0):
    """wave: 1-D float array -> [frames, num_mel_bins] log-mel energies (float64)."""
    x = np.asarray(wave, dtype=np.float64)
    shift, size, padded = frame_params(sample_rate, frame_length_ms, frame_shift_ms)
    if x.shape[0] < size:
        return np.zeros((0, num_mel_bins))
    m = 1 + (x.shape[0] - size) // shift                                   # snip_edges
    idx = np.arange(size)[None, :] + shift * np.arange(m)[:, None]
    fr = x[idx]
    fr = fr - fr.mean(axis=1, keepdims=True)                               # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)                 # replicate-padded shift by one
    fr = fr - preemphasis * prev
    fr = fr * hann_window(size)[None, :]
    fr = np.concatenate([fr, np.zeros((m, padded - size))], axis=1)
    spec = np.abs(np.fft.rfft(fr, axis=1)) ** 2                            # use_power
    mel = spec @ mel_banks(num_mel_bins, padded, sample_rate, low_freq, high_freq).T
    return np.log(np.maximum(mel, EPS))                                    # use_log_fbank


def wav2fbank(wave, *, swin=True, sample_rate=16000.0, melbins=128, norm_mean=-4.1426, norm_std=3.2001, target_length=1024):
    """AVE/dataloader.py:236-270 for one waveform segment: Swin backbones take 224 mel bins at a 4.4 ms shift and 224 frames,
    the CLIP backbones `melbins` at 10 ms and target_length // 10 frames; normalisation by the dataset statistics."""
    fb = kaldi_fbank(wave, sample_rate, 224 if swin else melbins, 4.4 if swin else 10.0)
    fb = (fb - norm_mean) / (norm_std * 2)
    tl = 224 if swin else int(target_length * (1 / 10))
    out = np.zeros((tl, fb.shape[1]))
    n = min(tl, fb.shape[0])
    out[:n] = fb[:n]
    return out
