"""Audio front end on the device: the `_wav2fbank` step of the reference's data loader (AVE/dataloader.py:204-272) for waveforms
that are already in memory -- Kaldi-compatible log-mel filterbank features, dataset normalisation, padding / cropping to the models'
spectrogram shape -- as ONE launch of stg_fbank for all segments of a batch.

File decoding (torchaudio.load), the choice of the 1-second segment (:231-233) and mixup of two waveforms (:211-229: a weighted sum of
two sample arrays) stay with the caller: they are host I/O and two lines of tensor arithmetic.  `waveform - waveform.mean()` (:208)
needs no counterpart: the per-frame DC removal of the filterbank makes the features invariant to a constant offset.

No CPU fallback: CPU tensors raise.  The oracle (oracle/fbank.py) is not imported here.
"""
import math

import numpy as np
import torch

from . import _lib

_tables = {}


def _frame_params(sample_rate, frame_length_ms, frame_shift_ms):
    shift = int(sample_rate * frame_shift_ms * 0.001)
    size = int(sample_rate * frame_length_ms * 0.001)
    return shift, size, 1 << (size - 1).bit_length()


def _mel_weights(num_bins, padded, sample_rate, low_freq, high_freq):
    """Triangular mel weights [num_bins, padded / 2 + 1] (kaldi.py get_mel_banks without VTLN; last column = the zero Nyquist pad)."""
    nfft = padded // 2
    if high_freq <= 0.0:
        high_freq += 0.5 * sample_rate

    def mel(f):
        return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    lo, hi = float(mel(low_freq)), float(mel(high_freq))
    d = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    m = mel(sample_rate / padded * np.arange(nfft, dtype=np.float64))[None, :]
    w = np.maximum(0.0, np.minimum((m - (lo + b * d)) / d, ((lo + (b + 2.0) * d) - m) / d))
    return np.concatenate([w, np.zeros((num_bins, 1))], axis=1)


def _get_tables(device, sample_rate, num_mel_bins, frame_length_ms, frame_shift_ms, low_freq, high_freq):
    key = (str(device), float(sample_rate), int(num_mel_bins), float(frame_length_ms), float(frame_shift_ms), float(low_freq), float(high_freq))
    t = _tables.get(key)
    if t is None:
        shift, size, padded = _frame_params(sample_rate, frame_length_ms, frame_shift_ms)
        win = 0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(size, dtype=np.float64) / (size - 1))       # torch.hann_window(periodic=False)
        mw = _mel_weights(num_mel_bins, padded, sample_rate, low_freq, high_freq)
        t = _tables[key] = (shift, size, padded, torch.tensor(win, dtype=torch.float32, device=device),
                            torch.tensor(mw, dtype=torch.float32, device=device).contiguous())
    return t


def fbank(wave, *, sample_rate=16000.0, num_mel_bins=128, frame_shift=10.0, frame_length=25.0, preemphasis=0.97, low_freq=20.0,
          high_freq=0.0, norm_mean=0.0, norm_std=0.5, target_frames=None):
    """wave: fp32 [S, n] GPU tensor of S equally long waveform segments -> fp32 [S, target_frames, num_mel_bins]:
    kaldi.fbank(htk_compat=True, use_energy=False, window_type='hanning', dither=0) per segment, then (x - norm_mean) / (2 norm_std),
    rows beyond the segment's frames zero, frames beyond target_frames dropped (defaults: no normalisation, every frame)."""
    if not wave.is_cuda:
        raise RuntimeError("stgcma.audio.fbank runs on MI355X only: move the waveforms to the GPU (no CPU fallback)")
    if wave.dim() == 1:
        wave = wave[None]
    if wave.dim() != 2 or wave.dtype != torch.float32 or wave.stride(1) != 1:
        raise RuntimeError("fbank: expected fp32 [segments, samples] with unit sample stride")
    shift, size, padded, win, mw = _get_tables(wave.device, sample_rate, num_mel_bins, frame_length, frame_shift, low_freq, high_freq)
    S, n = wave.shape
    frames = 1 + (n - size) // shift if n >= size else 0
    tf = frames if target_frames is None else int(target_frames)
    if tf <= 0:
        return torch.zeros((S, 0, num_mel_bins), dtype=torch.float32, device=wave.device)
    out = torch.empty((S, tf, num_mel_bins), dtype=torch.float32, device=wave.device)
    with torch.cuda.device(wave.device):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().stg_fbank(wave.data_ptr(), n, wave.stride(0), S, shift, size, padded, win.data_ptr(), mw.data_ptr(),
                                        int(num_mel_bins), float(preemphasis), float(norm_mean), float(norm_std), tf, out.data_ptr(), st),
                   "stg_fbank")
    return out


def wav2fbank(wave, model_type="MM-Swin-AVE-Base", *, sample_rate=16000.0, melbins=128, norm_mean=-4.1426, norm_std=3.2001,
              target_length=1024):
    """AVE/dataloader.py:236-270: the Swin backbones take 224 mel bins at a 4.4 ms shift padded / cropped to 224 frames, the CLIP
    backbones `melbins` bins at 10 ms and target_length // 10 frames.  wave [S, n] -> [S, frames, bins] (stack T segments of B clips
    and view as [B, T, frames, bins] for the models)."""
    swin = model_type in ("MM-Swin-AVE-Base", "MM-Swin-AVE-Large")
    return fbank(wave, sample_rate=sample_rate, num_mel_bins=224 if swin else melbins, frame_shift=4.4 if swin else 10.0,
                 norm_mean=norm_mean, norm_std=norm_std, target_frames=224 if swin else int(target_length * (1 / 10)))
