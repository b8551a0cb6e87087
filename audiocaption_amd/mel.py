"""Host-side constants of the log-mel front-end: Hann window, DFT twiddles, mel filterbank.

The arithmetic being replaced lives in torchaudio==0.13.1 (``MelSpectrogram`` / ``melscale_fbanks``,
called from reference cnn_encoder.py:338-348 and hf_wrapper.py:270-277), which the reference does not
vendor.  The tables are computed in fp32 with the same operation order torchaudio uses so that the
HIP kernel sees bit-identical filter weights; everything here is tiny, one-off, host-side setup.
"""
import math

import numpy as np
import torch
import torch.nn as nn


def _hz_to_mel(freq, mel_scale):
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    if freq >= min_log_hz:
        return min_log_hz / f_sp + math.log(freq / min_log_hz) / (math.log(6.4) / 27.0)
    return freq / f_sp


def _mel_to_hz(mels, mel_scale):
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    freqs = f_sp * mels
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm, mel_scale):
    """(n_freqs, n_mels) fp32 triangular filterbank."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min, mel_scale), _hz_to_mel(f_max, mel_scale), n_mels + 2)
    f_pts = _mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    if norm == "slaney":
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.contiguous()


class _OptionalBuffers(nn.Module):
    """Buffers that a checkpoint may or may not carry: absent keys are not reported as missing (the values are
    constants of the front-end, recomputed at construction)."""

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        n = len(missing_keys)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        del missing_keys[n:]


class MelSpectrogramBuffers(nn.Module):
    """The two buffers torchaudio's ``MelSpectrogram`` registers - ``spectrogram.window`` and ``mel_scale.fb`` - under the
    attribute name the reference encoders use (``melspec_extractor``, cnn_encoder.py:338-348, hf_wrapper.py:270-277):
    reference checkpoints carry ``...melspec_extractor.spectrogram.window`` / ``...melspec_extractor.mel_scale.fb`` (the
    trainer saves every buffer, run.py:209-216) and must load with ``strict=True``; checkpoints written here carry them
    too.  The HIP log-mel kernel reads its window and filterbank from these buffers.  Never called."""

    def __init__(self, sample_rate, n_fft, f_min, f_max, n_mels, norm, mel_scale):
        super().__init__()
        with torch.device("cpu"):   # real values even when the model is being built under a meta-device context (HF loaders)
            window = torch.hann_window(n_fft, periodic=True)
            fb = melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate, norm, mel_scale)
        self.spectrogram = _OptionalBuffers()
        self.spectrogram.register_buffer("window", window)
        self.mel_scale = _OptionalBuffers()
        self.mel_scale.register_buffer("fb", fb)

    def key(self):
        w, fb = self.spectrogram.window, self.mel_scale.fb
        return (w.device, w.data_ptr(), w._version, fb.data_ptr(), fb._version)


class MelTables:
    """Device-resident tables for ac_logmel.  ``window`` / ``fb``: take these tensors (the module buffers a checkpoint
    may have overwritten) instead of recomputing them."""

    def __init__(self, sample_rate, n_fft, hop, f_min, f_max, n_mels, norm, mel_scale, device, window=None, fb=None):
        if n_mels != 64:
            raise ValueError("the HIP log-mel kernel is built for 64 mel bins")
        self.sample_rate, self.n_fft, self.hop = sample_rate, n_fft, hop
        if fb is None:
            fb = melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate, norm, mel_scale)
        fb = fb.detach().float().cpu()
        nz = fb > 0
        lo = torch.where(nz.any(0), nz.float().argmax(0), torch.zeros(n_mels, dtype=torch.long))
        hi = torch.where(nz.any(0), n_fft // 2 - nz.flip(0).float().argmax(0),
                         torch.full((n_mels,), -1, dtype=torch.long))
        n = np.arange(n_fft, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * n / n_fft), -np.sin(2 * np.pi * n / n_fft)], axis=1)
        self.window = (torch.hann_window(n_fft, periodic=True) if window is None
                       else window.detach().float().cpu().contiguous()).to(device)
        self.twiddle = torch.from_numpy(tw.astype(np.float32)).contiguous().to(device)
        self.melfb = fb.to(device)
        self.mel_lo = lo.to(torch.int32).to(device)
        self.mel_hi = hi.to(torch.int32).to(device)
