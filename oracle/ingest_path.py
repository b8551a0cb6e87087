"""TEST INFRASTRUCTURE ONLY - CPU restatement of the waveform ingest in front of the hot path (SURVEY.md section 8(f)
rank 1): float16 HDF5 samples -> float32 (caption_dataset.py:131-145), ``torchaudio.functional.resample(wav, orig_sr,
target_sr)`` (caption_dataset.py:110-120; Clotho 44.1 kHz -> 32 kHz, train_data.yaml:11-12) and the zero-padding
collate with its ``min_duration`` blacklist (``WavPadCollate``, inference.py:81-111).

The collate is the reference's own code (pinned by restating it).  The resampler's arithmetic is torchaudio==0.13.1's
``_get_sinc_resample_kernel`` / ``_apply_sinc_resample_kernel`` (not vendored, not installed here); restated from the
published algorithm - windowed-sinc polyphase filter, ``lowpass_filter_width=6``, ``rolloff=0.99``, Hann window
(``sinc_interpolation``), computed in float64 and cast like torchaudio does when no dtype is given.  Pinned by an
independent witness (tests/test_ingest_oracle.py): the filter bank and resampled signals of the committed fixture
tests/golden/g13_resample.npz - the published prototype evaluated in float64 on the fine grid with numpy and run through
``scipy.signal.upfirdn`` by tests/golden/make_resample_golden.py, which imports neither this file nor the product's table
builder - plus closed-form properties (DC gain, tap count, sinusoid amplitudes).  torchaudio itself is not installed in
the build image, so no run of the LIBRARY backs the formula: **witness-pinned, not reference-pinned**.
The crop / pad to ``audio_duration`` (caption_dataset.py:121-129) is the reference's own code, restated.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """(new, 2*width + orig) float32 polyphase filter bank and ``width`` for the gcd-reduced frequencies."""
    gcd = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // gcd, int(new_freq) // gcd
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32)[:, 0], width, orig, new


def resample(wav, orig_freq, new_freq):
    """wav (..., L) float32 -> (..., ceil(new * L / orig))."""
    if orig_freq == new_freq:
        return wav
    kernel, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    shape = wav.shape
    x = wav.reshape(-1, shape[-1])
    length = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernel[:, None], stride=orig)            # (n, new, blocks)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


def wav_pad_collate(data_list, min_duration=0.32, sample_rate=32000):
    """``WavPadCollate.__call__`` (inference.py:81-111): items shorter than min_duration are blacklisted, the rest are
    zero-padded to the longest."""
    min_length = int(min_duration * sample_rate)
    aids, wavs, lens, black = [], [], [], []
    for aid, wav in data_list:
        if wav is None or len(wav) < min_length:
            black.append(aid)
            continue
        aids.append(aid)
        wavs.append(wav)
        lens.append(wav.shape[0])
    out = np.zeros((len(wavs), max(lens)))
    for i, w in enumerate(wavs):
        out[i, :len(w)] = w
    return {"aid": np.array(aids), "wav": out, "wav_len": np.array(lens), "blacklist_aid": black}


def crop_or_pad(wav, num_audio_samples, rng):
    """``process_waveform`` after the resampler (caption_dataset.py:121-129): a random window of num_audio_samples from a
    longer clip (``random.randint`` is inclusive on both ends), zeros behind a shorter one."""
    n = wav.shape[0]
    if n > num_audio_samples:
        start = rng.randint(0, n - num_audio_samples)
        return wav[start:start + num_audio_samples]
    if n < num_audio_samples:
        return np.concatenate([wav, np.zeros(num_audio_samples - n, dtype=wav.dtype)])
    return wav


def ingest(data_list, orig_sr, target_sr, min_duration=0.32, audio_duration=None, rng=None):
    """float16/float32 clips at orig_sr -> what the model's input_dict needs: resample each clip, crop / pad it to
    ``audio_duration`` when that is set (one ``rng.randint`` per longer clip, in order), then collate."""
    import random
    rng = rng if rng is not None else random
    items = []
    for aid, wav in data_list:
        if wav is None:
            items.append((aid, None))
            continue
        w = torch.as_tensor(np.array(wav, dtype=np.float32))
        y = resample(w[None], orig_sr, target_sr)[0].numpy()
        if audio_duration is not None:
            y = crop_or_pad(y, int(audio_duration * target_sr), rng)
        items.append((aid, y))
    return wav_pad_collate(items, min_duration, target_sr)
