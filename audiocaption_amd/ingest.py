"""Waveform ingest in front of the hot path, MI355X path: a list of float16 / float32
clips of different lengths at the dataset's sample rate becomes the resident ``wav`` (B, Lmax) float32 tensor and the
``wav_len`` array the models read - conversion, resampling (``torchaudio.functional.resample`` semantics,
caption_dataset.py:110-120) and zero-padding (``WavPadCollate``, inference.py:81-111, with its ``min_duration``
blacklist) in one kernel pass (csrc/ingest.hip).  The host only concatenates the raw samples into one pinned buffer."""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream


def _sinc_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio==0.13.1 ``_get_sinc_resample_kernel`` (sinc_interpolation / Hann), float64 then float32."""
    gcd = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // gcd, int(new_freq) // gcd
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None] / orig
    t = (torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx) * base
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    return k.to(torch.float32).contiguous(), width, orig, new


class WaveformIngest:

    def __init__(self, orig_sr, target_sr, min_duration=0.32, device="cuda"):
        self.orig_sr, self.target_sr = int(orig_sr), int(target_sr)
        self.min_length = int(min_duration * target_sr)
        self.device = torch.device(device)
        if self.orig_sr != self.target_sr:
            k, self.width, self.orig, self.new = _sinc_kernel(self.orig_sr, self.target_sr)
            nz = k != 0
            lo = torch.where(nz.any(1), nz.float().argmax(1), torch.zeros(k.shape[0], dtype=torch.long))
            hi = torch.where(nz.any(1), k.shape[1] - nz.flip(1).float().argmax(1), torch.zeros(k.shape[0], dtype=torch.long))
            self.kernel = k.to(self.device)
            self.tap_lo, self.tap_hi = lo.to(torch.int32).to(self.device), hi.to(torch.int32).to(self.device)
        else:
            self.width, self.orig, self.new = 0, 1, 1
            self.kernel = self.tap_lo = self.tap_hi = None

    def out_length(self, n):
        return int(math.ceil(self.new * n / self.orig))

    def __call__(self, data_list):
        """data_list: [(audio_id, 1-D float16/float32 array or None), ...] -> {"aid", "wav" (device), "wav_len",
        "blacklist_aid"} with the keys of ``WavPadCollate``."""
        if self.device.type != "cuda":
            raise _lib.HipLibraryError("WaveformIngest runs on a ROCm device; there is no CPU fallback")
        lib = _lib.load()
        aids, clips, lens, black = [], [], [], []
        for aid, wav in data_list:
            olen = 0 if wav is None else self.out_length(len(wav))
            if wav is None or olen < self.min_length:
                black.append(aid)
                continue
            aids.append(aid)
            clips.append(np.asarray(wav))
            lens.append(olen)
        if not clips:
            raise ValueError("every clip is shorter than min_duration")
        half = all(c.dtype == np.float16 for c in clips)
        dt = np.float16 if half else np.float32
        offs = np.zeros(len(clips) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(c) for c in clips])
        host = torch.empty(int(offs[-1]), dtype=torch.float16 if half else torch.float32).pin_memory()
        hv = host.numpy()
        for c, o in zip(clips, offs[:-1]):
            hv[o:o + len(c)] = c.astype(dt, copy=False)
        src = host.to(self.device, non_blocking=True)
        B, lmax = len(clips), max(lens)
        out = torch.empty(B, lmax, device=self.device, dtype=torch.float32)
        off_dev = torch.from_numpy(offs).to(self.device)
        len_dev = torch.tensor(lens, dtype=torch.int32, device=self.device)
        check(lib.ac_ingest_resample(ptr(src), int(half), ptr(off_dev), ptr(self.kernel), ptr(self.tap_lo), ptr(self.tap_hi),
                                     ptr(out), ptr(len_dev), B, lmax, self.orig, self.new, self.width, stream()),
              "ac_ingest_resample")
        return {"aid": np.array(aids), "wav": out, "wav_len": np.array(lens), "blacklist_aid": black}
