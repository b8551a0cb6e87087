"""Waveform ingest in front of the hot path, MI355X path: a list of float16 / float32
clips of different lengths at the dataset's sample rate becomes the resident ``wav`` (B, Lmax) float32 tensor and the
``wav_len`` array the models read - conversion, resampling (``torchaudio.functional.resample`` semantics,
caption_dataset.py:110-120), the random crop / zero pad to ``audio_duration`` (caption_dataset.py:121-129) and
zero-padding (``WavPadCollate``, inference.py:81-111, with its ``min_duration`` blacklist) in one kernel pass
(csrc/ingest.hip).  The host only concatenates the raw samples into one pinned buffer and draws the crop offsets.

The HDF5 side (``read_from_h5``, caption_dataset.py:131-145) stays with the caller: h5py is not part of this image; its
float16 arrays are taken as they are (the float16 -> float32 conversion happens in the kernel)."""
import math
import random

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

    def __init__(self, orig_sr, target_sr, min_duration=0.32, device="cuda", audio_duration=None, rng=None):
        """``audio_duration`` (seconds, caption_dataset.py:58-66): every clip comes out with exactly
        ``int(audio_duration * target_sr)`` samples - longer clips are cropped at an offset drawn uniformly from
        ``0 .. excess`` INCLUSIVE like ``random.randint(0, excess)`` (caption_dataset.py:124; one draw per longer clip in
        batch order, like the dataset's ``__getitem__`` calls), shorter ones zero-padded.  ``rng``: the ``random`` module
        (default), a ``random.Random``, a ``numpy.random.Generator`` or a ``numpy.random.RandomState`` - each is asked through
        its own inclusive-range call (``_draw_offset``).

        One instance serves one caller at a time: ``__call__`` holds a lock (the two pinned staging buffers, their events and
        the turn counter are shared state); loader threads that should pack in parallel take one instance each."""
        import threading
        self._lock = threading.Lock()
        self.orig_sr, self.target_sr = int(orig_sr), int(target_sr)
        self.min_length = int(min_duration * target_sr)
        self.num_audio_samples = int(audio_duration * target_sr) if audio_duration is not None else None
        self.rng = rng if rng is not None else random
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

    def _draw_offset(self, excess):
        """Uniform integer in [0, excess], whatever the generator's own convention for the upper bound is."""
        rng = self.rng
        if hasattr(rng, "integers"):                       # numpy.random.Generator: upper bound exclusive
            return int(rng.integers(0, excess + 1))
        if hasattr(rng, "randrange"):                      # random / random.Random: randint(0, n) == randrange(n + 1)
            return int(rng.randint(0, excess))
        if hasattr(rng, "randint"):                        # numpy.random.RandomState: upper bound exclusive
            return int(rng.randint(0, excess + 1))
        raise TypeError("rng must be the random module, a random.Random, a numpy Generator or a numpy RandomState")

    def _staging(self, n, dtype):
        """One of two page-locked staging buffers of at least n elements (kept between calls, grown by doubling); the one
        handed out is the one whose last upload was issued longer ago, after that upload has finished."""
        if not hasattr(self, "_staging_bufs"):
            self._staging_bufs, self._staging_events, self._staging_turn = [None, None], [None, None], 0
        self._staging_turn ^= 1
        k = self._staging_turn
        if self._staging_events[k] is not None:
            self._staging_events[k].synchronize()
        buf = self._staging_bufs[k]
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            buf = self._staging_bufs[k] = torch.empty(max(n, 2 * (buf.numel() if buf is not None and buf.dtype == dtype else 0)),
                                                      dtype=dtype).pin_memory()
        return buf

    def _pool(self):
        if getattr(self, "_threads", None) is None:
            import concurrent.futures
            import os
            self._threads = concurrent.futures.ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 4) // 2)))
        return self._threads

    def out_length(self, n):
        return int(math.ceil(self.new * n / self.orig))

    def __call__(self, data_list):
        """data_list: [(audio_id, 1-D float16/float32 array or None), ...] -> {"aid", "wav" (device), "wav_len",
        "blacklist_aid"} with the keys of ``WavPadCollate``."""
        if self.device.type != "cuda":
            raise _lib.HipLibraryError("WaveformIngest runs on a ROCm device; there is no CPU fallback")
        with self._lock:
            return self._ingest(data_list)

    def _ingest(self, data_list):
        lib = _lib.load()
        aids, clips, lens, black, starts, kept = [], [], [], [], [], []
        n_fix = self.num_audio_samples
        for aid, wav in data_list:
            olen = 0 if wav is None else self.out_length(len(wav))
            # the dataset crops / pads BEFORE the collate judges the length (caption_dataset.py:121-129, inference.py:95-99)
            final = olen if (n_fix is None or wav is None) else n_fix
            if wav is None or final < self.min_length:
                black.append(aid)
                continue
            aids.append(aid)
            clips.append(np.asarray(wav))
            lens.append(olen)
            starts.append(self._draw_offset(olen - n_fix) if n_fix is not None and olen > n_fix else 0)
            kept.append(final)
        if not clips:
            raise ValueError("every clip is shorter than min_duration")
        half = all(c.dtype == np.float16 for c in clips)
        dt = np.float16 if half else np.float32
        offs = np.zeros(len(clips) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(c) for c in clips])
        # Host packing is the slow half of this row (56 MB per 64 ten-second float16 clips): the pinned staging buffer is kept
        # between calls (a fresh page-locked allocation costs more than the copy into it) and the clips are copied into it by
        # a few threads (numpy releases the GIL inside the copy).  Two buffers alternate, so that the asynchronous upload of one
        # batch may still be reading its buffer while the next batch is being packed.
        host = self._staging(int(offs[-1]), torch.float16 if half else torch.float32)
        hv = host.numpy()

        def put(k):
            hv[offs[k]:offs[k + 1]] = clips[k] if clips[k].dtype == dt else clips[k].astype(dt)

        if len(clips) >= 8 and int(offs[-1]) >= (1 << 21):
            list(self._pool().map(put, range(len(clips))))
        else:
            for k in range(len(clips)):
                put(k)
        src = host[:int(offs[-1])].to(self.device, non_blocking=True)
        self._uploaded = torch.cuda.Event()
        self._uploaded.record()
        self._staging_events[self._staging_turn] = self._uploaded
        B, lmax = len(clips), max(kept)
        out = torch.empty(B, lmax, device=self.device, dtype=torch.float32)
        off_dev = torch.from_numpy(offs).to(self.device)
        meta = torch.tensor([lens, starts], dtype=torch.int32).to(self.device)   # one upload: resampled lengths, crop offsets
        check(lib.ac_ingest_resample(ptr(src), int(half), ptr(off_dev), ptr(self.kernel), ptr(self.tap_lo), ptr(self.tap_hi),
                                     ptr(out), ptr(meta[0]), ptr(meta[1]) if n_fix is not None else None, B, lmax, self.orig,
                                     self.new, self.width, stream()), "ac_ingest_resample")
        return {"aid": np.array(aids), "wav": out, "wav_len": np.array(kept), "blacklist_aid": black}
