"""Writes tests/golden/g13_resample.npz: the polyphase filter bank and resampled signals of
``torchaudio.functional.resample`` (torchaudio 0.13.1 defaults: sinc_interpolation = Hann window, lowpass_filter_width 6,
rolloff 0.99; call site caption_dataset.py:110-120) for the rate pairs the ingest row meets, from an INDEPENDENT evaluation:
numpy / scipy only, the published windowed-sinc prototype h(k) = (f0 / orig) sinc(f k) cos^2(pi f k / 12),
f = 0.99 min(orig, new) / (orig new), k clipped to |f k| <= 6, evaluated in float64 on the FINE grid (rate orig * new) -

* ``bank_<o>_<n>`` [new][2 width + orig]: bank[i][j] = h((j - width) new - i orig), cast to float32 like torchaudio does;
* ``y_<o>_<n>``: scipy.signal.upfirdn(h, x, up=new) sampled every ``orig`` fine steps (float64), for the seeded ``x_<o>_<n>``.

Neither ``audiocaption_amd.ingest._sinc_kernel`` nor ``oracle/ingest_path.py`` is imported here: the product's table and the
oracle's are both checked AGAINST this file (tests/test_ingest_oracle.py, tests/test_gpu_ingest.py).  torchaudio itself is
not installed in the build image, so the formula is the published one, not a run of the library.

    python tests/golden/make_resample_golden.py
"""
import math
import os

import numpy as np
from scipy.signal import upfirdn

PAIRS = [(44100, 32000), (48000, 16000), (16000, 32000)]
WIDTH, ROLLOFF = 6, 0.99


def prototype(k, o, n):
    base = min(o, n) * ROLLOFF
    t = np.clip(k.astype(np.float64) / (o * n) * base, -float(WIDTH), float(WIDTH))
    sinc = np.where(t == 0, 1.0, np.sin(np.pi * t) / np.where(t == 0, 1.0, np.pi * t))
    return sinc * np.cos(t * np.pi / WIDTH / 2) ** 2 * (base / o)


def main():
    out = {}
    for orig, new in PAIRS:
        g = math.gcd(orig, new)
        o, n = orig // g, new // g
        base = min(o, n) * ROLLOFF
        width = int(math.ceil(WIDTH * o / base))
        i = np.arange(n)[:, None]
        j = np.arange(2 * width + o)[None]
        out[f"bank_{orig}_{new}"] = prototype((j - width) * n - i * o, o, n).astype(np.float32)
        out[f"width_{orig}_{new}"] = np.int64(width)
        K = int(math.ceil(WIDTH * o * n / base))
        h = prototype(np.arange(-K, K + 1), o, n)
        rng = np.random.default_rng(orig + new)
        L = 3001
        x = rng.standard_normal((2, L))
        x[1] = np.sin(2 * np.pi * 440.0 * np.arange(L) / orig) + 0.3 * x[1]
        want_len = int(math.ceil(n * L / o))
        fine = upfirdn(h, x, up=n, down=1, axis=1)
        out[f"x_{orig}_{new}"] = x.astype(np.float32)
        out[f"y_{orig}_{new}"] = upfirdn(h, x.astype(np.float32).astype(np.float64), up=n, down=1, axis=1)[:, K + o * np.arange(want_len)]
        del fine
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g13_resample.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
