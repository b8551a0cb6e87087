"""CPU tests of the ingest oracle: the polyphase machinery of the resampler against
scipy.signal.upfirdn, closed-form properties of the restated windowed-sinc prototype (its formula is the one part no
installed package can witness: torchaudio is not vendored), and the reference's own padding collate."""
import math

import numpy as np
import pytest
import torch


def test_resample_kernel_geometry_and_dc_gain():
    from oracle import ingest_path as I
    k, width, orig, new = I.sinc_resample_kernel(44100, 32000)
    assert (orig, new, width) == (441, 320, 9) and k.shape == (320, 459)
    # every polyphase branch of a low-pass interpolator has (close to) unit DC gain
    np.testing.assert_allclose(k.sum(1).numpy(), 1.0, atol=2e-3)
    # the clamped Hann window leaves ~2 * 6 * orig / (0.99 * new) non-zero taps per phase
    nz = (k != 0).sum(1)
    assert int(nz.max()) <= 18 and int(nz.min()) >= 15
    k2, w2, o2, n2 = I.sinc_resample_kernel(16000, 32000)
    assert (o2, n2) == (1, 2) and w2 == math.ceil(6 / 0.99)


@pytest.mark.parametrize("orig,new", [(44100, 32000), (48000, 16000), (16000, 32000)])
def test_resample_length_and_sinusoid(orig, new):
    from oracle import ingest_path as I
    n = 30011
    t = torch.arange(n) / orig
    x = torch.sin(2 * math.pi * 1000.0 * t)                         # 1 kHz, far below both Nyquist rates
    y = I.resample(x[None], orig, new)[0]
    g = math.gcd(orig, new)
    assert y.shape[0] == math.ceil((new // g) * n / (orig // g))
    want = torch.sin(2 * math.pi * 1000.0 * torch.arange(y.shape[0]) / new)
    inner = slice(200, y.shape[0] - 200)                            # away from the zero-padded edges
    assert float((y[inner] - want[inner]).abs().max()) < 5e-3
    dc = I.resample(torch.ones(1, 20000), orig, new)[0]
    assert float((dc[200:-200] - 1.0).abs().max()) < 2e-3


def test_collate_matches_wav_pad_collate_semantics():
    from oracle import ingest_path as I
    rng = np.random.default_rng(0)
    items = [("a", rng.standard_normal(12000).astype(np.float32)), ("short", rng.standard_normal(100).astype(np.float32)),
             ("none", None), ("b", rng.standard_normal(20000).astype(np.float32))]
    out = I.wav_pad_collate(items, min_duration=0.32, sample_rate=32000)
    assert out["aid"].tolist() == ["a", "b"] and out["blacklist_aid"] == ["short", "none"]
    assert out["wav"].shape == (2, 20000) and out["wav_len"].tolist() == [12000, 20000]
    assert np.all(out["wav"][0, 12000:] == 0) and np.array_equal(out["wav"][1], items[3][1].astype(np.float64))


@pytest.mark.parametrize("orig,new", [(44100, 32000), (48000, 32000), (16000, 32000), (22050, 32000), (32000, 16000)])
def test_resample_polyphase_machinery_vs_scipy_upfirdn(orig, new):
    """Independent witness for everything in the resampler EXCEPT the prototype filter's formula: scipy.signal.upfirdn
    (zero-stuff by `new`, one long FIR at the rate orig * new, keep every `orig`-th sample) driven by the windowed-sinc
    prototype h(k) = scale * sinc(f k) * cos^2(pi f k / 12), f = 0.99 min(orig, new) / (orig new), evaluated in float64 on the
    fine grid - against the oracle's bank of `new` polyphase kernels, padding, block reshape and output length
    (torchaudio 0.13.1 _get_sinc_resample_kernel / _apply_sinc_resample_kernel as restated in oracle/ingest_path.py).
    Phase order, tap alignment and length conventions are where a polyphase restatement goes wrong; the formula itself
    stays 'restated from the published algorithm'."""
    from scipy.signal import upfirdn
    from oracle import ingest_path as I
    g = math.gcd(orig, new)
    o, n = orig // g, new // g
    base = min(o, n) * 0.99
    K = int(math.ceil(6 * o * n / base))                      # support of the prototype on the fine grid
    k = np.arange(-K, K + 1, dtype=np.float64)
    t = np.clip(k / (o * n) * base, -6.0, 6.0)
    h = np.where(t == 0, 1.0, np.sin(np.pi * t) / np.where(t == 0, 1.0, np.pi * t)) * np.cos(t * np.pi / 12) ** 2 * (base / o)
    rng = np.random.default_rng(orig + new)
    L = 3001
    x = rng.standard_normal((2, L))
    x[1] = np.sin(2 * np.pi * 440.0 * np.arange(L) / orig) + 0.3 * x[1]
    want_len = int(math.ceil(n * L / o))
    fine = upfirdn(h, x, up=n, down=1, axis=1)                 # fine[K + j] = sum_m x[m] h(j - m n)
    want = fine[:, K + o * np.arange(want_len)]
    got = I.resample(torch.from_numpy(x).float(), orig, new).numpy()
    assert got.shape == (2, want_len)
    err = float(np.abs(got - want).max())
    print(f"resample {orig}->{new}: oracle vs scipy.upfirdn max|diff| {err:.2e} (signal max {float(np.abs(want).max()):.2f})")
    assert err < 2e-5                                          # float32 kernel + float32 convolution vs float64


@pytest.mark.parametrize("orig,new", [(44100, 32000), (48000, 16000), (16000, 32000)])
def test_filter_bank_and_resampler_vs_committed_fixture(orig, new):
    """tests/golden/g13_resample.npz (tests/golden/make_resample_golden.py: numpy / scipy only, the published windowed-sinc
    prototype evaluated in float64 on the fine grid - neither the product's table builder nor the oracle is imported there):
    the PRODUCT's filter bank (audiocaption_amd.ingest._sinc_kernel, what csrc/ingest.hip walks) and the oracle's equal the
    fixture's to float32 rounding, and the oracle's resampled signals equal scipy.signal.upfirdn's."""
    import os
    from audiocaption_amd.ingest import _sinc_kernel
    from oracle import ingest_path as I
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_resample.npz"))
    bank, width = g[f"bank_{orig}_{new}"], int(g[f"width_{orig}_{new}"])
    k, w, o, n = _sinc_kernel(orig, new)
    assert (w, k.shape) == (width, bank.shape) and float(np.abs(k.numpy() - bank).max()) <= 1e-7
    ko, wo, _, _ = I.sinc_resample_kernel(orig, new)
    assert wo == width and float(np.abs(ko.numpy() - bank).max()) <= 1e-7
    y = I.resample(torch.from_numpy(g[f"x_{orig}_{new}"]), orig, new).numpy()
    assert y.shape == g[f"y_{orig}_{new}"].shape and float(np.abs(y - g[f"y_{orig}_{new}"]).max()) < 2e-6


def test_crop_or_pad_to_audio_duration_follows_the_dataset():
    """caption_dataset.py:121-129: a longer clip loses all but a random window (random.randint, inclusive bounds, one draw
    per longer clip in order), a shorter one is zero-padded, an exact one passes; every kept clip then has
    int(audio_duration * target_sr) samples for the collate."""
    import random
    from oracle import ingest_path as I
    rng = np.random.default_rng(3)
    clips = [("long", rng.standard_normal(52000).astype(np.float32)), ("short", rng.standard_normal(9000).astype(np.float32)),
             ("none", None), ("exact", rng.standard_normal(32000).astype(np.float32)),
             ("long2", rng.standard_normal(40001).astype(np.float32))]
    out = I.ingest(clips, 32000, 32000, audio_duration=1.0, rng=random.Random(7))
    assert out["aid"].tolist() == ["long", "short", "exact", "long2"] and out["blacklist_aid"] == ["none"]
    assert out["wav"].shape == (4, 32000) and out["wav_len"].tolist() == [32000] * 4
    r = random.Random(7)
    s0 = r.randint(0, 52000 - 32000)
    s1 = r.randint(0, 40001 - 32000)
    assert np.array_equal(out["wav"][0], clips[0][1][s0:s0 + 32000].astype(np.float64))
    assert np.array_equal(out["wav"][1][:9000], clips[1][1].astype(np.float64)) and not out["wav"][1][9000:].any()
    assert np.array_equal(out["wav"][2], clips[3][1].astype(np.float64))
    assert np.array_equal(out["wav"][3], clips[4][1][s1:s1 + 32000].astype(np.float64))
