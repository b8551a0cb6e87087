"""CPU tests of the ingest oracle.  The resampling filter is PARITY UNPINNED
(torchaudio is not vendored): closed-form properties of the restated windowed-sinc algorithm, plus the reference's own
padding collate."""
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
