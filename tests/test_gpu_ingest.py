"""GPU test of the ingest kernel (float16 -> float32, resampling, padding collate in one pass) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("orig,target,half", [(44100, 32000, True), (32000, 32000, True), (48000, 16000, False)])
def test_ingest_vs_oracle(orig, target, half):
    from audiocaption_amd import build
    from audiocaption_amd.ingest import WaveformIngest
    from oracle import ingest_path as I
    build.build()
    rng = np.random.default_rng(1)
    dt = np.float16 if half else np.float32
    items = [("a", (0.1 * rng.standard_normal(orig * 2 + 17)).astype(dt)), ("tiny", (0.1 * rng.standard_normal(300)).astype(dt)),
             ("b", (0.1 * rng.standard_normal(orig + 5)).astype(dt)), ("none", None),
             ("c", (0.1 * rng.standard_normal(int(orig * 1.3))).astype(dt))]
    want = I.ingest(items, orig, target)
    got = WaveformIngest(orig, target)(items)
    assert got["aid"].tolist() == want["aid"].tolist() and got["blacklist_aid"] == want["blacklist_aid"]
    assert got["wav_len"].tolist() == want["wav_len"].tolist()
    assert got["wav"].is_cuda and got["wav"].dtype == torch.float32 and tuple(got["wav"].shape) == want["wav"].shape
    d = float((got["wav"].cpu().double() - torch.from_numpy(want["wav"])).abs().max())
    print(f"ingest {orig}->{target} max|diff| {d:.3e}")
    assert d < 2e-6


def test_ingest_feeds_the_model(hip_model):
    """44.1 kHz float16 clips -> ingest -> captions: the step in front of the hot path plugs straight into it."""
    from audiocaption_amd.ingest import WaveformIngest
    rng = np.random.default_rng(2)
    items = [(f"clip{i}", (0.1 * rng.standard_normal(int(44100 * s))).astype(np.float16)) for i, s in enumerate((3.0, 2.2, 0.1))]
    batch = WaveformIngest(44100, 32000)(items)
    assert batch["blacklist_aid"] == ["clip2"] and batch["wav_len"].tolist() == [96000, 70400]
    out = hip_model({"mode": "inference", "wav": batch["wav"], "wav_len": batch["wav_len"], "specaug": False,
                     "sample_method": "greedy", "max_length": 5})
    assert tuple(out["seq"].shape) == (2, 5)


@pytest.mark.parametrize("orig,target", [(44100, 32000), (32000, 32000)])
def test_ingest_crop_and_pad_to_audio_duration_vs_oracle(orig, target):
    """``audio_duration`` (caption_dataset.py:121-129): longer clips cropped at the dataset's random offset, shorter ones
    zero-padded, inside the same kernel pass - against the oracle with the same seeded ``random.Random``."""
    import random
    from audiocaption_amd import build
    from audiocaption_amd.ingest import WaveformIngest
    from oracle import ingest_path as I
    build.build()
    rng = np.random.default_rng(5)
    items = [("long", (0.1 * rng.standard_normal(int(orig * 2.7))).astype(np.float16)), ("none", None),
             ("short", (0.1 * rng.standard_normal(int(orig * 0.6))).astype(np.float16)),
             ("long2", (0.1 * rng.standard_normal(int(orig * 1.5) + 3)).astype(np.float16))]
    want = I.ingest(items, orig, target, audio_duration=1.25, rng=random.Random(11))
    got = WaveformIngest(orig, target, audio_duration=1.25, rng=random.Random(11))(items)
    n = int(1.25 * target)
    assert got["aid"].tolist() == want["aid"].tolist() == ["long", "short", "long2"] and got["blacklist_aid"] == ["none"]
    assert got["wav_len"].tolist() == want["wav_len"].tolist() == [n] * 3 and tuple(got["wav"].shape) == (3, n)
    d = float((got["wav"].cpu().double() - torch.from_numpy(want["wav"])).abs().max())
    print(f"ingest {orig}->{target} with audio_duration: max|diff| {d:.3e}")
    assert d < 2e-6
    assert float(got["wav"][1, int(0.6 * target) + 2:].abs().max()) == 0.0          # the short clip's zero tail


@pytest.mark.parametrize("orig,new", [(44100, 32000), (48000, 16000), (16000, 32000)])
def test_ingest_kernel_vs_committed_resample_fixture(orig, new):
    """The kernel's resampled samples against tests/golden/g13_resample.npz (scipy.signal.upfirdn in float64 driven by the
    published prototype: independent of the product's table builder and of the oracle)."""
    import os
    from audiocaption_amd import build
    from audiocaption_amd.ingest import WaveformIngest
    build.build()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_resample.npz"))
    x, y = g[f"x_{orig}_{new}"], g[f"y_{orig}_{new}"]
    got = WaveformIngest(orig, new, min_duration=0.0)([("a", x[0]), ("b", x[1])])
    assert tuple(got["wav"].shape) == y.shape
    d = float((got["wav"].cpu().double().numpy() - y).__abs__().max())
    print(f"ingest kernel {orig}->{new} vs upfirdn fixture: max|diff| {d:.3e}")
    assert d < 3e-6
