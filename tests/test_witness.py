"""CPU: the two third-party pieces of the path pinned by INDEPENDENT code.

torchaudio's MelSpectrogram / AmplitudeToDB (cnn_encoder.py:338-350, hf_wrapper.py:270-279) and efficientnet_pytorch's
EfficientNet-B2 (hf_wrapper.py:225-241, eff_latent_encoder.py:74-186) are not vendored by the reference.
``tests/golden/g10_logmel.npz`` / ``g11_effb2.npz`` were written by ``tests/golden/make_witness.py`` from
``transformers.audio_utils`` and ``transformers.EfficientNetModel`` - implementations this repository did not write.
The first group of tests holds the oracles to those files on any machine; the second group re-runs the witnesses live
where ``transformers`` is installed (the build container) and checks that the committed files still are what they
produce."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import cpu_path as O
from oracle import effb2_path as E

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_witness as W  # noqa: E402  (input generators + the witness wrappers; imports transformers lazily)


@pytest.fixture(scope="module")
def g10(golden_dir):
    return np.load(os.path.join(golden_dir, "g10_logmel.npz"))


@pytest.fixture(scope="module")
def g11(golden_dir):
    return np.load(os.path.join(golden_dir, "g11_effb2.npz"))


def _db_close(got, want, name):
    """The witness is float64; torchaudio (and the oracle) work in float32, so bins 80 dB below a clip's peak keep ~3
    digits: bar 5e-3 dB on the maximum, 2e-4 dB on the 99th percentile."""
    d = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))
    print(f"[{name}] max|diff| {d.max():.3e} dB, 99th percentile {np.percentile(d, 99):.3e} dB")
    assert d.max() < 5e-3 and np.percentile(d, 99) < 2e-4


def test_oracle_filterbanks_vs_witness_fixture(g10):
    assert float(np.abs(O.mel_filterbank().numpy() - g10["fb_slaney"]).max()) < 1e-6
    assert float(np.abs(E.mel_filterbank_htk().numpy() - g10["fb_htk"]).max()) < 1e-5


def test_oracle_logmel_vs_witness_fixture(g10):
    wav32, wav16 = W.logmel_inputs()
    _db_close(O.logmel(torch.from_numpy(wav32), 32000).numpy(), g10["cnn14_db"], "Cnn14 log-mel")
    got = E.logmel_effb2(torch.from_numpy(wav16)).numpy()
    _db_close(got, g10["effb2_db"], "EffB2 log-mel (top_db 120 over the batch)")
    # the silent clip sits on the floor the loud clips set
    assert float(g10["effb2_db"][2].min()) == pytest.approx(float(g10["effb2_db"].max()) - 120.0, abs=1e-4)
    assert float(got.min()) == pytest.approx(float(g10["effb2_db"].min()), abs=1e-3)


def test_product_side_tables_vs_witness_fixture(g10):
    from audiocaption_amd.mel import melscale_fbanks
    assert float(np.abs(melscale_fbanks(513, 50.0, 14000.0, 64, 32000, "slaney", "slaney").numpy()
                        - g10["fb_slaney"]).max()) < 1e-6
    assert float(np.abs(melscale_fbanks(257, 0.0, 8000.0, 64, 16000, None, "htk").numpy() - g10["fb_htk"]).max()) < 1e-5


@pytest.mark.parametrize("name", ["lms10", "lms30", "sq260"])
def test_oracle_effb2_vs_witness_fixture(g11, name):
    state = __import__("audiocaption_amd.procedural", fromlist=["x"])
    st = state.to_torch(state.effb2_state(W.EFF_PREFIX))
    x = torch.from_numpy(W.effb2_inputs()[name])
    got = E.effb2_from_logmel(st, x, W.EFF_PREFIX).numpy()
    want = g11[name]
    assert got.shape == want.shape
    rel = float(np.abs(got - want).max() / np.abs(want).max())
    print(f"[EfficientNet-B2 {name} {tuple(x.shape)} -> {want.shape}] max|diff| / max|want| {rel:.3e}")
    assert rel < 5e-6


# ---- live: only where the witnesses can run -----------------------------------------------------------------------
def test_witness_fixtures_are_what_transformers_produces(g10, g11):
    pytest.importorskip("transformers")
    wav32, wav16 = W.logmel_inputs()
    w_c, fb_c = W.witness_logmel_cnn14(wav32)
    w_e, fb_e = W.witness_logmel_effb2(wav16)
    assert np.array_equal(w_c.astype(np.float32), g10["cnn14_db"]) and np.array_equal(w_e.astype(np.float32), g10["effb2_db"])
    assert np.array_equal(fb_c.astype(np.float32), g10["fb_slaney"]) and np.array_equal(fb_e.astype(np.float32), g10["fb_htk"])
    from audiocaption_amd import procedural as P
    model = W.witness_effnet(P.to_torch(P.effb2_state(W.EFF_PREFIX)))
    x = torch.from_numpy(W.effb2_inputs()["lms10"]).unsqueeze(1)
    want = W.effnet_features(model, x).mean(dim=2).transpose(1, 2).numpy()
    assert float(np.abs(want - g11["lms10"]).max()) < 1e-5 * float(np.abs(want).max())   # conv threading: not bit-stable


def test_every_backbone_tensor_is_mapped_by_name():
    """506 tensors of the efficientnet_pytorch layout (eff_latent_encoder.py:263-290) <-> 506 of the witness model."""
    pytest.importorskip("transformers")
    from audiocaption_amd import procedural as P
    st = P.effb2_state(W.EFF_PREFIX)
    keys = [W.hf_key(k) for k in st]
    assert len(keys) == len(set(keys)) == 506
