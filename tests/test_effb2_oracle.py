"""CPU tests of the EffB2 oracle (efficientnet_pytorch / torchaudio are not vendored; numeric pinning by independent
witnesses is in tests/test_witness.py): the structure
facts the reference itself pins (eff_latent_encoder.py:74-186,263-290) and closed-form
properties of the restated algorithms."""
import math

import numpy as np
import pytest
import torch


def test_b2_structure_matches_what_the_reference_pins():
    from oracle import effb2_path as E
    blocks = E.block_list()
    assert len(blocks) == 23
    # widths 32 -> 16, 24, 48, 88, 120, 208, 352 -> 1408; repeats 2, 3, 3, 4, 4, 5, 2 (analytic)
    assert E.STEM_OUT == 32 and E.HEAD_OUT == 1408
    couts = [b["cout"] for b in blocks]
    assert couts == [16] * 2 + [24] * 3 + [48] * 3 + [88] * 4 + [120] * 4 + [208] * 5 + [352] * 2
    # blocks 0-1 have no _expand_conv, blocks 2-22 do (eff_latent_encoder.py:267-285)
    assert [b["expand"] for b in blocks] == [1, 1] + [6] * 21
    # static "same" padding computed for the 260-px chain 260 -> 130 -> 130 -> 65 -> 33 -> 17 -> 17 -> 9 -> 9
    assert E.STEM_PAD == (0, 1)
    strided = [(i, b["k"], b["pad"]) for i, b in enumerate(blocks) if b["stride"] == 2]
    assert strided == [(2, 3, (0, 1)), (5, 5, (2, 2)), (8, 3, (1, 1)), (16, 5, (2, 2))]
    assert all(b["pad"] == ((b["k"] - 1) // 2,) * 2 for b in blocks if b["stride"] == 1)
    # squeeze-excite width = max(1, int(block input filters * 0.25)) (eff_latent_encoder.py:104-108)
    assert [b["se"] for b in blocks[:6]] == [8, 4, 4, 6, 6, 6]
    assert [b["skip"] for b in blocks[:4]] == [False, True, False, True]


def test_state_dict_keys_and_shapes_are_efficientnet_pytorchs(state_effb2):
    import audiocaption_amd as A
    model = A.init_model_from_config(A.effb2_trm_config(4981), print_fn=lambda s: None)
    sd = model.state_dict()
    # plus torchaudio's two MelSpectrogram buffers under the reference's attribute name (hf_wrapper.py:270-277)
    mel = {"encoder.melspec_extractor.spectrogram.window": (512,), "encoder.melspec_extractor.mel_scale.fb": (257, 64)}
    assert set(sd) == set(state_effb2) | set(mel)
    for k, shape in mel.items():
        assert tuple(sd[k].shape) == shape
    for k, v in sd.items():
        if not k.endswith("num_batches_tracked") and k not in mel:
            assert tuple(v.shape) == tuple(state_effb2[k].shape), k
    p = "encoder.backbone.eff_net."
    assert sd[p + "_conv_stem.weight"].shape == (32, 1, 3, 3)             # in_channels changed to 1 (hf_wrapper.py:240)
    assert p + "_blocks.0._expand_conv.weight" not in sd and p + "_blocks.2._expand_conv.weight" in sd
    assert sd[p + "_blocks.2._depthwise_conv.weight"].shape == (96, 1, 3, 3)
    assert sd[p + "_blocks.5._depthwise_conv.weight"].shape == (144, 1, 5, 5)
    assert sd[p + "_blocks.2._se_reduce.weight"].shape == (4, 96, 1, 1) and sd[p + "_blocks.2._se_reduce.bias"].shape == (4,)
    assert sd[p + "_conv_head.weight"].shape == (1408, 352, 1, 1)
    assert sd["decoder.classifier.weight"].data_ptr() == sd["decoder.word_embedding.weight"].data_ptr()  # tied
    n = sum(v.numel() for k, v in sd.items() if "eff_net" in k and v.dtype.is_floating_point and "running" not in k)
    assert abs(n - 7.70e6) < 0.05e6                                        # EfficientNet-B2 without its classifier


@pytest.mark.parametrize("seconds,frames,valid", [(10, 32, 31), (30, 94, 93)])
def test_output_geometry(state_effb2, seconds, frames, valid):
    """(B, 1, 64, T) -> (B, 1408, 2, T') -> (B, T', 1408); T' = 32 / 94 and len = 31 / 93."""
    from oracle import effb2_path as E
    T = seconds * 100 + 1
    x = torch.zeros(1, 1, 64, T)
    x[..., ::7] = -20.0
    y = E.extract_features(state_effb2, x)
    assert y.shape == (1, 1408, 2, frames)
    assert int(E.effb2_feat_len([seconds * 16000])[0]) == valid


def test_htk_filterbank_and_top_db():
    from oracle import effb2_path as E
    fb = E.mel_filterbank_htk()
    assert fb.shape == (257, 64) and float(fb.min()) >= 0.0
    # un-normalised triangles: neighbouring filters sum to 1 between the first and the last centre
    m = np.linspace(0.0, 2595.0 * math.log10(1.0 + 8000.0 / 700.0), 66)
    f = 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    freqs = np.linspace(0, 8000, 257)
    inside = (freqs > f[1]) & (freqs < f[-2])
    np.testing.assert_allclose(fb.sum(1).numpy()[inside], 1.0, atol=1e-4)
    # top_db acts on the maximum of the WHOLE batch: a loud clip lifts the floor of a silent one
    wav = torch.zeros(2, 16000)
    wav[0] = torch.sin(2 * math.pi * 1000.0 * torch.arange(16000) / 16000.0)
    db = E.logmel_effb2(wav)
    assert float(db[1].max()) == pytest.approx(float(db.max()) - 120.0) and float(db[1].min()) == float(db[1].max())
    alone = E.logmel_effb2(wav[1:])
    assert float(alone.max()) == pytest.approx(-100.0)                     # clamp(1e-10) with nothing louder around


def test_mbconv_known_answers(state_effb2):
    """Zero depthwise + zero SE weights: the gate is sigmoid(bias) and a skip block returns its input plus the
    projected constant."""
    from oracle import effb2_path as E
    st = dict(state_effb2)
    p = "encoder.backbone.eff_net._blocks.1."
    st[p + "_depthwise_conv.weight"] = torch.zeros_like(st[p + "_depthwise_conv.weight"])
    x = torch.randn(1, 1, 64, 101)
    _, outs = E.extract_features(st, x, return_blocks=True)
    _, ref = E.extract_features(state_effb2, x, return_blocks=True)
    delta = outs[1] - outs[0]
    assert torch.equal(outs[0], ref[0])
    # with a zero depthwise kernel the branch is position independent up to nothing at all: constant per channel
    assert float((delta - delta[:, :, :1, :1]).abs().max()) < 1e-5
