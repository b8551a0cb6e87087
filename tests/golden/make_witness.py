"""Fixtures for the two THIRD-PARTY pieces of the path, produced by independent code - not by anything in this repo.

SURVEY.md section 8(c): the log-mel front-end (``torchaudio==0.13.1`` MelSpectrogram / AmplitudeToDB, call sites
cnn_encoder.py:338-350,418-419 and hf_wrapper.py:270-279,292-293) and the EfficientNet-B2 backbone
(``efficientnet_pytorch==0.7.1``, call sites hf_wrapper.py:225-241, eff_latent_encoder.py:42-57,74-186) are not
vendored by the reference and not installed here, so the imported reference cannot pin them.  The container does hold
an unrelated implementation of both algorithms: ``transformers.audio_utils`` (numpy STFT, mel filter banks with the
slaney / htk scales and slaney normalisation, dB conversion with a range clamp) and ``transformers.EfficientNetModel``
(the Keras-derived EfficientNet).  This script runs THOSE on seeded inputs and writes

    g10_logmel.npz   log-mel of both front-ends (Cnn14: 32 kHz, slaney/slaney, 50-14000 Hz, n_fft 1024, hop 320, no
                     top_db; EffB2: 16 kHz, htk, 0-8000 Hz, n_fft 512, hop 160, top_db 120 over the whole batch)
    g11_effb2.npz    EfficientNet-B2 features (B2 = width 1.1, depth 1.2, 260-px static padding chain, 1 input
                     channel) of the procedural weights, mean over mel as the reference reduces them, for 10 s / 30 s
                     log-mels and a square 260 x 260 input

and asserts on the way that the oracles (oracle/cpu_path.py, oracle/effb2_path.py) agree with the witnesses.  The GPU
tests then compare the HIP kernels against these files: vectors no file of this repository computed.

Run in the build container (needs ``transformers``):   python tests/golden/make_witness.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from audiocaption_amd import procedural as P  # noqa: E402  (inputs + weights only: pure numpy generators)

EFF_PREFIX = "encoder.backbone.eff_net."


# ---------------------------------------------------------------------------------------------------------
# witness 1: transformers.audio_utils
# ---------------------------------------------------------------------------------------------------------
def witness_logmel_cnn14(wav):
    """(B, L) float32 @ 32 kHz -> (B, 64, T) dB, torchaudio MelSpectrogram(sr 32000, n_fft = win 1024, hop 320, f_min 50,
    f_max 14000, 64 mels, norm slaney, mel_scale slaney, power 2) + AmplitudeToDB(top_db None)."""
    from transformers import audio_utils as AU
    fb = AU.mel_filter_bank(513, 64, 50.0, 14000.0, 32000, norm="slaney", mel_scale="slaney")
    win = AU.window_function(1024, "hann", periodic=True)
    out = [AU.spectrogram(np.asarray(w, dtype=np.float64), win, 1024, 320, fft_length=1024, power=2.0, center=True,
                          pad_mode="reflect", onesided=True, mel_filters=fb, mel_floor=1e-10, log_mel="dB",
                          reference=1.0, min_value=1e-10, db_range=None, dtype=np.float64) for w in wav]
    return np.stack(out), fb


def witness_logmel_effb2(wav):
    """(B, L) float32 @ 16 kHz -> (B, 64, T) dB, torchaudio defaults (htk scale, no norm) at n_fft = win 512, hop 160,
    0-8000 Hz, then AmplitudeToDB(top_db 120) whose clamp is the maximum of the WHOLE batch minus 120 dB."""
    from transformers import audio_utils as AU
    fb = AU.mel_filter_bank(257, 64, 0.0, 8000.0, 16000, norm=None, mel_scale="htk")
    win = AU.window_function(512, "hann", periodic=True)
    db = np.stack([AU.spectrogram(np.asarray(w, dtype=np.float64), win, 512, 160, fft_length=512, power=2.0,
                                  center=True, pad_mode="reflect", onesided=True, mel_filters=fb, mel_floor=1e-10,
                                  log_mel="dB", reference=1.0, min_value=1e-10, db_range=None, dtype=np.float64)
                   for w in wav])
    return np.maximum(db, db.max() - 120.0), fb


# ---------------------------------------------------------------------------------------------------------
# witness 2: transformers.EfficientNetModel with the procedural weights mapped by NAME
# ---------------------------------------------------------------------------------------------------------
_BLOCK_MAP = {"_expand_conv": "expansion.expand_conv", "_bn0": "expansion.expand_bn",
              "_depthwise_conv": "depthwise_conv.depthwise_conv", "_bn1": "depthwise_conv.depthwise_norm",
              "_se_reduce": "squeeze_excite.reduce", "_se_expand": "squeeze_excite.expand",
              "_project_conv": "projection.project_conv", "_bn2": "projection.project_bn"}


def hf_key(k):
    """efficientnet_pytorch state-dict key (eff_latent_encoder.py:263-290) -> transformers.EfficientNetModel key."""
    assert k.startswith(EFF_PREFIX)
    parts = k[len(EFF_PREFIX):].split(".")
    if parts[0] == "_blocks":
        return "encoder.blocks." + parts[1] + "." + _BLOCK_MAP[parts[2]] + "." + ".".join(parts[3:])
    top = {"_conv_stem": "embeddings.convolution", "_bn0": "embeddings.batchnorm", "_conv_head": "encoder.top_conv",
           "_bn1": "encoder.top_bn"}[parts[0]]
    return top + "." + ".".join(parts[1:])


def witness_effnet(state):
    from transformers import EfficientNetConfig, EfficientNetModel
    cfg = EfficientNetConfig(num_channels=1, image_size=260, width_coefficient=1.1, depth_coefficient=1.2,
                             depthwise_padding=[5, 8, 16], hidden_dim=1408, batch_norm_eps=1e-3)
    model = EfficientNetModel(cfg).eval()
    mapped = {hf_key(k): v.reshape(()) if k.endswith("num_batches_tracked") else v
              for k, v in state.items() if k.startswith(EFF_PREFIX)}
    want = model.state_dict()
    assert set(mapped) == set(want), (sorted(set(mapped) ^ set(want))[:8])
    for k, v in mapped.items():
        assert tuple(v.shape) == tuple(want[k].shape), k
    model.load_state_dict(mapped, strict=True)
    return model


def effnet_features(model, x):
    """x (B, 1, F, T) -> (B, 1408, F', T'): the backbone without its pooling head (``extract_features``)."""
    with torch.no_grad():
        return model(pixel_values=x).last_hidden_state


def effb2_inputs():
    """Seeded backbone inputs (dB-like log-mels): (2, 64, 1001), (1, 64, 3001) and a square (1, 260, 260)."""
    return {"lms10": P.synthetic_logmel(2, 1001, 64, seed=4321), "lms30": P.synthetic_logmel(1, 3001, 64, seed=4322),
            "sq260": P.synthetic_logmel(1, 260, 260, seed=4323)}


def logmel_inputs():
    wav32 = P.synthetic_wav(2, 64000, seed=777, varied=True)                        # 2 s @ 32 kHz
    wav32 = np.concatenate([wav32, P.synthetic_wav(1, 64000, seed=778)], 0)         # + plain noise
    wav16 = P.synthetic_wav(3, 48000, seed=779, varied=True, sample_rate=16000)     # 3 s @ 16 kHz
    wav16[2] *= np.float32(1e-4)        # nearly silent: its floor is set by the batch maximum
    return wav32, wav16


def main():
    from oracle import cpu_path as O
    from oracle import effb2_path as E
    report = []

    wav32, wav16 = logmel_inputs()
    w_c, fb_c = witness_logmel_cnn14(wav32)
    w_e, fb_e = witness_logmel_effb2(wav16)
    o_c = O.logmel(torch.from_numpy(wav32), 32000).double().numpy()
    o_e = E.logmel_effb2(torch.from_numpy(wav16)).double().numpy()
    d = [float(np.abs(O.mel_filterbank().double().numpy() - fb_c).max()),
         float(np.abs(E.mel_filterbank_htk().double().numpy() - fb_e).max()),
         float(np.abs(o_c - w_c).max()), float(np.abs(o_e - w_e).max())]
    report.append(f"mel filterbank slaney/slaney oracle vs transformers.audio_utils  max|diff| {d[0]:.3e}")
    report.append(f"mel filterbank htk/None     oracle vs transformers.audio_utils  max|diff| {d[1]:.3e}")
    report.append(f"log-mel Cnn14 front-end     oracle vs transformers.audio_utils  max|diff| {d[2]:.3e} dB")
    report.append(f"log-mel EffB2 front-end     oracle vs transformers.audio_utils  max|diff| {d[3]:.3e} dB")
    # the witness is float64; the oracles follow torchaudio's float32 arithmetic (f32 linspace of the htk points, f32 STFT):
    # the largest dB differences sit in bins 85 dB below the clip's peak, where a 512-point f32 transform has ~3 digits left
    q = [float(np.percentile(np.abs(o_c - w_c), 99)), float(np.percentile(np.abs(o_e - w_e), 99))]
    report.append(f"  99th percentile of |diff|: Cnn14 {q[0]:.3e} dB, EffB2 {q[1]:.3e} dB")
    assert d[0] < 1e-6 and d[1] < 1e-5 and d[2] < 5e-3 and d[3] < 5e-3 and max(q) < 2e-4, report
    np.savez_compressed(os.path.join(HERE, "g10_logmel.npz"), cnn14_db=w_c.astype(np.float32),
                        effb2_db=w_e.astype(np.float32), fb_slaney=fb_c.astype(np.float32),
                        fb_htk=fb_e.astype(np.float32))

    state = P.to_torch(P.effb2_state(EFF_PREFIX))
    model = witness_effnet(state)
    out = {}
    for name, x in effb2_inputs().items():
        xt = torch.from_numpy(x).unsqueeze(1)
        want = effnet_features(model, xt)
        got = E.extract_features(state, xt)
        assert want.shape == got.shape, (want.shape, got.shape)
        rel = float((want - got).abs().max() / want.abs().max())
        report.append(f"EfficientNet-B2 features {name} {tuple(xt.shape)} -> {tuple(want.shape)}  oracle vs "
                      f"transformers.EfficientNetModel  max|diff|/max|want| {rel:.3e}")
        assert rel < 5e-6, report
        out[name] = want.mean(dim=2).transpose(1, 2).contiguous().numpy()       # 'b c f t -> b t c' mean
    np.savez_compressed(os.path.join(HERE, "g11_effb2.npz"), **out)
    text = "\n".join(report)
    print(text)
    with open(os.path.join(HERE, "REPORT_witness.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
