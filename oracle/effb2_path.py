"""TEST INFRASTRUCTURE ONLY - CPU restatement of the EffB2-Transformer captioner's encoder (SURVEY.md section 8, rows
A8 / A17: ``EfficientNetB2.forward`` hf_wrapper.py:287-315 == cnn_encoder.py:811-839, ``_EffiNet.forward``
hf_wrapper.py:229-232, ``get_effb2_model`` :235-241).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

**Pinned by independent witnesses, not by the reference** (``tests/golden/make_witness.py`` ->
``tests/golden/g10_logmel.npz`` / ``g11_effb2.npz``, re-checked by ``tests/test_witness.py``): ``extract_features`` agrees
with ``transformers.EfficientNetModel`` (B2 coefficients, ``depthwise_padding=[5, 8, 16]``, one input channel) carrying the
same procedural weights mapped key by key (506 tensors) to 5e-7 of the output's maximum on 64 x 1001, 64 x 3001 and
260 x 260 inputs; ``logmel_effb2`` agrees with ``transformers.audio_utils`` (htk scale, 0-8000 Hz, n_fft 512, hop 160,
clamp at batch maximum - 120 dB) to 1.5e-3 dB (6e-5 dB 99th percentile).  Why a witness is needed:
the backbone's arithmetic lives in the un-vendored third-party package
``efficientnet_pytorch==0.7.1`` (requirements.txt:8: ``EfficientNet.extract_features``, ``MBConvBlock.forward``,
``Conv2dStaticSamePadding``, ``utils.get_model_params / round_filters / round_repeats``) and the mel front-end in
``torchaudio==0.13.1`` (``MelSpectrogram`` defaults: HTK mel scale, no filter normalisation; ``AmplitudeToDB(top_db=120)``);
neither is installed here and the reference has no test or golden vector at either boundary.  This file restates their
PUBLISHED algorithms, anchored on what the reference itself pins:

* the layer list, channel arithmetic, squeeze-excite widths and the static "same" padding chain are spelled out by the
  reference's own re-implementation of the constructor (eff_latent_encoder.py:74-186 with prune_ratio 0) and by its
  state-dict key list (eff_latent_encoder.py:263-290: blocks 0-1 without ``_expand_conv``, blocks 2-22 with);
* the call sites: 1 input channel (hf_wrapper.py:240), ``b f t -> b 1 f t`` (mel is H, time is W), mean over mel
  after ``extract_features`` (hf_wrapper.py:229-232), ``attn_emb_len = (L // 160 + 1) // 32``, ``fc_emb =
  mean_with_lens`` (hf_wrapper.py:303-308).

MBConv (efficientnet_pytorch model.py, EfficientNet paper sec. 4 / MobileNetV3 SE): [1x1 expand + BN + swish when
expand_ratio != 1] -> depthwise k x k (stride s, static same padding) + BN + swish -> SE: global mean -> 1x1 (bias) ->
swish -> 1x1 (bias) -> sigmoid gate -> 1x1 project + BN -> + input when stride 1 and in == out channels.  BatchNorm
eps 1e-3, eval mode.  swish(x) = x * sigmoid(x).
"""
import math

import torch
import torch.nn.functional as F

from . import cpu_path as O

# (repeats, kernel, stride, expand, in, out) of EfficientNet-B0; B2 = width 1.1, depth 1.2, resolution 260
_B0 = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
       (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
WIDTH, DEPTH, IMAGE, SE_RATIO, BN_EPS = 1.1, 1.2, 260, 0.25, 1e-3


def round_filters(filters, width=WIDTH, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats, depth=DEPTH):
    return int(math.ceil(depth * repeats))


def same_pad(image, k, s):
    """Conv2dStaticSamePadding: (before, after) zeros for an axis of the CONSTRUCTION-time size ``image``."""
    out = math.ceil(image / s)
    pad = max((out - 1) * s + (k - 1) + 1 - image, 0)
    return pad // 2, pad - pad // 2


def block_list():
    """The 23 MBConv blocks of EfficientNet-B2: dicts with cin, cout, expand, k, stride, se, pad (before, after)."""
    image = math.ceil(IMAGE / 2)  # after the stride-2 stem
    blocks = []
    for (r, k, s, e, i, o) in _B0:
        cin, cout = round_filters(i), round_filters(o)
        for j in range(round_repeats(r)):
            stride = s if j == 0 else 1
            bi = cin if j == 0 else cout
            blocks.append({"cin": bi, "cout": cout, "expand": e, "k": k, "stride": stride,
                           "se": max(1, int(bi * SE_RATIO)), "pad": same_pad(image, k, stride),
                           "skip": stride == 1 and bi == cout})
            image = math.ceil(image / stride)
    return blocks


STEM_OUT, HEAD_OUT = round_filters(32), round_filters(1280)
STEM_PAD = same_pad(IMAGE, 3, 2)


def swish(x):
    return x * torch.sigmoid(x)


def _bn(x, state, prefix):
    return F.batch_norm(x, state[prefix + ".running_mean"], state[prefix + ".running_var"], state[prefix + ".weight"],
                        state[prefix + ".bias"], False, 0.0, BN_EPS)


def _conv_same(x, w, pad, stride=1, groups=1, bias=None):
    x = F.pad(x, (pad[0], pad[1], pad[0], pad[1]))
    return F.conv2d(x, w, bias, stride=stride, groups=groups)


def extract_features(state, x, prefix="encoder.backbone.eff_net.", return_blocks=False):
    """x (B, 1, F, T) -> (B, 1408, F', T')."""
    x = swish(_bn(_conv_same(x, state[prefix + "_conv_stem.weight"], STEM_PAD, 2), state, prefix + "_bn0"))
    outs = []
    for i, b in enumerate(block_list()):
        p = f"{prefix}_blocks.{i}."
        inp = x
        if b["expand"] != 1:
            x = swish(_bn(F.conv2d(x, state[p + "_expand_conv.weight"]), state, p + "_bn0"))
        x = swish(_bn(_conv_same(x, state[p + "_depthwise_conv.weight"], b["pad"], b["stride"], groups=x.shape[1]),
                      state, p + "_bn1"))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = swish(F.conv2d(sq, state[p + "_se_reduce.weight"], state[p + "_se_reduce.bias"]))
        sq = F.conv2d(sq, state[p + "_se_expand.weight"], state[p + "_se_expand.bias"])
        x = torch.sigmoid(sq) * x
        x = _bn(F.conv2d(x, state[p + "_project_conv.weight"]), state, p + "_bn2")
        if b["skip"]:
            x = x + inp
        outs.append(x)
    x = swish(_bn(F.conv2d(x, state[prefix + "_conv_head.weight"]), state, prefix + "_bn1"))
    return (x, outs) if return_blocks else x


# ---------------------------------------------------------------------------------------------------------
# mel front-end of EfficientNetB2 (hf_wrapper.py:270-279): torchaudio defaults = HTK scale, norm None, power 2
# ---------------------------------------------------------------------------------------------------------
def mel_filterbank_htk(sample_rate=16000, n_fft=512, n_mels=64, f_min=0.0, f_max=None):
    f_max = float(sample_rate // 2) if f_max is None else f_max
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def logmel_effb2(wav, sample_rate=16000, top_db=120.0):
    """wav (B, L) -> (B, 64, T) dB.  AmplitudeToDB(top_db=120) on a 3-D (batch, mel, time) input clamps at the maximum
    of the WHOLE batch minus top_db (torchaudio.functional.amplitude_to_DB packs the leading axis as channels)."""
    n_fft, hop = 32 * sample_rate // 1000, 10 * sample_rate // 1000
    window = torch.hann_window(n_fft, periodic=True)
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    power = spec.abs().pow(2.0)
    mel = torch.matmul(power.transpose(-1, -2), mel_filterbank_htk(sample_rate, n_fft)).transpose(-1, -2)
    db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    return torch.max(db, db.max() - top_db)


def effb2_from_logmel(state, lms, prefix="encoder.backbone.eff_net."):
    """lms (B, 64, T) -> attn_emb (B, T', 1408): ``reduce(extract_features(b 1 f t), 'b c f t -> b t c', 'mean')``."""
    x = extract_features(state, lms.unsqueeze(1), prefix)
    return x.mean(dim=2).transpose(1, 2)


def effb2_feat_len(wav_len, hop=160, ratio=32):
    return O.cnn14_feat_len(wav_len, hop, ratio)


def encoder_forward(state, wav, wav_len, prefix="encoder."):
    attn = effb2_from_logmel(state, logmel_effb2(wav), prefix + "backbone.eff_net.")
    lens = effb2_feat_len(wav_len)
    mask = (torch.arange(attn.shape[1])[None, :] < lens[:, None]).unsqueeze(-1)
    fc = (attn * mask).sum(1) / lens.unsqueeze(1)
    return {"attn_emb": attn, "fc_emb": fc, "attn_emb_len": lens}


def caption_forward(state, wav, wav_len, sample_method="beam", beam_size=3, max_length=20, temp=1.0):
    """``Effb2TrmCaptioningModel.forward`` (hf_wrapper.py:1162-1181): encoder, then the shared decoding routines."""
    enc = encoder_forward(state, wav, wav_len)
    if sample_method == "beam":
        out = O.beam_search(state, enc["attn_emb"], enc["attn_emb_len"], beam_size, max_length, temp)
    else:
        out = O.greedy_decode(state, enc["attn_emb"], enc["attn_emb_len"], max_length)
    out.update(enc)
    return out
