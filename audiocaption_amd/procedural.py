"""Procedural (checkpoint-free) weights and synthetic inputs for the Cnn14Rnn-Trm path.

No checkpoint of the reference can be fetched offline (361 MB), so every
parity test, the smoke test and bench.py use weights generated here.  A tensor's values are
a pure function of (base seed, state-dict key, shape): the same key gives the same numbers
in the reference (loaded through ``load_state_dict`` by tests/golden/make_golden.py), in the
CPU oracle and in the HIP path.

State-dict key layout follows SURVEY.md §2.4 (reference ``state_dict()`` of
``TransformerModel(CrnnEncoder(Cnn14Encoder, RnnEncoder), TransformerDecoder)``).
"""
import math
import zlib

import numpy as np

BASE_SEED = 1234

END_BETA = 1.5

CNN14_CHANNELS = [1, 64, 128, 256, 512, 1024, 2048]


def _rng(key, seed=BASE_SEED):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def _normal(key, shape, std, seed=BASE_SEED):
    return (_rng(key, seed).standard_normal(size=shape, dtype=np.float32) * np.float32(std))


def _uniform(key, shape, lo, hi, seed=BASE_SEED):
    u = _rng(key, seed).random(size=shape, dtype=np.float32)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


def _bn(prefix, c, out, seed):
    out[prefix + ".weight"] = _uniform(prefix + ".weight", (c,), 0.9, 1.1, seed)
    out[prefix + ".bias"] = _uniform(prefix + ".bias", (c,), -0.1, 0.1, seed)
    out[prefix + ".running_mean"] = _uniform(prefix + ".running_mean", (c,), -0.1, 0.1, seed)
    out[prefix + ".running_var"] = _uniform(prefix + ".running_var", (c,), 0.8, 1.2, seed)
    out[prefix + ".num_batches_tracked"] = np.array(0, dtype=np.int64)


def cnn14_state(prefix="", seed=BASE_SEED, with_fc1=True):
    """Cnn14Encoder tensors (reference ctor: cnn_encoder.py:330-368)."""
    out = {}
    p = prefix + "bn0"
    _bn(p, 64, out, seed)
    # log-mel of the synthetic waveform sits around -25 dB with ~5.6 dB spread per bin: give bn0
    # statistics of that order so the conv stack sees O(1) inputs, as a trained bn0 would.
    out[p + ".running_mean"] = _uniform(p + ".running_mean", (64,), -30.0, -20.0, seed)
    out[p + ".running_var"] = _uniform(p + ".running_var", (64,), 25.0, 40.0, seed)
    for b in range(6):
        cin, cout = CNN14_CHANNELS[b], CNN14_CHANNELS[b + 1]
        for j, ci in ((1, cin), (2, cout)):
            k = f"{prefix}conv_block{b + 1}.conv{j}.weight"
            out[k] = _normal(k, (cout, ci, 3, 3), math.sqrt(2.0 / (9 * ci)), seed)
            _bn(f"{prefix}conv_block{b + 1}.bn{j}", cout, out, seed)
    if with_fc1:
        k = prefix + "fc1.weight"
        out[k] = _normal(k, (2048, 2048), math.sqrt(1.0 / 2048), seed)
        out[prefix + "fc1.bias"] = _uniform(prefix + "fc1.bias", (2048,), -0.01, 0.01, seed)
    return out


def gru_state(prefix="", input_size=2048, hidden=256, layers=3, seed=BASE_SEED):
    """nn.GRU(bidirectional) tensors under ``network.`` (reference rnn_encoder.py:23-29)."""
    out = {}
    for l in range(layers):
        in_l = input_size if l == 0 else 2 * hidden
        for suf in ("", "_reverse"):
            # a real Cnn14 output is O(1); keep the pre-activations O(1) too so the gates are
            # not saturated (saturated gates would hide recurrence bugs from the parity tests)
            k = f"{prefix}network.weight_ih_l{l}{suf}"
            out[k] = _normal(k, (3 * hidden, in_l), 1.0 / math.sqrt(in_l), seed)
            k = f"{prefix}network.weight_hh_l{l}{suf}"
            out[k] = _normal(k, (3 * hidden, hidden), 1.0 / math.sqrt(hidden), seed)
            for b in ("bias_ih", "bias_hh"):
                k = f"{prefix}network.{b}_l{l}{suf}"
                out[k] = _uniform(k, (3 * hidden,), -0.1, 0.1, seed)
    return out


def positional_encoding(d_model=256, max_len=100):
    """Sinusoid table, closed form of reference model_util.py:172-179 (shape (max_len,1,d))."""
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    position = np.arange(0, max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2).astype(np.float32)
                      * np.float32(-math.log(10000.0) / d_model)).astype(np.float32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe[:, None, :]


def decoder_state(prefix="", vocab_size=4368, d_model=256, attn_emb_dim=512, nlayers=2,
                  dim_ff=1024, seed=BASE_SEED, tie_weights=False):
    """TransformerDecoder tensors (reference transformer_decoder.py:13-49)."""
    out = {}
    d = d_model
    k = prefix + "word_embedding.weight"
    # std 1/sqrt(d): after the decoder's sqrt(d) scaling the token embedding is O(1), the same
    # order as the positional encoding and the sub-layer outputs, so the greedy sequence depends
    # on position and audio and not only on the previous token (which would cycle).
    out[k] = _normal(k, (vocab_size, d), 1.0 / math.sqrt(d), seed)
    out[prefix + "pos_encoder.pe"] = positional_encoding(d, 100)
    for l in range(nlayers):
        lp = f"{prefix}model.layers.{l}."
        for att in ("self_attn", "multihead_attn"):
            k = lp + att + ".in_proj_weight"
            out[k] = _normal(k, (3 * d, d), 1.0 / math.sqrt(d), seed)
            k = lp + att + ".in_proj_bias"
            out[k] = _uniform(k, (3 * d,), -0.05, 0.05, seed)
            k = lp + att + ".out_proj.weight"
            out[k] = _normal(k, (d, d), 1.0 / math.sqrt(d), seed)
            k = lp + att + ".out_proj.bias"
            out[k] = _uniform(k, (d,), -0.05, 0.05, seed)
        k = lp + "linear1.weight"
        out[k] = _normal(k, (dim_ff, d), math.sqrt(2.0 / d), seed)
        out[lp + "linear1.bias"] = _uniform(lp + "linear1.bias", (dim_ff,), -0.05, 0.05, seed)
        k = lp + "linear2.weight"
        out[k] = _normal(k, (d, dim_ff), 1.0 / math.sqrt(dim_ff), seed)
        out[lp + "linear2.bias"] = _uniform(lp + "linear2.bias", (d,), -0.05, 0.05, seed)
        for n in ("norm1", "norm2", "norm3"):
            out[lp + n + ".weight"] = _uniform(lp + n + ".weight", (d,), 0.9, 1.1, seed)
            out[lp + n + ".bias"] = _uniform(lp + n + ".bias", (d,), -0.1, 0.1, seed)
    if tie_weights:
        out[prefix + "classifier.weight"] = out[prefix + "word_embedding.weight"]
    else:
        k = prefix + "classifier.weight"
        out[k] = _normal(k, (vocab_size, d), 1.0 / math.sqrt(d), seed)
    if not tie_weights:
        # Untrained weights almost never rank <end> first.  Align the <end> row with the last
        # LayerNorm's bias (a constant component of every decoder output): its logit becomes
        # beta*(1 + ~1.1*g), g~N(0,1) per step, against a max of ~4.1 over the other rows, so
        # clips stop at different steps and finished/unfinished rows coexist in a batch.
        b3 = out[f"{prefix}model.layers.{nlayers - 1}.norm3.bias"]
        row = (END_BETA / float(np.dot(b3, b3))) * b3
        out[prefix + "classifier.weight"][2] = row.astype(np.float32)
    k = prefix + "attn_proj.0.weight"
    out[k] = _normal(k, (d, attn_emb_dim), math.sqrt(2.0 / attn_emb_dim), seed)
    out[prefix + "attn_proj.0.bias"] = _uniform(prefix + "attn_proj.0.bias", (d,), -0.05, 0.05, seed)
    out[prefix + "attn_proj.3.weight"] = _uniform(prefix + "attn_proj.3.weight", (d,), 0.9, 1.1, seed)
    out[prefix + "attn_proj.3.bias"] = _uniform(prefix + "attn_proj.3.bias", (d,), -0.1, 0.1, seed)
    return out


# A second, HIGH-ENTROPY decoder draw for the decode fixtures g4b / g5b (tests/golden/make_golden.py): larger token
# embeddings and positional encodings make the next token depend on the whole prefix (few repeated tokens, clips and beams
# that differ), a larger <end> logit makes beams finish early while others continue (the -1000 path of base.py:317), and
# closely ranked candidates make the parent beam change at most steps (prev_beam != identity: the KV-cache re-gather).
DIVERSE = {"greedy": {"seed": 325, "emb_scale": 3.0, "pe_scale": 4.0, "end_beta": 3.0},
           "beam": {"seed": 255, "emb_scale": 2.0, "pe_scale": 2.0, "end_beta": 3.0}}


def decoder_state_diverse(kind, prefix="decoder.", vocab_size=4981):
    """``decoder_state`` re-drawn with the DIVERSE[kind] seed and scales (kind: "greedy" | "beam")."""
    c = DIVERSE[kind]
    d = decoder_state(prefix, vocab_size, seed=c["seed"])
    d[prefix + "word_embedding.weight"] = d[prefix + "word_embedding.weight"] * np.float32(c["emb_scale"])
    d[prefix + "pos_encoder.pe"] = d[prefix + "pos_encoder.pe"] * np.float32(c["pe_scale"])
    b3 = d[f"{prefix}model.layers.1.norm3.bias"]
    cw = d[prefix + "classifier.weight"].copy()
    cw[2] = ((c["end_beta"] / float(np.dot(b3, b3))) * b3).astype(np.float32)
    d[prefix + "classifier.weight"] = cw
    return d


def cnn14rnn_trm_state(vocab_size=4368, seed=BASE_SEED):
    """Full state dict of the Cnn14Rnn-Trm captioner."""
    out = {}
    out.update(cnn14_state("encoder.cnn.", seed))
    out.update(gru_state("encoder.rnn.", 2048, 256, 3, seed))
    out.update(decoder_state("decoder.", vocab_size, 256, 512, 2, 1024, seed))
    return out


# (repeats, kernel, stride, expand, in, out) of EfficientNet-B0 (Tan & Le 2019, table 1); B2 scales width by 1.1 and
# depth by 1.2.  The reference spells the same construction out in eff_latent_encoder.py:74-186.
_EFFNET_B0 = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
              (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


def effb2_round_filters(filters, width=1.1, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def effb2_blocks():
    """[(cin, cout, expand, kernel, stride, se_channels)] for the 23 MBConv blocks of EfficientNet-B2."""
    out = []
    for (r, k, s, e, i, o) in _EFFNET_B0:
        cin, cout = effb2_round_filters(i), effb2_round_filters(o)
        for j in range(int(math.ceil(1.2 * r))):
            bi = cin if j == 0 else cout
            out.append((bi, cout, e, k, s if j == 0 else 1, max(1, int(bi * 0.25))))
    return out


def effb2_state(prefix="encoder.backbone.eff_net.", seed=BASE_SEED):
    """EfficientNet-B2 tensors with 1 input channel and no classifier top (hf_wrapper.py:235-241), keys as
    efficientnet_pytorch names them (eff_latent_encoder.py:263-290)."""
    out = {}

    def conv(key, shape, fan_in, gain=2.0):
        out[key] = _normal(key, shape, math.sqrt(gain / fan_in), seed)

    # the stem sees raw dB values (~ -60..0): a small stem keeps the activations O(1) like a trained one would
    k = prefix + "_conv_stem.weight"
    out[k] = _normal(k, (32, 1, 3, 3), math.sqrt(2.0 / 9) / 20.0, seed)
    _bn(prefix + "_bn0", 32, out, seed)
    for i, (cin, cout, e, ks, s, se) in enumerate(effb2_blocks()):
        p = f"{prefix}_blocks.{i}."
        mid = cin * e
        if e != 1:
            conv(p + "_expand_conv.weight", (mid, cin, 1, 1), cin)
            _bn(p + "_bn0", mid, out, seed)
        conv(p + "_depthwise_conv.weight", (mid, 1, ks, ks), ks * ks)
        _bn(p + "_bn1", mid, out, seed)
        conv(p + "_se_reduce.weight", (se, mid, 1, 1), mid, 1.0)
        out[p + "_se_reduce.bias"] = _uniform(p + "_se_reduce.bias", (se,), -0.1, 0.1, seed)
        conv(p + "_se_expand.weight", (mid, se, 1, 1), se, 1.0)
        out[p + "_se_expand.bias"] = _uniform(p + "_se_expand.bias", (mid,), -0.1, 0.1, seed)
        conv(p + "_project_conv.weight", (cout, mid, 1, 1), mid, 1.0)
        _bn(p + "_bn2", cout, out, seed)
    conv(prefix + "_conv_head.weight", (effb2_round_filters(1280), effb2_blocks()[-1][1], 1, 1), effb2_blocks()[-1][1])
    _bn(prefix + "_bn1", effb2_round_filters(1280), out, seed)
    return out


def effb2_trm_state(vocab_size=4981, seed=BASE_SEED):
    """State dict of the EffB2-Transformer captioner (``Effb2TrmConfig`` defaults, hf_wrapper.py:1115-1141: d_model
    256, 2 layers, tied word embedding / classifier, attn_emb_dim 1408)."""
    out = {}
    out.update(effb2_state("encoder.backbone.eff_net.", seed))
    out.update(decoder_state("decoder.", vocab_size, 256, 1408, 2, 1024, seed, tie_weights=True))
    return out


def trm_encoder_state(prefix="", attn_feat_dim=2048, d_model=256, nlayers=2, dim_ff=1024, seed=BASE_SEED):
    """TransformerEncoder tensors (reference transformer_encoder.py:66-86)."""
    out = {}
    d = d_model
    k = prefix + "attn_proj.0.weight"
    out[k] = _normal(k, (d, attn_feat_dim), math.sqrt(2.0 / attn_feat_dim), seed)
    out[prefix + "attn_proj.0.bias"] = _uniform(prefix + "attn_proj.0.bias", (d,), -0.05, 0.05, seed)
    out[prefix + "attn_proj.3.weight"] = _uniform(prefix + "attn_proj.3.weight", (d,), 0.9, 1.1, seed)
    out[prefix + "attn_proj.3.bias"] = _uniform(prefix + "attn_proj.3.bias", (d,), -0.1, 0.1, seed)
    out[prefix + "cls_token"] = _normal(prefix + "cls_token", (d,), 1.0, seed)
    for l in range(nlayers):
        lp = f"{prefix}model.layers.{l}."
        out[lp + "self_attn.in_proj_weight"] = _normal(lp + "self_attn.in_proj_weight", (3 * d, d), 1.0 / math.sqrt(d), seed)
        out[lp + "self_attn.in_proj_bias"] = _uniform(lp + "self_attn.in_proj_bias", (3 * d,), -0.05, 0.05, seed)
        out[lp + "self_attn.out_proj.weight"] = _normal(lp + "self_attn.out_proj.weight", (d, d), 1.0 / math.sqrt(d), seed)
        out[lp + "self_attn.out_proj.bias"] = _uniform(lp + "self_attn.out_proj.bias", (d,), -0.05, 0.05, seed)
        out[lp + "linear1.weight"] = _normal(lp + "linear1.weight", (dim_ff, d), math.sqrt(2.0 / d), seed)
        out[lp + "linear1.bias"] = _uniform(lp + "linear1.bias", (dim_ff,), -0.05, 0.05, seed)
        out[lp + "linear2.weight"] = _normal(lp + "linear2.weight", (d, dim_ff), 1.0 / math.sqrt(dim_ff), seed)
        out[lp + "linear2.bias"] = _uniform(lp + "linear2.bias", (d,), -0.05, 0.05, seed)
        for n in ("norm1", "norm2"):
            out[lp + n + ".weight"] = _uniform(lp + n + ".weight", (d,), 0.9, 1.1, seed)
            out[lp + n + ".bias"] = _uniform(lp + n + ".bias", (d,), -0.1, 0.1, seed)
    return out


def synthetic_wav(batch, n_samples, seed=BASE_SEED, varied=False, sample_rate=32000):
    """wav = clip(0.1*N(0,1), -1, 1), fp32, shape (B, L).

    ``varied=True`` (parity tests) gives every clip its own gain, a tone and a slow amplitude
    envelope so that different clips encode to visibly different features."""
    w = _normal(f"wav/{batch}/{n_samples}", (batch, n_samples), 0.1, seed)
    if varied:
        r = _rng(f"wavvar/{batch}/{n_samples}", seed)
        t = np.arange(n_samples, dtype=np.float64) / sample_rate
        for b in range(batch):
            gain = 10.0 ** r.uniform(-1.5, 0.3)
            f0 = r.uniform(100.0, 6000.0)
            fm = r.uniform(0.1, 2.0)
            env = 0.55 + 0.45 * np.sin(2 * np.pi * fm * t + r.uniform(0, 6.28))
            tone = r.uniform(0.02, 0.3) * np.sin(2 * np.pi * f0 * t * (1.0 + 0.2 * np.sin(2 * np.pi * 0.3 * t)))
            w[b] = (gain * env * w[b] + env[::-1] * tone).astype(np.float32)
    return np.clip(w, -1.0, 1.0).astype(np.float32)


def synthetic_logmel(batch, n_frames, n_mels=64, seed=BASE_SEED):
    """A log-mel-like input (dB scale) for fixtures that start downstream of the mel front-end:
    per-bin noise (5.6 dB, what |N(0,1)|^2 gives) around -25 dB plus, per clip, a smooth
    spectral tilt and temporal envelope of +-12 dB so that clips differ from one another."""
    base = _normal(f"lms/{batch}/{n_frames}", (batch, n_mels, n_frames), 5.6, seed)
    r = _rng(f"lmsvar/{batch}/{n_frames}", seed)
    f = np.linspace(0.0, 1.0, n_mels)[None, :, None]
    t = np.linspace(0.0, 1.0, n_frames)[None, None, :]
    a = r.uniform(-12.0, 12.0, size=(batch, 1, 1))
    b = r.uniform(-12.0, 12.0, size=(batch, 1, 1))
    c = r.uniform(0.5, 4.0, size=(batch, 1, 1))
    ph = r.uniform(0.0, 6.28, size=(batch, 1, 1))
    shape = a * (f - 0.5) * 2.0 + b * np.sin(2 * np.pi * c * t + ph) * (0.5 + f)
    return (base - np.float32(25.0) + shape.astype(np.float32)).astype(np.float32)


def to_torch(state):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}
