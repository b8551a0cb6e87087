"""CPU oracle for the Cnn14Rnn-Trm captioning hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, with plain fp32 PyTorch CPU ops, what the reference computes on the path
    wav -> log-mel -> Cnn14 -> bi-GRU -> Transformer decoder -> greedy / beam search.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product path (``audiocaption_amd``) never does and fails loudly when its HIP
library is missing.

Pinning status:
* everything from the log-mel input onward is PINNED: ``tests/golden/make_golden.py`` imports
  the reference from /root/reference in the build container, loads the same procedural weights
  and checks every function below against the reference modules; the fixtures it wrote are in
  ``tests/golden/*.npz`` and are re-checked by ``tests/test_oracle_golden.py``.
* ``logmel``: its arithmetic lives in torchaudio==0.13.1 (requirements.txt:5), which is not
  vendored in the reference and not installed here, so the imported reference cannot pin it.
  It follows torchaudio's published MelSpectrogram/AmplitudeToDB semantics (call sites
  cnn_encoder.py:338-350,418-419) and is PINNED BY AN INDEPENDENT WITNESS instead:
  ``tests/golden/make_witness.py`` runs ``transformers.audio_utils`` (mel_filter_bank +
  spectrogram, float64) on seeded inputs, writes ``tests/golden/g10_logmel.npz`` and asserts
  agreement (filterbank 2e-8, log-mel 3.4e-4 dB max / 2e-5 dB 99th percentile);
  ``tests/test_witness.py`` re-checks this file against that fixture on any machine, next to
  the float64 numpy DFT and closed-form answers of ``tests/test_logmel_oracle.py``.

All functions take a flat ``state`` dict of torch tensors keyed as the reference's
``state_dict()``.
"""
import math

import torch
import torch.nn.functional as F

PAD_IDX, START_IDX, END_IDX = 0, 1, 2  # reference base.py:12-15


# ----------------------------------------------------------------------------------------
# log-mel front-end (torchaudio semantics restated; pinned by the transformers.audio_utils witness, see header)
# ----------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if f >= min_log_hz:
        return min_log_mel + math.log(f / min_log_hz) / logstep
    return f / f_sp


def _mel_to_hz_slaney(mels):
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    freqs = f_sp * mels
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def mel_filterbank(sample_rate=32000, n_fft=1024, n_mels=64, f_min=50.0, f_max=14000.0):
    """(n_freqs, n_mels) slaney-scale, slaney-normalised triangular filters.

    torchaudio.functional.melscale_fbanks(norm="slaney", mel_scale="slaney") as called through
    MelSpectrogram at cnn_encoder.py:338-348 (f_min=50, f_max=14000, n_mels=64).
    """
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return fb * enorm.unsqueeze(0)


def logmel(wav, sample_rate=32000):
    """wav (B, L) fp32 -> log-mel (B, 64, T), T = L // hop + 1.

    Spectrogram(power=2, periodic Hann, center=True, reflect pad) -> mel matmul ->
    10*log10(clamp(x, 1e-10))   (reference cnn_encoder.py:418-419; AmplitudeToDB top_db=None).
    """
    n_fft = 32 * sample_rate // 1000
    hop = 10 * sample_rate // 1000
    f_max = {32000: 14000.0, 16000: 8000.0}[sample_rate]
    window = torch.hann_window(n_fft, periodic=True)
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.abs().pow(2.0)  # (B, n_freqs, T)
    fb = mel_filterbank(sample_rate, n_fft, 64, 50.0, f_max)
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return 10.0 * torch.log10(torch.clamp(mel, min=1e-10))


# ----------------------------------------------------------------------------------------
# Cnn14 (reference cnn_encoder.py:414-464, ConvBlock :59-75); eval mode only
# ----------------------------------------------------------------------------------------
def _bn_eval(x, state, prefix, eps=1e-5):
    return F.batch_norm(x, state[prefix + ".running_mean"], state[prefix + ".running_var"],
                        state[prefix + ".weight"], state[prefix + ".bias"], False, 0.0, eps)


def cnn14_feat_len(wav_len, hop=320, ratio=32):
    """attn_emb_len = floor((floor(L/hop)+1)/32)  (cnn_encoder.py:446-450)."""
    wav_len = torch.as_tensor(wav_len)
    n = torch.div(wav_len, hop, rounding_mode="floor") + 1
    return torch.div(n, ratio, rounding_mode="floor")


def cnn14_from_logmel(state, lms, prefix="encoder.cnn.", return_blocks=False):
    """lms (B, 64, T) -> attn_emb (B, T//32, 2048).  bn0 acts on the mel axis."""
    x = lms.transpose(1, 2).unsqueeze(1)  # (B, 1, T, 64)
    x = _bn_eval(x.transpose(1, 3), state, prefix + "bn0").transpose(1, 3)
    blocks = []
    for b in range(1, 7):
        p = f"{prefix}conv_block{b}."
        x = F.relu(_bn_eval(F.conv2d(x, state[p + "conv1.weight"], padding=1), state, p + "bn1"))
        x = F.relu(_bn_eval(F.conv2d(x, state[p + "conv2.weight"], padding=1), state, p + "bn2"))
        if b < 6:
            x = F.avg_pool2d(x, kernel_size=(2, 2))
        blocks.append(x)
    attn_emb = torch.mean(x, dim=3).transpose(1, 2)
    if return_blocks:
        return attn_emb, blocks
    return attn_emb


def cnn14_forward(state, wav, wav_len, prefix="encoder.cnn."):
    lms = logmel(wav, 32000)
    return {"attn_emb": cnn14_from_logmel(state, lms, prefix),
            "attn_emb_len": cnn14_feat_len(wav_len)}


# ----------------------------------------------------------------------------------------
# RnnEncoder: packed 3-layer bi-GRU + length-aware mean (rnn_encoder.py:34-49,
# model_util.py:10-27,41-63).  Written as explicit cell recurrences, gate order r,z,n.
# ----------------------------------------------------------------------------------------
def _gru_direction(x, lens, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (B, T, I); rows beyond lens[b] produce zeros, state starts at 0 at each clip's first
    valid step (for the reverse direction: its own last valid frame)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = F.linear(x, w_ih, b_ih)  # (B, T, 3H)
    out = torch.zeros(B, T, H)
    h = torch.zeros(B, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gx[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gx[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gx[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h_new = (1.0 - z) * n + z * h
        valid = (t < lens).unsqueeze(1)
        h = torch.where(valid, h_new, h)
        out[:, t] = torch.where(valid, h_new, torch.zeros_like(h_new))
    return out


def gru_forward(state, attn, attn_len, prefix="encoder.rnn.", num_layers=3):
    """attn (B, T, 2048), attn_len (B,) -> attn_emb (B, max(len), 512), fc_emb (B, 512)."""
    lens = torch.as_tensor(attn_len).long()
    t_out = int(lens.max())
    x = attn[:, :t_out]
    for l in range(num_layers):
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            p = f"{prefix}network."
            outs.append(_gru_direction(
                x, lens, state[f"{p}weight_ih_l{l}{suf}"], state[f"{p}weight_hh_l{l}{suf}"],
                state[f"{p}bias_ih_l{l}{suf}"], state[f"{p}bias_hh_l{l}{suf}"], rev))
        x = torch.cat(outs, dim=-1)  # inter-layer dropout is inactive in eval mode
    mask = (torch.arange(t_out)[None, :] < lens[:, None]).unsqueeze(-1)
    fc_emb = (x * mask).sum(1) / lens.unsqueeze(1)
    return {"attn_emb": x, "fc_emb": fc_emb, "attn_emb_len": lens}


# ----------------------------------------------------------------------------------------
# TransformerDecoder (transformer_decoder.py:80-103): post-LN nn.TransformerDecoderLayer x2
# ----------------------------------------------------------------------------------------
def _mha(q_in, k_in, v_in, w, b, wo, bo, nhead, mask_add):
    """q_in (N, Tq, d), k_in/v_in (N, Tk, d); mask_add broadcastable to (N, h, Tq, Tk)."""
    N, Tq, d = q_in.shape
    Tk = k_in.shape[1]
    hd = d // nhead
    q = F.linear(q_in, w[:d], b[:d]).view(N, Tq, nhead, hd).transpose(1, 2)
    k = F.linear(k_in, w[d:2 * d], b[d:2 * d]).view(N, Tk, nhead, hd).transpose(1, 2)
    v = F.linear(v_in, w[2 * d:], b[2 * d:]).view(N, Tk, nhead, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd) + mask_add
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(N, Tq, d)
    return F.linear(o, wo, bo)


def decoder_memory(state, attn_emb, prefix="decoder."):
    """attn_proj = Linear -> ReLU -> (Dropout) -> LayerNorm  (transformer_decoder.py:38-43,86)."""
    p = F.relu(F.linear(attn_emb, state[prefix + "attn_proj.0.weight"], state[prefix + "attn_proj.0.bias"]))
    return F.layer_norm(p, (p.shape[-1],), state[prefix + "attn_proj.3.weight"], state[prefix + "attn_proj.3.bias"])


def decoder_forward(state, word, attn_emb, attn_emb_len, cap_padding_mask=None, prefix="decoder.",
                    nlayers=2, nhead=4):
    """word (N, T) int64 -> {"embed": (N, T, d), "logit": (N, T, V)}."""
    d = state[prefix + "word_embedding.weight"].shape[1]
    N, T = word.shape
    mem = decoder_memory(state, attn_emb, prefix)
    Ts = mem.shape[1]
    x = state[prefix + "word_embedding.weight"][word] * math.sqrt(d)
    x = x + state[prefix + "pos_encoder.pe"][:T, 0][None]
    neg = float("-inf")
    causal = torch.zeros(T, T).masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), neg)
    self_mask = causal[None, None]
    if cap_padding_mask is not None:
        self_mask = self_mask + torch.zeros(N, 1, 1, T).masked_fill(cap_padding_mask[:, None, None, :], neg)
    lens = torch.as_tensor(attn_emb_len)
    mem_pad = ~(torch.arange(Ts)[None, :] < lens[:, None])
    mem_mask = torch.zeros(N, 1, 1, Ts).masked_fill(mem_pad[:, None, None, :], neg)
    for l in range(nlayers):
        lp = f"{prefix}model.layers.{l}."
        sa = _mha(x, x, x, state[lp + "self_attn.in_proj_weight"], state[lp + "self_attn.in_proj_bias"],
                  state[lp + "self_attn.out_proj.weight"], state[lp + "self_attn.out_proj.bias"], nhead, self_mask)
        x = F.layer_norm(x + sa, (d,), state[lp + "norm1.weight"], state[lp + "norm1.bias"])
        ca = _mha(x, mem, mem, state[lp + "multihead_attn.in_proj_weight"], state[lp + "multihead_attn.in_proj_bias"],
                  state[lp + "multihead_attn.out_proj.weight"], state[lp + "multihead_attn.out_proj.bias"], nhead, mem_mask)
        x = F.layer_norm(x + ca, (d,), state[lp + "norm2.weight"], state[lp + "norm2.bias"])
        ff = F.linear(F.relu(F.linear(x, state[lp + "linear1.weight"], state[lp + "linear1.bias"])),
                      state[lp + "linear2.weight"], state[lp + "linear2.bias"])
        x = F.layer_norm(x + ff, (d,), state[lp + "norm3.weight"], state[lp + "norm3.bias"])
    return {"embed": x, "logit": F.linear(x, state[prefix + "classifier.weight"])}


# ----------------------------------------------------------------------------------------
# TransformerEncoder (transformer_encoder.py:93-116): attn_proj, cls token, post-LN nn.TransformerEncoderLayer x2
# ----------------------------------------------------------------------------------------
def transformer_encoder_forward(state, attn, attn_len, prefix="", nlayers=2, nhead=4):
    """attn (N, T, A), attn_len (N,) -> attn_emb (N, T+1, d), fc_emb = attn_emb[:, 0], attn_emb_len = attn_len + 1."""
    x = decoder_memory(state, attn, prefix)   # the same Linear -> ReLU -> (Dropout) -> LayerNorm stack
    N, T, d = x.shape
    x = torch.cat([state[prefix + "cls_token"].reshape(1, 1, d).repeat(N, 1, 1), x], dim=1)
    lens = torch.as_tensor(attn_len).clone() + 1
    pad = ~(torch.arange(T + 1)[None, :] < lens[:, None])
    mask = torch.zeros(N, 1, 1, T + 1).masked_fill(pad[:, None, None, :], float("-inf"))
    for l in range(nlayers):
        lp = f"{prefix}model.layers.{l}."
        sa = _mha(x, x, x, state[lp + "self_attn.in_proj_weight"], state[lp + "self_attn.in_proj_bias"],
                  state[lp + "self_attn.out_proj.weight"], state[lp + "self_attn.out_proj.bias"], nhead, mask)
        x = F.layer_norm(x + sa, (d,), state[lp + "norm1.weight"], state[lp + "norm1.bias"])
        ff = F.linear(F.relu(F.linear(x, state[lp + "linear1.weight"], state[lp + "linear1.bias"])),
                      state[lp + "linear2.weight"], state[lp + "linear2.bias"])
        x = F.layer_norm(x + ff, (d,), state[lp + "norm2.weight"], state[lp + "norm2.bias"])
    return {"attn_emb": x, "fc_emb": x[:, 0], "attn_emb_len": lens}


# ----------------------------------------------------------------------------------------
# greedy decoding (base.py:152-218, transformer_model.py:34-57)
# ----------------------------------------------------------------------------------------
def greedy_decode(state, attn_emb, attn_emb_len, max_length=20, prefix="decoder.",
                  start_idx=START_IDX, end_idx=END_IDX, pad_idx=PAD_IDX, force_steps=False):
    """Returns seq (B, max_length) int64 (end_idx after a clip finishes), logit (B, L, V),
    sampled_logprob (B, L), embed (B, L, d) and ``steps`` = decoder calls executed.  Columns after
    the early stop are left at their init (logit/embed there are unspecified in the reference,
    which uses torch.empty: base.py:124-127); here they are zeros."""
    B = attn_emb.shape[0]
    V, d = state[prefix + "classifier.weight"].shape
    seq = torch.full((B, max_length), end_idx, dtype=torch.long)
    logit = torch.zeros(B, max_length, V)
    logprob = torch.zeros(B, max_length)
    embed = torch.zeros(B, max_length, d)
    unfinished = None
    steps = 0
    for t in range(max_length):
        word = torch.cat([torch.full((B, 1), start_idx, dtype=torch.long), seq[:, :t]], dim=1)
        out = decoder_forward(state, word, attn_emb, attn_emb_len, word == pad_idx, prefix)
        logit_t = out["logit"][:, -1]
        lp, w = torch.max(torch.log_softmax(logit_t, dim=1), 1)
        logit[:, t], seq[:, t], logprob[:, t], embed[:, t] = logit_t, w, lp, out["embed"][:, -1]
        steps += 1
        un_t = seq[:, t] != end_idx
        unfinished = un_t if t == 0 else unfinished & un_t
        seq[:, t][~unfinished] = end_idx
        if unfinished.sum() == 0 and not force_steps:
            break
    return {"seq": seq, "logit": logit, "sampled_logprob": logprob, "embed": embed, "steps": steps}


# ----------------------------------------------------------------------------------------
# beam search (base.py:254-361, transformer_model.py:59-86)
# ----------------------------------------------------------------------------------------
def beam_search(state, attn_emb, attn_emb_len, beam_size=3, max_length=20, temp=1.0, prefix="decoder.",
                start_idx=START_IDX, end_idx=END_IDX, pad_idx=PAD_IDX, n_best=False, n_best_size=None, trace=None):
    """base.py:254-361.  n_best: "seq" is (B, n_best_size, max_length), the finished beams of a clip by descending
    length-normalised score (base.py:258-263,354-358).  trace (a list, fixtures only): receives one record per (clip,
    step) - the parent beam of every kept candidate and which of them ended."""
    B = attn_emb.shape[0]
    n_best_size = beam_size if n_best_size is None else n_best_size
    nbest_seq = torch.full((B, n_best_size, max_length), end_idx, dtype=torch.long)
    V = state[prefix + "classifier.weight"].shape[0]
    lens = torch.as_tensor(attn_emb_len)
    out_seq = torch.full((B, max_length), end_idx, dtype=torch.long)
    scores = torch.zeros(B)
    for i in range(B):
        mem_i = attn_emb[i:i + 1].repeat(beam_size, 1, 1)
        len_i = lens[i:i + 1].repeat(beam_size)
        topk_logprob = torch.zeros(beam_size)
        seq = None
        done = []
        for t in range(max_length):
            start = torch.full((beam_size, 1), start_idx, dtype=torch.long)
            word = start if t == 0 else torch.cat([start, seq], dim=1)
            logit_t = decoder_forward(state, word, mem_i, len_i, word == pad_idx, prefix)["logit"][:, -1]
            lp = torch.log_softmax(torch.log_softmax(logit_t, dim=1) / temp, dim=1)
            lp = topk_logprob.unsqueeze(1) + lp
            if t == 0:
                topk_logprob, topk_words = lp[0].topk(beam_size, 0, True, True)
            else:
                topk_logprob, topk_words = lp.view(-1).topk(beam_size, 0, True, True)
            prev_beam = torch.div(topk_words, V, rounding_mode="trunc")
            next_word = topk_words % V
            seq = next_word.unsqueeze(1) if t == 0 else torch.cat([seq[prev_beam], next_word.unsqueeze(1)], dim=1)
            is_end = next_word == end_idx
            if t == max_length - 1:
                is_end = torch.ones_like(is_end)
            if trace is not None:
                cand = (lp[0] if t == 0 else lp.view(-1)).topk(beam_size + 1).values
                trace.append({"clip": i, "t": t, "prev_beam": prev_beam.tolist(), "ended": is_end.tolist(),
                              "margin": float((cand[:-1] - cand[1:]).min())})   # smallest gap among the kept and to the first cut
            for b in range(beam_size):
                if is_end[b]:
                    done.append({"seq": seq[b].clone(), "score": topk_logprob[b].item() / (t + 1)})
            topk_logprob = topk_logprob.clone()
            topk_logprob[is_end] -= 1000
            if len(done) == beam_size:
                break
        done = sorted(done, key=lambda x: -x["score"])
        best = done[0]["seq"]
        out_seq[i, :len(best)] = best
        scores[i] = done[0]["score"]
        for j, d in enumerate(done[:n_best_size]):
            nbest_seq[i, j, :len(d["seq"])] = d["seq"]
    return {"seq": nbest_seq if n_best else out_seq, "score": scores}


# ----------------------------------------------------------------------------------------
# whole path (what bench.py's cpu_baseline leg times)
# ----------------------------------------------------------------------------------------
def caption_forward(state, wav, wav_len, sample_method="greedy", beam_size=3, max_length=20,
                    force_steps=False):
    enc = cnn14_forward(state, wav, wav_len)
    enc = gru_forward(state, enc["attn_emb"], enc["attn_emb_len"])
    if sample_method == "beam":
        out = beam_search(state, enc["attn_emb"], enc["attn_emb_len"], beam_size, max_length)
    else:
        out = greedy_decode(state, enc["attn_emb"], enc["attn_emb_len"], max_length, force_steps=force_steps)
    out.update(enc)
    return out
