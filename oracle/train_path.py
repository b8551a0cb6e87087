"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's TRAINING step for the Cnn14Rnn-Trm path
(SURVEY.md section 8, rows A13-A16).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product path (``audiocaption_amd``) never does.

What it restates (plain torch-CPU fp32 functional ops, gradients by torch.autograd):

* frozen Cnn14 in train mode: F.dropout(p=0.2) after every conv block, BatchNorm in eval mode
  (cnn_encoder.py:431-444, crnn_trm_encoder.py:194-202);
* 3-layer bi-GRU with inter-layer dropout under pack_padded_sequence semantics (rnn_encoder.py:34-49);
* scheduled-sampling stepwise forward: step t feeds either cap[:, :t+1] or [start] + own greedy tokens, the
  decoder re-runs on the whole prefix with fresh dropout and only the last position's logit is kept
  (base.py:131-137,152-199, transformer_model.py:34-57, transformer_decoder.py:80-103); ss_ratio == 1 is the single
  teacher-forced pass of ``seq_forward`` (transformer_model.py:20-32);
* LabelSmoothingLoss (loss.py:51-74), clip_grad_norm_ (run.py:125), torch.optim.Adam with L2 weight decay
  (cnn14rnn_trm.yaml:42-46), ExponentialDecayScheduler (lr_scheduler.py:22-42).

Dropout cannot be bit-compared with the reference (torch's generator); with every p = 0 this file is pinned against
gradients produced by the reference itself (tests/golden/g8_train.npz, made by tests/golden/make_golden.py).
With p > 0 the masks are the counter hash of csrc/train.hip (splitmix64 of seed and element index), restated in
``drop_mask`` below, so the HIP path can be compared with this file with dropout ACTIVE.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_path as O

# operation codes that, together with the per-step base seed, give each dropout site its own stream
OP_CNN_BLOCK = 1        # + block index 0..5
OP_SPECAUG = 9
OP_GRU_LAYER = 10       # + layer index
OP_MEM = 20
OP_EMB_A, OP_EMB_B = 21, 22
OP_LAYER = 30           # + 10 * layer + {0: self-attn P, 1: dropout1, 2: cross-attn P, 3: dropout2, 4: ffn, 5: dropout3}


def op_seed(base_seed, op):
    return ((int(base_seed) << 16) + int(op)) & 0xFFFFFFFFFFFFFFFF


def drop_mask(seed, idx0, count, p):
    """float32 vector of ``count`` multipliers (0 or 1/(1-p)) for element indices idx0.. (csrc/train.hip drop_hash)."""
    if p <= 0.0:
        return np.ones(count, dtype=np.float32)
    with np.errstate(over="ignore"):
        idx = np.arange(idx0, idx0 + count, dtype=np.uint64)
        z = idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(32)).astype(np.uint64)
    t = min(int(float(np.float32(p)) * 4294967296.0), 4294967295)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(u >= t, scale, np.float32(0.0)).astype(np.float32)


def _mask_t(seed, idx0, shape, p):
    return torch.from_numpy(drop_mask(seed, idx0, int(np.prod(shape)), p).reshape(shape))


# ---------------------------------------------------------------------------------------------------------
# frozen Cnn14, train mode
# ---------------------------------------------------------------------------------------------------------
def specaug_stripes(seed, B, T, F=64, time_width=64, time_num=2, freq_width=8, freq_num=2):
    """(B, time_num + freq_num, 2) int32 (begin, length) stripes: torchlibrosa's SpecAugmentation as the reference
    configures it (cnn_encoder.py:352-354; DropStripes: width ~ randint(0, drop_width), begin ~ randint(0, total - width),
    all time stripes of the batch first, then all mel stripes).  PARITY UNPINNED for the draws themselves (torchlibrosa is
    not vendored and uses torch's generator); the masking arithmetic is what is compared."""
    rng = np.random.default_rng(int(seed) & 0xFFFFFFFFFFFFFFFF)
    out = np.zeros((B, time_num + freq_num, 2), dtype=np.int32)
    for b in range(B):
        for k in range(time_num):
            d = int(rng.integers(0, time_width))
            out[b, k] = (int(rng.integers(0, T - d)), d)
    for b in range(B):
        for k in range(freq_num):
            d = int(rng.integers(0, freq_width))
            out[b, time_num + k] = (int(rng.integers(0, F - d)), d)
    return out


def apply_specaug(lms, stripes):
    """lms (B, 64, T): zero the striped frames / mel bins (before bn0, cnn_encoder.py:423-429)."""
    out = lms.clone()
    for b in range(lms.shape[0]):
        for k in range(2):
            bg, ln = int(stripes[b, k, 0]), int(stripes[b, k, 1])
            out[b, :, bg:bg + ln] = 0.0
        for k in range(2, 4):
            bg, ln = int(stripes[b, k, 0]), int(stripes[b, k, 1])
            out[b, bg:bg + ln, :] = 0.0
    return out


def cnn14_train_from_logmel(state, lms, base_seed, p=0.2, rows_per_clip=None, prefix="encoder.cnn.", specaug=False):
    """As cpu_path.cnn14_from_logmel plus F.dropout(p) after every block.  The mask of block b is indexed over the
    HIP path's activation layout [clip][row < rows_per_clip[b]][w][c] (rows beyond the valid ones exist there as
    zero padding), so ``rows_per_clip`` (6 ints, the Hp of the level the block's output lives at) must be given
    when p > 0."""
    if specaug:
        lms = apply_specaug(lms, specaug_stripes(op_seed(base_seed, OP_SPECAUG), lms.shape[0], lms.shape[2]))
    x = lms.transpose(1, 2).unsqueeze(1)
    x = O._bn_eval(x.transpose(1, 3), state, prefix + "bn0").transpose(1, 3)
    for b in range(1, 7):
        q = f"{prefix}conv_block{b}."
        x = F.relu(O._bn_eval(F.conv2d(x, state[q + "conv1.weight"], padding=1), state, q + "bn1"))
        x = F.relu(O._bn_eval(F.conv2d(x, state[q + "conv2.weight"], padding=1), state, q + "bn2"))
        if b < 6:
            x = F.avg_pool2d(x, kernel_size=(2, 2))
        if p > 0:
            B, C, Hv, W = x.shape
            hp = rows_per_clip[b - 1]
            m = _mask_t(op_seed(base_seed, OP_CNN_BLOCK + b - 1), 0, (B, hp, W, C), p)[:, :Hv]
            x = x * m.permute(0, 3, 1, 2)
    return torch.mean(x, dim=3).transpose(1, 2)


# ---------------------------------------------------------------------------------------------------------
# GRU with inter-layer dropout
# ---------------------------------------------------------------------------------------------------------
def gru_train_forward(state, attn, attn_len, base_seed, p=0.5, prefix="encoder.rnn.", num_layers=3):
    lens = torch.as_tensor(attn_len).long()
    B, T, _ = attn.shape
    x = attn
    for l in range(num_layers):
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            q = f"{prefix}network."
            outs.append(O._gru_direction(x, lens, state[f"{q}weight_ih_l{l}{suf}"], state[f"{q}weight_hh_l{l}{suf}"],
                                         state[f"{q}bias_ih_l{l}{suf}"], state[f"{q}bias_hh_l{l}{suf}"], rev))
        x = torch.cat(outs, dim=-1)
        if l < num_layers - 1 and p > 0:
            x = x * _mask_t(op_seed(base_seed, OP_GRU_LAYER + l), 0, (B, T, x.shape[-1]), p)
    # The reference truncates to the longest clip (pad_packed_sequence); frames beyond a clip's length are zero
    # and masked as attention keys, so all T frames are kept here, like the HIP path does (static shapes; the
    # dropout mask of the audio memory is indexed over T frames per clip).
    return x


# ---------------------------------------------------------------------------------------------------------
# decoder pass with dropout (row space of csrc/train.hip: pass t holds rows row0 + n*L + l)
# ---------------------------------------------------------------------------------------------------------
def _mha_train(q_in, k_in, v_in, w, b, wo, bo, nhead, mask_add, pmask):
    N, Tq, d = q_in.shape
    Tk = k_in.shape[1]
    hd = d // nhead
    q = F.linear(q_in, w[:d], b[:d]).view(N, Tq, nhead, hd).transpose(1, 2)
    k = F.linear(k_in, w[d:2 * d], b[d:2 * d]).view(N, Tk, nhead, hd).transpose(1, 2)
    v = F.linear(v_in, w[2 * d:], b[2 * d:]).view(N, Tk, nhead, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd) + mask_add
    a = torch.softmax(s, dim=-1) * pmask
    o = torch.matmul(a, v).transpose(1, 2).reshape(N, Tq, d)
    return F.linear(o, wo, bo)


KINK = 1e-5


def _relu_at_kinks(pre, theirs):
    """ReLU whose gate, where the pre-activation is within KINK of zero, is taken from ``theirs`` (the activation the
    implementation under test kept for the same cells; None = plain ReLU).  The derivative of ReLU jumps at zero: a
    pre-activation of +-1e-6 - inside the forward tolerance - switches a whole gradient path on or off, so a gradient
    comparison has to put both sides on the same side of every such kink.  Values are unchanged to within KINK."""
    gate = pre > 0
    if theirs is not None:
        gate = torch.where(pre.detach().abs() < KINK, theirs.reshape(pre.shape) > 0, gate)
    return pre * gate


def decoder_pass(state, word, attn_emb, attn_emb_len, pad_idx, base_seed, p, row0, mrow0, seq0, pl, prefix="decoder.",
                 nlayers=2, nhead=4, relu_gates=None):
    """One decoder call of the training loop on tokens ``word`` (N, L); returns the (N, L, d) outputs.
    relu_gates: optional {"mem": (N*Tm, d), "ffn": [per layer (all rows of all passes, F)]} activations of the
    implementation under test, consulted only at ReLU kinks (see ``_relu_at_kinks``)."""
    d = state[prefix + "word_embedding.weight"].shape[1]
    N, L = word.shape
    Tm = attn_emb.shape[1]
    a = _relu_at_kinks(F.linear(attn_emb, state[prefix + "attn_proj.0.weight"], state[prefix + "attn_proj.0.bias"]),
                       relu_gates["mem"] if relu_gates else None)
    a = a * _mask_t(op_seed(base_seed, OP_MEM), mrow0 * d, (N, Tm, d), p)
    mem = F.layer_norm(a, (d,), state[prefix + "attn_proj.3.weight"], state[prefix + "attn_proj.3.bias"])
    x = state[prefix + "word_embedding.weight"][word] * _mask_t(op_seed(base_seed, OP_EMB_A), row0 * d, (N, L, d), p)
    x = x * math.sqrt(d) + state[prefix + "pos_encoder.pe"][:L, 0][None]
    x = x * _mask_t(op_seed(base_seed, OP_EMB_B), row0 * d, (N, L, d), p)
    neg = float("-inf")
    causal = torch.zeros(L, L).masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), neg)
    self_mask = causal[None, None] + torch.zeros(N, 1, 1, L).masked_fill((word == pad_idx)[:, None, None, :], neg)
    lens = torch.as_tensor(attn_emb_len)
    mem_mask = torch.zeros(N, 1, 1, Tm).masked_fill(~(torch.arange(Tm)[None, :] < lens[:, None])[:, None, None, :], neg)

    def rowmask(op, width=d):
        return _mask_t(op_seed(base_seed, op), row0 * width, (N, L, width), p)

    def pmask(op, ptk, tk):
        return _mask_t(op_seed(base_seed, op), seq0 * nhead * pl * ptk, (N, nhead, pl, ptk), p)[:, :, :L, :tk]

    for l in range(nlayers):
        lp = f"{prefix}model.layers.{l}."
        op = OP_LAYER + 10 * l
        sa = _mha_train(x, x, x, state[lp + "self_attn.in_proj_weight"], state[lp + "self_attn.in_proj_bias"],
                        state[lp + "self_attn.out_proj.weight"], state[lp + "self_attn.out_proj.bias"], nhead,
                        self_mask, pmask(op + 0, pl, L))
        x = F.layer_norm(x + sa * rowmask(op + 1), (d,), state[lp + "norm1.weight"], state[lp + "norm1.bias"])
        ca = _mha_train(x, mem, mem, state[lp + "multihead_attn.in_proj_weight"],
                        state[lp + "multihead_attn.in_proj_bias"], state[lp + "multihead_attn.out_proj.weight"],
                        state[lp + "multihead_attn.out_proj.bias"], nhead, mem_mask, pmask(op + 2, Tm, Tm))
        x = F.layer_norm(x + ca * rowmask(op + 3), (d,), state[lp + "norm2.weight"], state[lp + "norm2.bias"])
        pre = F.linear(x, state[lp + "linear1.weight"], state[lp + "linear1.bias"])
        dm = rowmask(op + 4, pre.shape[-1])
        theirs = None
        if relu_gates:      # kept AFTER dropout there: a dropped cell says nothing (and carries no gradient either)
            theirs = relu_gates["ffn"][l][row0:row0 + N * L].reshape(pre.shape)
            theirs = torch.where(dm > 0, theirs, pre.detach())
        hdn = _relu_at_kinks(pre, theirs) * dm
        ff = F.linear(hdn, state[lp + "linear2.weight"], state[lp + "linear2.bias"])
        x = F.layer_norm(x + ff * rowmask(op + 5), (d,), state[lp + "norm3.weight"], state[lp + "norm3.bias"])
    return x


def train_forward(state, attn_emb, attn_emb_len, cap, use_cap, base_seed=0, p_dec=0.2, start_idx=O.START_IDX,
                  pad_idx=O.PAD_IDX, teacher_forcing=False, prefix="decoder.", relu_gates=None):
    """Scheduled-sampling forward.  cap (N, Tc) int64; use_cap[t] = the draw ``random.random() < ss_ratio`` of
    step t (transformer_model.py:44).  Returns logit (N, Tc-1, V) and seq (N, Tc-1) (the greedy tokens).
    teacher_forcing=True is ``seq_forward``: one pass over cap[:, :-1], every position classified."""
    N, Tc = cap.shape
    T = Tc - 1
    Tm = attn_emb.shape[1]
    cls = state[prefix + "classifier.weight"]
    if teacher_forcing:
        x = decoder_pass(state, cap[:, :-1], attn_emb, attn_emb_len, pad_idx, base_seed, p_dec, 0, 0, 0, T, prefix,
                         relu_gates=relu_gates)
        logit = F.linear(x, cls)
        return {"logit": logit, "seq": logit.argmax(-1)}
    seq = torch.zeros(N, T, dtype=torch.long)
    logits = []
    row0 = 0
    for t in range(T):
        L = t + 1
        if use_cap[t]:
            word = cap[:, :L]
        else:
            word = torch.cat([torch.full((N, 1), start_idx, dtype=torch.long), seq[:, :t]], dim=1)
        x = decoder_pass(state, word, attn_emb, attn_emb_len, pad_idx, base_seed, p_dec, row0, t * N * Tm, t * N, T,
                         prefix, relu_gates=relu_gates)
        logit_t = F.linear(x[:, -1], cls)
        seq[:, t] = logit_t.detach().argmax(-1)
        logits.append(logit_t)
        row0 += N * L
    return {"logit": torch.stack(logits, dim=1), "seq": seq}


def label_smoothing_loss(logit, tgt, tgt_len, smoothing=0.1):
    """loss.py:51-74, reduction "mean"."""
    V = logit.shape[-1]
    lp = torch.log_softmax(logit, dim=-1)
    q = torch.full_like(lp, smoothing / (V - 1))
    q.scatter_(-1, tgt.unsqueeze(-1), 1.0 - smoothing)
    loss = torch.sum(-q * lp, dim=-1)
    mask = (torch.arange(logit.shape[1])[None, :] < torch.as_tensor(tgt_len)[:, None]).float()
    return (loss * mask).sum() / mask.sum()


TRAINABLE_PREFIXES = ("encoder.rnn.", "decoder.")


def trainable_keys(state):
    return [k for k in state if k.startswith(TRAINABLE_PREFIXES) and not k.endswith("pos_encoder.pe")]


def train_step_grads(state, cnn_attn, attn_len, cap, cap_len, use_cap, base_seed=0, p_dec=0.2, p_rnn=0.5,
                     smoothing=0.1, teacher_forcing=False, relu_gates=None):
    """Loss and gradients of one batch given the (frozen) Cnn14 output ``cnn_attn`` (B, T', 2048).
    relu_gates: see ``decoder_pass``."""
    keys = trainable_keys(state)
    st = dict(state)
    for k in keys:
        st[k] = state[k].detach().clone().requires_grad_(True)
    attn_emb = gru_train_forward(st, cnn_attn, attn_len, base_seed, p_rnn)
    out = train_forward(st, attn_emb, attn_len, cap, use_cap, base_seed, p_dec, teacher_forcing=teacher_forcing,
                        relu_gates=relu_gates)
    loss = label_smoothing_loss(out["logit"], cap[:, 1:], torch.as_tensor(cap_len) - 1, smoothing)
    grads = torch.autograd.grad(loss, [st[k] for k in keys], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(st[k])) for k, gr in zip(keys, grads)}
    return {"loss": loss.detach(), "logit": out["logit"].detach(), "seq": out["seq"], "grads": g,
            "attn_emb": attn_emb.detach()}


def clip_and_adam(params, grads, exp_avg, exp_avg_sq, step, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6,
                  max_norm=1.0):
    """clip_grad_norm_ (run.py:125) then one torch.optim.Adam update (L2 decay added to the gradient); dicts of
    tensors, updated in place; returns the total gradient norm before clipping."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0) if max_norm > 0 else torch.tensor(1.0)
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for k in params:
        g = grads[k] * coef + weight_decay * params[k]
        exp_avg[k].mul_(b1).add_(g, alpha=1 - b1)
        exp_avg_sq[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = exp_avg_sq[k].sqrt() / math.sqrt(bc2) + eps
        params[k].addcdiv_(exp_avg[k], denom, value=-lr / bc1)
    return total


def exponential_decay_lr(step_count, base_lr, final_lr, total_iters, warmup_iters):
    """ExponentialDecayScheduler._get_closed_form_lr (lr_scheduler.py:22-42) for its 1-based ``_step_count``."""
    if step_count <= warmup_iters:
        coeff = step_count / warmup_iters if step_count < warmup_iters else 1.0
        return coeff * base_lr
    base = (final_lr / base_lr) ** (1 / (total_iters - warmup_iters))
    return base_lr * base ** (step_count - warmup_iters)
