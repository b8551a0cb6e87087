"""Cnn14 waveform encoder, MI355X path.  Plugin-compatible with the reference class
``captioning.models.cnn_encoder.Cnn14Encoder`` (cnn_encoder.py:330-464): same constructor, same
``forward(input_dict) -> {"fc_emb", "attn_emb", "attn_emb_len"}`` contract, same ``state_dict()`` keys
(SURVEY.md §2.4), same ``load_pretrained`` hook.

The torch.nn sub-modules below only OWN the parameters (so checkpoints, ``.to()``, optimizers and
``named_parameters`` behave exactly like the reference); the forward pass never calls them: it runs
log-mel -> bn0 -> 12 x (conv3x3 + BN + ReLU [+ pool]) -> mean over mel entirely in the HIP kernels of
``csrc/logmel.hip`` and ``csrc/conv3x3.hip`` and fails loudly if they are unavailable.
"""
import os

import torch
import torch.nn as nn

from . import kernels as K
from .mel import MelSpectrogramBuffers, MelTables

CHANNELS = [1, 64, 128, 256, 512, 1024, 2048]


class ConvBlock(nn.Module):
    """Parameter container for two bias-free 3x3 convs + BatchNorms (reference cnn_encoder.py:32-57)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        for conv in (self.conv1, self.conv2):
            nn.init.xavier_uniform_(conv.weight)


def cnn14_feat_len(wav_len, hop, ratio=32):
    """attn_emb_len = floor((floor(L / hop) + 1) / 32) as a CPU int64 tensor (cnn_encoder.py:446-450)."""
    wav_len = torch.as_tensor(wav_len).cpu()
    n = torch.div(wav_len, hop, rounding_mode="floor") + 1
    return torch.div(n, ratio, rounding_mode="floor").long()


def wino1d_covers(cout):
    """Layers the F(2,3) kernel runs: 128-channel column tiles, and the 64-channel / 16-column form of conv2 of block 1
    (AUDIOCAPTION_W1_C64=0 sends that layer back to the direct split-bf16 kernel).  Anything else: direct split-bf16."""
    return cout % 128 == 0 or (cout == 64 and os.environ.get("AUDIOCAPTION_W1_C64", "1") != "0")


def _wino1d_clip_chunk(B, Hp, W, Cin):
    """Clips one F(2,3) launch may take: the kernel addresses its input with 32-bit BYTE offsets through one buffer descriptor
    ((B * Hp + 16) * W * Cin * 4 < 2^31, csrc/conv3x3_wino1d.hip) - 127 ten-second clips at conv2 of block 1."""
    return max(1, min(B, ((1 << 31) - 1) // (W * Cin * 4 * Hp) - 1))


def _conv_wino1d(x, w, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1, need=None, splitk_buf=None,
                 dropout=None):
    """The "wino1d" tier's launcher (``_pack`` packs a layer's weights for the kernel ``wino1d_covers`` names).
    ``splitk_buf(floats) -> tensor``: workspace provider for the K-sliced launches of single clips.  A batch whose input
    exceeds the kernel's 2 GiB addressing range is convolved in clip chunks (clips do not interact; rows are per clip)."""
    if not wino1d_covers(Cout):
        K.conv3x3_bn_relu_bf16x3_gw(x, w, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode)
        if dropout is not None:
            K.dropout_(out, out.numel() if mode != 1 else B * (Hp // 2) * (W // 2) * Cout, *dropout)
        return out
    chunk = _wino1d_clip_chunk(B, Hp, W, Cin)
    if chunk < B:
        if dropout is not None:
            raise ValueError("F(2,3) conv: the dropout epilogue indexes the whole output buffer; batches beyond 2 GiB of input "
                             "are not supported in train mode")
        in_clip = Hp * W * Cin
        out_clip = {0: Hp * W * Cout, 1: (Hp // 2) * (W // 2) * Cout, 2: H * Cout}[mode]
        xf, of = x.reshape(-1), out.reshape(-1)
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            nd = (need[0][b0:b0 + nb], need[1], need[2]) if need is not None else None
            _conv_wino1d(xf[b0 * in_clip:(b0 + nb) * in_clip], w, scale, shift, of[b0 * out_clip:(b0 + nb) * out_clip], nb, Hp, H, W,
                         Cin, Cout, mode, map_mode, need=nd, splitk_buf=splitk_buf)
        return out
    if dropout is not None:   # F.dropout on the block's output in the kernel's epilogue (no extra pass over the buffer)
        return K.conv3x3_bn_relu_wino1d(x, w, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, dropout=dropout)
    ws = None
    if splitk_buf is not None:
        floats = K.wino1d_splitk_floats(B, Hp, W, Cin, Cout)
        ws = splitk_buf(floats) if floats else None
    return K.conv3x3_bn_relu_wino1d(x, w, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, need=need, workspace=ws)


WINO = ("wino43", "wino1d")   # the Winograd tiers: same layouts, epilogues, ragged-batch and dropout hooks
W43_MIN_WORKGROUPS = int(os.environ.get("AUDIOCAPTION_W43_MIN_WG", "192"))


def wino43_covers(W, cout):
    """Layers the F(4,3) kernel runs (one 512-register wave per SIMD, csrc/conv3x3_wino43.hip): the full-width forms of
    conv blocks 2-5 and the column-tile form of block 6 (2 mel columns).  Block 1 has its own kernel
    (csrc/conv3x3_block1_w4.hip)."""
    return W in (32, 16, 8, 4, 2) and cout % 128 == 0


def _conv_wino43(x, w, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode=-1, need=None, splitk_buf=None,
                 dropout=None, cold=False):
    """The "wino43" tier's launcher: ``w`` is (F(2,3) pack, F(4,3) pack or None).  F(4,3) when the kernel covers the layer
    and the launch fills the chip (a workgroup owns a CU: single clips run the K-sliced F(2,3) form instead)."""
    w23, w43, lazy = (tuple(w) + (None,))[:3] if isinstance(w, tuple) else (w, None, None)
    # (AUDIOCAPTION_RAGGED_EXACT=1 only: batches of uneven lengths then keep block 6 on F(2,3) - the caller passes it the F(2,3)
    # pack only: its tiles are taller than a clip, nothing can be skipped there anyway, and the quad-wide input window of
    # F(4,3) would cost every layer upstream four more valid rows per clip - ``rows_needed``.  In the default mode ragged
    # batches run F(4,3) on block 6 too and valid frames are within 5e-5 of the dense run, not bit-identical)
    if w43 is not None and (mode != 1 if W == 2 else mode != 2) and Hp % 4 == 0 \
            and K.wino43_workgroups(B, Hp, W, Cout) >= W43_MIN_WORKGROUPS:
        return K.conv3x3_bn_relu_wino43(x, w43, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, need=need,
                                        dropout=dropout)
    # Few pixels, heavy weights, COLD weights: blocks 5-6 in the training step at the reference's per-GPU batch of 4 - the
    # step touches a few hundred MB between two uses of these 360 MB of weights, so they come from HBM every iteration, and
    # the dropout epilogue of the F(2,3) kernel has no K-sliced form: 2 x 390 us for block 6 of four clips; the
    # weight-streaming direct form takes ~110 (tools/small_batch_bench.py, tools/train_bench.py --batch 4: 4.30 -> 3.79 ms
    # per step).  Inference calls re-use the weights from the memory-side cache call after call, where the K-sliced F(2,3)
    # form is as fast or faster (B = 1: 36 + 61 us vs 27 + 55 for block 6, 34 + 48 vs 39 + 53 for block 5; B = 4: 116 vs
    # 165): they keep it.
    if lazy is not None and need is None and splitk_buf is not None and SKINNY and (cold or dropout is not None) \
            and B * Hp * W <= SKINNY_MAX_PX:
        n = K.skinny_workspace_floats(B, Hp, W, Cin, Cout)
        if n > 0:
            return K.conv3x3_bn_relu_skinny(x, lazy.get(), scale, shift, out, B, Hp, H, W, Cin, Cout, mode, splitk_buf(n),
                                            dropout=dropout)
    return _conv_wino1d(x, w23, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, need=need, splitk_buf=splitk_buf,
                        dropout=dropout)


# conv1 of block 1 inside the one-kernel block: "mfma" (split-bf16 product on the matrix cores) | "valu" (the f32 chain of
# ac_conv3x3_first, bit-identical to the two-kernel form)
BLOCK1_CONV1 = os.environ.get("AUDIOCAPTION_BLOCK1_CONV1", "mfma")
SKINNY = os.environ.get("AUDIOCAPTION_SKINNY", "1") != "0"
SKINNY_MAX_PX = int(os.environ.get("AUDIOCAPTION_SKINNY_MAX_PX", "1024"))   # pixels (B * Hp * W) up to which it is the faster form


class _LazyPack:
    """The direct split-bf16 pack of a layer's weights (csrc/conv3x3_skinny.hip), built on first use: only small launches
    need it, and conv2 of block 6 alone is 151 MB."""

    def __init__(self, weight):
        self._weight, self._pk = weight, None

    def get(self):
        if self._pk is None:
            self._pk = K.pack_conv_weight_bf16x3_frag(self._weight.detach().float())
        return self._pk


def ragged_exact():
    """AUDIOCAPTION_RAGGED_EXACT=1: a ragged batch returns the BITS of the run that convolves all its padding (``rows_needed(
    quads=True)``, block 6 on F(2,3)).  Default 0: every conv skips what lies beyond the rows an output frame DEPENDS on; the
    F(4,3) kernels then form the last live quad of a clip from input rows that were themselves skipped (zeros) beyond that
    line - rows whose contribution cancels in exact arithmetic and to the tier's 2^-16 operand error here: valid frames
    within 5e-5 of the dense run (the bar the tier is held to anyway), a fifth fewer rows convolved on Clotho-shaped batches."""
    return os.environ.get("AUDIOCAPTION_RAGGED_EXACT", "0") == "1"


def rows_needed(block, conv, quads=False):
    """(mul, add): output rows of conv ``conv`` (1 | 2) of conv block ``block`` (1..6) that can reach an output frame below a
    clip's own ``attn_emb_len`` = mul * attn_emb_len + add.  Block 6 is not pooled: its second conv needs exactly the
    frames, its first one row more; every 3x3 conv upstream adds a row, every pooling doubles (cnn_encoder.py:431-441).
    ``quads`` (the "wino43" tier): the F(4,3) kernel forms every output row of a QUAD from all six input rows of the quad
    (the rows a given output does not depend on cancel in exact arithmetic, not bit for bit), so a conv of blocks 2-5 that
    needs N rows wants its input valid up to the end of the last quad + 1 <= N + 4 rows instead of N + 1: results below
    ``attn_emb_len`` then stay bit-identical to the full convolution."""
    if block == 6:
        return (1, 0) if conv == 2 else (1, 1)
    if not quads:
        mul, add = 1 << (6 - block), (1 << (8 - block)) - 4
        return (mul, add) if conv == 2 else (mul, add + 1)
    need = 2 * 2                      # conv2 of block 5 = 2 * (frames + 2): tracked as mul * frames + add
    mul = 2
    for b in range(5, block, -1):     # walk up: conv1 of block b (+4), pooled output of block b - 1 (+4), doubled
        need = 2 * (need + 8)
        mul *= 2
    if block == 1:                    # conv2 of block 1 runs the F(2,3) kernel; what it feeds (conv1 of block 2) is F(4,3)
        return (mul, need) if conv == 2 else (mul, need + 1)
    return (mul, need) if conv == 2 else (mul, need + 4)


def conv_kernel(algo):
    """The launcher of a conv tier (``Cnn14.conv_algo``)."""
    return {"winograd": K.conv3x3_bn_relu_winograd, "direct": K.conv3x3_bn_relu,
            "bf16x3": K.conv3x3_bn_relu_bf16x3_gw, "bf16x3_lds": K.conv3x3_bn_relu_bf16x3,
            "f16x2": K.conv3x3_bn_relu_f16x2_gw, "wino1d": _conv_wino1d, "wino43": _conv_wino43}[algo]


class Cnn14Encoder(nn.Module):

    def __init__(self, sample_rate=32000, freeze=False):
        super().__init__()
        sr_to_fmax = {32000: 14000, 16000: 8000}
        self.sample_rate = sample_rate
        self.n_fft = 32 * sample_rate // 1000
        self.hop_length = 10 * sample_rate // 1000
        self.f_min, self.f_max = 50.0, float(sr_to_fmax[sample_rate])
        self.melspec_extractor = MelSpectrogramBuffers(sample_rate, self.n_fft, self.f_min, self.f_max, 64, "slaney",
                                                       "slaney")
        self.bn0 = nn.BatchNorm2d(64)
        for b in range(6):
            setattr(self, f"conv_block{b + 1}", ConvBlock(CHANNELS[b], CHANNELS[b + 1]))
        self.downsample_ratio = 32
        self.fc1 = nn.Linear(2048, 2048, bias=True)
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.zeros_(self.fc1.bias)
        self.fc_emb_size = 2048
        self.freeze = freeze
        # Conv tiers.  "wino43" (default): Winograd along time on split-bf16 operands, f32 activations - f32-grade parity
        # with the fp32 reference (logits within 1e-4, identical token ids): F(4,3) (1.5 bf16 MFMA products per f32 product)
        # on conv blocks 2-5 when the launch fills the chip, F(2,3) (two products) elsewhere.  "wino1d": F(2,3) everywhere.  "bf16x3": the direct form on split-bf16 operands, three products (same accuracy).  "f16x2" (opt-in,
        # half-precision gate): fp16 activations (kept as fp16 in HBM), fp16 hi + lo weights, two fp16 MFMA products -
        # identical token ids, logits within 1e-3 (BASELINE.json's half-precision bar), NOT reference precision.
        # "winograd": F(2x2,3x3) on the f32 MFMA, exact f32.  "direct": 9-tap f32 implicit GEMM.  "bf16x3_lds": bf16x3
        # with an LDS weight ring (kept for ablations).  The train-mode forward never uses "f16x2".
        self.conv_algo = os.environ.get("AUDIOCAPTION_CONV_ALGO", "wino43")
        self.f16x2_min_frames = int(os.environ.get("AUDIOCAPTION_F16X2_MIN_FRAMES", "10"))
        # The "f16x2" tier is MIXED: conv_block6 (K = 9216 / 18432, two pixels per frame to average over - half of the
        # tier's logit error by the per-layer breakdown of DESIGN.md section 4) runs on the split-bf16 kernel with f32
        # activations; block 5's pooled output is then written as f32.  "f16x2" here restores the pure fp16 tier.
        self.f16x2_block6 = os.environ.get("AUDIOCAPTION_F16X2_BLOCK6", "bf16x3")
        self._tables = None
        self._tables_key = None
        self._packed = {}   # conv tier -> (key of the tensors it was packed from, packed weights)
        self._bufs = {}

    # ---- checkpoint hook (reference cnn_encoder.py:376-412: PANNs / COLA / BLAT layouts) ----------
    def load_pretrained(self, pretrained, output_fn=print):
        checkpoint = torch.load(pretrained, map_location="cpu")
        if "model" in checkpoint:
            sd = checkpoint["model"]
            if any(k.startswith("backbone.") for k in sd):  # COLA
                sd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
        elif "state_dict" in checkpoint:  # BLAT
            sd = {k.replace("audio_encoder.", ""): v for k, v in checkpoint["state_dict"].items()
                  if "audio_encoder" in k}
        else:
            raise Exception("Unkown checkpoint format")
        own = self.state_dict()
        loaded = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        output_fn(f"Loading pre-trained model, with mismatched keys {[k for k in sd if k not in loaded]}\n")
        own.update(loaded)
        self.load_state_dict(own, strict=True)
        if self.freeze:
            for name, param in self.named_parameters():
                param.requires_grad = name not in loaded

    # ---- weight packing for the kernels (cached; invalidated by in-place updates / .to()) ----------
    def _pack(self, device, algo=None):
        algo = algo or self.conv_algo
        tensors = [self.bn0.weight, self.bn0.bias, self.bn0.running_mean, self.bn0.running_var]
        for b in range(6):
            blk = getattr(self, f"conv_block{b + 1}")
            for conv, bn in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2)):
                tensors += [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        # Keyed on the tensors themselves (address, torch version counter, and the per-tensor counter HIP optimisers
        # bump), NOT on the global parameter generation: an optimiser step on the GRU / decoder must not repack the
        # frozen Cnn14 - captured training graphs hold the addresses of this pack (and a repack costs ~0.3 s).
        mixed = algo == "f16x2" and self.f16x2_block6 == "bf16x3"
        key = tuple((t.data_ptr(), t._version, K._lib.tensor_generation(t)) for t in tensors) + (algo, mixed)
        if not isinstance(self._packed, dict):
            self._packed = {}
        hit = self._packed.get(algo)
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            pk = {"bn0": K.fold_bn(self.bn0.weight.float(), self.bn0.bias.float(), self.bn0.running_mean.float(),
                                   self.bn0.running_var.float(), self.bn0.eps), "convs": []}
            for b in range(6):
                blk = getattr(self, f"conv_block{b + 1}")
                for j, (conv, bn) in enumerate(((blk.conv1, blk.bn1), (blk.conv2, blk.bn2))):
                    w = conv.weight.detach().float()
                    if b == 0 and j == 0:
                        wp = w.reshape(64, 9).contiguous()
                    elif algo == "winograd":
                        wp = K.pack_conv_weight_winograd(w)
                    elif algo == "direct":
                        wp = K.pack_conv_weight(w)
                    elif algo == "bf16x3" or (algo in WINO and not wino1d_covers(w.shape[0])):
                        wp = K.pack_conv_weight_bf16x3_frag(w)
                    elif algo == "wino1d":
                        wp = K.pack_conv_weight_wino1d_frag(w)
                    elif algo == "wino43":   # both packs: small launches (single clips) take the K-sliced F(2,3) form
                        wp = (K.pack_conv_weight_wino1d_frag(w),
                              K.pack_conv_weight_wino43_frag(w) if wino43_covers(64 >> b, w.shape[0]) else None,
                              _LazyPack(conv.weight) if b >= 4 else None)   # blocks 5-6 (W = 4, 2): the skinny form
                    elif algo == "bf16x3_lds":
                        wp = K.pack_conv_weight_bf16x3(w)
                    elif algo == "f16x2" and mixed and b == 5:
                        wp, inv = K.pack_conv_weight_bf16x3_frag(w), None
                    elif algo == "f16x2":
                        wp, inv = K.pack_conv_weight_f16x2_frag(w)
                    else:
                        raise ValueError(f"unknown conv_algo {algo!r}")
                    sc, sh = K.fold_bn(bn.weight.float(), bn.bias.float(), bn.running_mean.float(),
                                       bn.running_var.float(), bn.eps)
                    if algo == "f16x2" and not (b == 0 and j == 0) and inv is not None:
                        sc = (sc * inv).contiguous()
                    pk["convs"].append((wp, sc, sh))
            pk["mixed"] = mixed
            if algo == "wino43":
                pk["b1c2_f43"] = K.pack_conv_weight_wino43_frag(self.conv_block1.conv2.weight.detach().float())
        self._packed[algo] = (key, pk)
        return pk

    def _buf(self, name, numel, device, dtype=torch.float32):
        b = self._bufs.get((name, dtype))
        if b is None or b.numel() < numel or b.device != device:
            b = torch.empty(numel, device=device, dtype=dtype)
            self._bufs[(name, dtype)] = b
        return b

    def capture_token(self, algo):
        """(identity, references) of everything a captured graph of this encoder's launches addresses: the packed weights
        of ``algo`` and the shared activation buffers.  A graph owner keeps the references for as long as the graph lives
        and re-captures when the identity changes (a larger batch shape re-allocates the buffers)."""
        hit = self._packed.get(algo)
        ident = (id(hit[1]) if hit is not None else None,) + tuple(sorted((k[0], str(k[1]), b.data_ptr())
                                                                           for k, b in self._bufs.items()))
        return ident, (hit, dict(self._bufs), self._tables)

    def geometry(self, n_samples):
        """Valid (H) and physical (Hp) row counts of the 6 resolution levels for an L-sample batch."""
        T = n_samples // self.hop_length + 1
        H = [T >> k for k in range(6)]
        if H[5] < 1:
            raise ValueError(f"clips of {n_samples} samples are shorter than one Cnn14 output frame")
        hp6 = (H[5] + 4) & ~3  # >= H6 + 1 and a multiple of 4 (a row quad of F(4,3) / a pooling pair never straddles clips)
        Hp = [hp6 << (5 - k) for k in range(6)]
        return T, H, Hp

    def logmel_front(self, wav):
        """The first kernel of ``encode`` on its own (log-mel + bn0 in the conv stack's layout), so that a caller can run it
        on another stream ahead of the convolutions (TransformerModel.forward_async); pass the result back as ``x0``."""
        dev = wav.device
        self._ensure_tables(dev)
        pk = self._pack(dev, self.effective_algo(None, False, None))
        Hp = self.geometry(wav.shape[1])[2]
        return K.logmel(wav, self._tables, pk["bn0"][0], pk["bn0"][1], rows_per_clip=Hp[0], channels_last=True)

    def _ensure_tables(self, dev):
        mkey = self.melspec_extractor.key()
        if self._tables is None or self._tables.window.device != dev or self._tables_key != mkey:
            self._tables = MelTables(self.sample_rate, self.n_fft, self.hop_length, self.f_min, self.f_max, 64,
                                     "slaney", "slaney", dev, window=self.melspec_extractor.spectrogram.window,
                                     fb=self.melspec_extractor.mel_scale.fb)
            self._tables_key = mkey

    def encode(self, wav, dropout=None, specaug=None, train=False, min_frames=None, algo=None, overflow=None,
               clip_frames=None, x0=None, block6_f23=False):
        """wav (B, L) on the ROCm device -> attn_emb (B, T // 32, 2048).

        ``dropout = (p, op_code, seed_dev_ptr)``: the train-mode forward of the frozen network, F.dropout(p) after
        every conv block (cnn_encoder.py:431-442); the mask of block b is the counter hash of csrc/train.hip with seed
        op_code + b (+ the device-side step seed) over the block's output buffer.
        ``specaug``: int32 device tensor (B, 4, 2) of (begin, length) stripes - 2 over time, 2 over mel - masked on the
        log-mel as the reference's SpecAugmentation does in train mode (cnn_encoder.py:423-425).
        ``train``: this is the forward of a training step (TrainEngine), with or without dropout.
        ``min_frames``: output frames of the shortest clip of the batch (``forward`` passes it).  The "f16x2" tier's
        logit error grows as clips get shorter (fewer frames for the decoder's attention to average the fp16 rounding
        over: 3e-4 at 10 s, 6e-4 at 3 s, up to 1.7e-3 at 1 s - DESIGN.md section 4), so a batch that contains a clip of
        fewer than ``f16x2_min_frames`` (10, = 3.2 s) frames runs on the split-bf16 tier (3e-5 at any length).
        ``algo``: conv tier of this call (default ``self.conv_algo``).  ``overflow``: int32 device word the fp16 tier ORs
        with 1 when an activation left the fp16 range (``forward`` returns it as ``f16_overflow``).
        ``clip_frames``: int32 device tensor (B,) of every clip's own ``attn_emb_len`` - ragged batches: the "wino1d" tier
        skips the conv rows that lie beyond what a clip's own length can bring to one of its output frames (the reference
        convolves the zero padding, collate_func.py:29-32); frames below ``attn_emb_len`` are bit-identical, frames at or
        beyond it - which no temporal encoder reads (model_util.py:10-27) - are then NOT the reference's values."""
        if wav.dim() != 2:
            raise ValueError("wav must be (batch, samples)")
        dev = wav.device
        self._ensure_tables(dev)
        # the "f16x2" tier keeps its activations in HBM as fp16; the train-mode forward (dropout on f32 block outputs,
        # parity pinned by tests/golden/g8_train.npz) stays on the split-bf16 tier
        algo = self.effective_algo(algo, train or dropout is not None, min_frames)
        pk = self._pack(dev, algo)
        B, L = wav.shape
        T, H, Hp = self.geometry(L)
        if x0 is None:   # (given: computed by logmel_front on another stream; bn0 is the same for every tier)
            x0 = K.logmel(wav, self._tables, pk["bn0"][0], pk["bn0"][1], rows_per_clip=Hp[0], channels_last=True)
        if specaug is not None:
            K.specaug_(x0, specaug, pk["bn0"][1], B, Hp[0], T)
        return self.conv_stack(x0, B, H, Hp, pk, algo, dropout, overflow=overflow,
                               clip_frames=clip_frames if algo in WINO else None, block6_f23=block6_f23)

    def effective_algo(self, algo=None, train=False, min_frames=None):
        """The conv tier a call runs on: the fp16-activation tier is left for the train-mode forward and for batches
        that contain a very short clip (see ``encode``)."""
        algo = algo or self.conv_algo
        if algo == "f16x2" and train:
            return "bf16x3"
        if algo == "f16x2" and min_frames is not None and min_frames < self.f16x2_min_frames:
            return "bf16x3"
        return algo

    def conv_stack(self, x0, B, H, Hp, pk, algo, dropout=None, blocks=None, overflow=None, clip_frames=None,
                   block6_f23=False):
        """The six conv blocks on a bn0-normalised log-mel x0 [B*Hp[0]][64] -> attn_emb (B, H[5], 2048).
        ``blocks``: a list that receives a float32 (B, C, H, W) copy of every pooled block output (tests)."""
        dev = x0.device
        act = torch.float16 if algo == "f16x2" else torch.float32
        full = self._buf("full", B * Hp[0] * 64 * 64, dev, act)      # conv1 outputs (largest: level 1)
        pooled = self._buf("pooled", B * Hp[1] * 32 * 64, dev, act)  # block outputs (largest: block 1)
        W = 64
        conv = conv_kernel(algo)
        if algo == "f16x2":
            import functools
            conv = functools.partial(conv, overflow=overflow)
        fuse1 = algo == "f16x2" and os.environ.get("AUDIOCAPTION_FUSE_BLOCK1", "1") != "0"
        # "wino43": conv block 1 in one kernel (conv1 computed into the F(4,3) staging of conv2) when the launch gives every
        # CU a persistent workgroup; single clips keep conv_first + the K-sliced F(2,3) form
        fuse1_w4 = algo == "wino43" and os.environ.get("AUDIOCAPTION_FUSE_BLOCK1", "1") != "0" and Hp[0] % 8 == 0 \
            and B * Hp[0] // 8 >= W43_MIN_WORKGROUPS and pk.get("b1c2_f43") is not None
        if algo in WINO and os.environ.get("AUDIOCAPTION_W1_SPLITK", "1") != "0":
            import functools   # single clips: layers of a few workgroups run K-sliced over a shared workspace
            conv = functools.partial(conv, splitk_buf=lambda n: self._buf("w1_splitk", n, dev))
            if algo == "wino43" and dropout is not None:
                conv = functools.partial(conv, cold=True)   # the training step: weights come from HBM every iteration

        def need(block, j):   # ragged batches: the rows of this layer a clip's own length can bring to an output frame
            return {"need": (clip_frames,) + rows_needed(block, j, quads=algo == "wino43" and ragged_exact())} \
                if clip_frames is not None and algo in WINO else {}

        mixed = algo == "f16x2" and pk.get("mixed", False)
        for b in range(6):
            cin, cout = CHANNELS[b], CHANNELS[b + 1]
            w1, s1, t1 = pk["convs"][2 * b]
            w2, s2, t2 = pk["convs"][2 * b + 1]
            if b == 5 and block6_f23 and isinstance(w1, tuple):
                w1, w2 = (w1[0], None), (w2[0], None)
            pool_out = pooled
            if mixed and b == 4:     # block 5 hands block 6 (split-bf16, f32 activations) an f32 pooled output
                pool_out = self._buf("pooled32", B * Hp[5] * 2 * CHANNELS[5], dev, torch.float32)
            if mixed and b == 5:
                conv = K.conv3x3_bn_relu_bf16x3_gw
                full = self._buf("full32", B * Hp[5] * 2 * CHANNELS[6], dev, torch.float32)
            if b == 0 and fuse1_w4:
                K.conv3x3_block1_wino43(x0, w1, s1, t1, pk["b1c2_f43"], s2, t2, pooled, B, Hp[0], H[0],
                                        dropout=(dropout[0], dropout[1], dropout[2]) if dropout is not None else None,
                                        conv1=BLOCK1_CONV1, **need(1, 2))
            elif b == 0 and fuse1:   # conv1 is computed inside conv2's kernel: its 64-channel output never reaches HBM
                K.conv3x3_block1_f16x2(x0, w1, s1, t1, w2, s2, t2, pooled, B, Hp[0], H[0], W, overflow=overflow)
            elif b == 0:
                K.conv3x3_first(x0, w1, s1, t1, full, B, Hp[0], H[0], W, overflow=overflow)
            else:
                conv(pooled, w1, s1, t1, full, B, Hp[b], H[b], W, cin, cout, 0, **need(b + 1, 1))
            if b < 5:
                fused_drop = dropout is not None and algo in WINO and not (b == 0 and fuse1)
                if not (b == 0 and (fuse1 or fuse1_w4)):
                    kw = need(b + 1, 2) if algo not in WINO or wino1d_covers(cout) else {}
                    if fused_drop:   # the block's F.dropout in the conv kernel's epilogue
                        kw = {"dropout": (dropout[0], dropout[1] + b, dropout[2])}
                    conv(full, w2, s2, t2, pool_out, B, Hp[b], H[b], W, cout, cout, 1, **kw)
                pooled = pool_out
                W //= 2
                if dropout is not None and not fused_drop:
                    K.dropout_(pooled, B * Hp[b + 1] * W * cout, dropout[0], dropout[1] + b, dropout[2])
                if blocks is not None:
                    blk = pooled[:B * Hp[b + 1] * W * cout].reshape(B, Hp[b + 1], W, cout)[:, :H[b + 1]]
                    blocks.append(blk.permute(0, 3, 1, 2).float().clone())
            else:
                attn = torch.empty(B, H[5], cout, device=dev, dtype=torch.float32)
                if dropout is None:
                    conv(full, w2, s2, t2, attn, B, Hp[b], H[b], W, cout, cout, 2, **need(6, 2))
                else:  # dropout sits between the last block and the mean over mel bins
                    last = self._buf("last", B * Hp[5] * W * cout, dev)
                    if algo in WINO:
                        conv(full, w2, s2, t2, last, B, Hp[b], H[b], W, cout, cout, 0,
                             dropout=(dropout[0], dropout[1] + b, dropout[2]))
                    else:
                        conv(full, w2, s2, t2, last, B, Hp[b], H[b], W, cout, cout, 0)
                        K.dropout_(last, B * Hp[5] * W * cout, dropout[0], dropout[1] + b, dropout[2])
                    K.rows_mean_w(last, attn, B, Hp[5], H[5], W, cout)
        return attn

    def forward(self, input_dict, skip_fc=False):
        if self.training:
            raise NotImplementedError(
                "Cnn14Encoder (HIP path): in train mode the (frozen) Cnn14 only runs inside the whole-model training "
                "step (audiocaption_amd.train.TrainEngine); SpecAugment and the backward through the convolutions "
                "are not built")
        wav = input_dict["wav"]
        feat_length = cnn14_feat_len(input_dict["wav_len"], self.hop_length, self.downsample_ratio)
        min_frames = int(feat_length.min())
        # ``conv_algo`` in the input dict overrides the tier for this call (the model re-runs a batch whose fp16
        # activations overflowed on the f32-activation tier)
        algo = self.effective_algo(input_dict.get("conv_algo"), False, min_frames)
        flag = torch.zeros(1, device=wav.device, dtype=torch.int32) if algo == "f16x2" else None
        # inside a composite encoder (skip_fc: CrnnEncoder / Cnn14TransformerEncoder mask by length) the rows a clip's own
        # length cannot bring to an output frame are not convolved; stand-alone, attn_emb is the reference's everywhere
        ragged = skip_fc and algo in WINO and os.environ.get("AUDIOCAPTION_SKIP_DEAD_ROWS", "1") != "0" \
            and int(feat_length.min()) < int(feat_length.max())
        frames = K.upload(feat_length, wav.device, torch.int32) if ragged else None
        # AUDIOCAPTION_RAGGED_EXACT=1: a batch of uneven lengths keeps block 6 on the F(2,3) kernel whether or not rows are
        # skipped (same kernels = same bits with AUDIOCAPTION_SKIP_DEAD_ROWS on and off; see _conv_wino43, ragged_exact)
        uneven = skip_fc and ragged_exact() and int(feat_length.min()) < int(feat_length.max())
        attn_emb = self.encode(wav, min_frames=min_frames, algo=algo, overflow=flag, clip_frames=frames,
                               x0=input_dict.get("_logmel"), block6_f23=uneven)
        out = {"attn_emb": attn_emb, "attn_emb_len": feat_length}
        if flag is not None:
            # non-zero: an activation exceeded the fp16 range (65504) and this result must not be used -
            # TransformerModel re-runs such a batch on the split-bf16 tier; stand-alone callers check it themselves
            out["f16_overflow"] = flag
        if not skip_fc:
            # Cnn14's own clip embedding (cnn_encoder.py:451-456); CrnnEncoder discards it.
            lens = K.upload(feat_length, wav.device, torch.int32)
            pooled = K.mean_with_lens(attn_emb, lens, add_max=True)
            out["fc_emb"] = K.linear(pooled, self.fc1.weight.float(), self.fc1.bias.float(), relu=True)
        return out


# The reference keeps its EfficientNet-B2 encoder in the same module (cnn_encoder.py:770-839); re-export ours so that
# ``captioning.models.cnn_encoder.EfficientNetB2`` resolves after ``compat.install()``.
from .effnet_encoder import EfficientNetB2  # noqa: E402,F401
