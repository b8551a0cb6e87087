"""EfficientNet-B2 audio encoder, MI355X path.  Plugin-compatible with the reference class ``EfficientNetB2``
(``captioning.models.cnn_encoder.EfficientNetB2`` cnn_encoder.py:770-839 == hf_wrapper.py:262-315): same constructor
keywords, ``forward(input_dict) -> {"fc_emb", "attn_emb", "attn_emb_len"}``, ``fc_emb_size`` 1408, and the
``state_dict()`` keys of ``backbone.eff_net.*`` exactly as ``efficientnet_pytorch`` names them
(eff_latent_encoder.py:263-290), so the published checkpoint loads.

The nn modules below only OWN the parameters.  The forward pass is csrc/logmel.hip (HTK mel, top_db clamp),
csrc/effnet.hip (stem, depthwise + squeeze sums, squeeze-excite gate, ``ac_pointwise_conv`` for the 1x1 convolutions
(BatchNorm folded into the weight rows; swish, the squeeze-excite gate and the residual in the GEMM's prologue /
epilogue), channels-last ``[clip][time][mel][C]``.  The backbone's arithmetic lives in the un-vendored
``efficientnet_pytorch==0.7.1``; ``oracle/effb2_path.py`` restates its published algorithm, and both are held to
``transformers.EfficientNetModel`` / ``transformers.audio_utils`` outputs (tests/golden/g10_logmel.npz, g11_effb2.npz).
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib
from . import kernels as K
from ._lib import check, ptr, stream
from .cnn_encoder import cnn14_feat_len
from .mel import MelSpectrogramBuffers, MelTables

# (repeats, kernel, stride, expand, in, out) of EfficientNet-B0; B2: width x1.1, depth x1.2, resolution 260
_B0 = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
       (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
BN_EPS, BN_MOM = 1e-3, 0.01


def round_filters(filters, width=1.1, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def same_pad(image, k, s):
    """Static "same" padding of efficientnet_pytorch's Conv2dStaticSamePadding for a construction-time size."""
    out = math.ceil(image / s)
    pad = max((out - 1) * s + (k - 1) + 1 - image, 0)
    return pad // 2, pad - pad // 2


class _Conv(nn.Module):
    """Weight (and optional bias) holder with Conv2d's parameter names."""

    def __init__(self, cin, cout, k, groups=1, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out")
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None


class MBConvBlock(nn.Module):

    def __init__(self, cin, cout, expand, k, stride, image):
        super().__init__()
        mid = cin * expand
        self.cin, self.cout, self.mid, self.expand, self.k, self.stride = cin, cout, mid, expand, k, stride
        self.pad = same_pad(image, k, stride)
        self.skip = stride == 1 and cin == cout
        if expand != 1:
            self._expand_conv = _Conv(cin, mid, 1)
            self._bn0 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        self._depthwise_conv = _Conv(mid, mid, k, groups=mid)
        self._bn1 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        se = max(1, int(cin * 0.25))
        self._se_reduce = _Conv(mid, se, 1, bias=True)
        self._se_expand = _Conv(se, mid, 1, bias=True)
        self._project_conv = _Conv(mid, cout, 1)
        self._bn2 = nn.BatchNorm2d(cout, momentum=BN_MOM, eps=BN_EPS)


class EfficientNet(nn.Module):
    """Parameter layout of ``efficientnet_pytorch.EfficientNet`` (B2, 1 input channel, include_top False)."""

    def __init__(self):
        super().__init__()
        image = 260
        self._conv_stem = _Conv(1, round_filters(32), 3)
        self.stem_pad = same_pad(image, 3, 2)
        self._bn0 = nn.BatchNorm2d(round_filters(32), momentum=BN_MOM, eps=BN_EPS)
        image = math.ceil(image / 2)
        blocks = []
        for (r, k, s, e, i, o) in _B0:
            cin, cout = round_filters(i), round_filters(o)
            for j in range(int(math.ceil(1.2 * r))):
                stride = s if j == 0 else 1
                blocks.append(MBConvBlock(cin if j == 0 else cout, cout, e, k, stride, image))
                image = math.ceil(image / stride)
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = _Conv(blocks[-1].cout, round_filters(1280), 1)
        self._bn1 = nn.BatchNorm2d(round_filters(1280), momentum=BN_MOM, eps=BN_EPS)


class _EffiNet(nn.Module):

    def __init__(self):
        super().__init__()
        self.eff_net = EfficientNet()


# matrix-bound 1x1 convolutions: "pw" (default) = ac_pw_gemm_bf16x3 (weights pre-split in MFMA fragment order, activations
# split once per 256 output channels), "bf16x3" = ac_gemm_bf16x3 (both operands split per tile), "f32" = exact f32 MFMA
GEMM_ALGO = os.environ.get("AUDIOCAPTION_EFFB2_GEMM", "pw")


# MBConv expand -> depthwise -> squeeze sums as ONE kernel (csrc/effnet_fused.hip) for the blocks with at least
# FUSE_MIN_ROWS positions in the batch.  Measured per block at 128 clips x 10 s (tools/effb2_block_bench.py): the four
# blocks at 501 x 32 and 250 x 16 positions per clip run 1.25-1.45x faster fused (the 6x expanded tensor is neither
# written nor read back); from 125 x 8 positions down the fused kernel's LDS-resident band is too small to keep the
# vector ALUs busy and the two-kernel chain is as fast or faster.  "0" keeps the chain everywhere.
FUSE_EXPAND_DW = os.environ.get("AUDIOCAPTION_EFFB2_FUSE", "1") != "0"
FUSE_MIN_ROWS = int(os.environ.get("AUDIOCAPTION_EFFB2_FUSE_MIN_ROWS", "400000"))


# squeeze-excite gate: "kernel" = ac_effnet_se_gate_t (one launch per block), "gemm" = two small ac_gemm launches
SE_GATE = os.environ.get("AUDIOCAPTION_EFFB2_SE", "kernel")


def _fold(bn):
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return scale, bn.bias.detach().float() - bn.running_mean.detach().float() * scale


class EfficientNetB2(nn.Module):

    def __init__(self, n_mels=64, win_length=32, hop_length=10, f_min=0, freeze=False):
        super().__init__()
        if n_mels != 64 or win_length != 32 or hop_length != 10:
            raise NotImplementedError("EfficientNetB2 (HIP path): n_mels 64, 32 ms windows, 10 ms hop only "
                                      "(the configuration every reference config uses)")
        self.sample_rate = 16000
        self.n_fft = win_length * self.sample_rate // 1000
        self.hop_length = 10 * self.sample_rate // 1000
        self.f_min = float(f_min)
        self.top_db = 120.0
        self.melspec_extractor = MelSpectrogramBuffers(self.sample_rate, self.n_fft, self.f_min, self.sample_rate // 2, 64,
                                                       None, "htk")
        self.backbone = _EffiNet()
        self.fc_emb_size = self.backbone.eff_net._conv_head.weight.shape[0]
        self.downsample_ratio = 32
        if freeze:
            for p in self.parameters():
                p.requires_grad = False
        self._packed, self._packed_key, self._tables, self._bufs = None, None, None, {}
        self._buf_gen = 0
        self._tables_key = None
        self._graphs = {}   # (device, wav shape, packed-weights key) -> captured encoder

    # ---- weight packing (cached) ---------------------------------------------------------------------------
    def _pack(self):
        net = self.backbone.eff_net
        key = tuple((t.data_ptr(), t._version, _lib.tensor_generation(t))
                    for t in list(net.parameters()) + list(net.buffers()))
        if self._packed is not None and key == self._packed_key:
            return self._packed

        def pointwise(conv, bn):
            sc, sh = _fold(bn)
            w = (conv.weight.detach().float().reshape(conv.weight.shape[0], -1) * sc[:, None]).contiguous()   # BN folded
            wfrag = None
            if w.is_cuda and GEMM_ALGO == "pw":
                # split into bf16 hi + lo and laid out in MFMA fragment order once (csrc/pw_gemm.hip)
                lib = _lib.load()
                n, k = w.shape
                wfrag = torch.empty(lib.ac_pw_gemm_packed_bytes(n, k), device=w.device, dtype=torch.uint8)
                check(lib.ac_pw_gemm_pack(ptr(w), ptr(wfrag), n, k, stream()), "ac_pw_gemm_pack")
            return w, sh.contiguous(), wfrag

        with torch.no_grad():
            sc, sh = _fold(net._bn0)
            pk = {"stem": (net._conv_stem.weight.detach().float().reshape(-1, 9).contiguous(), sc.contiguous(),
                           sh.contiguous()), "blocks": []}
            for blk in net._blocks:
                d = {}
                if blk.expand != 1:
                    d["expand"] = pointwise(blk._expand_conv, blk._bn0)
                sc, sh = _fold(blk._bn1)
                # [C][1][k mel][k time] -> [k time][k mel][C]
                d["dw"] = (blk._depthwise_conv.weight.detach().float()[:, 0].permute(2, 1, 0).contiguous(),
                           sc.contiguous(), sh.contiguous())
                d["se"] = (blk._se_reduce.weight.detach().float().reshape(blk._se_reduce.weight.shape[0], -1).contiguous(),
                           blk._se_reduce.bias.detach().float().contiguous(),
                           blk._se_expand.weight.detach().float().reshape(blk.mid, -1).contiguous(),
                           blk._se_expand.bias.detach().float().contiguous(),
                           blk._se_expand.weight.detach().float().reshape(blk.mid, -1).t().contiguous())   # [S][C]
                d["project"] = pointwise(blk._project_conv, blk._bn2)
                pk["blocks"].append(d)
            pk["head"] = pointwise(net._conv_head, net._bn1)
        self._packed, self._packed_key = pk, key
        return pk

    def _buf(self, name, numel, device):
        b = self._bufs.get(name)
        if b is None or b.numel() < numel or b.device != device:
            b = torch.empty(numel, device=device, dtype=torch.float32)
            self._bufs[name] = b
            self._buf_gen += 1      # captured graphs hold the old address: they re-capture (``_encode_graph``)
        return b

    @staticmethod
    def _gemm(x, w, bias, y, M, N, Kd, act=0, beta=0.0, a_scale=None, a_rows=0, wfrag=None):
        lib = _lib.load()
        # two kernels for the same contract (tools/pointwise_bench.py, tools/pointwise_vs_pw.py): the full-resolution stages
        # (>= 0.4 M rows against a few KB of weights) are HBM-bound -> the streaming kernel that reads each activation once
        # (at 129 k rows the split-bf16 kernel is already ahead: 48 -> 288 expand 79 -> 64 us); everything later
        # -> the LDS-tiled x W^T GEMM, which shares each weight tile between 64 or 128 rows
        if M >= 400000 or (M >= 100000 and Kd <= 32) or wfrag is None and M >= 100000 and Kd <= 64:
            check(lib.ac_pointwise_conv(ptr(x), ptr(w), ptr(bias), ptr(y), M, N, Kd, act, beta, ptr(a_scale), a_rows,
                                        stream()), "ac_pointwise_conv")
        elif wfrag is not None and GEMM_ALGO == "pw":
            # matrix-bound layers: activation-stationary split-bf16 kernel over weights pre-split in fragment order
            check(lib.ac_pw_gemm_bf16x3(ptr(x), ptr(wfrag), ptr(bias), ptr(y), M, N, Kd, act, beta, ptr(a_scale), a_rows,
                                        stream()), "ac_pw_gemm_bf16x3")
        elif GEMM_ALGO in ("bf16x3", "pw"):
            # the late 1x1 convolutions are matrix-bound in f32 (65-90 TFLOP/s on the exact-f32 MFMA): split-bf16
            # operands (2^-16), f32 accumulation; small products are forwarded to the exact-f32 kernels by the library
            check(lib.ac_gemm_bf16x3(ptr(x), Kd, 1, ptr(w), 1, Kd, ptr(y), N, M, N, Kd, ptr(bias), act, beta, 1, 0.0, 0, None,
                                     0, ptr(a_scale), a_rows, stream()), "ac_gemm_bf16x3")
        else:
            check(lib.ac_gemm(ptr(x), Kd, 1, ptr(w), 1, Kd, ptr(y), N, M, N, Kd, ptr(bias), act, beta, 1, 0.0, 0, None, 0,
                              ptr(a_scale), a_rows, stream()), "ac_gemm")

    def logmel(self, wav):
        """wav (B, L) -> log-mel dB [B][T][64] (time-major), clamped at (batch max - 120 dB) like AmplitudeToDB."""
        dev = wav.device
        mkey = self.melspec_extractor.key()
        if self._tables is None or self._tables.window.device != dev or self._tables_key != mkey:
            self._tables = MelTables(self.sample_rate, self.n_fft, self.hop_length, self.f_min, self.sample_rate // 2, 64,
                                     None, "htk", dev, window=self.melspec_extractor.spectrogram.window,
                                     fb=self.melspec_extractor.mel_scale.fb)
            self._tables_key = mkey
        x = K.logmel(wav, self._tables, channels_last=True)      # (B*T, 64)
        scratch = self._buf("maxscratch", 1024, dev)
        check(_lib.load().ac_top_db_clamp(ptr(x), x.numel(), self.top_db, ptr(scratch), 1024, stream()),
              "ac_top_db_clamp")
        return x

    def features(self, x, B, T, F=64):
        """x: log-mel [B][T][F] -> attn_emb (B, T', 1408)."""
        lib = _lib.load()
        dev = x.device
        pk = self._pack()
        net = self.backbone.eff_net
        s = stream()
        w, sc, sh = pk["stem"]
        pb, pa = net.stem_pad
        To, Fo = (T + pb + pa - 3) // 2 + 1, (F + pb + pa - 3) // 2 + 1
        C = w.shape[0]
        # three rotating activation buffers sized for the largest tensor of the chain (block 2's expanded input)
        big = B * To * Fo * max(blk.mid for blk in net._blocks[:3])
        cur = self._buf("act_a", big, dev)
        check(lib.ac_effnet_stem(ptr(x), ptr(w), ptr(sc), ptr(sh), ptr(cur), B, T, F, C, pb, pa, s), "ac_effnet_stem")
        T, F = To, Fo
        mid_buf, dw_buf = self._buf("act_b", big, dev), self._buf("act_c", big, dev)
        nxt = self._buf("act_d", big, dev)
        # squeeze sums of all 23 blocks in one buffer, cleared by ONE fill per forward (the depthwise kernels accumulate
        # into their block's slice with atomics)
        pool_all = self._buf("se_pool", B * sum(blk.mid for blk in net._blocks), dev)
        pool_all.zero_()
        pool_off = 0
        gate = self._buf("se_gate", B * 2112, dev)
        sq_buf = self._buf("se_squeezed", B * 128, dev)
        for blk, d in zip(net._blocks, pk["blocks"]):
            rows = B * T * F
            xin = cur
            wd, sc, sh = d["dw"]
            pb, pa = blk.pad
            To, Fo = (T + pb + pa - blk.k) // blk.stride + 1, (F + pb + pa - blk.k) // blk.stride + 1
            pool = pool_all[pool_off:pool_off + B * blk.mid]
            pool_off += B * blk.mid
            fused = False
            if blk.expand != 1 and FUSE_EXPAND_DW and rows >= FUSE_MIN_ROWS:
                # expand -> depthwise -> squeeze sums in one kernel: the expanded tensor (6x the block input) stays in LDS
                w, b, _ = d["expand"]
                rc = lib.ac_effnet_expand_depthwise(ptr(xin), ptr(w), ptr(b), ptr(wd), ptr(sc), ptr(sh), ptr(dw_buf),
                                                    ptr(pool), 1.0 / (To * Fo), B, T, F, blk.cin, blk.mid, blk.k,
                                                    blk.stride, pb, pa, s)
                if rc != _lib.AC_ERR_ARG:      # AC_ERR_ARG: one row band does not fit the LDS budget -> two-kernel chain
                    check(rc, "ac_effnet_expand_depthwise")
                    fused = True
            if not fused:
                if blk.expand != 1:
                    w, b, wf = d["expand"]
                    self._gemm(xin, w, b, mid_buf, rows, blk.mid, blk.cin, act=2, wfrag=wf)
                    xmid = mid_buf
                else:
                    xmid = xin
                check(lib.ac_effnet_depthwise(ptr(xmid), ptr(wd), ptr(sc), ptr(sh), ptr(dw_buf), ptr(pool), 1.0 / (To * Fo),
                                              B, T, F, blk.mid, blk.k, blk.stride, pb, pa, s), "ac_effnet_depthwise")
            # squeeze-excite gate for all clips at once: two small GEMMs (swish, then sigmoid, in the epilogues)
            w1, b1, w2, b2, w2t = d["se"]
            Sq = w1.shape[0]
            if SE_GATE == "kernel":
                # one launch, a workgroup per clip (pool already holds the means: pool_scale above)
                check(lib.ac_effnet_se_gate_t(ptr(pool), 1.0, ptr(w1), ptr(b1), ptr(w2t), ptr(b2), ptr(gate), B, blk.mid,
                                              Sq, s), "ac_effnet_se_gate_t")
            else:
                check(lib.ac_gemm(ptr(pool), blk.mid, 1, ptr(w1), 1, blk.mid, ptr(sq_buf), Sq, B, Sq, blk.mid, ptr(b1), 2,
                                  0.0, 1, 0.0, 0, None, 0, None, 0, s), "ac_gemm(se reduce)")
                check(lib.ac_gemm(ptr(sq_buf), Sq, 1, ptr(w2), 1, Sq, ptr(gate), blk.mid, B, blk.mid, Sq, ptr(b2), 3, 0.0,
                                  1, 0.0, 0, None, 0, None, 0, s), "ac_gemm(se expand)")
            w, b, wf = d["project"]
            rows_o = B * To * Fo
            if blk.skip:
                # x <- x + project(gate * dw): accumulate in place on the block input
                self._gemm(dw_buf, w, b, xin, rows_o, blk.cout, blk.mid, beta=1.0, a_scale=gate, a_rows=To * Fo, wfrag=wf)
            else:
                self._gemm(dw_buf, w, b, nxt, rows_o, blk.cout, blk.mid, a_scale=gate, a_rows=To * Fo, wfrag=wf)
                cur, nxt = nxt, cur
            T, F = To, Fo
        w, b, wf = pk["head"]
        rows = B * T * F
        Ch = w.shape[0]
        self._gemm(cur, w, b, mid_buf, rows, Ch, w.shape[1], act=2, wfrag=wf)
        attn = torch.empty(B, T, Ch, device=dev, dtype=torch.float32)
        K.rows_mean_w(mid_buf, attn, B, T, T, F, Ch)            # mean over mel: 'b c f t -> b t c'
        return attn

    def _encode(self, wav):
        B, L = wav.shape
        return self.features(self.logmel(wav), B, L // self.hop_length + 1)

    def _encode_graph(self, wav):
        """log-mel + backbone replayed from a HIP graph.  The ~170 launches of this encoder cost the host 8.4 ms per
        batch - more than the device needs - so a shape that comes back (second use) is captured once over a static
        input buffer and replayed; weights repacked by a checkpoint load or an in-place update start a new graph."""
        self._pack()
        key = (wav.device, tuple(wav.shape), self._packed_key)
        st = self._graphs.pop(key, None)
        if st is None:
            st = {"uses": 0, "graph": None}
            while len(self._graphs) >= 4:       # a handful of shapes (fixed-length batches); ragged streams stay eager
                self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = st
        st["uses"] += 1
        if st["uses"] < 2:
            return self._encode(wav)
        if st["graph"] is not None and st["buf_gen"] != self._buf_gen:
            # a larger shape re-allocated the shared activation buffers since this graph was captured: its launches
            # point at freed memory.  Drop it and capture again over the current buffers.
            st["graph"] = None
        if st["graph"] is None:
            st["wav"] = wav.clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st["attn"] = self._encode(st["wav"])
            st["graph"], st["buf_gen"] = graph, self._buf_gen
            st["hold"] = (dict(self._bufs), self._packed)   # what the graph's launches address stays alive with it
        st["wav"].copy_(wav)
        st["graph"].replay()
        return st["attn"].clone()

    def forward(self, input_dict):
        if self.training:
            raise NotImplementedError("EfficientNetB2 (HIP path): inference only; training this encoder "
                                      "(BatchNorm statistics, drop-connect, backward) is not built")
        wav = input_dict["wav"]
        if not wav.is_cuda:
            raise _lib.HipLibraryError("the HIP path needs tensors on a ROCm device; there is no CPU fallback")
        wav = K.f32c(wav)
        B, L = wav.shape
        attn_emb = self._encode_graph(wav) if os.environ.get("AUDIOCAPTION_ENCODER_GRAPH", "1") != "0" \
            else self._encode(wav)
        feat_length = cnn14_feat_len(input_dict["wav_len"], self.hop_length, self.downsample_ratio)
        lens = K.upload(feat_length, wav.device, torch.int32)
        fc_emb = K.mean_with_lens(attn_emb, lens)
        return {"fc_emb": fc_emb, "attn_emb": attn_emb, "attn_emb_len": feat_length}
