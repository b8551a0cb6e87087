"""Encoder-decoder captioning model, MI355X path.  Plugin-compatible with the reference classes
``captioning.models.base.CaptionModel`` (base.py:24-361) and
``captioning.models.transformer_model.TransformerModel`` (transformer_model.py:11-86):

* ``model(input_dict) -> dict`` with ``seq`` (int64, CPU, (B, max_len)), ``logit``, ``sampled_logprob``,
  ``embed`` plus the encoder outputs; ``input_dict`` keys as ``Runner._forward`` builds them (run.py:29-51);
* class-level ``set_index(start_idx, end_idx, pad_idx)`` and the defaults pad 0 / start 1 / end 2 /
  max_length 20 (base.py:10-21);
* the decoder must be a ``TransformerDecoder`` (base.py:42-46).

Differences in schedule, not in results: greedy search runs entirely on the device (one C call, no
per-step host synchronisation, KV cache); beam search batches ALL clips and beams into one decoder
call per step (the reference loops over clips, base.py:266) with the reference's per-clip bookkeeping
(done beams, -1000 trick, length-normalised score, early exit) done on the host from one small
device->host copy per step.
"""
import ctypes

import numpy as np
import os
import weakref

import torch
import torch.nn as nn

from . import _lib
from . import kernels as K
from ._lib import check, ptr, stream
from .transformer_decoder import TransformerDecoder


class CaptionMetaMixin:
    pad_idx = 0
    start_idx = 1
    end_idx = 2
    max_length = 20

    @classmethod
    def set_index(cls, start_idx, end_idx, pad_idx):
        cls.start_idx = start_idx
        cls.end_idx = end_idx
        cls.pad_idx = pad_idx


class CaptionModel(nn.Module, CaptionMetaMixin):

    compatible_decoders = (TransformerDecoder,)

    def __init__(self, encoder, decoder, **kwargs):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.vocab_size = decoder.vocab_size
        self.train_forward_keys = ["cap", "cap_len", "ss_ratio"]
        self.inference_forward_keys = ["sample_method", "max_length", "temp"]
        if kwargs.get("freeze_encoder", False):
            for param in self.encoder.parameters():
                param.requires_grad = False
        names = [c.__name__ for c in self.compatible_decoders]
        assert isinstance(self.decoder, self.compatible_decoders), \
            f"{self.decoder.__class__.__name__} is incompatible with {self.__class__.__name__}, " \
            f"please use decoder in {names} "

    def forward(self, input_dict):
        if input_dict["mode"] == "train":
            # the whole training forward (frozen Cnn14 with dropout, GRU, scheduled-sampling decoder) is one engine
            # call; ``logit`` comes back attached to autograd by a single bridge node (audiocaption_amd/train.py)
            from .train import train_forward
            return train_forward(self, input_dict)
        encoder_output_dict = self.encoder(input_dict)
        output = self.forward_decoder(input_dict, encoder_output_dict)
        flags = _device_flags(output)
        if flags is not None:
            redo = self._check_flags(flags.cpu(), input_dict)
            if redo is not None:
                return redo
        return output

    def _check_flags(self, host_flags, input_dict):
        """host_flags = [fp16 overflow of the conv tier, split-GRU partner timeout] read back with the results."""
        if int(host_flags[1]) != 0:
            algos = []
            for m in self.encoder.modules():   # the word is sticky on the device: clear it so that the NEXT batch is judged on its own
                ws = getattr(m, "_split_ws", None)
                if ws is not None:
                    ws.view(torch.int32)[:1].zero_()
                if getattr(m, "gru_algo", None) == "split":
                    algos.append(m)
            # The split kernel needs its four workgroups per (clip, direction) co-resident; on a GPU shared with another
            # process (or over-subscribed by other streams) a partner may never start.  Once that has happened the
            # encoder moves to the single-workgroup kernel for good and the batch is run again (AUDIOCAPTION_GRU_FALLBACK=0:
            # raise instead).
            if algos and os.environ.get("AUDIOCAPTION_GRU_FALLBACK", "1") != "0" and not input_dict.get("_gru_retry"):
                import warnings
                for m in algos:
                    m.gru_algo = "single"
                warnings.warn("split GRU kernel: a workgroup's partner never started (GPU shared with another process?); this "
                              "encoder now uses the single-workgroup kernel (AUDIOCAPTION_GRU_ALGO=single)")
                return self.forward(dict(input_dict, _gru_retry=True))
            raise _lib.HipLibraryError("split GRU kernel: a workgroup's partner never started (GPU shared with another "
                                       "process?); set AUDIOCAPTION_GRU_ALGO=single")
        if int(host_flags[0]) != 0:
            return self._rerun_wide(input_dict)
        return None

    def _rerun_wide(self, input_dict):
        """An activation of the fp16-activation conv tier left the fp16 range (the kernels raise ``f16_overflow``
        instead of storing inf): the batch is run again on the split-bf16 tier, whose activations are f32."""
        if input_dict.get("conv_algo") == "bf16x3":
            raise _lib.HipLibraryError("f16_overflow raised by a tier without fp16 activations")
        return self.forward(dict(input_dict, conv_algo="bf16x3"))

    def forward_decoder(self, input_dict, encoder_output_dict):
        if input_dict["mode"] == "train":
            raise NotImplementedError("mode='train' is handled by CaptionModel.forward as one fused step")
        elif input_dict["mode"] == "inference":
            output = self.inference_forward(self._inference_dict(input_dict, encoder_output_dict))
        else:
            raise Exception("mode should be either 'train' or 'inference'")
        output.update(encoder_output_dict)
        return output

    def _inference_dict(self, input_dict, encoder_output_dict):
        """The decoding request with the reference's defaults (base.py:89-101) merged with the encoder's outputs."""
        forward_dict = {"mode": "inference"}
        default_args = {"sample_method": "greedy", "max_length": self.max_length, "temp": 1.0}
        for key in self.inference_forward_keys:
            forward_dict[key] = input_dict.get(key, default_args[key])
        if forward_dict["sample_method"] == "beam":
            forward_dict["beam_size"] = input_dict.get("beam_size", 3)
            forward_dict["n_best"] = input_dict.get("n_best", False)
            forward_dict["n_best_size"] = input_dict.get("n_best_size", forward_dict["beam_size"])
        forward_dict.update(encoder_output_dict)
        return forward_dict

    def inference_forward(self, input_dict):
        method = input_dict["sample_method"]
        if method == "beam":
            return self.beam_search(input_dict)
        if method == "greedy":
            return self.greedy_search(input_dict)
        raise NotImplementedError(
            f"sample_method={method!r}: only 'greedy' and 'beam' are on the accelerated path "
            "(dbs / gumbel / top-k / top-p sampling are out of scope, SURVEY.md §2.1 row 7)")


def _device_flags(enc):
    """The encoder's device-side status words as one int32[2] tensor (None when the encoder raises none)."""
    a, b = enc.get("f16_overflow"), enc.get("gru_error")
    if a is None and b is None:
        return None
    dev = (a if a is not None else b).device
    z = torch.zeros(1, device=dev, dtype=torch.int32)
    return torch.cat([a.view(1).to(torch.int32) if a is not None else z, b.view(1).to(torch.int32) if b is not None else z])


def _masked_stream(dev, first_cu, n_cus):
    """A HIP stream restricted to CU-mask bits first_cu .. first_cu + n_cus - 1 (``ac_stream_create_cu_mask``), wrapped for
    torch.  The stream lives as long as the process (a handful per model; never destroyed under work in flight)."""
    import ctypes
    from . import _lib
    out = ctypes.c_void_p()
    with torch.cuda.device(dev):
        check(_lib.load().ac_stream_create_cu_mask(int(first_cu), int(n_cus), ctypes.addressof(out)), "ac_stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value, device=dev)


def _cu_split():
    """AUDIOCAPTION_DECODE_CUS="n" or "n,exclusive": the greedy decode chains of the throughput mode run on a stream
    restricted to the first n CU-mask bits; with ",exclusive" the encoder stream is restricted to the others.  0 = off."""
    v = os.environ.get("AUDIOCAPTION_DECODE_CUS", "0").split(",")
    return int(v[0] or 0), len(v) > 1 and v[1].strip().lower().startswith("ex")


def decode_group():
    """Submissions ``forward_async`` decodes as ONE greedy chain at most (AUDIOCAPTION_DECODE_GROUP, default 4).  Beside a
    running encoder every launch of a chain waits for compute units the conv workgroups give back, so a chain advances at a
    quarter of its stand-alone speed and costs the encoder stream CU time per LAUNCH, not per row: one chain per 2 / 3 / 4 / 5
    batches of 64 clips = 13.2 / 13.9 / 14.3 / 14.2 k clips/s over 40 steps (round 4, wino43 tier)."""
    return max(1, int(os.environ.get("AUDIOCAPTION_DECODE_GROUP", "4")))


def _make_streams(dev):
    """(encoder stream, decode stream, chain stream) of the throughput mode.  AUDIOCAPTION_STREAM_PRIORITIES="enc,dec" sets
    their HIP priorities (lower = more urgent; default 0,0).  The chain stream (greedy decode chains) is the decode stream
    unless AUDIOCAPTION_DECODE_CUS restricts it to a few compute units (see ``_cu_split``)."""
    pe, pd = (int(v) for v in os.environ.get("AUDIOCAPTION_STREAM_PRIORITIES", "0,0").split(","))
    n_dec, exclusive = _cu_split()
    enc_s, dec_s = torch.cuda.Stream(dev, priority=pe), torch.cuda.Stream(dev, priority=pd)
    chain_s = dec_s
    if n_dec > 0:
        total = torch.cuda.get_device_properties(dev).multi_processor_count
        chain_s = _masked_stream(dev, 0, n_dec)
        if exclusive:
            enc_s = _masked_stream(dev, n_dec, total - n_dec)
    return enc_s, dec_s, chain_s


class PendingCaption:
    """Handle of a batch submitted with ``TransformerModel.forward_async``: ``result()`` blocks until the
    batch's token ids have reached the host and returns the same dict ``model(input_dict)`` would."""

    def __init__(self, model=None):
        self._model = model
        self._done = self._seq = self._lp = self._out = self._release = self._result = None
        self._flag = self._input = self._cnt = None

    def _fill(self, done_event, host_seq, host_logprob, output, release, host_flag=None, host_cnt=None):
        self._done, self._seq, self._lp, self._out, self._release = done_event, host_seq, host_logprob, output, release
        self._flag = host_flag
        self._cnt = host_cnt   # shared chain: this batch's unfinished counts, to blank the steps it would not have run

    def result(self):
        if getattr(self, "_lazy", None) is not None:   # beam search: the host-driven loop runs now, on the decode stream
            self._model._run_lazy(self)
        if self._result is not None:    # asked again: the staging buffers have gone back to the pool, return the copy
            return dict(self._result)
        if self._done is None:          # still waiting for a partner batch (pair decode): decode it on its own now
            self._model._flush_held()
        self._done.synchronize()
        out = dict(self._out)
        out["seq"] = self._seq.clone()
        out["sampled_logprob"] = self._lp.clone()
        if getattr(self, "_cnt", None) is not None:
            # A chain shared with other batches runs until ALL of them have finished; the reference stops a batch when its own
            # rows have (base.py:167) and leaves the later columns at their initial 0: step t ran iff rows were unfinished after t - 1
            ran = torch.ones(self._cnt.shape[0], dtype=torch.bool)
            ran[1:] = self._cnt[:-1] > 0
            out["sampled_logprob"][:, ~ran] = 0
        flags = self._flag.clone() if self._flag is not None else None
        if self._release is not None:  # hand the pinned staging buffers back to the pool
            self._release()
            self._release = None
        if flags is not None:          # fp16 range exceeded in the conv tier: blocking re-run on f32 activations
            redo = self._model._check_flags(flags, self._input)
            if redo is not None:
                out = redo
        self._result = out
        self._seq = self._lp = self._cnt = None
        return dict(out)


class TransformerModel(CaptionModel):

    def __init__(self, encoder, decoder, **kwargs):
        super().__init__(encoder, decoder, **kwargs)
        self._streams = None
        self._pinned = {}
        self._held = None   # forward_async: submitted batches whose decode waits for partner batches (a list)

    # ---- throughput mode: encoder of batch i+1 overlaps the (latency-bound) decode of batch i ---------
    def forward_async(self, input_dict, pair=None):
        """Submit one greedy-decoding batch without waiting for it.  The encoder runs on one HIP stream, the
        decoder (a chain of ~370 tiny latency-bound kernels) on another, so consecutive submissions overlap:
        the decode of batch i fills the gaps of the matrix-bound encoder of batch i+1.  Results are identical
        to ``model(input_dict)``; only the schedule differs.  Returns a ``PendingCaption``.

        ``pair`` (default: AUDIOCAPTION_PAIR_DECODE = "auto"): up to AUDIOCAPTION_DECODE_GROUP (4) consecutive submissions of
        the same shape can be decoded as ONE chain (rows are independent: same tokens, same logits; the chain is latency-bound, so 128 rows cost what 64
        do).  Beside a running encoder a chain advances at a quarter of its stand-alone speed (its ~200 dependent launches
        each wait for workgroup slots: 8.5 ms instead of 1.9 ms per batch against an encoder of 6.3 ms,
        tools/stream_timeline.py), so under load the CHAIN bounds the step - one chain per two or three batches hands the
        bound back to the encoder: 6.78 -> 6.37 ms per step.  "auto" holds a batch for its partner only while the decode stream is
        still busy with earlier chains (it could not have started anyway); an idle decode stream decodes on submission, so
        single requests are not delayed.  True / "1" always waits for a full group, False / "0" never.  A batch without a
        partner is decoded on its own as soon as its ``result()`` is asked for."""
        method = input_dict.get("sample_method", "greedy")
        if input_dict.get("mode") != "inference" or method not in ("greedy", "beam"):
            raise NotImplementedError("forward_async: greedy or beam inference only; use model(input_dict) otherwise")
        if method == "beam":
            return self._forward_async_beam(input_dict)
        if pair is None:
            pair = os.environ.get("AUDIOCAPTION_PAIR_DECODE", "auto")
        pair = {True: "1", False: "0"}.get(pair, pair)
        if pair not in ("0", "1", "auto"):
            raise ValueError(f"AUDIOCAPTION_PAIR_DECODE={pair!r}: '0', '1' or 'auto'")
        dev = input_dict["wav"].device
        if self._streams is None or self._streams[0].device != dev:
            self._streams = _make_streams(dev)
        enc_s, dec_s, chain_s = self._streams
        cur = torch.cuda.current_stream(dev)
        enc_s.wait_stream(cur)  # inputs produced on the caller's stream
        # A composite encoder (CrnnEncoder) runs in two halves: the convolutions on the encoder stream, the GRU (six launches of
        # latency-bound work: 0.4 ms in which the matrix cores idle) at the head of the decode chain on the decode stream, under
        # the next batch's convolutions.  Measured: 6.20 -> 6.00 ms per step (10.3 k -> 10.7 k clips/s).  It was SLOWER (6.05 vs
        # 5.83 ms) while every batch still had its own decode chain and the decode stream was the bottleneck; the split-GRU
        # kernel's workgroup groups form by start order, so waiting for slots beside conv workgroups cannot deadlock
        # (csrc/gru.hip).  AUDIOCAPTION_GRU_STREAM=encoder keeps the whole encoder on one stream.
        split_enc = hasattr(self.encoder, "forward_front") and os.environ.get("AUDIOCAPTION_GRU_STREAM", "decode") == "decode"
        # The log-mel kernel (FFT on the vector ALUs and LDS: nothing the matrix-bound convolutions wait for) runs on a third
        # stream, ahead of its batch's convolutions and under the previous batch's: 6.02 -> 5.94 ms per step, steady state
        # 10.8 k -> 11.05 k clips/s.  (conv1 of block 1 beside it, into its own buffers: no gain - it is bound by 1 GB of HBM
        # writes.)  AUDIOCAPTION_LOGMEL_STREAM=encoder keeps it on the encoder stream.
        submitted = input_dict   # what the caller passed (kept for a possible re-run)
        cnn = getattr(self.encoder, "cnn", None)
        if (split_enc and hasattr(cnn, "logmel_front") and "_logmel" not in input_dict
                and os.environ.get("AUDIOCAPTION_LOGMEL_STREAM", "front") == "front"):
            if getattr(self, "_pre_stream", None) is None or self._pre_stream.device != dev:
                self._pre_stream = torch.cuda.Stream(dev)
            pre_s = self._pre_stream
            pre_s.wait_stream(cur)
            with torch.cuda.stream(pre_s):
                x0 = cnn.logmel_front(input_dict["wav"])
                x0.record_stream(enc_s)
                pre_done = torch.cuda.Event()
                pre_done.record(pre_s)
            enc_s.wait_event(pre_done)
            input_dict = dict(input_dict, _logmel=x0)
        with torch.cuda.stream(enc_s):
            enc = self.encoder.forward_front(input_dict) if split_enc else self.encoder(input_dict)
            enc_done = torch.cuda.Event()
            enc_done.record(enc_s)
        # "auto" pairing asks whether the decode stream is idle: asked BEFORE this batch's own GRU is queued on it (afterwards
        # the stream is never idle, and "auto" would hold every lone submission until result())
        dec_was_idle = chain_s.query() and dec_s.query()
        if split_enc:
            with torch.cuda.stream(dec_s):
                dec_s.wait_event(enc_done)
                for t in enc["feats"].values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(dec_s)
                enc = self.encoder.forward_back(enc)
                enc_done = torch.cuda.Event()
                enc_done.record(dec_s)
        max_length = int(input_dict.get("max_length", self.max_length))
        item = (PendingCaption(self), enc, enc_done, max_length)
        item[0]._input = submitted
        # submissions per chain at most: see decode_group()
        gmax = decode_group() if pair != "0" else 1
        held = self._held or []
        if held:
            h = held[0]
            same = (h[3] == max_length and h[1]["attn_emb"].shape == enc["attn_emb"].shape
                    and h[1]["attn_emb"].device == enc["attn_emb"].device)
            if not same:
                self._decode_group(held)
                held = []
        held = held + [item]
        # "auto": a lone batch waits only while the decode stream is still busy; held batches go as soon as it is idle
        if len(held) >= gmax or (pair == "auto" and dec_was_idle):
            self._held = None
            self._decode_group(held)
        else:
            self._held = held
        return item[0]

    def _forward_async_beam(self, input_dict):
        """Beam search is a host-driven loop (it asks the device now and then whether any clip is still searching), so
        only the ENCODER is submitted here, on the encoder stream; the search itself runs on the decode stream when
        ``result()`` is called - under the encoders of the batches submitted after this one."""
        dev = input_dict["wav"].device
        if self._streams is None or self._streams[0].device != dev:
            self._streams = _make_streams(dev)
        enc_s, dec_s = self._streams[:2]
        enc_s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(enc_s):
            enc = self.encoder(input_dict)
            enc_done = torch.cuda.Event()
            enc_done.record(enc_s)
        pending = PendingCaption(self)
        pending._lazy = (dict(input_dict), enc, enc_done)
        if getattr(self, "_lazy_queue", None) is None:
            self._lazy_queue = []
        # weak references: a handle the caller dropped without asking for its result() must not keep its encoder outputs alive
        self._lazy_queue = [r for r in self._lazy_queue if r() is not None and r()._lazy is not None]
        self._lazy_queue.append(weakref.ref(pending))
        return pending

    @staticmethod
    def _beam_group_key(input_dict, enc):
        """Submissions that may share one search: the same search parameters and the same memory geometry."""
        a = enc["attn_emb"]
        return (tuple(a.shape[1:]), a.device, input_dict.get("beam_size", 3), input_dict.get("max_length"),
                input_dict.get("temp", 1.0), bool(input_dict.get("n_best", False)), input_dict.get("n_best_size"))

    def _run_lazy(self, pending):
        """The beam search of a submitted batch, on the decode stream.  AUDIOCAPTION_BEAM_GROUP=n (default 2) searches up to n
        consecutive submissions with the same search parameters as ONE batch and splits the results (clips are independent:
        the same token ids as separate searches; a clip that has its `beam` finished beams is retired whatever the other
        clips do).  Beside running encoders a search is slowed by every one of its ~260 dependent launches waiting for
        workgroup slots, so one search per two submissions wins: EffB2-Trm (128 clips, beam 3) 7.60 -> 7.08 ms per
        submission (16.8 k -> 18.1 k clips/s); 3 / 4 submissions per search: 7.16 / 7.14 ms.  (In round 2, before the
        decode-side changes of round 3, the grouped search measured slower: 8.8 vs 7.9 ms.)"""
        refs = getattr(self, "_lazy_queue", None) or []
        queue = [q for q in (r() for r in refs) if q is not None and q._lazy is not None]
        # AUDIOCAPTION_BEAM_CONCURRENT=n (default 1): the searches of n consecutive submissions run side by side, each on
        # its own decode stream over its own static buffers, driven in lockstep by this thread.  Measured on EffB2-Trm
        # (128 clips, beam 3): 9.4 ms per submission with two searches in flight, 8.9 with three, 7.7 one by one - the
        # search over 384 rows is not waiting for CU slots, it shares the device's throughput with the next encoders.
        conc = max(1, int(os.environ.get("AUDIOCAPTION_BEAM_CONCURRENT", "1")))
        limit = max(1, int(os.environ.get("AUDIOCAPTION_BEAM_GROUP", "2")))
        merge = limit > 1
        if not merge:
            limit = conc
        group = [pending]
        if pending in queue:
            i = queue.index(pending)
            key = self._beam_group_key(pending._lazy[0], pending._lazy[1])
            for q in queue[i + 1:]:
                if len(group) >= limit or q._lazy is None or self._beam_group_key(q._lazy[0], q._lazy[1]) != key:
                    break
                group.append(q)
        items = [g._lazy for g in group]
        for g in group:
            g._lazy = None
        self._lazy_queue = [r for r in refs if r() is not None and r()._lazy is not None]
        enc_s, dec_s = self._streams[:2]
        if len(items) > 1 and not merge:
            return self._run_concurrent(group, items)
        with torch.cuda.stream(dec_s):
            for _, enc, enc_done in items:
                dec_s.wait_event(enc_done)
                for t in enc.values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(dec_s)
            if len(items) == 1:
                outs = [self.forward_decoder(items[0][0], items[0][1])]
            else:
                rows = [it[1]["attn_emb"].shape[0] for it in items]
                merged = dict(items[0][1])
                merged["attn_emb"] = torch.cat([it[1]["attn_emb"] for it in items], 0)
                merged["fc_emb"] = torch.cat([it[1]["fc_emb"] for it in items], 0)
                merged["attn_emb_len"] = torch.cat([torch.as_tensor(it[1]["attn_emb_len"]).cpu() for it in items], 0)
                whole = self.forward_decoder(items[0][0], merged)
                outs, r0 = [], 0
                for n in rows:
                    outs.append({k: (v[r0:r0 + n] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == sum(rows)
                                     else v) for k, v in whole.items()})
                    r0 += n
            dec_s.synchronize()
            for g, (input_dict, enc, _), out in zip(group, items, outs):
                flags = _device_flags(enc)
                if flags is not None:
                    redo = self._check_flags(flags.cpu(), input_dict)
                    if redo is not None:
                        out = redo
                g._result = out

    def _run_concurrent(self, group, items):
        """Beam searches of several submissions side by side: one decode stream and one set of static buffers each."""
        dev = items[0][1]["attn_emb"].device
        if getattr(self, "_dec_streams", None) is None or self._dec_streams[0].device != dev:
            self._dec_streams = [self._streams[1]]
        while len(self._dec_streams) < len(items):
            self._dec_streams.append(torch.cuda.Stream(dev, priority=self._streams[1].priority))
        streams = self._dec_streams[:len(items)]
        runs = []
        for slot, ((input_dict, enc, enc_done), st) in enumerate(zip(items, streams)):
            with torch.cuda.stream(st):
                st.wait_event(enc_done)
                for t in enc.values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(st)
                runs.append(self._beam_begin(self._inference_dict(input_dict, enc), slot=slot))
        alive = True
        while alive:
            alive = False
            for run, st in zip(runs, streams):
                with torch.cuda.stream(st):
                    alive = self._beam_advance(run) or alive
        for g, (input_dict, enc, _), run, st in zip(group, items, runs, streams):
            with torch.cuda.stream(st):
                out = self._beam_finish(run)
                out.update(enc)
                flags = _device_flags(enc)
                if flags is not None:
                    redo = self._check_flags(flags.cpu(), input_dict)
                    if redo is not None:
                        out = redo
            g._result = out

    def _flush_held(self):
        held, self._held = self._held, None
        if held:
            self._decode_group(held)

    def _decode_group(self, items):
        """One greedy chain over the rows of all ``items`` (pending, enc, enc_done, max_length) on the decode stream."""
        enc_s, _, dec_s = self._streams   # the chain stream (the decode stream unless AUDIOCAPTION_DECODE_CUS restricts it)
        max_length = items[0][3]
        with torch.cuda.stream(dec_s):
            for _, enc, ev, _ in items:
                dec_s.wait_event(ev)
                for t in enc.values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(dec_s)
            if len(items) == 1:
                attn, lens = items[0][1]["attn_emb"], items[0][1]["attn_emb_len"]
            else:   # the decoder copies every batch into its rows of the chain's static buffer: no concatenation launch
                attn, lens = [it[1]["attn_emb"] for it in items], [it[1]["attn_emb_len"] for it in items]
            res = self.decoder.greedy(attn, lens, max_length, self.start_idx, self.end_idx, self.pad_idx)
            same_b = len({it[1]["attn_emb"].shape[0] for it in items}) == 1
            if len(items) > 1 and same_b:   # rows still unfinished after step t, per batch: one reduction for the whole chain
                cnts = (res["seq"].view(len(items), -1, max_length) != self.end_idx).sum(1).to(torch.int32)
            r0 = 0
            staged = []
            for pending, enc, _, _ in items:
                B = enc["attn_emb"].shape[0]
                rows = slice(r0, r0 + B)
                r0 += B
                pool = self._pinned.setdefault((B, max_length), [])
                if not pool:
                    pool.append((torch.empty(B, max_length, dtype=torch.int64).pin_memory(),
                                 torch.empty(B, max_length, dtype=torch.float32).pin_memory(),
                                 torch.zeros(2, dtype=torch.int32).pin_memory(),
                                 torch.zeros(max_length, dtype=torch.int32).pin_memory()))
                host_seq, host_lp, host_flag, host_cnt = pool.pop()
                host_seq.copy_(res["seq"][rows], non_blocking=True)
                host_lp.copy_(res["sampled_logprob"][rows], non_blocking=True)
                dflags = _device_flags(enc)
                if dflags is not None:
                    host_flag.copy_(dflags, non_blocking=True)
                else:
                    host_flag.zero_()
                if len(items) == 1:
                    cnt = res["unfinished_cnt"]
                elif same_b:
                    cnt = cnts[len(staged)]
                else:   # rows still unfinished after step t: finished rows hold end_idx (csrc/decoder.hip greedy_pick)
                    cnt = (res["seq"][rows] != self.end_idx).sum(0).to(torch.int32)
                if len(items) > 1:
                    host_cnt.copy_(cnt, non_blocking=True)
                out = {"logit": res["logit"][rows], "embed": res["embed"][rows], "unfinished_cnt": cnt}
                out.update(enc)
                staged.append((pending, host_seq, host_lp, host_flag, host_cnt, len(items) > 1, out, pool))
            done = torch.cuda.Event()
            done.record(dec_s)
        for pending, host_seq, host_lp, host_flag, host_cnt, shared, out, pool in staged:
            pending._fill(done, host_seq, host_lp, out,
                          (lambda pool=pool, a=host_seq, b=host_lp, c=host_flag, d=host_cnt: pool.append((a, b, c, d))), host_flag,
                          host_cnt if shared else None)

    # ---- greedy (base.py:152-218) -----------------------------------------------------------------
    def greedy_search(self, input_dict):
        if input_dict.get("temp", 1.0) != 1.0:
            pass  # greedy argmax does not depend on temp (base.py:216-218 ignores it for "greedy")
        args = (input_dict["attn_emb"], input_dict["attn_emb_len"], int(input_dict["max_length"]), self.start_idx, self.end_idx,
                self.pad_idx)
        # the blocking call: nothing else runs on the GPU while this batch decodes -> the one-launch form where it applies
        res = self.decoder.greedy(*args, alone=True)
        if "cluster_error" in res and int(res["cluster_error"].item() & 0xffffffff) != 0:
            import warnings
            warnings.warn("one-launch greedy search: a workgroup's partners never started (another process on the GPU?); "
                          "decoding this batch with the launch chain")
            res = self.decoder.greedy(*args, mode="chain")
        return {
            "seq": res["seq"].cpu(),                          # the reference keeps seq on the CPU (base.py:122)
            "logit": res["logit"],
            "sampled_logprob": res["sampled_logprob"].cpu(),  # CPU as in base.py:126
            "embed": res["embed"],
            "unfinished_cnt": res["unfinished_cnt"],
        }

    # ---- beam search (base.py:254-361), all clips batched ---------------------------------------------
    def beam_search(self, input_dict):
        run = self._beam_begin(input_dict)
        while self._beam_advance(run):
            pass
        return self._beam_finish(run)

    def _beam_begin(self, input_dict, slot=0):
        """Set a search up on the current stream: static buffers of this (shape, slot), inputs copied in.  ``slot``
        separates the buffers of searches that run concurrently on different streams."""
        dec = self.decoder
        attn_emb = input_dict["attn_emb"]
        dev = attn_emb.device
        B, Tm, _ = attn_emb.shape
        beam = int(input_dict["beam_size"])
        max_length = int(input_dict["max_length"])
        temp = float(input_dict["temp"])
        n_best = bool(input_dict.get("n_best", False))
        n_best_size = int(input_dict.get("n_best_size", beam))
        V = self.vocab_size
        R = B * beam
        A = attn_emb.shape[2]

        # Everything per step stays on the device: decoder step + scores + per-clip top-k (ac_trm_beam_step), the
        # per-clip bookkeeping of base.py:290-323 (ac_trm_beam_update) and the KV-cache re-gather
        # (ac_trm_beam_reorder).  The host only asks after steps 8, 12 and 16 whether any clip is still searching, so the
        # ~14 launches x max_length steps are four fixed launch sequences: like the greedy chain they are captured per
        # shape (second use) into HIP graphs over static buffers and replayed - a host-launched chain of ~280 small
        # dependent kernels cannot keep up once it shares the device with the next batches' encoders.
        lib = _lib.load()
        ld = max_length + 1
        cap = beam * max_length                    # upper bound of finished beams per clip
        use_graph = os.environ.get("AUDIOCAPTION_DECODE_GRAPH", "1") != "0"
        key = (dev, B, Tm, A, beam, max_length, temp, self.start_idx, self.end_idx, self.pad_idx, dec._weights_key(), slot)
        if getattr(self, "_beam_state", None) is None:
            self._beam_state = {}
        states = self._beam_state
        st = states.pop(key, None)
        if st is None:
            i32 = dict(device=dev, dtype=torch.int32)
            f32 = dict(device=dev, dtype=torch.float32)
            tok0 = torch.full((R, ld), self.end_idx, **i32)
            tok0[:, 0] = self.start_idx
            mask0 = torch.zeros(R, ld, device=dev, dtype=torch.uint8)
            if self.start_idx == self.pad_idx or self.end_idx == self.pad_idx:
                mask0 = (tok0 == self.pad_idx).to(torch.uint8)
            ws_n = lib.ac_trm_workspace_floats(ctypes.byref(dec.weights()), R, max_length)
            st = {"uses": 0, "graphs": {},
                  "attn_emb": torch.empty(B, Tm, A, **f32), "mem_len": torch.empty(B, **i32),
                  "memkv": torch.empty(dec.nlayers, B * Tm, 2 * dec.d_model, **f32),
                  "tmp": torch.empty(B * Tm, dec.d_model, **f32), "ws": torch.empty(ws_n, **f32),
                  "tok0": tok0, "tok1": torch.full((R, ld), self.end_idx, **i32), "mask0": mask0,
                  "tok": [torch.empty(R, ld, **i32) for _ in range(2)],
                  "mask": torch.empty(R, ld, device=dev, dtype=torch.uint8),
                  "cum": torch.empty(R, **f32), "active": torch.empty(B, **i32), "done_cnt": torch.empty(B, **i32),
                  "done_seq": torch.empty(B, cap, max_length, **i32), "done_score": torch.empty(B, cap, **f32),
                  "src_row": torch.empty(R, **i32), "n_active": torch.empty(1, **i32),
                  "top_val": torch.empty(B, beam, **f32), "top_idx": torch.empty(B, beam, **i32)}
        states[key] = st                       # most recently used last
        while len(states) > 6:
            states.pop(next(iter(states)))
        st["uses"] += 1
        st["attn_emb"].copy_(K.f32c(attn_emb))
        st["mem_len"].copy_(K.upload(input_dict["attn_emb_len"], dev, torch.int32))
        tok, mask, cum, active = st["tok"], st["mask"], st["cum"], st["active"]
        done_cnt, done_seq, done_score = st["done_cnt"], st["done_seq"], st["done_score"]
        src_row, n_active, ws = st["src_row"], st["n_active"], st["ws"]
        w = ctypes.byref(dec.weights())

        def segment(t0, t1):
            """Steps t0 .. t1 - 1 on the current stream (capturable: fixed launches over the static buffers)."""
            if t0 == 0:
                tok[0].copy_(st["tok0"])
                tok[1].copy_(st["tok1"])
                mask.copy_(st["mask0"])
                cum.zero_()
                active.fill_(1)
                done_cnt.zero_()
                n_active.fill_(B)
                check(lib.ac_trm_memory(w, ptr(st["attn_emb"]), B, Tm, ptr(st["memkv"]), ptr(st["tmp"]), stream()),
                      "ac_trm_memory")
            for t in range(t0, t1):
                check(lib.ac_trm_beam_step(w, ptr(st["memkv"]), ptr(st["mem_len"]), B, beam, Tm, max_length, t, float(temp),
                                           ptr(tok[t & 1]), ptr(mask), ptr(cum), ptr(st["top_val"]), ptr(st["top_idx"]),
                                           ptr(ws), stream()), "ac_trm_beam_step")
                check(lib.ac_trm_beam_update(ptr(st["top_val"]), ptr(st["top_idx"]), ptr(tok[t & 1]), ptr(tok[(t + 1) & 1]),
                                             ptr(mask), ptr(cum), ptr(active), ptr(done_cnt), ptr(done_seq),
                                             ptr(done_score), ptr(src_row), ptr(n_active), B, beam, V, max_length, t,
                                             self.end_idx, self.pad_idx, cap, stream()), "ac_trm_beam_update")
                if t + 1 < max_length:
                    check(lib.ac_trm_beam_reorder(w, R, max_length, t, ptr(src_row), ptr(ws), stream()),
                          "ac_trm_beam_reorder")

        # AUDIOCAPTION_BEAM_SEGMENTS="0,8,12,16" (default): the steps after which the host asks whether any clip is still searching
        # ("0": never - one launch sequence for the whole search, no host synchronisation inside it)
        seg = [int(v) for v in os.environ.get("AUDIOCAPTION_BEAM_SEGMENTS", "0,8,12,16").split(",") if v.strip() != ""]
        bounds = sorted({0} | {b for b in seg if 0 < b < max_length}) + [max_length]
        return {"st": st, "segment": segment, "bounds": bounds, "next": 0, "use_graph": use_graph and st["uses"] >= 2,
                "dev": dev, "B": B, "beam": beam, "max_length": max_length, "cap": cap, "V": V, "n_best": n_best,
                "n_best_size": n_best_size}

    def _beam_advance(self, run):
        """Launch the next segment of steps on the current stream; False when the search is over (all segments done, or
        every clip has its `beam` finished beams, base.py:318-323 - asked between segments only)."""
        i, bounds, st = run["next"], run["bounds"], run["st"]
        if i >= len(bounds) - 1:
            return False
        if i > 0 and int(st["n_active"].item()) == 0:
            run["next"] = len(bounds)
            return False
        t0, t1 = bounds[i], bounds[i + 1]
        if not run["use_graph"]:
            run["segment"](t0, t1)
        else:
            graph = st["graphs"].get(t0)
            if graph is None:
                torch.cuda.synchronize(run["dev"])
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    run["segment"](t0, t1)
                st["graphs"][t0] = graph
            graph.replay()
        run["next"] = i + 1
        return True

    def _beam_finish(self, run):
        st, dec = run["st"], self.decoder
        B, max_length, cap, V, dev = run["B"], run["max_length"], run["cap"], run["V"], run["dev"]
        n_best, n_best_size = run["n_best"], run["n_best_size"]
        done_cnt, done_seq, done_score = st["done_cnt"], st["done_seq"], st["done_score"]
        counts = done_cnt.cpu().numpy()
        seqs = done_seq.cpu().numpy()
        scores = done_score.cpu().numpy()

        if n_best:
            seq = torch.full((B, n_best_size, max_length), self.end_idx, dtype=torch.long)
        else:
            seq = torch.full((B, max_length), self.end_idx, dtype=torch.long)
        for i in range(B):
            n = min(int(counts[i]), cap)
            order = sorted(range(n), key=lambda j: -scores[i, j])   # stable: ties keep the append order
            if n_best:
                for j, o in enumerate(order[:n_best_size]):
                    seq[i, j] = torch.from_numpy(seqs[i, o].astype(np.int64))
            else:
                seq[i] = torch.from_numpy(seqs[i, order[0]].astype(np.int64))
        # logit / embed / sampled_logprob are not filled by the reference's beam search either
        # (base.py:124-127 leaves them at torch.empty / zeros)
        return {
            "seq": seq,
            "logit": torch.empty(B, max_length, V, device=dev),
            "sampled_logprob": torch.zeros(B, max_length),
            "embed": torch.empty(B, max_length, dec.d_model, device=dev),
        }
