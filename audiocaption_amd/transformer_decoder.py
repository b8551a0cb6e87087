"""Transformer caption decoder, MI355X path.  Plugin-compatible with the reference class
``captioning.models.transformer_decoder.TransformerDecoder`` (transformer_decoder.py:11-103) and its
base ``captioning.models.BaseDecoder`` (models/__init__.py:64-92): same constructor, attributes
(``vocab_size``, ``d_model``, ``emb_dim``, ``fc_emb_dim``, ``attn_emb_dim``), ``state_dict()`` keys and
``forward({"word", "attn_emb", "attn_emb_len", "cap_padding_mask"}) -> {"embed", "logit"}``.

The nn modules own the parameters only.  All arithmetic runs in csrc/decoder.hip + csrc/gemm.hip:
``memory()`` projects the audio features once, ``forward`` / ``greedy`` / ``beam_*`` run the per-position
decoder step against a self-attention KV cache.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from . import _lib
from . import kernels as K
from ._lib import check, f32c, ptr, stream


class PositionalEncoding(nn.Module):
    """Sinusoid table kept as a frozen Parameter named ``pe`` so that it appears in checkpoints exactly
    like the reference's (model_util.py:167-186)."""

    def __init__(self, d_model, dropout=0.1, max_len=100):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_parameter("pe", nn.Parameter(pe.unsqueeze(1), requires_grad=False))


class BaseDecoder(nn.Module):

    def __init__(self, emb_dim, vocab_size, fc_emb_dim, attn_emb_dim, dropout=0.2, tie_weights=False):
        super().__init__()
        self.emb_dim = emb_dim
        self.vocab_size = vocab_size
        self.fc_emb_dim = fc_emb_dim
        self.attn_emb_dim = attn_emb_dim
        self.tie_weights = tie_weights
        self.word_embedding = nn.Embedding(vocab_size, emb_dim)
        self.in_dropout = nn.Dropout(dropout)


class TransformerDecoder(BaseDecoder):

    def __init__(self, emb_dim, vocab_size, fc_emb_dim, attn_emb_dim, dropout, freeze=False,
                 tie_weights=False, **kwargs):
        super().__init__(emb_dim, vocab_size, fc_emb_dim, attn_emb_dim, dropout=dropout, tie_weights=tie_weights)
        self.d_model = emb_dim
        self.nhead = kwargs.get("nhead", self.d_model // 64)
        self.nlayers = kwargs.get("nlayers", 2)
        self.dim_feedforward = kwargs.get("dim_feedforward", self.d_model * 4)
        self.pos_encoder = PositionalEncoding(self.d_model, dropout)
        layer = nn.TransformerDecoderLayer(d_model=self.d_model, nhead=self.nhead,
                                           dim_feedforward=self.dim_feedforward, dropout=dropout)
        self.model = nn.TransformerDecoder(layer, self.nlayers)
        self.classifier = nn.Linear(self.d_model, vocab_size, bias=False)
        if tie_weights:
            self.classifier.weight = self.word_embedding.weight
        self.attn_proj = nn.Sequential(nn.Linear(self.attn_emb_dim, self.d_model), nn.ReLU(),
                                       nn.Dropout(dropout), nn.LayerNorm(self.d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.freeze = freeze
        if freeze:
            for p in self.parameters():
                p.requires_grad = False
        self._w = None
        self._w_key = None
        self._w_keep = None
        self._ws = {}
        self._greedy_state = None

    def load_pretrained(self, pretrained, output_fn=print):
        """Reference transformer_decoder.py:56-72: take the ``decoder.*`` entries of a model checkpoint."""
        checkpoint = torch.load(pretrained, map_location="cpu")
        if "model" in checkpoint:
            checkpoint = checkpoint["model"]
        sd = {k[8:]: v for k, v in checkpoint.items() if k.startswith("decoder.")} or checkpoint
        own = self.state_dict()
        loaded = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        own.update(loaded)
        self.load_state_dict(own, strict=True)
        if self.freeze:
            for name, param in self.named_parameters():
                param.requires_grad = name not in loaded

    # ------------------------------------------------------------------------------------------
    def _weights_key(self):
        return tuple((t.data_ptr(), t._version, t.dtype) for t in self.parameters()) + (_lib.param_generation(),)

    def weights(self):
        """ac_trm_weights struct of device pointers (rebuilt when a parameter changes)."""
        key = self._weights_key()
        if self._w is not None and key == self._w_key:
            return self._w
        keep = []

        def P(t):
            t = f32c(t.detach())
            keep.append(t)
            return ctypes.c_void_p(ptr(t).value)

        if self.nlayers > _lib.AC_MAX_LAYERS:
            raise ValueError("too many decoder layers for the HIP path")
        w = _lib.AcTrmWeights()
        w.d_model, w.nhead, w.nlayers, w.dim_ff = self.d_model, self.nhead, self.nlayers, self.dim_feedforward
        w.vocab, w.max_pos, w.attn_emb_dim = self.vocab_size, self.pos_encoder.pe.shape[0], self.attn_emb_dim
        w.emb, w.pe, w.cls_w = P(self.word_embedding.weight), P(self.pos_encoder.pe), P(self.classifier.weight)
        w.proj_w, w.proj_b = P(self.attn_proj[0].weight), P(self.attn_proj[0].bias)
        w.proj_ln_w, w.proj_ln_b = P(self.attn_proj[3].weight), P(self.attn_proj[3].bias)
        for i, L in enumerate(self.model.layers):
            wl = w.layer[i]
            wl.sa_in_w, wl.sa_in_b = P(L.self_attn.in_proj_weight), P(L.self_attn.in_proj_bias)
            wl.sa_out_w, wl.sa_out_b = P(L.self_attn.out_proj.weight), P(L.self_attn.out_proj.bias)
            wl.ca_in_w, wl.ca_in_b = P(L.multihead_attn.in_proj_weight), P(L.multihead_attn.in_proj_bias)
            wl.ca_out_w, wl.ca_out_b = P(L.multihead_attn.out_proj.weight), P(L.multihead_attn.out_proj.bias)
            wl.l1_w, wl.l1_b, wl.l2_w, wl.l2_b = P(L.linear1.weight), P(L.linear1.bias), P(L.linear2.weight), P(L.linear2.bias)
            wl.n1_w, wl.n1_b = P(L.norm1.weight), P(L.norm1.bias)
            wl.n2_w, wl.n2_b = P(L.norm2.weight), P(L.norm2.bias)
            wl.n3_w, wl.n3_b = P(L.norm3.weight), P(L.norm3.bias)
        # fragment-packed copy of the step projections (read by every decode position)
        lib = _lib.load()
        n = lib.ac_trm_step_pack_floats(ctypes.byref(w))
        if n <= 0:
            raise _lib.HipLibraryError("ac_trm_step_pack_floats rejected the decoder configuration")
        pk = torch.empty(n, device=self.word_embedding.weight.device, dtype=torch.float32)
        check(lib.ac_trm_pack_step_weights(ctypes.byref(w), ptr(pk), stream()), "ac_trm_pack_step_weights")
        keep.append(pk)
        w.step_pk = ctypes.c_void_p(pk.data_ptr())
        self._cluster_pk = None   # per-part blobs of the one-launch greedy search, packed on first use (cluster_pack)
        self._w, self._w_key, self._w_keep = w, key, keep
        return w

    def cluster_pack(self):
        """Weights in the layout of the one-launch greedy search (csrc/decoder_cluster.hip: per part, per layer, row-major
        [k][column] slices), or None when the decoder's shape is not the one that kernel covers.  Cached with ``weights()``."""
        w = self.weights()
        if getattr(self, "_cluster_pk", None) is None:
            lib = _lib.load()
            n = lib.ac_trm_cluster_pack_floats(ctypes.byref(w))
            if n <= 0:
                self._cluster_pk = False
            else:
                pk = torch.empty(n, device=self.word_embedding.weight.device, dtype=torch.float32)
                check(lib.ac_trm_cluster_pack(ctypes.byref(w), ptr(pk), stream()), "ac_trm_cluster_pack")
                self._cluster_pk = pk
        return self._cluster_pk if self._cluster_pk is not False else None

    def cluster_covers(self, rows, Tm, max_length, device=None):
        """The one-launch greedy search takes this problem: shape covered, the row's keys and values fit the LDS, and the
        clusters are resident together - one workgroup per CU, four per row: rows <= compute units // 4 of ``device`` (64 on
        a whole MI355X; a CU-masked or partitioned device has fewer and would run the clusters in rounds, slower than the
        launch chain).  AUDIOCAPTION_CLUSTER_MAX_ROWS overrides the bound."""
        if self.d_model != 256 or self.nhead != 4 or self.dim_feedforward != 1024 or max_length > 32:
            return False
        vq = (self.vocab_size + 3) // 4
        if 3 * vq >= self.vocab_size:     # a vocabulary quarter without a column (csrc/decoder_cluster.hip cluster_shape_ok)
            return False
        nlc = (vq + 63) // 64 * 64
        lds = 4 * (256 + 192 + 64 + 256 + 5 * 256 + ((max(Tm, 32) + 3) & ~3) + 256 + 16 + 1024 + nlc
                   + 2 * self.nlayers * (max_length + Tm) * 64) + 64
        env = os.environ.get("AUDIOCAPTION_CLUSTER_MAX_ROWS")
        if env is not None:
            max_rows = int(env)
        else:
            dev = device if device is not None else self.word_embedding.weight.device
            max_rows = torch.cuda.get_device_properties(dev).multi_processor_count // 4 if dev.type == "cuda" else 0
        return lds <= 159 * 1024 and rows <= max_rows

    def workspace(self, rows, max_len, device):
        lib = _lib.load()
        n = lib.ac_trm_workspace_floats(ctypes.byref(self.weights()), rows, max_len)
        if n <= 0:
            raise _lib.HipLibraryError("ac_trm_workspace_floats rejected the decoder configuration")
        ws = self._ws.get(device)
        if ws is None or ws.numel() < n:
            ws = torch.empty(n, device=device, dtype=torch.float32)
            self._ws[device] = ws
        return ws

    def memory(self, attn_emb):
        """attn_emb (R, Tm, attn_emb_dim) -> memkv (nlayers, R*Tm, 2*d): attn_proj + cross-attn K/V."""
        lib = _lib.load()
        attn_emb = f32c(attn_emb)
        R, Tm, _ = attn_emb.shape
        memkv = torch.empty(self.nlayers, R * Tm, 2 * self.d_model, device=attn_emb.device, dtype=torch.float32)
        tmp = torch.empty(R * Tm, self.d_model, device=attn_emb.device, dtype=torch.float32)
        check(lib.ac_trm_memory(ctypes.byref(self.weights()), ptr(attn_emb), R, Tm, ptr(memkv), ptr(tmp), stream()),
              "ac_trm_memory")
        return memkv

    def forward(self, input_dict):
        if self.training:
            raise NotImplementedError(
                "TransformerDecoder (HIP path): in train mode the decoder only runs inside the whole-model training "
                "step (audiocaption_amd.train.TrainEngine / TransformerModel.forward with mode='train')")
        lib = _lib.load()
        attn_emb = input_dict["attn_emb"]
        dev = attn_emb.device
        word = input_dict["word"].to(dev)
        N, T = word.shape
        Tm = attn_emb.shape[1]
        mem_len = K.upload(input_dict["attn_emb_len"], dev, torch.int32)
        mask = input_dict.get("cap_padding_mask")
        mask_u8 = None if mask is None else mask.to(dev).to(torch.uint8).contiguous()
        memkv = self.memory(attn_emb)
        tokens = word.to(torch.int32).contiguous()
        embed = torch.empty(N, T, self.d_model, device=dev, dtype=torch.float32)
        logit = torch.empty(N, T, self.vocab_size, device=dev, dtype=torch.float32)
        ws = self.workspace(N, T, dev)
        check(lib.ac_trm_forward_tokens(ctypes.byref(self.weights()), ptr(memkv), ptr(mem_len), N, Tm, ptr(tokens),
                                        ptr(mask_u8), T, ptr(embed), ptr(logit), ptr(ws), stream()),
              "ac_trm_forward_tokens")
        return {"embed": embed, "logit": logit}

    def _greedy_launch(self, st, max_length, start_idx, end_idx, pad_idx):
        """Memory preparation + the whole on-device greedy loop on the current stream (capturable)."""
        lib = _lib.load()
        B, Tm, _ = st["attn_emb"].shape
        w = ctypes.byref(self.weights())
        check(lib.ac_trm_memory(w, ptr(st["attn_emb"]), B, Tm, ptr(st["memkv"]), ptr(st["tmp"]), stream()),
              "ac_trm_memory")
        if st.get("cluster_ws") is not None:
            check(lib.ac_trm_greedy_cluster(w, ptr(st["cluster_pk"]), ptr(st["memkv"]), ptr(st["mem_len"]), B, Tm, max_length,
                                            start_idx, end_idx, pad_idx, ptr(st["seq"]), ptr(st["logit"]),
                                            ptr(st["sampled_logprob"]), ptr(st["embed"]), ptr(st["unfinished_cnt"]),
                                            ptr(st["cluster_ws"]), int(st["early_stop"]), stream()), "ac_trm_greedy_cluster")
            return
        check(lib.ac_trm_greedy(w, ptr(st["memkv"]), ptr(st["mem_len"]), B, Tm, max_length, start_idx, end_idx,
                                pad_idx, ptr(st["seq"]), ptr(st["logit"]), ptr(st["sampled_logprob"]),
                                ptr(st["embed"]), ptr(st["unfinished_cnt"]), ptr(st["ws"]), stream()),
              "ac_trm_greedy")

    def greedy(self, attn_emb, attn_emb_len, max_length, start_idx, end_idx, pad_idx, alone=False, mode=None):
        """On-device greedy search.  Returns device tensors seq (int64), logit, logprob, embed, cnt.  ``attn_emb`` /
        ``attn_emb_len``: one batch, or lists of batches of the same (frames, width) decoded as one chain.

        Two forms of the same search.  "chain": ten launches per step (csrc/decoder.hip) - ~90 us per step whatever the row
        count, the form that shares the GPU with the next batch's encoder (``forward_async``).  "cluster": ONE persistent
        launch, a cluster of four workgroups per row (csrc/decoder_cluster.hip) - about half the time per step when
        nothing else runs on the GPU and the rows fit (``cluster_covers``).  ``mode`` (default: AUDIOCAPTION_GREEDY =
        "auto"): "auto" takes the cluster form when the caller says the decode runs ``alone`` (the blocking ``model()``
        call) and the problem is covered.  The returned dict then carries "cluster_error" (device int32 word, non-zero:
        a workgroup's partners never started - rerun with mode="chain").

        The ~370 short launches of a decode are latency-bound, so the fixed launch sequence (memory
        preparation + max_length decoder steps) is captured once per shape (on its second use) into a HIP graph over
        static buffers and replayed; inputs are copied in, outputs are cloned out (fresh tensors per call, as the
        reference returns).  Set AUDIOCAPTION_DECODE_GRAPH=0 to launch eagerly."""
        parts = list(attn_emb) if isinstance(attn_emb, (list, tuple)) else [attn_emb]   # several batches, one chain: each is
        dev = parts[0].device                                                           # copied into its rows of the static buffer
        B, (Tm, A) = sum(p_.shape[0] for p_ in parts), parts[0].shape[1:]
        use_graph = os.environ.get("AUDIOCAPTION_DECODE_GRAPH", "1") != "0"
        mode = mode or os.environ.get("AUDIOCAPTION_GREEDY", "auto")
        if mode not in ("auto", "chain", "cluster"):
            raise ValueError(f"AUDIOCAPTION_GREEDY={mode!r}: 'auto', 'chain' or 'cluster'")
        wanted = mode == "cluster" or (mode == "auto" and alone)   # the weights are repacked for it only when it can be chosen
        covered = wanted and self.cluster_covers(B, Tm, max_length, dev) and self.cluster_pack() is not None
        if mode == "cluster" and not covered:
            raise _lib.HipLibraryError("the one-launch greedy search does not cover this decoder shape / row count")
        cluster = wanted and covered
        # AUDIOCAPTION_CLUSTER_EARLY_STOP=0: the one-launch form runs all max_length steps like the launch chain (benchmarks that
        # compare the two at equal work; the outputs are the same either way)
        early = os.environ.get("AUDIOCAPTION_CLUSTER_EARLY_STOP", "1") != "0"
        key = (dev, B, Tm, max_length, start_idx, end_idx, pad_idx, cluster, early, self._weights_key())
        if self._greedy_state is None:
            self._greedy_state = {}
        states = self._greedy_state
        st = states.pop(key, None)
        if st is None:
            f32 = dict(device=dev, dtype=torch.float32)
            ws_n = _lib.load().ac_trm_workspace_floats(ctypes.byref(self.weights()), B, max_length)
            st = {
                "key": key, "graph": None, "uses": 0,
                "attn_emb": torch.empty(B, Tm, A, **f32), "mem_len": torch.empty(B, device=dev, dtype=torch.int32),
                "memkv": torch.empty(self.nlayers, B * Tm, 2 * self.d_model, **f32),
                "tmp": torch.empty(B * Tm, self.d_model, **f32), "ws": torch.empty(ws_n, **f32),
                "seq": torch.empty(B, max_length, device=dev, dtype=torch.int64),
                "logit": torch.empty(B, max_length, self.vocab_size, **f32),
                "sampled_logprob": torch.empty(B, max_length, **f32),
                "embed": torch.empty(B, max_length, self.d_model, **f32),
                "unfinished_cnt": torch.empty(max_length, device=dev, dtype=torch.int32),
            }
            if cluster:
                nb = _lib.load().ac_trm_cluster_workspace_bytes(B)
                st["cluster_ws"] = torch.zeros((nb + 7) // 8, device=dev, dtype=torch.int64)   # (the error word is reset by every call)
                st["cluster_pk"] = self.cluster_pack()
                st["early_stop"] = early
        states[key] = st                       # most recently used last; batches of changing length keep 8 shapes
        while len(states) > 8:
            states.pop(next(iter(states)))
        st["uses"] += 1
        r0 = 0
        for p_ in parts:
            st["attn_emb"][r0:r0 + p_.shape[0]].copy_(p_)
            r0 += p_.shape[0]
        if isinstance(attn_emb_len, (list, tuple)):
            attn_emb_len = torch.cat([torch.as_tensor(l_).cpu().reshape(-1) for l_ in attn_emb_len])   # host side
        st["mem_len"].copy_(K.upload(attn_emb_len, dev, torch.int32))
        if not use_graph or st["uses"] < 2:
            # first batch of this shape: plain launches (also the warm-up a capture needs); a shape that never comes
            # back is never captured
            self._greedy_launch(st, max_length, start_idx, end_idx, pad_idx)
        else:
            if st["graph"] is None:
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._greedy_launch(st, max_length, start_idx, end_idx, pad_idx)
                st["graph"] = graph
            st["graph"].replay()
        out = {k: st[k].clone() for k in ("seq", "logit", "sampled_logprob", "embed", "unfinished_cnt")}
        if cluster:
            out["cluster_error"] = st["cluster_ws"][:1].clone()
        return out

    def beam_step(self, memkv, mem_len, B, beam, Tm, max_length, t, temp, tokens, mask, cum, ws):
        lib = _lib.load()
        dev = memkv.device
        top_val = torch.empty(B, beam, device=dev, dtype=torch.float32)
        top_idx = torch.empty(B, beam, device=dev, dtype=torch.int32)
        check(lib.ac_trm_beam_step(ctypes.byref(self.weights()), ptr(memkv), ptr(mem_len), B, beam, Tm, max_length,
                                   t, float(temp), ptr(tokens), ptr(mask), ptr(cum), ptr(top_val), ptr(top_idx),
                                   ptr(ws), stream()), "ac_trm_beam_step")
        return top_val, top_idx

    def beam_reorder(self, R, max_length, t, src_row, ws):
        lib = _lib.load()
        check(lib.ac_trm_beam_reorder(ctypes.byref(self.weights()), R, max_length, t, ptr(src_row), ptr(ws),
                                      stream()), "ac_trm_beam_reorder")
