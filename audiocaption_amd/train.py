"""Training step of the Cnn14Rnn-Trm captioner on the MI355X path.

What the reference does per iteration (run.py:77-148): ``model.train()``; ``output = model(input_dict)`` with
``mode="train"`` - the frozen Cnn14 (dropout active, BatchNorm in eval mode), the 3-layer bi-GRU and the
scheduled-sampling stepwise decoder (base.py:131-199, transformer_model.py:34-57) -; LabelSmoothingLoss;
``loss.backward()``; ``clip_grad_norm_``; ``Adam.step()``; under DDP the gradients are all-reduced.

Here the whole step is HIP kernels driven from this file through the C ABI (csrc/train.hip):

* every decoder prefix pass of the scheduled-sampling loop lives in one ROW SPACE (pass t owns rows
  ``off_t + n*(t+1) + l``); the forward runs pass by pass (the greedy token of step t feeds step t+1 on the device,
  no host synchronisation), the backward runs ONCE over all rows because the passes are independent given the tokens;
* weight gradients are split-K MFMA GEMMs over all rows that accumulate straight into ONE flat gradient buffer whose
  views are the parameters' ``.grad`` - so the DDP all-reduce is one RCCL call and clip + Adam are three launches;
* dropout masks come from a counter hash, regenerated in the backward (nothing stored) and reproducible by the CPU
  oracle; the per-step seed lives in device memory so that a captured HIP graph draws fresh masks when replayed.

``TrainEngine`` is the fast path (``engine.step(batch)``); ``TransformerModel.forward`` with ``mode="train"`` goes
through the same engine and returns ``logit`` attached to torch.autograd by ONE bridge node, so the reference's
runner (``loss.backward()``, any torch optimizer) works unchanged.
"""
import os
import random

import torch

from . import _lib
from ._lib import check
from .cnn_encoder import cnn14_feat_len

# dropout site codes (low 16 bits of the seed; oracle/train_path.py restates them)
OP_CNN_BLOCK = 1
OP_SPECAUG = 9
OP_GRU_LAYER = 10
OP_MEM = 20
OP_EMB_A, OP_EMB_B = 21, 22
OP_LAYER = 30

D = 256
H = 256

# Optional per-launch observer (bench.py: roofline of the training GEMMs): called as hook(phase, info) with phase
# "pre" / "post" around every ac_gemm launch of an EAGER step; info = {"phase": "forward" | "backward", "M", "N", "K",
# "flops"}.  Never set while a graph is being captured.
GEMM_HOOK = None


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatParams:
    """The trainable parameters re-pointed into one flat fp32 buffer (``p.data`` become views) plus a flat gradient
    buffer of the same layout.  GRU tensors are ordered so that the two directions of each kind are adjacent
    ([W_ih fwd; W_ih rev] is one (2*3H, In) matrix)."""

    def __init__(self, model):
        rnn = model.encoder.rnn.network
        names = []
        for l in range(rnn.num_layers):
            for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                names += [f"encoder.rnn.network.{kind}_l{l}", f"encoder.rnn.network.{kind}_l{l}_reverse"]
        named = dict(model.named_parameters())
        for k, p in named.items():
            if k.startswith("decoder.") and p.requires_grad:
                names.append(k)
        for k in names:
            if not named[k].requires_grad:
                raise NotImplementedError(f"TrainEngine: parameter {k} is frozen; partial freezing of the GRU/decoder "
                                          "is not built")
        for k, p in named.items():
            if p.requires_grad and k not in names:
                raise NotImplementedError(
                    f"TrainEngine: {k} requires grad, but only the GRU and the decoder are trainable on the HIP path "
                    "(the reference configs freeze the Cnn14: freeze_cnn / freeze_cnn_bn, cnn14rnn_trm.yaml:11-13)")
        self.names = names
        self.params = [named[k] for k in names]
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += _align(p.numel())
        self.total = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad_views = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.grad_views.append(self.grad[o:o + p.numel()].view(p.shape))
        self.index = {k: i for i, k in enumerate(names)}
        # the GRU tensors come first, the decoder's after them: [0, decoder_offset) is final only after the GRU's
        # backward through time, [decoder_offset, total) as soon as the decoder's backward is done
        self.decoder_offset = next((o for k, o in zip(names, self.offsets) if k.startswith("decoder.")), off)
        # tied parameters (``tie_weights=True``: decoder.classifier.weight IS decoder.word_embedding.weight) appear once
        # in named_parameters(); the other name resolves to the same slot, so both uses accumulate into one gradient
        by_id = {id(p): i for i, p in enumerate(self.params)}
        for k, p in model.named_parameters(remove_duplicate=False):
            if k not in self.index and id(p) in by_id:
                self.index[k] = by_id[id(p)]

    def intact(self):
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def p(self, name):
        return self.flat.data_ptr() + 4 * self.offsets[self.index[name]]

    def g(self, name):
        return self.grad.data_ptr() + 4 * self.offsets[self.index[name]]

    def attach_grads(self):
        """Fast path: make the views of the flat buffer the parameters' ``.grad`` (no copies)."""
        for p, v in zip(self.params, self.grad_views):
            p.grad = v


def dist_world_size(process_group=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(process_group)


class _RsAgWork:
    """Handle of a reduce-scatter + all-gather pair (and the all-reduce of the few trailing elements): ``wait()`` like a
    torch work handle.  On RCCL both collectives are issued at once (stream-ordered on the communicator's stream); a
    backend whose asynchronous operations may run concurrently (gloo) gets the all-gather issued once the scatter is in."""

    def __init__(self, first, then, extra):
        self._first, self._then, self._extra = first, then, extra

    def wait(self):
        for w in self._first:
            if w is not None:
                w.wait()
        for w in self._then():
            if w is not None:
                w.wait()
        if self._extra is not None:
            self._extra.wait()
        return True


_RS_SHARDS = {}


def default_grad_sync(world, backend):
    """Gradient exchange when AUDIOCAPTION_GRAD_SYNC is not set: on RCCL with four or more ranks the sum goes as a direct
    reduce-scatter + all-gather ("rs_ag": on a fully connected xGMI node every peer receives its shard over its own link,
    and the two halves are separate collectives later work can sit between); a ring all-reduce otherwise (two ranks share
    one link either way; gloo has no reduce_scatter_tensor fast path)."""
    return "rs_ag" if backend == "nccl" and world >= 4 else "all_reduce"


def allreduce_flat_gradients(flat_grad, process_group=None, async_op=False, algo=None):
    """Sum (a slice of) the flat gradient buffer over the ranks - the reference's DDP gradient all-reduce
    (run_ddp.py:98-108) as ONE collective per slice instead of per-parameter buckets - and return the world size (or,
    with ``async_op``, the work handle, None for a single rank); the division by the world size is folded into the clip
    coefficient (``clip_grad_norm_(..., grad_div=world)``).  No-op without an initialised process group.

    ``algo`` (default AUDIOCAPTION_GRAD_SYNC, else ``default_grad_sync``): "rs_ag" spells the sum out as ``reduce_scatter_tensor`` into
    this rank's 1/N shard followed by ``all_gather_into_tensor`` back into the slice (the < N trailing elements that do not
    divide go through a tiny all-reduce).  Same sums, same bytes per link as a ring all-reduce; it exists so that the two
    halves are separate collectives on the wire of a fully connected xGMI node (7 links per GPU: each of the N - 1 peers
    receives its shard directly) and so that later work can sit between them.  Bit-equal to "all_reduce" for two ranks
    (tests/test_train_oracle.py, tests/test_gpu_train_ddp.py)."""
    import torch.distributed as dist
    world = dist_world_size(process_group)
    work = None
    if world > 1:
        algo = algo or os.environ.get("AUDIOCAPTION_GRAD_SYNC") or default_grad_sync(world, dist.get_backend(process_group))
        if algo == "all_reduce":
            work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=process_group, async_op=async_op)
        elif algo == "rs_ag":
            n = flat_grad.numel()
            body = n - n % world
            key = (flat_grad.device, body // world)
            shard = _RS_SHARDS.get(key)
            if shard is None:
                shard = _RS_SHARDS[key] = torch.empty(body // world, device=flat_grad.device, dtype=flat_grad.dtype)
            head, tail = flat_grad[:body], flat_grad[body:]
            extra = dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=process_group, async_op=async_op) if n > body else None
            ordered = dist.get_backend(process_group) == "nccl"
            first = [dist.reduce_scatter_tensor(shard, head, op=dist.ReduceOp.SUM, group=process_group, async_op=async_op)] if body else []

            def gather():
                return [dist.all_gather_into_tensor(head, shard, group=process_group, async_op=async_op)] if body else []

            if not async_op:
                gather()
            elif ordered:
                second = gather()
                work = _RsAgWork(first + second, lambda: [], extra)
            else:
                work = _RsAgWork(first, gather, extra)
        else:
            raise ValueError(f"AUDIOCAPTION_GRAD_SYNC={algo!r}: 'all_reduce' or 'rs_ag'")
    return work if async_op else world


class _SideStream:
    """The backward's OFF-CHAIN work - weight-gradient products dW += dy^T x and bias column sums: nothing downstream reads
    them before the optimiser - on a second stream beside the input-gradient chain (dx products, attention / LayerNorm /
    GRU backward kernels: each a few tens of microseconds of latency-bound work that leaves most of the chip idle).
    ``fork()`` makes the side stream wait for everything queued on the current stream and returns its handle; ``join()``
    makes the current stream wait for the side stream - called before a buffer the side stream still reads is written
    again, before kernels that accumulate into the same gradients, and at the end of every captured part.  Works inside HIP
    graph capture (the side stream joins the capture through the event waits).  AUDIOCAPTION_TRAIN_SIDE_STREAM=0: off."""

    def __init__(self):
        self.enabled = os.environ.get("AUDIOCAPTION_TRAIN_SIDE_STREAM", "1") != "0"
        self.stream = None
        self.dirty = False

    def fork(self):
        cur = torch.cuda.current_stream()
        if not self.enabled or GEMM_HOOK is not None:   # (the per-launch timing hook records events on the current stream)
            return cur.cuda_stream
        if self.stream is None or self.stream.device != cur.device:
            self.stream = torch.cuda.Stream(cur.device)
        self.stream.wait_stream(cur)
        self.dirty = True
        return self.stream.cuda_stream

    def join(self):
        if self.dirty:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.dirty = False


_WS_POISON = os.environ.get("AUDIOCAPTION_WS_POISON", "0") == "1"
_DX_SPLITK = os.environ.get("AUDIOCAPTION_TRAIN_DX_SPLITK", "1") != "0"   # development: 0 = input gradients never split-K


class _Ws:
    """Named device buffers shared by every batch shape of an engine: a buffer is re-allocated only when a shape needs
    more than it holds, and ``gen`` counts those re-allocations (captured graphs hold raw addresses and are re-captured
    when it moves)."""

    def __init__(self, device):
        self.device = device
        self.t = {}
        self.gen = 0

    def f(self, name, *shape):
        n = 1
        for s in shape:
            n *= int(s)
        b = self.t.get(name)
        if b is None or b.numel() < n:
            b = torch.empty(max(n, 1), device=self.device, dtype=torch.float32)
            if _WS_POISON:
                b.fill_(float("nan"))      # development aid: a read of a cell no kernel wrote shows up as NaN
            self.t[name] = b
            self.gen += 1
        return b.data_ptr()

    def i(self, name, n):
        b = self.t.get(name)
        if b is None or b.numel() < n:
            b = torch.empty(max(n, 1), device=self.device, dtype=torch.int32)
            self.t[name] = b
            self.gen += 1
        return b.data_ptr()

    def tensor(self, name):
        return self.t[name]


class TrainEngine:

    def __init__(self, model, seed=0):
        from .crnn_trm_encoder import CrnnEncoder
        if not isinstance(model.encoder, CrnnEncoder) or not model.encoder.freeze_cnn_bn:
            raise NotImplementedError(
                "TrainEngine: built for the reference's training recipe, CrnnEncoder(freeze_cnn=True, "
                "freeze_cnn_bn=True) (cnn14rnn_trm.yaml:9-13); the backward through the Cnn14 is not built")
        dec = model.decoder
        if dec.d_model != D or dec.nhead * 64 != D or model.encoder.rnn.hidden_size != H:
            raise NotImplementedError("TrainEngine: d_model 256 / head_dim 64 / GRU hidden 256 only")
        self.model = model
        self.lib = _lib.load()
        self.flat = None
        self._wsg = None
        self.seed = int(seed)          # bumped after every forward (fresh dropout masks per step)
        self._seed_ptr = None
        self._saved = None
        self._states = {}
        self._phase = "forward"
        # GEMMs of the step: "bf16x3" = split-bf16 operands on the bf16 MFMA for the large products (the backward over
        # all rows, teacher forcing), "f32" = exact f32 MFMA everywhere
        # "pw" (default) = "bf16x3" with the products against a WEIGHT matrix (x W^T and dy W, two thirds of the GEMM time)
        # on ac_pw_gemm_bf16x3: weights re-split into MFMA fragment order once per iteration (one launch for the whole
        # table), activations split once per 256 output columns; the same three-product arithmetic, bit-identical results
        self.gemm_algo = os.environ.get("AUDIOCAPTION_TRAIN_GEMM", "pw")
        self._pw = {}          # (weight address, N, K, transposed) -> packed-fragment tensor
        self._pw_table = None  # device copy of the pack records (ac_pw_gemm_pack_table)
        self._pw_hold = []     # tables captured graphs may still reference
        self.gru_algo = os.environ.get("AUDIOCAPTION_GRU_ALGO", "split")   # forward recurrence kernel (see RnnEncoder)

    # ---- small launch helpers (raw addresses; s = stream handle) ------------------------------------------
    def _gemm(self, s, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias=None, relu=0, beta=0.0, splitk=1, drop_p=0.0,
              seed=0, row0=0):
        hook = GEMM_HOOK
        if hook is not None:
            info = {"phase": self._phase, "M": M, "N": N, "K": K, "flops": 2.0 * M * N * K}
            hook("pre", info)
        if self.gemm_algo in ("bf16x3", "pw"):
            # split-bf16 operands (2^-16), f32 accumulation; small or unaligned products fall through to exact f32
            check(self.lib.ac_gemm_bf16x3(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias, relu, beta, splitk, drop_p, seed,
                                          self._seed_ptr, row0, None, 0, s), "ac_gemm_bf16x3")
        else:
            check(self.lib.ac_gemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias, relu, beta, splitk, drop_p, seed,
                                   self._seed_ptr, row0, None, 0, s), "ac_gemm")
        if hook is not None:
            hook("post", info)

    # ---- products against a weight matrix on the activation-stationary kernel (csrc/pw_gemm.hip) ----
    def _pw_frag(self, s, W, N, K, transposed):
        """Packed fragments of W[N][K] (transposed: of W^T, for dy W).  A new layer is packed on the spot and joins the table
        that ``_pw_pack_all`` repacks at the start of every iteration."""
        key = (W, N, K, transposed)
        frag = self._pw.get(key)
        if frag is None:
            n, k = (K, N) if transposed else (N, K)
            frag = torch.empty(self.lib.ac_pw_gemm_packed_bytes(n, k), device=self.flat.flat.device, dtype=torch.uint8)
            self._pw[key] = frag
            self._pw_table = None
            s_n, s_k = (1, K) if transposed else (K, 1)
            check(self.lib.ac_pw_gemm_pack_strided(W, s_n, s_k, frag.data_ptr(), n, k, s), "ac_pw_gemm_pack_strided")
        return frag.data_ptr()

    def _pw_pack_all(self, s):
        """Re-split every registered weight (they changed in the optimiser step): one launch over a device-resident table."""
        if self.gemm_algo != "pw" or not self._pw:
            return
        self._pw_build()
        table, count = self._pw_table
        check(self.lib.ac_pw_gemm_pack_table(table.data_ptr(), count, s), "ac_pw_gemm_pack_table")

    def _pw_build(self):
        """The device-resident table of pack records; built outside any graph capture (it is a host upload)."""
        if self._pw_table is None and self._pw:
            import struct
            rec = b""
            for (W, N, K, transposed), frag in self._pw.items():
                n, k = (K, N) if transposed else (N, K)
                s_n, s_k = (1, K) if transposed else (K, 1)
                rec += struct.pack("<QQqqii", W, frag.data_ptr(), s_n, s_k, n, k)
            table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(self.flat.flat.device)
            self._pw_table = (table, len(self._pw))
            self._pw_hold.append(table)     # a captured graph may still replay an older table

    @staticmethod
    def _pw_ok(M, N, K, ldx, ldy, *ptrs):
        # enough 32-row x 256-column workgroups to occupy the chip: the few-row products of the free-running passes stay
        # on the small-tile exact-f32 kernels ac_gemm_bf16x3 forwards them to
        return (float(M) * N * K >= 3.0e7 and ((M + 31) // 32) * ((N + 255) // 256) >= 128 and N % 4 == 0 and K % 4 == 0 and
                ldx % 4 == 0 and ldy % 4 == 0 and all(q is None or q % 16 == 0 for q in ptrs))

    def _lin(self, s, x, W, b, y, M, N, K, ldx=None, ldy=None, relu=0, drop_p=0.0, seed=0, row0=0):
        """y[M][N] = x[M][K] W[N][K]^T + b"""
        ldx, ldy = ldx or K, ldy or N
        if self.gemm_algo == "pw" and self._pw_ok(M, N, K, ldx, ldy, x, W, b, y):
            hook = GEMM_HOOK
            if hook is not None:
                info = {"phase": self._phase, "M": M, "N": N, "K": K, "flops": 2.0 * M * N * K}
                hook("pre", info)
            check(self.lib.ac_pw_gemm_bf16x3_ex(x, ldx, self._pw_frag(s, W, N, K, False), b, y, ldy, M, N, K, relu, 0.0, None,
                                                0, drop_p, seed, self._seed_ptr, row0, s), "ac_pw_gemm_bf16x3_ex")
            if hook is not None:
                hook("post", info)
            return
        self._gemm(s, x, ldx, 1, W, 1, K, y, ldy, M, N, K, b, relu, 0.0, 1, drop_p, seed, row0)

    def _lin_dx(self, s, dy, W, dx, M, N, K, beta=0.0, lddy=None, lddx=None):
        """dx[M][K] (+)= dy[M][N] W[N][K]"""
        lddy, lddx = lddy or N, lddx or K
        if self.gemm_algo == "pw" and self._pw_ok(M, K, N, lddy, lddx, dy, W, dx):
            hook = GEMM_HOOK
            if hook is not None:
                info = {"phase": self._phase, "M": M, "N": K, "K": N, "flops": 2.0 * M * N * K}
                hook("pre", info)
            check(self.lib.ac_pw_gemm_bf16x3_ex(dy, lddy, self._pw_frag(s, W, N, K, True), None, dx, lddx, M, K, N, 0, beta,
                                                None, 0, 0.0, 0, None, 0, s), "ac_pw_gemm_bf16x3_ex")
            if hook is not None:
                hook("post", info)
            return
        # few output tiles over a long reduction (the classifier's dx: 672 x 256 outputs over 4981 vocabulary entries; the
        # GRU's dx: 992 x 512 over 1536 gate columns): slices of the reduction on separate workgroups (atomic accumulation,
        # like the weight gradients) instead of 44 / 128 workgroups walking all of it
        tiles = ((M + 63) // 64) * ((K + 63) // 64)
        sk = max(1, min((255 + tiles) // tiles, N // 256)) if tiles < 200 and beta in (0.0, 1.0) and _DX_SPLITK else 1
        self._gemm(s, dy, lddy, 1, W, K, 1, dx, lddx, M, K, N, None, 0, beta, sk)

    @staticmethod
    def _splitk(M, N, K):
        blocks = ((M + 63) // 64) * ((N + 63) // 64)
        sk = max(1, min(512 // max(blocks, 1), K // 128))
        return sk

    def _lin_dw(self, s, dy, lddy, x, ldx, dW, rows, N, K):
        """dW[N][K] += dy[rows][N]^T x[rows][K]"""
        sk = self._splitk(N, K, rows)
        self._gemm(s, dy, 1, lddy, x, ldx, 1, dW, K, N, K, rows, None, 0, 1.0, sk)

    def _colsum(self, s, x, ld, out, M, N):
        check(self.lib.ac_colsum(x, ld, out, M, N, s), "ac_colsum")

    # ---------------------------------------------------------------------------------------------------
    def _ensure(self, device):
        if self.flat is None or not self.flat.intact() or self.flat.flat.device != device:
            self.flat = FlatParams(self.model)
            self._states = {}
            self._pw, self._pw_table = {}, None
            _lib.bump_param_generation(self.flat.params)

    @staticmethod
    def _layout(N, T, Tm, teacher_forcing, device):
        """Static index tables of the row space."""
        if teacher_forcing:
            passes = [(T, 0)]
        else:
            passes, off = [], 0
            for t in range(T):
                passes.append((t + 1, off))
                off += N * (t + 1)
        R = sum(N * L for L, _ in passes)
        S = len(passes) * N
        pos = torch.empty(R, dtype=torch.int32)
        qrow0 = torch.empty(S, dtype=torch.int32)
        qlen = torch.empty(S, dtype=torch.int32)
        cls_rows = torch.empty(N * T, dtype=torch.int32)
        for t, (L, off) in enumerate(passes):
            pos[off:off + N * L] = torch.arange(L, dtype=torch.int32).repeat(N)
            qrow0[t * N:(t + 1) * N] = off + torch.arange(N, dtype=torch.int32) * L
            qlen[t * N:(t + 1) * N] = L
            if teacher_forcing:
                cls_rows[:] = torch.arange(N * T, dtype=torch.int32)
            else:
                cls_rows[t::T] = off + torch.arange(N, dtype=torch.int32) * L + t
        lay = {"passes": passes, "R": R, "S": S, "ones": torch.ones(max(T, 1), dtype=torch.int32, device=device)}
        for k, v in (("pos", pos), ("qrow0", qrow0), ("qlen", qlen), ("cls_rows", cls_rows),
                     ("mrow0", torch.arange(S, dtype=torch.int32) * Tm), ("mklen", torch.full((S,), Tm, dtype=torch.int32))):
            lay[k] = v.to(device)
        return lay

    # ---- host side of one iteration: shapes, static input buffers, scheduled-sampling draws, seed -------------
    def _prepare(self, input_dict):
        model = self.model
        enc, dec = model.encoder, model.decoder
        if not model.training:
            raise RuntimeError("TrainEngine needs model.train() (run.py:79)")
        specaug = bool(input_dict.get("specaug", False)) and enc.cnn.training
        wav = input_dict["wav"]
        dev = wav.device
        if not wav.is_cuda:
            raise _lib.HipLibraryError("the training step needs tensors on a ROCm device; there is no CPU fallback")
        self._ensure(dev)
        cap = input_dict["cap"]
        N, Tc = cap.shape
        T = Tc - 1
        ss_ratio = input_dict["ss_ratio"]
        teacher_forcing = ss_ratio == 1
        hook = input_dict.get("_cnn_attn")    # parity-test hook: start downstream of the (un-pinned) mel front-end
        if hook is None:
            Tq = enc.cnn.geometry(wav.shape[1])[1][5]
        else:
            Tq = hook.shape[1]
        p_dec = float(dec.in_dropout.p)
        p_rnn = float(enc.rnn.network.dropout)
        p_cnn = 0.2 if enc.cnn.training else 0.0
        key = (dev, N, Tc, tuple(wav.shape) if hook is None else ("hook", Tq), teacher_forcing, p_dec, p_rnn, p_cnn,
               model.start_idx, model.pad_idx, specaug and hook is None)
        st = self._states.get(key)
        if st is None:
            lay = self._layout(N, T, Tq, teacher_forcing, dev)
            S = lay["S"]
            if self._wsg is None or self._wsg.device != dev:
                self._wsg = _Ws(dev)
            st = {"key": key, "ws": self._wsg, "lay": lay, "N": N, "T": T, "Tc": Tc, "Tq": Tq, "teacher_forcing": teacher_forcing,
                  "p_dec": p_dec, "p_rnn": p_rnn, "p_cnn": p_cnn, "graphs": {}, "steps": 0,
                  "wav": torch.empty_like(wav, dtype=torch.float32) if hook is None else None,
                  "cnn_attn_in": torch.empty(N, Tq, 2048, device=dev) if hook is not None else None,
                  "cap": torch.empty(N, Tc, device=dev, dtype=torch.int64),
                  "specaug": torch.zeros(N, 4, 2, device=dev, dtype=torch.int32) if (specaug and hook is None) else None,
                  "specaug_pinned": [],
                  # one pinned staging block -> one device block (int32 words):
                  #   seed (int64) | lens [N] | tgt_len [N] | use_cap [T] | mvalid [S]
                  # a ring of staging blocks, each guarded by an event, because the host runs ahead of the device
                  "ring": [(torch.zeros(2 + 2 * N + max(T, 1) + S, dtype=torch.int32).pin_memory(), torch.cuda.Event())
                           for _ in range(4)],
                  "small": torch.zeros(2 + 2 * N + max(T, 1) + S, device=dev, dtype=torch.int32)}
            self._states[key] = st
            # real batches are padded to their longest clip / caption: shapes keep changing, so the per-shape states
            # (index tables, static inputs, a captured graph) are kept for the 16 most recently used shapes only
            while len(self._states) > 16:
                self._states.pop(next(iter(self._states)))
        else:
            self._states[key] = self._states.pop(key)     # most recently used last
        if wav.shape[0] != N:
            raise ValueError("cap and wav batch sizes differ")
        lens = cnn14_feat_len(input_dict["wav_len"], enc.cnn.hop_length, enc.cnn.downsample_ratio)
        if int(lens.min()) < 1 or int(lens.max()) > Tq:
            raise ValueError("attn_len must lie in [1, attn.size(1)]")
        # one draw per step, exactly the reference's call pattern (transformer_model.py:44)
        use_cap = [1] * T if teacher_forcing else [int(random.random() < ss_ratio) for _ in range(T)]
        use_cap = input_dict.get("_use_cap", use_cap)
        # passes whose prefix is the model's own prediction (pass 0 always starts from <start> alone)
        st["free_ts"] = [] if teacher_forcing else [t for t in range(1, T) if not int(use_cap[t])]
        S = st["lay"]["S"]
        base_seed = int(input_dict.get("dropout_seed", self.seed))
        self.seed = base_seed + 1
        h, ev = st["ring"][st["steps"] % len(st["ring"])]
        ev.synchronize()                       # the copy that last used this staging block has completed
        h[:2].view(torch.int64)[0] = base_seed
        h[2:2 + N] = lens.to(torch.int32)
        if "cap_len" in input_dict:
            h[2 + N:2 + 2 * N] = (torch.as_tensor(input_dict["cap_len"]).to(torch.int32) - 1).clamp(max=T)
        h[2 + 2 * N:2 + 2 * N + T] = torch.as_tensor(use_cap, dtype=torch.int32)
        h[2 + 2 * N + max(T, 1):] = lens.to(torch.int32).repeat(S // N)
        st["small"].copy_(h, non_blocking=True)
        ev.record()
        if st["specaug"] is not None:
            # SpecAugment stripes of this iteration (torchlibrosa draws them with torch's generator; here a numpy stream
            # seeded by the dropout seed): pinned staging kept alive for a few iterations, the host runs ahead
            from .kernels import specaug_stripes
            T_frames = wav.shape[1] // enc.cnn.hop_length + 1
            host = torch.from_numpy(specaug_stripes((base_seed << 16) + OP_SPECAUG, N, T_frames)).pin_memory()
            st["specaug_pinned"] = (st["specaug_pinned"] + [host])[-4:]
            st["specaug"].copy_(host, non_blocking=True)
        if hook is None:
            st["wav"].copy_(wav, non_blocking=True)
        else:
            st["cnn_attn_in"].copy_(hook, non_blocking=True)
        st["cap"].copy_(cap, non_blocking=True)
        st["lens_host"] = lens
        return st

    # ---- forward launches (no host synchronisation, capturable) ---------------------------------------------------
    def _launch_forward(self, st, free=True):
        model, lib, fp = self.model, self.lib, self.flat
        enc, dec = model.encoder, model.decoder
        s = _lib.stream()
        self._phase = "forward"
        ws, lay = st["ws"], st["lay"]
        N, T, Tc, Tq = st["N"], st["T"], st["Tc"], st["Tq"]
        p_dec, p_rnn, p_cnn = st["p_dec"], st["p_rnn"], st["p_cnn"]
        teacher_forcing = st["teacher_forcing"]
        small = st["small"].data_ptr()
        self._seed_ptr = small
        self._pw_pack_all(s)        # the optimiser moved the weights since the last iteration
        lens_p, ucap, mvalid = small + 8, small + 4 * (2 + 2 * N), small + 4 * (2 + 2 * N + max(T, 1))
        R, S, passes = lay["R"], lay["S"], lay["passes"]
        NP = len(passes)
        Tm = Tq
        B = N
        rows_g = B * Tq
        # frozen Cnn14 (train mode: dropout after every block)
        if st["cnn_attn_in"] is not None:
            cnn_attn = st["cnn_attn_in"]
        else:
            cnn_attn = enc.cnn.encode(st["wav"], dropout=(p_cnn, OP_CNN_BLOCK, self._seed_ptr) if p_cnn > 0 else None,
                                      specaug=st["specaug"], train=True)
        st["cnn_attn"] = cnn_attn
        Cin = cnn_attn.shape[2]

        # ---- GRU, saving the gates -------------------------------------------------------------------
        nl = enc.rnn.num_layers
        x_in = cnn_attn.data_ptr()
        in_dim = Cin
        gru = []
        pre = "encoder.rnn.network."
        for l in range(nl):
            w_ih, w_hh = fp.p(f"{pre}weight_ih_l{l}"), fp.p(f"{pre}weight_hh_l{l}")
            b_ih, b_hh = fp.p(f"{pre}bias_ih_l{l}"), fp.p(f"{pre}bias_hh_l{l}")
            gx = ws.f(f"gx{l}", rows_g, 6 * H)
            self._lin(s, x_in, w_ih, b_ih, gx, rows_g, 6 * H, in_dim)
            out = ws.f(f"gru_out{l}", rows_g, 2 * H)
            save = ws.f(f"gru_save{l}", rows_g, 2 * 4 * H)
            if self.gru_algo == "split":
                # four workgroups per (clip, direction), W_hh register resident, straight from the flat parameter buffer
                xch = ws.f("gru_xch", (lib.ac_gru_split_workspace_bytes(B) + 3) // 4)
                if not st.get("gru_xch_zeroed"):
                    ws.tensor("gru_xch")[:1].zero_()       # the sticky error word (first word of the workspace)
                    st["gru_xch_zeroed"] = True
                check(lib.ac_gru_layer_split(gx, w_hh, b_hh, lens_p, out, save, xch, B, Tq, H, s), "ac_gru_layer_split")
            else:
                whhT = ws.f(f"whhT{l}", 2 * 3 * H * H)
                check(lib.ac_gru_pack_whh(w_hh, whhT, H, s), "ac_gru_pack_whh")
                check(lib.ac_gru_layer_train(gx, whhT, b_hh, lens_p, out, save, B, Tq, H, s), "ac_gru_layer_train")
            nxt = out
            if l < nl - 1 and p_rnn > 0:
                nxt = ws.f(f"gru_drop{l}", rows_g, 2 * H)
                check(lib.ac_dropout(out, nxt, rows_g * 2 * H, p_rnn, OP_GRU_LAYER + l, self._seed_ptr, 0, s),
                      "ac_dropout")
            gru.append({"x": x_in, "in_dim": in_dim, "out": out})
            x_in, in_dim = nxt, 2 * H
        # attn_emb (B, Tq, 512).  The reference truncates it to the longest clip (pad_packed_sequence); the frames
        # beyond a clip's length are zero and masked as keys, so keeping all Tq frames changes no result and keeps
        # the shapes static.
        attn_emb = x_in
        A = 2 * H

        # ---- decoder: audio memory (shared by all passes up to its dropout mask) ---------------------
        dp = "decoder."
        nlay = dec.nlayers
        V = dec.vocab_size
        F = dec.dim_feedforward
        rows_m = N * Tm
        Rm = NP * rows_m
        mem_a = ws.f("mem_a", rows_m, D)
        self._lin(s, attn_emb, fp.p(dp + "attn_proj.0.weight"), fp.p(dp + "attn_proj.0.bias"), mem_a, rows_m, D, A, relu=1)
        mem_pre, mem = ws.f("mem_pre", Rm, D), ws.f("mem", Rm, D)
        check(lib.ac_dropadd_ln_fwd(mem_a, None, fp.p(dp + "attn_proj.3.weight"), fp.p(dp + "attn_proj.3.bias"), mem_pre,
                                    mem, 0, Rm, rows_m, D, p_dec, OP_MEM, self._seed_ptr, 1e-5, s), "ac_dropadd_ln_fwd")
        kv = []
        for l in range(nlay):
            lp = f"{dp}model.layers.{l}."
            kvb = ws.f(f"kv{l}", Rm, 2 * D)
            self._lin(s, mem, fp.p(lp + "multihead_attn.in_proj_weight") + 4 * D * D,
                      fp.p(lp + "multihead_attn.in_proj_bias") + 4 * D, kvb, Rm, 2 * D, D)
            kv.append(kvb)

        # ---- the passes ------------------------------------------------------------------------------
        word = ws.i("word", R)
        seq = ws.i("seq", N * T)
        logit = ws.f("logit", N * T, V)
        pos, qrow0, qlen = lay["pos"].data_ptr(), lay["qrow0"].data_ptr(), lay["qlen"].data_ptr()
        mrow0, mklen = lay["mrow0"].data_ptr(), lay["mklen"].data_ptr()
        emb, pe, cls = fp.p(dp + "word_embedding.weight"), dec.pos_encoder.pe.data_ptr(), fp.p(dp + "classifier.weight")
        nh = dec.nhead
        P1 = [ws.f(f"P1_{l}", S * nh * T * T) for l in range(nlay)]     # attention probabilities, kept per layer
        P2 = [ws.f(f"P2_{l}", S * nh * T * Tm) for l in range(nlay)]
        x0 = ws.f("x_l0", R, D)
        acts = []
        for l in range(nlay):
            a = {k: ws.f(f"{k}{l}", R, w) for k, w in (("qkv", 3 * D), ("ctx1", D), ("sa", D), ("pre1", D), ("x1", D),
                                                       ("q2", D), ("ctx2", D), ("ca", D), ("pre2", D), ("x2", D),
                                                       ("hdn", F), ("ff", D), ("pre3", D))}
            a["x3"] = ws.f(f"x_l{l + 1}", R, D)
            acts.append(a)
        st.update(V=V, F=F, gru=gru, kv=kv, acts=acts, x0=x0, word=word, P1=P1, P2=P2, mem=mem, mem_pre=mem_pre,
                  mem_a=mem_a, attn_emb=attn_emb)
        st["ctx"] = dict(seq=seq, logit=logit, pos=pos, qrow0=qrow0, qlen=qlen, mrow0=mrow0, mklen=mklen, emb=emb, pe=pe,
                         cls=cls, ucap=ucap, mvalid=mvalid, Tm=Tm)
        # The reference runs T sequential decoder passes (pass t on the prefix of length t + 1).  Which passes are
        # teacher forced is known on the host (the scheduled-sampling draws), and a teacher-forced pass depends on nothing
        # the model produced - so ALL passes first run teacher forced as ONE batch over the whole row space (26 launches
        # over 7 392 rows instead of 21 x 26 over 32 ... 704), and only the free-running passes (15 % at ss_ratio 0.85)
        # are then re-run in order on their own predictions, overwriting their rows.  Same results: a pass reads earlier
        # passes only through `seq`, and the dropout masks are a function of (seed, row), not of the launch.
        if teacher_forcing:
            self._decoder_passes(st, 0, 1, ucap)
        else:
            self._decoder_passes(st, 0, NP, lay["ones"].data_ptr())
            if free:
                for t in st["free_ts"]:
                    self._decoder_passes(st, t, t + 1, ucap)

    def _decoder_passes(self, st, ta, tb, ucap_ptr):
        """Decoder passes [ta, tb) of the row space as one batch: prefixes (teacher tokens where ``ucap_ptr[t]`` is set,
        else the model's own earlier predictions), embedding, the decoder layers, the classifier on the last position of
        every sequence -> logit[:, t], arg-max -> seq[:, t]."""
        model, lib, fp = self.model, self.lib, self.flat
        dec = model.decoder
        s = _lib.stream()
        self._phase = "forward"
        ws, lay, c = st["ws"], st["lay"], st["ctx"]
        N, T, Tc = st["N"], st["T"], st["Tc"]
        V, F, acts, kv, P1, P2, x0, word = st["V"], st["F"], st["acts"], st["kv"], st["P1"], st["P2"], st["x0"], st["word"]
        p_dec, teacher_forcing = st["p_dec"], st["teacher_forcing"]
        passes = lay["passes"]
        NP = len(passes)
        nlay, nh, Tm = dec.nlayers, dec.nhead, c["Tm"]
        dp = "decoder."
        off = passes[ta][1]
        nr = sum(N * passes[t][0] for t in range(ta, tb))
        lmax = max(passes[t][0] for t in range(ta, tb))
        seq0, nseq = ta * N, (tb - ta) * N
        o4 = 4 * off * D
        cap_p = st["cap"].data_ptr()
        for t in range(ta, tb):
            L, off_t = passes[t]
            check(lib.ac_build_prefix(cap_p, Tc, c["seq"], T, ucap_ptr, 0 if teacher_forcing else t, model.start_idx, word,
                                      off_t, N, L, s), "ac_build_prefix")
        check(lib.ac_embed_fwd(c["emb"], c["pe"], word, c["pos"], x0, off, nr, D, p_dec, OP_EMB_A, p_dec, OP_EMB_B,
                               self._seed_ptr, s), "ac_embed_fwd")
        qrow0, qlen, mrow0, mklen, mvalid = c["qrow0"], c["qlen"], c["mrow0"], c["mklen"], c["mvalid"]
        x = x0
        for l in range(nlay):
            lp = f"{dp}model.layers.{l}."
            a = acts[l]
            op = OP_LAYER + 10 * l
            qkv = a["qkv"] + 4 * off * 3 * D
            self._lin(s, x + o4, fp.p(lp + "self_attn.in_proj_weight"), fp.p(lp + "self_attn.in_proj_bias"), qkv, nr,
                      3 * D, D)
            check(lib.ac_attn_seq_fwd(a["qkv"], 3 * D, a["qkv"] + 4 * D, 3 * D, a["qkv"] + 8 * D, 3 * D, a["ctx1"], D,
                                      P1[l], T, T, qrow0, qlen, qrow0, qlen, None, word, model.pad_idx, 1, seq0, nseq, nh,
                                      64, lmax, lmax, p_dec, op + 0, self._seed_ptr, s), "ac_attn_seq_fwd")
            self._lin(s, a["ctx1"] + o4, fp.p(lp + "self_attn.out_proj.weight"), fp.p(lp + "self_attn.out_proj.bias"),
                      a["sa"] + o4, nr, D, D)
            check(lib.ac_dropadd_ln_fwd(a["sa"], x, fp.p(lp + "norm1.weight"), fp.p(lp + "norm1.bias"), a["pre1"],
                                        a["x1"], off, nr, 0, D, p_dec, op + 1, self._seed_ptr, 1e-5, s), "ln1")
            self._lin(s, a["x1"] + o4, fp.p(lp + "multihead_attn.in_proj_weight"),
                      fp.p(lp + "multihead_attn.in_proj_bias"), a["q2"] + o4, nr, D, D)
            check(lib.ac_attn_seq_fwd(a["q2"], D, kv[l], 2 * D, kv[l] + 4 * D, 2 * D, a["ctx2"], D, P2[l], T, Tm, qrow0,
                                      qlen, mrow0, mklen, mvalid, None, 0, 0, seq0, nseq, nh, 64, lmax, Tm, p_dec, op + 2,
                                      self._seed_ptr, s), "ac_attn_seq_fwd(cross)")
            self._lin(s, a["ctx2"] + o4, fp.p(lp + "multihead_attn.out_proj.weight"),
                      fp.p(lp + "multihead_attn.out_proj.bias"), a["ca"] + o4, nr, D, D)
            check(lib.ac_dropadd_ln_fwd(a["ca"], a["x1"], fp.p(lp + "norm2.weight"), fp.p(lp + "norm2.bias"),
                                        a["pre2"], a["x2"], off, nr, 0, D, p_dec, op + 3, self._seed_ptr, 1e-5, s),
                  "ln2")
            self._lin(s, a["x2"] + o4, fp.p(lp + "linear1.weight"), fp.p(lp + "linear1.bias"),
                      a["hdn"] + 4 * off * F, nr, F, D, relu=1, drop_p=p_dec, seed=op + 4, row0=off)
            self._lin(s, a["hdn"] + 4 * off * F, fp.p(lp + "linear2.weight"), fp.p(lp + "linear2.bias"), a["ff"] + o4,
                      nr, D, F)
            check(lib.ac_dropadd_ln_fwd(a["ff"], a["x2"], fp.p(lp + "norm3.weight"), fp.p(lp + "norm3.bias"),
                                        a["pre3"], a["x3"], off, nr, 0, D, p_dec, op + 5, self._seed_ptr, 1e-5, s),
                  "ln3")
            x = a["x3"]
        logit, seq, cls = c["logit"], c["seq"], c["cls"]
        if teacher_forcing:
            self._lin(s, x, cls, None, logit, N * T, V, D)
        elif tb - ta == NP:
            # every pass at once: the last position of every sequence, gathered in (clip, step) order = logit's layout
            xlast = ws.f("xlast", N * T, D)
            check(lib.ac_gather_rows(x, lay["cls_rows"].data_ptr(), xlast, N * T, D, s), "ac_gather_rows")
            self._lin(s, xlast, cls, None, logit, N * T, V, D)
            check(lib.ac_argmax_rows(logit, V, N * T, V, seq, 1, s), "ac_argmax_rows")
        else:
            for t in range(ta, tb):
                L, off_t = passes[t]
                # classifier on the last position of every sequence of this pass -> logit[:, t]
                self._lin(s, x + 4 * off_t * D + 4 * t * D, cls, None, logit + 4 * t * V, N, V, D, ldx=L * D, ldy=T * V)
                check(lib.ac_argmax_rows(logit + 4 * t * V, T * V, N, V, seq + 4 * t, T, s), "ac_argmax_rows")

    def _outputs(self, st):
        N, T, V = st["N"], st["T"], st["V"]
        out = {"logit": st["ws"].tensor("logit")[:N * T * V].view(N, T, V), "attn_emb_len": st["lens_host"]}
        if not st["teacher_forcing"]:
            out["seq"] = st["ws"].tensor("seq")[:N * T].view(N, T)
        return out

    def forward(self, input_dict):
        """The reference's ``model(input_dict)`` for mode "train": returns ``logit`` (N, T, V) [+ ``seq``] as fresh
        device tensors and keeps what the backward needs."""
        st = self._prepare(input_dict)
        self._launch_forward(st)
        self._saved = st
        out = self._outputs(st)
        out["logit"] = out["logit"].clone()
        if "seq" in out:
            out["seq"] = out["seq"].to(torch.int64)
        return out

    # ---- backward -----------------------------------------------------------------------------------------
    def backward(self, dlogit):
        """d(loss)/d(parameters) of the last forward into the (zeroed) flat gradient buffer, given d(loss)/d(logit)
        (N, T, V).  The kernels accumulate (split-K atomics, bias column sums), so the buffer is cleared first."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("TrainEngine.backward without a forward")
        self._saved = None
        dlogit = dlogit.contiguous()
        if dlogit.dtype != torch.float32:
            dlogit = dlogit.float()
        self._launch_backward(sv, dlogit.data_ptr())

    def _launch_backward(self, sv, dl, part="all"):
        """``part``: "all", or the two halves of it - "head" (classifier, decoder layers, audio memory, attn_proj: every
        decoder gradient is final after it, and d(loss)/d(GRU output) is in the workspace) and "gru" (the backward through
        time of the three GRU layers) - so that the all-reduce of the decoder's gradients can run under the GRU's."""
        if part == "gru":
            return self._launch_backward_gru(sv)
        model, lib, fp = self.model, self.lib, self.flat
        enc, dec = model.encoder, model.decoder
        s = _lib.stream()
        self._phase = "backward"
        ws, lay = sv["ws"], sv["lay"]
        N, T, Tq, V, F = sv["N"], sv["T"], sv["Tq"], sv["V"], sv["F"]
        B, Tm = N, Tq
        p_dec, p_rnn = sv["p_dec"], sv["p_rnn"]
        R, S = lay["R"], lay["S"]
        NP = len(lay["passes"])
        rows_m = N * Tm
        Rm = NP * rows_m
        nh = dec.nhead
        nlay = dec.nlayers
        dp = "decoder."
        fp.grad.zero_()
        self._seed_ptr = sv["small"].data_ptr()
        side = self._side = getattr(self, "_side", None) or _SideStream()
        qrow0, qlen = lay["qrow0"].data_ptr(), lay["qlen"].data_ptr()
        mrow0, mklen = lay["mrow0"].data_ptr(), lay["mklen"].data_ptr()
        cls_rows = lay["cls_rows"].data_ptr()
        NT = N * T

        # ---- classifier --------------------------------------------------------------------------------
        xtop = sv["acts"][-1]["x3"]
        xlast = ws.f("xlast", NT, D)
        check(lib.ac_gather_rows(xtop, cls_rows, xlast, NT, D, s), "ac_gather_rows")
        self._lin_dw(side.fork(), dl, V, xlast, D, fp.g(dp + "classifier.weight"), NT, V, D)
        dxlast = ws.f("dxlast", NT, D)
        self._lin_dx(s, dl, fp.p(dp + "classifier.weight"), dxlast, NT, V, D)
        dx = ws.f("dx_a", R, D)
        dres = ws.f("dx_b", R, D)
        ws.tensor("dx_a")[:R * D].zero_()
        check(lib.ac_scatter_add_rows(dxlast, cls_rows, dx, NT, D, s), "ac_scatter_add_rows")
        dsub = ws.f("dsub", R, D)
        dhdn = ws.f("dhdn", R, F)
        dctx = ws.f("dctx", R, D)
        dq2 = ws.f("dq2", R, D)
        dqkv = ws.f("dqkv", R, 3 * D)
        dmem = ws.f("dmem", Rm, D)
        dkv = ws.f("dkv", Rm, 2 * D)
        scale = 1.0 / (1.0 - p_dec) if p_dec > 0 else 1.0
        for l in reversed(range(nlay)):
            lp = f"{dp}model.layers.{l}."
            a = sv["acts"][l]
            op = OP_LAYER + 10 * l
            x_in = sv["x0"] if l == 0 else sv["acts"][l - 1]["x3"]
            # norm3 / feed-forward
            side.join()   # dsub (and, a layer on, dhdn / dq2 / dkv / dqkv) are about to be written again
            check(lib.ac_dropadd_ln_bwd(dx, a["pre3"], fp.p(lp + "norm3.weight"), dsub, dres, 0, None, 0,
                                        fp.g(lp + "norm3.weight"), fp.g(lp + "norm3.bias"), R, D, p_dec, op + 5,
                                        self._seed_ptr, 1e-5, s), "ln3 bwd")
            s2 = side.fork()
            self._lin_dw(s2, dsub, D, a["hdn"], F, fp.g(lp + "linear2.weight"), R, D, F)
            self._colsum(s2, dsub, D, fp.g(lp + "linear2.bias"), R, D)
            self._lin_dx(s, dsub, fp.p(lp + "linear2.weight"), dhdn, R, D, F)
            check(lib.ac_mask_pos_scale(dhdn, a["hdn"], R * F, scale, s), "ac_mask_pos_scale")
            s2 = side.fork()
            self._lin_dw(s2, dhdn, F, a["x2"], D, fp.g(lp + "linear1.weight"), R, F, D)
            self._colsum(s2, dhdn, F, fp.g(lp + "linear1.bias"), R, F)
            self._lin_dx(s, dhdn, fp.p(lp + "linear1.weight"), dres, R, F, D, beta=1.0)
            dx, dres = dres, dx                                   # dx = d(x2)
            # norm2 / cross attention
            side.join()
            check(lib.ac_dropadd_ln_bwd(dx, a["pre2"], fp.p(lp + "norm2.weight"), dsub, dres, 0, None, 0,
                                        fp.g(lp + "norm2.weight"), fp.g(lp + "norm2.bias"), R, D, p_dec, op + 3,
                                        self._seed_ptr, 1e-5, s), "ln2 bwd")
            s2 = side.fork()
            self._lin_dw(s2, dsub, D, a["ctx2"], D, fp.g(lp + "multihead_attn.out_proj.weight"), R, D, D)
            self._colsum(s2, dsub, D, fp.g(lp + "multihead_attn.out_proj.bias"), R, D)
            self._lin_dx(s, dsub, fp.p(lp + "multihead_attn.out_proj.weight"), dctx, R, D, D)
            kvl = sv["kv"][l]
            check(lib.ac_attn_seq_bwd(a["q2"], D, kvl, 2 * D, kvl + 4 * D, 2 * D, sv["P2"][l], T, Tm, dctx, D, dq2, D, dkv,
                                      2 * D, dkv + 4 * D, 2 * D, qrow0, qlen, mrow0, mklen, 0, S, nh, 64, T, Tm, p_dec,
                                      op + 2, self._seed_ptr, s), "ac_attn_seq_bwd(cross)")
            w_in, g_in = fp.p(lp + "multihead_attn.in_proj_weight"), fp.g(lp + "multihead_attn.in_proj_weight")
            b_in_g = fp.g(lp + "multihead_attn.in_proj_bias")
            s2 = side.fork()
            self._lin_dw(s2, dq2, D, a["x1"], D, g_in, R, D, D)
            self._colsum(s2, dq2, D, b_in_g, R, D)
            self._lin_dw(s2, dkv, 2 * D, sv["mem"], D, g_in + 4 * D * D, Rm, 2 * D, D)
            self._colsum(s2, dkv, 2 * D, b_in_g + 4 * D, Rm, 2 * D)
            self._lin_dx(s, dq2, w_in, dres, R, D, D, beta=1.0)
            self._lin_dx(s, dkv, w_in + 4 * D * D, dmem, Rm, 2 * D, D, beta=0.0 if l == nlay - 1 else 1.0)
            dx, dres = dres, dx                                   # dx = d(x1)
            # norm1 / self attention
            side.join()
            check(lib.ac_dropadd_ln_bwd(dx, a["pre1"], fp.p(lp + "norm1.weight"), dsub, dres, 0, None, 0,
                                        fp.g(lp + "norm1.weight"), fp.g(lp + "norm1.bias"), R, D, p_dec, op + 1,
                                        self._seed_ptr, 1e-5, s), "ln1 bwd")
            s2 = side.fork()
            self._lin_dw(s2, dsub, D, a["ctx1"], D, fp.g(lp + "self_attn.out_proj.weight"), R, D, D)
            self._colsum(s2, dsub, D, fp.g(lp + "self_attn.out_proj.bias"), R, D)
            self._lin_dx(s, dsub, fp.p(lp + "self_attn.out_proj.weight"), dctx, R, D, D)
            check(lib.ac_attn_seq_bwd(a["qkv"], 3 * D, a["qkv"] + 4 * D, 3 * D, a["qkv"] + 8 * D, 3 * D, sv["P1"][l], T, T,
                                      dctx, D, dqkv, 3 * D, dqkv + 4 * D, 3 * D, dqkv + 8 * D, 3 * D, qrow0, qlen, qrow0,
                                      qlen, 0, S, nh, 64, T, T, p_dec, op + 0, self._seed_ptr, s), "ac_attn_seq_bwd")
            s2 = side.fork()
            self._lin_dw(s2, dqkv, 3 * D, x_in, D, fp.g(lp + "self_attn.in_proj_weight"), R, 3 * D, D)
            self._colsum(s2, dqkv, 3 * D, fp.g(lp + "self_attn.in_proj_bias"), R, 3 * D)
            self._lin_dx(s, dqkv, fp.p(lp + "self_attn.in_proj_weight"), dres, R, 3 * D, D, beta=1.0)
            dx, dres = dres, dx                                   # dx = d(layer input)
        side.join()   # the embedding's gradient may be the classifier's (tied weights): no two writers at once
        check(lib.ac_embed_bwd(dx, sv["word"], fp.g(dp + "word_embedding.weight"), R, D, p_dec, OP_EMB_A, p_dec, OP_EMB_B,
                               self._seed_ptr, s), "ac_embed_bwd")

        # ---- audio memory -> attn_proj -> GRU output ---------------------------------------------------
        da_rep = ws.f("da_rep", Rm, D)
        check(lib.ac_dropadd_ln_bwd(dmem, sv["mem_pre"], fp.p(dp + "attn_proj.3.weight"), da_rep, None, 0, sv["mem_a"],
                                    rows_m, fp.g(dp + "attn_proj.3.weight"), fp.g(dp + "attn_proj.3.bias"), Rm, D, p_dec,
                                    OP_MEM, self._seed_ptr, 1e-5, s), "mem ln bwd")
        da = ws.f("da", rows_m, D)
        check(lib.ac_sum_replicas(da_rep, da, rows_m * D, NP, s), "ac_sum_replicas")
        A = 2 * H
        s2 = side.fork()
        self._lin_dw(s2, da, D, sv["attn_emb"], A, fp.g(dp + "attn_proj.0.weight"), rows_m, D, A)
        self._colsum(s2, da, D, fp.g(dp + "attn_proj.0.bias"), rows_m, D)
        rows_g = B * Tq
        dout = ws.f("gru_dout", rows_g, A)
        self._lin_dx(s, da, fp.p(dp + "attn_proj.0.weight"), dout, rows_m, D, A)
        if part == "all":
            self._launch_backward_gru(sv)
        side.join()   # every decoder gradient is final (the all-reduce of part "head" reads them next)

    def _launch_backward_gru(self, sv):
        model, lib, fp = self.model, self.lib, self.flat
        enc = model.encoder
        s = _lib.stream()
        self._phase = "backward"
        ws = sv["ws"]
        N, Tq = sv["N"], sv["Tq"]
        B, p_rnn, A = N, sv["p_rnn"], 2 * H
        self._seed_ptr = sv["small"].data_ptr()
        rows_g = B * Tq
        dout = ws.f("gru_dout", rows_g, A)
        lens_p = sv["small"].data_ptr() + 8
        nl = enc.rnn.num_layers
        side = self._side = getattr(self, "_side", None) or _SideStream()
        pre = "encoder.rnn.network."
        for l in reversed(range(nl)):
            g = sv["gru"][l]
            # two sets of the recurrence's gradient buffers: layer l's weight gradients (side stream) read set l % 2 while layer
            # l - 1's backward through time fills the other one
            k_ = l & 1
            dgx, dgh, hprev = ws.f(f"dgx{k_}", rows_g, 6 * H), ws.f(f"dgh{k_}", rows_g, 6 * H), ws.f(f"hprev{k_}", rows_g, 2 * H)
            if l + 2 < nl:
                side.join()   # this set was read by layer l + 2's weight gradients
            if l < nl - 1 and p_rnn > 0:
                check(lib.ac_dropout(dout, dout, rows_g * A, p_rnn, OP_GRU_LAYER + l, self._seed_ptr, 0, s), "ac_dropout")
            check(lib.ac_gru_layer_bwd(dout, g["out"], ws.f(f"gru_save{l}", rows_g, 8 * H), fp.p(f"{pre}weight_hh_l{l}"),
                                       lens_p, dgx, dgh, hprev, B, Tq, H, s), "ac_gru_layer_bwd")
            s2 = side.fork()
            self._lin_dw(s2, dgx, 6 * H, g["x"], g["in_dim"], fp.g(f"{pre}weight_ih_l{l}"), rows_g, 6 * H, g["in_dim"])
            self._colsum(s2, dgx, 6 * H, fp.g(f"{pre}bias_ih_l{l}"), rows_g, 6 * H)
            self._colsum(s2, dgh, 6 * H, fp.g(f"{pre}bias_hh_l{l}"), rows_g, 6 * H)
            for d_ in range(2):
                # dW_hh[dir] (3H, H) += dgh[:, dir]^T hprev[:, dir]
                sk = self._splitk(3 * H, H, rows_g)
                self._gemm(s2, dgh + 4 * d_ * 3 * H, 1, 6 * H, hprev + 4 * d_ * H, 2 * H, 1,
                           fp.g(f"{pre}weight_hh_l{l}") + 4 * d_ * 3 * H * H, H, 3 * H, H, rows_g, None, 0, 1.0, sk)
            if l > 0:
                self._lin_dx(s, dgx, fp.p(f"{pre}weight_ih_l{l}"), dout, rows_g, 6 * H, g["in_dim"])
        side.join()

    # ---- fast path: forward + loss + backward (+ gradient all-reduce) + clip + Adam -----------------------
    def _launch_part(self, st, smoothing, part):
        """One capturable piece of an iteration:
        "fwd0"       frozen Cnn14, GRU, audio memory and EVERY decoder pass teacher forced as one batch;
        ("pass", t)  free-running pass t re-run on the model's own predictions (only the passes the draws made so);
        "tail"       label-smoothing loss (mean over the valid target tokens, counted on the device) + the whole backward;
        "tail_head" / "gru"   the same in two halves (see ``_launch_backward``) when gradients are all-reduced."""
        self._seed_ptr = st["small"].data_ptr()
        if part == "fwd0":
            return self._launch_forward(st, free=False)
        if isinstance(part, tuple):
            return self._decoder_passes(st, part[1], part[1] + 1, st["ctx"]["ucap"])
        if part == "gru":
            return self._launch_backward(st, None, "gru")
        N, T, Tc, V = st["N"], st["T"], st["Tc"], st["V"]
        ws = st["ws"]
        logit = ws.f("logit", N * T, V)
        dlogit = ws.f("dlogit", N * T, V)
        row_loss, loss = ws.f("row_loss", N * T), ws.f("loss", 1)
        tgt_len = st["small"].data_ptr() + 4 * (2 + N)
        check(self.lib.ac_label_smoothing_loss(logit, st["cap"].data_ptr() + 8, Tc, tgt_len, N, T, V, float(smoothing), 0.0,
                                               row_loss, loss, dlogit, 0.0, None, _lib.stream()),
              "ac_label_smoothing_loss")
        self._launch_backward(st, dlogit, "all" if part == "tail" else "head")

    # ---- frozen Cnn14 one iteration ahead ---------------------------------------------------------------------------
    def prefetch_cnn(self, input_dict, seed):
        """Launch the frozen Cnn14 forward (log-mel, SpecAugment, conv stack with its dropout) of a batch on the side stream
        NOW, for the iteration that will run with dropout seed ``seed``: it is matrix-bound and does not depend on any
        trainable parameter, so it overlaps the latency-bound GRU / decoder forward and backward of the iteration in
        flight (3 of ~7.5 ms at batch 32).  ``step`` picks the result up when it is handed the same batch - the same ``wav``
        storage at the same version counter and shape, the same SpecAugment flag, train / eval mode and dropout seed
        (``_pf_key``); anything else (a loader that refills one static tensor in place, an explicit ``dropout_seed``, a
        train-mode ``forward`` in between that advanced the seed) discards the look-ahead and the Cnn14 forward is run
        again for the batch in hand.  Hand ``next_batch`` fresh tensors (or at least do not write into them before the
        step that consumes them).  The masks are those the in-line forward would draw (same counter hash, same seed
        word): same loss, same gradients."""
        model = self.model
        enc = model.encoder
        wav = input_dict["wav"]
        dev = wav.device
        if not wav.is_cuda:
            raise _lib.HipLibraryError("the training step needs tensors on a ROCm device; there is no CPU fallback")
        if getattr(self, "_cnn_stream", None) is None or self._cnn_stream.device != dev:
            self._cnn_stream = torch.cuda.Stream(dev)
            self._pf_seeds = [torch.zeros(1, device=dev, dtype=torch.int64) for _ in range(3)]
            self._pf_turn = 0
        side = self._cnn_stream
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)                     # the batch may have been produced on the caller's stream
        self._pf_turn = (self._pf_turn + 1) % len(self._pf_seeds)
        seed_dev = self._pf_seeds[self._pf_turn]
        specaug = bool(input_dict.get("specaug", False)) and enc.cnn.training
        p_cnn = 0.2 if enc.cnn.training else 0.0
        with torch.cuda.stream(side):
            seed_dev.fill_(int(seed))
            sa = None
            if specaug:
                from .kernels import specaug_stripes
                T_frames = wav.shape[1] // enc.cnn.hop_length + 1
                host = torch.from_numpy(specaug_stripes((int(seed) << 16) + OP_SPECAUG, wav.shape[0], T_frames)).pin_memory()
                sa = host.to(dev, non_blocking=True)
                self._pf_pinned = (getattr(self, "_pf_pinned", []) + [host])[-4:]
            attn = enc.cnn.encode(wav.float(), dropout=(p_cnn, OP_CNN_BLOCK, seed_dev.data_ptr()) if p_cnn > 0 else None,
                                  specaug=sa, train=True)
            ev = torch.cuda.Event()
            ev.record(side)
        attn.record_stream(cur)
        self._pf = {"wav": wav, "attn": attn, "event": ev, "seed": int(seed), "key": self._pf_key(input_dict, seed)}

    def _pf_key(self, input_dict, seed):
        """What a look-ahead Cnn14 result was computed FROM: the step that consumes it must present the same."""
        wav = input_dict["wav"]
        cnn = self.model.encoder.cnn
        return (wav.data_ptr(), wav._version, tuple(wav.shape), wav.dtype, bool(input_dict.get("specaug", False)),
                bool(cnn.training), int(seed))

    def step(self, input_dict, optimizer, smoothing=0.1, max_grad_norm=1.0, process_group=None, use_graph=True,
             next_batch=None):
        """One training iteration (run.py:106-126) without leaving the HIP path; returns the loss as a device scalar.

        The launches of an iteration are latency-bound at the reference's batch sizes, so for each batch shape they are
        captured into HIP graphs over static buffers (from the second iteration of that shape on) and replayed; the
        per-iteration data - audio, captions, lengths, the scheduled-sampling draws and the dropout seed - are copied
        into those buffers first.  An iteration replays: one graph for the encoder + all decoder passes teacher forced,
        one small graph per free-running pass the draws produced (captured the first time pass t is free), one graph for
        loss + backward.  Gradients land in the flat buffer (= the parameters' ``.grad``); with ``torch.distributed``
        initialised they are summed over the ranks by two all-reduces (the decoder's slice under the GRU backward, then
        the GRU's) and the division by the world size is folded into the clip coefficient; clip + Adam are three launches
        on the flat buffers."""
        from .optim import FusedAdam, clip_grad_norm_
        if "cap_len" not in input_dict:
            raise KeyError("cap_len")
        # ``next_batch``: the batch of the FOLLOWING iteration - its frozen Cnn14 forward is launched on a side stream under
        # this iteration's GRU / decoder work (``prefetch_cnn``).  The convolutions share one set of activation buffers, so
        # with a look-ahead in play this iteration's own Cnn14 forward goes through the side stream as well.
        if input_dict.get("_cnn_attn") is None and (next_batch is not None or getattr(self, "_pf", None) is not None):
            pf = getattr(self, "_pf", None)
            want = int(input_dict.get("dropout_seed", self.seed))   # the caller's seed wins over the one guessed a step ahead
            if pf is None or pf["key"] != self._pf_key(input_dict, want):
                if pf is not None:
                    self.lookahead_misses = getattr(self, "lookahead_misses", 0) + 1   # a look-ahead that could not be used
                self.prefetch_cnn(input_dict, want)
                pf = self._pf
            self._pf = None
            torch.cuda.current_stream(input_dict["wav"].device).wait_event(pf["event"])
            input_dict = dict(input_dict, _cnn_attn=pf["attn"], dropout_seed=want)
        st = self._prepare(input_dict)
        if next_batch is not None:
            # the next iteration's own seed if it carries one, else the one it will draw by default
            self.prefetch_cnn(next_batch, int(next_batch.get("dropout_seed", self.seed)))
        st["steps"] += 1
        world = dist_world_size(process_group)
        # Several ranks: the backward runs as TWO parts - up to the end of the decoder's backward, then the GRU's backward
        # through time - and the all-reduce of the decoder's gradients (45 % of the 42.8 MB, final after part one) is
        # issued between them, so that it travels under the GRU backward; the GRU's gradients follow.
        parts = ["fwd0"] + [("pass", t) for t in st["free_ts"]] + (["tail_head", "gru"] if world > 1 else ["tail"])
        graphs = st.setdefault("graphs", {})
        eager = (not use_graph) or st["steps"] < 2      # first iteration of this shape: eager (also the warm-up)
        self._pw_build()                                # (the eager iteration registered the layers; never inside a capture)
        works = []
        for part in parts:
            if eager:
                self._launch_part(st, smoothing, part)
            else:
                # A captured graph holds RAW ADDRESSES: the flat parameter storage, the shared workspace, and - through
                # the frozen Cnn14 - its packed weights and its activation buffers, which are shared by all batch shapes
                # and re-allocated when a larger one arrives.  All of them are part of the key, and the state keeps
                # references to what its graphs address, so a stale graph is neither replayed nor left pointing at
                # recycled memory.
                cnn = getattr(self.model.encoder, "cnn", None)
                cnn_algo = cnn.effective_algo(None, True) if cnn is not None and hasattr(cnn, "_pack") else None
                if cnn_algo is not None and part == "fwd0":
                    # weight repacks (host uploads, float64 transforms of the Winograd tier) must not land inside a
                    # capture: pack the frozen Cnn14 for the tier the train-mode forward uses now (cached: a no-op
                    # unless the Cnn14's own tensors changed)
                    cnn._pack(st["cap"].device, cnn_algo)
                ident, refs = cnn.capture_token(cnn_algo) if cnn_algo is not None else (None, None)
                gkey = (smoothing, _lib.param_generation_flat(self), st["ws"].gen, ident, world > 1)
                hit = graphs.get(part)
                if hit is None or hit[1] != gkey:
                    if part != "fwd0" and (graphs.get("fwd0") is None or graphs["fwd0"][1] != gkey):
                        raise RuntimeError("TrainEngine: graph parts out of order")   # fwd0 refreshes the shared context
                    torch.cuda.synchronize(st["cap"].device)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self._launch_part(st, smoothing, part)
                    ident, refs = cnn.capture_token(cnn_algo) if cnn_algo is not None else (None, None)
                    hit = graphs[part] = (graph, (smoothing, _lib.param_generation_flat(self), st["ws"].gen, ident,
                                                  world > 1), refs)
                hit[0].replay()
            if world > 1 and part in ("tail_head", "gru"):
                o = self.flat.decoder_offset
                piece = self.flat.grad[o:] if part == "tail_head" else self.flat.grad[:o]
                works.append(allreduce_flat_gradients(piece, process_group, async_op=True))
        for w_ in works:
            if w_ is not None:
                w_.wait()
        self.flat.attach_grads()
        clip = clip_grad_norm_(self.flat.params, max_grad_norm, grad_div=float(world), scale_now=False)
        # A partner timeout of the split GRU kernel (four cooperating workgroups per (clip, direction); GPU shared or
        # over-subscribed) leaves stale hidden values behind: its sticky error word is folded into the clip state's
        # non-finite flag ON THE DEVICE, so the update is skipped exactly like a NaN loss (run.py:123) without a host
        # synchronisation; ``gru_timeout()`` reads it for callers that want to raise.
        err = self._gru_error_word(st)
        if err is not None:
            hit = (err != 0).to(clip.state.dtype)
            err.zero_()           # in stream order: the word judges ONE iteration, the next one starts clean
            if world > 1:
                # every rank must skip together: this rank's gradients are already in the others' all-reduced sums
                import torch.distributed as dist
                dist.all_reduce(hit, op=dist.ReduceOp.MAX, group=process_group)
            clip.state[3:4].add_(hit)
            if getattr(self, "_gru_timeouts", None) is None or self._gru_timeouts.device != hit.device:
                self._gru_timeouts = torch.zeros(1, device=hit.device, dtype=torch.float32)
            self._gru_timeouts.add_(hit.to(torch.float32))
        if getattr(self, "_skipped", None) is None or self._skipped.device != clip.state.device:
            self._skipped = torch.zeros(1, device=clip.state.device, dtype=torch.float32)
        self._skipped.add_((clip.state[3:4] != 0).to(torch.float32))   # updates skipped so far (non-finite gradients, GRU timeouts)
        if isinstance(optimizer, FusedAdam):
            optimizer.step(clip=clip)
        else:
            # Any other torch optimiser: the clip coefficient is applied to the gradients first.  A non-finite gradient
            # norm (state[3] != 0) must skip the WHOLE update like the reference does (run.py:123) - scaling by 0 would
            # turn the NaNs into NaN * 0 = NaN and the optimiser would write them into parameters and moments - so
            # this path reads the flag back (one host synchronisation per step; FusedAdam skips on the device).
            if float(clip.state[3].item()) != 0.0:
                self.flat.grad.zero_()
            else:
                check(self.lib.ac_scale_by_coef(self.flat.grad.data_ptr(), self.flat.total, clip.state.data_ptr(),
                                                _lib.stream()), "ac_scale_by_coef")
                optimizer.step()
                _lib.bump_param_generation(self.flat.params)
        out = self._outputs(st)
        return {"loss": st["ws"].tensor("loss")[0], "total_norm": clip.total_norm, "logit": out["logit"],
                "seq": out.get("seq"), "skipped_updates": self._skipped[0]}


def _gru_error_word(self, st):
    """The split-GRU kernel's sticky error word of this engine's workspace as a 1-element int32 device view, or None."""
    if self.gru_algo != "split" or not st.get("gru_xch_zeroed"):
        return None
    return st["ws"].tensor("gru_xch").view(torch.int32)[:1]


def gru_timeout(self):
    """True if a split-GRU launch of this engine timed out waiting for a partner workgroup since the last call (reads the
    device: synchronises).  ``step`` folds the kernel's error word into that iteration's skip flag and clears it, counting
    the hits; a word raised by a forward that no ``step`` followed is picked up here.  The affected iterations' updates
    were skipped on the device (on every rank under DDP)."""
    hit = False
    cnt = getattr(self, "_gru_timeouts", None)
    if cnt is not None and float(cnt.item()) != 0.0:
        hit = True
        cnt.zero_()
    for st in self._states.values():
        err = _gru_error_word(self, st)
        if err is not None and int(err.item()) != 0:
            hit = True
            err.zero_()
    return hit


def skipped_updates(self):
    """Optimiser updates ``step`` has skipped on the device so far (non-finite gradient norm, run.py:123, or a split-GRU
    partner timeout); reads the device."""
    c = getattr(self, "_skipped", None)
    return int(c.item()) if c is not None else 0


TrainEngine._gru_error_word = _gru_error_word
TrainEngine.gru_timeout = gru_timeout
TrainEngine.skipped_updates = skipped_updates


class _TrainBridge(torch.autograd.Function):
    """One autograd node for the whole HIP training forward: ``logit`` depends on every trainable parameter; its
    backward runs TrainEngine.backward and hands autograd the views of the flat gradient buffer, so ``.grad``
    accumulation, optimizers and torch's DistributedDataParallel hooks behave as with the reference model."""

    @staticmethod
    def forward(ctx, engine, logit, *params):
        ctx.engine = engine
        return logit.view_as(logit)

    @staticmethod
    def backward(ctx, dlogit):
        ctx.engine.backward(dlogit)
        return (None, None) + tuple(ctx.engine.flat.grad_views)


def train_forward(model, input_dict):
    """``TransformerModel.forward`` for ``mode == "train"`` (base.py:73-112 -> train_forward)."""
    engine = getattr(model, "_train_engine", None)
    if engine is None:
        engine = model._train_engine = TrainEngine(model)
    out = engine.forward(input_dict)
    if torch.is_grad_enabled():
        out["logit"] = _TrainBridge.apply(engine, out["logit"], *engine.flat.params)
    return out
