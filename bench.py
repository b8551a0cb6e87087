#!/usr/bin/env python
"""Headline benchmark: clips/s for encode + greedy decode of the Cnn14Rnn-Trm captioner on synthetic
10 s @ 32 kHz clips (BASELINE.json metric, configs[1]: batch 64 per GPU, max_length 20).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL when N > 1): launched under ``torch.distributed.run`` the ranks are read
from the environment; launched plainly with ``--gpus N > 1`` the script re-executes itself under
``torch.distributed.run`` with N ranks.  Clips are independent, so the batch is sharded across ranks with no data-path
collective ("scaling": "weak").  A step is one pass of the hot path (log-mel -> Cnn14 -> bi-GRU -> greedy Transformer
decoding, token ids back on the host) over one resident batch.  Rank 0 prints ONE JSON line with

* ``value``: the throughput of the DEFAULT conv tier - "wino43", f32-grade: logits within 1e-4 of the fp32 reference,
  identical token ids (``config.precision_gate``) - over exactly K timed steps that rotate over four resident input
  batches;
* ``roofline``: the dominant kernel (conv2 + BN + ReLU + 2x2 pool of blocks 2-5) timed live with HIP events on its
  launch stream; ``achieved`` = algorithmic direct-convolution FLOPs / time, ``frac`` = achieved / the dense peak of the
  pipe the products run on (the roofline fraction); ``mfma_issue_frac`` = MFMA FLOPs actually ISSUED / that peak (pipe
  utilisation: the split-bf16 F(4,3) form issues 1.5 bf16 products per algorithmic f32 product);
* flat scalars measured in THIS run: ``value_f32_exact`` (Winograd on the f32 MFMA), ``value_bf16x3_direct``,
  ``value_f16x2_half_precision_gate`` (the opt-in fp16 tier: NOT reference precision), ``value_blocking_model_call`` (the
  reference's own call, one blocking ``model(input_dict)`` per step), ``latency_b1_greedy_ms`` / ``latency_b1_beam3_ms``
  (one 10 s clip through ``model()``, what demo.py / the HF surface run), ``train_clips_per_s`` (BASELINE configs[3]),
  ``effb2_trm_clips_per_s`` (configs[2]), ``clotho_shape_clips_per_s`` / ``clotho_shape_no_skip_clips_per_s`` (32 ragged
  15-30 s clips, zero-padded: with / without dead-row skipping);
* ``cpu_baseline``: the oracle (CPU restatement of the reference) on all physical host cores, median of 5 passes over a
  bounded 32-clip sample.
The per-tier rooflines, the steady-state window, the decoder GEMM table, the training-step and EffB2 objects go to
``gpurun_out/bench_details.json`` (``details_file``) so that the printed line stays short.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch


def _decode_group():
    from audiocaption_amd.transformer_model import decode_group
    return decode_group()

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 / fp16 MFMA peak (v_mfma_f32_32x32x16_{bf16,f16})
HBM_PEAK_GBS = 8000.0

# MFMA work issued per algorithmic (direct-convolution f32) FLOP, per conv tier: Winograd F(2x2,3x3) needs 16 instead of
# 36 products per tile; the split-bf16 path issues three bf16 products per f32 product (hi*hi, hi*lo, lo*hi); the fp16
# tier two (x16*w_hi, x16*w_lo)
TIERS = {
    "wino43": {"conv_algo": "wino43", "linear_algo": "bf16x3", "issue_ratio": 1.5, "peak": BF16_MFMA_PEAK_TFLOPS,
               "dtype": "bf16x3",
               "kernel": "conv3x3_w4_kernel<POOL> (F(4,3) Winograd along time on split-bf16 operands: 18 transformed "
                         "products x 3 bf16 MFMAs per 36 direct f32 products, f32 accumulate; one 512-register wave per SIMD)",
               "gate": "f32 gate: logits within 1e-4 of the fp32 CPU reference, identical token ids (measured 4e-5; split-bf16 "
                       "operands carry 16 significant bits, activations stay f32 in HBM, f32 accumulation) - asserted end "
                       "to end by tests/test_gpu_model.py"},
    "wino1d": {"conv_algo": "wino1d", "linear_algo": "bf16x3", "issue_ratio": 2.0, "peak": BF16_MFMA_PEAK_TFLOPS,
               "dtype": "bf16x3",
               "kernel": "conv3x3_w1_kernel<POOL> (F(2,3) Winograd along time on split-bf16 operands: 12 transformed "
                         "products x 3 bf16 MFMAs per 18 direct f32 products, f32 accumulate)",
               "gate": "f32 gate: logits within 1e-4 of the fp32 CPU reference, identical token ids (measured 4e-5; split-bf16 "
                       "operands carry 16 significant bits, activations stay f32 in HBM, f32 accumulation) - asserted end "
                       "to end by tests/test_gpu_model.py"},
    "f32": {"conv_algo": "winograd", "linear_algo": "f32", "issue_ratio": 1.0 / 2.25, "peak": FP32_MFMA_PEAK_TFLOPS,
            "dtype": "f32", "kernel": "conv3x3_wino_kernel<POOL> (Winograd F(2x2,3x3), v_mfma_f32_32x32x2_f32)",
            "gate": "f32 end to end: logits within 1e-4 of the fp32 CPU reference, identical token ids (measured 4e-6)"},
    "bf16x3": {"conv_algo": "bf16x3", "linear_algo": "bf16x3", "issue_ratio": 3.0, "peak": BF16_MFMA_PEAK_TFLOPS,
               "dtype": "bf16x3",
               "kernel": "conv3x3_gw_kernel<128, POOL, PREC 0 (split bf16 x 3), 2x2 waves, 128 px>",
               "gate": "f32 gate: logits within 1e-4 of the fp32 CPU reference, identical token ids (measured 3e-5; "
                       "operands carry 16 significant bits, f32 accumulation)"},
    "f16x2": {"conv_algo": "f16x2", "linear_algo": "bf16x3", "issue_ratio": 2.0, "peak": BF16_MFMA_PEAK_TFLOPS,
              "dtype": "f16x2",
              "kernel": "conv3x3_gw_kernel<128, POOL, PREC 1 (fp16 x 2), 1x4 waves, 256-pixel column-tile blocks>",
              "gate": "BASELINE.json north_star half-precision gate: identical greedy/beam token ids, logits within 1e-3 of "
                      "the fp32 CPU reference (worst measured over 5 seeds x 6 clip lengths: "
                      "tests/test_gpu_model.py::test_default_tier_logit_error_by_clip_length, bar 5e-4)"},
}
ALGO_TO_TIER = {"winograd": "f32", "direct": "f32", "bf16x3": "bf16x3", "bf16x3_lds": "bf16x3", "f16x2": "f16x2",
                "wino1d": "wino1d", "wino43": "wino43"}


# ---------------------------------------------------------------------------------------------------------------------
# launch plumbing: one process per GPU
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_spawn(n):
    """``python bench.py --gpus N`` outside a launcher: re-execute under torch.distributed.run with N ranks on this node
    (the command the driver itself uses); the children see WORLD_SIZE and do the work."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class _CStdoutToStderr:
    """RCCL prints a version banner on the C-level stdout when a communicator is created; stdout of this script carries
    exactly ONE JSON line, so file descriptor 1 points at stderr while communicators are being set up."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class Ranks:
    """world / rank / device of this process and the two collectives the bench contract needs (barrier, MAX)."""

    def __init__(self, need_gpu=True):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = None
        self.gpu = need_gpu
        dev_index = 0
        if need_gpu:
            n_dev = torch.cuda.device_count()
            if n_dev == 0:
                raise SystemExit("bench.py: no ROCm device visible (the hot path has no CPU fallback)")
            dev_index = self.local_rank % n_dev
            torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index) if need_gpu else torch.device("cpu")
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # one rank per GPU over RCCL ("nccl").  AUDIOCAPTION_BENCH_BACKEND=gloo lets the multi-process flow be
            # exercised where there are fewer GPUs than ranks (ranks then share a device; RCCL refuses that)
            self.backend = os.environ.get("AUDIOCAPTION_BENCH_BACKEND", "nccl" if need_gpu else "gloo")
            if self.backend == "nccl":
                if torch.cuda.device_count() < self.world:
                    raise SystemExit(f"bench.py: {self.world} ranks over RCCL need {self.world} GPUs, "
                                     f"{torch.cuda.device_count()} visible")
                with _CStdoutToStderr():
                    dist.init_process_group("nccl", device_id=self.dev)
                    t = torch.zeros(1, device=self.dev)
                    dist.all_reduce(t)                 # communicator created here, not inside a timed region
                    torch.cuda.synchronize()
            else:
                dist.init_process_group(self.backend)
            self.dist = dist

    def describe(self):
        """One record per rank (gathered on every rank): which device it runs on - so that an N-GPU line shows that the
        ranks really sat on N different GPUs."""
        me = {"rank": self.rank, "local_rank": self.local_rank, "device": str(self.dev), "host": socket.gethostname(),
              "pid": os.getpid()}
        if self.gpu:
            pr = torch.cuda.get_device_properties(self.dev)
            me.update({"name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None),
                       "uuid": str(getattr(pr, "uuid", "")) or None, "pci_bus_id": getattr(pr, "pci_bus_id", None),
                       "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))})
        if self.dist is None:
            return [me]
        got = [None] * self.world
        self.dist.all_gather_object(got, me)
        return got

    def sync_all(self):
        if self.gpu:
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.gpu:
            torch.cuda.synchronize()

    def max_seconds(self, seconds):
        if self.dist is None:
            return float(seconds)
        dev = self.dev if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_steps(ranks, run, steps):
    """The bench contract's timed region: barrier + synchronize on both sides, MAX over ranks."""
    ranks.sync_all()
    t0 = time.perf_counter()
    out = run(steps)
    ranks.sync_all()
    ranks.last_own = time.perf_counter() - t0   # this rank's own clock (the line reports the MAX over ranks)
    return ranks.max_seconds(ranks.last_own), out


def collective_record(ranks, dev, nccl_floats=10_700_000):
    """What makes an N > 1 line self-describing about the node's collectives: on RCCL the all-reduce / reduce-scatter +
    all-gather rates on a buffer of the training step's gradient size; on gloo (CPU ranks, tests) the same two spellings of
    the sum on a small buffer, checked against each other.  ``grad_sync_default``: what the training step picks when
    AUDIOCAPTION_GRAD_SYNC is not set (audiocaption_amd/train.py default_grad_sync)."""
    from audiocaption_amd.train import allreduce_flat_gradients, default_grad_sync
    world = ranks.world
    if ranks.backend == "nccl":
        rec = time_allreduce(ranks, torch.zeros(nccl_floats, device=dev))
    else:
        rec = {"backend": ranks.backend, "world_size": world}
        if world > 1:
            a = torch.arange(1003, dtype=torch.float32) * (ranks.rank + 1)
            b = a.clone()
            t0 = time.perf_counter()
            allreduce_flat_gradients(a, algo="all_reduce")
            t1 = time.perf_counter()
            allreduce_flat_gradients(b, algo="rs_ag")
            t2 = time.perf_counter()
            rec.update({"all_reduce_ms": (t1 - t0) * 1e3, "rs_ag_ms": (t2 - t1) * 1e3, "bytes": a.numel() * 4,
                        "rs_ag_equals_all_reduce": bool(torch.equal(a, b))})
    rec["grad_sync_default"] = default_grad_sync(world, ranks.backend)
    return rec


def seconds_per_rank(ranks, own_seconds, dev):
    own = torch.tensor([own_seconds], dtype=torch.float64, device=dev if ranks.backend == "nccl" else "cpu")
    if ranks.world == 1:
        return [float(own)]
    every = [torch.zeros_like(own) for _ in range(ranks.world)]
    ranks.dist.all_gather(every, own)
    return [float(t) for t in every]


def bench_stub(args, ranks):
    """Launch-plumbing check without the HIP path (tests/test_bench_launch.py): the "step" is a small CPU matmul, everything
    else - ranks from the environment or self-spawned, barrier, MAX-over-ranks timing, rank 0 printing one line with
    n_gpus = world - is the code the real modes run."""
    x = torch.randn(128, 128, generator=torch.Generator().manual_seed(ranks.rank))

    def run(n):
        y = x
        for _ in range(n):
            y = torch.tanh(y @ x * 1e-2)
        return y

    run(args.warmup)
    elapsed, _ = timed_steps(ranks, run, args.steps)
    who = ranks.describe()      # collective: every rank calls it
    per_rank = seconds_per_rank(ranks, ranks.last_own, "cpu")
    rccl = collective_record(ranks, "cpu")
    return {"ranks": who, "seconds_per_rank": per_rank, "rccl": rccl, "metric": "stub steps/sec (launch plumbing only)", "value": ranks.world * args.steps / elapsed, "unit": "steps/s",
            "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "stub (no HIP path): one 128x128 CPU matmul per step", "backend": ranks.backend,
                       "requested_gpus": args.gpus}}


# ---------------------------------------------------------------------------------------------------------------------
# training step (BASELINE configs[3])
# ---------------------------------------------------------------------------------------------------------------------
def bench_train(args, ranks, steps, warmup, with_rccl=False, roofline=True):
    """Training iterations of the reference's recipe (run.py:77-148; cnn14rnn_trm.yaml): frozen Cnn14 with dropout,
    bi-GRU, scheduled-sampling decoder (ss_ratio 0.85), LabelSmoothingLoss(0.1), backward, gradient all-reduce
    (RCCL on the flat gradient buffer), clip_grad_norm_(1.0), Adam(5e-4, weight_decay 1e-6) - everything
    inside the timed region, synthetic AudioCaps-shape batches resident in HBM."""
    import random
    import numpy as np
    import audiocaption_amd as A
    from audiocaption_amd import build, procedural as P
    from audiocaption_amd.optim import FusedAdam
    from audiocaption_amd import train as T
    build.build()
    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    vocab, cap_len = 4981, 22   # AudioCaps vocabulary (cnn14rnn_trm.yaml:31); <bos> + 20 words + <eos>
    model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
    model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
    model = model.to(dev).train()
    B, L = args.train_batch, int(args.seconds * 32000)
    wav = torch.from_numpy(P.synthetic_wav(B, L, seed=P.BASE_SEED + 100 + rank)).to(dev)
    g = torch.Generator().manual_seed(1000 + rank)
    cap = torch.randint(4, vocab, (B, cap_len), generator=g)
    lens = torch.randint(8, cap_len + 1, (B,), generator=g)
    lens[0] = cap_len
    cap[:, 0] = 1
    for i, n in enumerate(lens.tolist()):
        cap[i, n - 1] = 2
        cap[i, n:] = 0
    batch = {"mode": "train", "wav": wav, "wav_len": [L] * B, "specaug": False, "cap": cap.to(dev),
             "cap_len": np.asarray(lens), "ss_ratio": 0.85}
    engine = T.TrainEngine(model, seed=rank * 1000003)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-6)
    random.seed(rank)
    # the frozen Cnn14 forward of iteration i + 1 runs on a side stream under iteration i (TrainEngine.prefetch_cnn): a data
    # loader that hands over the next batch early (AUDIOCAPTION_TRAIN_LOOKAHEAD=0: the plain loop)
    look_ahead = os.environ.get("AUDIOCAPTION_TRAIN_LOOKAHEAD", "1") != "0"

    r = None

    def run(n):
        last = None
        for _ in range(n):
            last = engine.step(batch, opt, smoothing=0.1, max_grad_norm=1.0, next_batch=batch if look_ahead else None)
        return last

    run(warmup)
    elapsed, r = timed_steps(ranks, run, steps)
    loss = float(r["loss"])
    n_param = engine.flat.total

    # roofline of the backward GEMMs (dgrad / wgrad of the decoder and the GRU; exact-f32 MFMA): one EAGER iteration with
    # HIP events around every ac_gemm launch of the backward (the replayed graph cannot be instrumented)
    roof = None
    try:
        if not roofline:
            raise RuntimeError("roofline not requested")
        ev = []

        def hook(phase, info):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            if phase == "pre":
                ev.append([e, None, info])
            else:
                ev[-1][1] = e

        engine.step(batch, opt, smoothing=0.1, max_grad_norm=1.0, use_graph=False)   # eager launches warmed up
        T.GEMM_HOOK = hook
        engine.step(batch, opt, smoothing=0.1, max_grad_norm=1.0, use_graph=False)
        T.GEMM_HOOK = None
        torch.cuda.synchronize()
        bwd = [(a.elapsed_time(b), i) for a, b, i in ev if i["phase"] == "backward"]
        fwd = [(a.elapsed_time(b), i) for a, b, i in ev if i["phase"] == "forward"]
        fl_b, ms_b = sum(i["flops"] for _, i in bwd), sum(t for t, _ in bwd)
        fl_f, ms_f = sum(i["flops"] for _, i in fwd), sum(t for t, _ in fwd)
        big = max(bwd, key=lambda x: x[1]["flops"])
        if os.environ.get("AUDIOCAPTION_BENCH_GEMM_LIST"):   # development: every GEMM launch of the iteration, slowest first
            rows = sorted(((t, i["phase"], i["M"], i["N"], i["K"]) for t, i in bwd + fwd), reverse=True)
            with open(os.environ["AUDIOCAPTION_BENCH_GEMM_LIST"], "w") as f:
                for t, ph, M_, N_, K_ in rows:
                    f.write(f"{t * 1e3:8.1f} us  {ph:8s} M {M_:6d} N {N_:6d} K {K_:6d}  {2.0 * M_ * N_ * K_ / t / 1e9:7.1f} TFLOP/s\n")
        split = engine.gemm_algo in ("bf16x3", "pw")
        peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        roof = {"bound": "mfma",
                "kernel": ("pw_bf16x3_kernel for dy W (ac_pw_gemm_bf16x3: weights re-split into MFMA fragment order once per "
                           "iteration, activations split once per 256 columns) + gemm_bf16x3_kernel for dy^T x (ac_gemm_bf16x3: "
                           "both operands split into bf16 hi + lo at staging); three v_mfma_f32_32x32x16_bf16 per product, f32 "
                           "accumulation; small / unaligned products on the exact-f32 kernels" if engine.gemm_algo == "pw" else
                           "gemm_bf16x3_kernel (ac_gemm_bf16x3: operands split into bf16 hi + lo at staging, three "
                           "v_mfma_f32_32x32x16_bf16 per product, f32 accumulation; small / unaligned products on the exact-f32 "
                           "kernels)" if split else "gemm_general / gemm_nt / gemm_kk (ac_gemm, exact f32 v_mfma_f32_32x32x2_f32)")
                          + f": the {len(bwd)} dgrad + wgrad launches of one backward pass",
                "achieved": fl_b / ms_b / 1e9, "peak": peak, "unit": "TFLOP/s",
                "frac": fl_b / ms_b / 1e9 * (3.0 if split else 1.0) / peak, "algorithmic_frac": fl_b / ms_b / 1e9 / peak,
                "issued_per_algorithmic": 3.0 if split else 1.0, "traffic": None,
                "gemm_ms_per_step": ms_b, "gflop_per_step": fl_b / 1e9, "launches": len(bwd),
                "largest": {"M": big[1]["M"], "N": big[1]["N"], "K": big[1]["K"], "ms": big[0],
                            "tflops": big[1]["flops"] / big[0] / 1e9},
                "forward_gemms": {"achieved": fl_f / ms_f / 1e9 if ms_f > 0 else None, "gemm_ms_per_step": ms_f,
                                  "gflop_per_step": fl_f / 1e9, "launches": len(fwd),
                                  "frac": fl_f / ms_f / 1e9 * (3.0 if split else 1.0) / peak if ms_f > 0 else None,
                                  "note": "all decoder passes teacher forced as one batch + the free-running passes re-run"},
                "note": "HIP events around every ac_gemm of one eager iteration (the timed steps replay a HIP graph)"}
    except Exception as e:  # noqa: BLE001 - a secondary measurement never costs the line
        T.GEMM_HOOK = None
        roof = {"error": f"{type(e).__name__}: {e}"}

    rccl = None
    if with_rccl:
        rccl = time_allreduce(ranks, engine.flat.grad)
    res = {
        "metric": "clips/sec trained (forward+backward+Adam), Cnn14_Rnn-Trm, AudioCaps-shape batches",
        "value": world * B * steps / elapsed, "unit": "clips/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16x3 (frozen convolutions: F(4,3) Winograd on split-bf16 operands, the same kernels as inference; large GEMMs on split-bf16 operands; f32 "
                 "accumulation, f32 elsewhere)"
                 if engine.gemm_algo in ("bf16x3", "pw") else "bf16x3 frozen convolutions, f32 everything trained",
        "data": "synthetic",
        "config": {"workload": f"training step, batch {B} per GPU ({world * B} global), {args.seconds:g} s @ 32 kHz clips, "
                               f"captions of {cap_len} tokens, vocab {vocab}, scheduled sampling 0.85, dropout on "
                               "(BASELINE configs[3]; the reference's own recipe is global batch 32)",
                   "trainable_parameters": int(sum(p.numel() for p in engine.flat.params)),
                   "gradient_sync": ("all-reduce of the flat %.1f MB gradient buffer per step (RCCL), decoder half "
                                     "overlapped with the GRU backward" % (n_param * 4e-6))
                   if world > 1 else "single GPU: none",
                   "last_loss": loss},
        "roofline": roof,
    }
    if rccl is not None:
        res["rccl"] = rccl
    if world > 1:
        res["ranks"] = ranks.describe()
        res["config"]["gradient_sync_algo"] = os.environ.get("AUDIOCAPTION_GRAD_SYNC", "all_reduce")
    return res


def time_allreduce(ranks, flat_grad):
    """One all-reduce of the flat gradient buffer timed on its own (the collective of run_ddp.py:98-108).  With a single
    rank a 1-rank RCCL communicator is created for the measurement, so that the RCCL call itself has executed on this
    hardware (the data path of a 1-rank all-reduce is a device copy)."""
    import torch.distributed as dist
    own_group = False
    try:
        buf = flat_grad.clone()
        with _CStdoutToStderr():
            if ranks.dist is None:
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                        device_id=ranks.dev)
                own_group = True
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        world = dist.get_world_size()
        nbytes = buf.numel() * 4
        # the same sum spelled as reduce_scatter + all_gather (AUDIOCAPTION_GRAD_SYNC=rs_ag, audiocaption_amd/train.py)
        from audiocaption_amd.train import allreduce_flat_gradients
        rs_ms = None
        if world > 1:
            for _ in range(2):
                allreduce_flat_gradients(buf, algo="rs_ag")
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                allreduce_flat_gradients(buf, algo="rs_ag")
            e1.record()
            torch.cuda.synchronize()
            rs_ms = e0.elapsed_time(e1) / n
        bus = lambda t: (2.0 * (world - 1) / world) * nbytes / (t * 1e-3) / 1e9 if world > 1 and t else None
        return {"world_size": world, "backend": dist.get_backend(), "all_reduce_ms": ms, "bytes": nbytes,
                "bus_gbs": bus(ms), "rs_ag_ms": rs_ms, "rs_ag_bus_gbs": bus(rs_ms),
                "grad_sync": os.environ.get("AUDIOCAPTION_GRAD_SYNC") or "default (see grad_sync_default)",
                "xgmi_note": "8 GPUs fully connected, 7 links x ~153 GB/s per GPU: a ring is bound by ONE link per hop, "
                             "direct reduce-scatter / all-gather can use all 7"}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if own_group and dist.is_initialized():
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# EffB2-Transformer (BASELINE configs[2] / configs[4])
# ---------------------------------------------------------------------------------------------------------------------
def bench_effb2(args, ranks, steps, warmup):
    """EffB2-Transformer captioner (``Effb2TrmCaptioningModel`` shape, hf_wrapper.py:1115-1181): 16 kHz waveforms ->
    log-mel (HTK, top_db 120) -> EfficientNet-B2 -> 2-layer Transformer decoder, beam search (the wrapper's default,
    beam 3) with token ids back on the host; clips sharded over the ranks, no data-path collective."""
    import audiocaption_amd as A
    from audiocaption_amd import build, procedural as P
    build.build()
    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    vocab = 4981
    model = A.init_model_from_config(A.effb2_trm_config(vocab), print_fn=lambda s: None)
    model.load_state_dict(P.to_torch(P.effb2_trm_state(vocab)), strict=True)
    model = model.eval().to(dev)
    B, L = args.effb2_batch, int(args.seconds * 16000)
    wav = torch.from_numpy(P.synthetic_wav(B, L, seed=P.BASE_SEED + 200 + rank, sample_rate=16000)).to(dev)
    inp = {"mode": "inference", "wav": wav, "wav_len": [L] * B, "specaug": False, "max_length": args.max_length}
    if args.beam > 0:
        inp.update(sample_method="beam", beam_size=args.beam)
    else:
        inp.update(sample_method="greedy")

    def run_steps(n):
        """Throughput mode: every encoder is submitted up front on the encoder stream; the (host-driven) beam searches
        run one after the other on the decode stream underneath them.  --sync-steps: one blocking call per step."""
        if getattr(args, "sync_steps", False):
            last = None
            for _ in range(n):
                last = model(dict(inp))
            return last
        pend = [model.forward_async(dict(inp)) for _ in range(n)]
        last = None
        for p_ in pend:
            last = p_.result()
        return last

    # one-time setup: every chain shape a greedy run can produce (a lone batch, a full group, a shorter last group) is used
    # twice, so that no HIP-graph capture lands in the warm-up or the timed steps (as in the Cnn14 mode)
    gmax = _decode_group()
    for n_prime in [1 + 2 * gmax] + [1 + gmax + k for k in range(1, gmax) for _ in (0, 1)] + [1, 1]:
        run_steps(n_prime)
    run_steps(max(warmup, 2))
    elapsed, out = timed_steps(ranks, run_steps, steps)
    assert tuple(out["seq"].shape) == (B, args.max_length)
    # encoder alone (HBM-bound: SURVEY section 8(d)(iv) prices it at ~100 MB of activation traffic per 10 s clip)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        model.encoder(inp)
    e1.record()
    torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / 3
    alg_bytes = 100e6 * (args.seconds / 10.0) * B
    traffic, tsrc = (None, None)
    if args.seconds == 10.0:   # PMC-measured HBM bytes per clip (collected in separate --pmc passes), scaled to B
        per_clip, tsrc = pmc_traffic(EFFB2_TRAFFIC_FILE, "hbm_bytes_per_clip", False)
        traffic = per_clip * B if per_clip else None
    return {
        "encoder_roofline": {"bound": "hbm", "achieved": alg_bytes / (enc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg_bytes / (enc_ms * 1e-3) / 8e12, "traffic": traffic, "traffic_source": tsrc,
                             "encoder_ms": enc_ms,
                             "note": "algorithmic activation bytes (100 MB per 10 s clip, SURVEY 8(d)) / measured time of "
                                     "log-mel + EfficientNet-B2 for the whole batch"},
        "metric": "clips/sec encode+decode, EffB2-Transformer", "value": world * B * steps / elapsed, "unit": "clips/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (1x1 convolutions on split-bf16 operands, f32 accumulation)"
                 if os.environ.get("AUDIOCAPTION_EFFB2_GEMM", "pw") in ("bf16x3", "pw") else "f32", "data": "synthetic",
        "config": {"workload": f"EffB2-Trm, batch {B} per GPU, {args.seconds:g} s @ 16 kHz synthetic clips, "
                               + (f"beam search (beam {args.beam})" if args.beam > 0 else "greedy")
                               + f", max_length {args.max_length}, vocab {vocab} (BASELINE configs[2]; configs[4] with "
                                 "--seconds 30 --beam 4 --gpus 8)",
                   "global_batch": world * B, "parity": "efficientnet_pytorch / torchaudio are not vendored by the reference: "
                   "HIP path and oracle/effb2_path.py are held to transformers.EfficientNetModel / audio_utils witnesses "
                   "(tests/golden/g10_logmel.npz, g11_effb2.npz)",
                   "sharding": f"clips sharded over {world} rank(s), no data-path collective",
                   "schedule": "blocking model() per step" if getattr(args, "sync_steps", False) else
                               "forward_async: encoders on one HIP stream, the beam searches on a second one under the "
                               "following encoders"},
    }


def _gemm_set(lib, dev, M, shapes, nl, e0, e1, P, S):
    """The decoder's layer GEMMs at M rows on ac_pw_gemm_bf16x3 (the training step's kernel for x W^T) and on the exact-f32
    ac_gemm: (per-GEMM rows, total flops, total us, total us exact f32), layer GEMMs weighted by the number of layers."""
    rows, tot_flops, tot_us, tot_us_f32 = [], 0.0, 0.0, 0.0

    def time_call(call):
        for _ in range(3):
            call()
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 20

    for name, N, Kd in shapes:
        x, w = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
        b, y = torch.randn(N, device=dev), torch.empty(M, N, device=dev)
        wfrag = torch.empty(lib.ac_pw_gemm_packed_bytes(N, Kd), device=dev, dtype=torch.uint8)
        lib.ac_pw_gemm_pack(P(w), P(wfrag), N, Kd, S)
        us = time_call(lambda: lib.ac_pw_gemm_bf16x3(P(x), P(wfrag), P(b), P(y), M, N, Kd, 0, 0.0, None, 0, S))
        us_f32 = time_call(lambda: lib.ac_gemm(P(x), Kd, 1, P(w), 1, Kd, P(y), N, M, N, Kd, P(b), 0, 0.0, 1, 0.0, 0, None, 0,
                                               None, 0, S))
        fl = 2.0 * M * N * Kd
        rows.append({"gemm": f"{name} ({M} x {N} x {Kd})", "us": us, "tflops": fl / us / 1e6, "us_exact_f32": us_f32,
                     "tflops_exact_f32": fl / us_f32 / 1e6})
        k = nl if name != "classifier" else 1
        tot_flops += fl * k
        tot_us += us * k
        tot_us_f32 += us_f32 * k
    return rows, tot_flops, tot_us, tot_us_f32


def _decoder_rooflines(model, dev, B, vocab, max_length):
    """Decode-step weight bandwidth and teacher-forced GEMM utilisation."""
    import ctypes
    from audiocaption_amd import _lib
    lib = _lib.load()
    dec = model.decoder
    d, ffn, nl = dec.d_model, 1024, 2
    Tm = 31
    attn = torch.randn(B, Tm, dec.attn_emb_dim, device=dev)
    lens = torch.full((B,), Tm, dtype=torch.int64)
    for _ in range(3):   # third use replays the captured graph
        dec.greedy(attn, lens, max_length, model.start_idx, model.end_idx, model.pad_idx)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dec.greedy(attn, lens, max_length, model.start_idx, model.end_idx, model.pad_idx)
    e1.record()
    torch.cuda.synchronize()
    chain_ms = e0.elapsed_time(e1) / 5
    step_us = chain_ms * 1e3 / max_length
    # the one-launch form (csrc/decoder_cluster.hip: what the blocking model() call and single clips take), same method
    cluster_us = None
    if dec.cluster_covers(B, Tm, max_length) and dec.cluster_pack() is not None:
        for _ in range(3):
            dec.greedy(attn, lens, max_length, model.start_idx, model.end_idx, model.pad_idx, mode="cluster")
        e0.record()
        for _ in range(5):
            dec.greedy(attn, lens, max_length, model.start_idx, model.end_idx, model.pad_idx, mode="cluster")
        e1.record()
        torch.cuda.synchronize()
        cluster_us = e0.elapsed_time(e1) / 5 * 1e3 / max_length
    # weights one cached step touches: per layer self-attn in/out projections, cross-attn q/out (memory K/V are
    # projected once per batch), the two FFN matrices; then the classifier
    step_bytes = 4.0 * (nl * (3 * d * d + d * d + 2 * d * d + 2 * d * ffn) + vocab * d)
    step_flops = 2.0 * B * step_bytes / 4.0
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    shapes = [("self-attn qkv", 3 * d, d), ("attn out", d, d), ("ffn1", ffn, d), ("ffn2", d, ffn), ("classifier", vocab, d)]

    def gemm_set(M, sh):
        return _gemm_set(lib, dev, M, sh, nl, e0, e1, P, S)

    tf_single = gemm_set(B * 21, shapes)    # one teacher-forced pass over a 22-token caption: B x 21 positions
    # what the training step launches: ALL 21 prefix passes as one batch (231 rows per clip); the classifier only sees
    # the last position of each pass (B x 21 rows), so it is not part of this set
    tf_all = gemm_set(B * 231, shapes[:-1])
    rows, tot_flops, tot_us, tot_us_f32 = tf_single
    M = B * 21

    return {
        "decode_step": {"bound": "latency (weight stream)", "us_per_step": step_us, "rows": B,
                        "weight_bytes_per_step": step_bytes, "achieved": step_bytes / (step_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": step_bytes / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "mfma_frac": step_flops / (step_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        "us_per_step_one_launch": cluster_us,
                        "note": "one KV-cached greedy step over the whole batch (memory projection and output copies of the "
                                "call included, / max_length): us_per_step = the ten-launch chain replayed from a HIP graph (the "
                                "form that runs beside the next batch's encoder), us_per_step_one_launch = the persistent "
                                "cluster kernel the blocking call takes; neither HBM nor the matrix cores are the limit at 64 rows"},
        # One convention (as for the conv kernels): achieved = algorithmic f32 FLOPs / time; frac = MFMA FLOPs ISSUED / the
        # dense peak of the pipe the kernel runs on - ac_pw_gemm_bf16x3 issues three bf16 products per f32 product on the
        # bf16 pipe (2.5 PFLOP/s), ac_gemm one f32 product on the f32 pipe (157.3 TFLOP/s).
        "teacher_forced_gemms": {
            "bound": "mfma", "rows": M, "kernel": "pw_bf16x3_kernel (ac_pw_gemm_bf16x3: the training step's x W^T / dy W kernel)",
            "achieved": tot_flops / tot_us / 1e6, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": 3.0 * tot_flops / tot_us / 1e6 / BF16_MFMA_PEAK_TFLOPS, "issued_per_algorithmic": 3.0, "dtype": "bf16x3",
            "exact_f32": {"kernel": "ac_gemm (v_mfma_f32_32x32x2_f32)", "achieved": tot_flops / tot_us_f32 / 1e6,
                          "peak": FP32_MFMA_PEAK_TFLOPS, "frac": tot_flops / tot_us_f32 / 1e6 / FP32_MFMA_PEAK_TFLOPS},
            "per_gemm": rows,
            "all_passes_batched": {
                "rows": B * 231, "achieved": tf_all[1] / tf_all[2] / 1e6, "peak": BF16_MFMA_PEAK_TFLOPS,
                "frac": 3.0 * tf_all[1] / tf_all[2] / 1e6 / BF16_MFMA_PEAK_TFLOPS,
                "exact_f32": {"achieved": tf_all[1] / tf_all[3] / 1e6, "peak": FP32_MFMA_PEAK_TFLOPS,
                              "frac": tf_all[1] / tf_all[3] / 1e6 / FP32_MFMA_PEAK_TFLOPS},
                "per_gemm": tf_all[0],
                "note": "the same layer GEMMs at the row count the training step really launches them with: every one of "
                        "the 21 prefix passes of scheduled sampling teacher forced as ONE batch "
                        "(TrainEngine._decoder_passes), 231 rows per clip; layer GEMMs only - the classifier sees the last "
                        "position of each pass (the B x 21 rows above)"},
            "note": "ALGORITHMIC f32 FLOPs of the decoder's layer GEMMs at M = batch x 21 caption positions (layer GEMMs "
                    "weighted x2 layers) over kernel time"},
    }


# ---------------------------------------------------------------------------------------------------------------------
# headline: Cnn14Rnn-Trm greedy inference (BASELINE configs[1])
# ---------------------------------------------------------------------------------------------------------------------
def sustained_mfma_tflops(dev):
    """The dense bf16 matrix rate this part sustains (csrc/probe.hip: bare v_mfma_f32_32x32x16_bf16 loops on every CU, no
    memory traffic), measured in this run: a few launches of ~4 ms until the clock has settled, the last one reported."""
    from audiocaption_amd import _lib
    lib = _lib.load()
    blocks, iters = 512, 12000
    out = torch.empty(blocks * 256, device=dev)
    rate = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.ac_mfma_bf16_probe(out.data_ptr(), blocks, iters, _lib.stream()), "ac_mfma_bf16_probe")
        e1.record()
        torch.cuda.synchronize()
        rate = blocks * 4 * iters * 32 * 32768.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return rate


# HBM bytes per launch of a tier's dominant conv kernel come from rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in
# their own runs, tools/round_profiles.sh + tools/collect_profiles.py): a FILE per tier, named here with the round that measured
# it.  No fallback chain: a file that has gone missing is an error for the headline tier (the number would silently be another
# round's), and `traffic_source` travels in the top-level line.
TRAFFIC_FILES = {"wino43": "r06_traffic_wino43.json", "wino1d": "r03_traffic_wino1d.json", "bf16x3": "r01_traffic_bf16x3.json",
                 "f16x2": "r02_traffic_f16x2.json", "winograd": None}
EFFB2_TRAFFIC_FILE = "r06_traffic_effb2.json"


def pmc_traffic(name, key, required):
    """(value of `key`, source string) from profiles/<name>; (None, None) when the tier has no PMC file; raises when a file
    that should exist does not."""
    if name is None:
        return None, None
    tpath = os.path.join(REPO, "profiles", name)
    if not os.path.exists(tpath):
        if required:
            raise FileNotFoundError(f"{tpath}: the PMC traffic summary this bench line cites is missing "
                                    "(tools/round_profiles.sh + tools/collect_profiles.py write it)")
        return None, None
    with open(tpath) as f:
        return json.load(f).get(key), f"profiles/{name} (separate rocprofv3 --pmc passes of the same command, not this run)"


def conv_roofline(tier, events, required_traffic=False):
    """Roofline object of the dominant conv kernel from the HIP events the launch hook collected.  ONE convention:
    ``achieved`` = ALGORITHMIC direct-convolution f32 FLOPs / kernel time; ``frac`` = achieved / the dense peak of the pipe the
    kernel runs on (the roofline fraction); ``mfma_issue_frac`` = MFMA FLOPs actually ISSUED (achieved x the tier's products per
    direct product) / that peak (pipe utilisation).  The exact-f32 tier is the one exception, and says so: Winograd F(2x2,3x3)
    issues 2.25x FEWER products than the direct form on the same f32 pipe, so algorithmic / peak exceeds 1 and means nothing as
    a roofline fraction - its ``frac`` is the issued fraction, the algorithmic ratio is kept as ``algorithmic_over_peak``."""
    t = TIERS[tier]
    flops = sum(2.0 * 9 * i["Cin"] * i["Cout"] * i["H"] * i["W"] * i["B"] for _, _, i in events)
    ms = sum(s.elapsed_time(e) for s, e, _ in events)
    n = len(events)
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic, tsrc = pmc_traffic(TRAFFIC_FILES.get(t["conv_algo"]), "hbm_bytes_per_launch", required_traffic)
    issue = achieved * t["issue_ratio"] / t["peak"]
    out = {"bound": "mfma", "achieved": achieved, "peak": t["peak"], "unit": "TFLOP/s",
           "frac": achieved / t["peak"], "mfma_issue_frac": issue,
           "issued_per_algorithmic": t["issue_ratio"], "traffic": traffic, "traffic_source": tsrc,
           "kernel": t["kernel"], "launches_timed": n, "avg_launch_ms": ms / n if n else None,
           "algorithmic_gflop_per_launch": flops / n / 1e9 if n else None}
    if t["issue_ratio"] < 1.0:
        out["algorithmic_over_peak"] = out["frac"]
        out["frac"] = issue
        out["frac_note"] = "issued / peak: this tier issues fewer products than the direct form (see conv_roofline)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0, help="clip duration")
    ap.add_argument("--max-length", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tiers", "--no-f32-path", dest="no_tiers", action="store_true",
                    help="skip the measurements of the other conv tiers")
    ap.add_argument("--no-steady-state", action="store_true")
    ap.add_argument("--steady-seconds", type=float, default=2.0, help="length of the steady-state window")
    ap.add_argument("--sync-steps", action="store_true", help="blocking model(input_dict) per step (no overlap)")
    ap.add_argument("--cpu-clips", type=int, default=32, help="clips per CPU-baseline pass")
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--mode", choices=["infer", "train", "effb2"], default="infer",
                    help="train: the JSON line is the TRAINING step (BASELINE configs[3]: forward + backward + Adam, "
                         "gradients all-reduced over RCCL when N > 1); effb2: EffB2-Transformer inference "
                         "(BASELINE configs[2] / configs[4] with --seconds 30 --beam 4)")
    ap.add_argument("--effb2-batch", type=int, default=128, help="clips per GPU per step in the EffB2 measurement")
    ap.add_argument("--beam", type=int, default=3, help="beam size of the EffB2 measurement (0: greedy)")
    ap.add_argument("--no-effb2", action="store_true", help="skip the secondary EffB2-Trm measurement")
    ap.add_argument("--no-ingest", action="store_true", help="skip the host-to-device / ingest measurements")
    ap.add_argument("--no-ragged", action="store_true", help="skip the secondary Clotho-shape (ragged batch) measurement")
    ap.add_argument("--clotho-shape", action="store_true",
                    help="ragged Clotho-shape set (SURVEY 8(d)): durations ~ U[15 s, 30 s] zero-padded to the batch "
                         "maximum, wav_len = true lengths, clips dealt to the ranks by total duration")
    ap.add_argument("--train-batch", type=int, default=32, help="clips per GPU per training step")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--stub-workload", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: spawn the N ranks ourselves
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_spawn(args.gpus))

    ranks = Ranks(need_gpu=not args.stub_workload)
    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    if rank == 0 and args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus reports {world}",
              file=sys.stderr)

    if args.stub_workload:
        res = bench_stub(args, ranks)
        if rank == 0:
            print(json.dumps(res), flush=True)
        ranks.finish()
        return

    import audiocaption_amd as A
    from audiocaption_amd import build, kernels as K, procedural as P
    build.build()

    if args.mode in ("train", "effb2"):
        if args.mode == "train":
            res = bench_train(args, ranks, args.steps, max(args.warmup, 3), with_rccl=True)
        else:
            res = bench_effb2(args, ranks, args.steps, max(args.warmup, 1))
            if world > 1:
                res["ranks"] = ranks.describe()
        if rank == 0:
            print(json.dumps(res), flush=True)
        ranks.finish()
        return

    vocab = 4368  # Clotho v2 (eg_configs/clotho_v2/waveform/cnn14rnn_trm.yaml:31)
    state = P.to_torch(P.cnn14rnn_trm_state(vocab))
    model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
    model.load_state_dict(state, strict=True)
    model = model.eval().to(dev)

    L = int(args.seconds * 32000)
    B = args.batch
    audio_seconds = args.seconds * B
    NROT = 4   # resident input batches the timed steps rotate over (4 x 82 MB at the default shape: beyond the 256 MiB
    #            Infinity Cache, so no step finds its waveform on chip)
    if args.clotho_shape:
        # the same global set on every rank (seeded), dealt out by total duration: encoder cost ~ duration
        import numpy as np
        from audiocaption_amd.sharding import shard_by_duration
        rng = np.random.default_rng(P.BASE_SEED)
        dur = rng.uniform(15.0, 30.0, size=world * B)
        mine = shard_by_duration(dur.tolist(), world)[rank]
        B = len(mine)
        wav_len = [int(dur[i] * 32000) for i in mine]
        L = max(wav_len)
        wavs = []
        for k in range(NROT):
            full = P.synthetic_wav(B, L, seed=P.BASE_SEED + rank + 1000 * k)
            for j, n in enumerate(wav_len):
                full[j, n:] = 0.0
            wavs.append(torch.from_numpy(full).to(dev))
        audio_seconds = float(sum(wav_len)) / 32000.0
    else:
        wavs = [torch.from_numpy(P.synthetic_wav(B, L, seed=P.BASE_SEED + rank + 1000 * k)).to(dev) for k in range(NROT)]
        wav_len = [L] * B
    inputs = [{"mode": "inference", "wav": w_, "wav_len": wav_len, "specaug": False, "sample_method": "greedy",
               "max_length": args.max_length} for w_ in wavs]
    wav = wavs[0]

    def run_steps(n, sync=None):
        """n passes of the hot path over the rotating resident batches.  Default: throughput mode (forward_async: the
        encoder of step i+1 overlaps the latency-bound decode of step i on a second stream; every step is fully processed
        and its token ids are on the host before the timed region ends).  sync: one blocking model() per step."""
        if args.sync_steps if sync is None else sync:
            last = None
            for i in range(n):
                last = model(dict(inputs[i % NROT]))
            return last
        pend = [model.forward_async(dict(inputs[i % NROT])) for i in range(n)]
        last = None
        for p_ in pend:
            last = p_.result()
        return last

    cnn = model.encoder.cnn
    default_algo = cnn.conv_algo
    default_tier = ALGO_TO_TIER[default_algo]
    linear_algo = K.LINEAR_ALGO

    def measure(tier, steps, warmup, sync=None):
        """K timed steps of one conv tier with HIP events around every launch of its dominant kernel."""
        cnn.conv_algo = TIERS[tier]["conv_algo"] if tier != default_tier else default_algo
        K.LINEAR_ALGO = "f32" if tier == "f32" else linear_algo   # the exact-f32 tier: f32 GEMMs as well
        try:
            if warmup:
                run_steps(warmup, sync)
            events = []

            def hook(phase, info):
                # the dominant kernel instance: conv2 + BN + ReLU + 2x2 pool of blocks 2-5 (4 launches per step)
                if info["mode"] == 1 and info["Cout"] % 128 == 0:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()  # on torch's current stream == the kernel's launch stream
                    if phase == "pre":
                        events.append([e, None, dict(info)])
                    else:
                        events[-1][1] = e

            K.CONV_LAUNCH_HOOK = hook
            try:
                elapsed, out = timed_steps(ranks, lambda n: run_steps(n, sync), steps)
            finally:
                K.CONV_LAUNCH_HOOK = None
            return elapsed, out, events
        finally:
            cnn.conv_algo = default_algo
            K.LINEAR_ALGO = linear_algo

    # One-time setup, not a warm-up step: the decode chain is replayed from a HIP graph that is captured on the SECOND
    # use of a shape (audiocaption_amd/transformer_decoder.py), so the shape is used twice here and neither the W warm-up
    # steps nor the K timed steps contain a graph capture - the same role as loading the weights.
    if not args.sync_steps:
        # forward_async decodes up to AUDIOCAPTION_DECODE_GROUP submissions as one chain: every chain shape a run can
        # produce (a lone batch, a full group, the shorter groups left at the end of a run) is used twice here
        gmax = _decode_group()
        for n_prime in [1 + 2 * gmax] + [1 + gmax + k for k in range(1, gmax) for _ in (0, 1)] + [1, 1]:
            run_steps(n_prime)
    # ---- timed region: exactly K steps of the default tier ----
    elapsed, out, events = measure(default_tier, args.steps, args.warmup)
    elapsed_own = ranks.last_own
    ref_steps = min(int((out["unfinished_cnt"].cpu() > 0).sum().item()) + 1, args.max_length)
    # the headline line cites this round's PMC file: missing = error (AUDIOCAPTION_TRAFFIC_OPTIONAL=1: the collection passes
    # themselves, tools/round_profiles.sh, which run before the file exists)
    headline_roof = conv_roofline(default_tier, events, required_traffic=os.environ.get("AUDIOCAPTION_TRAFFIC_OPTIONAL") != "1")
    tiers = {default_tier: {"value": world * B * args.steps / elapsed, "unit": "clips/s", "ms_per_step": elapsed / args.steps * 1e3,
                            "steps": args.steps, "conv_algo": default_algo, "dtype": TIERS[default_tier]["dtype"],
                            "precision_gate": TIERS[default_tier]["gate"], "roofline": headline_roof}}

    # ---- steady state: >= 2 s of steps (DVFS under sustained MFMA load) ----
    steady = None
    if not args.no_steady_state:
        n_ss = int(min(1200, max(args.steps, math.ceil(args.steady_seconds / (elapsed / args.steps)))))
        ss_elapsed, _, ss_events = measure(default_tier, n_ss, 0)
        ss_roof = conv_roofline(default_tier, ss_events)
        steady = {"steps": n_ss, "seconds": ss_elapsed, "ms_per_step": ss_elapsed / n_ss * 1e3,
                  "value": world * B * n_ss / ss_elapsed, "unit": "clips/s",
                  "roofline_frac": ss_roof["frac"], "roofline_avg_launch_ms": ss_roof["avg_launch_ms"]}

    # ---- the other conv tiers, same hook, same schedule ----
    extra = {}
    if not args.no_tiers:
        for tier in ("f32", "bf16x3", "f16x2", "wino1d", "wino43"):
            if tier in tiers:
                continue
            n_t = max(5, args.steps // 2)
            try:
                t_el, _, t_ev = measure(tier, n_t, 2)
                tiers[tier] = {"value": world * B * n_t / t_el, "unit": "clips/s", "ms_per_step": t_el / n_t * 1e3,
                               "steps": n_t, "conv_algo": TIERS[tier]["conv_algo"], "dtype": TIERS[tier]["dtype"],
                               "precision_gate": TIERS[tier]["gate"], "roofline": conv_roofline(tier, t_ev)}
            except Exception as e:  # noqa: BLE001
                tiers[tier] = {"error": f"{type(e).__name__}: {e}"}

    # ---- the reference's own call: one blocking model(input_dict) per step (run.py:45,51; base.py:212-224) ----
    # The blocking call decodes with the one-launch cluster kernel (csrc/decoder_cluster.hip), which ends like the reference's
    # loop when every row has emitted <end> - with these synthetic clips after a few steps.  The figure reported as
    # value_blocking_model_call runs ALL max_length steps (AUDIOCAPTION_CLUSTER_EARLY_STOP=0: the same decode work as the
    # headline and as earlier rounds); "as_called" is the call as a user gets it; "launch_chain" the ten-launches-per-step form.
    blocking = None
    try:
        n_b = max(5, args.steps // 2)

        def blocking_run(**env):
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                el, _, _ = measure(default_tier, n_b, 2, sync=True)
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            return {"value": world * B * n_b / el, "unit": "clips/s", "ms_per_step": el / n_b * 1e3, "steps": n_b}

        blocking = blocking_run(AUDIOCAPTION_CLUSTER_EARLY_STOP="0")
        blocking["decode_steps_executed"] = args.max_length
        blocking["as_called"] = blocking_run()
        blocking["launch_chain"] = blocking_run(AUDIOCAPTION_GREEDY="chain")
    except Exception as e:  # noqa: BLE001
        blocking = {"error": f"{type(e).__name__}: {e}"}

    # ---- batch-1 latency of the blocking call: what demo.py / the HF surface model(audio, audio_length) run ----
    latency = {}
    try:
        one = {"mode": "inference", "wav": wavs[0][:1].contiguous(), "wav_len": wav_len[:1], "specaug": False,
               "max_length": args.max_length}
        os.environ["AUDIOCAPTION_CLUSTER_EARLY_STOP"] = "0"   # all max_length steps, like the launch chain (equal work)
        for name, kw in (("greedy", {"sample_method": "greedy"}), ("beam3", {"sample_method": "beam", "beam_size": 3}),
                         ("greedy_as_called", {"sample_method": "greedy"})):
            if name == "greedy_as_called":
                os.environ.pop("AUDIOCAPTION_CLUSTER_EARLY_STOP", None)
            for _ in range(3):
                model(dict(one, **kw))
            ts = []
            for _ in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o1 = model(dict(one, **kw))
                _ = o1["seq"].cpu()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            latency[name] = sorted(ts)[len(ts) // 2]
    except Exception as e:  # noqa: BLE001
        latency["error"] = f"{type(e).__name__}: {e}"
    finally:
        os.environ.pop("AUDIOCAPTION_CLUSTER_EARLY_STOP", None)

    # the log-mel kernel on its own: HBM-bound (1.54 MB per 10 s clip: waveform in, log-mel out)
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pk = cnn._pack(dev)
    hp0 = cnn.geometry(L)[2][0]
    K.logmel(wav, cnn._tables, pk["bn0"][0], pk["bn0"][1], rows_per_clip=hp0, channels_last=True)
    m0.record()
    for i in range(12):
        K.logmel(wavs[i % NROT], cnn._tables, pk["bn0"][0], pk["bn0"][1], rows_per_clip=hp0, channels_last=True)
    m1.record()
    torch.cuda.synchronize()
    mel_ms = m0.elapsed_time(m1) / 12
    mel_bytes = 1.54e6 * (args.seconds / 10.0) * B
    extra["mel_roofline"] = {"bound": "hbm", "achieved": mel_bytes / (mel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": mel_bytes / (mel_ms * 1e-3) / 8e12, "traffic": None, "kernel": "logmel_kernel<1024>",
                             "avg_launch_ms": mel_ms,
                             "note": "wave-per-frame 1024-point FFT + mel + dB + bn0 in one pass; 1.54 MB algorithmic bytes "
                                     "per 10 s clip but 0.09 GFLOP of FFT per clip, so the kernel is FFT-issue bound, not "
                                     "HBM bound"}
    # the decoder's dense GEMMs.  One cached decode step is a latency chain over ~12 MB of
    # weights (no matrix-bound regime exists at 64 rows); the "MFMA utilisation on the decoder GEMMs" figure is only
    # meaningful on the teacher-forced training shape (M = B x 21 rows), measured here on the training step's GEMM kernels.
    try:
        extra["decoder_roofline"] = _decoder_rooflines(model, dev, B, vocab, args.max_length)
    except Exception as e:  # secondary measurement: never lose the headline line over it
        extra["decoder_roofline"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_train:
        # secondary: the training step (BASELINE configs[3]) on this GPU
        del out
        try:   # a secondary measurement must never cost the headline line
            tr = bench_train(args, ranks, args.steps, 3, with_rccl=True)   # the step count of `--mode train` (the look-ahead's first step is not overlapped)
            extra["train_step"] = {k: tr[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config",
                                                       "roofline", "rccl") if k in tr}
        except Exception as e:  # noqa: BLE001
            extra["train_step"] = {"error": f"{type(e).__name__}: {e}"}
        try:   # the reference recipe's per-GPU batch at 8 ranks (run_ddp.py:68-69: batch 32 over 8 GPUs = 4 clips per GPU)
            import copy
            a4 = copy.copy(args)
            a4.train_batch = 4
            t4 = bench_train(a4, ranks, args.steps, 3, with_rccl=False, roofline=False)
            extra["train_step_b4"] = {k: t4[k] for k in ("value", "unit", "ms_per_step", "steps") if k in t4}
        except Exception as e:  # noqa: BLE001
            extra["train_step_b4"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_ragged and not args.clotho_shape and not args.sync_steps:
        # secondary: a Clotho-shaped ragged batch (durations ~ U[15 s, 30 s], zero-padded to the longest) with
        # and without dead-row skipping - 32 clips, two resident batches, forward_async like the headline
        try:
            import numpy as np
            rng = np.random.default_rng(P.BASE_SEED)
            dur = rng.uniform(15.0, 30.0, size=32)
            r_len = [int(d_ * 32000) for d_ in dur]
            r_L = max(r_len)
            r_in = []
            for k in range(2):
                full = P.synthetic_wav(32, r_L, seed=P.BASE_SEED + 77 + k)
                for j, n in enumerate(r_len):
                    full[j, n:] = 0.0
                r_in.append({"mode": "inference", "wav": torch.from_numpy(full).to(dev), "wav_len": r_len, "specaug": False,
                             "sample_method": "greedy", "max_length": args.max_length})

            def r_steps(n):
                pend = [model.forward_async(dict(r_in[i % 2])) for i in range(n)]
                for p_ in pend:
                    last = p_.result()
                return last

            ragged = {"clips": 32, "live_fraction": float(sum(r_len)) / (32.0 * r_L), "steps": 10}
            keep = os.environ.get("AUDIOCAPTION_SKIP_DEAD_ROWS")
            for tag, flag in (("skip", "1"), ("no_skip", "0")):
                os.environ["AUDIOCAPTION_SKIP_DEAD_ROWS"] = flag
                for _ in range(2):   # every chain shape the timed run produces, used twice: no graph capture inside it
                    r_steps(10)
                t_r, _ = timed_steps(ranks, r_steps, 10)
                ragged[tag] = {"value": 32 * 10 / t_r, "unit": "clips/s", "ms_per_step": t_r / 10 * 1e3}
            if keep is None:
                os.environ.pop("AUDIOCAPTION_SKIP_DEAD_ROWS", None)
            else:
                os.environ["AUDIOCAPTION_SKIP_DEAD_ROWS"] = keep
            del r_in
            extra["ragged"] = ragged
        except Exception as e:  # noqa: BLE001
            extra["ragged"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_ingest and not args.clotho_shape and not args.sync_steps:
        # secondary: the pipeline the reference's runner actually drives (run.py:22-27: every batch arrives in host memory and
        # goes to the device inside the step) and the ingest row in front of it (float16 44.1 kHz clips
        # -> float32 32 kHz batch: caption_dataset.py:110-145 + inference.py:81-111).  NOT the headline: `value` is quoted
        # with resident inputs as the contract asks.
        try:
            import collections
            import numpy as np
            host = [w_.cpu().pin_memory() for w_ in wavs]
            keep = collections.deque(maxlen=6)   # the device copies of the steps in flight stay referenced

            def h2d_steps(n):
                pend = []
                for i in range(n):
                    w_dev = host[i % NROT].to(dev, non_blocking=True)   # PCIe inside the step, like run.py:22-27
                    keep.append(w_dev)
                    pend.append(model.forward_async(dict(inputs[i % NROT], wav=w_dev)))
                last = None
                for p_ in pend:
                    last = p_.result()
                return last

            h2d_steps(5)
            t_h, _ = timed_steps(ranks, h2d_steps, args.steps)
            extra["with_h2d"] = {"value": B * args.steps / t_h, "unit": "clips/s", "ms_per_step": t_h / args.steps * 1e3,
                                 "bytes_per_step": int(host[0].numel() * 4),
                                 "note": "pinned host batch -> device inside every step (run.py:22-27), forward_async"}
            keep.clear()
            del host
            from audiocaption_amd.ingest import WaveformIngest
            rng = np.random.default_rng(P.BASE_SEED + 5)
            clips = [(f"c{i}", (0.1 * rng.standard_normal(int(44100 * args.seconds))).astype(np.float16)) for i in range(B)]
            ing = WaveformIngest(44100, 32000, device=dev)

            def ingest_steps(n):
                out_ = None
                for _ in range(n):
                    out_ = ing(clips)
                torch.cuda.synchronize(dev)
                return out_

            ingest_steps(2)
            # host-side work (threaded packing of float16 samples into pinned memory): best of three windows of 5 batches, so
            # that one scheduling hiccup of the host does not decide the figure
            t_i = min(timed_steps(ranks, ingest_steps, 5)[0] for _ in range(3))
            # the kernel alone (HIP events): what is left when the host-side packing of the float16 samples is taken out
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            src = torch.from_numpy(np.concatenate([c for _, c in clips])).to(dev)
            offs = torch.arange(B + 1, device=dev, dtype=torch.int64) * len(clips[0][1])
            olen = torch.full((B,), ing.out_length(len(clips[0][1])), dtype=torch.int32, device=dev)
            dst = torch.empty(B, int(olen[0]), device=dev)
            from audiocaption_amd import _lib as _L
            lib_ = _L.load()

            def kern():
                _L.check(lib_.ac_ingest_resample(_L.ptr(src), 1, _L.ptr(offs), _L.ptr(ing.kernel), _L.ptr(ing.tap_lo), _L.ptr(ing.tap_hi),
                                                 _L.ptr(dst), _L.ptr(olen), None, B, dst.shape[1], ing.orig, ing.new, ing.width,
                                                 _L.stream()), "ac_ingest_resample")
            kern()
            e0.record()
            for _ in range(10):
                kern()
            e1.record()
            torch.cuda.synchronize(dev)
            k_ms = e0.elapsed_time(e1) / 10
            extra["ingest"] = {"value": B * 5 / t_i, "unit": "clips/s", "ms_per_batch": t_i / 5 * 1e3,
                               "kernel_clips_per_s": B / (k_ms * 1e-3), "kernel_ms": k_ms,
                               "kernel_gbs": (src.numel() * 2 + dst.numel() * 4) / (k_ms * 1e-3) / 1e9,
                               "note": f"best of 3 windows of 5 batches; {B} float16 clips x {args.seconds:g} s @ 44.1 kHz -> float32 @ 32 kHz (WaveformIngest: host packing "
                                       "into one pinned buffer + H2D + one kernel); kernel_* = the kernel alone on resident samples"}
            del src, dst
        except Exception as e:  # noqa: BLE001
            extra["with_h2d"] = extra.get("with_h2d") or {"error": f"{type(e).__name__}: {e}"}
            extra["ingest"] = extra.get("ingest") or {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_effb2:
        # secondary: EffB2-Transformer inference (BASELINE configs[2])
        try:
            # the same step / warm-up counts as `--mode effb2`: the last grouped beam search of a run drains without an
            # encoder beside it, so a run of half the steps reads ~3 % lower for the same steady state
            eb = bench_effb2(args, ranks, args.steps, max(args.warmup, 1))
            extra["effb2_trm"] = {k: eb[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config",
                                                      "encoder_roofline")}
        except Exception as e:  # noqa: BLE001
            extra["effb2_trm"] = {"error": f"{type(e).__name__}: {e}"}
    multi = None
    if world > 1:
        # N > 1: the inference path has no collective (clips are sharded), so the line carries what makes the run
        # self-describing - where every rank sat, its own time, and the node's collective rates on a buffer of the
        # training step's gradient size (42.8 MB) - next to the max-over-ranks headline
        multi = {"ranks": ranks.describe(), "seconds_per_rank": seconds_per_rank(ranks, elapsed_own, dev),
                 "rccl": collective_record(ranks, dev)}
    sustained = {}
    if rank == 0 and TIERS[default_tier]["peak"] == BF16_MFMA_PEAK_TFLOPS:
        try:   # context for `frac` (denominator: the nominal dense peak): what a bare MFMA loop reaches on this part
            sm = sustained_mfma_tflops(dev)
            sustained = {"sustained_mfma_tflops_measured": sm,
                         "frac_of_sustained": headline_roof["achieved"] * headline_roof["issued_per_algorithmic"] / sm,
                         "sustained_note": "csrc/probe.hip: v_mfma_f32_32x32x16_bf16 only, all CUs, no memory traffic, "
                                           "measured in this run (the part holds ~1.7 of 2.4 GHz under matrix load)"}
        except Exception as e:  # noqa: BLE001
            sustained = {"sustained_mfma_tflops_measured": None, "sustained_note": f"{type(e).__name__}: {e}"}
    if rank == 0:
        clips = world * B * args.steps
        val = lambda d_, k="value": (d_ or {}).get(k) if isinstance(d_, dict) else None
        result = {
            "metric": "clips/sec (10 s @ 32 kHz) encode+greedy-decode, Cnn14_Rnn-Trm",
            "value": clips / elapsed,
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": TIERS[default_tier]["dtype"],
            "data": "synthetic",
            "config": {"workload": f"Cnn14Rnn-Trm greedy decode, batch {B} per GPU, {args.seconds:g} s @ 32 kHz "
                                   f"synthetic clips, max_length {args.max_length}, vocab {vocab} (BASELINE configs[1])",
                       "global_batch": world * B, "decode_steps_executed": args.max_length,
                       "decode_steps_reference_would_run": ref_steps, "conv_algo": default_algo,
                       "input_set": ("Clotho-shape: ragged 15-30 s clips, zero-padded, duration-balanced sharding; "
                                     "%.0f s of audio per step on rank 0" % audio_seconds) if args.clotho_shape else
                                    f"fixed-length, {NROT} resident batches in rotation",
                       "precision_gate": "logits within 1e-4 of the fp32 CPU reference, identical token ids"
                                         if default_tier != "f16x2" else "HALF-PRECISION gate only: logits within 1e-3",
                       "precision": {
                           "wino43": "f32 activations in HBM; 3x3 convolutions as Winograd along time on split-bf16 operands "
                                     "(x = hi + lo, hi*hi + hi*lo + lo*hi, f32 accumulate: 2^-16 operand error) - F(4,3) on conv "
                                     "blocks 1-6 (block 1 fused: conv1 computed inside conv2's staging; block 6 as column "
                                     "tiles; input transform in f32, filter transform in f64, both BEFORE the split) - GRU "
                                     "input projections on split-bf16 operands, everything else f32",
                           "wino1d": "f32 activations in HBM; 3x3 convolutions as F(2,3) Winograd along time on split-bf16 "
                                     "operands (x = hi + lo, hi*hi + hi*lo + lo*hi, f32 accumulate: 2^-16 operand error), "
                                     "GRU input projections on split-bf16 operands, everything else f32",
                           "bf16x3": "f32 activations; direct 3x3 convolutions on split-bf16 operands (2^-16), f32 accumulate",
                           "f32": "f32 end to end (Winograd F(2x2,3x3) on the f32 MFMA)",
                           "f16x2": "fp16 activations in HBM, fp16 hi+lo weights: NOT reference precision (logits within "
                                    "1e-3); batches with a clip under 3.2 s or an fp16 overflow re-run on bf16x3"}[default_tier],
                       "sharding": f"clips sharded over {world} rank(s), no data-path collective",
                       "schedule": "blocking model() per step" if args.sync_steps else
                                   "forward_async: encoder of step i+1 under the decode chain of step i (two HIP streams)"},
            "roofline": dict({k: headline_roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "mfma_issue_frac",
                                                            "issued_per_algorithmic", "traffic", "traffic_source", "avg_launch_ms",
                                                            "launches_timed", "kernel")}, **sustained),
            # flat scalars, all measured in this run
            "value_f32_exact": val(tiers.get("f32")),
            "value_bf16x3_direct": val(tiers.get("bf16x3")),
            "value_f16x2_half_precision_gate": val(tiers.get("f16x2")),
            "value_wino1d_f23_everywhere": val(tiers.get("wino1d")),
            "value_blocking_model_call": val(blocking),
            "value_blocking_model_call_as_called": val((blocking or {}).get("as_called")),
            "value_blocking_model_call_launch_chain": val((blocking or {}).get("launch_chain")),
            "ms_blocking_model_call": val(blocking, "ms_per_step"),
            "latency_b1_greedy_ms": latency.get("greedy"),
            "latency_b1_greedy_as_called_ms": latency.get("greedy_as_called"),
            "latency_b1_beam3_ms": latency.get("beam3"),
            "steady_state_value": val(steady),
            "train_clips_per_s": val(extra.get("train_step")),
            "train_ms_per_step": val(extra.get("train_step"), "ms_per_step"),
            "train_ms_per_step_b4": val(extra.get("train_step_b4"), "ms_per_step"),
            "train_clips_per_s_b4": val(extra.get("train_step_b4")),
            "effb2_trm_clips_per_s": val(extra.get("effb2_trm")),
            "logmel_hbm_frac": extra["mel_roofline"]["frac"],
            "value_with_h2d": val(extra.get("with_h2d")),
            "ingest_clips_per_s": val(extra.get("ingest")),
            "ingest_kernel_clips_per_s": val(extra.get("ingest"), "kernel_clips_per_s"),
            "clotho_shape_clips_per_s": val((extra.get("ragged") or {}).get("skip")),
            "clotho_shape_no_skip_clips_per_s": val((extra.get("ragged") or {}).get("no_skip")),
        }
        if multi is not None:
            result.update({"ranks": multi["ranks"], "seconds_per_rank": multi["seconds_per_rank"], "rccl": multi["rccl"]})
        details = {"tiers": tiers, "steady_state": steady, "blocking_model_call": blocking, "latency_b1_ms": latency,
                   "rooflines_other": {"logmel": extra["mel_roofline"], "decoder": extra["decoder_roofline"]},
                   "train_step": extra.get("train_step"), "train_step_b4": extra.get("train_step_b4"), "effb2_trm": extra.get("effb2_trm"), "ragged": extra.get("ragged"),
                   "with_h2d": extra.get("with_h2d"), "ingest": extra.get("ingest")}
        if not args.no_cpu_baseline and world == 1:
            try:
                from oracle import cpu_path as O  # the CPU restatement, timed as a reported baseline only
                phys = _physical_cores()
                torch.set_num_threads(phys)
                nc = args.cpu_clips
                cwav = torch.from_numpy(P.synthetic_wav(max(nc, 1), L)[:nc])
                O.caption_forward(state, cwav[:2], [L] * 2, "greedy", max_length=args.max_length, force_steps=True)   # warm-up
                t_enc, t_dec = [], []
                for _ in range(args.cpu_reps):   # encode and decode timed separately
                    c0 = time.perf_counter()
                    enc = O.cnn14_forward(state, cwav, [L] * nc)
                    enc = O.gru_forward(state, enc["attn_emb"], enc["attn_emb_len"])
                    c1 = time.perf_counter()
                    O.greedy_decode(state, enc["attn_emb"], enc["attn_emb_len"], args.max_length, force_steps=True)
                    c2 = time.perf_counter()
                    t_enc.append(c1 - c0)
                    t_dec.append(c2 - c1)
                med = lambda v: sorted(v)[len(v) // 2]
                me, md = med(t_enc), med(t_dec)
                result["cpu_baseline"] = {
                    "value": nc / (me + md), "unit": "clips/s", "cores": phys, "kind": "port",
                    "encode_clips_per_s": nc / me, "decode_clips_per_s": nc / md, "host_cpu_count": os.cpu_count(),
                    "sample": f"oracle/cpu_path.py (fp32 torch {torch.__version__} CPU ops) on all {phys} physical cores: log-mel "
                              f"+ Cnn14 + bi-GRU, then greedy decoding that re-runs the decoder on the whole prefix for "
                              f"{args.max_length} steps like the reference; one batch of {nc} clips x {args.seconds:g} s per "
                              f"pass, median of {args.cpu_reps} passes after a warm-up pass"}
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            dpath = os.path.join(REPO, "gpurun_out", "bench_details.json")
            with open(dpath, "w") as f:
                json.dump({"line": result, "details": details}, f, indent=1)
            result["details_file"] = "gpurun_out/bench_details.json (per-tier rooflines, steady state, decoder GEMMs, train_step, effb2_trm)"
        except OSError:
            pass
        print(json.dumps(result), flush=True)
    ranks.finish()


def _physical_cores():
    """Physical cores of the host (SMT siblings counted once); falls back to os.cpu_count()."""
    try:
        seen = set()
        for cpu in os.listdir("/sys/devices/system/cpu"):
            tp = os.path.join("/sys/devices/system/cpu", cpu, "topology")
            if cpu.startswith("cpu") and cpu[3:].isdigit() and os.path.isdir(tp):
                with open(os.path.join(tp, "physical_package_id")) as f1, open(os.path.join(tp, "core_id")) as f2:
                    seen.add((f1.read().strip(), f2.read().strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


if __name__ == "__main__":
    main()
