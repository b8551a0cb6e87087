#!/usr/bin/env python
"""Development experiment: what does a chain of small dependent kernels on a second stream cost the one-workgroup-per-CU conv
kernels of the encoder stream - per LAUNCH, per CU-microsecond, or per microsecond of chain?  The chain is synthetic
(tools/tax_corunner.hip: `blocks` workgroups of `threads` threads with a chosen register / LDS footprint that stay resident for `spin` us asleep, issuing FMAs or streaming loads), replayed
from a HIP graph beside the Cnn14 stack of 64 ten-second clips.
    python tools/tax_probe.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiocaption_amd as A
from audiocaption_amd import _lib, procedural as P

vocab = 4368
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
model = model.eval().to("cuda:0")
B = 64
wav = torch.from_numpy(P.synthetic_wav(B, 320000, varied=True)).cuda()
import subprocess
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libtaxco.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-O2", "-w", "-o", so, os.path.join(here, "tax_corunner.hip")])
co = ctypes.CDLL(so)
buf = torch.zeros(1024 * 4096 * 4 + 65536, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.Stream()
NB = 12   # conv batches per measurement


def chain_graph(K, vgprs, blocks, threads, lds, spin_us, mode):
    def launch(stream_ptr):
        rc = co.corun(ctypes.c_void_p(buf.data_ptr()), vgprs, blocks, threads, lds, int(spin_us * 2000), mode, ctypes.c_void_p(stream_ptr))
        assert rc == 0, rc
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for _ in range(2):
            launch(side.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K):
            launch(torch.cuda.current_stream().cuda_stream)
    return g


def measure(graph, replays_per_batch):
    with torch.cuda.stream(main):
        for _ in range(3):
            model.encoder.cnn.encode(wav)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph is not None:
        with torch.cuda.stream(side):
            s0.record()
    with torch.cuda.stream(main):
        e0.record()
    for _ in range(NB):
        if graph is not None:
            with torch.cuda.stream(side):
                for _ in range(replays_per_batch):
                    graph.replay()
        with torch.cuda.stream(main):
            model.encoder.cnn.encode(wav)
    with torch.cuda.stream(main):
        e1.record()
    if graph is not None:
        with torch.cuda.stream(side):
            s1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / NB, (s0.elapsed_time(s1) / NB if graph is not None else 0.0)


base, _ = measure(None, 0)
base2, _ = measure(None, 0)
print(f"conv stack alone: {base:.3f} / {base2:.3f} ms per batch of {B}")
base = min(base, base2)
K = 50
MODES = {0: "asleep", 1: "FMAs", 2: "loads", 3: "stream"}
for vgprs, blocks, threads, lds, spin, mode in ((24, 1, 64, 0, 2, 0),                                     # launches alone
                                                # the register / LDS cliff (asleep: no contention, only residency)
                                                (24, 256, 256, 0, 8, 0), (56, 256, 256, 0, 8, 0), (64, 256, 256, 0, 8, 0),
                                                (72, 256, 256, 0, 8, 0), (96, 256, 256, 0, 8, 0), (128, 256, 256, 0, 8, 0),
                                                (56, 256, 256, 32768, 8, 0), (56, 256, 256, 65536, 8, 0), (56, 256, 256, 90112, 8, 0),
                                                # work that moves bytes: sharing a CU vs waiting for one
                                                (56, 256, 256, 0, 16, 3), (128, 256, 256, 0, 16, 3), (56, 256, 256, 24576, 16, 3),
                                                (128, 256, 256, 24576, 16, 3), (56, 768, 256, 0, 8, 3), (128, 768, 256, 0, 8, 3),
                                                (56, 256, 256, 0, 16, 1), (128, 256, 256, 0, 16, 1), (56, 64, 256, 0, 16, 3),
                                                (128, 64, 256, 0, 16, 3)):
    g = chain_graph(K, vgprs, blocks, threads, lds, spin, mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(10):
            g.replay()
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / 10 * 1e3
    for reps in (1, 2):
        ms, side_ms = measure(g, reps)
        n = K * reps
        print(f"{n:3d} launches per batch of {blocks:4d} x {threads:3d} threads, {vgprs:3d} VGPRs, {lds // 1024:2d} KB LDS, {spin} us {MODES[mode]:6s} "
              f"({alone * reps:.3f} ms alone): conv {ms:.3f} ms (+{ms - base:.3f}), chain {side_ms:.3f} ms per batch; "
              f"tax = {(ms - base) / (alone * reps):.2f} of the chain's own time", flush=True)
