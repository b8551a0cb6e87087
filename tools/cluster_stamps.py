#!/usr/bin/env python
"""Development tool: stage timestamps inside the one-launch greedy search (csrc/decoder_cluster.hip built with
-DAC_CLUSTER_STAMPS): where the ~65 us of a step go for row 0 / part 0 at step 5.
    python tools/cluster_stamps.py --build     # here (no GPU): tools/bin/libcluster_stamps.so
    python tools/cluster_stamps.py [B]         # on the GPU box"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "bin", "libcluster_stamps.so")
if "--build" in sys.argv:
    from audiocaption_amd import build as B
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = [B._hipcc(), "-x", "hip", os.path.join(B.CSRC, "decoder_cluster.hip"), "-shared", "-o", SO, "-DAC_CLUSTER_STAMPS"] + B.FLAGS + B.NO_PACKED_F32
    subprocess.check_call(cmd)
    print("built", SO)
    sys.exit(0)
import torch
import audiocaption_amd as A
from audiocaption_amd import _lib, procedural as P
from audiocaption_amd._lib import ptr, stream

Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dec = A.TransformerDecoder(emb_dim=256, vocab_size=4368, fc_emb_dim=512, attn_emb_dim=512, dropout=0.2, nlayers=2)
dec.load_state_dict(P.to_torch(P.decoder_state("", 4368)))
dec = dec.eval().cuda()
torch.manual_seed(0)
attn = torch.randn(Bn, 31, 512, device="cuda")
lens = torch.full((Bn,), 31)
for _ in range(2):
    dec.greedy(attn, lens, 20, 1, 2, 0, mode="cluster")
torch.cuda.synchronize()
st = list(dec._greedy_state.values())[-1]
lib = ctypes.CDLL(SO)
lib.ac_trm_greedy_cluster.restype = ctypes.c_int
lib.ac_trm_greedy_cluster.argtypes = _lib.SIGNATURES["ac_trm_greedy_cluster"][1]
w = ctypes.byref(dec.weights())
for rep in range(3):
    rc = lib.ac_trm_greedy_cluster(w, ptr(st["cluster_pk"]), ptr(st["memkv"]), ptr(st["mem_len"]), Bn, 31, 20, 1, 2, 0, ptr(st["seq"]),
                                   ptr(st["logit"]), ptr(st["sampled_logprob"]), ptr(st["embed"]), ptr(st["unfinished_cnt"]),
                                   ptr(st["cluster_ws"]), 0, stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert lib.ac_cluster_stamps_read(buf) == 0
t = [int(x) for x in buf]
names = {0: "embed", 40: "layers done", 41: "classifier (logits of the quarter)", 42: "max / sum-exp of the quarter", 43: "exchange 4", 44: "bookkeeping"}
for l in range(2):
    for k, n in ((1, "qkv product"), (2, "self attention"), (3, "out product"), (4, "exchange 1"), (5, "LN1"), (6, "cross q product"),
                 (7, "cross attention"), (8, "out product + exchange 2"), (9, "LN2"), (10, "ffn 1"), (11, "ffn 2"), (12, "exchange 3")):
        names[k + 12 * l] = f"layer {l}: {n}"
order = sorted(k for k in names if t[k])
prev = t[0]
print(f"B = {Bn}; row 0, part 0, step 5; microseconds (100 MHz clock)")
for k in order:
    print(f"  {names[k]:42s} +{(t[k] - prev) / 100:6.2f}   at {(t[k] - t[0]) / 100:6.2f}")
    prev = t[k]
