#!/usr/bin/env python
"""Time the on-device greedy decode alone (memory prep + 20 steps, HIP-graph replay).  Development tool."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiocaption_amd as A
from audiocaption_amd import build, procedural as P

build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MODE = sys.argv[2] if len(sys.argv) > 2 else "chain"   # "chain" (csrc/decoder.hip) | "cluster" (csrc/decoder_cluster.hip)
dec = A.TransformerDecoder(emb_dim=256, vocab_size=4368, fc_emb_dim=512, attn_emb_dim=512, dropout=0.2, nlayers=2)
dec.load_state_dict(P.to_torch(P.decoder_state("", 4368)))
dec = dec.eval().cuda()
attn = torch.randn(B, 31, 512, device="cuda")
lens = torch.full((B,), 31)
for _ in range(3):
    out = dec.greedy(attn, lens, 20, 1, 2, 0, mode=MODE)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    out = dec.greedy(attn, lens, 20, 1, 2, 0, mode=MODE)
torch.cuda.synchronize()
steps = int((out["unfinished_cnt"] > 0).sum()) + 1
print(f"B={B} {MODE}: greedy decode {1e3 * (time.perf_counter() - t0) / n:.3f} ms per batch, {min(steps, 20)} steps "
      f"(graph={os.environ.get('AUDIOCAPTION_DECODE_GRAPH', '1')})")
