#!/usr/bin/env python
"""Time the replayed greedy-chain graph alone (no output clones) at a row count, wide route vs narrow route.  Development tool.
usage: decode_wide_bench.py ROWS [Tm]      env: AUDIOCAPTION_DEC_WIDE_MIN / _BM / _CLS_NTB"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiocaption_amd as A
from audiocaption_amd import build, procedural as P

build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Tm = int(sys.argv[2]) if len(sys.argv) > 2 else 31
dec = A.TransformerDecoder(emb_dim=256, vocab_size=4368, fc_emb_dim=512, attn_emb_dim=512, dropout=0.2, nlayers=2)
dec.load_state_dict(P.to_torch(P.decoder_state("", 4368)))
dec = dec.eval().cuda()
attn = torch.randn(B, Tm, 512, device="cuda")
lens = torch.full((B,), Tm)
for _ in range(3):
    out = dec.greedy(attn, lens, 20, 1, 2, 0, mode="chain")
torch.cuda.synchronize()
st = next(reversed(dec._greedy_state.values()))
g = st["graph"]
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / n
print(f"rows={B} Tm={Tm} wide_min={os.environ.get('AUDIOCAPTION_DEC_WIDE_MIN', 'default')} bm={os.environ.get('AUDIOCAPTION_DEC_WIDE_BM', 'auto')}: "
      f"{ms:.3f} ms per 20-step chain = {1e3 * ms / 20:.1f} us per step (incl. memory projection)")
