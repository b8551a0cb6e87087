"""Development tool: the encoder (log-mel + Cnn14 + GRU) launched eagerly vs replayed from one HIP graph - how much of the
encoder's wall time is launch gaps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiocaption_amd as A
from audiocaption_amd import build, procedural as P

build.build()
vocab = 4368
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
model = model.eval().cuda()
B, L = 64, 320000
wav = torch.from_numpy(P.synthetic_wav(B, L, seed=1)).cuda()
d = {"wav": wav, "wav_len": [L] * B}


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for _ in range(3):
        ref = model.encoder(d)
    print(f"eager encoder           {timed(lambda: model.encoder(d)):.3f} ms")
    print(f"eager cnn only          {timed(lambda: model.encoder.cnn(d, skip_fc=True)):.3f} ms")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model.encoder(d)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model.encoder(d)
    print(f"graph replay encoder    {timed(g.replay):.3f} ms")
    g.replay()
    torch.cuda.synchronize()
    print("equal", bool(torch.equal(out["attn_emb"], ref["attn_emb"])))
