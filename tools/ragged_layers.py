"""Development tool: per-layer conv time of a ragged (Clotho-shape) batch with and without dead-row skipping."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import audiocaption_amd as A
from audiocaption_amd import build, kernels as K, procedural as P

build.build()
vocab = 4368
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
model = model.eval().cuda()
B = 32
rng = np.random.default_rng(0)
dur = rng.uniform(15.0, 30.0, B)
L = int(32000 * dur.max())
lens = [int(32000 * d) for d in dur]
wav = torch.from_numpy(P.synthetic_wav(B, L, seed=1)).cuda()
for i, n in enumerate(lens):
    wav[i, n:] = 0
cnn = model.encoder.cnn
print(f"{B} clips, {dur.mean():.1f} s mean of {dur.max():.1f} s max: {dur.sum() / (B * dur.max()):.3f} live")


def run(skip):
    os.environ["AUDIOCAPTION_SKIP_DEAD_ROWS"] = "1" if skip else "0"
    ev = []

    def hook(phase, info):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if phase == "pre":
            ev.append([e, None, dict(info)])
        else:
            ev[-1][1] = e

    d = {"wav": wav, "wav_len": lens}
    with torch.no_grad():
        cnn(d, skip_fc=True)
        cnn(d, skip_fc=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            cnn(d, skip_fc=True)
        e1.record()
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1) / 5
        K.CONV_LAUNCH_HOOK = hook
        cnn(d, skip_fc=True)
        K.CONV_LAUNCH_HOOK = None
        torch.cuda.synchronize()
    return total, [(i["Cin"], i["Cout"], i["W"], i["mode"], a.elapsed_time(b)) for a, b, i in ev]


t1, l1 = run(True)
t0, l0 = run(False)
print(f"cnn total: skip {t1:.3f} ms, no skip {t0:.3f} ms")
for a, b in zip(l1, l0):
    print(f"  {a[0]:5d}->{a[1]:5d} W {a[2]:2d} mode {a[3]}: {a[4] * 1e3:7.0f} us vs {b[4] * 1e3:7.0f} us  ({a[4] / b[4]:.2f})")
print(f"  hooked layers: {sum(x[4] for x in l1):.3f} vs {sum(x[4] for x in l0):.3f} ms")

