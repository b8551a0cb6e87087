"""Development tool: N training iterations (HIP-graph replay, weight gradients on the side stream) on a fixed batch - the loss
must stay finite and fall, the gradient norm must stay sane, no split-GRU timeout may be raised."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import audiocaption_amd as A
from audiocaption_amd import build, procedural as P
from audiocaption_amd.optim import FusedAdam
from audiocaption_amd.train import TrainEngine

build.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(4981)), strict=True)
model = model.cuda().train()
B, L = 16, 160000
wav = torch.from_numpy(P.synthetic_wav(B, L, seed=3, varied=True)).cuda()
g = torch.Generator().manual_seed(1)
cap = torch.randint(4, 4981, (B, 12), generator=g)
cap[:, 0], cap[:, -1] = 1, 2
batch = {"mode": "train", "wav": wav, "wav_len": [L] * B, "specaug": False, "cap": cap.cuda(), "cap_len": np.array([12] * B),
         "ss_ratio": 0.85}
eng = TrainEngine(model, seed=7)
opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-6)
losses, norms = [], []
for i in range(n):
    r = eng.step(batch, opt, smoothing=0.1, max_grad_norm=1.0)
    if i % 10 == 0 or i == n - 1:
        losses.append(float(r["loss"]))
        norms.append(float(r["total_norm"]))
print("loss", [round(v, 3) for v in losses[:4]], "...", [round(v, 3) for v in losses[-3:]])
print("norm max", max(norms), "timeout", eng.gru_timeout())
ok = all(np.isfinite(losses)) and losses[-1] < losses[0] - 2.0 and max(norms) < 1e3 and not eng.gru_timeout()
print("TRAIN SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
