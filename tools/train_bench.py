"""Development probe: time of one training iteration (TrainEngine.step) on synthetic AudioCaps-shape batches."""
import argparse
import random
import time

import numpy as np
import torch

import audiocaption_amd as A
from audiocaption_amd import procedural as Pr
from audiocaption_amd.optim import FusedAdam
from audiocaption_amd.train import TrainEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--cap-len", type=int, default=22)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--vocab", type=int, default=4981)
args = ap.parse_args()

B, L = args.batch, int(args.seconds * 32000)
model = A.init_model_from_config(A.cnn14rnn_trm_config(args.vocab), print_fn=lambda s: None)
model.load_state_dict(Pr.to_torch(Pr.cnn14rnn_trm_state(args.vocab)), strict=True)
model = model.to("cuda:0").train()
wav = torch.from_numpy(Pr.synthetic_wav(B, L, seed=1)).cuda()
g = torch.Generator().manual_seed(0)
cap = torch.randint(4, args.vocab, (B, args.cap_len), generator=g)
cap[:, 0], cap[:, -1] = 1, 2
batch = {"mode": "train", "wav": wav, "wav_len": [L] * B, "specaug": False, "cap": cap.cuda(),
         "cap_len": np.array([args.cap_len] * B), "ss_ratio": 0.85}
eng = TrainEngine(model)
opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-6)
random.seed(0)
for _ in range(3):
    r = eng.step(batch, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    r = eng.step(batch, opt)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"B={B} cap_len={args.cap_len}: {1e3 * t / args.steps:.2f} ms/step ({B * args.steps / t:.0f} clips/s), "
      f"host submit {1e3 * t_host / args.steps:.2f} ms/step, loss {float(r['loss']):.4f}")
