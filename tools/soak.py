"""Development soak: repeated inference steps must give identical tokens; a few hundred training iterations must stay
finite and keep reducing the loss; EffB2 beam search must be repeatable."""
import random
import numpy as np
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as Pr
from audiocaption_amd.optim import FusedAdam
from audiocaption_amd.train import TrainEngine

model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
model.load_state_dict(Pr.to_torch(Pr.cnn14rnn_trm_state(4981)), strict=True)
model = model.cuda().eval()
B, L = 64, 320000
wav = torch.from_numpy(Pr.synthetic_wav(B, L, seed=3, varied=True)).cuda()
inp = {"mode": "inference", "wav": wav, "wav_len": [L] * B, "specaug": False, "sample_method": "greedy", "max_length": 20}
ref = model(dict(inp))["seq"]
bad = 0
pend = [model.forward_async(dict(inp)) for _ in range(100)]
for p in pend:
    bad += int(not torch.equal(p.result()["seq"], ref))
for _ in range(50):
    bad += int(not torch.equal(model(dict(inp))["seq"], ref))
bref = model(dict(inp, sample_method="beam", beam_size=3))["seq"]
for _ in range(20):
    bad += int(not torch.equal(model(dict(inp, sample_method="beam", beam_size=3))["seq"], bref))
print("inference mismatches over 170 repeats:", bad)

model.train()
g = torch.Generator().manual_seed(0)
Bt = 16
cap = torch.randint(4, 4981, (Bt, 14), generator=g)
cap[:, 0], cap[:, -1] = 1, 2
batch = {"mode": "train", "wav": wav[:Bt].contiguous(), "wav_len": [L] * Bt, "specaug": True, "cap": cap.cuda(),
         "cap_len": np.array([14] * Bt), "ss_ratio": 0.8}
eng = TrainEngine(model)
opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-6)
random.seed(0)
losses = []
for it in range(300):
    r = eng.step(batch, opt)
    if it % 50 == 0 or it == 299:
        losses.append(float(r["loss"]))
print("training losses every 50 iterations:", [f"{v:.3f}" for v in losses], "finite:", all(np.isfinite(losses)))
assert bad == 0 and all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0]
print("soak OK")
