"""Development: the training look-ahead (frozen Cnn14 of the next iteration on a side stream) against the plain loop, step by
step: losses, time per step, skipped updates (GRU partner timeouts)."""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocaption_amd as A
from audiocaption_amd import procedural as Pr
from audiocaption_amd.optim import FusedAdam
from audiocaption_amd.train import TrainEngine

ss = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
state = Pr.to_torch(Pr.cnn14rnn_trm_state(4981))
model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
model.load_state_dict(state, strict=True)
model = model.to("cuda:0")
B, L = 2, 96000
batches = []
for k in range(2):
    wav = torch.from_numpy(Pr.synthetic_wav(B, L, seed=4 + k, varied=True)).cuda()
    cap = torch.tensor([[1, 9 + k, 30, 2, 0], [1, 7, 7 + k, 12, 2]])
    batches.append({"mode": "train", "wav": wav, "wav_len": [L, L - 16000 * k], "specaug": True, "cap": cap.cuda(),
                    "cap_len": np.array([4, 5]), "ss_ratio": ss})
for look in ((False,) if os.environ.get('PROBE_LOOK', '1') == '0' else (True,)):
    model.load_state_dict(state, strict=True)
    model.train()
    random.seed(3)
    eng = TrainEngine(model, seed=77)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    out = []
    for it in range(6):
        nxt = batches[(it + 1) % 2] if look and it < 5 else None
        if os.environ.get("PROBE_SYNC", "0") == "1":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = eng.step(batches[it % 2], opt, next_batch=nxt, use_graph=os.environ.get("PROBE_GRAPH", "1") == "1")
        loss = float(r["loss"])
        out.append((round(loss, 6), round(float(r["total_norm"]), 5), round(float(eng.flat.grad.abs().sum()), 3), round(float(eng.flat.flat.abs().sum()), 3)))
    print("look-ahead" if look else "plain     ", out, "skipped", eng.skipped_updates(), "gru timeouts", eng.gru_timeout())
