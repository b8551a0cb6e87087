"""Development tool: when the 6-step trajectory of tools/train_repeat.py comes out in two variants, WHAT differs after step 3?"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audiocaption_amd as A
from audiocaption_amd import procedural as Pr
from audiocaption_amd.optim import FusedAdam
from audiocaption_amd.train import TrainEngine
state = Pr.to_torch(Pr.cnn14rnn_trm_state(4981))
model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None).to("cuda:0")
B, L = 2, 96000
batches = []
for k in range(2):
    wav = torch.from_numpy(Pr.synthetic_wav(B, L, seed=4 + k, varied=True)).cuda()
    cap = torch.tensor([[1, 9 + k, 30, 2, 0], [1, 7, 7 + k, 12, 2]])
    batches.append({"mode": "train", "wav": wav, "wav_len": [L, L - 16000 * k], "specaug": True, "cap": cap.cuda(),
                    "cap_len": np.array([4, 5]), "ss_ratio": 1.0})
runs = []
for rep in range(24):
    model.load_state_dict(state, strict=True); model.train(); random.seed(3)
    eng = TrainEngine(model, seed=77)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    losses, snaps, attn = [], [], []
    for it in range(4):
        torch.cuda.synchronize()
        r = eng.step(batches[it % 2], opt, use_graph=False)
        losses.append(float(r["loss"]))
        snaps.append(eng.flat.flat.clone())
        st_ = list(eng._states.values())[-1]
        attn.append(st_["cnn_attn"].clone())
    runs.append((losses, snaps, attn, list(eng.flat.names), [v.numel() for v in eng.flat.grad_views]))
    print(rep, losses, [f"{float(a.double().sum()):.6f}" for a in attn])
ref = runs[0]
for i, r in enumerate(runs[1:], 1):
    if abs(r[0][3] - ref[0][3]) > 1e-4:
        print("run", i, "differs at step 4:", r[0][3], "vs", ref[0][3])
        for st in range(3):
            d = (r[1][st] - ref[1][st]).abs()
            big = (d > 5e-4).nonzero().flatten()
            print(f"  after step {st + 1}: params differing by > 5e-4: {big.numel()}, max diff {float(d.max()):.2e}, mean {float(d.mean()):.2e}")
            if r[2][st] is not None and ref[2][st] is not None:
                print(f"     cnn_attn of step {st + 1}: max diff {float((r[2][st] - ref[2][st]).abs().max()):.3e}")
            if 0 < big.numel() < 20:
                offs = np.cumsum([0] + r[4])
                for b in big.tolist():
                    k = int(np.searchsorted(offs, b, side="right") - 1)
                    print("      ", r[3][k] if k < len(r[3]) else "?", b - offs[k] if k < len(offs) else "")
        break
