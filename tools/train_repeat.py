"""Development tool: how reproducible is a 6-step training trajectory run to run (eager / graph), per conv route?"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audiocaption_amd as A
from audiocaption_amd import procedural as Pr
from audiocaption_amd.optim import FusedAdam
from audiocaption_amd.train import TrainEngine
state = Pr.to_torch(Pr.cnn14rnn_trm_state(4981))
model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
model = model.to("cuda:0")
B, L = 2, 96000
batches = []
for k in range(2):
    wav = torch.from_numpy(Pr.synthetic_wav(B, L, seed=4 + k, varied=True)).cuda()
    cap = torch.tensor([[1, 9 + k, 30, 2, 0], [1, 7, 7 + k, 12, 2]])
    batches.append({"mode": "train", "wav": wav, "wav_len": [L, L - 16000 * k], "specaug": True, "cap": cap.cuda(),
                    "cap_len": np.array([4, 5]), "ss_ratio": 1.0})
def run(use_graph):
    model.load_state_dict(state, strict=True); model.train(); random.seed(3)
    eng = TrainEngine(model, seed=77)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=float(os.environ.get('EPS', '1e-8')))
    out = []
    for it in range(6):
        torch.cuda.synchronize()
        r = eng.step(batches[it % 2], opt, use_graph=use_graph)
        out.append(float(r["loss"]))
        if os.environ.get("DUMP"):
            out[-1] = (out[-1], float(eng._saved["cnn_attn"].double().sum()) if hasattr(eng, "_saved") and "cnn_attn" in eng._saved else None)
    return out
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    print("eager", run(False)); print("graph", run(True))
