"""Development tool: host time of forward_async submissions (is the host ahead of the device?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as P
vocab = 4368
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
model = model.eval().to("cuda:0")
B = 64
wav = torch.from_numpy(P.synthetic_wav(B, 320000, varied=True)).cuda()
inp = {"mode": "inference", "wav": wav, "wav_len": [320000] * B, "specaug": False, "sample_method": "greedy", "max_length": 20}
for _ in range(4): model.forward_async(dict(inp)).result()
torch.cuda.synchronize()
t0 = time.perf_counter(); ts = []
pend = []
for _ in range(12):
    pend.append(model.forward_async(dict(inp))); ts.append(time.perf_counter() - t0)
print("submission times (ms):", " ".join(f"{t * 1e3:.1f}" for t in ts))
rs = []
for p in pend:
    p.result(); rs.append(time.perf_counter() - t0)
print("result times (ms):    ", " ".join(f"{t * 1e3:.1f}" for t in rs))
