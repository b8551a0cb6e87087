import time, torch, sys
sys.path.insert(0, '/root/repo')
import audiocaption_amd as A
from audiocaption_amd import procedural as P
vocab = 4981
model = A.init_model_from_config(A.effb2_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.effb2_trm_state(vocab)), strict=True)
model = model.eval().cuda()
B, L = 128, 160000
wav = torch.from_numpy(P.synthetic_wav(B, L, seed=5, sample_rate=16000)).cuda()
inp = {"mode": "inference", "wav": wav, "wav_len": [L] * B, "specaug": False, "max_length": 20, "sample_method": "beam", "beam_size": 3}
for _ in range(2): model(dict(inp))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): model.encoder(dict(inp))
th = time.perf_counter() - t0
torch.cuda.synchronize(); print(f"encoder: host submit {th/3*1e3:.2f} ms, total {(time.perf_counter()-t0)/3*1e3:.2f} ms per call")
enc = model.encoder(dict(inp)); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): model.forward_decoder(dict(inp), enc)
torch.cuda.synchronize(); print(f"beam decode alone: {(time.perf_counter()-t0)/3*1e3:.2f} ms per call")
t0 = time.perf_counter()
pend = [model.forward_async(dict(inp)) for _ in range(6)]
print(f"submitted 6 encoders at {(time.perf_counter()-t0)*1e3:.1f} ms")
for i, p in enumerate(pend):
    p.result(); print(f"result {i} at {(time.perf_counter()-t0)*1e3:.1f} ms")
