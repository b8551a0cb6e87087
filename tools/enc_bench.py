"""Development tool: the encoder alone (log-mel + Cnn14 + GRU), back to back on one stream, vs the whole step."""
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
for name, fn in (("encoder only", lambda: model.encoder(dict(inp))),
                 ("cnn only", lambda: model.encoder.cnn.encode(wav)),
                 ("blocking model()", lambda: model(dict(inp)))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
