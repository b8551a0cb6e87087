"""Development tool: throughput mode with the decode of TWO consecutive batches run as one 128-row chain."""
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
enc_s, dec_s = torch.cuda.Stream(), torch.cuda.Stream()

def run(nsteps, group):
    outs = []
    held = []
    for i in range(nsteps):
        with torch.cuda.stream(enc_s):
            enc = model.encoder(dict(inp))
            ev = torch.cuda.Event(); ev.record(enc_s)
        held.append((enc, ev))
        if len(held) == group:
            with torch.cuda.stream(dec_s):
                for _, e in held: dec_s.wait_event(e)
                attn = torch.cat([h[0]["attn_emb"] for h in held], 0) if group > 1 else held[0][0]["attn_emb"]
                lens = torch.cat([torch.as_tensor(h[0]["attn_emb_len"]) for h in held], 0)
                res = model.decoder.greedy(attn, lens, 20, model.start_idx, model.end_idx, model.pad_idx)
                outs.append(res["seq"].to("cpu", non_blocking=True))
            held = []
    torch.cuda.synchronize()
    return outs

for group in (2, 4, 8, 24, 2, 24):
    run(4, group)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = run(24, group)
    dt = (time.perf_counter() - t0) / 24
    print(f"decode every {group} batch(es): {dt * 1e3:.3f} ms per 64-clip batch, {64 / dt:.0f} clips/s")
