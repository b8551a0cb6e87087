"""Development probe: blocking model() vs forward_async, repeated; counts runs whose logits differ from the blocking call's.
    python tools/w1_async_probe.py <tier> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as P

vocab = 4981
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
model = model.eval().cuda()
cnn = model.encoder.cnn
cnn.conv_algo = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
big = torch.from_numpy(P.synthetic_wav(4, 320000, varied=True)).cuda()
model({"mode": "inference", "wav": big, "wav_len": [320000, 280000, 160000, 300000], "specaug": False,
       "sample_method": "greedy", "max_length": 20})
wavs = [torch.from_numpy(P.synthetic_wav(3, 48000, seed=s_, varied=True)).cuda() for s_ in (1, 2, 3)]
inputs = [{"mode": "inference", "wav": w, "wav_len": [48000, 40000, 33000], "specaug": False,
           "sample_method": "greedy", "max_length": 8} for w in wavs]
want = [model(dict(i)) for i in inputs]
bad = {"attn_emb": 0, "logit": 0, "seq": 0, "embed": 0}
worst = 0.0
for r in range(reps):
    pend = [model.forward_async(dict(i)) for i in inputs]
    got = [p.result() for p in pend]
    for k in range(3):
        for key in bad:
            if not torch.equal(want[k][key], got[k][key]):
                bad[key] += 1
        worst = max(worst, float((want[k]["logit"] - got[k]["logit"]).abs().max()))
print(sys.argv[1], "async results differing from blocking over", reps * 3, "batches:", bad, "worst logit diff", worst)
# where do the differences start?
shown = 0
for r in range(reps):
    pend = [model.forward_async(dict(i)) for i in inputs]
    got = [p.result() for p in pend]
    for k in range(3):
        d = (want[k]["embed"] - got[k]["embed"]).abs().amax(-1)      # (B, steps)
        if float(d.max()) > 0 and shown < 6:
            shown += 1
            dl = (want[k]["logit"] - got[k]["logit"]).abs().amax(-1)
            print(f"batch {k}: embed diff per (row, step):\n", d.cpu().numpy().round(5), "\n logit diff:\n", dl.cpu().numpy().round(5))
