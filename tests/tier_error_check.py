"""Checker script (lives under tests/ because it uses the oracle): logit error of the conv tiers against the CPU oracle as a function of the clip length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as P
from oracle import cpu_path as O   # checker
vocab = 4368
state = P.to_torch(P.cnn14rnn_trm_state(vocab))
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(state, strict=True)
model = model.eval().cuda()
tiers = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16x2"]
for sec in (0.5, 1, 1.5, 2, 3, 4, 5, 6, 8):
    L = int(32000 * sec)
    lens = [L, int(L * 0.8)]
    for seed in (9, 10, 11):
        wav = P.synthetic_wav(2, L, seed=seed, varied=True)
        wav[1, lens[1]:] = 0
        wav = torch.from_numpy(wav)
        want = O.caption_forward(state, wav, lens, "greedy", max_length=8)
        st = want["steps"]
        for tier in tiers:
            model.encoder.cnn.conv_algo = tier
            out = model({"mode": "inference", "wav": wav.cuda(), "wav_len": lens, "specaug": False,
                         "sample_method": "greedy", "max_length": 8})
            print(f"{sec} s seed {seed} {tier}: tokens equal {bool(torch.equal(out['seq'][:, :st], want['seq'][:, :st]))} "
                  f"logit diff {float((out['logit'][:, :st].cpu() - want['logit'][:, :st]).abs().max()):.2e} "
                  f"frames {want['attn_emb_len'].tolist()}")
