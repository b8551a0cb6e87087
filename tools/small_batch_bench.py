#!/usr/bin/env python
"""Development tool: the Cnn14 conv stack at small batches (single clips, the reference's per-GPU training batch of 4) with
and without the weight-streaming kernel of conv blocks 5-6 (csrc/conv3x3_skinny.hip; AUDIOCAPTION_SKINNY=0 turns it off)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import cnn_encoder as CE, kernels as K, procedural as P

model = A.init_model_from_config(A.cnn14rnn_trm_config(4368), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(4368)), strict=True)
model = model.eval().to("cuda:0")
cnn = model.encoder.cnn
for B in (1, 2, 4, 8, 16):
    wav = torch.from_numpy(P.synthetic_wav(B, 320000, varied=True)).cuda()
    inp = {"wav": wav, "wav_len": [320000] * B}
    res = {}
    for flag in (True, False):
        CE.SKINNY = flag
        for _ in range(3):
            cnn(inp, skip_fc=True)
        per = {}

        def hook(phase, info, per=per):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            per.setdefault((info["algo"], info["W"], info["Cin"], info["mode"]), []).append(e)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            cnn(inp, skip_fc=True)
        torch.cuda.synchronize()
        res[flag] = (time.perf_counter() - t0) / 10 * 1e3
        K.CONV_LAUNCH_HOOK = hook
        cnn(inp, skip_fc=True)
        K.CONV_LAUNCH_HOOK = None
        torch.cuda.synchronize()
        res[(flag, "layers")] = {f"{k[0]}W{k[1]}c{k[2]}m{k[3]}": round(v[0].elapsed_time(v[1]) * 1e3) for k, v in per.items() if k[1] <= 4}
    print(f"B={B}: encoder {res[True]:.3f} ms with the skinny kernel, {res[False]:.3f} ms without")
    print("   skinny us:", res[(True, "layers")])
    print("   before us:", res[(False, "layers")])
