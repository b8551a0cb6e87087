"""Development probe: the greedy decode chain beside synthetic co-runners (VGPR-hungry MFMA burners, no memory traffic)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as P

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libpollute.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-O2", "-w", "-o", so, os.path.join(here, "lds_polluter.hip")])
lib = ctypes.CDLL(so)
vocab = 4981
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
model = model.eval().cuda()
model.encoder.cnn.conv_algo = "bf16x3"
wav = torch.from_numpy(P.synthetic_wav(3, 48000, seed=1, varied=True)).cuda()
inp = {"mode": "inference", "wav": wav, "wav_len": [48000, 40000, 33000], "specaug": False, "sample_method": "greedy", "max_length": 8}
enc = model.encoder(dict(inp))
dec = model.decoder
args = (enc["attn_emb"], enc["attn_emb_len"], 8, model.start_idx, model.end_idx, model.pad_idx)
for _ in range(3):
    want = dec.greedy(*args)
s2 = torch.cuda.Stream()
sink = torch.zeros(4, device="cuda")

for vg, lds in ((224, 1024), (128, 1024), (128, 81920), (128, 60000), (1224, 1024)):
    bad, worst = 0, 0.0
    for r in range(30):
        with torch.cuda.stream(s2):
            lib.burn(ctypes.c_void_p(sink.data_ptr()), vg, 4096, lds, 4000, ctypes.c_void_p(s2.cuda_stream))
        got = dec.greedy(*args)
        torch.cuda.synchronize()
        d = float((got["logit"] - want["logit"]).abs().max())
        bad += int(d != 0.0)
        worst = max(worst, d)
    print(f"co-runner: {'VALU' if vg > 1000 else 'MFMA'} burner, {vg % 1000} VGPRs, {lds} B LDS per 256-thread workgroup: decodes differing {bad}/30, worst {worst}")
