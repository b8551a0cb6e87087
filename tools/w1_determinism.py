"""Development probe: is the wino1d conv tier bit-reproducible run to run, alone and beside other work on the GPU?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as P, kernels as K

vocab = 4368
model = A.init_model_from_config(A.cnn14rnn_trm_config(vocab), print_fn=lambda s: None)
model.load_state_dict(P.to_torch(P.cnn14rnn_trm_state(vocab)), strict=True)
model = model.eval().cuda()
cnn = model.encoder.cnn
wav = torch.from_numpy(P.synthetic_wav(3, 48000, seed=1, varied=True)).cuda()
inp = {"wav": wav, "wav_len": [48000, 40000, 33000], "specaug": False}
for algo in ("wino1d", "bf16x3"):
    cnn.conv_algo = algo
    ref = cnn(dict(inp))["attn_emb"].clone()
    bad = 0
    for i in range(20):
        out = cnn(dict(inp))["attn_emb"]
        bad += int(not torch.equal(out, ref))
    print(algo, "alone: runs differing from the first:", bad)
    # beside a busy second stream
    s2 = torch.cuda.Stream()
    x = torch.randn(4096, 4096, device="cuda")
    bad = 0
    for i in range(20):
        with torch.cuda.stream(s2):
            for _ in range(20):
                y = torch.sin(x) * 1.0001
        out = cnn(dict(inp))["attn_emb"]
        torch.cuda.synchronize()
        d = float((out - ref).abs().max())
        bad += int(d != 0.0)
        if d != 0.0 and bad < 4:
            print("   max diff", d)
    print(algo, "beside a second stream: runs differing:", bad)
# layer level
torch.manual_seed(0)
for (B, H, Hp, W, Cin, Cout, mode) in [(3, 150, 160, 32, 64, 128, 0), (3, 150, 160, 32, 128, 128, 1), (3, 75, 80, 16, 128, 256, 0), (3, 37, 40, 8, 256, 512, 1), (3, 18, 20, 4, 512, 1024, 1), (3, 9, 10, 2, 1024, 2048, 0), (3, 9, 10, 2, 2048, 2048, 2)]:
    x = torch.randn(B * Hp, W, Cin, device="cuda")
    x.view(B, Hp, W, Cin)[:, H:] = 0
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * (2.0 / (9 * Cin)) ** 0.5
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
    shape = (B * Hp, W, Cout) if mode == 0 else ((B * Hp // 2, W // 2, Cout) if mode == 1 else (B, H, Cout))
    wp = K.pack_conv_weight_wino1d_frag(w)
    outs = []
    for i in range(12):
        out = torch.full(shape, 7.0, device="cuda")
        if i % 2:
            with torch.cuda.stream(s2):
                for _ in range(10):
                    y = torch.sin(x) * 1.0001
        K.conv3x3_bn_relu_wino1d(x, wp, sc, sh, out, B, Hp, H, W, Cin, Cout, mode)
        torch.cuda.synchronize()
        outs.append(out)
    bad = sum(int(not torch.equal(o, outs[0])) for o in outs)
    print(f"layer W={W} {Cin}->{Cout} mode{mode}: differing runs {bad}", [float((o - outs[0]).abs().max()) for o in outs if not torch.equal(o, outs[0])][:3])
