"""Development probe: does the greedy decode chain give the same logits while another stream runs (a) an LDS-scribbling
kernel, (b) conv kernels of each tier?"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as P, kernels as K

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
W = int(os.environ.get("PROBE_W", "16"))
B, H, Hp, Cin, Cout = 16 * 16 // W, 250, 256, 128, 256
x = torch.randn(B * Hp, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.03
sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
out = torch.empty(B * Hp, W, Cout, device="cuda")
packs = {"wino1d": K.pack_conv_weight_wino1d_frag(w), "bf16x3": K.pack_conv_weight_bf16x3_frag(w)}
fns = {"wino1d": K.conv3x3_bn_relu_wino1d, "bf16x3": K.conv3x3_bn_relu_bf16x3_gw}
torch.cuda.synchronize()
for name in (sys.argv[1:] or ["wino1d", "bf16x3"]):
    bad, worst = 0, 0.0
    for r in range(40):
        with torch.cuda.stream(s2):
            for _ in range(6):
                fns[name](x, packs[name], sc, sh, out, B, Hp, H, W, Cin, Cout, 0)
        got = dec.greedy(*args)
        torch.cuda.synchronize()
        d = float((got["logit"] - want["logit"]).abs().max())
        bad += int(d != 0.0)
        worst = max(worst, d)
    print(f"decode beside {name} conv launches on a second stream: differing {bad}/40, worst logit diff {worst}")

# which stage: the memory preparation (attn_proj + LayerNorm + cross K/V) alone, compared bit for bit
import ctypes as C
from audiocaption_amd import _lib
lib = _lib.load()
st = list(dec._greedy_state.values())[-1]
Bq, Tm, _ = st["attn_emb"].shape
w_ = C.byref(dec.weights())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
PP = lambda t: C.c_void_p(t.data_ptr())
memkv, tmp = torch.empty_like(st["memkv"]), torch.empty_like(st["tmp"])
lib.ac_trm_memory(w_, PP(st["attn_emb"]), Bq, Tm, PP(memkv), PP(tmp), S())
torch.cuda.synchronize()
ref_kv, ref_tmp = memkv.clone(), tmp.clone()
bad_kv = bad_tmp = 0
for r in range(40):
    with torch.cuda.stream(s2):
        for _ in range(6):
            fns["wino1d"](x, packs["wino1d"], sc, sh, out, B, Hp, H, W, Cin, Cout, 0)
    memkv.fill_(-7.0); tmp.fill_(-7.0)
    lib.ac_trm_memory(w_, PP(st["attn_emb"]), Bq, Tm, PP(memkv), PP(tmp), S())
    torch.cuda.synchronize()
    bad_kv += int(not torch.equal(memkv, ref_kv))
    bad_tmp += int(not torch.equal(tmp, ref_tmp))
    if not torch.equal(tmp, ref_tmp) and bad_tmp <= 3:
        d = (tmp - ref_tmp).abs()
        print("  attn_proj+LN rows differing:", d.amax(1).nonzero().flatten().tolist(), "max", float(d.max()))
print(f"memory preparation beside wino1d conv: attn_proj+LN differing {bad_tmp}/40, K/V differing {bad_kv}/40")
# the plain linear alone
xx = torch.randn(12, 512, device="cuda"); ww = torch.randn(256, 512, device="cuda"); bb = torch.randn(256, device="cuda")
yy = torch.empty(12, 256, device="cuda")
lib.ac_linear(PP(xx), PP(ww), PP(bb), PP(yy), 12, 256, 512, 512, 512, 256, 1, S()); torch.cuda.synchronize()
ref = yy.clone(); badl = 0
for r in range(40):
    with torch.cuda.stream(s2):
        for _ in range(6):
            fns["wino1d"](x, packs["wino1d"], sc, sh, out, B, Hp, H, W, Cin, Cout, 0)
    yy.fill_(-7.0)
    for _ in range(20):
        lib.ac_linear(PP(xx), PP(ww), PP(bb), PP(yy), 12, 256, 512, 512, 512, 256, 1, S())
    torch.cuda.synchronize()
    badl += int(not torch.equal(yy, ref))
print(f"ac_linear (gemm_skinny 12 x 256 x 512) beside wino1d conv: differing {badl}/40")

# one decode step beside the conv kernel: which workspace regions differ?
args1 = (enc["attn_emb"], enc["attn_emb_len"], 1, model.start_idx, model.end_idx, model.pad_idx)
os.environ["AUDIOCAPTION_DECODE_GRAPH"] = "0"
for _ in range(2):
    dec.greedy(*args1)
st1 = [v for k, v in dec._greedy_state.items() if k[3] == 1][-1]
torch.cuda.synchronize()
ref_ws = st1["ws"].clone()
ref_st = {k: st1[k].clone() for k in ("memkv", "tmp", "attn_emb", "mem_len")}
R, d, ff, V, nl = 3, 256, 1024, vocab, 2
sizes = [("x", R * d), ("x2", R * d), ("qkv", R * 3 * d), ("att", R * d), ("tmp", R * d), ("ff", R * ff), ("q2", R * d), ("lg", R * V),
         ("cacheA", 2 * nl * R * 1 * d), ("cacheB", 2 * nl * R * 1 * d)]
al = lambda n: (n + 3) & ~3
from collections import Counter
hits = Counter()
for r in range(60):
    with torch.cuda.stream(s2):
        for _ in range(3):
            fns["wino1d"](x, packs["wino1d"], sc, sh, out, B, Hp, H, W, Cin, Cout, 0)
    dec.greedy(*args1)
    torch.cuda.synchronize()
    for nm in ("memkv", "tmp", "attn_emb", "mem_len"):
        if not torch.equal(st1[nm], ref_st[nm]):
            dd = (st1[nm].float() - ref_st[nm].float()).abs()
            hits[nm + " differs, max " + str(round(float(dd.max()), 4))] += 1
    off = 0
    for name, n in sizes:
        a, b = st1["ws"][off:off + n], ref_ws[off:off + n]
        if not torch.equal(a, b):
            dd = (a - b).abs()
            if name.startswith("cache"):
                per = dd.reshape(2 * nl, R, d).amax(-1)
                hits[name + " sets(K0,V0,K1,V1) x rows: " + str((per > 0).int().tolist())] += 1
            else:
                rows = dd.reshape(R, -1).amax(-1)
                hits[name + " rows " + str((rows > 0).int().tolist())] += 1
        off += al(n)
for k, v in sorted(hits.items()):
    print(f"  {v:3d}/60  {k}")
