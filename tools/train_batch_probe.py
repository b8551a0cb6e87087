"""Development probe: is the train-mode Cnn14 forward (TrainEngine) independent of the batch a clip sits in?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as Pr
from audiocaption_amd.train import TrainEngine
state = Pr.to_torch(Pr.cnn14rnn_trm_state(4981))
model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
model.load_state_dict(state, strict=True)
model = model.cuda().train()
for m in model.decoder.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if isinstance(m, torch.nn.MultiheadAttention):
        m.dropout = 0.0
model.encoder.rnn.network.dropout = 0.0
model.encoder.cnn.eval()
B, L, Tc = 32, 320000, 22
wav = torch.from_numpy(Pr.synthetic_wav(B, L, seed=31, varied=True)).cuda()
g = torch.Generator().manual_seed(4)
cap = torch.randint(4, 4981, (B, Tc), generator=g)
cap_len = torch.randint(8, Tc + 1, (B,), generator=g)
cap[:, 0] = 1
for i, n in enumerate(cap_len.tolist()):
    cap[i, n - 1] = 2
    cap[i, n:] = 0
eng = TrainEngine(model)
def fwd(sl):
    n = sl.stop - sl.start
    out = eng.forward({"mode": "train", "wav": wav[sl].contiguous(), "wav_len": [L] * n, "specaug": False,
                       "cap": cap[sl].cuda(), "cap_len": cap_len[sl].numpy(), "ss_ratio": 1})
    return eng._saved["cnn_attn"].clone(), out["logit"].clone()
a_full, l_full = fwd(slice(0, B))
a_a, l_a = fwd(slice(0, 16))
a_b, l_b = fwd(slice(16, B))
a_full2, _ = fwd(slice(0, B))
print("algo", model.encoder.cnn.conv_algo, "cnn_attn: full vs first half", float((a_full[:16] - a_a).abs().max()), "second half",
      float((a_full[16:] - a_b).abs().max()), "full twice", float((a_full - a_full2).abs().max()), "max", float(a_full.abs().max()))
print("logit: first half", float((l_full[:16] - l_a).abs().max()), "second half", float((l_full[16:] - l_b).abs().max()))
cnn = model.encoder.cnn
cnn.eval()
with torch.no_grad():
    e_full = cnn.encode(wav)
    e_a = cnn.encode(wav[:16].contiguous())
print("eval-mode encode: full vs first half", float((e_full[:16] - e_a).abs().max()), " train-mode vs eval-mode", float((a_full - e_full).abs().max()))

# gradient linearity per parameter tensor
from audiocaption_amd.loss import _launch
cap_len[0] = cap_len[16] = Tc
def grads(sl):
    n = sl.stop - sl.start
    out = eng.forward({"mode": "train", "wav": wav[sl].contiguous(), "wav_len": [L] * n, "specaug": False,
                       "cap": cap[sl].cuda(), "cap_len": cap_len[sl].numpy(), "ss_ratio": 1})
    tl = (cap_len[sl] - 1)
    count = float(tl.sum())
    logit = out["logit"]
    dlogit = torch.empty_like(logit)
    loss, _ = _launch(logit, cap[sl][:, 1:].cuda(), tl.to(device="cuda", dtype=torch.int32), 0.1, 1.0 / count, dlogit, 1.0 / count, None)
    eng.backward(dlogit)
    return count, float(loss), [v.double().clone() for v in eng.flat.grad_views]
model.train(); model.encoder.cnn.eval()
c, loss, gf = grads(slice(0, B))
ca, la, ga = grads(slice(0, 16))
cb, lb, gb = grads(slice(16, B))
c2, loss2, gf2 = grads(slice(0, B))
gmax = max(float(g_.abs().max()) for g_ in gf)
rows = []
for name, f, a, b, f2 in zip(eng.flat.names, gf, ga, gb, gf2):
    comb = (ca * a + cb * b) / c
    rows.append((float((comb - f).abs().max()) / gmax, float((f - f2).abs().max()) / gmax, name))
rows.sort(reverse=True)
for e, e2, n in rows[:8]:
    print(f"{e:.3e} (full twice: {e2:.3e})  {n}")

# which run is off?  the same slices under the other f32-grade tier
res = {}
for algo in ("wino1d", "bf16x3", "wino1d"):
    cnn.conv_algo = algo
    for nm, sl in (("full", slice(0, B)), ("A", slice(0, 16)), ("B", slice(16, B))):
        _, lo, g_ = grads(sl)
        key = (algo, nm)
        flat = torch.cat([t.flatten() for t in g_])
        if key in res:
            print(algo, nm, "repeat: max diff", float((flat - res[key][1]).abs().max()))
        res[key] = (lo, flat)
for nm in ("full", "A", "B"):
    a, b = res[("wino1d", nm)], res[("bf16x3", nm)]
    print(nm, "loss wino1d", a[0], "bf16x3", b[0], " grad diff / max", float((a[1] - b[1]).abs().max() / b[1].abs().max()))

# downstream of the conv stack only (hook): the same features -> the same gradients?  and the sensitivity to the features
def grads_hook(attn, sl):
    n = sl.stop - sl.start
    out = eng.forward({"mode": "train", "wav": wav[sl].contiguous(), "wav_len": [L] * n, "specaug": False, "_cnn_attn": attn,
                       "cap": cap[sl].cuda(), "cap_len": cap_len[sl].numpy(), "ss_ratio": 1})
    tl = (cap_len[sl] - 1)
    count = float(tl.sum())
    logit = out["logit"]
    dlogit = torch.empty_like(logit)
    loss, _ = _launch(logit, cap[sl][:, 1:].cuda(), tl.to(device="cuda", dtype=torch.int32), 0.1, 1.0 / count, dlogit, 1.0 / count, None)
    eng.backward(dlogit)
    return count, torch.cat([v.double().flatten() for v in eng.flat.grad_views])
feats = {}
with torch.no_grad():
    for algo in ("wino1d", "bf16x3", "winograd"):
        cnn.conv_algo = algo
        feats[algo] = cnn.encode(wav).clone()
print("features: wino1d vs winograd", float((feats["wino1d"] - feats["winograd"]).abs().max()), " bf16x3 vs winograd",
      float((feats["bf16x3"] - feats["winograd"]).abs().max()))
G = {}
for algo in feats:
    c_, gF = grads_hook(feats[algo], slice(0, B))
    ca_, gA = grads_hook(feats[algo][:16].contiguous(), slice(0, 16))
    cb_, gB = grads_hook(feats[algo][16:].contiguous(), slice(16, B))
    G[algo] = gF
    print(algo, "hooked: linearity error", float(((ca_ * gA + cb_ * gB) / c_ - gF).abs().max() / gF.abs().max()))
print("grad(full) wino1d vs winograd features:", float((G["wino1d"] - G["winograd"]).abs().max() / G["winograd"].abs().max()),
      " bf16x3 vs winograd:", float((G["bf16x3"] - G["winograd"]).abs().max() / G["winograd"].abs().max()))
