"""Development probe: where does the training step depend on the batch a clip sits in?  (features through the hook)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiocaption_amd as A
from audiocaption_amd import procedural as Pr
from audiocaption_amd.train import TrainEngine
from audiocaption_amd.loss import _launch
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
cnn = model.encoder.cnn
B, L, Tc = 32, 320000, 22
wav = torch.from_numpy(Pr.synthetic_wav(B, L, seed=31, varied=True)).cuda()
g = torch.Generator().manual_seed(4)
cap = torch.randint(4, 4981, (B, Tc), generator=g)
cap_len = torch.randint(8, Tc + 1, (B,), generator=g)
cap_len[0] = cap_len[16] = Tc
cap[:, 0] = 1
for i, n in enumerate(cap_len.tolist()):
    cap[i, n - 1] = 2
    cap[i, n:] = 0
eng = TrainEngine(model)
feats = {}
with torch.no_grad():
    for algo in ("wino1d", "bf16x3"):
        cnn.conv_algo = algo
        feats[algo] = cnn.encode(wav).clone()
def run(attn, sl):
    n = sl.stop - sl.start
    out = eng.forward({"mode": "train", "wav": wav[sl].contiguous(), "wav_len": [L] * n, "specaug": False, "_cnn_attn": attn,
                       "cap": cap[sl].cuda(), "cap_len": cap_len[sl].numpy(), "ss_ratio": 1})
    tl = (cap_len[sl] - 1)
    count = float(tl.sum())
    logit = out["logit"].clone()
    dlogit = torch.empty_like(logit)
    _launch(logit, cap[sl][:, 1:].cuda(), tl.to(device="cuda", dtype=torch.int32), 0.1, 1.0 / count, dlogit, 1.0 / count, None)
    saved0 = 0
    saved = {k: v.clone() for k, v in (eng._saved or {}).items() if isinstance(v, torch.Tensor)}
    eng.backward(dlogit)
    return count, logit, [v.double().clone() for v in eng.flat.grad_views], saved
print("engine: gemm_algo", eng.gemm_algo)
for algo in ("wino1d", "bf16x3"):
    c, lF, gF, sF = run(feats[algo], slice(0, B))
    ca, lA, gA, sA = run(feats[algo][:16].contiguous(), slice(0, 16))
    cb, lB, gB, sB = run(feats[algo][16:].contiguous(), slice(16, B))
    dl = torch.cat([(lF[:16] - lA).abs().amax((1, 2)), (lF[16:] - lB).abs().amax((1, 2))])
    print(algo, "forward logits, per clip max|full - half|:", [f"{v:.1e}" for v in dl.tolist()])
    for k in sF:
        if k in sA and sF[k].shape[0] == B and sA[k].shape[0] == 16 and sF[k].dtype.is_floating_point:
            print("   saved", k, "first half diff", float((sF[k][:16] - sA[k]).abs().max()))
    gmax = max(float(t.abs().max()) for t in gF)
    worst = []
    for name, f, a, b in zip(eng.flat.names, gF, gA, gB):
        comb = (ca * a + cb * b) / c
        d = comb - f
        # is the difference a rescaling of the gradient?
        alpha = float((d * f).sum() / (f * f).sum())
        worst.append((float(d.abs().max()) / gmax, alpha, name))
    worst.sort(reverse=True)
    for e, al, n in worst[:5]:
        print(f"   {e:.2e}  projection of the difference on the gradient {al:+.2e}  {n}")
