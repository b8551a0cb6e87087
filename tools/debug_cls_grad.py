import os, random, sys, numpy as np, torch
sys.path.insert(0, "tests")
from audiocaption_amd import procedural as Pr
import audiocaption_amd as A
from audiocaption_amd.loss import LabelSmoothingLoss
from test_gpu_train import _set_dropout, _cnn_attn_f32
g8 = dict(np.load("tests/golden/g8_train.npz"))
state = Pr.to_torch(Pr.cnn14rnn_trm_state(4981))
model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
model.load_state_dict(state, strict=True)
model = model.to("cuda:0").train()
_set_dropout(model, 0.0, 0.0, False)
lms = torch.from_numpy(Pr.synthetic_logmel(4, 1001)).cuda()
cnn_attn = _cnn_attn_f32(model, lms)
cap = torch.from_numpy(g8["cap"]).cuda()
cap_len = g8["cap_len"]
out = model({"mode": "train", "wav": torch.zeros(4, 320000, device="cuda"), "wav_len": g8["wav_len"].tolist(),
             "specaug": False, "cap": cap, "cap_len": cap_len, "ss_ratio": 1, "_cnn_attn": cnn_attn})
loss = LabelSmoothingLoss(smoothing=0.1)({"logit": out["logit"], "tgt": cap[:, 1:], "tgt_len": torch.as_tensor(cap_len - 1)})
loss.backward()
gr = dict(model.named_parameters())["decoder.classifier.weight"].grad
print("norm f32", float(gr.norm()), "norm f64", float(gr.double().norm()), "fixture", float(g8["tf_gnorm/decoder.classifier.weight"]))
print("sum f64", float(gr.double().sum()), "fixture", float(g8["tf_gsum/decoder.classifier.weight"]))
# oracle on CPU
from oracle import train_path as OT, cpu_path as O
o = OT.train_step_grads(state, cnn_attn.cpu(), O.cnn14_feat_len(g8["wav_len"].tolist()), cap.cpu(), cap_len, [], p_dec=0.0, p_rnn=0.0, teacher_forcing=True)
og = o["grads"]["decoder.classifier.weight"]
print("oracle norm f32", float(og.norm()), "f64", float(og.double().norm()))
d = (gr.cpu().double() - og.double()).abs()
print("max abs diff", float(d.max()), "max|og|", float(og.abs().max()), "rows with diff>1e-5:", torch.nonzero(d.max(1).values > 1e-5).flatten().tolist()[:20])
