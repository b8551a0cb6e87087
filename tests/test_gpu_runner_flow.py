"""The reference runner's loop (python_scripts/train_eval/run.py:77-148, base.py:212-305) replayed on the HIP classes,
built from a reference-style config through the dotted names the reference's YAML files use: ingest -> train iterations
(model(input_dict) -> loss -> backward -> clip -> optimizer -> scheduler, SWA) -> eval with beam search -> tokenizer ->
prediction file.  What a maintainer gets after `compat.install()`."""
import json
import pickle
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_reference_style_training_and_inference_loop(tmp_path, state4981):
    from audiocaption_amd import config as C
    from audiocaption_amd.ingest import WaveformIngest
    from audiocaption_amd.text import DictTokenizer, write_predictions
    from audiocaption_amd.trainer import ScheduledSampling, SwaAverager
    cfg = {
        "model": C.cnn14rnn_trm_config(4981),                       # eg_configs/audiocaps/waveform/cnn14rnn_trm.yaml
        "loss": {"type": "captioning.losses.loss.LabelSmoothingLoss", "args": {"smoothing": 0.1}},
        "optimizer": {"type": "audiocaption_amd.optim.FusedAdam", "args": {"lr": 5e-4, "weight_decay": 1e-6}},
        "lr_scheduler": {"type": "captioning.utils.lr_scheduler.ExponentialDecayScheduler", "args": {"final_lrs": 5e-7}},
        "scheduled_sampling": {"use": True, "mode": "linear", "final_ratio": 0.7},
        "trainer": {"max_grad_norm": 1.0},
        "inference_args": {"sample_method": "beam", "beam_size": 3},
    }
    model = C.init_model_from_config(cfg["model"], print_fn=lambda s: None)
    model.load_state_dict(state4981, strict=True)
    model = model.cuda()
    words = {w: i for i, w in enumerate(["<pad>", "<start>", "<end>", "<unk>"] + [f"w{i}" for i in range(4977)])}
    vocab = tmp_path / "vocab.pkl"
    vocab.write_bytes(pickle.dumps(words))
    tokenizer = DictTokenizer(str(vocab))
    model.set_index(tokenizer.bos, tokenizer.eos, tokenizer.pad)
    loss_fn = C.get_cls_from_str(cfg["loss"]["type"])(**cfg["loss"]["args"])
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = C.get_cls_from_str(cfg["optimizer"]["type"])(params, **cfg["optimizer"]["args"])
    iterations = 6
    sched = C.get_cls_from_str(cfg["lr_scheduler"]["type"])(optimizer, total_iters=iterations,
                                                            warmup_iters=max(iterations // 5, 1),
                                                            **cfg["lr_scheduler"]["args"])
    ss = ScheduledSampling(total_iters=iterations, **cfg["scheduled_sampling"])
    swa = SwaAverager(model)
    # data: 44.1 kHz float16 clips -> ingest (resample + pad) ; captions through the tokenizer
    rng = np.random.default_rng(0)
    clips = [(f"clip{i}", (0.1 * rng.standard_normal(int(44100 * d))).astype(np.float16)) for i, d in enumerate((3.0, 2.4, 2.7))]
    batch = WaveformIngest(44100, 32000)(clips)
    caps = tokenizer(["w5 w9 w100 w7", "w8 w8 w20", "w1 w2 w3 w4 w5"])
    random.seed(1)
    losses = []
    model.train()
    for it in range(iterations):
        ss_ratio = ss.step()
        sched.step()
        optimizer.zero_grad()
        output = model({"mode": "train", "wav": batch["wav"], "wav_len": batch["wav_len"], "specaug": False,
                        "cap": torch.as_tensor(caps["cap"]).cuda(), "cap_len": caps["cap_len"], "ss_ratio": ss_ratio})
        output["tgt"] = torch.as_tensor(caps["cap"])[:, 1:].cuda()
        output["tgt_len"] = torch.as_tensor(caps["cap_len"] - 1)
        loss = loss_fn(output)
        assert not torch.isnan(loss)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), cfg["trainer"]["max_grad_norm"])   # torch's own, as run.py:125
        optimizer.step()
        losses.append(float(loss))
        if it >= 3:
            swa.update_parameters(model)
    print("losses", [f"{v:.3f}" for v in losses], "lr", optimizer.param_groups[0]["lr"])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert swa.n_averaged == 3
    # evaluation: BaseRunner._inference + the prediction file
    model.eval()
    with torch.no_grad():
        out = model({"mode": "inference", "wav": batch["wav"], "wav_len": batch["wav_len"], "specaug": False,
                     **cfg["inference_args"]})
    seqs = out["seq"].cpu().numpy()
    key2pred = {aid: [text] for aid, text in zip(batch["aid"], tokenizer.decode(seqs))}
    write_predictions(key2pred, str(tmp_path / "pred.json"))
    pred = json.loads((tmp_path / "pred.json").read_text())["predictions"]
    assert [p["filename"] for p in pred] == ["clip0", "clip1", "clip2"] and all(isinstance(p["tokens"], str) for p in pred)
    # the averaged weights load back into a fresh model (run.py:350-355 saves them as swa.pth)
    fresh = C.init_model_from_config(cfg["model"], print_fn=lambda s: None)
    fresh.load_state_dict({k: v.cpu() for k, v in swa.state_dict().items()}, strict=True)
