"""Generate golden fixtures by running the REFERENCE (imported from /root/reference) on CPU.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

Recipe: inert ``sys.modules`` stubs for the non-arithmetic modules the
reference imports (toml, h5py, wandb, torchvision, torchlibrosa) and a stand-in
``torchaudio.transforms`` whose MelSpectrogram returns a preset log-mel tensor (so no golden
ever depends on the stand-in: every fixture starts at the log-mel input), then the model is
built through the reference's own ``init_model_from_config`` from its AudioCaps YAML, the
procedural weights of ``audiocaption_amd.procedural`` are loaded with ``load_state_dict`` and
the reference modules are run.  Each result is also compared with ``oracle/cpu_path.py`` here,
which is what pins the oracle.  Only inputs/outputs are written (``tests/golden/*.npz``).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

_PRESET = {"lms": None}


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("toml", loads=lambda s: {}, load=lambda f: {})
    mod("h5py")
    mod("wandb", run=None)
    mod("torchvision")

    class SpecAugmentation(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    tl = mod("torchlibrosa")
    tl.augmentation = mod("torchlibrosa.augmentation", SpecAugmentation=SpecAugmentation)

    class MelSpectrogram(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, wav):
            return _PRESET["lms"]

    class AmplitudeToDB(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    ta = mod("torchaudio")
    ta.transforms = mod("torchaudio.transforms", MelSpectrogram=MelSpectrogram, AmplitudeToDB=AmplitudeToDB)
    ta.functional = mod("torchaudio.functional")


def main():
    _install_stubs()
    torch.manual_seed(0)
    from captioning.utils import train_util  # noqa: E402  (reference)
    from audiocaption_amd import procedural as P
    from oracle import cpu_path as O

    out_dir = os.path.dirname(os.path.abspath(__file__))
    cfg = train_util.load_config(os.path.join(REF, "eg_configs/audiocaps/waveform/cnn14rnn_trm.yaml"))
    V = cfg["model"]["decoder"]["args"]["vocab_size"]
    model = train_util.init_model_from_config(cfg["model"], print_fn=lambda s: None)
    state_np = P.cnn14rnn_trm_state(vocab_size=V)
    state = P.to_torch(state_np)
    missing = set(model.state_dict().keys()) ^ set(state.keys())
    assert not missing, missing
    model.load_state_dict(state, strict=True)
    model.eval()
    torch.set_grad_enabled(False)

    report = {}

    def cmp(name, a, b):
        d = float((a - b).abs().max())
        report[name] = d
        print(f"  oracle vs reference  {name:28s} max|diff| = {d:.3e}")
        return d

    # ---- G1: log-mel -> Cnn14 blocks -> attn_emb -------------------------------------------
    lms = torch.from_numpy(P.synthetic_logmel(2, 1001))
    _PRESET["lms"] = lms
    cnn = model.encoder.cnn
    x = lms.transpose(1, 2).unsqueeze(1)
    x = cnn.bn0(x.transpose(1, 3)).transpose(1, 3)
    ref_blocks = []
    for b in range(1, 7):
        x = getattr(cnn, f"conv_block{b}")(x, pool_size=(2, 2) if b < 6 else (1, 1), pool_type="avg")
        ref_blocks.append(x)
    ref_cnn = cnn({"wav": torch.zeros(2, 320000), "wav_len": [320000, 320000], "specaug": False})
    o_attn, o_blocks = O.cnn14_from_logmel(state, lms, return_blocks=True)
    for b in range(6):
        cmp(f"G1 conv_block{b + 1}", o_blocks[b], ref_blocks[b])
    cmp("G1 attn_emb", o_attn, ref_cnn["attn_emb"])
    assert torch.equal(ref_cnn["attn_emb_len"], O.cnn14_feat_len([320000, 320000]))
    g1 = {"attn_emb": ref_cnn["attn_emb"].numpy()}
    for b in range(6):
        blk = ref_blocks[b]
        g1[f"block{b + 1}_sum"] = blk.double().sum(dim=(2, 3)).numpy()  # (B, C) checksums
        g1[f"block{b + 1}_abs_sum"] = blk.double().abs().sum(dim=(2, 3)).numpy()
        g1[f"block{b + 1}_corner"] = blk[:, :8, :4, :2].numpy()
    np.savez_compressed(os.path.join(out_dir, "g1_cnn14.npz"), **g1)

    # ---- G2: GRU on ragged lengths -----------------------------------------------------------
    attn = ref_cnn["attn_emb"]
    for tag, lens in (("full", [31, 31]), ("ragged", [31, 20]), ("short", [7, 25])):
        ref_rnn = model.encoder.rnn({"attn": attn, "attn_len": torch.tensor(lens)})
        o_rnn = O.gru_forward(state, attn, lens)
        cmp(f"G2 {tag} attn_emb", o_rnn["attn_emb"], ref_rnn["attn_emb"])
        cmp(f"G2 {tag} fc_emb", o_rnn["fc_emb"], ref_rnn["fc_emb"])
        np.savez_compressed(os.path.join(out_dir, f"g2_gru_{tag}.npz"), lens=np.array(lens),
                            attn_emb=ref_rnn["attn_emb"].numpy(), fc_emb=ref_rnn["fc_emb"].numpy())

    # ---- G4 inputs: B=4 ragged wav_len (reference smoke shapes cnn_encoder.py:845-849) -------
    wav_len = [320000, 280000, 160000, 300000]
    lms4 = torch.from_numpy(P.synthetic_logmel(4, 1001))
    _PRESET["lms"] = lms4
    inp = {"mode": "inference", "wav": torch.zeros(4, 320000), "wav_len": wav_len, "specaug": False}
    ref_g = model(dict(inp, sample_method="greedy", max_length=20))
    enc_attn, enc_len = ref_g["attn_emb"], ref_g["attn_emb_len"]
    o_enc = O.gru_forward(state, O.cnn14_from_logmel(state, lms4), O.cnn14_feat_len(wav_len))
    cmp("G4 encoder attn_emb", o_enc["attn_emb"], enc_attn)
    cmp("G4 encoder fc_emb", o_enc["fc_emb"], ref_g["fc_emb"])
    assert torch.equal(enc_len, o_enc["attn_emb_len"]), (enc_len, o_enc["attn_emb_len"])

    # ---- G3: decoder, full (teacher-forced) sequence incl. a pad token ------------------------
    g = torch.Generator().manual_seed(7)
    word = torch.randint(4, V, (4, 12), generator=g)
    word[:, 0] = 1
    word[1, 5] = 0   # a generated pad-id token is masked as a key (transformer_model.py:55)
    word[2, 9:] = 0
    dec_in = {"word": word, "attn_emb": enc_attn, "attn_emb_len": enc_len, "cap_padding_mask": word == 0}
    ref_d = model.decoder(dec_in)
    o_d = O.decoder_forward(state, word, enc_attn, enc_len, word == 0)
    cmp("G3 decoder embed", o_d["embed"], ref_d["embed"])
    cmp("G3 decoder logit", o_d["logit"], ref_d["logit"])
    top_v, top_i = ref_d["logit"].topk(8, dim=-1)
    np.savez_compressed(os.path.join(out_dir, "g3_decoder.npz"), word=word.numpy(),
                        attn_emb=enc_attn.numpy(), attn_emb_len=enc_len.numpy(),
                        embed=ref_d["embed"].numpy(), logit_top_val=top_v.numpy(), logit_top_idx=top_i.numpy(),
                        logit_row0=ref_d["logit"][0].numpy())

    # ---- G4: greedy ------------------------------------------------------------------------
    o_g = O.greedy_decode(state, o_enc["attn_emb"], o_enc["attn_emb_len"], 20)
    print("  reference greedy seq:\n", ref_g["seq"].numpy())
    assert torch.equal(o_g["seq"], ref_g["seq"]), (o_g["seq"], ref_g["seq"])
    steps = o_g["steps"]
    cmp("G4 greedy logit", o_g["logit"][:, :steps], ref_g["logit"][:, :steps])
    cmp("G4 greedy logprob", o_g["sampled_logprob"][:, :steps], ref_g["sampled_logprob"][:, :steps])
    cmp("G4 greedy embed", o_g["embed"][:, :steps], ref_g["embed"][:, :steps])
    top2 = ref_g["logit"][:, :steps].topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1])
    print(f"  greedy steps executed {steps}; min top1-top2 logit gap {float(gap.min()):.3e}")
    np.savez_compressed(
        os.path.join(out_dir, "g4_greedy.npz"), wav_len=np.array(wav_len), steps=np.array(steps),
        attn_emb=enc_attn.numpy(), fc_emb=ref_g["fc_emb"].numpy(), attn_emb_len=enc_len.numpy(),
        seq=ref_g["seq"].numpy(), sampled_logprob=ref_g["sampled_logprob"][:, :steps].numpy(),
        embed=ref_g["embed"][:, :steps].numpy(), top2_gap=gap.numpy(),
        logit_top_val=ref_g["logit"][:, :steps].topk(8, dim=-1).values.numpy(),
        logit_top_idx=ref_g["logit"][:, :steps].topk(8, dim=-1).indices.numpy())

    # ---- G5: beam 3 and 4 ---------------------------------------------------------------------
    g5 = {"attn_emb_len": enc_len.numpy()}
    for k in (3, 4):
        ref_b = model(dict(inp, sample_method="beam", beam_size=k, max_length=20))
        o_b = O.beam_search(state, o_enc["attn_emb"], o_enc["attn_emb_len"], k, 20)
        print(f"  reference beam-{k} seq:\n", ref_b["seq"].numpy())
        assert torch.equal(o_b["seq"], ref_b["seq"]), (o_b["seq"], ref_b["seq"])
        g5[f"seq_beam{k}"] = ref_b["seq"].numpy()
        g5[f"score_beam{k}"] = o_b["score"].numpy()
    np.savez_compressed(os.path.join(out_dir, "g5_beam.npz"), **g5)

    # ---- G4b / G5b: the same searches on HIGH-ENTROPY decoder draws (audiocaption_amd/procedural.py DIVERSE) ----------
    # g4 / g5 above repeat one token and hardly ever change the parent beam, so a wrong KV-cache re-gather could pass
    # them.  These draws are checked HERE to exercise what they are for, on the reference's own outputs.
    def max_repeat(seqs):
        worst = 0
        for row in seqs.tolist():
            toks = [t for t in row if t != 2]
            if toks:
                worst = max(worst, int(np.bincount(toks).max()))
        return worst

    sb = dict(state)
    sb.update(P.to_torch(P.decoder_state_diverse("greedy", vocab_size=V)))
    model.load_state_dict(sb, strict=True)
    ref_gb = model(dict(inp, sample_method="greedy", max_length=20))
    o_gb = O.greedy_decode(sb, o_enc["attn_emb"], o_enc["attn_emb_len"], 20)
    assert torch.equal(o_gb["seq"], ref_gb["seq"])
    stb = o_gb["steps"]
    cmp("G4b greedy logit", o_gb["logit"][:, :stb], ref_gb["logit"][:, :stb])
    ends = [int((row == 2).nonzero()[0]) if (row == 2).any() else 20 for row in ref_gb["seq"]]
    top2b = ref_gb["logit"][:, :stb].topk(2, dim=-1).values
    gapb = top2b[..., 0] - top2b[..., 1]
    print(f"  G4b greedy: steps {stb}, first <end> per clip {ends}, most repeated token x{max_repeat(ref_gb['seq'])}, "
          f"min top1-top2 gap {float(gapb.min()):.2e}\n", ref_gb["seq"].numpy())
    assert len(set(ends)) >= 3 and max_repeat(ref_gb["seq"]) <= 4 and float(gapb.min()) > 5e-4 and stb >= 15
    np.savez_compressed(
        os.path.join(out_dir, "g4b_greedy.npz"), steps=np.array(stb), seq=ref_gb["seq"].numpy(), top2_gap=gapb.numpy(),
        sampled_logprob=ref_gb["sampled_logprob"][:, :stb].numpy(),
        logit_top_val=ref_gb["logit"][:, :stb].topk(8, dim=-1).values.numpy(),
        logit_top_idx=ref_gb["logit"][:, :stb].topk(8, dim=-1).indices.numpy())

    sb = dict(state)
    sb.update(P.to_torch(P.decoder_state_diverse("beam", vocab_size=V)))
    model.load_state_dict(sb, strict=True)
    g5b = {}
    for k in (3, 4):
        ref_b = model(dict(inp, sample_method="beam", beam_size=k, max_length=20))
        trace = []
        o_b = O.beam_search(sb, o_enc["attn_emb"], o_enc["attn_emb_len"], k, 20, trace=trace)
        assert torch.equal(o_b["seq"], ref_b["seq"]), (o_b["seq"], ref_b["seq"])
        ident = list(range(k))
        per_clip = [[r for r in trace if r["clip"] == i] for i in range(4)]
        reorder = [sum(1 for r in rs if r["t"] > 0 and r["prev_beam"] != ident) for rs in per_clip]
        first_end = [min((r["t"] for r in rs if any(r["ended"])), default=99) for rs in per_clip]
        last_t = [max(r["t"] for r in rs) for rs in per_clip]
        early_and_on = sum(1 for i in range(4) if first_end[i] < 5 and last_t[i] > first_end[i])
        margin = min(r["margin"] for r in trace)
        distinct = len(set(map(tuple, ref_b["seq"].tolist())))
        print(f"  G5b beam {k}: distinct results {distinct}/4, steps with a changed parent per clip {reorder}, first finished "
              f"beam at t = {first_end} of {last_t}, most repeated token x{max_repeat(ref_b['seq'])}, min candidate margin "
              f"{margin:.2e}\n", ref_b["seq"].numpy())
        # >= 3 distinct captions, the parent beam changes on >= 5 steps of every clip, a beam finishes before t = 5 while
        # the search goes on (the -1000 path, base.py:317), margins far above the f32 noise of the logits (4e-6)
        assert distinct >= 3 and min(reorder) >= 5 and early_and_on >= 1 and margin > 1e-4
        if k == 3:
            assert max_repeat(ref_b["seq"]) <= 3
        g5b[f"seq_beam{k}"] = ref_b["seq"].numpy()
        g5b[f"score_beam{k}"] = o_b["score"].numpy()
        g5b[f"reorder_steps_beam{k}"] = np.array(reorder)
        nb = model(dict(inp, sample_method="beam", beam_size=k, max_length=20, n_best=True, n_best_size=k))
        o_nb = O.beam_search(sb, o_enc["attn_emb"], o_enc["attn_emb_len"], k, 20, n_best=True, n_best_size=k)
        assert torch.equal(o_nb["seq"], nb["seq"])
        g5b[f"nbest_beam{k}"] = nb["seq"].numpy()
    np.savez_compressed(os.path.join(out_dir, "g5b_beam.npz"), **g5b)
    model.load_state_dict(state, strict=True)

    # ---- G7: LabelSmoothingLoss known answers (loss.py:51-74) ---------------------------------
    sys.modules["wandb"].run = None
    from captioning.losses.loss import LabelSmoothingLoss
    logit = ref_d["logit"][:, :11]
    tgt = word[:, 1:]
    tgt_len = torch.tensor([11, 9, 8, 11])
    loss = LabelSmoothingLoss(smoothing=0.1)({"logit": logit, "tgt": tgt, "tgt_len": tgt_len})
    np.savez_compressed(os.path.join(out_dir, "g7_loss.npz"), tgt=tgt.numpy(), tgt_len=tgt_len.numpy(),
                        loss=np.array(float(loss)))
    print(f"  G7 label-smoothing loss = {float(loss):.6f}")


    # ---- G9: TransformerEncoder (transformer_encoder.py:64-116), the alternative temporal encoder -----------
    from captioning.models.transformer_encoder import TransformerEncoder
    trm_np = P.trm_encoder_state()
    trm_state = P.to_torch(trm_np)
    trm = TransformerEncoder(spec_dim=-1, fc_feat_dim=2048, attn_feat_dim=2048, d_model=256)
    assert set(trm.state_dict()) == set(trm_state)
    trm.load_state_dict(trm_state, strict=True)
    trm.eval()
    g9 = {}
    for tag, lens in (("full", [31, 31]), ("ragged", [31, 20])):
        lens_t = torch.tensor(lens)
        ref_t = trm({"attn": attn, "attn_len": lens_t})
        assert lens_t.tolist() == [v + 1 for v in lens]      # the reference increments the caller's tensor in place
        o_t = O.transformer_encoder_forward(trm_state, attn, lens)
        cmp(f"G9 {tag} attn_emb", o_t["attn_emb"], ref_t["attn_emb"])
        assert torch.equal(o_t["attn_emb_len"], ref_t["attn_emb_len"])
        g9[f"{tag}_lens"] = np.array(lens)
        g9[f"{tag}_attn_emb"] = ref_t["attn_emb"].numpy()
        g9[f"{tag}_attn_emb_len"] = ref_t["attn_emb_len"].numpy()
    np.savez_compressed(os.path.join(out_dir, "g9_trm_encoder.npz"), **g9)

    # ---- G8: one TRAINING step (A13-A16), dropout p = 0 so that it is deterministic --------------------
    # scheduled-sampling forward (base.py:131-199, transformer_model.py:34-57), LabelSmoothingLoss, backward,
    # clip_grad_norm_(1.0) and one torch.optim.Adam(lr 5e-4, weight_decay 1e-6) update, all by the reference.
    import copy
    import random
    from oracle import train_path as OT
    torch.set_grad_enabled(True)
    cfg0 = copy.deepcopy(cfg["model"])
    cfg0["decoder"]["args"]["dropout"] = 0.0
    cfg0["encoder"]["rnn"]["args"]["dropout"] = 0.0
    model0 = train_util.init_model_from_config(cfg0, print_fn=lambda s: None)
    model0.load_state_dict(state, strict=True)
    model0.train()
    model0.encoder.cnn.eval()   # no F.dropout inside the frozen Cnn14 (cnn_encoder.py:432-442) for this fixture
    gen = torch.Generator().manual_seed(11)
    cap = torch.randint(4, V, (4, 13), generator=gen)
    cap_len = np.array([13, 10, 8, 12])
    cap[:, 0] = 1
    for i, n in enumerate(cap_len):
        cap[i, n - 1] = 2
        cap[i, n:] = 0
    _PRESET["lms"] = lms4
    loss_fn = LabelSmoothingLoss(smoothing=0.1)
    g8 = {"cap": cap.numpy(), "cap_len": cap_len, "wav_len": np.array(wav_len)}
    trainable = [(k, p_) for k, p_ in model0.named_parameters() if p_.requires_grad]
    assert sorted(k for k, _ in trainable) == sorted(OT.trainable_keys(state)), "trainable key sets differ"
    idx_gen = np.random.default_rng(3)
    sample_idx = {k: idx_gen.integers(0, p_.numel(), size=min(64, p_.numel())) for k, p_ in trainable}
    for tag, ss_ratio in (("ss", 0.7), ("tf", 1)):
        model0.load_state_dict(state, strict=True)
        model0.zero_grad(set_to_none=True)
        random.seed(5)
        use_cap = [random.random() < ss_ratio for _ in range(cap.shape[1] - 1)] if ss_ratio != 1 else []
        random.seed(5)
        out = model0({"mode": "train", "wav": torch.zeros(4, 320000), "wav_len": wav_len, "specaug": False,
                      "cap": cap, "cap_len": cap_len, "ss_ratio": ss_ratio})
        out["tgt"] = cap[:, 1:]
        out["tgt_len"] = torch.as_tensor(cap_len - 1)
        loss = loss_fn(out)
        loss.backward()
        total_norm = torch.nn.utils.clip_grad_norm_(model0.parameters(), 1.0)
        raw = {k: p_.grad.detach().clone() / min(1.0, float(1.0 / (total_norm + 1e-6))) for k, p_ in trainable}
        opt = torch.optim.Adam([p_ for _, p_ in trainable], lr=5e-4, weight_decay=1e-6)
        before = {k: p_.detach().clone() for k, p_ in trainable}
        opt.step()
        # the same step in the oracle
        o = OT.train_step_grads(state, O.cnn14_from_logmel(state, lms4), O.cnn14_feat_len(wav_len), cap, cap_len,
                                use_cap, p_dec=0.0, p_rnn=0.0, teacher_forcing=(ss_ratio == 1))
        cmp(f"G8 {tag} logit", o["logit"], out["logit"].detach())
        report[f"G8 {tag} loss"] = abs(float(o["loss"]) - float(loss))
        worst_g = 0.0
        for k, _ in trainable:
            scale = float(raw[k].abs().max()) + 1e-12
            worst_g = max(worst_g, float((o["grads"][k] - raw[k]).abs().max()) / scale)
        report[f"G8 {tag} grads (rel. to max)"] = worst_g
        print(f"  oracle vs reference  G8 {tag} loss {float(loss):.6f}  worst relative grad diff {worst_g:.3e}")
        if ss_ratio != 1:
            assert torch.equal(o["seq"], out["seq"]), "greedy tokens of the training forward differ"
        params = {k: state[k].clone() for k, _ in trainable}
        m1 = {k: torch.zeros_like(v) for k, v in params.items()}
        m2 = {k: torch.zeros_like(v) for k, v in params.items()}
        o_norm = OT.clip_and_adam(params, o["grads"], m1, m2, 1)
        report[f"G8 {tag} grad norm (relative)"] = abs(float(o_norm) - float(total_norm)) / float(total_norm)
        worst_p = max(float((params[k] - p_.detach()).abs().max()) for k, p_ in trainable)
        report[f"G8 {tag} params after Adam"] = worst_p
        print(f"  G8 {tag}: grad norm {float(total_norm):.6f}; worst |param diff| after Adam {worst_p:.3e}")
        g8[f"{tag}_use_cap"] = np.array(use_cap, dtype=np.int32)
        g8[f"{tag}_loss"] = np.array(float(loss))
        g8[f"{tag}_total_norm"] = np.array(float(total_norm))
        if "seq" in out:
            g8[f"{tag}_seq"] = out["seq"].numpy()
        g8[f"{tag}_logit_top_val"] = out["logit"].detach().topk(8, dim=-1).values.numpy()
        g8[f"{tag}_logit_top_idx"] = out["logit"].detach().topk(8, dim=-1).indices.numpy()
        for k, p_ in trainable:
            g8[f"{tag}_gnorm/{k}"] = np.array(float(raw[k].double().norm()))  # f32 CPU norms lose 2e-4 on 1.3M elements
            g8[f"{tag}_gsum/{k}"] = np.array(float(raw[k].double().sum()))
            g8[f"{tag}_gsample/{k}"] = raw[k].reshape(-1)[sample_idx[k]].numpy()
            g8[f"{tag}_delta/{k}"] = (p_.detach() - before[k]).reshape(-1)[sample_idx[k]].numpy()
    for k in sample_idx:
        g8[f"sample_idx/{k}"] = sample_idx[k]
    np.savez_compressed(os.path.join(out_dir, "g8_train.npz"), **g8)
    torch.set_grad_enabled(False)

    with open(os.path.join(out_dir, "REPORT.txt"), "w") as f:
        f.write("max |oracle - reference| per fixture (written by make_golden.py, torch %s)\n" % torch.__version__)
        for k, v in report.items():
            f.write(f"{k:32s} {v:.3e}\n")
        f.write(f"greedy steps {steps}, min top1-top2 gap {float(gap.min()):.3e}\n")
    worst = max(report.values())
    print(f"worst oracle-vs-reference diff: {worst:.3e}")
    assert worst < 2e-4, "oracle does not match the reference"


if __name__ == "__main__":
    main()
