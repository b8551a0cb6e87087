"""GPU parity tests of the DEFAULT kernel route at sizes that fill the chip, and of the decode side at the Clotho vocabulary.

The tier tests of tests/test_gpu_model.py force the F(4,3) kernel onto few-clip batches (``W43_MIN_WORKGROUPS = 1``) because
the default routes such launches to the K-sliced F(2,3) form; here nothing is overridden: the batch is what decides, the
launch hook records which kernel family every conv layer took, and the result is held to the oracle directly
(oracle/cpu_path.py: identical greedy ids, logits within 1e-4 - SURVEY 8(d)'s fp32 gate; reference code
cnn_encoder.py:414-464, base.py:152-218,254-325)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _maxdiff(name, got, want):
    got, want = torch.as_tensor(got).float().cpu(), torch.as_tensor(want).float().cpu()
    d = float((got - want).abs().max())
    print(f"[{name}] max|diff| {d:.3e} (|want| max {float(want.abs().max()):.3e})")
    return d


def _record_conv_launches():
    """Context manager: the (algo, W, Cin, Cout, mode) of every conv launch, in order."""
    import contextlib
    from audiocaption_amd import kernels as K

    @contextlib.contextmanager
    def cm():
        seen = []

        def hook(phase, info):
            if phase == "pre":
                seen.append((info["algo"], info["W"], info["Cin"], info["Cout"], info["mode"]))

        saved = K.CONV_LAUNCH_HOOK
        K.CONV_LAUNCH_HOOK = hook
        try:
            yield seen
        finally:
            K.CONV_LAUNCH_HOOK = saved
    return cm()


def _oracle_in_chunks(state, wav, wav_len, chunk=16):
    """oracle/cpu_path.caption_forward over clip chunks (clips do not interact; bounds the CPU side's memory)."""
    from oracle import cpu_path as O
    outs = [O.caption_forward(state, wav[i:i + chunk], wav_len[i:i + chunk], "greedy") for i in range(0, len(wav_len), chunk)]
    steps = max(o["steps"] for o in outs)
    T = max(o["attn_emb"].shape[1] for o in outs)

    def cat(key, pad_steps=False):
        parts = []
        for o in outs:
            t = o[key]
            if pad_steps and t.shape[1] < steps:
                shape = list(t.shape)
                shape[1] = steps - t.shape[1]
                t = torch.cat([t, torch.zeros(shape, dtype=t.dtype)], 1)
            parts.append(t)
        return torch.cat(parts, 0)

    assert all(o["attn_emb"].shape[1] == T for o in outs)
    return {"attn_emb": cat("attn_emb"), "attn_emb_len": cat("attn_emb_len"), "per_chunk": outs, "chunk": chunk}


def _check_against_chunks(out, ref, wav_len):
    """Ids and logits of the batched HIP run against the per-chunk oracle runs, each over the steps ITS chunk decoded (the
    reference stops a batch when every clip of it has emitted <end>: base.py:206-211).  Clip by clip, step by step: the ids
    must agree wherever the oracle's own top-1 / top-2 margin exceeds 2e-3 (a difference at a near-tie ends the comparison of
    that clip: what follows was decoded from another prefix); logits are compared on the common prefix."""
    worst, compared, ties = 0.0, 0, 0
    seq, logit = out["seq"].cpu(), out["logit"].cpu()
    for k, o in enumerate(ref["per_chunk"]):
        lo = k * ref["chunk"]
        st = o["steps"]
        top2 = o["logit"][:, :st].topk(2, -1).values
        gap = top2[..., 0] - top2[..., 1]
        for r in range(o["seq"].shape[0]):
            for t in range(st):
                worst = max(worst, float((logit[lo + r, t] - o["logit"][r, t]).abs().max()))
                if int(seq[lo + r, t]) != int(o["seq"][r, t]):
                    assert float(gap[r, t]) <= 2e-3, f"clip {lo + r} step {t}: ids differ at a margin of {float(gap[r, t]):.2e}"
                    ties += 1
                    break
                compared += 1
    print(f"worst |logit diff| {worst:.3e} over {compared} (clip, step) pairs with equal ids; {ties} clips left at a near-tie")
    assert compared >= 8 * len(wav_len) and ties <= len(wav_len) // 8
    return worst


@pytest.mark.parametrize("B,seconds,all_f43", [(16, 4.0, False), (48, 10.0, True)])
def test_default_route_vs_oracle(hip_model, state4981, B, seconds, all_f43):
    """No override of any routing threshold.  16 x 4 s: conv block 1 fused + blocks 2-3 on the F(4,3) kernel, blocks 4-6 (too
    few workgroups for one per CU) on F(2,3).  48 x 10 s: every layer on the F(4,3) family, the batch the bench runs at
    (64) in everything but the count.  wav -> ids against the oracle: ids identical, logits within 1e-4."""
    from audiocaption_amd import cnn_encoder as CE
    from audiocaption_amd import procedural as P
    assert hip_model.encoder.cnn.conv_algo == "wino43" and CE.W43_MIN_WORKGROUPS == 192, "this test is about the defaults"
    L = int(seconds * 32000)
    wav_len = [L - 3200 * (i % 5) * (1 if i % 3 else 0) for i in range(B)]   # a few lengths, the longest first
    wav = P.synthetic_wav(B, L, varied=True)
    for i, n in enumerate(wav_len):
        wav[i, n:] = 0.0
    wav = torch.from_numpy(wav)
    with _record_conv_launches() as seen:
        out = hip_model({"mode": "inference", "wav": wav.cuda(), "wav_len": wav_len, "specaug": False,
                         "sample_method": "greedy", "max_length": 20})
    torch.cuda.synchronize()
    algos = [a for a, *_ in seen]
    print("conv launches:", seen)
    assert algos[0] == "block1_w4" and len(seen) == 11, "block 1 fused, then conv1 / conv2 of blocks 2-6"
    if all_f43:
        assert set(algos[1:]) == {"wino43"}
    else:
        assert algos[1:5] == ["wino43"] * 4 and set(algos[5:]) == {"wino1d"}, algos
    ref = _oracle_in_chunks(state4981, wav, wav_len)
    assert torch.equal(out["attn_emb_len"], ref["attn_emb_len"])
    assert _maxdiff("attn_emb", out["attn_emb"], ref["attn_emb"]) < 5e-4
    assert _check_against_chunks(out, ref, wav_len) < 1e-4


def test_bench_batch_launches_the_f43_family_on_every_layer(hip_model):
    """At the bench's batch (64 x 10 s) the default tier means: block 1 in its fused kernel, conv1 and conv2 of blocks 2-6
    on the F(4,3) kernel with the epilogue each layer needs (conv1: full output; conv2: + 2x2 pool; block 6: + mean over
    mel) - asserted from the launch hook, not from the configuration."""
    from audiocaption_amd import procedural as P
    wav = torch.from_numpy(P.synthetic_wav(64, 320000, varied=True)).cuda()
    with _record_conv_launches() as seen:
        hip_model.encoder({"wav": wav, "wav_len": [320000] * 64, "specaug": False})
    torch.cuda.synchronize()
    want = [("block1_w4", 64, 1, 64, 1)]
    cin = 64
    for blk, (w, cout) in enumerate(((32, 128), (16, 256), (8, 512), (4, 1024), (2, 2048)), start=2):
        want.append(("wino43", w, cin, cout, 0))
        want.append(("wino43", w, cout, cout, 1 if blk < 6 else 2))
        cin = cout
    assert seen == want, seen


@pytest.fixture(scope="module")
def clotho_models():
    """Product models at the Clotho vocabulary (4368, BASELINE configs[0] / configs[1]): the plain draw and the two
    high-entropy decoder draws of the g4b / g5b fixtures."""
    import audiocaption_amd as A
    from audiocaption_amd import procedural as P
    V = 4368
    base = P.to_torch(P.cnn14rnn_trm_state(V))
    models = {}
    for kind in ("plain", "greedy", "beam"):
        st = dict(base)
        if kind != "plain":
            st.update(P.to_torch(P.decoder_state_diverse(kind, vocab_size=V)))
        m = A.init_model_from_config(A.cnn14rnn_trm_config(V), print_fn=lambda s: None)
        m.load_state_dict(st, strict=True)
        models[kind] = (m.eval().to("cuda:0"), st)
    return models


@pytest.mark.parametrize("mode", ["chain", "cluster"])
@pytest.mark.parametrize("kind", ["plain", "greedy"])
def test_greedy_at_the_clotho_vocabulary_vs_oracle(clotho_models, golden_dir, kind, mode, monkeypatch):
    """The headline's vocabulary (4368 rows of classifier and embedding: other GEMM tails than 4981) on the g4 encoder
    outputs: greedy ids identical to the oracle's, logits / log-probabilities within 1e-4 (base.py:152-218)."""
    import os
    from oracle import cpu_path as O
    monkeypatch.setenv("AUDIOCAPTION_GREEDY", mode)
    model, st = clotho_models[kind]
    g4 = dict(np.load(os.path.join(golden_dir, "g4_greedy.npz")))
    attn, alen = torch.from_numpy(g4["attn_emb"]), torch.from_numpy(g4["attn_emb_len"])
    want = O.greedy_decode(st, attn, alen, 20)
    enc = {"attn_emb": attn.cuda(), "attn_emb_len": alen, "fc_emb": torch.from_numpy(g4["fc_emb"]).cuda()}
    for _ in range(3):   # eager, capture, replay
        out = model.forward_decoder({"mode": "inference", "sample_method": "greedy", "max_length": 20}, enc)
        s = want["steps"]
        assert torch.equal(out["seq"][:, :s].cpu(), want["seq"][:, :s])
        assert out["logit"].shape[-1] == 4368
        assert _maxdiff(f"V=4368 greedy logits [{kind}]", out["logit"][:, :s], want["logit"][:, :s]) < 1e-4
        assert _maxdiff(f"V=4368 greedy logprob [{kind}]", out["sampled_logprob"][:, :s], want["sampled_logprob"][:, :s]) < 1e-4


@pytest.mark.parametrize("kind", ["plain", "beam"])
@pytest.mark.parametrize("beam", [3, 4])
def test_beam_at_the_clotho_vocabulary_vs_oracle(clotho_models, golden_dir, kind, beam):
    """Beam search at V = 4368 against the oracle (base.py:254-361): best captions and n-best lists identical."""
    import os
    from oracle import cpu_path as O
    model, st = clotho_models[kind]
    g4 = dict(np.load(os.path.join(golden_dir, "g4_greedy.npz")))
    attn, alen = torch.from_numpy(g4["attn_emb"]), torch.from_numpy(g4["attn_emb_len"])
    enc = {"attn_emb": attn.cuda(), "attn_emb_len": alen, "fc_emb": torch.from_numpy(g4["fc_emb"]).cuda()}
    req = {"mode": "inference", "sample_method": "beam", "beam_size": beam, "max_length": 20}
    want = O.beam_search(st, attn, alen, beam_size=beam, max_length=20)["seq"]
    for _ in range(3):
        out = model.forward_decoder(dict(req), enc)
        np.testing.assert_array_equal(out["seq"].numpy(), want.numpy())
    wantn = O.beam_search(st, attn, alen, beam_size=beam, max_length=20, n_best=True, n_best_size=beam)["seq"]
    outn = model.forward_decoder(dict(req, n_best=True, n_best_size=beam), enc)
    np.testing.assert_array_equal(outn["seq"].numpy(), wantn.numpy())


@pytest.mark.parametrize("route", ["default", "wide", "hybrid"])
@pytest.mark.parametrize("kind", ["plain", "beam"])
def test_beam_over_512_rows_takes_the_tiled_classifier_and_matches_the_oracle(clotho_models, golden_dir, kind, route, monkeypatch):
    """From 512 decode rows on (beam search over grouped batches) the classifier runs as LayerNorm + the tiled exact-f32 GEMM
    instead of the 16 x 16-tile projection (csrc/decoder.hip classifier_step): 176 clips x beam 3 = 528 rows - the g4 clips
    repeated - must give every copy the oracle's caption of its clip (base.py:254-361), like the 12-row search does.  The two
    opt-in routes of csrc/decoder_wide.hip (every projection / only the joined QKV projection and the classifier on the
    three-plane bf16 kernels; three beams share a clip's audio memory: row_div = 3) are held to the same captions."""
    import os
    from oracle import cpu_path as O
    if route != "default":
        monkeypatch.setenv("AUDIOCAPTION_DECODE_GRAPH", "0")     # eager launches: the route switch is read by every call
        monkeypatch.setenv("AUDIOCAPTION_DEC_WIDE_MIN" if route == "wide" else "AUDIOCAPTION_DEC_HYBRID", "128" if route == "wide" else "1")
    model, st = clotho_models[kind]
    g4 = dict(np.load(os.path.join(golden_dir, "g4_greedy.npz")))
    attn, alen = torch.from_numpy(g4["attn_emb"]), torch.from_numpy(g4["attn_emb_len"])
    want = O.beam_search(st, attn, alen, beam_size=3, max_length=20)["seq"]
    reps = (176 + attn.shape[0] - 1) // attn.shape[0]
    enc = {"attn_emb": attn.repeat(reps, 1, 1).cuda(), "attn_emb_len": alen.repeat(reps),
           "fc_emb": torch.from_numpy(g4["fc_emb"]).repeat(reps, 1).cuda()}
    assert enc["attn_emb"].shape[0] * 3 >= 512
    out = model.forward_decoder({"mode": "inference", "sample_method": "beam", "beam_size": 3, "max_length": 20}, enc)
    np.testing.assert_array_equal(out["seq"].numpy(), want.repeat(reps, 1).numpy())
