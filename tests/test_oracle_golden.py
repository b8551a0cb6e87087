"""CPU: the oracle (oracle/cpu_path.py) against the fixtures written by tests/golden/make_golden.py
from the REFERENCE's own outputs.  This is what pins the oracle on the GPU box, where the reference
does not exist."""
import os

import numpy as np
import torch

from audiocaption_amd import procedural as P
from oracle import cpu_path as O

WAV_LEN = [320000, 280000, 160000, 300000]


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_g1_cnn14_blocks_and_attn_emb(golden_dir, state4981):
    g = _load(golden_dir, "g1_cnn14.npz")
    lms = torch.from_numpy(P.synthetic_logmel(2, 1001))
    attn, blocks = O.cnn14_from_logmel(state4981, lms, return_blocks=True)
    np.testing.assert_allclose(attn.numpy(), g["attn_emb"], rtol=0, atol=2e-5)
    for b in range(6):
        blk = blocks[b]
        np.testing.assert_allclose(blk.double().sum(dim=(2, 3)).numpy(), g[f"block{b + 1}_sum"], rtol=1e-6, atol=1e-3)
        np.testing.assert_allclose(blk[:, :8, :4, :2].numpy(), g[f"block{b + 1}_corner"], rtol=0, atol=2e-5)


def test_g2_gru(golden_dir, state4981):
    attn = torch.from_numpy(_load(golden_dir, "g1_cnn14.npz")["attn_emb"])
    for tag in ("full", "ragged", "short"):
        g = _load(golden_dir, f"g2_gru_{tag}.npz")
        out = O.gru_forward(state4981, attn, g["lens"].tolist())
        np.testing.assert_allclose(out["attn_emb"].numpy(), g["attn_emb"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(out["fc_emb"].numpy(), g["fc_emb"], rtol=0, atol=5e-6)
        assert out["attn_emb"].shape[1] == max(g["lens"])


def test_g3_decoder(golden_dir, state4981):
    g = _load(golden_dir, "g3_decoder.npz")
    word = torch.from_numpy(g["word"])
    out = O.decoder_forward(state4981, word, torch.from_numpy(g["attn_emb"]), torch.from_numpy(g["attn_emb_len"]),
                            word == 0)
    np.testing.assert_allclose(out["embed"].numpy(), g["embed"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(out["logit"][0].numpy(), g["logit_row0"], rtol=0, atol=5e-5)
    idx = torch.from_numpy(g["logit_top_idx"])
    np.testing.assert_allclose(out["logit"].gather(-1, idx).numpy(), g["logit_top_val"], rtol=0, atol=5e-5)


def test_g4_greedy_tokens_identical(golden_dir, state4981):
    g = _load(golden_dir, "g4_greedy.npz")
    out = O.greedy_decode(state4981, torch.from_numpy(g["attn_emb"]), torch.from_numpy(g["attn_emb_len"]), 20)
    assert g["top2_gap"].min() > 1e-4, "fixture must have a top-1/top-2 gap far above fp32 noise"
    np.testing.assert_array_equal(out["seq"].numpy(), g["seq"])
    steps = int(g["steps"])
    assert out["steps"] == steps
    np.testing.assert_allclose(out["sampled_logprob"][:, :steps].numpy(), g["sampled_logprob"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["embed"][:, :steps].numpy(), g["embed"], rtol=0, atol=5e-5)


def test_g4_feat_len_formula(golden_dir):
    g = _load(golden_dir, "g4_greedy.npz")
    np.testing.assert_array_equal(O.cnn14_feat_len(WAV_LEN).numpy(), g["attn_emb_len"])
    np.testing.assert_array_equal(g["attn_emb_len"], [31, 27, 15, 29])


def test_g5_beam_tokens_identical(golden_dir, state4981):
    g4 = _load(golden_dir, "g4_greedy.npz")
    g = _load(golden_dir, "g5_beam.npz")
    for k in (3, 4):
        out = O.beam_search(state4981, torch.from_numpy(g4["attn_emb"]), torch.from_numpy(g4["attn_emb_len"]), k, 20)
        np.testing.assert_array_equal(out["seq"].numpy(), g[f"seq_beam{k}"])


def _diverse_state(state4981, kind):
    from audiocaption_amd import procedural as P
    st = dict(state4981)
    st.update(P.to_torch(P.decoder_state_diverse(kind, vocab_size=4981)))
    return st


def test_g4b_g5b_high_entropy_decode_fixtures(golden_dir, state4981):
    """The second decoder draw (procedural.DIVERSE): the reference's greedy / beam-3 / beam-4 / n-best ids on searches
    whose parent beam changes at >= 5 steps per clip and whose beams finish at different steps (make_golden.py asserts
    those properties on the reference's outputs when it writes the files)."""
    g4 = _load(golden_dir, "g4_greedy.npz")
    attn, lens = torch.from_numpy(g4["attn_emb"]), torch.from_numpy(g4["attn_emb_len"])
    gb = _load(golden_dir, "g4b_greedy.npz")
    out = O.greedy_decode(_diverse_state(state4981, "greedy"), attn, lens, 20)
    np.testing.assert_array_equal(out["seq"].numpy(), gb["seq"])
    assert gb["top2_gap"].min() > 5e-4 and len({int((r == 2).argmax()) for r in gb["seq"]}) >= 3
    steps = int(gb["steps"])
    idx = torch.from_numpy(gb["logit_top_idx"])
    np.testing.assert_allclose(out["logit"][:, :steps].gather(-1, idx).numpy(), gb["logit_top_val"], rtol=0, atol=5e-5)
    g5 = _load(golden_dir, "g5b_beam.npz")
    st = _diverse_state(state4981, "beam")
    for k in (3, 4):
        assert g5[f"reorder_steps_beam{k}"].min() >= 5 and len(set(map(tuple, g5[f"seq_beam{k}"].tolist()))) >= 3
        np.testing.assert_array_equal(O.beam_search(st, attn, lens, k, 20)["seq"].numpy(), g5[f"seq_beam{k}"])
        np.testing.assert_array_equal(O.beam_search(st, attn, lens, k, 20, n_best=True, n_best_size=k)["seq"].numpy(),
                                      g5[f"nbest_beam{k}"])


def test_g7_label_smoothing_known_answer(golden_dir, state4981):
    g3 = _load(golden_dir, "g3_decoder.npz")
    g = _load(golden_dir, "g7_loss.npz")
    word = torch.from_numpy(g3["word"])
    logit = O.decoder_forward(state4981, word, torch.from_numpy(g3["attn_emb"]),
                              torch.from_numpy(g3["attn_emb_len"]), word == 0)["logit"][:, :11]
    tgt, tgt_len = torch.from_numpy(g["tgt"]), torch.from_numpy(g["tgt_len"])
    # closed form of loss.py:58-74 with smoothing 0.1
    V = logit.shape[-1]
    lp = torch.log_softmax(logit, -1)
    q = torch.full_like(lp, 0.1 / (V - 1)).scatter_(-1, tgt.unsqueeze(-1), 0.9)
    mask = torch.arange(11)[None] < tgt_len[:, None]
    loss = (-(q * lp).sum(-1) * mask).sum() / mask.sum()
    assert abs(float(loss) - float(g["loss"])) < 1e-4


def test_g9_transformer_encoder_oracle_vs_reference(golden_dir):
    """TransformerEncoder: oracle against the reference's outputs, full and ragged lengths."""
    import os
    import numpy as np
    import torch
    from audiocaption_amd import procedural as P
    from oracle import cpu_path as O
    g1 = dict(np.load(os.path.join(golden_dir, "g1_cnn14.npz")))
    g9 = dict(np.load(os.path.join(golden_dir, "g9_trm_encoder.npz")))
    st = P.to_torch(P.trm_encoder_state())
    attn = torch.from_numpy(g1["attn_emb"])
    for tag in ("full", "ragged"):
        lens = g9[f"{tag}_lens"].tolist()
        out = O.transformer_encoder_forward(st, attn, lens)
        assert out["attn_emb"].shape == (2, 32, 256)
        assert float((out["attn_emb"] - torch.from_numpy(g9[f"{tag}_attn_emb"])).abs().max()) < 2e-5
        assert out["attn_emb_len"].tolist() == g9[f"{tag}_attn_emb_len"].tolist() == [v + 1 for v in lens]
        assert torch.equal(out["fc_emb"], out["attn_emb"][:, 0])
