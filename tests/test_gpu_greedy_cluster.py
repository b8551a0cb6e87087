"""GPU tests of the one-launch greedy search (csrc/decoder_cluster.hip: a cluster of four workgroups per row) against the
launch chain (csrc/decoder.hip) and the oracle (base.py:152-218 restated in oracle/cpu_path.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _enc(B, Tm, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    attn = torch.randn(B, Tm, 512, generator=g) * 0.5
    lens = torch.randint(max(1, Tm // 3), Tm + 1, (B,), generator=g) if ragged else torch.full((B,), Tm)
    lens[0] = Tm
    for b in range(B):
        attn[b, int(lens[b]):] = 0.0
    return attn, lens


@pytest.mark.parametrize("B,Tm", [(1, 31), (3, 15), (64, 31), (37, 31), (5, 94)])
def test_cluster_equals_chain(hip_model, B, Tm):
    """The same search by both forms on random encoder outputs (ragged memory lengths; 64 rows = every CU holds a workgroup;
    Tm = 94: 30 s clips): ids, stop bookkeeping and log-probabilities equal,
    logits and embeddings within 2e-5 (the forms sum in different orders), eager + captured + replayed."""
    dec = hip_model.decoder
    attn, lens = _enc(B, Tm, seed=B * 131 + Tm)
    args = (attn.cuda(), lens, 20, hip_model.start_idx, hip_model.end_idx, hip_model.pad_idx)
    want = dec.greedy(*args, mode="chain")
    torch.cuda.synchronize()
    for it in range(3):
        got = dec.greedy(*args, mode="cluster")
        torch.cuda.synchronize()
        assert int(got["cluster_error"].item()) == 0
        cnt_w, cnt_g = want["unfinished_cnt"].cpu().numpy(), got["unfinished_cnt"].cpu().numpy()
        np.testing.assert_array_equal(cnt_g, cnt_w)
        np.testing.assert_array_equal(got["seq"].cpu().numpy(), want["seq"].cpu().numpy())
        steps = int(np.argmax(cnt_w == 0)) + 1 if (cnt_w == 0).any() else 20
        d = float((got["logit"][:, :steps] - want["logit"][:, :steps]).abs().max())
        e = float((got["embed"][:, :steps] - want["embed"][:, :steps]).abs().max())
        lp = float((got["sampled_logprob"] - want["sampled_logprob"]).abs().max())
        print(f"B={B} Tm={Tm} pass {it}: steps {steps}, max|dlogit| {d:.2e} |dembed| {e:.2e} |dlogprob| {lp:.2e}")
        assert d < 2e-5 and e < 2e-5 and lp < 2e-5


def test_cluster_stops_like_the_reference(diverse_models, golden_dir):
    """The draw whose clips stop at steps 3 / 10 / 19 / 10: columns beyond the batch's last step keep the reference's initial
    values (seq = <end>, log-probability 0), unfinished_cnt is the chain's, and a batch whose rows all stop early leaves the
    later columns untouched although the clusters run one or two steps past the reference's stop."""
    import os
    model = diverse_models["greedy"]
    g4 = dict(np.load(os.path.join(golden_dir, "g4_greedy.npz")))
    attn, alen = torch.from_numpy(g4["attn_emb"]), torch.from_numpy(g4["attn_emb_len"])
    for rows in ([0, 1, 2, 3], [0, 1, 3], [0]):
        a, l = attn[rows].cuda(), alen[rows]
        args = (a, l, 20, model.start_idx, model.end_idx, model.pad_idx)
        want = model.decoder.greedy(*args, mode="chain")
        got = model.decoder.greedy(*args, mode="cluster")
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got["unfinished_cnt"].cpu().numpy(), want["unfinished_cnt"].cpu().numpy())
        np.testing.assert_array_equal(got["seq"].cpu().numpy(), want["seq"].cpu().numpy())
        assert float((got["sampled_logprob"] - want["sampled_logprob"]).abs().max()) < 2e-5
        cnt = want["unfinished_cnt"].cpu().numpy()
        if (cnt == 0).any():
            stop = int(np.argmax(cnt == 0)) + 1
            assert float(got["sampled_logprob"][:, stop:].abs().max()) == 0.0 if stop < 20 else True


def test_blocking_call_takes_the_cluster_form_and_agrees_with_the_oracle(hip_model, state4981, monkeypatch):
    """model(input_dict) (run.py:45 / demo.py:48) decodes with the one-launch form by default: wav -> ids against the oracle."""
    from audiocaption_amd import procedural as P
    from oracle import cpu_path as O
    monkeypatch.delenv("AUDIOCAPTION_GREEDY", raising=False)
    wav_len = [320000, 280000, 160000, 300000]
    wav = P.synthetic_wav(4, 320000, varied=True)
    for i, n in enumerate(wav_len):
        wav[i, n:] = 0.0
    wav = torch.from_numpy(wav)
    want = O.caption_forward(state4981, wav, wav_len, "greedy")
    seen = []
    orig = hip_model.decoder._greedy_launch

    def spy(st, *a):
        seen.append(st.get("cluster_ws") is not None)
        return orig(st, *a)

    monkeypatch.setattr(hip_model.decoder, "_greedy_launch", spy)
    out = hip_model({"mode": "inference", "wav": wav.cuda(), "wav_len": wav_len, "specaug": False,
                     "sample_method": "greedy", "max_length": 20})
    assert seen and all(seen), "the blocking call should have used the cluster kernel"
    st = want["steps"]
    assert float((out["logit"][:, :st].cpu() - want["logit"][:, :st]).abs().max()) < 1e-4
    assert torch.equal(out["seq"][:, :st], want["seq"][:, :st])


def test_cluster_refuses_what_it_does_not_cover(hip_model):
    """More rows than CUs / 4 would run the clusters in rounds (slower than the launch chain): mode="cluster" says so, "auto"
    takes the chain."""
    from audiocaption_amd import _lib
    dec = hip_model.decoder
    attn, lens = _enc(100, 31, seed=5)
    args = (attn.cuda(), lens, 20, hip_model.start_idx, hip_model.end_idx, hip_model.pad_idx)
    with pytest.raises(_lib.HipLibraryError):
        dec.greedy(*args, mode="cluster")
    out = dec.greedy(*args, alone=True)
    assert "cluster_error" not in out and out["seq"].shape == (100, 20)
