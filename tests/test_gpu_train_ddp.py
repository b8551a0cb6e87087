"""Two-rank data-parallel training step ON the GPU box: both ranks share cuda:0 and all-reduce the flat gradient buffer
through gloo (RCCL refuses two ranks on one device; the collective call, the world-size folding into the clip
coefficient and the fused Adam are the same code the 8-GPU launch runs).  With dropout off, teacher forcing and equal
caption lengths, the 2 x 2-clip data-parallel step must equal the single-process 4-clip step."""
import os
import random

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _model():
    import audiocaption_amd as A
    from audiocaption_amd import procedural as Pr
    model = A.init_model_from_config(A.cnn14rnn_trm_config(4981), print_fn=lambda s: None)
    model.load_state_dict(Pr.to_torch(Pr.cnn14rnn_trm_state(4981)), strict=True)
    model = model.cuda().train()
    for m in model.decoder.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    model.encoder.rnn.network.dropout = 0.0
    model.encoder.cnn.eval()
    return model


def _batch(sl):
    from audiocaption_amd import procedural as Pr
    L = 96000
    wav = torch.from_numpy(Pr.synthetic_wav(4, L, seed=41, varied=True))[sl].cuda()
    g = torch.Generator().manual_seed(6)
    cap = torch.randint(4, 4981, (4, 9), generator=g)
    cap[:, 0], cap[:, -1] = 1, 2
    n = wav.shape[0]
    return {"mode": "train", "wav": wav, "wav_len": [L] * n, "specaug": False, "cap": cap[sl].cuda(),
            "cap_len": np.array([9] * n), "ss_ratio": 1}


def _one_step(sl):
    from audiocaption_amd.optim import FusedAdam
    from audiocaption_amd.train import TrainEngine
    model = _model()
    eng = TrainEngine(model)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-6)
    r = eng.step(_batch(sl), opt, smoothing=0.1, max_grad_norm=1.0, use_graph=False)
    torch.cuda.synchronize()
    return float(r["loss"]), float(r["total_norm"]), eng.flat.flat.detach().cpu().clone()


def _worker(rank, world, port, out_dir, algo="all_reduce"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["AUDIOCAPTION_GRAD_SYNC"] = algo
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loss, norm, flat = _one_step(slice(2 * rank, 2 * rank + 2))
        tag = "" if algo == "all_reduce" else "_" + algo
        torch.save({"loss": loss, "norm": norm, "flat": flat}, os.path.join(out_dir, f"rank{rank}{tag}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step(tmp_path):
    from audiocaption_amd import build
    build.build()
    port = 29700 + random.randint(0, 200)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    dr = (r0["flat"] - r1["flat"]).abs()
    print(f"rank0 vs rank1 parameters: max|diff| {float(dr.max()):.3e}, differing entries {int((dr > 0).sum())}")
    assert torch.equal(r0["flat"], r1["flat"]), "ranks diverged after the all-reduced update"
    assert r0["norm"] == pytest.approx(r1["norm"], rel=1e-6)
    loss, norm, flat = _one_step(slice(0, 4))
    assert 0.5 * (r0["loss"] + r1["loss"]) == pytest.approx(loss, rel=1e-5)
    assert r0["norm"] == pytest.approx(norm, rel=1e-4)
    d = float((r0["flat"] - flat).abs().max())
    print(f"2-rank vs single-process parameters after one step: max|diff| {d:.3e}")
    # Adam's first step is lr * sign-like: near-zero gradients may flip, everything else must agree
    frac = float(((r0["flat"] - flat).abs() > 1e-5).float().mean())
    assert frac < 1e-3


def test_two_rank_reduce_scatter_all_gather_equals_all_reduce(tmp_path):
    """AUDIOCAPTION_GRAD_SYNC=rs_ag (reduce_scatter + all_gather over the two flat-buffer slices, issued between the
    backward parts like the all-reduce) leaves bit-identical parameters on both ranks, equal to the all-reduce's."""
    from audiocaption_amd import build
    build.build()
    port = 29400 + random.randint(0, 90)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, port + 100, str(tmp_path), "rs_ag"), nprocs=2, join=True)
    a0, b0, b1 = (torch.load(tmp_path / f) for f in ("rank0.pt", "rank0_rs_ag.pt", "rank1_rs_ag.pt"))
    assert torch.equal(b0["flat"], b1["flat"]), "ranks diverged"
    # the collective itself is bit-equal (tests/test_train_oracle.py); the two runs differ at most by the backward's
    # atomic accumulation order
    d = (a0["flat"] - b0["flat"]).abs()
    print(f"rs_ag vs all_reduce parameters after one step: bit-equal {bool(torch.equal(a0['flat'], b0['flat']))}, max|diff| {float(d.max()):.3e}")
    assert float((d > 1e-5).float().mean()) < 1e-3 and a0["norm"] == pytest.approx(b0["norm"], rel=1e-5)


def _graph_worker(rank, world, port, out_dir):
    """Three iterations with use_graph=True: the first eager, the second captures BOTH graphs of the two-part step (decoder
    part, GRU part - the decoder's gradients are all-reduced between them), the third replays them."""
    from audiocaption_amd.optim import FusedAdam
    from audiocaption_amd.train import TrainEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        eng = TrainEngine(model)
        opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-6)
        losses = []
        for _ in range(3):
            losses.append(float(eng.step(_batch(slice(2 * rank, 2 * rank + 2)), opt, smoothing=0.1, max_grad_norm=1.0)["loss"]))
        st = next(iter(eng._states.values()))
        assert {"fwd0", "tail_head", "gru"} <= set(st["graphs"])
        torch.cuda.synchronize()
        torch.save({"losses": losses, "flat": eng.flat.flat.detach().cpu().clone()}, os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_graph_steps_keep_the_ranks_identical(tmp_path):
    from audiocaption_amd import build
    from audiocaption_amd.optim import FusedAdam
    from audiocaption_amd.train import TrainEngine
    build.build()
    port = 29950 + random.randint(0, 40)
    mp.spawn(_graph_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    assert torch.equal(r0["flat"], r1["flat"]), "ranks diverged"
    # against three single-process iterations on the union batch (eager, one part): same trajectory up to rounding
    model = _model()
    eng = TrainEngine(model)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-6)
    losses = [float(eng.step(_batch(slice(0, 4)), opt, smoothing=0.1, max_grad_norm=1.0, use_graph=False)["loss"])
              for _ in range(3)]
    for k in range(3):
        assert 0.5 * (r0["losses"][k] + r1["losses"][k]) == pytest.approx(losses[k], rel=2e-4), k
    assert losses[2] < losses[0]
    frac = float(((r0["flat"] - eng.flat.flat.cpu()).abs() > 5e-5).float().mean())
    print(f"entries off by more than 5e-5 after three steps: {frac:.2e}")
    assert frac < 5e-3
