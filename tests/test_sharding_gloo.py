"""CPU, world_size 2 (gloo): the multi-GPU path of bench.py is "shard the clips, no data-path collective,
max-over-ranks timing".  The sharding / seeding / reduction logic is exercised here with two processes;
the per-rank compute is replaced by the CPU oracle on tiny clips (the HIP path needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audiocaption_amd import procedural as P
    from audiocaption_amd.sharding import shard_clips, reduce_max_seconds
    # every rank owns a disjoint, seed-distinct shard of the global clip list
    n_global, L = 6, 9600
    mine = shard_clips(n_global, rank, world)
    wav = P.synthetic_wav(len(mine), L, seed=P.BASE_SEED + rank)
    elapsed = 0.5 + rank  # pretend rank 1 is the slow one
    worst = reduce_max_seconds(elapsed)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, float(wav.sum())))
    if rank == 0:
        np.save(os.path.join(out_dir, "res.npy"), np.array([worst], dtype=np.float64))
        all_ids = sorted(i for ids, _ in gathered for i in ids)
        assert all_ids == list(range(n_global)), all_ids
        assert gathered[0][1] != gathered[1][1], "ranks must not process identical clips"
    dist.barrier()
    dist.destroy_process_group()


def test_clip_sharding_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    worst = float(np.load(tmp_path / "res.npy")[0])
    assert worst == pytest.approx(1.5)  # MAX over ranks, as bench.py reports


def test_shard_clips_balanced():
    from audiocaption_amd.sharding import shard_clips, shard_by_duration
    for n, w in ((64, 8), (65, 8), (3, 4)):
        shards = [shard_clips(n, r, w) for r in range(w)]
        assert sorted(i for s in shards for i in s) == list(range(n))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
    # Clotho-shape clips are balanced by total samples, not by count
    durations = [30, 15, 16, 29, 22, 23, 17, 28]
    parts = shard_by_duration(durations, 2)
    loads = [sum(durations[i] for i in p) for p in parts]
    assert sorted(i for p in parts for i in p) == list(range(8))
    assert abs(loads[0] - loads[1]) <= 2
