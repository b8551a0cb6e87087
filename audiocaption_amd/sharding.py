"""Clip-level data parallelism for inference: clips are independent, so a batch or a
dataset is sharded over ranks with NO data-path collective; only timing (bench.py) is reduced."""
import torch
import torch.distributed as dist


def shard_clips(n_clips, rank, world_size):
    """Indices of the clips rank `rank` owns (contiguous, sizes differ by at most one)."""
    base, rem = divmod(n_clips, world_size)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def shard_by_duration(durations, world_size):
    """Greedy longest-first partition balancing total duration per rank (encoder cost ~ duration)."""
    order = sorted(range(len(durations)), key=lambda i: -durations[i])
    parts, loads = [[] for _ in range(world_size)], [0.0] * world_size
    for i in order:
        r = loads.index(min(loads))
        parts[r].append(i)
        loads[r] += durations[i]
    return [sorted(p) for p in parts]


def reduce_max_seconds(seconds, device=None):
    """MAX over ranks of a wall-clock interval (the bench contract); identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
