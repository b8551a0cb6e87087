"""CPU, 2 ranks (gloo): ``python bench.py --gpus 2`` outside any launcher must start two ranks itself and report
``n_gpus: 2`` (round 1's bench.py parsed --gpus and ignored it).  The per-step work is a stub (--stub-workload: the HIP
path needs a GPU); the launch plumbing - self-spawn under torch.distributed.run, ranks from the environment, barrier,
MAX-over-ranks timing, rank 0 printing ONE JSON line - is the code every real mode of bench.py runs."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=240):
    env = dict(os.environ, AUDIOCAPTION_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, cwd=REPO, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_self_spawns_two_ranks():
    res = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--stub-workload"])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["warmup"] == 1
    assert res["config"]["backend"] == "gloo" and res["config"]["requested_gpus"] == 2
    assert res["scaling"] == "weak" and res["value"] > 0 and res["ms_per_step"] > 0
    # the line says where every rank sat (two different processes, ranks 0 and 1)
    assert [r["rank"] for r in res["ranks"]] == [0, 1] and res["ranks"][0]["pid"] != res["ranks"][1]["pid"]


def test_bench_under_a_launcher_reads_the_ranks_from_the_environment():
    """The driver's own command: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N."""
    env = dict(os.environ, AUDIOCAPTION_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29731", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--stub-workload"], env=env, cwd=REPO, timeout=240, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2


def test_bench_gpus_8_stub_describes_every_rank():
    """The driver's 8-GPU command on CPU ranks (gloo): the line carries one record per rank, per-rank seconds and the
    collective timing object, so that the first real 8-GPU run describes itself."""
    res = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--stub-workload"], timeout=600)
    assert res["n_gpus"] == 8 and [r["rank"] for r in res["ranks"]] == list(range(8))
    assert len({r["pid"] for r in res["ranks"]}) == 8 and len(res["seconds_per_rank"]) == 8
    assert res["rccl"]["world_size"] == 8 and res["rccl"]["backend"] == "gloo"
    assert res["rccl"]["grad_sync_default"] == "all_reduce"          # rs_ag only on RCCL with >= 4 ranks
    assert res["ms_per_step"] * 1e-3 * res["steps"] >= max(res["seconds_per_rank"]) - 1e-9


def test_default_gradient_sync_by_backend_and_world_size():
    from audiocaption_amd.train import default_grad_sync
    assert default_grad_sync(8, "nccl") == "rs_ag" and default_grad_sync(4, "nccl") == "rs_ag"
    assert default_grad_sync(2, "nccl") == "all_reduce" and default_grad_sync(8, "gloo") == "all_reduce"


def test_bench_single_rank_stub():
    res = _run(["--steps", "2", "--warmup", "0", "--stub-workload"])
    assert res["n_gpus"] == 1


@pytest.mark.gpu
def test_bench_two_ranks_real_workload_on_one_gpu():
    """The N > 1 flow of the REAL inference bench (clips sharded over ranks, no data-path collective, barrier + MAX timing,
    rank records, per-rank seconds) with two gloo ranks sharing the only GPU of the box - RCCL refuses two ranks on one
    device, everything else is the code an 8-GPU run executes."""
    res = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--seconds", "2", "--max-length", "6",
                "--no-tiers", "--no-train", "--no-effb2", "--no-steady-state", "--no-cpu-baseline"], timeout=600)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 8 and res["value"] > 0
    assert [r["rank"] for r in res["ranks"]] == [0, 1] and len(res["seconds_per_rank"]) == 2
    assert res["rccl"]["world_size"] == 2 and res["scaling"] == "weak"
    assert abs(res["value"] - 8 * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]
