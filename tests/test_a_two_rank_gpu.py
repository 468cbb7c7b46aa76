"""The N > 1 path with REAL registrations: benchmarks/config5.py under torch.distributed.run with two ranks
sharing the one GPU of the test box (BENCH_DEVICE=0, gloo for the gather), against the same program run as a single rank.
The file name sorts first on purpose: the ranks are child processes, and the pytest process has not touched the GPU yet when
it starts them (children are started, never exec'd into)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.children_only]

ARGS = ["--candidates", "12", "--submaps", "3", "--casts-per-submap", "2", "--n-azimuth", "256", "--capacity", "65536", "--min-good", "0"]


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_ranks_register_their_shards_and_gather():
    env = dict(os.environ, BENCH_DEVICE="0", BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "benchmarks/config5.py", "--gpus", "2"] + ARGS, env)
    one = _run([sys.executable, "benchmarks/config5.py", "--gpus", "1"] + ARGS, env)
    assert two["n_gpus"] == 2 and two["collective_ranks"] == 2 and two["all_reported"] and one["all_reported"]
    assert two["candidates"] == 12 and two["within_5cm_of_truth"] >= 10
    # sharding must not change any result: same acceptance and accuracy as the single-rank run
    for k in ("accepted(fitness<0.3)", "within_5cm_of_truth", "median_pose_err_m", "mean_iterations"):
        assert two[k] == one[k], (k, two[k], one[k])


def test_rccl_executes_the_gather_in_a_one_rank_group():
    """`sharding.gather_rows` / `max_over_ranks` on their `nccl` branch: device tensors, all_gather + all_reduce through RCCL.  The pool's
    boxes have ONE GPU, so the communicator has one rank (benchmarks/rccl_probe.py forces the collective that a one-rank group would skip)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(29900 + (os.getpid() % 90)))
    r = subprocess.run([sys.executable, "benchmarks/rccl_probe.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK world=1 backend=nccl" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_rccl_with_two_ranks_needs_two_devices():
    """Why the two-rank test above gathers over gloo: RCCL refuses two ranks on one device ("Duplicate GPU detected").  If a future RCCL
    accepts them the probe must simply succeed."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + (os.getpid() % 200)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), "benchmarks/rccl_probe.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode == 0:
        assert "RCCL_OK world=2" in r.stdout
    else:
        assert "Duplicate GPU detected" in r.stderr + r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_py_runs_as_two_ranks():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU) - here both ranks on device 0 with gloo
    for the barrier and the max-over-ranks reduce: the base pairs are cast once and shared through /dev/shm, every rank times its own
    batches, rank 0 prints ONE line whose value is the whole job's (both ranks' pairs over the slower rank's time)."""
    env = dict(os.environ, BENCH_DEVICE="0", BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29300 + (os.getpid() % 250)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", "2", "--pairs", "128", "--steps", "2", "--warmup", "1", "--bases", "2", "--no-h2d", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["config"]["pairs_per_gpu_per_step"] == 128
    assert abs(d["value"] - 2 * 128 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]  # whole-job pairs over the max-over-ranks step time
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"mi355_bench_bases_{port}_")]  # the shared file is gone again
