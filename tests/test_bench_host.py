"""CPU test of bench.py's host plumbing for several ranks on one node: the base pairs are ray-cast ONCE, by local
rank 0, and handed to the other local ranks through a file in /dev/shm - every rank gets the same arrays, and a rank other than 0 does
not cast.  (The GPU side of `bench.py --gpus N` - pairs sharded over ranks, barrier + max-over-ranks timing - is
tests/test_sharding.py's gloo tests and tests/test_a_two_rank_gpu.py.)"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import bench
from lidar_graph_slam_amd import synth
calls = []
real = synth.make_pair_job
def counted(spec):
    calls.append(1)
    return real(spec)
synth.make_pair_job = counted            # (a pool of one worker calls it in-process: n_base = 1)
t0 = time.time()
pairs = bench.base_pairs(int(os.environ["RANK"]), 1)
s, t, g = pairs[0]
print(json.dumps({"rank": int(os.environ["RANK"]), "casts": len(calls), "n_src": int(len(s)), "n_tgt": int(len(t)),
                  "sum": float(np.asarray(s, np.float64).sum() + np.asarray(t, np.float64).sum()), "g": np.asarray(g).round(9).tolist()}))
"""


def test_base_pairs_are_cast_once_per_node_and_shared(tmp_path):
    port = str(40000 + os.getpid() % 20000)
    shared = f"/dev/shm/mi355_bench_bases_{port}_none_{os.getpid()}_1.npz"  # (bench.shared_bases_path: port, torchrun's run id, the launcher's pid)
    if os.path.exists(shared):
        os.remove(shared)
    procs = []
    try:
        for r in (1, 0):  # rank 1 first: it has to wait for rank 0's file
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", WORLD_SIZE="2", MASTER_PORT=port, MASTER_ADDR="127.0.0.1", TORCHELASTIC_RUN_ID="none")
            procs.append(subprocess.Popen([sys.executable, "-c", _CHILD, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = []
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, e[-2000:]
            outs.append(json.loads(o.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if os.path.exists(shared):
            os.remove(shared)
    by_rank = {o["rank"]: o for o in outs}
    assert by_rank[0]["casts"] == 1 and by_rank[1]["casts"] == 0  # cast once, by local rank 0
    assert by_rank[0]["n_src"] == by_rank[1]["n_src"] > 60000 and by_rank[0]["n_tgt"] == by_rank[1]["n_tgt"]
    assert by_rank[0]["sum"] == by_rank[1]["sum"] and by_rank[0]["g"] == by_rank[1]["g"]  # the same arrays, bit for bit
    np.testing.assert_allclose(np.asarray(by_rank[0]["g"])[3], [0, 0, 0, 1])


def test_rank_bootstrap_with_eight_local_ranks():
    """`bench.py --gpus 8` as the driver launches it on an 8-GPU node, as far as a machine without one can run it:
    torch.distributed.run with 8 local ranks, gloo, BENCH_BOOTSTRAP_ONLY (the GPU section stubbed).  Local rank 0 casts the base pair once
    while 7 ranks wait on /dev/shm; all 8 join the process group, pass the barrier, the shared file disappears, the max-over-ranks reduce
    returns the slowest rank's value and rank 0 alone prints.  No scaling curve has been measured anywhere - this shows the host side only."""
    env = dict(os.environ, BENCH_BOOTSTRAP_ONLY="1", BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    port = 28000 + (os.getpid() % 900)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", "8", "--bases", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["bootstrap_only"] and d["n_gpus"] == 8 and d["local_world"] == 8
    assert d["base_pairs_identical_on_every_rank"] and not d["shared_file_left_behind"]
    assert abs(d["max_over_ranks"] - 1.007) < 1e-9  # rank 7's value
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"mi355_bench_bases_{port}_")]
