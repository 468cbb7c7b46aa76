"""Does RCCL run the gather of the sharded path on THIS box?  Two ranks on device 0 (the pool's boxes have one GPU): RCCL may
refuse two ranks on one device; then a one-rank communicator at least executes the all_gather through RCCL.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 benchmarks/rccl_probe.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.getcwd())
from lidar_graph_slam_amd import sharding  # noqa: E402

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
n = 37
tgt = np.random.default_rng(1).integers(0, 5, n)
mine = sharding.shard_by_target(tgt, world)[rank]
rows = np.stack([mine.astype(np.float64), tgt[mine].astype(np.float64), np.full(len(mine), rank, np.float64)], axis=1)
out = sharding.gather_rows(mine, rows, n, force_collective=True)
assert not np.isnan(out).any() and (out[:, 0] == np.arange(n)).all() and (out[:, 1] == tgt).all(), out
t = sharding.max_over_ranks(1.0 + rank)
assert t == float(world), t
if rank == 0:
    print(f"RCCL_OK world={world} backend={dist.get_backend()} gathered {n} rows")
dist.destroy_process_group()
