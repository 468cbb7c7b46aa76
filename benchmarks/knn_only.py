"""Tuning aid: the preparation chain of a batch ALONE (prefilter + grid + k-NN covariances of 1 024 clouds), kernel families timed with
the library's own HIP-event brackets — for A/B of k-NN kernel variants whose outputs may be wrong (ablations)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

torch.cuda.init()
from lidar_graph_slam_amd import _lib as L  # noqa: E402
from lidar_graph_slam_amd import synth  # noqa: E402
from lidar_graph_slam_amd.context import Context  # noqa: E402
from lidar_graph_slam_amd.registration import BatchRegistration  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = Context(0)
dev = torch.device("cuda", 0)
base = []
for i in range(4):
    T_gt = synth.se3(1.0 - 0.1 * i, 0.1 + 0.05 * i, 0.0, 0.0, 0.0, np.deg2rad(2.0 - 0.4 * i))
    T_tgt = synth.se3(7.0 * i, 0.3 * i, 0.0, 0.0, 0.0, 0.02 * i)
    base.append(synth.make_pair(T_gt=T_gt, T_target=T_tgt, seed=11 + 17 * i)[1])
cap = base[0].shape[0]
tgt = torch.stack([torch.from_numpy(base[p % 4]) for p in range(P)]).to(dev)
counts = np.full(P, cap, dtype=np.int32)
torch.cuda.synchronize()
b = BatchRegistration(P, P, cap, ctx=ctx, max_iterations=1)
b.set_timing(True)
b.set_side_streams(False)
pre = L.prefilter_params(min_distance=1.0, leaf_size=0.2, mean_k=0)
for it in range(4):
    b.set_targets_dev(tgt.data_ptr(), counts, pre)
    ctx.synchronize()
    kt = b.kernel_times()
    if it:
        print("  ".join(f"{k.split('(')[0]} {v[0] / max(v[1], 1):.3f}" for k, v in kt.items()))
