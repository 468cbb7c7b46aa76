"""Tuning aid for BASELINE configs[3] (per-scan odometry): wall time of every callback_cloud of a short stream, split by key-frame /
non-key-frame scans.  Run under `rocprofv3 --kernel-trace --stats` to see how much of the wall time the GPU is busy."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from lidar_graph_slam_amd import synth  # noqa: E402
from lidar_graph_slam_amd.filters import PointsPreFiltering  # noqa: E402
from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
method = sys.argv[2] if len(sys.argv) > 2 else "FAST_GICP"
street = synth.street(n * 0.8 + 10.0)
poses = synth.trajectory(n, seed=3)
pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
scans = [pre.callback_sensor_points(synth.cast_scan(street, P, seed=300 + i, ground_max_range=20.0)) for i, P in enumerate(poses)]
sm = LidarScanMatcher(registration_method=method)
ts = []
for s in scans:
    t0 = time.perf_counter()
    o = sm.callback_cloud(s)
    ts.append((time.perf_counter() - t0, o["new_key_frame"], o["target_points"], o["iterations"]))
kf = [t[0] for t in ts[25:] if t[1]]
nk = [t[0] for t in ts[25:] if not t[1]]
print(f"{method}: scans {n}, after warm-up: key-frame scans {1e3 * np.mean(kf):.2f} ms ({len(kf)}), others {1e3 * np.mean(nk):.2f} ms ({len(nk)}), "
      f"mean {1e3 * np.mean([t[0] for t in ts[25:]]):.2f} ms, iterations {np.mean([t[3] for t in ts[25:]]):.1f}, target points {ts[-1][2]}")
