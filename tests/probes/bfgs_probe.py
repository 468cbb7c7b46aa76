"""Probe (run by hand on the GPU box, optionally under MI355_LIB=<another build>): pclomp's BFGS GICP through the registration object
against the oracle after ONE outer iteration and at convergence, on the shifted pairs of tests/test_gpu_gicp_bfgs_batch.py."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidar_graph_slam_amd import registration as R, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

s, t, T = synth.make_pair(n_azimuth=256)
s, t = s[np.abs(s[:, :3]).sum(1) > 0], t[np.abs(t[:, :3]).sum(1) > 0]
shifts = [np.eye(4, dtype=np.float32), synth.se3(0.2, -0.1, 0, 0, 0, 0.01).astype(np.float32), synth.se3(-0.3, 0.2, 0.05, 0.01, -0.01, -0.02).astype(np.float32)]
for mi in (1, 64):
    for j, sh in enumerate(shifts):
        tj = O.transform(t, sh)
        reg = R.GeneralizedIterativeClosestPoint(max_iterations=mi)
        reg.setInputTarget(tj)
        reg.setInputSource(s)
        reg.align()
        ref = O.gicp_bfgs_align(s, tj, O.gicp_bfgs_params(max_iterations=mi))
        dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
        print(f"max_iterations {mi} shift {j}: gpu vs oracle {dt:.6f} m {dr:.2e} rad; iterations {reg._result['iterations']} / {ref['iterations']}; "
              f"cost {reg._result['error']:.9g} / {ref['error']:.9g}")

tj = O.transform(t, shifts[1])
reg = R.GeneralizedIterativeClosestPoint()
reg.setInputTarget(tj)
reg.setInputSource(s)
ts = []
for _ in range(7):
    t0 = time.perf_counter()
    reg.align()
    ts.append(time.perf_counter() - t0)
print(f"align of one pair ({len(s)} x {len(tj)} points, {reg._result['iterations']} outer iterations): median {np.median(ts) * 1e3:.2f} ms, min {np.min(ts) * 1e3:.2f} ms")
