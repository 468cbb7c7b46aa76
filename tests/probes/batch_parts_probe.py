"""Child-process probe of tests/test_a_batch_parts.py: one FAST_GICP batch (64 pairs against 5 targets, pairs dealt to targets
round-robin so that every part of the pipeline needs every part of the targets; then 61 pairs, one target each) registered with
forced iterations and with early stopping, side streams on and off; everything a caller can observe goes to an .npz.
MI355_BATCH_PARTS (read once per process) sets the number of parts the batch is cut into."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401  (torch's ROCm runtime first: tests/conftest.py)

if torch.cuda.is_available():
    torch.cuda.init()
from lidar_graph_slam_amd import _lib as L  # noqa: E402
from lidar_graph_slam_amd import synth  # noqa: E402
from lidar_graph_slam_amd.registration import BatchRegistration  # noqa: E402

out_path = sys.argv[1]
src, tgt, _ = synth.make_pair(n_azimuth=64)
pre = L.prefilter_params(min_distance=1.0, leaf_size=0.2, mean_k=0)
S, U = 64, 5
cap = src.shape[0]
srcs = np.zeros((S, cap, 4), np.float32)
tgts = np.zeros((U, cap, 4), np.float32)
ns, nt = [], []
for i in range(S):
    n = cap - 13 * i
    srcs[i, :n] = src[:n]
    ns.append(n)
for u in range(U):
    n = cap - 29 * u
    tgts[u, :n] = tgt[:n]
    nt.append(n)
rng = np.random.default_rng(5)
guesses = np.stack([synth.se3(*(rng.uniform(-0.2, 0.2, 2)), 0.0, 0.0, 0.0, rng.uniform(-0.02, 0.02)) for _ in range(S)]).astype(np.float32)
res = {}
for force in (1, 0):
    b = BatchRegistration(S, U, cap, max_iterations=12 if force else 40, force_iterations=force)
    b.set_timing(True)
    for mode in (True, False):
        b.set_side_streams(mode)
        b.set_targets(tgts, nt, pre)
        b.set_sources(srcs, ns, pre)
        b.align(np.arange(S) % U, guesses, with_fitness=True)
        r, fit = b.results(with_fitness=True, raw=True)
        key = f"f{force}_s{int(mode)}"
        res[key + "_T"] = np.array(r["T"])
        res[key + "_it"] = np.array(r["iterations"])
        res[key + "_conv"] = np.array(r["converged"])
        res[key + "_err"] = np.array(r["error"])
        res[key + "_n"] = np.stack([np.array(r["n_source"]), np.array(r["n_target"]), np.array(r["n_inliers"])])
        res[key + "_fit"] = fit
        chk, bad = b.verify_correspondences()
        res[key + "_verify"] = np.array([chk, bad])
        d = b.diagnostics()
        res[key + "_fc"] = d["first_converged"]
        res[key + "_knn"] = np.concatenate([d["knn_points"], d["knn_handed_over"]])
        kt = b.kernel_times()
        res[key + "_launches_knn"] = np.array([kt["k_knn_cov_tile"][1]])
    b.close()
# fewer live slots than the batch was made for, identity mapping (one target per pair)
b = BatchRegistration(S, S, cap, max_iterations=8, force_iterations=1)
b.set_live(61, 61)
b.set_targets(np.concatenate([tgts] * 13)[:61], (nt * 13)[:61], pre)
b.set_sources(srcs[:61], ns[:61], pre)
b.align()
res["ident_T"] = np.array(b.results(raw=True)["T"])
b.close()
np.savez(out_path, **res)
print("PROBE_OK", os.environ.get("MI355_BATCH_PARTS", "default"))
