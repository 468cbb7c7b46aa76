"""Child-process probe of tests/test_a_mask_rows.py: a FAST_GICP batch of 16 pairs whose capacity is NOT a multiple of the 256-point
tile of k_gicp_linearize (4 096 + 100) and whose first source fills it, registered with forced iterations; poses, iteration counts and
the incremental-correspondence probe go to an .npz.  MI355_CORR_MASK is read once per process."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401  (torch's ROCm runtime first: tests/conftest.py)

if torch.cuda.is_available():
    torch.cuda.init()
from lidar_graph_slam_amd import synth  # noqa: E402
from lidar_graph_slam_amd.registration import BatchRegistration  # noqa: E402

out_path = sys.argv[1]
src, tgt, _ = synth.make_pair(n_azimuth=64)
src, tgt = src[np.abs(src[:, :3]).sum(1) > 0], tgt[np.abs(tgt[:, :3]).sum(1) > 0]
S = 16
cap = 4096 + 100
rng = np.random.default_rng(9)
srcs = np.zeros((S, cap, 4), np.float32)
tgts = np.zeros((S, cap, 4), np.float32)
ns, nt = [], []
for i in range(S):
    # pair 0 fills the capacity (points repeated with a small jitter), the others are ragged
    n = cap if i == 0 else len(src) - 37 * i
    pick = np.arange(n) % len(src)
    srcs[i, :n] = src[pick]
    if i == 0:
        srcs[i, len(src):n, :3] += rng.normal(0.0, 0.01, (n - len(src), 3)).astype(np.float32)
    ns.append(n)
    m = len(tgt) - 11 * i
    tgts[i, :m] = tgt[:m]
    nt.append(m)
res = {}
b = BatchRegistration(S, S, cap, max_iterations=14, force_iterations=1)
b.set_targets(tgts, nt)
b.set_sources(srcs, ns)
b.align()
r = b.results(raw=True)
res["T"] = np.array(r["T"])
res["it"] = np.array(r["iterations"])
res["err"] = np.array(r["error"])
res["inl"] = np.array(r["n_inliers"])
res["verify"] = np.array(b.verify_correspondences())
b.close()
np.savez(out_path, **res)
print("PROBE_OK", os.environ.get("MI355_CORR_MASK", "default"))
