"""Parity bands of BASELINE.md section 4, in one place.

* fixed iteration count, same inputs:  ||dt|| <= 1 mm, angle <= 1e-4 rad  (`assert_fixed`)
* early stopping, same inputs:         the convergence band, measured as fast_gicp's is_converged measures a step
                                       (SURVEY F9): max|dt_i| < 0.01 m, max|dR - I| < 2e-3   (`assert_in_band`)
Every comparison is also appended to gpurun_out/parity_deltas.jsonl (scratch) so that the worst observed values can be
quoted in BASELINE.md; `benchmarks/summarize_parity.py` turns it into profiles/rNN_parity_deltas.txt."""
import json
import os

from lidar_graph_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAND_T, BAND_R = 0.01, 2e-3  # transformation_epsilon (lidar_scan_matcher.param.yaml:11), fast_gicp's rotation_epsilon_ default


def _log(tag, **kw):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_deltas.jsonl"), "a") as f:
            f.write(json.dumps({"tag": tag, **kw}) + "\n")
    except OSError:
        pass


def deltas(T_a, T_b):
    bt, br = synth.band_error(T_a, T_b)
    dt, dr = synth.pose_error(T_a, T_b)
    return bt, br, dt, dr


def assert_in_band(T_a, T_b, tag, scale=1.0):
    """`scale` > 1 only where the two runs did NOT see the same inputs (a stream whose sub-maps were assembled from poses that
    already differ inside the band); the call site says why."""
    bt, br, dt, dr = deltas(T_a, T_b)
    _log(tag, band_t=bt, band_r=br, dt=dt, dr=dr, scale=scale)
    if not os.environ.get("PARITY_LOG_ONLY"):  # a measuring run records every delta and judges none
        assert bt <= scale * BAND_T and br <= scale * BAND_R, (tag, bt, br, dt, dr)
    return bt, br


def assert_fixed(T_a, T_b, tag, t=1e-3, r=1e-4):
    bt, br, dt, dr = deltas(T_a, T_b)
    _log(tag, band_t=bt, band_r=br, dt=dt, dr=dr, fixed=True)
    if not os.environ.get("PARITY_LOG_ONLY"):
        assert dt <= t and dr <= r, (tag, dt, dr)
    return dt, dr
