"""F3 inside F4 (gicp.hip, round 5): for a pair whose steps have become small, k_gicp_linearize checks the search margins of the points
it reads anyway and k_gicp_research searches the few that have left theirs — k_gicp_correspond is not launched for such a pair.
Exactness does not depend on it: with the fold on (default) and off (MI355_CORR_FOLD=0: every pair through k_gicp_correspond every
iteration) the incremental correspondences equal a from-scratch search at the final poses, and the poses agree to fp64 summation
order.  The batch's capacity is not a multiple of the linearize tile and one source fills it (the per-wave lists of the last tile).
MI355_CORR_FOLD is read once per process, so both arms are child processes; the file name sorts first like the other children-only tests."""
import os
import subprocess
import sys

import numpy as np
import pytest

from lidar_graph_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.children_only]


def _probe(fold, tmp_path):
    out = os.path.join(str(tmp_path), f"fold{fold}.npz")
    env = dict(os.environ, MI355_CORR_FOLD=str(fold))
    r = subprocess.run([sys.executable, "tests/probes/fold_probe.py", out], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PROBE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


def test_the_folded_margin_check_is_exact(tmp_path):
    on, off = _probe(1, tmp_path), _probe(0, tmp_path)
    assert on["research_launches"][0] == 14 and off["research_launches"][0] == 0  # one k_gicp_research launch per iteration, or none
    for arm in (on, off):
        assert arm["verify"][0] > 0 and arm["verify"][1] == 0  # incremental == from scratch at the final poses
        assert (arm["it"] == 14).all()
        np.testing.assert_array_equal(arm["T"], arm["T2"])
    np.testing.assert_array_equal(on["inl"], off["inl"])
    for p in range(len(on["T"])):
        dt, dr = synth.pose_error(on["T"][p], off["T"][p])
        assert dt < 1e-6 and dr < 1e-7, (p, dt, dr)
    np.testing.assert_allclose(on["err"], off["err"], rtol=1e-9)
