"""The "margin-safe" bits of the incremental correspondence search (gicp.hip: k_gicp_linearize writes one word per 64 points, the next
k_gicp_correspond reads them) are rows of ceil(capacity / 64) words per pair.  The last 256-point tile of a pair reaches past its row
when the capacity is not a multiple of the tile; until round 5 its words landed in the NEXT pair's row as all-ones ("safe"), and that
pair then skipped the margin check of up to 192 of its points.  Here the capacity is 4 096 + 100 and the first source fills it: the
incremental correspondences must equal a from-scratch search, and the poses must be BIT-equal to a run without the bits
(MI355_CORR_MASK=0: the bits only skip checks that would pass).  The switch is read once per process, so both arms are child
processes; the file name sorts first like the other children-only tests."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.children_only]


def _probe(mask, tmp_path):
    out = os.path.join(str(tmp_path), f"mask{mask}.npz")
    env = dict(os.environ, MI355_CORR_MASK=str(mask))
    r = subprocess.run([sys.executable, "tests/probes/mask_rows_probe.py", out], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PROBE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


def test_margin_bits_stay_inside_their_pairs_row(tmp_path):
    on, off = _probe(1, tmp_path), _probe(0, tmp_path)
    for arm in (on, off):
        assert arm["verify"][0] > 0 and arm["verify"][1] == 0  # incremental == from scratch at the final poses
        assert (arm["it"] == 14).all()
    for k in ("T", "err", "inl"):
        np.testing.assert_array_equal(on[k], off[k], err_msg=k)
