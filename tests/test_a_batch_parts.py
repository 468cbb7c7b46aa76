"""The pipeline inside mi355_batch_* (api.hip, round 4): a batch is cut into parts whose preparation chains and LM loops overlap on
internal streams.  Parts only reorder independent work: everything a caller can observe - poses, iteration counts, errors, fitness,
the incremental-correspondence probe, the iterations-to-convergence diagnostics - must be BIT-EQUAL whatever the number of parts,
with the chains on side streams or not, with forced iterations or early stopping (whose polls are non-blocking since round 4).
The number of parts is read once per process (MI355_BATCH_PARTS), so every variant is a child process; the file name sorts first
like the other children-only tests."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.children_only]


def _probe(parts, tmp_path):
    out = os.path.join(str(tmp_path), f"parts{parts}.npz")
    env = dict(os.environ, MI355_BATCH_PARTS=str(parts))
    r = subprocess.run([sys.executable, "tests/probes/batch_parts_probe.py", out], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PROBE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


def test_parts_only_reorder_independent_work(tmp_path):
    one = _probe(1, tmp_path)
    # sanity of the reference run itself: registrations are registrations, the incremental search is exact, early stopping stops
    assert one["f1_s1_verify"][0] > 0 and one["f1_s1_verify"][1] == 0 and one["f0_s1_verify"][1] == 0
    assert (one["f0_s1_conv"] == 1).all() and one["f0_s1_it"].max() < 40 and (one["f1_s1_it"] == 12).all()
    assert one["f1_s1_launches_knn"][0] == 2
    for parts in (3, 4):
        got = _probe(parts, tmp_path)
        # it really ran in parts: one k-NN launch per part of the 64 sources and per non-empty part of the 5 targets
        assert got["f1_s1_launches_knn"][0] == parts + -(-5 // -(-5 // parts)), parts
        for k in sorted(one):
            if k.endswith("_launches_knn"):
                continue
            np.testing.assert_array_equal(got[k], one[k], err_msg=f"{k} with {parts} parts")
    # side streams on / off inside one process, forced and early-stopped
    for f in (0, 1):
        for k in ("T", "it", "conv", "err", "fit", "fc"):
            np.testing.assert_array_equal(one[f"f{f}_s1_{k}"], one[f"f{f}_s0_{k}"], err_msg=k)
