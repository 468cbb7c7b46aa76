"""The order in which the tile searches take the points (cloud.hip, build_tile_order: 2-D Hilbert curve over the two long axes of a
cloud, 3-D for a cloud that is not 2.5-D) decides which 64 points share a wave, never a result.  Batches use it by default; the
k-NN / covariance parity tests of tests/test_gpu_filters.py call the single-cloud entry points, which do not — so they are run once
more in a child process with MI355_TILE_ORDER=2 (single clouds too): the lattice with its ties, the sparse uniform volume (the 3-D
curve), duplicates, the cloud of exactly k points, every template width and the four full-size clouds, every point against the
oracle's neighbourhood.  (Child process: the switch is read once per process.  The file name sorts first like the other
children-only test: the pytest process has not touched the GPU yet when it starts the child.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.children_only]


def test_covariances_are_exact_along_the_tile_order():
    env = dict(os.environ, MI355_TILE_ORDER="2", MI355_KNN_STATS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_filters.py", "-q", "-x", "-m", "gpu", "-k", "covariances or sor", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]
