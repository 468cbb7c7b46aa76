"""The order in which the tile searches take the points (cloud.hip, build_tile_order: 2-D Hilbert curve over the two long axes of a
cloud, 3-D for a cloud that is not 2.5-D) decides which 64 points share a wave, never a result.  Batches use it by default; the
k-NN / covariance parity tests of tests/test_gpu_filters.py call the single-cloud entry points, which do not — so they are run once
more in a child process with MI355_TILE_ORDER=2 (single clouds too): the lattice with its ties, the sparse uniform volume (the 3-D
curve), duplicates, the cloud of exactly k points, every template width and the four full-size clouds, every point against the
oracle's neighbourhood.  (Child process: the switch is read once per process.  The file name sorts first like the other
children-only test: the pytest process has not touched the GPU yet when it starts the child.)
The second test does the same for a launch geometry the small test batches never see (MI355_BLOCKS_PER_PAIR)."""
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


def test_incremental_correspondences_are_exact_when_a_block_walks_many_tiles():
    """k_gicp_correspond searches its queue a block's worth at a time and carries what is left to the block's next tile.  The small
    batches of tests/test_gpu_gicp.py give every block one tile; MI355_BLOCKS_PER_PAIR=1 / 2 (the geometry of a 4 096-pair batch)
    makes one block walk all / half of a pair's tiles, leftovers and all, and the probe that compares the incremental
    correspondences with a search of every point from scratch (mi355_batch_verify_correspondences) must still find no mismatch."""
    for g in ("1", "2"):
        env = dict(os.environ, MI355_BLOCKS_PER_PAIR=g)
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_gicp.py", "-q", "-x", "-m", "gpu", "-k", "incremental_correspondences or side_streams",
                            "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (g, r.stdout[-3000:] + r.stderr[-2000:])
        assert " passed" in r.stdout and "failed" not in r.stdout, (g, r.stdout[-2000:])
