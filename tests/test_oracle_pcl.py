"""Pins the C++ oracle's PCL-piece restatements with analytic known answers and
against the independent NumPy/SciPy restatement (oracle/numpy_ref.py).  PARITY UNPINNED upstream: see oracle/oracle.h."""
import numpy as np
import pytest

from oracle import numpy_ref as NR
from oracle import oracle_py as O


def _boundary_points(n=200000, seed=5):
    """Points whose fp32 squared norm depends on the association: x*x + (y*y + z*z) (Eigen's unrolled 3-element redux) differs
    from (x*x + y*y) + z*z, scaled onto the 1.0 m threshold so that the two orders disagree about the predicate."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    v *= np.float32(1.0) + rng.integers(-2, 3, size=(n, 1)).astype(np.float32) * np.float32(2.0 ** -24)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    eigen = np.sqrt(x * x + (y * y + z * z))
    other = np.sqrt((x * x + y * y) + z * z)
    pick = (1.0 < eigen.astype(np.float64)) != (1.0 < other.astype(np.float64))
    pts = np.zeros((int(pick.sum()), 4), np.float32)
    pts[:, :3] = v[pick]
    pts[:, 3] = np.arange(len(pts))
    return pts, (1.0 < eigen[pick].astype(np.float64))


def test_distance_filter_sums_squares_in_eigens_order():
    pts, keep = _boundary_points()
    assert len(pts) > 100 and 0 < keep.sum() < len(pts)
    np.testing.assert_array_equal(O.distance_filter(pts, 1.0), pts[keep])
    np.testing.assert_array_equal(NR.distance_filter(pts, 1.0), pts[keep])


def test_distance_filter_strict_and_order():
    pts = np.array([[1, 0, 0, 1], [0.5, 0, 0, 2], [0, 2, 0, 3], [0, 0, 0, 4], [np.nan, 0, 0, 5], [0.6, 0.8, 0, 6]], dtype=np.float32)
    out = O.distance_filter(pts, 1.0)
    # strict '<': the two points at exactly 1.0 m are dropped; NaN compares false and is dropped; order preserved
    assert out[:, 3].tolist() == [3.0]
    assert np.array_equal(O.distance_filter(pts, 0.4)[:, 3], np.array([1, 2, 3, 6], dtype=np.float32))


def test_voxel_grid_known_answer():
    # 10 hand-placed points, leaf 1.0: voxels (0,0,0) x3, (1,0,0) x2, (0,1,0) x1, (2,2,1) x4
    pts = np.array([
        [0.1, 0.1, 0.1, 1.0], [0.2, 0.3, 0.4, 2.0], [0.9, 0.8, 0.7, 3.0],
        [1.5, 0.5, 0.5, 4.0], [1.1, 0.1, 0.9, 6.0],
        [0.5, 1.5, 0.5, 7.0],
        [2.1, 2.2, 1.3, 1.0], [2.9, 2.8, 1.7, 1.0], [2.5, 2.5, 1.5, 2.0], [2.5, 2.5, 1.5, 4.0],
    ], dtype=np.float32)
    out, ovf = O.voxel_grid(pts[::-1].copy(), 1.0)  # input order must not matter for the centroid set/order
    assert not ovf
    exp = np.array([
        [0.4, 0.4, 0.4, 2.0],        # idx 0
        [1.3, 0.3, 0.7, 5.0],        # idx 1
        [0.5, 1.5, 0.5, 7.0],        # idx dx = 3
        [2.5, 2.5, 1.5, 2.0],        # idx 2 + 2*3 + 1*9 = 17
    ], dtype=np.float32)
    assert out.shape == exp.shape
    np.testing.assert_allclose(out, exp, atol=1e-6)


def test_voxel_grid_vs_numpy(medium_pair):
    src, _, _ = medium_pair
    src = O.distance_filter(src, 1.0)
    for leaf in (0.1, 0.2, 0.5):
        a, ovf = O.voxel_grid(src, leaf)
        b = NR.voxel_grid(src, leaf)
        assert not ovf and a.shape == b.shape
        np.testing.assert_allclose(a, b, atol=2e-5)
        # idempotence: every point alone in its voxel maps to itself (set-wise)
        a2, _ = O.voxel_grid(a, leaf)
        assert len(a2) <= len(a)


def test_voxel_grid_overflow_passthrough():
    pts = np.array([[0, 0, 0, 1], [1000, 1000, 1000, 2], [-1000, 5, 7, 3]], dtype=np.float32)
    out, ovf = O.voxel_grid(pts, 0.001)  # 2e6^3 cells > INT32_MAX
    assert ovf and np.array_equal(out, pts)


def test_voxel_grid_empty_and_single():
    out, ovf = O.voxel_grid(np.zeros((0, 4), np.float32), 0.2)
    assert len(out) == 0 and not ovf
    one = np.array([[1, 2, 3, 4]], dtype=np.float32)
    out, _ = O.voxel_grid(one, 0.2)
    np.testing.assert_array_equal(out, one)


def test_knn_kdtree_equals_bruteforce(small_pair):
    s, t, _ = small_pair
    for k in (1, 20, 31):
        i1, d1 = O.knn(t, s, k)
        i2, d2 = O.knn(t, s, k, brute_force=True)
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(d1, d2)
        assert (np.diff(d1, axis=1) >= 0).all()


def test_knn_ties_lowest_index():
    pts = np.zeros((8, 4), np.float32)
    pts[:, 0] = [1, 1, -1, -1, 2, 2, 0.5, 0.5]  # pairs of duplicates
    q = np.zeros((1, 4), np.float32)
    i, d = O.knn(pts, q, 4)
    assert i[0].tolist() == [6, 7, 0, 1]
    np.testing.assert_allclose(d[0], [0.25, 0.25, 1, 1])


def test_knn_vs_scipy(medium_pair):
    from scipy.spatial import cKDTree

    src, tgt, _ = medium_pair
    tgt = O.distance_filter(tgt, 1.0)
    src = O.distance_filter(src, 1.0)[:3000]
    i, d2 = O.knn(tgt, src, 5)
    dd, ii = cKDTree(tgt[:, :3].astype(np.float64)).query(src[:, :3].astype(np.float64), k=5)
    np.testing.assert_allclose(d2, dd * dd, rtol=1e-5, atol=1e-9)
    assert (i == ii).mean() > 0.999  # fp32 vs fp64 distances may swap near-ties


def test_sor_planted_outlier():
    rng = np.random.default_rng(0)
    g = np.stack(np.meshgrid(np.arange(20), np.arange(20)), -1).reshape(-1, 2) * 0.1
    pts = np.zeros((401, 4), np.float32)
    pts[:400, :2] = g + rng.normal(0, 0.002, g.shape)
    pts[400] = [1.0, 1.0, 3.0, 9.0]  # 3 m above the patch
    out, keep, md = O.sor(pts, 30, 1.2)
    assert not keep[400] and md[400] > 2.9
    out2, keep2, md2 = NR.sor(pts, 30, 1.2)
    np.testing.assert_array_equal(keep, keep2)
    np.testing.assert_allclose(md, md2, rtol=1e-5)
    assert np.array_equal(out, pts[keep])  # order preserved


def test_sor_vs_numpy(medium_pair):
    src, _, _ = medium_pair
    v, _ = O.voxel_grid(O.distance_filter(src, 1.0), 0.1)
    out, keep, md = O.sor(v, 30, 1.2)
    out2, keep2, md2 = NR.sor(v, 30, 1.2)
    np.testing.assert_allclose(md, md2, rtol=2e-5)
    assert (keep != keep2).sum() <= 2  # points sitting on the threshold within fp32 rounding
    assert 0.5 * len(v) < len(out) < len(v)


def test_transform_and_fitness(small_pair):
    s, t, T = small_pair
    T32 = T.astype(np.float32)
    p = O.transform(s, T32)
    ref = s[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    np.testing.assert_allclose(p[:, :3], ref, atol=2e-5)
    np.testing.assert_array_equal(p[:, 3], s[:, 3])
    f = O.fitness(s, t, T32)
    np.testing.assert_allclose(f, NR.fitness(s, t, T32), rtol=1e-4)
    # max_range is compared with the SQUARED distance, as PCL does
    f_small = O.fitness(s, t, T32, max_range=0.01)
    assert f_small <= 0.01
    assert O.fitness(s, t, T32, max_range=-1.0) == np.finfo(np.float64).max
