"""Size-independent properties of the HIP path at BASELINE.json's full size (65 536-ray scans): the oracle is too slow to
run every full-size case, so these check what must hold whatever the input — idempotence and order of the voxel grid, the
self-neighbour and sortedness of k-NN lists, the eigen-structure of PLANE covariances, inverse consistency of the
registration, and independence of a batch's results from what else is in the batch."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_pair():
    src, tgt, T = synth.make_pair()  # 64 rings x 1024 azimuth
    assert len(src) == 65536
    return src, tgt, T


@pytest.fixture(scope="module")
def F():
    from lidar_graph_slam_amd import filters

    return filters


def _voxel(F, cloud, leaf):
    vg = F.VoxelGrid()
    vg.setLeafSize(leaf, leaf, leaf)
    vg.setInputCloud(cloud)
    return vg.filter()


def test_voxel_grid_idempotent_sorted_unique(F, full_pair):
    src = F.distance_filter(full_pair[0], 1.0)
    for leaf in (0.2, 0.5):
        v1 = _voxel(F, src, leaf)
        assert 0 < len(v1) < len(src)
        # one point per occupied voxel, in ascending PCL voxel index
        lo = src[:, :3].min(axis=0)
        inv = np.float32(1.0) / np.float32(leaf)
        mn = np.floor(lo * inv).astype(np.int64)
        mx = np.floor(src[:, :3].max(axis=0) * inv).astype(np.int64)
        div = mx - mn + 1
        ijk = np.floor(v1[:, :3] * inv).astype(np.int64) - mn
        idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
        assert np.all(np.diff(idx) > 0)
        # every centroid lies inside the bounding box of the points
        assert np.all(v1[:, :3] >= lo - 1e-4) and np.all(v1[:, :3] <= src[:, :3].max(axis=0) + 1e-4)
        # a second pass moves nothing: each voxel now holds exactly its own centroid
        v2 = _voxel(F, v1, leaf)
        assert len(v2) == len(v1)
        np.testing.assert_allclose(v2, v1, atol=2e-6 * np.abs(v1).max())


def test_knn_lists_sorted_and_self_first(F, full_pair):
    cloud = _voxel(F, F.distance_filter(full_pair[0], 1.0), 0.2)
    idx, d2 = F.knn(cloud, cloud, 20)
    assert idx.shape == (len(cloud), 20)
    assert np.all(d2[:, 0] == 0.0) and np.all(idx[:, 0] == np.arange(len(cloud)))  # a point is its own nearest neighbour
    assert np.all(np.diff(d2, axis=1) >= 0)
    assert np.all((idx >= 0) & (idx < len(cloud)))
    assert all(len(set(row)) == 20 for row in idx[:: max(1, len(idx) // 500)])  # no neighbour twice
    # the reported distances are the distances
    j = idx[:, 7]
    d = ((cloud[:, :3].astype(np.float64) - cloud[j, :3].astype(np.float64)) ** 2).sum(axis=1)
    np.testing.assert_allclose(d2[:, 7], d, rtol=2e-6, atol=1e-9)
    # k-NN is consistent under k: the first 8 of the 20 are the 8-NN
    idx8, d8 = F.knn(cloud, cloud, 8)
    np.testing.assert_array_equal(d8, d2[:, :8])


def test_plane_covariances_have_the_plane_spectrum(F, full_pair):
    cloud = _voxel(F, F.distance_filter(full_pair[0], 1.0), 0.2)
    c = F.covariances(cloud, 20, 3).astype(np.float64)  # PLANE
    assert np.abs(c - c.transpose(0, 2, 1)).max() == 0.0
    w = np.linalg.eigvalsh(c)
    np.testing.assert_allclose(w, np.tile([1e-3, 1.0, 1.0], (len(c), 1)), rtol=2e-5, atol=2e-7)


def test_registration_inverse_consistency_and_batch_independence(full_pair):
    from lidar_graph_slam_amd import _lib as L
    from lidar_graph_slam_amd.registration import BatchRegistration

    src, tgt, T_gt = full_pair
    cap = 65536
    pre = L.prefilter_params(min_distance=1.0, leaf_size=0.2, mean_k=0)
    kw = dict(max_iterations=20, force_iterations=1)
    counts = np.array([cap, cap], np.int32)
    both = np.stack([src, tgt])  # pair 0: src -> tgt, pair 1: tgt -> src
    b = BatchRegistration(2, 2, cap, **kw)
    b.set_targets(np.stack([tgt, src]), counts, pre)
    b.set_sources(both, counts, pre)
    b.align()
    r = b.results()
    T01, T10 = r[0]["T"].astype(np.float64), r[1]["T"].astype(np.float64)
    dt, dr = synth.pose_error(T01 @ T10, np.eye(4))
    assert dt < 0.02 and dr < 1e-3, (dt, dr)  # forward and backward registration are inverse to each other (GICP is not symmetric: cm level)
    dt, dr = synth.pose_error(T01, T_gt)
    assert dt < 0.02 and dr < 1e-3, (dt, dr)
    # the same pair inside a different batch (other neighbours, other slot) gives bit-identical results
    b3 = BatchRegistration(3, 3, cap, **kw)
    c3 = np.array([cap, cap, cap], np.int32)
    b3.set_targets(np.stack([src, src, tgt]), c3, pre)
    b3.set_sources(np.stack([tgt, tgt[::-1].copy(), src]), c3, pre)
    b3.align()
    r3 = b3.results()
    np.testing.assert_array_equal(r3[2]["T"], r[0]["T"])
    np.testing.assert_array_equal(r3[0]["T"], r[1]["T"])
    # ... and the caller's point order does not matter beyond rounding (tgt reversed in slot 1)
    dt, dr = synth.pose_error(r3[1]["T"], r3[0]["T"])
    assert dt < 2e-4 and dr < 2e-5, (dt, dr)
