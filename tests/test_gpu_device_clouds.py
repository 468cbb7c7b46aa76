"""Device-resident clouds (mi355_cloud): transform / concatenation on the device are bit-equal to the oracle's
pcl::transformPointCloud, a key frame's covariances follow it rigidly (C' = R C R^T), and a registration fed from device
clouds gives what the host-pointer entry points give."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def test_transform_and_concat_are_bit_equal_to_the_oracle(small_pair):
    from lidar_graph_slam_amd.registration import DeviceCloud

    s, t, _ = small_pair
    T1 = synth.se3(1.5, -0.3, 0.1, 0.01, -0.02, 0.3).astype(np.float32)
    T2 = synth.se3(-4.0, 2.0, 0.0, 0.0, 0.0, -1.2).astype(np.float32)
    ds, dt_ = DeviceCloud.upload(s), DeviceCloud.upload(t)
    assert len(ds) == len(s) and not ds.has_covariances
    np.testing.assert_array_equal(ds.download(), s)
    np.testing.assert_array_equal(ds.transformed(T1).download(), O.transform(s, T1))
    both = DeviceCloud.concat([ds, dt_, ds], [T1, T2, np.eye(4)])
    np.testing.assert_array_equal(both.download(), np.concatenate([O.transform(s, T1), O.transform(t, T2), s]))
    with pytest.raises(Exception):
        DeviceCloud.concat([ds, dt_], [T1, T2], carry_covariances=True)  # no covariances to carry yet


def test_covariances_follow_the_key_frame_rigidly(small_pair):
    from lidar_graph_slam_amd import filters as F
    from lidar_graph_slam_amd.registration import DeviceCloud, FastGICP

    s, t, T_gt = small_pair
    T = synth.se3(3.0, 1.0, -0.2, 0.02, 0.01, 0.7).astype(np.float32)
    ds = DeviceCloud.upload(s)
    ds.compute_covariances(20)
    assert ds.has_covariances
    moved = ds.transformed(T)
    assert moved.has_covariances
    # a registration that takes the moved cloud as its TARGET uses the carried covariances: same pose as the host path, whose
    # covariances are recomputed on the moved points (a rigid motion only rotates a neighbourhood covariance)
    src = O.transform(t, synth.se3(0.05, -0.02, 0.0, 0.0, 0.0, 0.004).astype(np.float32) @ T)
    a = FastGICP(max_iterations=10, force_iterations=1)
    a.setInputTarget(moved)
    a.setInputSource(DeviceCloud.upload(src))
    a.align()
    b = FastGICP(max_iterations=10, force_iterations=1)
    b.setInputTarget(O.transform(s, T))
    b.setInputSource(src)
    b.align()
    dt, dr = synth.pose_error(a.getFinalTransformation(), b.getFinalTransformation())
    assert dt <= 2e-4 and dr <= 2e-5, (dt, dr)
    # and numerically: R C R^T of the device covariances vs covariances computed on the transformed cloud
    c0 = F.covariances(s, 20).astype(np.float64)
    c1 = F.covariances(O.transform(s, T), 20).astype(np.float64)
    R = T[:3, :3].astype(np.float64)
    np.testing.assert_allclose(R @ c0 @ R.T, c1, atol=2e-5)


def test_source_covariances_are_computed_once_and_cached(small_pair):
    from lidar_graph_slam_amd.registration import DeviceCloud, FastGICP

    s, t, _ = small_pair
    ds, dt_ = DeviceCloud.upload(s), DeviceCloud.upload(t)
    reg = FastGICP(max_iterations=8, force_iterations=1)
    reg.setInputTarget(dt_)
    reg.setInputSource(ds)
    assert ds.has_covariances and not dt_.has_covariances  # the source keeps what the alignment needed anyway
    reg.align()
    T1 = reg.getFinalTransformation().copy()
    reg.setInputSource(ds)  # second time: the cached covariances are permuted in, not recomputed
    reg.align()
    np.testing.assert_array_equal(reg.getFinalTransformation(), T1)
    ref = O.gicp_align(s, t, O.gicp_params(max_iterations=8, force_iterations=1))
    dt, dr = synth.pose_error(T1, ref["T"])
    assert dt <= 1e-3 and dr <= 1e-4


def test_prefilter_of_a_device_cloud_is_bit_equal_to_the_oracle(medium_pair):
    """mi355_cloud_prefilter: the sub-map VoxelGrid of graph_based_slam.cpp:311-313 (and the whole points_prefiltering chain)
    without leaving the device."""
    from lidar_graph_slam_amd.registration import DeviceCloud

    src, _, _ = medium_pair
    d = DeviceCloud.upload(src)
    np.testing.assert_array_equal(d.prefiltered(leaf_size=0.5).download(), O.voxel_grid(src, 0.5)[0])
    np.testing.assert_array_equal(d.prefiltered(min_distance=1.0, leaf_size=0.2).download(), O.voxel_grid(O.distance_filter(src, 1.0), 0.2)[0])
    np.testing.assert_array_equal(d.prefiltered(min_distance=1.0).download(), O.distance_filter(src, 1.0))
    empty = DeviceCloud.upload(np.zeros((0, 4), np.float32)).prefiltered(leaf_size=0.5)
    assert len(empty) == 0
