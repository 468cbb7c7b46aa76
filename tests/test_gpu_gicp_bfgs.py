"""GPU parity of registration_method "GICP" (pclomp::GeneralizedIterativeClosestPoint, lidar_scan_matcher.cpp:73-96,
graph_based_slam.cpp:120-141): the HIP correspondence / cost kernels + the host BFGS against the oracle's restatement."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from lidar_graph_slam_amd import registration

    return registration


def test_one_outer_iteration_matches_the_oracle(R, medium_pair):
    """Same exact correspondences, same cost: after ONE outer iteration (a full BFGS minimisation) the poses agree far inside the
    convergence band — what separates them is fp32 vs fp64 covariances in the Mahalanobis matrices and the summation order."""
    s, t, _ = medium_pair
    reg = R.GeneralizedIterativeClosestPoint(max_iterations=1)
    reg.setInputTarget(t)
    reg.setInputSource(s)
    reg.align()
    ref = O.gicp_bfgs_align(s, t, O.gicp_bfgs_params(max_iterations=1))
    assert reg.hasConverged() and ref["converged"] and reg._result["iterations"] == ref["iterations"] == 1
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
    assert dt <= 2e-3 and dr <= 2e-4, (dt, dr)
    assert abs(reg._result["error"] - ref["error"]) <= 1e-3 * ref["error"]


def test_align_from_identity_and_with_a_rotation_guess(R, medium_pair):
    s, t, T = medium_pair
    reg = R.GeneralizedIterativeClosestPoint()
    reg.setMaxCorrespondenceDistance(2.0)
    reg.setMaximumIterations(64)
    reg.setMaximumOptimizerIterations(20)
    reg.setTransformationEpsilon(0.01)
    reg.setEuclideanFitnessEpsilon(1e-8)
    reg.setRANSACIterations(0)
    reg.setInputTarget(t)
    reg.setInputSource(s)
    reg.align()
    ref = O.gicp_bfgs_align(s, t)
    assert reg.hasConverged() and ref["converged"]
    assert abs(reg._result["iterations"] - ref["iterations"]) <= 1
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
    assert dt <= 0.01 and dr <= 2e-3, (dt, dr)  # both stop inside the convergence band
    dt, dr = synth.pose_error(reg.getFinalTransformation(), T)
    assert dt < 0.02 and dr < 1e-3, (dt, dr)
    np.testing.assert_allclose(reg.getFitnessScore(), O.fitness(s, t, reg.getFinalTransformation()), rtol=1e-9)
    g = synth.se3(0, 0, 0, 0, 0, 0.01).astype(np.float32)
    reg.align(g)
    refg = O.gicp_bfgs_align(s, t, guess=g)
    dt, dr = synth.pose_error(reg.getFinalTransformation(), refg["T"])
    assert reg.hasConverged() and dt <= 0.02 and dr <= 2e-3, (dt, dr)  # two stopping points, each within a band of the minimum


def test_nothing_in_range_does_not_converge(R, small_pair):
    s, t, _ = small_pair
    far = O.transform(s, synth.se3(500.0, 0, 0, 0, 0, 0).astype(np.float32))
    reg = R.GeneralizedIterativeClosestPoint()
    reg.setInputTarget(t)
    reg.setInputSource(far)
    reg.align()
    assert not reg.hasConverged() and reg._result["iterations"] == 0
    np.testing.assert_array_equal(reg.getFinalTransformation(), np.eye(4, dtype=np.float32))


def test_only_the_built_options_are_accepted(R):
    from lidar_graph_slam_amd import _lib as L

    with pytest.raises(L.Mi355Error):
        R.GeneralizedIterativeClosestPoint(ransac_iterations=5)


def test_scan_matcher_accepts_gicp(R):
    """lidar_scan_matcher.cpp:73-96: "GICP" builds pclomp's class with the yaml's GICP block (5 m correspondence distance) and the
    second scan is registered against the first from the identity.  Policy: the node's pose IS that registration's result.
    Engine: one outer iteration with those parameters matches the oracle to millimetres.  (Over several outer iterations two
    correct executions of this engine may part ways by centimetres: its cost moves the points in fp32, Fletcher's line search then
    works on a function with 1e-7 relative noise and ends with "no progress" wherever the noise first hides the descent — the
    oracle and the GPU, which sum in different orders, stop in different places.  test_align_* holds them to the band at 2 m.)"""
    from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher

    scene = synth.street(40.0)
    poses = synth.trajectory(3, seed=3)
    scans = [O.voxel_grid(O.distance_filter(synth.cast_scan(scene, P, n_azimuth=256, seed=50 + i, ground_max_range=20.0), 1.0), 0.2)[0]
             for i, P in enumerate(poses)]
    sm = LidarScanMatcher(registration_method="GICP")
    assert sm.callback_cloud(scans[0])["first_scan"]
    out = sm.callback_cloud(scans[1])
    reg = R.GeneralizedIterativeClosestPoint(max_correspondence_distance=5.0, max_iterations=64, max_optimizer_iterations=20,
                                             transformation_epsilon=0.01, euclidean_fitness_epsilon=1e-8, k_correspondences=20)
    reg.setInputTarget(scans[0])
    reg.setInputSource(scans[1])
    reg.align()
    assert out["converged"] == reg.hasConverged() and out["iterations"] == reg._result["iterations"]
    np.testing.assert_array_equal(out["pose"], reg.getFinalTransformation())
    one = R.GeneralizedIterativeClosestPoint(max_correspondence_distance=5.0, max_iterations=1)
    one.setInputTarget(scans[0])
    one.setInputSource(scans[1])
    one.align()
    ref = O.gicp_bfgs_align(scans[1], scans[0], O.gicp_bfgs_params(max_correspondence_distance=5.0, max_iterations=1))
    dt, dr = synth.pose_error(one.getFinalTransformation(), ref["T"])
    assert dt <= 5e-3 and dr <= 5e-4, (dt, dr)
    out = sm.callback_cloud(scans[2])  # from the previous pose: whatever upstream's handling of a translated guess gives, it is finite
    assert np.isfinite(out["pose"]).all()
