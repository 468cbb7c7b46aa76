"""GPU parity of the per-scan odometry loop (BASELINE config 4 shape): ScanMatcherCore (C++ policy + HIP registration)
vs an independent restatement of LidarScanMatcher::callback_cloud that registers with the oracle."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth
from oracle import oracle_py as O
from ref_scan_matcher import RefScanMatcher
import parity

pytestmark = pytest.mark.gpu


def _stream(n_scans, n_azimuth=128):
    scene = synth.Scene(1234)
    poses = synth.trajectory(n_scans, seed=3)
    scans = []
    for i in range(n_scans):
        raw = synth.cast_scan(scene, poses[i], n_azimuth=n_azimuth, seed=100 + i)
        scans.append(O.voxel_grid(O.distance_filter(raw, 1.0), 0.2)[0])  # what points_prefiltering would publish (SOR off here)
    return poses, scans


@pytest.mark.parametrize("method,reuse", [("FAST_GICP", False), ("FAST_GICP", True), ("NDT_OMP", False)])
def test_stream_matches_reference_policy(method, reuse):
    """reuse = False (the default): target covariances recomputed over the merged sub-map as fast_gicp does — held to the
    oracle's pose scan by scan; True (opt-in, not the reference's arithmetic): each key frame's covariances rotated into the
    sub-map — held to the policy decisions and the ground truth only."""
    from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher

    poses, scans = _stream(10)
    sm = LidarScanMatcher(registration_method=method, max_scan_accumulate_num=4, reuse_key_frame_covariances=reuse)
    ref = RefScanMatcher(method=method, max_scan_accumulate_num=4)
    n_kf = 0
    for i, s in enumerate(scans):
        a = sm.callback_cloud(s)
        b = ref.callback_cloud(s)
        assert a["first_scan"] == b["first_scan"], i
        if a["first_scan"]:
            continue
        assert a["converged"] and b["converged"], (i, a["converged"], b["converged"])
        # both stop inside the convergence band of each scan (measured as is_converged measures it).
        # With rotated per-key-frame covariances (reuse) a target point's neighbourhood holds its own scan's points only; on
        # these sparse 8 k-ray scans that moves the optimum by centimetres and the two streams then see different key frames -
        # that opt-in mode is held to the ground truth below instead (and, at full size, in test_gpu_fullsize.py)
        if not (method == "FAST_GICP" and reuse):
            parity.assert_in_band(a["pose"], b["pose"], f"scan_matcher_small/{method}/scan{i}")
        # the poses drive the key-frame gate; a scan sitting on the 1.0 m threshold could flip it, the stream avoids that
        assert a["new_key_frame"] == b["new_key_frame"], i
        n_kf += a["new_key_frame"]
        assert a["target_points"] == len(ref.target)
        gt = np.linalg.inv(poses[0]) @ poses[i]
        dt, dr = synth.pose_error(a["pose"], gt)
        if method == "FAST_GICP":  # NDT with 2 m cells on 8k-ray scans drifts (identically in the oracle): parity only
            assert dt < 0.15 and dr < 0.01, (i, dt, dr)  # odometry stays on the ground truth
    assert n_kf >= 3
    tgt = sm.target_cloud()
    assert tgt.shape == ref.target.shape  # the device-assembled sub-map is the reference's up to the key-frame poses (band)
    np.testing.assert_allclose(tgt[:, :3], ref.target[:, :3], rtol=0, atol=0.05)
    assert sm.callback_cloud(scans[-1])["n_key_frames"] == len(ref.key_frames)


def test_unknown_method_fails_at_construction():
    from lidar_graph_slam_amd import _lib as L
    from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher

    with pytest.raises(L.Mi355Error):
        LidarScanMatcher(registration_method="GICP_OMP")
