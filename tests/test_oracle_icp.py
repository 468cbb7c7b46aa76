"""Pins the oracle's pcl::IterativeClosestPoint restatement (PARITY UNPINNED upstream: PCL is absent): the rigid fit
against Kabsch by SVD, the loop against an independent NumPy restatement.  (Point-to-point ICP does not recover the ground truth on
these ring-patterned scans — from a 5 cm guess it walks 18 cm AWAY: the scan lines of two poses attract each other — which
is why the front end registers with GICP / NDT; the back end's default is restated for interface completeness.)"""
import numpy as np

from lidar_graph_slam_amd import synth
from oracle import numpy_ref as N
from oracle import oracle_py as O


def test_rigid_fit_is_kabsch():
    rng = np.random.default_rng(0)
    for _ in range(20):
        P = rng.normal(size=(40, 3)) * rng.uniform(0.1, 30, 3)
        T = synth.se3(*rng.uniform(-2, 2, 3), *rng.uniform(-0.6, 0.6, 3))
        Q = P @ T[:3, :3].T + T[:3, 3] + rng.normal(size=P.shape) * 0.02
        s = np.zeros(17)
        s[0], s[1:4], s[4:7] = len(P), P.sum(0), Q.sum(0)
        s[7:16] = (P[:, :, None] * Q[:, None, :]).sum(0).reshape(9)
        Tr, rc = O.rigid_from_sums(s)
        assert rc == 0
        mp, mq = P.mean(0), Q.mean(0)
        U, _, Vt = np.linalg.svd((P - mp).T @ (Q - mq))
        R = Vt.T @ np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))]) @ U.T
        np.testing.assert_allclose(Tr[:3, :3], R, atol=1e-12)
        np.testing.assert_allclose(Tr[:3, 3], mq - R @ mp, atol=1e-12)
    assert O.rigid_from_sums(np.r_[2.0, np.zeros(16)])[1] == 1  # fewer than 3 correspondences


def test_loop_vs_independent_numpy_restatement(small_pair):
    s, t, T_gt = small_pair
    for guess in (np.eye(4, dtype=np.float32), (T_gt @ synth.se3(0.05, -0.03, 0.0, 0, 0, 0.004)).astype(np.float32)):
        for k in (5, 10, 15):  # iteration by iteration the two restatements walk the same path
            r = O.icp_align(s, t, O.icp_params(max_iterations=k), guess=guess)
            T, conv, it = N.icp_align(s, t, guess=guess, max_iterations=k)
            assert r["converged"] and conv and r["iterations"] == it
            dt, dr = synth.pose_error(r["T"], T)
            assert dt < 5e-6 and dr < 5e-7, (k, dt, dr)
        # with PCL's stopping rules (relative MSE change 1e-6, incremental step 1e-4 m) the iteration at which the loop stops
        # hinges on the last bits of the distances; both stop near the same pose
        r = O.icp_align(s, t, O.icp_params(), guess=guess)
        T, conv, it = N.icp_align(s, t, guess=guess)
        assert r["converged"] and conv
        dt, dr = synth.pose_error(r["T"], T)
        assert dt < 0.02 and dr < 5e-4, (dt, dr)


def test_criteria(small_pair):
    s, t, _ = small_pair
    r = O.icp_align(s, t, O.icp_params(max_iterations=3))
    assert r["converged"] and r["iterations"] == 3  # reaching max_iterations counts as converged (DefaultConvergenceCriteria)
    far = s.copy()
    far[:, :3] += 1000.0
    r = O.icp_align(far, t, O.icp_params())
    assert not r["converged"] and r["iterations"] == 0  # no correspondence within 30 m: "not enough correspondences"
