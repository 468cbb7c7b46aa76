"""CPU tests of the PRODUCT's host/device maths header (csrc/devmath.hpp compiled with g++): se3_exp, pivoted LDL^T, the
symmetric 6x6 SVD solve, covariance regularisation, fp32 3x3 helpers — against SciPy / NumPy / the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from scipy.linalg import expm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hm") / "libhost_math.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           "-o", out, os.path.join(ROOT, "tests", "host_math_harness.cpp")])
    lib = C.CDLL(out)
    lib.hm_dist2f.restype = C.c_float
    lib.hm_is_converged.argtypes = [C.c_void_p, C.c_double, C.c_double]
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_se3_exp(hm):
    rng = np.random.default_rng(0)
    for scale in (1e-8, 1e-4, 0.2, 2.0):
        d = rng.normal(size=6) * scale
        T = np.empty(16)
        hm.hm_se3_exp(_p(d), _p(T))
        A = np.zeros((4, 4))
        A[:3, :3] = [[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]]
        A[:3, 3] = d[3:]
        np.testing.assert_allclose(T.reshape(4, 4), expm(A), atol=1e-12)


def test_ldlt6_and_svd6(hm):
    rng = np.random.default_rng(1)
    for _ in range(10):
        J = rng.normal(size=(40, 6)) * rng.uniform(0.01, 100, 6)
        H = np.ascontiguousarray(J.T @ J)
        b = rng.normal(size=6)
        x = np.empty(6)
        assert hm.hm_ldlt6(_p(H), _p(b), _p(x)) == 1
        np.testing.assert_allclose(x, np.linalg.solve(H, b), rtol=1e-8)
        hm.hm_svd6(_p(H), _p(b), _p(x))
        np.testing.assert_allclose(x, np.linalg.solve(H, b), rtol=1e-7)
    Z = np.zeros((6, 6))
    x = np.full(6, 7.0)
    assert hm.hm_ldlt6(_p(Z), _p(np.ones(6)), _p(x)) == 0 and not x.any()  # singular -> zero step, like Eigen's LDLT
    # indefinite symmetric (NDT Hessians are): the SVD solve is the plain inverse
    S = rng.normal(size=(6, 6))
    S = np.ascontiguousarray(S + S.T)
    b = rng.normal(size=6)
    hm.hm_svd6(_p(S), _p(b), _p(x))
    np.testing.assert_allclose(x, np.linalg.solve(S, b), rtol=1e-8)


def test_icp_rigid_fit_is_the_kabsch_fit(hm):
    """icp_fit.hpp icp_rigid_from_sums (the fit k_icp_advance runs on the device for every pair of a batch): the least-squares rigid
    transform of the matched pairs, from their 17 sums, equals the SVD (Kabsch / Umeyama) solution."""
    rng = np.random.default_rng(21)
    for trial in range(20):
        n = int(rng.integers(3, 400))
        P = rng.normal(size=(n, 3)) * rng.uniform(0.5, 30.0, 3) + rng.uniform(-50, 50, 3)
        ang = rng.uniform(-0.3, 0.3, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
        Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
        Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
        Rt, tt = Rz @ Ry @ Rx, rng.uniform(-2, 2, 3)
        Q = P @ Rt.T + tt + rng.normal(0, 0.02, (n, 3))
        sums = np.concatenate([[n], P.sum(0), Q.sum(0), (P[:, :, None] * Q[:, None, :]).sum(0).ravel(), [((Q - P) ** 2).sum()]])
        T = np.empty(16)
        assert hm.hm_icp_rigid(_p(np.ascontiguousarray(sums)), _p(T)) == 1
        T = T.reshape(4, 4)
        mp, mq = P.mean(0), Q.mean(0)
        U, _, Vt = np.linalg.svd((Q - mq).T @ (P - mp))
        D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
        Rk = U @ D @ Vt
        np.testing.assert_allclose(T[:3, :3], Rk, atol=1e-9)
        np.testing.assert_allclose(T[:3, 3], mq - Rk @ mp, atol=1e-8)
    few = np.zeros(17)
    few[0] = 2
    assert hm.hm_icp_rigid(_p(few), _p(np.empty(16))) == 0  # "Not enough correspondences found"


def test_ndt_newton_solve_is_the_svd_solve(hm):
    """ndt_mt.hpp ndt_newton_solve: the pivoted-elimination shortcut (well-conditioned Hessians, definite or not) and the Jacobi
    fallback both give JacobiSVD(H).solve(rhs) — the rank-truncating solve for a singular H included."""
    rng = np.random.default_rng(11)
    x, y = np.empty(6), np.empty(6)
    for sign in (1.0, -1.0):
        for _ in range(20):
            A = rng.normal(size=(6, 6))
            H = np.ascontiguousarray(sign * (A @ A.T + 0.1 * np.eye(6)) * rng.uniform(1e-2, 1e4))
            b = rng.normal(size=6)
            hm.hm_ndt_newton(_p(H), _p(b), _p(x))
            hm.hm_svd6(_p(H), _p(b), _p(y))
            np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-12 * np.abs(y).max())
            np.testing.assert_allclose(x, np.linalg.solve(H, b), rtol=1e-8)
    for _ in range(20):  # indefinite (NDT Hessians are, away from the optimum)
        S = rng.normal(size=(6, 6))
        S = np.ascontiguousarray(S + S.T)
        b = rng.normal(size=6)
        hm.hm_ndt_newton(_p(S), _p(b), _p(x))
        hm.hm_svd6(_p(S), _p(b), _p(y))
        np.testing.assert_allclose(x, y, rtol=1e-8, atol=1e-11 * np.abs(y).max())
    # near-singular and singular Hessians take the Jacobi path: the same numbers as svd6_solve_sym, bit for bit
    Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
    R5 = np.ascontiguousarray(Q @ np.diag([3.0, 2.0, 1.0, 0.5, 0.1, 0.0]) @ Q.T)
    R5 = np.ascontiguousarray(0.5 * (R5 + R5.T))
    ill = np.ascontiguousarray(Q @ np.diag([1.0, 1.0, 1.0, 1.0, 1.0, 1e-12]) @ Q.T)
    ill = np.ascontiguousarray(0.5 * (ill + ill.T))
    for H in (R5, ill, np.zeros((6, 6))):
        b = rng.normal(size=6)
        hm.hm_ndt_newton(_p(H), _p(b), _p(x))
        hm.hm_svd6(_p(H), _p(b), _p(y))
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("method,name", [(0, "NONE"), (1, "MIN_EIG"), (2, "NORMALIZED_MIN_EIG"), (3, "PLANE"), (4, "FROBENIUS")])
def test_regularisation_vs_numpy(hm, method, name):
    rng = np.random.default_rng(2)
    for _ in range(20):
        P = rng.normal(size=(20, 3)) * rng.uniform(0.01, 1.0, 3)
        c = np.cov(P.T, bias=True)
        c6 = np.array([c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]])
        out = np.empty(6)
        hm.hm_regularize(_p(c6), method, _p(out))
        got = np.array([[out[0], out[1], out[2]], [out[1], out[3], out[4]], [out[2], out[4], out[5]]])
        if name == "NONE":
            ref = c
        elif name == "FROBENIUS":
            Ci = np.linalg.inv(c + 1e-3 * np.eye(3))
            ref = np.linalg.inv(Ci / np.linalg.norm(Ci))
        else:
            U, s, Vt = np.linalg.svd(c)
            vals = {"PLANE": np.array([1, 1, 1e-3]), "MIN_EIG": np.maximum(s, 1e-3), "NORMALIZED_MIN_EIG": np.maximum(s / s.max(), 1e-3)}[name]
            ref = U @ np.diag(vals) @ Vt
        np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-11)


def _cov6(c):
    return np.array([c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]])


def test_plane_shortcut_matches_jacobi(hm):
    """PLANE = I - 0.999 n n^T with n from smallest_eigenvector_sym3 (Newton on the separated root + cross products + one
    2x2 rotation) against the full Jacobi path, over the neighbourhood shapes a LiDAR cloud has: planes, lines (one ring),
    blobs, tiny and huge scales, rank-deficient inputs.  Degenerate inputs must defer (return 0) so they resolve as before."""
    rng = np.random.default_rng(7)
    n_fast = 0
    worst = 0.0
    shapes = [(1.0, 1.0, 0.01), (1.0, 0.02, 0.01), (1.0, 0.5, 0.2), (1.0, 1.0, 1e-4), (1.0, 1e-3, 5e-4), (1.0, 0.3, 0.29), (1.0, 0.99, 0.5),
              (1.0, 0.05, 0.049)]
    for sx, sy, sz in shapes:
        for scale in (1e-3, 1.0, 50.0):
            for _ in range(40):
                Q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
                P = (rng.normal(size=(20, 3)) * np.array([sx, sy, sz]) * scale) @ Q.T + rng.uniform(-80, 80, 3)
                c6 = _cov6(np.cov(P.T, bias=True))
                n = np.empty(3)
                ok = hm.hm_smallest_eigvec(_p(c6), _p(n))
                ref = np.empty(6)
                hm.hm_plane_jacobi(_p(c6), _p(ref))
                got = np.empty(6)
                hm.hm_regularize(_p(c6), 3, _p(got))
                worst = max(worst, float(np.abs(got - ref).max()))
                np.testing.assert_allclose(got, ref, atol=2e-9)
                if ok:
                    n_fast += 1
                    assert abs(np.linalg.norm(n) - 1) < 1e-12
                    w, V = np.linalg.eigh(np.cov(P.T, bias=True))
                    assert abs(abs(n @ V[:, 0]) - 1) < 1e-9 / max((w[1] - w[0]) / w[2], 1e-6) * 1e-3 + 1e-12
    assert n_fast > 0.95 * len(shapes) * 3 * 40, n_fast
    # degenerate inputs defer to the Jacobi path: zero matrix, isotropic, exactly collinear (rank 1), duplicated point
    for c6 in (np.zeros(6), _cov6(np.eye(3)), _cov6(np.outer([1, 2, 3], [1, 2, 3]).astype(float)), _cov6(1e-30 * np.eye(3))):
        assert hm.hm_smallest_eigvec(_p(c6), _p(np.empty(3))) == 0
        ref, got = np.empty(6), np.empty(6)
        hm.hm_plane_jacobi(_p(c6), _p(ref))
        hm.hm_regularize(_p(c6), 3, _p(got))
        np.testing.assert_array_equal(got, ref)
    # exactly planar (rank 2, smallest eigenvalue 0): the shortcut applies and the normal is exact
    P = np.c_[rng.normal(size=(20, 2)), np.zeros(20)]
    n = np.empty(3)
    assert hm.hm_smallest_eigvec(_p(_cov6(np.cov(P.T, bias=True))), _p(n)) == 1
    np.testing.assert_allclose(np.abs(n), [0, 0, 1], atol=1e-15)


def test_fp32_helpers_and_specified_rounding(hm):
    rng = np.random.default_rng(3)
    a = rng.normal(size=(3, 3)).astype(np.float32)
    S = a @ a.T + np.eye(3, dtype=np.float32)
    s6 = np.array([S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]], dtype=np.float32)
    out = np.empty(6, dtype=np.float32)
    hm.hm_sym3_inverse_f(_p(s6), _p(out))
    inv = np.linalg.inv(S.astype(np.float64))
    np.testing.assert_allclose(out, [inv[0, 0], inv[0, 1], inv[0, 2], inv[1, 1], inv[1, 2], inv[2, 2]], rtol=2e-5)
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
    hm.hm_sym3_rotate_f(_p(np.ascontiguousarray(R.reshape(9))), _p(s6), _p(out))
    ref = R.astype(np.float64) @ S.astype(np.float64) @ R.astype(np.float64).T
    np.testing.assert_allclose(out, [ref[0, 0], ref[0, 1], ref[0, 2], ref[1, 1], ref[1, 2], ref[2, 2]], rtol=2e-5, atol=2e-6)
    # dist2f is the specified fmaf(dz,dz, fmaf(dy,dy, dx*dx)): bit-equal to the oracle's k-NN distances
    from oracle import oracle_py as O
    pts = rng.uniform(-50, 50, (64, 4)).astype(np.float32)
    q = rng.uniform(-50, 50, (1, 4)).astype(np.float32)
    idx, d2 = O.knn(pts, q, 64, brute_force=True)
    mine = np.array([hm.hm_dist2f(_p(np.ascontiguousarray(q[0])), _p(np.ascontiguousarray(pts[j]))) for j in idx[0]], dtype=np.float32)
    np.testing.assert_array_equal(mine, d2[0])


def test_is_converged(hm):
    T = np.eye(4)
    T[0, 3] = 0.009
    assert hm.hm_is_converged(_p(np.ascontiguousarray(T.reshape(16))), 2e-3, 0.01) == 1
    T[0, 3] = 0.011
    assert hm.hm_is_converged(_p(np.ascontiguousarray(T.reshape(16))), 2e-3, 0.01) == 0
    T[0, 3] = 0
    T[0, 1], T[1, 0] = -0.0025, 0.0025
    assert hm.hm_is_converged(_p(np.ascontiguousarray(T.reshape(16))), 2e-3, 0.01) == 0


@pytest.mark.parametrize("cost_id", [0, 1, 2])
def test_bfgs_machine_is_pcls_bfgs_class(hm, cost_id):
    """csrc/bfgs_dev.hpp - PCL's BFGS (GSL vector_bfgs2) as a resumable state machine that RETURNS for every cost evaluation, what the
    device advances for all pairs of a "GICP" batch - against the same algorithm in upstream's functor-calling shape
    (tests/bfgs_host_ref.hpp, the product's host code until round 4) on three analytic costs (an anisotropic quadratic, a Rosenbrock
    chain, quadratic + cosines), from random starts and under every inner-iteration budget: the accepted points, the cost, the number
    of inner iterations and the status are BIT-EQUAL (a value is a pure function of the step evaluated); the machine evaluates less -
    one (f, gradient) evaluation per distinct step, where the class calls f() and df() apart."""
    hm.hm_bfgs_minimize.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(100 + cost_id)
    fewer = 0
    for trial in range(12):
        x = rng.normal(0.0, 1.0 if trial < 8 else 4.0, 6)
        for budget in (1, 2, 5, 20, 300):
            a, b = np.zeros(10), np.zeros(10)
            hm.hm_bfgs_minimize(0, cost_id, _p(x), budget, _p(a))
            hm.hm_bfgs_minimize(1, cost_id, _p(x), budget, _p(b))
            np.testing.assert_array_equal(a[:9], b[:9], err_msg=f"trial {trial}, budget {budget}")
            assert b[9] <= a[9]
            fewer += int(b[9] < a[9])
            assert a[8] in (0.0, 3.0, -2.0) and (a[7] == budget or a[8] != -2.0)  # Running only when the budget ran out
    assert fewer > 30
    # ... and it minimises: the quadratic's minimum is known in closed form
    if cost_id == 0:
        H = np.diag(1.0 + 3.0 * np.arange(6))
        H[0, 5] = H[5, 0] = 0.2
        c = (1.0 + 3.0 * np.arange(6)) * 0.3 * (np.arange(6) + 1)
        out = np.zeros(10)
        hm.hm_bfgs_minimize(1, 0, _p(np.zeros(6)), 300, _p(out))
        assert np.abs(H @ out[:6] - c).max() < 1e-2  # testGradient(1e-2) is the stopping rule


def test_bfgs_pose_helpers(hm):
    """applyState (fp32 quaternion product of the three axis rotations onto a 4x4), the start vector of a matrix and the value /
    gradient from the 14 sums of a cost pass (bfgs_dev.hpp) against NumPy in fp64."""
    rng = np.random.default_rng(7)
    for _ in range(50):
        x = np.concatenate([rng.normal(0, 2.0, 3), rng.uniform(-1.2, 1.2, 3)])
        t = np.eye(4, dtype=np.float32)
        t[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        t[:3, 3] = rng.normal(0, 5.0, 3).astype(np.float32)
        t0 = t.copy()
        hm.hm_bfgs_apply_state(_p(t), _p(x))
        cr, sr, cp, sp, cy, sy = np.cos(x[3]), np.sin(x[3]), np.cos(x[4]), np.sin(x[4]), np.cos(x[5]), np.sin(x[5])
        Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
        want = t0.astype(np.float64)
        want[:3, :3] = Rz @ Ry @ Rx @ want[:3, :3]
        want[:3, 3] += x[:3]
        np.testing.assert_allclose(t, want, atol=5e-6)
        # the parameters of a matrix built by applyState from the identity are the parameters it was built from
        e = np.eye(4, dtype=np.float32)
        hm.hm_bfgs_apply_state(_p(e), _p(x))
        back = np.zeros(6)
        hm.hm_bfgs_x_from_matrix(_p(e), _p(back))
        np.testing.assert_allclose(back, x, atol=2e-6)
        # gradient of the rotation part: g[3 + a] = sum_ij dR/d(angle a)_ij * Racc_ji, against central differences of R
        h = rng.normal(size=14)
        h[0] = 50.0
        f = np.zeros(1)
        g = np.zeros(6)
        hm.hm_bfgs_cost_from_sums(_p(h), _p(x), _p(f), _p(g))
        assert f[0] == h[1] / h[0]
        np.testing.assert_allclose(g[:3], h[2:5] * 2.0 / h[0], rtol=1e-15)
        Racc = (h[5:14] * 2.0 / h[0]).reshape(3, 3)

        def rot(a):
            c, s = np.cos(a), np.sin(a)
            return (np.array([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]]) @ np.array([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
                    @ np.array([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]]))
        for a in range(3):
            d = np.zeros(3)
            d[a] = 1e-6
            dR = (rot(x[3:] + d) - rot(x[3:] - d)) / 2e-6
            np.testing.assert_allclose(g[3 + a], np.sum(dR * Racc.T), atol=1e-7 * (1 + np.abs(Racc).sum()))
