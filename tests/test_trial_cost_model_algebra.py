"""The algebra behind k_gicp_step (gicp.hip), restated in NumPy and checked against the definitions it replaces.

With the correspondences and Mahalanobis matrices M of an LM iteration frozen (FastGICP::compute_error), the cost at a trial
pose x_t = delta * x0 is a quadratic form of W = [dR - I | dt]:  d_t = d0 - W a~,  a~ = (x0 p, 1),  d0 = q - x0 p, so
    e_t = y0 - 2 <W, L> + sum W_rc W_se Q[rs][ce],   L = sum (M d0) a~^T,   Q = sum M (x) a~ a~^T,
and the normal equations of FastGICP::linearize, J = [skew(a) | -I], are linear in the same sums.  The device code accumulates
L and Q in the linearisation pass and reads H, b and every trial cost off them; the GPU parity test
(tests/test_gpu_gicp.py::test_trial_cost_model_is_compute_error) holds that code to the pass over the points and to the oracle.
This file pins the formulas themselves (index conventions included) on random data, without a GPU."""
import numpy as np


def _skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def _rot(w):
    th = np.linalg.norm(w)
    K = _skew(w / th)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _sums(a, d0, M):
    """L (3 x 4) and Q (3, 3, 4, 4) as the kernel defines them."""
    at = np.concatenate([a, np.ones((len(a), 1))], axis=1)
    Md0 = np.einsum("nij,nj->ni", M, d0)
    L = np.einsum("nr,nc->rc", Md0, at)
    Q = np.einsum("nrs,nc,ne->rsce", M, at, at)
    return L, Q


def _eps(k, i, p):
    return ((k - i) * (i - p) * (p - k)) // 2


def _normal_equations_from_sums(L, Q):
    """model_normal_entry of gicp.hip: rows / columns 0-2 rotation, 3-5 translation."""
    H = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        for j in range(3):
            H[3 + i, 3 + j] = Q[i, j, 3, 3]
            H[i, j] = sum(_eps(k, i, p) * _eps(l, j, q) * Q[k, l, p, q] for k in range(3) for l in range(3) for p in range(3) for q in range(3))
            H[i, 3 + j] = H[3 + j, i] = sum(_eps(k, i, p) * Q[k, j, p, 3] for k in range(3) for p in range(3))
        b[i] = -sum(_eps(k, i, p) * L[k, p] for k in range(3) for p in range(3))
        b[3 + i] = -L[i, 3]
    return H, b


def _data(seed, n=400, scale=60.0):
    g = np.random.default_rng(seed)
    a = g.uniform(-scale, scale, (n, 3))          # x0 p
    d0 = g.normal(0, 0.05, (n, 3))                # q - x0 p
    A = g.normal(0, 1, (n, 3, 3))
    M = np.einsum("nij,nkj->nik", A, A) * 50 + np.eye(3) * 100  # SPD "Mahalanobis" matrices
    return a, d0, M


def test_trial_cost_is_the_quadratic_form_of_the_sums():
    a, d0, M = _data(1)
    y0 = np.einsum("ni,nij,nj->", d0, M, d0)
    L, Q = _sums(a, d0, M)
    g = np.random.default_rng(2)
    for step in (1e-6, 1e-3, 0.05, 0.8):
        dR, dt = _rot(g.normal(0, 1, 3) * step * 0.1), g.normal(0, 1, 3) * step
        # the reference's loop: d_t = q - (dR a + dt) = d0 - (dR - I) a - dt
        dtl = d0 - (a @ (dR - np.eye(3)).T) - dt
        direct = np.einsum("ni,nij,nj->", dtl, M, dtl)
        W = np.concatenate([dR - np.eye(3), dt[:, None]], axis=1)
        model = y0 - 2 * np.sum(W * L) + np.einsum("rc,se,rsce->", W, W, Q)
        assert abs(model - direct) <= 1e-10 * max(direct, y0), (step, model, direct)
        # ... and its DECREASE keeps its digits where the difference of the two sums does not have to
        assert abs((y0 - model) - (y0 - direct)) <= 1e-6 * abs(y0 - direct) + 1e-9 * y0


def test_normal_equations_are_linear_in_the_same_sums():
    for seed, scale in ((3, 60.0), (4, 2000.0)):  # a map frame kilometres from the origin too
        a, d0, M = _data(seed, scale=scale)
        L, Q = _sums(a, d0, M)
        H, b = _normal_equations_from_sums(L, Q)
        Href = np.zeros((6, 6))
        bref = np.zeros(6)
        for ai, di, Mi in zip(a, d0, M):
            J = np.concatenate([_skew(ai), -np.eye(3)], axis=1)  # fast_gicp: dtdx0 = [skew(transed_mean_A) | -I]
            Href += J.T @ Mi @ J
            bref += J.T @ Mi @ di
        np.testing.assert_allclose(H, Href, rtol=1e-10, atol=1e-10 * np.abs(Href).max())
        np.testing.assert_allclose(b, bref, rtol=1e-9, atol=1e-10 * np.abs(bref).max())
        assert np.allclose(H, H.T)


def test_the_split_between_matrix_cores_and_registers_covers_every_entry():
    """k_gicp_linearize: 8 x 8 outer products (rows M_xx M_xy M_xz M_yy M_yz M_zz (M d0)_x (M d0)_y; columns ax ax, ax ay, ax az, ay ay,
    ay az, ax, ay, az) + 12 register sums (M x az az, sum M d0, (M d0)_z x a) + sum M (H_tt).  model_entry must find each of the 72 sums."""
    rows = ["Mxx", "Mxy", "Mxz", "Myy", "Myz", "Mzz", "Md0x", "Md0y", "Md0z"]
    cols = ["xx", "xy", "xz", "yy", "yz", "zz", "x", "y", "z", "1"]
    mf_rows, mf_cols = rows[:8], ["xx", "xy", "xz", "yy", "yz", "x", "y", "z"]
    have = {(r, c) for r in mf_rows for c in mf_cols}
    have |= {(r, "zz") for r in rows[:6]}                      # M x az az
    have |= {(r, "1") for r in rows[6:]}                       # sum M d0
    have |= {("Md0z", c) for c in ("x", "y", "z")}             # (M d0)_z x a
    have |= {(r, "1") for r in rows[:6]}                       # sum M
    need = {(r, c) for r in rows[:6] for c in cols} | {(r, c) for r in rows[6:] for c in ("x", "y", "z", "1")}
    assert need <= have and len(need) == 72
