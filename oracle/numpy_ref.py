"""ORACLE CROSS-CHECK — TEST INFRASTRUCTURE ONLY.

An independent NumPy/SciPy restatement of the same algorithms as oracle/*.cpp, sharing no code with it
(scipy.spatial.cKDTree for neighbours, numpy.linalg for SVD / eigen / solve, scipy.linalg.expm for the
SE(3) exponential).  Because the reference has no golden vectors (PARITY UNPINNED), the C++
oracle is pinned by requiring agreement with this file on small seeded inputs (tests/test_oracle_*.py) and
the agreed outputs are committed as fixtures under tests/golden/.

Specs followed: SURVEY.md §8a rows P1, P2, P5, F2-F9, N1-N3 ([UPSTREAM-RECALL]); call sites in
/root/reference/points_prefiltering/src/points_prefiltering.cpp:102-140 and
/root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:37-72,149-172.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm
from scipy.spatial import cKDTree


# ---------------------------------------------------------------- PCL pieces
def distance_filter(pts: np.ndarray, min_distance: float) -> np.ndarray:
    nrm = np.sqrt(pts[:, 0] * pts[:, 0] + (pts[:, 1] * pts[:, 1] + pts[:, 2] * pts[:, 2])).astype(np.float32)  # Eigen's 3-vector redux order
    return pts[min_distance < nrm.astype(np.float64)]


def voxel_grid(pts: np.ndarray, leaf: float) -> np.ndarray:
    """pcl::VoxelGrid centroids (all fields), ascending linear voxel index.  fp32 index arithmetic; centroids
    here in float64 then cast (the C++ oracle accumulates in fp32, so compare with ~1e-5 tolerance)."""
    pts = np.asarray(pts, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn = pts[:, :3].min(axis=0)
    mx = pts[:, :3].max(axis=0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div_b = max_b - min_b + 1
    ijk = (np.floor(pts[:, :3] * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div_b[0] + ijk[:, 2] * div_b[0] * div_b[1]
    order = np.argsort(idx, kind="stable")
    uniq, start, counts = np.unique(idx[order], return_index=True, return_counts=True)
    sums = np.add.reduceat(pts[order].astype(np.float64), start, axis=0)
    return (sums / counts[:, None]).astype(np.float32)


def sor(pts: np.ndarray, mean_k: int, stddev_mul: float):
    xyz = pts[:, :3].astype(np.float64)
    d, _ = cKDTree(xyz).query(xyz, k=mean_k + 1)
    dist = d[:, 1:].mean(axis=1).astype(np.float32).astype(np.float64)
    n = len(pts)
    s, sq = dist.sum(), (dist * dist).sum()
    mean = s / n
    std = np.sqrt((sq - s * s / n) / (n - 1))
    keep = ~(dist > mean + stddev_mul * std)
    return pts[keep], keep, dist


def fitness(src: np.ndarray, tgt: np.ndarray, T: np.ndarray, max_range: float = np.inf) -> float:
    p = src[:, :3].astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3].astype(np.float64)
    d, _ = cKDTree(tgt[:, :3].astype(np.float64)).query(p, k=1)
    d2 = d * d
    ok = d2 <= max_range
    return float(d2[ok].mean()) if ok.any() else float(np.finfo(np.float64).max)


# ---------------------------------------------------------------- fast_gicp
def covariances(pts: np.ndarray, k: int = 20, method: str = "PLANE") -> np.ndarray:
    xyz = pts[:, :3].astype(np.float64)
    _, nn = cKDTree(xyz).query(xyz, k=k)
    nb = xyz[nn]  # [n,k,3]
    c = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c) / k
    if method == "NONE":
        return cov
    if method == "FROBENIUS":
        C = cov + 1e-3 * np.eye(3)
        Ci = np.linalg.inv(C)
        return np.linalg.inv(Ci / np.linalg.norm(Ci, axis=(1, 2), keepdims=True))
    U, s, Vt = np.linalg.svd(cov)
    if method == "PLANE":
        vals = np.broadcast_to(np.array([1.0, 1.0, 1e-3]), s.shape)
    elif method == "MIN_EIG":
        vals = np.maximum(s, 1e-3)
    elif method == "NORMALIZED_MIN_EIG":
        vals = np.maximum(s / s.max(axis=1, keepdims=True), 1e-3)
    else:
        raise ValueError(method)
    return np.einsum("nij,nj,njk->nik", U, vals, Vt)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def se3_exp(d: np.ndarray) -> np.ndarray:
    """exp of the 4x4 twist [[skew(w), v],[0,0]] — equals fast_gicp's se3_exp (so3 quaternion + V matrix)."""
    A = np.zeros((4, 4))
    A[:3, :3] = skew(d[:3])
    A[:3, 3] = d[3:]
    return expm(A)


def gicp_linearize(src, src_cov, tgt, tgt_cov, T, max_corr):
    """H (6x6), b (6), error, correspondences — F3 + F4 with float64 throughout except the NN query, which uses
    the fp32 transform of the fp32 point like the C++ oracle."""
    T = np.asarray(T, dtype=np.float64)
    Tf = T.astype(np.float32)
    s32 = src[:, :3].astype(np.float32)
    q = (Tf[:3, :3] @ s32.T).T + Tf[:3, 3]
    d, j = cKDTree(tgt[:, :3].astype(np.float64)).query(q.astype(np.float64), k=1)
    ok = d * d < max_corr * max_corr
    corr = np.where(ok, j, -1)
    R = T[:3, :3]
    a = src[ok, :3].astype(np.float64) @ R.T + T[:3, 3]
    e = tgt[j[ok], :3].astype(np.float64) - a
    RCR = tgt_cov[j[ok]] + R @ src_cov[ok] @ R.T
    M = np.linalg.inv(RCR)
    J = np.zeros((len(a), 3, 6))
    J[:, 0, 1], J[:, 0, 2] = -a[:, 2], a[:, 1]
    J[:, 1, 0], J[:, 1, 2] = a[:, 2], -a[:, 0]
    J[:, 2, 0], J[:, 2, 1] = -a[:, 1], a[:, 0]
    J[:, :, 3:] = -np.eye(3)
    MJ = M @ J
    H = np.einsum("nki,nkj->ij", J, MJ)
    Me = np.einsum("nij,nj->ni", M, e)
    b = np.einsum("nki,nk->i", J, Me)
    err = float(np.einsum("ni,ni->", e, Me))
    return H, b, err, corr


def _gicp_error(src, tgt, corr, M_all, T):
    ok = corr >= 0
    a = src[ok, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    e = tgt[corr[ok], :3].astype(np.float64) - a
    return float(np.einsum("ni,nij,nj->", e, M_all, e))


def gicp_align(src, tgt, *, k=20, max_iterations=64, transformation_epsilon=0.01, rotation_epsilon=2e-3, max_corr=2.0,
               method="PLANE", lm_max_iterations=10, lm_init_lambda_factor=1e-9, force_iterations=False, guess=None):
    """LsqRegistration::computeTransformation + step_lm (F6, F7, F9)."""
    src_cov = covariances(src, k, method)
    tgt_cov = covariances(tgt, k, method)
    x0 = np.eye(4) if guess is None else np.asarray(guess, dtype=np.float64).copy()
    lam = -1.0
    converged = False
    iters = 0

    def is_conv(delta):
        r = np.abs(delta[:3, :3] - np.eye(3)).max() / rotation_epsilon
        t = np.abs(delta[:3, 3]).max() / transformation_epsilon
        return max(r, t) < 1

    for i in range(max_iterations):
        if converged:
            break
        iters = i + 1
        H, b, y0, corr = gicp_linearize(src, src_cov, tgt, tgt_cov, x0, max_corr)
        ok = corr >= 0
        M = np.linalg.inv(tgt_cov[corr[ok]] + x0[:3, :3] @ src_cov[ok] @ x0[:3, :3].T)
        if lam < 0:
            lam = lm_init_lambda_factor * np.abs(np.diag(H)).max()
        nu = 2.0
        accepted = False
        delta = np.eye(4)
        for _ in range(lm_max_iterations):
            d = np.linalg.solve(H + lam * np.eye(6), -b)
            delta = se3_exp(d)
            xi = delta @ x0
            yi = _gicp_error(src, tgt, corr, M, xi)
            rho = (y0 - yi) / d.dot(lam * d - b)
            if rho < 0:
                if not force_iterations and is_conv(delta):
                    accepted = True
                    break
                lam *= nu
                nu *= 2
                continue
            x0 = xi
            lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
            accepted = True
            break
        if not accepted and not force_iterations:
            break
        converged = (not force_iterations) and is_conv(delta)
    return x0.astype(np.float32), converged, iters


# ---------------------------------------------------------------- NDT
def ndt_cells(tgt: np.ndarray, resolution: float, min_points: int = 6, eig_mult: float = 0.01):
    """VoxelGridCovariance leaves (N1): dict linear_idx -> (n, mean, cov, icov) for n >= min_points and valid."""
    pts = np.asarray(tgt, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(resolution)
    min_b = np.floor(pts[:, :3].min(axis=0) * inv).astype(np.int64)
    max_b = np.floor(pts[:, :3].max(axis=0) * inv).astype(np.int64)
    div_b = max_b - min_b + 1
    ijk = (np.floor(pts[:, :3] * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div_b[0] + ijk[:, 2] * div_b[0] * div_b[1]
    out = {}
    order = np.argsort(idx, kind="stable")
    uniq, start, counts = np.unique(idx[order], return_index=True, return_counts=True)
    p64 = pts[order, :3].astype(np.float64)
    for u, s, c in zip(uniq, start, counts):
        if c < min_points:
            continue
        q = p64[s : s + c]
        mean = q.mean(axis=0)
        cov = np.cov(q.T, bias=True) * (c - 1.0) / c  # population covariance times (n-1)/n, as PCL computes it
        w, V = np.linalg.eigh(cov)
        if w[0] < 0 or w[1] < 0 or w[2] <= 0:
            continue
        lo = eig_mult * w[2]
        if w[0] < lo:
            w = w.copy()
            w[0] = lo
            if w[1] < lo:
                w[1] = lo
            cov = V @ np.diag(w) @ V.T
        out[int(u)] = (int(c), mean, cov, np.linalg.inv(cov))
    return out, min_b, div_b


def rot_xyz(roll, pitch, yaw):
    cx, sx, cy, sy, cz, sz = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def ndt_score(src, cells, min_b, div_b, resolution, p6, outlier_ratio=0.55, frozen_cells=None):
    """NDT score (eq. 6.9/6.10 [Magnusson 2009]) over DIRECT7 neighbourhoods, float64.  With ``frozen_cells`` (a list of
    per-point cell-key lists) the neighbourhood assignment is held fixed so finite differences are smooth."""
    c1 = 10 * (1 - outlier_ratio)
    c2 = outlier_ratio / resolution ** 3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    R = rot_xyz(*p6[3:])
    x = src[:, :3].astype(np.float64) @ R.T + np.asarray(p6[:3])
    inv = 1.0 / resolution
    disp = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    max_b = min_b + div_b - 1
    score = 0.0
    assign = []
    for i in range(len(x)):
        if frozen_cells is None:
            ijk = np.floor(x[i].astype(np.float32) * np.float32(inv)).astype(np.int64)
            keys = []
            for dsp in disp:
                c = ijk + dsp
                if np.any(c < min_b) or np.any(c > max_b):
                    continue
                key = int((c[0] - min_b[0]) + (c[1] - min_b[1]) * div_b[0] + (c[2] - min_b[2]) * div_b[0] * div_b[1])
                if key in cells:
                    keys.append(key)
            assign.append(keys)
        else:
            keys = frozen_cells[i]
        for key in keys:
            _, mean, _, icov = cells[key]
            xt = x[i] - mean
            e = np.exp(-d2 * xt @ icov @ xt / 2)
            if d2 * e > 1 or d2 * e < 0 or np.isnan(e):
                continue
            score += -d1 * e
    return score, assign


class NdtModel:
    """Vectorised, float64, and written independently of oracle_ndt.cpp: the NDT score of Magnusson 2009 eq. 6.9 over DIRECT7
    neighbourhoods, and a Newton optimiser that takes its gradient and Hessian from FINITE DIFFERENCES of that score (no
    analytic derivative code is shared with the oracle or the HIP path).  Used to tell an optimiser bug from a property of
    the scene (tests/test_oracle_ndt.py::test_ring_trap_*)."""

    def __init__(self, tgt, resolution=2.0, outlier_ratio=0.55, min_points=6, eig_mult=0.01):
        cells, self.min_b, self.div_b = ndt_cells(tgt, resolution, min_points, eig_mult)
        self.res = float(resolution)
        self.keys = np.array(sorted(cells), dtype=np.int64)
        self.mean = np.array([cells[k][1] for k in self.keys]).reshape(-1, 3)
        self.icov = np.array([cells[k][3] for k in self.keys]).reshape(-1, 3, 3)
        c1 = 10 * (1 - outlier_ratio)
        c2 = outlier_ratio / resolution ** 3
        d3 = -np.log(c2)
        self.d1 = -np.log(c1 + c2) - d3
        self.d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / self.d1)
        self.disp = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])

    def _transform(self, src, p6):
        return src[:, :3].astype(np.float64) @ rot_xyz(*p6[3:]).T + np.asarray(p6[:3], dtype=np.float64)

    def assign(self, src, p6):
        """[n, 7] indices into the cell arrays (-1: no usable cell there)."""
        x = self._transform(src, p6)
        ijk = np.floor(x.astype(np.float32) * np.float32(1.0 / self.res)).astype(np.int64)
        c = ijk[:, None, :] + self.disp[None, :, :] - self.min_b[None, None, :]
        ok = np.all((c >= 0) & (c < self.div_b[None, None, :]), axis=2)
        key = c[..., 0] + c[..., 1] * self.div_b[0] + c[..., 2] * self.div_b[0] * self.div_b[1]
        pos = np.searchsorted(self.keys, key)
        pos = np.minimum(pos, len(self.keys) - 1)
        ok &= self.keys[pos] == key
        return np.where(ok, pos, -1)

    def score(self, src, p6, assign=None):
        a = self.assign(src, p6) if assign is None else assign
        x = self._transform(src, p6)
        ok = a >= 0
        idx = np.where(ok, a, 0)
        xt = x[:, None, :] - self.mean[idx]
        q = np.einsum("nki,nkij,nkj->nk", xt, self.icov[idx], xt)
        e = np.exp(-self.d2 * q / 2)
        ok &= ~((self.d2 * e > 1) | (self.d2 * e < 0) | np.isnan(e))
        return float(np.sum(np.where(ok, -self.d1 * e, 0.0)))

    def derivatives(self, src, p6, h_t=1e-4, h_r=1e-5):
        """(score, gradient, Hessian) at p6 by central differences with the neighbourhood assignment held fixed."""
        p6 = np.asarray(p6, dtype=np.float64)
        a = self.assign(src, p6)
        h = np.array([h_t] * 3 + [h_r] * 3)
        f0 = self.score(src, p6, a)
        g = np.zeros(6)
        H = np.zeros((6, 6))
        fp, fm = np.zeros(6), np.zeros(6)
        for i in range(6):
            e = np.zeros(6)
            e[i] = h[i]
            fp[i], fm[i] = self.score(src, p6 + e, a), self.score(src, p6 - e, a)
            g[i] = (fp[i] - fm[i]) / (2 * h[i])
            H[i, i] = (fp[i] - 2 * f0 + fm[i]) / h[i] ** 2
        for i in range(6):
            for j in range(i + 1, 6):
                ei, ej = np.zeros(6), np.zeros(6)
                ei[i], ej[j] = h[i], h[j]
                v = (self.score(src, p6 + ei + ej, a) - self.score(src, p6 + ei - ej, a) - self.score(src, p6 - ei + ej, a)
                     + self.score(src, p6 - ei - ej, a)) / (4 * h[i] * h[j])
                H[i, j] = H[j, i] = v
        return f0, g, H

    def align(self, src, p0, step_size=0.1, transformation_epsilon=0.01, max_iterations=64):
        """Newton iterations as NDT's computeTransformation specifies them: direction = solve(H, -g),
        normalised, reversed when it does not increase the score, step length = the Newton step clamped to
        [transformation_epsilon / 2, step_size] (the line search of the ndt_omp lineage never iterates — row N4, legacy
        variant); converged when the step falls below transformation_epsilon.  Returns (p6, iterations, converged)."""
        p = np.asarray(p0, dtype=np.float64).copy()
        it = 0
        while True:
            _, g, H = self.derivatives(src, p)
            U, s, Vt = np.linalg.svd(H)
            keep = s > s.max() * 6 * np.finfo(float).eps
            d = -(Vt.T[:, keep] @ ((U[:, keep].T @ g) / s[keep]))
            nd = float(np.linalg.norm(d))
            if nd == 0 or np.isnan(nd):
                return p, it, not np.isnan(nd)
            d /= nd
            if g @ d < 0:  # not an ascent direction of the score: reverse (d_phi_0 >= 0 branch of computeStepLengthMT)
                d = -d
            step = min(max(nd, transformation_epsilon / 2), step_size)
            p = p + step * d
            if it > max_iterations or (it > 0 and step < transformation_epsilon):
                return p, it + 1, True
            it += 1


# ---------------------------------------------------------------- wire formats
_PF_DTYPE = {1: "i1", 2: "u1", 3: "<i2", 4: "<u2", 5: "<i4", 6: "<u4", 7: "<f4", 8: "<f8"}


def pointcloud2_unpack(msg: dict) -> np.ndarray:
    """pcl::fromROSMsg into pcl::PointXYZI, restated with NumPy byte views: [n, 4] float32 (x, y, z, intensity)."""
    raw = np.frombuffer(msg["data"], dtype=np.uint8)
    w, h, ps, rs = msg["width"], msg["height"], msg["point_step"], msg["row_step"]
    out = np.zeros((w * h, 4), np.float32)
    rows = raw[: rs * h].reshape(h, rs)[:, : w * ps].reshape(h * w, ps)
    for k, name in enumerate(("x", "y", "z", "intensity")):
        if name not in msg["fields"]:
            continue
        off, dt = msg["fields"][name]
        if name == "intensity" and dt != 7:
            continue  # pcl::fromROSMsg maps a field only when its datatype is the point type's (FLOAT32): warns, leaves 0
        size = np.dtype(_PF_DTYPE[dt]).itemsize
        out[:, k] = np.ascontiguousarray(rows[:, off : off + size]).view(_PF_DTYPE[dt])[:, 0].astype(np.float32)
    return out


def pointcloud2_pack_xyzi(points: np.ndarray) -> bytes:
    """pcl::toROSMsg of pcl::PointXYZI points: 32-byte records {x, y, z, 1.0f, intensity, 0, 0, 0}."""
    p = np.asarray(points, dtype=np.float32)
    rec = np.zeros((len(p), 8), np.float32)
    rec[:, :3] = p[:, :3]
    rec[:, 3] = 1.0
    rec[:, 4] = p[:, 3]
    return rec.tobytes()


def pcd_ascii(points: np.ndarray) -> str:
    """What pcl::io::savePCDFile(path, cloud) (ASCII) writes for pcl::PointXYZI points: `fs.precision(8)` stream output."""
    p = np.asarray(points, dtype=np.float32)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
            f"WIDTH {len(p)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(p)}\nDATA ascii\n")
    fmt = lambda v: "nan" if np.isnan(v) else ("%.8g" % float(v))  # noqa: E731
    return head + "".join(" ".join(fmt(v) for v in row) + "\n" for row in p)


# ---------------------------------------------------------------- pcl::IterativeClosestPoint
def icp_align(src, tgt, guess=None, max_correspondence_distance=30.0, max_iterations=100, transformation_epsilon=1e-8,
              euclidean_fitness_epsilon=1e-6):
    """Independent restatement (cKDTree + Kabsch by numpy SVD) of the ICP loop oracle_icp.cpp restates: (T float32, converged,
    iterations)."""
    from scipy.spatial import cKDTree

    tree = cKDTree(np.asarray(tgt[:, :3], dtype=np.float64))
    T = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, dtype=np.float32).copy()
    cur = (src[:, :3].astype(np.float32) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    prev_mse, it = np.finfo(np.float64).max, 0
    rot_thr = 1.0 - transformation_epsilon
    while True:
        d, j = tree.query(cur.astype(np.float64), k=1)
        ok = d * d <= max_correspondence_distance ** 2
        if ok.sum() < 3:
            return T, False, it
        P, Q = cur[ok].astype(np.float64), np.asarray(tgt[j[ok], :3], dtype=np.float64)
        mp, mq = P.mean(0), Q.mean(0)
        U, _, Vt = np.linalg.svd((P - mp).T @ (Q - mq))
        R = Vt.T @ np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))]) @ U.T
        Ti = np.eye(4)
        Ti[:3, :3], Ti[:3, 3] = R, mq - R @ mp
        Ti = Ti.astype(np.float32)
        cur = (cur @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        T = (Ti @ T).astype(np.float32)
        it += 1
        if it >= max_iterations:
            return T, True, it
        cos_angle = 0.5 * (float(Ti[0, 0]) + float(Ti[1, 1]) + float(Ti[2, 2]) - 1.0)
        if cos_angle >= rot_thr and float((Ti[:3, 3].astype(np.float64) ** 2).sum()) <= transformation_epsilon:
            return T, True, it
        mse = float((d[ok] ** 2).mean())
        if abs(mse - prev_mse) < 1e-12 or abs(mse - prev_mse) / prev_mse < euclidean_fitness_epsilon:
            return T, True, it
        prev_mse = mse


# ---------------------------------------------------------------- pclomp / PCL 1.8 BFGS GICP, for the far-from-origin question
def _euler_zyx(x, dt):
    """R = Rz(x5) Ry(x4) Rx(x3) with every entry computed in `dt` (PCL's applyState builds it from AngleAxisf)."""
    cx, sx, cy, sy, cz, sz = (dt(np.cos(dt(x[3]))), dt(np.sin(dt(x[3]))), dt(np.cos(dt(x[4]))), dt(np.sin(dt(x[4]))), dt(np.cos(dt(x[5]))),
                              dt(np.sin(dt(x[5]))))
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dt)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dt)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dt)
    return Rz @ Ry @ Rx


def memoryless_bfgs(fdf, x, max_iter, gtol):
    """The solver CLASS of pcl/registration/bfgs.h (GSL's vector_bfgs2): no Hessian estimate is kept - the next direction is
    p = g - A dx - B dg from the last step alone, normalised and pointed downhill, the first one the normalised steepest descent - here
    with SciPy's strong-Wolfe line search (c1 = 0.01, c2 just above it) where upstream has Fletcher's bracketing / sectioning.  Shares
    the two direction formulas with oracle_gicp_bfgs.cpp and nothing else."""
    import warnings

    from scipy.optimize import line_search

    f, g = fdf(x)
    p = -g / np.linalg.norm(g)
    alpha1, delta_f = 1.0, 0.0
    for _ in range(max_iter):
        fp0 = float(p @ g)
        if fp0 == 0 or not np.isfinite(fp0):
            break
        if delta_f < 0:
            alpha1 = min(1.0, 2.0 * max(-delta_f, 1e-15 * abs(f)) / (-fp0))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = line_search(lambda y: fdf(y)[0], lambda y: fdf(y)[1], x, p, gfk=g, old_fval=f, old_old_fval=f + 0.5 * alpha1 * (-fp0), c1=0.01,
                            c2=0.011, maxiter=100)[0]
        if a is None:
            break  # (upstream: "no progress")
        xn = x + a * p
        fn, gn = fdf(xn)
        dx, dg = xn - x, gn - g
        delta_f = fn - f
        dxg, dgg, dxdg = dx @ gn, dg @ gn, dx @ dg
        B = dxg / dxdg if dxdg != 0 else 0.0
        A = -(1.0 + (dg @ dg) / dxdg) * B + dgg / dxdg if dxdg != 0 else 0.0
        p = gn - A * dx - B * dg
        x, f, g = xn, fn, gn
        p = p / np.linalg.norm(p) * (-1.0 if p @ g > 0 else 1.0)
        if np.linalg.norm(g) < gtol:
            break
    return x


def gicp_bfgs_align(src, tgt, *, placement=np.float32, solver="full", k=20, max_corr=2.0, max_iterations=64, max_optimizer_iterations=20,
                    transformation_epsilon=0.01, rotation_epsilon=2e-3):
    """PCL 1.8's GICP outer loop (correspondences at the current pose, Mahalanobis matrices, minimise f(x) = 1/m sum r^T M r over
    x = (translation, Euler angles ABOUT THE ORIGIN), stop when no entry of the pose matrix moved by its epsilon) from the identity guess,
    written to ask ONE question of oracle_gicp_bfgs.cpp: when it stops converging far from the map origin, is that the fp32 in which
    upstream transforms the points (`placement`: the dtype of R(x), of the transformed point and of the residual's subtraction) or the
    inner solver?  solver = "full": SciPy's BFGS (keeps an inverse-Hessian estimate); "memoryless": memoryless_bfgs above.  Both run at
    most max_optimizer_iterations steps to a gradient norm of 1e-2, as upstream.  Returns (T float64, converged, outer iterations)."""
    from scipy.optimize import minimize

    dt = placement
    src_cov, tgt_cov = covariances(src, k, "PLANE"), covariances(tgt, k, "PLANE")
    s, t = src[:, :3].astype(dt), tgt[:, :3].astype(dt)
    tree = cKDTree(tgt[:, :3].astype(np.float64))
    trans = np.eye(4, dtype=dt)
    it, conv = 0, False
    while not conv:
        q = (s @ trans[:3, :3].T + trans[:3, 3]).astype(dt)
        d, j = tree.query(q.astype(np.float64))
        ok = d * d < max_corr * max_corr
        if ok.sum() < 4:
            break
        R = trans[:3, :3].astype(np.float64)
        M = np.linalg.inv(tgt_cov[j[ok]] + R @ src_cov[ok] @ R.T)
        p, qq = s[ok], t[j[ok]]
        m = len(p)

        def fdf(x):
            pp = (p @ _euler_zyx(x, dt).T + x[:3].astype(dt)).astype(dt)
            r = (pp - qq).astype(np.float64)
            Mr = np.einsum("nij,nj->ni", M, r)
            g = np.zeros(6)
            g[:3] = 2.0 / m * Mr.sum(0)
            A = 2.0 / m * np.einsum("ni,nj->ij", p.astype(np.float64), Mr)
            for a in range(3):  # d/d angle of sum (R p) . (M r): the derivative of R by central differences in fp64
                xp, xm = np.array(x, np.float64), np.array(x, np.float64)
                xp[3 + a] += 1e-6
                xm[3 + a] -= 1e-6
                g[3 + a] = float((((_euler_zyx(xp, np.float64) - _euler_zyx(xm, np.float64)) / 2e-6) * A.T).sum())
            return float(np.einsum("ni,ni->", r, Mr)) / m, g

        x0 = np.array([trans[0, 3], trans[1, 3], trans[2, 3], np.arctan2(trans[2, 1], trans[2, 2]), np.arcsin(-trans[2, 0]),
                       np.arctan2(trans[1, 0], trans[0, 0])], np.float64)
        if solver == "full":
            x = minimize(fdf, x0, jac=True, method="BFGS", options={"maxiter": max_optimizer_iterations, "gtol": 1e-2, "norm": 2}).x
        else:
            x = memoryless_bfgs(fdf, x0, max_optimizer_iterations, 1e-2)
        prev = trans.copy()
        trans = np.eye(4, dtype=dt)
        trans[:3, :3] = _euler_zyx(x, dt)
        trans[:3, 3] = x[:3].astype(dt)
        delta = max(np.abs(prev[:3, :3].astype(np.float64) - trans[:3, :3]).max() / rotation_epsilon,
                    np.abs(prev[:3, 3].astype(np.float64) - trans[:3, 3]).max() / transformation_epsilon)
        it += 1
        conv = it >= max_iterations or delta < 1
    return trans.astype(np.float64), conv, it
