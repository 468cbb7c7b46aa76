"""Deterministic synthetic KITTI-shaped LiDAR data.

The reference ships no data, bag or fixture, and there is no network, so every test and
benchmark input comes from this seeded ray-caster:

* scene  — ground plane z = -1.73 m, two building walls (y = +10 m / y = -12 m, 8 m high) with 1.0 m-deep
  recesses every 10 m, 60 axis-aligned boxes (cars 4 x 1.8 x 1.5 m, poles 0.3 x 0.3 x 5 m) placed by
  ``rng(seed)`` within 60 m along both kerbs (the driving lane stays free), maximum range 80 m;
* sensor — 64 rings, elevation +2.0 deg ... -24.8 deg (HDL-64E-like), ``n_azimuth`` steps (1024 -> exactly
  65 536 rays); range noise N(0, sigma), intensity U[0, 1).  Rays without a return within 80 m are emitted as
  the all-zero point, as LiDAR drivers do in organised mode, so the raw point count stays 65 536; the
  reference's distance filter (points_prefiltering.cpp:102-112, min_distance_cloud 1.0) drops them.
  (SURVEY.md §8(d) proposed re-casting them onto an 80 m dome; such points move WITH the sensor and bias every
  registration towards the identity — measured 0.4 m of bias on the 16 k-point pair — so they are not used.)

A *pair* is a target cast from pose I and a source RE-RAY-CAST from pose ``T_gt`` (not a rigid copy, so no
exact correspondences exist), each expressed in its own sensor frame; aligning source to target recovers
``T_gt``.  Output clouds are float32 [n, 4] = (x, y, z, intensity), the packed device layout.
"""
from __future__ import annotations

import os

import numpy as np

GROUND_Z = -1.73
MAX_RANGE = 80.0


def se3(tx=0.0, ty=0.0, tz=0.0, roll=0.0, pitch=0.0, yaw=0.0) -> np.ndarray:
    """4x4 float64 pose, R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = (tx, ty, tz)
    return T


def pose_error(T_est: np.ndarray, T_ref: np.ndarray):
    """(translation error [m], rotation angle error [rad]) between two 4x4 poses."""
    T_est = np.asarray(T_est, dtype=np.float64).reshape(4, 4)
    T_ref = np.asarray(T_ref, dtype=np.float64).reshape(4, 4)
    dt = float(np.linalg.norm(T_est[:3, 3] - T_ref[:3, 3]))
    dR = T_est[:3, :3] @ T_ref[:3, :3].T
    # atan2 of (|axis part|, cos): accurate near zero, where arccos of a rounded trace is not
    sn = 0.5 * np.linalg.norm([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
    cs = (np.trace(dR) - 1.0) / 2.0
    return dt, float(np.arctan2(sn, cs))


def band_error(T_a: np.ndarray, T_b: np.ndarray):
    """The difference of two poses measured the way fast_gicp's is_converged measures a step: with
    delta = T_a * T_b^-1, (max |delta.t_i|, max |(delta.R - I)_ij|).  Two correct runs that stop one LM iteration apart
    differ by one step that passed `max|t_i| < transformation_epsilon (0.01 m) and max|R - I| < rotation_epsilon (2e-3)`:
    THIS is the "convergence band" of BASELINE.md section 4, component-wise, not a Euclidean norm / rotation angle."""
    T_a = np.asarray(T_a, dtype=np.float64).reshape(4, 4)
    T_b = np.asarray(T_b, dtype=np.float64).reshape(4, 4)
    d = T_a @ np.linalg.inv(T_b)
    return float(np.abs(d[:3, 3]).max()), float(np.abs(d[:3, :3] - np.eye(3)).max())


class Scene:
    """Axis-aligned boxes + ground plane."""

    def __init__(self, seed: int = 1234, n_boxes: int = 60, x_extent: float = 400.0, box_x_range=(-60.0, 60.0)):
        """``box_x_range``: where along the street the parked cars / poles stand.  The default (+-60 m, 60 boxes) is the scene
        of the pair tests and golden fixtures; a long odometry stream needs them along its whole path (``street()``)."""
        rng = np.random.default_rng(seed)
        boxes = []
        # walls: 5 m flush / 5 m recessed by 1.0 m, alternating along x
        for y_wall, sign in ((10.0, 1.0), (-12.0, -1.0)):
            x = -x_extent
            k = 0
            while x < x_extent:
                depth = 1.0 if (k % 2) else 0.0
                y0 = y_wall + sign * depth
                lo_y, hi_y = (y0, y0 + 2.0) if sign > 0 else (y0 - 2.0, y0)
                boxes.append((x, lo_y, GROUND_Z, x + 5.0, hi_y, GROUND_Z + 8.0))
                x += 5.0
                k += 1
        for i in range(n_boxes):
            cx = rng.uniform(box_x_range[0], box_x_range[1])
            # parked at the kerbs: the driving lane |y| < 2.6 m stays free, so no sensor pose on the path sits inside or
            # against an object (a scan with 14 k returns inside 1 m is not a LiDAR scan)
            cy = rng.uniform(3.6, 7.0) if rng.uniform() < 0.45 else rng.uniform(-9.0, -3.6)
            if i % 3 == 2:  # pole
                sx, sy, sz = 0.3, 0.3, 5.0
            else:  # car, either heading
                sx, sy, sz = (4.0, 1.8, 1.5) if rng.uniform() < 0.7 else (1.8, 4.0, 1.5)
            boxes.append((cx - sx / 2, cy - sy / 2, GROUND_Z, cx + sx / 2, cy + sy / 2, GROUND_Z + sz))
        b = np.asarray(boxes, dtype=np.float64)
        self.lo = b[:, :3]
        self.hi = b[:, 3:]

    def cast(self, origin: np.ndarray, dirs: np.ndarray) -> np.ndarray:
        """Range along each unit direction (world frame) to the first hit; MAX_RANGE when nothing is hit."""
        n = dirs.shape[0]
        t_hit = np.full(n, MAX_RANGE, dtype=np.float64)
        dz = dirs[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (GROUND_Z - origin[2]) / dz
        ok = (dz < 0) & (tg > 0)
        t_hit = np.where(ok & (tg < t_hit), tg, t_hit)
        d = np.where(np.abs(dirs) < 1e-12, 1e-12, dirs)
        inv = 1.0 / d
        # only boxes that can be within range of this origin (walls span the whole trajectory)
        near = (self.hi[:, 0] > origin[0] - MAX_RANGE) & (self.lo[:, 0] < origin[0] + MAX_RANGE)
        blo, bhi = self.lo[near] - origin[None, :], self.hi[near] - origin[None, :]
        # slab test, chunked over boxes to bound memory
        for s in range(0, blo.shape[0], 32):
            tmin = np.full((n, min(32, blo.shape[0] - s)), -np.inf)
            tmax = np.full_like(tmin, np.inf)
            for ax in range(3):
                t0 = blo[None, s : s + 32, ax] * inv[:, ax, None]
                t1 = bhi[None, s : s + 32, ax] * inv[:, ax, None]
                np.maximum(tmin, np.minimum(t0, t1), out=tmin)
                np.minimum(tmax, np.maximum(t0, t1), out=tmax)
            hit = (tmax >= tmin) & (tmax > 0)
            t_enter = np.where(tmin > 0, tmin, tmax)  # origin inside a box: exit distance
            t_box = np.where(hit, t_enter, np.inf).min(axis=1)
            t_hit = np.minimum(t_hit, t_box)
        return t_hit


def street(length: float, seed: int = 1234, boxes_per_metre: float = 0.5) -> Scene:
    """A street scene with parked cars / poles along ``length`` metres (x from -20 m), for odometry streams (BASELINE config 4:
    1000 scans at 0.8 m = 800 m of path; the default Scene has objects within +-60 m only, beyond which translation along
    the street is observable from the 5 m wall recesses alone)."""
    return Scene(seed, n_boxes=int(boxes_per_metre * (length + 40.0)), x_extent=length + 120.0, box_x_range=(-20.0, length + 20.0))


def sensor_dirs(n_rings: int = 64, n_azimuth: int = 1024) -> np.ndarray:
    elev = np.deg2rad(np.linspace(2.0, -24.8, n_rings))
    az = np.linspace(0.0, 2.0 * np.pi, n_azimuth, endpoint=False)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (n_rings, n_azimuth))], axis=-1)
    return d.reshape(-1, 3)


def cast_scan(scene: Scene, T_world_sensor: np.ndarray, *, n_rings: int = 64, n_azimuth: int = 1024,
              noise_sigma: float = 0.02, seed: int = 1, jitter_sigma: float = 0.002, ground_max_range: float | None = None) -> np.ndarray:
    """One scan in the SENSOR frame, float32 [n_rings*n_azimuth, 4].  ``noise_sigma`` acts along the ray; ``jitter_sigma``
    is a small isotropic 3-D jitter (beam divergence / angular encoder noise) so that the returns of one azimuth column
    are not EXACTLY coplanar — exactly singular neighbourhoods make eigenvalue-sign tests implementation-defined.
    ``ground_max_range``: ground returns beyond this range are dropped (no-return rays).  A real road does not return at the
    grazing angles the far rings hit it with; a perfectly flat, perfectly reflective ground out to 80 m lays concentric rings
    around EVERY sensor pose, and matching ring to ring pulls NDT (and, weakly, GICP) towards the identity."""
    T = np.asarray(T_world_sensor, dtype=np.float64)
    d_s = sensor_dirs(n_rings, n_azimuth)
    d_w = d_s @ T[:3, :3].T
    rng_ = scene.cast(T[:3, 3], d_w)
    rng = np.random.default_rng(seed)
    if noise_sigma > 0:
        rng_ = np.where(rng_ >= MAX_RANGE, rng_, rng_ + rng.normal(0.0, noise_sigma, size=rng_.shape))
    no_return = rng_ >= MAX_RANGE
    if ground_max_range is not None:
        hit_z = T[2, 3] + d_w[:, 2] * rng_
        no_return = no_return | ((np.abs(hit_z - GROUND_Z) < 0.05) & (rng_ > ground_max_range))
    pts = d_s * rng_[:, None]
    if jitter_sigma > 0:
        pts = pts + rng.normal(0.0, jitter_sigma, size=pts.shape)
    pts[no_return] = 0.0
    out = np.empty((pts.shape[0], 4), dtype=np.float32)
    out[:, :3] = pts.astype(np.float32)
    out[:, 3] = rng.uniform(0.0, 1.0, size=pts.shape[0]).astype(np.float32)
    return out


def default_pair_gt() -> np.ndarray:
    """BASELINE configs 1/2: 1.0 m forward, 0.1 m left, 2 deg yaw."""
    return se3(1.0, 0.1, 0.0, 0.0, 0.0, np.deg2rad(2.0))


def make_pair(*, n_rings: int = 64, n_azimuth: int = 1024, T_gt: np.ndarray | None = None, scene_seed: int = 1234,
              noise_sigma: float = 0.02, seed: int = 1, T_target: np.ndarray | None = None):
    """(source, target, T_gt): target cast at ``T_target`` (default I), source at ``T_target @ T_gt``."""
    scene = Scene(scene_seed)
    T_gt = default_pair_gt() if T_gt is None else np.asarray(T_gt, dtype=np.float64)
    T_t = np.eye(4) if T_target is None else np.asarray(T_target, dtype=np.float64)
    tgt = cast_scan(scene, T_t, n_rings=n_rings, n_azimuth=n_azimuth, noise_sigma=noise_sigma, seed=seed)
    src = cast_scan(scene, T_t @ T_gt, n_rings=n_rings, n_azimuth=n_azimuth, noise_sigma=noise_sigma, seed=seed + 7919)
    return src, tgt, T_gt


def bench_pair_spec(i: int, rank: int = 0) -> dict:
    """Arguments of make_pair for the i-th distinct scene / pose of the benchmark batch (bench.py): its own street scene (box
    placement `scene_seed`), its own sensor pose on it and its own relative motion - 0.6 ... 1.2 m forward, +-0.12 m sideways,
    +-2.4 deg of yaw (BASELINE configs[1] is the i = 0 motion: 1.0 m, 0.1 m, 2 deg) - and its own noise seeds."""
    r = np.random.default_rng(9000 + 131 * i + 7919 * rank)
    if i == 0 and rank == 0:
        T_gt = default_pair_gt()
    else:
        T_gt = se3(r.uniform(0.6, 1.2), r.uniform(-0.12, 0.12), 0.0, 0.0, 0.0, np.deg2rad(r.uniform(-2.4, 2.4)))
    T_tgt = se3(r.uniform(-35.0, 35.0), r.uniform(-1.2, 1.2), 0.0, 0.0, 0.0, r.uniform(-0.06, 0.06))
    return dict(T_gt=T_gt, T_target=T_tgt, scene_seed=1234 + 17 * i + 1009 * rank, seed=11 + 17 * i + 101 * rank)


def make_pair_job(spec: dict):
    """make_pair(**spec) as a picklable job for a pool of spawned workers (ray casting is single-threaded NumPy)."""
    return make_pair(**spec)


_STREETS = {}


def cast_job(spec):
    """One scan of a street scene as a picklable job for a pool of spawned workers: spec = (street length, 4x4 sensor pose, noise
    seed, n_azimuth, ground_max_range).  The scene is built once per worker process."""
    length, pose, seed, n_azimuth, ground_max_range = spec
    if length not in _STREETS:
        _STREETS[length] = street(length)
    return cast_scan(_STREETS[length], np.asarray(pose, dtype=np.float64), n_azimuth=n_azimuth, seed=seed, ground_max_range=ground_max_range)


def cast_jobs_in_child(jobs, workers: int | None = None):
    """[cast_job(j) for j in jobs], computed by ONE child interpreter that owns the worker pool.  For callers whose own process may
    have initialised the GPU already (a pytest session): the caller starts a single plain child (python -m lidar_graph_slam_amd._cast_child), which never
    touches the GPU and is the one that spawns workers; jobs and scans travel through a temporary directory."""
    import pickle
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory(prefix="mi355_cast_") as d:
        with open(os.path.join(d, "jobs.pkl"), "wb") as f:
            pickle.dump(list(jobs), f)
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.run([sys.executable, "-m", "lidar_graph_slam_amd._cast_child", d] + ([str(workers)] if workers else []),
                       check=True, env=env, cwd=root)
        z = np.load(os.path.join(d, "scans.npz"))
        return [z[f"s{i}"] for i in range(len(z.files))]


def trajectory(n: int, *, speed: float = 8.0, hz: float = 10.0, seed: int = 3) -> np.ndarray:
    """[n,4,4] KITTI-like path: `speed` m/s at `hz` scans/s, gentle yaw-rate changes <= 10 deg/s, staying in the corridor."""
    rng = np.random.default_rng(seed)
    poses = np.empty((n, 4, 4))
    x = y = yaw = 0.0
    dt = 1.0 / hz
    rate = 0.0
    for i in range(n):
        poses[i] = se3(x, y, 0.0, 0.0, 0.0, yaw)
        if i % 20 == 0:
            rate = np.deg2rad(rng.uniform(-10.0, 10.0))
        # steer back towards the corridor axis so the path stays between the walls
        rate_eff = rate - 0.3 * yaw - 0.02 * y
        yaw += rate_eff * dt
        x += speed * dt * np.cos(yaw)
        y += speed * dt * np.sin(yaw)
    return poses


def random_perturbation(rng: np.random.Generator, max_t: float = 2.0, max_deg: float = 5.0) -> np.ndarray:
    """Seeded pose offset <= max_t metres / max_deg degrees (config 5 candidates)."""
    t = rng.uniform(-1.0, 1.0, 3)
    t = t / max(np.linalg.norm(t), 1e-9) * rng.uniform(0.0, max_t)
    t[2] *= 0.1
    ang = np.deg2rad(rng.uniform(-max_deg, max_deg))
    return se3(t[0], t[1], t[2], 0.0, 0.0, ang)

