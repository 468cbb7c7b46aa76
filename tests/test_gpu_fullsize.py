"""The BASELINE configs at their STATED sizes: every test here runs the HIP path through the C-ABI on
full-size inputs and checks it against the oracle on the same inputs and/or against the known ground truth.

  config 2  noise-free 64 k-point pair: <= 5 mm / 5e-4 rad vs truth, <= 1 mm / 1e-4 rad vs the oracle
  config 3  64 k-point scan vs a ~1 M-point target, 1.0 m NDT cells, vs O.ndt_align
  config 4  KITTI-shaped stream through ScanMatcherCore, window 20: 1 000 scans x 64 k rays (the stated size) with FAST_GICP in
            the default mode - first 24 scans against the reference-policy restatement registering with the oracle, all of them
            against the ground truth (ATE, key frames), with ground returns cut at 20 m and out to 80 m; and a 100-scan stream
            for both engines and the opt-in carried-covariance mode
  config 5  the per-GPU share of the loop-closure batch (512 candidates, 8 sub-maps, 0.5 m voxel) vs truth, a sample vs the
            oracle (pose and fitness)
"""
import os

import numpy as np
import pytest

import parity
from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _pf(c, leaf=0.2):
    return O.voxel_grid(O.distance_filter(c, 1.0), leaf)[0]


def test_config2_noise_free_pair_meets_the_stated_truth_tolerance():
    from lidar_graph_slam_amd.filters import PointsPreFiltering
    from lidar_graph_slam_amd.registration import FastGICP

    scene = synth.Scene(1234)
    T_gt = synth.default_pair_gt()
    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
    for gmr, tol_t in ((None, 5e-3), (20.0, 2e-3)):
        tgt = synth.cast_scan(scene, np.eye(4), noise_sigma=0.0, jitter_sigma=0.0, seed=1, ground_max_range=gmr)
        src = synth.cast_scan(scene, T_gt, noise_sigma=0.0, jitter_sigma=0.0, seed=7920, ground_max_range=gmr)
        s, t = pre.callback_sensor_points(src), pre.callback_sensor_points(tgt)
        np.testing.assert_array_equal(s, _pf(src))  # the prefilter is bit-exact at full size
        reg = FastGICP(max_iterations=20, force_iterations=1)
        reg.setInputTarget(t)
        reg.setInputSource(s)
        reg.align()
        dt, dr = synth.pose_error(reg.getFinalTransformation(), T_gt)
        assert dt <= tol_t and dr <= 5e-4, (gmr, dt, dr)  # <= 5 mm / 5e-4 rad on the noise-free scene
        ref = O.gicp_align(s, t, O.gicp_params(max_iterations=20, force_iterations=1))
        dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
        assert dt <= 1e-3 and dr <= 1e-4, (gmr, dt, dr)


def test_config3_ndt_scan_vs_1m_point_submap():
    from lidar_graph_slam_amd.registration import NormalDistributionsTransform

    scene = synth.Scene(1234)
    poses = [synth.se3(1.0 * i, 0.02 * i, 0, 0, 0, 0.002 * i) for i in range(16)]
    parts = []
    for i, P in enumerate(poses):
        c = synth.cast_scan(scene, P, seed=200 + i)
        c = c[np.abs(c[:, :3]).sum(1) > 0]
        c[:, :3] = (c[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3]).astype(np.float32)
        parts.append(c)
    sub = np.concatenate(parts, axis=0)
    assert 0.9e6 < len(sub) < 1.1e6, len(sub)
    T_true = synth.se3(7.5, 0.15, 0, 0, 0, 0.015)
    scan = synth.cast_scan(scene, T_true, seed=999)
    scan = scan[np.abs(scan[:, :3]).sum(1) > 0]
    guess = synth.se3(7.2, 0.1, 0, 0, 0, 0.01).astype(np.float32)
    ndt = NormalDistributionsTransform(resolution=1.0)
    ndt.setInputTarget(sub)
    ndt.setInputSource(scan)
    ndt.align(guess)
    ref = O.ndt_align(scan, sub, O.ndt_params(resolution=1.0), guess=guess)
    cells = ndt.cells()
    ref_cells = O.ndt_cells(sub, O.ndt_params(resolution=1.0))
    assert abs(len(cells["idx"]) - len(ref_cells["idx"])) <= 2  # leaves (a voxel on the eigenvalue-sign edge may differ)
    assert ndt.hasConverged() and ref["converged"]
    assert abs(ndt.result["iterations"] - ref["iterations"]) <= 1
    parity.assert_in_band(ndt.getFinalTransformation(), ref["T"], "config3/ndt_1M")  # inside the convergence band of the last step
    dt, dr = synth.pose_error(ndt.getFinalTransformation(), T_true)
    assert dt <= 0.05 and dr <= 2e-3, (dt, dr)


@pytest.fixture(scope="module")
def street_stream():
    n = 100
    scene = synth.street(n * 0.8 + 10.0)
    poses = synth.trajectory(n, seed=3)
    from lidar_graph_slam_amd.filters import PointsPreFiltering

    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
    scans = [pre.callback_sensor_points(synth.cast_scan(scene, poses[i], seed=300 + i, ground_max_range=20.0)) for i in range(n)]
    return poses, scans


def test_config4_gicp_stream_matches_the_reference_policy(street_stream):
    """Scan by scan against the restatement of callback_cloud registering with the ORACLE, in the DEFAULT mode
    (reuse_key_frame_covariances = 0): the target covariances are recomputed over the merged sub-map exactly as fast_gicp does."""
    from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher
    from ref_scan_matcher import RefScanMatcher

    poses, scans = street_stream
    sm = LidarScanMatcher(registration_method="FAST_GICP")
    ref = RefScanMatcher(method="FAST_GICP")
    for i in range(24):
        a, b = sm.callback_cloud(scans[i]), ref.callback_cloud(scans[i])
        assert a["first_scan"] == b["first_scan"]
        if a["first_scan"]:
            continue
        assert a["converged"] and b["converged"], i
        parity.assert_in_band(a["pose"], b["pose"], f"config4_100/FAST_GICP/scan{i}")
        assert a["new_key_frame"] == b["new_key_frame"], i
        assert a["target_points"] == len(ref.target), i
    tgt = sm.target_cloud()
    # the device-assembled sub-map is the reference's up to the key-frame poses, which agree inside the convergence band
    assert tgt.shape == ref.target.shape
    np.testing.assert_allclose(tgt[:, :3], ref.target[:, :3], rtol=0, atol=0.03)
    np.testing.assert_array_equal(tgt[:, 3], ref.target[:, 3])


@pytest.mark.parametrize("method,reuse", [("FAST_GICP", False), ("FAST_GICP", True), ("NDT_OMP", False)])
def test_config4_stream_tracks_the_ground_truth(street_stream, method, reuse):
    """100 scans x 64 k rays, window 20 (80 m of path): odometry error against the ground truth at every scan.  NDT_OMP at 2 m
    cells — the reference's default — tracks on this scene (the corridor scene of round 1 it could not track on is analysed in
    tests/test_oracle_ndt.py::test_ring_trap_is_a_property_of_the_scene_not_of_the_optimiser) with the scale bias NDT has on
    sparse far-field returns: every registration comes out ~0.5 % short (0.795 m for a true 0.8 m step), which the oracle
    reproduces scan for scan (0.307 m at scan 90, 0.368 m at scan 99 on this very stream).  GICP has no such bias."""
    from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher

    poses, scans = street_stream
    sm = LidarScanMatcher(registration_method=method, reuse_key_frame_covariances=reuse)
    n_kf, worst = 0, 0.0
    for i, s in enumerate(scans):
        o = sm.callback_cloud(s)
        if o["first_scan"]:
            continue
        assert o["converged"], (method, i)
        n_kf += o["new_key_frame"]
        dt, dr = synth.pose_error(o["pose"], np.linalg.inv(poses[0]) @ poses[i])
        worst = max(worst, dt)
        path = 0.8 * i
        bound = 0.10 + 0.006 * path if method == "NDT_OMP" else 0.05 + 0.002 * path
        assert dt < bound and dr < 0.02, (method, i, dt, dr, bound)
    # key frames every ~1.0 m of 0.8 m steps: every second scan, +-
    assert 40 <= n_kf <= 60, n_kf
    assert o["target_points"] > 150_000  # 20 key frames in the window


def _cast_for_stream(args):
    """Worker of the 1 000-scan stream (spawned processes: NumPy only, never the GPU)."""
    length, i, pose, gmr = args
    from lidar_graph_slam_amd import synth as S

    return S.cast_scan(S.street(length), pose, seed=300 + i, ground_max_range=gmr)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("ground_max_range", [20.0, None])
def test_config4_at_its_stated_size_1000_scans(ground_max_range):
    """BASELINE configs[3] at its stated length: 1 000 scans x 64 k rays (800 m of path at 0.8 m/scan), key frame 1.0 m,
    window 20, FAST_GICP in the default (reference-arithmetic) mode, through ScanMatcherCore:
      * scans 0..23 scan by scan against RefScanMatcher registering with the oracle, live (inside the convergence band),
      * all 1 000 scans against the ORACLE's trajectory of the same stream (tests/golden/stream1000_gmr*.npz, made by
        tests/golden/make_stream_golden.py from RefScanMatcher + oracle alone): same converged / key-frame decisions scan for
        scan, pose inside the band (two bands for at most 1 % of the scans) while the two sub-map histories are the same, and the same drift,
      * all 1 000 scans against the ground truth: drift bound, ATE (RMSE of the translation error), key-frame count.
    ground_max_range = 20 m is the street scene of the 100-scan tests; None keeps the ground returns out to 80 m as SURVEY 8(d)
    specifies the scene (the rings that trap NDT, tests/test_oracle_ndt.py; GICP tracks through them - and drifts far less:
    the ground rings constrain yaw / pitch, 0.08 m after 800 m against 0.6 % of the path without them, oracle and GPU alike).
    The scans are ray-cast by a pool of spawned CPU workers while the GPU registers."""
    import multiprocessing as mp

    from lidar_graph_slam_amd.filters import PointsPreFiltering
    from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher
    from ref_scan_matcher import RefScanMatcher

    n = 1000
    length = n * 0.8 + 10.0
    poses = synth.trajectory(n, seed=3)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"stream1000_gmr{'80' if ground_max_range is None else int(ground_max_range)}.npz"))
    assert int(gold["scans"]) == n
    same_history = True  # until a converged / key-frame decision differs, GPU and oracle register against the same sub-maps
    over_one_band = 0
    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
    sm = LidarScanMatcher(registration_method="FAST_GICP")
    ref = RefScanMatcher(method="FAST_GICP")
    err = np.zeros(n)
    n_kf = 0
    workers = max(2, min(48, (os.cpu_count() or 8) - 2))
    with mp.get_context("spawn").Pool(workers) as pool:
        raws = pool.imap(_cast_for_stream, [(length, i, poses[i], ground_max_range) for i in range(n)], chunksize=2)
        for i, raw in enumerate(raws):
            scan = pre.callback_sensor_points(raw)
            o = sm.callback_cloud(scan)
            if i < 24:
                b = ref.callback_cloud(scan)
                assert o["first_scan"] == b["first_scan"]
                if not o["first_scan"]:
                    assert o["converged"] and b["converged"], i
                    parity.assert_in_band(o["pose"], b["pose"], f"config4_1000/gmr{ground_max_range}/scan{i}")
                    assert o["new_key_frame"] == b["new_key_frame"], i
                    assert o["target_points"] == len(ref.target), i
            assert len(scan) == int(gold["n_points"][i]), i  # the HIP prefilter feeds the stream the oracle's scans
            if o["first_scan"]:
                continue
            assert o["converged"], i
            n_kf += o["new_key_frame"]
            # against the oracle's trajectory: identical decisions and in-band poses as long as the histories agree; a single
            # flipped key-frame decision (a scan sitting on the 1.0 m gate) gives the two streams different sub-maps from then
            # on, after which they are two valid odometry runs and only their drift is comparable
            if same_history:
                bt, br, dt_o, dr_o = parity.deltas(o["pose"], gold["pose"][i])
                parity._log(f"config4_1000/gmr{ground_max_range}/vs_golden/scan{i}", band_t=bt, band_r=br, dt=dt_o, dr=dr_o, scale=2.0)
                if bool(gold["new_key_frame"][i]) != o["new_key_frame"] or not gold["converged"][i]:
                    same_history = False
                    first_split = i
                elif not os.environ.get("PARITY_LOG_ONLY"):
                    # two bands, not one: by now the two streams do NOT register the same inputs - their sub-maps were assembled from
                    # key-frame poses that already differ inside the band (parity.assert_in_band's rule for `scale`), and the
                    # differences feed back: a convergence test that flips on a last-ulp difference of a covariance moves a pose by
                    # one LM step, and each of two converged runs is only known to sit within one band of the optimum.  (Observed:
                    # 1e-5 m over the first hundreds of scans, 1e-4 .. 1e-3 m later, single scans of 3 - 19 mm.)  One band must
                    # still hold for all but 1 % of the scans.
                    assert bt <= 2.0 * parity.BAND_T and br <= 2.0 * parity.BAND_R, (i, bt, br)
                    over_one_band += bt > parity.BAND_T or br > parity.BAND_R
            dt, dr = synth.pose_error(o["pose"], np.linalg.inv(poses[0]) @ poses[i])
            err[i] = dt
            dt_gold, dr_gold = synth.pose_error(gold["pose"][i], gold["truth"][i])
            # drift: at most 1 % of the path and 0.06 rad (LiDAR odometry without loop closure; with the ground cut at 20 m yaw is
            # weakly observable and the ORACLE ends 5.4 m / 0.036 rad off after 800 m), and not worse than the oracle's own by
            # more than a quarter of a percent of the path / 5 mrad
            assert dt < 0.10 + 0.01 * (0.8 * i) and dr < 0.06, (i, dt, dr)
            assert dt < dt_gold + 0.10 + 0.0025 * (0.8 * i) and dr < dr_gold + 5e-3, (i, dt, dt_gold, dr, dr_gold)
    ate = float(np.sqrt(np.mean(err**2)))
    err_gold = np.array([synth.pose_error(gold["pose"][i], gold["truth"][i])[0] for i in range(n)])
    ate_gold = float(np.sqrt(np.mean(err_gold**2)))
    parity._log(f"config4_1000/gmr{ground_max_range}/summary", ate_m=ate, end_err_m=float(err[-1]), key_frames=n_kf, path_m=0.8 * (n - 1),
                oracle_ate_m=ate_gold, oracle_end_err_m=float(err_gold[-1]), oracle_key_frames=int(gold["new_key_frame"].sum()) - 1,
                same_history_until_scan=(n if same_history else first_split), scans_over_one_band=int(over_one_band))
    assert over_one_band <= n // 100, over_one_band
    assert ate < max(1.25 * ate_gold, ate_gold + 0.05), (ate, ate_gold)
    assert abs(n_kf - (int(gold["new_key_frame"].sum()) - 1)) <= 5
    assert 420 <= n_kf <= 580, n_kf  # a key frame every 1.0 m of 0.8 m steps: every second scan, +-
    assert o["n_key_frames"] == n_kf + 1 and o["target_points"] > 150_000


@pytest.mark.parametrize("method,n", [("FAST_GICP", 512), ("NDT_OMP", 128), ("ICP", 128)])
def test_config5_per_gpu_share_of_the_loop_closure_batch(method, n):
    """FAST_GICP: the 512-candidate per-GPU share of BASELINE configs[4].  NDT_OMP (graph_based_slam.cpp:101-119, 1.0 m cells as
    graph_based_slam.param.yaml:18): the same candidates through the batched NDT, every pair's More-Thuente state on the device.  ICP - the back
    end's yaml default (graph_based_slam.param.yaml:9, graph_based_slam.cpp:142-151: 30 m, 100 iterations, 1e-8, 1e-6): the candidates through the
    batched ICP, its first iterations on the wave-cooperative tiles (128 pairs >= 8), the settled ones on search margins."""
    from lidar_graph_slam_amd import loop_batch
    from lidar_graph_slam_amd.filters import PointsPreFiltering

    S, casts = 8, 8
    scene = synth.street(20.0 * S + 20.0)
    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
    rng = np.random.default_rng(5)
    tgt_of_src = np.arange(n) % S
    cast_of = (np.arange(n) // S) % casts
    true_off = [synth.random_perturbation(np.random.default_rng(1000 + s * 64 + c), 1.0, 2.0) for s in range(S) for c in range(casts)]
    wrong = [synth.random_perturbation(rng, 1.5, 4.0) for _ in range(n)]
    sub_pose = lambda s: synth.se3(20.0 * s, 0.0, 0.0)  # noqa: E731
    true_pose = lambda p: sub_pose(int(tgt_of_src[p])) @ true_off[int(tgt_of_src[p]) * casts + int(cast_of[p])]  # noqa: E731
    graph_pose = lambda p: true_pose(p) @ wrong[p]  # noqa: E731

    def place(c, T):
        o = c.copy()
        o[:, :3] = (c[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        return o

    scans, submaps = {}, {}

    def scan_at(s, c):
        if (s, c) not in scans:
            scans[(s, c)] = pre.callback_sensor_points(synth.cast_scan(scene, sub_pose(s) @ true_off[s * casts + c], seed=9000 + s * 64 + c, ground_max_range=30.0))
        return scans[(s, c)]

    def load_source(p):
        return place(scan_at(int(tgt_of_src[p]), int(cast_of[p])), graph_pose(p))

    def load_target(s):
        if s not in submaps:
            parts = []
            for j in range(-2, 3):
                P = sub_pose(s) @ synth.se3(1.0 * j, 0.0, 0.0)
                parts.append(place(pre.callback_sensor_points(synth.cast_scan(scene, P, seed=500 + 10 * s + j, ground_max_range=30.0)), P))
            submaps[s] = np.concatenate(parts, axis=0)
        return submaps[s]

    kw = {"FAST_GICP": dict(max_iterations=64), "NDT_OMP": dict(max_iterations=100, resolution=1.0, step_size=0.1, transformation_epsilon=0.01),
          "ICP": dict(max_iterations=100, max_correspondence_distance=30.0, transformation_epsilon=1e-8, euclidean_fitness_epsilon=1e-6)}[method]
    out = loop_batch.register_sharded(tgt_of_src, load_source, load_target, 100_000, target_prefilter=loop_batch.voxel_prefilter(0.5), method=method, **kw)
    assert out["reported"].all() and (out["status"] == 0).all()
    errs = np.array([synth.pose_error(out["T"][p], true_pose(p) @ np.linalg.inv(graph_pose(p))) for p in range(n)])
    accepted = out["converged"] & (out["fitness"] < 0.3)  # graph_based_slam.cpp:321-328
    if method == "FAST_GICP":
        assert accepted.sum() >= 0.98 * n and (errs[accepted, 0] < 0.05).mean() > 0.98, (accepted.sum(), np.median(errs[:, 0]))
    elif method == "ICP":  # point-to-point with a 30 m range on voxel-filtered surfaces: decimetres, like the oracle (benchmarks/config5.py: median 5 - 9 cm)
        assert accepted.sum() >= 0.9 * n and np.median(errs[accepted, 0]) < 0.15, (accepted.sum(), np.median(errs[:, 0]))
    else:  # NDT with 1 m cells from up to 1.5 m / 4 deg off: coarser, and some candidates stay in a neighbouring basin — like the oracle
        assert accepted.sum() >= 0.9 * n and np.median(errs[accepted, 0]) < 0.10, (accepted.sum(), np.median(errs[:, 0]))
    # a sample against the oracle: same voxel-filtered sub-map, same candidate, identity guess (graph_based_slam.cpp:311-322)
    for p in (0, 77, 203, 511) if n > 511 else (0, 37, 77, 127):
        t = O.voxel_grid(load_target(int(tgt_of_src[p])), 0.5)[0]
        assert out["n_target"][p] == len(t)
        if method == "FAST_GICP":
            ref = O.gicp_align(load_source(p), t, O.gicp_params(max_iterations=64))
        elif method == "ICP":
            ref = O.icp_align(load_source(p), t, O.icp_params(**kw))
            assert abs(int(out["iterations"][p]) - ref["iterations"]) <= 2, (p, out["iterations"][p], ref["iterations"])
        else:
            ref = O.ndt_align(load_source(p), t, O.ndt_params(**kw))
            assert abs(int(out["iterations"][p]) - ref["iterations"]) <= 2, (p, out["iterations"][p], ref["iterations"])
        parity.assert_in_band(out["T"][p], ref["T"], f"config5/{method}/pair{p}")  # both stop inside the convergence band
        np.testing.assert_allclose(out["fitness"][p], O.fitness(load_source(p), t, out["T"][p]), rtol=1e-6)
