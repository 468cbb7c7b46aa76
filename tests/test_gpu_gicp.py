"""GPU parity: the HIP GICP path vs the oracle, through the C-ABI.

Tolerance (BASELINE.md §4 / SURVEY.md §8c): same init, fixed iteration count -> ||dt|| <= 1 mm, angle(dR) <= 1e-4 rad.
The kernel computes the per-point Mahalanobis and Jacobian products in fp32 (the oracle in fp64) and accumulates the
normal equations in fp64, so H, b and the error agree to ~1e-5 relative, correspondences exactly."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-3, 1e-4


@pytest.fixture(scope="module")
def R():
    from lidar_graph_slam_amd import registration

    return registration


def _reg(R, s, t, **kw):
    reg = R.FastGICP(**kw)
    reg.setInputTarget(t)
    reg.setInputSource(s)
    return reg


def test_linearize_parity(R, small_pair):
    s, t, T = small_pair
    cs, ct = O.covariances(s), O.covariances(t)
    reg = _reg(R, s, t)
    for guess in (np.eye(4), T, synth.se3(0.3, -0.1, 0.02, 0.01, -0.01, 0.02)):
        H, b, err, corr = reg.linearize(guess)
        H2, b2, err2, corr2, _ = O.gicp_linearize(s, cs, t, ct, guess, 2.0)
        np.testing.assert_array_equal(corr, corr2)  # exact NN parity incl. the max-distance gate
        np.testing.assert_allclose(H, H2, rtol=2e-5, atol=2e-5 * np.abs(H2).max())
        np.testing.assert_allclose(b, b2, rtol=2e-5, atol=2e-5 * np.abs(b2).max())
        np.testing.assert_allclose(err, err2, rtol=2e-5)


def test_trial_cost_model_is_compute_error(R, small_pair, medium_pair):
    """F5: the optimisation prices LsqRegistration::step_lm's trial poses from sums of the linearisation pass (no pass over the
    points).  That model is FastGICP::compute_error exactly - same frozen correspondences and Mahalanobis matrices - so it must
    agree with the pass over the points (the library's own probe kernel) to fp64 rounding, and with the oracle's compute_error to
    what the fp32 Mahalanobis matrices of the kernel allow (the tolerance of test_linearize_parity)."""
    for (s, t, T) in (small_pair, medium_pair):
        cs, ct = O.covariances(s), O.covariances(t)
        reg = _reg(R, s, t)
        T_lin = synth.se3(0.3, -0.1, 0.02, 0.01, -0.01, 0.02) @ T
        _, _, err0, corr, mahal = O.gicp_linearize(s, cs, t, ct, T_lin, 2.0)
        steps = [np.eye(4), synth.se3(1e-4, -2e-4, 5e-5, 1e-5, 2e-5, -1e-5), synth.se3(0.02, -0.01, 0.005, 1e-3, -2e-3, 1.5e-3),
                 synth.se3(-0.25, 0.08, -0.02, -0.008, 0.01, -0.015), synth.se3(1.0, -0.5, 0.2, 0.05, -0.04, 0.1)]
        for D in steps:
            T_trial = D @ T_lin
            model, direct = reg.trial_cost(T_lin, T_trial)
            ref = O.gicp_error(s, t, corr, mahal, T_trial)
            # model vs the pass over the points: the same fp32 matrices, fp64 sums in a different order
            assert abs(model - direct) <= 1e-9 * max(direct, err0), (model, direct)
            np.testing.assert_allclose(direct, ref, rtol=2e-5)
            np.testing.assert_allclose(model, ref, rtol=2e-5)
        # the DECREASE of a small step - what the rho test divides - keeps its digits: no difference of two large sums
        D = synth.se3(2e-5, -1e-5, 1e-5, 2e-6, -1e-6, 1e-6)
        model, direct = reg.trial_cost(T_lin, D @ T_lin)
        base, base_direct = reg.trial_cost(T_lin, T_lin)
        assert abs((base - model) - (base_direct - direct)) <= 1e-6 * abs(base_direct - direct) + 1e-12 * base_direct


def test_align_fixed_iterations(R, small_pair):
    s, t, T = small_pair
    reg = _reg(R, s, t, max_iterations=10, force_iterations=1)
    reg.align()
    ref = O.gicp_align(s, t, O.gicp_params(max_iterations=10, force_iterations=1))
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    assert reg.result["iterations"] == 10
    assert reg.result["n_source"] == len(s) and reg.result["n_target"] == len(t)
    np.testing.assert_allclose(reg.result["H"], ref["H"], rtol=1e-3, atol=1e-3 * np.abs(ref["H"]).max())


def test_align_early_stop_and_guess(R, small_pair):
    s, t, T = small_pair
    reg = _reg(R, s, t)
    reg.align()
    ref = O.gicp_align(s, t, O.gicp_params())
    assert reg.hasConverged() and ref["converged"]
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
    assert dt <= 0.01 and dr <= 2e-3  # inside the convergence band (F9)
    assert abs(reg.result["iterations"] - ref["iterations"]) <= 1
    # previous pose as the initial guess (lidar_scan_matcher.cpp:165)
    g = ref["T"]
    reg.align(g)
    ref2 = O.gicp_align(s, t, O.gicp_params(), guess=g)
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref2["T"])
    assert reg.hasConverged() and dt <= 0.01 and dr <= 2e-3
    out = reg.align(g, return_cloud=True)
    np.testing.assert_array_equal(out, O.transform(s, reg.getFinalTransformation()))


def test_set_target_always_rebuilds(R, small_pair):
    # the node mutates the same cloud object in place and re-passes it (lidar_scan_matcher.cpp:187,209,212)
    s, t, T = small_pair
    buf = t.copy()
    reg = _reg(R, s, buf, max_iterations=8, force_iterations=1)
    reg.align()
    T1 = reg.getFinalTransformation().copy()
    shift = synth.se3(0.5, 0.2, 0.0, 0, 0, 0.01).astype(np.float32)
    buf[:] = O.transform(t, shift)
    reg.setInputTarget(buf)
    reg.align()
    T2 = reg.getFinalTransformation()
    ref = O.gicp_align(s, buf, O.gicp_params(max_iterations=8, force_iterations=1))
    dt, dr = synth.pose_error(T2, ref["T"])
    assert dt <= TOL_T and dr <= TOL_R
    assert synth.pose_error(T1, T2)[0] > 0.3


def test_gauss_newton_and_regularisations(R, small_pair):
    from lidar_graph_slam_amd import _lib as L

    s, t, _ = small_pair
    for kw in (dict(optimizer=L.OPT_GN), dict(regularization_method=L.REG_MIN_EIG), dict(regularization_method=L.REG_FROBENIUS), dict(k_correspondences=10)):
        reg = _reg(R, s, t, max_iterations=6, force_iterations=1, **kw)
        reg.align()
        ref = O.gicp_align(s, t, O.gicp_params(max_iterations=6, force_iterations=1, **kw))
        dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
        assert dt <= TOL_T and dr <= 2 * TOL_R, (kw, dt, dr)


def test_fitness_score(R, small_pair):
    s, t, T = small_pair
    reg = _reg(R, s, t)
    f0 = reg.getFitnessScore()  # before align: final_transformation_ = identity
    np.testing.assert_allclose(f0, O.fitness(s, t, np.eye(4, dtype=np.float32)), rtol=1e-9)
    reg.align()
    Tf = reg.getFinalTransformation()
    np.testing.assert_allclose(reg.getFitnessScore(), O.fitness(s, t, Tf), rtol=1e-9)
    np.testing.assert_allclose(reg.getFitnessScore(0.01), O.fitness(s, t, Tf, 0.01), rtol=1e-9)
    assert reg.getFitnessScore(-1.0) == np.finfo(np.float64).max


def test_no_correspondences_and_errors(R, small_pair):
    from lidar_graph_slam_amd import _lib as L

    s, t, _ = small_pair
    reg = _reg(R, s, t, max_iterations=3)
    far = synth.se3(0, 0, 60.0).astype(np.float32)
    reg.align(far)  # nothing within 2 m: H = b = 0 -> zero step -> "converged" at the guess, exactly as upstream behaves
    ref = O.gicp_align(s, t, O.gicp_params(max_iterations=3), guess=far)
    assert reg.hasConverged() == ref["converged"]
    np.testing.assert_array_equal(reg.getFinalTransformation(), ref["T"])
    assert reg.result["n_inliers"] == 0
    reg2 = R.FastGICP()
    with pytest.raises(L.Mi355Error):
        reg2.setInputSource(s[:5])  # fewer points than k_correspondences
    reg3 = R.FastGICP()
    reg3.setInputSource(s)
    with pytest.raises(L.Mi355Error):
        reg3.align()  # no target


def test_full_size_pair_recovers_truth(R):
    src, tgt, T = synth.make_pair()
    from lidar_graph_slam_amd.filters import PointsPreFiltering

    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
    s, t = pre.callback_sensor_points(src), pre.callback_sensor_points(tgt)
    reg = _reg(R, s, t, max_iterations=20, force_iterations=1)
    reg.align()
    ref = O.gicp_align(s, t, O.gicp_params(max_iterations=20, force_iterations=1))
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    dt, dr = synth.pose_error(reg.getFinalTransformation(), T)
    assert dt < 0.015 and dr < 5e-4


def test_batch_matches_single(R, small_pair):
    s, t, T = small_pair
    cap = max(len(s), len(t)) + 7
    nS, nT = 5, 2
    rng = np.random.default_rng(0)
    srcs = np.zeros((nS, cap, 4), np.float32)
    tgts = np.zeros((nT, cap, 4), np.float32)
    ns, nt = [], []
    shifts = [np.eye(4, dtype=np.float32), synth.se3(0.2, -0.1, 0, 0, 0, 0.01).astype(np.float32)]
    for j in range(nT):
        tj = O.transform(t, shifts[j])
        tgts[j, : len(tj)] = tj
        nt.append(len(tj))
    for i in range(nS):
        si = s[: len(s) - 11 * i]  # ragged sizes
        srcs[i, : len(si)] = si
        ns.append(len(si))
    tgt_of_src = [0, 1, 0, 1, 1]
    b = R.BatchRegistration(nS, nT, cap, max_iterations=8, force_iterations=1)
    b.set_timing(True)
    b.set_targets(tgts, nt)
    b.set_sources(srcs, ns)
    b.align(tgt_of_src, with_fitness=True)
    res, fit = b.results(with_fitness=True)
    for i in range(nS):
        ref = O.gicp_align(srcs[i, : ns[i]], tgts[tgt_of_src[i], : nt[tgt_of_src[i]]], O.gicp_params(max_iterations=8, force_iterations=1))
        dt, dr = synth.pose_error(res[i]["T"], ref["T"])
        assert dt <= TOL_T and dr <= TOL_R, (i, dt, dr)
        np.testing.assert_allclose(fit[i], O.fitness(srcs[i, : ns[i]], tgts[tgt_of_src[i], : nt[tgt_of_src[i]]], res[i]["T"]), rtol=1e-9)
    assert all(r["status"] == 0 for r in res)
    kt = b.kernel_times()  # per-kernel-family HIP-event timing (bench.py's roofline block)
    assert kt["k_gicp_linearize"][1] == 8 and kt["k_gicp_linearize"][0] > 0
    assert kt["k_gicp_correspond"][1] == 8 and kt["k_knn_cov_tile"][1] == 2 and kt["k_fitness"][1] == 1
    assert b.kernel_times() == {}  # counters reset by the read


def test_batch_of_many_converging_pairs(R, small_pair):
    """>= 8 pairs without forced iterations: correspondence tiles in the first iterations and the LM trials in two phases (the first
    trial alone, the rest only for the pairs that rejected it) — pair by pair the same pose and iteration count as the oracle.  One
    pair starts at its optimum, where first trials do get rejected."""
    s, t, T = small_pair
    cap = max(len(s), len(t)) + 3
    nS = 10
    srcs = np.zeros((nS, cap, 4), np.float32)
    tgts = np.zeros((1, cap, 4), np.float32)
    tgts[0, : len(t)] = t
    ns = []
    for i in range(nS):
        si = O.transform(s[: len(s) - 7 * i], synth.se3(0.03 * i, -0.02 * i, 0.01 * i, 0, 0, 0.002 * i).astype(np.float32))
        srcs[i, : len(si)] = si
        ns.append(len(si))
    p = O.gicp_params(max_iterations=30)
    first = O.gicp_align(srcs[0, : ns[0]], t, p)
    opt = O.transform(srcs[0, : ns[0]], first["T"].astype(np.float32))  # pair 9: already aligned
    srcs[9, : len(opt)] = opt
    srcs[9, len(opt):] = 0
    ns[9] = len(opt)
    b = R.BatchRegistration(nS, 1, cap, max_iterations=30)
    b.set_targets(tgts, [len(t)])
    b.set_sources(srcs, ns)
    b.align([0] * nS)
    res = b.results()
    for i in range(nS):
        ref = O.gicp_align(srcs[i, : ns[i]], t, p)
        dt, dr = synth.pose_error(res[i]["T"], ref["T"])
        assert dt <= TOL_T and dr <= TOL_R, (i, dt, dr)
        assert res[i]["converged"] == ref["converged"] and abs(res[i]["iterations"] - ref["iterations"]) <= 1, (i, res[i], ref)


@pytest.mark.parametrize("force,iters", [(1, 20), (1, 6), (0, 30)])
def test_incremental_correspondences_are_exact(R, medium_pair, force, iters):
    """update_correspondences (fast_gicp F3) is INCREMENTAL here: a point keeps its correspondent while it provably cannot have
    changed (margins from the nearest / runner-up gap).  After the last iteration the correspondences held that way must equal a
    search of every point from scratch at the same pose — for pairs that ran 20 forced iterations, 6 (still moving) and an
    early-stopped run, incl. one pair that starts at its optimum and one far off."""
    src, tgt, T = medium_pair
    s = O.voxel_grid(O.distance_filter(src, 1.0), 0.2)[0]
    t = O.voxel_grid(O.distance_filter(tgt, 1.0), 0.2)[0]
    nS = 12
    cap = max(len(s), len(t)) + 5
    srcs = np.zeros((nS, cap, 4), np.float32)
    tgts = np.zeros((2, cap, 4), np.float32)
    tgts[0, : len(t)] = t
    tgts[1, : len(t) - 40] = t[40:]
    ns = []
    for i in range(nS):
        D = synth.se3(0.04 * i, -0.03 * i, 0.01 * i, 0, 0, 0.003 * i) if i < nS - 1 else synth.se3(0.9, 0.5, 0.0, 0, 0, 0.04)
        si = O.transform(s[: len(s) - 11 * i], D.astype(np.float32))
        srcs[i, : len(si)] = si
        ns.append(len(si))
    opt = O.transform(s, np.asarray(T, np.float32))  # pair 5: starts (nearly) aligned
    srcs[5] = 0
    srcs[5, : len(opt)] = opt
    ns[5] = len(opt)
    b = R.BatchRegistration(nS, 2, cap, max_iterations=iters, force_iterations=force)
    b.set_targets(tgts, [len(t), len(t) - 40])
    b.set_sources(srcs, ns)
    b.align([i % 2 for i in range(nS)])
    res = b.results()
    checked, bad = b.verify_correspondences()
    assert checked == sum(ns) and bad == 0, (checked, bad)
    res2 = b.results()  # the probe leaves the results alone
    for r, r2 in zip(res, res2):
        np.testing.assert_array_equal(r["T"], r2["T"])
        assert r["iterations"] == r2["iterations"] and r["converged"] == r2["converged"]
    b.close()


def test_batch_flags_pairs_with_too_few_points(R, small_pair):
    """a cloud that holds fewer than k_correspondences points (here: handed over that way; in production a voxel
    filter can shrink a sparse key-frame sub-map that far) must not be registered with made-up neighbours: the pair reports
    status MI355_PAIR_TOO_FEW_POINTS, keeps its guess and converged = 0; the other pairs are unaffected."""
    s, t, _ = small_pair
    cap = max(len(s), len(t))
    srcs = np.zeros((3, cap, 4), np.float32)
    tgts = np.zeros((2, cap, 4), np.float32)
    srcs[0, : len(s)] = s
    srcs[1, :7] = s[:7]          # 7 < k = 20 points
    srcs[2, : len(s)] = s
    tgts[0, : len(t)] = t
    tgts[1, :5] = t[:5]          # target with 5 points
    b = R.BatchRegistration(3, 2, cap, max_iterations=6, force_iterations=1)
    b.set_targets(tgts, [len(t), 5])
    b.set_sources(srcs, [len(s), 7, len(s)])
    guess = np.tile(np.eye(4, dtype=np.float32), (3, 1, 1))
    guess[1, 0, 3] = 0.25
    b.align([0, 0, 1], guesses=guess)
    res = b.results()
    assert [r["status"] for r in res] == [0, 1, 1]
    assert not res[1]["converged"] and res[1]["iterations"] == 0 and not res[2]["converged"]
    np.testing.assert_array_equal(res[1]["T"], guess[1])
    ref = O.gicp_align(s, t, O.gicp_params(max_iterations=6, force_iterations=1))
    dt, dr = synth.pose_error(res[0]["T"], ref["T"])
    assert dt <= TOL_T and dr <= TOL_R


def test_device_inputs_follow_the_callers_stream(R, small_pair):
    """mi355_set_stream(ctx, 0, own_stream=0) means torch's (legacy) default stream, not a private one: device
    inputs written by an UNSYNCHRONISED torch kernel on the current stream must be seen complete by mi355_batch_set_*_dev."""
    import torch

    from lidar_graph_slam_amd.context import Context

    s, t, _ = small_pair
    cap = max(len(s), len(t))
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # 0 on torch's default stream
    src_h = torch.zeros((1, cap, 4), dtype=torch.float32)
    tgt_h = torch.zeros((1, cap, 4), dtype=torch.float32)
    src_h[0, : len(s)] = torch.from_numpy(s)
    tgt_h[0, : len(t)] = torch.from_numpy(t)
    src_h, tgt_h = src_h.pin_memory(), tgt_h.pin_memory()
    ref = None
    for trial in range(3):
        b = R.BatchRegistration(1, 1, cap, ctx=ctx, max_iterations=6, force_iterations=1)
        # a long chain of torch kernels that finishes by producing the inputs, handed over without any synchronisation
        big = torch.ones((64, 1 << 20), device=dev)
        for _ in range(20):
            big = big * 1.0001 + 0.0
        shift = (big.sum() * 0.0).to(torch.float32)  # 0.0, but only known when the chain has run
        src_d = src_h.to(dev, non_blocking=True) + shift
        tgt_d = tgt_h.to(dev, non_blocking=True) + shift
        b.set_targets_dev(tgt_d.data_ptr(), [len(t)])
        b.set_sources_dev(src_d.data_ptr(), [len(s)])
        b.align()
        T = b.results()[0]["T"]
        if ref is None:
            ref = O.gicp_align(s, t, O.gicp_params(max_iterations=6, force_iterations=1))["T"]
        dt, dr = synth.pose_error(T, ref)
        assert dt <= TOL_T and dr <= TOL_R, (trial, dt, dr)
        b.close()
    ctx.set_stream(None)


def test_side_streams_only_reorder_independent_work(R, small_pair):
    """mi355_batch_set_side_streams: the preparation chains of targets and sources on two overlapping side streams (default) or one
    after the other on the batch's stream (what bench.py's isolated pass uses) — bit-identical poses, and the timers see every
    kernel family either way."""
    s, t, _ = small_pair
    cap = max(len(s), len(t))
    n = 9  # >= 8 pairs: the first iterations go through k_gicp_correspond_tile
    srcs = np.zeros((n, cap, 4), np.float32)
    tgts = np.zeros((n, cap, 4), np.float32)
    ns, nt = [], []
    for i in range(n):
        si, ti = s[: len(s) - 7 * i], t[: len(t) - 2 * i]
        srcs[i, : len(si)], tgts[i, : len(ti)] = si, ti
        ns.append(len(si))
        nt.append(len(ti))
    b = R.BatchRegistration(n, n, cap, max_iterations=5, force_iterations=1)
    b.set_timing(True)
    out = []
    for mode in (True, False, None):
        b.set_side_streams(mode)
        b.set_targets(tgts, nt)
        b.set_sources(srcs, ns)
        b.align()
        out.append(np.stack([r["T"] for r in b.results()]))
        kt = b.kernel_times()
        assert kt["k_gicp_correspond"][1] == 5 and kt["k_gicp_correspond_tile"][1] == 4 and kt["k_knn_cov_tile"][1] == 2  # tiles: the first 4 iterations
    np.testing.assert_array_equal(out[0], out[1])
    np.testing.assert_array_equal(out[0], out[2])
    ref = O.gicp_align(srcs[3, : ns[3]], tgts[3, : nt[3]], O.gicp_params(max_iterations=5, force_iterations=1))
    dt, dr = synth.pose_error(out[0][3], ref["T"])
    assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    b.close()


def test_batches_on_their_own_streams(R, small_pair):
    """mi355_batch_set_stream: two batch objects on two HIP streams, used alternately with device-resident inputs and their
    results collected late (the double buffering of bench.py --inflight 2) give bit-identical poses to a batch on the context
    stream — the side streams of the preparation and the per-batch stream only reorder independent work."""
    import ctypes as C

    s, t, _ = small_pair
    cap = max(len(s), len(t))
    n = 3
    srcs = np.zeros((n, cap, 4), np.float32)
    tgts = np.zeros((n, cap, 4), np.float32)
    ns, nt = [], []
    for i in range(n):
        si, ti = s[: len(s) - 5 * i], t[: len(t) - 3 * i]
        srcs[i, : len(si)], tgts[i, : len(ti)] = si, ti
        ns.append(len(si))
        nt.append(len(ti))
    kw = dict(max_iterations=6, force_iterations=1)
    ref = R.BatchRegistration(n, n, cap, **kw)
    ref.set_targets(tgts, nt)
    ref.set_sources(srcs, ns)
    ref.align()
    want = [r["T"] for r in ref.results()]

    # device buffers and streams straight from the HIP runtime the library itself uses (no torch in this process: a second
    # HIP runtime initialised after the first one sees no GPU)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]

    def to_device(a):
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), a.nbytes) == 0
        assert hip.hipMemcpy(ptr, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        return ptr.value

    d_src, d_tgt = to_device(srcs), to_device(tgts)
    streams = []
    for _ in range(2):
        st = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(st)) == 0
        streams.append(st.value)
    batches = [R.BatchRegistration(n, n, cap, **kw) for _ in range(2)]
    for b, st in zip(batches, streams):
        b.set_stream(st)
    for rep in range(3):  # A, B, A: the third use re-enters a batch whose previous results were never fetched
        b = batches[rep % 2]
        b.set_sources_dev(d_src, ns)
        b.set_targets_dev(d_tgt, nt)
        b.align()
    for b in batches:
        got = [r["T"] for r in b.results()]
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
    batches[0].set_stream(None)  # back to the context stream
    batches[0].set_sources_dev(d_src, ns)
    batches[0].set_targets_dev(d_tgt, nt)
    batches[0].align()
    for g, w in zip([r["T"] for r in batches[0].results()], want):
        np.testing.assert_array_equal(g, w)


def test_icp_matches_the_oracle(R, small_pair, medium_pair):
    """pcl::IterativeClosestPoint (the back end's default method, graph_based_slam.cpp:142-151) on the device vs
    the oracle's restatement: iteration by iteration the same pose; with PCL's stopping rules both converge near the same
    pose (the stopping iteration hinges on the last bits of the mean squared distance)."""
    s, t, T_gt = small_pair
    src, tgt, _ = medium_pair
    s2, t2 = O.voxel_grid(O.distance_filter(src, 1.0), 0.2)[0], O.voxel_grid(O.distance_filter(tgt, 1.0), 0.2)[0]
    for a, b in ((s, t), (s2, t2)):
        for guess in (None, (T_gt @ synth.se3(0.05, -0.03, 0.0, 0, 0, 0.004)).astype(np.float32)):
            for k in (1, 4, 12):
                icp = R.IterativeClosestPoint(max_iterations=k)
                icp.setInputTarget(b)
                icp.setInputSource(a)
                icp.align(guess)
                ref = O.icp_align(a, b, O.icp_params(max_iterations=k), guess=guess)
                assert icp.hasConverged() and icp.result["iterations"] == ref["iterations"] <= k  # (the stopping rules may end it earlier)
                dt, dr = synth.pose_error(icp.getFinalTransformation(), ref["T"])
                assert dt <= 2e-5 and dr <= 2e-6, (k, dt, dr)
            icp = R.IterativeClosestPoint()
            icp.setInputTarget(b)
            icp.setInputSource(a)
            icp.align(guess)
            ref = O.icp_align(a, b, O.icp_params(), guess=guess)
            assert icp.hasConverged() and ref["converged"]
            dt, dr = synth.pose_error(icp.getFinalTransformation(), ref["T"])
            assert dt <= 0.02 and dr <= 1e-3, (dt, dr)
            np.testing.assert_allclose(icp.getFitnessScore(), O.fitness(a, b, icp.getFinalTransformation()), rtol=1e-6)
    far = s.copy()
    far[:, :3] += 1000.0  # nothing within 30 m: "Not enough correspondences found"
    icp = R.IterativeClosestPoint()
    icp.setInputTarget(t)
    icp.setInputSource(far)
    icp.align()
    assert not icp.hasConverged() and icp.result["iterations"] == 0
    with pytest.raises(Exception):
        bad = R.IterativeClosestPoint(ransac_iterations=5)
        bad.setInputTarget(t)
