"""GPU parity: pclomp::GeneralizedIterativeClosestPoint ("GICP": lidar_scan_matcher.cpp:73-96, graph_based_slam.cpp:120-141) for many
pairs at once (MI355_METHOD_GICP in mi355_batch_create, round 5) — every pair's Fletcher line search / BFGS update advanced on the
device (csrc/bfgs_dev.hpp), one correspondence pass + one cost evaluation per pair and round — against the registration object (a batch
of one) and the oracle."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from lidar_graph_slam_amd import registration

    return registration


@pytest.mark.parametrize("max_iterations", [1, 64])
def test_batch_gicp_matches_single_and_oracle(R, medium_pair, max_iterations):
    s, t, T = medium_pair  # (the pair tests/test_gpu_gicp_bfgs.py holds the registration object to the oracle on, at the same tolerances)
    s, t = s[np.abs(s[:, :3]).sum(1) > 0], t[np.abs(t[:, :3]).sum(1) > 0]
    cap = max(len(s), len(t)) + 7
    nS, nT = 9, 2  # (>= 8 pairs: the batch searches by wave-cooperative tiles first, the registration object - a batch of one - per lane)
    srcs = np.zeros((nS, cap, 4), np.float32)
    tgts = np.zeros((nT, cap, 4), np.float32)
    shifts = [np.eye(4, dtype=np.float32), synth.se3(0.2, -0.1, 0, 0, 0, 0.01).astype(np.float32)]
    nt, ns = [], []
    for j in range(nT):
        tj = O.transform(t, shifts[j])
        tgts[j, : len(tj)] = tj
        nt.append(len(tj))
    for i in range(nS):
        si = s[: len(s) - 13 * i]
        srcs[i, : len(si)] = si
        ns.append(len(si))
    ns[5] = 0  # an empty source
    srcs[6, : ns[6], :3] += 1000.0  # nothing within 2 m: "Need at least 4 points to estimate a transform"
    tgt_of_src = [0, 1, 0, 1, 1, 0, 0, 1, 0]
    guesses = np.tile(np.eye(4, dtype=np.float32), (nS, 1, 1))
    guesses[2] = synth.se3(0, 0, 0, 0, 0, 0.01)  # (PCL 1.8's GICP composes a guess rotation-only-safely: tests/test_gpu_gicp_bfgs.py)
    kw = dict(max_iterations=max_iterations)
    b = R.BatchRegistration(nS, nT, cap, method="GICP", **kw)
    b.set_targets(tgts, nt)
    b.set_sources(srcs, ns)
    b.align(tgt_of_src, guesses=guesses, with_fitness=True)
    res, fit = b.results(with_fitness=True)
    single = R.GeneralizedIterativeClosestPoint(**kw)
    for i in (0, 1, 2, 3, 4, 7, 8):
        si, ti = srcs[i, : ns[i]], tgts[tgt_of_src[i], : nt[tgt_of_src[i]]]
        single.setInputTarget(ti)
        single.setInputSource(si)
        single.align(guesses[i])
        ref = O.gicp_bfgs_align(si, ti, O.gicp_bfgs_params(**kw), guess=guesses[i])
        r = res[i]
        assert r["converged"] and single.hasConverged() and ref["converged"], i
        # the same kernels, the same scalar code, the same sorted clouds and the same blocks per pair: the registration object IS a
        # batch of one
        dt, dr = synth.pose_error(r["T"], single.getFinalTransformation())
        assert dt <= 1e-6 and dr <= 1e-7, (i, dt, dr)
        assert r["iterations"] == single._result["iterations"], (i, r["iterations"], single._result["iterations"])
        dt, dr = synth.pose_error(r["T"], ref["T"])
        if max_iterations == 1:  # one outer iteration = one full BFGS minimisation on the same correspondences
            # (pair 0 - no shift, no guess - is the case tests/test_gpu_gicp_bfgs.py holds to 2 mm.  From a shifted start one minimisation
            # ends wherever testGradient(1e-2) / the line search's noise floor first stops it: millimetres apart for two correct
            # executions, tests/probes/bfgs_probe.py - the functor-calling host code of round 4 lands on the same digits as this batch)
            tol_t, tol_r = (2e-3, 2e-4) if i == 0 else (1e-2, 1e-3)
            assert r["iterations"] == ref["iterations"] == 1 and dt <= tol_t and dr <= tol_r, (i, dt, dr)
            assert abs(r["error"] - ref["error"]) <= 2e-3 * ref["error"]
        else:  # both stop inside the convergence band (the cost moves the points in fp32: tests/test_gpu_gicp_bfgs.py)
            assert abs(r["iterations"] - ref["iterations"]) <= 2 and dt <= 0.02 and dr <= 2e-3, (i, dt, dr, r["iterations"], ref["iterations"])
        assert r["n_source"] == ns[i] and r["n_target"] == nt[tgt_of_src[i]] and 0 < r["n_inliers"] <= ns[i]
        np.testing.assert_allclose(fit[i], O.fitness(si, ti, r["T"]), rtol=1e-6)
    assert res[5]["iterations"] == 0 and res[5]["n_source"] == 0 and not res[5]["converged"]
    assert res[6]["iterations"] == 0 and not res[6]["converged"] and res[6]["n_inliers"] == 0
    np.testing.assert_array_equal(np.asarray(res[6]["T"]), guesses[6])
    # a second align on the same object is bit-identical (fixed summation order, states reset)
    b.align(tgt_of_src, guesses=guesses, with_fitness=False)
    res2 = b.results()
    for i in range(5):
        np.testing.assert_array_equal(np.asarray(res2[i]["T"]), np.asarray(res[i]["T"]))
        assert res2[i]["iterations"] == res[i]["iterations"]
    b.close()


def test_batch_gicp_rejects_unbuilt_options(R):
    from lidar_graph_slam_amd import _lib as L

    with pytest.raises(L.Mi355Error):
        R.BatchRegistration(2, 2, 1024, method="GICP", ransac_iterations=3)


def test_multi_entry_takes_gicp(small_pair):
    """mi355_multi_batch_register with MI355_METHOD_GICP: two lanes on device 0, pairs dealt by sub-map, the same poses as one batch."""
    from lidar_graph_slam_amd.multi import MultiDevice
    from lidar_graph_slam_amd.registration import BatchRegistration

    s, t, _ = small_pair
    targets = [t, O.transform(t, synth.se3(0.1, 0.05, 0, 0, 0, 0.004).astype(np.float32))]
    sources = [s, s[:-50], s[:-100], s[:-150]]
    tgt_of_src = np.array([0, 1, 1, 0])
    md = MultiDevice([0, 0])
    res, _, lane = md.batch_register(sources, targets, tgt_of_src, method="GICP", raw=True, max_iterations=8)
    md.close()
    cap = max(len(s), len(t))
    pk = lambda cl: (np.stack([np.pad(c, ((0, cap - len(c)), (0, 0))) for c in cl]), np.array([len(c) for c in cl], np.int32))  # noqa: E731
    b = BatchRegistration(4, 2, cap, method="GICP", max_iterations=8)
    b.set_targets(*pk(targets))
    b.set_sources(*pk(sources))
    b.align(list(tgt_of_src))
    one = b.results(raw=True)
    b.close()
    assert sorted(set(lane.tolist())) == [0, 1]
    for p in range(4):
        assert res["converged"][p] and res["iterations"][p] == one["iterations"][p]
        np.testing.assert_allclose(res["T"][p], one["T"][p], rtol=0, atol=2e-6)


def test_batch_gicp_on_loop_closure_pairs_lands_where_the_oracle_lands():
    """BASELINE config 5's shape with "GICP" (graph_based_slam.cpp:120-141,297-322): candidates at wrong graph poses against 0.5 m
    voxel-filtered five-key-frame sub-maps, identity guess.  pclomp's GICP optimises (translation, Euler angles about the map ORIGIN):
    near the origin it recovers the truth to centimetres, tens of metres out it stops short - the restated reference (the oracle)
    exactly like the batch, which is what this test pins (benchmarks/config5.py --method GICP therefore runs on sub-maps near the
    origin)."""
    from lidar_graph_slam_amd import loop_batch
    from lidar_graph_slam_amd.registration import BatchRegistration

    S = 3  # sub-maps at x = 0, 20, 40 m
    length = 20.0 * S + 20.0

    def place(c, T):
        o = c.copy()
        o[:, :3] = (c[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        return o

    pre = lambda r: O.voxel_grid(O.distance_filter(r, 1.0), 0.2)[0]  # noqa: E731
    sub_pose = lambda s: synth.se3(20.0 * s, 0.0, 0.0)  # noqa: E731
    sources, submaps, expected = [], [], []
    for s in range(S):
        true_off = synth.random_perturbation(np.random.default_rng(1000 + s * 64), 1.0, 2.0)
        wrong = synth.random_perturbation(np.random.default_rng(5 + s), 1.5, 4.0)
        cand = pre(synth.cast_job((length, sub_pose(s) @ true_off, 9000 + s * 64, 512, 30.0)))
        kf = [pre(synth.cast_job((length, sub_pose(s) @ synth.se3(1.0 * j, 0, 0), 500 + 10 * s + j, 512, 30.0))) for j in range(-2, 3)]
        submaps.append(np.concatenate([place(kf[j + 2], sub_pose(s) @ synth.se3(1.0 * j, 0, 0)) for j in range(-2, 3)], axis=0))
        true_pose = sub_pose(s) @ true_off
        graph_pose = true_pose @ wrong
        sources.append(place(cand, graph_pose))
        expected.append(true_pose @ np.linalg.inv(graph_pose))
    cap = max(max(len(c) for c in sources), max(len(c) for c in submaps))
    pk = lambda cl: (np.stack([np.pad(c, ((0, cap - len(c)), (0, 0))) for c in cl]), np.array([len(c) for c in cl], np.int32))  # noqa: E731
    kw = dict(max_iterations=100, max_correspondence_distance=2.0, max_optimizer_iterations=20, transformation_epsilon=0.01)  # yaml:11-12,24-29
    b = BatchRegistration(S, S, cap, method="GICP", euclidean_fitness_epsilon=1e-8, **kw)
    b.set_targets(*pk(submaps), loop_batch.voxel_prefilter(0.5))
    b.set_sources(*pk(sources))
    b.align(list(range(S)), with_fitness=True)
    res, fit = b.results(with_fitness=True)
    b.close()
    for s in range(S):
        t = O.voxel_grid(submaps[s], 0.5)[0]
        ref = O.gicp_bfgs_align(sources[s], t, O.gicp_bfgs_params(**kw))
        assert res[s]["n_target"] == len(t) and res[s]["converged"] == ref["converged"]
        dt, dr = synth.pose_error(res[s]["T"], ref["T"])
        et, er = synth.pose_error(res[s]["T"], expected[s])
        ot, orr = synth.pose_error(ref["T"], expected[s])
        print(f"sub-map at x = {20 * s} m: GPU vs oracle {dt:.4f} m / {dr:.1e} rad, iterations {res[s]['iterations']} / {ref['iterations']}; vs the truth GPU {et:.3f} m, oracle {ot:.3f} m")
        # two correct executions of this engine stop inside a band of each other (tests/test_gpu_gicp_bfgs.py); what is asserted is that
        # the GPU is as good as the oracle against the truth, not worse by more than that band (which widens with the distance from the
        # origin as the engine's own error does: 40 m out both stop ~6 cm from the truth and from each other, an outer iteration apart)
        assert abs(res[s]["iterations"] - ref["iterations"]) <= 2 and dt <= 0.04 + 0.5 * ot and dr <= 4e-3, (s, dt, dr, ot)
        assert et <= ot + 0.04 + 0.5 * ot, (s, et, ot)
        np.testing.assert_allclose(fit[s], O.fitness(sources[s], t, res[s]["T"]), rtol=1e-6)
