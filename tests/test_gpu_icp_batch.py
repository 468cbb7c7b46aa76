"""GPU parity: pcl::IterativeClosestPoint for many pairs at once (MI355_METHOD_ICP in mi355_batch_create, round 5) — the back end's
default engine (graph_based_slam.param.yaml:9, graph_based_slam.cpp:142-151) with every pair's fit and PCL's convergence criteria
advanced on the device — against the registration object (a batch of one, the source in the caller's order) and the oracle."""
import numpy as np
import parity
import pytest

from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from lidar_graph_slam_amd import registration

    return registration


@pytest.mark.parametrize("max_iterations", [6, 100])
def test_batch_icp_matches_single_and_oracle(R, small_pair, max_iterations):
    s, t, T = small_pair
    cap = max(len(s), len(t)) + 7
    nS, nT = 9, 2  # (>= 8 pairs: the batch searches by wave-cooperative tiles while a pair moves, the registration object - a batch of one - per lane)
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
    srcs[6, : ns[6], :3] += 1000.0  # nothing within 30 m: "Not enough correspondences found"
    tgt_of_src = [0, 1, 0, 1, 1, 0, 0, 1, 0]
    guesses = np.tile(np.eye(4, dtype=np.float32), (nS, 1, 1))
    guesses[2] = (T @ synth.se3(0.05, -0.03, 0.0, 0, 0, 0.004)).astype(np.float32)
    guesses[3] = synth.se3(0.5, -0.1, 0, 0, 0, -0.01)
    kw = dict(max_iterations=max_iterations)
    b = R.BatchRegistration(nS, nT, cap, method="ICP", **kw)
    b.set_targets(tgts, nt)
    b.set_sources(srcs, ns)
    b.align(tgt_of_src, guesses=guesses, with_fitness=True)
    res, fit = b.results(with_fitness=True)
    single = R.IterativeClosestPoint(**kw)
    for i in (0, 1, 2, 3, 4, 7, 8):
        si, ti = srcs[i, : ns[i]], tgts[tgt_of_src[i], : nt[tgt_of_src[i]]]
        single.setInputTarget(ti)
        single.setInputSource(si)
        single.align(guesses[i])
        ref = O.icp_align(si, ti, O.icp_params(**kw), guess=guesses[i])
        r = res[i]
        assert r["converged"] and single.hasConverged() and ref["converged"], i
        # the same kernels and the same scalar code: the batch differs from the registration object in the order of the source
        # points (Morton-sorted against the caller's) only, i.e. in the order of its fp64 sums
        dt, dr = synth.pose_error(r["T"], single.getFinalTransformation())
        assert dt <= 1e-6 and dr <= 1e-7, (i, dt, dr)
        assert r["iterations"] == single.result["iterations"], (i, r["iterations"], single.result["iterations"])
        dt, dr = synth.pose_error(r["T"], ref["T"])
        if max_iterations == 6:  # fixed iteration count: iteration by iteration the oracle's pose
            assert r["iterations"] == ref["iterations"] and dt <= 2e-5 and dr <= 2e-6, (i, dt, dr)
        else:  # PCL's stopping rules: the stopping iteration hinges on the last bits of the mean squared distance - SURVEY 8(c)'s band,
            parity.assert_in_band(r["T"], ref["T"], f"icp_batch/pair{i}")  # the measured worst case in profiles/rNN_parity_deltas.txt
        assert r["n_source"] == ns[i] and r["n_target"] == nt[tgt_of_src[i]]
        np.testing.assert_allclose(fit[i], O.fitness(si, ti, r["T"]), rtol=1e-6)
    assert res[5]["iterations"] == 0 and res[5]["n_source"] == 0 and not res[5]["converged"]
    assert res[6]["iterations"] == 0 and not res[6]["converged"]
    np.testing.assert_array_equal(np.asarray(res[6]["T"]), guesses[6])
    # a second align on the same object (identity guesses) is bit-identical for the pairs that had the identity guess already
    b.align(tgt_of_src, with_fitness=False)
    res2 = b.results()
    for i in (0, 1, 4):
        np.testing.assert_array_equal(np.asarray(res2[i]["T"]), np.asarray(res[i]["T"]))
    b.close()


def test_batch_icp_and_fitness_on_sparse_targets_with_fewer_hash_levels(R):
    """a sparse target (points ~9 m apart: every point alone in its cell at the fine levels) makes the device-sized voxel hash
    keep fewer coarse levels, so ICP's 30 m correspondence search and the fitness query walk fine cells shell by shell - and since round
    6 the first iterations go through the wave-cooperative tiles, whose boxes then hold hundreds of cells.  Twelve pairs (>= 8: the tile
    path), fixed iteration count: iteration by iteration the oracle's poses, the oracle's fitness - and the batch must not take seconds."""
    import time

    rng = np.random.default_rng(91)
    nT, nS, n = 3, 12, 3000
    cap = n + 5
    tgts = np.zeros((nT, cap, 4), np.float32)
    for j in range(nT):
        tgts[j, :n, :3] = rng.uniform([-125, -125, -20], [125, 125, 20], (n, 3))
    srcs = np.zeros((nS, cap, 4), np.float32)
    ns, tgt_of_src, truth = [], [], []
    for i in range(nS):
        j = i % nT
        m = n - 37 * i
        pick = rng.choice(n, m, replace=False)
        Tt = synth.se3(0.4 + 0.05 * i, -0.3, 0.1, 0.0, 0.0, 0.004 * (i + 1))
        src = O.transform(tgts[j, pick], np.linalg.inv(Tt).astype(np.float32))
        src[:, :3] += rng.normal(0, 0.01, (m, 3)).astype(np.float32)
        srcs[i, :m] = src
        ns.append(m)
        tgt_of_src.append(j)
        truth.append(Tt)
    kw = dict(max_iterations=8)
    b = R.BatchRegistration(nS, nT, cap, method="ICP", **kw)
    b.set_targets(tgts, [n] * nT)
    b.set_sources(srcs, ns)
    b.align(tgt_of_src, with_fitness=True)
    b.results(with_fitness=True)  # (warm: allocations)
    t0 = time.perf_counter()
    b.set_targets(tgts, [n] * nT)
    b.set_sources(srcs, ns)
    b.align(tgt_of_src, with_fitness=True)
    res, fit = b.results(with_fitness=True)
    seconds = time.perf_counter() - t0
    for i in range(nS):
        si, ti = srcs[i, : ns[i]], tgts[tgt_of_src[i], :n]
        ref = O.icp_align(si, ti, O.icp_params(**kw))
        dt, dr = synth.pose_error(res[i]["T"], ref["T"])
        assert res[i]["iterations"] == ref["iterations"] and dt <= 2e-5 and dr <= 2e-6, (i, dt, dr)
        assert synth.pose_error(res[i]["T"], truth[i])[0] < 0.05
        np.testing.assert_allclose(fit[i], O.fitness(si, ti, res[i]["T"]), rtol=1e-6)
    assert seconds < 1.0, seconds  # (a performance cliff here would be seconds: 12 pairs x 8 iterations x 3 000 points)
    print(f"sparse-target ICP batch: {seconds * 1e3:.1f} ms")
    b.close()
