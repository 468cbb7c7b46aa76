"""GPU parity: the HIP NDT path (VoxelGridCovariance leaves, DIRECT7 derivatives, More-Thuente alignment) vs the oracle."""
import numpy as np
import pytest

from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from lidar_graph_slam_amd import registration

    return registration


@pytest.mark.parametrize("res", [1.0, 2.0])
def test_cells(R, small_pair, res):
    s, t, _ = small_pair
    reg = R.NormalDistributionsTransform(resolution=res)
    reg.setInputTarget(t)
    c = reg.cells()
    ref = O.ndt_cells(t, O.ndt_params(resolution=res))
    np.testing.assert_array_equal(c["idx"], ref["idx"])
    np.testing.assert_array_equal(c["min_b"], ref["min_b"])
    np.testing.assert_array_equal(c["div_b"], ref["div_b"])
    # exactly coplanar voxels have a smallest eigenvalue of +-1e-17 and may be accepted by one side only (see test_oracle_ndt)
    same = c["n"] == ref["n"]
    assert (~same).sum() <= 0.05 * len(same) + 1
    assert ((c["n"][~same] == -1) | (ref["n"][~same] == -1)).all()
    ok = same & (ref["n"] >= 6)
    np.testing.assert_allclose(c["mean"][ok], ref["mean"][ok], atol=1e-12)
    np.testing.assert_allclose(c["cov"][ok], ref["cov"][ok], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(c["icov"][ok], ref["icov"][ok], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("res", [1.0, 2.0])
def test_derivatives(R, small_pair, res):
    s, t, _ = small_pair
    reg = R.NormalDistributionsTransform(resolution=res)
    reg.setInputTarget(t)
    reg.setInputSource(s)
    for p6 in ([0, 0, 0, 0, 0, 0], [0.8, 0.05, 0.01, 0.004, -0.003, 0.03], [1.0, 0.1, 0, 0, 0, 0.035]):
        sc, g, H = reg.derivatives(p6)
        sc2, g2, H2 = O.ndt_derivatives(s, t, p6, O.ndt_params(resolution=res))
        np.testing.assert_allclose(sc, sc2, rtol=1e-4)
        np.testing.assert_allclose(g, g2, rtol=1e-3, atol=1e-3 * np.abs(g2).max())
        np.testing.assert_allclose(H, H2, rtol=1e-3, atol=1e-3 * np.abs(H2).max())
        _, _, H64 = reg.derivatives(p6, hessian_fp64=True)
        _, _, H64r = O.ndt_derivatives(s, t, p6, O.ndt_params(resolution=res), hessian_fp64=True)
        np.testing.assert_allclose(H64, H64r, rtol=1e-3, atol=1e-3 * np.abs(H64r).max())


def test_align(R, medium_pair):
    src, tgt, T = medium_pair
    s = O.voxel_grid(O.distance_filter(src, 1.0), 0.2)[0]
    t = O.voxel_grid(O.distance_filter(tgt, 1.0), 0.2)[0]
    reg = R.NormalDistributionsTransform()
    reg.setTransformationEpsilon(0.01)
    reg.setStepSize(0.1)
    reg.setResolution(2.0)
    reg.setMaximumIterations(64)
    reg.setNeighborhoodSearchMethod("DIRECT7")
    reg.setNumThreads(4)
    reg.setInputTarget(t)
    reg.setInputSource(s)
    reg.align()
    ref = O.ndt_align(s, t, O.ndt_params(resolution=2.0))
    assert reg.hasConverged() and ref["converged"]
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref["T"])
    # both stop inside the step-size convergence band (eps 0.01); fp32 per-point maths may add/remove one line-search trial
    assert dt <= 0.01 and dr <= 2e-3, (dt, dr)
    dt, dr = synth.pose_error(reg.getFinalTransformation(), T)
    assert dt < 0.08 and dr < 2e-3  # NDT accuracy on this scene, see tests/test_oracle_ndt.py
    np.testing.assert_allclose(reg.getFitnessScore(), O.fitness(s, t, reg.getFinalTransformation()), rtol=1e-9)
    # with a guess, 1 m cells (BASELINE config 3 resolution)
    g = synth.se3(0.9, 0.08, 0, 0, 0, np.deg2rad(1.8)).astype(np.float32)
    reg.setResolution(1.0)
    reg.setInputTarget(t)
    reg.align(g)
    ref1 = O.ndt_align(s, t, O.ndt_params(resolution=1.0), guess=g)
    dt, dr = synth.pose_error(reg.getFinalTransformation(), ref1["T"])
    assert reg.hasConverged() and dt <= 0.01 and dr <= 2e-3, (dt, dr)
    assert abs(reg.getTransformationProbability() - ref1["trans_probability"]) < 0.05 * abs(ref1["trans_probability"])


@pytest.mark.parametrize("legacy", [1, 0])
def test_batch_matches_single_and_oracle(R, small_pair, legacy):
    """Batch mode with NDT_OMP (graph_based_slam.cpp:101-119 offers it for loop candidates): every pair's Newton /
    More-Thuente state advances on the device.  Ragged sources, two targets, guesses; both line-search variants (with
    legacy = 0 the interval refinement and the fp64 computeHessian requests run on the device too)."""
    s, t, T = small_pair
    cap = max(len(s), len(t)) + 7
    nS, nT = 6, 2
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
    ns[5] = 0  # an empty source: no derivatives, reported as the reference reports a zero Newton step
    tgt_of_src = [0, 1, 0, 1, 1, 0]
    guesses = np.tile(np.eye(4, dtype=np.float32), (nS, 1, 1))
    guesses[2] = synth.se3(0.3, 0.05, 0, 0, 0, 0.01)
    guesses[3] = synth.se3(0.5, -0.1, 0, 0, 0, -0.01)
    kw = dict(resolution=2.0, step_size=0.1, transformation_epsilon=0.01, max_iterations=64, legacy_mt_skip=legacy)
    b = R.BatchRegistration(nS, nT, cap, method="NDT_OMP", **kw)
    b.set_targets(tgts, nt)
    b.set_sources(srcs, ns)
    b.align(tgt_of_src, guesses=guesses, with_fitness=True)
    res, fit = b.results(with_fitness=True)
    single = R.NormalDistributionsTransform(**kw)
    for i in range(nS - 1):
        si, ti = srcs[i, : ns[i]], tgts[tgt_of_src[i], : nt[tgt_of_src[i]]]
        single.setInputTarget(ti)
        single.setInputSource(si)
        single.align(guesses[i])
        ref = O.ndt_align(si, ti, O.ndt_params(**kw), guess=guesses[i])
        r = res[i]
        assert r["converged"] and single.hasConverged() and ref["converged"], i
        # same state machine, same per-point maths: the batch differs from the single path only in the summation order of
        # the points (Morton-sorted vs caller order) and in device vs host libm for the 6 angles
        dt, dr = synth.pose_error(r["T"], single.getFinalTransformation())
        assert dt <= 2e-3 and dr <= 5e-4, (i, dt, dr)
        assert abs(r["iterations"] - single._result["iterations"]) <= 1, (i, r["iterations"], single._result["iterations"])
        dt, dr = synth.pose_error(r["T"], ref["T"])
        assert dt <= 0.01 and dr <= 2e-3, (i, dt, dr)  # the convergence band of test_align
        assert abs(r["trans_probability"] - ref["trans_probability"]) < 0.05 * abs(ref["trans_probability"])
        assert r["n_source"] == ns[i] and r["n_target"] == nt[tgt_of_src[i]]
        np.testing.assert_allclose(fit[i], O.fitness(si, ti, r["T"]), rtol=1e-9)
    assert res[5]["iterations"] == 0 and res[5]["n_source"] == 0
    # a second align on the same object (new guesses) reuses the targets' leaves
    b.align(tgt_of_src, with_fitness=False)
    res2 = b.results()
    dt, dr = synth.pose_error(res2[0]["T"], res[0]["T"])
    assert dt == 0 and dr == 0  # pairs 0, 1, 4 had the identity guess already: bit-identical rerun
    b.close()


def test_batch_rejects_other_methods(R):
    from lidar_graph_slam_amd import _lib as L

    with pytest.raises(L.Mi355Error):
        R.BatchRegistration(2, 2, 128, method="VGICP")  # the four engines of graph_based_slam.cpp:80-152 and nothing else


def test_batch_keeps_its_own_targets_for_a_later_fitness(R, small_pair):
    """an NDT batch builds the targets' point hash - what getFitnessScore searches - only when mi355_batch_align is
    asked for a fitness.  Device inputs must stay valid until the stream has passed the align that follows them (include/mi355_lidar.h);
    a LATER align with a fitness, after the caller has reused its buffers, must still score against the targets that were set - the batch
    keeps its own copy of targets that came without a prefilter."""
    import torch

    from lidar_graph_slam_amd.context import Context

    s, t, _ = small_pair
    cap = max(len(s), len(t)) + 5
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    kw = dict(resolution=2.0, step_size=0.1, transformation_epsilon=0.01, max_iterations=64)
    tgt_h = np.zeros((2, cap, 4), np.float32)
    src_h = np.zeros((2, cap, 4), np.float32)
    t2 = O.transform(t, synth.se3(0.2, -0.1, 0, 0, 0, 0.01).astype(np.float32))
    tgt_h[0, : len(t)], tgt_h[1, : len(t2)] = t, t2
    src_h[0, : len(s)], src_h[1, : len(s) - 17] = s, s[: len(s) - 17]
    b = R.BatchRegistration(2, 2, cap, ctx=ctx, method="NDT_OMP", **kw)
    tgt_d, src_d = torch.from_numpy(tgt_h).to(dev), torch.from_numpy(src_h).to(dev)
    b.set_targets_dev(tgt_d.data_ptr(), [len(t), len(t2)])
    b.set_sources_dev(src_d.data_ptr(), [len(s), len(s) - 17])
    b.align([0, 1], with_fitness=False)
    first = np.stack([np.asarray(r["T"]) for r in b.results()])
    tgt_d.fill_(1.0e4)  # the caller reuses its buffers (behind the align on the same stream)
    src_d.fill_(-1.0e4)
    torch.cuda.synchronize()
    b.align([0, 1], with_fitness=True)
    res, fit = b.results(with_fitness=True)
    np.testing.assert_array_equal(np.stack([np.asarray(r["T"]) for r in res]), first)
    np.testing.assert_allclose(fit[0], O.fitness(s, t, first[0]), rtol=1e-9)
    np.testing.assert_allclose(fit[1], O.fitness(s[: len(s) - 17], t2, first[1]), rtol=1e-9)
    b.close()
    ctx.set_stream(None)
