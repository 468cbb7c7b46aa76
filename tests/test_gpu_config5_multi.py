"""BASELINE configs[4] at its stated list size through the C-ABI entry built for it (missing 2): 4 096 prefiltered
loop-closure candidates against 64 sub-maps (graph_based_slam.cpp:297-322 generalised from one candidate per tick to a candidate
list) through mi355_multi_batch_register - two and four lanes, all on device 0 (the box has one GPU: the partition by sub-map, the
per-lane host threads, the per-lane uploads through pinned staging and the gather are those of an 8-GPU node).  Results must be
those of ONE BatchRegistration on a 256-pair sample, inside the convergence band of the oracle on four pairs, and the acceptance
rule of graph_based_slam.cpp:321-328 (converged and fitness < score_threshold) must accept what lies near the truth.
The 832 scans behind the list (64 x 8 candidate scans, 64 x 5 key frames) are ray-cast by a pool of CPU workers under a child interpreter."""
import time

import numpy as np
import pytest

import parity
from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

N, S, CASTS = 4096, 64, 8


def _place(c, T):
    o = c.copy()
    o[:, :3] = (c[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    return o


@pytest.fixture(scope="module")
def config5():
    length = 20.0 * S + 20.0
    sub_pose = lambda s: synth.se3(20.0 * s, 0.0, 0.0)  # noqa: E731
    true_off = [synth.random_perturbation(np.random.default_rng(1000 + s * 64 + c), 1.0, 2.0) for s in range(S) for c in range(CASTS)]
    rng = np.random.default_rng(5)
    wrong = [synth.random_perturbation(rng, 1.5, 4.0) for _ in range(N)]
    tgt_of_src = np.arange(N) % S
    cast_of = (np.arange(N) // S) % CASTS
    jobs = [(length, sub_pose(s) @ true_off[s * CASTS + c], 9000 + s * 64 + c, 1024, 30.0) for s in range(S) for c in range(CASTS)]
    jobs += [(length, sub_pose(s) @ synth.se3(1.0 * j, 0.0, 0.0), 500 + 10 * s + j, 1024, 30.0) for s in range(S) for j in range(-2, 3)]
    # the 832 scans are cast by ONE plain child interpreter that owns the worker pool and never touches the GPU; this process - which
    # other tests of the session may have put on the GPU already - starts nothing but that child
    raw = synth.cast_jobs_in_child(jobs)
    import torch  # torch's ROCm runtime before the library's: a module-scoped fixture is built BEFORE the
                  # function-scoped autouse fixture of conftest.py that sees to it, and this one prefilters on the GPU

    if torch.cuda.is_available():
        torch.cuda.init()
    from lidar_graph_slam_amd.filters import PointsPreFiltering

    pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
    scans = [pre.callback_sensor_points(r) for r in raw]
    cand = {(s, c): scans[s * CASTS + c] for s in range(S) for c in range(CASTS)}
    submaps = []
    for s in range(S):
        parts = [_place(scans[S * CASTS + 5 * s + (j + 2)], sub_pose(s) @ synth.se3(1.0 * j, 0.0, 0.0)) for j in range(-2, 3)]
        submaps.append(np.concatenate(parts, axis=0))
    true_pose = lambda p: sub_pose(int(tgt_of_src[p])) @ true_off[int(tgt_of_src[p]) * CASTS + int(cast_of[p])]  # noqa: E731
    graph_pose = lambda p: true_pose(p) @ wrong[p]  # noqa: E731
    sources = [_place(cand[(int(tgt_of_src[p]), int(cast_of[p]))], graph_pose(p)) for p in range(N)]
    expected = [true_pose(p) @ np.linalg.inv(graph_pose(p)) for p in range(N)]
    return sources, submaps, tgt_of_src, expected


@pytest.mark.parametrize("lanes", [2, 4])
def test_config5_at_its_stated_size_through_the_multi_entry(config5, lanes):
    from lidar_graph_slam_amd import loop_batch
    from lidar_graph_slam_amd.multi import MultiDevice, shard_by_target
    from lidar_graph_slam_amd.registration import BatchRegistration

    sources, submaps, tgt_of_src, expected = config5
    pf = loop_batch.voxel_prefilter(0.5)  # graph_based_slam.cpp:61,311-313
    md = MultiDevice([0] * lanes)
    t0 = time.perf_counter()
    res, fit, lane = md.batch_register(sources, submaps, tgt_of_src, target_prefilter=pf, with_fitness=True, raw=True, max_iterations=64)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    res2, fit2, _ = md.batch_register(sources, submaps, tgt_of_src, target_prefilter=pf, with_fitness=True, raw=True, max_iterations=64)
    t_again = time.perf_counter() - t0
    np.testing.assert_array_equal(lane, shard_by_target(tgt_of_src, lanes))
    assert [md.lane_allocations(i) for i in range(lanes)] == [1] * lanes  # the second call re-used every lane's batch object
    np.testing.assert_array_equal(np.asarray(res2["T"]), np.asarray(res["T"]))  # ... and reproduced the first bit for bit
    np.testing.assert_array_equal(fit2, fit)
    md.close()
    print(f"config 5, {lanes} lanes on one MI355X: {N / t_again:.0f} pairs/s including uploads ({t_again * 1e3:.0f} ms; first call {t_first * 1e3:.0f} ms with allocations)")
    assert (np.asarray(res["status"]) == 0).all()
    errs = np.array([synth.pose_error(res["T"][p], expected[p]) for p in range(N)])
    accepted = (np.asarray(res["converged"]) != 0) & (fit < 0.3)  # graph_based_slam.cpp:321-328, score_threshold 0.3
    # (versus the TRUTH: along 1.3 km of street some sub-maps constrain the along-street direction weakly - the oracle lands on the same
    # poses, below; benchmarks/config5.py uses the same 90 % gate)
    assert accepted.sum() >= 0.98 * N and (errs[accepted, 0] < 0.05).mean() >= 0.90 and np.median(errs[:, 0]) < 0.03, (accepted.sum(), np.median(errs[:, 0]))
    # ---- a 256-pair sample, sub-map by sub-map, through ONE BatchRegistration
    sample = np.sort(np.random.default_rng(11).choice(N, 256, replace=False))
    used = sorted({int(tgt_of_src[p]) for p in sample})
    local = {t: k for k, t in enumerate(used)}
    cap = max(max(len(sources[p]) for p in sample), max(len(submaps[t]) for t in used))
    pk = lambda cl: (np.stack([np.pad(c, ((0, cap - len(c)), (0, 0))) for c in cl]), np.array([len(c) for c in cl], np.int32))  # noqa: E731
    b = BatchRegistration(len(sample), len(used), cap, max_iterations=64)
    b.set_targets(*pk([submaps[t] for t in used]), pf)
    b.set_sources(*pk([sources[p] for p in sample]))
    b.align([local[int(tgt_of_src[p])] for p in sample], with_fitness=True)
    one, one_fit = b.results(with_fitness=True, raw=True)
    b.close()
    for k, p in enumerate(sample):
        assert res["iterations"][p] == one["iterations"][k] and res["converged"][p] == one["converged"][k], p
        assert res["n_source"][p] == one["n_source"][k] and res["n_target"][p] == one["n_target"][k], p
        np.testing.assert_allclose(res["T"][p], one["T"][k], rtol=0, atol=2e-6)  # (same kernels, same clouds; the batch size sets the block count per pair)
        np.testing.assert_allclose(fit[p], one_fit[k], rtol=1e-9)
    # ---- against the oracle: same voxel-filtered sub-map, same candidate, identity guess (graph_based_slam.cpp:311-322).  Four fixed
    # pairs, and (two lanes: the oracle takes ~2 s per pair) the EIGHT ACCEPTED PAIRS FARTHEST FROM THE TRUTH plus eight random ones:
    # where the GPU lands decimetres from the truth - sub-maps that constrain the along-street direction weakly - the oracle must
    # land there too, or the misfits are ours.
    picks = [0, 1203, 2777, 4095]
    if lanes == 2:
        acc_idx = np.flatnonzero(accepted)
        picks += [int(p) for p in acc_idx[np.argsort(errs[acc_idx, 0])[-8:]]]
        picks += [int(p) for p in np.random.default_rng(23).choice(N, 8, replace=False)]
    worst_truth = 0.0
    for p in picks:
        t = O.voxel_grid(submaps[int(tgt_of_src[p])], 0.5)[0]
        assert res["n_target"][p] == len(t)
        ref = O.gicp_align(sources[p], t, O.gicp_params(max_iterations=64))
        parity.assert_in_band(res["T"][p], ref["T"], f"config5_multi/{lanes}lanes/pair{p}")
        np.testing.assert_allclose(fit[p], O.fitness(sources[p], t, res["T"][p]), rtol=1e-6)
        worst_truth = max(worst_truth, float(errs[p, 0]))
    print(f"config 5, {lanes} lanes: {len(picks)} pairs inside the oracle's convergence band, the farthest of them {worst_truth:.3f} m from the truth")
