"""Multi-device below the C-ABI (VERDICT r2 missing 1): mi355_multi_batch_register and the multi-lane
LoopClosureCore shard a candidate list by sub-map over the lanes of ONE process.  A lane is a context + stream + batch object
on a device; the box has one GPU, so the two-lane case puts both lanes on device 0 — the host threads, the partition, the
per-lane uploads and the gather are the ones an 8-GPU node runs.  Results must be those of a single batch, pair for pair."""
import numpy as np
import pytest

from lidar_graph_slam_amd import _lib as L
from lidar_graph_slam_amd import synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _candidates(n_targets=3, n_sources=10):
    scene = synth.Scene(1234)
    pf = lambda c: O.voxel_grid(O.distance_filter(c, 1.0), 0.2)[0]  # noqa: E731
    targets, sources, tgt_of_src, truth = [], [], [], []
    poses = [synth.se3(6.0 * t, 0.3 * t, 0, 0, 0, 0.02 * t) for t in range(n_targets)]
    for t, P in enumerate(poses):
        c = pf(synth.cast_scan(scene, P, n_azimuth=256, seed=10 + t))
        targets.append(c)
    rng = np.random.default_rng(3)
    for s in range(n_sources):
        t = int(rng.integers(0, n_targets)) if s else n_targets - 1  # skewed on purpose
        d = synth.random_perturbation(np.random.default_rng(100 + s), 0.4, 1.5)
        sources.append(pf(synth.cast_scan(scene, poses[t] @ d, n_azimuth=256, seed=50 + s)))
        tgt_of_src.append(t)
        truth.append(d)
    return sources, targets, np.asarray(tgt_of_src), truth


@pytest.mark.parametrize("method", ["FAST_GICP", "NDT_OMP", "ICP"])
def test_sharded_registration_equals_the_single_batch(method):
    from lidar_graph_slam_amd.multi import MultiDevice, shard_by_target
    from lidar_graph_slam_amd.registration import BatchRegistration

    sources, targets, tgt_of_src, truth = _candidates()
    kw = {"FAST_GICP": dict(max_iterations=30), "NDT_OMP": dict(max_iterations=30, resolution=1.0), "ICP": dict(max_iterations=40)}[method]
    cap = max(len(c) for c in sources + targets)
    S, U = len(sources), len(targets)
    pk = lambda cl: (np.stack([np.pad(c, ((0, cap - len(c)), (0, 0))) for c in cl]), np.array([len(c) for c in cl], np.int32))  # noqa: E731
    b = BatchRegistration(S, U, cap, method=method, **kw)
    b.set_targets(*pk(targets))
    b.set_sources(*pk(sources))
    b.align(tgt_of_src, with_fitness=True)
    single, single_fit = b.results(with_fitness=True)
    b.close()
    for lanes in ([0], [0, 0], [0, 0, 0]):
        md = MultiDevice(lanes)
        assert md.n_devices == len(lanes)
        for call in range(2):  # the second call reuses every lane's batch object
            res, fit, lane = md.batch_register(sources, targets, tgt_of_src, method=method, **kw)
            # ... really: allocated once per busy lane, whatever the padding bytes of the parameter structs hold
            assert [md.lane_allocations(i) for i in range(len(lanes))] == [1 if i in set(lane.tolist()) else 0 for i in range(len(lanes))], call
            np.testing.assert_array_equal(lane, shard_by_target(tgt_of_src, len(lanes)))
            assert len(set(lane.tolist())) == min(len(lanes), U)  # every lane that can have work has some
            for p in range(S):
                assert res[p]["status"] == 0 and res[p]["converged"] == single[p]["converged"]
                assert res[p]["iterations"] == single[p]["iterations"], (lanes, p)
                assert res[p]["n_source"] == len(sources[p]) and res[p]["n_target"] == len(targets[tgt_of_src[p]])
                # the same kernels on the same clouds: what differs is the slot a pair occupies
                np.testing.assert_allclose(res[p]["T"], single[p]["T"], rtol=0, atol=2e-6)
                np.testing.assert_allclose(fit[p], single_fit[p], rtol=1e-9)
        # fewer candidates than before, one sub-map only: the grown batch runs its live slots only
        sub = [p for p in range(S) if tgt_of_src[p] == tgt_of_src[0]]
        res, fit, lane = md.batch_register([sources[p] for p in sub], [targets[tgt_of_src[0]]], np.zeros(len(sub), np.int32), method=method, **kw)
        assert len(set(lane.tolist())) == 1
        assert max(md.lane_allocations(i) for i in range(len(lanes))) == 1  # a smaller request fits the grown batch
        for k, p in enumerate(sub):
            np.testing.assert_allclose(res[k]["T"], single[p]["T"], rtol=0, atol=2e-6)
        md.close()
    dt = [synth.pose_error(np.asarray(single[p]["T"]), truth[p])[0] for p in range(S)]
    assert np.median(dt) < (0.05 if method == "FAST_GICP" else 0.25)  # and the registrations are registrations


def test_bad_arguments_are_refused():
    from lidar_graph_slam_amd.multi import MultiDevice

    with pytest.raises(L.Mi355Error):
        MultiDevice([0, 99])  # no such device
    md = MultiDevice([0])
    c = np.zeros((50, 4), np.float32)
    with pytest.raises(L.Mi355Error):
        md.batch_register([c, c], [c], [0, 1])  # tgt_of_src out of range
    with pytest.raises(L.Mi355Error):
        md.batch_register([c, c], [c], None)  # mapping required


@pytest.mark.parametrize("method", ["FAST_GICP", "ICP"])
def test_multi_lane_loop_closure_equals_single_lane(method):
    """LoopClosureCore over two lanes (mi355_loop_closure_create_multi): candidates dealt by sub-map, every lane assembles,
    voxel-filters and registers its own - the results are the single-lane ones, query for query."""
    from lidar_graph_slam_amd.loop_closure import LoopClosure
    from ref_scan_matcher import matrix_to_pose
    from test_gpu_loop_closure import _out_and_back

    scene = synth.Scene(1234)
    truth = _out_and_back()
    kw = dict(search_key_frame_num=2, search_for_candidate_threshold=6.0, accumulate_distance_threshold=50.0, max_iteration=30, registration_method=method)
    one, two = LoopClosure(**kw), LoopClosure(device_ids=[0, 0], **kw)
    accum = 0.0
    for i, T in enumerate(truth):
        cloud = O.distance_filter(synth.cast_scan(scene, T, n_rings=32, n_azimuth=360, noise_sigma=0.01, seed=100 + i), 1.0)
        if i:
            accum += float(np.linalg.norm(truth[i][:3, 3] - truth[i - 1][:3, 3]))
        drift = synth.se3(0.25, -0.15, 0.0, 0.0, 0.0, 0.01) if i >= len(truth) // 2 else np.eye(4)
        p, q = matrix_to_pose((drift @ T).astype(np.float32))
        assert one.add_key_frame(cloud, p, q, accum) == two.add_key_frame(cloud, p, q, accum) == i
    queries = list(range(len(truth) - 6, len(truth))) + [3]
    a, b = one.close_loops(queries), two.close_loops(queries)
    assert len({g["candidate_id"] for g in a if g["candidate_id"] >= 0}) >= 2  # there is something to deal
    for g, h in zip(a, b):
        assert (g["query_id"], g["candidate_id"], g["converged"], g["accepted"], g["iterations"], g["target_points"]) == \
               (h["query_id"], h["candidate_id"], h["converged"], h["accepted"], h["iterations"], h["target_points"])
        np.testing.assert_allclose(h["T"], g["T"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(h["fitness"], g["fitness"], rtol=1e-9)
        np.testing.assert_allclose(h["between"], g["between"], rtol=0, atol=1e-5)
    # a second call on both: lanes keep their key frames and batch objects
    b2 = two.close_loops(queries[:3])
    for g, h in zip(a[:3], b2):
        np.testing.assert_allclose(h["T"], g["T"], rtol=0, atol=2e-6)
