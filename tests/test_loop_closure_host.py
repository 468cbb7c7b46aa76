"""Host logic of LoopClosureCore through the C-ABI (no GPU is touched before close()): the candidate gate of
graph_based_slam.cpp:256-280 against an independent restatement, parameter defaults against graph_based_slam.param.yaml."""
import numpy as np

from lidar_graph_slam_amd import _lib as L
from lidar_graph_slam_amd.loop_closure import LoopClosure
from ref_loop_closure import RefLoopClosure


def test_default_params_match_reference_yaml():
    p = L.LoopClosureParams()
    L.lib().mi355_loop_closure_default_params(p)
    # graph_based_slam.param.yaml:4,6-8,12,14,21,23 and the leaf of graph_based_slam.cpp:61
    assert (p.search_key_frame_num, p.score_threshold, p.search_for_candidate_threshold, p.accumulate_distance_threshold) == (20, 0.3, 15.0, 100.0)
    assert (p.submap_leaf, p.max_iteration, p.transformation_epsilon, p.max_correspondence_distance, p.correspondence_randomness) == (0.5, 100, 0.01, 2.0, 20)
    # yaml:9 registration_method "ICP", yaml:17-18 step_size / ndt_resolution
    assert (p.registration_method, p.step_size, p.ndt_resolution) == (L.METHOD_ICP, 0.1, 1.0)


def test_unknown_registration_method_is_refused():
    import pytest

    with pytest.raises(L.Mi355Error):
        LoopClosure(registration_method="GICP_OMP")  # get_registration leaves the pointer null for unknown strings (:153)
    LoopClosure(registration_method="GICP")  # pclomp's BFGS GICP (graph_based_slam.cpp:120-141)


def test_candidate_gate_matches_restatement():
    rng = np.random.default_rng(3)
    kw = dict(search_for_candidate_threshold=12.0, accumulate_distance_threshold=40.0)
    lc, ref = LoopClosure(**kw), RefLoopClosure(**kw)
    # a random walk that crosses itself: positions in a 40 m box, accumulated distance growing
    pos = np.cumsum(rng.normal(0, 2.0, size=(120, 3)) * [1, 1, 0.05], axis=0)
    accum = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(pos, axis=0), axis=1))])
    cloud = np.zeros((3, 4), np.float32)
    q_id = np.array([0.0, 0.0, 0.0, 1.0])
    for p, a in zip(pos, accum):
        assert lc.add_key_frame(cloud, p, q_id, a) == ref.add_key_frame(cloud, p, q_id, a)
    got = [lc.candidate(q) for q in range(len(pos))]
    want = [ref.candidate(q) for q in range(len(pos))]
    assert got == want
    assert any(c >= 0 for c in got) and any(c < 0 for c in got)
    # moving a key frame moves the gate's answer with it (adjust_pose, :417-432)
    q = max(i for i, c in enumerate(got) if c >= 0)
    far = pos[got[q]] + [500.0, 0, 0]
    lc.set_pose(got[q], far, q_id)
    ref.kf[got[q]] = (ref.kf[got[q]][0], far, ref.kf[got[q]][2], ref.kf[got[q]][3])
    assert lc.candidate(q) == ref.candidate(q) != got[q]
    # queries without a candidate produce no registration and need no GPU
    none = [i for i, c in enumerate(got) if c < 0][:3]
    res = lc.close_loops(none)
    assert [r["candidate_id"] for r in res] == [-1] * len(none) and not any(r["accepted"] for r in res)
