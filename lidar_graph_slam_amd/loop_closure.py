"""Python handle on LoopClosureCore (csrc/host/loop_closure_core.cpp): the ROS-free, batched registration half of
GraphBasedSLAM::optimization_callback (graph_based_slam.cpp:238-341).  The policy itself is C++, like the node it mirrors;
this file only marshals."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L


class LoopClosure:
    """Parameters are the back end's ROS parameters (graph_based_slam.param.yaml:3-23); the 0.5 m sub-map leaf is the
    constant of graph_based_slam.cpp:61.  Key frames and the candidate gate live on the host: no GPU is needed before close()."""

    def __init__(self, search_key_frame_num=20, score_threshold=0.3, search_for_candidate_threshold=15.0, accumulate_distance_threshold=100.0,
                 submap_leaf=0.5, max_iteration=100, transformation_epsilon=0.01, max_correspondence_distance=2.0, correspondence_randomness=20,
                 registration_method="ICP", step_size=0.1, ndt_resolution=1.0, correspondence_distance=2.0, max_optimizer_iteration=20,
                 ransac_iteration=0, device_id=0, device_ids=None):
        self._lib = L.lib()
        # graph_based_slam.param.yaml:9 defaults to "ICP"; every engine ("ICP" / "FAST_GICP" / "NDT_OMP" / "GICP" = pclomp's BFGS GICP)
        # goes through one batch per lane.
        methods = {"FAST_GICP": L.METHOD_FAST_GICP, "NDT_OMP": L.METHOD_NDT_OMP, "ICP": L.METHOD_ICP, "GICP": L.METHOD_GICP}
        if registration_method not in methods:
            raise L.Mi355Error(L.EINVAL, f"registration_method {registration_method!r}: FAST_GICP, NDT_OMP, ICP or GICP")
        self.params = L.LoopClosureParams(search_key_frame_num, score_threshold, search_for_candidate_threshold, accumulate_distance_threshold,
                                          submap_leaf, max_iteration, transformation_epsilon, max_correspondence_distance, correspondence_randomness,
                                          methods[registration_method], step_size, ndt_resolution, correspondence_distance, max_optimizer_iteration,
                                          ransac_iteration)
        # device_ids: one lane per entry (mi355_loop_closure_create_multi); a close() deals its candidates to the lanes by sub-map
        ids = np.ascontiguousarray([device_id] if device_ids is None else list(device_ids), dtype=np.int32)
        self.handle = self._lib.mi355_loop_closure_create_multi(ids.ctypes.data, len(ids), C.byref(self.params))
        if not self.handle:
            raise L.Mi355Error(L.EINVAL, "mi355_loop_closure_create: bad parameters (only ransac_iteration = 0 is built)")

    def _check(self, rc):
        if rc != L.OK:
            raise L.Mi355Error(rc, self._lib.mi355_loop_closure_last_error(self.handle).decode())

    @staticmethod
    def _pose7(position, quaternion_xyzw):
        return np.ascontiguousarray(np.concatenate([np.asarray(position, np.float64), np.asarray(quaternion_xyzw, np.float64)]))

    def add_key_frame(self, cloud, position, quaternion_xyzw, accum_distance) -> int:
        v, keep = L.view(cloud)
        p = self._pose7(position, quaternion_xyzw)
        kid = self._lib.mi355_loop_closure_add_key_frame(self.handle, C.byref(v), p.ctypes.data, float(accum_distance))
        if kid < 0:
            raise L.Mi355Error(L.EINVAL, self._lib.mi355_loop_closure_last_error(self.handle).decode())
        return kid

    def set_pose(self, key_frame_id, position, quaternion_xyzw):
        p = self._pose7(position, quaternion_xyzw)
        self._check(self._lib.mi355_loop_closure_set_pose(self.handle, int(key_frame_id), p.ctypes.data))

    def candidate(self, query_id) -> int:
        c = C.c_int(-1)
        self._check(self._lib.mi355_loop_closure_candidate(self.handle, int(query_id), C.byref(c)))
        return c.value

    def close_loops(self, query_ids) -> list[dict]:
        q = np.ascontiguousarray(query_ids, dtype=np.int32)
        res = (L.LoopResult * max(len(q), 1))()
        self._check(self._lib.mi355_loop_closure_close(self.handle, q.ctypes.data, len(q), res))
        return [{
            "query_id": r.query_id, "candidate_id": r.candidate_id, "converged": bool(r.converged), "accepted": bool(r.accepted),
            "iterations": r.iterations, "fitness": r.fitness, "T": np.array(r.transform, dtype=np.float32).reshape(4, 4),
            "between": np.array(r.between, dtype=np.float64).reshape(4, 4), "target_points": r.target_points,
        } for r in res[: len(q)]]

    def close(self):
        if getattr(self, "handle", None):
            self._lib.mi355_loop_closure_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may be gone already: leave the handles to the process teardown
            return
        try:
            self.close()
        except Exception:
            pass
