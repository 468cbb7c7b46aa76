"""Python handle on ScanMatcherCore (csrc/host/scan_matcher_core.cpp): the ROS-free LidarScanMatcher::callback_cloud
(lidar_scan_matcher.cpp:122-250).  The policy itself is C++, like the node it mirrors; this file only marshals."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L


class LidarScanMatcher:
    """Parameters are the node's ROS parameters (lidar_scan_matcher.param.yaml:3-19)."""

    def __init__(self, registration_method="NDT_OMP", displacement=1.0, max_scan_accumulate_num=20, max_iteration=64, omp_num_thread=4,
                 transformation_epsilon=0.01, step_size=0.1, ndt_resolution=2.0, max_correspondence_distance=2.0, correspondence_randomness=20,
                 reuse_key_frame_covariances=False, correspondence_distance=5.0, max_optimizer_iteration=20, device_id=0):
        method = {"FAST_GICP": L.METHOD_FAST_GICP, "NDT_OMP": L.METHOD_NDT_OMP, "GICP": L.METHOD_GICP}.get(registration_method, -1)
        self.params = L.ScanMatcherParams(method, displacement, max_scan_accumulate_num, max_iteration, omp_num_thread, transformation_epsilon,
                                          step_size, ndt_resolution, max_correspondence_distance, correspondence_randomness,
                                          1 if reuse_key_frame_covariances else 0, correspondence_distance, max_optimizer_iteration)
        self._lib = L.lib()
        self.handle = self._lib.mi355_scan_matcher_create(device_id, C.byref(self.params))
        if not self.handle:
            raise L.Mi355Error(L.EINVAL, f"mi355_scan_matcher_create failed (method {registration_method!r}; GPU present?)")

    def callback_cloud(self, cloud, base_to_sensor=None) -> dict:
        v, keep = L.view(cloud)
        T = None if base_to_sensor is None else np.ascontiguousarray(base_to_sensor, dtype=np.float32).reshape(16)
        out = L.ScanOutput()
        rc = self._lib.mi355_scan_matcher_callback_cloud(self.handle, C.byref(v), T.ctypes.data if T is not None else None, C.byref(out))
        if rc != L.OK:
            raise L.Mi355Error(rc, self._lib.mi355_scan_matcher_last_error(self.handle).decode())
        return {
            "first_scan": bool(out.first_scan), "converged": bool(out.converged), "new_key_frame": bool(out.new_key_frame),
            "pose": np.array(out.pose, dtype=np.float32).reshape(4, 4), "key_frame_id": int(out.key_frame_id),
            "target_points": int(out.target_points), "iterations": int(out.iterations), "n_key_frames": int(out.n_key_frames),
            "accum_distance": float(out.accum_distance),
        }

    def target_cloud(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._lib.mi355_scan_matcher_target(self.handle, None, 0, C.byref(n))
        out = np.empty((max(n.value, 1), 4), dtype=np.float32)
        rc = self._lib.mi355_scan_matcher_target(self.handle, out.ctypes.data, out.shape[0], C.byref(n))
        assert rc == L.OK
        return out[: n.value]

    def close(self):
        if getattr(self, "handle", None):
            self._lib.mi355_scan_matcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may be gone already: leave the handles to the process teardown
            return
        try:
            self.close()
        except Exception:
            pass
