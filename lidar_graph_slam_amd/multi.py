"""Python handle on mi355_multi_* (include/mi355_lidar.h, csrc/host/multi.cpp): sharded batch registration over several
devices from ONE process — one context + stream + host thread per device, pairs dealt to devices by sub-map, results gathered
by D2H.  The logic is C++ below the C-ABI (the loop-closure caller is a C++ node); this file only marshals.  The
one-process-per-GPU form of the same partition, with the results all-gathered over RCCL, is sharding.py / loop_batch.py."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L


def shard_by_target(tgt_of_src, n_lanes: int) -> np.ndarray:
    """lane_of_pair for every pair (the C++ rule; equals sharding.shard_by_target)."""
    t = np.ascontiguousarray(tgt_of_src, dtype=np.int32)
    out = np.empty(len(t), dtype=np.int32)
    rc = L.lib().mi355_multi_shard_by_target(t.ctypes.data, len(t), int(n_lanes), out.ctypes.data)
    if rc != L.OK:
        raise L.Mi355Error(rc, "mi355_multi_shard_by_target")
    return out


def _method_and_params(method, params):
    """(method code, parameter struct) of one of the four engines graph_based_slam.cpp:80-152 builds."""
    makers = {"FAST_GICP": (L.METHOD_FAST_GICP, L.gicp_params), "NDT_OMP": (L.METHOD_NDT_OMP, L.ndt_params), "ICP": (L.METHOD_ICP, L.icp_params),
              "GICP": (L.METHOD_GICP, L.gicp_bfgs_params)}
    if method not in makers:
        raise L.Mi355Error(L.EINVAL, f"registration method {method!r}: FAST_GICP, NDT_OMP, ICP or GICP")
    code, make = makers[method]
    return code, make(**params)


class MultiDevice:
    def __init__(self, device_ids=(0,)):
        self._lib = L.lib()
        ids = np.ascontiguousarray(list(device_ids), dtype=np.int32)
        self.handle = self._lib.mi355_multi_create(ids.ctypes.data, len(ids))
        if not self.handle:
            raise L.Mi355Error(L.EHIP, f"mi355_multi_create({list(device_ids)}) failed: no such HIP device (there is no CPU fallback)")
        self.n_devices = self._lib.mi355_multi_device_count(self.handle)

    def batch_register(self, sources, targets, tgt_of_src=None, guesses=None, *, method="FAST_GICP", source_prefilter=None,
                       target_prefilter=None, with_fitness=True, raw=False, **params):
        """sources / targets: lists of float32 [n,4] (or [n,8] PointXYZI-shaped) clouds.  Returns (list of result dicts - or, raw=True,
        ONE record array over the C structs -, fitness array or None, lane_of_pair)."""
        m, p = _method_and_params(method, params)
        sv = (L.CloudView * len(sources))()
        tv = (L.CloudView * len(targets))()
        keep = []
        for arr, views in ((sources, sv), (targets, tv)):
            for i, c in enumerate(arr):
                v, k = L.view(c)
                views[i] = v
                keep.append(k)
        t = None if tgt_of_src is None else np.ascontiguousarray(tgt_of_src, dtype=np.int32)
        g = None if guesses is None else np.ascontiguousarray(guesses, dtype=np.float32).reshape(len(sources), 16)
        res = (L.Result * len(sources))()
        fit = np.zeros(len(sources), dtype=np.float64)
        lane = np.full(len(sources), -1, dtype=np.int32)
        rc = self._lib.mi355_multi_batch_register(
            self.handle, m, C.byref(p), sv, len(sources), tv, len(targets), None if t is None else t.ctypes.data, None if g is None else g.ctypes.data,
            None if source_prefilter is None else C.byref(source_prefilter), None if target_prefilter is None else C.byref(target_prefilter),
            1 if with_fitness else 0, res, fit.ctypes.data if with_fitness else None, lane.ctypes.data)
        if rc != L.OK:
            raise L.Mi355Error(rc, self._lib.mi355_multi_last_error(self.handle).decode())
        out = np.frombuffer(res, dtype=L.RESULT_DTYPE, count=len(sources)) if raw else [r.to_dict() for r in res]
        return out, (fit if with_fitness else None), lane

    def lane_context(self, lane: int):
        """A Context object over lane `lane`'s mi355_ctx (owned by the multi object): for device clouds that live on that lane
        (DeviceCloud.upload(cloud, ctx=md.lane_context(i)))."""
        from .context import Context

        h = self._lib.mi355_multi_context(self.handle, int(lane))
        if not h:
            raise L.Mi355Error(L.EINVAL, f"no lane {lane}")
        c = Context.__new__(Context)
        c._lib, c.handle, c.device_id, c._borrowed = self._lib, h, None, True
        return c

    def batch_register_clouds(self, sources, targets, tgt_of_src=None, guesses=None, *, method="FAST_GICP", source_prefilter=None,
                              target_prefilter=None, with_fitness=True, raw=False, **params):
        """The same for DeviceCloud objects that already live on the lanes: source p and its sub-map on the lane
        shard_by_target deals pair p to (nothing crosses PCIe but the results)."""
        m, p = _method_and_params(method, params)
        sh = (C.c_void_p * len(sources))(*[c.handle for c in sources])
        th = (C.c_void_p * len(targets))(*[c.handle for c in targets])
        t = None if tgt_of_src is None else np.ascontiguousarray(tgt_of_src, dtype=np.int32)
        g = None if guesses is None else np.ascontiguousarray(guesses, dtype=np.float32).reshape(len(sources), 16)
        res = (L.Result * len(sources))()
        fit = np.zeros(len(sources), dtype=np.float64)
        lane = np.full(len(sources), -1, dtype=np.int32)
        rc = self._lib.mi355_multi_batch_register_clouds(
            self.handle, m, C.byref(p), sh, len(sources), th, len(targets), None if t is None else t.ctypes.data, None if g is None else g.ctypes.data,
            None if source_prefilter is None else C.byref(source_prefilter), None if target_prefilter is None else C.byref(target_prefilter),
            1 if with_fitness else 0, res, fit.ctypes.data if with_fitness else None, lane.ctypes.data)
        if rc != L.OK:
            raise L.Mi355Error(rc, self._lib.mi355_multi_last_error(self.handle).decode())
        out = np.frombuffer(res, dtype=L.RESULT_DTYPE, count=len(sources)) if raw else [r.to_dict() for r in res]
        return out, (fit if with_fitness else None), lane

    def lane_allocations(self, lane: int) -> int:
        """Times lane `lane`'s batch object has been (re)created: stays 1 while requests fit and keep method / parameters."""
        return int(self._lib.mi355_multi_lane_allocations(self.handle, int(lane)))

    def close(self):
        if getattr(self, "handle", None):
            self._lib.mi355_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may be gone already: leave the handles to the process teardown
            return
        try:
            self.close()
        except Exception:
            pass
