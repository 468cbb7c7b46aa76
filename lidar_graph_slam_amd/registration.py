"""pcl::Registration mirror: the objects `lidar_scan_matcher` / `graph_based_slam` hold as `registration_`
(lidar_scan_matcher.hpp:99, graph_based_slam.hpp:108), running on the MI355X through the C-ABI.

Method names are the ones the reference calls (lidar_scan_matcher.cpp:48-52,65-70,149,162,165-172;
graph_based_slam.cpp:315-322): setInputTarget / setInputSource / align / hasConverged / getFinalTransformation /
getFitnessScore plus the engine-specific setters.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L
from .context import Context, default_context


class DeviceCloud:
    """mi355_cloud: a cloud that lives in HBM (the key-frame ring of SURVEY 8f-2): uploaded once, transformed / concatenated on
    the device (lidar_scan_matcher.cpp:199-210, 275-294), handed to a registration object without crossing PCIe again."""

    def __init__(self, handle, ctx: Context):
        if not handle:
            raise L.Mi355Error(L.EHIP, ctx._lib.mi355_last_error(ctx.handle).decode())
        self.handle, self.ctx = handle, ctx

    @classmethod
    def upload(cls, cloud, ctx: Context | None = None):
        ctx = ctx or default_context()
        v, keep = L.view(cloud)
        return cls(ctx._lib.mi355_cloud_upload(ctx.handle, C.byref(v)), ctx)

    def transformed(self, T):
        T = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        return DeviceCloud(self.ctx._lib.mi355_cloud_transform(self.handle, T.ctypes.data), self.ctx)

    @staticmethod
    def concat(clouds, Ts, carry_covariances=False):
        ctx = clouds[0].ctx
        hs = (C.c_void_p * len(clouds))(*[c.handle for c in clouds])
        T = np.ascontiguousarray(np.asarray(Ts, dtype=np.float32).reshape(len(clouds), 16))
        return DeviceCloud(ctx._lib.mi355_cloud_concat(ctx.handle, hs, T.ctypes.data, len(clouds), 1 if carry_covariances else 0), ctx)

    def prefiltered(self, min_distance=0.0, leaf_size=0.0, mean_k=0, stddev_mul=0.0):
        """PointsPreFiltering's chain on the device: a new DeviceCloud (VoxelGrid alone = graph_based_slam.cpp:311-313)."""
        pf = L.PrefilterParams(float(min_distance), float(leaf_size), int(mean_k), float(stddev_mul))
        return DeviceCloud(self.ctx._lib.mi355_cloud_prefilter(self.handle, C.byref(pf)), self.ctx)

    def compute_covariances(self, k=20, method=L.REG_PLANE):
        self.ctx.check(self.ctx._lib.mi355_cloud_compute_covariances(self.handle, int(k), int(method)))

    def __len__(self):
        return int(self.ctx._lib.mi355_cloud_size(self.handle))

    @property
    def has_covariances(self):
        return bool(self.ctx._lib.mi355_cloud_has_covariances(self.handle))

    def download(self):
        out = np.empty((max(len(self), 1), 4), dtype=np.float32)
        self.ctx.check(self.ctx._lib.mi355_cloud_download(self.handle, out.ctypes.data, out.shape[0]))
        return out[: len(self)]

    def close(self):
        if getattr(self, "handle", None):
            self.ctx._lib.mi355_cloud_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may be gone already: leave the handles to the process teardown
            return
        try:
            self.close()
        except Exception:
            pass


class _Registration:
    _method = None

    def __init__(self, params, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.params = params
        self.handle = self.ctx._lib.mi355_reg_create(self.ctx.handle, self._method, C.byref(params))
        if not self.handle:
            raise L.Mi355Error(L.EINVAL, "mi355_reg_create failed")
        self._result = None
        self._dirty = False
        self._keep = []

    def _push(self):
        if self._dirty:
            self.ctx.check(self.ctx._lib.mi355_reg_set_params(self.handle, C.byref(self.params)))
            self._dirty = False

    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self.params, k, v)
        self._dirty = True

    # -- pcl::Registration interface ----------------------------------------------------------------------------
    def setInputTarget(self, cloud):
        self._push()
        if isinstance(cloud, DeviceCloud):
            self.ctx.check(self.ctx._lib.mi355_reg_set_target_cloud(self.handle, cloud.handle))
            return
        v, keep = L.view(cloud)
        self.ctx.check(self.ctx._lib.mi355_reg_set_target(self.handle, C.byref(v)))

    def setInputSource(self, cloud):
        self._push()
        if isinstance(cloud, DeviceCloud):
            self._n_src = len(cloud)
            self.ctx.check(self.ctx._lib.mi355_reg_set_source_cloud(self.handle, cloud.handle))
            return
        v, keep = L.view(cloud)
        self._n_src = v.n
        self.ctx.check(self.ctx._lib.mi355_reg_set_source(self.handle, C.byref(v)))

    def align(self, guess=None, return_cloud: bool = False):
        """align(output, guess).  Returns the aligned cloud when ``return_cloud`` (the node never uses it)."""
        self._push()
        g = None
        if guess is not None:
            g = np.ascontiguousarray(guess, dtype=np.float32).reshape(16)
        out = np.empty((max(self._n_src, 1), 4), dtype=np.float32) if return_cloud else None
        r = L.Result()
        self.ctx.check(self.ctx._lib.mi355_reg_align(self.handle, g.ctypes.data if g is not None else None, C.byref(r),
                                                     out.ctypes.data if out is not None else None))
        self._result = r.to_dict()
        return out[: self._n_src] if return_cloud else None

    def hasConverged(self) -> bool:
        return bool(self._result and self._result["converged"])

    def getFinalTransformation(self) -> np.ndarray:
        return self._result["T"] if self._result else np.eye(4, dtype=np.float32)

    def getFitnessScore(self, max_range: float = np.finfo(np.float64).max) -> float:
        s = C.c_double(0)
        self.ctx.check(self.ctx._lib.mi355_reg_fitness(self.handle, C.c_double(max_range), C.byref(s)))
        return s.value

    @property
    def result(self):
        return self._result

    def setMaximumIterations(self, n):
        self._set(max_iterations=int(n))

    def setTransformationEpsilon(self, e):
        self._set(transformation_epsilon=float(e))

    def setNumThreads(self, n):
        self._set(num_threads=int(n))  # accepted for interface parity; the GPU path has no thread count

    def close(self):
        if getattr(self, "handle", None):
            self.ctx._lib.mi355_reg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may be gone already: leave the handles to the process teardown
            return
        try:
            self.close()
        except Exception:
            pass


class FastGICP(_Registration):
    """fast_gicp::FastGICP<PointXYZI,PointXYZI> (lidar_scan_matcher.cpp:37-54)."""

    _method = L.METHOD_FAST_GICP

    def __init__(self, ctx: Context | None = None, **params):
        super().__init__(L.gicp_params(**params), ctx)

    def setCorrespondenceRandomness(self, k):
        self._set(k_correspondences=int(k))

    def setMaxCorrespondenceDistance(self, d):
        self._set(max_correspondence_distance=float(d))

    def setRotationEpsilon(self, e):
        self._set(rotation_epsilon=float(e))

    def setRegularizationMethod(self, m):
        self._set(regularization_method=int(m))

    def setLSQType(self, t):
        self._set(optimizer=int(t))

    def linearize(self, T):
        """FastGICP::linearize at pose T: (H 6x6, b 6, error, correspondences in the caller's point order)."""
        self._push()
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        H = np.empty(36)
        b = np.empty(6)
        err = C.c_double(0)
        corr = np.full(max(self._n_src, 1), -2, dtype=np.int32)
        self.ctx.check(self.ctx._lib.mi355_reg_gicp_linearize(self.handle, T.ctypes.data, H.ctypes.data, b.ctypes.data, C.byref(err), corr.ctypes.data))
        return H.reshape(6, 6), b, err.value, corr[: self._n_src]

    def trial_cost(self, T_lin, T_trial):
        """FastGICP::compute_error(T_trial) with the correspondences / Mahalanobis matrices of a linearisation at T_lin (parity probe):
        (cost as the optimisation prices its LM trials - from the sums of the linearisation pass -, cost by a pass over the points)."""
        A = np.ascontiguousarray(T_lin, dtype=np.float64).reshape(16)
        B = np.ascontiguousarray(T_trial, dtype=np.float64).reshape(16)
        m, d = C.c_double(0), C.c_double(0)
        self.ctx.check(self.ctx._lib.mi355_reg_gicp_trial_cost(self.handle, A.ctypes.data, B.ctypes.data, C.byref(m), C.byref(d)))
        return m.value, d.value


class NormalDistributionsTransform(_Registration):
    """pclomp::NormalDistributionsTransform<PointXYZI,PointXYZI> with DIRECT7 (lidar_scan_matcher.cpp:55-72)."""

    _method = L.METHOD_NDT_OMP

    def __init__(self, ctx: Context | None = None, **params):
        super().__init__(L.ndt_params(**params), ctx)

    def setStepSize(self, s):
        self._set(step_size=float(s))

    def setResolution(self, r):
        self._set(resolution=float(r))

    def setNeighborhoodSearchMethod(self, m):
        if m != "DIRECT7":
            raise ValueError("the reference selects pclomp::DIRECT7 (lidar_scan_matcher.cpp:69); other modes are not built")

    def getTransformationProbability(self):
        return self._result["trans_probability"] if self._result else 0.0

    def derivatives(self, p6, hessian_fp64=False):
        self._push()
        p6 = np.ascontiguousarray(p6, dtype=np.float64).reshape(6)
        score = C.c_double(0)
        g = np.empty(6)
        H = np.empty(36)
        self.ctx.check(self.ctx._lib.mi355_reg_ndt_derivatives(self.handle, p6.ctypes.data, C.byref(score), g.ctypes.data, H.ctypes.data,
                                                               1 if hessian_fp64 else 0))
        return score.value, g, H.reshape(6, 6)

    def cells(self):
        n = C.c_size_t(0)
        self.ctx.check(self.ctx._lib.mi355_reg_ndt_num_cells(self.handle, C.byref(n)))
        m = max(n.value, 1)
        idx = np.empty(m, dtype=np.int64)
        cnt = np.empty(m, dtype=np.int32)
        mean = np.empty((m, 3))
        cov = np.empty((m, 3, 3))
        icov = np.empty((m, 3, 3))
        min_b = np.empty(3, dtype=np.int32)
        div_b = np.empty(3, dtype=np.int32)
        self.ctx.check(self.ctx._lib.mi355_reg_ndt_cells(self.handle, idx.ctypes.data, cnt.ctypes.data, mean.ctypes.data, cov.ctypes.data,
                                                         icov.ctypes.data, min_b.ctypes.data, div_b.ctypes.data))
        k = n.value
        return {"idx": idx[:k], "n": cnt[:k], "mean": mean[:k], "cov": cov[:k], "icov": icov[:k], "min_b": min_b, "div_b": div_b}


class IterativeClosestPoint(_Registration):
    """pcl::IterativeClosestPoint<PointXYZI,PointXYZI>: the back end's default registration method
    (graph_based_slam.param.yaml:9; graph_based_slam.cpp:142-151: the defaults here are that block's setter values)."""

    _method = L.METHOD_ICP

    def __init__(self, ctx: Context | None = None, **params):
        super().__init__(L.icp_params(**params), ctx)

    def setMaxCorrespondenceDistance(self, d):
        self._set(max_correspondence_distance=float(d))

    def setEuclideanFitnessEpsilon(self, e):
        self._set(euclidean_fitness_epsilon=float(e))

    def setRANSACIterations(self, n):
        self._set(ransac_iterations=int(n))


class GeneralizedIterativeClosestPoint(_Registration):
    """pclomp::GeneralizedIterativeClosestPoint<PointXYZI,PointXYZI> — registration_method "GICP" of both nodes
    (lidar_scan_matcher.cpp:73-96, graph_based_slam.cpp:120-141): PCL 1.8's GICP, a BFGS minimisation per re-correspondence."""

    _method = L.METHOD_GICP

    def __init__(self, ctx: Context | None = None, **params):
        super().__init__(L.gicp_bfgs_params(**params), ctx)

    def setMaxCorrespondenceDistance(self, d):
        self._set(max_correspondence_distance=float(d))

    def setMaximumOptimizerIterations(self, n):
        self._set(max_optimizer_iterations=int(n))

    def setRotationEpsilon(self, e):
        self._set(rotation_epsilon=float(e))

    def setEuclideanFitnessEpsilon(self, e):
        self._set(euclidean_fitness_epsilon=float(e))

    def setRANSACIterations(self, n):
        self._set(ransac_iterations=int(n))

    def setCorrespondenceRandomness(self, k):
        self._set(k_correspondences=int(k))


class BatchRegistration:
    """Many independent scan pairs per launch (BASELINE configs 2 and 5): device-resident clouds, one registration per pair —
    FAST_GICP (default), NDT_OMP, ICP or GICP (pclomp's BFGS GICP), the four engines GraphBasedSLAM::get_registration offers
    (graph_based_slam.cpp:80-152; ICP is that node's yaml default)."""

    def __init__(self, n_sources: int, n_targets: int, capacity: int, ctx: Context | None = None, method: str = "FAST_GICP", **params):
        self.ctx = ctx or default_context()
        if method == "FAST_GICP":
            self.params, code = L.gicp_params(**params), L.METHOD_FAST_GICP
        elif method == "NDT_OMP":
            self.params, code = L.ndt_params(**params), L.METHOD_NDT_OMP
        elif method == "ICP":
            self.params, code = L.icp_params(**params), L.METHOD_ICP
        elif method == "GICP":
            self.params, code = L.gicp_bfgs_params(**params), L.METHOD_GICP
        else:
            raise L.Mi355Error(L.EINVAL, f"batch registration method {method!r} (FAST_GICP, NDT_OMP, ICP or GICP)")
        self.method = method
        self.n_sources, self.n_targets, self.capacity = int(n_sources), int(n_targets), int(capacity)
        self.handle = self.ctx._lib.mi355_batch_create(self.ctx.handle, code, C.byref(self.params), self.n_sources, self.n_targets,
                                                       self.capacity)
        if not self.handle:
            raise L.Mi355Error(L.EINVAL, self.ctx._lib.mi355_last_error(self.ctx.handle).decode())

    @staticmethod
    def _counts(c, n):
        c = np.ascontiguousarray(c, dtype=np.int32)
        assert c.shape == (n,), c.shape
        return c

    def set_targets_dev(self, data_ptr: int, counts, prefilter=None):
        c = self._counts(counts, self.n_targets)
        self.ctx.check(self.ctx._lib.mi355_batch_set_targets_dev(self.handle, C.c_void_p(data_ptr), c.ctypes.data, C.byref(prefilter) if prefilter else None))

    def set_sources_dev(self, data_ptr: int, counts, prefilter=None):
        c = self._counts(counts, self.n_sources)
        self.ctx.check(self.ctx._lib.mi355_batch_set_sources_dev(self.handle, C.c_void_p(data_ptr), c.ctypes.data, C.byref(prefilter) if prefilter else None))

    def set_targets(self, points, counts, prefilter=None):
        p = np.ascontiguousarray(points, dtype=np.float32)
        assert p.shape == (self.n_targets, self.capacity, 4), p.shape
        c = self._counts(counts, self.n_targets)
        self.ctx.check(self.ctx._lib.mi355_batch_set_targets(self.handle, p.ctypes.data, c.ctypes.data, C.byref(prefilter) if prefilter else None))

    def set_sources(self, points, counts, prefilter=None):
        p = np.ascontiguousarray(points, dtype=np.float32)
        assert p.shape == (self.n_sources, self.capacity, 4), p.shape
        c = self._counts(counts, self.n_sources)
        self.ctx.check(self.ctx._lib.mi355_batch_set_sources(self.handle, p.ctypes.data, c.ctypes.data, C.byref(prefilter) if prefilter else None))

    def set_targets_clouds(self, clouds, prefilter=None):
        hs = (C.c_void_p * self.n_targets)(*[c.handle for c in clouds])
        self.ctx.check(self.ctx._lib.mi355_batch_set_targets_clouds(self.handle, hs, C.byref(prefilter) if prefilter else None))

    def set_sources_clouds(self, clouds, prefilter=None):
        hs = (C.c_void_p * self.n_sources)(*[c.handle for c in clouds])
        self.ctx.check(self.ctx._lib.mi355_batch_set_sources_clouds(self.handle, hs, C.byref(prefilter) if prefilter else None))

    def set_live(self, n_sources: int, n_targets: int):
        """Use only the first n_sources / n_targets slots until the next call (mi355_batch_set_live): a batch object kept for
        requests of varying size.  Invalidates the sources, targets and results set before."""
        self.ctx.check(self.ctx._lib.mi355_batch_set_live(self.handle, int(n_sources), int(n_targets)))
        self.n_sources, self.n_targets = int(n_sources), int(n_targets)

    def set_stream(self, hip_stream: int | None):
        """Run this batch on its own HIP stream (None: back to the context's); batches on different streams overlap."""
        if hip_stream is None:
            self.ctx.check(self.ctx._lib.mi355_batch_set_stream(self.handle, None, 1))
        else:
            self.ctx.check(self.ctx._lib.mi355_batch_set_stream(self.handle, C.c_void_p(hip_stream), 0))

    def align(self, tgt_of_src=None, guesses=None, with_fitness=False):
        t = None if tgt_of_src is None else np.ascontiguousarray(tgt_of_src, dtype=np.int32)
        g = None if guesses is None else np.ascontiguousarray(guesses, dtype=np.float32).reshape(self.n_sources, 16)
        self.ctx.check(self.ctx._lib.mi355_batch_align(self.handle, t.ctypes.data if t is not None else None, g.ctypes.data if g is not None else None,
                                                       1 if with_fitness else 0))

    def results(self, with_fitness=False, raw=False):
        """Synchronises and returns the n_sources results: a list of dicts, or (raw=True) ONE NumPy record array over the C structs
        (fields T [4,4], converged, iterations, error, H [6,6], trans_probability, n_source, n_target, n_inliers, status, evaluations) - what a
        caller in a loop wants: building 1 024 dicts costs milliseconds of host time with the GPU idle."""
        res = (L.Result * self.n_sources)()
        fit = np.empty(self.n_sources) if with_fitness else None
        self.ctx.check(self.ctx._lib.mi355_batch_results(self.handle, res, fit.ctypes.data if fit is not None else None))
        out = np.frombuffer(res, dtype=L.RESULT_DTYPE, count=self.n_sources) if raw else [r.to_dict() for r in res]
        return (out, fit) if with_fitness else out

    def verify_correspondences(self):
        """(points checked, mismatches) of the incremental correspondence search against a from-scratch search at the final poses."""
        a, b = C.c_longlong(0), C.c_longlong(0)
        self.ctx.check(self.ctx._lib.mi355_batch_verify_correspondences(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def diagnostics(self):
        """After align() of a FAST_GICP batch (synchronises): {"first_converged": int32 [n_sources] - iterations run when fast_gicp's
        is_converged first held, -1 = never; "knn_points" / "knn_handed_over": (targets, sources) points of the last covariance
        launches and how many the tile search handed over to the one-wave-per-query kernel}."""
        fc = np.empty(self.n_sources, dtype=np.int32)
        pts, fb = np.zeros(2, dtype=np.int64), np.zeros(2, dtype=np.int64)
        self.ctx.check(self.ctx._lib.mi355_batch_diagnostics(self.handle, fc.ctypes.data, pts.ctypes.data, fb.ctypes.data))
        return {"first_converged": fc, "knn_points": pts, "knn_handed_over": fb}

    def get_cloud(self, targets: bool, index: int, with_cov: bool = False):
        """Parity probe (synchronises): cloud `index` of the targets / sources as the batch registers it - the points the prefilter left,
        in its output order, float32 [n, 4]; with_cov: also their regularised covariances, float32 [n, 6] (xx xy xz yy yz zz)."""
        n = C.c_size_t(0)
        pts = np.zeros((self.capacity, 4), dtype=np.float32)
        cov = np.zeros((self.capacity, 6), dtype=np.float32) if with_cov else None
        self.ctx.check(self.ctx._lib.mi355_batch_get_cloud(self.handle, 1 if targets else 0, int(index), pts.ctypes.data,
                                                           cov.ctypes.data if with_cov else None, self.capacity, C.byref(n)))
        return (pts[: n.value].copy(), cov[: n.value].copy()) if with_cov else pts[: n.value].copy()

    def set_side_streams(self, on: bool | None):
        """Prepare targets and sources on two overlapping side streams (True, the default), or one after the other on the batch's
        own stream (False: what a profiler wants); None = the library default."""
        self.ctx.check(self.ctx._lib.mi355_batch_set_side_streams(self.handle, -1 if on is None else (1 if on else 0)))

    def set_timing(self, enable: bool = True):
        """Record HIP events around the hot kernels of this batch (bench.py's roofline block)."""
        self.ctx.check(self.ctx._lib.mi355_batch_set_timing(self.handle, 1 if enable else 0))

    def kernel_times(self):
        """{kernel family: (total ms, launches)} since the last call; synchronises the batch."""
        buf = (L.KernelTime * 32)()
        n = C.c_int(0)
        self.ctx.check(self.ctx._lib.mi355_batch_kernel_times(self.handle, buf, 32, C.byref(n)))
        return {buf[i].name.decode(): (buf[i].total_ms, buf[i].launches) for i in range(min(n.value, 32))}

    def close(self):
        if getattr(self, "handle", None):
            self.ctx._lib.mi355_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may be gone already: leave the handles to the process teardown
            return
        try:
            self.close()
        except Exception:
            pass
