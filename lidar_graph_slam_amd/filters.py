"""PCL-filter mirror of the reference's prefilter node (points_prefiltering.cpp:65-148), running on the MI355X.

Class and method names follow the PCL objects the node holds (`pcl::VoxelGrid`, `pcl::StatisticalOutlierRemoval`,
points_prefiltering.cpp:146-148) so that the parity tests read like the reference's call sites:
``vg.setLeafSize(l, l, l); vg.setInputCloud(c); out = vg.filter()``.  Clouds are float32 [n,4] (x,y,z,intensity)
or [n,8] in pcl::PointXYZI memory layout.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .context import Context, default_context


def _out(n):
    return np.empty((max(n, 1), 4), dtype=np.float32)


def distance_filter(cloud, min_distance: float, ctx: Context | None = None) -> np.ndarray:
    """PointsPreFiltering::distance_filter (points_prefiltering.cpp:102-112)."""
    ctx = ctx or default_context()
    v, keep = L.view(cloud)
    out = _out(v.n)
    n = C.c_size_t(0)
    ctx.check(ctx._lib.mi355_distance_filter(ctx.handle, C.byref(v), C.c_double(min_distance), out.ctypes.data, out.shape[0], C.byref(n)))
    return out[: n.value].copy()


def transform_point_cloud(cloud, T, ctx: Context | None = None) -> np.ndarray:
    """pcl::transformPointCloud (lidar_scan_matcher.cpp:282,291)."""
    ctx = ctx or default_context()
    v, keep = L.view(cloud)
    T = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
    out = _out(v.n)
    ctx.check(ctx._lib.mi355_transform(ctx.handle, C.byref(v), T.ctypes.data, out.ctypes.data))
    return out[: v.n].copy()


class VoxelGrid:
    """pcl::VoxelGrid<PointXYZI> (points_prefiltering.cpp:118-120; graph_based_slam.cpp:61,311-313)."""

    def __init__(self, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.leaf = None
        self.cloud = None
        self.overflow = False

    def setLeafSize(self, lx, ly, lz):
        if not (lx == ly == lz):
            raise ValueError("the reference only uses cubic leaves (leaf_size_, leaf_size_, leaf_size_)")
        self.leaf = float(lx)

    def setInputCloud(self, cloud):
        self.cloud = cloud

    def filter(self) -> np.ndarray:
        v, keep = L.view(self.cloud)
        out = _out(v.n)
        n = C.c_size_t(0)
        ovf = C.c_int(0)
        self.ctx.check(self.ctx._lib.mi355_voxel_grid(self.ctx.handle, C.byref(v), C.c_float(self.leaf), out.ctypes.data, out.shape[0], C.byref(n),
                                                      C.byref(ovf)))
        self.overflow = bool(ovf.value)
        return out[: n.value].copy()


class StatisticalOutlierRemoval:
    """pcl::StatisticalOutlierRemoval<PointXYZI> (points_prefiltering.cpp:136-139)."""

    def __init__(self, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.mean_k = 1
        self.stddev_mul = 0.0
        self.cloud = None
        self.keep_mask = None
        self.mean_distances = None

    def setMeanK(self, k):
        self.mean_k = int(k)

    def setStddevMulThresh(self, m):
        self.stddev_mul = float(m)

    def setInputCloud(self, cloud):
        self.cloud = cloud

    def filter(self) -> np.ndarray:
        v, keep = L.view(self.cloud)
        out = _out(v.n)
        mask = np.zeros(max(v.n, 1), dtype=np.uint8)
        md = np.zeros(max(v.n, 1), dtype=np.float32)
        n = C.c_size_t(0)
        self.ctx.check(self.ctx._lib.mi355_sor(self.ctx.handle, C.byref(v), self.mean_k, C.c_double(self.stddev_mul), out.ctypes.data, out.shape[0],
                                               C.byref(n), mask.ctypes.data, md.ctypes.data))
        self.keep_mask = mask[: v.n].astype(bool)
        self.mean_distances = md[: v.n]
        return out[: n.value].copy()


class PointsPreFiltering:
    """The node's callback pipeline (points_prefiltering.cpp:65-87): distance filter -> VoxelGrid -> SOR, all on device.
    Parameter names are the node's ROS parameters (points_prefiltering.launch.xml:2-13)."""

    def __init__(self, leaf_size=0.1, mean_k=30, stddev=1.2, min_distance_cloud=1.0, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.params = L.prefilter_params(min_distance=min_distance_cloud, leaf_size=leaf_size, mean_k=mean_k, stddev_mul=stddev)

    def callback_sensor_points(self, cloud) -> np.ndarray:
        v, keep = L.view(cloud)
        out = _out(v.n)
        n = C.c_size_t(0)
        self.ctx.check(self.ctx._lib.mi355_prefilter(self.ctx.handle, C.byref(v), C.byref(self.params), out.ctypes.data, out.shape[0], C.byref(n)))
        return out[: n.value].copy()


def knn(points, queries, k: int, ctx: Context | None = None):
    """pcl::search::KdTree::nearestKSearch for every query: (indices [m,k], squared distances [m,k])."""
    ctx = ctx or default_context()
    vp, k1 = L.view(points)
    vq, k2 = L.view(queries)
    idx = np.empty((max(vq.n, 1), k), dtype=np.int32)
    d2 = np.empty((max(vq.n, 1), k), dtype=np.float32)
    ctx.check(ctx._lib.mi355_knn(ctx.handle, C.byref(vp), C.byref(vq), int(k), idx.ctypes.data, d2.ctypes.data))
    return idx[: vq.n], d2[: vq.n]


def covariances(points, k: int = 20, method: int = L.REG_PLANE, ctx: Context | None = None) -> np.ndarray:
    """fast_gicp calculate_covariances: [n,3,3] float32 (caller's point order)."""
    ctx = ctx or default_context()
    vp, k1 = L.view(points)
    c6 = np.empty((max(vp.n, 1), 6), dtype=np.float32)
    ctx.check(ctx._lib.mi355_covariances(ctx.handle, C.byref(vp), int(k), int(method), c6.ctypes.data))
    c6 = c6[: vp.n]
    out = np.empty((vp.n, 3, 3), dtype=np.float32)
    out[:, 0, 0], out[:, 0, 1], out[:, 0, 2] = c6[:, 0], c6[:, 1], c6[:, 2]
    out[:, 1, 0], out[:, 1, 1], out[:, 1, 2] = c6[:, 1], c6[:, 3], c6[:, 4]
    out[:, 2, 0], out[:, 2, 1], out[:, 2, 2] = c6[:, 2], c6[:, 4], c6[:, 5]
    return out
