"""Wire / on-disk formats at the edges of the hot path: sensor_msgs/PointCloud2 <-> device clouds
(pcl::fromROSMsg / pcl::toROSMsg at lidar_scan_matcher.cpp:127,145,196,203) and the PCD writer behind the back end's save_map
service (graph_based_slam.cpp:469-501).  A PointCloud2 is described by a plain dict with the message's own field names, so a
ROS node hands over `msg.data`, `msg.point_step`, ... without this package importing ROS."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .context import Context, default_context
from .registration import DeviceCloud

FLOAT32 = 7


def cloud_from_pointcloud2(msg: dict, ctx: Context | None = None) -> DeviceCloud:
    """msg: {data: bytes-like, width, height, point_step, row_step, fields: {name: (offset, datatype)}, is_bigendian}."""
    ctx = ctx or default_context()
    f = msg["fields"]
    for name in ("x", "y", "z"):
        if f[name][1] != FLOAT32:
            raise ValueError("x, y, z must be FLOAT32 (what every LiDAR driver publishes and pcl::fromROSMsg maps without conversion)")
    raw = np.frombuffer(msg["data"], dtype=np.uint8)
    oi, it = f.get("intensity", (-1, FLOAT32))
    if raw.size < msg["row_step"] * msg["height"]:
        raise ValueError(f"PointCloud2 data holds {raw.size} bytes, row_step * height = {msg['row_step'] * msg['height']}")
    m = L.PointCloud2(raw.ctypes.data if raw.size else None, msg["width"], msg["height"], msg["point_step"], msg["row_step"], f["x"][0], f["y"][0],
                      f["z"][0], oi, it, 1 if msg.get("is_bigendian") else 0, raw.size)
    h = ctx._lib.mi355_cloud_from_pointcloud2(ctx.handle, C.byref(m))
    if not h:
        raise L.Mi355Error(L.EINVAL, ctx._lib.mi355_last_error(ctx.handle).decode())
    return DeviceCloud(h, ctx)


def cloud_to_pointcloud2(cloud: DeviceCloud) -> dict:
    """pcl::toROSMsg of a pcl::PointCloud<pcl::PointXYZI>: the dict form of the message (data as bytes)."""
    n = len(cloud)
    out = np.zeros(max(n, 1) * 32, dtype=np.uint8)
    cloud.ctx.check(cloud.ctx._lib.mi355_cloud_to_pointcloud2(cloud.handle, out.ctypes.data, out.size))
    return {"data": out[: n * 32].tobytes(), "width": n, "height": 1, "point_step": 32, "row_step": 32 * n, "is_bigendian": False, "is_dense": True,
            "fields": {"x": (0, FLOAT32), "y": (4, FLOAT32), "z": (8, FLOAT32), "intensity": (16, FLOAT32)}}


def save_pcd(path: str, cloud, binary: bool = False):
    """pcl::io::savePCDFile(path, cloud, binary): `cloud` is a DeviceCloud or a host array [n, 4] / [n, 8]."""
    lib = L.lib()
    if isinstance(cloud, DeviceCloud):
        cloud.ctx.check(lib.mi355_cloud_save_pcd(cloud.handle, path.encode(), 1 if binary else 0))
        return
    v, keep = L.view(cloud)
    if lib.mi355_pcd_write(path.encode(), C.byref(v), 1 if binary else 0) != L.OK:
        raise OSError(f"could not write {path}")
