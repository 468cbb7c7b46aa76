"""the PCD writer (host-only, CPU test) and PointCloud2 <-> device clouds (GPU tests), against NumPy restatements."""
import numpy as np
import pytest

from oracle import numpy_ref as N


def _cloud(n=257, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-80, 80, (n, 4)).astype(np.float32)
    p[:, 3] = rng.uniform(0, 255, n).astype(np.float32)
    p[3] = (1.0, -2.5e-7, 123456.79, 0.0)
    p[4, 1] = np.nan
    return p


def test_pcd_writer_ascii_and_binary(tmp_path):
    from lidar_graph_slam_amd import wire

    p = _cloud()
    a = str(tmp_path / "a.pcd")
    wire.save_pcd(a, p)
    assert open(a).read() == N.pcd_ascii(p)
    xyzi8 = np.zeros((len(p), 8), np.float32)  # the pcl::PointXYZI memory layout: same file
    xyzi8[:, :3], xyzi8[:, 3], xyzi8[:, 4] = p[:, :3], 1.0, p[:, 3]
    wire.save_pcd(a, xyzi8)
    assert open(a).read() == N.pcd_ascii(p)
    b = str(tmp_path / "b.pcd")
    wire.save_pcd(b, p, binary=True)
    raw = open(b, "rb").read()
    head, body = raw.split(b"DATA binary\n", 1)
    assert head.decode() + "DATA binary\n" == N.pcd_ascii(p)[: len(head) + 12].replace("DATA ascii", "DATA binar")[: len(head)] + "DATA binary\n"
    np.testing.assert_array_equal(np.frombuffer(body, np.float32).reshape(-1, 4), p)
    wire.save_pcd(a, p[:0])
    assert "POINTS 0\nDATA ascii\n" in open(a).read()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["velodyne_xyzir", "pcl_xyzi", "u8_intensity_2d", "no_intensity"])
def test_pointcloud2_round_trip(layout):
    from lidar_graph_slam_amd import wire

    p = _cloud(300)
    n = len(p)
    expect_intensity_dropped = False
    if layout == "velodyne_xyzir":  # x y z intensity(f32) ring(u16) time(f32): point_step 22, unaligned fields
        ps, fields = 22, {"x": (0, 7), "y": (4, 7), "z": (8, 7), "intensity": (12, 7)}
        w, h, rs = n, 1, 22 * n + 10
    elif layout == "pcl_xyzi":
        ps, fields = 32, {"x": (0, 7), "y": (4, 7), "z": (8, 7), "intensity": (16, 7)}
        w, h, rs = n, 1, 32 * n
    elif layout == "u8_intensity_2d":  # organised cloud, 8-bit intensity, padded rows
        ps, fields = 16, {"x": (4, 7), "y": (8, 7), "z": (0, 7), "intensity": (13, 2)}
        w, h, rs = 30, 10, 16 * 30 + 7
        p[:, 3] = np.floor(p[:, 3])
        expect_intensity_dropped = True  # pcl::fromROSMsg does not convert a UINT8 field into PointXYZI's float: intensity stays 0
    else:
        ps, fields = 12, {"x": (0, 7), "y": (4, 7), "z": (8, 7)}
        w, h, rs = n, 1, 12 * n
        p[:, 3] = 0
    raw = np.random.default_rng(1).integers(0, 255, rs * h, dtype=np.uint8)
    rows = raw.reshape(h, rs)
    for i in range(n):
        rec = rows[i // w, (i % w) * ps : (i % w + 1) * ps]
        for k, name in enumerate(("x", "y", "z", "intensity")):
            if name in fields:
                off, dt = fields[name]
                rec[off : off + (4 if dt == 7 else 1)] = np.frombuffer(np.float32(p[i, k]).tobytes() if dt == 7 else np.uint8(p[i, k]).tobytes(), np.uint8)
    msg = {"data": raw.tobytes(), "width": w, "height": h, "point_step": ps, "row_step": rs, "fields": fields, "is_bigendian": False}
    ref = N.pointcloud2_unpack(msg)
    if expect_intensity_dropped:
        p[:, 3] = 0
    np.testing.assert_array_equal(ref[~np.isnan(ref)], p[~np.isnan(p)])
    with pytest.raises(ValueError):  # a message whose data is shorter than row_step * height is refused, not read past its end
        wire.cloud_from_pointcloud2({**msg, "data": msg["data"][:-1]})
    cloud = wire.cloud_from_pointcloud2(msg)
    got = cloud.download()
    np.testing.assert_array_equal(got, ref)
    back = wire.cloud_to_pointcloud2(cloud)
    assert back["point_step"] == 32 and back["width"] == n and back["row_step"] == 32 * n
    assert back["data"] == N.pointcloud2_pack_xyzi(ref)
    np.testing.assert_array_equal(N.pointcloud2_unpack(back), ref)
