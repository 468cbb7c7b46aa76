"""TEST INFRASTRUCTURE: an independent NumPy restatement of LidarScanMatcher::callback_cloud
(/root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:122-250) that registers with the ORACLE, used to check
the C++ ScanMatcherCore + HIP registration scan by scan.  Shares no code with csrc/host/scan_matcher_core.cpp."""
import numpy as np

from oracle import oracle_py as O


def matrix_to_pose(T):
    """convert_matrix_to_pose (lidar_graph_slam_utils.hpp:105-116): float matrix -> double position + Eigen quaternion."""
    from scipy.spatial.transform import Rotation

    T = np.asarray(T, dtype=np.float32).astype(np.float64)
    q = Rotation.from_matrix(T[:3, :3]).as_quat()  # x y z w
    if q[3] < 0:
        q = -q
    return T[:3, 3].copy(), q


def pose_to_matrix(p, q):
    """geometry_pose_to_matrix (lidar_graph_slam_utils.hpp:42-48)."""
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = p
    return T.astype(np.float32)


class RefScanMatcher:
    def __init__(self, method="FAST_GICP", displacement=1.0, max_scan_accumulate_num=20, max_iteration=64, transformation_epsilon=0.01,
                 step_size=0.1, ndt_resolution=2.0, max_correspondence_distance=2.0, correspondence_randomness=20):
        self.method = method
        self.displacement, self.window = displacement, max_scan_accumulate_num
        self.gp = O.gicp_params(k_correspondences=correspondence_randomness, max_iterations=max_iteration, transformation_epsilon=transformation_epsilon,
                                max_correspondence_distance=max_correspondence_distance)
        self.np_ = O.ndt_params(transformation_epsilon=transformation_epsilon, step_size=step_size, resolution=ndt_resolution, max_iterations=max_iteration)
        self.target = None
        self.key_frames = []
        self.prev = np.eye(4, dtype=np.float32)
        self.key_pose = np.eye(4, dtype=np.float32)
        self.accum = 0.0

    def callback_cloud(self, cloud, base_to_sensor=None):
        B = np.eye(4, dtype=np.float32) if base_to_sensor is None else np.asarray(base_to_sensor, dtype=np.float32)
        base_cloud = O.transform(cloud, B)
        if self.target is None:
            self.key_frames.append((cloud.copy(), matrix_to_pose(self.prev), 0.0))
            self.target = base_cloud.copy()
            return {"first_scan": True, "converged": False, "pose": self.prev.copy(), "new_key_frame": True}
        if self.method == "FAST_GICP":
            r = O.gicp_align(base_cloud, self.target, self.gp, guess=self.prev)
        else:
            r = O.ndt_align(base_cloud, self.target, self.np_, guess=self.prev)
        if not r["converged"]:
            return {"first_scan": False, "converged": False, "pose": self.prev.copy(), "new_key_frame": False}
        T = r["T"]
        self.prev = T.copy()
        delta = float(np.linalg.norm(T[:3, 3].astype(np.float64) - self.key_pose[:3, 3].astype(np.float64)))
        new_kf = False
        if self.displacement <= delta:
            self.key_pose = T.copy()
            self.accum += delta
            self.key_frames.append((base_cloud.copy(), matrix_to_pose(T), self.accum))
            parts = []
            for idx in range(self.window):
                k = len(self.key_frames) - 1 - idx
                if k < 0:
                    continue
                c, (p, q), _ = self.key_frames[k]
                parts.append(O.transform(c, pose_to_matrix(p, q)))
            self.target = np.concatenate(parts, axis=0)
            new_kf = True
        return {"first_scan": False, "converged": True, "pose": T.copy(), "new_key_frame": new_kf}
