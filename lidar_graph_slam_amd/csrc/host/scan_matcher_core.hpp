// ScanMatcherCore — ROS-free restatement of LidarScanMatcher::callback_cloud
// (/root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:122-250) over plain structs: the per-scan sequence
// (transform to base frame -> setInputSource -> align(prev pose) -> converged? -> key-frame test -> sliding sub-map of
// the last N key frames -> setInputTarget) and its state, with the registration running on the MI355X.
// Publishers, TF broadcasting and message (de)serialisation are ROS plumbing and stay out.
#pragma once
#include <cmath>
#include <string>

#include "pcl_like.hpp"

namespace mi355 {

struct Pose {  // geometry_msgs::msg::Pose
  double px = 0, py = 0, pz = 0;
  double qx = 0, qy = 0, qz = 0, qw = 1;
};

// lidar_graph_slam_utils::convert_matrix_to_pose (lidar_graph_slam_utils.hpp:105-116): Eigen::Quaterniond(Matrix3d)
inline Pose convert_matrix_to_pose(const Matrix4f& m) {
  Pose p;
  p.px = (double)m[3]; p.py = (double)m[7]; p.pz = (double)m[11];
  double R[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r][c] = (double)m[4 * r + c];
  double q[3];
  double t = R[0][0] + R[1][1] + R[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    p.qw = 0.5 * t;
    t = 0.5 / t;
    p.qx = (R[2][1] - R[1][2]) * t;
    p.qy = (R[0][2] - R[2][0]) * t;
    p.qz = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    p.qw = (R[k][j] - R[j][k]) * t;
    q[j] = (R[j][i] + R[i][j]) * t;
    q[k] = (R[k][i] + R[i][k]) * t;
    p.qx = q[0]; p.qy = q[1]; p.qz = q[2];
  }
  return p;
}

// lidar_graph_slam_utils::geometry_pose_to_matrix (lidar_graph_slam_utils.hpp:42-48): tf2::fromMsg -> Affine3d -> float
inline Matrix4f geometry_pose_to_matrix(const Pose& p) {
  const double x = p.qx, y = p.qy, z = p.qz, w = p.qw;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Matrix4f m = identity4();
  m[0] = (float)(1 - (tyy + tzz)); m[1] = (float)(txy - twz);       m[2] = (float)(txz + twy);        m[3] = (float)p.px;
  m[4] = (float)(txy + twz);       m[5] = (float)(1 - (txx + tzz)); m[6] = (float)(tyz - twx);        m[7] = (float)p.py;
  m[8] = (float)(txz - twy);       m[9] = (float)(tyz + twx);       m[10] = (float)(1 - (txx + tyy)); m[11] = (float)p.pz;
  return m;
}

inline Matrix4f matmul(const Matrix4f& a, const Matrix4f& b) {  // Eigen Matrix4f product
  Matrix4f c{};
  for (int r = 0; r < 4; r++)
    for (int s = 0; s < 4; s++) {
      float acc = 0.f;
      for (int k = 0; k < 4; k++) acc += a[4 * r + k] * b[4 * k + s];
      c[4 * r + s] = acc;
    }
  return c;
}

struct KeyFrame {  // lidar_graph_slam_msgs/msg/KeyFrame.msg:1-5 (header dropped)
  PointCloud cloud;           // host copy: filled by callers that hand key frames over on the host (LoopClosureCore::add_key_frame)
                              // and by ScanMatcherCore when a key frame leaves the sliding window (its `device` is then released)
  DeviceCloud::Ptr device;    // the cloud in HBM (with its GICP covariances once it has been a registration source); null once evicted
  Pose pose;
  double accum_distance = 0.0;
  long id = 0;
};

struct ScanMatcherParams {  // lidar_scan_matcher.param.yaml:3-19
  std::string registration_method = "NDT_OMP";
  double displacement = 1.0;
  int max_scan_accumulate_num = 20;
  int max_iteration = 64;
  int omp_num_thread = 4;
  double transformation_epsilon = 0.01;
  double step_size = 0.1;
  double ndt_resolution = 2.0;
  double max_correspondence_distance = 2.0;
  int correspondence_randomness = 20;
  // "GICP" block of the yaml (:21-26): pclomp's BFGS GICP
  double correspondence_distance = 5.0;
  double euclidean_fitness_epsilon = 1e-8;
  int max_optimizer_iteration = 20;
  bool use_reciprocal_correspondences = false;
  // false (default) = the reference's arithmetic: fast_gicp's setInputTarget (lidar_scan_matcher.cpp:212) recomputes the
  // k-NN covariances over the merged sub-map, whose neighbourhoods also hold points of the other key frames; the sub-map
  // is still assembled on the device.
  // true = opt-in optimisation, NOT the reference's arithmetic: a key frame's covariances are computed once
  // (when the scan is registered) and ROTATED into every sub-map it is part of (C' = R C R^T) instead of being recomputed
  // over the ~3 x 10^5-point union at every key frame.  Measured deviation: DESIGN.md section 3.
  bool reuse_key_frame_covariances = false;
};

struct ScanOutput {
  bool first_scan = false;   // became the initial target; no pose published (cpp:133-160)
  bool converged = false;    // false => scan dropped, pose frozen (cpp:167-170)
  bool new_key_frame = false;
  Matrix4f pose = identity4();  // translation_ (map <- base)
  long key_frame_id = -1;
  size_t target_points = 0;
  int iterations = 0;
};

class ScanMatcherCore {
 public:
  ScanMatcherCore(Context& ctx, const ScanMatcherParams& p);
  // `msg` cloud in the sensor frame; base_to_sensor = the TF the node looks up (identity when the lookup fails, cpp:259-271)
  ScanOutput callback_cloud(const PointCloud& input_cloud, const Matrix4f& base_to_sensor);
  // NOTE: KeyFrame::device is non-null only while the key frame is inside the sliding window (the last max_scan_accumulate_num
  // frames); when a frame leaves it its cloud is downloaded into KeyFrame::cloud and `device` is reset - read clouds through
  // key_frame_cloud(i), which serves either.
  const std::vector<KeyFrame>& key_frames() const { return key_frame_array_; }
  // key_frame.cloud as the reference holds it (:145,196): from the host copy, or downloaded while the frame is still in the window
  PointCloud key_frame_cloud(size_t i) const {
    const KeyFrame& kf = key_frame_array_.at(i);
    if (!kf.device) return kf.cloud;
    PointCloud c;
    kf.device->download(c);
    return c;
  }
  // the sliding sub-map (target_cloud_), downloaded on demand
  PointCloudPtr target_cloud() const;
  size_t target_points() const { return target_cloud_ ? target_cloud_->size() : 0; }
  double accum_distance() const { return accum_distance_; }

 private:
  Context& ctx_;
  ScanMatcherParams params_;
  Registration::Ptr registration_;
  DeviceCloud::Ptr target_cloud_;  // device-resident: assembled by a gather kernel, never staged through the host
  bool gicp_ = false;
  std::vector<KeyFrame> key_frame_array_;
  Matrix4f prev_translation_ = identity4(), key_frame_ = identity4(), translation_ = identity4();
  double accum_distance_ = 0.0;
  long id_ = 0;
};

}  // namespace mi355
