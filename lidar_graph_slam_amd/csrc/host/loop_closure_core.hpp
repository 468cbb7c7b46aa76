// LoopClosureCore — ROS-free restatement of the registration half of GraphBasedSLAM::optimization_callback
// (/root/reference/graph_based_slam/src/graph_based_slam.cpp:238-328) over plain structs, batched: the reference closes
// one loop per 1 Hz tick (latest key frame vs the sub-map around ONE candidate); here any number of query key frames are
// registered against the sub-maps around their candidates in one mi355_batch_* launch sequence (BASELINE config 5).
// The pose graph (GTSAM iSAM2, :331-349) stays with the caller, which receives the relative pose of the loop factor.
#pragma once
#include <memory>
#include <vector>

#include "scan_matcher_core.hpp"

namespace mi355 {

struct LoopClosureParams {  // graph_based_slam.param.yaml:3-23; sub-map leaf hard-coded at graph_based_slam.cpp:61
  int search_key_frame_num = 20;
  double score_threshold = 0.3;
  double search_for_candidate_threshold = 15.0;
  double accumulate_distance_threshold = 100.0;
  float submap_leaf = 0.5f;
  int max_iteration = 100;
  double transformation_epsilon = 0.01;
  double max_correspondence_distance = 2.0;
  int correspondence_randomness = 20;
  // get_registration (:74-155): "FAST_GICP" and "NDT_OMP" register all candidates in one batch; "ICP" — the yaml's default
  // (graph_based_slam.param.yaml:9), pcl::IterativeClosestPoint with the values hard-coded at :145-149 — runs pair by pair
  // on the device-resident clouds (its Horn fit is a 4x4 eigenproblem on the host per iteration), and so does "GICP"
  // (pclomp's BFGS GICP, :120-141: the quasi-Newton steps are host logic).
  int registration_method = MI355_METHOD_ICP;
  double step_size = 0.1;        // yaml:17 (NDT_OMP)
  double ndt_resolution = 1.0;   // yaml:18
  double correspondence_distance = 2.0;    // yaml:26 (GICP)
  int max_optimizer_iteration = 20;        // yaml:23
  double euclidean_fitness_epsilon = 1e-8; // yaml:27
};

struct LoopResult {
  int query_id = -1, candidate_id = -1;
  bool converged = false, accepted = false;
  int iterations = 0;
  double fitness = 0.0;
  Matrix4f transform = identity4();
  std::array<double, 16> between{};  // pose_from.between(pose_to), row-major
  size_t target_points = 0;
};

class LoopClosureCore {
 public:
  LoopClosureCore(int device_id, const LoopClosureParams& p) : device_id_(device_id), params_(p) {}
  ~LoopClosureCore();
  // key_frame_callback's bookkeeping (:354-380): the cloud is in the key frame's own frame
  int add_key_frame(PointCloud cloud, const Pose& pose, double accum_distance);
  void set_pose(int id, const Pose& pose) { key_frames_.at((size_t)id).pose = pose; }
  // :256-280 with `query` in the role of the latest key frame; -1 = no candidate
  int candidate(int query) const;
  // :282-341 for every query, one batch
  std::vector<LoopResult> close(const std::vector<int>& queries);
  size_t size() const { return key_frames_.size(); }

 private:
  int device_id_;
  LoopClosureParams params_;
  std::vector<KeyFrame> key_frames_;   // KeyFrame::device is filled (one upload per key frame) by the first close() that needs it
  std::unique_ptr<Context> ctx_;       // created by the first close(): the bookkeeping above needs no GPU
  // the batch object outlives close(): its buffers (tens of GB at 512 candidates) are allocated once and grown only when a
  // call needs more sources, sub-maps or capacity
  mi355_batch* batch_ = nullptr;
  std::shared_ptr<Registration> pairwise_;  // registration_method == ICP or GICP: one candidate at a time
  int batch_S_ = 0, batch_U_ = 0;
  size_t batch_cap_ = 0;
};

}  // namespace mi355
