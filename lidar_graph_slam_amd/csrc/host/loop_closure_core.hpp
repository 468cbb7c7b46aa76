// LoopClosureCore — ROS-free restatement of the registration half of GraphBasedSLAM::optimization_callback
// (/root/reference/graph_based_slam/src/graph_based_slam.cpp:238-328) over plain structs, batched: the reference closes
// one loop per 1 Hz tick (latest key frame vs the sub-map around ONE candidate); here any number of query key frames are
// registered against the sub-maps around their candidates in one mi355_batch_* launch sequence (BASELINE config 5).
// The pose graph (GTSAM iSAM2, :331-349) stays with the caller, which receives the relative pose of the loop factor.
#pragma once
#include <memory>
#include <vector>

#include "lanes.hpp"
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
  // get_registration (:74-155): every engine registers all candidates of a lane in ONE batch - "FAST_GICP", "NDT_OMP", "ICP" (the
  // yaml's default, graph_based_slam.param.yaml:9: pcl::IterativeClosestPoint with the values hard-coded at :145-149) and "GICP"
  // (pclomp's BFGS GICP, :120-141); the per-pair optimiser states advance on the device.
  int registration_method = MI355_METHOD_ICP;
  double step_size = 0.1;        // yaml:17 (NDT_OMP)
  double ndt_resolution = 1.0;   // yaml:18
  double correspondence_distance = 2.0;    // yaml:26 (GICP)
  int max_optimizer_iteration = 20;        // yaml:23
  double euclidean_fitness_epsilon = 1e-8; // yaml:27
  int ransac_iteration = 0;                // yaml "ransac_iteration" (setRANSACIterations, :139,:150): only 0 — no rejector — is built
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
  // device_ids: one lane (context + stream + batch object) per entry; the candidates of one close() are dealt to the lanes by
  // sub-map (lanes.hpp) and every lane registers its share under its own host thread.  The same ordinal may appear twice.
  LoopClosureCore(const std::vector<int>& device_ids, const LoopClosureParams& p);
  LoopClosureCore(int device_id, const LoopClosureParams& p) : LoopClosureCore(std::vector<int>{device_id}, p) {}
  ~LoopClosureCore();
  // key_frame_callback's bookkeeping (:354-380): the cloud is in the key frame's own frame
  int add_key_frame(PointCloud cloud, const Pose& pose, double accum_distance);
  void set_pose(int id, const Pose& pose) { key_frames_.at((size_t)id).pose = pose; }
  // :256-280 with `query` in the role of the latest key frame; -1 = no candidate
  int candidate(int query) const;
  // :282-341 for every query, one batch
  std::vector<LoopResult> close(const std::vector<int>& queries);
  size_t size() const { return key_frames_.size(); }
  int lanes() const { return (int)lanes_.size(); }

 private:
  struct LaneState {
    explicit LaneState(int device) : lane(device) {}
    Lane lane;                                // first member: destroyed last (its context outlives everything below)
    std::vector<DeviceCloud::Ptr> key_frame;  // this lane's copies: one upload per key frame and lane, by the first close() that needs it
  };
  // one lane's share of a close(): rows of `out` (all with a candidate) -> res / fit at the same positions
  void register_rows(LaneState& ls, const std::vector<int>& rows, const std::vector<LoopResult>& out, std::vector<mi355_result>& res, std::vector<double>& fit);
  LoopClosureParams params_;
  std::vector<KeyFrame> key_frames_;   // host copies (KeyFrame::cloud); no GPU is touched before the first close()
  std::vector<std::unique_ptr<LaneState>> lanes_;
};

}  // namespace mi355
