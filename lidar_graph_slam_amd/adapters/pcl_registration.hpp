// PCL-typed adapter: lets the reference's nodes keep `pcl::Registration<PointType, PointType>::Ptr registration_`
// (lidar_scan_matcher.hpp:99, graph_based_slam.hpp:108) and swap only the three `new ...` expressions
// (lidar_scan_matcher.cpp:38-39,56-57; graph_based_slam.cpp:84-85,103-104).  Header-only, pure forwarding to the C-ABI.
//
// PCL, Eigen and ROS 2 are absent from this repository's build image: the header is type-checked against
// a stub of PCL's surface (tests/pcl_stub, tests/test_adapter_compiles.py) and is otherwise the binding a maintainer adds on
// a machine that has PCL — see INTEGRATION.md.
//
// Two properties of pcl::Registration shape it:
//  * `getFitnessScore()` is NOT virtual.  The back end calls it through the base pointer (graph_based_slam.cpp:321), which
//    runs PCL's CPU loop over `tree_`; the device statistic (mi355_reg_fitness, same definition) is reached through
//    mi355_pcl::getFitnessScore(registration) below — the one-line change INTEGRATION.md shows for that call site.
//  * `align()` -> `initCompute()` rebuilds `tree_` (a CPU kd-tree) over the whole target whenever the target changed —
//    on the 1.3 M-point sub-map of every key frame, more CPU time than the whole device alignment.  The adapter installs a
//    LAZY kd-tree with force_no_recompute: nothing is built unless somebody does call PCL's own getFitnessScore.
#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>
#include <pcl/search/kdtree.h>

#include <algorithm>
#include <cstddef>
#include <limits>
#include <stdexcept>
#include <vector>

#include "mi355_lidar.h"

namespace mi355_pcl {

namespace detail {
inline mi355_cloud_view view(const pcl::PointCloud<pcl::PointXYZI>& c) {
  // pcl::PointXYZI: float data[4] at +0, intensity at +16, sizeof == 32
  return mi355_cloud_view{c.points.data(), c.points.size(), sizeof(pcl::PointXYZI), offsetof(pcl::PointXYZI, intensity)};
}
inline void check(mi355_ctx* ctx, int rc) {
  if (rc != MI355_OK) throw std::runtime_error(mi355_last_error(ctx));
}
// pcl::search::KdTree that indexes its cloud on the first query instead of in setInputCloud
class LazyKdTree : public pcl::search::KdTree<pcl::PointXYZI> {
 public:
  using Base = pcl::search::KdTree<pcl::PointXYZI>;
  using CloudConstPtr = Base::PointCloudConstPtr;
  void setInputCloud(const CloudConstPtr& cloud, const pcl::IndicesConstPtr& indices = pcl::IndicesConstPtr()) override {
    pending_ = cloud;
    pending_indices_ = indices;
    stale_ = true;
  }
  int nearestKSearch(const pcl::PointXYZI& p, int k, pcl::Indices& idx, std::vector<float>& d2) const override {
    if (stale_) {
      const_cast<LazyKdTree*>(this)->Base::setInputCloud(pending_, pending_indices_);
      stale_ = false;
    }
    return Base::nearestKSearch(p, k, idx, d2);
  }
 private:
  CloudConstPtr pending_;
  pcl::IndicesConstPtr pending_indices_;
  mutable bool stale_ = false;
};
}  // namespace detail

// what both engines share; the device fitness score is public so that the free function below can reach it
class DeviceRegistration : public pcl::Registration<pcl::PointXYZI, pcl::PointXYZI, float> {
 public:
  virtual double deviceFitnessScore(double max_range) = 0;
};

template <int METHOD, typename Params>
class RegistrationBase : public DeviceRegistration {
 public:
  using Base = pcl::Registration<pcl::PointXYZI, pcl::PointXYZI, float>;
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
  using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;
  using Matrix4 = typename Base::Matrix4;

  explicit RegistrationBase(int device_id = 0) : ctx_(mi355_create(device_id)), lazy_tree_(new detail::LazyKdTree()) {
    if (!ctx_) throw std::runtime_error("mi355_create failed: no HIP device");
    reg_ = mi355_reg_create(ctx_, METHOD, nullptr);
    if (!reg_) {
      mi355_destroy(ctx_);
      throw std::runtime_error("mi355_reg_create failed");
    }
    this->setSearchMethodTarget(lazy_tree_, /*force_no_recompute=*/true);  // initCompute() builds no CPU tree
  }
  ~RegistrationBase() override {
    mi355_reg_destroy(reg_);
    mi355_destroy(ctx_);
  }

  // The node clears and refills the SAME cloud object and passes the same pointer again (lidar_scan_matcher.cpp:187-212):
  // always rebuild, never short-circuit on pointer identity.
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {
    Base::setInputTarget(cloud);
    lazy_tree_->setInputCloud(cloud);  // remembered, not indexed
    push();
    const mi355_cloud_view v = detail::view(*cloud);
    detail::check(ctx_, mi355_reg_set_target(reg_, &v));
  }
  void setInputSource(const PointCloudSourceConstPtr& cloud) override {
    Base::setInputSource(cloud);
    push();
    const mi355_cloud_view v = detail::view(*cloud);
    detail::check(ctx_, mi355_reg_set_source(reg_, &v));
  }
  // same statistic as pcl::Registration::getFitnessScore (mean squared 1-NN distance of the aligned source), on the device.
  // Hides the base's (non-virtual) member: reached only through a pointer of THIS type or mi355_pcl::getFitnessScore().
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) { return deviceFitnessScore(max_range); }
  double deviceFitnessScore(double max_range) override {
    double s = 0;
    detail::check(ctx_, mi355_reg_fitness(reg_, max_range, &s));
    return s;
  }

 protected:
  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {
    push();
    float g[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) g[4 * r + c] = guess(r, c);  // Eigen is column-major, the C-ABI row-major
    mi355_result res;
    std::vector<float> packed(4 * std::max<std::size_t>(this->input_->size(), 1));
    detail::check(ctx_, mi355_reg_align(reg_, g, &res, packed.data()));
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) this->final_transformation_(r, c) = res.T[4 * r + c];
    this->converged_ = res.converged != 0;
    this->nr_iterations_ = res.iterations;
    onResult(res);
    output = *this->input_;
    for (std::size_t i = 0; i < output.size(); i++) {
      output[i].x = packed[4 * i]; output[i].y = packed[4 * i + 1]; output[i].z = packed[4 * i + 2];
    }
  }
  virtual void onResult(const mi355_result&) {}
  void push() {
    if (dirty_) {
      detail::check(ctx_, mi355_reg_set_params(reg_, &p_));
      dirty_ = false;
    }
  }
  mi355_ctx* ctx_;
  mi355_reg* reg_ = nullptr;
  pcl::shared_ptr<detail::LazyKdTree> lazy_tree_;
  Params p_;
  bool dirty_ = true;
};

// `registration_->getFitnessScore()` through the base pointer is PCL's CPU loop (the member is not virtual); this is the
// device version for any registration object, falling back to PCL's for engines that are not ours (graph_based_slam.cpp:321).
inline double getFitnessScore(const pcl::Registration<pcl::PointXYZI, pcl::PointXYZI, float>::Ptr& reg,
                              double max_range = std::numeric_limits<double>::max()) {
  if (auto* dev = dynamic_cast<DeviceRegistration*>(reg.get())) return dev->deviceFitnessScore(max_range);
  return reg->getFitnessScore(max_range);
}

// replaces `new fast_gicp::FastGICP<PointType, PointType>` (lidar_scan_matcher.cpp:38-39)
class FastGICP : public RegistrationBase<MI355_METHOD_FAST_GICP, mi355_gicp_params> {
 public:
  using Ptr = pcl::shared_ptr<FastGICP>;
  explicit FastGICP(int device_id = 0) : RegistrationBase(device_id) { mi355_gicp_default_params(&p_); reg_name_ = "mi355::FastGICP"; }
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; max_iterations_ = n; dirty_ = true; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; transformation_epsilon_ = e; dirty_ = true; }
  void setMaxCorrespondenceDistance(double d) { p_.max_correspondence_distance = d; corr_dist_threshold_ = d; dirty_ = true; }
  void setNumThreads(int n) { p_.num_threads = n; dirty_ = true; }
};

// the values of pclomp::NeighborSearchMethod (ndt_omp.h), for callers that do not have ndt_omp's header at hand
enum NeighborSearchMethod { KDTREE = 0, DIRECT26 = 1, DIRECT7 = 2, DIRECT1 = 3 };

// replaces `new pclomp::NormalDistributionsTransform<PointType, PointType>` (lidar_scan_matcher.cpp:56-57)
class NormalDistributionsTransform : public RegistrationBase<MI355_METHOD_NDT_OMP, mi355_ndt_params> {
 public:
  using Ptr = pcl::shared_ptr<NormalDistributionsTransform>;
  explicit NormalDistributionsTransform(int device_id = 0) : RegistrationBase(device_id) { mi355_ndt_default_params(&p_); reg_name_ = "mi355::NDT"; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; transformation_epsilon_ = e; dirty_ = true; }
  void setStepSize(double s) { p_.step_size = s; dirty_ = true; }
  void setResolution(float r) { p_.resolution = r; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; max_iterations_ = n; dirty_ = true; }
  // pclomp::NeighborSearchMethod is {KDTREE, DIRECT26, DIRECT7, DIRECT1}; the node asks for DIRECT7 (lidar_scan_matcher.cpp:69) and
  // DIRECT7 is the only search the device path has: anything else is refused, never silently replaced (host/pcl_like.hpp does the same)
  template <typename T> void setNeighborhoodSearchMethod(T method) {
    if (static_cast<int>(method) != static_cast<int>(DIRECT7))
      throw std::invalid_argument("mi355::NDT: only pclomp::DIRECT7 (lidar_scan_matcher.cpp:69) is built");
  }
  void setNumThreads(int n) { p_.num_threads = n; dirty_ = true; }
  double getTransformationProbability() const { return trans_probability_; }
 protected:
  void onResult(const mi355_result& r) override { trans_probability_ = r.trans_probability; }
 private:
  double trans_probability_ = 0;
};

// replaces `new pcl::IterativeClosestPoint<PointType, PointType>` (graph_based_slam.cpp:143-144: the back end's default engine)
class IterativeClosestPoint : public RegistrationBase<MI355_METHOD_ICP, mi355_icp_params> {
 public:
  using Ptr = pcl::shared_ptr<IterativeClosestPoint>;
  explicit IterativeClosestPoint(int device_id = 0) : RegistrationBase(device_id) { mi355_icp_default_params(&p_); reg_name_ = "mi355::ICP"; }
  void setMaxCorrespondenceDistance(double d) { p_.max_correspondence_distance = d; corr_dist_threshold_ = d; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; max_iterations_ = n; dirty_ = true; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; transformation_epsilon_ = e; dirty_ = true; }
  void setEuclideanFitnessEpsilon(double e) { p_.euclidean_fitness_epsilon = e; euclidean_fitness_epsilon_ = e; dirty_ = true; }
  void setRANSACIterations(int n) { p_.ransac_iterations = n; ransac_iterations_ = n; dirty_ = true; }  // only 0 is built
};

// replaces `new pclomp::GeneralizedIterativeClosestPoint<PointType, PointType>` (lidar_scan_matcher.cpp:74-75, graph_based_slam.cpp:122-123)
class GeneralizedIterativeClosestPoint : public RegistrationBase<MI355_METHOD_GICP, mi355_gicp_bfgs_params> {
 public:
  using Ptr = pcl::shared_ptr<GeneralizedIterativeClosestPoint>;
  explicit GeneralizedIterativeClosestPoint(int device_id = 0) : RegistrationBase(device_id) {
    mi355_gicp_bfgs_default_params(&p_);
    reg_name_ = "mi355::GICP";
  }
  void setMaxCorrespondenceDistance(double d) { p_.max_correspondence_distance = d; corr_dist_threshold_ = d; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; max_iterations_ = n; dirty_ = true; }
  void setMaximumOptimizerIterations(int n) { p_.max_optimizer_iterations = n; dirty_ = true; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; transformation_epsilon_ = e; dirty_ = true; }
  void setRotationEpsilon(double e) { p_.rotation_epsilon = e; dirty_ = true; }
  void setEuclideanFitnessEpsilon(double e) { p_.euclidean_fitness_epsilon = e; euclidean_fitness_epsilon_ = e; dirty_ = true; }
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; dirty_ = true; }
  void setRANSACIterations(int n) { p_.ransac_iterations = n; ransac_iterations_ = n; dirty_ = true; }  // only 0 is built
  void setUseReciprocalCorrespondences(bool) {}  // a base-class flag GICP's computeTransformation never reads; the nodes set false
};

}  // namespace mi355_pcl
