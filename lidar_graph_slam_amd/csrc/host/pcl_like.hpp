// C++ host mirror of the PCL-facing interface the reference's nodes program against, implemented over the C-ABI
// (include/mi355_lidar.h).  Same method names and call order as the reference's call sites so that node code ports by
// changing the concrete type only:
//   pcl::VoxelGrid / pcl::StatisticalOutlierRemoval      points_prefiltering.cpp:118-120,136-139,146-148
//   pcl::Registration::setInputTarget/Source/align/...   lidar_scan_matcher.cpp:149,162,165-172; graph_based_slam.cpp:315-322
//   fast_gicp::FastGICP setters                          lidar_scan_matcher.cpp:48-52
//   pclomp::NormalDistributionsTransform setters         lidar_scan_matcher.cpp:65-70
// No PCL, Eigen or ROS types: clouds are std::vector<mi355::PointXYZI> (the 32-byte pcl::PointXYZI memory layout),
// matrices are row-major float[16].  A PCL-typed adapter lives in adapters/pcl_registration.hpp.
#pragma once
#include <array>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mi355_lidar.h"

namespace mi355 {

struct alignas(16) PointXYZI {  // pcl::PointXYZI: float data[4] = {x,y,z,1}; float intensity; 12 bytes padding
  float x, y, z, w;
  float intensity, pad[3];
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");

using PointCloud = std::vector<PointXYZI>;
using PointCloudPtr = std::shared_ptr<PointCloud>;
using Matrix4f = std::array<float, 16>;  // row-major

inline Matrix4f identity4() {
  Matrix4f m{};
  m[0] = m[5] = m[10] = m[15] = 1.f;
  return m;
}

inline mi355_cloud_view view_of(const PointCloud& c) { return mi355_cloud_view{c.data(), c.size(), sizeof(PointXYZI), offsetof(PointXYZI, intensity)}; }

// packed float4 (x,y,z,intensity) -> PointXYZI
inline void unpack(const float* packed, size_t n, PointCloud& out) {
  out.resize(n);
  for (size_t i = 0; i < n; i++) {
    out[i].x = packed[4 * i]; out[i].y = packed[4 * i + 1]; out[i].z = packed[4 * i + 2]; out[i].w = 1.f;
    out[i].intensity = packed[4 * i + 3];
    out[i].pad[0] = out[i].pad[1] = out[i].pad[2] = 0.f;
  }
}

class Error : public std::runtime_error {
 public:
  Error(int code, const std::string& what) : std::runtime_error(what), code(code) {}
  int code;
};

class Context {
 public:
  explicit Context(int device_id = 0) : ctx_(mi355_create(device_id)) {
    if (!ctx_) throw Error(MI355_EHIP, "mi355_create failed: no HIP device (there is no CPU fallback)");
  }
  ~Context() { mi355_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  mi355_ctx* get() const { return ctx_; }
  void check(int rc) const {
    if (rc != MI355_OK) throw Error(rc, mi355_last_error(ctx_));
  }

 private:
  mi355_ctx* ctx_;
};

// ---- device-resident clouds (include/mi355_lidar.h: mi355_cloud) ------------------------------------------------------
// A key frame's cloud lives in HBM from the moment the scan arrives: transformed, concatenated into sub-maps and handed to the
// registration without crossing PCIe again (the reference does all of that on the CPU, lidar_scan_matcher.cpp:187-212).
class DeviceCloud {
 public:
  using Ptr = std::shared_ptr<DeviceCloud>;
  static Ptr upload(Context& ctx, const PointCloud& cloud) {
    const mi355_cloud_view v = view_of(cloud);
    return wrap(ctx, mi355_cloud_upload(ctx.get(), &v));
  }
  static Ptr upload_view(Context& ctx, const mi355_cloud_view& v) { return wrap(ctx, mi355_cloud_upload(ctx.get(), &v)); }
  // pcl::transformPointCloud (lidar_scan_matcher.cpp:275-294)
  Ptr transformed(const Matrix4f& T) const { return wrap(ctx_, mi355_cloud_transform(h_, T.data())); }
  // T_0 c_0 ++ T_1 c_1 ++ ... (lidar_scan_matcher.cpp:199-210; graph_based_slam.cpp:297-309)
  static Ptr concat(Context& ctx, const std::vector<Ptr>& clouds, const std::vector<Matrix4f>& Ts, bool carry_covariances) {
    std::vector<mi355_cloud*> hs;
    std::vector<float> T16;
    for (size_t i = 0; i < clouds.size(); i++) {
      hs.push_back(clouds[i]->h_);
      T16.insert(T16.end(), Ts[i].begin(), Ts[i].end());
    }
    return wrap(ctx, mi355_cloud_concat(ctx.get(), hs.data(), T16.data(), (int)hs.size(), carry_covariances ? 1 : 0));
  }
  ~DeviceCloud() { mi355_cloud_destroy(h_); }
  DeviceCloud(const DeviceCloud&) = delete;
  DeviceCloud& operator=(const DeviceCloud&) = delete;
  size_t size() const { return mi355_cloud_size(h_); }
  bool has_covariances() const { return mi355_cloud_has_covariances(h_) != 0; }
  void compute_covariances(int k, int method) { ctx_.check(mi355_cloud_compute_covariances(h_, k, method)); }
  // points_prefiltering.cpp:66-100 on the device (VoxelGrid alone: graph_based_slam.cpp:311-313)
  Ptr prefiltered(const mi355_prefilter_params& pf) const { return wrap(ctx_, mi355_cloud_prefilter(h_, &pf)); }
  void download(PointCloud& out) const {
    std::vector<float> packed(4 * std::max<size_t>(size(), 1));
    ctx_.check(mi355_cloud_download(h_, packed.data(), packed.size() / 4));
    unpack(packed.data(), size(), out);
  }
  mi355_cloud* get() const { return h_; }

 private:
  DeviceCloud(Context& ctx, mi355_cloud* h) : ctx_(ctx), h_(h) {}
  static Ptr wrap(Context& ctx, mi355_cloud* h) {
    if (!h) throw Error(MI355_EHIP, mi355_last_error(ctx.get()));
    return Ptr(new DeviceCloud(ctx, h));
  }
  Context& ctx_;
  mi355_cloud* h_;
};

// ---- filters --------------------------------------------------------------------------------------------------------
class VoxelGrid {  // pcl::VoxelGrid<PointXYZI>
 public:
  explicit VoxelGrid(Context& ctx) : ctx_(ctx) {}
  void setLeafSize(float lx, float ly, float lz) {
    if (lx != ly || ly != lz) throw Error(MI355_EINVAL, "cubic leaves only (the reference passes leaf_size_ three times)");
    leaf_ = lx;
  }
  void setInputCloud(const PointCloudPtr& cloud) { input_ = cloud; }
  void filter(PointCloud& output) {
    if (!input_) throw Error(MI355_ESTATE, "setInputCloud has not been called");
    std::vector<float> out(4 * std::max<size_t>(input_->size(), 1));
    size_t n = 0;
    int overflow = 0;
    const mi355_cloud_view v = view_of(*input_);
    ctx_.check(mi355_voxel_grid(ctx_.get(), &v, leaf_, out.data(), out.size() / 4, &n, &overflow));
    unpack(out.data(), n, output);
  }

 private:
  Context& ctx_;
  float leaf_ = 0.1f;
  PointCloudPtr input_;
};

class StatisticalOutlierRemoval {  // pcl::StatisticalOutlierRemoval<PointXYZI>
 public:
  explicit StatisticalOutlierRemoval(Context& ctx) : ctx_(ctx) {}
  void setMeanK(int k) { mean_k_ = k; }
  void setStddevMulThresh(double m) { stddev_ = m; }
  void setInputCloud(const PointCloudPtr& cloud) { input_ = cloud; }
  void filter(PointCloud& output) {
    if (!input_) throw Error(MI355_ESTATE, "setInputCloud has not been called");
    std::vector<float> out(4 * std::max<size_t>(input_->size(), 1));
    size_t n = 0;
    const mi355_cloud_view v = view_of(*input_);
    ctx_.check(mi355_sor(ctx_.get(), &v, mean_k_, stddev_, out.data(), out.size() / 4, &n, nullptr, nullptr));
    unpack(out.data(), n, output);
  }

 private:
  Context& ctx_;
  int mean_k_ = 1;
  double stddev_ = 0.0;
  PointCloudPtr input_;
};

// ---- registration ---------------------------------------------------------------------------------------------------
class Registration {  // pcl::Registration<PointXYZI, PointXYZI>
 public:
  using Ptr = std::shared_ptr<Registration>;
  virtual ~Registration() {
    if (reg_) mi355_reg_destroy(reg_);
  }
  // NB: always rebuilds — the node mutates the target cloud in place and re-passes the same pointer
  // (lidar_scan_matcher.cpp:187,209,212), so pointer identity must not short-circuit.
  void setInputTarget(const PointCloudPtr& cloud) {
    push_params();
    target_ = cloud;
    const mi355_cloud_view v = view_of(*cloud);
    ctx_.check(mi355_reg_set_target(reg_, &v));
  }
  void setInputSource(const PointCloudPtr& cloud) {
    push_params();
    source_ = cloud;
    const mi355_cloud_view v = view_of(*cloud);
    ctx_.check(mi355_reg_set_source(reg_, &v));
  }
  // the same from clouds that already live on the device
  void setInputTarget(const DeviceCloud::Ptr& cloud) {
    push_params();
    target_.reset();
    ctx_.check(mi355_reg_set_target_cloud(reg_, cloud->get()));
  }
  void setInputSource(const DeviceCloud::Ptr& cloud) {
    push_params();
    source_.reset();
    ctx_.check(mi355_reg_set_source_cloud(reg_, cloud->get()));
  }
  void align(PointCloud& output) { align(output, identity4()); }
  void align(PointCloud& output, const Matrix4f& guess) {
    push_params();
    std::vector<float> out(4 * std::max<size_t>(source_ ? source_->size() : 0, 1));
    ctx_.check(mi355_reg_align(reg_, guess.data(), &result_, out.data()));
    unpack(out.data(), source_ ? source_->size() : 0, output);
  }
  // align without materialising the aligned cloud (the node never uses it: lidar_scan_matcher.cpp:164-165)
  void align(const Matrix4f& guess) {
    push_params();
    ctx_.check(mi355_reg_align(reg_, guess.data(), &result_, nullptr));
  }
  bool hasConverged() const { return result_.converged != 0; }
  Matrix4f getFinalTransformation() const {
    Matrix4f m;
    std::memcpy(m.data(), result_.T, sizeof(result_.T));
    return m;
  }
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double s = 0;
    ctx_.check(mi355_reg_fitness(reg_, max_range, &s));
    return s;
  }
  const mi355_result& result() const { return result_; }

 protected:
  Registration(Context& ctx, int method, const void* params) : ctx_(ctx), reg_(mi355_reg_create(ctx.get(), method, params)) {
    if (!reg_) throw Error(MI355_EINVAL, "mi355_reg_create failed");
    std::memset(&result_, 0, sizeof(result_));
    const Matrix4f I = identity4();
    std::memcpy(result_.T, I.data(), sizeof(result_.T));
  }
  virtual const void* params_ptr() const = 0;
  void push_params() {
    if (dirty_) {
      ctx_.check(mi355_reg_set_params(reg_, params_ptr()));
      dirty_ = false;
    }
  }
  Context& ctx_;
  mi355_reg* reg_;
  mi355_result result_;
  PointCloudPtr source_, target_;
  bool dirty_ = false;
};

class FastGICP : public Registration {  // fast_gicp::FastGICP<PointXYZI, PointXYZI>
 public:
  explicit FastGICP(Context& ctx) : Registration(ctx, MI355_METHOD_FAST_GICP, nullptr) { mi355_gicp_default_params(&p_); }
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; dirty_ = true; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; dirty_ = true; }
  void setRotationEpsilon(double e) { p_.rotation_epsilon = e; dirty_ = true; }
  void setMaxCorrespondenceDistance(double d) { p_.max_correspondence_distance = d; dirty_ = true; }
  void setRegularizationMethod(int m) { p_.regularization_method = m; dirty_ = true; }
  void setNumThreads(int n) { p_.num_threads = n; dirty_ = true; }  // interface parity; the GPU path has no thread count

 protected:
  const void* params_ptr() const override { return &p_; }
  mi355_gicp_params p_;
};

enum NeighborSearchMethod { KDTREE, DIRECT26, DIRECT7, DIRECT1 };

class NormalDistributionsTransform : public Registration {  // pclomp::NormalDistributionsTransform<PointXYZI, PointXYZI>
 public:
  explicit NormalDistributionsTransform(Context& ctx) : Registration(ctx, MI355_METHOD_NDT_OMP, nullptr) { mi355_ndt_default_params(&p_); }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; dirty_ = true; }
  void setStepSize(double s) { p_.step_size = s; dirty_ = true; }
  void setResolution(float r) { p_.resolution = r; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; dirty_ = true; }
  void setNeighborhoodSearchMethod(NeighborSearchMethod m) {
    if (m != DIRECT7) throw Error(MI355_EINVAL, "only pclomp::DIRECT7 (lidar_scan_matcher.cpp:69) is built");
  }
  void setNumThreads(int n) { p_.num_threads = n; dirty_ = true; }
  double getTransformationProbability() const { return result_.trans_probability; }

 protected:
  const void* params_ptr() const override { return &p_; }
  mi355_ndt_params p_;
};

class GeneralizedIterativeClosestPoint : public Registration {  // pclomp::GeneralizedIterativeClosestPoint<PointXYZI, PointXYZI>
 public:
  explicit GeneralizedIterativeClosestPoint(Context& ctx) : Registration(ctx, MI355_METHOD_GICP, nullptr) { mi355_gicp_bfgs_default_params(&p_); }
  void setMaxCorrespondenceDistance(double d) { p_.max_correspondence_distance = d; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; dirty_ = true; }
  void setMaximumOptimizerIterations(int n) { p_.max_optimizer_iterations = n; dirty_ = true; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; dirty_ = true; }
  void setRotationEpsilon(double e) { p_.rotation_epsilon = e; dirty_ = true; }
  void setEuclideanFitnessEpsilon(double e) { p_.euclidean_fitness_epsilon = e; dirty_ = true; }
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; dirty_ = true; }
  void setRANSACIterations(int n) { p_.ransac_iterations = n; dirty_ = true; }  // only 0 is built
  void setUseReciprocalCorrespondences(bool on) {  // a pcl::Registration flag GICP's computeTransformation never reads
    if (on) throw Error(MI355_EINVAL, "reciprocal correspondences are not built (the nodes set false)");
  }

 protected:
  const void* params_ptr() const override { return &p_; }
  mi355_gicp_bfgs_params p_;
};

class IterativeClosestPoint : public Registration {  // pcl::IterativeClosestPoint<PointXYZI, PointXYZI>
 public:
  explicit IterativeClosestPoint(Context& ctx) : Registration(ctx, MI355_METHOD_ICP, nullptr) { mi355_icp_default_params(&p_); }
  void setMaxCorrespondenceDistance(double d) { p_.max_correspondence_distance = d; dirty_ = true; }
  void setMaximumIterations(int n) { p_.max_iterations = n; dirty_ = true; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; dirty_ = true; }
  void setEuclideanFitnessEpsilon(double e) { p_.euclidean_fitness_epsilon = e; dirty_ = true; }
  void setRANSACIterations(int n) { p_.ransac_iterations = n; dirty_ = true; }  // only 0 is built (graph_based_slam.cpp:149)

 protected:
  const void* params_ptr() const override { return &p_; }
  mi355_icp_params p_;
};

}  // namespace mi355
