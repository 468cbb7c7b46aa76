// See scan_matcher_core.hpp.  Line references are to /root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp.
#include "scan_matcher_core.hpp"

#include "../../../include/mi355_scan_matcher.h"

namespace mi355 {

ScanMatcherCore::ScanMatcherCore(Context& ctx, const ScanMatcherParams& p) : ctx_(ctx), params_(p) {
  if (p.registration_method == "FAST_GICP") {  // :37-54
    auto gicp = std::make_shared<FastGICP>(ctx);
    gicp->setCorrespondenceRandomness(p.correspondence_randomness);
    gicp->setMaximumIterations(p.max_iteration);
    gicp->setTransformationEpsilon(p.transformation_epsilon);
    gicp->setMaxCorrespondenceDistance(p.max_correspondence_distance);
    if (0 < p.omp_num_thread) gicp->setNumThreads(p.omp_num_thread);
    registration_ = gicp;
    gicp_ = true;
  } else if (p.registration_method == "NDT_OMP") {  // :55-72
    auto ndt = std::make_shared<NormalDistributionsTransform>(ctx);
    ndt->setTransformationEpsilon(p.transformation_epsilon);
    ndt->setStepSize(p.step_size);
    ndt->setResolution((float)p.ndt_resolution);
    ndt->setMaximumIterations(p.max_iteration);
    ndt->setNeighborhoodSearchMethod(DIRECT7);
    if (0 < p.omp_num_thread) ndt->setNumThreads(p.omp_num_thread);
    registration_ = ndt;
  } else if (p.registration_method == "GICP") {  // :73-96
    auto gicp = std::make_shared<GeneralizedIterativeClosestPoint>(ctx);
    gicp->setMaxCorrespondenceDistance(p.correspondence_distance);
    gicp->setMaximumIterations(p.max_iteration);
    gicp->setUseReciprocalCorrespondences(p.use_reciprocal_correspondences);
    gicp->setMaximumOptimizerIterations(p.max_optimizer_iteration);
    gicp->setTransformationEpsilon(p.transformation_epsilon);
    gicp->setEuclideanFitnessEpsilon(p.euclidean_fitness_epsilon);
    gicp->setCorrespondenceRandomness(p.correspondence_randomness);
    registration_ = gicp;
    gicp_ = true;  // same PLANE k-NN covariances as FAST_GICP: a key frame's can be carried into the sub-map
  } else {
    // the reference leaves registration_ null and crashes at the first callback (:97); fail at construction instead
    throw Error(MI355_EINVAL, "registration_method must be FAST_GICP, NDT_OMP or GICP");
  }
}

PointCloudPtr ScanMatcherCore::target_cloud() const {
  auto out = std::make_shared<PointCloud>();
  if (target_cloud_) target_cloud_->download(*out);
  return out;
}

ScanOutput ScanMatcherCore::callback_cloud(const PointCloud& input_cloud, const Matrix4f& base_to_sensor) {
  ScanOutput out;
  // the scan crosses PCIe once; everything below works on device clouds
  DeviceCloud::Ptr sensor_cloud = DeviceCloud::upload(ctx_, input_cloud);
  DeviceCloud::Ptr base_to_sensor_cloud = sensor_cloud->transformed(base_to_sensor);  // :129-131

  if (!target_cloud_) {  // :133-160
    prev_translation_ = identity4();
    key_frame_ = identity4();
    KeyFrame key_frame;
    key_frame.pose = convert_matrix_to_pose(prev_translation_);
    key_frame.id = id_++;
    key_frame.device = sensor_cloud;  // key frame #0 keeps the SENSOR-frame cloud (:145) — reproduced as is
    key_frame_array_.emplace_back(std::move(key_frame));
    target_cloud_ = base_to_sensor_cloud;
    registration_->setInputTarget(target_cloud_);
    out.first_scan = true;
    out.new_key_frame = true;
    out.key_frame_id = 0;
    out.target_points = target_cloud_->size();
    return out;
  }

  registration_->setInputSource(base_to_sensor_cloud);  // :162 (FAST_GICP: the scan's covariances stay with the device cloud)
  registration_->align(prev_translation_);              // :165
  out.iterations = registration_->result().iterations;
  if (!registration_->hasConverged()) {  // :167-170 — scan dropped, pose frozen
    out.converged = false;
    out.pose = prev_translation_;
    out.target_points = target_cloud_->size();
    return out;
  }
  out.converged = true;
  translation_ = registration_->getFinalTransformation();  // :172
  prev_translation_ = translation_;                        // :178
  out.pose = translation_;

  const double dx = (double)translation_[3] - (double)key_frame_[3];
  const double dy = (double)translation_[7] - (double)key_frame_[7];
  const double dz = (double)translation_[11] - (double)key_frame_[11];
  const double delta = std::sqrt(dx * dx + dy * dy + dz * dz);  // :180-182
  if (params_.displacement <= delta) {                          // :183
    key_frame_ = translation_;
    accum_distance_ += delta;

    KeyFrame key_frame;
    key_frame.pose = convert_matrix_to_pose(key_frame_);
    key_frame.accum_distance = accum_distance_;
    key_frame.id = id_++;
    key_frame.device = base_to_sensor_cloud;  // :196
    key_frame_array_.emplace_back(std::move(key_frame));

    // :187,199-210 clear the target and refill it with the last N key frames, each moved by its own pose: ONE device
    // cloud built by transform kernels writing at running offsets
    std::vector<DeviceCloud::Ptr> parts;
    std::vector<Matrix4f> poses;
    const bool carry = gicp_ && params_.reuse_key_frame_covariances;
    const int sub_map_size = (int)key_frame_array_.size();
    for (int idx = 0; idx < params_.max_scan_accumulate_num; idx++) {
      if ((sub_map_size - 1 - idx) < 0) continue;
      const KeyFrame& kf = key_frame_array_[(size_t)(sub_map_size - 1 - idx)];
      // key frame #0 was the first target, never a source: its covariances are computed here, once
      if (carry && !kf.device->has_covariances()) kf.device->compute_covariances(params_.correspondence_randomness, MI355_REG_PLANE);
      parts.push_back(kf.device);
      poses.push_back(geometry_pose_to_matrix(kf.pose));
    }
    target_cloud_ = DeviceCloud::concat(ctx_, parts, poses, carry);
    registration_->setInputTarget(target_cloud_);  // :212
    // only the last max_scan_accumulate_num key frames are ever read again: the one that has just left the window moves to
    // the host (where the reference keeps every key_frame.cloud, :196) and gives its HBM back - points and covariances
    const int leaving = sub_map_size - 1 - params_.max_scan_accumulate_num;
    if (leaving >= 0 && key_frame_array_[(size_t)leaving].device) {
      KeyFrame& old = key_frame_array_[(size_t)leaving];
      old.device->download(old.cloud);
      old.device.reset();
    }
    out.new_key_frame = true;
    out.key_frame_id = key_frame_array_.back().id;
  }
  out.target_points = target_cloud_->size();
  return out;
}

}  // namespace mi355

// ---- C-ABI (include/mi355_scan_matcher.h) ----------------------------------------------------------------------------
struct mi355_scan_matcher {
  std::unique_ptr<mi355::Context> ctx;
  std::unique_ptr<mi355::ScanMatcherCore> core;
  std::string err;
};

extern "C" mi355_scan_matcher* mi355_scan_matcher_create(int device_id, const mi355_scan_matcher_params* p) {
  if (!p) return nullptr;
  auto* h = new mi355_scan_matcher();
  try {
    h->ctx.reset(new mi355::Context(device_id));
    mi355::ScanMatcherParams sp;
    sp.registration_method = p->registration_method == MI355_METHOD_FAST_GICP ? "FAST_GICP"
                             : (p->registration_method == MI355_METHOD_NDT_OMP ? "NDT_OMP" : (p->registration_method == MI355_METHOD_GICP ? "GICP" : "?"));
    sp.correspondence_distance = p->correspondence_distance;
    sp.max_optimizer_iteration = p->max_optimizer_iteration;
    sp.displacement = p->displacement;
    sp.max_scan_accumulate_num = p->max_scan_accumulate_num;
    sp.max_iteration = p->max_iteration;
    sp.omp_num_thread = p->omp_num_thread;
    sp.transformation_epsilon = p->transformation_epsilon;
    sp.step_size = p->step_size;
    sp.ndt_resolution = p->ndt_resolution;
    sp.max_correspondence_distance = p->max_correspondence_distance;
    sp.correspondence_randomness = p->correspondence_randomness;
    sp.reuse_key_frame_covariances = p->reuse_key_frame_covariances != 0;
    h->core.reset(new mi355::ScanMatcherCore(*h->ctx, sp));
  } catch (const std::exception&) {
    delete h;
    return nullptr;
  }
  return h;
}

extern "C" void mi355_scan_matcher_destroy(mi355_scan_matcher* h) {
  if (!h) return;
  h->core.reset();
  h->ctx.reset();
  delete h;
}

extern "C" const char* mi355_scan_matcher_last_error(const mi355_scan_matcher* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int mi355_scan_matcher_callback_cloud(mi355_scan_matcher* h, const mi355_cloud_view* cloud, const float base_to_sensor16[16],
                                                 mi355_scan_output* out) {
  if (!h || !cloud || !out) return MI355_EINVAL;
  if (cloud->n && !cloud->data) return MI355_EINVAL;
  if (cloud->stride < 16 || cloud->stride % 4 || cloud->intensity_offset < 12 || cloud->intensity_offset + 4 > cloud->stride) return MI355_EINVAL;
  try {
    mi355::PointCloud in(cloud->n);
    const char* base = (const char*)cloud->data;
    for (size_t i = 0; i < cloud->n; i++) {
      const float* p = (const float*)(base + i * cloud->stride);
      in[i].x = p[0]; in[i].y = p[1]; in[i].z = p[2]; in[i].w = 1.f;
      in[i].intensity = *(const float*)(base + i * cloud->stride + cloud->intensity_offset);
      in[i].pad[0] = in[i].pad[1] = in[i].pad[2] = 0.f;
    }
    mi355::Matrix4f T = mi355::identity4();
    if (base_to_sensor16) std::memcpy(T.data(), base_to_sensor16, sizeof(float) * 16);
    const mi355::ScanOutput o = h->core->callback_cloud(in, T);
    out->first_scan = o.first_scan;
    out->converged = o.converged;
    out->new_key_frame = o.new_key_frame;
    std::memcpy(out->pose, o.pose.data(), sizeof(out->pose));
    out->key_frame_id = o.key_frame_id;
    out->target_points = o.target_points;
    out->iterations = o.iterations;
    out->n_key_frames = (long)h->core->key_frames().size();
    out->accum_distance = h->core->accum_distance();
    return MI355_OK;
  } catch (const mi355::Error& e) {
    h->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    h->err = e.what();
    return MI355_EINVAL;
  }
}

extern "C" int mi355_scan_matcher_target(mi355_scan_matcher* h, float* out_packed, size_t cap, size_t* n) {
  if (!h || !n) return MI355_EINVAL;
  *n = h->core->target_points();
  if (!out_packed) return MI355_OK;
  if (*n > cap) return MI355_ECAPACITY;
  const auto t = h->core->target_cloud();
  for (size_t i = 0; i < *n; i++) {
    out_packed[4 * i] = (*t)[i].x; out_packed[4 * i + 1] = (*t)[i].y; out_packed[4 * i + 2] = (*t)[i].z; out_packed[4 * i + 3] = (*t)[i].intensity;
  }
  return MI355_OK;
}
