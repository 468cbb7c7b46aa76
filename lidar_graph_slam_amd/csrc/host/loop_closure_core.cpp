// See loop_closure_core.hpp.  Line references are to /root/reference/graph_based_slam/src/graph_based_slam.cpp.
#include "loop_closure_core.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

#include "../../../include/mi355_loop_closure.h"

namespace mi355 {

int LoopClosureCore::add_key_frame(PointCloud cloud, const Pose& pose, double accum_distance) {
  KeyFrame kf;
  kf.cloud = std::move(cloud);
  kf.pose = pose;
  kf.accum_distance = accum_distance;
  kf.id = (long)key_frames_.size();
  key_frames_.emplace_back(std::move(kf));
  return (int)key_frames_.size() - 1;
}

int LoopClosureCore::candidate(int query) const {
  if (query < 0 || (size_t)query >= key_frames_.size()) throw Error(MI355_EINVAL, "query key frame out of range");
  const KeyFrame& latest = key_frames_[(size_t)query];
  double min_dist = std::numeric_limits<double>::max();
  int min_id = -1;
  for (int id = 0; id < (int)key_frames_.size(); id++) {  // :264-280 (every key frame is tested, also later ones)
    const KeyFrame& kf = key_frames_[(size_t)id];
    if ((latest.accum_distance - kf.accum_distance) < params_.accumulate_distance_threshold) continue;
    const double dx = latest.pose.px - kf.pose.px, dy = latest.pose.py - kf.pose.py, dz = latest.pose.pz - kf.pose.pz;
    const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (dist < params_.search_for_candidate_threshold && dist < min_dist) {
      min_dist = dist;
      min_id = id;
    }
  }
  return min_id;
}

LoopClosureCore::LoopClosureCore(const std::vector<int>& device_ids, const LoopClosureParams& p) : params_(p) {
  if (device_ids.empty()) throw Error(MI355_EINVAL, "LoopClosureCore needs at least one device");
  if (p.ransac_iteration != 0) throw Error(MI355_EINVAL, "ransac_iteration: only 0 (no RANSAC rejector, the yaml's value) is built");
  for (int d : device_ids) lanes_.emplace_back(new LaneState(d));
}

LoopClosureCore::~LoopClosureCore() = default;  // LaneState: clouds and registration objects go before their lane's context

// double-precision rigid transform from a geometry Pose (what gtsam::Pose3(Rot3::Quaternion, Point3) holds)
static void pose_to_matrix_d(const Pose& p, double M[16]) {
  const double n = std::sqrt(p.qx * p.qx + p.qy * p.qy + p.qz * p.qz + p.qw * p.qw);
  const double x = p.qx / n, y = p.qy / n, z = p.qz / n, w = p.qw / n;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
                       2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y)};
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) M[4 * r + c] = R[3 * r + c];
  }
  M[3] = p.px; M[7] = p.py; M[11] = p.pz;
  M[12] = M[13] = M[14] = 0.0;
  M[15] = 1.0;
}

void LoopClosureCore::register_rows(LaneState& ls, const std::vector<int>& rows, const std::vector<LoopResult>& out, std::vector<mi355_result>& res,
                                    std::vector<double>& fit) {
  Context& ctx = ls.lane.ctx();
  ls.key_frame.resize(key_frames_.size());
  auto on_device = [&](int k) -> const DeviceCloud::Ptr& {  // one upload per key frame and lane, ever
    if (!ls.key_frame[(size_t)k]) ls.key_frame[(size_t)k] = DeviceCloud::upload(ctx, key_frames_[(size_t)k].cloud);
    return ls.key_frame[(size_t)k];
  };
  // sub-maps (:297-309): key frames min_id - N .. min_id + N, each moved by its own pose, concatenated in index order —
  // transform kernels writing one device cloud per sub-map
  const int n_kf = (int)key_frames_.size();
  std::vector<int> uniq;  // candidate ids of this lane, in order of first use
  for (int i : rows)
    if (std::find(uniq.begin(), uniq.end(), out[(size_t)i].candidate_id) == uniq.end()) uniq.push_back(out[(size_t)i].candidate_id);
  std::vector<DeviceCloud::Ptr> submap(uniq.size());
  for (size_t u = 0; u < uniq.size(); u++) {
    std::vector<DeviceCloud::Ptr> parts;
    std::vector<Matrix4f> poses;
    for (int idx = -params_.search_key_frame_num; idx <= params_.search_key_frame_num; idx++) {
      const int k = uniq[u] + idx;
      if (k < 0 || n_kf <= k) continue;
      parts.push_back(on_device(k));
      poses.push_back(geometry_pose_to_matrix(key_frames_[(size_t)k].pose));
    }
    submap[u] = DeviceCloud::concat(ctx, parts, poses, /*carry_covariances=*/false);  // voxel-filtered next: new points
  }
  // sources (:248-252): the query key frame moved by its pose
  std::vector<DeviceCloud::Ptr> source;
  std::vector<int> tgt_of_src;
  for (int i : rows) {
    const int q = out[(size_t)i].query_id;
    source.push_back(on_device(q)->transformed(geometry_pose_to_matrix(key_frames_[(size_t)q].pose)));
    tgt_of_src.push_back((int)(std::find(uniq.begin(), uniq.end(), out[(size_t)i].candidate_id) - uniq.begin()));
  }
  const int S = (int)source.size();
  std::vector<mi355_result> r((size_t)S);
  std::vector<double> f((size_t)S);
  const mi355_prefilter_params pf{0.0, params_.submap_leaf, 0, 0.0};  // voxel_grid_.filter (:311-313): VoxelGrid only
  {
    mi355_gicp_params gp{};
    mi355_gicp_default_params(&gp);  // fast_gicp defaults, then the setters of :92-96
    gp.k_correspondences = params_.correspondence_randomness;
    gp.max_iterations = params_.max_iteration;
    gp.transformation_epsilon = params_.transformation_epsilon;
    gp.max_correspondence_distance = params_.max_correspondence_distance;
    mi355_ndt_params np{};
    mi355_ndt_default_params(&np);   // :108-116
    np.transformation_epsilon = params_.transformation_epsilon;
    np.step_size = params_.step_size;
    np.resolution = params_.ndt_resolution;
    np.max_iterations = params_.max_iteration;
    mi355_icp_params ip{};
    mi355_icp_default_params(&ip);   // :145-149 (hard-coded there; the yaml's ransac_iteration is read for "GICP" only)
    ip.max_correspondence_distance = 30;
    ip.max_iterations = 100;
    ip.transformation_epsilon = 1e-8;
    ip.euclidean_fitness_epsilon = 1e-6;
    ip.ransac_iterations = 0;
    mi355_gicp_bfgs_params bp{};
    mi355_gicp_bfgs_default_params(&bp);  // pclomp's BFGS GICP, :133-139
    bp.max_correspondence_distance = params_.correspondence_distance;
    bp.max_iterations = params_.max_iteration;
    bp.max_optimizer_iterations = params_.max_optimizer_iteration;
    bp.transformation_epsilon = params_.transformation_epsilon;
    bp.euclidean_fitness_epsilon = params_.euclidean_fitness_epsilon;
    bp.ransac_iterations = params_.ransac_iteration;  // :131,:139 (only 0 is built: mi355_loop_closure_create refuses anything else)
    const int method = params_.registration_method;
    const void* mp = method == MI355_METHOD_NDT_OMP ? (const void*)&np
                     : method == MI355_METHOD_ICP   ? (const void*)&ip
                     : method == MI355_METHOD_GICP  ? (const void*)&bp
                                                    : (const void*)&gp;
    const size_t mps = method == MI355_METHOD_NDT_OMP ? sizeof(np) : method == MI355_METHOD_ICP ? sizeof(ip) : method == MI355_METHOD_GICP ? sizeof(bp) : sizeof(gp);
    std::vector<mi355_cloud*> th, sh;
    for (auto& c : submap) th.push_back(c->get());
    for (auto& c : source) sh.push_back(c->get());
    // the lane's batch object outlives close() (grow-only) and runs its live slots only
    ls.lane.register_batch(method, mp, mps, sh, th, tgt_of_src,
                           nullptr /* align(*output): identity guess, :318 */, nullptr, &pf, /*with_fitness=*/true, r.data(), f.data());
  }
  for (int s = 0; s < S; s++) {
    res[(size_t)rows[(size_t)s]] = r[(size_t)s];
    fit[(size_t)rows[(size_t)s]] = f[(size_t)s];
  }
}

std::vector<LoopResult> LoopClosureCore::close(const std::vector<int>& queries) {
  std::vector<LoopResult> out(queries.size());
  std::vector<int> rows, cand_of_row;  // rows of `out` that are registered, and their candidates
  for (size_t i = 0; i < queries.size(); i++) {
    out[i].query_id = queries[i];
    out[i].candidate_id = candidate(queries[i]);
    if (out[i].candidate_id >= 0) {
      rows.push_back((int)i);
      cand_of_row.push_back(out[i].candidate_id);
    }
  }
  if (rows.empty()) return out;

  // deal the registered rows to the lanes by candidate (= sub-map): a sub-map is assembled, filtered and indexed on one lane only
  std::vector<mi355_result> res(queries.size());
  std::vector<double> fit(queries.size(), 0.0);
  const auto shards = shard_by_target(cand_of_row.data(), (int)rows.size(), (int)lanes_.size());
  std::vector<int> busy;
  for (int l = 0; l < (int)lanes_.size(); l++)
    if (!shards[(size_t)l].empty()) busy.push_back(l);
  for_each_lane(busy, [&](int l) {
    std::vector<int> mine;
    for (int k : shards[(size_t)l]) mine.push_back(rows[(size_t)k]);
    register_rows(*lanes_[(size_t)l], mine, out, res, fit);
  });

  const int S = (int)rows.size();
  const std::vector<int>& src_of = rows;
  for (int s = 0; s < S; s++) {
    LoopResult& r = out[(size_t)src_of[(size_t)s]];
    const mi355_result& m = res[(size_t)src_of[(size_t)s]];
    std::copy(m.T, m.T + 16, r.transform.begin());
    r.converged = m.converged != 0;
    r.iterations = m.iterations;
    r.fitness = fit[(size_t)src_of[(size_t)s]];
    r.target_points = m.n_target;
    r.accepted = r.converged && !(params_.score_threshold < r.fitness);  // :328
    // :331-340  pose_from = transform * pose(query) through geometry_msgs::Pose, pose_to = pose(candidate)
    const Pose from = convert_matrix_to_pose(matmul(r.transform, geometry_pose_to_matrix(key_frames_[(size_t)r.query_id].pose)));
    double A[16], B[16];
    pose_to_matrix_d(from, A);
    pose_to_matrix_d(key_frames_[(size_t)r.candidate_id].pose, B);
    // between = A^-1 * B, A rigid
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) {
        double acc = 0.0;
        for (int k = 0; k < 3; k++) acc += A[4 * k + i] * B[4 * k + j];
        r.between[(size_t)(4 * i + j)] = acc;
      }
      double acc = 0.0;
      for (int k = 0; k < 3; k++) acc += A[4 * k + i] * (B[4 * k + 3] - A[4 * k + 3]);
      r.between[(size_t)(4 * i + 3)] = acc;
    }
    r.between[12] = r.between[13] = r.between[14] = 0.0;
    r.between[15] = 1.0;
  }
  return out;
}

}  // namespace mi355

// ---- C-ABI (include/mi355_loop_closure.h) ----------------------------------------------------------------------------
struct mi355_loop_closure {
  std::unique_ptr<mi355::LoopClosureCore> core;
  std::string err;
};

template <typename F>
static int lc_guarded(mi355_loop_closure* h, F&& f) {
  try {
    f();
    return MI355_OK;
  } catch (const mi355::Error& e) {
    h->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    h->err = e.what();
    return MI355_EINVAL;
  }
}

static mi355::Pose pose_from7(const double p[7]) {
  mi355::Pose q;
  q.px = p[0]; q.py = p[1]; q.pz = p[2];
  q.qx = p[3]; q.qy = p[4]; q.qz = p[5]; q.qw = p[6];
  return q;
}

extern "C" void mi355_loop_closure_default_params(mi355_loop_closure_params* p) {
  if (!p) return;
  const mi355::LoopClosureParams d;
  p->search_key_frame_num = d.search_key_frame_num;
  p->score_threshold = d.score_threshold;
  p->search_for_candidate_threshold = d.search_for_candidate_threshold;
  p->accumulate_distance_threshold = d.accumulate_distance_threshold;
  p->submap_leaf = d.submap_leaf;
  p->max_iteration = d.max_iteration;
  p->transformation_epsilon = d.transformation_epsilon;
  p->max_correspondence_distance = d.max_correspondence_distance;
  p->correspondence_randomness = d.correspondence_randomness;
  p->registration_method = d.registration_method;
  p->step_size = d.step_size;
  p->ndt_resolution = d.ndt_resolution;
  p->correspondence_distance = d.correspondence_distance;
  p->max_optimizer_iteration = d.max_optimizer_iteration;
  p->ransac_iteration = d.ransac_iteration;
}

extern "C" mi355_loop_closure* mi355_loop_closure_create(int device_id, const mi355_loop_closure_params* p) {
  return mi355_loop_closure_create_multi(&device_id, 1, p);
}

extern "C" mi355_loop_closure* mi355_loop_closure_create_multi(const int* device_ids, int n_devices, const mi355_loop_closure_params* p) {
  if (!p || !device_ids || n_devices <= 0 || p->search_key_frame_num < 0 || !(p->submap_leaf >= 0.f)) return nullptr;
  if (p->ransac_iteration != 0) return nullptr;  // only "no RANSAC rejector" (the yaml's value, graph_based_slam.param.yaml:29) is built
  if (p->registration_method != MI355_METHOD_FAST_GICP && p->registration_method != MI355_METHOD_NDT_OMP && p->registration_method != MI355_METHOD_ICP &&
      p->registration_method != MI355_METHOD_GICP)
    return nullptr;  // unknown strings leave the reference with a null registration too (:153)
  mi355::LoopClosureParams lp;
  lp.search_key_frame_num = p->search_key_frame_num;
  lp.score_threshold = p->score_threshold;
  lp.search_for_candidate_threshold = p->search_for_candidate_threshold;
  lp.accumulate_distance_threshold = p->accumulate_distance_threshold;
  lp.submap_leaf = p->submap_leaf;
  lp.max_iteration = p->max_iteration;
  lp.transformation_epsilon = p->transformation_epsilon;
  lp.max_correspondence_distance = p->max_correspondence_distance;
  lp.correspondence_randomness = p->correspondence_randomness;
  lp.registration_method = p->registration_method;
  lp.step_size = p->step_size;
  lp.ndt_resolution = p->ndt_resolution;
  lp.correspondence_distance = p->correspondence_distance;
  lp.max_optimizer_iteration = p->max_optimizer_iteration;
  lp.ransac_iteration = p->ransac_iteration;
  auto* h = new mi355_loop_closure();
  try {
    h->core.reset(new mi355::LoopClosureCore(std::vector<int>(device_ids, device_ids + n_devices), lp));
  } catch (const std::exception&) {
    delete h;
    return nullptr;
  }
  return h;
}

extern "C" void mi355_loop_closure_destroy(mi355_loop_closure* h) { delete h; }

extern "C" const char* mi355_loop_closure_last_error(const mi355_loop_closure* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int mi355_loop_closure_add_key_frame(mi355_loop_closure* h, const mi355_cloud_view* cloud, const double pose7[7], double accum_distance) {
  if (!h || !cloud || !pose7 || (cloud->n && !cloud->data)) return -1;
  if (cloud->stride < 16 || cloud->stride % 4 || cloud->intensity_offset < 12 || cloud->intensity_offset + 4 > cloud->stride) return -1;
  try {
    mi355::PointCloud in(cloud->n);
    const char* base = (const char*)cloud->data;
    for (size_t i = 0; i < cloud->n; i++) {
      const float* p = (const float*)(base + i * cloud->stride);
      in[i].x = p[0]; in[i].y = p[1]; in[i].z = p[2]; in[i].w = 1.f;
      in[i].intensity = *(const float*)(base + i * cloud->stride + cloud->intensity_offset);
      in[i].pad[0] = in[i].pad[1] = in[i].pad[2] = 0.f;
    }
    return h->core->add_key_frame(std::move(in), pose_from7(pose7), accum_distance);
  } catch (const std::exception& e) {
    h->err = e.what();
    return -1;
  }
}

extern "C" int mi355_loop_closure_set_pose(mi355_loop_closure* h, int key_frame_id, const double pose7[7]) {
  if (!h || !pose7) return MI355_EINVAL;
  return lc_guarded(h, [&] {
    if (key_frame_id < 0 || (size_t)key_frame_id >= h->core->size()) throw mi355::Error(MI355_EINVAL, "key frame id out of range");
    h->core->set_pose(key_frame_id, pose_from7(pose7));
  });
}

extern "C" int mi355_loop_closure_candidate(const mi355_loop_closure* h, int query_id, int* candidate_id) {
  if (!h || !candidate_id) return MI355_EINVAL;
  return lc_guarded(const_cast<mi355_loop_closure*>(h), [&] { *candidate_id = h->core->candidate(query_id); });
}

extern "C" int mi355_loop_closure_close(mi355_loop_closure* h, const int* query_ids, int n_queries, mi355_loop_result* results) {
  if (!h || n_queries < 0 || (n_queries && (!query_ids || !results))) return MI355_EINVAL;
  return lc_guarded(h, [&] {
    const std::vector<mi355::LoopResult> r = h->core->close(std::vector<int>(query_ids, query_ids + n_queries));
    for (int i = 0; i < n_queries; i++) {
      mi355_loop_result& o = results[i];
      o.query_id = r[(size_t)i].query_id;
      o.candidate_id = r[(size_t)i].candidate_id;
      o.converged = r[(size_t)i].converged;
      o.accepted = r[(size_t)i].accepted;
      o.iterations = r[(size_t)i].iterations;
      o.fitness = r[(size_t)i].fitness;
      std::copy(r[(size_t)i].transform.begin(), r[(size_t)i].transform.end(), o.transform);
      std::copy(r[(size_t)i].between.begin(), r[(size_t)i].between.end(), o.between);
      o.target_points = r[(size_t)i].target_points;
    }
  });
}
