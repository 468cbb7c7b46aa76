// Multi-device below the C-ABI (SURVEY 8b "mi355_batch_align(ctx_or_multi, ...)", 8e process model): ONE host process,
// one mi355_ctx + stream per device ("lane"), scan pairs dealt to lanes by sub-map, every lane registers its share with
// one batch launch sequence, the fixed-size results come back by D2H — single process, so no collective is needed.
// The caller this serves is the loop-closure back end (graph_based_slam.cpp:311-322), a C++ node that cannot call
// torch.distributed; the torchrun form of the same partition is lidar_graph_slam_amd/sharding.py (one process per GPU).
#pragma once
#include <algorithm>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <thread>
#include <vector>

#include "pcl_like.hpp"

namespace mi355 {

// Deal pairs to `world` lanes so that all pairs of one target (sub-map) land on one lane — each sub-map is uploaded, voxel-
// filtered and indexed once — greedily balancing the pair counts: targets by descending pair count (ties: lower target id),
// each to the least-loaded lane (ties: lower lane).  Per lane the pair indices come sorted by (target, pair index).  The same
// rule as sharding.shard_by_target (Python, one process per GPU), so both forms produce the same partition.
inline std::vector<std::vector<int>> shard_by_target(const int* tgt_of_src, int n_pairs, int world) {
  std::map<int, int> count;
  for (int p = 0; p < n_pairs; p++) count[tgt_of_src[p]]++;
  std::vector<std::pair<int, int>> groups(count.begin(), count.end());  // (target, pairs), ascending target
  std::stable_sort(groups.begin(), groups.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second > b.second; });
  std::vector<long> load((size_t)world, 0);
  std::map<int, int> owner;
  for (const auto& g : groups) {
    const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    owner[g.first] = r;
    load[(size_t)r] += g.second;
  }
  std::vector<std::vector<int>> out((size_t)world);
  for (int p = 0; p < n_pairs; p++) out[(size_t)owner[tgt_of_src[p]]].push_back(p);
  for (auto& v : out) std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return tgt_of_src[a] < tgt_of_src[b]; });
  return out;
}

// Field-wise equality of the parameter structs (they contain padding bytes, and callers fill stack locals field by field: a
// memcmp over the raw struct may differ where no parameter does and would rebuild a lane's batch - tens of GB - at every call).
inline bool same_params(const mi355_gicp_params& a, const mi355_gicp_params& b) {
  return a.k_correspondences == b.k_correspondences && a.max_iterations == b.max_iterations && a.transformation_epsilon == b.transformation_epsilon &&
         a.rotation_epsilon == b.rotation_epsilon && a.max_correspondence_distance == b.max_correspondence_distance &&
         a.regularization_method == b.regularization_method && a.optimizer == b.optimizer && a.lm_max_iterations == b.lm_max_iterations &&
         a.lm_init_lambda_factor == b.lm_init_lambda_factor && a.force_iterations == b.force_iterations && a.num_threads == b.num_threads;
}
inline bool same_params(const mi355_ndt_params& a, const mi355_ndt_params& b) {
  return a.transformation_epsilon == b.transformation_epsilon && a.step_size == b.step_size && a.resolution == b.resolution &&
         a.max_iterations == b.max_iterations && a.outlier_ratio == b.outlier_ratio && a.min_points_per_voxel == b.min_points_per_voxel &&
         a.min_covar_eigvalue_mult == b.min_covar_eigvalue_mult && a.legacy_mt_skip == b.legacy_mt_skip && a.num_threads == b.num_threads;
}

inline bool same_params(const mi355_icp_params& a, const mi355_icp_params& b) {
  return a.max_correspondence_distance == b.max_correspondence_distance && a.max_iterations == b.max_iterations &&
         a.transformation_epsilon == b.transformation_epsilon && a.euclidean_fitness_epsilon == b.euclidean_fitness_epsilon &&
         a.rotation_epsilon == b.rotation_epsilon && a.ransac_iterations == b.ransac_iterations;
}

inline bool same_params(const mi355_gicp_bfgs_params& a, const mi355_gicp_bfgs_params& b) {
  return a.max_correspondence_distance == b.max_correspondence_distance && a.max_iterations == b.max_iterations &&
         a.max_optimizer_iterations == b.max_optimizer_iterations && a.transformation_epsilon == b.transformation_epsilon &&
         a.rotation_epsilon == b.rotation_epsilon && a.euclidean_fitness_epsilon == b.euclidean_fitness_epsilon &&
         a.k_correspondences == b.k_correspondences && a.ransac_iterations == b.ransac_iterations;
}

// One device's share: its context and a batch object that outlives the calls (tens of GB at 512 candidates: allocated once,
// grown only when a call needs more sources, sub-maps or capacity).
class Lane {
 public:
  explicit Lane(int device) : device_(device) {}
  ~Lane() {
    if (batch_) mi355_batch_destroy(batch_);
  }
  Lane(const Lane&) = delete;
  Lane& operator=(const Lane&) = delete;
  int device() const { return device_; }
  Context& ctx() {  // created by the first use: bookkeeping needs no GPU
    if (!ctx_) ctx_.reset(new Context(device_));
    return *ctx_;
  }
  // Register source s against target tgt_of_src[s], all in one batch launch sequence.  The clouds are either device clouds of THIS
  // lane's context (sources / targets) or host views (src_views / tgt_views: uploaded through the batch's pinned staging).
  // method: MI355_METHOD_FAST_GICP (params = mi355_gicp_params), MI355_METHOD_NDT_OMP (mi355_ndt_params), MI355_METHOD_ICP (mi355_icp_params)
  // or MI355_METHOD_GICP (mi355_gicp_bfgs_params).
  void register_batch(int method, const void* params, size_t params_size, const std::vector<mi355_cloud*>& sources, const std::vector<mi355_cloud*>& targets,
                      const std::vector<int>& tgt_of_src, const float* guesses16, const mi355_prefilter_params* source_prefilter,
                      const mi355_prefilter_params* target_prefilter, bool with_fitness, mi355_result* results, double* fitness) {
    (void)params_size;
    size_t cap = 1;
    for (mi355_cloud* c : sources) cap = std::max(cap, mi355_cloud_size(c));
    for (mi355_cloud* c : targets) cap = std::max(cap, mi355_cloud_size(c));
    if (!prepare(method, params, (int)sources.size(), (int)targets.size(), cap)) return;
    ctx().check(mi355_batch_set_targets_clouds(batch_, targets.data(), target_prefilter));
    ctx().check(mi355_batch_set_sources_clouds(batch_, sources.data(), source_prefilter));
    ctx().check(mi355_batch_align(batch_, tgt_of_src.data(), guesses16, with_fitness ? 1 : 0));
    ctx().check(mi355_batch_results(batch_, results, with_fitness ? fitness : nullptr));
  }
  void register_batch_views(int method, const void* params, const std::vector<mi355_cloud_view>& src_views, const std::vector<mi355_cloud_view>& tgt_views,
                            const std::vector<int>& tgt_of_src, const float* guesses16, const mi355_prefilter_params* source_prefilter,
                            const mi355_prefilter_params* target_prefilter, bool with_fitness, mi355_result* results, double* fitness) {
    size_t cap = 1;
    for (const auto& v : src_views) cap = std::max(cap, v.n);
    for (const auto& v : tgt_views) cap = std::max(cap, v.n);
    if (!prepare(method, params, (int)src_views.size(), (int)tgt_views.size(), cap)) return;
    // the sub-maps first: their chains (voxel filter, index, covariances) run while the candidates cross PCIe
    ctx().check(mi355_batch_set_targets_views(batch_, tgt_views.data(), target_prefilter));
    ctx().check(mi355_batch_set_sources_views(batch_, src_views.data(), source_prefilter));
    ctx().check(mi355_batch_align(batch_, tgt_of_src.data(), guesses16, with_fitness ? 1 : 0));
    ctx().check(mi355_batch_results(batch_, results, with_fitness ? fitness : nullptr));
  }

 private:
  // The lane's batch object, sized for (S, U, cap): created by the first call, re-created only when a call needs more sources,
  // sub-maps or capacity, or another method / other parameters (grow-only); then narrowed to the live slots.  false: nothing to do.
  bool prepare(int method, const void* params, int S, int U, size_t cap) {
    if (S == 0) return false;
    const bool same = batch_ && method == method_ &&
                      (method == MI355_METHOD_FAST_GICP ? same_params(gp_, *(const mi355_gicp_params*)params)
                       : method == MI355_METHOD_NDT_OMP ? same_params(np_, *(const mi355_ndt_params*)params)
                       : method == MI355_METHOD_GICP    ? same_params(bp_, *(const mi355_gicp_bfgs_params*)params)
                                                        : same_params(ip_, *(const mi355_icp_params*)params));
    if (!same || S > S_ || U > U_ || cap > cap_) {  // grow-only
      if (batch_) mi355_batch_destroy(batch_);
      batch_ = nullptr;
      if (!same) S_ = U_ = 0, cap_ = 0;
      S_ = std::max(S, S_); U_ = std::max(U, U_); cap_ = std::max(cap, cap_);
      method_ = method;
      if (method == MI355_METHOD_FAST_GICP) gp_ = *(const mi355_gicp_params*)params;
      else if (method == MI355_METHOD_NDT_OMP) np_ = *(const mi355_ndt_params*)params;
      else if (method == MI355_METHOD_GICP) bp_ = *(const mi355_gicp_bfgs_params*)params;
      else ip_ = *(const mi355_icp_params*)params;
      batch_ = mi355_batch_create(ctx().get(), method, params, S_, U_, cap_);
      allocations_++;
      if (!batch_) {
        S_ = U_ = 0;
        cap_ = 0;
        throw Error(MI355_EHIP, mi355_last_error(ctx().get()));
      }
    }
    // the batch runs its first `live` slots only: a slot beyond them costs nothing (no empty cloud, no NDT leaf build)
    ctx().check(mi355_batch_set_live(batch_, S, U));
    return true;
  }

  int device_;
  std::unique_ptr<Context> ctx_;
  mi355_batch* batch_ = nullptr;
  int S_ = 0, U_ = 0, method_ = -1;
  size_t cap_ = 0;
  mi355_gicp_params gp_{};
  mi355_ndt_params np_{};
  mi355_icp_params ip_{};
  mi355_gicp_bfgs_params bp_{};
  int allocations_ = 0;

 public:
  int allocations() const { return allocations_; }  // times the batch object has been (re)created: stays 1 while requests fit
};

// Run f(lane index) for every lane that has work: inline for one lane, one host thread per lane otherwise (a handle is used
// by one thread at a time; HIP's current device is per thread and every C-ABI call selects its context's device).
inline void for_each_lane(const std::vector<int>& busy, const std::function<void(int)>& f) {
  if (busy.size() <= 1) {
    for (int l : busy) f(l);
    return;
  }
  std::vector<std::exception_ptr> err(busy.size());
  std::vector<std::thread> th;
  for (size_t k = 0; k < busy.size(); k++)
    th.emplace_back([&, k] {
      try {
        f(busy[k]);
      } catch (...) {
        err[k] = std::current_exception();
      }
    });
  for (auto& t : th) t.join();
  for (auto& e : err)
    if (e) std::rethrow_exception(e);
}

}  // namespace mi355
