// mi355_multi_* (include/mi355_lidar.h): sharded batch registration over several devices from ONE host process.
// Caller: the loop-closure registration of /root/reference/graph_based_slam/src/graph_based_slam.cpp:311-322, generalised
// from one candidate per tick to a candidate list (BASELINE configs[4]).  Host C++ only: every device call goes through
// the C-ABI of the single-device library.
#include <memory>
#include <string>

#include "lanes.hpp"

struct mi355_multi {
  std::vector<std::unique_ptr<mi355::Lane>> lanes;
  std::string err;
};

extern "C" mi355_multi* mi355_multi_create(const int* device_ids, int n_devices) {
  if (!device_ids || n_devices <= 0) return nullptr;
  auto* m = new mi355_multi();
  try {
    for (int i = 0; i < n_devices; i++) {
      m->lanes.emplace_back(new mi355::Lane(device_ids[i]));
      m->lanes.back()->ctx();  // fail here, not at the first registration, when a device does not exist
    }
  } catch (const std::exception&) {
    delete m;
    return nullptr;
  }
  return m;
}

extern "C" void mi355_multi_destroy(mi355_multi* m) { delete m; }
extern "C" const char* mi355_multi_last_error(const mi355_multi* m) { return m ? m->err.c_str() : "null handle (no HIP device?)"; }
extern "C" int mi355_multi_device_count(const mi355_multi* m) { return m ? (int)m->lanes.size() : 0; }
extern "C" mi355_ctx* mi355_multi_context(mi355_multi* m, int lane) {
  if (!m || lane < 0 || (size_t)lane >= m->lanes.size()) return nullptr;
  return m->lanes[(size_t)lane]->ctx().get();
}

extern "C" int mi355_multi_lane_allocations(const mi355_multi* m, int lane) {
  if (!m || lane < 0 || (size_t)lane >= m->lanes.size()) return -1;
  return m->lanes[(size_t)lane]->allocations();
}

extern "C" int mi355_multi_shard_by_target(const int* tgt_of_src, int n_pairs, int n_lanes, int* lane_of_pair) {
  if (n_pairs < 0 || n_lanes <= 0 || (n_pairs && (!tgt_of_src || !lane_of_pair))) return MI355_EINVAL;
  const auto shards = mi355::shard_by_target(tgt_of_src, n_pairs, n_lanes);
  for (int l = 0; l < n_lanes; l++)
    for (int p : shards[(size_t)l]) lane_of_pair[p] = l;
  return MI355_OK;
}

namespace {
// What both forms of mi355_multi_batch_register share: argument checks, the partition, one host thread per busy lane, the gather.
// per_lane(lane, pairs of the lane sorted by (target, pair), local target index of every one of them, distinct targets in that
// order, results, fitness) registers the lane's share.
template <typename PerLane>
int multi_register(mi355_multi* m, int method, const void* params, int n_sources, int n_targets, const int* tgt_of_src, int with_fitness,
                   mi355_result* results, double* fitness, int* lane_of_pair, PerLane&& per_lane) {
  if (!m) return MI355_EINVAL;
  try {
    if (n_sources <= 0 || n_targets <= 0 || !results) throw mi355::Error(MI355_EINVAL, "mi355_multi_batch_register: NULL / empty argument");
    if (with_fitness && !fitness) throw mi355::Error(MI355_EINVAL, "with_fitness needs a fitness array");
    if (method != MI355_METHOD_FAST_GICP && method != MI355_METHOD_NDT_OMP && method != MI355_METHOD_ICP && method != MI355_METHOD_GICP)
      throw mi355::Error(MI355_EINVAL, "batch mode registers with FAST_GICP, NDT_OMP, ICP or GICP");
    if (!tgt_of_src && n_sources != n_targets) throw mi355::Error(MI355_EINVAL, "tgt_of_src required when n_sources != n_targets");
    mi355_gicp_params gp{};
    mi355_gicp_default_params(&gp);
    mi355_ndt_params np{};
    mi355_ndt_default_params(&np);
    if (params && method == MI355_METHOD_FAST_GICP) gp = *(const mi355_gicp_params*)params;
    if (params && method == MI355_METHOD_NDT_OMP) np = *(const mi355_ndt_params*)params;
    mi355_icp_params ip{};
    mi355_icp_default_params(&ip);
    if (params && method == MI355_METHOD_ICP) ip = *(const mi355_icp_params*)params;
    mi355_gicp_bfgs_params bp{};
    mi355_gicp_bfgs_default_params(&bp);
    if (params && method == MI355_METHOD_GICP) bp = *(const mi355_gicp_bfgs_params*)params;
    const void* pp = method == MI355_METHOD_FAST_GICP ? (const void*)&gp
                     : method == MI355_METHOD_NDT_OMP ? (const void*)&np
                     : method == MI355_METHOD_GICP    ? (const void*)&bp
                                                      : (const void*)&ip;
    std::vector<int> map((size_t)n_sources);
    for (int p = 0; p < n_sources; p++) {
      map[(size_t)p] = tgt_of_src ? tgt_of_src[p] : p;
      if (map[(size_t)p] < 0 || map[(size_t)p] >= n_targets) throw mi355::Error(MI355_EINVAL, "tgt_of_src out of range");
    }
    const int L = (int)m->lanes.size();
    const auto shards = mi355::shard_by_target(map.data(), n_sources, L);
    std::vector<int> busy;
    for (int l = 0; l < L; l++) {
      if (!shards[(size_t)l].empty()) busy.push_back(l);
      if (lane_of_pair)
        for (int p : shards[(size_t)l]) lane_of_pair[p] = l;
    }
    mi355::for_each_lane(busy, [&](int l) {
      const std::vector<int>& mine = shards[(size_t)l];  // sorted by (target, pair)
      std::vector<int> local_of((size_t)n_targets, -1), local_map, my_targets;
      for (int p : mine) {
        const int t = map[(size_t)p];
        if (local_of[(size_t)t] < 0) {
          local_of[(size_t)t] = (int)my_targets.size();
          my_targets.push_back(t);
        }
        local_map.push_back(local_of[(size_t)t]);
      }
      std::vector<mi355_result> res(mine.size());
      std::vector<double> fit(mine.size());
      per_lane(*m->lanes[(size_t)l], pp, mine, local_map, my_targets, res.data(), fit.data());
      for (size_t k = 0; k < mine.size(); k++) {  // the gather: fixed-size records back to the caller's pair order
        results[mine[k]] = res[k];
        if (with_fitness) fitness[mine[k]] = fit[k];
      }
    });
    return MI355_OK;
  } catch (const mi355::Error& e) {
    m->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    m->err = e.what();
    return MI355_EINVAL;
  }
}
std::vector<float> lane_guesses(const float* guesses16, const std::vector<int>& mine) {
  std::vector<float> g;
  if (guesses16)
    for (int p : mine) g.insert(g.end(), guesses16 + 16 * (size_t)p, guesses16 + 16 * (size_t)p + 16);
  return g;
}
}  // namespace

// Host views: every lane sends its sub-maps and candidates through its batch object's pinned staging (mi355_batch_set_*_views: one
// 2-D copy per run of slots, packing overlapped with the transfers) - each cloud crosses PCIe once, to the lane that registers it.
extern "C" int mi355_multi_batch_register(mi355_multi* m, int method, const void* params, const mi355_cloud_view* sources, int n_sources,
                                          const mi355_cloud_view* targets, int n_targets, const int* tgt_of_src, const float* guesses16,
                                          const mi355_prefilter_params* source_prefilter, const mi355_prefilter_params* target_prefilter,
                                          int with_fitness, mi355_result* results, double* fitness, int* lane_of_pair) {
  if (m && (!sources || !targets)) {
    m->err = "mi355_multi_batch_register: NULL sources / targets";
    return MI355_EINVAL;
  }
  return multi_register(m, method, params, n_sources, n_targets, tgt_of_src, with_fitness, results, fitness, lane_of_pair,
                        [&](mi355::Lane& lane, const void* pp, const std::vector<int>& mine, const std::vector<int>& local_map,
                            const std::vector<int>& my_targets, mi355_result* res, double* fit) {
                          std::vector<mi355_cloud_view> sv, tv;
                          for (int p : mine) sv.push_back(sources[p]);
                          for (int t : my_targets) tv.push_back(targets[t]);
                          const std::vector<float> g = lane_guesses(guesses16, mine);
                          lane.register_batch_views(method, pp, sv, tv, local_map, guesses16 ? g.data() : nullptr, source_prefilter, target_prefilter,
                                                    with_fitness != 0, res, fit);
                        });
}

// Device clouds: nothing is uploaded; every cloud must live on the lane that registers it (mi355_batch_set_*_clouds refuses a cloud
// of another context).
extern "C" int mi355_multi_batch_register_clouds(mi355_multi* m, int method, const void* params, mi355_cloud* const* sources, int n_sources,
                                                 mi355_cloud* const* targets, int n_targets, const int* tgt_of_src, const float* guesses16,
                                                 const mi355_prefilter_params* source_prefilter, const mi355_prefilter_params* target_prefilter,
                                                 int with_fitness, mi355_result* results, double* fitness, int* lane_of_pair) {
  if (m && (!sources || !targets)) {
    m->err = "mi355_multi_batch_register_clouds: NULL sources / targets";
    return MI355_EINVAL;
  }
  return multi_register(m, method, params, n_sources, n_targets, tgt_of_src, with_fitness, results, fitness, lane_of_pair,
                        [&](mi355::Lane& lane, const void* pp, const std::vector<int>& mine, const std::vector<int>& local_map,
                            const std::vector<int>& my_targets, mi355_result* res, double* fit) {
                          std::vector<mi355_cloud*> sh, th;
                          for (int p : mine) sh.push_back(sources[p]);
                          for (int t : my_targets) th.push_back(targets[t]);
                          const std::vector<float> g = lane_guesses(guesses16, mine);
                          lane.register_batch(method, pp, 0, sh, th, local_map, guesses16 ? g.data() : nullptr, source_prefilter, target_prefilter,
                                              with_fitness != 0, res, fit);
                        });
}
