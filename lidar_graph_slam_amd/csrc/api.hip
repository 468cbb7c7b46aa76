// The C-ABI of libmi355_lidar.so (include/mi355_lidar.h).  Thin: argument checking, host<->device staging and
// sequencing of the engine calls.  No CPU fallback: without a HIP device mi355_create() fails.
#include <algorithm>
#include <cmath>
#include <memory>

#include "engine.hpp"

using namespace mi355;

namespace mi355 {
thread_local KernelTimers* g_timers = nullptr;
thread_local int g_family_override = -1;
const char* const kKernelFamilyName[KF_COUNT] = {"k_knn_cov_tile", "k_knn_hard", "k_voxel_centroids", "sort_pairs(k_sort_*)",
                                                 "hash_build(k_hash_build|k_gather_insert)", "k_gicp_correspond", "k_gicp_correspond_tile", "k_gicp_linearize",
                                                 "k_gicp_step", "k_fitness",
                                                 "sor(sort,hash,k_knn_cov_tile<SOR>,k_knn_hard<SOR>,threshold,compact)",
                                                 "ndt_target_prep(k_ndt_bounds,keys,heads,scan,size,clear)", "ndt_target_sort(k_sort_tile,k_sort_merge)", "k_ndt_leaves(k_ndt_leaf_sums,k_ndt_leaf_finalize)", "k_ndt_derivatives", "k_ndt_advance"};
const Tuning& tuning() {
  static const Tuning t = [] {
    Tuning v;
    auto geti = [](const char* name, int& dst) { if (const char* e = getenv(name)) dst = atoi(e); };
    auto getf = [](const char* name, float& dst) { if (const char* e = getenv(name)) { const float f = (float)atof(e); if (f > 0.f) dst = f; } };
    geti("MI355_BLOCKS_PER_PAIR", v.blocks_per_pair);
    geti("MI355_PPT", v.corr_ppt);
    geti("MI355_CORR_TILES", v.corr_tiles);
    geti("MI355_HASH_SLOTS", v.hash_slots_per_cell);
    geti("MI355_CORR_BLOCKS", v.corr_blocks);
    geti("MI355_CORR_STATS", v.corr_stats);
    geti("MI355_CORR_MASK", v.corr_mask);
    geti("MI355_CORR_GROUPS", v.corr_groups);
    getf("MI355_MASK_SLACK", v.mask_slack);
    getf("MI355_CORR_R1", v.corr_r1);
    getf("MI355_ICP_R1_CELLS", v.icp_r1_cells);
    getf("MI355_ICP_MIN_MOVE", v.icp_min_move);
    getf("MI355_ICP_RMID_CELLS", v.icp_rmid_cells);
    getf("MI355_ICP_R1_SIGMA", v.icp_r1_sigma);
    getf("MI355_ICP_BALL_SCALE", v.icp_ball_scale);
    getf("MI355_ICP_BALL_PAD", v.icp_ball_pad);
    getf("MI355_MARGIN_STEP", v.margin_step);
    geti("MI355_BLOCKS_LIN", v.blocks_lin);
    geti("MI355_LIN_PPT", v.lin_ppt);
    geti("MI355_KNN_TILES", v.knn_tiles);
    geti("MI355_KNN_NEED_DIV", v.knn_need_div);
    geti("MI355_TILE_WAVES", v.tile_waves);
    geti("MI355_HARD_WAVES", v.hard_waves);
    getf("MI355_TILE_RSCALE", v.tile_rscale);
    geti("MI355_TILE_ORDER", v.tile_order);
    getf("MI355_TILE_GATHER", v.tile_gather);
    geti("MI355_BATCH_SIDE_STREAMS", v.side_streams);
    geti("MI355_BATCH_PARTS", v.batch_parts);
    getf("MI355_PREP_CUS", v.prep_cus);
    getf("MI355_LM_CUS", v.lm_cus);
    getf("MI355_GRID_CELL", v.grid_cell);
    geti("MI355_KNN_STATS", v.knn_stats);
    geti("MI355_NDT_WAVES", v.ndt_waves);
    return v;
  }();
  return t;
}
}  // namespace mi355

namespace {

template <typename F>
int guarded(mi355_ctx* ctx, F&& f) {
  try {
    f();
    return MI355_OK;
  } catch (const ArgError& e) {
    if (ctx) ctx->err = e.what();
    return e.code;
  } catch (const HipError& e) {
    if (ctx) ctx->err = e.what();
    return MI355_EHIP;
  } catch (const std::exception& e) {
    if (ctx) ctx->err = e.what();
    return MI355_EINVAL;
  }
}

void check_view(const mi355_cloud_view* v) {
  if (!v) throw ArgError(MI355_EINVAL, "cloud view is NULL");
  if (v->n && !v->data) throw ArgError(MI355_EINVAL, "cloud data is NULL");
  if (v->stride < 16 || v->stride % 4) throw ArgError(MI355_EINVAL, "stride must be >= 16 and a multiple of 4");
  if (v->intensity_offset < 12 || v->intensity_offset + 4 > v->stride) throw ArgError(MI355_EINVAL, "bad intensity_offset");
}

// host (strided) -> packed float4 pinned staging -> device
struct Uploader {
  PinnedBuf<float4> h;
  void upload(hipStream_t s, const mi355_cloud_view* v, float4* d_dst) {
    check_view(v);
    if (v->n == 0) return;
    h.ensure(v->n);
    const char* base = (const char*)v->data;
    for (size_t i = 0; i < v->n; i++) {
      const float* p = (const float*)(base + i * v->stride);
      h.p[i] = make_float4(p[0], p[1], p[2], *(const float*)(base + i * v->stride + v->intensity_offset));
    }
    MI_HIP(hipMemcpyAsync(d_dst, h.p, v->n * sizeof(float4), hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));  // staging buffer is reused by the next upload
  }
};

// scratch for the single-cloud filter entry points
struct FilterScratch {
  Workspace ws;
  Uploader up;
  DevBuf<float4> a, b;
  DevBuf<int> na, nb;
  PinnedBuf<int> h_n;
  CloudSet set;
  DevBuf<int> idx;
  DevBuf<float> f;
  DevBuf<uint8_t> mask;
};

}  // namespace

struct mi355_reg {
  mi355_ctx* ctx = nullptr;
  int method = MI355_METHOD_FAST_GICP;
  mi355_gicp_params gp;
  mi355_ndt_params np;
  mi355_icp_params ip;
  IcpScratch icp;
  mi355_gicp_bfgs_params bp;  // MI355_METHOD_GICP
  BfgsScratch bfgs;
  // NDT (and, for the source, ICP) register without the voxel hash of the cloud; only getFitnessScore needs it: built on demand
  bool src_grid_stale = false, tgt_grid_stale = false;
  Workspace ws;
  Uploader up;
  DevBuf<float4> stage_src, stage_tgt;
  DevBuf<int> cnt_src, cnt_tgt;
  CloudSet src, tgt;
  GicpBatch gb;
  NdtTargets* ndt = nullptr;
  NdtSolver* ndt_solver = nullptr;
  size_t n_src = 0, n_tgt = 0;
  bool has_src = false, has_tgt = false, aligned = false;
  float last_T[16];
  DevBuf<float4> aligned_dev;
  ~mi355_reg() {
    if (ndt) ndt_targets_destroy(ndt);
    if (ndt_solver) ndt_solver_destroy(ndt_solver);
  }
};

struct mi355_cloud {
  mi355_ctx* ctx = nullptr;
  size_t n = 0;
  DevBuf<float4> pts;  // caller's point order
  DevBuf<Cov6> cov;    // caller's point order; valid when has_cov
  bool has_cov = false;
  int cov_k = 0, cov_method = 0;
};

struct mi355_batch {
  mi355_ctx* ctx = nullptr;
  int method = MI355_METHOD_FAST_GICP;
  mi355_gicp_params gp;
  mi355_ndt_params np;       // method == MI355_METHOD_NDT_OMP
  mi355_icp_params ip;       // method == MI355_METHOD_ICP
  IcpBatch* icp = nullptr;
  int icp_max_src = 0;       // no source holds more points than this (the counts before the prefilter)
  mi355_gicp_bfgs_params bp;  // method == MI355_METHOD_GICP
  BfgsBatch* bfgs = nullptr;
  int bfgs_max_src = 0;
  NdtTargets* ndt = nullptr;  // every target's voxel Gaussians, built by one launch sequence per mi355_batch_set_targets*
  NdtSolver* ndt_solver = nullptr;
  int ndt_max_src = 0;        // no source holds more points than this (the counts before the prefilter: nothing is read back)
  const float4* ndt_tgt_pts = nullptr;  // the targets as the leaf builder took them (after the prefilter): their NN voxel hash -
  const int* ndt_tgt_cnt = nullptr;     // what getFitnessScore searches - is built only when a fitness is asked for
  int n_sources = 0, n_targets = 0;          // live slots (mi355_batch_set_live; = the allocation after mi355_batch_create)
  int alloc_sources = 0, alloc_targets = 0;  // slots the buffers were sized for
  size_t cap = 0;
  // THE PIPELINE (round 4).  A batch of >= 512 FAST_GICP pairs is cut into 2 - 4 PARTS: contiguous shares of the sources and of the
  // targets.  The preparation chains (prefilter -> sort -> voxel hash -> k-NN covariances) of the targets run part after part on one
  // side stream, those of the sources on another, and every part's LM loop on a stream of its own, released by the events of the
  // chains it needs: while part i is in its LM iterations (streaming kernels, bound by HBM) the chains of part i + 1 index the next
  // clouds (latency-bound searches) - what a caller used to get only from several batch objects in flight.  A part is the unit of
  // every launch; a pair's launch geometry is that of the whole batch (GicpBatch::S_geometry), so results are bit-equal whatever the
  // number of parts.  Small batches and NDT batches have one part.
  struct Part {
    int s0 = 0, ns = 0, t0 = 0, nt = 0;  // live ranges of sources / targets
    int cap_s = 0, cap_t = 0;            // slots this part was sized for
    CloudSet src;   // this part's sources (owned; pair indices are part-local)
    CloudSet tgt;   // this part's targets: a WINDOW into mi355_batch::tgt (one cloud-major array: a pair may register against any target)
    GicpBatch gb;
    Workspace ws[2];        // 0 = the targets' chain, 1 = the sources' chain
    CloudSet sor_set[2];
    DevBuf<int> cnt_in[2];
    std::vector<int> h_counts[2];
    hipStream_t lm_stream = nullptr;
    hipEvent_t prep_done[2] = {nullptr, nullptr}, knn_done[2] = {nullptr, nullptr}, lm_done = nullptr;
    bool prep_pending[2] = {false, false}, knn_pending[2] = {false, false};
    int it_enqueued = 0;
    bool lm_finished = false;
    ~Part() {
      if (lm_stream) { (void)hipStreamSynchronize(lm_stream); (void)hipStreamDestroy(lm_stream); }
      for (auto& e : prep_done) if (e) (void)hipEventDestroy(e);
      for (auto& e : knn_done) if (e) (void)hipEventDestroy(e);
      if (lm_done) (void)hipEventDestroy(lm_done);
    }
  };
  std::vector<std::unique_ptr<Part>> parts;
  struct Side {  // the stream the chains of one side (0 targets, 1 sources) run on, forked from the batch's stream at call time
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr;
    bool pending = false;  // work enqueued since the last join
  };
  Side side[2];
  CloudSet tgt;   // ALL targets
  KernelTimers timers;  // mi355_batch_set_timing / mi355_batch_kernel_times
  int side_streams = -1;  // mi355_batch_set_side_streams: -1 = the library default (Tuning::side_streams), 0 / 1 = off / on
  DevBuf<float4> host_stage_tgt, host_stage_src;  // device copies of host inputs, one per side: a side's next upload is ordered behind the align that joined its last chain
  DevBuf<float4> cloud_stage_src, cloud_stage_tgt;  // mi355_batch_set_*_clouds: the slots assembled on the device
  // mi355_batch_set_*_views: host clouds of any stride are packed into two pinned buffers that take turns - while one crosses PCIe
  // (one 2-D copy for a run of slots) the host packs the next run into the other
  PinnedBuf<float4> pin[2];
  hipEvent_t pin_free[2] = {nullptr, nullptr};
  bool pin_used[2] = {false, false};
  hipStream_t own_stream = nullptr;  // mi355_batch_set_stream: this batch's stream instead of the context's
  bool has_own_stream = false;
  hipStream_t stream() const { return has_own_stream ? own_stream : ctx->stream; }
  bool use_side_streams() const { return (side_streams >= 0 ? side_streams : tuning().side_streams) != 0; }
  // where part i's LM loop runs: its own stream when the chains run beside it, else the batch's
  hipStream_t lm_stream(size_t i) const { return (use_side_streams() && parts.size() > 1) ? parts[i]->lm_stream : stream(); }
  bool has_src = false, has_tgt = false, aligned = false, with_fitness = false;
  // every internal stream has finished what was enqueued on it
  void quiesce() {
    for (auto& sd : side)
      if (sd.stream) MI_HIP(hipStreamSynchronize(sd.stream));
    for (auto& pt : parts)
      if (pt->lm_stream) MI_HIP(hipStreamSynchronize(pt->lm_stream));
    MI_HIP(hipStreamSynchronize(stream()));
    for (auto& sd : side) sd.pending = false;
    for (auto& pt : parts) pt->prep_pending[0] = pt->prep_pending[1] = pt->knn_pending[0] = pt->knn_pending[1] = false;
  }
  ~mi355_batch() {
    if (ndt) ndt_targets_destroy(ndt);
    if (ndt_solver) ndt_solver_destroy(ndt_solver);
    if (icp) icp_batch_destroy(icp);
    if (bfgs) bfgs_batch_destroy(bfgs);
    parts.clear();
    for (auto& e : pin_free)
      if (e) (void)hipEventDestroy(e);
    for (auto& sd : side) {
      if (sd.stream) { (void)hipStreamSynchronize(sd.stream); (void)hipStreamDestroy(sd.stream); }
      if (sd.fork) (void)hipEventDestroy(sd.fork);
    }
  }
};

struct CtxExtras {
  FilterScratch fs;
};
static CtxExtras* extras(mi355_ctx* ctx);

extern "C" int mi355_abi_version(void) { return MI355_ABI_VERSION; }

// ---- defaults ------------------------------------------------------------------------------------------------------
extern "C" void mi355_gicp_default_params(mi355_gicp_params* p) {
  if (!p) return;
  p->k_correspondences = 20;
  p->max_iterations = 64;
  p->transformation_epsilon = 0.01;
  p->rotation_epsilon = 2e-3;
  p->max_correspondence_distance = 2.0;
  p->regularization_method = MI355_REG_PLANE;
  p->optimizer = MI355_OPT_LM;
  p->lm_max_iterations = 10;
  p->lm_init_lambda_factor = 1e-9;
  p->force_iterations = 0;
  p->num_threads = 0;
}
extern "C" void mi355_ndt_default_params(mi355_ndt_params* p) {
  if (!p) return;
  p->transformation_epsilon = 0.01;
  p->step_size = 0.1;
  p->resolution = 2.0;
  p->max_iterations = 64;
  p->outlier_ratio = 0.55;
  p->min_points_per_voxel = 6;
  p->min_covar_eigvalue_mult = 0.01;
  // ndt_omp (the engine the reference links: thirdparty/ndt_omp, koide3 lineage = PCL 1.8's NDT) evaluates
  // `interval_converged = (step_max - step_min) > 0`: the More-Thuente loop never iterates and the step length is the Newton
  // step clamped to [transformation_epsilon / 2, step_size].  That engine is what MI355_METHOD_NDT_OMP stands for, so it is the
  // default; 0 selects the `< 0` test of PCL >= 1.10 (the line search iterates).
  p->legacy_mt_skip = 1;
  p->num_threads = 0;
}
extern "C" void mi355_gicp_bfgs_default_params(mi355_gicp_bfgs_params* p) {  // lidar_scan_matcher.param.yaml / graph_based_slam.param.yaml "GICP" block
  if (!p) return;
  p->max_correspondence_distance = 2.0;
  p->max_iterations = 64;
  p->max_optimizer_iterations = 20;
  p->transformation_epsilon = 0.01;
  p->rotation_epsilon = 2e-3;
  p->euclidean_fitness_epsilon = 1e-8;
  p->k_correspondences = 20;
  p->ransac_iterations = 0;
}

extern "C" void mi355_icp_default_params(mi355_icp_params* p) {  // graph_based_slam.cpp:146-150
  if (!p) return;
  p->max_correspondence_distance = 30.0;
  p->max_iterations = 100;
  p->transformation_epsilon = 1e-8;
  p->euclidean_fitness_epsilon = 1e-6;
  p->rotation_epsilon = 0.0;
  p->ransac_iterations = 0;
}
extern "C" void mi355_prefilter_default_params(mi355_prefilter_params* p) {
  if (!p) return;
  p->min_distance = 1.0;
  p->leaf_size = 0.1;
  p->mean_k = 30;
  p->stddev_mul = 1.2;
}

// ---- context -------------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
static std::mutex g_mu;
static std::map<mi355_ctx*, CtxExtras*> g_extras;
static CtxExtras* extras(mi355_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_extras.find(ctx);
  if (it != g_extras.end()) return it->second;
  auto* e = new CtxExtras();
  g_extras[ctx] = e;
  return e;
}

extern "C" mi355_ctx* mi355_create(int device_id) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return nullptr;
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  auto* ctx = new mi355_ctx();
  ctx->device = device_id;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return nullptr;
  }
  ctx->own_stream = true;
  ctx->grid_cell = tuning().grid_cell;  // base cell of the voxel hash (m); results do not depend on it
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
    ctx->compute_units = prop.multiProcessorCount;
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
  }
  return ctx;
}

extern "C" void mi355_destroy(mi355_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_extras.find(ctx);
    if (it != g_extras.end()) {
      delete it->second;
      g_extras.erase(it);
    }
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* mi355_last_error(const mi355_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context (no HIP device?)"; }

extern "C" int mi355_set_stream(mi355_ctx* ctx, void* hip_stream, int own_stream) {
  if (!ctx) return MI355_EINVAL;
  return guarded(ctx, [&] {
    MI_HIP(hipSetDevice(ctx->device));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) MI_HIP(hipStreamDestroy(ctx->stream));
    if (own_stream) {
      MI_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      ctx->own_stream = true;
    } else {
      // the caller's stream exactly as given: a NULL handle IS the legacy default stream (what torch's default stream is),
      // so work enqueued there by the caller is ordered with ours
      ctx->stream = (hipStream_t)hip_stream;
      ctx->own_stream = false;
    }
  });
}

extern "C" int mi355_synchronize(mi355_ctx* ctx) {
  if (!ctx) return MI355_EINVAL;
  return guarded(ctx, [&] { MI_HIP(hipStreamSynchronize(ctx->stream)); });
}

extern "C" int mi355_set_grid_cell(mi355_ctx* ctx, float cell) {
  if (!ctx || !(cell > 0.f)) return MI355_EINVAL;
  ctx->grid_cell = cell;
  return MI355_OK;
}

extern "C" int mi355_device_info(mi355_ctx* ctx, char* name, size_t name_cap, int* compute_units) {
  if (!ctx) return MI355_EINVAL;
  if (name && name_cap) snprintf(name, name_cap, "%s", ctx->name);
  if (compute_units) *compute_units = ctx->compute_units;
  return MI355_OK;
}

// ---- filters -------------------------------------------------------------------------------------------------------
namespace {
struct OneCloud {
  FilterScratch& fs;
  size_t n;
  OneCloud(mi355_ctx* ctx, const mi355_cloud_view* in) : fs(extras(ctx)->fs), n(in->n) {
    MI_HIP(hipSetDevice(ctx->device));
    check_view(in);
    const size_t cap = std::max<size_t>(n, 1);
    fs.a.ensure(cap); fs.b.ensure(cap);
    fs.na.ensure(1); fs.nb.ensure(1);
    ws_reserve(fs.ws, 1, cap);
    fs.up.upload(ctx->stream, in, fs.a.p);
    const int ni = (int)n;
    MI_HIP(hipMemcpyAsync(fs.na.p, &ni, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
  }
  size_t cap() const { return std::max<size_t>(n, 1); }
};

size_t download(mi355_ctx* ctx, const float4* d_pts, const int* d_cnt, float* out, size_t out_cap) {
  int cnt = 0;
  MI_HIP(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MI_HIP(hipStreamSynchronize(ctx->stream));
  if ((size_t)cnt > out_cap) throw ArgError(MI355_ECAPACITY, "output capacity too small");
  if (cnt > 0) {
    MI_HIP(hipMemcpyAsync(out, d_pts, (size_t)cnt * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
  }
  return (size_t)cnt;
}
}  // namespace

extern "C" int mi355_distance_filter(mi355_ctx* ctx, const mi355_cloud_view* in, double min_distance, float* out, size_t out_cap, size_t* out_n) {
  if (!ctx || !out_n) return MI355_EINVAL;
  return guarded(ctx, [&] {
    OneCloud oc(ctx, in);
    distance_filter(ctx->stream, oc.fs.ws, oc.fs.a.p, oc.fs.na.p, oc.fs.b.p, oc.fs.nb.p, 1, oc.cap(), min_distance);
    *out_n = download(ctx, oc.fs.b.p, oc.fs.nb.p, out, out_cap);
  });
}

extern "C" int mi355_voxel_grid(mi355_ctx* ctx, const mi355_cloud_view* in, float leaf, float* out, size_t out_cap, size_t* out_n, int* overflow) {
  if (!ctx || !out_n || !(leaf > 0.f)) return MI355_EINVAL;
  return guarded(ctx, [&] {
    OneCloud oc(ctx, in);
    voxel_grid(ctx->stream, oc.fs.ws, oc.fs.a.p, oc.fs.na.p, oc.fs.b.p, oc.fs.nb.p, 1, oc.cap(), leaf);
    *out_n = download(ctx, oc.fs.b.p, oc.fs.nb.p, out, out_cap);
    if (overflow) {
      GridMeta m;
      MI_HIP(hipMemcpy(&m, oc.fs.ws.vmeta.p, sizeof(m), hipMemcpyDeviceToHost));
      *overflow = (oc.n > 0) ? m.overflow : 0;
    }
  });
}

extern "C" int mi355_sor(mi355_ctx* ctx, const mi355_cloud_view* in, int mean_k, double stddev_mul, float* out, size_t out_cap, size_t* out_n,
                         uint8_t* keep_mask, float* mean_dist) {
  if (!ctx || !out_n) return MI355_EINVAL;
  return guarded(ctx, [&] {
    OneCloud oc(ctx, in);
    if (oc.n == 0) { *out_n = 0; return; }
    if ((size_t)mean_k + 1 > oc.n) throw ArgError(MI355_EINVAL, "mean_k + 1 exceeds the number of points");
    oc.fs.mask.ensure(oc.cap());
    oc.fs.f.ensure(oc.cap());
    sor_filter(ctx->stream, oc.fs.ws, oc.fs.set, oc.fs.a.p, oc.fs.na.p, oc.fs.b.p, oc.fs.nb.p, 1, oc.cap(), mean_k, stddev_mul, ctx->grid_cell,
               oc.fs.mask.p, oc.fs.f.p);
    *out_n = download(ctx, oc.fs.b.p, oc.fs.nb.p, out, out_cap);
    if (keep_mask) MI_HIP(hipMemcpy(keep_mask, oc.fs.mask.p, oc.n, hipMemcpyDeviceToHost));
    if (mean_dist) MI_HIP(hipMemcpy(mean_dist, oc.fs.f.p, oc.n * sizeof(float), hipMemcpyDeviceToHost));
  });
}

// distance -> voxel -> SOR on device; returns the buffer holding the result (a or b of `fs`-like scratch)
static void run_prefilter(hipStream_t s, Workspace& ws, CloudSet& sor_set, float grid_cell, const mi355_prefilter_params& p, int B, size_t cap,
                          const float4* in, const int* cnt_in, DevBuf<float4>& a, DevBuf<float4>& b, DevBuf<int>& na, DevBuf<int>& nb,
                          const float4** out, const int** cnt_out, CloudSet* index_next = nullptr) {
  // index_next: the set the caller indexes the result into next, on the same workspace (voxel_grid prepares that index's keys when
  // it is the last stage)
  a.ensure((size_t)B * cap); b.ensure((size_t)B * cap);
  na.ensure(B); nb.ensure(B);
  const float4* cur = in;
  const int* ncur = cnt_in;
  auto next = [&](float4** dst, int** ndst) {
    if (cur == a.p) { *dst = b.p; *ndst = nb.p; } else { *dst = a.p; *ndst = na.p; }
  };
  float4* dst;
  int* ndst;
  const bool fused = p.min_distance > 0 && p.leaf_size > 0;  // distance filter inside the voxel grid's key pass
  if (p.min_distance > 0 && !fused) {
    next(&dst, &ndst);
    distance_filter(s, ws, cur, ncur, dst, ndst, B, cap, p.min_distance);
    cur = dst; ncur = ndst;
  }
  if (p.leaf_size > 0) {
    next(&dst, &ndst);
    voxel_grid(s, ws, cur, ncur, dst, ndst, B, cap, (float)p.leaf_size, fused ? p.min_distance : -1.0, p.mean_k > 0 ? nullptr : index_next, grid_cell);
    cur = dst; ncur = ndst;
  }
  if (p.mean_k > 0) {
    next(&dst, &ndst);
    sor_filter(s, ws, sor_set, cur, ncur, dst, ndst, B, cap, p.mean_k, p.stddev_mul, grid_cell, nullptr, nullptr);
    cur = dst; ncur = ndst;
  }
  *out = cur;
  *cnt_out = ncur;
}

extern "C" int mi355_prefilter(mi355_ctx* ctx, const mi355_cloud_view* in, const mi355_prefilter_params* p, float* out, size_t out_cap, size_t* out_n) {
  if (!ctx || !out_n || !p) return MI355_EINVAL;
  return guarded(ctx, [&] {
    OneCloud oc(ctx, in);
    if (oc.n == 0) { *out_n = 0; return; }
    // ping-pong between two extra stages so the input in fs.a stays intact
    static thread_local int dummy = 0;
    (void)dummy;
    FilterScratch& fs = oc.fs;
    const float4* res;
    const int* nres;
    run_prefilter(ctx->stream, fs.ws, fs.set, ctx->grid_cell, *p, 1, oc.cap(), fs.a.p, fs.na.p, fs.ws.stage_a, fs.ws.stage_b, fs.ws.count_a,
                  fs.ws.count_b, &res, &nres);
    *out_n = download(ctx, res, nres, out, out_cap);
  });
}

extern "C" int mi355_transform(mi355_ctx* ctx, const mi355_cloud_view* in, const float T16[16], float* out) {
  if (!ctx || !T16 || !out) return MI355_EINVAL;
  return guarded(ctx, [&] {
    OneCloud oc(ctx, in);
    transform_points(ctx->stream, oc.fs.a.p, oc.fs.b.p, oc.n, T16);
    if (oc.n) MI_HIP(hipMemcpyAsync(out, oc.fs.b.p, oc.n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
  });
}

// ---- stage-level ---------------------------------------------------------------------------------------------------
extern "C" int mi355_knn(mi355_ctx* ctx, const mi355_cloud_view* pts, const mi355_cloud_view* queries, int k, int* idx, float* d2) {
  if (!ctx || !idx || !d2) return MI355_EINVAL;
  return guarded(ctx, [&] {
    OneCloud oc(ctx, pts);
    check_view(queries);
    if (k < 1 || (size_t)k > oc.n) throw ArgError(MI355_EINVAL, "k must be in [1, n]");
    FilterScratch& fs = oc.fs;
    if (fs.set.B != 1 || fs.set.cap != oc.cap()) fs.set.alloc(1, oc.cap());
    build_grid(ctx->stream, fs.ws, fs.a.p, fs.na.p, fs.set, ctx->grid_cell);
    const size_t m = queries->n;
    if (m == 0) return;
    fs.b.ensure(m);
    fs.up.upload(ctx->stream, queries, fs.b.p);
    fs.idx.ensure(m * k);
    fs.f.ensure(m * k);
    knn_query(ctx->stream, fs.set, fs.b.p, (int)m, k, fs.idx.p, fs.f.p);
    MI_HIP(hipMemcpyAsync(idx, fs.idx.p, m * k * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(d2, fs.f.p, m * k * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
  });
}

extern "C" int mi355_covariances(mi355_ctx* ctx, const mi355_cloud_view* pts, int k, int method, float* cov6) {
  if (!ctx || !cov6) return MI355_EINVAL;
  return guarded(ctx, [&] {
    OneCloud oc(ctx, pts);
    if (k < 1 || (size_t)k > oc.n) throw ArgError(MI355_EINVAL, "k must be in [1, n]");
    FilterScratch& fs = oc.fs;
    if (fs.set.B != 1 || fs.set.cap != oc.cap()) fs.set.alloc(1, oc.cap());
    build_grid(ctx->stream, fs.ws, fs.a.p, fs.na.p, fs.set, ctx->grid_cell);
    build_tile_order(ctx->stream, fs.ws, fs.set, true);
    compute_covariances(ctx->stream, fs.set, k, method);
    std::vector<Cov6> c(oc.n);
    std::vector<int> orig(oc.n);
    MI_HIP(hipMemcpyAsync(c.data(), fs.set.cov.p, c.size() * sizeof(Cov6), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(orig.data(), fs.set.orig.p, oc.n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < oc.n; i++) {  // back to the caller's point order
      float* o = cov6 + 6 * (size_t)orig[i];
      for (int t = 0; t < 6; t++) o[t] = c[i].v[t];
    }
  });
}

// FAST_GICP and the BFGS GICP prepare their clouds the same way (voxel hash + PLANE-regularised k-NN covariances)
static bool needs_covariances(int method) { return method == MI355_METHOD_FAST_GICP || method == MI355_METHOD_GICP; }
static void adopt_bfgs_params(mi355_reg* r) {  // the covariance path reads gp
  r->gp.k_correspondences = r->bp.k_correspondences;
  r->gp.regularization_method = MI355_REG_PLANE;
}

static void reg_build_ndt_target(mi355_reg* reg, bool keep_fp64_leaves = false);

// ---- registration object -------------------------------------------------------------------------------------------
extern "C" mi355_reg* mi355_reg_create(mi355_ctx* ctx, int method, const void* params) {
  if (!ctx || (method != MI355_METHOD_FAST_GICP && method != MI355_METHOD_NDT_OMP && method != MI355_METHOD_ICP && method != MI355_METHOD_GICP))
    return nullptr;
  auto* r = new mi355_reg();
  r->ctx = ctx;
  r->method = method;
  mi355_gicp_default_params(&r->gp);
  mi355_ndt_default_params(&r->np);
  mi355_icp_default_params(&r->ip);
  mi355_gicp_bfgs_default_params(&r->bp);
  if (params) {
    if (method == MI355_METHOD_FAST_GICP) r->gp = *(const mi355_gicp_params*)params;
    else if (method == MI355_METHOD_NDT_OMP) r->np = *(const mi355_ndt_params*)params;
    else if (method == MI355_METHOD_GICP) r->bp = *(const mi355_gicp_bfgs_params*)params;
    else r->ip = *(const mi355_icp_params*)params;
  }
  if (method == MI355_METHOD_GICP) {
    if (r->bp.ransac_iterations != 0 || r->bp.k_correspondences < 1) {
      ctx->err = "GICP: only setRANSACIterations(0) is built; k_correspondences must be positive";
      delete r;
      return nullptr;
    }
    adopt_bfgs_params(r);
  }
  if (method == MI355_METHOD_ICP && r->ip.ransac_iterations != 0) {
    ctx->err = "ICP: only setRANSACIterations(0) (graph_based_slam.cpp:150) is built";
    delete r;
    return nullptr;
  }
  if (method == MI355_METHOD_NDT_OMP) {
    r->ndt = ndt_targets_create();
    r->ndt_solver = ndt_solver_create();
  }
  for (int i = 0; i < 16; i++) r->last_T[i] = (i % 5 == 0) ? 1.f : 0.f;
  return r;
}

extern "C" void mi355_reg_destroy(mi355_reg* reg) {
  if (!reg) return;
  (void)hipSetDevice(reg->ctx->device);
  (void)hipStreamSynchronize(reg->ctx->stream);
  delete reg;
}

extern "C" int mi355_reg_set_params(mi355_reg* reg, const void* params) {
  if (!reg || !params) return MI355_EINVAL;
  if (reg->method == MI355_METHOD_FAST_GICP) {
    const int old_k = reg->gp.k_correspondences, old_m = reg->gp.regularization_method;
    reg->gp = *(const mi355_gicp_params*)params;
    if (old_k != reg->gp.k_correspondences || old_m != reg->gp.regularization_method) { reg->src.has_cov = false; reg->tgt.has_cov = false; }
  } else if (reg->method == MI355_METHOD_ICP) {
    reg->ip = *(const mi355_icp_params*)params;
    if (reg->ip.ransac_iterations != 0) {
      reg->ctx->err = "ICP: only setRANSACIterations(0) (graph_based_slam.cpp:150) is built";
      return MI355_EINVAL;
    }
  } else if (reg->method == MI355_METHOD_GICP) {
    const int old_k = reg->bp.k_correspondences;
    const mi355_gicp_bfgs_params nb = *(const mi355_gicp_bfgs_params*)params;
    if (nb.ransac_iterations != 0 || nb.k_correspondences < 1) {
      reg->ctx->err = "GICP: only setRANSACIterations(0) is built; k_correspondences must be positive";
      return MI355_EINVAL;
    }
    reg->bp = nb;
    adopt_bfgs_params(reg);
    if (old_k != nb.k_correspondences) { reg->src.has_cov = false; reg->tgt.has_cov = false; }
  } else {
    // pclomp re-initialises its voxel grid only when the grid itself changes (setResolution and friends), and does so from
    // the target it already holds: the other setters (iterations, epsilon, step size, threads) leave the target alone
    const mi355_ndt_params old = reg->np;
    reg->np = *(const mi355_ndt_params*)params;
    const bool grid_changed = old.resolution != reg->np.resolution || old.min_points_per_voxel != reg->np.min_points_per_voxel ||
                              old.min_covar_eigvalue_mult != reg->np.min_covar_eigvalue_mult;
    if (grid_changed && reg->has_tgt)
      return guarded(reg->ctx, [&] {
        MI_HIP(hipSetDevice(reg->ctx->device));
        reg_build_ndt_target(reg);
      });
  }
  return MI355_OK;
}

// N1 of a registration object: its target is a set of ONE.  The fp64 leaves (computeHessian after a refined line search, the parity
// read-back of every occupied voxel) are built when something asks for them: the staged target points stay on the device.
static void reg_build_ndt_target(mi355_reg* reg, bool keep_fp64_leaves) {
  const int bound = (int)reg->n_tgt;
  ws_reserve(reg->ws, 1, std::max<size_t>(reg->n_tgt, 1));
  ndt_build_targets(reg->ctx->stream, reg->ws, *reg->ndt, reg->stage_tgt.p, reg->n_tgt, reg->cnt_tgt.p, &bound, 1, std::max<size_t>(reg->n_tgt, 1), reg->np,
                    keep_fp64_leaves || reg->np.legacy_mt_skip == 0);
}
static void reg_need_fp64_leaves(mi355_reg* reg) {
  if (!ndt_targets_have_fp64_leaves(*reg->ndt)) reg_build_ndt_target(reg, true);
}

static void reg_upload(mi355_reg* reg, const mi355_cloud_view* cloud, DevBuf<float4>& stage, DevBuf<int>& cnt) {
  mi355_ctx* ctx = reg->ctx;
  MI_HIP(hipSetDevice(ctx->device));
  check_view(cloud);
  const size_t cap = std::max<size_t>(cloud->n, 1);
  stage.ensure(cap);
  cnt.ensure(1);
  reg->up.upload(ctx->stream, cloud, stage.p);
  const int ni = (int)cloud->n;
  MI_HIP(hipMemcpyAsync(cnt.p, &ni, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  MI_HIP(hipStreamSynchronize(ctx->stream));
}

extern "C" int mi355_reg_set_target(mi355_reg* reg, const mi355_cloud_view* cloud) {
  if (!reg) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_upload(reg, cloud, reg->stage_tgt, reg->cnt_tgt);
    reg->n_tgt = cloud->n;
    reg->has_tgt = false;
    reg->aligned = false;
    if (reg->n_tgt == 0) throw ArgError(MI355_EINVAL, "empty target cloud");
    if (needs_covariances(reg->method)) {
      if ((size_t)reg->gp.k_correspondences > reg->n_tgt) throw ArgError(MI355_EINVAL, "target has fewer points than k_correspondences");
      reg->tgt.alloc(1, reg->n_tgt);
      build_grid(ctx->stream, reg->ws, reg->stage_tgt.p, reg->cnt_tgt.p, reg->tgt, ctx->grid_cell);
      compute_covariances(ctx->stream, reg->tgt, reg->gp.k_correspondences, reg->gp.regularization_method);
    } else if (reg->method == MI355_METHOD_ICP) {  // the NN structure is all point-to-point ICP needs
      reg->tgt.alloc(1, reg->n_tgt);
      build_grid(ctx->stream, reg->ws, reg->stage_tgt.p, reg->cnt_tgt.p, reg->tgt, ctx->grid_cell);
    } else {
      reg_build_ndt_target(reg);
      reg->tgt.alloc(1, reg->n_tgt);
      reg->tgt_grid_stale = true;  // getFitnessScore needs the NN structure of the target as well: mi355_reg_fitness builds it
    }
    reg->has_tgt = true;
  });
}

extern "C" int mi355_reg_set_source(mi355_reg* reg, const mi355_cloud_view* cloud) {
  if (!reg) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_upload(reg, cloud, reg->stage_src, reg->cnt_src);
    reg->n_src = cloud->n;
    reg->has_src = false;
    reg->aligned = false;
    if (reg->n_src == 0) throw ArgError(MI355_EINVAL, "empty source cloud");
    reg->src.alloc(1, reg->n_src);
    reg->src_grid_stale = !needs_covariances(reg->method);
    if (!reg->src_grid_stale) build_grid(ctx->stream, reg->ws, reg->stage_src.p, reg->cnt_src.p, reg->src, ctx->grid_cell);
    if (needs_covariances(reg->method)) {
      if ((size_t)reg->gp.k_correspondences > reg->n_src) throw ArgError(MI355_EINVAL, "source has fewer points than k_correspondences");
      compute_covariances(ctx->stream, reg->src, reg->gp.k_correspondences, reg->gp.regularization_method);
    }
    gicp_alloc(reg->gb, 1, reg->n_src, ctx->compute_units);
    reg->has_src = true;
  });
}

static void reg_require(mi355_reg* reg) {
  if (!reg->has_tgt) throw ArgError(MI355_ESTATE, "setInputTarget has not been called");
  if (!reg->has_src) throw ArgError(MI355_ESTATE, "setInputSource has not been called");
  MI_HIP(hipSetDevice(reg->ctx->device));
  if (needs_covariances(reg->method)) {
    if (!reg->src.has_cov) compute_covariances(reg->ctx->stream, reg->src, reg->gp.k_correspondences, reg->gp.regularization_method);
    if (!reg->tgt.has_cov) compute_covariances(reg->ctx->stream, reg->tgt, reg->gp.k_correspondences, reg->gp.regularization_method);
  }
}

extern "C" int mi355_reg_align(mi355_reg* reg, const float guess16[16], mi355_result* result, float* aligned_out) {
  if (!reg || !result) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_require(reg);
    if (reg->method == MI355_METHOD_FAST_GICP) {
      gicp_align(ctx->stream, reg->gb, reg->src, reg->tgt, reg->gp, nullptr, guess16);
      gicp_results(ctx->stream, reg->gb, reg->src, reg->tgt, result, nullptr);
    } else if (reg->method == MI355_METHOD_GICP) {
      float I[16];
      for (int i = 0; i < 16; i++) I[i] = (i % 5 == 0) ? 1.f : 0.f;
      gicp_bfgs_align(ctx->stream, reg->bfgs, reg->src, reg->tgt, reg->n_src, reg->bp, guess16 ? guess16 : I, result);
      result->n_target = (int)reg->n_tgt;
    } else if (reg->method == MI355_METHOD_ICP) {
      float I[16];
      for (int i = 0; i < 16; i++) I[i] = (i % 5 == 0) ? 1.f : 0.f;
      icp_align(ctx->stream, reg->icp, reg->tgt, reg->stage_src.p, reg->cnt_src.p, reg->n_src, reg->ip, guess16 ? guess16 : I, result);
      result->n_source = (int)reg->n_src;
      result->n_target = (int)reg->n_tgt;
    } else {
      float I[16];
      for (int i = 0; i < 16; i++) I[i] = (i % 5 == 0) ? 1.f : 0.f;
      // a batch of one: the More-Thuente state advances on the device, nothing is read back between evaluations
      if (reg->np.legacy_mt_skip == 0) reg_need_fp64_leaves(reg);
      ndt_solve(ctx->stream, *reg->ndt, *reg->ndt_solver, 1, reg->stage_src.p, reg->n_src, reg->cnt_src.p, (int)reg->n_src, nullptr, guess16 ? guess16 : I, reg->np);
      ndt_solver_results(*reg->ndt_solver, result, nullptr);
      result->n_source = (int)reg->n_src;
      result->n_target = (int)reg->n_tgt;
    }
    memcpy(reg->last_T, result->T, sizeof(reg->last_T));
    reg->aligned = true;
    if (aligned_out) {  // pcl::transformPointCloud(*input_, output, final_transformation_), caller's point order
      reg->aligned_dev.ensure(reg->n_src);
      transform_points(ctx->stream, reg->stage_src.p, reg->aligned_dev.p, reg->n_src, result->T);
      MI_HIP(hipMemcpyAsync(aligned_out, reg->aligned_dev.p, reg->n_src * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
      MI_HIP(hipStreamSynchronize(ctx->stream));
    }
  });
}

extern "C" int mi355_reg_fitness(mi355_reg* reg, double max_range, double* score) {
  if (!reg || !score) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_require(reg);
    if (reg->src_grid_stale) {
      build_grid(ctx->stream, reg->ws, reg->stage_src.p, reg->cnt_src.p, reg->src, ctx->grid_cell);
      reg->src_grid_stale = false;
    }
    if (reg->tgt_grid_stale) {
      build_grid(ctx->stream, reg->ws, reg->stage_tgt.p, reg->cnt_tgt.p, reg->tgt, ctx->grid_cell);
      reg->tgt_grid_stale = false;
    }
    const int zero = 0;
    MI_HIP(hipMemcpyAsync(reg->gb.tgt_of_src.p, &zero, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    gicp_fitness(ctx->stream, reg->gb, reg->src, reg->tgt, max_range, false, reg->last_T);
    double f[2];
    MI_HIP(hipMemcpyAsync(f, reg->gb.fitness.p, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    *score = f[1] > 0 ? f[0] / f[1] : 1.7976931348623157e308;
  });
}

extern "C" int mi355_reg_gicp_linearize(mi355_reg* reg, const double T16[16], double* H36, double* b6, double* error, int* corr) {
  if (!reg || !T16 || !H36 || !b6 || !error) return MI355_EINVAL;
  if (reg->method != MI355_METHOD_FAST_GICP) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_require(reg);
    gicp_linearize_once(ctx->stream, reg->gb, reg->src, reg->tgt, reg->gp, T16, H36, b6, error, corr);
  });
}

extern "C" int mi355_reg_gicp_trial_cost(mi355_reg* reg, const double T_lin16[16], const double T_trial16[16], double* model_cost, double* direct_cost) {
  if (!reg || !T_lin16 || !T_trial16 || !model_cost || !direct_cost) return MI355_EINVAL;
  if (reg->method != MI355_METHOD_FAST_GICP) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_require(reg);
    gicp_trial_cost_probe(ctx->stream, reg->gb, reg->src, reg->tgt, reg->gp, T_lin16, T_trial16, model_cost, direct_cost);
  });
}

extern "C" int mi355_reg_ndt_derivatives(mi355_reg* reg, const double p6[6], double* score, double* grad6, double* hess36, int hessian_fp64) {
  if (!reg || !p6 || !score || !grad6 || !hess36) return MI355_EINVAL;
  if (reg->method != MI355_METHOD_NDT_OMP) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_require(reg);
    if (hessian_fp64) reg_need_fp64_leaves(reg);
    ndt_derivatives(ctx->stream, *reg->ndt, *reg->ndt_solver, reg->stage_src.p, reg->n_src, reg->np, p6, score, grad6, hess36, hessian_fp64 != 0);
  });
}

extern "C" int mi355_reg_ndt_num_cells(mi355_reg* reg, size_t* n_cells) {
  if (!reg || !n_cells || reg->method != MI355_METHOD_NDT_OMP) return MI355_EINVAL;
  return guarded(reg->ctx, [&] {
    if (!reg->has_tgt) throw ArgError(MI355_ESTATE, "setInputTarget has not been called");
    MI_HIP(hipSetDevice(reg->ctx->device));
    reg_need_fp64_leaves(reg);  // (the read-back lists every occupied voxel, the default build only those that get a leaf)
    *n_cells = ndt_num_cells(reg->ctx->stream, *reg->ndt, 0);
  });
}

extern "C" int mi355_reg_ndt_cells(mi355_reg* reg, int64_t* leaf_idx, int* leaf_n, double* mean3, double* cov9, double* icov9, int* min_b3,
                                   int* div_b3) {
  if (!reg || reg->method != MI355_METHOD_NDT_OMP) return MI355_EINVAL;
  return guarded(reg->ctx, [&] {
    if (!reg->has_tgt) throw ArgError(MI355_ESTATE, "setInputTarget has not been called");
    MI_HIP(hipSetDevice(reg->ctx->device));
    reg_need_fp64_leaves(reg);
    ndt_cells(reg->ctx->stream, *reg->ndt, leaf_idx, leaf_n, mean3, cov9, icov9, min_b3, div_b3);
  });
}

// ---- device-resident clouds ----------------------------------------------------------------------------------------
static mi355_cloud* new_cloud(mi355_ctx* ctx, size_t n) {
  auto* c = new mi355_cloud();
  c->ctx = ctx;
  c->n = n;
  c->pts.ensure(std::max<size_t>(n, 1));
  return c;
}

extern "C" mi355_cloud* mi355_cloud_upload(mi355_ctx* ctx, const mi355_cloud_view* points) {
  if (!ctx || !points) return nullptr;
  mi355_cloud* c = nullptr;
  const int rc = guarded(ctx, [&] {
    MI_HIP(hipSetDevice(ctx->device));
    check_view(points);
    c = new_cloud(ctx, points->n);
    extras(ctx)->fs.up.upload(ctx->stream, points, c->pts.p);
  });
  if (rc != MI355_OK) {
    delete c;
    return nullptr;
  }
  return c;
}

extern "C" void mi355_cloud_destroy(mi355_cloud* cloud) {
  if (!cloud) return;
  (void)hipSetDevice(cloud->ctx->device);
  (void)hipStreamSynchronize(cloud->ctx->stream);  // kernels that still read the cloud
  delete cloud;
}

extern "C" size_t mi355_cloud_size(const mi355_cloud* cloud) { return cloud ? cloud->n : 0; }
extern "C" int mi355_cloud_has_covariances(const mi355_cloud* cloud) { return (cloud && cloud->has_cov) ? 1 : 0; }

extern "C" int mi355_cloud_download(mi355_cloud* cloud, float* out_packed, size_t capacity) {
  if (!cloud || (cloud->n && !out_packed)) return MI355_EINVAL;
  return guarded(cloud->ctx, [&] {
    if (cloud->n > capacity) throw ArgError(MI355_ECAPACITY, "output capacity too small");
    MI_HIP(hipSetDevice(cloud->ctx->device));
    if (cloud->n) MI_HIP(hipMemcpyAsync(out_packed, cloud->pts.p, cloud->n * sizeof(float4), hipMemcpyDeviceToHost, cloud->ctx->stream));
    MI_HIP(hipStreamSynchronize(cloud->ctx->stream));
  });
}

extern "C" mi355_cloud* mi355_cloud_concat(mi355_ctx* ctx, mi355_cloud* const* clouds, const float* T16s, int n, int carry_covariances) {
  if (!ctx || n < 0 || (n && (!clouds || !T16s))) return nullptr;
  mi355_cloud* out = nullptr;
  const int rc = guarded(ctx, [&] {
    MI_HIP(hipSetDevice(ctx->device));
    size_t total = 0;
    for (int i = 0; i < n; i++) {
      if (!clouds[i] || clouds[i]->ctx != ctx) throw ArgError(MI355_EINVAL, "cloud belongs to another context");
      if (carry_covariances && !clouds[i]->has_cov) throw ArgError(MI355_ESTATE, "carry_covariances: an input cloud carries no covariances");
      if (carry_covariances && (clouds[i]->cov_k != clouds[0]->cov_k || clouds[i]->cov_method != clouds[0]->cov_method))
        throw ArgError(MI355_ESTATE, "carry_covariances: the input clouds' covariances were computed with different parameters");
      total += clouds[i]->n;
    }
    out = new_cloud(ctx, total);
    if (carry_covariances && n > 0) {
      out->cov.ensure(std::max<size_t>(total, 1));
      out->has_cov = true;
      out->cov_k = clouds[0]->cov_k;
      out->cov_method = clouds[0]->cov_method;
    }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
      transform_points_cov(ctx->stream, clouds[i]->pts.p, out->has_cov ? clouds[i]->cov.p : nullptr, out->pts.p + off,
                           out->has_cov ? out->cov.p + off : nullptr, clouds[i]->n, T16s + 16 * (size_t)i);
      off += clouds[i]->n;
    }
  });
  if (rc != MI355_OK) {
    delete out;
    return nullptr;
  }
  return out;
}

extern "C" mi355_cloud* mi355_cloud_transform(mi355_cloud* cloud, const float T16[16]) {
  if (!cloud || !T16) return nullptr;
  mi355_cloud* one[1] = {cloud};
  return mi355_cloud_concat(cloud->ctx, one, T16, 1, cloud->has_cov ? 1 : 0);
}

extern "C" mi355_cloud* mi355_cloud_prefilter(mi355_cloud* cloud, const mi355_prefilter_params* p) {
  if (!cloud || !p) return nullptr;
  mi355_ctx* ctx = cloud->ctx;
  mi355_cloud* out = nullptr;
  const int rc = guarded(ctx, [&] {
    MI_HIP(hipSetDevice(ctx->device));
    out = new mi355_cloud();
    out->ctx = ctx;
    if (cloud->n == 0) return;
    FilterScratch& fs = extras(ctx)->fs;
    fs.na.ensure(1);
    fs.h_n.ensure(1);
    *fs.h_n.p = (int)cloud->n;
    MI_HIP(hipMemcpyAsync(fs.na.p, fs.h_n.p, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    const float4* res;
    const int* nres;
    run_prefilter(ctx->stream, fs.ws, fs.set, ctx->grid_cell, *p, 1, cloud->n, cloud->pts.p, fs.na.p, fs.ws.stage_a, fs.ws.stage_b, fs.ws.count_a,
                  fs.ws.count_b, &res, &nres);
    MI_HIP(hipMemcpyAsync(fs.h_n.p, nres, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    out->n = (size_t)*fs.h_n.p;
    out->pts.ensure(std::max<size_t>(out->n, 1));
    if (out->n) MI_HIP(hipMemcpyAsync(out->pts.p, res, out->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
  });
  if (rc != MI355_OK) {
    delete out;
    return nullptr;
  }
  return out;
}

extern "C" int mi355_cloud_compute_covariances(mi355_cloud* cloud, int k, int method) {
  if (!cloud) return MI355_EINVAL;
  mi355_ctx* ctx = cloud->ctx;
  return guarded(ctx, [&] {
    if (cloud->has_cov && cloud->cov_k == k && cloud->cov_method == method) return;
    if (cloud->n == 0 || (size_t)k > cloud->n || k < 1) throw ArgError(MI355_EINVAL, "k must be in [1, n]");
    MI_HIP(hipSetDevice(ctx->device));
    FilterScratch& fs = extras(ctx)->fs;
    fs.na.ensure(1);
    const int ni = (int)cloud->n;
    MI_HIP(hipMemcpyAsync(fs.na.p, &ni, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));  // &ni
    if (fs.set.B != 1 || fs.set.cap != cloud->n) fs.set.alloc(1, cloud->n);
    build_grid(ctx->stream, fs.ws, cloud->pts.p, fs.na.p, fs.set, ctx->grid_cell);
    build_tile_order(ctx->stream, fs.ws, fs.set, true);
    compute_covariances(ctx->stream, fs.set, k, method);
    cloud->cov.ensure(cloud->n);
    cov_to_caller_order(ctx->stream, fs.set, cloud->n, cloud->cov.p);
    cloud->has_cov = true;
    cloud->cov_k = k;
    cloud->cov_method = method;
  });
}

// ---- wire / on-disk formats -----------------------------------------------------------------------------------------
extern "C" mi355_cloud* mi355_cloud_from_pointcloud2(mi355_ctx* ctx, const mi355_pointcloud2* m) {
  if (!ctx || !m) return nullptr;
  mi355_cloud* c = nullptr;
  const int rc = guarded(ctx, [&] {
    const size_t n = (size_t)m->width * m->height;
    if (n && !m->data) throw ArgError(MI355_EINVAL, "PointCloud2 data is NULL");
    if (m->is_bigendian) throw ArgError(MI355_EINVAL, "big-endian PointCloud2 is not supported");
    static const int size_of[9] = {0, 1, 1, 2, 2, 4, 4, 4, 8};
    if (m->off_intensity >= 0 && (m->intensity_datatype < 1 || m->intensity_datatype > 8)) throw ArgError(MI355_EINVAL, "bad intensity datatype");
    const int isz = m->off_intensity >= 0 ? size_of[m->intensity_datatype] : 0;
    if (m->off_x < 0 || m->off_y < 0 || m->off_z < 0 || (uint32_t)std::max(m->off_x, std::max(m->off_y, m->off_z)) + 4 > m->point_step ||
        (m->off_intensity >= 0 && (uint32_t)(m->off_intensity + isz) > m->point_step) || (size_t)m->width * m->point_step > m->row_step)
      throw ArgError(MI355_EINVAL, "PointCloud2 field offsets / steps are inconsistent");
    const size_t bytes = (size_t)m->row_step * m->height;
    if (m->data_size < bytes) throw ArgError(MI355_EINVAL, "PointCloud2 data is shorter than row_step * height");
    // pcl::fromROSMsg maps `intensity` only when the field is a FLOAT32 (datatype 7) like pcl::PointXYZI's; else it stays 0
    const int off_intensity = (m->off_intensity >= 0 && m->intensity_datatype == 7) ? m->off_intensity : -1;
    MI_HIP(hipSetDevice(ctx->device));
    c = new_cloud(ctx, n);
    if (n == 0) return;
    FilterScratch& fs = extras(ctx)->fs;
    fs.mask.ensure(bytes);
    MI_HIP(hipMemcpyAsync(fs.mask.p, m->data, bytes, hipMemcpyHostToDevice, ctx->stream));
    pointcloud2_unpack(ctx->stream, fs.mask.p, c->pts.p, m->width, m->height, m->point_step, m->row_step, m->off_x, m->off_y, m->off_z,
                       off_intensity, 7);
    MI_HIP(hipStreamSynchronize(ctx->stream));  // the caller's message buffer and the raw staging are free again
  });
  if (rc != MI355_OK) {
    delete c;
    return nullptr;
  }
  return c;
}

extern "C" int mi355_cloud_to_pointcloud2(mi355_cloud* cloud, void* out_data, size_t capacity_bytes) {
  if (!cloud || (cloud->n && !out_data)) return MI355_EINVAL;
  mi355_ctx* ctx = cloud->ctx;
  return guarded(ctx, [&] {
    if (cloud->n * 32 > capacity_bytes) throw ArgError(MI355_ECAPACITY, "output capacity too small (32 bytes per point)");
    if (cloud->n == 0) return;
    MI_HIP(hipSetDevice(ctx->device));
    FilterScratch& fs = extras(ctx)->fs;
    fs.b.ensure(2 * cloud->n);
    pointcloud2_pack(ctx->stream, cloud->pts.p, fs.b.p, cloud->n);
    MI_HIP(hipMemcpyAsync(out_data, fs.b.p, cloud->n * 32, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
  });
}

// pcl::PCDWriter for PointXYZI: header + one line per point (ASCII, 8 significant digits as `fs.precision(8)` prints them)
// or the packed 16-byte records (binary).  Host only.
extern "C" int mi355_pcd_write(const char* path, const mi355_cloud_view* v, int binary) {
  if (!path || !v || (v->n && !v->data) || v->stride < 16 || v->intensity_offset + 4 > v->stride) return MI355_EINVAL;
  FILE* f = fopen(path, "wb");
  if (!f) return MI355_EINVAL;
  fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
             "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA %s\n", v->n, v->n, binary ? "binary" : "ascii");
  const char* base = (const char*)v->data;
  bool ok = true;
  for (size_t i = 0; i < v->n && ok; i++) {
    float rec[4];
    memcpy(rec, base + i * v->stride, 12);
    memcpy(rec + 3, base + i * v->stride + v->intensity_offset, 4);
    if (binary) {
      ok = fwrite(rec, sizeof(rec), 1, f) == 1;
    } else {
      for (int k = 0; k < 4; k++) {
        if (std::isnan(rec[k])) ok = ok && fprintf(f, "nan%s", k < 3 ? " " : "\n") > 0;
        else ok = ok && fprintf(f, "%.8g%s", (double)rec[k], k < 3 ? " " : "\n") > 0;
      }
    }
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? MI355_OK : MI355_EINVAL;
}

extern "C" int mi355_cloud_save_pcd(mi355_cloud* cloud, const char* path, int binary) {
  if (!cloud || !path) return MI355_EINVAL;
  std::vector<float> host(4 * std::max<size_t>(cloud->n, 1));
  const int rc = mi355_cloud_download(cloud, host.data(), host.size() / 4);
  if (rc != MI355_OK) return rc;
  const mi355_cloud_view v{host.data(), cloud->n, 16, 12};
  return mi355_pcd_write(path, &v, binary);
}

// caller-order device points of a cloud -> the registration object's staging (what the host-pointer entry points fill by upload)
static void reg_take_cloud(mi355_reg* reg, mi355_cloud* cloud, DevBuf<float4>& stage, DevBuf<int>& cnt) {
  mi355_ctx* ctx = reg->ctx;
  if (cloud->ctx != ctx) throw ArgError(MI355_EINVAL, "cloud belongs to another context");
  MI_HIP(hipSetDevice(ctx->device));
  stage.ensure(std::max<size_t>(cloud->n, 1));
  cnt.ensure(1);
  if (cloud->n) MI_HIP(hipMemcpyAsync(stage.p, cloud->pts.p, cloud->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
  const int ni = (int)cloud->n;
  MI_HIP(hipMemcpyAsync(cnt.p, &ni, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  MI_HIP(hipStreamSynchronize(ctx->stream));  // &ni
}

extern "C" int mi355_reg_set_source_cloud(mi355_reg* reg, mi355_cloud* cloud) {
  if (!reg || !cloud) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_take_cloud(reg, cloud, reg->stage_src, reg->cnt_src);
    reg->n_src = cloud->n;
    reg->has_src = false;
    reg->aligned = false;
    if (reg->n_src == 0) throw ArgError(MI355_EINVAL, "empty source cloud");
    reg->src.alloc(1, reg->n_src);
    reg->src_grid_stale = !needs_covariances(reg->method);
    if (!reg->src_grid_stale) build_grid(ctx->stream, reg->ws, reg->stage_src.p, reg->cnt_src.p, reg->src, ctx->grid_cell);
    if (needs_covariances(reg->method)) {
      if ((size_t)reg->gp.k_correspondences > reg->n_src) throw ArgError(MI355_EINVAL, "source has fewer points than k_correspondences");
      const bool cached = cloud->has_cov && cloud->cov_k == reg->gp.k_correspondences && cloud->cov_method == reg->gp.regularization_method;
      if (cached) {
        cov_from_caller_order(ctx->stream, reg->src, reg->n_src, cloud->cov.p);
      } else {  // compute once, keep with the cloud: it follows the key frame into every sub-map
        compute_covariances(ctx->stream, reg->src, reg->gp.k_correspondences, reg->gp.regularization_method);
        cloud->cov.ensure(reg->n_src);
        cov_to_caller_order(ctx->stream, reg->src, reg->n_src, cloud->cov.p);
        cloud->has_cov = true;
        cloud->cov_k = reg->gp.k_correspondences;
        cloud->cov_method = reg->gp.regularization_method;
      }
    }
    gicp_alloc(reg->gb, 1, reg->n_src, ctx->compute_units);
    reg->has_src = true;
  });
}

extern "C" int mi355_reg_set_target_cloud(mi355_reg* reg, mi355_cloud* cloud) {
  if (!reg || !cloud) return MI355_EINVAL;
  mi355_ctx* ctx = reg->ctx;
  return guarded(ctx, [&] {
    reg_take_cloud(reg, cloud, reg->stage_tgt, reg->cnt_tgt);
    reg->n_tgt = cloud->n;
    reg->has_tgt = false;
    reg->aligned = false;
    if (reg->n_tgt == 0) throw ArgError(MI355_EINVAL, "empty target cloud");
    if (needs_covariances(reg->method)) {
      if ((size_t)reg->gp.k_correspondences > reg->n_tgt) throw ArgError(MI355_EINVAL, "target has fewer points than k_correspondences");
      reg->tgt.alloc(1, reg->n_tgt);
      build_grid(ctx->stream, reg->ws, reg->stage_tgt.p, reg->cnt_tgt.p, reg->tgt, ctx->grid_cell);
      const bool carried = cloud->has_cov && cloud->cov_k == reg->gp.k_correspondences && cloud->cov_method == reg->gp.regularization_method;
      if (carried) cov_from_caller_order(ctx->stream, reg->tgt, reg->n_tgt, cloud->cov.p);
      else compute_covariances(ctx->stream, reg->tgt, reg->gp.k_correspondences, reg->gp.regularization_method);
    } else if (reg->method == MI355_METHOD_ICP) {
      reg->tgt.alloc(1, reg->n_tgt);
      build_grid(ctx->stream, reg->ws, reg->stage_tgt.p, reg->cnt_tgt.p, reg->tgt, ctx->grid_cell);
    } else {
      reg_build_ndt_target(reg);
      reg->tgt.alloc(1, reg->n_tgt);
      reg->tgt_grid_stale = true;
    }
    reg->has_tgt = true;
  });
}

// ---- batch ---------------------------------------------------------------------------------------------------------
using Part = mi355_batch::Part;

// A stream restricted to a share of the compute units (experiment: the latency-bound preparation chains and the bandwidth-bound LM
// loops of a pipelined batch on disjoint CUs).  The CU mask's bits interleave over the XCDs, so a contiguous run of bits is the same
// share of every XCD's CUs.  first: the share starts at bit 0, else it ends at the last bit.
static void create_stream(hipStream_t* s, int priority, float cu_share, bool first, int compute_units) {
  if (cu_share >= 1.f || cu_share <= 0.f || compute_units <= 0) {
    MI_HIP(hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority));
    return;
  }
  const int n = std::max(8, std::min(compute_units, (int)(cu_share * (float)compute_units + 0.5f) / 8 * 8));
  std::vector<uint32_t> mask((size_t)(compute_units + 31) / 32, 0u);
  for (int i = 0; i < n; i++) {
    const int bit = first ? i : compute_units - 1 - i;
    mask[(size_t)bit / 32] |= 1u << (bit % 32);
  }
  MI_HIP(hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data()));
}

// How many parts a batch is cut into (mi355_batch: THE PIPELINE).  One for NDT (its state machine advances all pairs together), for
// the per-lane k-NN path and for the counter builds (tuning aids that read one set of counters per launch).
static int batch_part_count(int method, int n_sources) {
  if (method != MI355_METHOD_FAST_GICP) return 1;
  const Tuning& tn = tuning();
  if (!tn.knn_tiles || tn.knn_stats || tn.corr_stats) return 1;
  if (tn.batch_parts > 0) return std::max(1, std::min(std::min(tn.batch_parts, n_sources), 8));
  return n_sources >= 512 ? 2 : 1;  // (measured at 2 048 pairs: 2 parts +3 %, 3 parts +-0, 4 parts -6 %: profiles/r04_parts_ab.txt)
}

// The live sources and targets dealt to the parts: ceil(n / parts) each, the last part(s) what is left - possibly nothing.
static void batch_layout(mi355_batch* b) {
  const int N = (int)b->parts.size();
  const int cs = div_up(b->n_sources, N), ct = div_up(b->n_targets, N);
  const CloudSet& T = b->tgt;
  for (int i = 0; i < N; i++) {
    Part& pt = *b->parts[(size_t)i];
    pt.s0 = std::min(i * cs, b->n_sources);
    pt.ns = std::min(cs, b->n_sources - pt.s0);
    pt.t0 = std::min(i * ct, b->n_targets);
    pt.nt = std::min(ct, b->n_targets - pt.t0);
    pt.src.B = pt.ns;
    pt.gb.S = pt.ns;
    CloudSet& w = pt.tgt;  // the window into the targets
    const size_t o = (size_t)pt.t0, n = (size_t)pt.nt;
    w.B = pt.nt;
    w.cap = T.cap;
    w.slots_cap = T.slots_cap;
    w.pts.alias(T.pts.p + o * T.cap, n * T.cap);
    w.orig.alias(T.orig.p + o * T.cap, n * T.cap);
    w.count.alias(T.count.p + o, n);
    w.meta.alias(T.meta.p + o, n);
    w.table.alias(T.table.p + o * T.slots_cap, n * T.slots_cap);
    if (T.cov.p) w.cov.alias(T.cov.p + o * T.cap, n * T.cap);
    w.built = w.has_cov = w.has_order = w.cell_keys_in_ws = w.order_keys_in_ws = false;
  }
}
static int target_part_of(const mi355_batch* b, int t) {
  const int N = (int)b->parts.size();
  return std::min(t / std::max(div_up(b->n_targets, N), 1), N - 1);
}

extern "C" mi355_batch* mi355_batch_create(mi355_ctx* ctx, int method, const void* params, int n_sources, int n_targets, size_t capacity) {
  if (!ctx || (method != MI355_METHOD_FAST_GICP && method != MI355_METHOD_NDT_OMP && method != MI355_METHOD_ICP && method != MI355_METHOD_GICP) ||
      n_sources <= 0 || n_targets <= 0 || capacity == 0) {
    if (ctx) ctx->err = "mi355_batch_create: bad argument (methods: FAST_GICP, NDT_OMP, ICP, GICP)";
    return nullptr;
  }
  auto* b = new mi355_batch();
  b->ctx = ctx;
  b->method = method;
  mi355_gicp_default_params(&b->gp);
  mi355_ndt_default_params(&b->np);
  mi355_icp_default_params(&b->ip);
  if (params && method == MI355_METHOD_FAST_GICP) b->gp = *(const mi355_gicp_params*)params;
  if (params && method == MI355_METHOD_NDT_OMP) b->np = *(const mi355_ndt_params*)params;
  if (params && method == MI355_METHOD_ICP) b->ip = *(const mi355_icp_params*)params;
  mi355_gicp_bfgs_default_params(&b->bp);
  if (params && method == MI355_METHOD_GICP) b->bp = *(const mi355_gicp_bfgs_params*)params;
  if (method == MI355_METHOD_GICP) {
    if (b->bp.ransac_iterations != 0 || b->bp.k_correspondences < 1) {
      ctx->err = "GICP: only setRANSACIterations(0) is built; k_correspondences must be positive";
      delete b;
      return nullptr;
    }
    b->gp.k_correspondences = b->bp.k_correspondences;  // the covariance path reads gp (as adopt_bfgs_params)
    b->gp.regularization_method = MI355_REG_PLANE;
  }
  if (method == MI355_METHOD_ICP && b->ip.ransac_iterations != 0) {
    ctx->err = "ICP: only setRANSACIterations(0) (graph_based_slam.cpp:150) is built";
    delete b;
    return nullptr;
  }
  b->n_sources = b->alloc_sources = n_sources;
  b->n_targets = b->alloc_targets = n_targets;
  b->cap = capacity;
  const int rc = guarded(ctx, [&] {
    MI_HIP(hipSetDevice(ctx->device));
    b->tgt.alloc(n_targets, capacity);
    if (needs_covariances(method)) b->tgt.cov.ensure((size_t)n_targets * capacity);  // (the parts' windows cannot grow it later)
    if (method == MI355_METHOD_NDT_OMP && !(b->np.resolution > 0)) throw ArgError(MI355_EINVAL, "ndt resolution must be positive");
    const int N = batch_part_count(method, n_sources);
    int prio_lo = 0, prio_hi = 0;
    MI_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    for (int i = 0; i < N; i++) {
      b->parts.emplace_back(new Part());
      Part& pt = *b->parts.back();
      pt.cap_s = div_up(n_sources, N);
      pt.cap_t = div_up(n_targets, N);
      pt.src.alloc(pt.cap_s, capacity);
      if (method != MI355_METHOD_FAST_GICP) gicp_alloc_fitness_only(pt.gb, n_sources, capacity);  // getFitnessScore is the point-to-point one for every method
      else gicp_alloc(pt.gb, pt.cap_s, capacity, ctx->compute_units, n_sources);
      for (int k = 0; k < 2; k++) {
        const int nb = k == 0 ? pt.cap_t : pt.cap_s;
        ws_reserve(pt.ws[k], nb, capacity);
        pt.cnt_in[k].ensure((size_t)nb);
        MI_HIP(hipEventCreateWithFlags(&pt.prep_done[k], hipEventDisableTiming));
        MI_HIP(hipEventCreateWithFlags(&pt.knn_done[k], hipEventDisableTiming));
      }
      MI_HIP(hipEventCreateWithFlags(&pt.lm_done, hipEventDisableTiming));
      // the LM loops run AHEAD of the (lowest-priority) preparation chains: their short kernels get the freed wave slots first
      if (N > 1) create_stream(&pt.lm_stream, prio_hi, tuning().lm_cus, true, ctx->compute_units);
    }
    if (method == MI355_METHOD_NDT_OMP) {
      b->ndt = ndt_targets_create();
      b->ndt_solver = ndt_solver_create();
    }
    if (method == MI355_METHOD_ICP) b->icp = icp_batch_create();
    if (method == MI355_METHOD_GICP) b->bfgs = bfgs_batch_create();
    for (int k = 0; k < 2; k++) {
      auto& sd = b->side[k];
      create_stream(&sd.stream, prio_lo, N > 1 ? tuning().prep_cus : 1.f, false, ctx->compute_units);
      MI_HIP(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
    }
    batch_layout(b);
  });
  if (rc != MI355_OK) {
    delete b;
    return nullptr;
  }
  return b;
}

extern "C" void mi355_batch_destroy(mi355_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  try { b->quiesce(); } catch (...) {}
  delete b;
}

static void batch_set(mi355_batch* b, bool targets, const float4* d_points, const int* counts, const mi355_prefilter_params* pf) {
  mi355_ctx* ctx = b->ctx;
  MI_HIP(hipSetDevice(ctx->device));
  if (!d_points || !counts) throw ArgError(MI355_EINVAL, "NULL points / counts");
  const int k = targets ? 0 : 1;
  auto& sd = b->side[k];
  const int B = targets ? b->n_targets : b->n_sources;
  for (int i = 0; i < B; i++)
    if (counts[i] < 0 || (size_t)counts[i] > b->cap) throw ArgError(MI355_EINVAL, "count exceeds capacity");
  // fork: everything already enqueued on the batch's stream (the producer of d_points, the previous align that still reads
  // these cloud sets) comes first
  const bool side_streams = b->use_side_streams();
  TimerScope timing(&b->timers);
  if (side_streams) {
    MI_HIP(hipEventRecord(sd.fork, b->stream()));
    MI_HIP(hipStreamWaitEvent(sd.stream, sd.fork, 0));
  }
  hipStream_t s = side_streams ? sd.stream : b->stream();
  const bool pipelined = b->parts.size() > 1;
  for (auto& ptp : b->parts) {
    Part& pt = *ptp;
    const int c0 = targets ? pt.t0 : pt.s0, n = targets ? pt.nt : pt.ns;
    if (n <= 0) continue;
    CloudSet& set = targets ? pt.tgt : pt.src;
    if (pt.prep_pending[k]) MI_HIP(hipEventSynchronize(pt.prep_done[k]));  // h_counts of a not-yet-joined call may still be read by its copy
    pt.h_counts[k].assign(counts + c0, counts + c0 + n);
    MI_HIP(hipMemcpyAsync(pt.cnt_in[k].p, pt.h_counts[k].data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    const float4* in = d_points + (size_t)c0 * b->cap;
    const float4* pts = in;
    const int* cnt = pt.cnt_in[k].p;
    Workspace& ws = pt.ws[k];
    if (pf && (pf->min_distance > 0 || pf->leaf_size > 0 || pf->mean_k > 0))
      run_prefilter(s, ws, pt.sor_set[k], ctx->grid_cell, *pf, n, b->cap, in, pt.cnt_in[k].p, ws.stage_a, ws.stage_b, ws.count_a, ws.count_b, &pts, &cnt,
                    (b->method == MI355_METHOD_NDT_OMP && targets) ? nullptr : &set);  // (NDT targets are not indexed next: their leaves are built)
    if (b->method == MI355_METHOD_NDT_OMP) {
      // NDT (one part).  Every target gets its voxel Gaussians (N1) - all of them by ONE launch sequence, sized from the counts the
      // caller gave (an upper bound of what a prefilter leaves): nothing is read back, nothing synchronises.  The NN voxel hash of the
      // targets' points is what getFitnessScore searches: built at align time, and only when a fitness is asked for.  The sources are
      // points in the voxel hash's Morton order (neighbouring lanes of the derivative pass read the same leaves).
      if (targets) {
        if (pts == in && d_points != b->host_stage_tgt.p && d_points != b->cloud_stage_tgt.p) {
          // no prefilter ran (and the points are not in the batch's own staging already): `pts` is the CALLER's buffer, and the targets'
          // point hash (getFitnessScore) is built from it at align time only - possibly at a LATER align than the one that follows this
          // call, by when the caller may have reused the memory (mi355_lidar.h: inputs stay valid until the stream has passed the next
          // align).  The batch keeps its own copy.
          ws.stage_a.ensure((size_t)n * b->cap);
          MI_HIP(hipMemcpyAsync(ws.stage_a.p, in, (size_t)n * b->cap * sizeof(float4), hipMemcpyDeviceToDevice, s));
          pts = ws.stage_a.p;
        }
        ndt_build_targets(s, ws, *b->ndt, pts, b->cap, cnt, pt.h_counts[k].data(), n, b->cap, b->np, b->np.legacy_mt_skip == 0);
        b->ndt_tgt_pts = pts;
        b->ndt_tgt_cnt = cnt;
        set.built = false;
      } else {
        build_grid(s, ws, pts, cnt, set, ctx->grid_cell, false);
        b->ndt_max_src = 0;
        for (int i = 0; i < n; i++) b->ndt_max_src = std::max(b->ndt_max_src, pt.h_counts[k][(size_t)i]);
      }
    } else if (b->method == MI355_METHOD_ICP) {
      // point-to-point ICP: the targets' NN voxel hash is all it needs; the sources are points in that hash's Morton order
      // (neighbouring lanes of k_icp_step walk the same cells of the target)
      build_grid(s, ws, pts, cnt, set, ctx->grid_cell, !targets);
      if (!targets) {
        build_tile_order(s, ws, set);  // the sources' compact 64-point patches: what k_icp_correspond_tile's waves take
        b->icp_max_src = 0;
        for (int i = 0; i < n; i++) b->icp_max_src = std::max(b->icp_max_src, pt.h_counts[k][(size_t)i]);
      }
    } else {
      build_grid(s, ws, pts, cnt, set, ctx->grid_cell, true);
      // (batches only: for ONE cloud - a registration object, the scan matcher's sub-map - the sort costs more than the search gains:
      // measured 2.19 instead of 1.99 ms per scan on the config-4 stream)
      build_tile_order(s, ws, set);
      // A batch of ONE part: the k-NN / covariance kernels of the two chains gain nothing from running at the same time (both are
      // bound by the same VALU and LDS), so the chain that comes second starts its k-NN when the other's is done - its prefilter,
      // sort and hash build still run beside that k-NN - and a kernel's duration is its own (what rocprofv3 reports).  With
      // several parts the chains of part i + 1 are meant to run beside the LM loop of part i anyway, and the 3 % that two
      // interleaved k-NN launches gain are taken.
      if (side_streams && !pipelined && pt.knn_pending[1 - k]) MI_HIP(hipStreamWaitEvent(s, pt.knn_done[1 - k], 0));
      compute_covariances(s, set, b->gp.k_correspondences, b->gp.regularization_method);
      if (side_streams) {
        MI_HIP(hipEventRecord(pt.knn_done[k], s));
        pt.knn_pending[k] = true;
      }
      if (b->method == MI355_METHOD_GICP && !targets) {
        b->bfgs_max_src = 0;
        for (int i = 0; i < n; i++) b->bfgs_max_src = std::max(b->bfgs_max_src, pt.h_counts[k][(size_t)i]);
      }
    }
    MI_HIP(hipEventRecord(pt.prep_done[k], s));
    pt.prep_pending[k] = true;
  }
  if (side_streams) sd.pending = true;
  if (targets) {
    b->tgt.built = b->method != MI355_METHOD_NDT_OMP;  // (NDT: the targets' point hash is built when a fitness is asked for)
    b->tgt.has_cov = needs_covariances(b->method);
  }
}

// Make stream `s` wait for the pending preparation chains of side k (all parts).
static void batch_wait_side(mi355_batch* b, hipStream_t s, int k) {
  for (auto& pt : b->parts)
    if (pt->prep_pending[k]) MI_HIP(hipStreamWaitEvent(s, pt->prep_done[k], 0));
}

extern "C" int mi355_batch_set_targets_dev(mi355_batch* b, const void* d_points, const int* counts, const mi355_prefilter_params* prefilter) {
  if (!b) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    batch_set(b, true, (const float4*)d_points, counts, prefilter);
    b->has_tgt = true;
    b->aligned = false;
  });
}

extern "C" int mi355_batch_set_sources_dev(mi355_batch* b, const void* d_points, const int* counts, const mi355_prefilter_params* prefilter) {
  if (!b) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    batch_set(b, false, (const float4*)d_points, counts, prefilter);
    b->has_src = true;
    b->aligned = false;
  });
}

static int batch_set_host(mi355_batch* b, bool targets, const float* points, const int* counts, const mi355_prefilter_params* pf) {
  if (!b || !points) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    MI_HIP(hipSetDevice(b->ctx->device));
    const int B = targets ? b->n_targets : b->n_sources;
    DevBuf<float4>& stage = targets ? b->host_stage_tgt : b->host_stage_src;
    stage.ensure((size_t)B * b->cap);
    // a second set call of the same side without an align in between: its last chains may still read the buffer
    batch_wait_side(b, b->stream(), targets ? 0 : 1);
    // only the filled part of every segment crosses PCIe (sources of 64 k points in segments sized for 300 k-point sub-maps)
    size_t widest = 0;
    if (counts)
      for (int i = 0; i < B; i++)
        if (counts[i] > 0) widest = std::max(widest, std::min((size_t)counts[i], b->cap));
    if (widest > 0)
      MI_HIP(hipMemcpy2DAsync(stage.p, b->cap * sizeof(float4), points, b->cap * sizeof(float4), widest * sizeof(float4), (size_t)B,
                              hipMemcpyHostToDevice, b->stream()));
    MI_HIP(hipStreamSynchronize(b->stream()));  // the caller's buffer is free when this call returns
    // The chains run on the side's own stream and this call returns: the other side's upload crosses PCIe while they run (with ONE
    // staging buffer the call had to wait for the chain - 186 ms per 2 048-pair step of host inputs instead of ~160).  The buffer is
    // written again by this side's next upload, on the batch's stream behind the align that joined these chains.
    batch_set(b, targets, stage.p, counts, pf);
    if (targets) b->has_tgt = true; else b->has_src = true;
    b->aligned = false;
  });
}
static int batch_set_clouds(mi355_batch* b, bool targets, mi355_cloud* const* clouds, const mi355_prefilter_params* pf) {
  if (!b || !clouds) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    MI_HIP(hipSetDevice(b->ctx->device));
    const int B = targets ? b->n_targets : b->n_sources;
    DevBuf<float4>& stage = targets ? b->cloud_stage_tgt : b->cloud_stage_src;  // stays valid until the next call: no sync needed
    stage.ensure((size_t)B * b->cap);
    batch_wait_side(b, b->stream(), targets ? 0 : 1);  // (a second set call of this side without an align in between)
    std::vector<int> counts((size_t)B);
    for (int i = 0; i < B; i++) {
      if (!clouds[i] || clouds[i]->ctx != b->ctx) throw ArgError(MI355_EINVAL, "NULL cloud / cloud of another context");
      if (clouds[i]->n > b->cap) throw ArgError(MI355_EINVAL, "cloud exceeds the batch capacity");
      counts[(size_t)i] = (int)clouds[i]->n;
      if (clouds[i]->n)
        MI_HIP(hipMemcpyAsync(stage.p + (size_t)i * b->cap, clouds[i]->pts.p, clouds[i]->n * sizeof(float4), hipMemcpyDeviceToDevice, b->stream()));
    }
    batch_set(b, targets, stage.p, counts.data(), pf);
    if (targets) b->has_tgt = true; else b->has_src = true;
    b->aligned = false;
  });
}
// Host clouds given as VIEWS (any stride, one per slot): packed into pinned memory a run of slots at a time - as many slots as fit
// kPinBytes at the run's widest cloud - and sent with one 2-D copy per run, the two pinned buffers taking turns, so the host packs
// run r + 1 while run r crosses PCIe.  (mi355_multi_batch_register used to upload every cloud through its own mi355_cloud: one
// pageable-to-pinned repack, one copy and one stream synchronisation per cloud.)
static int batch_set_views(mi355_batch* b, bool targets, const mi355_cloud_view* views, const mi355_prefilter_params* pf) {
  if (!b || !views) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    MI_HIP(hipSetDevice(b->ctx->device));
    constexpr size_t kPinBytes = 16u << 20;
    const int B = targets ? b->n_targets : b->n_sources;
    DevBuf<float4>& stage = targets ? b->host_stage_tgt : b->host_stage_src;
    stage.ensure((size_t)B * b->cap);
    std::vector<int> counts((size_t)B);
    for (int i = 0; i < B; i++) {
      check_view(&views[i]);
      if (views[i].n > b->cap) throw ArgError(MI355_EINVAL, "cloud exceeds the batch capacity");
      counts[(size_t)i] = (int)views[i].n;
    }
    batch_wait_side(b, b->stream(), targets ? 0 : 1);  // (a second set call of this side without an align in between)
    for (int k = 0; k < 2; k++)
      if (!b->pin_free[k]) MI_HIP(hipEventCreateWithFlags(&b->pin_free[k], hipEventDisableTiming));
    int run = 0;
    for (int i0 = 0; i0 < B;) {
      // the run [i0, i1): its widest cloud sets the pitch
      size_t width = std::max<size_t>(views[i0].n, 1);
      int i1 = i0 + 1;
      while (i1 < B) {
        const size_t w = std::max(width, std::max<size_t>(views[i1].n, 1));
        if (w * (size_t)(i1 + 1 - i0) * sizeof(float4) > kPinBytes) break;
        width = w;
        i1++;
      }
      const int k = run++ & 1;
      PinnedBuf<float4>& pin = b->pin[k];
      if (b->pin_used[k]) MI_HIP(hipEventSynchronize(b->pin_free[k]));  // its previous run (of this call or an earlier one) has left the buffer
      pin.ensure(std::max(width * (size_t)(i1 - i0), kPinBytes / sizeof(float4)));
      for (int i = i0; i < i1; i++) {
        const mi355_cloud_view& v = views[i];
        float4* dst = pin.p + (size_t)(i - i0) * width;
        const char* base = (const char*)v.data;
        if (v.stride == sizeof(float4) && v.intensity_offset == 12) {
          if (v.n) memcpy(dst, base, v.n * sizeof(float4));
        } else {
          for (size_t j = 0; j < v.n; j++) {
            const float* q = (const float*)(base + j * v.stride);
            dst[j] = make_float4(q[0], q[1], q[2], *(const float*)(base + j * v.stride + v.intensity_offset));
          }
        }
      }
      MI_HIP(hipMemcpy2DAsync(stage.p + (size_t)i0 * b->cap, b->cap * sizeof(float4), pin.p, width * sizeof(float4), width * sizeof(float4), (size_t)(i1 - i0),
                              hipMemcpyHostToDevice, b->stream()));
      MI_HIP(hipEventRecord(b->pin_free[k], b->stream()));
      b->pin_used[k] = true;
      i0 = i1;
    }
    // (the caller's buffers were read synchronously: they are free when this call returns; the pinned buffers are guarded by their events)
    batch_set(b, targets, stage.p, counts.data(), pf);
    if (targets) b->has_tgt = true; else b->has_src = true;
    b->aligned = false;
  });
}
extern "C" int mi355_batch_set_targets_views(mi355_batch* b, const mi355_cloud_view* views, const mi355_prefilter_params* pf) {
  return batch_set_views(b, true, views, pf);
}
extern "C" int mi355_batch_set_sources_views(mi355_batch* b, const mi355_cloud_view* views, const mi355_prefilter_params* pf) {
  return batch_set_views(b, false, views, pf);
}
extern "C" int mi355_batch_set_targets_clouds(mi355_batch* b, mi355_cloud* const* clouds, const mi355_prefilter_params* pf) {
  return batch_set_clouds(b, true, clouds, pf);
}
extern "C" int mi355_batch_set_sources_clouds(mi355_batch* b, mi355_cloud* const* clouds, const mi355_prefilter_params* pf) {
  return batch_set_clouds(b, false, clouds, pf);
}
extern "C" int mi355_batch_set_targets(mi355_batch* b, const float* points, const int* counts, const mi355_prefilter_params* pf) {
  return batch_set_host(b, true, points, counts, pf);
}
extern "C" int mi355_batch_set_sources(mi355_batch* b, const float* points, const int* counts, const mi355_prefilter_params* pf) {
  return batch_set_host(b, false, points, counts, pf);
}

extern "C" int mi355_batch_set_stream(mi355_batch* b, void* hip_stream, int use_context_stream) {
  if (!b) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    MI_HIP(hipSetDevice(b->ctx->device));
    b->quiesce();  // nothing of this batch may still be in flight on the old stream
    b->has_own_stream = !use_context_stream;
    b->own_stream = (hipStream_t)hip_stream;
  });
}

extern "C" int mi355_batch_align(mi355_batch* b, const int* tgt_of_src, const float* guesses16, int with_fitness) {
  if (!b) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    if (!b->has_src || !b->has_tgt) throw ArgError(MI355_ESTATE, "sources / targets not set");
    if (!tgt_of_src && b->n_sources != b->n_targets) throw ArgError(MI355_EINVAL, "tgt_of_src required when n_sources != n_targets");
    if (tgt_of_src)
      for (int p = 0; p < b->n_sources; p++)
        if (tgt_of_src[p] < 0 || tgt_of_src[p] >= b->n_targets) throw ArgError(MI355_EINVAL, "tgt_of_src out of range");
    MI_HIP(hipSetDevice(b->ctx->device));
    TimerScope timing(&b->timers);
    b->with_fitness = with_fitness != 0;
    hipStream_t main = b->stream();
    if (b->method == MI355_METHOD_NDT_OMP) {
      Part& pt = *b->parts[0];
      batch_wait_side(b, main, 0);
      batch_wait_side(b, main, 1);
      // blocks until every pair has converged (the More-Thuente states advance on the device; the host polls a counter)
      gicp_set_map(main, pt.gb, tgt_of_src, b->n_targets);
      ndt_solve(main, *b->ndt, *b->ndt_solver, b->n_sources, pt.src.pts.p, b->cap, pt.src.count.p, b->ndt_max_src, tgt_of_src, guesses16, b->np);
      if (b->with_fitness) {
        if (!b->tgt.built) {  // the NN voxel hash of the targets, from the points their leaves were built from (on the targets' workspace)
          build_grid(main, pt.ws[0], b->ndt_tgt_pts, b->ndt_tgt_cnt, pt.tgt, b->ctx->grid_cell, false);
          b->tgt.built = true;
        }
        std::vector<float> T((size_t)b->n_sources * 16);
        ndt_solver_results(*b->ndt_solver, nullptr, T.data());
        gicp_fitness(main, pt.gb, pt.src, b->tgt, 1.7976931348623157e308, false, T.data());
      }
    } else if (b->method == MI355_METHOD_ICP) {
      Part& pt = *b->parts[0];
      batch_wait_side(b, main, 0);
      batch_wait_side(b, main, 1);
      // blocks until every pair is done (the fits and PCL's convergence criteria run on the device; the host polls a counter)
      gicp_set_map(main, pt.gb, tgt_of_src, b->n_targets);
      icp_batch_align(main, *b->icp, b->n_sources, pt.src.pts.p, b->cap, pt.src.count.p, (size_t)b->icp_max_src, b->tgt, tgt_of_src, guesses16, b->ip,
                      pt.src.has_order ? (const int*)pt.src.tile_order.p : (const int*)nullptr);
      if (b->with_fitness) {
        std::vector<float> T((size_t)b->n_sources * 16);
        icp_batch_results(*b->icp, nullptr, T.data());
        size_t seed_pitch = 0;
        const int* seed = icp_batch_corr(*b->icp, &seed_pitch);
        gicp_fitness(main, pt.gb, pt.src, b->tgt, 1.7976931348623157e308, false, T.data(), seed, seed_pitch);
      }
    } else if (b->method == MI355_METHOD_GICP) {
      Part& pt = *b->parts[0];
      batch_wait_side(b, main, 0);
      batch_wait_side(b, main, 1);
      // blocks until every pair is done (the line searches and BFGS updates run on the device; the host polls a counter)
      gicp_set_map(main, pt.gb, tgt_of_src, b->n_targets);
      bfgs_batch_align(main, *b->bfgs, b->n_sources, pt.src, b->tgt, (size_t)b->bfgs_max_src, tgt_of_src, guesses16, b->bp);
      if (b->with_fitness) {
        std::vector<float> T((size_t)b->n_sources * 16);
        bfgs_batch_results(*b->bfgs, nullptr, T.data());
        size_t seed_pitch = 0;
        const int* seed = bfgs_batch_corr(*b->bfgs, &seed_pitch);
        gicp_fitness(main, pt.gb, pt.src, b->tgt, 1.7976931348623157e308, false, T.data(), seed, seed_pitch);
      }
    } else {
      // every part's LM loop on its stream, behind the chains it needs: its own sources' and those of the targets its pairs name
      const bool forked = b->lm_stream(0) != main;
      if (forked) MI_HIP(hipEventRecord(b->side[0].fork, main));  // (what the caller enqueued before this call comes first)
      for (size_t i = 0; i < b->parts.size(); i++) {
        Part& pt = *b->parts[i];
        pt.it_enqueued = 0;
        pt.lm_finished = pt.ns <= 0;
        if (pt.ns <= 0) continue;
        hipStream_t s = b->lm_stream(i);
        if (forked) MI_HIP(hipStreamWaitEvent(s, b->side[0].fork, 0));
        if (pt.prep_pending[1]) MI_HIP(hipStreamWaitEvent(s, pt.prep_done[1], 0));
        unsigned long long need = 0;
        for (int p = pt.s0; p < pt.s0 + pt.ns; p++) need |= 1ull << target_part_of(b, tgt_of_src ? tgt_of_src[p] : p);
        for (size_t j = 0; j < b->parts.size(); j++)
          if (((need >> j) & 1ull) && b->parts[j]->prep_pending[0]) MI_HIP(hipStreamWaitEvent(s, b->parts[j]->prep_done[0], 0));
        std::vector<int> map((size_t)pt.ns);  // pair -> target, batch-wide target ids (NULL: pair p registers against target p)
        for (int p = 0; p < pt.ns; p++) map[(size_t)p] = tgt_of_src ? tgt_of_src[pt.s0 + p] : pt.s0 + p;
        gicp_align_begin(s, pt.gb, pt.src, b->tgt, b->gp, map.data(), guesses16 ? guesses16 + 16 * (size_t)pt.s0 : nullptr);
      }
      const int max_it = b->gp.max_iterations;
      if (b->gp.force_iterations) {
        for (size_t i = 0; i < b->parts.size(); i++) {
          Part& pt = *b->parts[i];
          if (pt.ns > 0) gicp_align_iterate(b->lm_stream(i), pt.gb, pt.src, b->tgt, b->gp, 0, max_it);
        }
      } else {
        // Early stopping: iterations go out in groups of four, each group followed by a poll ("is any pair of this part still
        // optimising?") whose answer the host reads without blocking; the parts take turns, so while one part's answer is on its
        // way the others' next groups are enqueued.  At most GicpBatch::kPolls polls of a part are in flight - then the oldest is
        // waited for - i.e. at most two groups of (empty: converged pairs leave every kernel at once) iterations run past convergence.
        constexpr int kGroup = 4;
        bool any = true;
        while (any) {
          any = false;
          for (size_t i = 0; i < b->parts.size(); i++) {
            Part& pt = *b->parts[i];
            if (pt.lm_finished) continue;
            GicpBatch& gb = pt.gb;
            int n_active = 1;
            while (!pt.lm_finished && gicp_poll_query(gb, false, &n_active))
              if (n_active == 0) pt.lm_finished = true;
            if (!pt.lm_finished && pt.it_enqueued >= max_it) pt.lm_finished = true;
            while (!pt.lm_finished && gb.polls_issued - gb.polls_seen >= GicpBatch::kPolls)
              if (gicp_poll_query(gb, true, &n_active) && n_active == 0) pt.lm_finished = true;
            if (pt.lm_finished) continue;
            const int it1 = std::min(pt.it_enqueued + kGroup, max_it);
            gicp_align_iterate(b->lm_stream(i), gb, pt.src, b->tgt, b->gp, pt.it_enqueued, it1);
            pt.it_enqueued = it1;
            if (it1 < max_it) gicp_poll_active(b->lm_stream(i), gb);
            any = true;
          }
        }
        for (auto& pt : b->parts) {
          int n_active;
          while (gicp_poll_query(pt->gb, true, &n_active)) {}  // leave no poll in flight
        }
      }
      for (size_t i = 0; i < b->parts.size(); i++) {
        Part& pt = *b->parts[i];
        if (pt.ns <= 0) continue;
        hipStream_t s = b->lm_stream(i);
        if (tuning().corr_stats) gicp_corr_stats_print(s, pt.gb);
        if (b->with_fitness) gicp_fitness(s, pt.gb, pt.src, b->tgt, 1.7976931348623157e308, true, nullptr, (const int*)pt.gb.corr.p, pt.src.cap);
        if (s != main) {  // join: what the caller enqueues on the batch's stream next comes after this alignment
          MI_HIP(hipEventRecord(pt.lm_done, s));
          MI_HIP(hipStreamWaitEvent(main, pt.lm_done, 0));
        }
      }
      // (chains no pair needed - targets nobody registers against - are joined too)
      batch_wait_side(b, main, 0);
      batch_wait_side(b, main, 1);
    }
    for (auto& sd : b->side) sd.pending = false;
    for (auto& pt : b->parts) pt->prep_pending[0] = pt->prep_pending[1] = pt->knn_pending[0] = pt->knn_pending[1] = false;
    b->aligned = true;
  });
}

extern "C" int mi355_batch_results(mi355_batch* b, mi355_result* results, double* fitness) {
  if (!b || !results) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    if (!b->aligned) throw ArgError(MI355_ESTATE, "mi355_batch_align has not been called");
    if (fitness && !b->with_fitness) throw ArgError(MI355_ESTATE, "fitness was not requested in mi355_batch_align");
    MI_HIP(hipSetDevice(b->ctx->device));
    if (b->method != MI355_METHOD_FAST_GICP) {
      Part& pt = *b->parts[0];
      if (b->method == MI355_METHOD_NDT_OMP) ndt_solver_results(*b->ndt_solver, results, nullptr);
      else if (b->method == MI355_METHOD_GICP) bfgs_batch_results(*b->bfgs, results, nullptr);
      else icp_batch_results(*b->icp, results, nullptr);
      if (fitness) gicp_fitness_values(b->stream(), pt.gb, fitness);
    } else {
      for (auto& pt : b->parts)  // (the batch's stream has joined every part's)
        if (pt->ns > 0) gicp_results(b->stream(), pt->gb, pt->src, b->tgt, results + pt->s0, fitness ? fitness + pt->s0 : nullptr);
    }
  });
}

extern "C" int mi355_batch_verify_correspondences(mi355_batch* b, long long* n_checked, long long* n_mismatch) {
  if (!b || !n_checked || !n_mismatch) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    if (!b->aligned || b->method != MI355_METHOD_FAST_GICP) throw ArgError(MI355_ESTATE, "verify_correspondences: after mi355_batch_align of a FAST_GICP batch");
    MI_HIP(hipSetDevice(b->ctx->device));
    *n_checked = *n_mismatch = 0;
    for (auto& pt : b->parts) {
      if (pt->ns <= 0) continue;
      long long c = 0, m = 0;
      gicp_verify_correspondences(b->stream(), pt->gb, pt->src, b->tgt, b->gp, &c, &m);
      *n_checked += c;
      *n_mismatch += m;
    }
  });
}

extern "C" int mi355_batch_diagnostics(mi355_batch* b, int* first_converged, long long* knn_points, long long* knn_handed_over) {
  if (!b) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    if (b->method != MI355_METHOD_FAST_GICP) throw ArgError(MI355_ESTATE, "mi355_batch_diagnostics: a FAST_GICP batch");
    if (first_converged && !b->aligned) throw ArgError(MI355_ESTATE, "mi355_batch_diagnostics: after mi355_batch_align");
    MI_HIP(hipSetDevice(b->ctx->device));
    b->quiesce();
    long long np[2] = {0, 0}, nf[2] = {0, 0};
    for (auto& pt : b->parts) {
      if (first_converged && pt->ns > 0) gicp_first_converged(b->stream(), pt->gb, first_converged + pt->s0);
      for (int k = 0; k < 2; k++) {
        CloudSet& set = k == 0 ? pt->tgt : pt->src;
        if (set.B <= 0 || !set.has_cov || !set.fbn.p) continue;  // fbn [0, B): the hand-over queue lengths of the set's last covariance launch
        std::vector<int> fbn((size_t)set.B), cnt((size_t)set.B);
        MI_HIP(hipMemcpy(fbn.data(), set.fbn.p, sizeof(int) * (size_t)set.B, hipMemcpyDeviceToHost));
        MI_HIP(hipMemcpy(cnt.data(), set.count.p, sizeof(int) * (size_t)set.B, hipMemcpyDeviceToHost));
        for (int c = 0; c < set.B; c++) { np[k] += cnt[(size_t)c]; nf[k] += fbn[(size_t)c]; }
      }
    }
    for (int k = 0; k < 2; k++) {
      if (knn_points) knn_points[k] = np[k];
      if (knn_handed_over) knn_handed_over[k] = nf[k];
    }
  });
}

extern "C" int mi355_batch_get_cloud(mi355_batch* b, int targets, int index, float* out_xyzi, float* out_cov6, size_t capacity, size_t* n) {
  if (!b || !n) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    if (index < 0 || index >= (targets ? b->n_targets : b->n_sources)) throw ArgError(MI355_EINVAL, "mi355_batch_get_cloud: index");
    if (!(targets ? b->has_tgt : b->has_src)) throw ArgError(MI355_ESTATE, "mi355_batch_get_cloud: after mi355_batch_set_targets* / _sources*");
    MI_HIP(hipSetDevice(b->ctx->device));
    b->quiesce();
    for (auto& ptp : b->parts) {
      Part& pt = *ptp;
      const int c0 = targets ? pt.t0 : pt.s0, nc = targets ? pt.nt : pt.ns;
      if (index < c0 || index >= c0 + nc) continue;
      const CloudSet& set = targets ? pt.tgt : pt.src;
      if (!set.built) throw ArgError(MI355_ESTATE, "mi355_batch_get_cloud: this side holds no indexed clouds (NDT targets are voxel Gaussians)");
      const int c = index - c0;
      int cnt = 0;
      MI_HIP(hipMemcpy(&cnt, set.count.p + c, sizeof(int), hipMemcpyDeviceToHost));
      *n = (size_t)cnt;
      if ((size_t)cnt > capacity) throw ArgError(MI355_ECAPACITY, "mi355_batch_get_cloud: capacity");
      if (cnt == 0 || !out_xyzi) return;
      std::vector<float4> p((size_t)cnt);
      std::vector<int> orig((size_t)cnt);
      MI_HIP(hipMemcpy(p.data(), set.pts.p + (size_t)c * set.cap, sizeof(float4) * (size_t)cnt, hipMemcpyDeviceToHost));
      MI_HIP(hipMemcpy(orig.data(), set.orig.p + (size_t)c * set.cap, sizeof(int) * (size_t)cnt, hipMemcpyDeviceToHost));
      for (int i = 0; i < cnt; i++) memcpy(out_xyzi + 4 * (size_t)orig[(size_t)i], &p[(size_t)i], sizeof(float4));
      if (out_cov6) {
        if (!set.has_cov || !set.cov.p) throw ArgError(MI355_ESTATE, "mi355_batch_get_cloud: this engine computes no covariances");
        std::vector<Cov6> cv((size_t)cnt);
        MI_HIP(hipMemcpy(cv.data(), set.cov.p + (size_t)c * set.cap, sizeof(Cov6) * (size_t)cnt, hipMemcpyDeviceToHost));
        for (int i = 0; i < cnt; i++) memcpy(out_cov6 + 6 * (size_t)orig[(size_t)i], &cv[(size_t)i], sizeof(Cov6));
      }
      return;
    }
    throw ArgError(MI355_EINVAL, "mi355_batch_get_cloud: index outside every part");
  });
}

extern "C" int mi355_batch_set_side_streams(mi355_batch* b, int mode) {
  if (!b || mode < -1 || mode > 1) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    MI_HIP(hipSetDevice(b->ctx->device));
    b->quiesce();  // nothing of an earlier call may still be running on an internal stream when the mode changes
    b->side_streams = mode;
  });
}

extern "C" int mi355_batch_set_live(mi355_batch* b, int n_sources, int n_targets) {
  if (!b) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    if (n_sources <= 0 || n_sources > b->alloc_sources || n_targets <= 0 || n_targets > b->alloc_targets)
      throw ArgError(MI355_EINVAL, "mi355_batch_set_live: 1 <= live slots <= the slots given to mi355_batch_create");
    MI_HIP(hipSetDevice(b->ctx->device));
    b->quiesce();  // nothing of an earlier call may still be running when the slot counts change
    // every per-cloud / per-pair array is cloud-major, so a smaller batch is a prefix of the allocation (of every part's)
    b->n_sources = n_sources;
    b->n_targets = b->tgt.B = n_targets;
    batch_layout(b);
    b->has_src = b->has_tgt = b->aligned = false;
  });
}

extern "C" int mi355_batch_set_timing(mi355_batch* b, int enable) {
  if (!b) return MI355_EINVAL;
  b->timers.enabled = enable != 0;
  return MI355_OK;
}

extern "C" int mi355_batch_kernel_times(mi355_batch* b, mi355_kernel_time* out, int capacity, int* n_out) {
  if (!b || !n_out || (capacity > 0 && !out)) return MI355_EINVAL;
  return guarded(b->ctx, [&] {
    MI_HIP(hipSetDevice(b->ctx->device));
    b->quiesce();  // every recorded event is complete
    b->timers.collect();
    int n = 0;
    for (int f = 0; f < KF_COUNT; f++) {
      if (!b->timers.brackets[f]) continue;
      if (n < capacity) {
        snprintf(out[n].name, sizeof(out[n].name), "%s", kKernelFamilyName[f]);
        out[n].total_ms = b->timers.ms[f];
        out[n].launches = b->timers.brackets[f];
        out[n].reserved = 0;
      }
      n++;
    }
    *n_out = n;
    if (n <= capacity) b->timers.reset();  // a caller whose array was too small asks again with n entries and loses nothing
  });
}
