// Host-side plumbing shared by the translation units of libmi355_lidar.so: error handling, grow-only device
// buffers, the context object behind mi355_ctx.  gfx950 only — no CPU fallback exists: every entry point
// needs a HIP device and fails with MI355_EHIP when there is none.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi355_lidar.h"

namespace mi355 {

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct ArgError : std::runtime_error {
  int code;
  ArgError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define MI_HIP(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess)                                                                                  \
      throw ::mi355::HipError(std::string(#expr) + " -> " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

#define MI_LAUNCH_CHECK() MI_HIP(hipGetLastError())

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owner = true;  // false: a window into another buffer's memory (alias): never freed, never grown
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && owner) (void)hipFree(p);
    p = nullptr;
    n = 0;
    owner = true;
  }
  // grow-only; contents are NOT preserved
  void ensure(size_t need) {
    if (need <= n) return;
    if (!owner) throw std::logic_error("DevBuf::ensure: an alias cannot grow");
    release();
    size_t cap = need + need / 8 + 64;
    MI_HIP(hipMalloc((void**)&p, cap * sizeof(T)));
    n = cap;
  }
  // a window [q, q + m) into memory another DevBuf owns (the parts of a batch: one cloud-major array, one window per part)
  void alias(T* q, size_t m) {
    release();
    p = q;
    n = m;
    owner = false;
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
    std::swap(owner, o.owner);
  }
};

template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t n = 0;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void ensure(size_t need) {
    if (need <= n) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    MI_HIP(hipHostMalloc((void**)&p, need * sizeof(T), hipHostMallocDefault));
    n = need;
  }
};

inline int div_up(long a, long b) { return (int)((a + b - 1) / b); }

// ---- tuning knobs ------------------------------------------------------------------------------------------------
// Launch-geometry knobs, read ONCE per process from MI355_* environment variables (api.hip).  Results never depend on them.
struct Tuning {
  int blocks_per_pair = 0;    // MI355_BLOCKS_PER_PAIR   k_gicp_correspond blocks per pair (0 = from the batch size: 2 .. 64)
  int corr_ppt = 1;           // MI355_PPT               points per thread and pass in k_gicp_correspond (1 or 4)
  int blocks_lin = 0;         // MI355_BLOCKS_LIN        blocks per pair of the streaming kernels (0 = from the batch size)
  int lin_ppt = 1;            // MI355_LIN_PPT           points per thread and tile in k_gicp_linearize (1, 2, 4)
  int knn_tiles = 1;          // MI355_KNN_TILES         1 = wave-cooperative tiles with sub-tile lists (k_knn_cov_tile), 0 = per-lane k-NN search for
                              //                         every point (k_knn_self).  (Round 3's LDS cell bricks, measured slower, live in git history only.)
  int knn_need_div = 2;       // MI355_KNN_NEED_DIV      start level heuristic of the per-lane search
  int tile_waves = 8192;      // MI355_TILE_WAVES        resident one-wave blocks of k_knn_cov_tile over the batch
  int hard_waves = 64;        // MI355_HARD_WAVES        one-wave blocks of k_knn_hard per cloud of a batch (a lone big cloud gets thousands)
  int tile_order = 1;         // MI355_TILE_ORDER        1 = batches take the tiles of k_knn_cov_tile along a Hilbert curve (one more sort per cloud), 0 = along the voxel hash's Morton order, 2 = single clouds too (tests)
  float tile_rscale = 1.0f;   // MI355_TILE_RSCALE       first radius of a tile as a multiple of its density estimate
  float tile_gather = 1.25f;  // MI355_TILE_GATHER       first gather radius of a k-NN tile as a multiple of what its sub-tiles ask for, when the tiles follow build_tile_order (along the Morton order a wider gather was measured slower: 1.0)
  int side_streams = 1;       // MI355_BATCH_SIDE_STREAMS 0 = prepare targets and sources on the batch's own stream
  float prep_cus = 1.0f;      // MI355_PREP_CUS          fraction of the compute units the preparation chains of a pipelined batch may use (a CU mask on
                              //                         their streams: the LAST such share of the mask bits, i.e. of every XCD's CUs); 1 = all
  float lm_cus = 1.0f;        // MI355_LM_CUS            the same for the LM loops' streams (the FIRST such share of the bits); 1 = all
  int batch_parts = 0;        // MI355_BATCH_PARTS       parts a FAST_GICP batch is cut into (the pipeline of api.hip); 0 = by its size: 2 from 512 pairs, else 1
  float grid_cell = 0.5f;     // MI355_GRID_CELL         base cell of the voxel hash (m)
  int hash_slots_per_cell = 4; // MI355_HASH_SLOTS        table slots per occupied cell before rounding up to a power of two (load factor 1/8 - 1/4 at 4)
  int corr_blocks = 0;        // MI355_CORR_BLOCKS       blocks per pair of k_gicp_correspond alone; 0 = MI355_BLOCKS_PER_PAIR / its default (8 192 / pairs)
  int corr_tiles = 4;         // MI355_CORR_TILES        iterations (from the first) in which k_gicp_correspond_tile searches the moving pairs; 0 = never
  float corr_r1 = 0.5f;       // MI355_CORR_R1           first-round radius of a correspondence tile (m)
  int corr_groups = 1;        // MI355_CORR_GROUPS       0 = per-lane correspondence search also for batches of < 8 pairs (no lane groups)
  float margin_step = 0.03f;  // MI355_MARGIN_STEP       a pair tracks search margins once its last accepted step moved it by less than this (m at a 50 m lever arm)
  int corr_mask = 1;          // MI355_CORR_MASK         1 = the linearize pass marks the points whose search margin survives the step budgeted for the iteration
                              //                         and the next k_gicp_correspond reads those bits instead of the points (gicp.hip); 0 = the margin check of round 3
  float mask_slack = 2.f;     // MI355_MASK_SLACK        that budget: this many times what the trials of the previous iteration moved the pose by
  float icp_r1_cells = 2.f;   // MI355_ICP_R1_CELLS      first-round radius of ICP's correspondence tiles, in base cells of the target's voxel hash
  float icp_rmid_cells = 6.f; // MI355_ICP_RMID_CELLS    radius of a middle round for the lanes the first round left, in base cells (a value below the first radius turns it off; a reach of tens of metres never fits the
                              //                         last round: without it they go to the per-lane search; config 5 at 2 048 candidates: 14.3 -> 17.4 k pairs/s)
  float icp_r1_sigma = 3.f;   // MI355_ICP_R1_SIGMA      first-round radius of a pair's tiles = this many RMS correspondence distances of its last iteration, within [half a base cell, MI355_ICP_R1_CELLS] (config 5: +3 %; a huge value = fixed)
  float icp_ball_scale = 1.25f, icp_ball_pad = 0.02f;  // MI355_ICP_BALL_SCALE / _PAD  a settled ICP point is searched in the ball of radius scale x (distance to its last correspondent) + pad (m): its margin can be at most half of what the ball adds
  float icp_min_move = 0.005f; // MI355_ICP_MIN_MOVE     an ICP pair is searched by tiles while its last fit moved it by more than this (m at a 50 m lever arm)
  int corr_stats = 0;         // MI355_CORR_STATS        1 = count what the correspondence tiles scan, print after every align
  int ndt_waves = 32768;      // MI355_NDT_WAVES         waves of k_ndt_derivatives over a batch (each pair gets its share, >= 16, at most one point per lane)
  int knn_stats = 0;          // MI355_KNN_STATS         1 = after every covariance launch, synchronise and print to stderr how many
                              //                         points the tile kernel handed over to k_knn_hard (tuning aid, host side only)
};
const Tuning& tuning();

// ---- per-kernel-family timing (bench.py's roofline block) ----------------------------------------------------------
// HIP events recorded on the LAUNCHING stream around the launches of a family, only while a batch has timing enabled
// (mi355_batch_set_timing).  The durations are read after the streams have been synchronised (mi355_batch_kernel_times).
enum KernelFamily : int {
  KF_KNN_COV_TILE = 0, KF_KNN_HARD, KF_VOXEL_CENTROIDS, KF_SORT, KF_HASH_BUILD, KF_CORRESPOND, KF_CORRESPOND_TILE, KF_LINEARIZE, KF_STEP,
  KF_FITNESS, KF_SOR, KF_NDT_TARGET, KF_NDT_SORT, KF_NDT_LEAVES, KF_NDT_DERIV, KF_NDT_ADVANCE, KF_COUNT
};
extern const char* const kKernelFamilyName[KF_COUNT];
struct KernelTimers {
  struct Rec { int fam; hipEvent_t e0, e1; };
  bool enabled = false;
  std::vector<Rec> recs;
  size_t used = 0;
  double ms[KF_COUNT] = {0};
  int brackets[KF_COUNT] = {0};
  ~KernelTimers() {
    for (auto& r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  }
  // returns the record's INDEX (a bracket opened inside another one may grow `recs`), -1 when no event could be created
  long open(int fam, hipStream_t s) {
    if (used == recs.size()) {
      Rec r{fam, nullptr, nullptr};
      if (hipEventCreate(&r.e0) != hipSuccess) return -1;
      if (hipEventCreate(&r.e1) != hipSuccess) {
        (void)hipEventDestroy(r.e0);
        return -1;
      }
      recs.push_back(r);
    }
    const long i = (long)used++;
    recs[(size_t)i].fam = fam;
    (void)hipEventRecord(recs[(size_t)i].e0, s);
    return i;
  }
  void close(long i, hipStream_t s) { (void)hipEventRecord(recs[(size_t)i].e1, s); }
  // every recorded event must have completed (callers synchronise the streams first)
  void collect() {
    for (size_t i = 0; i < used; i++) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, recs[i].e0, recs[i].e1) == hipSuccess) { ms[recs[i].fam] += t; brackets[recs[i].fam]++; }
    }
    used = 0;
  }
  void reset() {
    for (int f = 0; f < KF_COUNT; f++) { ms[f] = 0; brackets[f] = 0; }
  }
};
extern thread_local KernelTimers* g_timers;  // set by the batch entry points while they enqueue work (api.hip)
extern thread_local int g_family_override;   // >= 0: every bracket opened meanwhile counts for this family (FamilyScope)
struct TimedLaunch {  // RAII bracket around the launches of one family on stream s
  KernelTimers* t = nullptr;
  long r = -1;
  hipStream_t s;
  TimedLaunch(int fam, hipStream_t s_) : s(s_) {
    if (g_timers && g_timers->enabled) {
      t = g_timers;
      r = t->open(g_family_override >= 0 ? g_family_override : fam, s);
    }
  }
  ~TimedLaunch() {
    if (r >= 0) t->close(r, s);
  }
};
struct FamilyScope {  // the launches of a composite stage (SOR = sort + hash + k-NN + threshold + compaction) under one name
  int prev;
  explicit FamilyScope(int fam) : prev(g_family_override) { g_family_override = fam; }
  ~FamilyScope() { g_family_override = prev; }
};
struct TimerScope {
  KernelTimers* prev;
  explicit TimerScope(KernelTimers* t) : prev(g_timers) { g_timers = t; }
  ~TimerScope() { g_timers = prev; }
};

}  // namespace mi355

struct mi355_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  float grid_cell = 0.5f;
  int compute_units = 0;
  char name[128] = {0};
};
