// Internal (C++) interface between the translation units of libmi355_lidar.so.
//
// Data layout in HBM.  Everything is *batched*: a CloudSet holds B clouds in fixed-capacity
// segments [c*cap, c*cap + count[c]) of shared arrays, with the per-cloud point count living in device memory, so
// that a whole pipeline (filter -> sort -> index -> covariances -> LM iterations) is enqueued without one host
// synchronisation.  A single registration object is simply a batch of one.
//   pts   float4 (x,y,z,intensity)  16 B/pt   sorted by the Morton code of the NN-grid base cell
//   orig  int32                      4 B/pt   index of the point in the caller's order
//   cov   Cov6 = 6 floats (xx,xy,xz,yy,yz,zz)  24 B/pt, unpadded (one dwordx4 + one dwordx2 load)
//   table HashSlot{key,start,end,-} 16 B/slot open-addressing voxel hash: (level, Morton cell key) -> point range
#pragma once
#include <stdint.h>

#include "common.hpp"
#include "devmath.hpp"

namespace mi355 {

constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;

struct HashSlot {
  uint32_t key;
  int start;
  int end;
  int pad;
};

constexpr int kMaxLevel = 4;

struct GridMeta {
  float ox, oy, oz;   // grid origin (bounding-box minimum)
  float inv_cell, cell;
  int dx, dy, dz;     // cells per axis
  uint32_t hshift;    // slot = (key * 0x9E3779B1) >> hshift
  uint32_t hmask;     // n_slots - 1
  int n;              // points in this cloud
  int overflow;       // VoxelGrid index overflow flag (pass-through)
  int top_level;      // coarsest level of the multi-level voxel hash (cell edge = cell * 2^level)
  int n_cells;        // occupied cells over all levels = entries of the hash
  int level_cells[kMaxLevel + 1];  // ... by level (k_count_cells; k_grid_hash_size drops the coarsest levels of a cloud whose cells
                                   // would not fit its share of the table: the searches then walk finer cells further, results unchanged)
  uint32_t pmask;     // probe-block mask: a probe sequence wraps inside the aligned block of pmask + 1 slots its first slot lies in (grid.hpp next_probe)
};
static_assert(sizeof(GridMeta) == 80, "GridMeta layout");

// Linear probing wraps inside aligned blocks of kProbeBlock slots (128 KB): a block of the table can then be BUILT IN LDS by one workgroup
// and written out with coalesced stores (k_hash_build) - no entry of it lies outside, whatever the collisions.
constexpr uint32_t kProbeBlockLog = 13, kProbeBlock = 1u << kProbeBlockLog;

struct Workspace {
  DevBuf<uint32_t> keys_a, keys_b;
  DevBuf<int> vals_a, vals_b;
  DevBuf<int> hist;
  DevBuf<uint32_t> bounds;  // [B][6] ordered-uint min xyz, max xyz
  DevBuf<int> flags, scan_local, scan_block, totals;
  DevBuf<float4> stage_a, stage_b;  // prefilter stages
  DevBuf<int> count_a, count_b;
  DevBuf<float> fscratch;
  DevBuf<double> dscratch;
  DevBuf<GridMeta> vmeta;  // voxel-grid meta
  DevBuf<int> sort_passes;  // [B] radix passes the keys of a segment need (set by k_voxel_meta / k_grid_meta)
};

struct CloudSet {
  int B = 0;
  size_t cap = 0;
  size_t slots_cap = 0;
  DevBuf<float4> pts;
  DevBuf<int> orig;
  DevBuf<int> count;
  DevBuf<Cov6> cov;   // [B*cap] regularised neighbourhood covariances
  DevBuf<int> nbr;  // [B*cap*k] k-NN lists (sorted indices) between the search and the covariance kernels
  DevBuf<int> fbq, fbn;  // [B*cap] / [2B] points the tile k-NN hands over to k_knn_hard, their number per cloud (+ the tile counters)
  DevBuf<float> fbhint;  // [B*cap] radius known to hold the k nearest neighbours of a handed-over point (0: unknown)
  DevBuf<int> tile_order;  // [B*cap] sorted-array indices in the order k_knn_cov_tile takes them (64 per tile)
  DevBuf<unsigned long long> knn_stats;  // Tuning::knn_stats only: counters of k_knn_cov_tile<STATS>
  DevBuf<GridMeta> meta;
  DevBuf<HashSlot> table;
  bool built = false, has_cov = false, has_order = false;
  bool cell_keys_in_ws = false;   // voxel_grid(..., index_next = this) -> build_grid, nothing in between on that workspace
  bool order_keys_in_ws = false;  // build_grid(..., tile_order_keys = true) -> build_tile_order, nothing in between on that workspace
  void alloc(int B_, size_t cap_);
};

// ---- sort.hip ------------------------------------------------------------------------------------------------
// Stable LSD radix sort of (key32, val32) pairs per cloud segment, `passes` x 8 bits from bit 0.
// Result ends up in (keys_a, vals_a) of the workspace.
// d_passes (optional, [B] on the device): 8-bit passes segment c really needs (its keys are below 2^(8 d_passes[c]) - 1, or below
// 2^(8 d_passes[c]) when every segment element carries a real key); honoured by the in-LDS tile path, ignored by the global passes.
// voxel (optional; only where sort_pairs_in_tiles(cap)): the keys are pcl::VoxelGrid's voxel indices of the points `pts` (grid.hpp
// voxel_index_key), computed by the tile sort itself while it loads a tile - keys_a / vals_a are not read.
struct VoxelKeySource {
  const float4* pts;
  const GridMeta* vmeta;
  double min_distance;
};
bool sort_pairs_in_tiles(size_t cap);  // segments of this capacity go through the in-LDS tile sort + merges (else: global LSD passes)
void sort_pairs(hipStream_t s, Workspace& ws, int B, size_t cap, const int* d_counts, int passes, const int* d_passes = nullptr,
                const VoxelKeySource* voxel = nullptr);
// Exclusive scan of ws.flags per cloud: position(i) = ws.scan_local[i] + ws.scan_block[c*nblk + i/1024]; totals[c].
void scan_flags(hipStream_t s, Workspace& ws, int B, size_t cap, const int* d_counts);
void scan_blocks(hipStream_t s, Workspace& ws, int B, size_t cap);  // the second stage alone (the caller's kernel did the first)
constexpr int kScanTile = 1024;

// ---- cloud.hip -----------------------------------------------------------------------------------------------
void ws_reserve(Workspace& ws, int B, size_t cap);
// Copy a (possibly strided) host/device cloud into packed float4 staging: see api.hip.
// distance filter (A1): in -> out compacted, order preserved.
void distance_filter(hipStream_t s, Workspace& ws, const float4* in, const int* cnt_in, float4* out, int* cnt_out, int B, size_t cap,
                     double min_distance);
// VoxelGrid (P1): in -> out centroids in ascending voxel index.  min_distance >= 0 fuses the distance filter (A1) in front of
// it: the points the filter drops take no part (same output as filter-then-grid, without the flag / scan / compact passes).
// index_next (optional): the CloudSet that build_grid indexes this filter's output into NEXT, on the same workspace with nothing in between
// (base cell index_cell): its geometry is fixed here and the centroid pass leaves the cell keys of build_grid's sort in the workspace.
void voxel_grid(hipStream_t s, Workspace& ws, const float4* in, const int* cnt_in, float4* out, int* cnt_out, int B, size_t cap, float leaf,
                double min_distance = -1.0, CloudSet* index_next = nullptr, float index_cell = 0.f);
// SOR (P2): in -> out, order preserved.  Optional diagnostics (device pointers, [B*cap]).
void sor_filter(hipStream_t s, Workspace& ws, CloudSet& scratch_set, const float4* in, const int* cnt_in, float4* out, int* cnt_out, int B,
                size_t cap, int mean_k, double stddev_mul, float cell, uint8_t* d_keep, float* d_mean_dist);
// Build the NN voxel-hash over `in` (any order): sorted points, orig map, hash table.
// tile_order_keys: the caller runs build_tile_order on the same workspace next — the gather pass leaves the keys of that sort in it
void build_grid(hipStream_t s, Workspace& ws, const float4* in, const int* cnt_in, CloudSet& out, float cell, bool tile_order_keys = false);
// k-NN covariances with regularisation (F2) for every point of a built CloudSet.
void compute_covariances(hipStream_t s, CloudSet& set, int k, int method);
// Optional, between build_grid and compute_covariances: the order in which the tile search takes the points (2-D Hilbert curve over the
// two long axes of the cloud: cloud.hip).  Changes which points share a wave, never a result.  ws: scratch of the sort.
// single_cloud: a caller with ONE cloud (registration object, cloud handle): only with MI355_TILE_ORDER=2 — there the sort costs more
// than the search gains; the switch exists so that the parity tests of the covariances can run on this path.
void build_tile_order(hipStream_t s, Workspace& ws, CloudSet& set, bool single_cloud = false);
// Exact k-NN of queries (packed float4 [m]) in cloud 0 of a built set; idx in the caller's point order.
void knn_query(hipStream_t s, const CloudSet& set, const float4* d_queries, int m, int k, int* d_idx, float* d_d2);
void transform_points(hipStream_t s, const float4* in, float4* out, size_t n, const float* T16_host);
// device-resident clouds: rigid transform of points (+ covariances, C' = R C R^T, when cov_in != nullptr)
void transform_points_cov(hipStream_t s, const float4* in, const Cov6* cov_in, float4* out, Cov6* cov_out, size_t n, const float* T16_host);
// sensor_msgs/PointCloud2 bytes (device) <-> packed float4
void pointcloud2_unpack(hipStream_t s, const unsigned char* d_raw, float4* out, uint32_t width, uint32_t height, uint32_t point_step,
                        uint32_t row_step, int ox, int oy, int oz, int oi, int itype);
void pointcloud2_pack(hipStream_t s, const float4* in, float4* out32, size_t n);
// covariances of a built single-cloud set <-> the caller's point order
void cov_to_caller_order(hipStream_t s, const CloudSet& set, size_t n, Cov6* out_caller);
void cov_from_caller_order(hipStream_t s, CloudSet& set, size_t n, const Cov6* in_caller);

// ---- gicp.hip ------------------------------------------------------------------------------------------------
struct GicpPairState;  // device-side LM state, one per source cloud
struct GicpBatch {
  int S = 0;           // sources (= pairs)
  int S_geometry = 0;  // the pair count the launch geometry (blocks per pair, search kernels) was chosen for: the WHOLE batch's when this
                       // object serves one part of a batch, so that a pair's blocks, and with them the order of its fp64 partial sums, do
                       // not depend on how the batch is cut into parts (results bit-equal whatever the number of parts)
  size_t cap = 0;
  int blocks_per_pair = 0;   // k_gicp_correspond geometry
  int blocks_tile = 1;  // blocks per pair of k_gicp_correspond_tile
  int blocks_lin = 0;        // streaming kernels geometry
  DevBuf<int> tgt_of_src;
  DevBuf<int> corr;          // [S*cap] target (sorted) index per sorted source point
  DevBuf<Cov6> mahal;        // [cap] Mahalanobis matrices of pair 0's last linearisation: written for the parity probes only (gicp_linearize_once)
  DevBuf<float4> qsearch;    // [S*cap] query position of the last full NN search + how far it may move (w)
  DevBuf<unsigned long long> safe_mask;  // [S][cap / 64] bit i of a pair: point i provably keeps its correspondence under EVERY trial pose of the
                                         // current LM iteration (written by the cost pass, read by the next k_gicp_correspond: gicp.hip)
  DevBuf<double> partials;   // [S][blocks][kPartStride = 96]: normal equations + trial-cost model (gicp.hip)
  DevBuf<GicpPairState> state;
  static constexpr int kPolls = 2;  // "any pair still active?" polls in flight (gicp_poll_active / gicp_poll_query)
  DevBuf<int> n_active;
  PinnedBuf<int> h_active;
  hipEvent_t poll_ev[kPolls] = {nullptr, nullptr};
  int polls_issued = 0, polls_seen = 0;
  ~GicpBatch() {
    for (auto& e : poll_ev)
      if (e) (void)hipEventDestroy(e);
  }
  DevBuf<double> fit_partials;
  DevBuf<double> fitness;    // [S][2] (sum, count)
  DevBuf<float> guesses, fit_T;        // device staging that must outlive the enqueued kernels
  std::vector<int> h_map;
  std::vector<float> h_guess, h_fit_T;
  int ppt = 1;              // points per thread and pass in k_gicp_correspond (Tuning::corr_ppt)
  DevBuf<unsigned long long> corr_stats;  // Tuning::corr_stats: counters of k_gicp_correspond_tile (tuning aid), else unallocated
};
// S_geometry: see GicpBatch (0 = S)
void gicp_alloc(GicpBatch& gb, int S, size_t cap, int compute_units, int S_geometry = 0);
void gicp_alloc_fitness_only(GicpBatch& gb, int S, size_t cap);  // for gicp_fitness with host poses only
void gicp_set_map(hipStream_t s, GicpBatch& gb, const int* tgt_of_src_host, int n_targets);
void gicp_fitness_values(hipStream_t s, GicpBatch& gb, double* fitness_out);
// The full LM optimisation of every pair.  guesses_host: S x 16 floats or nullptr.  With force_iterations everything is enqueued and
// the call returns; otherwise it returns once every pair has converged (iterations are enqueued in groups, the device is polled
// without ever leaving the GPU idle).  gicp.hip describes the pieces a caller with several batches in flight interleaves itself.
void gicp_align(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const int* tgt_of_src_host,
                const float* guesses_host);
void gicp_align_begin(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const int* tgt_of_src_host,
                      const float* guesses_host);
void gicp_align_iterate(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, int it0, int it1);
void gicp_poll_active(hipStream_t s, GicpBatch& gb);
bool gicp_poll_query(GicpBatch& gb, bool wait, int* n_active);
void gicp_corr_stats_print(hipStream_t s, GicpBatch& gb);  // Tuning::corr_stats only
// parity probe: incremental correspondences at the final pose vs a from-scratch search (batches of >= 8 pairs: k_gicp_correspond)
void gicp_verify_correspondences(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, long long* n_checked,
                                 long long* n_mismatch);
// seed_corr (optional): [S][seed_pitch] a correspondent per sorted source point from the engine's last iteration (-1: none) - its distance at the
// final pose bounds the search from the start; the score is the same sum of exact nearest distances
void gicp_fitness(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, double max_range, bool use_final_pose,
                  const float* T16_host_all, const int* seed_corr = nullptr, size_t seed_pitch = 0);
void gicp_results(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, mi355_result* out, double* fitness_out);
// per pair: the iteration count at which is_converged first held (-1: never); synchronises
void gicp_first_converged(hipStream_t s, GicpBatch& gb, int* out);
// single linearisation at a given pose (parity tests), pair 0 only
void gicp_linearize_once(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const double* T16,
                         double* H36, double* b6, double* err, int* corr_caller_order);
// linearise at T_lin, then the cost at T_trial through the trial-cost model and through a pass over the points (parity probe: gicp.hip)
void gicp_trial_cost_probe(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const double* T_lin16,
                           const double* T_trial16, double* model_cost, double* direct_cost);

// gicp.hip: pcl::IterativeClosestPoint's correspondences by FAST_GICP's wave-cooperative tiles (k_icp_correspond_tile): cur = [S][cpitch] the
// sources as moved so far, corr / qsearch the rows k_icp_step reads; pairs whose last fit moved them less than min_move are skipped
struct IcpPairState;
// tile_order (optional): [S][opitch] the sources' points in the order of their compact 64-point patches (cloud.hip build_tile_order)
// (moved_correspond_tiles: the same for any engine that keeps its moved sources in `cur` - jobs[pair] = the pair's point count, 0 = not now)
void moved_correspond_tiles(hipStream_t s, const float4* cur, size_t cpitch, const int* jobs, const CloudSet& tgt, const int* tgt_of_src, int* corr,
                            float4* qsearch, float max_corr_d2, float r1, float r_mid, int S, const int* tile_order, size_t opitch);
void icp_correspond_tiles(hipStream_t s, const float4* cur, size_t cpitch, const IcpPairState* state, const CloudSet& tgt, const int* tgt_of_src, int* corr,
                          float4* qsearch, float max_corr_d2, float r1, int S, float min_move, const int* tile_order, size_t opitch, float r_mid, float r1_sigma);

// ---- icp.hip -------------------------------------------------------------------------------------------------
// pcl::IterativeClosestPoint for S pairs at once (every pair's state in HBM, advanced on the device); a registration object is a
// batch of one.
struct IcpBatch;
IcpBatch* icp_batch_create();
void icp_batch_destroy(IcpBatch*);
const int* icp_batch_corr(const IcpBatch& b, size_t* pitch);  // the correspondents of the last iteration (seeds of the fitness search)
// src: [S][spitch] points in any order, counts on the device, none above max_count; tgt: built sets (the NN voxel hash is all
// point-to-point ICP needs); pair p registers against target tgt_of_src_host[p] (nullptr: p).  Blocks until every pair is done.
void icp_batch_align(hipStream_t s, IcpBatch& b, int S, const float4* d_src, size_t spitch, const int* d_src_counts, size_t max_count, const CloudSet& tgt,
                     const int* tgt_of_src_host, const float* guesses_host, const mi355_icp_params& p, const int* src_tile_order = nullptr);
void icp_batch_results(IcpBatch& b, mi355_result* out, float* T16_all);
struct IcpScratch {
  IcpBatch* batch = nullptr;
  ~IcpScratch();
};
// pcl::IterativeClosestPoint::align: source `d_src` (caller order, n points, the count also on the device) against a built target set
void icp_align(hipStream_t s, IcpScratch& sc, const CloudSet& tgt, const float4* d_src, const int* d_src_count, size_t n, const mi355_icp_params& p,
               const float* guess16, mi355_result* out);

// ---- gicp_bfgs.hip ---------------------------------------------------------------------------------------------
// pclomp::GeneralizedIterativeClosestPoint for S pairs at once (every pair's minimiser state in HBM, advanced on the device:
// bfgs_dev.hpp); a registration object is a batch of one.
struct BfgsBatch;
BfgsBatch* bfgs_batch_create();
void bfgs_batch_destroy(BfgsBatch*);
const int* bfgs_batch_corr(const BfgsBatch& b, size_t* pitch);  // the correspondents of the last outer iteration (seeds of the fitness search)
// src / tgt: built sets with PLANE covariances; pair p registers source p against target tgt_of_src_host[p] (nullptr: p); no source
// holds more than max_count points.  Blocks until every pair is done.
void bfgs_batch_align(hipStream_t s, BfgsBatch& b, int S, const CloudSet& src, const CloudSet& tgt, size_t max_count, const int* tgt_of_src_host,
                      const float* guesses_host, const mi355_gicp_bfgs_params& p);
void bfgs_batch_results(BfgsBatch& b, mi355_result* out, float* T16_all);
struct BfgsScratch {
  BfgsBatch* batch = nullptr;
  ~BfgsScratch();
};
// pclomp::GeneralizedIterativeClosestPoint::align: built source and target sets with PLANE covariances
void gicp_bfgs_align(hipStream_t s, BfgsScratch& sc, const CloudSet& src, const CloudSet& tgt, size_t n_src, const mi355_gicp_bfgs_params& p,
                     const float* guess16, mi355_result* out);

// ---- ndt.hip -------------------------------------------------------------------------------------------------
// U targets' voxel Gaussians (N1) in compact arrays; a registration object's target is a set of one.
struct NdtTargets;
NdtTargets* ndt_targets_create();
void ndt_targets_destroy(NdtTargets*);
bool ndt_targets_have_fp64_leaves(const NdtTargets& t);
// cloud u = d_pts[u * pitch ... + count[u]) in any point order; counts on the device, h_bound[u] >= count[u] known to the host (it
// sizes the allocations: nothing is read back, nothing synchronises).  ws: reserved for (U, cap), cap >= every count.
// keep_fp64_leaves: for computeHessian (legacy_mt_skip = 0) and the parity read-back (ndt_cells).
void ndt_build_targets(hipStream_t s, Workspace& ws, NdtTargets& t, const float4* d_pts, size_t pitch, const int* d_counts, const int* h_bound, int U,
                       size_t cap, const mi355_ndt_params& p, bool keep_fp64_leaves);
size_t ndt_num_cells(hipStream_t s, NdtTargets& t, int u);  // synchronises (reads the grid geometry back once per build)
void ndt_cells(hipStream_t s, NdtTargets& t, int64_t* leaf_idx, int* leaf_n, double* mean3, double* cov9, double* icov9, int* min_b3, int* div_b3);
// The Newton / More-Thuente state of P pairs, resident in HBM and advanced on the device (graph_based_slam.cpp:101-119 lets the
// loop-closure back end register its candidates with NDT_OMP; a registration object is a solver of one pair).
struct NdtSolver;
NdtSolver* ndt_solver_create();
void ndt_solver_destroy(NdtSolver*);
// src: [P][src_pitch] points (any order), counts on the device, none above max_count.  Blocks until every pair is done.
void ndt_solve(hipStream_t s, NdtTargets& t, NdtSolver& sv, int P, const float4* d_src, size_t src_pitch, const int* d_src_counts, int max_count,
               const int* tgt_of_src_host, const float* guesses_host, const mi355_ndt_params& p);
// out[i].evaluations = the passes over the source pair i asked for
void ndt_solver_results(NdtSolver& sv, mi355_result* out, float* T16_all);
int ndt_solver_rounds(const NdtSolver& sv);
// parity probe: computeDerivatives of one source against target 0 at the pose p6
void ndt_derivatives(hipStream_t s, NdtTargets& t, NdtSolver& sv, const float4* d_src, size_t n, const mi355_ndt_params& p, const double* p6,
                     double* score, double* grad6, double* hess36, bool hessian_fp64);

}  // namespace mi355
