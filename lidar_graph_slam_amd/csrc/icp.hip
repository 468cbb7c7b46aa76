// pcl::IterativeClosestPoint on gfx950: the back end's DEFAULT registration method
// (graph_based_slam.param.yaml:9; graph_based_slam.cpp:142-151, called at :315-322).  Point-to-point ICP is the same
// data-parallel pass as GICP's update_correspondences + linearize with a smaller reduction: per source point an exact 1-NN
// in the target's voxel hash (32 algorithmic bytes: source point + target point) and 17 fp64 sums (count, sum p, sum q,
// sum p q^T, sum d2), from which icp_fit.hpp fits the rigid transform and evaluates PCL's DefaultConvergenceCriteria.
// BATCHED (round 5): S pairs against the targets tgt_of_src names, every pair's state in HBM — k_icp_step moves and searches the
// sources of all pairs still iterating, k_icp_advance (one thread per pair) adds a pair's block partials in block order, fits and
// decides; nothing is read back between iterations, the host polls the number of unfinished pairs every few iterations.  A
// registration object is a batch of one (round 4: one launch + a 136-byte read-back + a stream synchronisation per iteration), so
// LoopClosureCore with the reference's default yaml is one launch sequence for all its candidates.
#include "engine.hpp"
#include "grid.hpp"
#include "icp_fit.hpp"
#include "primitives.hpp"

#ifndef ICP_SEEDED
#define ICP_SEEDED 1  // 0: every iteration searches from scratch (A/B: benchmarks/build_variant.sh)
#endif
namespace mi355 {

constexpr int kIcpBlock = 256;
constexpr int kIcpPartStride = 24;

struct IcpBatch {
  int S = 0, G = 1;
  DevBuf<IcpPairState> state;
  DevBuf<float4> cur;        // [S][pitch] the sources as moved so far (input_transformed)
  DevBuf<int> corr;          // [S][pitch] the correspondent of the previous iteration: the seed of the next search
  size_t pitch = 0;
  DevBuf<int> tgt_of_src, n_active;
  DevBuf<float> guesses;
  DevBuf<double> partials;   // [S][G][kIcpPartStride]
  PinnedBuf<int> h_active;
  PinnedBuf<IcpPairState> h_state;
  std::vector<int> h_map;
  std::vector<float> h_guess;
  int rounds = 0;
};
IcpBatch* icp_batch_create() { return new IcpBatch(); }
void icp_batch_destroy(IcpBatch* b) { delete b; }

__global__ void k_icp_begin(IcpPairState* __restrict__ st, const float* __restrict__ guesses, const int* __restrict__ scount,
                            const GridMeta* __restrict__ tmeta, const int* __restrict__ tgt_of_src, int S) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  float g[16];
  for (int k = 0; k < 16; k++) g[k] = guesses[16 * p + k];
  icp_begin(st[p], g, scount[p], tmeta[tgt_of_src[p]].n);
  if (scount[p] <= 0) { st[p].active = 0; }  // nothing to register: reported as PCL reports a failed fit (not converged, the guess)
}

// cur[i] <- Tinc * (in or cur)[i] (pcl::transformPointCloud arithmetic), then the correspondences and sums at the moved positions:
// PCL estimates the correspondences on the cloud as it stands, fits, THEN moves the cloud — so one launch = [move by the previous
// iteration's fit] + [correspondences and sums]; the first launch moves the input by the guess.  grid = xcd_grid(G, S): the blocks
// of a pair share an XCD (its target's points and hash table in ONE L2); blocks of finished pairs leave at once.
// The search is seeded with the previous iteration's correspondent (the cloud moves by millimetres once the first iterations are
// over): its distance bounds the walk through the voxel hash from the start, the result is the same exact nearest neighbour under
// the same tie rule (config 5 with ICP: 36 iterations per pair on average, every one of them a from-scratch walk before).
__global__ __launch_bounds__(kIcpBlock) void k_icp_step(const float4* __restrict__ src, size_t spitch, float4* __restrict__ cur, size_t cpitch,
                                                        int* __restrict__ corr, const IcpPairState* __restrict__ state, const float4* __restrict__ tpts,
                                                        const int* __restrict__ torig, const HashSlot* __restrict__ ttable,
                                                        const GridMeta* __restrict__ tmeta, size_t tcap, size_t tslots,
                                                        const int* __restrict__ tgt_of_src, float max_d2, int G, int S,
                                                        double* __restrict__ partials) {
  __shared__ double smem[4 * kIcpAcc];
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const IcpPairState& st = state[pair];
  if (!st.active) return;
  const int n = st.n;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.Tinc[k];
  const float4* in = st.moved ? cur + (size_t)pair * cpitch : src + (size_t)pair * spitch;
  float4* out = cur + (size_t)pair * cpitch;
  int* cr = corr + (size_t)pair * cpitch;
  const bool seeded = ICP_SEEDED && st.moved != 0;
  const GridView g = grid_view(tpts, torig, ttable, tmeta, tgt_of_src[pair], tcap, tslots);
  double acc[kIcpAcc];
#pragma unroll
  for (int k = 0; k < kIcpAcc; k++) acc[k] = 0.0;
  for (int i = bx * kIcpBlock + threadIdx.x; i < n; i += G * kIcpBlock) {
    const float4 p0 = in[i];
    const float4 p = make_float4(transform_row(T, p0.x, p0.y, p0.z), transform_row(T + 4, p0.x, p0.y, p0.z), transform_row(T + 8, p0.x, p0.y, p0.z), p0.w);
    out[i] = p;
    Nearest1 nn{g.orig, INFINITY, -1};
    float limit = max_d2;
    if (seeded) {
      const int j0 = cr[i];
      if (j0 >= 0) {
        const float4 q0 = g.pts[j0];
        const float d0 = dist2f(p.x, p.y, p.z, q0.x, q0.y, q0.z);
        if (d0 <= max_d2) { nn.best_d2 = d0; nn.best_j = j0; limit = d0; }
      }
    }
    // CorrespondenceEstimation::determineCorrespondences keeps a pair iff d2 <= max_dist^2
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) grid_search(g, p.x, p.y, p.z, nextafterf(limit, INFINITY), nn, 0, 2);
    const bool found = nn.best_j >= 0 && nn.best_d2 <= max_d2;
    cr[i] = found ? nn.best_j : -1;
    if (found) {
      const float4 q = g.pts[nn.best_j];
      acc[0] += 1.0;
      acc[1] += (double)p.x; acc[2] += (double)p.y; acc[3] += (double)p.z;
      acc[4] += (double)q.x; acc[5] += (double)q.y; acc[6] += (double)q.z;
      acc[7] += (double)p.x * (double)q.x; acc[8] += (double)p.x * (double)q.y; acc[9] += (double)p.x * (double)q.z;
      acc[10] += (double)p.y * (double)q.x; acc[11] += (double)p.y * (double)q.y; acc[12] += (double)p.y * (double)q.z;
      acc[13] += (double)p.z * (double)q.x; acc[14] += (double)p.z * (double)q.y; acc[15] += (double)p.z * (double)q.z;
      acc[16] += (double)nn.best_d2;
    }
  }
  block_sum<kIcpAcc, 4>(acc, smem);
  if (threadIdx.x == 0) {
    double* o = partials + ((size_t)pair * G + bx) * kIcpPartStride;
#pragma unroll
    for (int k = 0; k < kIcpAcc; k++) o[k] = acc[k];
  }
}

__device__ __forceinline__ int icp_blocks_dev(int n, int G) { return max(1, min(G, (n + kIcpBlock - 1) / kIcpBlock)); }

// One thread per pair: the pair's block partials added in block order (bitwise reproducible), the fit, the convergence criteria.
__global__ __launch_bounds__(64) void k_icp_advance(IcpPairState* __restrict__ state, const double* __restrict__ partials, IcpConsts c, int G, int S,
                                                    int* __restrict__ n_active) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= S) return;
  IcpPairState& st = state[pair];
  if (!st.active) return;
  const int nb = min(G, max(1, (st.n + kIcpBlock - 1) / kIcpBlock));  // blocks past the end of the cloud wrote nothing
  double sums[kIcpAcc];
#pragma unroll
  for (int k = 0; k < kIcpAcc; k++) sums[k] = 0.0;
  for (int b = 0; b < nb; b++) {
    const double* row = partials + ((size_t)pair * G + b) * kIcpPartStride;
#pragma unroll
    for (int k = 0; k < kIcpAcc; k++) sums[k] += row[k];
  }
  icp_advance(st, sums, c);
  if (n_active && st.active) atomicAdd(n_active, 1);
}

// src: [S][spitch] points (any order), counts on the device; tgt: the targets' NN voxel hash (built sets); pair p -> target
// tgt_of_src_host[p] (nullptr: p).  Blocks until every pair is done; the final states stay in the batch (icp_batch_results).
void icp_batch_align(hipStream_t s, IcpBatch& b, int S, const float4* d_src, size_t spitch, const int* d_src_counts, size_t max_count, const CloudSet& tgt,
                     const int* tgt_of_src_host, const float* guesses_host, const mi355_icp_params& p) {
  b.S = S;
  b.h_map.resize((size_t)S);
  b.h_guess.resize((size_t)S * 16);
  for (int i = 0; i < S; i++) {
    const int u = tgt_of_src_host ? tgt_of_src_host[i] : i;
    if (u < 0 || u >= tgt.B) throw ArgError(MI355_EINVAL, "tgt_of_src entry out of range");
    b.h_map[(size_t)i] = u;
    for (int k = 0; k < 16; k++) b.h_guess[(size_t)i * 16 + k] = guesses_host ? guesses_host[(size_t)i * 16 + k] : ((k % 5 == 0) ? 1.f : 0.f);
  }
  // blocks per pair: ~4 points per thread for a lone pair (the chip is not full anyway), about 8 k blocks over a batch
  const int Gcap = std::max(2, std::min(1024, 8192 / std::max(S, 1)));
  b.G = std::max(1, std::min(Gcap, div_up((long)std::max<size_t>(max_count, 1), kIcpBlock * 4)));
  b.pitch = std::max<size_t>(max_count, 1);
  b.state.ensure((size_t)S);
  b.h_state.ensure((size_t)S);
  b.cur.ensure((size_t)S * b.pitch);
  b.corr.ensure((size_t)S * b.pitch);
  b.tgt_of_src.ensure((size_t)S);
  b.guesses.ensure((size_t)S * 16);
  b.partials.ensure((size_t)S * b.G * kIcpPartStride);
  b.n_active.ensure(1);
  b.h_active.ensure(1);
  MI_HIP(hipMemcpyAsync(b.tgt_of_src.p, b.h_map.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(b.guesses.p, b.h_guess.data(), (size_t)S * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_icp_begin, dim3(div_up(S, 64)), dim3(64), 0, s, b.state.p, (const float*)b.guesses.p, d_src_counts, (const GridMeta*)tgt.meta.p,
                     (const int*)b.tgt_of_src.p, S);
  const double md2 = p.max_correspondence_distance * p.max_correspondence_distance;
  const float max_d2 = md2 > 3.0e38 ? INFINITY : (float)md2;
  IcpConsts c;
  c.max_iterations = p.max_iterations;
  c.transformation_epsilon = p.transformation_epsilon;
  c.euclidean_fitness_epsilon = p.euclidean_fitness_epsilon;
  c.rot_threshold = p.rotation_epsilon > 0 ? p.rotation_epsilon : 1.0 - p.transformation_epsilon;
  const int chunk = 8;
  const dim3 grid(xcd_grid(b.G, S));
  b.rounds = 0;
  for (int it = 0; it < std::max(p.max_iterations, 1);) {
    MI_HIP(hipMemsetAsync(b.n_active.p, 0, sizeof(int), s));
    for (int k = 0; k < chunk && it < std::max(p.max_iterations, 1); k++, it++) {
      TimedLaunch tl(KF_CORRESPOND, s);
      hipLaunchKernelGGL(k_icp_step, grid, dim3(kIcpBlock), 0, s, d_src, spitch, b.cur.p, b.pitch, b.corr.p, (const IcpPairState*)b.state.p, (const float4*)tgt.pts.p,
                         (const int*)tgt.orig.p, (const HashSlot*)tgt.table.p, (const GridMeta*)tgt.meta.p, tgt.cap, tgt.slots_cap,
                         (const int*)b.tgt_of_src.p, max_d2, b.G, S, b.partials.p);
      const bool last = k == chunk - 1 || it + 1 >= std::max(p.max_iterations, 1);
      hipLaunchKernelGGL(k_icp_advance, dim3(div_up(S, 64)), dim3(64), 0, s, b.state.p, (const double*)b.partials.p, c, b.G, S, last ? b.n_active.p : nullptr);
    }
    MI_LAUNCH_CHECK();
    b.rounds = it;
    MI_HIP(hipMemcpyAsync(b.h_active.p, b.n_active.p, sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (*b.h_active.p == 0) break;
  }
  MI_HIP(hipMemcpyAsync(b.h_state.p, b.state.p, (size_t)S * sizeof(IcpPairState), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void icp_batch_results(IcpBatch& b, mi355_result* out, float* T16_all) {
  for (int i = 0; i < b.S; i++) {
    const IcpPairState& st = b.h_state.p[i];
    if (out) {
      mi355_result& o = out[i];
      memset(&o, 0, sizeof(o));
      memcpy(o.T, st.final_T, sizeof(st.final_T));
      o.converged = st.converged;
      o.iterations = st.iterations;
      o.error = st.mse;
      o.n_source = st.n;
      o.n_target = st.n_target;
    }
    if (T16_all) memcpy(T16_all + (size_t)i * 16, st.final_T, 16 * sizeof(float));
  }
}

// pcl::IterativeClosestPoint::align of ONE pair: a batch of one (source d_src in the caller's order, its count on the device).
void icp_align(hipStream_t s, IcpScratch& sc, const CloudSet& tgt, const float4* d_src, const int* d_src_count, size_t n, const mi355_icp_params& p,
               const float* guess16, mi355_result* out) {
  if (!sc.batch) sc.batch = icp_batch_create();
  icp_batch_align(s, *sc.batch, 1, d_src, n, d_src_count, n, tgt, nullptr, guess16, p);
  icp_batch_results(*sc.batch, out, nullptr);
}
IcpScratch::~IcpScratch() {
  if (batch) icp_batch_destroy(batch);
}

}  // namespace mi355
