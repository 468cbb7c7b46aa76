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
#ifndef ICP_FAR_MARGINS
#define ICP_FAR_MARGINS 1  // 0 (A/B): a point whose seed is farther than two base cells is searched at every iteration, without a margin
#endif
#ifndef ICP_TILES
#define ICP_TILES 1  // 0: per-lane searches only (A/B)
#endif
#ifndef ICP_MARGINS
#define ICP_MARGINS 1  // 0: every point is searched at every iteration (round 5; A/B)
#endif
namespace mi355 {

constexpr int kIcpBlock = 256;
constexpr int kIcpTileIterations = 48;      // iterations (from the first) in which k_icp_correspond_tile is launched at all (it skips settled pairs itself)
constexpr int kIcpPartStride = 24;

struct IcpBatch {
  int S = 0, G = 1;
  DevBuf<IcpPairState> state;
  DevBuf<float4> cur;        // [S][pitch] the sources as moved so far (input_transformed)
  DevBuf<int> corr;          // [S][pitch] the correspondent of the previous iteration: the seed of the next search
  DevBuf<float4> qsearch;    // [S][pitch] where a point stood when it was last SEARCHED (x, y, z) and how far it may move from there
                             // before its correspondent can change (w; <= 0: search at every iteration)
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
const int* icp_batch_corr(const IcpBatch& b, size_t* pitch) { *pitch = b.pitch; return b.corr.p; }

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
// Round 6 - INCREMENTAL and still exact, as FAST_GICP's update_correspondences is here (gicp.hip): a search also learns the distance d2
// of the runner-up (or that there is none within the ball it covered), and while the point stands less than (d2 - d1) / 2 from where
// it was searched its nearest neighbour is provably the same point - nearer than any other by a strict margin, so no tie can arise - and,
// the margin being also below (max distance - d1), still within range.  Such a point costs its own load, its stored query and one
// gather; ~30 of an alignment's ~34 iterations move the cloud by less than a millimetre.
// cur[i] <- Tinc * (cur or in)[i] for every pair still iterating (pcl::transformPointCloud arithmetic): PCL moves the cloud by each
// iteration's fit before it estimates the next correspondences; the first launch moves the input by the guess.
__global__ __launch_bounds__(kIcpBlock) void k_icp_move(const float4* __restrict__ src, size_t spitch, float4* __restrict__ cur, size_t cpitch,
                                                        const IcpPairState* __restrict__ state, int G, int S) {
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const IcpPairState& st = state[pair];
  if (!st.active) return;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.Tinc[k];
  const float4* in = st.moved ? cur + (size_t)pair * cpitch : src + (size_t)pair * spitch;
  float4* out = cur + (size_t)pair * cpitch;
  for (int i = bx * kIcpBlock + threadIdx.x; i < st.n; i += G * kIcpBlock) {
    const float4 p0 = in[i];
    out[i] = make_float4(transform_row(T, p0.x, p0.y, p0.z), transform_row(T + 4, p0.x, p0.y, p0.z), transform_row(T + 8, p0.x, p0.y, p0.z), p0.w);
  }
}

// A wave pays for a search whenever ONE of its lanes needs one, so the points that have left their margins (a few per cent per
// iteration) are COMPACTED into an LDS queue and searched a block's worth at a time on dense waves - the scheme of k_gicp_correspond.
// Queue positions come from ballots and a prefix over the block's waves (not from an atomic counter): which thread searches which
// point, and with it the order of every thread's fp64 sums, is the same in every run.
constexpr int kIcpPpt = 2, kIcpTile = kIcpBlock * kIcpPpt;
__global__ __launch_bounds__(kIcpBlock) void k_icp_step(const float4* __restrict__ cur, size_t cpitch,
                                                        int* __restrict__ corr, float4* __restrict__ qsearch, const IcpPairState* __restrict__ state, const float4* __restrict__ tpts,
                                                        const int* __restrict__ torig, const HashSlot* __restrict__ ttable,
                                                        const GridMeta* __restrict__ tmeta, size_t tcap, size_t tslots,
                                                        const int* __restrict__ tgt_of_src, float max_d2, int G, int S,
                                                        double* __restrict__ partials, float ball_scale, float ball_pad) {
  __shared__ double smem[4 * kIcpAcc];
  __shared__ int queue[kIcpTile + kIcpBlock];  // what is left of the earlier tiles (< one block's worth) + this tile's
  __shared__ int wq[kIcpPpt][kIcpBlock / kWave];
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const IcpPairState& st = state[pair];
  if (!st.active) return;
  const int n = st.n;
  const float4* out = cur + (size_t)pair * cpitch;  // the pair's points where k_icp_move has put them
  int* cr = corr + (size_t)pair * cpitch;
  float4* qsr = qsearch + (size_t)pair * cpitch;
  const float max_d = sqrtf(max_d2);
  const GridView g = grid_view(tpts, torig, ttable, tmeta, tgt_of_src[pair], tcap, tslots);
  const int lane = lane_id(), wv = wave_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  double acc[kIcpAcc];
#pragma unroll
  for (int k = 0; k < kIcpAcc; k++) acc[k] = 0.0;
  auto add_pair = [&](const float4& p, const float4& q, float d2) {
    acc[0] += 1.0;
    acc[1] += (double)p.x; acc[2] += (double)p.y; acc[3] += (double)p.z;
    acc[4] += (double)q.x; acc[5] += (double)q.y; acc[6] += (double)q.z;
    acc[7] += (double)p.x * (double)q.x; acc[8] += (double)p.x * (double)q.y; acc[9] += (double)p.x * (double)q.z;
    acc[10] += (double)p.y * (double)q.x; acc[11] += (double)p.y * (double)q.y; acc[12] += (double)p.y * (double)q.z;
    acc[13] += (double)p.z * (double)q.x; acc[14] += (double)p.z * (double)q.y; acc[15] += (double)p.z * (double)q.z;
    acc[16] += (double)d2;
  };
  int nq = 0;  // queued points (block-uniform)
  for (int base = bx * kIcpTile; base < n; base += G * kIcpTile) {
    const bool last_tile = base + G * kIcpTile >= n;
    // ---- phase 1: a point that still stands inside the margin of its last search (k_icp_correspond_tile's, or this kernel's) keeps its correspondent
    int need_i[kIcpPpt], rank[kIcpPpt];
#pragma unroll
    for (int u = 0; u < kIcpPpt; u++) {
      const int i = base + u * kIcpBlock + (int)threadIdx.x;
      bool need = false;
      if (i < n) {
        const float4 p = out[i];
        bool keep = false;
        if (ICP_MARGINS) {  // (corr = -1 / a NaN margin: never searched, or left undecided by the tiles)
          const int j0 = cr[i];
          if (j0 >= 0) {
            const float4 qs = qsr[i];
            if (qs.w > 0.f && dist2f(p.x, p.y, p.z, qs.x, qs.y, qs.z) < qs.w * qs.w) {
              const float4 q0 = g.pts[j0];
              const float d0 = dist2f(p.x, p.y, p.z, q0.x, q0.y, q0.z);
              if (d0 <= max_d2) {  // (the margin keeps the point within range too; the test costs nothing)
                keep = true;
                add_pair(p, q0, d0);
              }
            }
          }
        }
        if (!keep) {
          if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) need = true;
          else cr[i] = -1;
        }
      }
      need_i[u] = need ? i : -1;
      const unsigned long long m = __ballot(need);
      rank[u] = __popcll(m & below);
      if (lane == 0) wq[u][wv] = __popcll(m);
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int u = 0; u < kIcpPpt; u++)
#pragma unroll
      for (int w = 0; w < kIcpBlock / kWave; w++) {
        const int t = wq[u][w];
        if (need_i[u] >= 0 && w == wv) queue[nq + total + rank[u]] = need_i[u];
        total += t;
      }
    nq += total;
    __syncthreads();
    // ---- phase 2: dense search of the queued points
    while (nq >= kIcpBlock || (last_tile && nq > 0)) {
      const int take = min(nq, kIcpBlock);
      nq -= take;
      if ((int)threadIdx.x < take) {
        const int i = queue[nq + threadIdx.x];
        const float4 p = out[i];
        Nearest1 nn{g.orig, INFINITY, -1};
        float limit = max_d2;
        if (ICP_SEEDED) {
          const int j0 = cr[i];
          if (j0 >= 0) {
            const float4 q0 = g.pts[j0];
            const float d0 = dist2f(p.x, p.y, p.z, q0.x, q0.y, q0.z);
            if (d0 <= max_d2) { nn.best_d2 = d0; nn.best_j = j0; limit = d0; }
          }
        }
        // CorrespondenceEstimation::determineCorrespondences keeps a pair iff d2 <= max_dist^2
        float margin = 0.f;  // (0: nothing is known about the runner-up - searched again next time, then from a close seed)
        if (ICP_MARGINS && nn.best_j >= 0 && limit < 4.f * g.m.cell * g.m.cell) {
          // a seed within two base cells: everything inside a ball a little larger than the seed distance - the nearest, and the runner-up
          // if it is in there (else it is at least the ball's radius away) - from a handful of cells of the level that fits the ball
          const float r = sqrtf(limit) * ball_scale + ball_pad;
          Nearest2 n2{g.orig, limit, nn.best_j, INFINITY, r * r};
          grid_ball_search(g, p.x, p.y, p.z, n2, 2.1f);
          nn.best_d2 = n2.d1;
          nn.best_j = n2.j1;
          if (n2.d1 <= max_d2) {
            const float d1 = sqrtf(n2.d1);
            margin = fminf(0.5f * (sqrtf(fminf(n2.d2nd, r * r)) - d1), max_d - d1) * (1.f - 1e-4f) - 1e-5f;  // (fp32 slack, as finish_search in gicp.hip)
          }
        } else if (ICP_MARGINS && ICP_FAR_MARGINS) {
          // a far seed, or none: the walk covers a little more than the seed distance (half a base cell), from the level whose cells are
          // about that long, and keeps the runner-up - a point metres from anything earns a margin too instead of being searched again
          // at every iteration
          const float lim_r = sqrtf(limit) + 0.5f * g.m.cell;
          const float lim2 = lim_r * lim_r;
          int level0 = 0;
          if (nn.best_j >= 0) {
            const float rel = sqrtf(limit) * g.m.inv_cell;
            level0 = (rel <= 1.f) ? 0 : min(g.m.top_level, (int)ceilf(log2f(rel)));
          }
          Nearest2 n2{g.orig, nn.best_d2, nn.best_j, INFINITY, lim2};
          const float ex2 = fminf(grid_search(g, p.x, p.y, p.z, lim2, n2, level0, 1), lim2);  // (cells beyond lim2 are pruned)
          nn.best_d2 = n2.d1;
          nn.best_j = n2.j1;
          if (n2.j1 >= 0 && n2.d1 <= max_d2) {
            const float d1 = sqrtf(n2.d1);
            margin = fminf(0.5f * (sqrtf(fminf(n2.d2nd, ex2)) - d1), max_d - d1) * (1.f - 1e-4f) - 1e-5f;
          }
        } else {
          grid_search(g, p.x, p.y, p.z, nextafterf(limit, INFINITY), nn, 0, 2);
        }
        if (ICP_MARGINS) qsr[i] = make_float4(p.x, p.y, p.z, margin);
        const bool found = nn.best_j >= 0 && nn.best_d2 <= max_d2;
        cr[i] = found ? nn.best_j : -1;
        if (found) add_pair(p, g.pts[nn.best_j], nn.best_d2);
      }
    }
    __syncthreads();  // the searched entries have been read: the next tile appends behind what is left
  }
  block_sum<kIcpAcc, 4>(acc, smem);
  if (threadIdx.x == 0) {
    double* o = partials + ((size_t)pair * G + bx) * kIcpPartStride;
#pragma unroll
    for (int k = 0; k < kIcpAcc; k++) o[k] = acc[k];
  }
}

// One thread per pair: the pair's block partials added in block order (bitwise reproducible), the fit, the convergence criteria.
__global__ __launch_bounds__(64) void k_icp_advance(IcpPairState* __restrict__ state, const double* __restrict__ partials, IcpConsts c, int G, int S,
                                                    int* __restrict__ n_active) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= S) return;
  IcpPairState& st = state[pair];
  if (!st.active) return;
  const int nb = min(G, max(1, (st.n + kIcpTile - 1) / kIcpTile));  // (the blocks past the end of the cloud wrote zeros)
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
                     const int* tgt_of_src_host, const float* guesses_host, const mi355_icp_params& p, const int* src_tile_order) {
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
  b.G = std::max(1, std::min(Gcap, div_up((long)std::max<size_t>(max_count, 1), kIcpTile * 2)));
  b.pitch = std::max<size_t>(max_count, 1);
  b.state.ensure((size_t)S);
  b.h_state.ensure((size_t)S);
  b.cur.ensure((size_t)S * b.pitch);
  b.corr.ensure((size_t)S * b.pitch);
  b.qsearch.ensure((size_t)S * b.pitch);
  b.tgt_of_src.ensure((size_t)S);
  b.guesses.ensure((size_t)S * 16);
  b.partials.ensure((size_t)S * b.G * kIcpPartStride);
  b.n_active.ensure(1);
  b.h_active.ensure(1);
  // "never searched": correspondent -1, NaN margin (all-ones bytes are both)
  MI_HIP(hipMemsetAsync(b.corr.p, 0xFF, (size_t)S * b.pitch * sizeof(int), s));
  MI_HIP(hipMemsetAsync(b.qsearch.p, 0xFF, (size_t)S * b.pitch * sizeof(float4), s));
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
      hipLaunchKernelGGL(k_icp_move, grid, dim3(kIcpBlock), 0, s, d_src, spitch, b.cur.p, b.pitch, (const IcpPairState*)b.state.p, b.G, S);
      // the pairs that still move by centimetres: wave-cooperative tiles (gicp.hip); the others live on the margins their searches left.
      // r1 = two base cells of the target's voxel hash: a sub-map voxelised at 0.5 m has its nearest point within that of nearly every query.
      // The tiles keep d2 < max^2 where PCL keeps d2 <= max^2: the next float up makes the two the same test.
      if (ICP_TILES && S >= 8 && it < kIcpTileIterations)
        icp_correspond_tiles(s, b.cur.p, b.pitch, (const IcpPairState*)b.state.p, tgt, (const int*)b.tgt_of_src.p, b.corr.p, b.qsearch.p,
                             nextafterf(max_d2, INFINITY), tuning().icp_r1_cells * tuning().grid_cell, S, tuning().icp_min_move, src_tile_order, spitch,
                             tuning().icp_rmid_cells * tuning().grid_cell, tuning().icp_r1_sigma);
      TimedLaunch tl(KF_CORRESPOND, s);
      hipLaunchKernelGGL(k_icp_step, grid, dim3(kIcpBlock), 0, s, (const float4*)b.cur.p, b.pitch, b.corr.p, b.qsearch.p, (const IcpPairState*)b.state.p, (const float4*)tgt.pts.p,
                         (const int*)tgt.orig.p, (const HashSlot*)tgt.table.p, (const GridMeta*)tgt.meta.p, tgt.cap, tgt.slots_cap,
                         (const int*)b.tgt_of_src.p, max_d2, b.G, S, b.partials.p, tuning().icp_ball_scale, tuning().icp_ball_pad);
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
