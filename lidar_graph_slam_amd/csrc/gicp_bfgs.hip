// pclomp::GeneralizedIterativeClosestPoint ("GICP": lidar_scan_matcher.cpp:73-96, graph_based_slam.cpp:120-141) on gfx950 —
// SURVEY.md 8f-4.  PCL 1.8's GICP: per OUTER iteration one correspondence + Mahalanobis pass over the source, then a BFGS
// minimisation (bfgs_dev.hpp) of f(x) = 1/m sum r^T M r over the six pose parameters, every evaluation of which is one
// streaming pass over the matched pairs:
//   k_bfgs_correspond   q = transformation_ * (guess * p) in fp32; exact 1-NN in the target's voxel hash, kept iff
//                       d2 < max_dist^2; M = (R C_src R^T + C_tgt)^-1 in fp64 (R = rotation of transformation_ * guess);
//                       64 algorithmic bytes per source point (point + covariance of source and correspondent);
//   k_bfgs_cost         r = T(x) p - q_tgt (fp32 transform, fp64 residual), 14 fp64 sums: m, f, the translation gradient
//                       and the 3x3 accumulator the rotation gradient is contracted from; 88 B per matched pair (source point,
//                       correspondence, target point, 48-byte fp64 Mahalanobis matrix);
//   k_bfgs_advance      one thread per pair: the pair's block partials added in block order (bitwise reproducible), value and
//                       gradient from the sums, the line search / BFGS update / outer-iteration bookkeeping up to the next point
//                       the cost is needed at.
// BATCHED (round 5): S pairs against the targets tgt_of_src names, every pair's minimiser state in HBM.  One ROUND = the three
// kernels over all pairs still iterating: pairs that start an outer iteration take the correspondence pass (the others' blocks
// leave at once), every pair takes one cost evaluation at the point ITS search asks for, then all searches advance.  Nothing is
// read back between rounds; the host polls the number of unfinished pairs every few rounds.  A registration object is a batch
// of one (rounds 1-4: PCL's functor-calling class on the host, a launch + a 112-byte read-back + a stream synchronisation per
// evaluation).  The covariances are the PLANE-regularised ones of gicp.hip (singular values (1, 1, 1e-3) = PCL's gicp_epsilon).
#include <cmath>
#include <cstring>

#include "bfgs_dev.hpp"
#include "engine.hpp"
#include "grid.hpp"
#include "primitives.hpp"

#ifndef BFGS_TILES
#define BFGS_TILES 1  // 0 (A/B): per-lane searches only
#endif
#ifndef BFGS_SEEDED
#define BFGS_SEEDED 1  // 0: every outer iteration searches from scratch (A/B: benchmarks/build_variant.sh)
#endif
namespace mi355 {

constexpr int kBfgsBlock = 256;
constexpr int kBfgsPartStride = 16;

struct BfgsConsts {
  int max_iterations, max_optimizer_iterations;
  double rotation_epsilon, transformation_epsilon;
};

struct BfgsPairState {
  float guess[16];   // base_transformation_
  float trans[16];   // transformation_: what the current outer iteration starts from
  float prev[16];    // previous_transformation_
  float T_eval[16];  // guess with applyState(req_x): what the next cost pass transforms the source by
  double R[9];       // rotation of double(transformation_) * double(guess): carries the source covariances of this outer iteration
  BfgsMin m;
  int active, converged, nr, need_corr;  // need_corr: the next round starts an outer iteration (correspondences + Mahalanobis matrices)
  int n, n_target, n_inliers, evaluations, searched, pad;  // searched: the pair has correspondences of an earlier iteration (the seeds)
  double last_f;
};

struct BfgsBatch {
  int S = 0, G = 1;
  size_t pitch = 0;
  DevBuf<BfgsPairState> state;
  DevBuf<int> corr;          // [S][pitch] correspondent (sorted target index) of every sorted source point, -1: none in range
  DevBuf<double> mahal;      // [S][pitch][6] fp64 Mahalanobis matrices of the current outer iteration
  DevBuf<float4> cur;        // [S][pitch] batches: transformation_ * (guess * p) of the pairs that start an outer iteration - what the tiles search at
  DevBuf<float4> qsearch;    // [S][pitch] batches: where a point's query stood at its last search (xyz), how far it may move before the answer can change (w)
  DevBuf<int> jobs;          // [S] batches: the point count of a pair that starts an outer iteration in this round, else 0
  DevBuf<double> partials;   // [S][G][kBfgsPartStride]
  DevBuf<int> tgt_of_src, n_active;
  DevBuf<float> guesses;
  PinnedBuf<int> h_active;
  PinnedBuf<BfgsPairState> h_state;
  std::vector<int> h_map;
  std::vector<float> h_guess;
  int rounds = 0;
};

BfgsBatch* bfgs_batch_create() { return new BfgsBatch(); }
void bfgs_batch_destroy(BfgsBatch* b) { delete b; }
const int* bfgs_batch_corr(const BfgsBatch& b, size_t* pitch) { *pitch = b.pitch; return b.corr.p; }

// Eigen's Matrix4f * Vector4f with w = 1, evaluated left to right (no fma: the oracle spells the same expression)
__device__ __forceinline__ float row_apply(const float* r, float x, float y, float z) { return r[0] * x + r[1] * y + r[2] * z + r[3]; }

// an outer iteration is scheduled: its correspondences carry R = rot(double(trans) * double(guess)), its minimisation starts at the
// parameters of trans (estimateRigidTransformationBFGS)
__device__ __forceinline__ void bfgs_schedule_outer(BfgsPairState& st) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double v = 0;
      for (int k = 0; k < 4; k++) v += (double)st.trans[4 * i + k] * (double)st.guess[4 * k + j];
      st.R[3 * i + j] = v;
    }
  for (int k = 0; k < 16; k++) st.prev[k] = st.trans[k];
  double x[6];
  bfgs_x_from_matrix(st.trans, x);
  bfgs_begin(st.m, x);
  for (int k = 0; k < 16; k++) st.T_eval[k] = st.guess[k];
  bfgs_apply_state(st.T_eval, st.m.req_x);
  st.need_corr = 1;
}

__global__ void k_bfgs_begin(BfgsPairState* __restrict__ state, const float* __restrict__ guesses, const int* __restrict__ scount,
                             const GridMeta* __restrict__ tmeta, const int* __restrict__ tgt_of_src, int S) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  BfgsPairState& st = state[p];
  for (int k = 0; k < 16; k++) {
    st.guess[k] = guesses[16 * p + k];
    st.trans[k] = (k % 5 == 0) ? 1.f : 0.f;
  }
  st.n = scount[p];
  st.n_target = tmeta[tgt_of_src[p]].n;
  st.converged = 0; st.nr = 0; st.n_inliers = 0; st.evaluations = 0; st.searched = 0; st.pad = 0;
  st.last_f = 0.0;
  st.active = st.n > 0 ? 1 : 0;
  bfgs_schedule_outer(st);
}

// Batches (round 6): the pairs that start an outer iteration put their moved sources where the wave-cooperative tiles of gicp.hip search
// them - q = transformation_ * (guess * p), the two fp32 products upstream makes - and say so in jobs[].
__global__ __launch_bounds__(kBfgsBlock) void k_bfgs_move(const float4* __restrict__ spts, size_t scap, const BfgsPairState* __restrict__ state,
                                                          float4* __restrict__ cur, size_t pitch, int* __restrict__ jobs, int G, int S) {
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const BfgsPairState& st = state[pair];
  const bool now = st.active && st.need_corr;
  if (bx == 0 && threadIdx.x == 0) jobs[pair] = now ? st.n : 0;
  if (!now) return;
  float guess[12], trans[12];
#pragma unroll
  for (int k = 0; k < 12; k++) { guess[k] = st.guess[k]; trans[k] = st.trans[k]; }
  const float4* sp = spts + (size_t)pair * scap;
  float4* out = cur + (size_t)pair * pitch;
  for (int i = bx * kBfgsBlock + threadIdx.x; i < st.n; i += G * kBfgsBlock) {
    const float4 p = sp[i];
    const float gx = row_apply(guess, p.x, p.y, p.z), gy = row_apply(guess + 4, p.x, p.y, p.z), gz = row_apply(guess + 8, p.x, p.y, p.z);
    out[i] = make_float4(row_apply(trans, gx, gy, gz), row_apply(trans + 4, gx, gy, gz), row_apply(trans + 8, gx, gy, gz), p.w);
  }
}

// The correspondences and Mahalanobis matrices of the pairs that start an outer iteration.  grid = xcd_grid(G, S): the blocks of a pair
// share an XCD (its target's points and hash table in ONE L2).  From the second outer iteration on the search is seeded with the
// previous correspondent: its distance bounds the walk through the voxel hash from the start (the same exact nearest neighbour under
// the same tie rule: icp.hip).
__global__ __launch_bounds__(kBfgsBlock) void k_bfgs_correspond(const float4* __restrict__ spts, const Cov6* __restrict__ scov, size_t scap,
                                                                const BfgsPairState* __restrict__ state, const float4* __restrict__ tpts,
                                                                const int* __restrict__ torig, const Cov6* __restrict__ tcov,
                                                                const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta,
                                                                size_t tcap, size_t tslots, const int* __restrict__ tgt_of_src, float max_d2, int G,
                                                                int S, int* __restrict__ corr, double* __restrict__ mahal, size_t pitch,
                                                                const float4* __restrict__ qsearch) {
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const BfgsPairState& st = state[pair];
  if (!st.active || !st.need_corr) return;
  const int n = st.n;
  const int t = tgt_of_src[pair];
  const GridView g = grid_view(tpts, torig, ttable, tmeta, t, tcap, tslots);
  const Cov6* tc = tcov + (size_t)t * tcap;
  float guess[12], trans[12];
  double R[9];
#pragma unroll
  for (int k = 0; k < 12; k++) { guess[k] = st.guess[k]; trans[k] = st.trans[k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = st.R[k];
  const bool seeded = BFGS_SEEDED && st.searched != 0;
  const float4* sp = spts + (size_t)pair * scap;
  const Cov6* sc = scov + (size_t)pair * scap;
  int* cr = corr + (size_t)pair * pitch;
  double* mh = mahal + (size_t)pair * pitch * 6;
  for (int i = bx * kBfgsBlock + threadIdx.x; i < n; i += G * kBfgsBlock) {
    const float4 p = sp[i];
    const float gx = row_apply(guess, p.x, p.y, p.z), gy = row_apply(guess + 4, p.x, p.y, p.z), gz = row_apply(guess + 8, p.x, p.y, p.z);
    const float qx = row_apply(trans, gx, gy, gz), qy = row_apply(trans + 4, gx, gy, gz), qz = row_apply(trans + 8, gx, gy, gz);
    Nearest1 nn{g.orig, INFINITY, -1};
    float limit = max_d2;
    // (batches: the tiles have been here - a margin that is not NaN says the correspondent in cr[i], or its absence, is this pose's exact answer;
    // what they left undecided - near-ties of their keys, boxes they could not scan - is searched here, from scratch)
    const bool decided = qsearch != nullptr && qsearch[(size_t)pair * pitch + i].w == qsearch[(size_t)pair * pitch + i].w;
    if (decided) {
      nn.best_j = cr[i];
      nn.best_d2 = nn.best_j >= 0 ? 0.f : INFINITY;  // (within range by the tiles' own test, the same d2 < max_dist^2)
    } else if (seeded && qsearch == nullptr) {
      const int j0 = cr[i];
      if (j0 >= 0) {
        const float4 q0 = g.pts[j0];
        const float d0 = dist2f(qx, qy, qz, q0.x, q0.y, q0.z);
        if (d0 < max_d2) { nn.best_d2 = d0; nn.best_j = j0; limit = nextafterf(d0, INFINITY); }
      }
    }
    if (!decided && isfinite(qx) && isfinite(qy) && isfinite(qz)) grid_search(g, qx, qy, qz, limit, nn, 0, 2);
    int j = -1;
    if (nn.best_j >= 0 && nn.best_d2 < max_d2) {
      j = nn.best_j;
      const Sym3<float> a = cov6_unpack(sc[i]), b = cov6_unpack(tc[j]);
      const Sym3<double> A = {(double)a.xx, (double)a.xy, (double)a.xz, (double)a.yy, (double)a.yz, (double)a.zz};
      const Sym3<double> RA = sym3_rotate(R, A);
      const Sym3<double> Sm = {RA.xx + (double)b.xx, RA.xy + (double)b.xy, RA.xz + (double)b.xz, RA.yy + (double)b.yy, RA.yz + (double)b.yz,
                               RA.zz + (double)b.zz};
      const Sym3<double> M = sym3_inverse(Sm);
      double* o = mh + 6 * (size_t)i;
      o[0] = M.xx; o[1] = M.xy; o[2] = M.xz; o[3] = M.yy; o[4] = M.yz; o[5] = M.zz;
    }
    cr[i] = j;
  }
}

// sums: [0] m, [1] sum r^T M r, [2..4] sum M r, [5..13] sum (base p) (M r)^T row-major, at the point the pair's search asks for
__global__ __launch_bounds__(kBfgsBlock) void k_bfgs_cost(const float4* __restrict__ spts, size_t scap, const BfgsPairState* __restrict__ state,
                                                          const float4* __restrict__ tpts, size_t tcap, const int* __restrict__ tgt_of_src,
                                                          const int* __restrict__ corr, const double* __restrict__ mahal, size_t pitch, int G, int S,
                                                          double* __restrict__ partials) {
  __shared__ double smem[4 * kBfgsAcc];
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const BfgsPairState& st = state[pair];
  if (!st.active) return;
  const int n = st.n;
  float T[12], base[12];
#pragma unroll
  for (int k = 0; k < 12; k++) { T[k] = st.T_eval[k]; base[k] = st.guess[k]; }
  const float4* sp = spts + (size_t)pair * scap;
  const float4* tp = tpts + (size_t)tgt_of_src[pair] * tcap;
  const int* cr = corr + (size_t)pair * pitch;
  const double* mh = mahal + (size_t)pair * pitch * 6;
  double acc[kBfgsAcc];
#pragma unroll
  for (int k = 0; k < kBfgsAcc; k++) acc[k] = 0.0;
  for (int i = bx * kBfgsBlock + threadIdx.x; i < n; i += G * kBfgsBlock) {
    const int j = cr[i];
    if (j < 0) continue;
    const float4 p = sp[i], q = tp[j];
    const float px = row_apply(T, p.x, p.y, p.z), py = row_apply(T + 4, p.x, p.y, p.z), pz = row_apply(T + 8, p.x, p.y, p.z);
    const double r0 = (double)(px - q.x), r1 = (double)(py - q.y), r2 = (double)(pz - q.z);
    const double* M = mh + 6 * (size_t)i;
    const double t0 = M[0] * r0 + M[1] * r1 + M[2] * r2, t1 = M[1] * r0 + M[3] * r1 + M[4] * r2, t2 = M[2] * r0 + M[4] * r1 + M[5] * r2;
    acc[0] += 1.0;
    acc[1] += r0 * t0 + r1 * t1 + r2 * t2;
    acc[2] += t0; acc[3] += t1; acc[4] += t2;
    // base_transformation_ * p_src, translation included (upstream's: gicp.hpp, OptimizationFunctorWithIndices::df)
    const double b0 = (double)row_apply(base, p.x, p.y, p.z), b1 = (double)row_apply(base + 4, p.x, p.y, p.z),
                 b2 = (double)row_apply(base + 8, p.x, p.y, p.z);
    acc[5] += b0 * t0; acc[6] += b0 * t1; acc[7] += b0 * t2;
    acc[8] += b1 * t0; acc[9] += b1 * t1; acc[10] += b1 * t2;
    acc[11] += b2 * t0; acc[12] += b2 * t1; acc[13] += b2 * t2;
  }
  block_sum<kBfgsAcc, 4>(acc, smem);
  if (threadIdx.x == 0) {
    double* out = partials + ((size_t)pair * G + bx) * kBfgsPartStride;
#pragma unroll
    for (int k = 0; k < kBfgsAcc; k++) out[k] = acc[k];
  }
}

// One thread per pair: GeneralizedIterativeClosestPoint::computeTransformation's bookkeeping around the minimiser.
__global__ __launch_bounds__(64) void k_bfgs_advance(BfgsPairState* __restrict__ state, const double* __restrict__ partials, BfgsConsts c, int G, int S,
                                                     int* __restrict__ n_active) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= S) return;
  BfgsPairState& st = state[pair];
  if (!st.active) return;
  double h[kBfgsAcc];
#pragma unroll
  for (int k = 0; k < kBfgsAcc; k++) h[k] = 0.0;
  for (int b = 0; b < G; b++) {
    const double* row = partials + ((size_t)pair * G + b) * kBfgsPartStride;
#pragma unroll
    for (int k = 0; k < kBfgsAcc; k++) h[k] += row[k];
  }
  st.evaluations++;
  st.n_inliers = (int)h[0];
  st.searched = 1;
  if (st.need_corr && h[0] < 4.0) {
    // "Need at least 4 points to estimate a transform": computeTransformation's loop is left, not converged (the first evaluation of
    // an outer iteration also tells how many correspondences there are)
    st.active = 0;
    return;
  }
  st.need_corr = 0;
  double f, g[6];
  bfgs_cost_from_sums(h, st.m.req_x, &f, g);
  const BfgsParams P;
  if (bfgs_feed(st.m, f, g, c.max_optimizer_iterations, P)) {  // the search wants the cost at another point: the next round evaluates it
#pragma unroll
    for (int k = 0; k < 16; k++) st.T_eval[k] = st.guess[k];
    bfgs_apply_state(st.T_eval, st.m.req_x);
  } else {
    // the inner loop is over: estimateRigidTransformationBFGS accepts NoProgress, Success and "out of optimizer iterations"
    const int result = st.m.result;
    if (!(result == BFGS_NOPROGRESS || result == BFGS_SUCCESS || st.m.inner == c.max_optimizer_iterations)) {
      st.active = 0;  // (cannot happen with this minimiser: kept as upstream's exit)
    } else {
      st.last_f = st.m.f;
#pragma unroll
      for (int k = 0; k < 16; k++) st.trans[k] = (k % 5 == 0) ? 1.f : 0.f;
      bfgs_apply_state(st.trans, st.m.x);
      double delta = 0;
      for (int k = 0; k < 4; k++)
        for (int l = 0; l < 4; l++) {
          const double ratio = (k < 3 && l < 3) ? 1.0 / c.rotation_epsilon : 1.0 / c.transformation_epsilon;
          delta = fmax(delta, ratio * fabs((double)st.prev[4 * k + l] - (double)st.trans[4 * k + l]));
        }
      st.nr++;
      if (st.nr >= c.max_iterations || delta < 1) {
        st.converged = 1;
#pragma unroll
        for (int k = 0; k < 16; k++) st.prev[k] = st.trans[k];
        st.active = 0;
      } else {
        bfgs_schedule_outer(st);
      }
    }
  }
  if (n_active && st.active) atomicAdd(n_active, 1);
}

// src / tgt: built sets with PLANE covariances (the sources' points and covariances in their sorted order); pair p -> target
// tgt_of_src_host[p] (nullptr: p); no source holds more than max_count points.  Blocks until every pair is done; the final states stay
// in the batch (bfgs_batch_results).
void bfgs_batch_align(hipStream_t s, BfgsBatch& b, int S, const CloudSet& src, const CloudSet& tgt, size_t max_count, const int* tgt_of_src_host,
                      const float* guesses_host, const mi355_gicp_bfgs_params& p) {
  if (!src.has_cov || !tgt.has_cov) throw ArgError(MI355_ESTATE, "covariances missing");
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
  b.G = std::max(1, std::min(Gcap, div_up((long)std::max<size_t>(max_count, 1), kBfgsBlock * 4)));
  b.pitch = std::max<size_t>(std::min(max_count, src.cap), 1);
  b.state.ensure((size_t)S);
  b.h_state.ensure((size_t)S);
  b.corr.ensure((size_t)S * b.pitch);
  b.mahal.ensure((size_t)S * b.pitch * 6);
  // A batch searches by tiles first (a handful of pairs cannot fill the chip with them: icp.hip); its sources follow their tile order.
  const bool tiles = BFGS_TILES && S >= 8;
  if (tiles) {
    b.cur.ensure((size_t)S * b.pitch);
    b.qsearch.ensure((size_t)S * b.pitch);
    b.jobs.ensure((size_t)S);
    // "never searched": correspondent -1, NaN margin (all-ones bytes are both)
    MI_HIP(hipMemsetAsync(b.corr.p, 0xFF, (size_t)S * b.pitch * sizeof(int), s));
    MI_HIP(hipMemsetAsync(b.qsearch.p, 0xFF, (size_t)S * b.pitch * sizeof(float4), s));
  }
  b.tgt_of_src.ensure((size_t)S);
  b.guesses.ensure((size_t)S * 16);
  b.partials.ensure((size_t)S * b.G * kBfgsPartStride);
  b.n_active.ensure(1);
  b.h_active.ensure(1);
  MI_HIP(hipMemcpyAsync(b.tgt_of_src.p, b.h_map.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(b.guesses.p, b.h_guess.data(), (size_t)S * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_bfgs_begin, dim3(div_up(S, 64)), dim3(64), 0, s, b.state.p, (const float*)b.guesses.p, (const int*)src.count.p,
                     (const GridMeta*)tgt.meta.p, (const int*)b.tgt_of_src.p, S);
  const float max_d2 = (float)(p.max_correspondence_distance * p.max_correspondence_distance);
  BfgsConsts c;
  c.max_iterations = p.max_iterations;
  c.max_optimizer_iterations = p.max_optimizer_iterations;
  c.rotation_epsilon = p.rotation_epsilon;
  c.transformation_epsilon = p.transformation_epsilon;
  const int chunk = 16;
  const dim3 grid(xcd_grid(b.G, S));
  b.rounds = 0;
  // every round is one evaluation per unfinished pair: at most max_iterations x (1 + max_optimizer_iterations x the line search's 200
  // trials) of them - the poll ends the loop long before
  const long max_rounds = (long)std::max(p.max_iterations, 1) * (1 + (long)std::max(p.max_optimizer_iterations, 1) * 201);
  for (long round = 0; round < max_rounds;) {
    MI_HIP(hipMemsetAsync(b.n_active.p, 0, sizeof(int), s));
    for (int k = 0; k < chunk && round < max_rounds; k++, round++) {
      if (tiles) {
        hipLaunchKernelGGL(k_bfgs_move, grid, dim3(kBfgsBlock), 0, s, (const float4*)src.pts.p, src.cap, (const BfgsPairState*)b.state.p, b.cur.p, b.pitch,
                           b.jobs.p, b.G, S);
        // first radius two base cells of the targets' voxel hash, a middle round at six when the range reaches that far (gicp.hip)
        moved_correspond_tiles(s, b.cur.p, b.pitch, (const int*)b.jobs.p, tgt, (const int*)b.tgt_of_src.p, b.corr.p, b.qsearch.p, max_d2,
                               2.f * tuning().grid_cell, 6.f * tuning().grid_cell, S,
                               (src.has_order && src.cap >= b.pitch) ? (const int*)src.tile_order.p : (const int*)nullptr, src.cap);
      }
      {
        TimedLaunch tl(KF_CORRESPOND, s);
        hipLaunchKernelGGL(k_bfgs_correspond, grid, dim3(kBfgsBlock), 0, s, (const float4*)src.pts.p, (const Cov6*)src.cov.p, src.cap,
                           (const BfgsPairState*)b.state.p, (const float4*)tgt.pts.p, (const int*)tgt.orig.p, (const Cov6*)tgt.cov.p,
                           (const HashSlot*)tgt.table.p, (const GridMeta*)tgt.meta.p, tgt.cap, tgt.slots_cap, (const int*)b.tgt_of_src.p, max_d2, b.G, S,
                           b.corr.p, b.mahal.p, b.pitch, tiles ? (const float4*)b.qsearch.p : (const float4*)nullptr);
      }
      hipLaunchKernelGGL(k_bfgs_cost, grid, dim3(kBfgsBlock), 0, s, (const float4*)src.pts.p, src.cap, (const BfgsPairState*)b.state.p,
                         (const float4*)tgt.pts.p, tgt.cap, (const int*)b.tgt_of_src.p, (const int*)b.corr.p, (const double*)b.mahal.p, b.pitch, b.G, S,
                         b.partials.p);
      hipLaunchKernelGGL(k_bfgs_advance, dim3(div_up(S, 64)), dim3(64), 0, s, b.state.p, (const double*)b.partials.p, c, b.G, S,
                         k == chunk - 1 || round + 1 >= max_rounds ? b.n_active.p : (int*)nullptr);
    }
    MI_LAUNCH_CHECK();
    b.rounds = (int)round;
    MI_HIP(hipMemcpyAsync(b.h_active.p, b.n_active.p, sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (*b.h_active.p == 0) break;
  }
  MI_HIP(hipMemcpyAsync(b.h_state.p, b.state.p, (size_t)S * sizeof(BfgsPairState), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void bfgs_batch_results(BfgsBatch& b, mi355_result* out, float* T16_all) {
  for (int i = 0; i < b.S; i++) {
    const BfgsPairState& st = b.h_state.p[i];
    // final_transformation_ = [R_prev R_guess | t_prev + t_guess]
    float T[16];
    memset(T, 0, sizeof(T));
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) T[4 * r + c] = st.prev[4 * r] * st.guess[c] + st.prev[4 * r + 1] * st.guess[4 + c] + st.prev[4 * r + 2] * st.guess[8 + c];
      T[4 * r + 3] = st.prev[4 * r + 3] + st.guess[4 * r + 3];
    }
    T[15] = 1.f;
    if (out) {
      mi355_result& o = out[i];
      memset(&o, 0, sizeof(o));
      memcpy(o.T, T, sizeof(T));
      o.converged = st.converged;
      o.iterations = st.nr;
      o.error = st.last_f;
      o.n_source = st.n;
      o.n_target = st.n_target;
      o.n_inliers = st.n_inliers;
      o.evaluations = st.evaluations;
    }
    if (T16_all) memcpy(T16_all + (size_t)i * 16, T, sizeof(T));
  }
}

// pclomp::GeneralizedIterativeClosestPoint::align of ONE pair: a batch of one.
void gicp_bfgs_align(hipStream_t s, BfgsScratch& sc, const CloudSet& src, const CloudSet& tgt, size_t n_src, const mi355_gicp_bfgs_params& p,
                     const float* guess16, mi355_result* out) {
  if (!sc.batch) sc.batch = bfgs_batch_create();
  bfgs_batch_align(s, *sc.batch, 1, src, tgt, n_src, nullptr, guess16, p);
  bfgs_batch_results(*sc.batch, out, nullptr);
}
BfgsScratch::~BfgsScratch() {
  if (batch) bfgs_batch_destroy(batch);
}

}  // namespace mi355
