// GICP on gfx950: the fast_gicp::FastGICP optimisation loop as a device-resident,
// batched Levenberg-Marquardt.  One launch covers every scan pair of the batch (blockIdx.y = pair).
//
// Per outer iteration (LsqRegistration::step_lm):
//   k_gicp_correspond(_tile)  F3: fp32 transform -> exact 1-NN in the target voxel-hash, incrementally (search margins).
//   k_gicp_linearize  F4: RCR = C_B + R C_A R^T -> Mahalanobis = RCR^-1 -> residual (fp64 transform) -> 21+6+1 normal-equation terms,
//                     accumulated in fp64 registers, reduced by wave shuffles + LDS to ONE partial per block (fixed order =>
//                     bitwise reproducible) + the 72 sums of the trial-cost model on the matrix cores (F5, see kModelSlot) + the
//                     search-margin bits of the next k_gicp_correspond.  Writes the Mahalanobis matrix (24 B/pt).
//   k_gicp_step       per pair: sums the partials, lambda init, LDL^T solve of (H + lambda I) d = -b for every lambda of step_lm's
//                     sequence, se3_exp, trial poses, their costs (F5 compute_error, through the model), rho test, accept / reject,
//                     lambda update, convergence test, iteration bookkeeping — no host round trip, no pass over the points.
// Algorithmic HBM bytes of k_gicp_linearize per source point: src pt 16 + src cov 24 + tgt pt 16 + tgt cov 24 = 80 B.
#include "engine.hpp"
#include "icp_fit.hpp"
#include "grid.hpp"
#include "primitives.hpp"
#include "tile.hpp"

namespace mi355 {
constexpr int kCorrStatsN = 8 + 4 * 64;  // MI355_CORR_STATS: 8 counters of the tile kernel + per iteration (searched, close seeds, by the whole wave, tracking pairs)

struct GicpPairState {
  Pose x0, xi, delta;
  double H[36], b[6], d[6];
  double y0, yi, lambda, nu;
  double final_H[36];
  double last_error;
  int iter, trial, active, converged, n_inliers, lm_failed, status;
  double last_step;  // size of the last accepted update (metres at 50 m lever arm): picks the search mode of the next iteration
  int first_conv;    // iterations run when is_converged first held (-1: never) - what fast_gicp runs without force_iterations
  int mask_valid;    // the "safe" bits of this pair (GicpBatch::safe_mask) hold for the pose the next k_gicp_correspond will see
  float slack_rot, slack_trans;  // the budget of the current iteration: its search-margin bits hold while no trial pose moves a point at distance r
                                 // from the origin by more than slack_rot * r + slack_trans (k_gicp_step sets and checks it; < 0: none yet)
};

// One Levenberg-Marquardt trial step.  step_lm's trial sequence is fully determined by (H, b, lambda): trial t uses
// lambda_t = lambda * 2 * 4 * ... * 2^t, so ALL trial poses are computed up front (one lane each), their costs come from the sums
// of the linearize pass (kModelSlot), and the decision walks the trials in order.
// Same result as the sequential loop, but no data-dependent number of passes over the points - none at all.
struct LmTrial {
  Pose xi, delta;
  double d[6];
  double lambda;
};

struct GicpDevParams {
  int max_iterations, optimizer, lm_max_iterations, force_iterations;
  double transformation_epsilon, rotation_epsilon, lm_init_lambda_factor;
  float max_corr_d2;
};

constexpr int kNLin = 8;        // sum M (6: H_tt) + cost + inlier count
constexpr int kSumCost = 6, kSumCount = 7;
constexpr int kNAcc = kNLin + 12;  // + the 12 sums of the model below that are kept in registers
constexpr int kPartStride = 96;  // doubles per (pair, block): 0..19 the register sums, 32..95 the 8 x 8 tile of the matrix cores
constexpr int kModelSlot = 32;
// The cost of an LM trial without a pass over the points.  With the correspondences and Mahalanobis matrices of the iteration frozen
// (FastGICP::compute_error), the cost at the trial pose x_t = delta_t x0 is a quadratic form of W = [dR - I | dt] (3 x 4):
//   d_t = q - x_t p = d0 - W a~        a~ = (x0 p, 1), d0 = q - x0 p (the residual linearize already has)
//   e_t = sum d_t^T M d_t = y0 - 2 <W, L> + sum_{r,s,c,e} W_rc W_se Q[rs][ce]
//   L[r][c] = sum (M d0)_r a~_c   (12 sums)        Q[rs][ce] = sum M_rs a~_c a~_e   (6 x 10 sums)
// - 72 sums over the points that do not depend on the trial.  k_gicp_linearize accumulates them in the pass it makes anyway, and
// k_gicp_step prices all the trials of step_lm from them: the cost pass (56 B per point and iteration, fp64-issue-bound on ten
// trials) is gone.  It is the same function of the pose as the reference's loop, evaluated as (decrease from y0), which is better
// conditioned than the difference of two sums the reference forms for its rho test.
// The normal equations are linear in the same sums - J = [skew(a) | -I] (fast_gicp_impl.hpp), H = sum J^T M J, b = sum J^T M d0 -
// so k_gicp_step forms H and b from them too (model_normal_equations) and the linearize pass carries no accumulator of its own for
// them: 20 fp64 sums per lane + the matrix-core tile instead of round 3's 29 (fp32 products) + round 4's 41.
// Where the 72 sums live: as the 9 x 10 matrix "model" (rows M_xx M_xy M_xz M_yy M_yz M_zz (M d0)_x (M d0)_y (M d0)_z, columns ax ax,
// ax ay, ax az, ay ay, ay az, az az, ax, ay, az, 1; a = x0 p) they are
//   - an 8 x 8 outer product per point on the matrix cores: rows M (6), (M d0)_x, (M d0)_y; columns ax ax, ax ay, ax az, ay ay, ay az,
//     ax, ay, az (one v_mfma_f64_16x16x4_f64 takes EIGHT points: four in its rows / columns 0-7, four in 8-15; the off-diagonal
//     blocks mix points and are ignored),
//   - 12 more fp64 accumulators per lane: M x az az (6), sum M d0 (3), (M d0)_z x (ax, ay, az) (3),
//   - and sum M itself, which the normal equations already hold (H_tt).
constexpr int kMfmaDim = 8, kNMfma = kMfmaDim * kMfmaDim;
constexpr int kModelRows = 9, kModelCols = 10, kNModel = kModelRows * kModelCols;
constexpr int kFeatStride = 9;  // LDS row stride of a point's 8 factors (doubles; odd: the lanes of a wave write whole rows)
constexpr int kGicpBlock = 256;
constexpr int kMaxTrials = 16;  // lm_max_iterations is capped here (fast_gicp default 10)

void gicp_alloc(GicpBatch& gb, int S_own, size_t cap, int compute_units, int S_geometry) {
  const Tuning& tn = tuning();
  gb.ppt = tn.corr_ppt == 1 ? 1 : 4;
  gb.S = S_own;
  gb.S_geometry = S_geometry > 0 ? S_geometry : S_own;
  gb.cap = cap;
  const int Sg = gb.S_geometry;  // the launch geometry below is that of the whole batch this object is (a part of)
  const int S = S_own;
  // One pair is worked on by 2 - 64 blocks of 256 threads that all land on one XCD (primitives.hpp).  In steady state a block's life is a chain of dependent loads (pair state, points, stored queries) around very little
  // work, so a big batch wants fewer blocks that each walk several tiles — measured at 1 024 pairs: 64 blocks per pair 74.5 ms per
  // step (k_gicp_correspond 12.9, the tile kernel 8.2), 16: 72.1 ms (12.2 / 7.4), 8: 72.4, 4: 73.5; at 128 pairs 64 is best
  // (12.7 ms per step against 13.2 at 16).  Round 3, with the per-lane kernel searching its queue a block's worth at a time (a block
  // that walks several tiles fills its queue) - about 8 k blocks over the batch: 1 024 pairs 16 / 8 / 6 blocks per pair 60.6 / 59.9 /
  // 59.6 ms per step, 2 048 pairs 16 / 8 / 4 / 2: 114.1 / 112.8 / 112.0 / 112.0 ms, 4 096 pairs 16 / 4 / 2: 222.4 / 215.2 / 214.8 ms
  // (benchmarks/bpp_sweep.sh).
  (void)compute_units;
  int G = tn.blocks_per_pair > 0 ? tn.blocks_per_pair : std::max(2, std::min(64, 8192 / std::max(Sg, 1)));
  G = std::max(1, std::min(G, div_up((long)cap, kGicpBlock * gb.ppt)));
  // (k_gicp_correspond alone, round 4, with the margin bits - MI355_CORR_BLOCKS, 2 048 pairs in two parts: 16 / 8 / 4 / 3 / 2 / 1 blocks per
  // pair 14.9 / 13.1 / 12.2 / 12.05 / 11.95 / 15.6 ms per step on 8 scenes; on the bench's 32 scenes 4 / 3 / 2: 10.8 / 11.1 / 11.5 ms and
  // the same step time - the default stays 8 192 / pairs)
  const int Gc = tn.corr_blocks > 0 ? tn.corr_blocks : G;
  gb.blocks_per_pair = std::max(1, std::min(Gc, div_up((long)cap, kGicpBlock * gb.ppt)));
  // (the tile search likes twice as many: 2 048 pairs 4 / 8 / 16 blocks per pair 16.0 / 15.7 / 15.9 ms per step, 4 096 pairs 2 / 6: 30.8 / 30.2)
  gb.blocks_tile = tn.blocks_per_pair > 0 ? G : std::max(1, std::min(std::max(4, std::min(64, 16384 / std::max(Sg, 1))), div_up((long)cap, kGicpBlock)));
  // the streaming kernels (linearize, error, fitness) want fewer, fatter blocks: 4 points per thread, fewer partials
  // few, long-lived blocks: each block streams several tiles (software-pipelined) and pays ONE 29-value fp64 reduction;
  // about 512 blocks over the batch for a few pairs; big batches 6 per pair (3 from 4 096 pairs on), never more than the tiles a cloud
  // can have.  (6 / 3 rather than the 4 of round 2: k_gicp_linearize holds 3 blocks per CU, 768 over the chip, and a launch of
  // pairs x 4 blocks ends in a part-filled last round - measured at 512 / 1 024 / 2 048 / 4 096 pairs, 4 against the value chosen:
  // 33.2 / 32.7, 59.2 / 58.8, 111.2 / 110.5, 214.6 / 212.7 ms per step, benchmarks/geometry_sweep.sh)
  const int per_pair = std::max(Sg <= 2048 ? 6 : 3, std::min(32, 512 / std::max(Sg, 1)));
  gb.blocks_lin = std::max(1, std::min(per_pair, div_up((long)cap, kGicpBlock * 2)));
  if (tn.blocks_lin > 0) gb.blocks_lin = tn.blocks_lin;
  gb.tgt_of_src.ensure(S);
  gb.corr.ensure((size_t)S * cap);
  gb.qsearch.ensure((size_t)S * cap);
  gb.safe_mask.ensure((size_t)S * (size_t)div_up((long)cap, 64));
  gb.partials.ensure((size_t)S * std::max(G, gb.blocks_lin) * kPartStride);
  gb.state.ensure(S);
  gb.n_active.ensure(GicpBatch::kPolls);
  gb.h_active.ensure(GicpBatch::kPolls);
  gb.fit_partials.ensure((size_t)S * std::max(G, gb.blocks_lin) * 2);
  gb.fitness.ensure((size_t)S * 2);
  if (tn.corr_stats) {
    gb.corr_stats.ensure(kCorrStatsN);
    MI_HIP(hipMemset(gb.corr_stats.p, 0, kCorrStatsN * sizeof(unsigned long long)));
  }
}

// What gicp_fitness alone needs (a batch that registers with another method still scores its pairs with it).
void gicp_alloc_fitness_only(GicpBatch& gb, int S, size_t cap) {
  gb.S = S;
  gb.S_geometry = S;
  gb.cap = cap;
  gb.blocks_per_pair = 1;
  gb.blocks_lin = std::max(1, std::min(std::max(4, std::min(32, 512 / std::max(S, 1))), div_up((long)cap, kGicpBlock * 2)));
  gb.tgt_of_src.ensure(S);
  gb.state.ensure(1);
  gb.fit_partials.ensure((size_t)S * gb.blocks_lin * 2);
  gb.fitness.ensure((size_t)S * 2);
}

void gicp_set_map(hipStream_t s, GicpBatch& gb, const int* tgt_of_src_host, int n_targets) {
  gb.h_map.resize(gb.S);
  for (int i = 0; i < gb.S; i++) {
    gb.h_map[i] = tgt_of_src_host ? tgt_of_src_host[i] : i;
    if (gb.h_map[i] < 0 || gb.h_map[i] >= n_targets) throw ArgError(MI355_EINVAL, "tgt_of_src out of range");
  }
  MI_HIP(hipMemcpyAsync(gb.tgt_of_src.p, gb.h_map.data(), gb.S * sizeof(int), hipMemcpyHostToDevice, s));
}

void gicp_fitness_values(hipStream_t s, GicpBatch& gb, double* fitness_out) {
  std::vector<double> fit((size_t)gb.S * 2);
  MI_HIP(hipMemcpyAsync(fit.data(), gb.fitness.p, fit.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  for (int p = 0; p < gb.S; p++) fitness_out[p] = fit[2 * p + 1] > 0 ? fit[2 * p] / fit[2 * p + 1] : 1.7976931348623157e308;
}

// scount / tcount (optional): a pair whose clouds hold fewer than k_min points after the prefilter is not registered —
// its covariances would be built from missing neighbours — and reports MI355_PAIR_TOO_FEW_POINTS instead.
__global__ void k_gicp_init(GicpPairState* st, const float* __restrict__ guesses, int S, int has_guess, const int* __restrict__ scount,
                            const int* __restrict__ tcount, const int* __restrict__ tgt_of_src, int k_min) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  GicpPairState s;
  s.x0 = has_guess ? pose_from_rowmajor4x4f(guesses + 16 * p) : pose_identity();  // Isometry3d(guess.cast<double>())
  s.xi = s.x0;
  s.delta = pose_identity();
  for (int i = 0; i < 36; i++) { s.H[i] = 0; s.final_H[i] = 0; }
  for (int i = 0; i < 6; i++) { s.b[i] = 0; s.d[i] = 0; }
  s.y0 = s.yi = 0;
  s.lambda = -1.0;
  s.nu = 2.0;
  s.last_error = 0;
  s.iter = 0; s.trial = 0; s.active = 1; s.converged = 0; s.n_inliers = 0; s.lm_failed = 0; s.status = MI355_PAIR_OK;
  if (scount && (scount[p] < k_min || tcount[tgt_of_src[p]] < k_min)) { s.active = 0; s.status = MI355_PAIR_TOO_FEW_POINTS; }
  s.last_step = 1.0e9;
  s.first_conv = -1;
  s.mask_valid = 0;
  s.slack_rot = s.slack_trans = -1.f;
  st[p] = s;
}

// No previous correspondence, NaN margin ("never searched"): for the LIVE points of every pair only — two memsets over the whole
// capacity moved 1.34 GB per align at 1 024 x 65 536 slots of which a quarter is live.
__global__ __launch_bounds__(256) void k_gicp_reset_search(const int* __restrict__ scount, int* __restrict__ corr, float4* __restrict__ qsearch, size_t scap) {
  const int pair = blockIdx.y;
  const int n = scount[pair];
  const size_t so = (size_t)pair * scap;
  const float nan = __uint_as_float(0xFFFFFFFFu);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    corr[so + i] = -1;
    qsearch[so + i] = make_float4(nan, nan, nan, nan);
  }
}

// ---- per-point pieces ----------------------------------------------------------------------------------------------

// e^T M e with the residual in fp64 and the stored fp32 Mahalanobis matrix: y0 (linearize) and yi (compute_error) must
// resolve cost decreases far below fp32 resolution or the LM rho test degenerates into a coin flip near convergence.
// fp64 multiply-adds are allowed to fuse here and in residual() (the library is otherwise built with -ffp-contract=off for the
// fp32 steps whose rounding the parity contract specifies): upstream is Eigen under -O3 -march=native, which fuses as it likes,
// so there is no contraction pattern to be faithful to, parity of the cost is a tolerance, and the all-trials cost pass is
// fp64-issue-bound: 83 -> 47 fp64 instructions per trial and pair of points.
__device__ __forceinline__ double cost64(double ex, double ey, double ez, float mxx, float mxy, float mxz, float myy, float myz, float mzz) {
#pragma clang fp contract(fast)
  const double a = (double)mxx * ex + (double)mxy * ey + (double)mxz * ez;
  const double b = (double)mxy * ex + (double)myy * ey + (double)myz * ez;
  const double c = (double)mxz * ex + (double)myz * ey + (double)mzz * ez;
  return ex * a + ey * b + ez * c;
}

__device__ __forceinline__ void residual64(const Pose& x, const float4& p, const float4& q, double& dax, double& day, double& daz, float& ax, float& ay,
                                           float& az, float& ex, float& ey, float& ez, double& dex, double& dey, double& dez) {
#pragma clang fp contract(fast)
  // transed_mean_A = trans * mean_A in fp64 (as upstream); the small residual is then handed to fp32
  const double px = (double)p.x, py = (double)p.y, pz = (double)p.z;
  dax = x.R[0] * px + x.R[1] * py + x.R[2] * pz + x.t[0];
  day = x.R[3] * px + x.R[4] * py + x.R[5] * pz + x.t[1];
  daz = x.R[6] * px + x.R[7] * py + x.R[8] * pz + x.t[2];
  ax = (float)dax; ay = (float)day; az = (float)daz;
  dex = (double)q.x - dax; dey = (double)q.y - day; dez = (double)q.z - daz;
  ex = (float)dex; ey = (float)dey; ez = (float)dez;
}
__device__ __forceinline__ void residual(const Pose& x, const float4& p, const float4& q, float& ax, float& ay, float& az, float& ex, float& ey,
                                         float& ez, double& dex, double& dey, double& dez) {
  double dax, day, daz;
  residual64(x, p, q, dax, day, daz, ax, ay, az, ex, ey, ez, dex, dey, dez);
}

// ---- correspondence search of one source point -----------------------------------------------------------------------
// State of a search: nearest (d1, j1), runner-up distance d2nd, the squared radius inside which everything has been seen.
struct NnState {
  float d1;
  int j1;
  float d2nd;
  float explored2;
  bool complete;  // explored2 covers what the result needs
};
constexpr int kLaneLevelCap = 2;  // per-lane searches stop at the 4 x base-cell level (2 m cells by default: shells r <= 1 there cover the 2 m range of
                                  // the reference configuration); capping at level 1 was measured: 16-40 % of the lanes then go to the cooperative
                                  // search in the first iterations, which serialises them — iteration 0 took 11.1 ms instead of 7.8 ms at 1 024 pairs

// From a finished search: the correspondence and how far the query may move before the answer can change.
__device__ __forceinline__ void finish_search(const NnState& st, float max_corr_d2, bool track_margin, int& j_out, float& margin_out) {
  if (!track_margin) {
    j_out = (st.j1 >= 0 && st.d1 < max_corr_d2) ? st.j1 : -1;
    margin_out = 1.0e-30f;  // the answer holds for THIS pose only: search again at the next one
    return;
  }
  const float max_corr = sqrtf(max_corr_d2);
  float margin;
  if (st.j1 >= 0 && st.d1 < max_corr_d2) {
    j_out = st.j1;
    // the correspondent stays the nearest while the query moves less than half the gap to the runner-up, and stays in range
    // while it moves less than its distance to the range limit: one margin covers both, so an unmoved point is never touched
    margin = fminf(0.5f * (sqrtf(fminf(st.d2nd, st.explored2)) - sqrtf(st.d1)), max_corr - sqrtf(st.d1));
  } else {
    // nothing within range: stays so until the query has moved by (distance to the nearest thing) - max_corr
    j_out = -1;
    margin = sqrtf(fminf(st.j1 >= 0 ? st.d1 : INFINITY, fmaxf(st.explored2, max_corr_d2))) - max_corr;
  }
  margin_out = margin * (1.f - 1e-4f) - 1e-5f;  // fp32 slack
}

// squared radius a correspondence search has to cover: max_corr, or a little past it when margins are tracked — a point
// with nothing in range then learns HOW far the nearest thing is and is not searched again at every iteration
__device__ __forceinline__ float search_reach2(const GridView& g, float max_corr_d2, bool track_margin) {
  if (!track_margin || !isfinite(max_corr_d2)) return max_corr_d2;
  const float ext = sqrtf(max_corr_d2) + 0.25f * g.m.cell + 0.25f;
  return ext * ext;
}

// Per-lane part: seeded with the previous correspondent, bounded to the fine levels.  st.complete == false: the query is handed
// to coop_search (the whole wave searches for it).
__device__ __forceinline__ NnState lane_search(const GridView& g, float qx, float qy, float qz, int jprev, float max_corr_d2, bool track_margin) {
  NnState st{INFINITY, -1, INFINITY, 0.f, false};
  const bool seeded = jprev >= 0 && jprev < g.m.n;
  float seed_d2 = INFINITY;
  if (seeded) {
    const float4 pp = g.pts[jprev];
    seed_d2 = dist2f(qx, qy, qz, pp.x, pp.y, pp.z);
  }
  if (!track_margin) {
    // The pose is still moving by more than any margin could absorb: plain exact 1-NN (prunes with the best distance, not
    // the runner-up), seeded with the previous correspondent.
    Nearest1 nn{g.orig, seed_d2, seeded ? jprev : -1};
    int level0 = 0;
    if (seeded) {
      const float rel = sqrtf(seed_d2) * g.m.inv_cell;
      level0 = (rel <= 1.f) ? 0 : min(g.m.top_level, (int)ceilf(log2f(rel)));
    }
    st.explored2 = grid_search(g, qx, qy, qz, max_corr_d2, nn, level0, 1, kLaneLevelCap);
    st.d1 = nn.best_d2;
    st.j1 = nn.best_j;
    st.complete = st.d1 < st.explored2 || st.explored2 >= max_corr_d2;
    return st;
  }
  Nearest2 nn{g.orig, seed_d2, seeded ? jprev : -1, INFINITY, INFINITY};
  if (seeded && seed_d2 < 0.36f * g.m.cell * g.m.cell) {
    // close seed: scan the ball a little larger than the seed distance (so that an isolated nearest neighbour still
    // earns a non-zero margin) — a handful of base cells
    const float r = sqrtf(seed_d2) * 1.05f + 0.005f;  // (1.25 d + 2 cm: larger margins, but 3 % more correspondence time over 20 iterations)
    nn.limit = r * r;
    grid_ball_search(g, qx, qy, qz, nn, 2.1f);
    st = NnState{nn.d1, nn.j1, nn.d2nd, nn.limit, true};
    return st;
  }
  const float reach2 = search_reach2(g, max_corr_d2, true);
  int level0 = 0;
  if (seeded && seed_d2 < max_corr_d2) {
    const float rel = sqrtf(seed_d2) * g.m.inv_cell;
    level0 = (rel <= 1.f) ? 0 : min(g.m.top_level, (int)ceilf(log2f(rel)));
  }
  // (cells farther than the reach are pruned: what the walk has covered beyond it says nothing about them)
  st.explored2 = fminf(grid_search(g, qx, qy, qz, reach2, nn, level0, 1, kLaneLevelCap), reach2);
  st.d1 = nn.d1;
  st.j1 = nn.j1;
  st.d2nd = nn.d2nd;
  // the runner-up matters up to the explored radius only (finish_search clamps it), the nearest must lie inside it
  st.complete = st.explored2 >= reach2 || (st.d1 < st.explored2);
  return st;
}

// Wave-cooperative part: ALL lanes work on one query (wave-uniform arguments).  The cells of the box around the closed ball
// of squared radius R2 are looked up by 64 lanes at once, their points are walked 64 at a time with kCoopDepth chunk loads in
// flight, every lane keeps the nearest and runner-up of its share (ties: lower caller index) and a wave reduction gives the
// exact nearest, its runner-up distance and explored2 = R2.  A lone lane doing this search walks the same cells one point
// at a time — 250 k cycles for a far-field query, and every wave of the early iterations has a few of them.
constexpr int kCoopDepth = 8;
constexpr int kCoopCells = 128;  // cells of the cube around the ball, two per lane
__device__ __forceinline__ NnState coop_search(const GridView& g, float qx, float qy, float qz, float R2) {
  const GridMeta& m = g.m;
  const int lane = lane_id();
  const float R = sqrtf(R2) * (1.f + 1e-5f) + 1e-5f;
  // the finest level at which the cube around the ball needs at most kCoopCells cells; cells the ball does not reach are
  // dropped before their probe, so a query above a dense surface scans the patch below it, not an 8 m cube
  auto base_cell = [&](float v, float o, int d) { return min(max((int)floorf(cell_coord(v, o, m.inv_cell)), 0), d - 1); };
  const int x0 = base_cell(qx - R, m.ox, m.dx), x1 = base_cell(qx + R, m.ox, m.dx);
  const int y0 = base_cell(qy - R, m.oy, m.dy), y1 = base_cell(qy + R, m.oy, m.dy);
  const int z0 = base_cell(qz - R, m.oz, m.dz), z1 = base_cell(qz + R, m.oz, m.dz);
  int L = 0;
  auto count = [&](int l) { return ((x1 >> l) - (x0 >> l) + 1) * ((y1 >> l) - (y0 >> l) + 1) * ((z1 >> l) - (z0 >> l) + 1); };
  while (L < m.top_level && count(L) > kCoopCells) L++;
  const int cx0 = x0 >> L, cy0 = y0 >> L, cz0 = z0 >> L;
  const int sx = (x1 >> L) - cx0 + 1, sy = (y1 >> L) - cy0 + 1, sz = (z1 >> L) - cz0 + 1;
  const int nc = sx * sy * sz;
  const bool whole = nc > kCoopCells;  // cannot happen for a radius of a few metres; the whole cloud is scanned then
  const float cell = m.cell * (float)(1 << L);
  float d1 = INFINITY, d2nd = INFINITY;
  int j1 = -1;
  for (int c0 = 0; c0 < (whole ? 1 : nc); c0 += kWave) {
    int cs = 0, ce = 0;
    if (!whole && c0 + lane < nc) {
      const int ci = c0 + lane;
      const int ix = cx0 + ci % sx, iy = cy0 + (ci / sx) % sy, iz = cz0 + ci / (sx * sy);
      // lower bound of the distance from the query to the cell's cube (the grid clamps outside queries: clamp the test too)
      const float lx = m.ox + (float)ix * cell, ly = m.oy + (float)iy * cell, lz = m.oz + (float)iz * cell;
      const float gx = fmaxf(fmaxf(lx - qx, qx - (lx + cell)), 0.f), gy = fmaxf(fmaxf(ly - qy, qy - (ly + cell)), 0.f),
                  gz = fmaxf(fmaxf(lz - qz, qz - (lz + cell)), 0.f);
      const float lb = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz) - 1.0e-3f * cell, 0.f);
      const bool edge = ix == 0 || iy == 0 || iz == 0 || ix == ((m.dx - 1) >> L) || iy == ((m.dy - 1) >> L) || iz == ((m.dz - 1) >> L);
      if (edge || lb * lb <= R2) {  // (border cells also hold the points the clamping put there)
        int s0, e0;
        if (cell_range(g, ((uint32_t)L << 27) | morton27((uint32_t)ix, (uint32_t)iy, (uint32_t)iz), s0, e0)) { cs = s0; ce = e0; }
      }
    }
    unsigned long long todo = whole ? 0ull : __ballot(ce > cs);
    int j = whole ? -kWave : 0, e = whole ? m.n : 0;
    auto advance = [&]() -> bool {
      j += kWave;
      if (j < e) return true;
      if (!todo) return false;
      const int cl = __builtin_ctzll(todo);
      todo &= todo - 1;
      j = __builtin_amdgcn_readlane(cs, cl);
      e = __builtin_amdgcn_readlane(ce, cl);
      return true;
    };
    for (;;) {
      int jb[kCoopDepth];
      bool have[kCoopDepth], in[kCoopDepth];
      float4 p[kCoopDepth];
#pragma unroll
      for (int u = 0; u < kCoopDepth; u++) {
        have[u] = advance();
        jb[u] = j;
        in[u] = have[u] && j + lane < e;
        p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in[u]) p[u] = g.pts[j + lane];
      }
      if (!have[0]) break;
#pragma unroll
      for (int u = 0; u < kCoopDepth; u++) {
        if (!have[u]) break;
        const float dd = dist2f(qx, qy, qz, p[u].x, p[u].y, p[u].z);
        if (in[u] && dd <= R2) {
          const int jj = jb[u] + lane;
          if (dd < d1 || (dd == d1 && j1 >= 0 && g.orig[jj] < g.orig[j1]) || j1 < 0) {
            if (j1 >= 0) d2nd = fminf(d2nd, d1);
            d1 = dd;
            j1 = jj;
          } else if (dd < d2nd) {
            d2nd = dd;
          }
        }
      }
      if (!have[kCoopDepth - 1]) break;
    }
  }
  // wave reduction: the nearest (ties: lowest caller index), then the runner-up = the best of everything else
  const float best = uniform_f(wave_min_all(d1));
  NnState st{INFINITY, -1, INFINITY, R2, true};
  if (best < INFINITY) {
    int oc = (d1 == best) ? g.orig[j1] : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) oc = min(oc, __shfl_xor(oc, off, kWave));
    const bool win = (d1 == best) && g.orig[max(j1, 0)] == oc;
    const int wl = __builtin_ctzll(__ballot(win));
    st.d1 = best;
    st.j1 = __builtin_amdgcn_readlane(j1, wl);
    st.d2nd = uniform_f(wave_min_all(lane == wl ? d2nd : d1));
  }
  return st;
}

// ---- a handful of pairs (the odometry stream: ONE scan against a 3 x 10^5-point sub-map) --------------------------------------
// 16 k queries are 256 waves: the chip is mostly idle and every wave runs as long as its slowest lane's walk through cells that
// hold 30+ points each — a chain of dependent loads.  Here a query gets an aligned GROUP of lanes: the cells of the cube around
// its ball are probed kGroupW at a time, their points scanned kGroupW at a time, the partial results merged by lane exchanges:
// more waves, shorter walks.  The ball comes from a seed — the previous correspondent, or any point of the finest
// occupied cell that holds the query — so it contains the nearest neighbour by construction; in margin mode it is the same
// 1.05 d + 5 mm ball the per-lane search explores around a close seed.  Control flow is wave-uniform (loops run while ANY group
// has work) so that the lane exchanges always find their partners active.
constexpr int kGroupW = 32;  // measured on the 60-scan stream: 4 lanes 2.9 ms per scan, 8: 2.45, 16: 2.25, 32: 2.10, 64: 2.17 (per-lane walks: 4.25)
constexpr int kGroupSlots = 64 / kGroupW;  // cells per lane: 64 cells of the cube around the ball per query

__device__ __forceinline__ float group_min_f(float v) {
#pragma unroll
  for (int off = 1; off < kGroupW; off <<= 1) v = fminf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ int group_min_i(int v) {
#pragma unroll
  for (int off = 1; off < kGroupW; off <<= 1) v = min(v, __shfl_xor(v, off, kWave));
  return v;
}

// valid, q, jprev: the same in the lanes of a group.  complete == false: the cube needs more than 64 cells even at the top level
// (it cannot for a finite reach); the caller searches such a query per lane.
__device__ __forceinline__ NnState group_search(const GridView& g, bool valid, float qx, float qy, float qz, int jprev, float max_corr_d2,
                                                bool track_margin, int* scratch) {
  const GridMeta& m = g.m;
  const int lane = lane_id(), sub = lane & (kGroupW - 1), gbase = lane & ~(kGroupW - 1);
  NnState out{INFINITY, -1, INFINITY, 0.f, true};
  if (m.n <= 0) {
    out.explored2 = INFINITY;
    return out;
  }
  auto base_cell = [&](float v, float o, int d) { return min(max((int)floorf(cell_coord(v, o, m.inv_cell)), 0), d - 1); };
  // ---- a seed: the previous correspondent, else a few points of the finest occupied cell around the query
  float seed_d2 = INFINITY;
  if (valid && jprev >= 0 && jprev < m.n) {
    const float4 pp = g.pts[jprev];
    seed_d2 = dist2f(qx, qy, qz, pp.x, pp.y, pp.z);
  }
  {
    bool need = valid && !(seed_d2 < INFINITY);
    const int bx = base_cell(qx, m.ox, m.dx), by = base_cell(qy, m.oy, m.dy), bz = base_cell(qz, m.oz, m.dz);
    for (int l = 0; l <= m.top_level; l++) {
      if (!__any(need)) break;
      int s0 = 0, e0 = 0;
      if (need && !cell_range(g, ((uint32_t)l << 27) | morton27((uint32_t)(bx >> l), (uint32_t)(by >> l), (uint32_t)(bz >> l)), s0, e0)) s0 = e0 = 0;
      e0 = min(e0, s0 + 4 * kGroupW);  // any point will do as a seed
      float best = INFINITY;
      for (int j = s0 + sub; __any(j < e0); j += kGroupW)
        if (j < e0) {
          const float4 pp = g.pts[j];
          best = fminf(best, dist2f(qx, qy, qz, pp.x, pp.y, pp.z));
        }
      best = group_min_f(best);
      if (need && best < INFINITY) {
        seed_d2 = best;
        need = false;
      }
    }
  }
  // ---- the ball
  const float limit2 = track_margin ? search_reach2(g, max_corr_d2, true) : max_corr_d2;
  float R2 = limit2;
  if (seed_d2 < limit2) {
    if (track_margin) {
      const float r = sqrtf(seed_d2) * 1.05f + 0.005f;  // as lane_search around a close seed: an isolated neighbour still earns a margin
      R2 = fminf(r * r, limit2);
    } else {
      R2 = seed_d2 * (1.f + 4e-6f) + 1e-12f;
    }
  }
  // ---- the cells of the cube around the ball, at the finest level that needs at most 64 of them; lane `sub` owns cells sub, sub + kGroupW, ...
  const float R = sqrtf(R2) * (1.f + 1e-5f) + 1e-5f;
  const int x0 = base_cell(qx - R, m.ox, m.dx), x1 = base_cell(qx + R, m.ox, m.dx);
  const int y0 = base_cell(qy - R, m.oy, m.dy), y1 = base_cell(qy + R, m.oy, m.dy);
  const int z0 = base_cell(qz - R, m.oz, m.dz), z1 = base_cell(qz + R, m.oz, m.dz);
  int L = 0;
  auto count = [&](int l) { return ((x1 >> l) - (x0 >> l) + 1) * ((y1 >> l) - (y0 >> l) + 1) * ((z1 >> l) - (z0 >> l) + 1); };
  while (L < m.top_level && count(L) > kGroupW * kGroupSlots) L++;
  const int cx0 = x0 >> L, cy0 = y0 >> L, cz0 = z0 >> L;
  const int sx = (x1 >> L) - cx0 + 1, sy = (y1 >> L) - cy0 + 1, sz = (z1 >> L) - cz0 + 1;
  const int nc = sx * sy * sz;
  const bool fits = isfinite(R2) && nc <= kGroupW * kGroupSlots;
  const float cell = m.cell * (float)(1 << L);
  int cs[kGroupSlots], ce[kGroupSlots];
  {
    uint32_t key[kGroupSlots];
    int4 sl[kGroupSlots];
    bool want[kGroupSlots];
#pragma unroll
    for (int t = 0; t < kGroupSlots; t++) {
      const int ci = t * kGroupW + sub;
      want[t] = false;
      key[t] = 0;
      cs[t] = ce[t] = 0;
      if (valid && fits && ci < nc) {
        const int ix = cx0 + ci % sx, iy = cy0 + (ci / sx) % sy, iz = cz0 + ci / (sx * sy);
        const float lx = m.ox + (float)ix * cell, ly = m.oy + (float)iy * cell, lz = m.oz + (float)iz * cell;
        const float gx = fmaxf(fmaxf(lx - qx, qx - (lx + cell)), 0.f), gy = fmaxf(fmaxf(ly - qy, qy - (ly + cell)), 0.f),
                    gz = fmaxf(fmaxf(lz - qz, qz - (lz + cell)), 0.f);
        const float lb = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz) - 1.0e-3f * cell, 0.f);
        const bool edge = ix == 0 || iy == 0 || iz == 0 || ix == ((m.dx - 1) >> L) || iy == ((m.dy - 1) >> L) || iz == ((m.dz - 1) >> L);
        want[t] = edge || lb * lb <= R2;  // (border cells also hold the points the clamping put there)
        key[t] = ((uint32_t)L << 27) | morton27((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
      }
      if (want[t]) sl[t] = first_slot(g, key[t]);  // the probes of a lane are independent: all in flight together
    }
#pragma unroll
    for (int t = 0; t < kGroupSlots; t++)
      if (want[t]) {
        int s0, e0;
        if (resolve_range(g, key[t], sl[t], s0, e0)) { cs[t] = s0; ce[t] = e0; }
      }
  }
  // ---- scan: the points of the group's cells as ONE flat sequence (cells in slot-major, lane order), kGroupW points per step and
  // kGroupDepth steps in flight.  (Until round 4 the group walked its up to 64 cells one after the other, a dependent load per cell
  // of ~30 points: 10 - 27 memory latencies per query against a 3 x 10^5-point sub-map.)  scratch: 2 x 64 ints of LDS per group -
  // exclusive prefix sums of the cell sizes and the cells' range starts; a position finds its cell by binary search.
  float d1 = INFINITY, d2nd = INFINITY;
  int j1 = -1;
  {
    constexpr int kCells = kGroupW * kGroupSlots;  // 64
    constexpr int kGroupDepth = 4;
    int run = 0;  // exclusive prefix over the group's cells, slot-major
    int excl[kGroupSlots];
#pragma unroll
    for (int t = 0; t < kGroupSlots; t++) {
      const int sz = max(ce[t] - cs[t], 0);
      int incl = sz;
#pragma unroll
      for (int off = 1; off < kGroupW; off <<= 1) {
        const int v = __shfl_up(incl, off, kWave);
        if (sub >= off) incl += v;
      }
      excl[t] = run + incl - sz;
      run += __shfl(incl, gbase + kGroupW - 1, kWave);
    }
    const int total = run;  // (the same in the lanes of a group)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int t = 0; t < kGroupSlots; t++) {
      scratch[t * kGroupW + sub] = excl[t];
      scratch[kCells + t * kGroupW + sub] = cs[t];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // Which walk?  A voxelised scan as the target has ~6 points per base cell: one step per cell leaves most of a group's lanes idle and
    // the flat sequence wins (a lone 64 k pair 1.46 -> 1.28 ms).  A sub-map of twenty overlapped scans has 30 - 60: a step per cell is a
    // full one, and the flat sequence only adds its binary searches (the odometry stream 2.0 -> 2.2 ms per scan with it): dense groups
    // keep the walk cell by cell.  (The choice is per wave: both groups of a wave take the same loop.)
    int nonempty = 0;
#pragma unroll
    for (int t = 0; t < kGroupSlots; t++) nonempty += __popcll(__ballot(ce[t] > cs[t]));
    const long all_points = (long)__shfl(total, 0, kWave) + (long)__shfl(total, kGroupW, kWave);
    const bool dense = all_points >= 32l * nonempty;  // (16 / 24 / 48: the stream 2.04 / 1.98 / 1.97 ms per scan, the lone pair 1.36 / 1.34 / 1.33 ms)
    if (dense) {
      // the non-empty cells, compacted into the group's scratch as (start, end) pairs, then walked with the loads of the NEXT cell's first
      // two steps in flight under the arithmetic of the current one (the walk used to be a dependent load per cell)
      int n_cells = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int t = 0; t < kGroupSlots; t++) {
        const bool has = ce[t] > cs[t];
        const unsigned long long mask = (__ballot(has) >> gbase) & ((kGroupW >= 64) ? ~0ull : ((1ull << (kGroupW & 63)) - 1ull));
        if (has) {
          const int pos = n_cells + __popcll(mask & ((1ull << sub) - 1ull));
          scratch[2 * pos] = cs[t];
          scratch[2 * pos + 1] = ce[t];
        }
        n_cells += __popcll(mask);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      auto consider = [&](bool in, int jq, const float4& pq) {
        const float dd = dist2f(qx, qy, qz, pq.x, pq.y, pq.z);
        if (in && dd <= R2) {
          if (j1 < 0 || dd < d1 || (dd == d1 && g.orig[jq] < g.orig[j1])) {
            if (j1 >= 0) d2nd = fminf(d2nd, d1);
            d1 = dd;
            j1 = jq;
          } else if (dd < d2nd) {
            d2nd = dd;
          }
        }
      };
      auto fetch = [&](int k, int& s0, int& e0, float4& p0, float4& p1) {
        s0 = e0 = 0;
        p0 = p1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < n_cells) {
          s0 = scratch[2 * k];
          e0 = scratch[2 * k + 1];
          if (s0 + sub < e0) p0 = g.pts[s0 + sub];
          if (s0 + sub + kGroupW < e0) p1 = g.pts[s0 + sub + kGroupW];
        }
      };
      int s_nx, e_nx;
      float4 p0_nx, p1_nx;
      fetch(0, s_nx, e_nx, p0_nx, p1_nx);
      const int n_max = max(__shfl(n_cells, 0, kWave), __shfl(n_cells, kGroupW, kWave));  // (both groups of the wave walk together)
      for (int k = 0; k < n_max; k++) {
        const int s0 = s_nx, e0 = e_nx;
        const float4 p0 = p0_nx, p1 = p1_nx;
        fetch(k + 1, s_nx, e_nx, p0_nx, p1_nx);
        consider(s0 + sub < e0, s0 + sub, p0);
        consider(s0 + sub + kGroupW < e0, s0 + sub + kGroupW, p1);
        for (int j = s0 + sub + 2 * kGroupW; __any(j < e0); j += 2 * kGroupW) {  // a cell of more than 64 points: the rest as before
          const bool in0 = j < e0, in1 = j + kGroupW < e0;
          float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
          if (in0) q0 = g.pts[j];
          if (in1) q1 = g.pts[j + kGroupW];
          consider(in0, j, q0);
          consider(in1, j + kGroupW, q1);
        }
      }
    } else
    for (int f0 = 0; __any(f0 < total); f0 += kGroupW * kGroupDepth) {
      int jj[kGroupDepth];
      bool in[kGroupDepth];
      float4 pp[kGroupDepth];
#pragma unroll
      for (int u = 0; u < kGroupDepth; u++) {
        const int f = f0 + u * kGroupW + sub;
        in[u] = f < total;
        jj[u] = 0;
        pp[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in[u]) {
          int lo = 0;
#pragma unroll
          for (int step = kCells / 2; step > 0; step >>= 1)
            if (scratch[lo + step] <= f) lo += step;
          jj[u] = scratch[kCells + lo] + (f - scratch[lo]);
          pp[u] = g.pts[jj[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < kGroupDepth; u++) {
        const float dd = dist2f(qx, qy, qz, pp[u].x, pp[u].y, pp[u].z);
        if (in[u] && dd <= R2) {
          if (j1 < 0 || dd < d1 || (dd == d1 && g.orig[jj[u]] < g.orig[j1])) {
            if (j1 >= 0) d2nd = fminf(d2nd, d1);
            d1 = dd;
            j1 = jj[u];
          } else if (dd < d2nd) {
            d2nd = dd;
          }
        }
      }
    }
  }
  // ---- merge the partial results of the lanes: the nearest (ties: lowest caller index), its runner-up = the best of everything else
  const float best = group_min_f(d1);
  out.explored2 = R2;
  out.complete = fits;
  if (best < INFINITY) {
    const int oc = group_min_i((d1 == best && j1 >= 0) ? g.orig[j1] : 0x7fffffff);
    const bool win = d1 == best && j1 >= 0 && g.orig[j1] == oc;
    constexpr unsigned long long kGroupMask = kGroupW >= 64 ? ~0ull : ((1ull << (kGroupW & 63)) - 1ull);
    const int wl = __builtin_ctzll((__ballot(win) >> gbase) & kGroupMask) & (kGroupW - 1);
    out.d1 = best;
    out.j1 = __shfl(j1, gbase + wl, kWave);
    out.d2nd = group_min_f(sub == wl ? d2nd : d1);
  }
  return out;
}

// (4 waves per SIMD: 131 VGPRs unconstrained, 127 without spills under the bound)
template <int kTile>
__global__ __launch_bounds__(kGicpBlock, 4) void k_gicp_correspond_group(const float4* __restrict__ spts, const int* __restrict__ scount, size_t scap,
                                                                      const float4* __restrict__ tpts, const int* __restrict__ torig,
                                                                      const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta,
                                                                      size_t tcap, size_t tslots, const int* __restrict__ tgt_of_src,
                                                                      const GicpPairState* __restrict__ state, int* __restrict__ corr,
                                                                      float4* __restrict__ qsearch, float max_corr_d2, int G, int S,
                                                                      double margin_step) {
  __shared__ int queue[kTile];
  __shared__ int gscratch[kGicpBlock / kGroupW][2 * kGroupW * kGroupSlots];  // group_search: prefix sums / range starts of a group's cells
  __shared__ int qcount;
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const GicpPairState* st = state + pair;
  if (!st->active) return;
  const bool track_margin = st->last_step < margin_step;
  const GridView g = grid_view(tpts, torig, ttable, tmeta, tgt_of_src[pair], tcap, tslots);
  const int n = scount[pair];
  const Pose x0 = st->x0;
  float Tf[12];
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 3; c++) Tf[4 * r + c] = (float)x0.R[3 * r + c];
    Tf[4 * r + 3] = (float)x0.t[r];
  }
  const size_t so = (size_t)pair * scap;
  constexpr int kQueriesPerRound = kGicpBlock / kGroupW;
  for (int base = bx * kTile; base < n; base += G * kTile) {
    if (threadIdx.x == 0) qcount = 0;
    __syncthreads();
    for (int u = threadIdx.x; u < kTile; u += kGicpBlock) {  // phase 1 of k_gicp_correspond: who needs a search?
      const int i = base + u;
      if (i >= n) continue;
      const float4 p = spts[so + i];
      const float qx = transform_row(Tf, p.x, p.y, p.z), qy = transform_row(Tf + 4, p.x, p.y, p.z), qz = transform_row(Tf + 8, p.x, p.y, p.z);
      const float4 qs = qsearch[so + i];
      if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) corr[so + i] = -1;
      else if (!(sqrtf(dist2f(qx, qy, qz, qs.x, qs.y, qs.z)) < qs.w)) queue[atomicAdd(&qcount, 1)] = i;
    }
    __syncthreads();
    const int nq = qcount;
    for (int t0 = 0; t0 < nq; t0 += kQueriesPerRound) {
      const int tq = t0 + (int)threadIdx.x / kGroupW;
      const bool valid = tq < nq;
      const int i = valid ? queue[tq] : 0;
      const float4 p = spts[so + i];
      const float qx = transform_row(Tf, p.x, p.y, p.z), qy = transform_row(Tf + 4, p.x, p.y, p.z), qz = transform_row(Tf + 8, p.x, p.y, p.z);
      NnState stq = group_search(g, valid, qx, qy, qz, valid ? corr[so + i] : -1, max_corr_d2, track_margin, gscratch[threadIdx.x / kGroupW]);
      // (a cube that does not fit 64 cells: one lane of the group walks the hash as k_gicp_correspond would)
      const bool lone = valid && !stq.complete && (threadIdx.x & (kGroupW - 1)) == 0;
      if (lone) stq = lane_search(g, qx, qy, qz, corr[so + i], max_corr_d2, true);
      if (valid && (threadIdx.x & (kGroupW - 1)) == 0) {
        int j;
        float margin;
        finish_search(stq, max_corr_d2, track_margin || lone, j, margin);
        corr[so + i] = j;
        qsearch[so + i] = make_float4(qx, qy, qz, margin);
      }
    }
    __syncthreads();
  }
}

// F3 update_correspondences, incremental and exact.  PPT = points per thread and pass (tile = 256 * PPT points).
// Occupancy: 5 waves per SIMD (96 VGPRs, 32 B of scratch in the search path) — measured at 1 024 pairs: 12.8 ms per step of 20
// launches against 13.6 ms at the 126 VGPRs the compiler takes unconstrained; 6 waves (80 VGPRs, 84 B of scratch): 13.9 ms.
template <int kLinPointsPerThread>
__global__ __launch_bounds__(kGicpBlock, 5) void k_gicp_correspond(const float4* __restrict__ spts,
                                                                const int* __restrict__ scount, size_t scap, const float4* __restrict__ tpts,
                                                                const int* __restrict__ torig,
                                                                const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta, size_t tcap,
                                                               size_t tslots, const int* __restrict__ tgt_of_src,
                                                               const GicpPairState* __restrict__ state, int* __restrict__ corr,
                                                                float4* __restrict__ qsearch, float max_corr_d2, int G, int S,
                                                                int tiles_took_the_moving_pairs, double margin_step,
                                                                unsigned long long* __restrict__ stats = nullptr, int iteration = 0,
                                                                const unsigned long long* __restrict__ safe_mask = nullptr, size_t mask_words = 0) {
  constexpr int kLinTile = kGicpBlock * kLinPointsPerThread;
  constexpr int kMaskTiles = 64;  // tiles of a block whose "safe" words are fetched in one go
  __shared__ int queue[kLinTile + kGicpBlock];  // what is left of the earlier tiles (< one block's worth) + this tile's
  __shared__ int qcount;
#ifndef CORR_UNSAFE_CAP
#define CORR_UNSAFE_CAP 2048  // (A/B: 1 024 brings the kernel's LDS to 6 KB - what is free on a CU beside eight k-NN waves)
#endif
  constexpr int kUnsafeCap = CORR_UNSAFE_CAP;  // points of a block that are not marked safe: listed when they are at most this many
  __shared__ int ulist[kUnsafeCap];
  __shared__ int ucount;
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const GicpPairState* st = state + pair;
  if (!st->active) return;
  // margins only pay once the updates are small enough to stay inside them (cm); before that every point is searched again:
  // k_gicp_correspond_tile does that for whole tiles at once (the first iterations), this kernel with the plain 1-NN search
  // A pair k_gicp_correspond_tile has just been through: only what it left behind (NaN margin: ties between its keys, boxes it
  // could not scan) is searched here, with margins like everything else it produced.
  const bool tile_pair = !(st->last_step < margin_step) && tiles_took_the_moving_pairs;
  const bool track_margin = st->last_step < margin_step || tile_pair;
  const int t = tgt_of_src[pair];
  const GridView g = grid_view(tpts, torig, ttable, tmeta, t, tcap, tslots);
  const int n = scount[pair];
  const Pose x0 = st->x0;
  float Tf[12];
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 3; c++) Tf[4 * r + c] = (float)x0.R[3 * r + c];
    Tf[4 * r + 3] = (float)x0.t[r];
  }
  const size_t so = (size_t)pair * scap;
  // The cost pass of the previous iteration has marked the points whose margin survives every trial pose it evaluated (safe_mask): a
  // pair that tracks margins looks at the OTHER points only - a few per cent of them once the pose has settled (the margins of a
  // voxelised cloud are spread over centimetres; the poses of the last iterations move by fractions of a millimetre).  The words of
  // all this block's tiles come in with one load per thread, their zero bits become a list in LDS, and the margin check runs over
  // that list: a block is then three memory latencies - words, listed points, the odd search - where the tile loop below is a chain
  // of dependent loads and three barriers per 256 points (0.5 ms per iteration at 2 048 pairs, whatever the bytes).
  constexpr int kWordsPerTile = kLinTile / kWave;
  const bool masked = safe_mask != nullptr && st->mask_valid && (st->last_step < margin_step) && (n <= kMaskTiles * G * kLinTile);
  bool use_list = false;
  int n_list = 0;
  if (threadIdx.x == 0) { qcount = 0; ucount = 0; }
  __syncthreads();
  if (masked) {
    const unsigned long long* mw = safe_mask + (size_t)pair * mask_words;
    for (int w = threadIdx.x; w < kMaskTiles * kWordsPerTile; w += kGicpBlock) {
      const int tile_no = w / kWordsPerTile;  // this block's tile_no-th tile: points (bx + tile_no G) kLinTile ..., and the word of them this thread reads
      const long first = (long)(bx + (long)tile_no * G) * kLinTile + (long)(w % kWordsPerTile) * kWave;
      if (first >= n) continue;
      unsigned long long bits = ~mw[first / kWave];  // (the cost pass sets the bits past the end of the cloud)
      if (!bits) continue;
      int pos = atomicAdd(&ucount, __popcll(bits));
      while (bits) {
        const int bit = __builtin_ctzll(bits);
        bits &= bits - 1;
        if (pos < kUnsafeCap) ulist[pos] = (int)first + bit;
        pos++;
      }
    }
    __syncthreads();
    n_list = ucount;
    use_list = n_list <= kUnsafeCap;  // (more than that - a pair that has just started to track margins: the tile loop)
  }
  for (int step = 0;; step++) {
    int base = 0;
    bool last_tile;
    if (use_list) {
      if (step * kLinTile >= n_list) break;
      last_tile = (step + 1) * kLinTile >= n_list;
    } else {
      base = (bx + step * G) * kLinTile;
      if (base >= n) break;
      last_tile = base + G * kLinTile >= n;
    }
    // ---- phase 1: which points of this tile need a new nearest-neighbour search?  
    // The last full search of a point left, at query position qs.xyz, a margin qs.w: while the query has moved less
    // than that, its nearest neighbour provably cannot change (triangle inequality on the nearest / runner-up
    // distances).  Points that do need a search are COMPACTED into an LDS queue so that the (divergent, latency-bound)
    // search runs on dense waves instead of stalling every wave of the block behind a few lanes: the queue is searched a
    // whole block's worth at a time, what is left waits for the next tile's (when 1 - 10 % of the points have left their
    // margins - the first iterations of margin tracking - a tile alone fills one or two of the block's four waves, partly).
#pragma unroll
    for (int u = 0; u < kLinPointsPerThread; u++) {
      int i;
      if (use_list) {
        const int at = step * kLinTile + u * kGicpBlock + (int)threadIdx.x;
        if (at >= n_list) continue;
        i = ulist[at];
      } else {
        i = base + u * kGicpBlock + threadIdx.x;
        if (i >= n) continue;
      }
      const float4 p = spts[so + i];
      const float qx = transform_row(Tf, p.x, p.y, p.z), qy = transform_row(Tf + 4, p.x, p.y, p.z), qz = transform_row(Tf + 8, p.x, p.y, p.z);
      const float4 qs = qsearch[so + i];
      const bool finite_q = isfinite(qx) && isfinite(qy) && isfinite(qz);
      const float moved = sqrtf(dist2f(qx, qy, qz, qs.x, qs.y, qs.z));
      // NaN margin (never searched) -> search.  The margin also covers "still within max_corr" (search_point), so a point
      // that passes this test needs neither its correspondence index nor its correspondent: 32 B per point in steady state.
      const bool need_search = tile_pair ? (qs.w != qs.w) : !(moved < qs.w);
      if (!finite_q) {
        corr[so + i] = -1;
      } else if (need_search) {
        queue[atomicAdd(&qcount, 1)] = i;
      }
    }
    __syncthreads();
    // ---- phase 2: dense search of the queued points
    int nq = qcount;
    if (stats && threadIdx.x == 0 && bx == 0 && step == 0 && track_margin && !tile_pair) atomicAdd(&stats[8 + 4 * min(iteration, 63) + 3], 1ull);
    const float reach_coop2 = search_reach2(g, max_corr_d2, true);
    while (nq >= kGicpBlock || (last_tile && nq > 0)) {  // whole waves go round together: the cooperative part needs all 64 lanes
      const int take = min(nq, kGicpBlock);
      if (stats && threadIdx.x == 0) atomicAdd(&stats[8 + 4 * min(iteration, 63)], (unsigned long long)take);
      nq -= take;
      const int tq = nq + threadIdx.x;
      const bool valid = (int)threadIdx.x < take;
      const int i = valid ? queue[tq] : 0;
      const float4 p = spts[so + i];
      const float qx = transform_row(Tf, p.x, p.y, p.z), qy = transform_row(Tf + 4, p.x, p.y, p.z), qz = transform_row(Tf + 8, p.x, p.y, p.z);
      NnState st{INFINITY, -1, INFINITY, 0.f, true};
      bool was_hard = false;
      if (valid) st = lane_search(g, qx, qy, qz, corr[so + i], max_corr_d2, track_margin);
      // the queries the fine levels could not finish, one at a time, by the whole wave
      unsigned long long hard = __ballot(valid && !st.complete);
      if (stats && lane_id() == 0 && hard) atomicAdd(&stats[8 + 4 * min(iteration, 63) + 2], (unsigned long long)__popcll(hard));
      while (hard) {
        const int hl = __builtin_ctzll(hard);
        hard &= hard - 1;
        const float hx = lane_f(qx, hl), hy = lane_f(qy, hl), hz = lane_f(qz, hl);
        // the cooperative search always covers the full reach: its result comes with a margin whatever the mode — a far-field
        // point with nothing in range keeps that answer while the pose moves less than its distance to the nearest thing
        const NnState cs = coop_search(g, hx, hy, hz, reach_coop2);
        if (lane_id() == hl) { st = cs; was_hard = true; }
      }
      if (valid) {
        int j;
        float margin;
        finish_search(st, max_corr_d2, track_margin || was_hard, j, margin);
        corr[so + i] = j;
        qsearch[so + i] = make_float4(qx, qy, qz, margin);
      }
    }
    __syncthreads();  // the searched entries have been read: the next tile appends behind what is left
    if (threadIdx.x == 0) qcount = nq;
    __syncthreads();
  }
}

// ---- the same for a pair whose pose still moves by more than any margin absorbs: every point is searched, by TILES -------
// One wave takes 64 consecutive source points (Morton order: a compact patch, and so is its rigid transform), gathers the
// target points of the cells around the patch into LDS once and every lane scans that list for its own query — the scheme
// of k_knn_cov_tile (tile.hpp) with a 2-element network.  A per-lane walk pays one dependent, divergent load per candidate
// and per lane; here a candidate costs one coalesced global load per WAVE and, per lane, one LDS broadcast + 9 VALU ops (6 with CORR_PK, below):
// the squared distance (fma form) with its low kCorrOrdBits mantissa bits replaced by the candidate's position in the list
// is ONE 32-bit key; v_min_u32 keeps the smallest key, v_med3_u32 the runner-up, no branch, no index bookkeeping.
//   round 1: radius r1 (most points have their correspondent within a few decimetres);
//   round 2: the lanes that found nothing there, radius = the full reach, skipping the box round 1 has seen;
//   a patch that straddles a jump of the Morton curve (its box then holds thousands of target points) is split into four
//   16-lane parts that go through the rounds one after the other.
// The candidate list is a kCorrCand-entry LDS ring that is scanned and reused whenever it fills up, so no box is too big.
// Exactness: the keys order candidates by distance up to 2^kCorrOrdBits ulps; a lane whose two smallest keys fall into the
// same or adjacent buckets (a tie or a near-tie: lattice clouds) repeats its search per lane, with exact distances and the
// lowest-caller-index tie rule.  The nearest's exact distance is recomputed with dist2f, the runner-up enters the margin as
// the lower edge of its bucket, so every margin is at most what the per-lane search would have granted.
#ifndef CORR_SCAN_BATCH
#define CORR_SCAN_BATCH 0  // 1: four candidates per step with the next step's LDS reads in flight - measured SLOWER (3.97 against 3.64 ms per
                           // 2 048-pair launch: a fifth more instructions, and four waves per SIMD hide the LDS latency already); kept for A/B builds
#endif
constexpr int kCorrOrdBits = 9;
constexpr int kCorrCand = 1 << kCorrOrdBits;
constexpr uint32_t kCorrOrdMask = (uint32_t)kCorrCand - 1u;
constexpr uint32_t kCorrEmpty = 0xFFFFFFFFu;

struct CorrBest {
  uint32_t k1, k2;  // smallest and second smallest key
  int j1;           // sorted target index behind k1
};

#ifndef CORR_PK
#define CORR_PK 1  // 0: one candidate per step from float4 entries (A/B builds)
#endif
// CORR_PK: the ring holds the candidates in PAIRS - (x0 x1 y0 y1) (z0 z1 w0 w1) per 32 bytes - so that a lane reads the coordinates of
// two candidates as aligned register pairs and takes both squared distances with packed fp32 instructions (v_pk_add / v_pk_mul /
// v_pk_fma_f32: 6 instead of 12 for two candidates, the same IEEE operations in the same order per candidate).
typedef float corr_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void corr_put(float4* cand, int c, float x, float y, float z, float w) {
#if CORR_PK
  float* f = reinterpret_cast<float*>(cand) + 8 * (c >> 1) + (c & 1);
  f[0] = x; f[2] = y; f[4] = z; f[6] = w;
#else
  cand[c] = make_float4(x, y, z, w);
#endif
}
__device__ __forceinline__ float corr_payload(const float4* cand, int c) {
#if CORR_PK
  return reinterpret_cast<const float*>(cand)[8 * (c >> 1) + 6 + (c & 1)];
#else
  return cand[c].w;
#endif
}
__device__ __forceinline__ void corr_scan(const float4* __restrict__ cand, int n, float qx, float qy, float qz, CorrBest& b) {
  const uint32_t before = b.k1;
  uint32_t k1 = b.k1, k2 = b.k2;
#if CORR_PK
  {
    // (the mask lives in a VGPR: with a literal mask AND a scalar position the and-or needs two constant-bus operands and the compiler
    // splits it into v_and + v_or)
    uint32_t keep = ~kCorrOrdMask;
    asm volatile("" : "+v"(keep));
    auto take = [&](float dd, int c) {
      const uint32_t key = (__float_as_uint(dd) & keep) | (uint32_t)c;
      k2 = max(min(k1, k2), min(max(k1, k2), key));  // median(k1, k2, key) -> v_med3_u32: the new runner-up
      k1 = min(k1, key);
    };
    const corr_v2f qxv = {qx, qx}, qyv = {qy, qy}, qzv = {qz, qz};
    const int np = n >> 1;
#pragma unroll 4
    for (int p = 0; p < np; p++) {
      const float4 a = cand[2 * p], c2 = cand[2 * p + 1];  // (x0 x1 y0 y1) (z0 z1 w0 w1)
      const corr_v2f px = {a.x, a.y}, py = {a.z, a.w}, pz = {c2.x, c2.y};
      const corr_v2f dx = qxv - px, dy = qyv - py, dz = qzv - pz;
      const corr_v2f dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      take(dd.x, 2 * p);
      take(dd.y, 2 * p + 1);
    }
    if (n & 1) {
      const float4 a = cand[2 * np], c2 = cand[2 * np + 1];
      const float dx = qx - a.x, dy = qy - a.z, dz = qz - c2.x;
      take(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)), n - 1);
    }
    if (k1 != before) b.j1 = __float_as_int(corr_payload(cand, (int)(k1 & kCorrOrdMask)));
    b.k1 = k1;
    b.k2 = k2;
    return;
  }
#endif
  auto one = [&](const float4& p, int c) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const uint32_t key = (__float_as_uint(dd) & ~kCorrOrdMask) | (uint32_t)c;
    k2 = max(min(k1, k2), min(max(k1, k2), key));  // median(k1, k2, key) -> v_med3_u32: the new runner-up
    k1 = min(k1, key);
  };
  // Four candidates per step, the four LDS reads of the NEXT step requested before this step's 40 VALU instructions run (the same
  // address in every lane: broadcasts).  Left to itself the compiler - at the 128 registers of four waves per SIMD - re-used one
  // register triple for the second half of an 8-way unrolled body: four exposed LDS latencies per eight candidates, about as long
  // as their arithmetic (PMC: 28 % VALU issue, 41 % waiting).  Reading up to 4 entries past n stays inside the ring's slack.
#if CORR_SCAN_BATCH
  int c = 0;
  float4 p0 = cand[0], p1 = cand[1], p2 = cand[2], p3 = cand[3];
  for (; c + 4 <= n; c += 4) {
    const float4 q0 = cand[c + 4], q1 = cand[c + 5], q2 = cand[c + 6], q3 = cand[c + 7];
    __builtin_amdgcn_sched_barrier(0);
    one(p0, c); one(p1, c + 1); one(p2, c + 2); one(p3, c + 3);
    __builtin_amdgcn_sched_barrier(0);
    p0 = q0; p1 = q1; p2 = q2; p3 = q3;
  }
  if (c < n) one(p0, c);
  if (c + 1 < n) one(p1, c + 1);
  if (c + 2 < n) one(p2, c + 2);
#else  // (A/B aid: the loop of round 3)
#pragma unroll 8
  for (int c = 0; c < n; c++) one(cand[c], c);
#endif
  if (k1 != before) b.j1 = __float_as_int(cand[k1 & kCorrOrdMask].w);
  b.k1 = k1;
  b.k2 = k2;
}

// Gather the points of `cells` inside `nb` (and outside `ob`) through the LDS ring, scanning whenever it is full.
__device__ __forceinline__ void corr_gather_scan(const GridView& g, const TileCells& cells_in, const TileBox& nb, const TileBox& ob, bool has_old,
                                                 float4* cand, float qx, float qy, float qz, CorrBest& best, int* scratch, const TileBox& ob2, bool has_old2) {
  const int lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  // the cells' points as one flat sequence (tile.hpp FlatCells): full 64-lane loads instead of one load per ~6-point cell
  FlatCells fc;
  fc.init(cells_in, scratch);
  int n = 0;
  for (int f0 = 0; f0 < fc.total; f0 += kWave * kGatherDepth) {
    int idx[kGatherDepth];
    float4 p[kGatherDepth];
#pragma unroll
    for (int u = 0; u < kGatherDepth; u++) {
      const int f = f0 + u * kWave + lane;
      idx[u] = 0;
      p[u] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
      if (f < fc.total) {
        idx[u] = fc.index_of(f, scratch);
        p[u] = g.pts[idx[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < kGatherDepth; u++) {
      if (f0 + u * kWave >= fc.total) break;
      const bool in = in_box(nb, p[u].x, p[u].y, p[u].z) && !(has_old && in_box(ob, p[u].x, p[u].y, p[u].z)) &&
                      !(has_old2 && in_box(ob2, p[u].x, p[u].y, p[u].z));
      const unsigned long long mask = __ballot(in);
      const int cnt = __popcll(mask);
      if (n + cnt > kCorrCand) {  // ring full: every lane scans what is there, then it is reused
        corr_scan(cand, n, qx, qy, qz, best);
        n = 0;
      }
      if (in) corr_put(cand, n + __popcll(mask & below), p[u].x, p[u].y, p[u].z, __int_as_float(idx[u]));
      n += cnt;
    }
  }
  corr_scan(cand, n, qx, qy, qz, best);
}

// upper bound of the distance behind a key (upper edge of its bucket + the fma form's rounding)
__device__ __forceinline__ float corr_key_ub(uint32_t k) { return __uint_as_float((k | kCorrOrdMask) + 2u); }

// The two rounds for the lanes of `act` (wave-uniform control flow).  explored2 = the squared radius inside which the lane
// has seen every target point; 0 = not decided here (left to the per-lane search).
// A round cannot run only when its box needs more than 64 cells at the top level of the hash (it never does for radii of metres).
constexpr int kCorrSplitPoints = 2048;  // a 64-lane patch whose round-1 cells hold more points than this is split into quarters
__device__ __forceinline__ void corr_rounds(const GridView& g, bool act, float qx, float qy, float qz, float R1, float R1_2, float reach, float reach2,
                                            bool one_round, float4* cand, CorrBest& best, float& explored2, unsigned long long* stats, int* scratch,
                                            float r_mid = 0.f) {
  const TileBox box = tile_bounds(act, qx, qy, qz);
  const TileBox nb = grow(box, R1 + tile_eps(box, reach));
  const TileCells cells = tile_cells(g, nb, 0x7fffffff);
  bool seen = false;
  if (cells.ok) {
    // everything in the box counts (round 2 skips this box), but only what lies within R1 is known to be complete
    CorrBest b = best;
    corr_gather_scan(g, cells, nb, nb, false, cand, qx, qy, qz, b, scratch, nb, false);
    if (act) { best = b; explored2 = R1_2; }
    seen = true;
  }
  const bool found1 = best.k1 != kCorrEmpty && corr_key_ub(best.k1) <= R1_2;
  bool act2 = act && !(seen && (one_round || found1));
  const unsigned long long m2 = __ballot(act2);
  if (stats && lane_id() == 0) {  // tuning aid (MI355_CORR_STATS)
    atomicAdd(&stats[0], 1ull);
    atomicAdd(&stats[1], (unsigned long long)cells.total);
    if (!cells.ok) atomicAdd(&stats[2], 1ull);
    if (m2) { atomicAdd(&stats[3], 1ull); atomicAdd(&stats[4], (unsigned long long)__popcll(m2)); }
  }
  if (!m2) return;
  // A middle round (ICP: a reach of tens of metres never fits the last round's 64 top-level cells, and whoever fails round 1 would go
  // straight to the per-lane search): the lanes round 1 left, radius r_mid, skipping round 1's box.
  TileBox nbm = nb;
  bool seen_mid = false;
  if (r_mid > R1 && r_mid < reach) {
    const TileBox boxm = tile_bounds(act2, qx, qy, qz);
    nbm = grow(boxm, r_mid + tile_eps(boxm, reach));
    const TileCells cellsm = tile_cells(g, nbm, 0x7fffffff);
    if (cellsm.ok) {
      CorrBest b = best;
      corr_gather_scan(g, cellsm, nbm, nb, seen, cand, qx, qy, qz, b, scratch, nb, false);
      if (act2) { best = b; explored2 = r_mid * r_mid; }
      seen_mid = true;
      act2 = act2 && !(best.k1 != kCorrEmpty && corr_key_ub(best.k1) <= r_mid * r_mid);
      if (!__ballot(act2)) return;
    }
  }
  const TileBox box2 = tile_bounds(act2, qx, qy, qz);
  const TileBox nb2 = grow(box2, reach + tile_eps(box2, reach));
  const TileCells cells2 = tile_cells(g, nb2, 0x7fffffff);
  if (stats && lane_id() == 0) {
    atomicAdd(&stats[5], (unsigned long long)cells2.total);
    if (!cells2.ok) atomicAdd(&stats[6], 1ull);
  }
  if (!cells2.ok) {
    if (act2) explored2 = 0.f;  // what the earlier rounds found beyond their radius proves nothing
    return;
  }
  CorrBest b = best;
  corr_gather_scan(g, cells2, nb2, nb, seen, cand, qx, qy, qz, b, scratch, nbm, seen_mid);
  if (act2) { best = b; explored2 = reach2; }
}

// The tiles of ONE pair (block bx of the pair's G): sp = the pair's source points, Tf the pose they are searched at (row-major 3 x 4),
// corr_p / qs_p the pair's rows of the correspondence and last-search arrays.  Shared by FAST_GICP's update_correspondences and by ICP's
// correspondence estimation (k_icp_correspond_tile below).
__device__ __forceinline__ void correspond_tiles_of_pair(const GridView& g, const float4* __restrict__ sp, int n, const float* Tf, int* __restrict__ corr_p,
                                                         float4* __restrict__ qs_p, float max_corr_d2, float r1, int G, int bx,
                                                         unsigned long long* __restrict__ stats, const int* __restrict__ order_p, float4* cand, int* scratch,
                                                         float r_mid = 0.f) {
  const int lane = lane_id();
  const float reach2 = search_reach2(g, max_corr_d2, true);
  const float reach = sqrtf(reach2);
  const bool one_round = !(r1 < reach);
  const float R1 = one_round ? reach : r1, R1_2 = one_round ? reach2 : r1 * r1;
  constexpr int kWavesPerBlock = kGicpBlock / kWave;
  for (int tile = bx * kWavesPerBlock + wave_id(); tile * kWave < n; tile += G * kWavesPerBlock) {
    int i = tile * kWave + lane;
    const bool valid = i < n;
    // the source's tiles in the order its k-NN search took them (cloud.hip, build_tile_order): compact patches, not runs of the Morton curve
    if (order_p && valid) i = order_p[i];
    const float4 p = sp[valid ? i : tile * kWave];
    const float qx = transform_row(Tf, p.x, p.y, p.z), qy = transform_row(Tf + 4, p.x, p.y, p.z), qz = transform_row(Tf + 8, p.x, p.y, p.z);
    const bool finite_q = valid && isfinite(qx) && isfinite(qy) && isfinite(qz);
    if (valid && !finite_q) corr_p[i] = -1;
    // a point whose last search left a margin it has not used up keeps its correspondence (far-field points mostly: nothing
    // within reach, and that stays so while the pose moves by decimetres)
    const float4 qs = qs_p[valid ? i : tile * kWave];
    const bool act = finite_q && !(sqrtf(dist2f(qx, qy, qz, qs.x, qs.y, qs.z)) < qs.w);
    if (!__ballot(act)) continue;
    CorrBest best{kCorrEmpty, kCorrEmpty, -1};
    float explored2 = 0.f;
    {
      // a patch across a jump of the Morton curve is mostly empty space between compact parts: its quarters go through the
      // rounds one after the other
      const TileBox box = tile_bounds(act, qx, qy, qz);
      const TileCells probe = tile_cells(g, grow(box, R1 + tile_eps(box, reach)), kCorrSplitPoints);
      const int parts = probe.ok ? 1 : 4;
      for (int part = 0; part < parts; part++) {
        const bool pact = act && (parts == 1 || (lane >> 4) == part);
        if (!__ballot(pact)) continue;
        corr_rounds(g, pact, qx, qy, qz, R1, R1_2, reach, reach2, one_round, cand, best, explored2, stats, scratch, r_mid);
      }
    }
    // what the keys could not decide (a tie or near-tie of the two smallest, a box that could not be scanned) is left to the
    // per-lane search of k_gicp_correspond, which runs next: NaN margin = "not searched"
    const bool near_tie = best.k2 != kCorrEmpty && ((best.k2 >> kCorrOrdBits) - (best.k1 >> kCorrOrdBits)) <= 1u;
    const bool left = act && (explored2 == 0.f || near_tie);
    if (stats) {
      const unsigned long long mleft = __ballot(left);
      if (lane == 0) atomicAdd(&stats[7], (unsigned long long)__popcll(mleft));
    }
    if (left) {
      corr_p[i] = -1;
      qs_p[i] = make_float4(qx, qy, qz, __uint_as_float(0x7fc00000u));
    } else if (act) {
      NnState stt{INFINITY, -1, INFINITY, explored2, true};
      if (best.k1 != kCorrEmpty) {
        const float4 w = g.pts[best.j1];
        stt.d1 = dist2f(qx, qy, qz, w.x, w.y, w.z);
        stt.j1 = best.j1;
        if (best.k2 != kCorrEmpty) stt.d2nd = __uint_as_float(((best.k2 >> kCorrOrdBits) - 1u) << kCorrOrdBits);  // lower edge, one bucket of slack
      }
      int j;
      float margin;
      finish_search(stt, max_corr_d2, true, j, margin);
      corr_p[i] = j;
      qs_p[i] = make_float4(qx, qy, qz, margin);
    }
  }
}

// (4 waves per SIMD = 126 VGPRs, no spills: 4.05 ms per launch at 1 024 pairs; unconstrained 135 VGPRs = 3 waves: 4.53 ms;
// 5 waves = 96 VGPRs + 48 B of scratch: 4.6 ms)
__global__ __launch_bounds__(kGicpBlock, 4) void k_gicp_correspond_tile(const float4* __restrict__ spts, const int* __restrict__ scount, size_t scap,
                                                                     const float4* __restrict__ tpts, const int* __restrict__ torig,
                                                                     const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta,
                                                                     size_t tcap, size_t tslots, const int* __restrict__ tgt_of_src,
                                                                     const GicpPairState* __restrict__ state, int* __restrict__ corr,
                                                                     float4* __restrict__ qsearch, float max_corr_d2, float r1, int G, int S,
                                                                     unsigned long long* __restrict__ stats, double margin_step,
                                                                     const int* __restrict__ tile_order) {
  __shared__ float4 cand_all[kGicpBlock / kWave][kCorrCand + 8];  // (+ 8: corr_scan reads ahead of the entries it uses)
  __shared__ int gscratch[kGicpBlock / kWave][2 * kWave];         // corr_gather_scan: prefix sums / range starts of a gather's cells
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const GicpPairState* st = state + pair;
  if (!st->active || st->last_step < margin_step) return;
  const int t = tgt_of_src[pair];
  const GridView g = grid_view(tpts, torig, ttable, tmeta, t, tcap, tslots);
  const int n = scount[pair];
  const Pose x0 = st->x0;
  float Tf[12];
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 3; c++) Tf[4 * r + c] = (float)x0.R[3 * r + c];
    Tf[4 * r + 3] = (float)x0.t[r];
  }
  const size_t so = (size_t)pair * scap;
  correspond_tiles_of_pair(g, spts + so, n, Tf, corr + so, qsearch + so, max_corr_d2, r1, G, bx, stats, tile_order ? tile_order + so : nullptr,
                           cand_all[wave_id()], gscratch[wave_id()]);
}

// pcl::IterativeClosestPoint's correspondence estimation by the same tiles (round 6).  `cur` holds the sources AS MOVED SO FAR, so the
// pose is the identity; a pair is searched by tiles while its last incremental fit still moved it by more than `min_move` (metres at a
// 50 m lever arm - every point then needs a new search anyway); afterwards the margins the searches left carry the correspondences
// (k_icp_step) and this kernel leaves at once.  What the tiles cannot decide (near-ties of their keys, patches whose box they cannot scan,
// points with nothing within r1 and a reach of tens of metres) is marked "not searched" and goes through k_icp_step's per-lane search.
__global__ __launch_bounds__(kGicpBlock, 4) void k_icp_correspond_tile(const float4* __restrict__ cur, size_t cpitch, const IcpPairState* __restrict__ state,
                                                                    const float4* __restrict__ tpts, const int* __restrict__ torig,
                                                                    const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta, size_t tcap,
                                                                    size_t tslots, const int* __restrict__ tgt_of_src, int* __restrict__ corr,
                                                                    float4* __restrict__ qsearch, float max_corr_d2, float r1, int G, int S, float min_move,
                                                                    const int* __restrict__ tile_order, size_t opitch, float r_mid, float r1_sigma) {
  __shared__ float4 cand_all[kGicpBlock / kWave][kCorrCand + 8];
  __shared__ int gscratch[kGicpBlock / kWave][2 * kWave];
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const IcpPairState& st = state[pair];
  if (!st.active) return;
  if (st.iterations > 0) {  // (the first search of a pair is always by tiles)
    const float tr = sqrtf(st.Tinc[3] * st.Tinc[3] + st.Tinc[7] * st.Tinc[7] + st.Tinc[11] * st.Tinc[11]);
    const float c = 0.5f * (st.Tinc[0] + st.Tinc[5] + st.Tinc[10] - 1.f);
    const float ang = sqrtf(fmaxf(2.f * (1.f - c), 0.f));  // ~ the rotation angle of the last fit
    if (!(tr + 50.f * ang >= min_move)) return;
  }
  const GridView g = grid_view(tpts, torig, ttable, tmeta, tgt_of_src[pair], tcap, tslots);
  // the first round's radius follows the pair: r1_sigma times the RMS correspondence distance of its last iteration, between half a base
  // cell and r1 - as the pair closes in, the box a patch scans shrinks with it (who lies farther goes through the middle round)
  if (r1_sigma > 0.f && st.iterations > 0 && st.mse > 0.0) r1 = fminf(r1, fmaxf(r1_sigma * sqrtf((float)st.mse), 0.5f * g.m.cell));
  const float I[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  const size_t so = (size_t)pair * cpitch;
  correspond_tiles_of_pair(g, cur + so, st.n, I, corr + so, qsearch + so, max_corr_d2, r1, G, bx, nullptr,
                           tile_order ? tile_order + (size_t)pair * opitch : nullptr, cand_all[wave_id()], gscratch[wave_id()], r_mid);
}
// The same tiles for an engine that keeps its sources where the search should find them (`cur`, identity pose) and says per pair whether it
// wants a search now (`jobs[pair]` = the pair's point count, 0 = not in this launch): pclomp's BFGS GICP at the start of an outer iteration
// (gicp_bfgs.hip).  A point inside the margin of its last search keeps its correspondent here too.
__global__ __launch_bounds__(kGicpBlock, 4) void k_moved_correspond_tile(const float4* __restrict__ cur, size_t cpitch, const int* __restrict__ jobs,
                                                                      const float4* __restrict__ tpts, const int* __restrict__ torig,
                                                                      const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta, size_t tcap,
                                                                      size_t tslots, const int* __restrict__ tgt_of_src, int* __restrict__ corr,
                                                                      float4* __restrict__ qsearch, float max_corr_d2, float r1, int G, int S,
                                                                      const int* __restrict__ tile_order, size_t opitch, float r_mid) {
  __shared__ float4 cand_all[kGicpBlock / kWave][kCorrCand + 8];
  __shared__ int gscratch[kGicpBlock / kWave][2 * kWave];
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const int n = jobs[pair];
  if (n <= 0) return;
  const GridView g = grid_view(tpts, torig, ttable, tmeta, tgt_of_src[pair], tcap, tslots);
  const float I[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  const size_t so = (size_t)pair * cpitch;
  correspond_tiles_of_pair(g, cur + so, n, I, corr + so, qsearch + so, max_corr_d2, r1, G, bx, nullptr,
                           tile_order ? tile_order + (size_t)pair * opitch : nullptr, cand_all[wave_id()], gscratch[wave_id()], r_mid);
}
void moved_correspond_tiles(hipStream_t s, const float4* cur, size_t cpitch, const int* jobs, const CloudSet& tgt, const int* tgt_of_src, int* corr,
                            float4* qsearch, float max_corr_d2, float r1, float r_mid, int S, const int* tile_order, size_t opitch) {
  const int G = std::max(1, std::min(std::max(4, std::min(64, 16384 / std::max(S, 1))), div_up((long)cpitch, kGicpBlock)));
  TimedLaunch tl(KF_CORRESPOND_TILE, s);
  hipLaunchKernelGGL(k_moved_correspond_tile, dim3(xcd_grid(G, S)), dim3(kGicpBlock), 0, s, cur, cpitch, jobs, (const float4*)tgt.pts.p, (const int*)tgt.orig.p,
                     (const HashSlot*)tgt.table.p, (const GridMeta*)tgt.meta.p, tgt.cap, tgt.slots_cap, tgt_of_src, corr, qsearch, max_corr_d2, r1, G, S, tile_order,
                     opitch, r_mid);
  MI_LAUNCH_CHECK();
}
void icp_correspond_tiles(hipStream_t s, const float4* cur, size_t cpitch, const IcpPairState* state, const CloudSet& tgt, const int* tgt_of_src, int* corr,
                          float4* qsearch, float max_corr_d2, float r1, int S, float min_move, const int* tile_order, size_t opitch, float r_mid, float r1_sigma) {
  const int G = std::max(1, std::min(std::max(4, std::min(64, 16384 / std::max(S, 1))), div_up((long)cpitch, kGicpBlock)));
  TimedLaunch tl(KF_CORRESPOND_TILE, s);
  hipLaunchKernelGGL(k_icp_correspond_tile, dim3(xcd_grid(G, S)), dim3(kGicpBlock), 0, s, cur, cpitch, state, (const float4*)tgt.pts.p, (const int*)tgt.orig.p,
                     (const HashSlot*)tgt.table.p, (const GridMeta*)tgt.meta.p, tgt.cap, tgt.slots_cap, tgt_of_src, corr, qsearch, max_corr_d2, r1, G, S, min_move, tile_order, opitch, r_mid, r1_sigma);
  MI_LAUNCH_CHECK();
}

// F4 linearize given the correspondences: Mahalanobis + normal equations, a pure streaming pass — per source point it
// reads src pt 16 + src cov 24(32) + tgt pt 16 + tgt cov 24(32) B (the 80 algorithmic bytes of SURVEY.md §8d) and the
// 4-byte correspondence and accumulates the sums below (the Mahalanobis matrices stay in registers: nothing reads them back).
// `model` (LM): the pass also accumulates the 72 sums of the trial-cost model (see kModelSlot) - 63 more fp64 accumulators per lane
// would halve the kernel's occupancy, so most of them are an outer product per point on the matrix cores: the lanes of a wave leave the
// 8 + 8 factors of their points in LDS, and 8 v_mfma_f64_16x16x4_f64 (8 points each: A = factors of the Mahalanobis side, B = of the
// transformed point) add the 8 x 8 products of all 64 points to an accumulator tile of 4 doubles per lane; 12 sums stay in registers.
// `safe_mask` (LM, pairs that track search margins): the pass marks the points whose margin survives a move of the pose by the slack
// k_gicp_step has budgeted for this iteration's trials (slack_rot / slack_trans of the pair's state; k_gicp_step validates the
// budget against the trials it then makes) - the next k_gicp_correspond reads those bits instead of the points.
typedef double mfma_acc_t __attribute__((ext_vector_type(4)));
#ifndef LIN_OCC
#define LIN_OCC 2  // resident blocks per CU (= waves per SIMD) the register allocation aims at
#endif
template <int kLinPointsPerThread>
__global__ __launch_bounds__(kGicpBlock, LIN_OCC) void k_gicp_linearize(const float4* __restrict__ spts, const Cov6* __restrict__ scov,
                                                               const int* __restrict__ scount, size_t scap, const float4* __restrict__ tpts,
                                                               const Cov6* __restrict__ tcov, size_t tcap, const int* __restrict__ tgt_of_src,
                                                               const GicpPairState* __restrict__ state, const int* __restrict__ corr,
                                                               Cov6* __restrict__ mahal, double* __restrict__ partials, int G, int S,
                                                               const float4* __restrict__ qsearch, unsigned long long* __restrict__ safe_mask,
                                                               size_t mask_words, double margin_step) {
  constexpr int kLinTile = kGicpBlock * kLinPointsPerThread;
  constexpr int kWaves = kGicpBlock / kWave;
  constexpr int kFeat = 2 * kWave * kFeatStride;
  __shared__ double smem[kWaves * kNAcc];
  __shared__ double feat[kWaves][kFeat];  // per wave: the factors of its 64 points; at the end: its accumulator tile
  int pair, bx;
  if (!xcd_item_chunk(G, S, pair, bx)) return;
  const GicpPairState* st = state + pair;
  if (!st->active) return;
  const int t = tgt_of_src[pair];
  const float4* tp = tpts + (size_t)t * tcap;
  const Cov6* tc = tcov + (size_t)t * tcap;
  const int n = scount[pair];
  const Pose x0 = st->x0;
  float Rf[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Rf[k] = (float)x0.R[k];
  const size_t so = (size_t)pair * scap;
  const float slack_rot = st->slack_rot, slack_trans = st->slack_trans;
  const bool mark = safe_mask != nullptr && st->last_step < margin_step && slack_rot >= 0.f;
  float Tf0[12];
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 3; c++) Tf0[4 * r + c] = Rf[3 * r + c];
    Tf0[4 * r + 3] = (float)x0.t[r];
  }
  double acc[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; k++) acc[k] = 0.0;
  mfma_acc_t macc = {0.0, 0.0, 0.0, 0.0};
  const int lane = lane_id(), wv = wave_id();
  double* featA = feat[wv] + lane * kFeatStride;
  double* featB = feat[wv] + kWave * kFeatStride + lane * kFeatStride;
  // Software pipeline over the block's tiles: the correspondence index, the source point and its covariance of the NEXT
  // tile are loaded while the current tile is computed, so the dependent gathers (target point / covariance at corr[i])
  // of a tile are issued as soon as its loop iteration starts — one exposed memory latency per tile instead of two.
  // (A third stage - the index two tiles ahead, the gathers one tile ahead - was built and measured in round 4: 176 VGPRs, two waves per
  // SIMD, the same 16.4 ms per step; bounded to three waves it spills and loses 6 %.  The kernel is bound by bytes, not by that latency.)
  struct Pre {
    int j;
    float4 p, qs;
    Cov6 c;
  };
  auto preload = [&](int base, int u) -> Pre {
    Pre r;
    const int i = base + u * kGicpBlock + threadIdx.x;
    r.j = -1;
    if (i < n) {
      r.j = corr[so + i];
      r.p = spts[so + i];
      r.c = scov[so + i];
      if (mark) r.qs = qsearch[so + i];
    }
    return r;
  };
  Pre nxt[kLinPointsPerThread];
#pragma unroll
  for (int u = 0; u < kLinPointsPerThread; u++) nxt[u] = preload(bx * kLinTile, u);
  for (int base = bx * kLinTile; base < n; base += G * kLinTile) {
    Pre cur[kLinPointsPerThread];
    float4 tq[kLinPointsPerThread];
    Cov6 tcv[kLinPointsPerThread];
#pragma unroll
    for (int u = 0; u < kLinPointsPerThread; u++) {
      cur[u] = nxt[u];
      const int j = max(cur[u].j, 0);
      tq[u] = tp[j];
      tcv[u] = tc[j];
    }
#pragma unroll
    for (int u = 0; u < kLinPointsPerThread; u++) nxt[u] = preload(base + G * kLinTile, u);
#pragma unroll
    for (int u = 0; u < kLinPointsPerThread; u++) {
      const int i = base + u * kGicpBlock + threadIdx.x;
      const int j = cur[u].j;
      const float4 p = cur[u].p;
      if (mark) {
        // Which points keep their correspondence wherever this iteration's step ends?  The last search of a point left a margin
        // (k_gicp_correspond); a point at x0 p moves by at most slack_rot |x0 p| + slack_trans under any pose within the budget, so
        // "moved so far + that < margin" settles it for the pose the next iteration linearises at - one bit per point, and the next
        // k_gicp_correspond reads the bits (1/128 of the bytes of its margin check) instead of the points and their stored queries.
        bool safe = i >= n;  // (past the end: nothing to look at)
        if (i < n) {
          const float4 qs = cur[u].qs;
          const float qx = transform_row(Tf0, p.x, p.y, p.z), qy = transform_row(Tf0 + 4, p.x, p.y, p.z), qz = transform_row(Tf0 + 8, p.x, p.y, p.z);
          // (no square roots: |x0 p| <= its 1-norm, and "moved < budget" is compared squared)
          const float reach = fabsf(qx) + fabsf(qy) + fabsf(qz);
          const float budget = qs.w - ((slack_rot * reach + slack_trans) * (1.f + 1e-5f) + 2.0e-5f);
          safe = isfinite(reach) && budget > 0.f && dist2f(qx, qy, qz, qs.x, qs.y, qs.z) < budget * budget * (1.f - 1e-6f);
        }
        const unsigned long long bits = __ballot(safe);
        // (the last tile of a pair may reach past the pair's row of words when the capacity is not a multiple of the tile: those words
        // would be the next pair's first ones, marked "safe" behind its back)
        const size_t word = (size_t)((base + u * kGicpBlock) / kWave + wv);
        if (lane == 0 && word < mask_words) safe_mask[(size_t)pair * mask_words + word] = bits;
      }
      const bool has = j >= 0;  // (not past the end, and a correspondence)
      if (has) {
        const Sym3<float> CA = cov6_unpack(cur[u].c);
        const Sym3<float> CB = cov6_unpack(tcv[u]);
        Sym3<float> RCR = sym3_rotate(Rf, CA);
        RCR.xx += CB.xx; RCR.xy += CB.xy; RCR.xz += CB.xz; RCR.yy += CB.yy; RCR.yz += CB.yz; RCR.zz += CB.zz;
        const Sym3<float> M = sym3_inverse(RCR);
        if (mahal) mahal[so + i] = cov6_pack(M.xx, M.xy, M.xz, M.yy, M.yz, M.zz);  // (the parity probe's pass over the points reads them; the optimisation does not)
        float ax, ay, az, ex, ey, ez;
        double dax, day, daz, dex, dey, dez;
        residual64(x0, p, tq[u], dax, day, daz, ax, ay, az, ex, ey, ez, dex, dey, dez);
        {
#pragma clang fp contract(fast)
          const double mxx = (double)M.xx, mxy = (double)M.xy, mxz = (double)M.xz, myy = (double)M.yy, myz = (double)M.yz, mzz = (double)M.zz;
          const double md0x = mxx * dex + mxy * dey + mxz * dez;  // M d0
          const double md0y = mxy * dex + myy * dey + myz * dez;
          const double md0z = mxz * dex + myz * dey + mzz * dez;
          acc[0] += mxx; acc[1] += mxy; acc[2] += mxz; acc[3] += myy; acc[4] += myz; acc[5] += mzz;
          acc[kSumCost] += dex * md0x + dey * md0y + dez * md0z;  // d0^T M d0 (cost64's operations)
          acc[kSumCount] += 1.0;
          // (straight to LDS: the factors kept in registers until the whole wave is here cost the kernel a wave per SIMD)
          featA[0] = mxx; featA[1] = mxy; featA[2] = mxz; featA[3] = myy; featA[4] = myz; featA[5] = mzz; featA[6] = md0x; featA[7] = md0y;
          featB[0] = dax * dax; featB[1] = dax * day; featB[2] = dax * daz; featB[3] = day * day; featB[4] = day * daz;
          featB[5] = dax; featB[6] = day; featB[7] = daz;
          const double azaz = daz * daz;
          acc[kNLin + 0] += mxx * azaz; acc[kNLin + 1] += mxy * azaz; acc[kNLin + 2] += mxz * azaz;
          acc[kNLin + 3] += myy * azaz; acc[kNLin + 4] += myz * azaz; acc[kNLin + 5] += mzz * azaz;
          acc[kNLin + 6] += md0x; acc[kNLin + 7] += md0y; acc[kNLin + 8] += md0z;
          acc[kNLin + 9] += md0z * dax; acc[kNLin + 10] += md0z * day; acc[kNLin + 11] += md0z * daz;
        }
      } else {
#pragma unroll
        for (int k = 0; k < kMfmaDim; k++) { featA[k] = 0.0; featB[k] = 0.0; }
      }
      if (__ballot(has)) {  // (wave-uniform)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the rows are read by other lanes
        // product step s takes points 8 s .. 8 s + 7: lane (r = lane & 15, k = lane >> 4) holds factor r & 7 of point 8 s + k (r < 8:
        // rows / columns 0-7 of the tile) or of point 8 s + 4 + k (rows / columns 8-15)
        const int r = lane & 15, kq = lane >> 4;
        const double* rowA = feat[wv] + (kq + 4 * (r >> 3)) * kFeatStride + (r & 7);
        const double* rowB = rowA + kWave * kFeatStride;
#pragma unroll 4
        for (int step = 0; step < kWave / 8; step++) {
          const double a = rowA[step * 8 * kFeatStride], b = rowB[step * 8 * kFeatStride];
          macc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, macc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next tile overwrites the rows
      }
    }
  }
  static_assert(kFeat >= kWave * kBlockSumRow, "the wave's factor rows double as the transpose buffer of the block sum");
  {
    const double total = block_sum_columns<kNAcc, kWaves>(acc, feat[wv], smem);
    if (threadIdx.x < kNAcc) partials[((size_t)pair * G + bx) * kPartStride + threadIdx.x] = total;
  }
  {
    // the four waves' accumulator tiles -> one 8 x 8: element v of lane l is row (l >> 4) + 4 v, column l & 15 (the f64 MFMA's own map);
    // the two diagonal blocks hold the two halves of the points
#pragma unroll
    for (int v = 0; v < 4; v++) feat[wv][((lane >> 4) + 4 * v) * 16 + (lane & 15)] = macc[v];
    __syncthreads();
    if (threadIdx.x < kNMfma) {
      const int i = threadIdx.x / kMfmaDim, j = threadIdx.x % kMfmaDim;
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < kWaves; w++) sum += feat[w][i * 16 + j] + feat[w][(i + 8) * 16 + j + 8];
      partials[((size_t)pair * G + bx) * kPartStride + kModelSlot + threadIdx.x] = sum;
    }
  }
}

__device__ __forceinline__ void lm_make_trial(const double* H, const double* b, const Pose& x0, double lambda, LmTrial& tr) {
  double A[36], nb[6];
  for (int i = 0; i < 36; i++) A[i] = H[i];
  for (int i = 0; i < 6; i++) { A[7 * i] += lambda; nb[i] = -b[i]; }
  ldlt6_solve(A, nb, tr.d);
  tr.delta = se3_exp(tr.d);
  tr.xi = pose_compose(tr.delta, x0);
  tr.lambda = lambda;
}

// After a step_optimize completed: iteration bookkeeping + convergence (F6).  State: GicpPairState itself, or the copy of these
// fields k_gicp_step keeps in registers (StepFields).
struct StepFields {
  int iter, active, converged, first_conv, lm_failed;
  double last_step;
  Pose delta;
};
template <typename State>
__device__ __forceinline__ void finish_step(State& s, const GicpDevParams& P, bool ok) {
  s.iter += 1;
  if (!ok) {  // "lm not converged!!" -> break (benchmark mode keeps iterating: fixed work per pair)
    s.lm_failed = 1;
    if (!P.force_iterations || s.iter >= P.max_iterations) s.active = 0;
    return;
  }
  const bool conv = is_converged(s.delta, P.rotation_epsilon, P.transformation_epsilon);
  s.converged = conv ? 1 : 0;
  if (conv && s.first_conv < 0) s.first_conv = s.iter;
  {
    double rmax = 0, tmax = 0;
    for (int k = 0; k < 9; k++) rmax = fmax(rmax, fabs(s.delta.R[k] - ((k % 4 == 0) ? 1.0 : 0.0)));
    for (int k = 0; k < 3; k++) tmax = fmax(tmax, fabs(s.delta.t[k]));
    s.last_step = tmax + 50.0 * rmax;
  }
  if ((conv && !P.force_iterations) || s.iter >= P.max_iterations) s.active = 0;
}

// Sum of one value over the G block partials of a pair, in block order (bitwise reproducible whatever the machine does).  ALL the
// loads go out before the first add - up to 32 at a time: a pair's 6 (a big batch) or 32 (a few pairs) partials are one memory
// latency, where round 3's 8-at-a-time loop with a one-by-one tail took four (a lone pair) or six (the bench's G = 6: all tail).
__device__ __forceinline__ double sum_partials(const double* __restrict__ p, int G) {
  double s = 0;
  for (int g0 = 0; g0 < G; g0 += 32) {
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; u++) v[u] = (g0 + u < G) ? p[(size_t)(g0 + u) * kPartStride] : 0.0;
#pragma unroll
    for (int u = 0; u < 32; u++)
      if (g0 + u < G) s += v[u];
  }
  return s;
}

// Entry idx = row * 10 + column of the 9 x 10 model matrix (see kModelSlot) from the summed partials of a pair.
MI_HD double model_entry(const double* sums, int idx) {
  const int r = idx / kModelCols, c = idx % kModelCols;
  const double* mf = sums + kModelSlot;  // the 8 x 8 tile
  if (r < 6) {
    if (c < 5) return mf[r * kMfmaDim + c];
    if (c == 5) return sums[kNLin + r];
    if (c < 9) return mf[r * kMfmaDim + c - 1];
    return sums[r];  // sum M = H_tt
  }
  if (c < 6) return 0.0;  // (never read)
  if (r < 8) return c < 9 ? mf[r * kMfmaDim + c - 1] : sums[kNLin + 6 + (r - 6)];
  return c < 9 ? sums[kNLin + 9 + (c - 6)] : sums[kNLin + 8];
}

// index of (r, s) among xx xy xz yy yz zz, and of (c, e), c, e in {x, y, z, 1}, among the ten columns of the model
MI_HD int model_sym3(int r, int s2) {
  const int lo = r < s2 ? r : s2, hi = r < s2 ? s2 : r;
  return lo * 3 - lo * (lo - 1) / 2 + (hi - lo);
}
MI_HD int model_sym4(int c, int e) { return (c < 3 && e < 3) ? model_sym3(c, e) : ((c == 3 && e == 3) ? 9 : 6 + (c < e ? c : e)); }

// The normal equations of FastGICP::linearize from the model m (9 x 10): J = [skew(a) | -I] per point (rows/cols 0-2 rotation, 3-5
// translation), H = sum J^T M J, b = sum J^T M d0.  With skew(a)_ki = -eps_kip a_p:
//   H_ij (rot, rot)     = sum_{k,l,p,q} eps_kip eps_ljq  sum(M_kl a_p a_q)
//   H_i,3+j (rot, trans) = sum_{k,p} eps_kip  sum(M_kj a_p)            H_3+i,3+j = sum(M_ij)
//   b_i (rot) = - sum_{k,p} eps_kip  sum((M d0)_k a_p)                 b_3+i = - sum((M d0)_i)
// Entry e < 36: H[e] (row-major 6 x 6); 36 <= e < 42: b[e - 36].
MI_HD double model_normal_entry(const double* m, int e) {
  auto eps = [](int k, int i, int p) { return ((k - i) * (i - p) * (p - k)) / 2; };  // Levi-Civita on {0, 1, 2}
  auto Q = [&](int k, int l, int p, int q) { return m[model_sym3(k, l) * kModelCols + model_sym4(p, q)]; };
  auto L = [&](int k, int p) { return m[(6 + k) * kModelCols + model_sym4(p, 3)]; };
  if (e >= 36) {
    const int i = e - 36;
    if (i >= 3) return -L(i - 3, 3);
    double v = 0.0;
    for (int k = 0; k < 3; k++)
      for (int p = 0; p < 3; p++) {
        const int s1 = eps(k, i, p);
        if (s1) v -= s1 * L(k, p);
      }
    return v;
  }
  const int i = e / 6, j = e % 6;
  if (i >= 3 && j >= 3) return Q(i - 3, j - 3, 3, 3);
  if (i < 3 && j < 3) {
    double v = 0.0;
    for (int k = 0; k < 3; k++)
      for (int p = 0; p < 3; p++) {
        const int s1 = eps(k, i, p);
        if (!s1) continue;
        for (int l = 0; l < 3; l++)
          for (int q = 0; q < 3; q++) {
            const int s2 = eps(l, j, q);
            if (s2) v += (s1 * s2) * Q(k, l, p, q);
          }
      }
    return v;
  }
  const int r = i < 3 ? i : j, t = (i < 3 ? j : i) - 3;  // (rotation index, translation index): H is symmetric
  double v = 0.0;
  for (int k = 0; k < 3; k++)
    for (int p = 0; p < 3; p++) {
      const int s1 = eps(k, r, p);
      if (s1) v += s1 * Q(k, t, p, 3);
    }
  return v;
}

// The decrease y0 - e_t of the cost at the trial pose delta x0, from the sums of k_gicp_linearize (see kModelSlot): m = the 9 x 10
// tile, rows M_xx M_xy M_xz M_yy M_yz M_zz (M d0)_xyz, columns ax ax, ax ay, ax az, ay ay, ay az, az az, ax, ay, az, 1.
MI_HD double model_decrease(const Pose& delta, const double* m) {
  double W[3][4];
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 3; c++) W[r][c] = delta.R[3 * r + c] - ((r == c) ? 1.0 : 0.0);
    W[r][3] = delta.t[r];
  }
  double lin = 0.0, quad = 0.0;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) lin += W[r][c] * m[(6 + r) * kModelCols + model_sym4(c, 3)];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s2 = 0; s2 < 3; s2++) {
      double q = 0.0;
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) q += W[r][c] * W[s2][e] * m[model_sym3(r, s2) * kModelCols + model_sym4(c, e)];
      quad += q;
    }
  return 2.0 * lin - quad;
}

// One LsqRegistration::step_lm per pair, from the sums of k_gicp_linearize: the normal equations, every trial of the lambda sequence
// (lane t: LDL^T solve, se3_exp, trial pose, its cost through the model), then lane 0 walks the trials as step_lm does - accept /
// reject by rho, lambda update, convergence test, iteration bookkeeping - no host round trip and no pass over the points.
// sums_only: the normal equations and y0 go to the state and nothing else happens (gicp_linearize_once).
// margin_step >= 0: the search-margin bits k_gicp_linearize wrote for this pair hold if the trials stayed inside the slack it had
// budgeted; the budget of the next iteration is slack_factor x what these trials took.
__global__ __launch_bounds__(64) void k_gicp_step(GicpPairState* __restrict__ state, const double* __restrict__ partials, int G, GicpDevParams P,
                                                  int sums_only, double margin_step, float slack_factor) {
  __shared__ double sums[kPartStride], mdl[kNModel];
  __shared__ double sH[36], sb[6], slambda;
  __shared__ Pose sx0;
  __shared__ LmTrial s_tr[kMaxTrials];
  __shared__ double dec[kMaxTrials];  // cost decrease y0 - e_t of every trial
  __shared__ int s_accepted;
  const int pair = blockIdx.x;
  GicpPairState* st = state + pair;
  const int l = threadIdx.x;
  // everything the kernel reads of the state, in flight together with the partial sums (the decision below used to find these fields
  // one dependent load at a time)
  const int ld_active = st->active, ld_iter = st->iter, ld_converged = st->converged, ld_first_conv = st->first_conv, ld_lm_failed = st->lm_failed;
  const double ld_lambda = st->lambda, ld_last_step = st->last_step;
  const float ld_slack_rot = st->slack_rot, ld_slack_trans = st->slack_trans;
  const double ld_x0 = (l < 9) ? st->x0.R[l] : st->x0.t[min(l, 11) - 9];
  if (!ld_active) return;
  const double* mine = partials + (size_t)pair * G * kPartStride;
  static_assert(kPartStride <= 2 * kWave && kNAcc <= kModelSlot && kModelSlot + kNMfma <= kPartStride, "two sums per lane cover a pair's partials");
  sums[l] = (l < kNAcc || l >= kModelSlot) ? sum_partials(mine + l, G) : 0.0;
  if (kWave + l < kPartStride) sums[kWave + l] = sum_partials(mine + kWave + l, G);
  __syncthreads();
  mdl[l] = model_entry(sums, l);
  if (kWave + l < kNModel) mdl[kWave + l] = model_entry(sums, kWave + l);
  __syncthreads();
  if (l < 36) sH[l] = model_normal_entry(mdl, l);
  else if (l < 42) sb[l - 36] = model_normal_entry(mdl, l);
  __syncthreads();
  if (P.optimizer == MI355_OPT_GN) {
    if (l == 0) {
      st->mask_valid = 0;
      GicpPairState s = *st;
      for (int i = 0; i < 36; i++) s.H[i] = sH[i];
      for (int i = 0; i < 6; i++) s.b[i] = sb[i];
      s.y0 = sums[kSumCost];
      s.n_inliers = (int)(sums[kSumCount] + 0.5);
      if (!sums_only) {
        double nb[6];
        for (int i = 0; i < 6; i++) nb[i] = -s.b[i];
        ldlt6_solve(s.H, nb, s.d);
        s.delta = se3_exp(s.d);
        s.x0 = pose_compose(s.delta, s.x0);
        for (int i = 0; i < 36; i++) s.final_H[i] = s.H[i];
        s.last_error = s.y0;
        finish_step(s, P, true);
      }
      *st = s;
    }
    return;
  }
  // LM: only the fields that change are touched (a copy of the 1 KB state through one lane's scratch cost 15 of the 30 us), and what
  // the decision reads of the state came in with the first loads of the kernel (ld_*): the walk itself waits for no memory.
  if (l == 0) {
    double lam = ld_lambda;
    if (lam < 0.0) {
      double mx = 0;
      for (int i = 0; i < 6; i++) mx = fmax(mx, fabs(sH[7 * i]));
      lam = P.lm_init_lambda_factor * mx;
    }
    slambda = lam;
    st->y0 = sums[kSumCost];
    st->n_inliers = (int)(sums[kSumCount] + 0.5);
    s_accepted = 0;
  }
  if (l < 12) ((l < 9) ? sx0.R[l] : sx0.t[l - 9]) = ld_x0;
  __syncthreads();
  if (l < 36) st->H[l] = sH[l];
  if (l < 6) st->b[l] = sb[l];
  if (sums_only) return;
  float slack_r = 0.f, slack_t = 0.f;
  if (l < P.lm_max_iterations) {
    // lambda of trial l: step_lm multiplies by nu = 2, 4, 8, ... after each rejection
    double lam = slambda, nu = 2.0;
    for (int i = 0; i < l; i++) { lam = nu * lam; nu = 2 * nu; }
    LmTrial tr;
    lm_make_trial(sH, sb, sx0, lam, tr);
    s_tr[l] = tr;
    dec[l] = model_decrease(tr.delta, mdl);
    // how far a point can move between x0 and this trial pose: x_t p - x0 p = (dR - I)(x0 p) + dt, so at most
    // ||dR - I||_F |x0 p| + |dt|
    double fr = 0, ft = 0;
    for (int k = 0; k < 9; k++) { const double v = tr.delta.R[k] - ((k % 4 == 0) ? 1.0 : 0.0); fr += v * v; }
    for (int k = 0; k < 3; k++) ft += tr.delta.t[k] * tr.delta.t[k];
    slack_r = (float)(sqrt(fr) * (1.0 + 1e-6)) + 1e-12f;
    slack_t = (float)(sqrt(ft) * (1.0 + 1e-6)) + 1e-12f;
  }
  slack_r = wave_reduce64(slack_r, [](float a, float b) { return fmaxf(a, b); });
  slack_t = wave_reduce64(slack_t, [](float a, float b) { return fmaxf(a, b); });
  __syncthreads();
  if (l == 0) {
    GicpPairState& s = *st;
    // the bits of this iteration's linearize pass (written only for a pair that tracked margins, under the budget in the state)
    const bool marked = margin_step >= 0.0 && ld_last_step < margin_step && ld_slack_rot >= 0.f;
    s.mask_valid = (marked && slack_r <= ld_slack_rot && slack_t <= ld_slack_trans) ? 1 : 0;
    s.slack_rot = slack_factor * slack_r + 1e-9f;
    s.slack_trans = slack_factor * slack_t + 1e-7f;
    StepFields f;
    f.iter = ld_iter; f.active = 1; f.converged = ld_converged; f.first_conv = ld_first_conv; f.lm_failed = ld_lm_failed; f.last_step = ld_last_step;
    bool finished = false, moved = false;
    double lam_out = slambda, lam_next = slambda, nu = 2.0;
    int t_last = 0;
    for (int t = 0; t < P.lm_max_iterations && !finished; t++) {
      const LmTrial& tr = s_tr[t];
      double denom = 0;
      for (int i = 0; i < 6; i++) denom += tr.d[i] * (tr.lambda * tr.d[i] - sb[i]);
      const double rho = dec[t] / denom;  // (y0 - yi) / denom
      t_last = t;
      if (rho < 0) {
        if (!P.force_iterations && is_converged(tr.delta, P.rotation_epsilon, P.transformation_epsilon)) {
          f.delta = tr.delta;  // step_lm returns true without moving x0 (and without touching lambda)
          s.delta = tr.delta;
          lam_out = tr.lambda;
          finish_step(f, P, true);
          finished = true;
        } else {
          lam_next = nu * tr.lambda;
          nu = 2 * nu;
        }
      } else {
        s.x0 = tr.xi;
        s.delta = tr.delta;
        f.delta = tr.delta;
        for (int i = 0; i < 6; i++) s.d[i] = tr.d[i];
        const double c = 2 * rho - 1;
        lam_out = tr.lambda * fmax(1.0 / 3.0, 1 - c * c * c);
        s_accepted = 1;  // final_H = H: by 36 lanes, below
        s.last_error = sums[kSumCost] - dec[t];
        s.yi = s.last_error;
        finish_step(f, P, true);
        finished = true;
        moved = true;
      }
    }
    if (!finished) {  // every trial rejected: "lm not converged!!"
      lam_out = lam_next;
      finish_step(f, P, false);
    }
    (void)moved;
    s.trial = t_last;
    s.lambda = lam_out;
    s.iter = f.iter;
    s.active = f.active;
    s.converged = f.converged;
    s.first_conv = f.first_conv;
    s.lm_failed = f.lm_failed;
    s.last_step = f.last_step;
  }
  __syncthreads();
  if (s_accepted && l < 36) st->final_H[l] = sH[l];
}

__device__ __forceinline__ double point_error(const Pose& x, const float4& p, const float4& q, const Cov6& m) {
  float ax, ay, az, ex, ey, ez;
  double dex, dey, dez;
  residual(x, p, q, ax, ay, az, ex, ey, ez, dex, dey, dez);
  return cost64(dex, dey, dez, m.v[0], m.v[1], m.v[2], m.v[3], m.v[4], m.v[5]);
}

// F5 compute_error as the reference writes it - a pass over the points at pose x with the frozen correspondences and Mahalanobis
// matrices of the last linearisation.  The optimisation prices its trials through the model (k_gicp_step); this kernel is the PROBE
// the parity tests compare that model with (gicp_cost_probe), for pair 0 of a registration object.
__global__ __launch_bounds__(kGicpBlock) void k_gicp_cost_probe(const float4* __restrict__ spts, const int* __restrict__ scount, const float4* __restrict__ tp,
                                                                const int* __restrict__ corr, const Cov6* __restrict__ mahal, Pose x,
                                                                double* __restrict__ out) {
  __shared__ double smem[4];
  const int n = scount[0];
  double acc[1] = {0.0};
  for (int i = blockIdx.x * kGicpBlock + threadIdx.x; i < n; i += gridDim.x * kGicpBlock) {
    const int j = corr[i];
    if (j >= 0) acc[0] += point_error(x, spts[i], tp[j], mahal[i]);
  }
  block_sum<1, 4>(acc, smem);
  if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}

__global__ void k_count_active(const GicpPairState* __restrict__ st, int S, int* __restrict__ n_active) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int p = threadIdx.x; p < S; p += blockDim.x) c += st[p].active ? 1 : 0;
  atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) *n_active = cnt;
}

static GicpDevParams dev_params(const mi355_gicp_params& p) {
  GicpDevParams P;
  P.max_iterations = p.max_iterations;
  P.optimizer = p.optimizer;
  if (p.lm_max_iterations < 1 || p.lm_max_iterations > kMaxTrials) throw ArgError(MI355_EINVAL, "lm_max_iterations must be in [1, 16]");
  P.lm_max_iterations = p.lm_max_iterations;
  P.force_iterations = p.force_iterations;
  P.transformation_epsilon = p.transformation_epsilon;
  P.rotation_epsilon = p.rotation_epsilon;
  P.lm_init_lambda_factor = p.lm_init_lambda_factor;
  const double m2 = p.max_correspondence_distance * p.max_correspondence_distance;
  P.max_corr_d2 = m2 > 3.0e38 ? INFINITY : (float)m2;
  return P;
}

// search: -1 = by iteration and batch size; gicp_linearize_once forces 0 = per-lane walks only, 1 = tiles (+ per-lane leftovers),
// 2 = lane groups
static void launch_linearize(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const GicpDevParams& P, int iteration,
                             int search = -1) {
  auto kc = gb.ppt == 1 ? k_gicp_correspond<1> : k_gicp_correspond<4>;
  const int lin_ppt = tuning().lin_ppt;
  auto kl = lin_ppt == 1 ? k_gicp_linearize<1> : (lin_ppt == 2 ? k_gicp_linearize<2> : k_gicp_linearize<4>);
  {
    const Tuning& tn = tuning();
    // a handful of pairs cannot fill the chip with tiles (their few hundred waves would each wait out their own gathers, and a
    // lone pair is usually a scan against a dense multi-key-frame sub-map): they get a group of lanes per query instead
    const bool tiles = search >= 0 ? search == 1 : (iteration < tn.corr_tiles && gb.S_geometry >= 8);
    const bool groups = search >= 0 ? search == 2 : (gb.S_geometry < 8 && tn.corr_groups != 0);
    if (tiles) {
      TimedLaunch tl(KF_CORRESPOND_TILE, s);
      hipLaunchKernelGGL(k_gicp_correspond_tile, dim3(xcd_grid(gb.blocks_tile, gb.S)), dim3(kGicpBlock), 0, s, src.pts.p, src.count.p, src.cap,
                         tgt.pts.p, tgt.orig.p, tgt.table.p, tgt.meta.p, tgt.cap, tgt.slots_cap, gb.tgt_of_src.p, gb.state.p, gb.corr.p,
                         gb.qsearch.p, P.max_corr_d2, tn.corr_r1, gb.blocks_tile, gb.S, gb.corr_stats.p, (double)tn.margin_step,
                         (src.has_order && src.cap == gb.cap) ? (const int*)src.tile_order.p : (const int*)nullptr);
    }
    TimedLaunch tl(KF_CORRESPOND, s);
    if (groups && isfinite(P.max_corr_d2)) {
#ifndef GROUP_TILE
#define GROUP_TILE 8
#endif
      // queries per block: a block searches 8 of them per round (a 32-lane group each), one round after the other - with 64 queries a
      // block was a chain of 8 rounds and a 16 k-point scan filled half the chip's wave slots (230 us per from-scratch launch of the stream)
      constexpr int kTile = GROUP_TILE;
      const int Gg = std::max(1, std::min(4096, div_up((long)gb.cap, kTile)));
      hipLaunchKernelGGL(k_gicp_correspond_group<kTile>, dim3(xcd_grid(Gg, gb.S)), dim3(kGicpBlock), 0, s, src.pts.p, src.count.p, src.cap, tgt.pts.p,
                         tgt.orig.p, tgt.table.p, tgt.meta.p, tgt.cap, tgt.slots_cap, gb.tgt_of_src.p, gb.state.p, gb.corr.p, gb.qsearch.p,
                         P.max_corr_d2, Gg, gb.S, (double)tn.margin_step);
    } else {
      hipLaunchKernelGGL(kc, dim3(xcd_grid(gb.blocks_per_pair, gb.S)), dim3(kGicpBlock), 0, s, src.pts.p, src.count.p, src.cap, tgt.pts.p, tgt.orig.p,
                         tgt.table.p, tgt.meta.p, tgt.cap, tgt.slots_cap, gb.tgt_of_src.p, gb.state.p, gb.corr.p, gb.qsearch.p, P.max_corr_d2,
                         gb.blocks_per_pair, gb.S, tiles ? 1 : 0, (double)tn.margin_step, gb.corr_stats.p, iteration,
                         (tn.corr_mask && search < 0) ? (const unsigned long long*)gb.safe_mask.p : (const unsigned long long*)nullptr,
                         (size_t)div_up((long)gb.cap, 64));
    }
  }
  TimedLaunch tl(KF_LINEARIZE, s);
  const bool mask = tuning().corr_mask && search < 0;
  hipLaunchKernelGGL(kl, dim3(xcd_grid(gb.blocks_lin, gb.S)), dim3(kGicpBlock), 0, s, src.pts.p, src.cov.p, src.count.p, src.cap,
                     tgt.pts.p, tgt.cov.p, tgt.cap, gb.tgt_of_src.p, gb.state.p, gb.corr.p, search >= 0 ? gb.mahal.p : (Cov6*)nullptr, gb.partials.p, gb.blocks_lin, gb.S,
                     (const float4*)gb.qsearch.p, mask ? gb.safe_mask.p : (unsigned long long*)nullptr,
                     (size_t)div_up((long)gb.cap, 64), (double)tuning().margin_step);
}

// ---- the LM optimisation of every pair of a batch, in three pieces a caller can interleave with those of other batches --------
// gicp_align_begin    pair -> target map, guesses, initial state (enqueue only)
// gicp_align_iterate  iterations [it0, it1): correspond -> linearize -> step (enqueue only; a pair that has
//                     converged leaves every kernel at once, so iterations enqueued past convergence cost launches, not work)
// gicp_poll_active    "is any pair still optimising?": the count goes to a pinned host slot behind an event that
//                     gicp_poll_query tests WITHOUT blocking (round 3 synchronised the stream every four iterations: the one host
//                     round trip of the loop, and with several parts / batches in flight it stalled the thread that should have
//                     been enqueueing the others)
void gicp_align_begin(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const int* tgt_of_src_host,
                      const float* guesses_host) {
  if (!src.has_cov || !tgt.has_cov) throw ArgError(MI355_ESTATE, "covariances missing");
  const int S = gb.S;
  gb.h_map.resize(S);
  for (int i = 0; i < S; i++) {
    gb.h_map[i] = tgt_of_src_host ? tgt_of_src_host[i] : i;
    if (gb.h_map[i] < 0 || gb.h_map[i] >= tgt.B) throw ArgError(MI355_EINVAL, "tgt_of_src out of range");
  }
  MI_HIP(hipMemcpyAsync(gb.tgt_of_src.p, gb.h_map.data(), S * sizeof(int), hipMemcpyHostToDevice, s));
  if (guesses_host) {
    gb.h_guess.assign(guesses_host, guesses_host + (size_t)S * 16);
    gb.guesses.ensure((size_t)S * 16);
    MI_HIP(hipMemcpyAsync(gb.guesses.p, gb.h_guess.data(), (size_t)S * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(k_gicp_init, dim3(div_up(S, 64)), dim3(64), 0, s, gb.state.p, (const float*)gb.guesses.p, S, guesses_host ? 1 : 0,
                     (const int*)src.count.p, (const int*)tgt.count.p, (const int*)gb.tgt_of_src.p, std::max(p.k_correspondences, 1));
  // no previous correspondences yet, NaN margins: the first pass searches every point
  hipLaunchKernelGGL(k_gicp_reset_search, dim3(std::max(1, std::min(64, div_up((long)src.cap, 1024))), S), dim3(256), 0, s, (const int*)src.count.p, gb.corr.p,
                     gb.qsearch.p, src.cap);
  (void)dev_params(p);  // (argument check before anything else is enqueued)
  gb.polls_issued = gb.polls_seen = 0;
  MI_LAUNCH_CHECK();
}

void gicp_align_iterate(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, int it0, int it1) {
  const int S = gb.S;
  const GicpDevParams P = dev_params(p);
  const Tuning& tn = tuning();
  for (int it = it0; it < it1; it++) {
    launch_linearize(s, gb, src, tgt, P, it);
    TimedLaunch tl(KF_STEP, s);
    hipLaunchKernelGGL(k_gicp_step, dim3(S), dim3(64), 0, s, gb.state.p, (const double*)gb.partials.p, gb.blocks_lin, P, 0,
                       tn.corr_mask ? (double)tn.margin_step : -1.0, tn.mask_slack);
  }
  MI_LAUNCH_CHECK();
}

void gicp_poll_active(hipStream_t s, GicpBatch& gb) {
  const int slot = gb.polls_issued % GicpBatch::kPolls;
  if (gb.polls_issued - gb.polls_seen >= GicpBatch::kPolls) throw std::logic_error("gicp_poll_active: every poll slot is in flight");
  gb.n_active.ensure(GicpBatch::kPolls);
  gb.h_active.ensure(GicpBatch::kPolls);
  if (!gb.poll_ev[slot]) MI_HIP(hipEventCreateWithFlags(&gb.poll_ev[slot], hipEventDisableTiming));
  hipLaunchKernelGGL(k_count_active, dim3(1), dim3(256), 0, s, gb.state.p, gb.S, gb.n_active.p + slot);
  MI_HIP(hipMemcpyAsync(gb.h_active.p + slot, gb.n_active.p + slot, sizeof(int), hipMemcpyDeviceToHost, s));
  MI_HIP(hipEventRecord(gb.poll_ev[slot], s));
  gb.polls_issued++;
}

// The oldest poll in flight: has its answer arrived (wait = true: block until it has)?  *n_active then holds it and the slot is free.
bool gicp_poll_query(GicpBatch& gb, bool wait, int* n_active) {
  if (gb.polls_seen >= gb.polls_issued) return false;
  const int slot = gb.polls_seen % GicpBatch::kPolls;
  if (wait) {
    MI_HIP(hipEventSynchronize(gb.poll_ev[slot]));
  } else {
    const hipError_t e = hipEventQuery(gb.poll_ev[slot]);
    if (e == hipErrorNotReady) return false;
    MI_HIP(e);
  }
  *n_active = gb.h_active.p[slot];
  gb.polls_seen++;
  return true;
}

// One batch by itself (the registration object, a batch of one part): iterations are enqueued four at a time, each group followed
// by a poll; the host looks at a poll's answer only after the NEXT group is in the queue, so the GPU never waits for the host - at
// most one group of (empty) iterations runs past convergence.
void gicp_align(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const int* tgt_of_src_host,
                const float* guesses_host) {
  gicp_align_begin(s, gb, src, tgt, p, tgt_of_src_host, guesses_host);
  if (p.force_iterations) {
    gicp_align_iterate(s, gb, src, tgt, p, 0, p.max_iterations);
    return;
  }
  constexpr int kGroup = 4;
  bool done = false;
  for (int it = 0; it < p.max_iterations && !done; it += kGroup) {
    const int it1 = std::min(it + kGroup, p.max_iterations);
    gicp_align_iterate(s, gb, src, tgt, p, it, it1);
    if (it1 >= p.max_iterations) break;
    gicp_poll_active(s, gb);
    int n_active = 1;
    // the poll behind the PREVIOUS group (blocking: its group has certainly been overtaken by the one just enqueued)
    while (gb.polls_issued - gb.polls_seen > 1)
      if (gicp_poll_query(gb, true, &n_active) && n_active == 0) done = true;
  }
  int n_active;
  while (gicp_poll_query(gb, true, &n_active)) {}  // leave no poll in flight (the slots are reused by the next alignment)
}

// ---- exactness probe of the incremental correspondence search (parity tests) -------------------------------------------------
__global__ void k_gicp_set_active(GicpPairState* st, int S, int value, int* __restrict__ saved) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  if (saved && value) saved[p] = st[p].active;
  st[p].active = (saved && !value) ? saved[p] : value;
  if (st[p].status != MI355_PAIR_OK) st[p].active = 0;
}
__global__ __launch_bounds__(256) void k_gicp_compare_corr(const int* __restrict__ scount, const GicpPairState* __restrict__ st, const int* __restrict__ a,
                                                          const int* __restrict__ b, size_t scap, unsigned long long* __restrict__ out) {
  const int pair = blockIdx.y;
  if (st[pair].status != MI355_PAIR_OK) return;
  const int n = scount[pair];
  unsigned long long checked = 0, bad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    checked++;
    bad += a[(size_t)pair * scap + i] != b[(size_t)pair * scap + i];
  }
  if (checked) atomicAdd(&out[0], checked);
  if (bad) atomicAdd(&out[1], bad);
}
// After gicp_align: the correspondences the INCREMENTAL search (margins) holds at the final pose against a search of
// every point from scratch at the same pose.  They must be identical.
void gicp_verify_correspondences(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, long long* n_checked,
                                 long long* n_mismatch) {
  const GicpDevParams P = dev_params(p);
  const int S = gb.S;
  DevBuf<int> saved, inc;
  DevBuf<unsigned long long> out;
  saved.ensure(S);
  inc.ensure((size_t)S * src.cap);
  out.ensure(2);
  MI_HIP(hipMemsetAsync(out.p, 0, 2 * sizeof(unsigned long long), s));
  auto kc = gb.ppt == 1 ? k_gicp_correspond<1> : k_gicp_correspond<4>;
  auto correspond = [&](bool with_mask) {
    hipLaunchKernelGGL(kc, dim3(xcd_grid(gb.blocks_per_pair, S)), dim3(kGicpBlock), 0, s, src.pts.p, src.count.p, src.cap, tgt.pts.p, tgt.orig.p, tgt.table.p,
                       tgt.meta.p, tgt.cap, tgt.slots_cap, gb.tgt_of_src.p, gb.state.p, gb.corr.p, gb.qsearch.p, P.max_corr_d2, gb.blocks_per_pair, S, 0,
                       (double)tuning().margin_step, (unsigned long long*)nullptr, 0,
                       (with_mask && tuning().corr_mask) ? (const unsigned long long*)gb.safe_mask.p : (const unsigned long long*)nullptr,
                       (size_t)div_up((long)gb.cap, 64));
  };
  hipLaunchKernelGGL(k_gicp_set_active, dim3(div_up(S, 64)), dim3(64), 0, s, gb.state.p, S, 1, saved.p);
  correspond(true);  // incremental: margins - and the "safe" bits - as the last iteration left them, at the final pose
  MI_HIP(hipMemcpyAsync(inc.p, gb.corr.p, (size_t)S * src.cap * sizeof(int), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_gicp_reset_search, dim3(std::max(1, std::min(64, div_up((long)src.cap, 1024))), S), dim3(256), 0, s, (const int*)src.count.p, gb.corr.p,
                     gb.qsearch.p, src.cap);
  correspond(false);  // from scratch: NaN margins, every point searched
  hipLaunchKernelGGL(k_gicp_compare_corr, dim3(16, S), dim3(256), 0, s, (const int*)src.count.p, gb.state.p, (const int*)inc.p, (const int*)gb.corr.p, src.cap, out.p);
  hipLaunchKernelGGL(k_gicp_set_active, dim3(div_up(S, 64)), dim3(64), 0, s, gb.state.p, S, 0, saved.p);
  unsigned long long h[2];
  MI_HIP(hipMemcpyAsync(h, out.p, sizeof(h), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  MI_LAUNCH_CHECK();
  *n_checked = (long long)h[0];
  *n_mismatch = (long long)h[1];
}

// ---- P5: getFitnessScore ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kGicpBlock) void k_fitness(const float4* __restrict__ spts, const int* __restrict__ scount, size_t scap,
                                                        const float4* __restrict__ tpts, const int* __restrict__ torig,
                                                        const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta, size_t tcap,
                                                        size_t tslots, const int* __restrict__ tgt_of_src, const GicpPairState* __restrict__ state,
                                                        const float* __restrict__ T_all, double* __restrict__ fit_partials, float search_d2,
                                                        double max_range, const int* __restrict__ seed, size_t seed_pitch) {
  __shared__ double smem[4 * 2];
  const int pair = blockIdx.y;
  const GridView g = grid_view(tpts, torig, ttable, tmeta, tgt_of_src[pair], tcap, tslots);
  const int n = scount[pair];
  float Tf[12];
  if (T_all) {
    for (int k = 0; k < 12; k++) Tf[k] = T_all[16 * pair + k];
  } else {
    const Pose x0 = state[pair].x0;  // final_transformation_ = x0.cast<float>()
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) Tf[4 * r + c] = (float)x0.R[3 * r + c];
      Tf[4 * r + 3] = (float)x0.t[r];
    }
  }
  double acc[2] = {0.0, 0.0};
  const size_t so = (size_t)pair * scap;
  for (int i = blockIdx.x * kGicpBlock + threadIdx.x; i < n; i += gridDim.x * kGicpBlock) {
    const float4 p = spts[so + i];
    const float qx = transform_row(Tf, p.x, p.y, p.z), qy = transform_row(Tf + 4, p.x, p.y, p.z), qz = transform_row(Tf + 8, p.x, p.y, p.z);
    Nearest1 nn{g.orig, INFINITY, -1};
    float lim = search_d2;
    int level0 = 0, level_step = 2;
    if (seed) {
      // the engine's last correspondent of this point: the final pose is a last small step away from where it was found, so its distance
      // bounds the walk from the start (the same exact nearest distance: the walk covers everything closer)
      const int j0 = seed[(size_t)pair * seed_pitch + i];
      if (j0 >= 0 && j0 < g.m.n) {
        const float4 q0 = g.pts[j0];
        const float d0 = dist2f(qx, qy, qz, q0.x, q0.y, q0.z);
        if (d0 <= search_d2) {
          nn.best_d2 = d0;
          nn.best_j = j0;
          lim = nextafterf(d0, INFINITY);
          const float rel = sqrtf(d0) * g.m.inv_cell;
          level0 = (rel <= 1.f) ? 0 : min(g.m.top_level, (int)ceilf(log2f(rel)));
          level_step = 1;
        }
      }
    }
    if (isfinite(qx) && isfinite(qy) && isfinite(qz)) grid_search(g, qx, qy, qz, lim, nn, level0, level_step);
    if (nn.best_j >= 0 && (double)nn.best_d2 <= max_range) { acc[0] += (double)nn.best_d2; acc[1] += 1.0; }
  }
  block_sum<2, 4>(acc, smem);
  if (threadIdx.x == 0) {
    fit_partials[((size_t)pair * gridDim.x + blockIdx.x) * 2] = acc[0];
    fit_partials[((size_t)pair * gridDim.x + blockIdx.x) * 2 + 1] = acc[1];
  }
}

__global__ void k_fitness_final(const double* __restrict__ fit_partials, double* __restrict__ fitness, int S, int G) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  double s = 0, c = 0;
  for (int g = 0; g < G; g++) { s += fit_partials[((size_t)p * G + g) * 2]; c += fit_partials[((size_t)p * G + g) * 2 + 1]; }
  fitness[2 * p] = s;
  fitness[2 * p + 1] = c;
}

void gicp_corr_stats_print(hipStream_t s, GicpBatch& gb) {
  if (!gb.corr_stats.p) return;
  unsigned long long h[kCorrStatsN];
  MI_HIP(hipMemcpyAsync(h, gb.corr_stats.p, sizeof(h), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  MI_HIP(hipMemsetAsync(gb.corr_stats.p, 0, sizeof(h), s));
  const double t = (double)std::max(1ull, h[0]);
  fprintf(stderr, "[mi355] correspondence tiles: %llu, %.0f points in round-1 cells each (%llu not run); round 2: %llu of them, %.1f lanes, "
                  "%.0f points in cells each (%llu not run); %llu lanes to the per-lane search\n",
          h[0], (double)h[1] / t, h[2], h[3], (double)h[4] / (double)std::max(1ull, h[3]), (double)h[5] / (double)std::max(1ull, h[3]), h[6], h[7]);
  fprintf(stderr, "[mi355] per-lane correspondence search by iteration (points searched / of them by the whole wave / pairs tracking margins):");
  for (int it = 0; it < 64; it++)
    if (h[8 + 4 * it] || h[8 + 4 * it + 3]) fprintf(stderr, " %d: %llu / %llu / %llu;", it, h[8 + 4 * it], h[8 + 4 * it + 2], h[8 + 4 * it + 3]);
  fprintf(stderr, "\n");
}

void gicp_fitness(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, double max_range, bool use_final_pose,
                  const float* T16_host_all, const int* seed_corr, size_t seed_pitch) {
  const float* dTp = nullptr;
  if (!use_final_pose) {
    gb.h_fit_T.assign(T16_host_all, T16_host_all + (size_t)gb.S * 16);
    gb.fit_T.ensure((size_t)gb.S * 16);
    MI_HIP(hipMemcpyAsync(gb.fit_T.p, gb.h_fit_T.data(), (size_t)gb.S * 16 * sizeof(float), hipMemcpyHostToDevice, s));
    dTp = gb.fit_T.p;
  }
  const float search_d2 = (max_range >= 3.0e38) ? INFINITY : (float)(max_range * (1.0 + 1e-6)) + 1e-30f;
  TimedLaunch tl(KF_FITNESS, s);
  hipLaunchKernelGGL(k_fitness, dim3(gb.blocks_lin, gb.S), dim3(kGicpBlock), 0, s, src.pts.p, src.count.p, src.cap, tgt.pts.p, tgt.orig.p,
                     tgt.table.p, tgt.meta.p, tgt.cap, tgt.slots_cap, gb.tgt_of_src.p, gb.state.p, dTp, gb.fit_partials.p, search_d2, max_range, seed_corr, seed_pitch);
  hipLaunchKernelGGL(k_fitness_final, dim3(div_up(gb.S, 64)), dim3(64), 0, s, gb.fit_partials.p, gb.fitness.p, gb.S, gb.blocks_lin);
  MI_LAUNCH_CHECK();
}

void gicp_results(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, mi355_result* out, double* fitness_out) {
  const int S = gb.S;
  std::vector<GicpPairState> st(S);
  std::vector<int> ns(S), nt(tgt.B), map(S);
  MI_HIP(hipMemcpyAsync(st.data(), gb.state.p, S * sizeof(GicpPairState), hipMemcpyDeviceToHost, s));
  MI_HIP(hipMemcpyAsync(ns.data(), src.count.p, S * sizeof(int), hipMemcpyDeviceToHost, s));
  MI_HIP(hipMemcpyAsync(nt.data(), tgt.count.p, tgt.B * sizeof(int), hipMemcpyDeviceToHost, s));
  MI_HIP(hipMemcpyAsync(map.data(), gb.tgt_of_src.p, S * sizeof(int), hipMemcpyDeviceToHost, s));
  std::vector<double> fit;
  if (fitness_out) {
    fit.resize((size_t)S * 2);
    MI_HIP(hipMemcpyAsync(fit.data(), gb.fitness.p, fit.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  MI_HIP(hipStreamSynchronize(s));
  for (int p = 0; p < S; p++) {
    mi355_result& r = out[p];
    memset(&r, 0, sizeof(r));
    pose_to_rowmajor4x4f(st[p].x0, r.T);
    r.converged = st[p].converged;
    r.iterations = st[p].iter;
    r.error = st[p].last_error;
    memcpy(r.H, st[p].final_H, sizeof(r.H));
    r.n_source = ns[p];
    r.n_target = nt[map[p]];
    r.n_inliers = st[p].n_inliers;
    r.status = st[p].status;
    if (fitness_out) fitness_out[p] = fit[2 * p + 1] > 0 ? fit[2 * p] / fit[2 * p + 1] : 1.7976931348623157e308;
  }
}

void gicp_first_converged(hipStream_t s, GicpBatch& gb, int* out) {
  std::vector<GicpPairState> st((size_t)gb.S);
  MI_HIP(hipMemcpyAsync(st.data(), gb.state.p, (size_t)gb.S * sizeof(GicpPairState), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  for (int p = 0; p < gb.S; p++) out[p] = st[(size_t)p].first_conv;
}

__global__ void k_set_pose(GicpPairState* st, Pose x) {
  st[0].x0 = x;
  st[0].active = 1;
  st[0].lambda = -1.0;
}

void gicp_linearize_once(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const double* T16,
                         double* H36, double* b6, double* err, int* corr_caller_order) {
  if (!src.has_cov || !tgt.has_cov) throw ArgError(MI355_ESTATE, "covariances missing");
  const int zero = 0;
  MI_HIP(hipMemcpyAsync(gb.tgt_of_src.p, &zero, sizeof(int), hipMemcpyHostToDevice, s));
  MI_HIP(hipStreamSynchronize(s));
  hipLaunchKernelGGL(k_gicp_init, dim3(1), dim3(64), 0, s, gb.state.p, (const float*)nullptr, 1, 0, (const int*)nullptr, (const int*)nullptr,
                     (const int*)nullptr, 0);
  Pose x;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) x.R[3 * r + c] = T16[4 * r + c];
    x.t[r] = T16[4 * r + 3];
  }
  hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(1), 0, s, gb.state.p, x);
  gb.mahal.ensure(gb.cap);  // (only this entry point - the parity probes - keeps the Mahalanobis matrices: pair 0's)
  MI_HIP(hipMemsetAsync(gb.corr.p, 0xFF, gb.cap * sizeof(int), s));
  MI_HIP(hipMemsetAsync(gb.qsearch.p, 0xFF, gb.cap * sizeof(float4), s));
  GicpDevParams P = dev_params(p);
  P.optimizer = MI355_OPT_LM;
  // this entry point exists for the parity tests: run ALL THREE correspondence searches from scratch — the per-lane walk, the
  // lane groups (what a lone pair gets in an alignment), the tile kernel (what iteration 0 of a batch gets) — and insist that
  // they agree, so that the caller's comparison with the oracle pins all of them
  std::vector<int> first;
  for (int search = 0; search <= 2; search++) {
    if ((search == 1 && !tuning().corr_tiles) || (search == 2 && !tuning().corr_groups)) continue;
    MI_HIP(hipMemsetAsync(gb.corr.p, 0xFF, gb.cap * sizeof(int), s));
    MI_HIP(hipMemsetAsync(gb.qsearch.p, 0xFF, gb.cap * sizeof(float4), s));
    launch_linearize(s, gb, src, tgt, P, 0, search);
    std::vector<int> got(gb.cap);
    int ns = 0;
    MI_HIP(hipMemcpyAsync(got.data(), gb.corr.p, gb.cap * sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(&ns, src.count.p, sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (first.empty()) {
      first.swap(got);
      continue;
    }
    for (int i = 0; i < ns; i++)
      if (got[(size_t)i] != first[(size_t)i])
        throw ArgError(MI355_ESTATE, search == 1 ? "tile and per-lane correspondence searches disagree" : "group and per-lane correspondence searches disagree");
  }
  hipLaunchKernelGGL(k_gicp_step, dim3(1), dim3(64), 0, s, gb.state.p, (const double*)gb.partials.p, gb.blocks_lin, P, 1, -1.0, 0.f);
  MI_LAUNCH_CHECK();
  GicpPairState st;
  MI_HIP(hipMemcpyAsync(&st, gb.state.p, sizeof(st), hipMemcpyDeviceToHost, s));
  int n = 0;
  MI_HIP(hipMemcpyAsync(&n, src.count.p, sizeof(int), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  memcpy(H36, st.H, sizeof(st.H));
  memcpy(b6, st.b, sizeof(st.b));
  *err = st.y0;
  if (corr_caller_order) {
    int nt = 0;
    MI_HIP(hipMemcpy(&nt, tgt.count.p, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> c(n), so(n), to(nt);
    MI_HIP(hipMemcpy(c.data(), gb.corr.p, n * sizeof(int), hipMemcpyDeviceToHost));
    MI_HIP(hipMemcpy(so.data(), src.orig.p, n * sizeof(int), hipMemcpyDeviceToHost));
    MI_HIP(hipMemcpy(to.data(), tgt.orig.p, nt * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) corr_caller_order[so[i]] = c[i] >= 0 ? to[c[i]] : -1;
  }
}

// Parity probe of the trial-cost model: linearise at T_lin (as gicp_linearize_once), then the cost at T_trial with those frozen
// correspondences and Mahalanobis matrices - once through the model the optimisation uses (the sums of k_gicp_linearize), once as
// FastGICP::compute_error writes it, a pass over the points.
void gicp_trial_cost_probe(hipStream_t s, GicpBatch& gb, const CloudSet& src, const CloudSet& tgt, const mi355_gicp_params& p, const double* T_lin16,
                           const double* T_trial16, double* model_cost, double* direct_cost) {
  double H[36], b[6], y0 = 0;
  gicp_linearize_once(s, gb, src, tgt, p, T_lin16, H, b, &y0, nullptr);
  const int G = gb.blocks_lin;
  std::vector<double> part((size_t)G * kPartStride);
  MI_HIP(hipMemcpyAsync(part.data(), gb.partials.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  auto to_pose = [](const double* T) {
    Pose x;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) x.R[3 * r + c] = T[4 * r + c];
      x.t[r] = T[4 * r + 3];
    }
    return x;
  };
  const Pose xl = to_pose(T_lin16), xt = to_pose(T_trial16);
  Pose inv;  // xl^-1 (a rigid transform)
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) inv.R[3 * r + c] = xl.R[3 * c + r];
  }
  for (int r = 0; r < 3; r++) inv.t[r] = -(inv.R[3 * r] * xl.t[0] + inv.R[3 * r + 1] * xl.t[1] + inv.R[3 * r + 2] * xl.t[2]);
  constexpr int kProbeBlocks = 64;
  DevBuf<double> out;
  out.ensure(kProbeBlocks);
  hipLaunchKernelGGL(k_gicp_cost_probe, dim3(kProbeBlocks), dim3(kGicpBlock), 0, s, (const float4*)src.pts.p, (const int*)src.count.p,
                     (const float4*)tgt.pts.p, (const int*)gb.corr.p, (const Cov6*)gb.mahal.p, xt, out.p);
  MI_LAUNCH_CHECK();
  double direct[kProbeBlocks];
  MI_HIP(hipMemcpyAsync(direct, out.p, sizeof(direct), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  double sums[kPartStride], m[kNModel];
  for (int k = 0; k < kPartStride; k++) {
    double sum = 0;
    for (int g = 0; g < G; g++) sum += part[(size_t)g * kPartStride + k];
    sums[k] = sum;
  }
  for (int k = 0; k < kNModel; k++) m[k] = model_entry(sums, k);
  *model_cost = y0 - model_decrease(pose_compose(xt, inv), m);
  double d = 0;
  for (int g = 0; g < kProbeBlocks; g++) d += direct[g];
  *direct_cost = d;
}

}  // namespace mi355
