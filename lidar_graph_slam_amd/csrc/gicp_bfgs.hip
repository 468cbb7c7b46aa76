// pclomp::GeneralizedIterativeClosestPoint ("GICP": lidar_scan_matcher.cpp:73-96, graph_based_slam.cpp:120-141) on gfx950 —
// SURVEY.md 8f-4.  PCL 1.8's GICP: per OUTER iteration one correspondence + Mahalanobis pass over the source, then a BFGS
// minimisation (host/bfgs.hpp) of f(x) = 1/m sum r^T M r over the six pose parameters, every evaluation of which is one
// streaming pass over the matched pairs:
//   k_bfgs_correspond   q = transformation_ * (guess * p) in fp32; exact 1-NN in the target's voxel hash, kept iff
//                       d2 < max_dist^2; M = (R C_src R^T + C_tgt)^-1 in fp64 (R = rotation of transformation_ * guess);
//                       64 algorithmic bytes per source point (point + covariance of source and correspondent);
//   k_bfgs_cost         r = T(x) p - q_tgt (fp32 transform, fp64 residual), 14 fp64 sums: m, f, the translation gradient
//                       and the 3x3 accumulator the rotation gradient is contracted from; 88 B per matched pair (source point,
//                       correspondence, target point, 48-byte fp64 Mahalanobis matrix).
// Both reduce in a fixed order (block partials, then one thread per sum), so a run is bitwise reproducible.  The covariances are
// the PLANE-regularised ones of gicp.hip (singular values (1, 1, 1e-3) = PCL's gicp_epsilon).  The loop is serial by nature
// (one pair, ~10^2 dependent evaluations): a launch + 112-byte read-back per evaluation.
#include <cmath>
#include <cstring>

#include "engine.hpp"
#include "grid.hpp"
#include "host/bfgs.hpp"
#include "primitives.hpp"

namespace mi355 {

constexpr int kBfgsAcc = 14;
constexpr int kBfgsBlock = 256;

struct Mat4f {
  float m[16];
};
struct Mat3d {
  double m[9];
};

// Eigen's Matrix4f * Vector4f with w = 1, evaluated left to right (no fma: the oracle spells the same expression)
__device__ __forceinline__ float row_apply(const float* r, float x, float y, float z) { return r[0] * x + r[1] * y + r[2] * z + r[3]; }

__global__ __launch_bounds__(kBfgsBlock) void k_bfgs_correspond(const float4* __restrict__ spts, const Cov6* __restrict__ scov, int n,
                                                                const float4* __restrict__ tpts, const int* __restrict__ torig,
                                                                const Cov6* __restrict__ tcov, const HashSlot* __restrict__ ttable,
                                                                const GridMeta* __restrict__ tmeta, size_t tcap, size_t tslots, Mat4f guess,
                                                                Mat4f trans, Mat3d R, float max_d2, int* __restrict__ corr,
                                                                double* __restrict__ mahal) {
  const int i = blockIdx.x * kBfgsBlock + threadIdx.x;
  if (i >= n) return;
  const GridView g = grid_view(tpts, torig, ttable, tmeta, 0, tcap, tslots);
  const float4 p = spts[i];
  const float gx = row_apply(guess.m, p.x, p.y, p.z), gy = row_apply(guess.m + 4, p.x, p.y, p.z), gz = row_apply(guess.m + 8, p.x, p.y, p.z);
  const float qx = row_apply(trans.m, gx, gy, gz), qy = row_apply(trans.m + 4, gx, gy, gz), qz = row_apply(trans.m + 8, gx, gy, gz);
  Nearest1 nn{g.orig, INFINITY, -1};
  if (isfinite(qx) && isfinite(qy) && isfinite(qz)) grid_search(g, qx, qy, qz, max_d2, nn, 0, 2);
  int j = -1;
  if (nn.best_j >= 0 && nn.best_d2 < max_d2) {
    j = nn.best_j;
    const Sym3<float> a = cov6_unpack(scov[i]), b = cov6_unpack(tcov[j]);
    const Sym3<double> A = {(double)a.xx, (double)a.xy, (double)a.xz, (double)a.yy, (double)a.yz, (double)a.zz};
    const Sym3<double> RA = sym3_rotate(R.m, A);
    const Sym3<double> S = {RA.xx + (double)b.xx, RA.xy + (double)b.xy, RA.xz + (double)b.xz, RA.yy + (double)b.yy, RA.yz + (double)b.yz,
                            RA.zz + (double)b.zz};
    const Sym3<double> M = sym3_inverse(S);
    double* o = mahal + 6 * (size_t)i;
    o[0] = M.xx; o[1] = M.xy; o[2] = M.xz; o[3] = M.yy; o[4] = M.yz; o[5] = M.zz;
  }
  corr[i] = j;
}

// sums: [0] m, [1] sum r^T M r, [2..4] sum M r, [5..13] sum (base p) (M r)^T row-major
__global__ __launch_bounds__(kBfgsBlock) void k_bfgs_cost(const float4* __restrict__ spts, int n, const float4* __restrict__ tpts,
                                                          const int* __restrict__ corr, const double* __restrict__ mahal, Mat4f T, Mat4f base,
                                                          int with_gradient, double* __restrict__ partials) {
  __shared__ double smem[4 * kBfgsAcc];
  double acc[kBfgsAcc];
#pragma unroll
  for (int k = 0; k < kBfgsAcc; k++) acc[k] = 0.0;
  for (int i = blockIdx.x * kBfgsBlock + threadIdx.x; i < n; i += gridDim.x * kBfgsBlock) {
    const int j = corr[i];
    if (j < 0) continue;
    const float4 p = spts[i], q = tpts[j];
    const float px = row_apply(T.m, p.x, p.y, p.z), py = row_apply(T.m + 4, p.x, p.y, p.z), pz = row_apply(T.m + 8, p.x, p.y, p.z);
    const double r0 = (double)(px - q.x), r1 = (double)(py - q.y), r2 = (double)(pz - q.z);
    const double* M = mahal + 6 * (size_t)i;
    const double t0 = M[0] * r0 + M[1] * r1 + M[2] * r2, t1 = M[1] * r0 + M[3] * r1 + M[4] * r2, t2 = M[2] * r0 + M[4] * r1 + M[5] * r2;
    acc[0] += 1.0;
    acc[1] += r0 * t0 + r1 * t1 + r2 * t2;
    if (with_gradient) {
      acc[2] += t0; acc[3] += t1; acc[4] += t2;
      // base_transformation_ * p_src, translation included (upstream's: gicp.hpp, OptimizationFunctorWithIndices::df)
      const double b0 = (double)row_apply(base.m, p.x, p.y, p.z), b1 = (double)row_apply(base.m + 4, p.x, p.y, p.z),
                   b2 = (double)row_apply(base.m + 8, p.x, p.y, p.z);
      acc[5] += b0 * t0; acc[6] += b0 * t1; acc[7] += b0 * t2;
      acc[8] += b1 * t0; acc[9] += b1 * t1; acc[10] += b1 * t2;
      acc[11] += b2 * t0; acc[12] += b2 * t1; acc[13] += b2 * t2;
    }
  }
  block_sum<kBfgsAcc, 4>(acc, smem);
  if (threadIdx.x == 0) {
    double* out = partials + (size_t)blockIdx.x * 16;
#pragma unroll
    for (int k = 0; k < kBfgsAcc; k++) out[k] = acc[k];
  }
}

__global__ void k_bfgs_reduce(const double* __restrict__ partials, int nblk, double* __restrict__ sums) {
  const int k = threadIdx.x;
  if (k >= kBfgsAcc) return;
  double s = 0;
  for (int b = 0; b < nblk; b++) s += partials[(size_t)b * 16 + k];
  sums[k] = s;
}

namespace {

// GeneralizedIterativeClosestPoint::applyState (PCL 1.8): rotation block <- R(x) * rotation block, translation column += x0..2,
// fp32; R(x) = AngleAxisf(x5, Z) * AngleAxisf(x4, Y) * AngleAxisf(x3, X), which Eigen evaluates as a quaternion product
void apply_state(float t[16], const Vec6& x) {
  const float hz = 0.5f * (float)x[5], hy = 0.5f * (float)x[4], hx = 0.5f * (float)x[3];
  const float qz[4] = {std::cos(hz), 0.f, 0.f, std::sin(hz)}, qy[4] = {std::cos(hy), 0.f, std::sin(hy), 0.f}, qx[4] = {std::cos(hx), std::sin(hx), 0.f, 0.f};
  auto qmul = [](const float* a, const float* b, float* o) {  // (w, x, y, z)
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  };
  float qzy[4], q[4];
  qmul(qz, qy, qzy);
  qmul(qzy, qx, q);
  const float w = q[0], X = q[1], Y = q[2], Z = q[3];
  const float tx = 2 * X, ty = 2 * Y, tz = 2 * Z;
  const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * X, txy = ty * X, txz = tz * X, tyy = ty * Y, tyz = tz * Y, tzz = tz * Z;
  const float R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  float o[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) o[3 * r + c] = R[3 * r] * t[c] + R[3 * r + 1] * t[4 + c] + R[3 * r + 2] * t[8 + c];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[4 * r + c] = o[3 * r + c];
  t[3] += (float)x[0]; t[7] += (float)x[1]; t[11] += (float)x[2];
}

// computeRDerivative: g[3..5] = tr(dR/d(roll, pitch, yaw) * Racc) for R = Rz(yaw) Ry(pitch) Rx(roll)
void rotation_gradient(const Vec6& x, const double Racc[9], Vec6& g) {
  const double cr = std::cos(x[3]), sr = std::sin(x[3]), cp = std::cos(x[4]), sp = std::sin(x[4]), cy = std::cos(x[5]), sy = std::sin(x[5]);
  const double d_roll[9] = {0, sr * sy + cr * cy * sp, cr * sy - cy * sr * sp, 0, -cy * sr + cr * sy * sp, -cr * cy - sr * sy * sp, 0, cr * cp, -cp * sr};
  const double d_pitch[9] = {-cy * sp, cy * cp * sr, cr * cy * cp, -sy * sp, cp * sr * sy, cr * cp * sy, -cp, -sr * sp, -cr * sp};
  const double d_yaw[9] = {-cp * sy, -cr * cy - sr * sy * sp, cy * sr - cr * sy * sp, cy * cp, -cr * sy + cy * sr * sp, sr * sy + cr * cy * sp, 0, 0, 0};
  auto contract = [&](const double* D) {
    double s = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) s += D[3 * i + j] * Racc[3 * j + i];
    return s;
  };
  g[3] = contract(d_roll); g[4] = contract(d_pitch); g[5] = contract(d_yaw);
}

// OptimizationFunctorWithIndices: one kernel pass per evaluation
struct CostOnDevice {
  hipStream_t s;
  BfgsScratch& sc;
  const CloudSet& src;
  const CloudSet& tgt;
  int n, blocks;
  Mat4f base;
  long evaluations = 0;

  void sums(const Vec6& x, bool with_gradient, double* f, Vec6* g) {
    evaluations++;
    Mat4f T = base;
    apply_state(T.m, x);
    hipLaunchKernelGGL(k_bfgs_cost, dim3(blocks), dim3(kBfgsBlock), 0, s, src.pts.p, n, tgt.pts.p, sc.corr.p, sc.mahal.p, T, base,
                       with_gradient ? 1 : 0, sc.partials.p);
    hipLaunchKernelGGL(k_bfgs_reduce, dim3(1), dim3(64), 0, s, sc.partials.p, blocks, sc.sums.p);
    MI_LAUNCH_CHECK();
    MI_HIP(hipMemcpyAsync(sc.h_sums.p, sc.sums.p, kBfgsAcc * sizeof(double), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    const double* h = sc.h_sums.p;
    const double m = h[0];
    if (f) *f = h[1] / m;
    if (g) {
      for (int k = 0; k < 3; k++) (*g)[k] = h[2 + k] * 2.0 / m;
      double Racc[9];
      for (int k = 0; k < 9; k++) Racc[k] = h[5 + k] * 2.0 / m;
      rotation_gradient(x, Racc, *g);
    }
  }
  double f(const Vec6& x) { double v; sums(x, false, &v, nullptr); return v; }
  void df(const Vec6& x, Vec6& g) { sums(x, true, nullptr, &g); }
  void fdf(const Vec6& x, double& fo, Vec6& g) { sums(x, true, &fo, &g); }
};

}  // namespace

void gicp_bfgs_align(hipStream_t s, BfgsScratch& sc, const CloudSet& src, const CloudSet& tgt, size_t n_src, const mi355_gicp_bfgs_params& p,
                     const float* guess16, mi355_result* out) {
  if (!src.has_cov || !tgt.has_cov) throw ArgError(MI355_ESTATE, "covariances missing");
  const int n = (int)n_src;
  sc.corr.ensure(n_src);
  sc.mahal.ensure(6 * n_src);
  const int blocks = std::max(1, std::min(1024, div_up(n, kBfgsBlock)));
  sc.partials.ensure((size_t)blocks * 16);
  sc.sums.ensure(16);
  sc.h_sums.ensure(16);
  Mat4f guess, trans, prev;
  memcpy(guess.m, guess16, sizeof(guess.m));
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(trans.m, I, sizeof(I));
  prev = trans;
  const float max_d2 = (float)(p.max_correspondence_distance * p.max_correspondence_distance);
  CostOnDevice cost{s, sc, src, tgt, n, blocks, guess};
  int nr = 0;
  bool converged = false;
  double last_f = 0;
  while (!converged) {
    // transform_R = double(transformation_) * double(guess); its rotation block carries the source covariances
    Mat3d R;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double v = 0;
        for (int k = 0; k < 4; k++) v += (double)trans.m[4 * i + k] * (double)guess.m[4 * k + j];
        R.m[3 * i + j] = v;
      }
    hipLaunchKernelGGL(k_bfgs_correspond, dim3(div_up(n, kBfgsBlock)), dim3(kBfgsBlock), 0, s, src.pts.p, src.cov.p, n, tgt.pts.p, tgt.orig.p,
                       tgt.cov.p, tgt.table.p, tgt.meta.p, tgt.cap, tgt.slots_cap, guess, trans, R, max_d2, sc.corr.p, sc.mahal.p);
    MI_LAUNCH_CHECK();
    prev = trans;
    // ---- estimateRigidTransformationBFGS
    Vec6 x = {(double)trans.m[3], (double)trans.m[7], (double)trans.m[11], std::atan2((double)trans.m[9], (double)trans.m[10]),
              std::asin(-(double)trans.m[8]), std::atan2((double)trans.m[4], (double)trans.m[0])};
    Bfgs<CostOnDevice> bfgs(cost);
    bfgs.minimizeInit(x);  // its first evaluation also tells how many correspondences there are
    if (sc.h_sums.p[0] < 4.0) break;  // "Need at least 4 points to estimate a transform": the loop is left, not converged
    int inner = 0;
    BfgsStatus result = BfgsStatus::Running;
    do {
      inner++;
      result = bfgs.minimizeOneStep(x);
      if (result != BfgsStatus::Success) break;
      result = bfgs.testGradient(1e-2);
    } while (result == BfgsStatus::Running && inner < p.max_optimizer_iterations);
    if (!(result == BfgsStatus::NoProgress || result == BfgsStatus::Success || inner == p.max_optimizer_iterations)) break;
    last_f = bfgs.value();
    memcpy(trans.m, I, sizeof(I));
    apply_state(trans.m, x);
    double delta = 0;
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 4; l++) {
        const double ratio = (k < 3 && l < 3) ? 1.0 / p.rotation_epsilon : 1.0 / p.transformation_epsilon;
        delta = std::max(delta, ratio * std::fabs((double)prev.m[4 * k + l] - (double)trans.m[4 * k + l]));
      }
    nr++;
    if (nr >= p.max_iterations || delta < 1) {
      converged = true;
      prev = trans;
    }
  }
  memset(out, 0, sizeof(*out));
  // final_transformation_ = [R_prev R_guess | t_prev + t_guess]
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out->T[4 * r + c] = prev.m[4 * r] * guess.m[c] + prev.m[4 * r + 1] * guess.m[4 + c] + prev.m[4 * r + 2] * guess.m[8 + c];
    out->T[4 * r + 3] = prev.m[4 * r + 3] + guess.m[4 * r + 3];
  }
  out->T[15] = 1.f;
  out->converged = converged ? 1 : 0;
  out->iterations = nr;
  out->error = last_f;
  out->n_source = n;
  out->n_inliers = (int)sc.h_sums.p[0];
}

}  // namespace mi355
