// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED.
//
// Restatement of fast_gicp::FastGICP<PointXYZI,PointXYZI> (class chain FastGICP : LsqRegistration :
// pcl::Registration) as the reference drives it from
// /root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:37-54 (construction + setters),
// :149,:212 (setInputTarget), :162 (setInputSource), :165 (align with the previous pose as guess),
// :167,:172 (hasConverged / getFinalTransformation) and
// /root/reference/graph_based_slam/src/graph_based_slam.cpp:84-100,315-322.
// The fast_gicp sources are an un-vendored submodule (/root/reference/.gitmodules:4-6); the algorithm below
// follows SURVEY.md §8a rows F1-F10 [UPSTREAM-RECALL].
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kdtree.hpp"
#include "linalg.hpp"
#include "oracle.h"

using namespace oracle;

extern "C" void orc_gicp_default_params(orc_gicp_params* p) {
  p->k_correspondences = 20;
  p->max_iterations = 64;
  p->transformation_epsilon = 0.01;  // reference yaml; fast_gicp's own default is 5e-4
  p->rotation_epsilon = 2e-3;
  p->max_correspondence_distance = 2.0;
  p->regularization_method = ORC_REG_PLANE;
  p->optimizer = ORC_OPT_LM;
  p->lm_max_iterations = 10;
  p->lm_init_lambda_factor = 1e-9;
  p->force_iterations = 0;
  p->num_threads = 0;
}

// ---- F8: so3_exp / se3_exp -------------------------------------------------------------------------------
static void quat_to_R(double w, double x, double y, double z, Mat3& R) {
  // Eigen::Quaterniond::toRotationMatrix (no normalisation)
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
  R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
}

static Iso3 se3_exp(const double a[6]) {
  const Vec3 omega = {{a[0], a[1], a[2]}};
  const double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  double imag_factor, real_factor;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = std::sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    imag_factor = std::sin(half_theta) / theta;
    real_factor = std::cos(half_theta);
  }
  Iso3 se3;
  quat_to_R(real_factor, imag_factor * omega[0], imag_factor * omega[1], imag_factor * omega[2], se3.R);
  const double theta = std::sqrt(theta_sq);
  Mat3 V;
  if (theta < 1e-10) {
    V = se3.R;
  } else {
    const Mat3 Omega = skew(omega);
    const Mat3 Omega_sq = mul(Omega, Omega);
    const double tsq = theta * theta;
    V = add(add(Mat3::identity(), scale(Omega, (1.0 - std::cos(theta)) / tsq)),
            scale(Omega_sq, (theta - std::sin(theta)) / (tsq * theta)));
  }
  se3.t = mul(V, Vec3{{a[3], a[4], a[5]}});
  return se3;
}

extern "C" void orc_se3_exp(const double d[6], double T16[16]) {
  const Iso3 x = se3_exp(d);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T16[4 * r + c] = x.R(r, c);
    T16[4 * r + 3] = x.t[r];
  }
  T16[12] = T16[13] = T16[14] = 0;
  T16[15] = 1;
}

// ---- F2: calculate_covariances -----------------------------------------------------------------------------
static void covariances_impl(const float* pts, size_t n, const KdTree& tree, int k, int method, double* covs9) {
#pragma omp parallel
  {
    std::vector<int> ki(k);
    std::vector<float> kd(k);
#pragma omp for schedule(guided, 8)
    for (long i = 0; i < (long)n; i++) {
      tree.knn(pts + 4 * i, k, ki.data(), kd.data());
      double mean[3] = {0, 0, 0};
      for (int j = 0; j < k; j++)
        for (int d = 0; d < 3; d++) mean[d] += (double)pts[4 * (size_t)ki[j] + d];
      for (int d = 0; d < 3; d++) mean[d] /= k;
      Mat3 cov = Mat3::zero();
      for (int j = 0; j < k; j++) {
        double c[3];
        for (int d = 0; d < 3; d++) c[d] = (double)pts[4 * (size_t)ki[j] + d] - mean[d];
        for (int r = 0; r < 3; r++)
          for (int s = 0; s < 3; s++) cov(r, s) += c[r] * c[s];
      }
      cov = scale(cov, 1.0 / k);  // divide by k, not k-1
      Mat3 out;
      if (method == ORC_REG_NONE) {
        out = cov;
      } else if (method == ORC_REG_FROBENIUS) {
        const double lambda = 1e-3;
        Mat3 C = add(cov, scale(Mat3::identity(), lambda));
        Mat3 C_inv = inverse(C);
        out = inverse(scale(C_inv, 1.0 / frobenius(C_inv)));
      } else {
        double sv[3];
        Mat3 U;
        eigen_sym3_desc(cov, sv, U);  // symmetric PSD: SVD == eigen-decomposition, U == V
        double values[3];
        if (method == ORC_REG_PLANE) {
          values[0] = 1; values[1] = 1; values[2] = 1e-3;
        } else if (method == ORC_REG_MIN_EIG) {
          for (int d = 0; d < 3; d++) values[d] = std::max(sv[d], 1e-3);
        } else {  // NORMALIZED_MIN_EIG
          const double mx = std::max(sv[0], std::max(sv[1], sv[2]));
          for (int d = 0; d < 3; d++) values[d] = std::max(sv[d] / mx, 1e-3);
        }
        out = Mat3::zero();
        for (int r = 0; r < 3; r++)
          for (int s = 0; s < 3; s++)
            for (int d = 0; d < 3; d++) out(r, s) += U(r, d) * values[d] * U(s, d);
      }
      std::memcpy(covs9 + 9 * (size_t)i, out.m, sizeof(out.m));
    }
  }
}

extern "C" int orc_covariances(const float* pts, size_t n, int k, int method, double* covs9, int num_threads) {
  if (k <= 0 || (size_t)k > n) return -1;
  const int saved = omp_get_max_threads();
  if (num_threads > 0) omp_set_num_threads(num_threads);
  KdTree tree;
  tree.build(pts, n, 4);
  covariances_impl(pts, n, tree, k, method, covs9);
  omp_set_num_threads(saved);
  return 0;
}

// ---- F3/F4/F5 ----------------------------------------------------------------------------------------------
struct GicpState {
  const float* src; size_t n;
  const float* tgt; size_t m;
  const double* src_cov; const double* tgt_cov;
  const KdTree* tgt_tree;
  double max_corr;
  std::vector<int> corr;
  std::vector<double> mahal;  // n x 9
};

static void iso_to_float_rows(const Iso3& x, float Tf[12]) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Tf[4 * r + c] = (float)x.R(r, c);
    Tf[4 * r + 3] = (float)x.t[r];
  }
}

// F3 update_correspondences: fp32 transform of the fp32 point, 1-NN in the target tree, d2 < max_corr^2,
// RCR = C_B + R C_A R^T, Mahalanobis = RCR^-1 (the 4x4 embedding with (3,3) patched to 1 then 0 reduces to this).
static void update_correspondences(GicpState& st, const Iso3& trans) {
  float Tf[12];
  iso_to_float_rows(trans, Tf);
  const float thr = (float)(st.max_corr * st.max_corr);
#pragma omp parallel for schedule(guided, 8)
  for (long i = 0; i < (long)st.n; i++) {
    const float* p = st.src + 4 * i;
    float q[3];
    for (int r = 0; r < 3; r++)
      q[r] = std::fmaf(Tf[4 * r], p[0], std::fmaf(Tf[4 * r + 1], p[1], std::fmaf(Tf[4 * r + 2], p[2], Tf[4 * r + 3])));
    int j;
    float d2;
    st.tgt_tree->knn(q, 1, &j, &d2);
    st.corr[i] = d2 < thr ? j : -1;
    if (st.corr[i] < 0) continue;
    Mat3 CA, CB;
    std::memcpy(CA.m, st.src_cov + 9 * (size_t)i, sizeof(CA.m));
    std::memcpy(CB.m, st.tgt_cov + 9 * (size_t)j, sizeof(CB.m));
    const Mat3 RCR = add(CB, mul(mul(trans.R, CA), transpose(trans.R)));
    const Mat3 M = inverse(RCR);
    std::memcpy(&st.mahal[9 * (size_t)i], M.m, sizeof(M.m));
  }
}

static double linearize(GicpState& st, const Iso3& trans, double* H36, double* b6) {
  update_correspondences(st, trans);
  const int nt = omp_get_max_threads();
  std::vector<double> Hs((size_t)nt * 36, 0.0), bs((size_t)nt * 6, 0.0), es(nt, 0.0);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)st.n; i++) {
    const int j = st.corr[i];
    if (j < 0) continue;
    const int tid = omp_get_thread_num();
    const Vec3 a = {{(double)st.src[4 * i], (double)st.src[4 * i + 1], (double)st.src[4 * i + 2]}};
    const Vec3 ta = mul(trans.R, a);
    const Vec3 tA = {{ta[0] + trans.t[0], ta[1] + trans.t[1], ta[2] + trans.t[2]}};
    const Vec3 e = {{(double)st.tgt[4 * (size_t)j] - tA[0], (double)st.tgt[4 * (size_t)j + 1] - tA[1],
                     (double)st.tgt[4 * (size_t)j + 2] - tA[2]}};
    Mat3 M;
    std::memcpy(M.m, &st.mahal[9 * (size_t)i], sizeof(M.m));
    const Vec3 Me = mul(M, e);
    es[tid] += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
    // J = [ skew(T p) | -I ]  (3 x 6), parameter order (rotation, translation)
    double J[3][6];
    const Mat3 S = skew(tA);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) { J[r][c] = S(r, c); J[r][3 + c] = (r == c) ? -1.0 : 0.0; }
    double MJ[3][6];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 6; c++) MJ[r][c] = M(r, 0) * J[0][c] + M(r, 1) * J[1][c] + M(r, 2) * J[2][c];
    double* H = &Hs[(size_t)tid * 36];
    double* b = &bs[(size_t)tid * 6];
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) H[6 * r + c] += J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c];
      b[r] += J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2];
    }
  }
  double err = 0;
  std::fill(H36, H36 + 36, 0.0);
  std::fill(b6, b6 + 6, 0.0);
  for (int t = 0; t < nt; t++) {
    err += es[t];
    for (int k = 0; k < 36; k++) H36[k] += Hs[(size_t)t * 36 + k];
    for (int k = 0; k < 6; k++) b6[k] += bs[(size_t)t * 6 + k];
  }
  return err;
}

static double compute_error(const float* src, size_t n, const float* tgt, const int* corr, const double* mahal,
                            const Iso3& trans) {
  const int nt = omp_get_max_threads();
  std::vector<double> es(nt, 0.0);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n; i++) {
    const int j = corr[i];
    if (j < 0) continue;
    const Vec3 a = {{(double)src[4 * i], (double)src[4 * i + 1], (double)src[4 * i + 2]}};
    const Vec3 ta = mul(trans.R, a);
    const Vec3 e = {{(double)tgt[4 * (size_t)j] - (ta[0] + trans.t[0]), (double)tgt[4 * (size_t)j + 1] - (ta[1] + trans.t[1]),
                     (double)tgt[4 * (size_t)j + 2] - (ta[2] + trans.t[2])}};
    Mat3 M;
    std::memcpy(M.m, mahal + 9 * (size_t)i, sizeof(M.m));
    const Vec3 Me = mul(M, e);
    es[omp_get_thread_num()] += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
  }
  double err = 0;
  for (double v : es) err += v;
  return err;
}

static Iso3 iso_from_T16d(const double* T) {
  Iso3 x;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) x.R(r, c) = T[4 * r + c];
    x.t[r] = T[4 * r + 3];
  }
  return x;
}

extern "C" int orc_gicp_linearize(const float* src, size_t n, const double* src_cov9, const float* tgt, size_t m,
                                  const double* tgt_cov9, const double T16[16], double max_corr, double* H36,
                                  double* b6, double* err, int* corr, double* mahal9) {
  KdTree tree;
  tree.build(tgt, m, 4);
  GicpState st{src, n, tgt, m, src_cov9, tgt_cov9, &tree, max_corr, std::vector<int>(n), std::vector<double>(9 * n)};
  *err = linearize(st, iso_from_T16d(T16), H36, b6);
  if (corr) std::memcpy(corr, st.corr.data(), n * sizeof(int));
  if (mahal9) std::memcpy(mahal9, st.mahal.data(), 9 * n * sizeof(double));
  return 0;
}

extern "C" int orc_gicp_error(const float* src, size_t n, const float* tgt, const int* corr, const double* mahal9,
                              const double T16[16], double* err) {
  *err = compute_error(src, n, tgt, corr, mahal9, iso_from_T16d(T16));
  return 0;
}

// F9 is_converged
static bool is_converged(const Iso3& delta, double rot_eps, double trans_eps) {
  double rmax = 0, tmax = 0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) rmax = std::max(rmax, std::fabs(delta.R(r, c) - (r == c ? 1.0 : 0.0)) / rot_eps);
    tmax = std::max(tmax, std::fabs(delta.t[r]) / trans_eps);
  }
  return std::max(rmax, tmax) < 1;
}

// F6/F7: LsqRegistration::computeTransformation + step_lm / step_gn
extern "C" int orc_gicp_align(const float* src, size_t n, const float* tgt, size_t m, const orc_gicp_params* p,
                              const float* guess16, orc_result* res) {
  if ((size_t)p->k_correspondences > n || (size_t)p->k_correspondences > m) return -1;
  const int saved = omp_get_max_threads();
  if (p->num_threads > 0) omp_set_num_threads(p->num_threads);
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  // F1: setInputSource / setInputTarget build one kd-tree each; F2: covariances are computed lazily at the
  // start of computeTransformation.
  KdTree src_tree, tgt_tree;
  src_tree.build(src, n, 4);
  tgt_tree.build(tgt, m, 4);
  std::vector<double> src_cov(9 * n), tgt_cov(9 * m);
  covariances_impl(src, n, src_tree, p->k_correspondences, p->regularization_method, src_cov.data());
  covariances_impl(tgt, m, tgt_tree, p->k_correspondences, p->regularization_method, tgt_cov.data());
  const auto t1 = clk::now();

  GicpState st{src, n, tgt, m, src_cov.data(), tgt_cov.data(), &tgt_tree, p->max_correspondence_distance,
               std::vector<int>(n), std::vector<double>(9 * n)};
  Iso3 x0 = from_rowmajor4x4f(guess16);
  double lm_lambda = -1.0;
  bool converged = false;
  int iters = 0;
  double H[36], b[6], last_err = 0;
  std::memset(res->H, 0, sizeof(res->H));
  for (int i = 0; i < p->max_iterations && !converged; i++) {
    iters = i + 1;
    Iso3 delta = Iso3::identity();
    bool ok = false;
    const double y0 = linearize(st, x0, H, b);
    if (p->optimizer == ORC_OPT_GN) {
      double d[6], nb[6];
      for (int k = 0; k < 6; k++) nb[k] = -b[k];
      ldlt6_solve(H, nb, d);
      delta = se3_exp(d);
      x0 = compose(delta, x0);
      std::memcpy(res->H, H, sizeof(H));
      last_err = y0;
      ok = true;
    } else {
      if (lm_lambda < 0.0) {
        double mx = 0;
        for (int k = 0; k < 6; k++) mx = std::max(mx, std::fabs(H[7 * k]));
        lm_lambda = p->lm_init_lambda_factor * mx;
      }
      double nu = 2.0;
      for (int it = 0; it < p->lm_max_iterations; it++) {
        double A[36], nb[6], d[6];
        std::memcpy(A, H, sizeof(A));
        for (int k = 0; k < 6; k++) { A[7 * k] += lm_lambda; nb[k] = -b[k]; }
        ldlt6_solve(A, nb, d);
        delta = se3_exp(d);
        const Iso3 xi = compose(delta, x0);
        const double yi = compute_error(src, n, tgt, st.corr.data(), st.mahal.data(), xi);
        double denom = 0;
        for (int k = 0; k < 6; k++) denom += d[k] * (lm_lambda * d[k] - b[k]);
        const double rho = (y0 - yi) / denom;
        if (getenv("ORC_DEBUG")) fprintf(stderr, "[orc] iter %d trial %d y0 %.9g yi %.9g rho %.4g lambda %.3g |d| %.3g\n", i, it, y0, yi, rho, lm_lambda, std::sqrt(d[0]*d[0]+d[1]*d[1]+d[2]*d[2]+d[3]*d[3]+d[4]*d[4]+d[5]*d[5]));
        if (rho < 0) {
          if (!p->force_iterations && is_converged(delta, p->rotation_epsilon, p->transformation_epsilon)) { ok = true; break; }
          lm_lambda = nu * lm_lambda;
          nu = 2 * nu;
          continue;
        }
        x0 = xi;
        lm_lambda = lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
        std::memcpy(res->H, H, sizeof(H));
        last_err = yi;
        ok = true;
        break;
      }
    }
    if (!ok && !p->force_iterations) break;  // "lm not converged!!" (benchmark mode keeps iterating: fixed work per pair)
    converged = !p->force_iterations && is_converged(delta, p->rotation_epsilon, p->transformation_epsilon);
  }
  const auto t2 = clk::now();
  to_rowmajor4x4f(x0, res->T);
  res->converged = converged ? 1 : 0;
  res->iterations = iters;
  res->error = last_err;
  res->trans_probability = 0;
  res->ms_setup = std::chrono::duration<double, std::milli>(t1 - t0).count();
  res->ms_align = std::chrono::duration<double, std::milli>(t2 - t1).count();
  omp_set_num_threads(saved);
  return 0;
}
