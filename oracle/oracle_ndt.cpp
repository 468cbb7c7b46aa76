// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED.
//
// Restatement of pclomp::NormalDistributionsTransform<PointXYZI,PointXYZI> (ndt_omp: PCL's NDT with OpenMP,
// fp32 per-point derivative maths and direct voxel neighbourhoods) + pclomp::VoxelGridCovariance as the
// reference drives them from /root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:55-72
// (construction: eps, step size, resolution, max iterations, DIRECT7, threads), :149,:212 (setInputTarget),
// :162,:165 (setInputSource / align) and /root/reference/graph_based_slam/src/graph_based_slam.cpp:103-119.
// The ndt_omp sources are an un-vendored submodule (/root/reference/.gitmodules:1-3); the algorithm follows
// SURVEY.md §8a rows N1-N5 [UPSTREAM-RECALL] and Magnusson 2009 (eq. 6.9-6.21), More & Thuente 1994.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>

#include "linalg.hpp"
#include "oracle.h"

using namespace oracle;

extern "C" void orc_ndt_default_params(orc_ndt_params* p) {
  p->transformation_epsilon = 0.01;
  p->step_size = 0.1;
  p->resolution = 2.0;
  p->max_iterations = 64;
  p->outlier_ratio = 0.55;
  p->min_points_per_voxel = 6;
  p->min_covar_eigvalue_mult = 0.01;
  p->legacy_mt_skip = 1;  /* the ndt_omp lineage (PCL 1.8): see oracle.h */
  p->num_threads = 0;
}

namespace {

struct Leaf {
  int nr_points = 0;
  double mean[3] = {0, 0, 0};
  Mat3 cov = Mat3::zero();
  Mat3 icov = Mat3::zero();
};

// N1 — pclomp::VoxelGridCovariance::applyFilter (filter(true) from NDT::init()).
struct NdtGrid {
  float inv_leaf;
  int min_b[3], max_b[3], div_b[3], mul[3];
  std::unordered_map<int64_t, Leaf> leaves;
  int min_points;

  void build(const float* tgt, size_t m, const orc_ndt_params& p) {
    min_points = p.min_points_per_voxel;
    inv_leaf = 1.0f / (float)p.resolution;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < m; i++)
      for (int d = 0; d < 3; d++) { mn[d] = std::min(mn[d], tgt[4 * i + d]); mx[d] = std::max(mx[d], tgt[4 * i + d]); }
    for (int d = 0; d < 3; d++) {
      min_b[d] = (int)std::floor(mn[d] * inv_leaf);
      max_b[d] = (int)std::floor(mx[d] * inv_leaf);
      div_b[d] = max_b[d] - min_b[d] + 1;
    }
    mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];
    leaves.clear();
    leaves.reserve(m / 4 + 16);
    std::unordered_map<int64_t, Mat3> sums;  // Σ p p^T kept beside the leaf until finalised
    for (size_t i = 0; i < m; i++) {
      const float* p4 = tgt + 4 * i;
      const int i0 = (int)(std::floor(p4[0] * inv_leaf) - (float)min_b[0]);
      const int i1 = (int)(std::floor(p4[1] * inv_leaf) - (float)min_b[1]);
      const int i2 = (int)(std::floor(p4[2] * inv_leaf) - (float)min_b[2]);
      const int64_t idx = (int64_t)i0 * mul[0] + (int64_t)i1 * mul[1] + (int64_t)i2 * mul[2];
      Leaf& leaf = leaves[idx];
      const double q[3] = {(double)p4[0], (double)p4[1], (double)p4[2]};
      for (int r = 0; r < 3; r++) {
        leaf.mean[r] += q[r];
        for (int c = 0; c < 3; c++) leaf.cov(r, c) += q[r] * q[c];
      }
      leaf.nr_points++;
    }
    for (auto& kv : leaves) {
      Leaf& leaf = kv.second;
      const double n = (double)leaf.nr_points;
      double pt_sum[3];
      for (int d = 0; d < 3; d++) { pt_sum[d] = leaf.mean[d]; leaf.mean[d] /= n; }
      if (leaf.nr_points < min_points) continue;
      // single-pass covariance: (Σpp^T − 2·pt_sum·μ^T)/n + μμ^T, then ·(n−1)/n
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          leaf.cov(r, c) = (leaf.cov(r, c) - 2 * (pt_sum[r] * leaf.mean[c])) / n + leaf.mean[r] * leaf.mean[c];
      leaf.cov = scale(leaf.cov, (n - 1.0) / n);
      // eigen-decomposition, ascending eigenvalues (SelfAdjointEigenSolver order)
      double w[3], V[9];
      Mat3 sym = leaf.cov;  // the solver reads the lower triangle; the matrix is symmetric by construction
      jacobi_eigen_sym<3>(sym.m, w, V);
      int o[3] = {0, 1, 2};
      std::sort(o, o + 3, [&](int a, int b) { return w[a] < w[b]; });
      double ev[3] = {w[o[0]], w[o[1]], w[o[2]]};
      if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) { leaf.nr_points = -1; continue; }
      const double min_ev = p.min_covar_eigvalue_mult * ev[2];
      if (ev[0] < min_ev) {
        ev[0] = min_ev;
        if (ev[1] < min_ev) ev[1] = min_ev;
        Mat3 E;
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) E(r, c) = V[r * 3 + o[c]];
        Mat3 rebuilt = Mat3::zero();  // evecs * diag(ev) * evecs^-1 (orthonormal => transpose)
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++)
            for (int d = 0; d < 3; d++) rebuilt(r, c) += E(r, d) * ev[d] * E(c, d);
        leaf.cov = rebuilt;
      }
      leaf.icov = inverse(leaf.cov);
      double mxc = -INFINITY, mnc = INFINITY;
      for (int k = 0; k < 9; k++) { mxc = std::max(mxc, leaf.icov.m[k]); mnc = std::min(mnc, leaf.icov.m[k]); }
      if (mxc == INFINITY || mnc == -INFINITY) leaf.nr_points = -1;
    }
  }

  // getNeighborhoodAtPoint7: the voxel holding the point and its 6 face neighbours, inside the grid bounds,
  // present and with nr_points >= min_points_per_voxel_.
  int neighborhood7(const float x[3], const Leaf* out[7]) const {
    static const int disp[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    int ijk[3];
    for (int d = 0; d < 3; d++) ijk[d] = (int)std::floor(x[d] * inv_leaf);
    int cnt = 0;
    for (int k = 0; k < 7; k++) {
      bool inside = true;
      int64_t idx = 0;
      for (int d = 0; d < 3; d++) {
        const int c = ijk[d] + disp[k][d];
        if (c < min_b[d] || c > max_b[d]) inside = false;
        idx += (int64_t)(c - min_b[d]) * mul[d];
      }
      if (!inside) continue;
      auto it = leaves.find(idx);
      if (it != leaves.end() && it->second.nr_points >= min_points) out[cnt++] = &it->second;
    }
    return cnt;
  }
};

struct AngleTables {
  double j[8][3];   // j_ang_a_ .. j_ang_h_
  double h[15][3];  // h_ang_a2_, a3_, b2_, b3_, c2_, c3_, d1_, d2_, d3_, e1_, e2_, e3_, f1_, f2_, f3_
  float jf[8][3];
  float hf[15][3];
};

// computeAngleDerivatives (eq. 6.19 / 6.21 [Magnusson 2009]); R = Rx(roll) Ry(pitch) Rz(yaw).
void angle_derivatives(const double p[6], AngleTables& t) {
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
  const double J[8][3] = {
      {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},
      {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
      {(-sy * cz), sy * sz, cy},
      {sx * cy * cz, (-sx * cy * sz), sx * sy},
      {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
      {(-cy * sz), (-cy * cz), 0},
      {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},
      {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
  const double Hh[15][3] = {
      {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},     // a2
      {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},  // a3
      {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                        // b2
      {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                        // b3
      {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},            // c2
      {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},            // c3
      {(-cy * cz), (cy * sz), (sy)},                                       // d1
      {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                        // d2
      {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                       // d3
      {(sy * sz), (sy * cz), 0},                                           // e1
      {(-sx * cy * sz), (-sx * cy * cz), 0},                               // e2
      {(cx * cy * sz), (cx * cy * cz), 0},                                 // e3
      {(-cy * cz), (cy * sz), 0},                                          // f1
      {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},           // f2
      {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};          // f3
  for (int a = 0; a < 8; a++)
    for (int d = 0; d < 3; d++) { t.j[a][d] = J[a][d]; t.jf[a][d] = (float)J[a][d]; }
  for (int a = 0; a < 15; a++)
    for (int d = 0; d < 3; d++) { t.h[a][d] = Hh[a][d]; t.hf[a][d] = (float)Hh[a][d]; }
}

// p6 -> Matrix4f exactly as  Translation<float> * AngleAxis<float>(roll,X) * AngleAxis<float>(pitch,Y) *
// AngleAxis<float>(yaw,Z)  builds it (fp32 products, left to right).
void pose_to_matrix_f(const double p[6], float T[16]) {
  const float a = (float)p[3], b = (float)p[4], c = (float)p[5];
  const float ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
  const float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
  const float Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
  const float Rz[9] = {cc, -sc, 0, sc, cc, 0, 0, 0, 1};
  float A[9], R[9];
  for (int r = 0; r < 3; r++)
    for (int s = 0; s < 3; s++) A[3 * r + s] = Rx[3 * r] * Ry[s] + Rx[3 * r + 1] * Ry[3 + s] + Rx[3 * r + 2] * Ry[6 + s];
  for (int r = 0; r < 3; r++)
    for (int s = 0; s < 3; s++) R[3 * r + s] = A[3 * r] * Rz[s] + A[3 * r + 1] * Rz[3 + s] + A[3 * r + 2] * Rz[6 + s];
  for (int r = 0; r < 3; r++) {
    for (int s = 0; s < 3; s++) T[4 * r + s] = R[3 * r + s];
    T[4 * r + 3] = (float)p[r];
  }
  T[12] = T[13] = T[14] = 0;
  T[15] = 1;
}

// Eigen::Matrix3f::eulerAngles(0,1,2) (Eigen 3.3 algorithm), fp32.
void euler_angles_012(const float T[16], float res[3]) {
  auto coeff = [&](int r, int c) { return T[4 * r + c]; };
  const int i = 0, j = 1, k = 2;  // even permutation: odd = 0
  res[0] = std::atan2(coeff(j, k), coeff(k, k));
  const float c2 = std::sqrt(coeff(i, i) * coeff(i, i) + coeff(i, j) * coeff(i, j));
  if (res[0] > 0.f) {
    if (res[0] > 0.f) res[0] -= (float)M_PI; else res[0] += (float)M_PI;
    res[1] = std::atan2(-coeff(i, k), -c2);
  } else {
    res[1] = std::atan2(-coeff(i, k), c2);
  }
  const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * coeff(k, i) - c1 * coeff(j, i), c1 * coeff(j, j) - s1 * coeff(k, j));
  for (int d = 0; d < 3; d++) res[d] = -res[d];
}

struct Ndt {
  const float* src; size_t n;
  NdtGrid grid;
  double gauss_d1, gauss_d2;
  AngleTables ang;

  void init_gauss(const orc_ndt_params& p) {
    const double gauss_c1 = 10 * (1 - p.outlier_ratio);
    const double gauss_c2 = p.outlier_ratio / std::pow(p.resolution, 3);
    const double gauss_d3 = -std::log(gauss_c2);
    gauss_d1 = -std::log(gauss_c1 + gauss_c2) - gauss_d3;
    gauss_d2 = -2 * std::log((-std::log(gauss_c1 * std::exp(-0.5) + gauss_c2) - gauss_d3) / gauss_d1);
  }

  // N3 — computeDerivatives with ndt_omp's fp32 per-point maths (point_gradient 4x6 / point_hessian 24x6 floats),
  // fp64 accumulators.  trans = the fp32-transformed source cloud.
  double derivatives(const float* trans, const double p[6], double* grad6, double* hess36, bool compute_hessian) {
    angle_derivatives(p, ang);
    const int nt = omp_get_max_threads();
    std::vector<double> sc(nt, 0.0), gr((size_t)nt * 6, 0.0), he((size_t)nt * 36, 0.0);
    const float d1 = (float)gauss_d1, d2 = (float)gauss_d2;
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < (long)n; idx++) {
      const int tid = omp_get_thread_num();
      const float* xt = trans + 4 * idx;
      const Leaf* nb[7];
      const int cnt = grid.neighborhood7(xt, nb);
      if (cnt == 0) continue;
      const float x[3] = {src[4 * idx], src[4 * idx + 1], src[4 * idx + 2]};
      // computePointDerivatives (float): point_gradient (3 rows used) and point_hessian blocks a..f
      float pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
      float xj[8];
      for (int a = 0; a < 8; a++) xj[a] = ang.jf[a][0] * x[0] + ang.jf[a][1] * x[1] + ang.jf[a][2] * x[2];
      pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4];
      pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
      float ph[6][6][3];  // ph[i][j] = d2T/dpi dpj (3-vector)
      std::memset(ph, 0, sizeof(ph));
      if (compute_hessian) {
        float xh[15];
        for (int a = 0; a < 15; a++) xh[a] = ang.hf[a][0] * x[0] + ang.hf[a][1] * x[1] + ang.hf[a][2] * x[2];
        const float A[3] = {0, xh[0], xh[1]}, B[3] = {0, xh[2], xh[3]}, C[3] = {0, xh[4], xh[5]};
        const float D[3] = {xh[6], xh[7], xh[8]}, E[3] = {xh[9], xh[10], xh[11]}, F[3] = {xh[12], xh[13], xh[14]};
        for (int d = 0; d < 3; d++) {
          ph[3][3][d] = A[d]; ph[4][3][d] = B[d]; ph[5][3][d] = C[d];
          ph[3][4][d] = B[d]; ph[4][4][d] = D[d]; ph[5][4][d] = E[d];
          ph[3][5][d] = C[d]; ph[4][5][d] = E[d]; ph[5][5][d] = F[d];
        }
      }
      for (int c = 0; c < cnt; c++) {
        const Leaf& cell = *nb[c];
        // x_trans in double, then cast to float (updateDerivatives, ndt_omp float overload)
        float xtr[3], ci[3][3];
        for (int d = 0; d < 3; d++) xtr[d] = (float)((double)xt[d] - cell.mean[d]);
        for (int r = 0; r < 3; r++)
          for (int s = 0; s < 3; s++) ci[r][s] = (float)cell.icov(r, s);
        float xc[3];  // x_trans4 * c_inv4 (row vector times matrix)
        for (int s = 0; s < 3; s++) xc[s] = xtr[0] * ci[0][s] + xtr[1] * ci[1][s] + xtr[2] * ci[2][s];
        float e_x_cov_x = std::exp(-d2 * (xtr[0] * xc[0] + xtr[1] * xc[1] + xtr[2] * xc[2]) * 0.5f);
        const float score_inc = -d1 * e_x_cov_x;
        e_x_cov_x = d2 * e_x_cov_x;
        if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
        sc[tid] += (double)score_inc;
        e_x_cov_x *= d1;
        float cg[3][6];  // c_inv4 * point_gradient4
        for (int r = 0; r < 3; r++)
          for (int k = 0; k < 6; k++) cg[r][k] = ci[r][0] * pg[0][k] + ci[r][1] * pg[1][k] + ci[r][2] * pg[2][k];
        float xcg[6];  // x_trans4 * (c_inv4 * point_gradient4)
        for (int k = 0; k < 6; k++) xcg[k] = xtr[0] * cg[0][k] + xtr[1] * cg[1][k] + xtr[2] * cg[2][k];
        for (int k = 0; k < 6; k++) gr[(size_t)tid * 6 + k] += (double)(e_x_cov_x * xcg[k]);
        if (compute_hessian) {
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
              const float xh_ij = xc[0] * ph[i][j][0] + xc[1] * ph[i][j][1] + xc[2] * ph[i][j][2];
              const float gg = pg[0][j] * cg[0][i] + pg[1][j] * cg[1][i] + pg[2][j] * cg[2][i];
              he[(size_t)tid * 36 + 6 * i + j] += (double)(e_x_cov_x * (-d2 * xcg[i] * xcg[j] + xh_ij + gg));
            }
        }
      }
    }
    double score = 0;
    std::fill(grad6, grad6 + 6, 0.0);
    if (compute_hessian) std::fill(hess36, hess36 + 36, 0.0);
    for (int t = 0; t < nt; t++) {
      score += sc[t];
      for (int k = 0; k < 6; k++) grad6[k] += gr[(size_t)t * 6 + k];
      if (compute_hessian)
        for (int k = 0; k < 36; k++) hess36[k] += he[(size_t)t * 36 + k];
    }
    return score;
  }

  // computeHessian / updateHessian — fp64 per-point maths, uses the angle tables of the last derivatives() call.
  void hessian_fp64(const float* trans, double* hess36) {
    const int nt = omp_get_max_threads();
    std::vector<double> he((size_t)nt * 36, 0.0);
#pragma omp parallel for schedule(static)
    for (long idx = 0; idx < (long)n; idx++) {
      const int tid = omp_get_thread_num();
      const float* xt = trans + 4 * idx;
      const Leaf* nb[7];
      const int cnt = grid.neighborhood7(xt, nb);
      if (cnt == 0) continue;
      const double x[3] = {(double)src[4 * idx], (double)src[4 * idx + 1], (double)src[4 * idx + 2]};
      double pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
      double xj[8], xh[15];
      for (int a = 0; a < 8; a++) xj[a] = x[0] * ang.j[a][0] + x[1] * ang.j[a][1] + x[2] * ang.j[a][2];
      for (int a = 0; a < 15; a++) xh[a] = x[0] * ang.h[a][0] + x[1] * ang.h[a][1] + x[2] * ang.h[a][2];
      pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4];
      pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
      double ph[6][6][3];
      std::memset(ph, 0, sizeof(ph));
      const double A[3] = {0, xh[0], xh[1]}, B[3] = {0, xh[2], xh[3]}, C[3] = {0, xh[4], xh[5]};
      const double D[3] = {xh[6], xh[7], xh[8]}, E[3] = {xh[9], xh[10], xh[11]}, F[3] = {xh[12], xh[13], xh[14]};
      for (int d = 0; d < 3; d++) {
        ph[3][3][d] = A[d]; ph[4][3][d] = B[d]; ph[5][3][d] = C[d];
        ph[3][4][d] = B[d]; ph[4][4][d] = D[d]; ph[5][4][d] = E[d];
        ph[3][5][d] = C[d]; ph[4][5][d] = E[d]; ph[5][5][d] = F[d];
      }
      for (int c = 0; c < cnt; c++) {
        const Leaf& cell = *nb[c];
        double xtr[3];
        for (int d = 0; d < 3; d++) xtr[d] = (double)xt[d] - cell.mean[d];
        double cx[3];  // c_inv * x_trans
        for (int r = 0; r < 3; r++) cx[r] = cell.icov(r, 0) * xtr[0] + cell.icov(r, 1) * xtr[1] + cell.icov(r, 2) * xtr[2];
        double e_x_cov_x = gauss_d2 * std::exp(-gauss_d2 * (xtr[0] * cx[0] + xtr[1] * cx[1] + xtr[2] * cx[2]) / 2);
        if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
        e_x_cov_x *= gauss_d1;
        double cg[3][6], xcg[6];
        for (int r = 0; r < 3; r++)
          for (int k = 0; k < 6; k++) cg[r][k] = cell.icov(r, 0) * pg[0][k] + cell.icov(r, 1) * pg[1][k] + cell.icov(r, 2) * pg[2][k];
        for (int k = 0; k < 6; k++) xcg[k] = xtr[0] * cg[0][k] + xtr[1] * cg[1][k] + xtr[2] * cg[2][k];
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) {
            double ch[3];  // c_inv * point_hessian block (3i, j)
            for (int r = 0; r < 3; r++)
              ch[r] = cell.icov(r, 0) * ph[i][j][0] + cell.icov(r, 1) * ph[i][j][1] + cell.icov(r, 2) * ph[i][j][2];
            const double xh_ij = xtr[0] * ch[0] + xtr[1] * ch[1] + xtr[2] * ch[2];
            const double gg = pg[0][j] * cg[0][i] + pg[1][j] * cg[1][i] + pg[2][j] * cg[2][i];
            he[(size_t)tid * 36 + 6 * i + j] += e_x_cov_x * (-gauss_d2 * xcg[i] * xcg[j] + xh_ij + gg);
          }
      }
    }
    std::fill(hess36, hess36 + 36, 0.0);
    for (int t = 0; t < nt; t++)
      for (int k = 0; k < 36; k++) hess36[k] += he[(size_t)t * 36 + k];
  }
};

// ---- N4: More-Thuente --------------------------------------------------------------------------------------
inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

bool update_interval_mt(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t,
                        double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  else if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  else if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

double trial_value_selection_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t,
                                double f_t, double g_t) {
  if (f_t > f_l) {  // Case 1
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {  // Case 2
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (std::fabs(g_t) <= std::fabs(g_l)) {  // Case 3
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_t_next = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    if (a_t > a_l) return std::min(a_t + 0.66 * (a_u - a_t), a_t_next);
    return std::max(a_t + 0.66 * (a_u - a_t), a_t_next);
  }
  // Case 4
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  const double w = std::sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

void transform_f(const float* src, size_t n, const float* T, float* out) { orc_transform(src, n, T, out); }

}  // namespace

extern "C" int orc_ndt_cells(const float* tgt, size_t m, const orc_ndt_params* p, int64_t* leaf_idx, int* leaf_n,
                             double* mean3, double* cov9, double* icov9, int* min_b3, int* div_b3, size_t* n_leaves) {
  NdtGrid g;
  g.build(tgt, m, *p);
  std::vector<int64_t> keys;
  keys.reserve(g.leaves.size());
  for (auto& kv : g.leaves) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  size_t o = 0;
  for (int64_t k : keys) {
    const Leaf& l = g.leaves[k];
    leaf_idx[o] = k;
    leaf_n[o] = l.nr_points;
    for (int d = 0; d < 3; d++) mean3[3 * o + d] = l.mean[d];
    std::memcpy(cov9 + 9 * o, l.cov.m, sizeof(l.cov.m));
    std::memcpy(icov9 + 9 * o, l.icov.m, sizeof(l.icov.m));
    o++;
  }
  for (int d = 0; d < 3; d++) { min_b3[d] = g.min_b[d]; div_b3[d] = g.div_b[d]; }
  *n_leaves = o;
  return 0;
}

extern "C" int orc_ndt_derivatives(const float* src, size_t n, const float* tgt, size_t m, const orc_ndt_params* p,
                                   const double p6[6], double* score, double* grad6, double* hess36,
                                   int hessian_fp64) {
  const int saved = omp_get_max_threads();
  if (p->num_threads > 0) omp_set_num_threads(p->num_threads);
  Ndt ndt;
  ndt.src = src; ndt.n = n;
  ndt.grid.build(tgt, m, *p);
  ndt.init_gauss(*p);
  float T[16];
  pose_to_matrix_f(p6, T);
  std::vector<float> trans(4 * n);
  transform_f(src, n, T, trans.data());
  *score = ndt.derivatives(trans.data(), p6, grad6, hess36, true);
  if (hessian_fp64) ndt.hessian_fp64(trans.data(), hess36);
  omp_set_num_threads(saved);
  return 0;
}

// N2 — computeTransformation + N4 computeStepLengthMT
extern "C" int orc_ndt_align(const float* src, size_t n, const float* tgt, size_t m, const orc_ndt_params* p,
                             const float* guess16, orc_result* res) {
  const int saved = omp_get_max_threads();
  if (p->num_threads > 0) omp_set_num_threads(p->num_threads);
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  Ndt ndt;
  ndt.src = src; ndt.n = n;
  ndt.grid.build(tgt, m, *p);
  const auto t1 = clk::now();
  ndt.init_gauss(*p);

  int nr_iterations = 0;
  bool converged = false;
  float final_T[16];
  std::memcpy(final_T, guess16, sizeof(final_T));
  std::vector<float> trans(4 * n);
  bool guess_is_identity = true;
  for (int k = 0; k < 16; k++)
    if (guess16[k] != ((k % 5 == 0) ? 1.f : 0.f)) guess_is_identity = false;
  if (!guess_is_identity) transform_f(src, n, guess16, trans.data());
  else std::memcpy(trans.data(), src, 4 * n * sizeof(float));

  double pv[6], delta_p[6], grad[6], hess[36];
  float eul[3];
  euler_angles_012(final_T, eul);
  pv[0] = (double)final_T[3]; pv[1] = (double)final_T[7]; pv[2] = (double)final_T[11];
  pv[3] = (double)eul[0]; pv[4] = (double)eul[1]; pv[5] = (double)eul[2];

  double score = ndt.derivatives(trans.data(), pv, grad, hess, true);
  double trans_probability = 0;
  while (!converged) {
    double ng[6];
    for (int k = 0; k < 6; k++) ng[k] = -grad[k];
    svd6_solve_sym(hess, ng, delta_p);
    double delta_p_norm = 0;
    for (int k = 0; k < 6; k++) delta_p_norm += delta_p[k] * delta_p[k];
    delta_p_norm = std::sqrt(delta_p_norm);
    if (delta_p_norm == 0 || delta_p_norm != delta_p_norm) {
      trans_probability = score / (double)n;
      converged = delta_p_norm == delta_p_norm;
      goto done;
    }
    for (int k = 0; k < 6; k++) delta_p[k] /= delta_p_norm;
    {
      // ---- computeStepLengthMT(p, step_dir = delta_p, step_init = delta_p_norm, step_max, step_min, ...) ----
      double* step_dir = delta_p;
      const double step_init = delta_p_norm, step_max = p->step_size, step_min = p->transformation_epsilon / 2;
      const double phi_0 = -score;
      double d_phi_0 = 0;
      for (int k = 0; k < 6; k++) d_phi_0 -= grad[k] * step_dir[k];
      double a_t = 0;
      if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) { delta_p_norm = 0; goto step_done; }
        d_phi_0 *= -1;
        for (int k = 0; k < 6; k++) step_dir[k] *= -1;
      }
      {
        const int max_step_iterations = 10;
        int step_iterations = 0;
        const double mu = 1.e-4, nu = 0.9;
        double a_l = 0, a_u = 0;
        double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
        double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu), g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
        bool interval_converged = p->legacy_mt_skip ? ((step_max - step_min) > 0) : ((step_max - step_min) < 0);
        bool open_interval = true;
        a_t = step_init;
        a_t = std::min(a_t, step_max);
        a_t = std::max(a_t, step_min);
        double x_t[6];
        for (int k = 0; k < 6; k++) x_t[k] = pv[k] + step_dir[k] * a_t;
        pose_to_matrix_f(x_t, final_T);
        transform_f(src, n, final_T, trans.data());
        score = ndt.derivatives(trans.data(), x_t, grad, hess, true);
        double phi_t = -score, d_phi_t = 0;
        for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * step_dir[k];
        double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        while (!interval_converged && step_iterations < max_step_iterations &&
               !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
          if (open_interval) a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
          else a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
          a_t = std::min(a_t, step_max);
          a_t = std::max(a_t, step_min);
          for (int k = 0; k < 6; k++) x_t[k] = pv[k] + step_dir[k] * a_t;
          pose_to_matrix_f(x_t, final_T);
          transform_f(src, n, final_T, trans.data());
          score = ndt.derivatives(trans.data(), x_t, grad, hess, false);
          phi_t = -score;
          d_phi_t = 0;
          for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * step_dir[k];
          psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
          d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
          if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = false;
            f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
            f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
          }
          if (open_interval) interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
          else interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
          step_iterations++;
        }
        if (step_iterations) ndt.hessian_fp64(trans.data(), hess);
        delta_p_norm = a_t;
      }
    step_done:;
    }
    for (int k = 0; k < 6; k++) { delta_p[k] *= delta_p_norm; pv[k] += delta_p[k]; }
    if (nr_iterations > p->max_iterations || (nr_iterations && (std::fabs(delta_p_norm) < p->transformation_epsilon)))
      converged = true;
    nr_iterations++;
  }
  trans_probability = score / (double)n;
done:
  const auto t2 = clk::now();
  std::memcpy(res->T, final_T, sizeof(final_T));
  res->converged = converged ? 1 : 0;
  res->iterations = nr_iterations;
  res->error = score;
  std::memcpy(res->H, hess, sizeof(hess));
  res->trans_probability = trans_probability;
  res->ms_setup = std::chrono::duration<double, std::milli>(t1 - t0).count();
  res->ms_align = std::chrono::duration<double, std::milli>(t2 - t1).count();
  omp_set_num_threads(saved);
  return 0;
}
