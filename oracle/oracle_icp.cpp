// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED.
//
// Restatement of pcl::IterativeClosestPoint<PointXYZI,PointXYZI> as the back end configures it — its DEFAULT registration
// method (/root/reference/graph_based_slam/config/graph_based_slam.param.yaml:9, graph_based_slam.cpp:142-151:
// setMaxCorrespondenceDistance(30), setMaximumIterations(100), setTransformationEpsilon(1e-8),
// setEuclideanFitnessEpsilon(1e-6), setRANSACIterations(0)) and calls it (graph_based_slam.cpp:315-322).  PCL is a system
// library absent from this image; the algorithm below is PCL's published one [UPSTREAM-RECALL]:
//   loop:  correspondences = 1-NN of every (already transformed) source point in the target, kept iff d2 <= max_dist^2
//          (CorrespondenceEstimation::determineCorrespondences); fewer than 3 -> not converged, stop;
//          incremental transform = least-squares rigid fit of the pairs (TransformationEstimationSVD = Umeyama without
//          scaling); the transformed source is moved by it (fp32, like pcl::transformPointCloud) and
//          final = incremental * final; ++iterations;
//          DefaultConvergenceCriteria: iterations >= max -> converged (ITERATIONS); incremental rotation cos >= rotation
//          threshold (1 - transformation_epsilon unless a rotation epsilon is set) AND incremental translation^2 <=
//          transformation_epsilon -> converged (TRANSFORM); |mse - prev| < 1e-12 -> ABS_MSE; |mse - prev| / prev <
//          euclidean_fitness_epsilon -> REL_MSE (mse = mean squared correspondence distance of this iteration).
// The rigid fit is evaluated with Horn's quaternion method in fp64 (the rotation Umeyama's SVD returns whenever the
// cross-covariance is not degenerate).
#include <omp.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "kdtree.hpp"
#include "linalg.hpp"
#include "oracle.h"

using namespace oracle;

extern "C" void orc_icp_default_params(orc_icp_params* p) {
  p->max_correspondence_distance = 30.0;
  p->max_iterations = 100;
  p->transformation_epsilon = 1e-8;
  p->euclidean_fitness_epsilon = 1e-6;
  p->rotation_epsilon = 0.0;
  p->num_threads = 0;
}

// sums[17]: n, sum p (3), sum q (3), sum p q^T (9, row-major p x q), sum d2  ->  rigid T16 (row-major double) with q ~ R p + t
extern "C" int orc_rigid_from_sums(const double* s, double* T16) {
  const double n = s[0];
  if (n < 3) return 1;
  const double mp[3] = {s[1] / n, s[2] / n, s[3] / n}, mq[3] = {s[4] / n, s[5] / n, s[6] / n};
  double S[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) S[i][j] = s[7 + 3 * i + j] - n * mp[i] * mq[j];
  const double N[16] = {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0],
                        S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2],
                        S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1],
                        S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]};
  double w[4], V[16];
  jacobi_eigen_sym<4>(N, w, V);
  int best = 0;
  for (int k = 1; k < 4; k++)
    if (w[k] > w[best]) best = k;
  double q[4] = {V[0 * 4 + best], V[1 * 4 + best], V[2 * 4 + best], V[3 * 4 + best]};
  const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (double& v : q) v /= nq;
  const double qw = q[0], x = q[1], y = q[2], z = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * qw), 2 * (x * z + y * qw),
                       2 * (x * y + z * qw), 1 - 2 * (x * x + z * z), 2 * (y * z - x * qw),
                       2 * (x * z - y * qw), 2 * (y * z + x * qw), 1 - 2 * (x * x + y * y)};
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T16[4 * r + c] = R[3 * r + c];
    T16[4 * r + 3] = mq[r] - (R[3 * r] * mp[0] + R[3 * r + 1] * mp[1] + R[3 * r + 2] * mp[2]);
  }
  T16[12] = T16[13] = T16[14] = 0;
  T16[15] = 1;
  return 0;
}

extern "C" int orc_icp_align(const float* src, size_t n, const float* tgt, size_t m, const orc_icp_params* p, const float* guess16, orc_result* res) {
  const int saved = omp_get_max_threads();
  if (p->num_threads > 0) omp_set_num_threads(p->num_threads);
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  KdTree tree;
  tree.build(tgt, m, 4);
  const auto t1 = clk::now();
  std::vector<float> cur(4 * n), nxt(4 * n);
  orc_transform(src, n, guess16, cur.data());  // input_transformed = guess * input (fp32)
  float final_T[16];
  std::memcpy(final_T, guess16, sizeof(final_T));
  const double max_d2 = p->max_correspondence_distance * p->max_correspondence_distance;
  const double rot_thr = p->rotation_epsilon > 0 ? p->rotation_epsilon : 1.0 - p->transformation_epsilon;
  int iterations = 0;
  bool converged = false;
  double prev_mse = 1.7976931348623157e308, mse = 0;
  for (;;) {
    const int nt = omp_get_max_threads();
    std::vector<double> part((size_t)nt * 17, 0.0);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
      int j;
      float d2;
      if (tree.knn(&cur[4 * i], 1, &j, &d2) < 1 || (double)d2 > max_d2) continue;
      double* a = &part[(size_t)omp_get_thread_num() * 17];
      const float* pp = &cur[4 * i];
      const float* qq = tgt + 4 * (size_t)j;
      a[0] += 1;
      for (int k = 0; k < 3; k++) { a[1 + k] += pp[k]; a[4 + k] += qq[k]; }
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) a[7 + 3 * r + c] += (double)pp[r] * (double)qq[c];
      a[16] += d2;
    }
    double s[17] = {0};
    for (int t = 0; t < nt; t++)
      for (int k = 0; k < 17; k++) s[k] += part[(size_t)t * 17 + k];
    if (s[0] < 3) { converged = false; break; }  // "Not enough correspondences found"
    double Td[16];
    orc_rigid_from_sums(s, Td);
    float Tf[16];
    for (int k = 0; k < 16; k++) Tf[k] = (float)Td[k];
    orc_transform(cur.data(), n, Tf, nxt.data());
    cur.swap(nxt);
    float F[16];  // final = incremental * final (Matrix4f product)
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        float acc = 0.f;
        for (int k = 0; k < 4; k++) acc += Tf[4 * r + k] * final_T[4 * k + c];
        F[4 * r + c] = acc;
      }
    std::memcpy(final_T, F, sizeof(F));
    iterations++;
    // DefaultConvergenceCriteria::hasConverged (max_iterations_similar_transforms_ = 0)
    if (iterations >= p->max_iterations) { converged = true; break; }
    const double cos_angle = 0.5 * ((double)Tf[0] + (double)Tf[5] + (double)Tf[10] - 1.0);
    const double tr2 = (double)Tf[3] * Tf[3] + (double)Tf[7] * Tf[7] + (double)Tf[11] * Tf[11];
    if (cos_angle >= rot_thr && tr2 <= p->transformation_epsilon) { converged = true; break; }
    mse = s[16] / s[0];
    if (std::fabs(mse - prev_mse) < 1e-12) { converged = true; break; }
    if (std::fabs(mse - prev_mse) / prev_mse < p->euclidean_fitness_epsilon) { converged = true; break; }
    prev_mse = mse;
  }
  const auto t2 = clk::now();
  std::memset(res, 0, sizeof(*res));
  std::memcpy(res->T, final_T, sizeof(final_T));
  res->converged = converged ? 1 : 0;
  res->iterations = iterations;
  res->error = mse;
  res->ms_setup = std::chrono::duration<double, std::milli>(t1 - t0).count();
  res->ms_align = std::chrono::duration<double, std::milli>(t2 - t1).count();
  omp_set_num_threads(saved);
  return 0;
}
