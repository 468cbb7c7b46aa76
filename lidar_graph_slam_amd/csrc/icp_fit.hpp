// The scalar part of pcl::IterativeClosestPoint (graph_based_slam.cpp:142-151): the rigid fit from the 17 sums of
// an iteration (Horn's quaternion method = the rotation Umeyama's SVD returns) and PCL's DefaultConvergenceCriteria, written ONCE
// as a resumable per-pair state.  Everything is MI_HD: k_icp_advance advances every pair of a batch on the device - a
// registration object is a batch of one - and tests/host_math_harness.cpp compiles the same text on the CPU.
#pragma once
#include <cmath>

#include "devmath.hpp"

namespace mi355 {

constexpr int kIcpAcc = 17;  // count, sum p (3), sum q (3), sum p q^T (9), sum d2

struct IcpConsts {
  int max_iterations;
  double transformation_epsilon, euclidean_fitness_epsilon, rot_threshold;
};

struct IcpPairState {
  float Tinc[16];     // what the next k_icp_step moves the cloud by: the guess, then each iteration's fit
  float final_T[16];  // final_transformation_
  int iterations, converged, active, moved;  // moved: the source has been through a step (the next one reads `cur`, not the input)
  int n, n_target, pad0, pad1;
  double prev_mse, mse;
};

// least-squares rigid fit q ~ R p + t from the sums (Horn 1987: largest eigenvector of the 4x4 quaternion matrix), fp64.
// Every index is a compile-time constant once the loops are unrolled (only the sweep loop stays rolled): registers on the device.
MI_HD bool icp_rigid_from_sums(const double* s, double* T16) {
  const double n = s[0];
  if (n < 3) return false;
  const double mp[3] = {s[1] / n, s[2] / n, s[3] / n}, mq[3] = {s[4] / n, s[5] / n, s[6] / n};
  double S[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) S[i][j] = s[7 + 3 * i + j] - n * mp[i] * mq[j];
  double A[16] = {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0],
                  S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2],
                  S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1],
                  S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]};
  double V[16];
#pragma unroll
  for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {  // cyclic Jacobi, symmetric 4x4
    double off = 0, diag = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) { const double v = A[4 * i + j] * A[4 * i + j]; if (i == j) diag += v; else off += v; }
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        const double apq = A[4 * p + q];
        if (apq != 0.0) {
          const double tau = (A[4 * q + q] - A[4 * p + p]) / (2.0 * apq);
          const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
#pragma unroll
          for (int k = 0; k < 4; k++) { const double a = A[4 * k + p], b = A[4 * k + q]; A[4 * k + p] = c * a - sn * b; A[4 * k + q] = sn * a + c * b; }
#pragma unroll
          for (int k = 0; k < 4; k++) { const double a = A[4 * p + k], b = A[4 * q + k]; A[4 * p + k] = c * a - sn * b; A[4 * q + k] = sn * a + c * b; }
#pragma unroll
          for (int k = 0; k < 4; k++) { const double a = V[4 * k + p], b = V[4 * k + q]; V[4 * k + p] = c * a - sn * b; V[4 * k + q] = sn * a + c * b; }
        }
      }
  }
  // the eigenvector of the largest eigenvalue (first of equals), selected without run-time indexing
  double best = A[0];
  double q[4] = {V[0], V[4], V[8], V[12]};
#pragma unroll
  for (int k = 1; k < 4; k++)
    if (A[5 * k] > best) {
      best = A[5 * k];
      q[0] = V[k]; q[1] = V[4 + k]; q[2] = V[8 + k]; q[3] = V[12 + k];
    }
  const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
  for (int k = 0; k < 4; k++) q[k] /= nq;
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                       2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 3; c++) T16[4 * r + c] = R[3 * r + c];
    T16[4 * r + 3] = mq[r] - (R[3 * r] * mp[0] + R[3 * r + 1] * mp[1] + R[3 * r + 2] * mp[2]);
  }
  T16[12] = T16[13] = T16[14] = 0;
  T16[15] = 1;
  return true;
}

MI_HD void icp_begin(IcpPairState& st, const float* guess16, int n_source, int n_target) {
  for (int k = 0; k < 16; k++) { st.Tinc[k] = guess16[k]; st.final_T[k] = guess16[k]; }
  st.iterations = 0;
  st.converged = 0;
  st.active = 1;
  st.moved = 0;
  st.n = n_source;
  st.n_target = n_target;
  st.pad0 = st.pad1 = 0;
  st.prev_mse = 1.7976931348623157e308;
  st.mse = 0;
}

// `sums`: the 17 sums of the step just made (the cloud moved by Tinc, correspondences at the moved positions).
MI_HD void icp_advance(IcpPairState& st, const double* sums, const IcpConsts& c) {
  st.moved = 1;
  double Td[16];
  if (!icp_rigid_from_sums(sums, Td)) {  // "Not enough correspondences found. Relax your threshold parameters."
    st.converged = 0;
    st.active = 0;
    return;
  }
  for (int k = 0; k < 16; k++) st.Tinc[k] = (float)Td[k];
  float F[16];  // final_transformation_ = transformation_ * final_transformation_ (Matrix4f)
  for (int r = 0; r < 4; r++)
    for (int cc = 0; cc < 4; cc++) {
      float a = 0.f;
      for (int k = 0; k < 4; k++) a += st.Tinc[4 * r + k] * st.final_T[4 * k + cc];
      F[4 * r + cc] = a;
    }
  for (int k = 0; k < 16; k++) st.final_T[k] = F[k];
  st.iterations++;
  st.mse = sums[16] / sums[0];  // (always: `error` of the result describes the correspondences of the LAST iteration, however it ended)
  // DefaultConvergenceCriteria::hasConverged (max_iterations_similar_transforms_ = 0, absolute MSE threshold 1e-12)
  bool done = false;
  if (st.iterations >= c.max_iterations) done = true;
  if (!done) {
    const double cos_angle = 0.5 * ((double)st.Tinc[0] + (double)st.Tinc[5] + (double)st.Tinc[10] - 1.0);
    const double tr2 = (double)st.Tinc[3] * st.Tinc[3] + (double)st.Tinc[7] * st.Tinc[7] + (double)st.Tinc[11] * st.Tinc[11];
    if (cos_angle >= c.rot_threshold && tr2 <= c.transformation_epsilon) done = true;
  }
  if (!done) {
    if (fabs(st.mse - st.prev_mse) < 1e-12) done = true;
    else if (fabs(st.mse - st.prev_mse) / st.prev_mse < c.euclidean_fitness_epsilon) done = true;
    else st.prev_mse = st.mse;
  }
  if (done) {
    st.converged = 1;
    st.active = 0;
  }
}

}  // namespace mi355
