// The scalar control of pclomp::NormalDistributionsTransform::computeTransformation: the Newton
// iteration, computeStepLengthMT (More & Thuente 1994 as PCL / ndt_omp restate it) and the convergence test, written ONCE
// as a resumable state machine over ~1.5 KB of per-pair state.  Every step that needs the point cloud — computeDerivatives
// (fp32, with or without the Hessian) or computeHessian (fp64) — is a REQUEST the executor answers with the 28 reduced sums;
// ndt_advance() consumes them and runs on to the next request.
// ndt_solve (ndt.hip) runs it for P pairs at once — a registration object is P = 1 —: k_ndt_advance advances every pair's state on
// the device, nothing is read back between requests (the host only polls the number of unfinished pairs every few rounds).
// Everything is MI_HD: tests/host_math_harness.cpp compiles the same text on the CPU.
#pragma once
#include <cmath>
#include <cstring>

#include "devmath.hpp"
#include "../../include/mi355_lidar.h"

namespace mi355 {

struct NdtAngles {     // eq. 6.19 / 6.21 [Magnusson 2009] at the pose of the requested evaluation
  float jf[8][3];
  float hf[15][3];
  double jd[8][3];
  double hd[15][3];
};

enum NdtRequest { NDT_REQ_NONE = 0, NDT_REQ_DERIV_HESS = 1, NDT_REQ_DERIV = 2, NDT_REQ_HESS64 = 3 };
enum NdtPhase { NDT_PH_FIRST = 0, NDT_PH_MT_FIRST = 1, NDT_PH_MT_LOOP = 2, NDT_PH_HESS64 = 3, NDT_PH_DONE = 4 };

struct NdtConsts {
  double gd1, gd2;                 // gauss_d1_, gauss_d2_
  double step_size, trans_eps;
  int max_iterations, legacy_mt_skip;
};

struct NdtPairState {
  int request, phase;
  int converged, early, nr_iterations, step_iterations, open_interval, interval_converged;
  int n, pad;                      // source points (trans_probability = score / n)
  double pv[6], dir[6], x_t[6];    // pose, search direction (unit), trial pose
  double grad[6], hess[36], score;
  double a_t, a_l, f_l, g_l, a_u, f_u, g_u;
  double phi_0, d_phi_0, phi_t, d_phi_t, psi_t, d_psi_t;
  double trans_probability;
  float T[16];                     // matrix of the requested evaluation == final_transformation_ when done
  NdtAngles ang;
};

MI_HD NdtConsts ndt_consts(const mi355_ndt_params& p) {
  NdtConsts c;
  const double c1 = 10 * (1 - p.outlier_ratio);
  const double c2 = p.outlier_ratio / pow(p.resolution, 3);
  const double d3 = -log(c2);
  c.gd1 = -log(c1 + c2) - d3;
  c.gd2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / c.gd1);
  c.step_size = p.step_size;
  c.trans_eps = p.transformation_epsilon;
  c.max_iterations = p.max_iterations;
  c.legacy_mt_skip = p.legacy_mt_skip;
  return c;
}

MI_HD void ndt_angle_tables(const double p[6], NdtAngles& t) {
  double cx, cy, cz, sx, sy, sz;
  if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
  if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
  if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
  // eq. 6.19 [Magnusson 2009]: rows a..h
  const double J[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy},
                          {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                          {-sy * cz, sy * sz, cy},
                          {sx * cy * cz, -sx * cy * sz, sx * sy},
                          {-cx * cy * cz, cx * cy * sz, -cx * sy},
                          {-cy * sz, -cy * cz, 0},
                          {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0},
                          {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  // eq. 6.21: a2 a3 b2 b3 c2 c3 d1 d2 d3 e1 e2 e3 f1 f2 f3
  const double H[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy},
                           {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},
                           {cx * cy * cz, -cx * cy * sz, cx * sy},
                           {sx * cy * cz, -sx * cy * sz, sx * sy},
                           {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0},
                           {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},
                           {-cy * cz, cy * sz, sy},
                           {-sx * sy * cz, sx * sy * sz, sx * cy},
                           {cx * sy * cz, -cx * sy * sz, -cx * cy},
                           {sy * sz, sy * cz, 0},
                           {-sx * cy * sz, -sx * cy * cz, 0},
                           {cx * cy * sz, cx * cy * cz, 0},
                           {-cy * cz, cy * sz, 0},
                           {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0},
                           {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  for (int a = 0; a < 8; a++)
    for (int d = 0; d < 3; d++) { t.jd[a][d] = J[a][d]; t.jf[a][d] = (float)J[a][d]; }
  for (int a = 0; a < 15; a++)
    for (int d = 0; d < 3; d++) { t.hd[a][d] = H[a][d]; t.hf[a][d] = (float)H[a][d]; }
}

// Translation<float> * AngleAxis<float>(roll,X) * AngleAxis<float>(pitch,Y) * AngleAxis<float>(yaw,Z), fp32 products
MI_HD void ndt_pose_to_matrix_f(const double p[6], float T[16]) {
  const float a = (float)p[3], b = (float)p[4], c = (float)p[5];
  const float ca = cosf(a), sa = sinf(a), cb = cosf(b), sb = sinf(b), cc = cosf(c), sc = sinf(c);
  const float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
  const float Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb};
  const float Rz[9] = {cc, -sc, 0, sc, cc, 0, 0, 0, 1};
  float M[9], R[9];
  for (int r = 0; r < 3; r++)
    for (int s = 0; s < 3; s++) M[3 * r + s] = Rx[3 * r] * Ry[s] + Rx[3 * r + 1] * Ry[3 + s] + Rx[3 * r + 2] * Ry[6 + s];
  for (int r = 0; r < 3; r++)
    for (int s = 0; s < 3; s++) R[3 * r + s] = M[3 * r] * Rz[s] + M[3 * r + 1] * Rz[3 + s] + M[3 * r + 2] * Rz[6 + s];
  for (int r = 0; r < 3; r++) {
    for (int s = 0; s < 3; s++) T[4 * r + s] = R[3 * r + s];
    T[4 * r + 3] = (float)p[r];
  }
  T[12] = T[13] = T[14] = 0.f;
  T[15] = 1.f;
}

// Matrix3f::eulerAngles(0,1,2), fp32
MI_HD void ndt_euler_angles_012(const float T[16], float res[3]) {
  res[0] = atan2f(T[4 * 1 + 2], T[4 * 2 + 2]);
  const float c2 = sqrtf(T[0] * T[0] + T[1] * T[1]);
  if (res[0] > 0.f) {
    res[0] -= 3.14159265358979323846f;
    res[1] = atan2f(-T[2], -c2);
  } else {
    res[1] = atan2f(-T[2], c2);
  }
  const float s1 = sinf(res[0]), c1 = cosf(res[0]);
  res[2] = atan2f(s1 * T[4 * 2 + 0] - c1 * T[4 * 1 + 0], c1 * T[4 * 1 + 1] - s1 * T[4 * 2 + 1]);
  for (int d = 0; d < 3; d++) res[d] = -res[d];
}

// ---- More-Thuente pieces (computeStepLengthMT's auxiliaries) -----------------------------------------------------------
MI_HD double ndt_psi(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
MI_HD double ndt_dpsi(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

MI_HD bool ndt_update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

MI_HD double ndt_cubic_min(double a0, double f0, double g0, double a1, double f1, double g1) {  // minimiser of the cubic through both
  const double z = 3 * (f1 - f0) / (a1 - a0) - g1 - g0;
  const double w = sqrt(z * z - g1 * g0);
  return a0 + (a1 - a0) * (w - g0 - z) / (g1 - g0 + 2 * w);
}

MI_HD double ndt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    const double a_c = ndt_cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return (fabs(a_c - a_l) < fabs(a_q - a_l)) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    const double a_c = ndt_cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return (fabs(a_c - a_t) >= fabs(a_s - a_t)) ? a_c : a_s;
  }
  if (fabs(g_t) <= fabs(g_l)) {
    const double a_c = ndt_cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double nxt = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
    return (a_t > a_l) ? fmin(a_t + 0.66 * (a_u - a_t), nxt) : fmax(a_t + 0.66 * (a_u - a_t), nxt);
  }
  return ndt_cubic_min(a_u, f_u, g_u, a_t, f_t, g_t);
}

// delta_p = JacobiSVD(H).solve(-g) (N2).  For a symmetric H the SVD solve is the eigen-decomposition solve, i.e. H^-1 rhs unless H is
// rank-deficient to working precision (singular values below 6 eps sigma_max are dropped).  NDT Hessians are indefinite away from the
// optimum, so the shortcut is Gaussian elimination with partial pivoting (registers, ~1 k instructions; every index a compile-time
// constant, the run-time pivot row applied as "for every candidate row: if (piv == c) swap"): when its pivots span less than 1e9
// and the residual confirms the solution, no singular value is anywhere near the SVD's truncation threshold and both give H^-1 rhs
// to ~cond * eps.  Everything else — rank-deficient, empty, NaN — takes the cyclic Jacobi, whose ~10^5 dependent fp64 instructions
// on ONE lane set the duration of k_ndt_advance in round 5's first measurement (7.7 of the 42 ms of a 1 024-pair step).
MI_HD bool lu6_solve(const double* A_in, const double* rhs, double* x, double& piv_min, double& piv_max) {
  constexpr int N = 6;
  double A[N * N], b[N];
#pragma unroll
  for (int i = 0; i < N * N; i++) A[i] = A_in[i];
#pragma unroll
  for (int i = 0; i < N; i++) b[i] = rhs[i];
  piv_min = 1.7976931348623157e308;
  piv_max = 0.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; k++) {
    int piv = k;
    double best = fabs(A[k * N + k]);
#pragma unroll
    for (int i = k + 1; i < N; i++)
      if (fabs(A[i * N + k]) > best) { best = fabs(A[i * N + k]); piv = i; }
#pragma unroll
    for (int c = k + 1; c < N; c++) {
      if (piv == c) {
#pragma unroll
        for (int j = 0; j < N; j++) { const double t = A[k * N + j]; A[k * N + j] = A[c * N + j]; A[c * N + j] = t; }
        { const double t = b[k]; b[k] = b[c]; b[c] = t; }
      }
    }
    piv_min = fmin(piv_min, best);
    piv_max = fmax(piv_max, best);
    if (!(best > 0.0)) ok = false;  // (also NaN)
    if (ok) {
      const double inv = 1.0 / A[k * N + k];
#pragma unroll
      for (int i = k + 1; i < N; i++) {
        const double f = A[i * N + k] * inv;
#pragma unroll
        for (int j = k + 1; j < N; j++) A[i * N + j] -= f * A[k * N + j];
        b[i] -= f * b[k];
      }
    }
  }
  if (!ok) return false;
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    double s = b[i];
#pragma unroll
    for (int j = i + 1; j < N; j++) s -= A[i * N + j] * x[j];
    x[i] = s / A[i * N + i];
  }
  return true;
}

MI_HD void ndt_newton_solve(const double* H, const double* rhs, double* x) {
  double lo, hi;
  if (lu6_solve(H, rhs, x, lo, hi) && lo > 1.0e-9 * hi) {
    // residual check: ||H x - rhs||_inf <= 1e-9 (||H||_inf ||x||_inf + ||rhs||_inf)
    double rmax = 0, hmax = 0, xmax = 0, bmax = 0;
    bool finite = true;
    for (int i = 0; i < 6; i++) {
      double r = -rhs[i], hr = 0;
      for (int j = 0; j < 6; j++) { r += H[6 * i + j] * x[j]; hr += fabs(H[6 * i + j]); }
      rmax = fmax(rmax, fabs(r));
      hmax = fmax(hmax, hr);
      xmax = fmax(xmax, fabs(x[i]));
      bmax = fmax(bmax, fabs(rhs[i]));
      finite = finite && (x[i] == x[i]);
    }
    if (finite && xmax < 1.0e300 && rmax <= 1.0e-9 * (hmax * xmax + bmax)) return;
  }
  svd6_solve_sym(H, rhs, x);
}

constexpr double kNdtMu = 1.e-4, kNdtNu = 0.9;
constexpr int kNdtMaxStepIterations = 10;

// ---- the state machine ---------------------------------------------------------------------------------------------------
// computeTransformation's prologue: the first computeDerivatives runs on the cloud moved by the GUESS MATRIX itself, at
// the pose read back from it (translation + eulerAngles(0,1,2)).
MI_HD void ndt_begin(NdtPairState& st, const float guess16[16], int n_source) {
  memset(&st, 0, sizeof(st));
  st.n = n_source;
  for (int k = 0; k < 16; k++) st.T[k] = guess16[k];
  float eul[3];
  ndt_euler_angles_012(guess16, eul);
  st.pv[0] = guess16[3]; st.pv[1] = guess16[7]; st.pv[2] = guess16[11];
  st.pv[3] = eul[0]; st.pv[4] = eul[1]; st.pv[5] = eul[2];
  ndt_angle_tables(st.pv, st.ang);
  st.request = NDT_REQ_DERIV_HESS;
  st.phase = NDT_PH_FIRST;
}

MI_HD void ndt_unpack_hessian(const double* sums, double* H) {
  int k = 7;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) { H[6 * i + j] = sums[k]; H[6 * j + i] = sums[k]; k++; }
}

// ask for computeDerivatives at pv + dir * a_t
MI_HD void ndt_request_trial(NdtPairState& st, int request, int phase) {
  for (int k = 0; k < 6; k++) st.x_t[k] = st.pv[k] + st.dir[k] * st.a_t;
  ndt_pose_to_matrix_f(st.x_t, st.T);
  ndt_angle_tables(st.x_t, st.ang);
  st.request = request;
  st.phase = phase;
}

// `sums`: the 28 reduced values of the request just answered (score, gradient, upper Hessian triangle).
MI_HD void ndt_advance(NdtPairState& st, const double* sums, const NdtConsts& c) {
  enum { NEWTON, MT_CHECK, END_ITER } go;
  const double step_max = c.step_size, step_min = c.trans_eps / 2;
  switch (st.phase) {
    case NDT_PH_FIRST:
      st.score = sums[0];
      for (int k = 0; k < 6; k++) st.grad[k] = sums[1 + k];
      ndt_unpack_hessian(sums, st.hess);
      go = NEWTON;
      break;
    case NDT_PH_MT_FIRST:
    case NDT_PH_MT_LOOP: {
      st.score = sums[0];
      for (int k = 0; k < 6; k++) st.grad[k] = sums[1 + k];
      if (st.phase == NDT_PH_MT_FIRST) ndt_unpack_hessian(sums, st.hess);
      st.phi_t = -st.score;
      st.d_phi_t = 0;
      for (int k = 0; k < 6; k++) st.d_phi_t -= st.grad[k] * st.dir[k];
      st.psi_t = ndt_psi(st.a_t, st.phi_t, st.phi_0, st.d_phi_0, kNdtMu);
      st.d_psi_t = ndt_dpsi(st.d_phi_t, st.d_phi_0, kNdtMu);
      if (st.phase == NDT_PH_MT_LOOP) {
        if (st.open_interval && (st.psi_t <= 0 && st.d_psi_t >= 0)) {
          st.open_interval = 0;
          st.f_l = st.f_l + st.phi_0 - kNdtMu * st.d_phi_0 * st.a_l; st.g_l = st.g_l + kNdtMu * st.d_phi_0;
          st.f_u = st.f_u + st.phi_0 - kNdtMu * st.d_phi_0 * st.a_u; st.g_u = st.g_u + kNdtMu * st.d_phi_0;
        }
        st.interval_converged = st.open_interval ? ndt_update_interval(st.a_l, st.f_l, st.g_l, st.a_u, st.f_u, st.g_u, st.a_t, st.psi_t, st.d_psi_t)
                                                 : ndt_update_interval(st.a_l, st.f_l, st.g_l, st.a_u, st.f_u, st.g_u, st.a_t, st.phi_t, st.d_phi_t);
        st.step_iterations++;
      }
      go = MT_CHECK;
      break;
    }
    case NDT_PH_HESS64:
      ndt_unpack_hessian(sums, st.hess);
      go = END_ITER;
      break;
    default:
      st.request = NDT_REQ_NONE;
      return;
  }
  bool searched = true;  // do_search of the iteration being finished (END_ITER entered from NEWTON with d_phi_0 == 0 clears it)
  for (;;) {
    if (go == NEWTON) {
      double ng[6], delta_p[6];
      for (int k = 0; k < 6; k++) ng[k] = -st.grad[k];
      ndt_newton_solve(st.hess, ng, delta_p);
      double nrm = 0;
      for (int k = 0; k < 6; k++) nrm += delta_p[k] * delta_p[k];
      nrm = sqrt(nrm);
      if (nrm == 0 || nrm != nrm) {
        st.trans_probability = st.score / (double)st.n;
        st.converged = nrm == nrm;
        st.early = 1;
        st.phase = NDT_PH_DONE;
        st.request = NDT_REQ_NONE;
        return;
      }
      for (int k = 0; k < 6; k++) st.dir[k] = delta_p[k] / nrm;
      // ---- computeStepLengthMT ----
      st.phi_0 = -st.score;
      st.d_phi_0 = 0;
      for (int k = 0; k < 6; k++) st.d_phi_0 -= st.grad[k] * st.dir[k];
      st.a_t = 0;
      searched = true;
      if (st.d_phi_0 >= 0) {
        if (st.d_phi_0 == 0) searched = false;
        else {
          st.d_phi_0 *= -1;
          for (int k = 0; k < 6; k++) st.dir[k] *= -1;
        }
      }
      if (!searched) { go = END_ITER; continue; }
      st.step_iterations = 0;
      st.a_l = 0; st.a_u = 0;
      st.f_l = ndt_psi(st.a_l, st.phi_0, st.phi_0, st.d_phi_0, kNdtMu); st.g_l = ndt_dpsi(st.d_phi_0, st.d_phi_0, kNdtMu);
      st.f_u = ndt_psi(st.a_u, st.phi_0, st.phi_0, st.d_phi_0, kNdtMu); st.g_u = ndt_dpsi(st.d_phi_0, st.d_phi_0, kNdtMu);
      // ndt_omp / PCL 1.8 test (step_max - step_min) > 0, which skips the interval refinement (mi355_ndt_params.legacy_mt_skip)
      st.interval_converged = c.legacy_mt_skip ? ((step_max - step_min) > 0) : ((step_max - step_min) < 0);
      st.open_interval = 1;
      st.a_t = fmax(fmin(nrm, step_max), step_min);
      ndt_request_trial(st, NDT_REQ_DERIV_HESS, NDT_PH_MT_FIRST);
      return;
    }
    if (go == MT_CHECK) {
      if (!st.interval_converged && st.step_iterations < kNdtMaxStepIterations && !(st.psi_t <= 0 && st.d_phi_t <= -kNdtNu * st.d_phi_0)) {
        st.a_t = st.open_interval ? ndt_trial_value(st.a_l, st.f_l, st.g_l, st.a_u, st.f_u, st.g_u, st.a_t, st.psi_t, st.d_psi_t)
                                  : ndt_trial_value(st.a_l, st.f_l, st.g_l, st.a_u, st.f_u, st.g_u, st.a_t, st.phi_t, st.d_phi_t);
        st.a_t = fmax(fmin(st.a_t, step_max), step_min);
        ndt_request_trial(st, NDT_REQ_DERIV, NDT_PH_MT_LOOP);
        return;
      }
      if (st.step_iterations) {  // the Hessian of the accepted point, in fp64 (computeHessian); T and the tables are that point's
        st.request = NDT_REQ_HESS64;
        st.phase = NDT_PH_HESS64;
        return;
      }
      go = END_ITER;
      continue;
    }
    // END_ITER: p += delta_p (direction scaled by the step found), convergence test
    const double step = searched ? st.a_t : 0.0;
    for (int k = 0; k < 6; k++) st.pv[k] += st.dir[k] * step;
    if (st.nr_iterations > c.max_iterations || (st.nr_iterations && (fabs(step) < c.trans_eps))) st.converged = 1;
    st.nr_iterations++;
    if (st.converged) {
      st.trans_probability = st.score / (double)st.n;
      st.phase = NDT_PH_DONE;
      st.request = NDT_REQ_NONE;
      return;
    }
    go = NEWTON;
  }
}

}  // namespace mi355
