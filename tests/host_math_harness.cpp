// TEST INFRASTRUCTURE: compiles the PRODUCT's shared host/device maths header (lidar_graph_slam_amd/csrc/devmath.hpp)
// with g++ so that the CPU suite can check it against the oracle / NumPy without a GPU.  Never shipped.
#include "../lidar_graph_slam_amd/csrc/devmath.hpp"
#include "../lidar_graph_slam_amd/csrc/bfgs_dev.hpp"
#include "../lidar_graph_slam_amd/csrc/icp_fit.hpp"
#include "../lidar_graph_slam_amd/csrc/ndt_mt.hpp"

#include "bfgs_host_ref.hpp"  // PCL's BFGS in its upstream, functor-calling shape (the product's until round 4): the reference of bfgs_dev.hpp

using namespace mi355;

namespace {
// analytic test costs over six parameters, value and gradient: 0 an anisotropic quadratic, 1 a Rosenbrock chain, 2 quadratic + cosines
struct TestCost {
  int id;
  long evals = 0;
  void eval(const double* x, double& f, double* g) {
    evals++;
    if (id == 0) {
      f = 0;
      for (int k = 0; k < 6; k++) {
        const double w = 1.0 + 3.0 * k, d = x[k] - 0.3 * (k + 1);
        f += 0.5 * w * d * d;
        g[k] = w * d;
      }
      f += 0.2 * x[0] * x[5];
      g[0] += 0.2 * x[5];
      g[5] += 0.2 * x[0];
    } else if (id == 1) {
      f = 0;
      for (int k = 0; k < 6; k++) g[k] = 0;
      for (int k = 0; k < 5; k++) {
        const double a = x[k + 1] - x[k] * x[k], b = 1 - x[k];
        f += 10 * a * a + b * b;
        g[k] += -40 * a * x[k] - 2 * b;
        g[k + 1] += 20 * a;
      }
    } else {
      f = 0;
      for (int k = 0; k < 6; k++) {
        const double d = x[k] - 0.1 * k;
        f += d * d + 0.3 * (1 - std::cos(3 * x[k] + k));
        g[k] = 2 * d + 0.9 * std::sin(3 * x[k] + k);
      }
    }
  }
  double f(const Vec6& x) { double v, g[6]; eval(x.data(), v, g); return v; }
  void df(const Vec6& x, Vec6& g) { double v; eval(x.data(), v, g.data()); }
  void fdf(const Vec6& x, double& v, Vec6& g) { eval(x.data(), v, g.data()); }
};
}  // namespace

extern "C" {
void hm_se3_exp(const double* d, double* T16) {
  const Pose p = se3_exp(d);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T16[4 * r + c] = p.R[3 * r + c];
    T16[4 * r + 3] = p.t[r];
  }
  T16[12] = T16[13] = T16[14] = 0;
  T16[15] = 1;
}
int hm_ldlt6(const double* A, const double* rhs, double* x) { return ldlt6_solve(A, rhs, x) ? 1 : 0; }
void hm_svd6(const double* H, const double* rhs, double* x) { svd6_solve_sym(H, rhs, x); }
// ICP's rigid fit from the 17 sums of an iteration (Horn's quaternion method); returns 0 when there are fewer than 3 correspondences
int hm_icp_rigid(const double* sums17, double* T16) { return icp_rigid_from_sums(sums17, T16) ? 1 : 0; }
// NDT's Newton solve: pivoted LDL^T for a definite, well-conditioned Hessian, the Jacobi SVD solve for everything else
void hm_ndt_newton(const double* H, const double* rhs, double* x) { ndt_newton_solve(H, rhs, x); }
void hm_regularize(const double* cov6, int method, double* out6) {
  const Sym3<double> c = {cov6[0], cov6[1], cov6[2], cov6[3], cov6[4], cov6[5]};
  const Sym3<double> r = regularize_covariance(c, method);
  out6[0] = r.xx; out6[1] = r.xy; out6[2] = r.xz; out6[3] = r.yy; out6[4] = r.yz; out6[5] = r.zz;
}
// the smallest-eigenvector shortcut of the PLANE regularisation: returns 1 and n[3], or 0 when it defers to the Jacobi path
int hm_smallest_eigvec(const double* cov6, double* n3) {
  const Sym3<double> c = {cov6[0], cov6[1], cov6[2], cov6[3], cov6[4], cov6[5]};
  return smallest_eigenvector_sym3(c, n3) ? 1 : 0;
}
// PLANE regularisation through the Jacobi eigen-decomposition only (what the shortcut must reproduce)
void hm_plane_jacobi(const double* cov6, double* out6) {
  const Sym3<double> c = {cov6[0], cov6[1], cov6[2], cov6[3], cov6[4], cov6[5]};
  double w[3], V[9];
  jacobi_eigen_sym3(c, w, V);
  int o[3];
  sort3_desc(w, o);
  const double values[3] = {1.0, 1.0, 1e-3};
  double out[6] = {0, 0, 0, 0, 0, 0};
  for (int d = 0; d < 3; d++) {
    const double ux = V[0 + o[d]], uy = V[3 + o[d]], uz = V[6 + o[d]], v = values[d];
    out[0] += ux * v * ux; out[1] += ux * v * uy; out[2] += ux * v * uz;
    out[3] += uy * v * uy; out[4] += uy * v * uz; out[5] += uz * v * uz;
  }
  for (int i = 0; i < 6; i++) out6[i] = out[i];
}
void hm_sym3_inverse_f(const float* a6, float* out6) {
  const Sym3<float> a = {a6[0], a6[1], a6[2], a6[3], a6[4], a6[5]};
  const Sym3<float> r = sym3_inverse(a);
  out6[0] = r.xx; out6[1] = r.xy; out6[2] = r.xz; out6[3] = r.yy; out6[4] = r.yz; out6[5] = r.zz;
}
void hm_sym3_rotate_f(const float* R9, const float* c6, float* out6) {
  const Sym3<float> c = {c6[0], c6[1], c6[2], c6[3], c6[4], c6[5]};
  const Sym3<float> r = sym3_rotate(R9, c);
  out6[0] = r.xx; out6[1] = r.xy; out6[2] = r.xz; out6[3] = r.yy; out6[4] = r.yz; out6[5] = r.zz;
}
int hm_is_converged(const double* T16, double rot_eps, double trans_eps) {
  Pose p;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) p.R[3 * r + c] = T16[4 * r + c];
    p.t[r] = T16[4 * r + 3];
  }
  return is_converged(p, rot_eps, trans_eps) ? 1 : 0;
}
float hm_dist2f(const float* q, const float* p) { return dist2f(q[0], q[1], q[2], p[0], p[1], p[2]); }
// estimateRigidTransformationBFGS's inner loop on an analytic cost, through the resumable machine of bfgs_dev.hpp (machine = 1) or through
// the functor-calling class (machine = 0).  out: x[6], f, inner, result, evaluations
void hm_bfgs_minimize(int machine, int cost_id, const double* x_in, int max_inner, double* out) {
  TestCost cost{cost_id};
  double xo[6], fo;
  int inner = 0, result;
  if (machine) {
    BfgsMin m;
    BfgsParams P;
    bfgs_begin(m, x_in);
    for (;;) {
      double f, g[6];
      cost.eval(m.req_x, f, g);
      if (!bfgs_feed(m, f, g, max_inner, P)) break;
    }
    for (int k = 0; k < 6; k++) xo[k] = m.x[k];
    fo = m.f; inner = m.inner; result = m.result;
  } else {
    Vec6 x;
    for (int k = 0; k < 6; k++) x[k] = x_in[k];
    Bfgs<TestCost> bfgs(cost);
    bfgs.minimizeInit(x);
    BfgsStatus r = BfgsStatus::Running;
    do {
      inner++;
      r = bfgs.minimizeOneStep(x);
      if (r != BfgsStatus::Success) break;
      r = bfgs.testGradient(1e-2);
    } while (r == BfgsStatus::Running && inner < max_inner);
    for (int k = 0; k < 6; k++) xo[k] = x[k];
    fo = bfgs.value(); result = (int)r;
  }
  for (int k = 0; k < 6; k++) out[k] = xo[k];
  out[6] = fo; out[7] = inner; out[8] = result; out[9] = (double)cost.evals;
}
// applyState on a 4 x 4 fp32 matrix (in place), the start vector of a matrix, the rotation part of the gradient
void hm_bfgs_apply_state(float* t16, const double* x6) { bfgs_apply_state(t16, x6); }
void hm_bfgs_x_from_matrix(const float* t16, double* x6) { bfgs_x_from_matrix(t16, x6); }
void hm_bfgs_cost_from_sums(const double* h14, const double* x6, double* f, double* g6) { bfgs_cost_from_sums(h14, x6, f, g6); }
}
