// TEST INFRASTRUCTURE: compiles the PRODUCT's shared host/device maths header (lidar_graph_slam_amd/csrc/devmath.hpp)
// with g++ so that the CPU suite can check it against the oracle / NumPy without a GPU.  Never shipped.
#include "../lidar_graph_slam_amd/csrc/devmath.hpp"
#include "../lidar_graph_slam_amd/csrc/icp_fit.hpp"
#include "../lidar_graph_slam_amd/csrc/ndt_mt.hpp"

using namespace mi355;

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
}
