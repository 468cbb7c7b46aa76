// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED.
//
// Restatement of pclomp::GeneralizedIterativeClosestPoint<PointXYZI,PointXYZI> — registration_method "GICP" of both nodes
// (/root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:73-96, /root/reference/graph_based_slam/src/
// graph_based_slam.cpp:120-141: setMaxCorrespondenceDistance, setMaximumIterations, setMaximumOptimizerIterations,
// setTransformationEpsilon, setEuclideanFitnessEpsilon, setRANSACIterations(0)).  pclomp's class is PCL 1.8's
// pcl::GeneralizedIterativeClosestPoint with OpenMP loops (thirdparty/ndt_omp is an empty submodule here); the algorithm
// below is that published one [UPSTREAM-RECALL]:
//   covariances  k = 20 neighbours (the point itself included), biased covariance, SVD, singular values replaced by
//                (1, 1, gicp_epsilon = 1e-3): the same matrix as fast_gicp's PLANE regularisation -> orc_covariances(PLANE);
//   outer loop   correspondences: q = transformation_ * (guess * p) in fp32, 1-NN in the target, kept iff d2 < max_dist^2;
//                Mahalanobis M_i = (R C_src R^T + C_tgt)^-1 with R the rotation of double(transformation_) * double(guess);
//                transformation_ <- BFGS minimum of f(x) = 1/m sum r^T M r, r = T(x) p - q_tgt, where T(x) = base (= guess)
//                with its rotation pre-multiplied by R(x) = Rz(x5) Ry(x4) Rx(x3) (fp32) and its translation shifted by x0..2;
//                delta = max |previous - transformation_| scaled by 1/rotation_epsilon (3x3 block) or 1/transformation_epsilon
//                (the rest); converged when the iteration count reaches max_iterations or delta < 1;
//   result       final = [R_t R_g | t_t + t_g]  (PCL 1.8 composes rotation and translation separately, as the cost does;
//                only the correspondence search applies transformation_ to the guess-moved point — upstream's quirk, kept).
//   inner solver pcl/registration/bfgs.h = GSL's vector_bfgs2 with Fletcher's line search (rho 0.01, sigma 0.01, tau1 9,
//                tau2 0.05, tau3 0.5, cubic interpolation), at most max_optimizer_iterations steps, stopped when the gradient
//                norm drops below 1e-2.
#include <omp.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "kdtree.hpp"
#include "linalg.hpp"
#include "oracle.h"

using namespace oracle;

extern "C" void orc_gicp_bfgs_default_params(orc_gicp_bfgs_params* p) {
  p->max_correspondence_distance = 2.0;  // yaml: correspondence_distance
  p->max_iterations = 64;
  p->max_optimizer_iterations = 20;
  p->transformation_epsilon = 0.01;
  p->rotation_epsilon = 2e-3;            // PCL default, not set by the nodes
  p->k_correspondences = 20;             // PCL default
  p->gicp_epsilon = 0.001;               // PCL default
  p->num_threads = 0;
}

namespace {

// ---- pcl/registration/bfgs.h -----------------------------------------------------------------------------------------
enum { BFGS_RUNNING = -2, BFGS_SUCCESS = 0, BFGS_NO_PROGRESS = 3 };

struct Cost {  // f and its gradient at x (6 doubles)
  virtual double f(const double* x) = 0;
  virtual void df(const double* x, double* g) = 0;
  virtual void fdf(const double* x, double& f, double* g) = 0;
  virtual ~Cost() {}
};

double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
  const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
  const double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
  const double c = 2 * (f1 - f0 - fp0);
  double zmin = zl, fmin = fl;
  if (fh < fmin) { zmin = zh; fmin = fh; }
  if (c > 0) {
    const double z = -fp0 / c;
    if (z > zl && z < zh) {
      const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
      if (f < fmin) { zmin = z; fmin = f; }
    }
  }
  return zmin;
}

inline double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }

// real roots of a x^2 + b x + c (gsl_poly_solve_quadratic): count, ascending
int solve_quadratic(double a, double b, double c, double& x0, double& x1) {
  if (a == 0) {
    if (b == 0) return 0;
    x0 = -c / b;
    return 1;
  }
  const double disc = b * b - 4 * a * c;
  if (disc > 0) {
    if (b == 0) {
      const double r = std::sqrt(-c / a);
      x0 = -r; x1 = r;
    } else {
      const double sgnb = (b > 0 ? 1 : -1);
      const double temp = -0.5 * (b + sgnb * std::sqrt(disc));
      const double r1 = temp / a, r2 = c / temp;
      if (r1 < r2) { x0 = r1; x1 = r2; } else { x0 = r2; x1 = r1; }
    }
    return 2;
  }
  if (disc == 0) {
    x0 = x1 = -0.5 * b / a;
    return 2;
  }
  return 0;
}

double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
  const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1;
  const double xi = fp0 + fp1 - 2 * (f1 - f0);
  const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
  double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl);
  auto check = [&](double z) {
    const double y = cubic(c0, c1, c2, c3, z);
    if (y < fmin) { zmin = z; fmin = y; }
  };
  check(zh);
  double z0 = 0, z1 = 0;
  const int n = solve_quadratic(3 * c3, 2 * c2, c1, z0, z1);
  if (n == 2) {
    if (z0 > zl && z0 < zh) check(z0);
    if (z1 > zl && z1 < zh) check(z1);
  } else if (n == 1) {
    if (z0 > zl && z0 < zh) check(z0);
  }
  return zmin;
}

double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order) {
  double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a);
  if (zmin > zmax) std::swap(zmin, zmax);
  double z;
  if (order > 2 && !std::isnan(fpb)) z = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax);
  else z = interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
  return a + z * (b - a);
}

struct Bfgs {
  Cost& cost;
  // parameters (the values gicp.hpp sets)
  int bracket_iters = 100, section_iters = 100, order = 3;
  double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1;
  // state
  double f = 0, delta_f = 0, fp0 = 0, pnorm = 0, g0norm = 0;
  double x0[6], g0[6], p[6], gradient[6], dx[6];
  // the wrapper's caches
  double x_alpha[6], g_alpha[6], f_alpha = 0, df_alpha = 0;
  double f_key = 0, df_key = 0, x_key = 0, g_key = 0;

  explicit Bfgs(Cost& c) : cost(c) {}

  static double dot(const double* a, const double* b) { double s = 0; for (int k = 0; k < 6; k++) s += a[k] * b[k]; return s; }
  static double norm(const double* a) { return std::sqrt(dot(a, a)); }
  double slope() const { return dot(g_alpha, p); }
  void move_to(double alpha) {
    if (alpha == x_key) return;
    for (int k = 0; k < 6; k++) x_alpha[k] = x0[k] + alpha * p[k];
    x_key = alpha;
  }
  double apply_f(double alpha) {
    if (alpha == f_key) return f_alpha;
    move_to(alpha);
    f_alpha = cost.f(x_alpha);
    f_key = alpha;
    return f_alpha;
  }
  double apply_df(double alpha) {
    if (alpha == df_key) return df_alpha;
    move_to(alpha);
    if (alpha != g_key) { cost.df(x_alpha, g_alpha); g_key = alpha; }
    df_alpha = slope();
    df_key = alpha;
    return df_alpha;
  }
  void apply_fdf(double alpha, double& fo, double& dfo) {
    if (alpha == f_key && alpha == df_key) { fo = f_alpha; dfo = df_alpha; return; }
    if (alpha == f_key || alpha == df_key) { fo = apply_f(alpha); dfo = apply_df(alpha); return; }
    move_to(alpha);
    cost.fdf(x_alpha, f_alpha, g_alpha);
    f_key = alpha; g_key = alpha;
    df_alpha = slope();
    df_key = alpha;
    fo = f_alpha; dfo = df_alpha;
  }
  void change_direction() {
    std::memcpy(x_alpha, x0, sizeof(x0)); x_key = 0;
    f_key = 0;
    std::memcpy(g_alpha, g0, sizeof(g0)); g_key = 0;
    df_alpha = slope(); df_key = 0;
  }

  void init(double* x) {
    delta_f = 0;
    std::memset(dx, 0, sizeof(dx));
    cost.fdf(x, f, gradient);
    std::memcpy(x0, x, sizeof(x0));
    std::memcpy(g0, gradient, sizeof(g0));
    g0norm = norm(g0);
    for (int k = 0; k < 6; k++) p[k] = gradient[k] * -1 / g0norm;
    pnorm = norm(p);
    fp0 = -g0norm;
    std::memcpy(x_alpha, x0, sizeof(x0)); x_key = 0;
    f_alpha = f; f_key = 0;
    std::memcpy(g_alpha, g0, sizeof(g0)); g_key = 0;
    df_alpha = slope(); df_key = 0;
  }

  int line_search(double alpha1, double& alpha_new) {
    double f0l, fp0l, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
    double alpha = alpha1, alpha_prev = 0.0;
    double a, b, fa, fb, fpa, fpb;
    int i = 0;
    apply_fdf(0.0, f0l, fp0l);
    falpha_prev = f0l; fpalpha_prev = fp0l;
    a = 0.0; b = alpha; fa = f0l; fb = 0.0; fpa = fp0l; fpb = 0.0;
    while (i++ < bracket_iters) {  // bracketing
      falpha = apply_f(alpha);
      if (falpha > f0l + alpha * rho * fp0l || falpha >= falpha_prev) {  // Fletcher's rho test
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
        b = alpha; fb = falpha; fpb = std::numeric_limits<double>::quiet_NaN();
        break;
      }
      fpalpha = apply_df(alpha);
      if (std::fabs(fpalpha) <= -sigma * fp0l) { alpha_new = alpha; return BFGS_SUCCESS; }  // Fletcher's sigma test
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha;
        b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
        break;
      }
      delta = alpha - alpha_prev;
      alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta, order);
      alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha; alpha = alpha_next;
    }
    while (i++ < section_iters) {  // sectioning of [a, b]
      delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
      falpha = apply_f(alpha);
      if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return BFGS_NO_PROGRESS;  // roundoff prevents progress
      if (falpha > f0l + rho * alpha * fp0l || falpha >= fa) {
        b = alpha; fb = falpha; fpb = std::numeric_limits<double>::quiet_NaN();
      } else {
        fpalpha = apply_df(alpha);
        if (std::fabs(fpalpha) <= -sigma * fp0l) { alpha_new = alpha; return BFGS_SUCCESS; }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
          b = a; fb = fa; fpb = fpa;
          a = alpha; fa = falpha; fpa = fpalpha;
        } else {
          a = alpha; fa = falpha; fpa = fpalpha;
        }
      }
    }
    return BFGS_SUCCESS;
  }

  int one_step(double* x) {
    double alpha = 0.0, alpha1;
    const double f0l = f;
    if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) {
      std::memset(dx, 0, sizeof(dx));
      return BFGS_NO_PROGRESS;
    }
    if (delta_f < 0) {
      const double del = std::max(-delta_f, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f0l));
      alpha1 = std::min(1.0, 2.0 * del / (-fp0));
    } else {
      alpha1 = std::fabs(step_size);
    }
    const int status = line_search(alpha1, alpha);
    if (status != BFGS_SUCCESS) return status;
    {  // updatePosition
      double fa, dfa;
      apply_fdf(alpha, fa, dfa);
      f = fa;
      std::memcpy(x, x_alpha, sizeof(x_alpha));
      std::memcpy(gradient, g_alpha, sizeof(g_alpha));
    }
    delta_f = f - f0l;
    double dx0[6], dg0[6];
    for (int k = 0; k < 6; k++) { dx0[k] = x[k] - x0[k]; dx[k] = dx0[k]; dg0[k] = gradient[k] - g0[k]; }
    const double dxg = dot(dx0, gradient), dgg = dot(dg0, gradient), dxdg = dot(dx0, dg0), dgnorm = norm(dg0);
    double A, B;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    } else {
      B = 0; A = 0;
    }
    for (int k = 0; k < 6; k++) p[k] = -A * dx0[k] + gradient[k] - B * dg0[k];
    std::memcpy(g0, gradient, sizeof(g0));
    std::memcpy(x0, x, sizeof(x0));
    g0norm = norm(g0);
    pnorm = norm(p);
    const double dir = (dot(p, gradient) > 0) ? -1.0 : 1.0;
    for (int k = 0; k < 6; k++) p[k] *= dir / pnorm;
    pnorm = norm(p);
    fp0 = dot(p, g0);
    change_direction();
    return BFGS_SUCCESS;
  }
  int test_gradient(double eps) const { return norm(gradient) < eps ? BFGS_SUCCESS : BFGS_RUNNING; }
};

// ---- GeneralizedIterativeClosestPoint::applyState (fp32: AngleAxisf products are quaternion products) ------------------
void apply_state(float t[16], const double* x) {
  const float hz = 0.5f * (float)x[5], hy = 0.5f * (float)x[4], hx = 0.5f * (float)x[3];
  // q = qz * qy * qx, (w, x, y, z)
  const float qz[4] = {std::cos(hz), 0.f, 0.f, std::sin(hz)}, qy[4] = {std::cos(hy), 0.f, std::sin(hy), 0.f}, qx[4] = {std::cos(hx), std::sin(hx), 0.f, 0.f};
  auto qmul = [](const float* a, const float* b, float* o) {
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

// R = Rz(psi) Ry(theta) Rx(phi): the three partial derivatives contracted with Racc (computeRDerivative)
void r_derivative(const double* x, const double Racc[9], double* g) {
  const double phi = x[3], theta = x[4], psi = x[5];
  const double cphi = std::cos(phi), sphi = std::sin(phi), ctheta = std::cos(theta), stheta = std::sin(theta), cpsi = std::cos(psi), spsi = std::sin(psi);
  const double dphi[9] = {0, sphi * spsi + cphi * cpsi * stheta, cphi * spsi - cpsi * sphi * stheta,
                          0, -cpsi * sphi + cphi * spsi * stheta, -cphi * cpsi - sphi * spsi * stheta,
                          0, cphi * ctheta, -ctheta * sphi};
  const double dtheta[9] = {-cpsi * stheta, cpsi * ctheta * sphi, cphi * cpsi * ctheta,
                            -spsi * stheta, ctheta * sphi * spsi, cphi * ctheta * spsi,
                            -ctheta, -sphi * stheta, -cphi * stheta};
  const double dpsi[9] = {-ctheta * spsi, -cphi * cpsi - sphi * spsi * stheta, cpsi * sphi - cphi * spsi * stheta,
                          cpsi * ctheta, -cphi * spsi + cpsi * sphi * stheta, sphi * spsi + cphi * cpsi * stheta,
                          0, 0, 0};
  auto inner = [&](const double* A) {  // matricesInnerProd: sum_ij A(j,i) Racc(i,j) = tr(A Racc)
    double r = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r += A[3 * j + i] * Racc[3 * i + j];
    return r;
  };
  g[3] = inner(dphi); g[4] = inner(dtheta); g[5] = inner(dpsi);
}

struct GicpCost : Cost {  // OptimizationFunctorWithIndices
  const float* src; const float* tgt;
  const std::vector<int>* isrc; const std::vector<int>* itgt;
  const std::vector<double>* mahal;  // n_src x 9
  float base[16];
  long evals = 0;

  void sums(const double* x, double* f, double* g) {
    evals++;
    float T[16];
    std::memcpy(T, base, sizeof(T));
    apply_state(T, x);
    const int m = (int)isrc->size();
    double fs = 0, g3[3] = {0, 0, 0}, Racc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < m; i++) {
      const float* p = src + 4 * (size_t)(*isrc)[(size_t)i];
      const float* q = tgt + 4 * (size_t)(*itgt)[(size_t)i];
      float pp[3];
      for (int r = 0; r < 3; r++) pp[r] = T[4 * r] * p[0] + T[4 * r + 1] * p[1] + T[4 * r + 2] * p[2] + T[4 * r + 3];
      const double res[3] = {(double)(pp[0] - q[0]), (double)(pp[1] - q[1]), (double)(pp[2] - q[2])};
      const double* M = mahal->data() + 9 * (size_t)(*isrc)[(size_t)i];
      double temp[3];
      for (int r = 0; r < 3; r++) temp[r] = M[3 * r] * res[0] + M[3 * r + 1] * res[1] + M[3 * r + 2] * res[2];
      fs += res[0] * temp[0] + res[1] * temp[1] + res[2] * temp[2];
      if (g) {
        for (int r = 0; r < 3; r++) g3[r] += temp[r];
        float pb[3];  // base_transformation_ * p_src (translation included: upstream's)
        for (int r = 0; r < 3; r++) pb[r] = base[4 * r] * p[0] + base[4 * r + 1] * p[1] + base[4 * r + 2] * p[2] + base[4 * r + 3];
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) Racc[3 * r + c] += (double)pb[r] * temp[c];
      }
    }
    if (f) *f = fs / m;
    if (g) {
      for (int r = 0; r < 3; r++) g[r] = g3[r] * 2.0 / m;
      for (double& v : Racc) v *= 2.0 / m;
      r_derivative(x, Racc, g);
    }
  }
  double f(const double* x) override { double v; sums(x, &v, nullptr); return v; }
  void df(const double* x, double* g) override { sums(x, nullptr, g); }
  void fdf(const double* x, double& fo, double* g) override { sums(x, &fo, g); }
};

}  // namespace

// f and gradient of the GICP cost for given correspondences (parity tests: numeric gradient, HIP reduction)
extern "C" int orc_gicp_bfgs_cost(const float* src, const float* tgt, const int* isrc, const int* itgt, int m, const double* mahal9_by_src,
                                  size_t n_src, const float* base16, const double* x6, double* f, double* g6) {
  GicpCost c;
  std::vector<int> a(isrc, isrc + m), b(itgt, itgt + m);
  std::vector<double> M(mahal9_by_src, mahal9_by_src + 9 * n_src);
  c.src = src; c.tgt = tgt; c.isrc = &a; c.itgt = &b; c.mahal = &M;
  std::memcpy(c.base, base16, sizeof(c.base));
  c.fdf(x6, *f, g6);
  return 0;
}

extern "C" int orc_gicp_bfgs_align(const float* src, size_t n, const float* tgt, size_t m, const orc_gicp_bfgs_params* p, const float* guess16,
                                   orc_result* res) {
  const int saved = omp_get_max_threads();
  if (p->num_threads > 0) omp_set_num_threads(p->num_threads);
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  std::memset(res, 0, sizeof(*res));
  float guess[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (guess16) std::memcpy(guess, guess16, sizeof(guess));
  std::memcpy(res->T, guess, sizeof(guess));
  if ((size_t)p->k_correspondences > n || (size_t)p->k_correspondences > m || p->k_correspondences < 1) {
    omp_set_num_threads(saved);
    return -1;
  }
  std::vector<double> csrc(9 * n), ctgt(9 * m);
  orc_covariances(src, n, p->k_correspondences, ORC_REG_PLANE, csrc.data(), p->num_threads);
  orc_covariances(tgt, m, p->k_correspondences, ORC_REG_PLANE, ctgt.data(), p->num_threads);
  KdTree tree;
  tree.build(tgt, m, 4);
  const auto t1 = clk::now();

  float trans[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, prev[16];
  std::memcpy(prev, trans, sizeof(prev));
  const float thr = (float)(p->max_correspondence_distance * p->max_correspondence_distance);
  std::vector<double> mahal(9 * n);
  for (size_t i = 0; i < n; i++) { double* M = &mahal[9 * i]; std::memset(M, 0, 72); M[0] = M[4] = M[8] = 1; }
  std::vector<int> nn(n);
  std::vector<float> nd(n);
  int nr = 0;
  bool converged = false;
  double last_f = 0;
  while (!converged) {
    // transform_R = double(transformation_) * double(guess)
    double TR[16];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += (double)trans[4 * i + k] * (double)guess[4 * k + j];
        TR[4 * i + j] = s;
      }
    Mat3 R;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) R(r, c) = TR[4 * r + c];
#pragma omp parallel for schedule(guided, 8)
    for (long i = 0; i < (long)n; i++) {
      const float* q0 = src + 4 * i;
      float q1[3], q2[3];
      for (int r = 0; r < 3; r++) q1[r] = guess[4 * r] * q0[0] + guess[4 * r + 1] * q0[1] + guess[4 * r + 2] * q0[2] + guess[4 * r + 3];
      for (int r = 0; r < 3; r++) q2[r] = trans[4 * r] * q1[0] + trans[4 * r + 1] * q1[1] + trans[4 * r + 2] * q1[2] + trans[4 * r + 3];
      int idx = -1;
      float d2 = 0;
      tree.knn(q2, 1, &idx, &d2);
      nn[(size_t)i] = idx;
      nd[(size_t)i] = d2;
      if (idx >= 0 && d2 < thr) {
        Mat3 C1, C2;
        std::memcpy(C1.m, &csrc[9 * (size_t)i], 72);
        std::memcpy(C2.m, &ctgt[9 * (size_t)idx], 72);
        const Mat3 Mi = inverse(add(mul(mul(R, C1), transpose(R)), C2));
        std::memcpy(&mahal[9 * (size_t)i], Mi.m, 72);
      }
    }
    std::vector<int> isrc, itgt;
    for (size_t i = 0; i < n; i++)
      if (nn[i] >= 0 && nd[i] < thr) { isrc.push_back((int)i); itgt.push_back(nn[i]); }
    std::memcpy(prev, trans, sizeof(prev));
    // ---- estimateRigidTransformationBFGS
    if (isrc.size() < 4) break;  // "Need at least 4 points to estimate a transform": caught upstream, loop left, not converged
    double x[6] = {(double)trans[3], (double)trans[7], (double)trans[11], std::atan2((double)trans[9], (double)trans[10]),
                   std::asin(-(double)trans[8]), std::atan2((double)trans[4], (double)trans[0])};
    GicpCost cost;
    cost.src = src; cost.tgt = tgt; cost.isrc = &isrc; cost.itgt = &itgt; cost.mahal = &mahal;
    std::memcpy(cost.base, guess, sizeof(guess));
    Bfgs bfgs(cost);
    bfgs.init(x);
    int inner = 0, result = BFGS_RUNNING;
    do {
      inner++;
      result = bfgs.one_step(x);
      if (result) break;
      result = bfgs.test_gradient(1e-2);
    } while (result == BFGS_RUNNING && inner < p->max_optimizer_iterations);
    if (!(result == BFGS_NO_PROGRESS || result == BFGS_SUCCESS || inner == p->max_optimizer_iterations)) break;  // SolverDidntConverge
    last_f = bfgs.f;
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(trans, I, sizeof(I));
    apply_state(trans, x);
    double delta = 0;
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 4; l++) {
        const double ratio = (k < 3 && l < 3) ? 1.0 / p->rotation_epsilon : 1.0 / p->transformation_epsilon;
        const double c = ratio * std::fabs((double)prev[4 * k + l] - (double)trans[4 * k + l]);
        if (c > delta) delta = c;
      }
    nr++;
    if (nr >= p->max_iterations || delta < 1) {
      converged = true;
      std::memcpy(prev, trans, sizeof(prev));
    }
  }
  // final = [R_prev R_guess | t_prev + t_guess]
  float F[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) F[4 * r + c] = prev[4 * r] * guess[c] + prev[4 * r + 1] * guess[4 + c] + prev[4 * r + 2] * guess[8 + c];
    F[4 * r + 3] = prev[4 * r + 3] + guess[4 * r + 3];
  }
  std::memcpy(res->T, F, sizeof(F));
  res->converged = converged ? 1 : 0;
  res->iterations = nr;
  res->error = last_f;
  const auto t2 = clk::now();
  res->ms_setup = std::chrono::duration<double, std::milli>(t1 - t0).count();
  res->ms_align = std::chrono::duration<double, std::milli>(t2 - t1).count();
  omp_set_num_threads(saved);
  return 0;
}
