// pcl/registration/bfgs.h as a RESUMABLE state machine - the six-parameter quasi-Newton minimiser
// pclomp::GeneralizedIterativeClosestPoint runs inside every outer iteration ("GICP": lidar_scan_matcher.cpp:73-96,
// graph_based_slam.cpp:120-141).  PCL's class is GSL's vector_bfgs2: a BFGS direction update around Fletcher's line search
// (bracketing by the rho test, sectioning with cubic / quadratic interpolation, the sigma test on the directional derivative).
// Upstream calls the cost functor from inside the search; here the search RETURNS whenever it needs the cost at a new point
// (bfgs_begin / bfgs_feed -> "evaluate at req_x"), and is fed the value and gradient there - so that one kernel pass evaluates
// the pending points of ALL pairs of a batch and one thread per pair advances its search (gicp_bfgs.hip: k_bfgs_advance).
// Rounds 1-4 ran the class-shaped original on the host, a launch and a read-back per evaluation (tests/bfgs_host_ref.hpp keeps
// it: tests/test_host_math.py holds this text to it, iterate by iterate, on analytic functions).
// Every evaluation supplies f AND the gradient (the cost pass computes both from the same sums), so upstream's separate caches
// of f, df and the slope collapse into one entry - the last evaluated step; values are pure functions of the step, so only the
// NUMBER of evaluations differs from upstream's f() / df() / fdf() calls, never a value.
// Everything is MI_HD: the device advances the pairs, tests/host_math_harness.cpp compiles the same text with g++.
#pragma once
#include <cmath>

#include "devmath.hpp"

namespace mi355 {

enum BfgsResult { BFGS_RUNNING = -2, BFGS_SUCCESS = 0, BFGS_NOPROGRESS = 3 };
enum BfgsStage { BFGS_AT_INIT = 0, BFGS_AT_BRACKET = 1, BFGS_AT_SECTION = 2, BFGS_AT_STEP = 3, BFGS_IDLE = 4 };

struct BfgsMin {
  double x[6];                       // the accepted point (minimizeOneStep's x)
  double x0[6], g0[6], p[6], grad[6];  // start of the current line, its gradient, the direction, gradient at x
  double f, delta_f, fp0, pnorm, g0norm;
  // the one evaluation that is remembered: step c_at along p from x0
  double c_at, c_f, c_slope, c_x[6], c_g[6];
  // what has been asked for: the cost at req_x = x0 + req_at p
  double req_at, req_x[6];
  // Fletcher's line search, its locals
  double alpha, alpha_prev, falpha_prev, fpalpha_prev, a, b, fa, fb, fpa, fpb, f0, lfp0, f_before, alpha_new;
  int i, inner, stage, result;
};

// the values gicp.hpp assigns to BFGS::Parameters (order 3 = cubic interpolation)
struct BfgsParams {
  int bracket_iters = 100, section_iters = 100, order = 3;
  double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1;
};

namespace bfgs_dev {

MI_HD double dot6(const double* a, const double* b) {
  double s = 0;
  for (int k = 0; k < 6; k++) s += a[k] * b[k];
  return s;
}
MI_HD double norm6(const double* a) { return sqrt(dot6(a, a)); }

// minimiser of the quadratic through (0, f0) with slope fp0 and (1, f1), restricted to [zl, zh]
MI_HD double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
  const double curv = f1 - f0 - fp0;
  const double ql = f0 + zl * (fp0 + zl * curv), qh = f0 + zh * (fp0 + zh * curv);
  double zmin = zl, fmin = ql;
  if (qh < fmin) { zmin = zh; fmin = qh; }
  if (2 * curv > 0) {
    const double z = -fp0 / (2 * curv);
    if (z > zl && z < zh && (f0 + z * (fp0 + z * curv)) < fmin) zmin = z;
  }
  return zmin;
}

// real roots of a x^2 + b x + c, ascending (gsl_poly_solve_quadratic)
MI_HD int quadratic_roots(double a, double b, double c, double& r0, double& r1) {
  if (a == 0) {
    if (b == 0) return 0;
    r0 = -c / b;
    return 1;
  }
  const double disc = b * b - 4 * a * c;
  if (disc < 0) return 0;
  if (disc == 0) {
    r0 = r1 = -0.5 * b / a;
    return 2;
  }
  if (b == 0) {
    const double r = sqrt(-c / a);
    r0 = -r; r1 = r;
    return 2;
  }
  const double t = -0.5 * (b + (b > 0 ? 1.0 : -1.0) * sqrt(disc));
  const double ra = t / a, rb = c / t;
  r0 = ra < rb ? ra : rb; r1 = ra < rb ? rb : ra;
  return 2;
}

// minimiser of the cubic through (0, f0, fp0) and (1, f1, fp1), restricted to [zl, zh]
MI_HD double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
  const double c2 = 3 * (f1 - f0) - 2 * fp0 - fp1, c3 = fp0 + fp1 - 2 * (f1 - f0);
  double zmin = zl, fmin = f0 + zl * (fp0 + zl * (c2 + zl * c3));
  {
    const double y = f0 + zh * (fp0 + zh * (c2 + zh * c3));
    if (y < fmin) { zmin = zh; fmin = y; }
  }
  double r0 = 0, r1 = 0;
  const int n = quadratic_roots(3 * c3, 2 * c2, fp0, r0, r1);
  if (n >= 1 && r0 > zl && r0 < zh) {
    const double y = f0 + r0 * (fp0 + r0 * (c2 + r0 * c3));
    if (y < fmin) { zmin = r0; fmin = y; }
  }
  if (n == 2 && r1 > zl && r1 < zh) {
    const double y = f0 + r1 * (fp0 + r1 * (c2 + r1 * c3));
    if (y < fmin) { zmin = r1; fmin = y; }
  }
  return zmin;
}

// next trial step inside [xmin, xmax] from the two known points a and b (fpb = NaN: slope at b unknown)
MI_HD double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order) {
  double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a);
  if (zmin > zmax) { const double t = zmin; zmin = zmax; zmax = t; }
  const double z = (order > 2 && !(fpb != fpb)) ? interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax)
                                                : interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
  return a + z * (b - a);
}

// the search wants the cost at step alpha of the current line: remembered -> false (go on), else the request is filed -> true
MI_HD bool want(BfgsMin& m, double alpha, int stage) {
  if (alpha == m.c_at) return false;
  m.req_at = alpha;
  for (int k = 0; k < 6; k++) m.req_x[k] = m.x0[k] + alpha * m.p[k];
  m.stage = stage;
  return true;
}

}  // namespace bfgs_dev

// BFGS::minimizeInit: the cost at x is asked for
MI_HD void bfgs_begin(BfgsMin& m, const double* x) {
  for (int k = 0; k < 6; k++) { m.x[k] = x[k]; m.req_x[k] = x[k]; }
  m.req_at = 0;
  m.delta_f = 0;
  m.inner = 0;
  m.result = BFGS_RUNNING;
  m.stage = BFGS_AT_INIT;
}

// The cost f and its gradient g at req_x have arrived.  Runs GeneralizedIterativeClosestPoint::estimateRigidTransformationBFGS's
//   do { inner++; result = minimizeOneStep(x); if (result != Success) break; result = testGradient(1e-2); } while (Running && inner < max)
// on until the next point is needed (true: evaluate at m.req_x) or the loop is over (false: m.result, m.inner, m.x, m.f).
MI_HD bool bfgs_feed(BfgsMin& m, double fe, const double* ge, int max_optimizer_iterations, const BfgsParams& P) {
  using namespace bfgs_dev;
  const double eps = 2.220446049250313e-16, nan = NAN;
  enum { START_STEP, BRACKET_LOOP, BRACKET_GOT, SECTION_LOOP, SECTION_GOT, LINE_DONE, STEP_GOT, INNER_END };
  int next;
  if (m.stage == BFGS_AT_INIT) {
    m.f = fe;
    for (int k = 0; k < 6; k++) { m.grad[k] = ge[k]; m.x0[k] = m.x[k]; m.g0[k] = ge[k]; }
    m.g0norm = norm6(m.g0);
    for (int k = 0; k < 6; k++) m.p[k] = m.grad[k] * -1 / m.g0norm;
    m.pnorm = norm6(m.p);
    m.fp0 = -m.g0norm;
    m.c_at = 0; m.c_f = m.f; m.c_slope = dot6(m.g0, m.p);
    for (int k = 0; k < 6; k++) { m.c_x[k] = m.x0[k]; m.c_g[k] = m.g0[k]; }
    next = START_STEP;
  } else {
    m.c_at = m.req_at; m.c_f = fe; m.c_slope = dot6(ge, m.p);
    for (int k = 0; k < 6; k++) { m.c_x[k] = m.req_x[k]; m.c_g[k] = ge[k]; }
    next = m.stage == BFGS_AT_BRACKET ? BRACKET_GOT : (m.stage == BFGS_AT_SECTION ? SECTION_GOT : STEP_GOT);
  }
  for (;;) {
    switch (next) {
      case START_STEP: {  // minimizeOneStep, up to the line search's first trial
        m.inner++;
        m.f_before = m.f;
        if (m.pnorm == 0.0 || m.g0norm == 0.0 || m.fp0 == 0) { m.result = BFGS_NOPROGRESS; next = INNER_END; break; }
        double alpha1;
        if (m.delta_f < 0) {
          const double del = fmax(-m.delta_f, 10 * eps * fabs(m.f_before));
          alpha1 = fmin(1.0, 2.0 * del / (-m.fp0));
        } else {
          alpha1 = fabs(P.step_size);
        }
        // lineSearch: value and slope at step 0 are the remembered ones (c_at == 0 here: the previous step's end, or minimizeInit's)
        m.f0 = m.c_f; m.lfp0 = m.c_slope;
        m.alpha = alpha1; m.alpha_prev = 0.0; m.falpha_prev = m.f0; m.fpalpha_prev = m.lfp0;
        m.a = 0.0; m.b = m.alpha; m.fa = m.f0; m.fb = 0.0; m.fpa = m.lfp0; m.fpb = 0.0;
        m.i = 0;
        m.alpha_new = 0.0;
        next = BRACKET_LOOP;
        break;
      }
      case BRACKET_LOOP:
        if (!(m.i++ < P.bracket_iters)) { next = SECTION_LOOP; break; }
        if (want(m, m.alpha, BFGS_AT_BRACKET)) return true;
        next = BRACKET_GOT;
        break;
      case BRACKET_GOT: {
        const double falpha = m.c_f;
        if (falpha > m.f0 + m.alpha * P.rho * m.lfp0 || falpha >= m.falpha_prev) {  // rho test failed: the minimum is behind us
          m.a = m.alpha_prev; m.fa = m.falpha_prev; m.fpa = m.fpalpha_prev;
          m.b = m.alpha; m.fb = falpha; m.fpb = nan;
          next = SECTION_LOOP;
          break;
        }
        const double fpalpha = m.c_slope;
        if (fabs(fpalpha) <= -P.sigma * m.lfp0) {  // sigma test: flat enough
          m.alpha_new = m.alpha;
          next = LINE_DONE;
          break;
        }
        if (fpalpha >= 0) {
          m.a = m.alpha; m.fa = falpha; m.fpa = fpalpha;
          m.b = m.alpha_prev; m.fb = m.falpha_prev; m.fpb = m.fpalpha_prev;
          next = SECTION_LOOP;
          break;
        }
        const double delta = m.alpha - m.alpha_prev;
        const double nx = interpolate(m.alpha_prev, m.falpha_prev, m.fpalpha_prev, m.alpha, falpha, fpalpha, m.alpha + delta, m.alpha + P.tau1 * delta, P.order);
        m.alpha_prev = m.alpha; m.falpha_prev = falpha; m.fpalpha_prev = fpalpha;
        m.alpha = nx;
        next = BRACKET_LOOP;
        break;
      }
      case SECTION_LOOP: {  // sectioning of the bracket [a, b]
        if (!(m.i++ < P.section_iters)) { next = LINE_DONE; break; }
        const double delta = m.b - m.a;
        m.alpha = interpolate(m.a, m.fa, m.fpa, m.b, m.fb, m.fpb, m.a + P.tau2 * delta, m.b - P.tau3 * delta, P.order);
        if (want(m, m.alpha, BFGS_AT_SECTION)) return true;
        next = SECTION_GOT;
        break;
      }
      case SECTION_GOT: {
        const double falpha = m.c_f;
        if ((m.a - m.alpha) * m.fpa <= eps) { m.result = BFGS_NOPROGRESS; next = INNER_END; break; }  // round-off prevents progress
        if (falpha > m.f0 + P.rho * m.alpha * m.lfp0 || falpha >= m.fa) {
          m.b = m.alpha; m.fb = falpha; m.fpb = nan;
        } else {
          const double fpalpha = m.c_slope;
          if (fabs(fpalpha) <= -P.sigma * m.lfp0) {
            m.alpha_new = m.alpha;
            next = LINE_DONE;
            break;
          }
          if (((m.b - m.a) >= 0 && fpalpha >= 0) || ((m.b - m.a) <= 0 && fpalpha <= 0)) { m.b = m.a; m.fb = m.fa; m.fpb = m.fpa; }
          m.a = m.alpha; m.fa = falpha; m.fpa = fpalpha;
        }
        next = SECTION_LOOP;
        break;
      }
      case LINE_DONE:  // the step is taken at alpha_new: value and gradient there
        if (want(m, m.alpha_new, BFGS_AT_STEP)) return true;
        next = STEP_GOT;
        break;
      case STEP_GOT: {
        m.f = m.c_f;
        for (int k = 0; k < 6; k++) { m.x[k] = m.c_x[k]; m.grad[k] = m.c_g[k]; }
        m.delta_f = m.f - m.f_before;
        // BFGS direction from the change of position and gradient
        double dx[6], dg[6];
        for (int k = 0; k < 6; k++) { dx[k] = m.x[k] - m.x0[k]; dg[k] = m.grad[k] - m.g0[k]; }
        const double dxg = dot6(dx, m.grad), dgg = dot6(dg, m.grad), dxdg = dot6(dx, dg), dgnorm = norm6(dg);
        double A = 0, B = 0;
        if (dxdg != 0) {
          B = dxg / dxdg;
          A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
        }
        for (int k = 0; k < 6; k++) m.p[k] = -A * dx[k] + m.grad[k] - B * dg[k];
        for (int k = 0; k < 6; k++) { m.g0[k] = m.grad[k]; m.x0[k] = m.x[k]; }
        m.g0norm = norm6(m.g0);
        m.pnorm = norm6(m.p);
        const double dir = (dot6(m.p, m.grad) > 0) ? -1.0 : 1.0;
        for (int k = 0; k < 6; k++) m.p[k] *= dir / m.pnorm;
        m.pnorm = norm6(m.p);
        m.fp0 = dot6(m.p, m.g0);
        // new direction: what is remembered describes step 0 again
        m.c_at = 0; m.c_slope = dot6(m.g0, m.p);
        for (int k = 0; k < 6; k++) { m.c_x[k] = m.x0[k]; m.c_g[k] = m.g0[k]; }
        // testGradient(1e-2)
        m.result = norm6(m.grad) < 1e-2 ? BFGS_SUCCESS : BFGS_RUNNING;
        next = (m.result == BFGS_RUNNING && m.inner < max_optimizer_iterations) ? START_STEP : INNER_END;
        break;
      }
      case INNER_END:
      default:
        m.stage = BFGS_IDLE;
        return false;
    }
  }
}

// ---- the pose side of pclomp's GICP (gicp.hpp) ---------------------------------------------------------------------------------
// GeneralizedIterativeClosestPoint::applyState (PCL 1.8): rotation block <- R(x) * rotation block, translation column += x0..2, fp32;
// R(x) = AngleAxisf(x5, Z) * AngleAxisf(x4, Y) * AngleAxisf(x3, X), which Eigen evaluates as a quaternion product.  The half-angle
// sines and cosines are taken in fp64 and rounded to fp32: the correctly rounded fp32 values (what the host's cosf / sinf return in all
// but halfway cases) on the host and on the device alike.
MI_HD void bfgs_apply_state(float* t, const double* x) {
  const float hz = 0.5f * (float)x[5], hy = 0.5f * (float)x[4], hx = 0.5f * (float)x[3];
  const float qz[4] = {(float)cos((double)hz), 0.f, 0.f, (float)sin((double)hz)}, qy[4] = {(float)cos((double)hy), 0.f, (float)sin((double)hy), 0.f},
              qx[4] = {(float)cos((double)hx), (float)sin((double)hx), 0.f, 0.f};
  float qzy[4], q[4];
  // (w, x, y, z) products
  qzy[0] = qz[0] * qy[0] - qz[1] * qy[1] - qz[2] * qy[2] - qz[3] * qy[3];
  qzy[1] = qz[0] * qy[1] + qz[1] * qy[0] + qz[2] * qy[3] - qz[3] * qy[2];
  qzy[2] = qz[0] * qy[2] + qz[2] * qy[0] + qz[3] * qy[1] - qz[1] * qy[3];
  qzy[3] = qz[0] * qy[3] + qz[3] * qy[0] + qz[1] * qy[2] - qz[2] * qy[1];
  q[0] = qzy[0] * qx[0] - qzy[1] * qx[1] - qzy[2] * qx[2] - qzy[3] * qx[3];
  q[1] = qzy[0] * qx[1] + qzy[1] * qx[0] + qzy[2] * qx[3] - qzy[3] * qx[2];
  q[2] = qzy[0] * qx[2] + qzy[2] * qx[0] + qzy[3] * qx[1] - qzy[1] * qx[3];
  q[3] = qzy[0] * qx[3] + qzy[3] * qx[0] + qzy[1] * qx[2] - qzy[2] * qx[1];
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
MI_HD void bfgs_rotation_gradient(const double* x, const double* Racc, double* g) {
  const double cr = cos(x[3]), sr = sin(x[3]), cp = cos(x[4]), sp = sin(x[4]), cy = cos(x[5]), sy = sin(x[5]);
  const double d_roll[9] = {0, sr * sy + cr * cy * sp, cr * sy - cy * sr * sp, 0, -cy * sr + cr * sy * sp, -cr * cy - sr * sy * sp, 0, cr * cp, -cp * sr};
  const double d_pitch[9] = {-cy * sp, cy * cp * sr, cr * cy * cp, -sy * sp, cp * sr * sy, cr * cp * sy, -cp, -sr * sp, -cr * sp};
  const double d_yaw[9] = {-cp * sy, -cr * cy - sr * sy * sp, cy * sr - cr * sy * sp, cy * cp, -cr * sy + cy * sr * sp, sr * sy + cr * cy * sp, 0, 0, 0};
  double s3 = 0, s4 = 0, s5 = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      s3 += d_roll[3 * i + j] * Racc[3 * j + i];
      s4 += d_pitch[3 * i + j] * Racc[3 * j + i];
      s5 += d_yaw[3 * i + j] * Racc[3 * j + i];
    }
  g[3] = s3; g[4] = s4; g[5] = s5;
}

// OptimizationFunctorWithIndices: value and gradient at x from the 14 sums of a cost pass (count, sum r^T M r, sum M r, sum (base p)(M r)^T)
constexpr int kBfgsAcc = 14;
MI_HD void bfgs_cost_from_sums(const double* h, const double* x, double* f, double* g) {
  const double m = h[0];
  *f = h[1] / m;
  for (int k = 0; k < 3; k++) g[k] = h[2 + k] * 2.0 / m;
  double Racc[9];
  for (int k = 0; k < 9; k++) Racc[k] = h[5 + k] * 2.0 / m;
  bfgs_rotation_gradient(x, Racc, g);
}

// estimateRigidTransformationBFGS's start: x = (translation, roll, pitch, yaw) of the fp32 matrix
MI_HD void bfgs_x_from_matrix(const float* t, double* x) {
  x[0] = (double)t[3]; x[1] = (double)t[7]; x[2] = (double)t[11];
  x[3] = atan2((double)t[9], (double)t[10]);
  x[4] = asin(-(double)t[8]);
  x[5] = atan2((double)t[4], (double)t[0]);
}

}  // namespace mi355
