// pcl/registration/bfgs.h — the six-parameter quasi-Newton minimiser pclomp::GeneralizedIterativeClosestPoint runs inside
// every outer iteration (lidar_scan_matcher.cpp:73-96 selects it as "GICP").  PCL's class is GSL's vector_bfgs2: a BFGS
// direction update around Fletcher's line search (bracketing by the rho test, sectioning with cubic / quadratic
// interpolation, the sigma test on the directional derivative), with a one-entry cache per quantity so that a trial step is
// never evaluated twice.  Host code: scalar logic on a handful of doubles; every evaluation of the cost is a kernel launch
// (gicp_bfgs.hip).  The functor supplies  double f(x),  void df(x, g),  void fdf(x, f, g)  over std::array<double, 6>.
#pragma once
#include <array>
#include <cmath>
#include <limits>
#include <utility>

namespace mi355 {

using Vec6 = std::array<double, 6>;

enum class BfgsStatus { Running = -2, Success = 0, NoProgress = 3 };

namespace bfgs_detail {

inline double dot6(const Vec6& a, const Vec6& b) {
  double s = 0;
  for (int k = 0; k < 6; k++) s += a[k] * b[k];
  return s;
}
inline double norm6(const Vec6& a) { return std::sqrt(dot6(a, a)); }

// minimiser of the quadratic through (0, f0) with slope fp0 and (1, f1), restricted to [zl, zh]
inline double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
  const double curv = f1 - f0 - fp0;
  auto q = [&](double z) { return f0 + z * (fp0 + z * curv); };
  double zmin = zl, fmin = q(zl);
  if (q(zh) < fmin) { zmin = zh; fmin = q(zh); }
  if (2 * curv > 0) {
    const double z = -fp0 / (2 * curv);
    if (z > zl && z < zh && q(z) < fmin) zmin = z;
  }
  return zmin;
}

// real roots of a x^2 + b x + c, ascending (gsl_poly_solve_quadratic)
inline int quadratic_roots(double a, double b, double c, double& r0, double& r1) {
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
    const double r = std::sqrt(-c / a);
    r0 = -r; r1 = r;
    return 2;
  }
  const double t = -0.5 * (b + (b > 0 ? 1.0 : -1.0) * std::sqrt(disc));
  const double ra = t / a, rb = c / t;
  r0 = std::min(ra, rb); r1 = std::max(ra, rb);
  return 2;
}

// minimiser of the cubic through (0, f0, fp0) and (1, f1, fp1), restricted to [zl, zh]
inline double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
  const double c2 = 3 * (f1 - f0) - 2 * fp0 - fp1, c3 = fp0 + fp1 - 2 * (f1 - f0);
  auto cub = [&](double z) { return f0 + z * (fp0 + z * (c2 + z * c3)); };
  double zmin = zl, fmin = cub(zl);
  auto consider = [&](double z) {
    const double y = cub(z);
    if (y < fmin) { zmin = z; fmin = y; }
  };
  consider(zh);
  double r0 = 0, r1 = 0;
  const int n = quadratic_roots(3 * c3, 2 * c2, fp0, r0, r1);
  if (n >= 1 && r0 > zl && r0 < zh) consider(r0);
  if (n == 2 && r1 > zl && r1 < zh) consider(r1);
  return zmin;
}

// next trial step inside [xmin, xmax] from the two known points a and b (fpb = NaN: slope at b unknown)
inline double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order) {
  double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a);
  if (zmin > zmax) std::swap(zmin, zmax);
  const double z = (order > 2 && !std::isnan(fpb)) ? interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax)
                                                   : interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
  return a + z * (b - a);
}

}  // namespace bfgs_detail

template <typename Functor>
class Bfgs {
 public:
  struct Parameters {  // the values gicp.hpp assigns (order 3 = cubic interpolation)
    int bracket_iters = 100, section_iters = 100, order = 3;
    double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1;
  } parameters;

  explicit Bfgs(Functor& functor) : fn_(functor) {}

  void minimizeInit(Vec6& x) {
    using namespace bfgs_detail;
    delta_f_ = 0;
    fn_.fdf(x, f_, gradient_);
    x0_ = x;
    g0_ = gradient_;
    g0norm_ = norm6(g0_);
    for (int k = 0; k < 6; k++) p_[k] = gradient_[k] * -1 / g0norm_;
    pnorm_ = norm6(p_);
    fp0_ = -g0norm_;
    trial_ = Trial{x0_, g0_, f_, dot6(g0_, p_), 0, 0, 0, 0};
  }

  BfgsStatus minimizeOneStep(Vec6& x) {
    using namespace bfgs_detail;
    const double f_before = f_;
    if (pnorm_ == 0.0 || g0norm_ == 0.0 || fp0_ == 0) return BfgsStatus::NoProgress;
    double alpha1;
    if (delta_f_ < 0) {
      const double del = std::max(-delta_f_, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f_before));
      alpha1 = std::min(1.0, 2.0 * del / (-fp0_));
    } else {
      alpha1 = std::fabs(parameters.step_size);
    }
    double alpha = 0;
    const BfgsStatus st = lineSearch(alpha1, alpha);
    if (st != BfgsStatus::Success) return st;
    {  // take the step
      double fa, dfa;
      valueAndSlope(alpha, fa, dfa);
      f_ = fa;
      x = trial_.x;
      gradient_ = trial_.g;
    }
    delta_f_ = f_ - f_before;
    // BFGS direction from the change of position and gradient
    Vec6 dx, dg;
    for (int k = 0; k < 6; k++) { dx[k] = x[k] - x0_[k]; dg[k] = gradient_[k] - g0_[k]; }
    const double dxg = dot6(dx, gradient_), dgg = dot6(dg, gradient_), dxdg = dot6(dx, dg), dgnorm = norm6(dg);
    double A = 0, B = 0;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    }
    for (int k = 0; k < 6; k++) p_[k] = -A * dx[k] + gradient_[k] - B * dg[k];
    g0_ = gradient_;
    x0_ = x;
    g0norm_ = norm6(g0_);
    pnorm_ = norm6(p_);
    const double dir = (dot6(p_, gradient_) > 0) ? -1.0 : 1.0;
    for (int k = 0; k < 6; k++) p_[k] *= dir / pnorm_;
    pnorm_ = norm6(p_);
    fp0_ = dot6(p_, g0_);
    trial_ = Trial{x0_, g0_, trial_.f, dot6(g0_, p_), 0, 0, 0, 0};  // new direction: the caches describe step 0 again
    return BfgsStatus::Success;
  }

  BfgsStatus testGradient(double epsilon) const { return bfgs_detail::norm6(gradient_) < epsilon ? BfgsStatus::Success : BfgsStatus::Running; }
  double value() const { return f_; }

 private:
  // the point x0 + alpha p with what has been evaluated there; *_at = the step each cached quantity belongs to
  struct Trial {
    Vec6 x, g;
    double f, slope;
    double x_at, f_at, g_at, slope_at;
  };

  void moveTo(double alpha) {
    if (alpha == trial_.x_at) return;
    for (int k = 0; k < 6; k++) trial_.x[k] = x0_[k] + alpha * p_[k];
    trial_.x_at = alpha;
  }
  double valueAt(double alpha) {
    if (alpha != trial_.f_at) {
      moveTo(alpha);
      trial_.f = fn_.f(trial_.x);
      trial_.f_at = alpha;
    }
    return trial_.f;
  }
  double slopeAt(double alpha) {
    if (alpha != trial_.slope_at) {
      moveTo(alpha);
      if (alpha != trial_.g_at) {
        fn_.df(trial_.x, trial_.g);
        trial_.g_at = alpha;
      }
      trial_.slope = bfgs_detail::dot6(trial_.g, p_);
      trial_.slope_at = alpha;
    }
    return trial_.slope;
  }
  void valueAndSlope(double alpha, double& f, double& slope) {
    if (alpha == trial_.f_at || alpha == trial_.slope_at) {
      f = valueAt(alpha);
      slope = slopeAt(alpha);
      return;
    }
    moveTo(alpha);
    fn_.fdf(trial_.x, trial_.f, trial_.g);
    trial_.f_at = trial_.g_at = trial_.slope_at = alpha;
    trial_.slope = bfgs_detail::dot6(trial_.g, p_);
    f = trial_.f;
    slope = trial_.slope;
  }

  // Fletcher's line search (GSL linear_minimize.c: minimize)
  BfgsStatus lineSearch(double alpha1, double& alpha_new) {
    using bfgs_detail::interpolate;
    const Parameters& P = parameters;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    double f0, fp0;
    valueAndSlope(0.0, f0, fp0);
    double alpha = alpha1, alpha_prev = 0.0, falpha_prev = f0, fpalpha_prev = fp0;
    double a = 0.0, b = alpha, fa = f0, fb = 0.0, fpa = fp0, fpb = 0.0;
    int i = 0;
    while (i++ < P.bracket_iters) {  // bracketing
      const double falpha = valueAt(alpha);
      if (falpha > f0 + alpha * P.rho * fp0 || falpha >= falpha_prev) {  // rho test failed: the minimum is behind us
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
        b = alpha; fb = falpha; fpb = nan;
        break;
      }
      const double fpalpha = slopeAt(alpha);
      if (std::fabs(fpalpha) <= -P.sigma * fp0) {  // sigma test: flat enough
        alpha_new = alpha;
        return BfgsStatus::Success;
      }
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha;
        b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
        break;
      }
      const double delta = alpha - alpha_prev;
      const double next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + P.tau1 * delta, P.order);
      alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha;
      alpha = next;
    }
    while (i++ < P.section_iters) {  // sectioning of the bracket [a, b]
      const double delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + P.tau2 * delta, b - P.tau3 * delta, P.order);
      const double falpha = valueAt(alpha);
      if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return BfgsStatus::NoProgress;  // round-off prevents progress
      if (falpha > f0 + P.rho * alpha * fp0 || falpha >= fa) {
        b = alpha; fb = falpha; fpb = nan;
      } else {
        const double fpalpha = slopeAt(alpha);
        if (std::fabs(fpalpha) <= -P.sigma * fp0) {
          alpha_new = alpha;
          return BfgsStatus::Success;
        }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) { b = a; fb = fa; fpb = fpa; }
        a = alpha; fa = falpha; fpa = fpalpha;
      }
    }
    return BfgsStatus::Success;
  }

  Functor& fn_;
  Vec6 x0_{}, g0_{}, p_{}, gradient_{};
  double f_ = 0, delta_f_ = 0, fp0_ = 0, pnorm_ = 0, g0norm_ = 0;
  Trial trial_{};
};

}  // namespace mi355
