// Small fixed-size maths shared by the HIP kernels (device) and the host-side control code.
// Everything is `MI_HD` so the same functions run inside kernels and in the C++ host mirror; the CPU unit
// tests (tests/test_host_math.py) compile this header with g++ to check it against the oracle without a GPU.
//
// Algorithms restated here (never copied — the upstream sources are not even vendored in the reference):
//   sym3 eigen / PLANE & friends regularisation ... fast_gicp calculate_covariances   
//   se3_exp / so3_exp ............................. fast_gicp so3.hpp                   (row F8)
//   ldlt6 ......................................... Eigen::LDLT used by step_lm / gn     (row F7)
//   is_converged .................................. LsqRegistration::is_converged       (row F9)
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MI_HD __host__ __device__ __forceinline__
#else
#define MI_HD inline
#endif

namespace mi355 {

// Symmetric 3x3 stored as (xx, xy, xz, yy, yz, zz).
template <typename T>
struct Sym3 {
  T xx, xy, xz, yy, yz, zz;
};

// Storage form of a symmetric 3x3 in HBM: the 6 unique floats, 24 bytes per record (no padding: the GICP streaming kernels
// are HBM-bound and these records are most of their bytes).  A record loads as one dwordx4 + one dwordx2.
struct Cov6 {
  float v[6];  // xx xy xz yy yz zz
};
MI_HD Cov6 cov6_pack(float xx, float xy, float xz, float yy, float yz, float zz) { return Cov6{{xx, xy, xz, yy, yz, zz}}; }
MI_HD Sym3<float> cov6_unpack(const Cov6& c) { return Sym3<float>{c.v[0], c.v[1], c.v[2], c.v[3], c.v[4], c.v[5]}; }

template <typename T>
MI_HD Sym3<T> sym3_inverse(const Sym3<T>& a) {
  // cofactor inverse (what a fixed-size 3x3 inverse evaluates)
  const T c00 = a.yy * a.zz - a.yz * a.yz;
  const T c01 = a.xz * a.yz - a.xy * a.zz;
  const T c02 = a.xy * a.yz - a.xz * a.yy;
  const T c11 = a.xx * a.zz - a.xz * a.xz;
  const T c12 = a.xy * a.xz - a.xx * a.yz;
  const T c22 = a.xx * a.yy - a.xy * a.xy;
  const T det = a.xx * c00 + a.xy * c01 + a.xz * c02;
  const T inv = T(1) / det;
  return {c00 * inv, c01 * inv, c02 * inv, c11 * inv, c12 * inv, c22 * inv};
}

// R C R^T for symmetric C and a 3x3 R given row-major.
template <typename T>
MI_HD Sym3<T> sym3_rotate(const T* R, const Sym3<T>& c) {
  // t = R * C
  T t[9];
  for (int r = 0; r < 3; r++) {
    const T a = R[3 * r], b = R[3 * r + 1], d = R[3 * r + 2];
    t[3 * r + 0] = a * c.xx + b * c.xy + d * c.xz;
    t[3 * r + 1] = a * c.xy + b * c.yy + d * c.yz;
    t[3 * r + 2] = a * c.xz + b * c.yz + d * c.zz;
  }
  Sym3<T> o;
  o.xx = t[0] * R[0] + t[1] * R[1] + t[2] * R[2];
  o.xy = t[0] * R[3] + t[1] * R[4] + t[2] * R[5];
  o.xz = t[0] * R[6] + t[1] * R[7] + t[2] * R[8];
  o.yy = t[3] * R[3] + t[4] * R[4] + t[5] * R[5];
  o.yz = t[3] * R[6] + t[4] * R[7] + t[5] * R[8];
  o.zz = t[6] * R[6] + t[7] * R[7] + t[8] * R[8];
  return o;
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 in fp64: A = V diag(w) V^T (columns of V).
// Eigenvalues unsorted.  Fixed rotation order (0,1),(0,2),(1,2) so host and device follow the same path.
MI_HD void jacobi_eigen_sym3(const Sym3<double>& s, double w[3], double V[9]) {
  double A[9] = {s.xx, s.xy, s.xz, s.xy, s.yy, s.yz, s.xz, s.yz, s.zz};
  for (int i = 0; i < 9; i++) V[i] = 0.0;
  V[0] = V[4] = V[8] = 1.0;
  for (int sweep = 0; sweep < 32; sweep++) {
    const double off = 2.0 * (A[1] * A[1] + A[2] * A[2] + A[5] * A[5]);
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-60 || off <= 1e-32 * diag) break;
    for (int pq = 0; pq < 3; pq++) {
      const int p = (pq == 2) ? 1 : 0;
      const int q = (pq == 0) ? 1 : 2;
      const double apq = A[3 * p + q];
      if (apq == 0.0) continue;
      const double app = A[3 * p + p], aqq = A[3 * q + q];
      const double tau = (aqq - app) / (2.0 * apq);
      const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
      for (int k = 0; k < 3; k++) {
        const double akp = A[3 * k + p], akq = A[3 * k + q];
        A[3 * k + p] = c * akp - sn * akq;
        A[3 * k + q] = sn * akp + c * akq;
      }
      for (int k = 0; k < 3; k++) {
        const double apk = A[3 * p + k], aqk = A[3 * q + k];
        A[3 * p + k] = c * apk - sn * aqk;
        A[3 * q + k] = sn * apk + c * aqk;
      }
      for (int k = 0; k < 3; k++) {
        const double vkp = V[3 * k + p], vkq = V[3 * k + q];
        V[3 * k + p] = c * vkp - sn * vkq;
        V[3 * k + q] = sn * vkp + c * vkq;
      }
    }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
}

MI_HD void sort3_desc(const double w[3], int o[3]) {
  o[0] = 0; o[1] = 1; o[2] = 2;
  if (w[o[0]] < w[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  if (w[o[1]] < w[o[2]]) { int t = o[1]; o[1] = o[2]; o[2] = t; }
  if (w[o[0]] < w[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
}

enum RegularizationMethod { REG_NONE = 0, REG_MIN_EIG = 1, REG_NORMALIZED_MIN_EIG = 2, REG_PLANE = 3, REG_FROBENIUS = 4 };

// Unit eigenvector of the SMALLEST eigenvalue of a symmetric positive semi-definite 3x3, in fp64, without the sweeps of
// the Jacobi iteration (which cost the covariance kernel more than the neighbour search itself: 5-6 sweeps x 3 rotations
// x 2 sqrt + 2 div in fp64, and a data-dependent trip count under SIMT).  PLANE regularisation needs nothing else:
// U diag(1, 1, 1e-3) U^T = I - (1 - 1e-3) n n^T.
//   1. the eigenvalue that is best separated from the other two (the smallest for a plane-like neighbourhood, the largest
//      for a line-like one; the sign of det(A - tr/3 I) tells which) is a simple, well-conditioned root of the
//      characteristic polynomial: Newton from outside the spectrum (0 or the trace) converges to it monotonically;
//   2. its eigenvector is the largest cross product of two rows of A - lambda I (error ~ eps / gap, like Jacobi's);
//   3. the other two eigenpairs live in the orthogonal complement: ONE exact Jacobi rotation of the projected 2x2.
// Returns false when the input is (numerically) degenerate — isotropic, rank <= 1, or the two smallest eigenvalues closer
// than 1e-9 of the trace — so that the caller takes the Jacobi path and degenerate inputs resolve exactly as before.
MI_HD bool smallest_eigenvector_sym3(const Sym3<double>& cv, double n[3]) {
  const double tr = cv.xx + cv.yy + cv.zz;
  if (!(tr > 0.0) || !(tr < 1.0e300)) return false;
  const double sc = 1.0 / tr;
  const double axx = cv.xx * sc, axy = cv.xy * sc, axz = cv.xz * sc, ayy = cv.yy * sc, ayz = cv.yz * sc, azz = cv.zz * sc;
  // characteristic polynomial of the trace-1 matrix:  p(l) = l^3 - l^2 + m l - d
  const double m = (axx * ayy - axy * axy) + (axx * azz - axz * axz) + (ayy * azz - ayz * ayz);
  const double d = axx * (ayy * azz - ayz * ayz) - axy * (axy * azz - ayz * axz) + axz * (axy * ayz - ayy * axz);
  // det(A - I/3) = -p(1/3): negative -> two eigenvalues above the mean, one below: the smallest is the separated one
  const double third = 1.0 / 3.0;
  const double pm = ((third - 1.0) * third + m) * third - d;
  const bool small_sep = pm > 0.0;
  double l = small_sep ? 0.0 : 1.0;
  {
    // Warm start (round 6): the first Newton steps in fp32 - an fp64 step costs ~35 instructions, most of them the division, and the
    // covariance epilogue of k_knn_cov_tile ran 6 - 10 of them per point.  From outside the spectrum the iteration is monotone (p is
    // concave below 1/3 and convex above: the tangent's root lies on the starting side of the polynomial's), so in exact arithmetic it
    // never passes the root; fp32 rounding may, by ~1e-7 relative, and the result is moved back OUT by 1e-5 relative before the fp64
    // iteration takes over - still monotone, now within 1e-5 of the root it converges to as before (|step| <= 2e-17).
    const float mf = (float)m, df = (float)d;
    float lf = small_sep ? 0.f : 1.f;
    for (int it = 0; it < 6; it++) {
      const float p = ((lf - 1.f) * lf + mf) * lf - df;
      const float dp = (3.f * lf - 2.f) * lf + mf;
      if (dp == 0.f) break;
      lf -= p / dp;
    }
    const float back = small_sep ? lf * (1.f - 1e-5f) : lf * (1.f + 1e-5f);
    if (back >= 0.f && back <= 1.f) l = (double)back;  // (NaN or an escape from [0, 1]: the cold start)
  }
  for (int it = 0; it < 24; it++) {
    const double p = ((l - 1.0) * l + m) * l - d;
    const double dp = (3.0 * l - 2.0) * l + m;
    if (dp == 0.0) break;
    const double step = p / dp;
    l -= step;
    if (fabs(step) <= 2.0e-17) break;
  }
  // eigenvector of l: rows of B = A - l I are orthogonal to it
  const double bxx = axx - l, byy = ayy - l, bzz = azz - l;
  const double c0x = axy * ayz - axz * byy, c0y = axz * axy - bxx * ayz, c0z = bxx * byy - axy * axy;  // r0 x r1
  const double c1x = axy * bzz - axz * ayz, c1y = axz * axz - bxx * bzz, c1z = bxx * ayz - axy * axz;  // r0 x r2
  const double c2x = byy * bzz - ayz * ayz, c2y = ayz * axz - axy * bzz, c2z = axy * ayz - byy * axz;  // r1 x r2
  const double n0 = c0x * c0x + c0y * c0y + c0z * c0z, n1 = c1x * c1x + c1y * c1y + c1z * c1z, n2 = c2x * c2x + c2y * c2y + c2z * c2z;
  double vx = c0x, vy = c0y, vz = c0z, nn = n0;
  if (n1 > nn) { vx = c1x; vy = c1y; vz = c1z; nn = n1; }
  if (n2 > nn) { vx = c2x; vy = c2y; vz = c2z; nn = n2; }
  // |cross| ~ (gap to the 2nd) * (gap to the 3rd) of the trace-1 matrix: below 1e-8 the root was not a separated one
  if (!(nn > 1.0e-16)) return false;
  const double inv = 1.0 / sqrt(nn);
  vx *= inv; vy *= inv; vz *= inv;
  if (small_sep) { n[0] = vx; n[1] = vy; n[2] = vz; return true; }
  // orthonormal basis (u, w) of the complement of v: u = v x e_i for the axis where |v_i| is smallest
  const double ax = fabs(vx), ay = fabs(vy), az = fabs(vz);
  double ux, uy, uz;
  if (ax <= ay && ax <= az) { ux = 0.0; uy = vz; uz = -vy; }
  else if (ay <= az) { ux = -vz; uy = 0.0; uz = vx; }
  else { ux = vy; uy = -vx; uz = 0.0; }
  const double iu = 1.0 / sqrt(ux * ux + uy * uy + uz * uz);
  ux *= iu; uy *= iu; uz *= iu;
  const double wx = vy * uz - vz * uy, wy = vz * ux - vx * uz, wz = vx * uy - vy * ux;
  // projected 2x2: [uAu uAw; uAw wAw]
  const double Aux = axx * ux + axy * uy + axz * uz, Auy = axy * ux + ayy * uy + ayz * uz, Auz = axz * ux + ayz * uy + azz * uz;
  const double Awx = axx * wx + axy * wy + axz * wz, Awy = axy * wx + ayy * wy + ayz * wz, Awz = axz * wx + ayz * wy + azz * wz;
  const double app = ux * Aux + uy * Auy + uz * Auz, aqq = wx * Awx + wy * Awy + wz * Awz, apq = ux * Awx + uy * Awy + uz * Awz;
  // one Jacobi rotation diagonalises it: eigenvalues app - t apq (vector c u - s w) and aqq + t apq (vector s u + c w)
  double c = 1.0, s = 0.0, t = 0.0;
  if (apq != 0.0) {
    const double tau = (aqq - app) / (2.0 * apq);
    t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    c = 1.0 / sqrt(1.0 + t * t);
    s = t * c;
  }
  const double e1 = app - t * apq, e2 = aqq + t * apq;
  if (!(fabs(e1 - e2) > 1.0e-9)) return false;  // the two smallest eigenvalues (nearly) coincide: no defined normal
  if (e1 < e2) { n[0] = c * ux - s * wx; n[1] = c * uy - s * wy; n[2] = c * uz - s * wz; }
  else { n[0] = s * ux + c * wx; n[1] = s * uy + c * wy; n[2] = s * uz + c * wz; }
  return true;
}

// fast_gicp covariance regularisation (row F2).  Input: raw neighbourhood covariance (divided by k).
MI_HD Sym3<double> regularize_covariance(const Sym3<double>& cov, int method) {
  if (method == REG_NONE) return cov;
  if (method == REG_FROBENIUS) {
    const double lambda = 1e-3;
    Sym3<double> C = cov;
    C.xx += lambda; C.yy += lambda; C.zz += lambda;
    const Sym3<double> Ci = sym3_inverse(C);
    const double fro = sqrt(Ci.xx * Ci.xx + Ci.yy * Ci.yy + Ci.zz * Ci.zz + 2.0 * (Ci.xy * Ci.xy + Ci.xz * Ci.xz + Ci.yz * Ci.yz));
    const double s = 1.0 / fro;
    const Sym3<double> Cn = {Ci.xx * s, Ci.xy * s, Ci.xz * s, Ci.yy * s, Ci.yz * s, Ci.zz * s};
    return sym3_inverse(Cn);
  }
  if (method == REG_PLANE) {
    double n[3];
    if (smallest_eigenvector_sym3(cov, n)) {
      const double f = 1.0 - 1e-3;
      return {1.0 - f * n[0] * n[0], -f * n[0] * n[1], -f * n[0] * n[2], 1.0 - f * n[1] * n[1], -f * n[1] * n[2], 1.0 - f * n[2] * n[2]};
    }
  }
  double w[3], V[9];
  jacobi_eigen_sym3(cov, w, V);
  int o[3];
  sort3_desc(w, o);  // singular-value order of the SVD of a symmetric PSD matrix
  double values[3];
  if (method == REG_PLANE) {
    values[0] = 1.0; values[1] = 1.0; values[2] = 1e-3;
  } else if (method == REG_MIN_EIG) {
    for (int d = 0; d < 3; d++) values[d] = fmax(w[o[d]], 1e-3);
  } else {
    const double mx = fmax(w[o[0]], fmax(w[o[1]], w[o[2]]));
    for (int d = 0; d < 3; d++) values[d] = fmax(w[o[d]] / mx, 1e-3);
  }
  Sym3<double> out = {0, 0, 0, 0, 0, 0};
  for (int d = 0; d < 3; d++) {
    const double ux = V[0 + o[d]], uy = V[3 + o[d]], uz = V[6 + o[d]], v = values[d];
    out.xx += ux * v * ux; out.xy += ux * v * uy; out.xz += ux * v * uz;
    out.yy += uy * v * uy; out.yz += uy * v * uz; out.zz += uz * v * uz;
  }
  return out;
}

// Rigid transform in fp64: R row-major (9) + t (3).
struct Pose {
  double R[9];
  double t[3];
};

MI_HD Pose pose_identity() {
  Pose p;
  for (int i = 0; i < 9; i++) p.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  p.t[0] = p.t[1] = p.t[2] = 0.0;
  return p;
}

MI_HD Pose pose_compose(const Pose& a, const Pose& b) {  // a * b
  Pose c;
  for (int r = 0; r < 3; r++) {
    for (int s = 0; s < 3; s++) c.R[3 * r + s] = a.R[3 * r] * b.R[s] + a.R[3 * r + 1] * b.R[3 + s] + a.R[3 * r + 2] * b.R[6 + s];
    c.t[r] = a.R[3 * r] * b.t[0] + a.R[3 * r + 1] * b.t[1] + a.R[3 * r + 2] * b.t[2] + a.t[r];
  }
  return c;
}

MI_HD Pose pose_from_rowmajor4x4f(const float* T) {
  Pose p;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) p.R[3 * r + c] = (double)T[4 * r + c];
    p.t[r] = (double)T[4 * r + 3];
  }
  return p;
}

MI_HD void pose_to_rowmajor4x4f(const Pose& p, float* T) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[4 * r + c] = (float)p.R[3 * r + c];
    T[4 * r + 3] = (float)p.t[r];
  }
  T[12] = T[13] = T[14] = 0.f;
  T[15] = 1.f;
}

// Row F8: so3_exp (quaternion, small-angle series below theta^2 = 1e-10) and se3_exp (V matrix).
MI_HD Pose se3_exp(const double a[6]) {
  const double ox = a[0], oy = a[1], oz = a[2];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  double imag_factor, real_factor;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    imag_factor = sin(half_theta) / theta;
    real_factor = cos(half_theta);
  }
  const double w = real_factor, x = imag_factor * ox, y = imag_factor * oy, z = imag_factor * oz;
  Pose p;
  {
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    p.R[0] = 1 - (tyy + tzz); p.R[1] = txy - twz;       p.R[2] = txz + twy;
    p.R[3] = txy + twz;       p.R[4] = 1 - (txx + tzz); p.R[5] = tyz - twx;
    p.R[6] = txz - twy;       p.R[7] = tyz + twx;       p.R[8] = 1 - (txx + tyy);
  }
  const double theta = sqrt(theta_sq);
  double V[9];
  if (theta < 1e-10) {
    for (int i = 0; i < 9; i++) V[i] = p.R[i];
  } else {
    const double O[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
    double O2[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) O2[3 * r + c] = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
    const double tsq = theta * theta;
    const double k1 = (1.0 - cos(theta)) / tsq, k2 = (theta - sin(theta)) / (tsq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + k1 * O[i] + k2 * O2[i];
  }
  for (int r = 0; r < 3; r++) p.t[r] = V[3 * r] * a[3] + V[3 * r + 1] * a[4] + V[3 * r + 2] * a[5];
  return p;
}

// Row F9.
MI_HD bool is_converged(const Pose& delta, double rot_eps, double trans_eps) {
  double rmax = 0, tmax = 0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) rmax = fmax(rmax, fabs(delta.R[3 * r + c] - (r == c ? 1.0 : 0.0)) / rot_eps);
    tmax = fmax(tmax, fabs(delta.t[r]) / trans_eps);
  }
  return fmax(rmax, tmax) < 1;
}

// 6x6 symmetric LDL^T with largest-diagonal pivoting; solves A x = rhs.
MI_HD bool ldlt6_solve(const double* A_in, const double* rhs, double* x, double* D_out = nullptr) {
  // Every index below is a compile-time constant once the loops are unrolled: the pivot row is CHOSEN at run time, but the swap is
  // written as "for every candidate row c: if (piv == c) swap" - so the matrices live in registers on the device.  (Indexed by the
  // run-time pivot they lived in scratch memory: 1.7 KB per lane of k_gicp_solve, every access a trip to memory, ~10 of the 15 us
  // that kernel took for a lone pair.)  Same operations in the same order as the plain loops: same bits.
  constexpr int N = 6;
  double A[N * N], L[N * N], D[N], y[N], z[N], r[N];
  int perm[N];
#pragma unroll
  for (int i = 0; i < N * N; i++) { A[i] = A_in[i]; L[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < N; i++) { perm[i] = i; r[i] = rhs[i]; }
  bool singular = false;
#pragma unroll
  for (int k = 0; k < N; k++) {
    int piv = k;
    double best = fabs(A[k * N + k]);
#pragma unroll
    for (int i = k + 1; i < N; i++)
      if (fabs(A[i * N + i]) > best) { best = fabs(A[i * N + i]); piv = i; }
#pragma unroll
    for (int c = k + 1; c < N; c++) {
      if (piv == c) {
#pragma unroll
        for (int j = 0; j < N; j++) { const double t = A[k * N + j]; A[k * N + j] = A[c * N + j]; A[c * N + j] = t; }
#pragma unroll
        for (int j = 0; j < N; j++) { const double t = A[j * N + k]; A[j * N + k] = A[j * N + c]; A[j * N + c] = t; }
#pragma unroll
        for (int j = 0; j < k; j++) { const double t = L[k * N + j]; L[k * N + j] = L[c * N + j]; L[c * N + j] = t; }
        { const int t = perm[k]; perm[k] = perm[c]; perm[c] = t; }
        { const double t = r[k]; r[k] = r[c]; r[c] = t; }  // r[i] = rhs[perm[i]]
      }
    }
    D[k] = A[k * N + k];
    L[k * N + k] = 1.0;
    if (D[k] == 0.0) singular = true;  // (e.g. no correspondences): Eigen's LDLT::solve returns the zero vector here
    if (!singular) {
#pragma unroll
      for (int i = k + 1; i < N; i++) L[i * N + k] = A[i * N + k] / D[k];
#pragma unroll
      for (int i = k + 1; i < N; i++)
#pragma unroll
        for (int j = k + 1; j < N; j++) A[i * N + j] -= L[i * N + k] * D[k] * L[j * N + k];
    }
  }
  if (D_out) {
#pragma unroll
    for (int i = 0; i < N; i++) D_out[i] = D[i];
  }
  if (singular) {
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = 0.0;
    return false;
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    double s = r[i];
#pragma unroll
    for (int j = 0; j < i; j++) s -= L[i * N + j] * y[j];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < N; i++) y[i] /= D[i];
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int j = i + 1; j < N; j++) s -= L[j * N + i] * z[j];
    z[i] = s;
  }
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = 0; j < N; j++)
      if (perm[i] == j) x[j] = z[i];
  return true;
}

// Cyclic Jacobi for a symmetric 6x6 (NDT's JacobiSVD(H).solve(-g): for a symmetric matrix the SVD solve is the
// eigen-decomposition solve over |lambda| above the rank threshold).  Every index is a compile-time constant once the loops are
// unrolled (only the sweep loop stays rolled), so A and V live in registers on the device — indexed by run-time p, q they lived in
// scratch memory, every access a trip to memory: 0.3 - 0.6 ms per solve on one lane.  Same operations in the same order: same bits.
MI_HD void svd6_solve_sym(const double* H, const double* rhs, double* x) {
  constexpr int N = 6;
  double A[N * N], V[N * N];
#pragma unroll
  for (int i = 0; i < N * N; i++) { A[i] = H[i]; V[i] = ((i % (N + 1)) == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0, diag = 0;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
      for (int j = 0; j < N; j++) {
        const double v = A[i * N + j] * A[i * N + j];
        if (i == j) diag += v; else off += v;
      }
    if (off <= 1e-60 || off <= 1e-32 * diag) break;
#pragma unroll
    for (int p = 0; p < N - 1; p++)
#pragma unroll
      for (int q = p + 1; q < N; q++) {
        const double apq = A[p * N + q];
        if (apq != 0.0) {
          const double app = A[p * N + p], aqq = A[q * N + q];
          const double tau = (aqq - app) / (2.0 * apq);
          const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
          for (int k = 0; k < N; k++) {
            const double akp = A[k * N + p], akq = A[k * N + q];
            A[k * N + p] = c * akp - s * akq;
            A[k * N + q] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < N; k++) {
            const double apk = A[p * N + k], aqk = A[q * N + k];
            A[p * N + k] = c * apk - s * aqk;
            A[q * N + k] = s * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < N; k++) {
            const double vkp = V[k * N + p], vkq = V[k * N + q];
            V[k * N + p] = c * vkp - s * vkq;
            V[k * N + q] = s * vkp + c * vkq;
          }
        }
      }
  }
  double wmax = 0;
#pragma unroll
  for (int i = 0; i < N; i++) wmax = fmax(wmax, fabs(A[i * N + i]));
  const double thr = wmax * 2.220446049250313e-16 * N;
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double wk = A[k * N + k];
    if (fabs(wk) > thr) {
      double dot = 0;
#pragma unroll
      for (int i = 0; i < N; i++) dot += V[i * N + k] * rhs[i];
      dot /= wk;
#pragma unroll
      for (int i = 0; i < N; i++) x[i] += V[i * N + k] * dot;
    }
  }
}

// Specified fp32 evaluation order shared with the oracle (oracle/kdtree.hpp, oracle_pcl.cpp::orc_transform):
//   d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx));   T*p row = fmaf(r0,x, fmaf(r1,y, fmaf(r2,z, t)))
MI_HD float dist2f(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}
MI_HD float transform_row(const float* Trow, float x, float y, float z) {
  return fmaf(Trow[0], x, fmaf(Trow[1], y, fmaf(Trow[2], z, Trow[3])));
}

}  // namespace mi355
