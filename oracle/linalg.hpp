// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/.  PARITY UNPINNED: the reference's numeric engines (fast_gicp, ndt_omp, PCL,
// Eigen) are un-vendored submodules / system libraries that are absent from /root/reference
// (.gitmodules:1-6), and the reference holds no golden vectors.
//
// Tiny fixed-size dense linear algebra in fp64 — the pieces of Eigen the restated algorithms
// need (Eigen itself is not installed in this image).  Row-major everywhere.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>

namespace oracle {

struct Mat3 {
  double m[9];
  double& operator()(int r, int c) { return m[3 * r + c]; }
  double operator()(int r, int c) const { return m[3 * r + c]; }
  static Mat3 zero() { Mat3 a; std::memset(a.m, 0, sizeof(a.m)); return a; }
  static Mat3 identity() { Mat3 a = zero(); a(0, 0) = a(1, 1) = a(2, 2) = 1.0; return a; }
};

struct Vec3 {
  double v[3];
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};

inline Mat3 mul(const Mat3& a, const Mat3& b) {
  Mat3 c = Mat3::zero();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) c(i, j) += a(i, k) * b(k, j);
  return c;
}
inline Mat3 transpose(const Mat3& a) {
  Mat3 t;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t(i, j) = a(j, i);
  return t;
}
inline Vec3 mul(const Mat3& a, const Vec3& x) {
  Vec3 y;
  for (int i = 0; i < 3; i++) y[i] = a(i, 0) * x[0] + a(i, 1) * x[1] + a(i, 2) * x[2];
  return y;
}
inline Mat3 add(const Mat3& a, const Mat3& b) {
  Mat3 c;
  for (int i = 0; i < 9; i++) c.m[i] = a.m[i] + b.m[i];
  return c;
}
inline Mat3 scale(const Mat3& a, double s) {
  Mat3 c;
  for (int i = 0; i < 9; i++) c.m[i] = a.m[i] * s;
  return c;
}
inline double det(const Mat3& a) {
  return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) +
         a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
}
// General 3x3 inverse by cofactors (what Eigen's fixed-size inverse() does for 3x3).
inline Mat3 inverse(const Mat3& a) {
  Mat3 c;
  c(0, 0) = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
  c(0, 1) = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
  c(0, 2) = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
  c(1, 0) = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
  c(1, 1) = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
  c(1, 2) = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
  c(2, 0) = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  c(2, 1) = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
  c(2, 2) = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
  const double d = a(0, 0) * c(0, 0) + a(0, 1) * c(1, 0) + a(0, 2) * c(2, 0);
  const double inv = 1.0 / d;
  for (int i = 0; i < 9; i++) c.m[i] *= inv;
  return c;
}
inline Mat3 skew(const Vec3& x) {
  Mat3 s = Mat3::zero();
  s(0, 1) = -x[2]; s(0, 2) = x[1];
  s(1, 0) = x[2];  s(1, 2) = -x[0];
  s(2, 0) = -x[1]; s(2, 1) = x[0];
  return s;
}
inline double frobenius(const Mat3& a) {
  double s = 0;
  for (int i = 0; i < 9; i++) s += a.m[i] * a.m[i];
  return std::sqrt(s);
}

// Cyclic Jacobi eigen-decomposition of a symmetric NxN matrix.  A = V diag(w) V^T, columns of V
// are eigenvectors.  Eigenvalues are returned UNSORTED; callers sort as they need.
template <int N>
inline void jacobi_eigen_sym(const double* A_in, double* w, double* V) {
  double A[N * N];
  std::memcpy(A, A_in, sizeof(A));
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) V[i * N + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) (i == j ? diag : off) += A[i * N + j] * A[i * N + j];
    if (off <= 1e-60 || off <= 1e-32 * diag) break;
    for (int p = 0; p < N - 1; p++)
      for (int q = p + 1; q < N; q++) {
        const double apq = A[p * N + q];
        if (apq == 0.0) continue;
        const double app = A[p * N + p], aqq = A[q * N + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < N; k++) {  // A <- A J (columns p,q)
          const double akp = A[k * N + p], akq = A[k * N + q];
          A[k * N + p] = c * akp - s * akq;
          A[k * N + q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; k++) {  // A <- J^T A (rows p,q)
          const double apk = A[p * N + k], aqk = A[q * N + k];
          A[p * N + k] = c * apk - s * aqk;
          A[q * N + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; k++) {
          const double vkp = V[k * N + p], vkq = V[k * N + q];
          V[k * N + p] = c * vkp - s * vkq;
          V[k * N + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < N; i++) w[i] = A[i * N + i];
}

// Eigen-decomposition of a symmetric 3x3 sorted by DESCENDING eigenvalue (= the singular-value order
// Eigen::JacobiSVD returns for a symmetric PSD input, where U == V).
inline void eigen_sym3_desc(const Mat3& a, double w[3], Mat3& U) {
  double ww[3], VV[9];
  jacobi_eigen_sym<3>(a.m, ww, VV);
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int i, int j) { return ww[i] > ww[j]; });
  for (int c = 0; c < 3; c++) {
    w[c] = ww[idx[c]];
    for (int r = 0; r < 3; r++) U(r, c) = VV[r * 3 + idx[c]];
  }
}

// 6x6 symmetric LDL^T with diagonal pivoting (restates what Eigen::LDLT<Matrix6d>::solve provides:
// the exact solution of (A) x = rhs for a symmetric positive semi-definite A).
inline bool ldlt6_solve(const double* A_in, const double* rhs, double* x) {
  constexpr int N = 6;
  double A[N * N];
  std::memcpy(A, A_in, sizeof(A));
  int perm[N];
  for (int i = 0; i < N; i++) perm[i] = i;
  double L[N * N] = {0}, D[N];
  for (int k = 0; k < N; k++) {
    int piv = k;
    double best = std::fabs(A[k * N + k]);
    for (int i = k + 1; i < N; i++)
      if (std::fabs(A[i * N + i]) > best) { best = std::fabs(A[i * N + i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < N; j++) std::swap(A[k * N + j], A[piv * N + j]);
      for (int j = 0; j < N; j++) std::swap(A[j * N + k], A[j * N + piv]);
      for (int j = 0; j < k; j++) std::swap(L[k * N + j], L[piv * N + j]);
      std::swap(perm[k], perm[piv]);
    }
    D[k] = A[k * N + k];
    L[k * N + k] = 1.0;
    if (D[k] == 0.0) {  // singular (e.g. no correspondences): Eigen's LDLT::solve returns the zero vector here
      for (int i = 0; i < N; i++) x[i] = 0.0;
      return false;
    }
    for (int i = k + 1; i < N; i++) L[i * N + k] = A[i * N + k] / D[k];
    for (int i = k + 1; i < N; i++)
      for (int j = k + 1; j < N; j++) A[i * N + j] -= L[i * N + k] * D[k] * L[j * N + k];
  }
  double y[N], z[N];
  for (int i = 0; i < N; i++) {
    double s = rhs[perm[i]];
    for (int j = 0; j < i; j++) s -= L[i * N + j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < N; i++) y[i] /= D[i];
  for (int i = N - 1; i >= 0; i--) {
    double s = y[i];
    for (int j = i + 1; j < N; j++) s -= L[j * N + i] * z[j];
    z[i] = s;
  }
  for (int i = 0; i < N; i++) x[perm[i]] = z[i];
  return true;
}

// Least-squares solve of a symmetric 6x6 system through its eigen-decomposition.  For a symmetric
// matrix the SVD is |lambda| with U = V*sign, so JacobiSVD(H).solve(rhs) == sum_i v_i (v_i.rhs)/lambda_i
// over singular values above Eigen's default rank threshold (eps * max|lambda|... times size).
inline void svd6_solve_sym(const double* H, const double* rhs, double* x) {
  constexpr int N = 6;
  double w[N], V[N * N];
  jacobi_eigen_sym<N>(H, w, V);
  double wmax = 0;
  for (int i = 0; i < N; i++) wmax = std::max(wmax, std::fabs(w[i]));
  const double thr = wmax * 2.220446049250313e-16 * N;
  for (int i = 0; i < N; i++) x[i] = 0;
  for (int k = 0; k < N; k++) {
    if (std::fabs(w[k]) <= thr) continue;
    double dot = 0;
    for (int i = 0; i < N; i++) dot += V[i * N + k] * rhs[i];
    dot /= w[k];
    for (int i = 0; i < N; i++) x[i] += V[i * N + k] * dot;
  }
}

// 4x4 rigid transform helper (row-major), fp64.
struct Iso3 {
  Mat3 R;
  Vec3 t;
  static Iso3 identity() { Iso3 x; x.R = Mat3::identity(); x.t = {{0, 0, 0}}; return x; }
};
inline Iso3 compose(const Iso3& a, const Iso3& b) {  // a * b
  Iso3 c;
  c.R = mul(a.R, b.R);
  Vec3 rt = mul(a.R, b.t);
  c.t = {{rt[0] + a.t[0], rt[1] + a.t[1], rt[2] + a.t[2]}};
  return c;
}
inline Iso3 from_rowmajor4x4f(const float* T) {
  Iso3 x;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) x.R(r, c) = (double)T[4 * r + c];
    x.t[r] = (double)T[4 * r + 3];
  }
  return x;
}
inline void to_rowmajor4x4f(const Iso3& x, float* T) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[4 * r + c] = (float)x.R(r, c);
    T[4 * r + 3] = (float)x.t[r];
  }
  T[12] = T[13] = T[14] = 0.f;
  T[15] = 1.f;
}

}  // namespace oracle
