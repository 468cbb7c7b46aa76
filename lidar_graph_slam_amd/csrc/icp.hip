// pcl::IterativeClosestPoint on gfx950: the back end's DEFAULT registration method
// (graph_based_slam.param.yaml:9; graph_based_slam.cpp:142-151, called at :315-322).  Point-to-point ICP is the same
// data-parallel pass as GICP's update_correspondences + linearize with a smaller reduction: per source point an exact 1-NN
// in the target's voxel hash (32 algorithmic bytes: source point + target point) and 17 fp64 sums (count, sum p, sum q,
// sum p q^T, sum d2) from which the host fits the rigid transform (Horn's quaternion method = the rotation Umeyama's SVD
// returns) and evaluates PCL's DefaultConvergenceCriteria.  One launch + a 136-byte read-back per iteration: the loop is
// serial by definition (every iteration moves the cloud it searches from).
#include "engine.hpp"
#include "grid.hpp"
#include "primitives.hpp"

namespace mi355 {

constexpr int kIcpAcc = 17;
constexpr int kIcpBlock = 256;

struct Mat4f16 {
  float m[16];
};

// cur[i] <- T_inc * cur[i] (pcl::transformPointCloud arithmetic), then the correspondence sums at the NEW positions... no:
// PCL estimates the correspondences on the cloud as it stands, fits, THEN moves the cloud.  So one launch = [move by the
// previous iteration's fit] + [correspondences and sums at the moved positions]; the first launch moves by the guess.
__global__ __launch_bounds__(kIcpBlock) void k_icp_step(const float4* __restrict__ in, float4* __restrict__ cur, int n, Mat4f16 T,
                                                        const float4* __restrict__ tpts, const int* __restrict__ torig,
                                                        const HashSlot* __restrict__ ttable, const GridMeta* __restrict__ tmeta, size_t tcap,
                                                        size_t tslots, float max_d2, double* __restrict__ partials) {
  __shared__ double smem[4 * kIcpAcc];
  const GridView g = grid_view(tpts, torig, ttable, tmeta, 0, tcap, tslots);
  double acc[kIcpAcc];
#pragma unroll
  for (int k = 0; k < kIcpAcc; k++) acc[k] = 0.0;
  for (int i = blockIdx.x * kIcpBlock + threadIdx.x; i < n; i += gridDim.x * kIcpBlock) {
    const float4 p0 = in[i];
    const float4 p = make_float4(transform_row(T.m, p0.x, p0.y, p0.z), transform_row(T.m + 4, p0.x, p0.y, p0.z),
                                 transform_row(T.m + 8, p0.x, p0.y, p0.z), p0.w);
    cur[i] = p;
    Nearest1 nn{g.orig, INFINITY, -1};
    // CorrespondenceEstimation::determineCorrespondences keeps a pair iff d2 <= max_dist^2
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) grid_search(g, p.x, p.y, p.z, nextafterf(max_d2, INFINITY), nn, 0, 2);
    if (nn.best_j >= 0 && nn.best_d2 <= max_d2) {
      const float4 q = g.pts[nn.best_j];
      acc[0] += 1.0;
      acc[1] += (double)p.x; acc[2] += (double)p.y; acc[3] += (double)p.z;
      acc[4] += (double)q.x; acc[5] += (double)q.y; acc[6] += (double)q.z;
      acc[7] += (double)p.x * (double)q.x; acc[8] += (double)p.x * (double)q.y; acc[9] += (double)p.x * (double)q.z;
      acc[10] += (double)p.y * (double)q.x; acc[11] += (double)p.y * (double)q.y; acc[12] += (double)p.y * (double)q.z;
      acc[13] += (double)p.z * (double)q.x; acc[14] += (double)p.z * (double)q.y; acc[15] += (double)p.z * (double)q.z;
      acc[16] += (double)nn.best_d2;
    }
  }
  block_sum<kIcpAcc, 4>(acc, smem);
  if (threadIdx.x == 0) {
    double* out = partials + (size_t)blockIdx.x * kIcpAcc;
#pragma unroll
    for (int k = 0; k < kIcpAcc; k++) out[k] = acc[k];
  }
}

__global__ void k_icp_reduce(const double* __restrict__ partials, int nblk, double* __restrict__ sums) {
  const int k = threadIdx.x;
  if (k >= kIcpAcc) return;
  double s = 0;
  for (int b = 0; b < nblk; b++) s += partials[(size_t)b * kIcpAcc + k];  // fixed order: bitwise reproducible
  sums[k] = s;
}

// least-squares rigid fit q ~ R p + t from the sums (Horn 1987: largest eigenvector of the 4x4 quaternion matrix), fp64
static bool rigid_from_sums(const double* s, double* T16) {
  const double n = s[0];
  if (n < 3) return false;
  const double mp[3] = {s[1] / n, s[2] / n, s[3] / n}, mq[3] = {s[4] / n, s[5] / n, s[6] / n};
  double S[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) S[i][j] = s[7 + 3 * i + j] - n * mp[i] * mq[j];
  double A[16] = {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0],
                  S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2],
                  S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1],
                  S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]};
  double V[16];
  for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {  // cyclic Jacobi, symmetric 4x4
    double off = 0, diag = 0;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) { const double v = A[4 * i + j] * A[4 * i + j]; if (i == j) diag += v; else off += v; }
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        const double apq = A[4 * p + q];
        if (apq == 0.0) continue;
        const double tau = (A[4 * q + q] - A[4 * p + p]) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
        for (int k = 0; k < 4; k++) { const double a = A[4 * k + p], b = A[4 * k + q]; A[4 * k + p] = c * a - sn * b; A[4 * k + q] = sn * a + c * b; }
        for (int k = 0; k < 4; k++) { const double a = A[4 * p + k], b = A[4 * q + k]; A[4 * p + k] = c * a - sn * b; A[4 * q + k] = sn * a + c * b; }
        for (int k = 0; k < 4; k++) { const double a = V[4 * k + p], b = V[4 * k + q]; V[4 * k + p] = c * a - sn * b; V[4 * k + q] = sn * a + c * b; }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; k++)
    if (A[5 * k] > A[5 * best]) best = k;
  double q[4] = {V[best], V[4 + best], V[8 + best], V[12 + best]};
  const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (double& v : q) v /= nq;
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                       2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T16[4 * r + c] = R[3 * r + c];
    T16[4 * r + 3] = mq[r] - (R[3 * r] * mp[0] + R[3 * r + 1] * mp[1] + R[3 * r + 2] * mp[2]);
  }
  T16[12] = T16[13] = T16[14] = 0;
  T16[15] = 1;
  return true;
}

void icp_align(hipStream_t s, IcpScratch& sc, const CloudSet& tgt, const float4* d_src, size_t n, const mi355_icp_params& p, const float* guess16,
               mi355_result* out) {
  const int nblk = std::max(1, std::min(1024, div_up((long)n, kIcpBlock * 4)));
  sc.cur.ensure(std::max<size_t>(n, 1));
  sc.partials.ensure((size_t)nblk * kIcpAcc);
  sc.sums.ensure(kIcpAcc);
  sc.h_sums.ensure(kIcpAcc);
  const double md2 = p.max_correspondence_distance * p.max_correspondence_distance;
  const float max_d2 = md2 > 3.0e38 ? INFINITY : (float)md2;
  const double rot_thr = p.rotation_epsilon > 0 ? p.rotation_epsilon : 1.0 - p.transformation_epsilon;
  Mat4f16 Tinc;
  memcpy(Tinc.m, guess16, sizeof(Tinc.m));
  float final_T[16];
  memcpy(final_T, guess16, sizeof(final_T));
  int iterations = 0;
  bool converged = false;
  double prev_mse = 1.7976931348623157e308, mse = 0;
  const float4* in = d_src;  // the first step moves the input by the guess, later ones move `cur` in place
  for (;;) {
    TimedLaunch tl(KF_CORRESPOND, s);
    hipLaunchKernelGGL(k_icp_step, dim3(nblk), dim3(kIcpBlock), 0, s, in, sc.cur.p, (int)n, Tinc, (const float4*)tgt.pts.p, (const int*)tgt.orig.p,
                       (const HashSlot*)tgt.table.p, (const GridMeta*)tgt.meta.p, tgt.cap, tgt.slots_cap, max_d2, sc.partials.p);
    hipLaunchKernelGGL(k_icp_reduce, dim3(1), dim3(64), 0, s, (const double*)sc.partials.p, nblk, sc.sums.p);
    MI_HIP(hipMemcpyAsync(sc.h_sums.p, sc.sums.p, kIcpAcc * sizeof(double), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    in = sc.cur.p;
    const double* sm = sc.h_sums.p;
    double Td[16];
    if (!rigid_from_sums(sm, Td)) {  // "Not enough correspondences found. Relax your threshold parameters."
      converged = false;
      break;
    }
    for (int k = 0; k < 16; k++) Tinc.m[k] = (float)Td[k];
    float F[16];  // final_transformation_ = transformation_ * final_transformation_ (Matrix4f)
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        float a = 0.f;
        for (int k = 0; k < 4; k++) a += Tinc.m[4 * r + k] * final_T[4 * k + c];
        F[4 * r + c] = a;
      }
    memcpy(final_T, F, sizeof(F));
    iterations++;
    // DefaultConvergenceCriteria::hasConverged (max_iterations_similar_transforms_ = 0, absolute MSE threshold 1e-12)
    if (iterations >= p.max_iterations) { converged = true; break; }
    const double cos_angle = 0.5 * ((double)Tinc.m[0] + (double)Tinc.m[5] + (double)Tinc.m[10] - 1.0);
    const double tr2 = (double)Tinc.m[3] * Tinc.m[3] + (double)Tinc.m[7] * Tinc.m[7] + (double)Tinc.m[11] * Tinc.m[11];
    if (cos_angle >= rot_thr && tr2 <= p.transformation_epsilon) { converged = true; break; }
    mse = sm[16] / sm[0];
    if (fabs(mse - prev_mse) < 1e-12) { converged = true; break; }
    if (fabs(mse - prev_mse) / prev_mse < p.euclidean_fitness_epsilon) { converged = true; break; }
    prev_mse = mse;
  }
  memset(out, 0, sizeof(*out));
  memcpy(out->T, final_T, sizeof(final_T));
  out->converged = converged ? 1 : 0;
  out->iterations = iterations;
  out->error = mse;
}

}  // namespace mi355
