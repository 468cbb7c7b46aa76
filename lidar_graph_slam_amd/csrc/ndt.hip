// NDT on gfx950: pclomp::NormalDistributionsTransform (DIRECT7) + pclomp::VoxelGridCovariance — SURVEY.md §8a rows N1-N5.
//
//   target build  : PCL voxel index per point -> batched radix sort -> one thread per occupied voxel accumulates
//                   sum / sum of outer products in fp64 (ascending point index), eigen-inflates (eq. 6.11 [Magnusson 2009])
//                   and inverts; usable voxels (>= min_points_per_voxel, valid eigenvalues) go into an open-addressing
//                   hash keyed by the linear voxel index, so DIRECT7 is 7 hash probes — no kd-tree.
//   derivatives   : one thread per source point: fp32 transform, 7-voxel gather, score + gradient(6) + Hessian(21 unique)
//                   with ndt_omp's fp32 per-point maths, accumulated in fp64 registers and reduced by a fixed-order
//                   wave/LDS tree (one partial per block).  Algorithmic bytes per point: 16 + 7 x 36.
//   line search   : More-Thuente on the host (scalar logic on 8 doubles); every trial costs one derivative launch and a
//                   224-byte read-back.
#include <cmath>

#include "engine.hpp"
#include "primitives.hpp"

namespace mi355 {

struct NdtLeafF {     // what the fp32 derivative kernel reads: 64 B, one per usable voxel
  double mean[3];
  float icov[9];
  float pad;
};
struct NdtLeafD {     // fp64 view for computeHessian and for parity read-back
  double mean[3];
  double cov[9];
  double icov[9];
  int n;              // nr_points (-1 = rejected)
  int pad;
};

struct NdtGridMeta {
  int min_b[3], max_b[3], div_b[3];
  float inv_leaf;
  uint32_t hshift, hmask;
  int n_leaves;
};

struct NdtTarget {
  size_t n_pts = 0, cap = 0;
  DevBuf<NdtLeafF> leaf_f;
  DevBuf<NdtLeafD> leaf_d;
  DevBuf<uint32_t> leaf_key;
  DevBuf<HashSlot> table;   // key -> leaf index in .start
  size_t slots = 0;
  DevBuf<NdtGridMeta> meta;
  NdtGridMeta h_meta;
  DevBuf<int> d_count;
  DevBuf<float4> trans;
  DevBuf<double> partials;
  DevBuf<double> sums;
  PinnedBuf<double> h_sums;
  int blocks = 0;
};

NdtTarget* ndt_target_create() { return new NdtTarget(); }
void ndt_target_destroy(NdtTarget* t) { delete t; }
size_t ndt_num_cells(NdtTarget& t) { return (size_t)t.h_meta.n_leaves; }

// ---- target build ---------------------------------------------------------------------------------------------------
__global__ void k_ndt_meta(const uint32_t* __restrict__ bounds, NdtGridMeta* __restrict__ meta, float resolution, int n_pts, uint32_t slots) {
  NdtGridMeta m;
  const float inv = 1.0f / resolution;
  for (int d = 0; d < 3; d++) {
    const float mn = ordered_to_float(bounds[d]), mx = ordered_to_float(bounds[3 + d]);
    m.min_b[d] = (int)floorf(mn * inv);
    m.max_b[d] = (int)floorf(mx * inv);
    m.div_b[d] = m.max_b[d] - m.min_b[d] + 1;
  }
  m.inv_leaf = inv;
  uint32_t lg = 0;
  while ((1u << lg) < slots) lg++;
  m.hshift = 32 - lg;
  m.hmask = slots - 1;
  m.n_leaves = 0;
  (void)n_pts;
  *meta = m;
}

__global__ __launch_bounds__(256) void k_ndt_keys(const float4* __restrict__ pts, const NdtGridMeta* __restrict__ meta, uint32_t* __restrict__ keys,
                                                  int* __restrict__ vals, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const NdtGridMeta m = *meta;
  const float4 p = pts[i];
  const int i0 = (int)(floorf(p.x * m.inv_leaf) - (float)m.min_b[0]);
  const int i1 = (int)(floorf(p.y * m.inv_leaf) - (float)m.min_b[1]);
  const int i2 = (int)(floorf(p.z * m.inv_leaf) - (float)m.min_b[2]);
  keys[i] = (uint32_t)(i0 + i1 * m.div_b[0] + i2 * m.div_b[0] * m.div_b[1]);
  vals[i] = i;
}

__global__ __launch_bounds__(256) void k_ndt_heads(const uint32_t* __restrict__ keys, int* __restrict__ flags, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || keys[i - 1] != keys[i]) ? 1 : 0;
}

__global__ __launch_bounds__(128) void k_ndt_leaves(const float4* __restrict__ pts, const uint32_t* __restrict__ keys, const int* __restrict__ vals,
                                                    const int* __restrict__ flags, const int* __restrict__ local, const int* __restrict__ blockoff,
                                                    const int* __restrict__ totals, NdtGridMeta* __restrict__ meta, NdtLeafF* __restrict__ leaf_f,
                                                    NdtLeafD* __restrict__ leaf_d, uint32_t* __restrict__ leaf_key, HashSlot* __restrict__ table,
                                                    int n, int min_points, double eig_mult) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) meta->n_leaves = totals[0];
  if (i >= n || !flags[i]) return;
  const int li = local[i] + blockoff[i / kScanTile];
  const uint32_t key = keys[i];
  double sx = 0, sy = 0, sz = 0;
  Sym3<double> sq = {0, 0, 0, 0, 0, 0};
  int j = i;
  for (; j < n && keys[j] == key; j++) {
    const float4 p = pts[vals[j]];
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    sx += x; sy += y; sz += z;
    sq.xx += x * x; sq.xy += x * y; sq.xz += x * z; sq.yy += y * y; sq.yz += y * z; sq.zz += z * z;
  }
  const int cnt = j - i;
  const double nn = (double)cnt;
  NdtLeafD L;
  L.n = cnt;
  L.pad = 0;
  const double mx = sx / nn, my = sy / nn, mz = sz / nn;
  L.mean[0] = mx; L.mean[1] = my; L.mean[2] = mz;
  for (int k = 0; k < 9; k++) { L.cov[k] = 0; L.icov[k] = 0; }
  bool usable = cnt >= min_points;
  if (usable) {
    // single pass covariance: (sum pp^T - 2 (pt_sum mean^T)) / n + mean mean^T, then * (n-1)/n
    const double f = (nn - 1.0) / nn;
    Sym3<double> c;
    c.xx = ((sq.xx - 2 * (sx * mx)) / nn + mx * mx) * f;
    c.xy = ((sq.xy - 2 * (sx * my)) / nn + mx * my) * f;
    c.xz = ((sq.xz - 2 * (sx * mz)) / nn + mx * mz) * f;
    c.yy = ((sq.yy - 2 * (sy * my)) / nn + my * my) * f;
    c.yz = ((sq.yz - 2 * (sy * mz)) / nn + my * mz) * f;
    c.zz = ((sq.zz - 2 * (sz * mz)) / nn + mz * mz) * f;
    double w[3], V[9];
    jacobi_eigen_sym3(c, w, V);
    int o[3];
    sort3_desc(w, o);  // o[2] smallest ... o[0] largest
    double ev[3] = {w[o[2]], w[o[1]], w[o[0]]};  // ascending, as SelfAdjointEigenSolver orders them
    if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) {
      L.n = -1;
      usable = false;
    } else {
      const double min_ev = eig_mult * ev[2];
      if (ev[0] < min_ev) {
        ev[0] = min_ev;
        if (ev[1] < min_ev) ev[1] = min_ev;
        const int oa[3] = {o[2], o[1], o[0]};
        Sym3<double> r = {0, 0, 0, 0, 0, 0};
        for (int d = 0; d < 3; d++) {
          const double ux = V[0 + oa[d]], uy = V[3 + oa[d]], uz = V[6 + oa[d]], v = ev[d];
          r.xx += ux * v * ux; r.xy += ux * v * uy; r.xz += ux * v * uz;
          r.yy += uy * v * uy; r.yz += uy * v * uz; r.zz += uz * v * uz;
        }
        c = r;
      }
      const Sym3<double> ic = sym3_inverse(c);
      L.cov[0] = c.xx; L.cov[1] = c.xy; L.cov[2] = c.xz; L.cov[3] = c.xy; L.cov[4] = c.yy; L.cov[5] = c.yz; L.cov[6] = c.xz; L.cov[7] = c.yz; L.cov[8] = c.zz;
      L.icov[0] = ic.xx; L.icov[1] = ic.xy; L.icov[2] = ic.xz; L.icov[3] = ic.xy; L.icov[4] = ic.yy; L.icov[5] = ic.yz; L.icov[6] = ic.xz; L.icov[7] = ic.yz; L.icov[8] = ic.zz;
      bool bad = false;
      for (int k = 0; k < 9; k++) bad = bad || isinf(L.icov[k]);
      if (bad) { L.n = -1; usable = false; }
    }
  }
  leaf_d[li] = L;
  leaf_key[li] = key;
  NdtLeafF F;
  for (int k = 0; k < 3; k++) F.mean[k] = L.mean[k];
  for (int k = 0; k < 9; k++) F.icov[k] = (float)L.icov[k];
  F.pad = 0.f;
  leaf_f[li] = F;
  if (usable) {
    const NdtGridMeta m = *meta;
    uint32_t h = (key * 0x9E3779B1u) >> m.hshift;
    for (;;) {
      const uint32_t prev = atomicCAS(&table[h].key, kEmptyKey, key);
      if (prev == kEmptyKey) { table[h].start = li; break; }
      h = (h + 1) & m.hmask;
    }
  }
}

__global__ void k_ndt_table_clear(HashSlot* t, size_t slots) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots) reinterpret_cast<int4*>(t)[i] = make_int4((int)kEmptyKey, 0, 0, 0);
}

// bounds helpers live in cloud.hip; re-declare the kernels we need through a tiny local copy to keep TUs independent
__global__ void k_ndt_bounds_init(uint32_t* b) {
  if (threadIdx.x < 6) b[threadIdx.x] = (threadIdx.x < 3) ? 0xFFFFFFFFu : 0u;
}
__global__ __launch_bounds__(256) void k_ndt_bounds(const float4* __restrict__ p, int n, uint32_t* __restrict__ bounds) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 v = p[i];
    mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z);
    mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z);
  }
  for (int d = 0; d < 3; d++)
    for (int off = 32; off > 0; off >>= 1) {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], off, kWave));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off, kWave));
    }
  if (lane_id() == 0 && mn[0] <= mx[0])
    for (int d = 0; d < 3; d++) {
      atomicMin(&bounds[d], float_to_ordered(mn[d]));
      atomicMax(&bounds[3 + d], float_to_ordered(mx[d]));
    }
}

void ndt_build_target(hipStream_t s, Workspace& ws, NdtTarget& t, const float4* d_pts, size_t n, const mi355_ndt_params& p) {
  if (!(p.resolution > 0)) throw ArgError(MI355_EINVAL, "ndt resolution must be positive");
  t.n_pts = n;
  ws_reserve(ws, 1, n);
  t.leaf_f.ensure(n); t.leaf_d.ensure(n); t.leaf_key.ensure(n);
  size_t slots = 16;
  while (slots < 2 * n) slots <<= 1;
  t.slots = slots;
  t.table.ensure(slots);
  t.meta.ensure(1);
  t.d_count.ensure(1);
  const int ni = (int)n;
  MI_HIP(hipMemcpyAsync(t.d_count.p, &ni, sizeof(int), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_ndt_bounds_init, dim3(1), dim3(64), 0, s, ws.bounds.p);
  hipLaunchKernelGGL(k_ndt_bounds, dim3(std::max(1, std::min(256, div_up((long)n, 1024)))), dim3(256), 0, s, d_pts, ni, ws.bounds.p);
  hipLaunchKernelGGL(k_ndt_meta, dim3(1), dim3(1), 0, s, ws.bounds.p, t.meta.p, (float)p.resolution, ni, (uint32_t)slots);
  const dim3 grid(div_up((long)n, 256));
  hipLaunchKernelGGL(k_ndt_keys, grid, dim3(256), 0, s, d_pts, t.meta.p, ws.keys_a.p, ws.vals_a.p, ni);
  sort_pairs(s, ws, 1, n, t.d_count.p, 4);
  hipLaunchKernelGGL(k_ndt_heads, grid, dim3(256), 0, s, ws.keys_a.p, ws.flags.p, ni);
  scan_flags(s, ws, 1, n, t.d_count.p);
  hipLaunchKernelGGL(k_ndt_table_clear, dim3(div_up((long)slots, 256)), dim3(256), 0, s, t.table.p, slots);
  hipLaunchKernelGGL(k_ndt_leaves, dim3(div_up((long)n, 128)), dim3(128), 0, s, d_pts, ws.keys_a.p, ws.vals_a.p, ws.flags.p, ws.scan_local.p,
                     ws.scan_block.p, ws.totals.p, t.meta.p, t.leaf_f.p, t.leaf_d.p, t.leaf_key.p, t.table.p, ni, p.min_points_per_voxel,
                     p.min_covar_eigvalue_mult);
  MI_LAUNCH_CHECK();
  MI_HIP(hipMemcpyAsync(&t.h_meta, t.meta.p, sizeof(NdtGridMeta), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void ndt_cells(hipStream_t s, NdtTarget& t, int64_t* leaf_idx, int* leaf_n, double* mean3, double* cov9, double* icov9, int* min_b3, int* div_b3) {
  const int nl = t.h_meta.n_leaves;
  std::vector<NdtLeafD> L(nl);
  std::vector<uint32_t> K(nl);
  MI_HIP(hipMemcpyAsync(L.data(), t.leaf_d.p, nl * sizeof(NdtLeafD), hipMemcpyDeviceToHost, s));
  MI_HIP(hipMemcpyAsync(K.data(), t.leaf_key.p, nl * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < nl; i++) {
    leaf_idx[i] = (int64_t)K[i];
    leaf_n[i] = L[i].n;
    memcpy(mean3 + 3 * i, L[i].mean, 3 * sizeof(double));
    memcpy(cov9 + 9 * i, L[i].cov, 9 * sizeof(double));
    memcpy(icov9 + 9 * i, L[i].icov, 9 * sizeof(double));
  }
  for (int d = 0; d < 3; d++) { min_b3[d] = t.h_meta.min_b[d]; div_b3[d] = t.h_meta.div_b[d]; }
}

// ---- derivatives ------------------------------------------------------------------------------------------------------
struct NdtAngles {
  float jf[8][3];
  float hf[15][3];
  double jd[8][3];
  double hd[15][3];
};
struct NdtPassParams {
  float T[16];
  float d1, d2;
  double gd1, gd2;
  int compute_hessian;
};

constexpr int kNdtAcc = 28;  // score + 6 gradient + 21 Hessian (upper triangle, row-major)
constexpr int kNdtBlock = 256;

__device__ __forceinline__ int ndt_lookup(const HashSlot* table, const NdtGridMeta& m, uint32_t key) {
  uint32_t h = (key * 0x9E3779B1u) >> m.hshift;
  for (;;) {
    const int4 sl = *reinterpret_cast<const int4*>(&table[h]);
    if ((uint32_t)sl.x == key) return sl.y;
    if ((uint32_t)sl.x == kEmptyKey) return -1;
    h = (h + 1) & m.hmask;
  }
}

// getNeighborhoodAtPoint7: the voxel holding the point and its 6 face neighbours (inside the grid, usable).
__device__ __forceinline__ int ndt_neighbors7(const HashSlot* table, const NdtGridMeta& m, float x, float y, float z, int out[7]) {
  const int ijk[3] = {(int)floorf(x * m.inv_leaf), (int)floorf(y * m.inv_leaf), (int)floorf(z * m.inv_leaf)};
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const int dx = (k == 1) - (k == 2), dy = (k == 3) - (k == 4), dz = (k == 5) - (k == 6);
    const int c0 = ijk[0] + dx, c1 = ijk[1] + dy, c2 = ijk[2] + dz;
    if (c0 < m.min_b[0] || c0 > m.max_b[0] || c1 < m.min_b[1] || c1 > m.max_b[1] || c2 < m.min_b[2] || c2 > m.max_b[2]) continue;
    const uint32_t key = (uint32_t)((c0 - m.min_b[0]) + (c1 - m.min_b[1]) * m.div_b[0] + (c2 - m.min_b[2]) * m.div_b[0] * m.div_b[1]);
    const int li = ndt_lookup(table, m, key);
    if (li >= 0) out[cnt++] = li;
  }
  return cnt;
}

// computeDerivatives with ndt_omp's fp32 per-point maths (N3).
__global__ __launch_bounds__(kNdtBlock) void k_ndt_derivatives(const float4* __restrict__ src, int n, const NdtLeafF* __restrict__ leaves,
                                                               const HashSlot* __restrict__ table, const NdtGridMeta* __restrict__ meta,
                                                               NdtAngles A, NdtPassParams P, double* __restrict__ partials) {
  __shared__ double smem[4 * kNdtAcc];
  const NdtGridMeta m = *meta;
  double acc[kNdtAcc];
#pragma unroll
  for (int k = 0; k < kNdtAcc; k++) acc[k] = 0.0;
  for (int idx = blockIdx.x * kNdtBlock + threadIdx.x; idx < n; idx += gridDim.x * kNdtBlock) {
    const float4 xp = src[idx];
    const float xt0 = transform_row(P.T, xp.x, xp.y, xp.z), xt1 = transform_row(P.T + 4, xp.x, xp.y, xp.z), xt2 = transform_row(P.T + 8, xp.x, xp.y, xp.z);
    int nb[7];
    const int cnt = ndt_neighbors7(table, m, xt0, xt1, xt2, nb);
    if (cnt == 0) continue;
    const float x[3] = {xp.x, xp.y, xp.z};
    // computePointDerivatives: non-trivial entries of the 3x6 point gradient and the 3-vectors a..f of the point Hessian
    float xj[8];
#pragma unroll
    for (int a = 0; a < 8; a++) xj[a] = A.jf[a][0] * x[0] + A.jf[a][1] * x[1] + A.jf[a][2] * x[2];
    float pg[3][6] = {{1, 0, 0, 0, xj[2], xj[5]}, {0, 1, 0, xj[0], xj[3], xj[6]}, {0, 0, 1, xj[1], xj[4], xj[7]}};
    float xh[15];
#pragma unroll
    for (int a = 0; a < 15; a++) xh[a] = A.hf[a][0] * x[0] + A.hf[a][1] * x[1] + A.hf[a][2] * x[2];
    // ph(i,j) for i,j in 3..5: a=(0,xh0,xh1) b=(0,xh2,xh3) c=(0,xh4,xh5) d=xh6..8 e=xh9..11 f=xh12..14
    for (int c = 0; c < cnt; c++) {
      const NdtLeafF L = leaves[nb[c]];
      const float xtr[3] = {(float)((double)xt0 - L.mean[0]), (float)((double)xt1 - L.mean[1]), (float)((double)xt2 - L.mean[2])};
      const float* ci = L.icov;
      float xc[3];
#pragma unroll
      for (int s2 = 0; s2 < 3; s2++) xc[s2] = xtr[0] * ci[s2] + xtr[1] * ci[3 + s2] + xtr[2] * ci[6 + s2];
      float e_x_cov_x = expf(-P.d2 * (xtr[0] * xc[0] + xtr[1] * xc[1] + xtr[2] * xc[2]) * 0.5f);
      const float score_inc = -P.d1 * e_x_cov_x;
      e_x_cov_x = P.d2 * e_x_cov_x;
      if (e_x_cov_x > 1.f || e_x_cov_x < 0.f || e_x_cov_x != e_x_cov_x) continue;
      acc[0] += (double)score_inc;
      e_x_cov_x *= P.d1;
      float cg[3][6];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 6; k++) cg[r][k] = ci[3 * r] * pg[0][k] + ci[3 * r + 1] * pg[1][k] + ci[3 * r + 2] * pg[2][k];
      float xcg[6];
#pragma unroll
      for (int k = 0; k < 6; k++) xcg[k] = xtr[0] * cg[0][k] + xtr[1] * cg[1][k] + xtr[2] * cg[2][k];
#pragma unroll
      for (int k = 0; k < 6; k++) acc[1 + k] += (double)(e_x_cov_x * xcg[k]);
      if (P.compute_hessian) {
        int hk = 7;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = i; j < 6; j++) {
            float xh_ij = 0.f;
            if (i >= 3) {  // j >= i >= 3: block (i-3, j-3) of {a b c; b d e; c e f}
              const int bi = i - 3, bj = j - 3;
              const int blk = (bi == 0) ? bj : ((bi == 1) ? (bj + 2) : 5);  // a=0 b=1 c=2 d=3 e=4 f=5
              float v0, v1, v2;
              if (blk == 0) { v0 = 0.f; v1 = xh[0]; v2 = xh[1]; }
              else if (blk == 1) { v0 = 0.f; v1 = xh[2]; v2 = xh[3]; }
              else if (blk == 2) { v0 = 0.f; v1 = xh[4]; v2 = xh[5]; }
              else if (blk == 3) { v0 = xh[6]; v1 = xh[7]; v2 = xh[8]; }
              else if (blk == 4) { v0 = xh[9]; v1 = xh[10]; v2 = xh[11]; }
              else { v0 = xh[12]; v1 = xh[13]; v2 = xh[14]; }
              xh_ij = xc[0] * v0 + xc[1] * v1 + xc[2] * v2;
            }
            const float gg = pg[0][j] * cg[0][i] + pg[1][j] * cg[1][i] + pg[2][j] * cg[2][i];
            acc[hk++] += (double)(e_x_cov_x * (-P.d2 * xcg[i] * xcg[j] + xh_ij + gg));
          }
      }
    }
  }
  block_sum<kNdtAcc, 4>(acc, smem);
  if (threadIdx.x == 0)
    for (int k = 0; k < kNdtAcc; k++) partials[(size_t)blockIdx.x * 32 + k] = acc[k];
}

// computeHessian / updateHessian: fp64 per-point maths.
__global__ __launch_bounds__(kNdtBlock) void k_ndt_hessian64(const float4* __restrict__ src, int n, const NdtLeafD* __restrict__ leaves,
                                                             const HashSlot* __restrict__ table, const NdtGridMeta* __restrict__ meta, NdtAngles A,
                                                             NdtPassParams P, double* __restrict__ partials) {
  __shared__ double smem[4 * kNdtAcc];
  const NdtGridMeta m = *meta;
  double acc[kNdtAcc];
#pragma unroll
  for (int k = 0; k < kNdtAcc; k++) acc[k] = 0.0;
  for (int idx = blockIdx.x * kNdtBlock + threadIdx.x; idx < n; idx += gridDim.x * kNdtBlock) {
    const float4 xp = src[idx];
    const float xt0 = transform_row(P.T, xp.x, xp.y, xp.z), xt1 = transform_row(P.T + 4, xp.x, xp.y, xp.z), xt2 = transform_row(P.T + 8, xp.x, xp.y, xp.z);
    int nb[7];
    const int cnt = ndt_neighbors7(table, m, xt0, xt1, xt2, nb);
    if (cnt == 0) continue;
    const double x[3] = {(double)xp.x, (double)xp.y, (double)xp.z};
    double xj[8], xh[15];
#pragma unroll
    for (int a = 0; a < 8; a++) xj[a] = x[0] * A.jd[a][0] + x[1] * A.jd[a][1] + x[2] * A.jd[a][2];
#pragma unroll
    for (int a = 0; a < 15; a++) xh[a] = x[0] * A.hd[a][0] + x[1] * A.hd[a][1] + x[2] * A.hd[a][2];
    double pg[3][6] = {{1, 0, 0, 0, xj[2], xj[5]}, {0, 1, 0, xj[0], xj[3], xj[6]}, {0, 0, 1, xj[1], xj[4], xj[7]}};
    for (int c = 0; c < cnt; c++) {
      const NdtLeafD& L = leaves[nb[c]];
      const double xtr[3] = {(double)xt0 - L.mean[0], (double)xt1 - L.mean[1], (double)xt2 - L.mean[2]};
      double ci[9];
#pragma unroll
      for (int k = 0; k < 9; k++) ci[k] = L.icov[k];
      double cx[3];
#pragma unroll
      for (int r = 0; r < 3; r++) cx[r] = ci[3 * r] * xtr[0] + ci[3 * r + 1] * xtr[1] + ci[3 * r + 2] * xtr[2];
      double e_x_cov_x = P.gd2 * exp(-P.gd2 * (xtr[0] * cx[0] + xtr[1] * cx[1] + xtr[2] * cx[2]) / 2);
      if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
      e_x_cov_x *= P.gd1;
      double cg[3][6], xcg[6];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 6; k++) cg[r][k] = ci[3 * r] * pg[0][k] + ci[3 * r + 1] * pg[1][k] + ci[3 * r + 2] * pg[2][k];
#pragma unroll
      for (int k = 0; k < 6; k++) xcg[k] = xtr[0] * cg[0][k] + xtr[1] * cg[1][k] + xtr[2] * cg[2][k];
      int hk = 7;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) {
          double xh_ij = 0.0;
          if (i >= 3) {
            const int bi = i - 3, bj = j - 3;
            const int blk = (bi == 0) ? bj : ((bi == 1) ? (bj + 2) : 5);
            double v0, v1, v2;
            if (blk == 0) { v0 = 0; v1 = xh[0]; v2 = xh[1]; }
            else if (blk == 1) { v0 = 0; v1 = xh[2]; v2 = xh[3]; }
            else if (blk == 2) { v0 = 0; v1 = xh[4]; v2 = xh[5]; }
            else if (blk == 3) { v0 = xh[6]; v1 = xh[7]; v2 = xh[8]; }
            else if (blk == 4) { v0 = xh[9]; v1 = xh[10]; v2 = xh[11]; }
            else { v0 = xh[12]; v1 = xh[13]; v2 = xh[14]; }
            // x_trans . (c_inv * point_hessian block)
            xh_ij = xtr[0] * (ci[0] * v0 + ci[1] * v1 + ci[2] * v2) + xtr[1] * (ci[3] * v0 + ci[4] * v1 + ci[5] * v2) +
                    xtr[2] * (ci[6] * v0 + ci[7] * v1 + ci[8] * v2);
          }
          const double gg = pg[0][j] * cg[0][i] + pg[1][j] * cg[1][i] + pg[2][j] * cg[2][i];
          acc[hk++] += e_x_cov_x * (-P.gd2 * xcg[i] * xcg[j] + xh_ij + gg);
        }
    }
  }
  block_sum<kNdtAcc, 4>(acc, smem);
  if (threadIdx.x == 0)
    for (int k = 0; k < kNdtAcc; k++) partials[(size_t)blockIdx.x * 32 + k] = acc[k];
}

__global__ __launch_bounds__(64) void k_ndt_reduce(const double* __restrict__ partials, int G, double* __restrict__ sums) {
  const int l = threadIdx.x;
  if (l >= kNdtAcc) return;
  double s = 0;
  for (int g = 0; g < G; g++) s += partials[(size_t)g * 32 + l];
  sums[l] = s;
}

// ---- host-side control (N2, N4) -----------------------------------------------------------------------------------------
namespace {

void angle_tables(const double p[6], NdtAngles& t) {
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
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
void pose_to_matrix_f(const double p[6], float T[16]) {
  const float a = (float)p[3], b = (float)p[4], c = (float)p[5];
  const float ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
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
void euler_angles_012(const float T[16], float res[3]) {
  auto co = [&](int r, int c) { return T[4 * r + c]; };
  res[0] = std::atan2(co(1, 2), co(2, 2));
  const float c2 = std::sqrt(co(0, 0) * co(0, 0) + co(0, 1) * co(0, 1));
  if (res[0] > 0.f) {
    res[0] -= (float)M_PI;
    res[1] = std::atan2(-co(0, 2), -c2);
  } else {
    res[1] = std::atan2(-co(0, 2), c2);
  }
  const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * co(2, 0) - c1 * co(1, 0), c1 * co(1, 1) - s1 * co(2, 1));
  for (int d = 0; d < 3; d++) res[d] = -res[d];
}

struct NdtRun {
  hipStream_t s;
  NdtTarget& t;
  const float4* src;
  int n;
  double gd1, gd2;
  NdtAngles ang;  // tables of the last derivative pass (computeHessian reuses them)
  float T[16];

  void gauss(const mi355_ndt_params& p) {
    const double c1 = 10 * (1 - p.outlier_ratio);
    const double c2 = p.outlier_ratio / std::pow(p.resolution, 3);
    const double d3 = -std::log(c2);
    gd1 = -std::log(c1 + c2) - d3;
    gd2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / gd1);
  }
  void prepare() {
    t.blocks = std::max(1, std::min(1024, div_up(n, kNdtBlock)));
    t.partials.ensure((size_t)t.blocks * 32);
    t.sums.ensure(32);
    t.h_sums.ensure(32);
  }
  NdtPassParams pass(bool hess) const {
    NdtPassParams P;
    memcpy(P.T, T, sizeof(T));
    P.d1 = (float)gd1; P.d2 = (float)gd2;
    P.gd1 = gd1; P.gd2 = gd2;
    P.compute_hessian = hess ? 1 : 0;
    return P;
  }
  void fetch() {
    hipLaunchKernelGGL(k_ndt_reduce, dim3(1), dim3(64), 0, s, t.partials.p, t.blocks, t.sums.p);
    MI_LAUNCH_CHECK();
    MI_HIP(hipMemcpyAsync(t.h_sums.p, t.sums.p, kNdtAcc * sizeof(double), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
  }
  static void unpack_hessian(const double* a, double* H) {
    int k = 7;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) { H[6 * i + j] = a[k]; H[6 * j + i] = a[k]; k++; }
  }
  // computeDerivatives at pose p with transform matrix T (already set)
  double derivatives(const double p[6], double* grad, double* hess, bool compute_hessian) {
    angle_tables(p, ang);
    hipLaunchKernelGGL(k_ndt_derivatives, dim3(t.blocks), dim3(kNdtBlock), 0, s, src, n, t.leaf_f.p, t.table.p, t.meta.p, ang, pass(compute_hessian),
                       t.partials.p);
    fetch();
    for (int k = 0; k < 6; k++) grad[k] = t.h_sums.p[1 + k];
    if (compute_hessian) unpack_hessian(t.h_sums.p, hess);
    return t.h_sums.p[0];
  }
  void hessian64(double* hess) {
    hipLaunchKernelGGL(k_ndt_hessian64, dim3(t.blocks), dim3(kNdtBlock), 0, s, src, n, t.leaf_d.p, t.table.p, t.meta.p, ang, pass(true), t.partials.p);
    fetch();
    unpack_hessian(t.h_sums.p, hess);
  }
};

inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  auto cubic = [](double a0, double f0, double g0, double a1, double f1, double g1) {  // minimiser of the cubic through (a0,f0,g0),(a1,f1,g1)
    const double z = 3 * (f1 - f0) / (a1 - a0) - g1 - g0;
    const double w = std::sqrt(z * z - g1 * g0);
    return a0 + (a1 - a0) * (w - g0 - z) / (g1 - g0 + 2 * w);
  };
  if (f_t > f_l) {
    const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) ? a_c : a_s;
  }
  if (std::fabs(g_t) <= std::fabs(g_l)) {
    const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double nxt = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    return (a_t > a_l) ? std::min(a_t + 0.66 * (a_u - a_t), nxt) : std::max(a_t + 0.66 * (a_u - a_t), nxt);
  }
  return cubic(a_u, f_u, g_u, a_t, f_t, g_t);
}

}  // namespace

void ndt_derivatives(hipStream_t s, Workspace&, NdtTarget& t, const float4* d_src, size_t n, const mi355_ndt_params& p, const double* p6, double* score,
                     double* grad6, double* hess36, bool hessian_fp64) {
  NdtRun run{s, t, d_src, (int)n};
  run.gauss(p);
  run.prepare();
  pose_to_matrix_f(p6, run.T);
  *score = run.derivatives(p6, grad6, hess36, true);
  if (hessian_fp64) run.hessian64(hess36);
}

void ndt_align(hipStream_t s, Workspace&, NdtTarget& t, const float4* d_src, size_t n, const mi355_ndt_params& p, const float* guess16,
               mi355_result* out) {
  NdtRun run{s, t, d_src, (int)n};
  run.gauss(p);
  run.prepare();
  memcpy(run.T, guess16, sizeof(run.T));  // transformPointCloud(output, output, guess): the first pass uses the guess matrix itself
  int nr_iterations = 0;
  bool converged = false;
  double pv[6], delta_p[6], grad[6], hess[36];
  float eul[3];
  euler_angles_012(guess16, eul);
  pv[0] = guess16[3]; pv[1] = guess16[7]; pv[2] = guess16[11];
  pv[3] = eul[0]; pv[4] = eul[1]; pv[5] = eul[2];
  float final_T[16];
  memcpy(final_T, guess16, sizeof(final_T));
  double score = run.derivatives(pv, grad, hess, true);
  double trans_probability = 0;
  bool early = false;
  while (!converged) {
    double ng[6];
    for (int k = 0; k < 6; k++) ng[k] = -grad[k];
    svd6_solve_sym(hess, ng, delta_p);
    double nrm = 0;
    for (int k = 0; k < 6; k++) nrm += delta_p[k] * delta_p[k];
    nrm = std::sqrt(nrm);
    if (nrm == 0 || nrm != nrm) {
      trans_probability = score / (double)n;
      converged = nrm == nrm;
      early = true;
      break;
    }
    for (int k = 0; k < 6; k++) delta_p[k] /= nrm;
    // ---- computeStepLengthMT ----
    double* dir = delta_p;
    const double step_init = nrm, step_max = p.step_size, step_min = p.transformation_epsilon / 2;
    const double phi_0 = -score;
    double d_phi_0 = 0;
    for (int k = 0; k < 6; k++) d_phi_0 -= grad[k] * dir[k];
    double a_t = 0;
    bool do_search = true;
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) do_search = false;
      else {
        d_phi_0 *= -1;
        for (int k = 0; k < 6; k++) dir[k] *= -1;
      }
    }
    if (do_search) {
      const int max_step_iterations = 10;
      int step_iterations = 0;
      const double mu = 1.e-4, nu = 0.9;
      double a_l = 0, a_u = 0;
      double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
      double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu), g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
      bool interval_converged = p.legacy_mt_skip ? ((step_max - step_min) > 0) : ((step_max - step_min) < 0);
      bool open_interval = true;
      a_t = std::max(std::min(step_init, step_max), step_min);
      double x_t[6];
      for (int k = 0; k < 6; k++) x_t[k] = pv[k] + dir[k] * a_t;
      pose_to_matrix_f(x_t, run.T);
      memcpy(final_T, run.T, sizeof(final_T));
      score = run.derivatives(x_t, grad, hess, true);
      double phi_t = -score, d_phi_t = 0;
      for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * dir[k];
      double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
      while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        a_t = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t) : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = std::max(std::min(a_t, step_max), step_min);
        for (int k = 0; k < 6; k++) x_t[k] = pv[k] + dir[k] * a_t;
        pose_to_matrix_f(x_t, run.T);
        memcpy(final_T, run.T, sizeof(final_T));
        score = run.derivatives(x_t, grad, hess, false);
        phi_t = -score;
        d_phi_t = 0;
        for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * dir[k];
        psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
        d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
          open_interval = false;
          f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
          f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
        }
        interval_converged = open_interval ? update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                                           : update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        step_iterations++;
      }
      if (step_iterations) run.hessian64(hess);
    }
    const double step = do_search ? a_t : 0.0;
    for (int k = 0; k < 6; k++) { delta_p[k] *= step; pv[k] += delta_p[k]; }
    if (nr_iterations > p.max_iterations || (nr_iterations && (std::fabs(step) < p.transformation_epsilon))) converged = true;
    nr_iterations++;
  }
  if (!early) trans_probability = score / (double)n;
  memset(out, 0, sizeof(*out));
  memcpy(out->T, final_T, sizeof(final_T));
  out->converged = converged ? 1 : 0;
  out->iterations = nr_iterations;
  out->error = score;
  memcpy(out->H, hess, sizeof(hess));
  out->trans_probability = trans_probability;
}

}  // namespace mi355
