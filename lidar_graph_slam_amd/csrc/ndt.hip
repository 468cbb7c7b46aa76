// NDT on gfx950: pclomp::NormalDistributionsTransform (DIRECT7) + pclomp::VoxelGridCovariance — SURVEY.md §8a rows N1-N5.
//
//   target build  : PCL voxel index per point -> batched radix sort -> one thread per occupied voxel accumulates
//                   sum / sum of outer products in fp64 (ascending point index), eigen-inflates (eq. 6.11 [Magnusson 2009])
//                   and inverts; usable voxels (>= min_points_per_voxel, valid eigenvalues) go into an open-addressing
//                   hash keyed by the linear voxel index, so DIRECT7 is 7 hash probes — no kd-tree.
//   derivatives   : one thread per source point: fp32 transform, 7-voxel gather, score + gradient(6) + Hessian(21 unique)
//                   with ndt_omp's fp32 per-point maths, accumulated in fp64 registers and reduced by a fixed-order
//                   wave/LDS tree (one partial per block).  Algorithmic bytes per point: 16 + 7 x 36.
//   control        : the Newton iteration + More-Thuente line search is the resumable state machine of ndt_mt.hpp.  One pair:
//                   the host advances it (one launch + 224-byte read-back per evaluation).  A batch of pairs: every pair's
//                   state lives in HBM and k_ndt_batch_advance advances all of them — no read-back between evaluations.
#include <cmath>

#include "engine.hpp"
#include "ndt_mt.hpp"
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
  DevBuf<int> done;  // blocks of the current request that have finished (ndt_last_block_reduce re-arms it)
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
  // Every lane fetches the point at its sorted position (one gather per lane, all in flight together); the lane at the head of a
  // voxel's run then adds the points of the lanes behind it IN ORDER, handed over by lane shuffles, and finishes from memory
  // whatever of its run lies beyond the wave (a head walking its whole run from memory — 30+ dependent gathers per voxel at
  // 2 m cells — took 0.57 ms per 3 x 10^5-point target).
  const int lane = lane_id();
  const bool valid = i < n;
  const bool head = valid && flags[i] != 0;
  const uint32_t key = valid ? keys[i] : 0xFFFFFFFEu;
  float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) mine = pts[vals[i]];
  double sx = 0, sy = 0, sz = 0;
  Sym3<double> sq = {0, 0, 0, 0, 0, 0};
  auto add = [&](float fx, float fy, float fz) {
    const double x = (double)fx, y = (double)fy, z = (double)fz;
    sx += x; sy += y; sz += z;
    sq.xx += x * x; sq.xy += x * y; sq.xz += x * z; sq.yy += y * y; sq.yz += y * z; sq.zz += z * z;
  };
  if (head) add(mine.x, mine.y, mine.z);
  int cnt = head ? 1 : 0;
  bool open = head;
  for (int d = 1; d < kWave; d++) {
    const uint32_t kk = __shfl_down(key, d, kWave);
    const float px = __shfl_down(mine.x, d, kWave), py = __shfl_down(mine.y, d, kWave), pz = __shfl_down(mine.z, d, kWave);
    const bool in_wave = lane + d < kWave;
    const bool take = open && in_wave && kk == key;
    if (take) { add(px, py, pz); cnt++; }
    open = take || (open && !in_wave);
    if (!__any(open && in_wave)) break;
  }
  if (!head) return;
  if (open) {  // the run continues past this wave
    int j = i + (kWave - lane);
    for (; j < n && keys[j] == key; j++) {
      const float4 p = pts[vals[j]];
      add(p.x, p.y, p.z);
    }
    cnt = j - i;
  }
  const int li = local[i] + blockoff[i / kScanTile];
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
  if (n == 0) {  // an empty target (an unused slot of a batch): no voxel exists, every lookup falls outside the grid
    t.leaf_f.ensure(1); t.leaf_d.ensure(1); t.leaf_key.ensure(1);
    t.slots = 16;
    t.table.ensure(t.slots);
    t.meta.ensure(1);
    NdtGridMeta m{};
    for (int d = 0; d < 3; d++) { m.min_b[d] = 0; m.max_b[d] = -1; m.div_b[d] = 0; }
    m.inv_leaf = 1.0f / (float)p.resolution;
    m.hshift = 28; m.hmask = 15; m.n_leaves = 0;
    t.h_meta = m;
    hipLaunchKernelGGL(k_ndt_table_clear, dim3(1), dim3(256), 0, s, t.table.p, t.slots);
    MI_HIP(hipMemcpyAsync(t.meta.p, &t.h_meta, sizeof(NdtGridMeta), hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
    return;
  }
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
// One pair: the block that finishes LAST adds the block partials, in block order (the order the batch path's
// k_ndt_batch_advance uses), and re-arms the counter — one launch and one 224-byte read-back per request instead of two launches.
__device__ __forceinline__ void ndt_last_block_reduce(const double* partials, int* done, double* __restrict__ sums) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    __threadfence();  // this block's partials are visible before the count says so
    is_last = atomicAdd(done, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  const volatile double* p = partials;
  if (threadIdx.x < kNdtAcc) {
    double sacc = 0;
    for (int gidx = 0; gidx < (int)gridDim.x; gidx++) sacc += p[(size_t)gidx * 32 + threadIdx.x];
    sums[threadIdx.x] = sacc;
  }
  if (threadIdx.x == 0) *done = 0;
}

// Block `block` of `nblocks` strides over the n source points; its 28 sums go to partials_row[0..27].  The single-pair
// kernel and the batch kernel (pair state in HBM) share this body, so a pair gets the same sums either way.
__device__ __forceinline__ void ndt_derivatives_body(const float4* __restrict__ src, int n, const NdtLeafF* __restrict__ leaves,
                                                     const HashSlot* __restrict__ table, const NdtGridMeta& m, const NdtAngles& A,
                                                     const NdtPassParams& P, int block, int nblocks, double* __restrict__ partials_row,
                                                     double* smem) {
  double acc[kNdtAcc];
#pragma unroll
  for (int k = 0; k < kNdtAcc; k++) acc[k] = 0.0;
  for (int idx = block * kNdtBlock + threadIdx.x; idx < n; idx += nblocks * kNdtBlock) {
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
    for (int k = 0; k < kNdtAcc; k++) partials_row[k] = acc[k];
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_derivatives(const float4* __restrict__ src, int n, const NdtLeafF* __restrict__ leaves,
                                                               const HashSlot* __restrict__ table, const NdtGridMeta* __restrict__ meta,
                                                               NdtAngles A, NdtPassParams P, double* __restrict__ partials,
                                                               int* __restrict__ done, double* __restrict__ sums) {
  __shared__ double smem[4 * kNdtAcc];
  const NdtGridMeta m = *meta;
  ndt_derivatives_body(src, n, leaves, table, m, A, P, blockIdx.x, gridDim.x, partials + (size_t)blockIdx.x * 32, smem);
  ndt_last_block_reduce(partials, done, sums);
}

// computeHessian / updateHessian: fp64 per-point maths.
__device__ __forceinline__ void ndt_hessian64_body(const float4* __restrict__ src, int n, const NdtLeafD* __restrict__ leaves,
                                                   const HashSlot* __restrict__ table, const NdtGridMeta& m, const NdtAngles& A,
                                                   const NdtPassParams& P, int block, int nblocks, double* __restrict__ partials_row, double* smem) {
  double acc[kNdtAcc];
#pragma unroll
  for (int k = 0; k < kNdtAcc; k++) acc[k] = 0.0;
  for (int idx = block * kNdtBlock + threadIdx.x; idx < n; idx += nblocks * kNdtBlock) {
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
    for (int k = 0; k < kNdtAcc; k++) partials_row[k] = acc[k];
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_hessian64(const float4* __restrict__ src, int n, const NdtLeafD* __restrict__ leaves,
                                                             const HashSlot* __restrict__ table, const NdtGridMeta* __restrict__ meta, NdtAngles A,
                                                             NdtPassParams P, double* __restrict__ partials, int* __restrict__ done,
                                                             double* __restrict__ sums) {
  __shared__ double smem[4 * kNdtAcc];
  const NdtGridMeta m = *meta;
  ndt_hessian64_body(src, n, leaves, table, m, A, P, blockIdx.x, gridDim.x, partials + (size_t)blockIdx.x * 32, smem);
  ndt_last_block_reduce(partials, done, sums);
}



// ---- executors of the ndt_mt.hpp state machine (N2, N4) ---------------------------------------------------------------------
namespace {

inline int ndt_blocks(int n) { return std::max(1, std::min(1024, div_up(n, kNdtBlock))); }

inline NdtPassParams pass_params(const float* T, const NdtConsts& c, bool hess) {
  NdtPassParams P;
  memcpy(P.T, T, sizeof(P.T));
  P.d1 = (float)c.gd1; P.d2 = (float)c.gd2;
  P.gd1 = c.gd1; P.gd2 = c.gd2;
  P.compute_hessian = hess ? 1 : 0;
  return P;
}

// one pair: answer a request with one launch and a 224-byte read-back
struct NdtRun {
  hipStream_t s;
  NdtTarget& t;
  const float4* src;
  int n;
  NdtConsts c;

  void prepare() {
    t.blocks = ndt_blocks(n);
    t.partials.ensure((size_t)t.blocks * 32);
    t.sums.ensure(32);
    t.h_sums.ensure(32);
    if (!t.done.p) {
      t.done.ensure(1);
      MI_HIP(hipMemsetAsync(t.done.p, 0, sizeof(int), s));
    }
  }
  const double* answer(int request, const float* T, const NdtAngles& ang) {
    const NdtPassParams P = pass_params(T, c, request == NDT_REQ_DERIV_HESS);
    if (request == NDT_REQ_HESS64)
      hipLaunchKernelGGL(k_ndt_hessian64, dim3(t.blocks), dim3(kNdtBlock), 0, s, src, n, t.leaf_d.p, t.table.p, t.meta.p, ang, P, t.partials.p,
                         t.done.p, t.sums.p);
    else
      hipLaunchKernelGGL(k_ndt_derivatives, dim3(t.blocks), dim3(kNdtBlock), 0, s, src, n, t.leaf_f.p, t.table.p, t.meta.p, ang, P, t.partials.p,
                         t.done.p, t.sums.p);
    MI_LAUNCH_CHECK();
    MI_HIP(hipMemcpyAsync(t.h_sums.p, t.sums.p, kNdtAcc * sizeof(double), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    return t.h_sums.p;
  }
};

void state_to_result(const NdtPairState& st, mi355_result* out) {
  memset(out, 0, sizeof(*out));
  memcpy(out->T, st.T, sizeof(st.T));
  out->converged = st.converged ? 1 : 0;
  out->iterations = st.nr_iterations;
  out->error = st.score;
  memcpy(out->H, st.hess, sizeof(st.hess));
  out->trans_probability = st.trans_probability;
  out->n_source = st.n;
}

}  // namespace

void ndt_derivatives(hipStream_t s, Workspace&, NdtTarget& t, const float4* d_src, size_t n, const mi355_ndt_params& p, const double* p6, double* score,
                     double* grad6, double* hess36, bool hessian_fp64) {
  NdtRun run{s, t, d_src, (int)n, ndt_consts(p)};
  run.prepare();
  float T[16];
  NdtAngles ang;
  ndt_pose_to_matrix_f(p6, T);
  ndt_angle_tables(p6, ang);
  const double* sums = run.answer(NDT_REQ_DERIV_HESS, T, ang);
  *score = sums[0];
  for (int k = 0; k < 6; k++) grad6[k] = sums[1 + k];
  ndt_unpack_hessian(sums, hess36);
  if (hessian_fp64) ndt_unpack_hessian(run.answer(NDT_REQ_HESS64, T, ang), hess36);
}

void ndt_align(hipStream_t s, Workspace&, NdtTarget& t, const float4* d_src, size_t n, const mi355_ndt_params& p, const float* guess16,
               mi355_result* out) {
  NdtRun run{s, t, d_src, (int)n, ndt_consts(p)};
  run.prepare();
  NdtPairState st;
  ndt_begin(st, guess16, (int)n);
  while (st.request != NDT_REQ_NONE) ndt_advance(st, run.answer(st.request, st.T, st.ang), run.c);
  state_to_result(st, out);
}

// ---- batch: P pairs, every pair's More-Thuente state in HBM -------------------------------------------------------------
struct NdtTargetRef {
  const NdtLeafF* leaf_f;
  const NdtLeafD* leaf_d;
  const HashSlot* table;
  const NdtGridMeta* meta;
};

struct NdtBatch {
  int P = 0, U = 0;
  std::vector<NdtTarget*> targets;
  std::vector<NdtTargetRef> h_refs;
  DevBuf<NdtTargetRef> refs;
  DevBuf<NdtPairState> states;
  DevBuf<int> tgt_of_src, counts, n_active;
  DevBuf<float> guesses;
  DevBuf<double> partials;
  PinnedBuf<int> h_active;
  std::vector<NdtPairState> h_states;
  std::vector<int> h_map, h_counts;
  std::vector<float> h_guess;
  int rounds = 0;  // evaluation rounds of the last align (diagnostics)
  ~NdtBatch() {
    for (auto* t : targets) ndt_target_destroy(t);
  }
};

NdtBatch* ndt_batch_create(int P, int U) {
  auto* b = new NdtBatch();
  b->P = P;
  b->U = U;
  b->targets.assign((size_t)U, nullptr);
  for (auto& t : b->targets) t = ndt_target_create();
  b->h_refs.resize((size_t)U);
  b->refs.ensure((size_t)U);
  b->states.ensure((size_t)P);
  b->tgt_of_src.ensure((size_t)P);
  b->counts.ensure((size_t)P);
  b->guesses.ensure((size_t)P * 16);
  b->n_active.ensure(1);
  b->h_active.ensure(1);
  return b;
}
void ndt_batch_destroy(NdtBatch* b) { delete b; }
void ndt_batch_set_live(NdtBatch& b, int P_live, int U_live) {  // arrays are pair-major / target-major: a prefix of the allocation
  b.P = P_live;
  b.U = U_live;
}
size_t ndt_batch_num_cells(NdtBatch& b, int u) { return ndt_num_cells(*b.targets[(size_t)u]); }
int ndt_batch_rounds(const NdtBatch& b) { return b.rounds; }

void ndt_batch_set_target(hipStream_t s, Workspace& ws, NdtBatch& b, int u, const float4* d_pts, size_t n, const mi355_ndt_params& p) {
  NdtTarget& t = *b.targets[(size_t)u];
  ndt_build_target(s, ws, t, d_pts, n, p);
  b.h_refs[(size_t)u] = NdtTargetRef{t.leaf_f.p, t.leaf_d.p, t.table.p, t.meta.p};
}

__global__ void k_ndt_batch_begin(NdtPairState* __restrict__ states, const float* __restrict__ guesses, const int* __restrict__ counts, int P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float g[16];
  for (int k = 0; k < 16; k++) g[k] = guesses[16 * p + k];
  ndt_begin(states[p], g, counts[p]);
}

__device__ __forceinline__ int ndt_blocks_dev(int n) { return max(1, min(1024, (n + kNdtBlock - 1) / kNdtBlock)); }

// computeDerivatives of every pair that asked for it (blocks of the other pairs leave at once).  grid = xcd_grid(G, P): the
// blocks of one pair share an XCD, so its target's leaves and hash table are fetched into ONE L2.
__global__ __launch_bounds__(kNdtBlock) void k_ndt_batch_derivatives(const float4* __restrict__ src, size_t src_cap,
                                                                     const NdtTargetRef* __restrict__ targets, const int* __restrict__ tgt_of_src,
                                                                     const NdtPairState* __restrict__ states, NdtConsts c, int P, int G,
                                                                     double* __restrict__ partials) {
  __shared__ double smem[4 * kNdtAcc];
  int pair, bx;
  if (!xcd_item_chunk(G, P, pair, bx)) return;
  const NdtPairState& st = states[pair];
  const int req = st.request;
  if (req != NDT_REQ_DERIV_HESS && req != NDT_REQ_DERIV) return;
  const int n = st.n, nb = ndt_blocks_dev(n);
  if (bx >= nb) return;
  NdtPassParams Pp;
  for (int k = 0; k < 16; k++) Pp.T[k] = st.T[k];
  Pp.d1 = (float)c.gd1; Pp.d2 = (float)c.gd2;
  Pp.gd1 = c.gd1; Pp.gd2 = c.gd2;
  Pp.compute_hessian = req == NDT_REQ_DERIV_HESS;
  const NdtTargetRef tr = targets[tgt_of_src[pair]];
  const NdtGridMeta m = *tr.meta;
  ndt_derivatives_body(src + (size_t)pair * src_cap, n, tr.leaf_f, tr.table, m, st.ang, Pp, bx, nb, partials + ((size_t)pair * G + bx) * 32, smem);
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_batch_hessian64(const float4* __restrict__ src, size_t src_cap,
                                                                   const NdtTargetRef* __restrict__ targets, const int* __restrict__ tgt_of_src,
                                                                   const NdtPairState* __restrict__ states, NdtConsts c, int P, int G,
                                                                   double* __restrict__ partials) {
  __shared__ double smem[4 * kNdtAcc];
  int pair, bx;
  if (!xcd_item_chunk(G, P, pair, bx)) return;
  const NdtPairState& st = states[pair];
  if (st.request != NDT_REQ_HESS64) return;
  const int n = st.n, nb = ndt_blocks_dev(n);
  if (bx >= nb) return;
  NdtPassParams Pp;
  for (int k = 0; k < 16; k++) Pp.T[k] = st.T[k];
  Pp.d1 = (float)c.gd1; Pp.d2 = (float)c.gd2;
  Pp.gd1 = c.gd1; Pp.gd2 = c.gd2;
  Pp.compute_hessian = 1;
  const NdtTargetRef tr = targets[tgt_of_src[pair]];
  const NdtGridMeta m = *tr.meta;
  ndt_hessian64_body(src + (size_t)pair * src_cap, n, tr.leaf_d, tr.table, m, st.ang, Pp, bx, nb, partials + ((size_t)pair * G + bx) * 32, smem);
}

// One wave per pair: lanes 0..27 add the pair's block partials in block order (the single path's order), lane 0 advances the
// pair's state machine to its next request.  n_active (optional) counts the pairs still asking.
__global__ __launch_bounds__(64) void k_ndt_batch_advance(NdtPairState* __restrict__ states, const double* __restrict__ partials, NdtConsts c, int P,
                                                          int G, int* __restrict__ n_active) {
  __shared__ double sums[32];
  const int pair = blockIdx.x, lane = threadIdx.x;
  NdtPairState& st = states[pair];
  if (st.request == NDT_REQ_NONE) return;
  const int nb = ndt_blocks_dev(st.n);
  if (lane < kNdtAcc) {
    double s = 0;
    for (int g = 0; g < nb; g++) s += partials[((size_t)pair * G + g) * 32 + lane];
    sums[lane] = s;
  }
  __syncthreads();
  if (lane == 0) {
    ndt_advance(st, sums, c);
    if (n_active && st.request != NDT_REQ_NONE) atomicAdd(n_active, 1);
  }
}

// src: [P][src_cap] points (any order), h_counts: points per source.  Blocks until every pair is done (the host polls the
// number of unfinished pairs every few rounds); results stay in b.h_states.
void ndt_batch_align(hipStream_t s, NdtBatch& b, const float4* d_src, size_t src_cap, const int* h_counts, const int* tgt_of_src_host,
                     const float* guesses_host, const mi355_ndt_params& p) {
  const int P = b.P;
  const NdtConsts c = ndt_consts(p);
  b.h_map.resize((size_t)P);
  b.h_counts.assign(h_counts, h_counts + P);
  b.h_guess.resize((size_t)P * 16);
  int G = 1;
  for (int i = 0; i < P; i++) {
    const int u = tgt_of_src_host ? tgt_of_src_host[i] : i;
    if (u < 0 || u >= b.U) throw ArgError(MI355_EINVAL, "tgt_of_src entry out of range");
    b.h_map[(size_t)i] = u;
    G = std::max(G, ndt_blocks(h_counts[i]));
    for (int k = 0; k < 16; k++) b.h_guess[(size_t)i * 16 + k] = guesses_host ? guesses_host[(size_t)i * 16 + k] : ((k % 5 == 0) ? 1.f : 0.f);
  }
  b.partials.ensure((size_t)P * G * 32);
  MI_HIP(hipMemcpyAsync(b.refs.p, b.h_refs.data(), (size_t)b.U * sizeof(NdtTargetRef), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(b.tgt_of_src.p, b.h_map.data(), (size_t)P * sizeof(int), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(b.counts.p, b.h_counts.data(), (size_t)P * sizeof(int), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(b.guesses.p, b.h_guess.data(), (size_t)P * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_ndt_batch_begin, dim3(div_up(P, 64)), dim3(64), 0, s, b.states.p, b.guesses.p, b.counts.p, P);
  // every request is one round; a Newton iteration is 1 round with the legacy line search (the default), up to 12 without
  const int per_iter = p.legacy_mt_skip ? 1 : (kNdtMaxStepIterations + 2);
  const long max_rounds = 1 + (long)(p.max_iterations + 2) * per_iter;
  const int chunk = 4;
  const dim3 grid(xcd_grid(G, P));
  b.rounds = 0;
  for (long r = 0; r < max_rounds;) {
    MI_HIP(hipMemsetAsync(b.n_active.p, 0, sizeof(int), s));
    for (int k = 0; k < chunk; k++, r++) {
      hipLaunchKernelGGL(k_ndt_batch_derivatives, grid, dim3(kNdtBlock), 0, s, d_src, src_cap, b.refs.p, b.tgt_of_src.p, b.states.p, c, P, G,
                         b.partials.p);
      if (!p.legacy_mt_skip)  // computeHessian is only ever requested after a refined line search
        hipLaunchKernelGGL(k_ndt_batch_hessian64, grid, dim3(kNdtBlock), 0, s, d_src, src_cap, b.refs.p, b.tgt_of_src.p, b.states.p, c, P, G,
                           b.partials.p);
      hipLaunchKernelGGL(k_ndt_batch_advance, dim3(P), dim3(64), 0, s, b.states.p, b.partials.p, c, P, G, (k == chunk - 1) ? b.n_active.p : nullptr);
    }
    MI_LAUNCH_CHECK();
    b.rounds = (int)r;
    MI_HIP(hipMemcpyAsync(b.h_active.p, b.n_active.p, sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (*b.h_active.p == 0) break;
  }
  b.h_states.resize((size_t)P);
  MI_HIP(hipMemcpyAsync(b.h_states.data(), b.states.p, (size_t)P * sizeof(NdtPairState), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void ndt_batch_results(NdtBatch& b, mi355_result* out, float* T16_all) {
  for (int i = 0; i < b.P; i++) {
    const NdtPairState& st = b.h_states[(size_t)i];
    if (out) {
      state_to_result(st, &out[i]);
      // a pair cut off by the round budget (it cannot happen within max_iterations) reports what it has, not converged
      if (st.request != NDT_REQ_NONE) out[i].converged = 0;
    }
    if (T16_all) memcpy(T16_all + (size_t)i * 16, st.T, 16 * sizeof(float));
  }
}

}  // namespace mi355
