// NDT on gfx950: pclomp::NormalDistributionsTransform (DIRECT7) + pclomp::VoxelGridCovariance — SURVEY.md §8a rows N1-N5.
//
//   target build  : BATCHED over U targets, one launch sequence, no host synchronisation (N1): PCL voxel index per point -> batched
//                   radix sort -> the lane at the head of a voxel's run accumulates sum / sum of outer products in fp64 (ascending point
//                   index), eigen-inflates (eq. 6.11 [Magnusson 2009]) and inverts; usable voxels (>= min_points_per_voxel, valid
//                   eigenvalues) go into an open-addressing hash of 8-byte slots (voxel index -> leaf) sized ON THE DEVICE to 4 x the
//                   occupied voxels, so DIRECT7 is 7 hash probes — no kd-tree.  A registration object's target is a batch of one.
//   leaf          : 48 B — fp32 mean RELATIVE TO THE VOXEL CENTRE (1e-7 m resolution there; an absolute fp32 mean would lose 8 µm at
//                   100 m) + the 6 unique terms of the inverse covariance.  The fp64 leaf (computeHessian, parity read-back) is kept
//                   only where it is asked for.
//   derivatives   : one thread per source point (N3).  The sums of computeDerivatives factor: with q = C^-1 (x' - mu) and
//                   w = d1 d2 exp(-d2/2 (x' - mu).q),  g_i = (SUM_cells w q) . J_i  and
//                   H_ij = J_i^T (SUM_cells w (C^-1 - d2 q q^T)) J_j + (SUM_cells w q) . H_ij^pt — the point Jacobian J and the
//                   point Hessian depend on the point only, so the 7 cells cost ~45 fp32 instructions each (3 + 6 running sums) and
//                   the 6 x 6 contraction is formed ONCE per point (upstream: per cell) before it is added to 28 fp64 accumulators,
//                   reduced by a fixed-order wave/LDS tree (one partial per block).  Algorithmic bytes per point: 16 + 7 x 36.
//   control       : the Newton iteration + More-Thuente line search is the resumable state machine of ndt_mt.hpp, advanced ON THE
//                   DEVICE for every pair (k_ndt_advance): nothing is read back between evaluations, the host polls one counter every
//                   few rounds.  One pair is a batch of one (round 4: 224-byte read-back + stream synchronisation per evaluation).
#include <cmath>

#include "engine.hpp"
#include "ndt_mt.hpp"
#include "primitives.hpp"

// compile-time switches of the derivative kernel (A/B builds: benchmarks/build_variant.sh)
#ifndef NDT_DERIV_WAVES
#define NDT_DERIV_WAVES 3   // waves per SIMD the derivative kernel is compiled for: 150 VGPRs, no scratch (4: 128 VGPRs + 36 spilled, +5 %)
#endif
#ifndef NDT_STAGED
#define NDT_STAGED 0        // 1 = the LDS image of ndt_derivatives_body (measured 24 % SLOWER than per-lane loads: profiles/r05_ndt_variants.txt)
#endif
#ifndef NDT_BLOCK
#define NDT_BLOCK 256       // threads per block of the derivative kernels
#endif

namespace mi355 {

struct NdtLeaf {      // what the fp32 derivative kernel reads: 48 B, one per occupied voxel (three dwordx4)
  float m[3];         // mean - centre of the voxel
  float ic[6];        // inverse covariance xx xy xz yy yz zz
  float pad[3];
};
static_assert(sizeof(NdtLeaf) == 48, "NdtLeaf is three 16-byte loads");
struct NdtLeafD {     // fp64 view for computeHessian and for parity read-back
  double mean[3];
  double cov[9];
  double icov[9];
  int n;              // nr_points (-1 = rejected)
  int pad;
};
struct NdtSlot {      // open addressing, key = linear voxel index (PCL's), leaf = index into the target's leaf arrays
  uint32_t key;
  int leaf;
};

struct NdtGridMeta {
  int min_b[3], max_b[3], div_b[3];
  float inv_leaf, leaf;
  uint32_t hshift, hmask;
  int n_leaves;       // occupied voxels (usable or not)
  int n_pts;
};

struct NdtTargetRef {  // one target inside the compact arrays of NdtTargets
  const NdtLeaf* leaf;
  const NdtLeafD* leaf_d;   // nullptr unless the fp64 leaves are kept
  const NdtSlot* table;
  const NdtGridMeta* meta;
};

// Sum of a voxel's points: sum p (3), sum pp^T (6) and the count, fp64.
struct NdtSums {
  double v[9];
  int cnt, head;   // head: sorted position of the piece's first element
};

struct NdtTargets {
  int U = 0;
  bool has_d = false;
  DevBuf<NdtLeaf> leaf;
  DevBuf<NdtLeafD> leaf_d;
  DevBuf<uint32_t> leaf_key;
  DevBuf<NdtSlot> table;
  DevBuf<NdtGridMeta> meta;
  DevBuf<NdtTargetRef> refs;
  DevBuf<NdtSums> acc, carry;        // per-leaf sums of the run's head piece / per-wave sums carried into a wave (build scratch)
  DevBuf<long long> base;            // [U][3] leaf / slot / carry offset of target u
  DevBuf<int> slots_alloc;           // [U]
  std::vector<long long> h_base;
  std::vector<int> h_slots;
  std::vector<NdtTargetRef> h_refs;
  std::vector<NdtGridMeta> h_meta;   // read back on demand (ndt_num_cells / ndt_cells)
  bool h_meta_valid = false;
};

NdtTargets* ndt_targets_create() { return new NdtTargets(); }
bool ndt_targets_have_fp64_leaves(const NdtTargets& t) { return t.has_d; }
void ndt_targets_destroy(NdtTargets* t) { delete t; }

static void ndt_read_meta(hipStream_t s, NdtTargets& t) {
  if (t.h_meta_valid) return;
  t.h_meta.resize((size_t)t.U);
  MI_HIP(hipMemcpyAsync(t.h_meta.data(), t.meta.p, (size_t)t.U * sizeof(NdtGridMeta), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
  t.h_meta_valid = true;
}
size_t ndt_num_cells(hipStream_t s, NdtTargets& t, int u) {
  ndt_read_meta(s, t);
  return (size_t)t.h_meta[(size_t)u].n_leaves;
}

// ---- target build ---------------------------------------------------------------------------------------------------
__global__ void k_ndt_bounds_init(uint32_t* b, int U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < U * 6) b[i] = ((i % 6) < 3) ? 0xFFFFFFFFu : 0u;
}
__global__ __launch_bounds__(256) void k_ndt_bounds(const float4* __restrict__ pts, size_t pitch, const int* __restrict__ counts,
                                                    uint32_t* __restrict__ bounds) {
  const int c = blockIdx.y;
  const int n = counts[c];
  const float4* p = pts + (size_t)c * pitch;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  constexpr int kU = 4;  // independent loads in flight per thread
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += kU * stride) {
    float4 v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const int i = i0 + u * stride;
      v[u] = (i < n) ? p[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {  // (fminf / fmaxf skip the NaN padding)
      mn[0] = fminf(mn[0], v[u].x); mn[1] = fminf(mn[1], v[u].y); mn[2] = fminf(mn[2], v[u].z);
      mx[0] = fmaxf(mx[0], v[u].x); mx[1] = fmaxf(mx[1], v[u].y); mx[2] = fmaxf(mx[2], v[u].z);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; d++)
    for (int off = 32; off > 0; off >>= 1) {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], off, kWave));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off, kWave));
    }
  if (lane_id() == 0 && mn[0] <= mx[0])
    for (int d = 0; d < 3; d++) {
      atomicMin(&bounds[c * 6 + d], float_to_ordered(mn[d]));
      atomicMax(&bounds[c * 6 + 3 + d], float_to_ordered(mx[d]));
    }
}

// one thread per target: grid geometry from the bounds, radix passes its voxel indices need
__global__ void k_ndt_meta(const uint32_t* __restrict__ bounds, const int* __restrict__ counts, NdtGridMeta* __restrict__ meta,
                           int* __restrict__ sort_passes, float resolution, int U) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= U) return;
  NdtGridMeta m;
  const float inv = 1.0f / resolution;
  const int n = counts[c];
  const bool empty = n <= 0 || !(ordered_to_float(bounds[c * 6]) <= ordered_to_float(bounds[c * 6 + 3]));
  for (int d = 0; d < 3; d++) {
    const float mn = ordered_to_float(bounds[c * 6 + d]), mx = ordered_to_float(bounds[c * 6 + 3 + d]);
    m.min_b[d] = empty ? 0 : (int)floorf(mn * inv);
    m.max_b[d] = empty ? -1 : (int)floorf(mx * inv);   // an empty target: every lookup falls outside the grid
    m.div_b[d] = m.max_b[d] - m.min_b[d] + 1;
  }
  m.inv_leaf = inv;
  m.leaf = resolution;
  m.hshift = 28;
  m.hmask = 15;
  m.n_leaves = 0;
  m.n_pts = max(n, 0);
  meta[c] = m;
  const unsigned long long cells = (unsigned long long)max(m.div_b[0], 1) * (unsigned long long)max(m.div_b[1], 1) * (unsigned long long)max(m.div_b[2], 1);
  int p = 1;
  while (p < 4 && cells > ((1ull << (8 * p)) - 1ull)) p++;
  sort_passes[c] = p;
}

__global__ __launch_bounds__(256) void k_ndt_keys(const float4* __restrict__ pts, size_t pitch, const int* __restrict__ counts,
                                                  const NdtGridMeta* __restrict__ meta, uint32_t* __restrict__ keys, int* __restrict__ vals,
                                                  size_t cap) {
  const int c = blockIdx.y;
  const int n = counts[c];
  const NdtGridMeta m = meta[c];
  const float4* p = pts + (size_t)c * pitch;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 v = p[i];
    const int i0 = (int)(floorf(v.x * m.inv_leaf) - (float)m.min_b[0]);
    const int i1 = (int)(floorf(v.y * m.inv_leaf) - (float)m.min_b[1]);
    const int i2 = (int)(floorf(v.z * m.inv_leaf) - (float)m.min_b[2]);
    keys[(size_t)c * cap + i] = (uint32_t)(i0 + i1 * m.div_b[0] + i2 * m.div_b[0] * m.div_b[1]);
    vals[(size_t)c * cap + i] = i;
  }
}

// flags[i] = 1 where sorted element i starts the run of a voxel that gets a leaf: every occupied voxel (all_voxels: the fp64 leaves
// are kept for the parity read-back, which lists the unusable voxels too) or only those with >= min_points points — the keys are
// sorted, so a run is that long iff the key min_points - 1 places further on is still the same.
__global__ __launch_bounds__(256) void k_ndt_heads(const uint32_t* __restrict__ keys, const int* __restrict__ counts, int* __restrict__ flags,
                                                   size_t cap, int min_points, int all_voxels) {
  const int c = blockIdx.y;
  const int n = counts[c];
  const uint32_t* k = keys + (size_t)c * cap;
  const int m1 = max(min_points, 1) - 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t ki = k[i];
    const bool head = i == 0 || k[i - 1] != ki;
    const bool leaf = head && (all_voxels || (i + m1 < n && k[i + m1] == ki));
    flags[(size_t)c * cap + i] = leaf ? 1 : 0;
  }
}

// one thread per target: the voxels that get a leaf are counted — size its hash (4 slots per leaf, within its allocation)
__global__ void k_ndt_size(const int* __restrict__ totals, const int* __restrict__ counts, const int* __restrict__ slots_alloc,
                           NdtGridMeta* __restrict__ meta, int U) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= U) return;
  const int nl = counts[c] > 0 ? totals[c] : 0;
  uint32_t slots = 16, lg = 4;
  while (slots < 4u * (uint32_t)nl && (int)(slots << 1) <= slots_alloc[c]) { slots <<= 1; lg++; }
  meta[c].n_leaves = nl;
  meta[c].hshift = 32 - lg;
  meta[c].hmask = slots - 1;
}

__global__ __launch_bounds__(256) void k_ndt_table_clear(NdtSlot* __restrict__ table, const long long* __restrict__ base,
                                                         const NdtGridMeta* __restrict__ meta) {
  const int c = blockIdx.y;
  const int slots = (int)meta[c].hmask + 1;
  int2* t = reinterpret_cast<int2*>(table + base[3 * c + 1]);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += gridDim.x * blockDim.x) t[i] = make_int2((int)kEmptyKey, -1);
}

constexpr int kLeafBlock = 256;

// Per-voxel sums by a SEGMENTED SCAN over the sorted points (round 4: the lane at the head of a run pulled the points of the lanes
// behind it one shuffle at a time — 63 dependent steps per wave where a 2 m voxel holds a hundred points).  Every lane fetches the
// point at its sorted position (one gather per lane, all in flight together) and forms its 9 products; six Hillis-Steele steps add
// them up inside the runs of the wave.  The last lane of a run's piece inside this wave owns the piece: the piece that starts at the
// run's head goes to the leaf's slot of `acc` (if the voxel gets a leaf), a piece that continues a run begun in an earlier wave goes
// to the wave's `carry` (cnt = 0: the wave starts with a head, nothing carried) — k_ndt_leaf_finalize adds the carries of the waves a
// run spans, in wave order.  Fixed tree, fixed order: bitwise reproducible.
__global__ __launch_bounds__(kLeafBlock) void k_ndt_leaf_sums(const float4* __restrict__ pts_all, size_t pitch, const uint32_t* __restrict__ keys_all,
                                                               const int* __restrict__ vals_all, const int* __restrict__ flags_all,
                                                               const int* __restrict__ local_all, const int* __restrict__ blockoff_all,
                                                               const int* __restrict__ counts, const long long* __restrict__ base,
                                                               NdtSums* __restrict__ acc_all, NdtSums* __restrict__ carry_all, size_t cap,
                                                               int nblk_scan) {
  const int c = blockIdx.y;
  const int n = counts[c];
  if ((int)(blockIdx.x * kLeafBlock) >= n) return;
  const float4* pts = pts_all + (size_t)c * pitch;
  const uint32_t* keys = keys_all + (size_t)c * cap;
  const int i = blockIdx.x * kLeafBlock + threadIdx.x;
  const int lane = lane_id();
  const bool valid = i < n;
  uint32_t key = 0xFFFFFFFEu, kprev = 0xFFFFFFFDu;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  int leaf_flag = 0;
  if (valid) {
    key = keys[i];
    kprev = i > 0 ? keys[i - 1] : ~key;
    p = pts[vals_all[(size_t)c * cap + i]];
    leaf_flag = flags_all[(size_t)c * cap + i];
  }
  const bool head = valid && key != kprev;
  const unsigned long long hm = __ballot(head);
  // first lane of this lane's piece: the last head at or below it, lane 0 if the run began in an earlier wave
  const unsigned long long below = hm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  const int start = below ? (63 - __clzll(below)) : 0;
  double v[9];
  {
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    v[0] = x; v[1] = y; v[2] = z;
    v[3] = x * x; v[4] = x * y; v[5] = x * z; v[6] = y * y; v[7] = y * z; v[8] = z * z;
  }
  int cnt = valid ? 1 : 0;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const bool take = lane - off >= start;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double t = __shfl_up(v[k], off, kWave);
      if (take) v[k] += t;
    }
    const int tc = __shfl_up(cnt, off, kWave);
    if (take) cnt += tc;
  }
  // the piece ends at this lane when the next lane starts a run, or is past the end, or this is lane 63
  const bool next_head = (lane == 63) ? true : (((hm >> (lane + 1)) & 1ull) != 0);
  const bool next_valid = (lane == 63) ? false : (i + 1 < n);
  const bool last = valid && (lane == 63 || next_head || !next_valid);
  const bool piece_has_head = ((hm >> start) & 1ull) != 0;
  // (values of the piece's first lane, needed by its last lane)
  const int li_lane = (valid && leaf_flag) ? local_all[(size_t)c * cap + i] + blockoff_all[(size_t)c * nblk_scan + i / kScanTile] : -1;
  const int li_start = __shfl(li_lane, start, kWave);
  const int wave_global = i >> 6;
  NdtSums* carry = carry_all + base[3 * c + 2];
  if (lane == 0 && valid && head) {  // nothing is carried into this wave
    NdtSums z;
#pragma unroll
    for (int k = 0; k < 9; k++) z.v[k] = 0.0;
    z.cnt = 0; z.head = 0;
    carry[wave_global] = z;
  }
  if (last) {
    NdtSums o;
#pragma unroll
    for (int k = 0; k < 9; k++) o.v[k] = v[k];
    o.cnt = cnt;
    o.head = i - (lane - start);
    if (!piece_has_head) carry[wave_global] = o;
    else if (li_start >= 0) acc_all[base[3 * c] + li_start] = o;
  }
}

// One thread per leaf: the run's total = its head piece + the carries of the waves it runs on into; mean, covariance, eigenvalue
// inflation, inverse (N1); the 48-byte leaf, the fp64 leaf where kept, the hash entry of a usable voxel.
__global__ __launch_bounds__(64) void k_ndt_leaf_finalize(const uint32_t* __restrict__ keys_all, const int* __restrict__ counts,
                                                          const NdtGridMeta* __restrict__ meta_all, const long long* __restrict__ base,
                                                          const NdtSums* __restrict__ acc_all, const NdtSums* __restrict__ carry_all,
                                                          NdtLeaf* __restrict__ leaf_all, NdtLeafD* __restrict__ leaf_d_all,
                                                          uint32_t* __restrict__ leaf_key_all, NdtSlot* __restrict__ table_all, size_t cap,
                                                          int min_points, double eig_mult) {
  const int c = blockIdx.y;
  const NdtGridMeta m = meta_all[c];
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= m.n_leaves) return;
  const int n = counts[c];
  const long long lbase = base[3 * c];
  const NdtSums h = acc_all[lbase + li];
  const NdtSums* carry = carry_all + base[3 * c + 2];
  double sx = h.v[0], sy = h.v[1], sz = h.v[2];
  Sym3<double> sq = {h.v[3], h.v[4], h.v[5], h.v[6], h.v[7], h.v[8]};
  int cnt = h.cnt;
  const uint32_t key = keys_all[(size_t)c * cap + h.head];
  if (((h.head + cnt) & 63) == 0) {  // the head piece ran to the end of its wave: the run may go on
    for (int w = (h.head + cnt) >> 6; w * 64 < n; w++) {
      const NdtSums cr = carry[w];
      if (cr.cnt == 0) break;
      sx += cr.v[0]; sy += cr.v[1]; sz += cr.v[2];
      sq.xx += cr.v[3]; sq.xy += cr.v[4]; sq.xz += cr.v[5]; sq.yy += cr.v[6]; sq.yz += cr.v[7]; sq.zz += cr.v[8];
      cnt += cr.cnt;
      if (cr.cnt < 64) break;
    }
  }
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
    Sym3<double> cv;
    cv.xx = ((sq.xx - 2 * (sx * mx)) / nn + mx * mx) * f;
    cv.xy = ((sq.xy - 2 * (sx * my)) / nn + mx * my) * f;
    cv.xz = ((sq.xz - 2 * (sx * mz)) / nn + mx * mz) * f;
    cv.yy = ((sq.yy - 2 * (sy * my)) / nn + my * my) * f;
    cv.yz = ((sq.yz - 2 * (sy * mz)) / nn + my * mz) * f;
    cv.zz = ((sq.zz - 2 * (sz * mz)) / nn + mz * mz) * f;
    double w[3], V[9];
    jacobi_eigen_sym3(cv, w, V);
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
        cv = r;
      }
      const Sym3<double> ic = sym3_inverse(cv);
      L.cov[0] = cv.xx; L.cov[1] = cv.xy; L.cov[2] = cv.xz; L.cov[3] = cv.xy; L.cov[4] = cv.yy; L.cov[5] = cv.yz; L.cov[6] = cv.xz; L.cov[7] = cv.yz; L.cov[8] = cv.zz;
      L.icov[0] = ic.xx; L.icov[1] = ic.xy; L.icov[2] = ic.xz; L.icov[3] = ic.xy; L.icov[4] = ic.yy; L.icov[5] = ic.yz; L.icov[6] = ic.xz; L.icov[7] = ic.yz; L.icov[8] = ic.zz;
      bool bad = false;
      for (int k = 0; k < 9; k++) bad = bad || isinf(L.icov[k]);
      if (bad) { L.n = -1; usable = false; }
    }
  }
  if (leaf_d_all) leaf_d_all[lbase + li] = L;
  if (leaf_key_all) leaf_key_all[lbase + li] = key;
  // the fp32 leaf: mean relative to the centre of the voxel, ((float)index + 0.5f) * leaf — the derivative kernel forms the same product
  const int d0 = max(m.div_b[0], 1), d1 = max(m.div_b[1], 1);
  const int a0 = (int)(key % (uint32_t)d0) + m.min_b[0], a1 = (int)((key / (uint32_t)d0) % (uint32_t)d1) + m.min_b[1],
            a2 = (int)(key / ((uint32_t)d0 * (uint32_t)d1)) + m.min_b[2];
  NdtLeaf F;
  F.m[0] = (float)(L.mean[0] - (double)(((float)a0 + 0.5f) * m.leaf));
  F.m[1] = (float)(L.mean[1] - (double)(((float)a1 + 0.5f) * m.leaf));
  F.m[2] = (float)(L.mean[2] - (double)(((float)a2 + 0.5f) * m.leaf));
  F.ic[0] = (float)L.icov[0]; F.ic[1] = (float)L.icov[1]; F.ic[2] = (float)L.icov[2];
  F.ic[3] = (float)L.icov[4]; F.ic[4] = (float)L.icov[5]; F.ic[5] = (float)L.icov[8];
  F.pad[0] = F.pad[1] = F.pad[2] = 0.f;
  leaf_all[lbase + li] = F;
  if (usable) {
    NdtSlot* table = table_all + base[3 * c + 1];
    uint32_t hh = (key * 0x9E3779B1u) >> m.hshift;
    for (;;) {
      const uint32_t prev = atomicCAS(&table[hh].key, kEmptyKey, key);
      if (prev == kEmptyKey) { table[hh].leaf = li; break; }
      hh = (hh + 1) & m.hmask;
    }
  }
}

// U clouds: cloud u = d_pts[u * pitch ... + count[u]) in any point order; d_counts on the device, h_bound[u] >= count[u] known to the
// host (the counts BEFORE a prefilter: what sizes the leaf and hash allocations — nothing is read back).  ws: reserved for (U, cap).
void ndt_build_targets(hipStream_t s, Workspace& ws, NdtTargets& t, const float4* d_pts, size_t pitch, const int* d_counts, const int* h_bound, int U,
                       size_t cap, const mi355_ndt_params& p, bool keep_fp64_leaves) {
  if (!(p.resolution > 0)) throw ArgError(MI355_EINVAL, "ndt resolution must be positive");
  t.U = U;
  t.has_d = keep_fp64_leaves;
  t.h_meta_valid = false;
  t.h_base.assign((size_t)U * 3, 0);
  t.h_slots.assign((size_t)U, 16);
  t.h_refs.resize((size_t)U);
  const int minp = std::max(p.min_points_per_voxel, 1);
  long long nleaf = 0, nslot = 0, ncarry = 0, max_bound = 1, max_leaves = 1;
  for (int u = 0; u < U; u++) {
    const long long b = std::max<long long>(1, std::min<long long>(h_bound[u], (long long)cap));
    // a voxel with a leaf holds >= min_points_per_voxel points (every occupied voxel when the fp64 leaves are kept for the read-back);
    // 4 slots per leaf is the aim, > 2 the floor
    const long long leaves = keep_fp64_leaves ? b : b / minp + 1;
    int slots = 16;
    while ((long long)slots < 2 * (b / minp) + 1 && slots < (1 << 30)) slots <<= 1;
    max_bound = std::max(max_bound, b);
    max_leaves = std::max(max_leaves, leaves);
    t.h_base[(size_t)u * 3] = nleaf;
    t.h_base[(size_t)u * 3 + 1] = nslot;
    t.h_base[(size_t)u * 3 + 2] = ncarry;
    t.h_slots[(size_t)u] = slots;
    nleaf += leaves;
    nslot += slots;
    ncarry += b / 64 + 1;
  }
  t.leaf.ensure((size_t)nleaf);
  t.acc.ensure((size_t)nleaf);
  t.carry.ensure((size_t)ncarry);
  if (keep_fp64_leaves) { t.leaf_d.ensure((size_t)nleaf); t.leaf_key.ensure((size_t)nleaf); }
  t.table.ensure((size_t)nslot);
  t.meta.ensure((size_t)U);
  t.refs.ensure((size_t)U);
  t.base.ensure((size_t)U * 3);
  t.slots_alloc.ensure((size_t)U);
  for (int u = 0; u < U; u++)
    t.h_refs[(size_t)u] = NdtTargetRef{t.leaf.p + t.h_base[(size_t)u * 3], keep_fp64_leaves ? t.leaf_d.p + t.h_base[(size_t)u * 3] : nullptr,
                                       t.table.p + t.h_base[(size_t)u * 3 + 1], t.meta.p + u};
  MI_HIP(hipMemcpyAsync(t.refs.p, t.h_refs.data(), (size_t)U * sizeof(NdtTargetRef), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(t.base.p, t.h_base.data(), (size_t)U * 3 * sizeof(long long), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(t.slots_alloc.p, t.h_slots.data(), (size_t)U * sizeof(int), hipMemcpyHostToDevice, s));
  int max_slots = 16;
  for (int u = 0; u < U; u++) max_slots = std::max(max_slots, t.h_slots[(size_t)u]);
  // a few blocks per cloud walk it (grid sized for the chip, not for the capacity)
  const int gx = std::max(1, std::min(div_up((long)max_bound, 1024), std::max(4, div_up(4096, U))));
  {
    TimedLaunch tl(KF_NDT_TARGET, s);
    hipLaunchKernelGGL(k_ndt_bounds_init, dim3(div_up(U * 6, 256)), dim3(256), 0, s, ws.bounds.p, U);
    hipLaunchKernelGGL(k_ndt_bounds, dim3(gx, U), dim3(256), 0, s, d_pts, pitch, d_counts, ws.bounds.p);
    hipLaunchKernelGGL(k_ndt_meta, dim3(div_up(U, 64)), dim3(64), 0, s, ws.bounds.p, d_counts, t.meta.p, ws.sort_passes.p, (float)p.resolution, U);
    hipLaunchKernelGGL(k_ndt_keys, dim3(gx, U), dim3(256), 0, s, d_pts, pitch, d_counts, t.meta.p, ws.keys_a.p, ws.vals_a.p, cap);
  }
  {
    FamilyScope fs(KF_NDT_SORT);
    sort_pairs(s, ws, U, cap, d_counts, 4, ws.sort_passes.p);
  }
  {
    TimedLaunch tl(KF_NDT_TARGET, s);
    hipLaunchKernelGGL(k_ndt_heads, dim3(gx, U), dim3(256), 0, s, ws.keys_a.p, d_counts, ws.flags.p, cap, p.min_points_per_voxel, keep_fp64_leaves ? 1 : 0);
    scan_flags(s, ws, U, cap, d_counts);
    hipLaunchKernelGGL(k_ndt_size, dim3(div_up(U, 64)), dim3(64), 0, s, ws.totals.p, d_counts, t.slots_alloc.p, t.meta.p, U);
    hipLaunchKernelGGL(k_ndt_table_clear, dim3(std::max(1, std::min(div_up(max_slots, 1024), std::max(4, div_up(4096, U)))), U), dim3(256), 0, s,
                       t.table.p, t.base.p, t.meta.p);
  }
  TimedLaunch tl(KF_NDT_LEAVES, s);
  hipLaunchKernelGGL(k_ndt_leaf_sums, dim3(div_up((long)max_bound, kLeafBlock), U), dim3(kLeafBlock), 0, s, d_pts, pitch, ws.keys_a.p, ws.vals_a.p,
                     ws.flags.p, ws.scan_local.p, ws.scan_block.p, d_counts, t.base.p, t.acc.p, t.carry.p, cap, div_up((long)cap, kScanTile));
  hipLaunchKernelGGL(k_ndt_leaf_finalize, dim3(div_up((long)max_leaves, 64), U), dim3(64), 0, s, ws.keys_a.p, d_counts, t.meta.p, t.base.p, t.acc.p,
                     t.carry.p, t.leaf.p, keep_fp64_leaves ? t.leaf_d.p : nullptr, keep_fp64_leaves ? t.leaf_key.p : nullptr, t.table.p, cap,
                     p.min_points_per_voxel, p.min_covar_eigvalue_mult);
  MI_LAUNCH_CHECK();
}

void ndt_cells(hipStream_t s, NdtTargets& t, int64_t* leaf_idx, int* leaf_n, double* mean3, double* cov9, double* icov9, int* min_b3, int* div_b3) {
  if (!t.has_d) throw ArgError(MI355_ESTATE, "ndt_cells: the fp64 leaves of this target were not kept");
  ndt_read_meta(s, t);
  const NdtGridMeta& hm = t.h_meta[0];
  const int nl = hm.n_leaves;
  std::vector<NdtLeafD> L((size_t)nl);
  std::vector<uint32_t> K((size_t)nl);
  if (nl > 0) {
    MI_HIP(hipMemcpyAsync(L.data(), t.leaf_d.p, (size_t)nl * sizeof(NdtLeafD), hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(K.data(), t.leaf_key.p, (size_t)nl * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
  }
  for (int i = 0; i < nl; i++) {
    leaf_idx[i] = (int64_t)K[(size_t)i];
    leaf_n[i] = L[(size_t)i].n;
    memcpy(mean3 + 3 * i, L[(size_t)i].mean, 3 * sizeof(double));
    memcpy(cov9 + 9 * i, L[(size_t)i].cov, 9 * sizeof(double));
    memcpy(icov9 + 9 * i, L[(size_t)i].icov, 9 * sizeof(double));
  }
  for (int d = 0; d < 3; d++) { min_b3[d] = hm.min_b[d]; div_b3[d] = hm.div_b[d]; }
}

// ---- derivatives ------------------------------------------------------------------------------------------------------
constexpr int kNdtAcc = 28;  // score + 6 gradient + 21 Hessian (upper triangle, row-major)
constexpr int kNdtBlock = NDT_BLOCK;
constexpr int kNdtPartStride = 32;

// getNeighborhoodAtPoint7: the voxel holding the point and its 6 face neighbours (inside the grid, usable).  All seven first probes
// are in flight together; a probe that lands on another voxel's slot walks on (rare at load factor <= 1/4).
__device__ __forceinline__ void ndt_neighbors7(const NdtSlot* __restrict__ table, const NdtGridMeta& m, const int ijk[3], int li[7]) {
  // coordinates relative to the grid (a point just outside it still has neighbours inside); one range test per axis and offset
  const int r0 = ijk[0] - m.min_b[0], r1 = ijk[1] - m.min_b[1], r2 = ijk[2] - m.min_b[2];
  const uint32_t d0 = (uint32_t)m.div_b[0], d1 = (uint32_t)m.div_b[1], d2 = (uint32_t)m.div_b[2];
  const bool x0 = (uint32_t)r0 < d0, xp = (uint32_t)(r0 + 1) < d0, xm = (uint32_t)(r0 - 1) < d0;
  const bool y0 = (uint32_t)r1 < d1, yp = (uint32_t)(r1 + 1) < d1, ym = (uint32_t)(r1 - 1) < d1;
  const bool z0 = (uint32_t)r2 < d2, zp = (uint32_t)(r2 + 1) < d2, zm = (uint32_t)(r2 - 1) < d2;
  const uint32_t sy = d0, sz = d0 * d1;
  const uint32_t khome = (uint32_t)r0 + (uint32_t)r1 * sy + (uint32_t)r2 * sz;
  const bool in[7] = {x0 && y0 && z0, xp && y0 && z0, xm && y0 && z0, x0 && yp && z0, x0 && ym && z0, x0 && y0 && zp, x0 && y0 && zm};
  const uint32_t key[7] = {khome, khome + 1u, khome - 1u, khome + sy, khome - sy, khome + sz, khome - sz};
  uint32_t h[7];
  int2 sl[7];
#pragma unroll
  for (int k = 0; k < 7; k++) {
    h[k] = (key[k] * 0x9E3779B1u) >> m.hshift;
    sl[k] = in[k] ? *reinterpret_cast<const int2*>(&table[h[k]]) : make_int2((int)kEmptyKey, -1);
  }
#pragma unroll
  for (int k = 0; k < 7; k++) {
    while ((uint32_t)sl[k].x != key[k] && (uint32_t)sl[k].x != kEmptyKey) {
      h[k] = (h[k] + 1) & m.hmask;
      sl[k] = *reinterpret_cast<const int2*>(&table[h[k]]);
    }
    li[k] = ((uint32_t)sl[k].x == key[k]) ? sl[k].y : -1;
  }
}

struct NdtPassParams {
  float T[16];
  float d1, d2;
  double gd1, gd2;
};

// computeDerivatives (N3) of the points block `block` of `nblocks` strides over; its 28 sums go to partials_row[0..27].
//
// WHERE THE LEAVES COME FROM.  The kernel that runs (NDT_STAGED = 0, the default) has every lane probe the hash for its seven DIRECT7
// voxels - all seven probes in flight - and fetch its (up to seven) 48-byte leaves itself.  The 64 points of a wave are neighbours in
// space (the sources of a batch are in the voxel hash's Morton order), so their neighbourhoods overlap almost entirely and the leaves
// come from L2: the counters have the HBM traffic of a launch at the source points alone, and the kernel bound by VALU issue
// (valu_issue_frac 0.70: profiles/r06_ndt_issue.json).
// The alternative behind NDT_STAGED = 1 - kept as an A/B build only - stages a wave's cells in LDS: the wave takes the box of its
// points' voxels grown by one voxel, lane c probes the hash for box cell c and copies that leaf into slot c of the wave's LDS image
// (<= kNdtSlots cells, two passes), and every lane reads its seven cells from there - two dependent global loads per WAVE instead of
// per lane.  Built in round 5 and measured 24 % SLOWER than the per-lane loads (profiles/r05_ndt_variants.txt): the image costs more
// instructions than the L2 hits it replaces.  Same arithmetic either way: the leaf bytes are the same.
constexpr int kNdtSlots = 128;  // cells of a wave's LDS image (48 B each; the image shares the wave's reduction scratch)
static_assert(kNdtSlots * 48 <= kWave * kBlockSumRow * 8, "the LDS image of a wave fits the scratch of its reduction");

template <bool HESS>
__device__ __forceinline__ void ndt_derivatives_body(const float4* __restrict__ src, int n, const NdtLeaf* __restrict__ leaves,
                                                     const NdtSlot* __restrict__ table, const NdtGridMeta& m, const NdtAngles& A,
                                                     const NdtPassParams& P, int block, int nblocks, double* __restrict__ partials_row,
                                                     double* wave_buf, double* smem) {
  double acc[kNdtAcc];
#pragma unroll
  for (int k = 0; k < kNdtAcc; k++) acc[k] = 0.0;
  const float nh_d2 = -0.5f * P.d2;
  float4* image = reinterpret_cast<float4*>(wave_buf);  // [kNdtSlots][3]
  const int lane = lane_id();
  for (int i0 = block * kNdtBlock; i0 < n; i0 += nblocks * kNdtBlock) {  // (block-uniform trip count: the waves reduce together)
    const int idx = i0 + (int)threadIdx.x;
    const bool valid = idx < n;
    const float4 xp = valid ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float xt0 = transform_row(P.T, xp.x, xp.y, xp.z), xt1 = transform_row(P.T + 4, xp.x, xp.y, xp.z), xt2 = transform_row(P.T + 8, xp.x, xp.y, xp.z);
    const int ijk[3] = {(int)floorf(xt0 * m.inv_leaf), (int)floorf(xt1 * m.inv_leaf), (int)floorf(xt2 * m.inv_leaf)};
    // ---- the wave's box of voxels, grown by one: wave-uniform
    auto imin = [](int a, int b) { return a < b ? a : b; };
    auto imax = [](int a, int b) { return a > b ? a : b; };
    int lo[3] = {0, 0, 0}, ext[3] = {0, 0, 0};
    if (NDT_STAGED) {
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int mn = wave_reduce64(valid ? ijk[d] : 0x7FFFFFFF, imin), mx = wave_reduce64(valid ? ijk[d] : (int)0x80000000, imax);
        lo[d] = mn - 1;
        ext[d] = (mx >= mn) ? (int)min((long long)mx - (long long)mn + 3ll, 1ll << 20) : 0;
      }
    }
    const long long ncell = (long long)ext[0] * ext[1] * ext[2];
    const bool staged = NDT_STAGED && ncell > 0 && ncell <= kNdtSlots;
    int li[7];
    if (staged) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the image of the previous trip has been read)
      for (int c0 = 0; c0 < (int)ncell; c0 += kWave) {
        const int c = c0 + lane;
        float4 L0 = make_float4(0.f, 0.f, 0.f, 0.f), L1 = L0, L2 = L0;
        if (c < (int)ncell) {
          const int c0x = lo[0] + c % ext[0], c1x = lo[1] + (c / ext[0]) % ext[1], c2x = lo[2] + c / (ext[0] * ext[1]);
          const bool in = !(c0x < m.min_b[0] || c0x > m.max_b[0] || c1x < m.min_b[1] || c1x > m.max_b[1] || c2x < m.min_b[2] || c2x > m.max_b[2]);
          if (in) {
            const uint32_t key = (uint32_t)((c0x - m.min_b[0]) + (c1x - m.min_b[1]) * m.div_b[0] + (c2x - m.min_b[2]) * m.div_b[0] * m.div_b[1]);
            uint32_t h = (key * 0x9E3779B1u) >> m.hshift;
            int2 sl = *reinterpret_cast<const int2*>(&table[h]);
            while ((uint32_t)sl.x != key && (uint32_t)sl.x != kEmptyKey) {
              h = (h + 1) & m.hmask;
              sl = *reinterpret_cast<const int2*>(&table[h]);
            }
            if ((uint32_t)sl.x == key) {
              const float4* Lp = reinterpret_cast<const float4*>(leaves + sl.y);
              L0 = Lp[0]; L1 = Lp[1]; L2 = Lp[2];
              L2.y = 1.f;  // present
            }
          }
          image[c * 3] = L0; image[c * 3 + 1] = L1; image[c * 3 + 2] = L2;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else if (valid) {
      ndt_neighbors7(table, m, ijk, li);
    }
    if (!valid) continue;
    // running sums over the (at most 7) voxels: score, a = SUM w q, M = SUM w (C^-1 - d2 q q^T)
    float sc = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    float mxx = 0.f, mxy = 0.f, mxz = 0.f, myy = 0.f, myz = 0.f, mzz = 0.f;
    bool any = false;
    const int s_home = (ijk[0] - lo[0]) + ((ijk[1] - lo[1]) + (ijk[2] - lo[2]) * ext[1]) * ext[0];
    // voxel centres along each axis: the voxel below, the point's own, the one above (the leaf builder's product, bit for bit)
    float cx[3], cy[3], cz[3];
#pragma unroll
    for (int o = 0; o < 3; o++) {
      cx[o] = ((float)(ijk[0] + o - 1) + 0.5f) * m.leaf;
      cy[o] = ((float)(ijk[1] + o - 1) + 0.5f) * m.leaf;
      cz[o] = ((float)(ijk[2] + o - 1) + 0.5f) * m.leaf;
    }
#pragma unroll
    for (int k = 0; k < 7; k++) {
      const int dx = (k == 1) - (k == 2), dy = (k == 3) - (k == 4), dz = (k == 5) - (k == 6);
      float4 L0, L1;
      float izz;
      if (staged) {
        const int sidx = s_home + dx + (dy + dz * ext[1]) * ext[0];
        const float4 L2 = image[sidx * 3 + 2];
        if (L2.y == 0.f) continue;
        L0 = image[sidx * 3]; L1 = image[sidx * 3 + 1];
        izz = L2.x;
      } else {
        if (li[k] < 0) continue;
        const float4* Lp = reinterpret_cast<const float4*>(leaves + li[k]);
        L0 = Lp[0]; L1 = Lp[1];
        izz = reinterpret_cast<const float*>(Lp)[8];
      }
      // x' - mean = (x' - centre) - (mean - centre)
      const float e0 = (xt0 - cx[dx + 1]) - L0.x;
      const float e1 = (xt1 - cy[dy + 1]) - L0.y;
      const float e2 = (xt2 - cz[dz + 1]) - L0.z;
      const float ixx = L0.w, ixy = L1.x, ixz = L1.y, iyy = L1.z, iyz = L1.w;
      const float q0 = fmaf(ixx, e0, fmaf(ixy, e1, ixz * e2));
      const float q1 = fmaf(ixy, e0, fmaf(iyy, e1, iyz * e2));
      const float q2 = fmaf(ixz, e0, fmaf(iyz, e1, izz * e2));
      const float s2 = fmaf(e0, q0, fmaf(e1, q1, e2 * q2));
      const float ex = __expf(nh_d2 * s2);
      const float score_inc = -P.d1 * ex;
      const float ed = P.d2 * ex;
      const bool ok = !(ed > 1.f || ed < 0.f || ed != ed);
      any = any || ok;
      const float w = ok ? P.d1 * ed : 0.f;
      sc += ok ? score_inc : 0.f;
      a0 = fmaf(w, q0, a0); a1 = fmaf(w, q1, a1); a2 = fmaf(w, q2, a2);
      if (HESS) {
        const float t = -P.d2 * w;
        const float t0 = t * q0, t1 = t * q1, t2 = t * q2;
        mxx = fmaf(w, ixx, fmaf(t0, q0, mxx)); mxy = fmaf(w, ixy, fmaf(t0, q1, mxy)); mxz = fmaf(w, ixz, fmaf(t0, q2, mxz));
        myy = fmaf(w, iyy, fmaf(t1, q1, myy)); myz = fmaf(w, iyz, fmaf(t1, q2, myz)); mzz = fmaf(w, izz, fmaf(t2, q2, mzz));
      }
    }
    if (!any) continue;
    // computePointDerivatives: the non-trivial entries of the 3 x 6 point Jacobian (columns 3..5) ...
    const float x0 = xp.x, x1 = xp.y, x2 = xp.z;
    float xj[8];
#pragma unroll
    for (int r = 0; r < 8; r++) xj[r] = fmaf(A.jf[r][0], x0, fmaf(A.jf[r][1], x1, A.jf[r][2] * x2));
    // J3 = (0, xj0, xj1)   J4 = (xj2, xj3, xj4)   J5 = (xj5, xj6, xj7)
    acc[0] += (double)sc;
    acc[1] += (double)a0;
    acc[2] += (double)a1;
    acc[3] += (double)a2;
    acc[4] += (double)fmaf(a1, xj[0], a2 * xj[1]);
    acc[5] += (double)fmaf(a0, xj[2], fmaf(a1, xj[3], a2 * xj[4]));
    acc[6] += (double)fmaf(a0, xj[5], fmaf(a1, xj[6], a2 * xj[7]));
    if (HESS) {
      // ... and the 3-vectors a..f of the point Hessian: a=(0,xh0,xh1) b=(0,xh2,xh3) c=(0,xh4,xh5) d=xh6..8 e=xh9..11 f=xh12..14
      float xh[15];
#pragma unroll
      for (int r = 0; r < 15; r++) xh[r] = fmaf(A.hf[r][0], x0, fmaf(A.hf[r][1], x1, A.hf[r][2] * x2));
      // N_j = M J_j
      const float n30 = fmaf(mxy, xj[0], mxz * xj[1]), n31 = fmaf(myy, xj[0], myz * xj[1]), n32 = fmaf(myz, xj[0], mzz * xj[1]);
      const float n40 = fmaf(mxx, xj[2], fmaf(mxy, xj[3], mxz * xj[4])), n41 = fmaf(mxy, xj[2], fmaf(myy, xj[3], myz * xj[4])),
                  n42 = fmaf(mxz, xj[2], fmaf(myz, xj[3], mzz * xj[4]));
      const float n50 = fmaf(mxx, xj[5], fmaf(mxy, xj[6], mxz * xj[7])), n51 = fmaf(mxy, xj[5], fmaf(myy, xj[6], myz * xj[7])),
                  n52 = fmaf(mxz, xj[5], fmaf(myz, xj[6], mzz * xj[7]));
      // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
      acc[7] += (double)mxx;  acc[8] += (double)mxy;  acc[9] += (double)mxz;  acc[10] += (double)n30; acc[11] += (double)n40; acc[12] += (double)n50;
      acc[13] += (double)myy; acc[14] += (double)myz; acc[15] += (double)n31; acc[16] += (double)n41; acc[17] += (double)n51;
      acc[18] += (double)mzz; acc[19] += (double)n32; acc[20] += (double)n42; acc[21] += (double)n52;
      // H_ij (i, j >= 3) = J_i . N_j + a . Hpt_ij
      acc[22] += (double)fmaf(xj[0], n31, fmaf(xj[1], n32, fmaf(a1, xh[0], a2 * xh[1])));
      acc[23] += (double)fmaf(xj[0], n41, fmaf(xj[1], n42, fmaf(a1, xh[2], a2 * xh[3])));
      acc[24] += (double)fmaf(xj[0], n51, fmaf(xj[1], n52, fmaf(a1, xh[4], a2 * xh[5])));
      acc[25] += (double)fmaf(xj[2], n40, fmaf(xj[3], n41, fmaf(xj[4], n42, fmaf(a0, xh[6], fmaf(a1, xh[7], a2 * xh[8])))));
      acc[26] += (double)fmaf(xj[2], n50, fmaf(xj[3], n51, fmaf(xj[4], n52, fmaf(a0, xh[9], fmaf(a1, xh[10], a2 * xh[11])))));
      acc[27] += (double)fmaf(xj[5], n50, fmaf(xj[6], n51, fmaf(xj[7], n52, fmaf(a0, xh[12], fmaf(a1, xh[13], a2 * xh[14])))));
    }
  }
  const double total = block_sum_columns<kNdtAcc, kNdtBlock / kWave>(acc, wave_buf, smem);
  if ((int)threadIdx.x < kNdtAcc) partials_row[threadIdx.x] = total;
}

// computeHessian / updateHessian: fp64 per-point maths (requested after a refined line search only: legacy_mt_skip = 0).
__device__ __forceinline__ void ndt_hessian64_body(const float4* __restrict__ src, int n, const NdtLeafD* __restrict__ leaves,
                                                   const NdtSlot* __restrict__ table, const NdtGridMeta& m, const NdtAngles& A,
                                                   const NdtPassParams& P, int block, int nblocks, double* __restrict__ partials_row,
                                                   double* wave_buf, double* smem) {
  double acc[kNdtAcc];
#pragma unroll
  for (int k = 0; k < kNdtAcc; k++) acc[k] = 0.0;
  for (int idx = block * kNdtBlock + threadIdx.x; idx < n; idx += nblocks * kNdtBlock) {
    const float4 xp = src[idx];
    const float xt0 = transform_row(P.T, xp.x, xp.y, xp.z), xt1 = transform_row(P.T + 4, xp.x, xp.y, xp.z), xt2 = transform_row(P.T + 8, xp.x, xp.y, xp.z);
    const int ijk[3] = {(int)floorf(xt0 * m.inv_leaf), (int)floorf(xt1 * m.inv_leaf), (int)floorf(xt2 * m.inv_leaf)};
    int nb[7];
    ndt_neighbors7(table, m, ijk, nb);
    bool any = false;
#pragma unroll
    for (int k = 0; k < 7; k++) any = any || nb[k] >= 0;
    if (!any) continue;
    const double x[3] = {(double)xp.x, (double)xp.y, (double)xp.z};
    double xj[8], xh[15];
#pragma unroll
    for (int a = 0; a < 8; a++) xj[a] = x[0] * A.jd[a][0] + x[1] * A.jd[a][1] + x[2] * A.jd[a][2];
#pragma unroll
    for (int a = 0; a < 15; a++) xh[a] = x[0] * A.hd[a][0] + x[1] * A.hd[a][1] + x[2] * A.hd[a][2];
    double pg[3][6] = {{1, 0, 0, 0, xj[2], xj[5]}, {0, 1, 0, xj[0], xj[3], xj[6]}, {0, 0, 1, xj[1], xj[4], xj[7]}};
    for (int c = 0; c < 7; c++) {
      if (nb[c] < 0) continue;
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
  const double total = block_sum_columns<kNdtAcc, kNdtBlock / kWave>(acc, wave_buf, smem);
  if ((int)threadIdx.x < kNdtAcc) partials_row[threadIdx.x] = total;
}

// ---- solver: P pairs, every pair's More-Thuente state in HBM (N2, N4) --------------------------------------------------------
struct NdtResultRec {  // what leaves the device per pair
  float T[16];
  int converged, iterations, evaluations, n, n_target, done, pad0, pad1;
  double score, trans_probability;
  double hess[36];
};

struct NdtSolver {
  int P_alloc = 0;
  DevBuf<NdtPairState> states;
  DevBuf<int> tgt_of_src, n_active, evals;
  DevBuf<float> guesses;
  DevBuf<double> partials;
  DevBuf<NdtResultRec> recs;
  PinnedBuf<int> h_active;
  PinnedBuf<NdtResultRec> h_recs;
  std::vector<int> h_map;
  std::vector<float> h_guess;
  int P = 0, rounds = 0;
};
NdtSolver* ndt_solver_create() { return new NdtSolver(); }
void ndt_solver_destroy(NdtSolver* s) { delete s; }
int ndt_solver_rounds(const NdtSolver& sv) { return sv.rounds; }

__device__ __forceinline__ int ndt_blocks_dev(int n, int G) { return max(1, min(G, (n + kNdtBlock - 1) / kNdtBlock)); }

__global__ void k_ndt_begin(NdtPairState* __restrict__ states, int* __restrict__ evals, const float* __restrict__ guesses,
                            const int* __restrict__ counts, int P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float g[16];
  for (int k = 0; k < 16; k++) g[k] = guesses[16 * p + k];
  ndt_begin(states[p], g, counts[p]);
  evals[p] = 0;
}

// computeDerivatives of every pair that asked for it (blocks of the other pairs leave at once).  grid = xcd_grid(G, P): the
// blocks of one pair share an XCD, so its target's leaves and hash table are fetched into ONE L2.
__global__ __launch_bounds__(kNdtBlock, NDT_DERIV_WAVES) void k_ndt_derivatives(const float4* __restrict__ src, size_t src_pitch,
                                                               const NdtTargetRef* __restrict__ targets, const int* __restrict__ tgt_of_src,
                                                               const NdtPairState* __restrict__ states, NdtConsts c, int P, int G,
                                                               double* __restrict__ partials) {
  __shared__ double wave_buf[(kNdtBlock / kWave) * kWave * kBlockSumRow];
  __shared__ double smem[(kNdtBlock / kWave) * kNdtAcc];
  int pair, bx;
  if (!xcd_item_chunk(G, P, pair, bx)) return;
  const NdtPairState& st = states[pair];
  const int req = st.request;
  if (req != NDT_REQ_DERIV_HESS && req != NDT_REQ_DERIV) return;
  const int n = st.n, nb = ndt_blocks_dev(n, G);
  if (bx >= nb) return;
  NdtPassParams Pp;
  for (int k = 0; k < 16; k++) Pp.T[k] = st.T[k];
  Pp.d1 = (float)c.gd1; Pp.d2 = (float)c.gd2;
  Pp.gd1 = c.gd1; Pp.gd2 = c.gd2;
  const NdtTargetRef tr = targets[tgt_of_src[pair]];
  const NdtGridMeta m = *tr.meta;
  double* row = partials + ((size_t)pair * G + bx) * kNdtPartStride;
  double* wb = wave_buf + wave_id() * kWave * kBlockSumRow;
  if (req == NDT_REQ_DERIV_HESS)
    ndt_derivatives_body<true>(src + (size_t)pair * src_pitch, n, tr.leaf, tr.table, m, st.ang, Pp, bx, nb, row, wb, smem);
  else
    ndt_derivatives_body<false>(src + (size_t)pair * src_pitch, n, tr.leaf, tr.table, m, st.ang, Pp, bx, nb, row, wb, smem);
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_hessian64(const float4* __restrict__ src, size_t src_pitch,
                                                             const NdtTargetRef* __restrict__ targets, const int* __restrict__ tgt_of_src,
                                                             const NdtPairState* __restrict__ states, NdtConsts c, int P, int G,
                                                             double* __restrict__ partials) {
  __shared__ double wave_buf[(kNdtBlock / kWave) * kWave * kBlockSumRow];
  __shared__ double smem[(kNdtBlock / kWave) * kNdtAcc];
  int pair, bx;
  if (!xcd_item_chunk(G, P, pair, bx)) return;
  const NdtPairState& st = states[pair];
  if (st.request != NDT_REQ_HESS64) return;
  const int n = st.n, nb = ndt_blocks_dev(n, G);
  if (bx >= nb) return;
  NdtPassParams Pp;
  for (int k = 0; k < 16; k++) Pp.T[k] = st.T[k];
  Pp.d1 = (float)c.gd1; Pp.d2 = (float)c.gd2;
  Pp.gd1 = c.gd1; Pp.gd2 = c.gd2;
  const NdtTargetRef tr = targets[tgt_of_src[pair]];
  const NdtGridMeta m = *tr.meta;
  ndt_hessian64_body(src + (size_t)pair * src_pitch, n, tr.leaf_d, tr.table, m, st.ang, Pp, bx, nb,
                     partials + ((size_t)pair * G + bx) * kNdtPartStride, wave_buf + wave_id() * kWave * kBlockSumRow, smem);
}

// One wave per pair: lanes 0..27 add the pair's block partials in block order, lane 0 advances the pair's state machine to its
// next request.  n_active (optional) counts the pairs still asking.
__global__ __launch_bounds__(64) void k_ndt_advance(NdtPairState* __restrict__ states, int* __restrict__ evals, const double* __restrict__ partials,
                                                    NdtConsts c, int P, int G, int* __restrict__ n_active) {
  __shared__ double sums[32];
  const int pair = blockIdx.x, lane = threadIdx.x;
  NdtPairState& st = states[pair];
  if (st.request == NDT_REQ_NONE) return;
  const int nb = ndt_blocks_dev(st.n, G);
  if (lane < kNdtAcc) {
    double s = 0;
    for (int g = 0; g < nb; g++) s += partials[((size_t)pair * G + g) * kNdtPartStride + lane];
    sums[lane] = s;
  }
  __syncthreads();
  if (lane == 0) {
    evals[pair]++;
    ndt_advance(st, sums, c);
    if (n_active && st.request != NDT_REQ_NONE) atomicAdd(n_active, 1);
  }
}

__global__ void k_ndt_results(const NdtPairState* __restrict__ states, const int* __restrict__ evals, const NdtTargetRef* __restrict__ targets,
                              const int* __restrict__ tgt_of_src, NdtResultRec* __restrict__ recs, int P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const NdtPairState& st = states[p];
  NdtResultRec r;
  for (int k = 0; k < 16; k++) r.T[k] = st.T[k];
  r.converged = st.converged;
  r.iterations = st.nr_iterations;
  r.evaluations = evals[p];
  r.n = st.n;
  r.n_target = targets[tgt_of_src[p]].meta->n_pts;
  r.done = st.request == NDT_REQ_NONE;
  r.pad0 = r.pad1 = 0;
  r.score = st.score;
  r.trans_probability = st.trans_probability;
  for (int k = 0; k < 36; k++) r.hess[k] = st.hess[k];
  recs[p] = r;
}

// Registers P sources (src: [P][src_pitch] points in any order, counts on the device, none above max_count) against the targets
// tgt_of_src names.  Blocks until every pair is done (the host polls the number of unfinished pairs every few rounds); the results
// stay in the solver (ndt_solver_results).
void ndt_solve(hipStream_t s, NdtTargets& t, NdtSolver& sv, int P, const float4* d_src, size_t src_pitch, const int* d_src_counts, int max_count,
               const int* tgt_of_src_host, const float* guesses_host, const mi355_ndt_params& p) {
  const NdtConsts c = ndt_consts(p);
  if (!p.legacy_mt_skip && !t.has_d) throw ArgError(MI355_ESTATE, "ndt: legacy_mt_skip = 0 needs targets built with their fp64 leaves");
  sv.P = P;
  sv.h_map.resize((size_t)P);
  sv.h_guess.resize((size_t)P * 16);
  for (int i = 0; i < P; i++) {
    const int u = tgt_of_src_host ? tgt_of_src_host[i] : i;
    if (u < 0 || u >= t.U) throw ArgError(MI355_EINVAL, "tgt_of_src entry out of range");
    sv.h_map[(size_t)i] = u;
    for (int k = 0; k < 16; k++) sv.h_guess[(size_t)i * 16 + k] = guesses_host ? guesses_host[(size_t)i * 16 + k] : ((k % 5 == 0) ? 1.f : 0.f);
  }
  // blocks per pair: enough blocks over the batch to fill the chip, several points per thread once there are (the 28-value block
  // reduction costs about as much as two points)
  const int wpb = kNdtBlock / kWave;
  const int Gcap = std::max(16 / wpb, std::min(4096 / wpb, tuning().ndt_waves / wpb / std::max(P, 1)));
  const int G = std::max(1, std::min(Gcap, div_up(std::max(max_count, 1), kNdtBlock)));
  sv.states.ensure((size_t)P);
  sv.evals.ensure((size_t)P);
  sv.tgt_of_src.ensure((size_t)P);
  sv.guesses.ensure((size_t)P * 16);
  sv.recs.ensure((size_t)P);
  sv.h_recs.ensure((size_t)P);
  sv.n_active.ensure(1);
  sv.h_active.ensure(1);
  sv.partials.ensure((size_t)P * G * kNdtPartStride);
  MI_HIP(hipMemcpyAsync(sv.tgt_of_src.p, sv.h_map.data(), (size_t)P * sizeof(int), hipMemcpyHostToDevice, s));
  MI_HIP(hipMemcpyAsync(sv.guesses.p, sv.h_guess.data(), (size_t)P * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_ndt_begin, dim3(div_up(P, 64)), dim3(64), 0, s, sv.states.p, sv.evals.p, sv.guesses.p, d_src_counts, P);
  // every request is one round; a Newton iteration is 1 round with the legacy line search (the default), up to 12 without
  const int per_iter = p.legacy_mt_skip ? 1 : (kNdtMaxStepIterations + 2);
  const long max_rounds = 1 + (long)(p.max_iterations + 2) * per_iter;
  const int chunk = 4;
  const dim3 grid(xcd_grid(G, P));
  sv.rounds = 0;
  for (long r = 0; r < max_rounds;) {
    MI_HIP(hipMemsetAsync(sv.n_active.p, 0, sizeof(int), s));
    for (int k = 0; k < chunk; k++, r++) {
      {
        TimedLaunch tl(KF_NDT_DERIV, s);
        hipLaunchKernelGGL(k_ndt_derivatives, grid, dim3(kNdtBlock), 0, s, d_src, src_pitch, t.refs.p, sv.tgt_of_src.p, sv.states.p, c, P, G, sv.partials.p);
        if (!p.legacy_mt_skip)  // computeHessian is only ever requested after a refined line search
          hipLaunchKernelGGL(k_ndt_hessian64, grid, dim3(kNdtBlock), 0, s, d_src, src_pitch, t.refs.p, sv.tgt_of_src.p, sv.states.p, c, P, G, sv.partials.p);
      }
      TimedLaunch tl(KF_NDT_ADVANCE, s);
      hipLaunchKernelGGL(k_ndt_advance, dim3(P), dim3(64), 0, s, sv.states.p, sv.evals.p, sv.partials.p, c, P, G, (k == chunk - 1) ? sv.n_active.p : nullptr);
    }
    MI_LAUNCH_CHECK();
    sv.rounds = (int)r;
    MI_HIP(hipMemcpyAsync(sv.h_active.p, sv.n_active.p, sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (*sv.h_active.p == 0) break;
  }
  hipLaunchKernelGGL(k_ndt_results, dim3(div_up(P, 64)), dim3(64), 0, s, sv.states.p, sv.evals.p, t.refs.p, sv.tgt_of_src.p, sv.recs.p, P);
  MI_LAUNCH_CHECK();
  MI_HIP(hipMemcpyAsync(sv.h_recs.p, sv.recs.p, (size_t)P * sizeof(NdtResultRec), hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void ndt_solver_results(NdtSolver& sv, mi355_result* out, float* T16_all) {
  for (int i = 0; i < sv.P; i++) {
    const NdtResultRec& r = sv.h_recs.p[i];
    if (out) {
      mi355_result& o = out[i];
      memset(&o, 0, sizeof(o));
      memcpy(o.T, r.T, sizeof(r.T));
      // a pair cut off by the round budget (it cannot happen within max_iterations) reports what it has, not converged
      o.converged = (r.converged && r.done) ? 1 : 0;
      o.iterations = r.iterations;
      o.error = r.score;
      memcpy(o.H, r.hess, sizeof(r.hess));
      o.trans_probability = r.trans_probability;
      o.n_source = r.n;
      o.n_target = r.n_target;
      o.evaluations = r.evaluations;
    }
    if (T16_all) memcpy(T16_all + (size_t)i * 16, r.T, 16 * sizeof(float));
  }
}

// ---- parity probe: computeDerivatives of ONE pair at a given pose, sums read back -------------------------------------------------
__global__ void k_ndt_probe_sum(const double* __restrict__ partials, int nb, double* __restrict__ sums) {
  const int lane = threadIdx.x;
  if (lane >= kNdtAcc) return;
  double s = 0;
  for (int g = 0; g < nb; g++) s += partials[(size_t)g * kNdtPartStride + lane];
  sums[lane] = s;
}

void ndt_derivatives(hipStream_t s, NdtTargets& t, NdtSolver& sv, const float4* d_src, size_t n, const mi355_ndt_params& p, const double* p6,
                     double* score, double* grad6, double* hess36, bool hessian_fp64) {
  if (hessian_fp64 && !t.has_d) throw ArgError(MI355_ESTATE, "ndt_derivatives: the fp64 leaves of this target were not kept");
  const NdtConsts c = ndt_consts(p);
  const int G = std::max(1, std::min(1024, div_up((long)std::max<size_t>(n, 1), kNdtBlock)));
  sv.states.ensure(1);
  sv.tgt_of_src.ensure(1);
  sv.partials.ensure((size_t)G * kNdtPartStride);
  DevBuf<double> sums;
  sums.ensure(32);
  // the pose's matrix and angle tables are host-made here (the state machine makes them on the device)
  NdtPairState st;
  memset(&st, 0, sizeof(st));
  st.n = (int)n;
  ndt_pose_to_matrix_f(p6, st.T);
  ndt_angle_tables(p6, st.ang);
  const int zero = 0;
  double h_sums[kNdtAcc];
  auto run = [&](int request) {
    st.request = request;
    MI_HIP(hipMemcpyAsync(sv.states.p, &st, sizeof(st), hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(sv.tgt_of_src.p, &zero, sizeof(int), hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
    if (request == NDT_REQ_HESS64)
      hipLaunchKernelGGL(k_ndt_hessian64, dim3(xcd_grid(G, 1)), dim3(kNdtBlock), 0, s, d_src, n, t.refs.p, sv.tgt_of_src.p, sv.states.p, c, 1, G, sv.partials.p);
    else
      hipLaunchKernelGGL(k_ndt_derivatives, dim3(xcd_grid(G, 1)), dim3(kNdtBlock), 0, s, d_src, n, t.refs.p, sv.tgt_of_src.p, sv.states.p, c, 1, G, sv.partials.p);
    const int nb = std::max(1, std::min(G, div_up((long)n, kNdtBlock)));
    hipLaunchKernelGGL(k_ndt_probe_sum, dim3(1), dim3(64), 0, s, sv.partials.p, nb, sums.p);
    MI_LAUNCH_CHECK();
    MI_HIP(hipMemcpyAsync(h_sums, sums.p, sizeof(h_sums), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
  };
  run(NDT_REQ_DERIV_HESS);
  *score = h_sums[0];
  for (int k = 0; k < 6; k++) grad6[k] = h_sums[1 + k];
  ndt_unpack_hessian(h_sums, hess36);
  if (hessian_fp64) {
    run(NDT_REQ_HESS64);
    ndt_unpack_hessian(h_sums, hess36);
  }
}

}  // namespace mi355
