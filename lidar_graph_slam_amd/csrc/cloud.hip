// Cloud-set kernels: the PCL-filter replacements (distance filter, VoxelGrid, StatisticalOutlierRemoval), the NN
// voxel-hash build, exact k-NN and the fast_gicp covariance estimation.  SURVEY.md §8a rows A1-A4, P1-P4, F1, F2.
#include "engine.hpp"
#include "grid.hpp"
#include "primitives.hpp"
#include "tile.hpp"

namespace mi355 {

void CloudSet::alloc(int B_, size_t cap_) {
  B = B_;
  cap = cap_;
  // The table of a cloud is sized on the device to 4 x its occupied cells (k_grid_hash_size): ~0.25 entries per point for a voxelised
  // scan.  Its share of the allocation holds 2 - 4 slots per point of capacity (until round 5: 8 - 16, the worst case of 5 levels
  // with every point alone in its cell at each of them - 8 MB per 65 536-point slot, half of a batch's footprint, for 0.5 MB used);
  // a cloud whose cells do not fit (points tens of metres apart) keeps as many of the coarse levels as do.
  size_t s = 16;
  while (s < 2 * cap) s <<= 1;
  slots_cap = s;
  pts.ensure((size_t)B * cap);
  orig.ensure((size_t)B * cap);
  count.ensure(B);
  meta.ensure(B);
  table.ensure((size_t)B * slots_cap);
  built = false;
  has_cov = false;
  has_order = false;
}

// Threads per block of the one-thread-per-point kernels without block-level cooperation (keys, heads, centroids, cell counts,
// gather + insert).  ONE WAVE per block: a 256-thread block waits for four free wave slots on one CU and gives them back together
// when its slowest wave is done, and these waves live ~2 us — measured on k_voxel_centroids at 1 024 clouds (PMC: 5.4 resident
// waves per CU of 32 at 256 threads): 3.5 ms per launch beside the other chain with 256- and 512-thread blocks, 3.9 ms with
// 1 024, 1.74 ms with 64 (alone: 1.77 / 1.69 ms at 256 / 64).
constexpr int kPointBlock = 64;
constexpr int kHashBlock = 64;  // (the hash build alone: 1.46 ms per launch at 256 threads, 1.59 at 64; beside the other chain 2.69 / 2.35 ms)

void ws_reserve(Workspace& ws, int B, size_t cap) {
  const size_t n = (size_t)B * cap;
  ws.keys_a.ensure(n); ws.keys_b.ensure(n);
  ws.vals_a.ensure(n); ws.vals_b.ensure(n);
  ws.flags.ensure(n);
  ws.scan_local.ensure(n);
  ws.scan_block.ensure((size_t)B * div_up((long)cap, kScanTile));
  ws.totals.ensure(B);
  ws.bounds.ensure((size_t)B * 6);
  ws.vmeta.ensure(B);
  ws.sort_passes.ensure(B);
}

// PointsPreFiltering::distance_filter's predicate (points_prefiltering.cpp:102-112): Eigen Vector3f::norm() — fp32 sqrt of
// x*x + (y*y + z*z) (the association of Eigen's unrolled 3-element redux, Redux.h redux_novec_unroller) — promoted to double, strict '<' (NaN compares false; min_distance = 0 still drops the all-zero no-return
// points).  A NEGATIVE min_distance means "no filter" (the voxel grid without a fused distance filter).
// ---- bounding boxes ---------------------------------------------------------------------------------------------
__global__ void k_bounds_init(uint32_t* bounds, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 6) bounds[i] = ((i % 6) < 3) ? 0xFFFFFFFFu : 0u;
}

__global__ __launch_bounds__(256) void k_bounds(const float4* __restrict__ pts, const int* __restrict__ counts, uint32_t* __restrict__ bounds,
                                                size_t cap, double min_distance) {
  const int c = blockIdx.y;
  const int n = counts[c];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  const float4* p = pts + (size_t)c * cap;
  // four independent loads per thread and trip (one load in flight per thread left this streaming reduction at 2 TB/s)
  constexpr int kU = 4;
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += kU * stride) {
    float4 v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const int i = i0 + u * stride;
      v[u] = (i < n) ? p[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (isfinite(v[u].x) && isfinite(v[u].y) && isfinite(v[u].z) && passes_distance(v[u], min_distance)) {
        mn[0] = fminf(mn[0], v[u].x); mn[1] = fminf(mn[1], v[u].y); mn[2] = fminf(mn[2], v[u].z);
        mx[0] = fmaxf(mx[0], v[u].x); mx[1] = fmaxf(mx[1], v[u].y); mx[2] = fmaxf(mx[2], v[u].z);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 3; d++)
    for (int off = 32; off > 0; off >>= 1) {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], off, kWave));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off, kWave));
    }
  __shared__ float red[4][6];
  if (lane_id() == 0)
    for (int d = 0; d < 3; d++) { red[wave_id()][d] = mn[d]; red[wave_id()][3 + d] = mx[d]; }
  __syncthreads();
  if (threadIdx.x < 6) {  // one atomic per block and component
    const int d = threadIdx.x;
    float v = red[0][d];
    for (int w = 1; w < 4; w++) v = (d < 3) ? fminf(v, red[w][d]) : fmaxf(v, red[w][d]);
    if (d < 3) { if (v < INFINITY) atomicMin(&bounds[c * 6 + d], float_to_ordered(v)); }
    else { if (v > -INFINITY) atomicMax(&bounds[c * 6 + d], float_to_ordered(v)); }
  }
}

static void compute_bounds(hipStream_t s, Workspace& ws, const float4* in, const int* cnt, int B, size_t cap, double min_distance = -1.0) {
  hipLaunchKernelGGL(k_bounds_init, dim3(div_up(B * 6, 256)), dim3(256), 0, s, ws.bounds.p, B);
  const int gx = std::max(1, std::min(16, div_up((long)cap, 4096)));
  hipLaunchKernelGGL(k_bounds, dim3(gx, B), dim3(256), 0, s, in, cnt, ws.bounds.p, cap, min_distance);
}

// ---- NN voxel-hash build ----------------------------------------------------------------------------------------
__global__ void k_grid_meta(const uint32_t* __restrict__ bounds, const int* __restrict__ counts, GridMeta* __restrict__ meta, int B, float cell,
                            int* __restrict__ sort_passes) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B) return;
  GridMeta m;
  const int n = counts[c];
  m.n = n;
  m.overflow = 0;
  float mn[3], mx[3];
  for (int d = 0; d < 3; d++) { mn[d] = ordered_to_float(bounds[c * 6 + d]); mx[d] = ordered_to_float(bounds[c * 6 + 3 + d]); }
  if (n <= 0 || !(mn[0] <= mx[0])) { mn[0] = mn[1] = mn[2] = 0.f; mx[0] = mx[1] = mx[2] = 0.f; }
  float cs = cell;
  int dx, dy, dz;
  for (;;) {
    const float inv = 1.0f / cs;
    dx = (int)floorf((mx[0] - mn[0]) * inv) + 1;
    dy = (int)floorf((mx[1] - mn[1]) * inv) + 1;
    dz = (int)floorf((mx[2] - mn[2]) * inv) + 1;
    if (dx <= 512 && dy <= 512 && dz <= 512) break;  // 9 Morton bits per axis
    cs *= 2.0f;                                       // larger extents: coarsen the base cell
  }
  m.ox = mn[0]; m.oy = mn[1]; m.oz = mn[2];
  m.cell = cs;
  m.inv_cell = 1.0f / cs;
  m.dx = dx; m.dy = dy; m.dz = dz;
  int top = 0;
  while (top < kMaxLevel && (((max(dx, max(dy, dz)) - 1) >> top) > 0)) top++;
  m.top_level = top;
  m.n_cells = 0;
  for (int l = 0; l <= kMaxLevel; l++) m.level_cells[l] = 0;
  // the hash size is set by k_grid_hash_size / k_hash_build once the occupied cells have been counted (after the sort)
  m.hshift = 28;
  m.hmask = 15;
  m.pmask = 15;
  meta[c] = m;
  // Morton keys of this cloud are below 2^(3 bits), bits = those of the largest cell coordinate (top_level is capped, this is not):
  // every point has a key, so ceil(3 bits / 8) radix passes order them (the sort's own padding follows the real elements and stays
  // behind them in a stable sort even on equal low bits)
  int bits = 0;
  while (((max(dx, max(dy, dz)) - 1) >> bits) > 0) bits++;
  sort_passes[c] = max(1, (3 * bits + 7) / 8);
}

__device__ __forceinline__ uint32_t cell_key(const GridMeta& m, float x, float y, float z) {
  const int ix = min(max((int)floorf(cell_coord(x, m.ox, m.inv_cell)), 0), m.dx - 1);
  const int iy = min(max((int)floorf(cell_coord(y, m.oy, m.inv_cell)), 0), m.dy - 1);
  const int iz = min(max((int)floorf(cell_coord(z, m.oz, m.inv_cell)), 0), m.dz - 1);
  return morton27((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}

__global__ __launch_bounds__(256) void k_cell_keys(const float4* __restrict__ pts, const GridMeta* __restrict__ meta, uint32_t* __restrict__ keys,
                                                   int* __restrict__ vals, size_t cap) {
  const int c = blockIdx.y;
  const GridMeta m = meta[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  const float4 p = pts[(size_t)c * cap + i];
  keys[(size_t)c * cap + i] = cell_key(m, p.x, p.y, p.z);
  vals[(size_t)c * cap + i] = i;
}

// Occupied cells over all levels of a cloud = entries the hash will hold: one per run of equal (key >> 3l) in the sorted keys.
// The table is then sized to the entries (load factor 0.125 - 0.25: the tile searches probe mostly EMPTY cells, and an
// unsuccessful probe walks to the end of its cluster — 6.6 dependent loads per 64-cell lookup at load factor 0.5), not to the
// worst case of 5 entries per point: a 16 k-point cloud has ~6 k occupied cells — a 0.5 MB table that stays in the XCD's L2
// while the cloud is searched, instead of 4 MB (the clears alone moved 3.2 GB per 1 024-cloud launch).
// (round 4: a few hundred blocks per cloud that walk it, one atomic per block.  The first version launched cap / 64 one-wave blocks per
// cloud - a million blocks per 1 024-cloud launch, four fifths of them past the end of their cloud - and took 0.43 ms to read 54 MB.)
constexpr int kCountBlock = 256, kCountPer = 8;
__global__ __launch_bounds__(kCountBlock) void k_count_cells(const uint32_t* __restrict__ keys, GridMeta* __restrict__ meta, size_t cap) {
  __shared__ int wsum[kCountBlock / kWave][kMaxLevel + 1];
  const int c = blockIdx.y;
  const int n = meta[c].n, top = meta[c].top_level;
  const uint32_t* k = keys + (size_t)c * cap;
  int cnt[kMaxLevel + 1];
#pragma unroll
  for (int l = 0; l <= kMaxLevel; l++) cnt[l] = 0;
  for (int base = blockIdx.x * kCountBlock * kCountPer; base < n; base += gridDim.x * kCountBlock * kCountPer) {
    const int i0 = base + threadIdx.x * kCountPer;
    uint32_t key[kCountPer + 1];
    key[0] = (i0 > 0 && i0 <= n) ? k[i0 - 1] : 0u;
#pragma unroll
    for (int u = 0; u < kCountPer; u++) key[u + 1] = (i0 + u < n) ? k[i0 + u] : 0u;
#pragma unroll
    for (int u = 0; u < kCountPer; u++) {
      const int i = i0 + u;
      if (i < n) {
#pragma unroll
        for (int l = 0; l <= kMaxLevel; l++)
          if (l <= top) cnt[l] += ((i == 0) || ((key[u] >> (3 * l)) != (key[u + 1] >> (3 * l)))) ? 1 : 0;
      }
    }
  }
#pragma unroll
  for (int l = 0; l <= kMaxLevel; l++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt[l] += __shfl_down(cnt[l], off, kWave);
    if (lane_id() == 0) wsum[wave_id()][l] = cnt[l];
  }
  __syncthreads();
  if (threadIdx.x <= kMaxLevel) {
    int total = 0;
    for (int w = 0; w < kCountBlock / kWave; w++) total += wsum[w][threadIdx.x];
    if (total > 0) atomicAdd(&meta[c].level_cells[threadIdx.x], total);
  }
}

// The table of one cloud from its occupied cells per level: which levels it keeps, how many slots it gets.
struct HashSize {
  int top, cells;
  uint32_t hshift, hmask, pmask;
};
__device__ __forceinline__ HashSize grid_hash_size(int top_level, const int* level_cells, uint32_t slots_cap, uint32_t per_cell) {
  // the levels whose cells fit 0.7 of the cloud's share of the table (level 0 always does: at most one cell per point)
  int top = top_level, cells = 0;
  for (int l = 0; l <= top; l++) {
    if (l > 0 && (uint32_t)(cells + level_cells[l]) > (slots_cap / 10u) * 7u) { top = l - 1; break; }
    cells += level_cells[l];
  }
  uint32_t slots = 16, lg = 4;
  while (slots < per_cell * (uint32_t)max(cells, 1) && slots < slots_cap) { slots <<= 1; lg++; }
  return HashSize{top, cells, 32 - lg, slots - 1, min(slots, kProbeBlock) - 1};
}
__global__ void k_grid_hash_size(GridMeta* __restrict__ meta, int B, uint32_t slots_cap, uint32_t per_cell) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B) return;
  const HashSize hs = grid_hash_size(meta[c].top_level, meta[c].level_cells, slots_cap, per_cell);
  meta[c].top_level = hs.top;
  meta[c].n_cells = hs.cells;
  meta[c].hshift = hs.hshift;
  meta[c].hmask = hs.hmask;
  meta[c].pmask = hs.pmask;
}

// (clear and insert of a cloud run on the XCD that searches it later — xcd_item_chunk: the table of a scan, 0.5 MB, is filled in ONE
// L2.  Measured at 1 024 clouds: FETCH_SIZE of k_gather_insert 371 -> 202 MB per launch, the time unchanged; WRITE_SIZE stays at
// 1.5 GB against 0.34 GB of points and indices — ~3 scattered 4-byte stores / CAS per point each reach the fabric as their own
// 32-byte write whichever L2 they come from.)
__global__ __launch_bounds__(256) void k_table_clear(HashSlot* __restrict__ table, const GridMeta* __restrict__ meta, size_t slots_cap, int G, int B) {
  int c, bx;
  if (!xcd_item_chunk(G, B, c, bx)) return;
  const uint32_t slots = meta[c].hmask + 1;
  int4* t = reinterpret_cast<int4*>(table + (size_t)c * slots_cap);
  for (uint32_t i = bx * blockDim.x + threadIdx.x; i < slots; i += G * blockDim.x) t[i] = make_int4((int)kEmptyKey, 0, 0, 0);
}

__device__ __forceinline__ HashSlot* table_insert(HashSlot* t, const GridMeta& m, uint32_t key) {
  uint32_t h = hash_slot(key, m.hshift);
  for (;;) {
    const uint32_t prev = atomicCAS(&t[h].key, kEmptyKey, key);
    if (prev == kEmptyKey || prev == key) return &t[h];
    h = next_probe(h, m.pmask);
  }
}

__device__ __forceinline__ uint32_t hilbert2(uint32_t x, uint32_t y) {  // 11 bits per axis -> 22-bit index along the curve
  uint32_t d = 0;
#pragma unroll
  for (int b = 10; b >= 0; b--) {
    const uint32_t s = 1u << b;
    const uint32_t rx = (x >> b) & 1u, ry = (y >> b) & 1u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0u) {  // rotate the quadrant
      if (rx == 1u) { x = (s - 1u) - (x & (s - 1u)); y = (s - 1u) - (y & (s - 1u)); }
      const uint32_t t = x; x = y; y = t;
    }
    x &= s - 1u; y &= s - 1u;
  }
  return d;
}
// 7 bits per axis -> 21-bit index along the 3-D Hilbert curve (Skilling's transpose form)
__device__ __forceinline__ uint32_t hilbert3(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t X[3] = {x, y, z};
  constexpr uint32_t M = 1u << 6;
#pragma unroll
  for (uint32_t Q = M; Q > 1u; Q >>= 1) {
    const uint32_t P = Q - 1u;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  uint32_t t = 0;
#pragma unroll
  for (uint32_t Q = M; Q > 1u; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1u;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  uint32_t h = 0;
#pragma unroll
  for (int b = 6; b >= 0; b--) h = (h << 3) | (((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u);
  return h;
}
// key of a point in the order the tile searches take the points of a cloud in (build_tile_order below): 2-D Hilbert index over the
// two longest axes of the bounding box (a, b), 2^11 cells over the longer one - a scan is 2.5-D.  A cloud that is not (its shortest
// extent more than half of the middle one: an object, a volume) gets the 3-D curve, 2^7 cells over the longest axis.
__device__ __forceinline__ uint32_t tile_order_key(const GridMeta& m, const float4& p) {
  const int d_min = min(m.dx, min(m.dy, m.dz)), d_max = max(m.dx, max(m.dy, m.dz)), d_mid = m.dx + m.dy + m.dz - d_min - d_max;
  if (2 * d_min > d_mid) {
    const float scale = 127.0f / ((float)max(d_max, 1) * m.cell);
    auto q = [&](float v) { return (uint32_t)min(max((int)(v * scale), 0), 127); };
    return hilbert3(q(p.x - m.ox), q(p.y - m.oy), q(p.z - m.oz));
  }
  const int drop = (m.dz <= m.dx && m.dz <= m.dy) ? 2 : ((m.dy <= m.dx) ? 1 : 0);
  const float ua = (drop == 0) ? p.y - m.oy : p.x - m.ox, ub = (drop == 2) ? p.y - m.oy : p.z - m.oz;
  const int da = (drop == 0) ? m.dy : m.dx, db = (drop == 2) ? m.dy : m.dz;
  const float scale = 2047.0f / ((float)max(max(da, db), 1) * m.cell);
  const uint32_t ia = (uint32_t)min(max((int)(ua * scale), 0), 2047), ib = (uint32_t)min(max((int)(ub * scale), 0), 2047);
  return hilbert2(ia, ib);
}

// (round 4: kGiGroups x 64 consecutive elements per wave, all their loads in flight together - with one element per lane a wave was
// a chain of dependent operations: sorted index -> point gather -> CAS -> slot store, 90 % of its cycles waiting)
#ifndef GI_GROUPS
#define GI_GROUPS 4
#endif
constexpr int kGiGroups = GI_GROUPS;
__global__ __launch_bounds__(256) void k_gather_insert(const float4* __restrict__ in, const uint32_t* __restrict__ keys, const int* __restrict__ vals,
                                                       const GridMeta* __restrict__ meta, float4* __restrict__ pts, int* __restrict__ orig,
                                                       HashSlot* __restrict__ table, size_t cap, size_t slots_cap, int G, int B,
                                                       uint32_t* __restrict__ order_keys, int* __restrict__ order_vals) {
  int c, bx;
  if (!xcd_item_chunk(G, B, c, bx)) return;
  const GridMeta m = meta[c];
  const int lane = lane_id();  // (one-wave blocks: kHashBlock)
  const size_t o = (size_t)c * cap;
  HashSlot* tab = table + (size_t)c * slots_cap;
  // (G blocks per cloud walk its chunks: one block per chunk of the CAPACITY was 262 k blocks per 1 024-cloud launch, four fifths of them
  // past the end of a voxelised cloud)
  for (int chunk = bx; chunk * (kWave * kGiGroups) < m.n; chunk += G) {
  const int i0 = chunk * (kWave * kGiGroups) + lane;
  uint32_t key_[kGiGroups], kprev_[kGiGroups], knext_[kGiGroups];
  int v_[kGiGroups];
  float4 pv_[kGiGroups];
#pragma unroll
  for (int g = 0; g < kGiGroups; g++) {
    const int i = i0 + g * kWave;
    const bool valid = i < m.n;
    v_[g] = valid ? vals[o + i] : 0;
    key_[g] = valid ? keys[o + i] : 0u;
    kprev_[g] = (valid && i > 0) ? keys[o + i - 1] : 0u;
    knext_[g] = (valid && i < m.n - 1) ? keys[o + i + 1] : 0u;
  }
#pragma unroll
  for (int g = 0; g < kGiGroups; g++) pv_[g] = (i0 + g * kWave < m.n) ? in[o + v_[g]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int g = 0; g < kGiGroups; g++) {
    const int i = i0 + g * kWave;
    const bool valid = i < m.n;
    if (valid) {
      pts[o + i] = pv_[g];
      orig[o + i] = v_[g];
      if (order_keys) {  // the (key, index) pairs build_tile_order sorts next, while the point is in a register
        order_keys[o + i] = tile_order_key(m, pv_[g]);
        order_vals[o + i] = i;
      }
    }
    const uint32_t key = key_[g], kprev = kprev_[g], knext = knext_[g];
    for (int l = 0; l <= m.top_level; l++) {  // one (start, end) entry per occupied cell of every level
      const bool head = valid && ((i == 0) || ((kprev >> (3 * l)) != (key >> (3 * l))));
      const bool tail = valid && ((i == m.n - 1) || ((knext >> (3 * l)) != (key >> (3 * l))));
      // A run that lies inside the group is entered by its head alone - one CAS and one 16-byte store instead of two of each (the
      // scattered 4-byte stores and CAS of this kernel reach the fabric as 32-byte writes of their own: 1.5 GB per 1 024-cloud launch
      // against 0.34 GB of points and indices); only a run that crosses a group boundary is entered from both ends.
      const unsigned long long heads = __ballot(head), tails = __ballot(tail);
      const unsigned long long tails_from_here = tails & ~((1ull << lane) - 1ull), heads_up_to_here = heads & ((2ull << lane) - 1ull);
      const bool tail_in_wave = tails_from_here != 0ull, head_in_wave = heads_up_to_here != 0ull;
      if (head) {
        HashSlot* sl = table_insert(tab, m, level_key(key, l));
        if (tail_in_wave) {
          const int t = __builtin_ctzll(tails_from_here);
          *reinterpret_cast<int4*>(sl) = make_int4((int)level_key(key, l), i, i + (t - lane) + 1, 0);
        } else {
          sl->start = i;
        }
      } else if (tail && !head_in_wave) {
        table_insert(tab, m, level_key(key, l))->end = i + 1;
      }
    }
  }
  }
}

// ---- the voxel hash of a batch of scan-sized clouds: one workgroup per cloud, the table built in LDS (round 6) -----------------------
// k_gather_insert above enters every cell with a CAS and scattered 4- / 16-byte stores into a table in HBM: per 2 048-cloud launch 2.5 GB
// written for 0.9 GB of points, indices and slots, 1.66 ms (+ 0.17 ms clearing the tables first, + the cell count and its atomics).
// Here ONE block owns a cloud from the sorted (cell key, index) pairs to the finished table:
//   pass A  gathers the points into Morton order (and leaves the keys of build_tile_order's sort), counting the occupied cells per level
//           on the way; thread 0 sizes the table (grid_hash_size);
//   then, per probe block of the table (kProbeBlock = 8 192 slots = 128 KB of LDS; a voxelised scan's table is four of them): clear the
//   LDS image, walk the sorted keys once more - a run of equal (key >> 3l) is a cell of level l: its first element enters `start`, its
//   last `end` - inserting with LDS atomics only the cells whose first slot lies in this probe block (probe sequences wrap inside their
//   block: next_probe), and write the image out with coalesced 16-byte stores.  HBM sees every slot written exactly once.
// A cloud costs (probe blocks) x (its points) key tests, so this is the path of BATCHES of scan-sized clouds (hash_build_in_blocks);
// lone and large clouds keep k_gather_insert.
constexpr int kHbThreads = 1024, kHbKeep = 16;  // a cloud of <= kHbKeep x 1 024 points keeps its keys in registers between the passes
__global__ __launch_bounds__(kHbThreads) void k_hash_build(const float4* __restrict__ in, const uint32_t* __restrict__ keys, const int* __restrict__ vals,
                                                           GridMeta* __restrict__ meta, float4* __restrict__ pts, int* __restrict__ orig,
                                                           HashSlot* __restrict__ table, size_t cap, size_t slots_cap, uint32_t per_cell,
                                                           uint32_t* __restrict__ order_keys, int* __restrict__ order_vals) {
  __shared__ int4 image[kProbeBlock];
  __shared__ int wcnt[kHbThreads / kWave][kMaxLevel + 1];
  __shared__ HashSize s_hs;
  const int c = blockIdx.x;
  const GridMeta m = meta[c];
  const int n = m.n;
  const size_t o = (size_t)c * cap;
  const uint32_t* k = keys + o;
  const int tid = threadIdx.x;
  // levels at which an element starts / ends a run of equal (key >> 3l), from the keys of its neighbours: all l <= msb(difference) / 3
  auto run_level = [](uint32_t diff) { return diff ? min((31 - __clz((int)diff)) / 3, kMaxLevel) : -1; };
  // ---- pass A: gather + occupied cells per level
  int cnt[kMaxLevel + 1];
#pragma unroll
  for (int l = 0; l <= kMaxLevel; l++) cnt[l] = 0;
  const bool keep = n <= kHbKeep * kHbThreads;
  uint32_t key_[kHbKeep];  // keep: element tid + 1 024 j -> its key and (head level + 1) | (tail level + 1) << 4
  uint32_t lev_[kHbKeep];
#pragma unroll
  for (int j = 0; j < kHbKeep; j++) { key_[j] = 0u; lev_[j] = 0u; }
  constexpr int kU = 4;  // elements per thread and trip: their index loads, then their point gathers, in flight together
  auto pass_a = [&](int i0, auto&& remember) {
    int v[kU];
    uint32_t kc[kU], kp[kU], kn[kU];
    float4 pv[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const int i = i0 + u * kHbThreads;
      const bool valid = i < n;
      v[u] = valid ? vals[o + i] : 0;
      kc[u] = valid ? k[i] : 0u;
      kp[u] = (valid && i > 0) ? k[i - 1] : 0u;
      kn[u] = (valid && i < n - 1) ? k[i + 1] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kU; u++) pv[u] = (i0 + u * kHbThreads < n) ? in[o + v[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const int i = i0 + u * kHbThreads;
      if (i >= n) continue;
      pts[o + i] = pv[u];
      orig[o + i] = v[u];
      if (order_keys) {
        order_keys[o + i] = tile_order_key(m, pv[u]);
        order_vals[o + i] = i;
      }
      const int lh = (i > 0) ? run_level(kp[u] ^ kc[u]) : kMaxLevel, lt = (i < n - 1) ? run_level(kn[u] ^ kc[u]) : kMaxLevel;
#pragma unroll
      for (int l = 0; l <= kMaxLevel; l++)
        if (l <= m.top_level) cnt[l] += (l <= lh) ? 1 : 0;
      remember(u, kc[u], (uint32_t)(lh + 1) | ((uint32_t)(lt + 1) << 4));
    }
  };
  if (keep) {
#pragma unroll
    for (int jb = 0; jb < kHbKeep / kU; jb++)
      if (tid + jb * kU * kHbThreads < n)
        pass_a(tid + jb * kU * kHbThreads, [&](int u, uint32_t key, uint32_t lev) {
#pragma unroll
          for (int uu = 0; uu < kU; uu++)
            if (uu == u) { key_[jb * kU + uu] = key; lev_[jb * kU + uu] = lev; }
        });
  } else {
    for (int i0 = tid; i0 < n; i0 += kU * kHbThreads) pass_a(i0, [](int, uint32_t, uint32_t) {});
  }
#pragma unroll
  for (int l = 0; l <= kMaxLevel; l++) {
    const int t = wave_reduce64(cnt[l], [](int a, int b) { return a + b; });
    if (lane_id() == 0) wcnt[wave_id()][l] = t;
  }
  __syncthreads();
  if (tid == 0) {
    int level_cells[kMaxLevel + 1];
    for (int l = 0; l <= kMaxLevel; l++) {
      int t = 0;
      for (int w = 0; w < kHbThreads / kWave; w++) t += wcnt[w][l];
      level_cells[l] = t;
      meta[c].level_cells[l] = t;
    }
    const HashSize hs = grid_hash_size(m.top_level, level_cells, (uint32_t)slots_cap, per_cell);
    meta[c].top_level = hs.top;
    meta[c].n_cells = hs.cells;
    meta[c].hshift = hs.hshift;
    meta[c].hmask = hs.hmask;
    meta[c].pmask = hs.pmask;
    s_hs = hs;
  }
  __syncthreads();
  const HashSize hs = s_hs;
  // ---- the table, one probe block at a time
  const uint32_t slots = hs.hmask + 1u, psize = hs.pmask + 1u, pieces = slots / psize;
  int4* tab = reinterpret_cast<int4*>(table + (size_t)c * slots_cap);
  // element i: key, run levels lh / lt (-1: none; capped at the levels the table keeps) -> its cells of probe block q
  auto enter = [&](uint32_t q, int i, uint32_t key, int lh, int lt) {
    lh = min(lh, hs.top); lt = min(lt, hs.top);
    for (int l = 0; l <= max(lh, lt); l++) {
      const uint32_t K = level_key(key, l);
      uint32_t h = hash_slot(K, hs.hshift);
      if ((h >> kProbeBlockLog) != q) continue;
      h &= hs.pmask;
      for (;;) {
        const uint32_t prev = atomicCAS(reinterpret_cast<uint32_t*>(&image[h].x), kEmptyKey, K);
        if (prev == kEmptyKey || prev == K) break;
        h = (h + 1u) & hs.pmask;
      }
      if (l <= lh) image[h].y = i;
      if (l <= lt) image[h].z = i + 1;
    }
  };
  for (uint32_t q = 0; q < pieces; q++) {
    for (uint32_t z = tid; z < psize; z += kHbThreads) image[z] = make_int4((int)kEmptyKey, 0, 0, 0);
    __syncthreads();
    if (keep) {
#pragma unroll
      for (int j = 0; j < kHbKeep; j++)
        if (lev_[j]) enter(q, tid + j * kHbThreads, key_[j], (int)(lev_[j] & 15u) - 1, (int)(lev_[j] >> 4) - 1);
    } else {
      for (int i = tid; i < n; i += kHbThreads) {
        const uint32_t key = k[i];
        enter(q, i, key, (i > 0) ? run_level(k[i - 1] ^ key) : kMaxLevel, (i < n - 1) ? run_level(k[i + 1] ^ key) : kMaxLevel);
      }
    }
    __syncthreads();
    for (uint32_t z = tid; z < psize; z += kHbThreads) tab[(size_t)q * psize + z] = image[z];
    __syncthreads();
  }
}
// which batches take k_hash_build: many clouds (a block per cloud must fill the chip) of at most 2^16 points (<= 16 probe blocks)
static bool hash_build_in_blocks(int B, size_t cap) {
  static const int mode = [] { const char* e = getenv("MI355_HASH_BLOCKS"); return e ? atoi(e) : -1; }();  // A/B aid: 0 = never, 1 = always
  if (mode >= 0) return mode != 0 && cap <= 65536;
  return B >= 64 && cap <= 65536;
}

__global__ void k_copy_counts(const int* in, int* out, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = in[i];
}

void build_grid(hipStream_t s, Workspace& ws, const float4* in, const int* cnt_in, CloudSet& out, float cell, bool tile_order_keys) {
  const int B = out.B;
  const size_t cap = out.cap;
  ws_reserve(ws, B, cap);
  const dim3 grid(div_up((long)cap, kHashBlock), B);
  if (out.cell_keys_in_ws) {  // voxel_grid(..., index_next = &out) on this workspace has left meta, keys and sort passes behind
    out.cell_keys_in_ws = false;
  } else {
    compute_bounds(s, ws, in, cnt_in, B, cap);
    hipLaunchKernelGGL(k_grid_meta, dim3(div_up(B, 64)), dim3(64), 0, s, ws.bounds.p, cnt_in, out.meta.p, B, cell, ws.sort_passes.p);
    hipLaunchKernelGGL(k_cell_keys, grid, dim3(kHashBlock), 0, s, in, out.meta.p, ws.keys_a.p, ws.vals_a.p, cap);
  }
  hipLaunchKernelGGL(k_copy_counts, dim3(div_up(B, 256)), dim3(256), 0, s, cnt_in, out.count.p, B);
  sort_pairs(s, ws, B, cap, cnt_in, 4, ws.sort_passes.p);
  TimedLaunch tl(KF_HASH_BUILD, s);
  const bool okeys = tile_order_keys && tuning().knn_tiles == 1 && tuning().tile_order;
  if (hash_build_in_blocks(B, cap)) {
    hipLaunchKernelGGL(k_hash_build, dim3(B), dim3(kHbThreads), 0, s, in, ws.keys_a.p, ws.vals_a.p, out.meta.p, out.pts.p, out.orig.p, out.table.p, cap,
                       out.slots_cap, (uint32_t)std::max(1, tuning().hash_slots_per_cell), okeys ? ws.keys_b.p : (uint32_t*)nullptr,
                       okeys ? ws.vals_b.p : (int*)nullptr);
  } else {
    hipLaunchKernelGGL(k_count_cells, dim3(std::max(1, std::min(std::max(8, 1024 / std::max(B, 1)), div_up((long)cap, kCountBlock * kCountPer))), B), dim3(kCountBlock), 0, s, ws.keys_a.p, out.meta.p, cap);
    hipLaunchKernelGGL(k_grid_hash_size, dim3(div_up(B, 64)), dim3(64), 0, s, out.meta.p, B, (uint32_t)out.slots_cap, (uint32_t)std::max(1, tuning().hash_slots_per_cell));
    const int Gc = std::max(1, std::min(32, (int)(out.slots_cap / 8192))), Gi = std::max(1, std::min(std::max(64, 8192 / std::max(B, 1)), div_up((long)cap, kHashBlock * kGiGroups)));  // (a lone big cloud: thousands of blocks)
    hipLaunchKernelGGL(k_table_clear, dim3(xcd_grid(Gc, B)), dim3(256), 0, s, out.table.p, out.meta.p, out.slots_cap, Gc, B);
    hipLaunchKernelGGL(k_gather_insert, dim3(xcd_grid(Gi, B)), dim3(kHashBlock), 0, s, in, ws.keys_a.p, ws.vals_a.p, out.meta.p, out.pts.p, out.orig.p,
                       out.table.p, cap, out.slots_cap, Gi, B, okeys ? ws.keys_b.p : (uint32_t*)nullptr, okeys ? ws.vals_b.p : (int*)nullptr);
  }
  MI_LAUNCH_CHECK();
  if (okeys) {  // sort_pairs reads the *_a buffers
    std::swap(ws.keys_a.p, ws.keys_b.p); std::swap(ws.keys_a.n, ws.keys_b.n);
    std::swap(ws.vals_a.p, ws.vals_b.p); std::swap(ws.vals_a.n, ws.vals_b.n);
  }
  out.built = true;
  out.has_cov = false;
  out.has_order = false;
  out.order_keys_in_ws = okeys;
}

// ---- k-NN covariances (F2) --------------------------------------------------------------------------------------
// next point at exactly tau2 whose caller index is the smallest one greater than `last` (tie resolution, rare path)
struct NextEq {
  static constexpr int kScanWidth = 1;
  const int* orig; float tau2; int last; int best_j; int best_o;
  __device__ __forceinline__ float bound() const { return tau2; }
  __device__ __forceinline__ float complete() const { return tau2; }
  __device__ __forceinline__ void flush() {}
  __device__ __forceinline__ void restart() {}
  __device__ __forceinline__ void visit(int j, float dd) {
    if (dd != tau2) return;
    const int oj = orig[j];
    if (oj > last && (best_j < 0 || oj < best_o)) { best_j = j; best_o = oj; }
  }
};

// Step 1: exact k-NN of every point in its own cloud -> neighbour lists (sorted-array indices, k ints per point, in no
// particular order — the covariance only needs the set).
//   pass A  distances of the k nearest by the branch-free median network (NearestKD) over the escalating shell search;
//   pass B  one ball query with the exact k-th distance collects the neighbour indices (ties at the k-th distance:
//           lowest caller index first, like the oracle).
template <int K>
__global__ __launch_bounds__(256) void k_knn_self(const float4* __restrict__ pts, const int* __restrict__ orig, const HashSlot* __restrict__ table,
                                                  const GridMeta* __restrict__ meta, int* __restrict__ nbr, size_t cap, size_t slots_cap, int k, int G,
                                                  int B, int need_div, const int* __restrict__ queue, const int* __restrict__ queue_n) {
  int c, bx;
  if (!xcd_item_chunk(G, B, c, bx)) return;
  const GridView g = grid_view(pts, orig, table, meta, c, cap, slots_cap);
  int i = bx * blockDim.x + threadIdx.x;
  if (queue) {  // only the points the tile search handed over
    if (i >= queue_n[c]) return;
    i = queue[(size_t)c * cap + i];
  }
  if (i >= g.m.n) return;
  const float4 q = g.pts[i];
  int* o = nbr + ((size_t)c * cap + i) * k;
  float tau2;
  {
    NearestKD<K> nk;
    nk.init(k);
    grid_search(g, q.x, q.y, q.z, INFINITY, nk, knn_start_level(g, q.x, q.y, q.z, k, need_div), 1);
    tau2 = nk.kth();
  }
  if (!(tau2 < INFINITY)) {  // fewer than k points in the whole cloud (the API rejects this; keep the lists defined)
    for (int j = 0; j < k; j++) o[j] = -1;
    return;
  }
  BallCollector bc{g.orig, o, k, tau2, 0, 0, -1};
  grid_ball_search(g, q.x, q.y, q.z, bc, 1.0f);
  if (bc.n_lt + bc.n_eq == k && bc.n_eq == 1) {
    o[k - 1] = bc.eq_first;  // the common case: exactly one point at the k-th distance
    return;
  }
  // ties at the k-th distance: take the (k - n_lt) lowest caller indices among the points at tau2 (rare, slow path)
  int filled = min(bc.n_lt, k);
  int last = -1;
  while (filled < k) {
    // next point at tau2 whose caller index is the smallest one greater than `last`
    NextEq ne{g.orig, tau2, last, -1, 0x7fffffff};
    grid_ball_search(g, q.x, q.y, q.z, ne, 1.0f);
    if (ne.best_j < 0) break;
    o[filled++] = ne.best_j;
    last = ne.best_o;
  }
  for (; filled < k; filled++) o[filled] = -1;
}

// Neighbourhood covariance in fp64 (divide by k, not k-1) + regularisation; point(t) returns the t-th neighbour.
// KU > 0: the two passes over the neighbours are unrolled KU times (guarded by t < k), so a caller that keeps the neighbour
// list in registers indexes it with compile-time constants; KU == 0: plain loops.
template <int KU, typename P>
__device__ __forceinline__ void covariance_of(int k, int method, Cov6* o, P&& point) {
  double mx = 0, my = 0, mz = 0;
  auto add_mean = [&](int t) {
    const float4 p = point(t);
    mx += (double)p.x; my += (double)p.y; mz += (double)p.z;
  };
  if constexpr (KU > 0) {
#pragma unroll
    for (int t = 0; t < KU; t++)
      if (t < k) add_mean(t);
  } else {
    for (int t = 0; t < k; t++) add_mean(t);
  }
  const double invk = 1.0 / (double)k;
  mx *= invk; my *= invk; mz *= invk;
  Sym3<double> cv = {0, 0, 0, 0, 0, 0};
  auto add_cov = [&](int t) {
    const float4 p = point(t);
    const double x = (double)p.x - mx, y = (double)p.y - my, z = (double)p.z - mz;
    cv.xx += x * x; cv.xy += x * y; cv.xz += x * z; cv.yy += y * y; cv.yz += y * z; cv.zz += z * z;
  };
  if constexpr (KU > 0) {
#pragma unroll
    for (int t = 0; t < KU; t++)
      if (t < k) add_cov(t);
  } else {
    for (int t = 0; t < k; t++) add_cov(t);
  }
  cv.xx *= invk; cv.xy *= invk; cv.xz *= invk; cv.yy *= invk; cv.yz *= invk; cv.zz *= invk;
  const Sym3<double> r = regularize_covariance(cv, method);
  *o = cov6_pack((float)r.xx, (float)r.xy, (float)r.xz, (float)r.yy, (float)r.yz, (float)r.zz);
}

// pcl::StatisticalOutlierRemoval's statistic of one point (P2): the k = mean_k + 1 nearest neighbours include the point itself at
// distance 0, which PCL skips (`for k = 1 .. mean_k: dist_sum += sqrt(nn_dists[k])`, double accumulation), so the mean over the
// other mean_k is the sum over all k divided by mean_k.  point(t) returns the t-th neighbour.
template <int KU, typename P>
__device__ __forceinline__ float sor_mean_distance(int k, const float4& q, P&& point) {
  double dist_sum = 0.0;
#pragma unroll
  for (int t = 0; t < KU; t++)
    if (t < k) {
      const float4 p = point(t);
      dist_sum += sqrt((double)dist2f(q.x, q.y, q.z, p.x, p.y, p.z));
    }
  return (float)(dist_sum / (double)(k - 1));
}

// Steps 1+2, fast path: k-NN and covariance of every point by the wave-cooperative tile search (tile.hpp).  One wave =
// 64 consecutive points of the Morton-sorted cloud.  The points of the grown tile box are gathered into LDS once; every
// group of kSubQ consecutive lanes (a SUB-TILE: 16 points that are neighbours in the kd order of the tile) then gets its own list of
// the candidates inside ITS grown bounding box, and a lane ranks only its sub-tile's list through a (k+2)-deep key network
// — well under half of the ~385 candidates of the whole tile for k = 20, which is what bounded the first version of
// this kernel (VALU-bound on 64 lanes x 385 candidates x 23 v_med3).  A key is the squared distance with the candidate's
// position in the tile list in the low kOrdBits mantissa bits, so the k smallest keys ARE the neighbour list and the
// covariance is computed straight from the LDS copies of those points.  The box radius starts from the tile's own point
// density and grows until it covers the k-th distance of every lane; a later round gathers only the new shell of the tile
// box and appends to each open sub-list what entered its own grown box.
// A lane's list is exact when (1) the radius covered around its sub-tile reaches its k-th distance and (2) its k-th and
// (k+1)-th keys differ in the distance bits that are left, i.e. truncation cannot have swapped a neighbour with a
// non-neighbour.  When exactly two candidates share the k-th truncated distance (~1 % of the points) the pair is ranked
// with the full distances; longer runs, lanes whose neighbourhood does not fit the lists and far-field points go to the
// fallback queue and are done by k_knn_hard + k_cov_from_knn.  A tile whose own box is too large (it straddles a jump of
// the Morton curve, or lies in the sparse far field) is split into halves, processed one after the other by the whole wave.
constexpr int kOrdBits = 10;
#ifndef KNN_TILE_PREFETCH
#define KNN_TILE_PREFETCH 1
#endif
constexpr bool kTilePrefetch = KNN_TILE_PREFETCH != 0;  // fetch the next tile under the current one (k_knn_cov_tile)
#ifndef KNN_TILE_CAND  // (KNN_*: overridable for A/B builds, benchmarks/build_variant.sh)
#define KNN_TILE_CAND 768
#define KNN_SUB_CAP 764
#define KNN_TILE_WAVES_PER_SIMD 2
#endif
#ifndef KNN_GATHER_DEPTH
#define KNN_GATHER_DEPTH 8  // 64-point chunk loads of the candidate gather in flight per wave
#endif
#ifndef KNN_COV_REGS
#define KNN_COV_REGS 1  // 0: the covariance epilogue reads its neighbours from LDS point by point (A/B builds)
#endif
#ifndef KNN_RANK_BATCH
#define KNN_RANK_BATCH 1  // 0: the rank loop of round 3 (A/B builds)
#endif
constexpr int kTileMaxCand = KNN_TILE_CAND;  // candidates per tile segment (list positions; at most 2^kOrdBits)
constexpr uint32_t kOrdMask = (1u << kOrdBits) - 1u;
constexpr int kTileMaxLoad = 8192;           // points in the covering cells per round (loads only)
constexpr int kSubQ = 16;                    // lanes (queries) per sub-tile (8: 10.9 + 1.8 ms per launch, 16: 10.7 + 1.7 ms, 32: 11.2 + 1.7 ms, 4: 12.3 + 2.5 ms)
constexpr int kNSub = kWave / kSubQ;
// candidates per sub-tile list.  12 KB tile list + 7.5 KB of sub-lists per wave = 8 waves per CU, the same footprint as
// the first split (1 024 + 8 x 220): measured at 1 024 clouds, lists that overflow are what sends most points to k_knn_hard —
// 220 entries: 2.65 % handed over, tile kernel 11.3 ms + hard 3.9 ms per launch; 476 entries with a 768-entry tile list:
// 1.08 %, 10.9 ms + 1.8 ms (604 / 640: 1.27 %, 11.1 + 2.0; 348 / 1 024: 1.24 %, 12.4 + 2.0).  (Those with 8-lane sub-tiles; the
// 16-lane sub-tiles in the tree have 4 lists of 764: 1.01 %.)
// Occupancy is LDS-bound at 2 waves per SIMD (18 KB per wave).  Measured alternatives that reach 3 waves: a 512-entry tile list
// with 508-entry sub-lists (12 KB): tile kernel 9.9 ms but 1.68 % handed over, k_knn_hard 2.6 ms — the same sum; sub-lists as
// per-sub-tile BITMASKS over the 768-entry tile list (12.7 KB, 168 VGPRs): same hand-over, 11.0 ms (11.3 ms with two candidate
// loads in flight) — walking set bits costs ~10 more VALU instructions per candidate than reading a 16-bit list entry, and the
// rank loop is VALU-bound, so the extra wave does not pay for it.
constexpr int kSubCap = KNN_SUB_CAP;
constexpr int kSubRow = kSubCap + 16;        // row pitch: the software-pipelined rank loop reads (and ignores) up to 15 entries past the end

// min / max over the aligned group of kSubQ lanes a lane belongs to (same value in all lanes of the group)
#if WAVE_REDUCE_DPP
__device__ __forceinline__ float group_min(float v) {
  static_assert(kSubQ == 16, "a sub-tile is one DPP row");
  return row_reduce16(v, [](float a, float b) { return fminf(a, b); });
}
__device__ __forceinline__ float group_max(float v) {
  return row_reduce16(v, [](float a, float b) { return fmaxf(a, b); });
}
#else
__device__ __forceinline__ float group_min(float v) {
#pragma unroll
  for (int off = 1; off < kSubQ; off <<= 1) v = fminf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int off = 1; off < kSubQ; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}
#endif

// Bitonic sort of the wave's lanes by a float key, ascending inside every aligned segment of S lanes.  Only (key, src) travel
// — src = the lane that held the point originally — so a compare-exchange stage costs two lane exchanges, not four.
template <int S>
__device__ __forceinline__ void segment_sort_lanes(float& key, int& src) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= S; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float pk = __shfl_xor(key, j, kWave);
      const int ps = __shfl_xor(src, j, kWave);
      const bool up = ((lane & k) == 0) || (k == S);  // the last merge is ascending in every segment
      const bool lower = (lane & j) == 0;
      // the lower lane of a pair keeps the smaller key in an ascending block; equal keys stay where they are
      const bool take = (lower == up) ? (pk < key) : (pk > key);
      key = take ? pk : key;
      src = take ? ps : src;
    }
  }
}
// Longest axis of the bounding box of the aligned segment of S lanes a lane belongs to (lanes with x = +inf do not count).
template <int S>
__device__ __forceinline__ int segment_longest_axis(float x, float y, float z) {
  const bool ok = x < INFINITY;
  float lx = ok ? x : INFINITY, ly = ok ? y : INFINITY, lz = ok ? z : INFINITY;
  float hx = ok ? x : -INFINITY, hy = ok ? y : -INFINITY, hz = ok ? z : -INFINITY;
#pragma unroll
  for (int off = 1; off < S; off <<= 1) {
    lx = fminf(lx, __shfl_xor(lx, off, kWave)); ly = fminf(ly, __shfl_xor(ly, off, kWave)); lz = fminf(lz, __shfl_xor(lz, off, kWave));
    hx = fmaxf(hx, __shfl_xor(hx, off, kWave)); hy = fmaxf(hy, __shfl_xor(hy, off, kWave)); hz = fmaxf(hz, __shfl_xor(hz, off, kWave));
  }
  const float ex = hx - lx, ey = hy - ly, ez = hz - lz;
  return (ex >= ey && ex >= ez) ? 0 : ((ey >= ez) ? 1 : 2);
}
// kd order of the 64 queries of a tile: split at the median along the longest axis, three levels deep, so that every aligned
// group of 8 / 16 lanes (a sub-tile) and every aligned half / quarter (what a tile is split into when its box is too big) is
// spatially compact.  64 consecutive points of the Morton curve are NOT: the curve jumps, and one ragged group of 8 sets
// the trip count of the whole wave's ranking loop (measured: mean sub-list 77 candidates, longest 182).
__device__ __forceinline__ void kd_order_tile(float& x, float& y, float& z, int& i) {
  const float x0 = x, y0 = y, z0 = z;
  const int i0 = i;
  int src = lane_id();
  auto level = [&](auto seg) {
    constexpr int S = decltype(seg)::value;
    const int axis = segment_longest_axis<S>(x, y, z);
    float key = (axis == 0) ? x : ((axis == 1) ? y : z);
    segment_sort_lanes<S>(key, src);
    x = __shfl(x0, src, kWave); y = __shfl(y0, src, kWave); z = __shfl(z0, src, kWave);
  };
  level(std::integral_constant<int, 64>{});
  level(std::integral_constant<int, 32>{});
  level(std::integral_constant<int, 16>{});
  i = __shfl(i0, src, kWave);
}

// Launch: one-wave blocks, G per cloud (all on the cloud's XCD); a wave takes tiles from the cloud's counter until none
// is left, so the waves of a cloud finish together whatever the tiles cost.
// SOR = true: the same search for pcl::StatisticalOutlierRemoval (P2) — k = mean_k + 1 neighbours including the point itself, and
// instead of the covariance the mean of the mean_k non-zero-rank distances goes to mean_dist[] in the caller's point order.
template <int K, bool SOR, bool STATS = false>
__global__ __launch_bounds__(64, KNN_TILE_WAVES_PER_SIMD) void k_knn_cov_tile(const float4* __restrict__ pts, const int* __restrict__ orig,
                                                     const HashSlot* __restrict__ table, const GridMeta* __restrict__ meta,
                                                     Cov6* __restrict__ cov, float* __restrict__ mean_dist, int* __restrict__ queue,
                                                     float* __restrict__ queue_hint, int* __restrict__ queue_n, int* __restrict__ tile_ctr, size_t cap,
                                                     size_t slots_cap, int k, int method, int G, int B, float r_scale, float gather_scale,
                                                     const int* __restrict__ tile_order = nullptr, unsigned long long* __restrict__ stats = nullptr) {
  int st_tiles = 0, st_segs = 0, st_rounds = 0, st_gathers = 0, st_cands = 0, st_build = 0, st_rank = 0, st_direct = 0;  // STATS only
  int st_fb[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // STATS only: lanes handed over, by reason
  long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = 0;  // STATS only: shader clocks per phase of the tile loop (8, 9: the cell lookup / the rest of a gather's set-up apart)
  auto phase = [&](int w) {
    if constexpr (STATS) {
      const long long t = clock64();
      ph[w] += t - ph_t;
      ph_t = t;
    }
  };
  if constexpr (STATS) ph_t = clock64();
  __shared__ float4 cand[kTileMaxCand];         // this wave's candidate list (x, y, z, sorted index)
  __shared__ uint16_t sub[kNSub][kSubRow];      // per sub-tile: positions in cand[] of the candidates inside its grown box
  __shared__ int gscratch[2 * kWave];           // tile_gather: prefix sums / range starts of the cells of a gather
#ifdef KNN_ABL_LDS_PAD  // (timing ablation: fewer resident waves per CU)
  __shared__ int lds_pad[KNN_ABL_LDS_PAD / 4];
  if (threadIdx.x == 0 && G < 0) lds_pad[G & 1] = 1;
#endif
  int c, bx;
  if (!xcd_item_chunk(G, B, c, bx)) return;
  const GridView g = grid_view(pts, orig, table, meta, c, cap, slots_cap);
  const int n = g.m.n;
  const int lane = lane_id();
  const int mysub = lane / kSubQ, sublane = lane % kSubQ;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int z = lane; z < kNSub * kSubRow; z += kWave) (&sub[0][0])[z] = 0;
  // The chain tile counter -> (tile order ->) points of a tile is three dependent memory operations, 5 % of a tile's time when it
  // runs at the top of the loop.  A wave therefore claims its NEXT tile while it starts on the current one, reads that tile's slice of
  // the order once its own first gather is back, and the points when it enters the covariance epilogue: all three arrive under work.
  // (Only where a wave takes several tiles: claiming ahead with one tile per wave - a lone 16 k-point cloud - leaves every second wave
  // without one: 262 instead of 211 us for that launch.)
  const bool ahead = kTilePrefetch && n > 4 * kWave * G;
  int tile_nx = 0;
  if (ahead && lane == 0) tile_nx = atomicAdd(&tile_ctr[c], 1);
  int i_nx = 0, pf = 0;  // next tile: this lane's point index (pf >= 1), its point (pf == 2)
  float4 q_nx = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch_index = [&](int t0) {
    int j = t0 + lane;
    if (tile_order && j < n) j = tile_order[(size_t)c * cap + j];  // tiles in an order of their own (build_tile_order)
    return j;
  };
  for (;;) {
  if (!ahead && lane == 0) tile_nx = atomicAdd(&tile_ctr[c], 1);
  const int tile0 = __builtin_amdgcn_readfirstlane(tile_nx) * kWave;
  if (tile0 >= n) break;
  if constexpr (STATS) st_tiles++;
  int t_segs = 0, t_cands = 0, t_fail_cells = 0, t_fail_list = 0;  // STATS only
  int i = pf >= 1 ? i_nx : fetch_index(tile0);
  float4 q = pf == 2 ? q_nx : g.pts[min(i, n - 1)];
  pf = 0;
  if (ahead && lane == 0) tile_nx = atomicAdd(&tile_ctr[c], 1);  // (a wave's last claim is a tile that does not exist)
  if (i >= n) q.x = q.y = q.z = INFINITY;  // past the end of the cloud: sorts to the last lanes, never active
#ifndef KNN_ABL_NO_KD  // (KNN_ABL_*: timing ablations of benchmarks/knn_only.py, never defined in the product build)
  // along the curve of build_tile_order every aligned group of 16 lanes is a compact patch already (measured: 8.7 ms without this, 9.4 ms with)
  if (!tile_order) kd_order_tile(q.x, q.y, q.z, i);
#endif
  const bool valid = i < n;
  phase(0);
  const float rcap = 16.f * g.m.cell, rmin = 0.6f * g.m.cell;
  bool fb = false;  // this lane goes to the fallback queue
  float hint = 0.f;  // ... with a radius that is known to hold its k nearest neighbours (0: not known)
  int seg = 0, len = kWave;
  while (seg < kWave) {
    const bool act = valid && lane >= seg && lane < seg + len;
    const int nact = __popcll(__ballot(act));
    bool split = false;
    if (nact > 0) {
      if constexpr (STATS) { st_segs++; t_segs++; }
      const TileBox bb = tile_bounds(act, q.x, q.y, q.z);
      const float eps = tile_eps(bb, rcap);
      float r0, e_max;  // first radius: k-NN radius of a surface with the segment's own point density
      {
        const float ex = bb.hx - bb.lx, ey = bb.hy - bb.ly, ez = bb.hz - bb.lz;
        const float e_min = fminf(fminf(ex, ey), ez);
        e_max = fmaxf(fmaxf(ex, ey), ez);
        const float e_mid = ex + ey + ez - e_min - e_max;
        const float area = fmaxf(e_max * e_mid, 0.16f * g.m.cell * g.m.cell);
        r0 = fminf(fmaxf(r_scale * sqrtf((float)k * area / (3.14159265f * (float)nact)), rmin), rcap);
      }
      // a segment that is a single sub-tile ranks the tile list directly (no sub-list, no kSubCap limit)
      const bool direct = len <= kSubQ;
      // this lane's sub-tile box as centre + half extent (the same in the kSubQ lanes of a sub-tile).  The half extent is
      // padded by eps, so that rounding of centre / extent can only ADD candidates; a sub-tile without active lanes matches
      // nothing (half extent -inf).
      float scx, scy, scz, shx, shy, shz;
      bool take;  // this sub-tile appends candidates in the next build pass
      {
        const float lx = group_min(act ? q.x : INFINITY), ly = group_min(act ? q.y : INFINITY), lz = group_min(act ? q.z : INFINITY);
        const float hx = group_max(act ? q.x : -INFINITY), hy = group_max(act ? q.y : -INFINITY), hz = group_max(act ? q.z : -INFINITY);
        take = lx <= hx;
        scx = take ? 0.5f * (lx + hx) : 0.f; scy = take ? 0.5f * (ly + hy) : 0.f; scz = take ? 0.5f * (lz + hz) : 0.f;
        shx = take ? 0.5f * (hx - lx) + eps : -INFINITY; shy = take ? 0.5f * (hy - ly) + eps : -INFINITY;
        shz = take ? 0.5f * (hz - lz) + eps : -INFINITY;
      }
      // Every sub-tile has its OWN radius: rs is what its list covers after the next build pass, rs_old what it covers now.
      // The tile list always covers the largest of them (R).
      float rs = r0, rs_old = -INFINITY;
      int slen = 0, sdone = 0;  // this lane's sub-list: entries appended / ranked so far (the same in the lanes of a sub-tile)
      SmallestKeys<K + 2> keys;
      keys.init();
      int n_ord = 0, n_done = 0;  // tile list: gathered / ranked directly (direct mode)
      TileBox ob = bb;
      float R = -1.f;
      bool has_old = false, open = act;
      phase(1);
      for (int round = 0;; round++) {
        if constexpr (STATS) st_rounds++;
        // ---- tile list: grow the gathered box to the largest radius a working sub-tile asks for
        // (the FIRST gather reaches gather_scale x further than the sub-tiles ask for: a round is a chain of dependent loads - 64 hash
        // probes, then the cell ranges - that two waves per SIMD cannot hide)
        const float Rw = uniform_f(wave_max_all(take ? rs : 0.f));
        const float Rn = (has_old || !(Rw > R)) ? Rw : fminf(gather_scale * Rw, rcap);
        if (Rw > R) {
          if constexpr (STATS) st_gathers++;
          const TileBox nb = grow(bb, Rn);
          phase(9);
          const TileCells cells = tile_cells(g, nb, kTileMaxLoad);
          phase(8);
          const int n_new = cells.ok ? tile_gather<KNN_GATHER_DEPTH>(g, cells, nb, ob, has_old, cand, n_ord, kTileMaxCand, gscratch) : -1;
          if constexpr (STATS) { if (!cells.ok) t_fail_cells++; else if (n_new < 0) t_fail_list++; }
          if (n_new < 0) {  // over a budget: split the segment if nothing has been gathered yet, else hand the open lanes over
            if (!has_old && Rw > rmin && Rn > 0.5f * e_max) {
              // clustered points (far field): the density guess overshoots and the radius, not the segment's own extent,
              // makes the box too big - start smaller, the rounds grow it back where needed
              rs = fmaxf(0.5f * rs, rmin);
              round = -1;
              continue;
            }
            if (!has_old && len > 8) split = true;
            else {
              if constexpr (STATS) st_fb[0] += __popcll(__ballot(open && !fb));
              fb = fb || open;
            }
            break;
          }
          ob = nb;
          has_old = true;
          R = Rn;
          if (ahead && pf == 0) {  // the counter's answer has been back for a while: the next tile's slice of the order goes out now
            const int t0 = __builtin_amdgcn_readfirstlane(tile_nx) * kWave;
            i_nx = t0 < n ? fetch_index(t0) : n;
            pf = 1;
          }
          if constexpr (STATS) st_cands += n_new - n_ord;
          n_ord = n_new;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the list was written by other lanes
        }
        phase(2);
        bool redo = false;
        if (direct) {
          if constexpr (STATS) st_direct += n_ord - n_done;
#pragma unroll 4
          for (int o = n_done; o < n_ord; o++) {
            const float4 p = cand[o];
            const uint32_t d = __float_as_uint(dist2f(q.x, q.y, q.z, p.x, p.y, p.z));
            keys.insert((d & ~kOrdMask) | (uint32_t)o);
          }
          n_done = n_ord;
        } else {
          // ---- sub-lists: every candidate of the tile list (old and new) that enters a working sub-tile's grown box this
          // round, i.e. lies inside (sub-box + rs) but not inside (sub-box + rs_old).  One step tests kSubQ candidates against
          // all kNSub boxes: lane (s, j) tests candidate base + j against the box of ITS sub-tile s, one ballot gives every
          // sub-tile its 8 hit bits, and the hits are appended to the sub-tile's list in candidate order.
          // (four steps per trip: the four LDS reads and box tests are independent, only the list offsets chain)
          for (int base = 0; base < n_ord; base += 4 * kSubQ) {
            if constexpr (STATS) st_build++;
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int o = base + u * kSubQ + sublane;
              const float4 p = cand[min(o, kTileMaxCand - 1)];
              // Chebyshev distance of p to the sub-box (<= 0 inside)
              const float m = fmaxf(fmaxf(fabsf(p.x - scx) - shx, fabsf(p.y - scy) - shy), fabsf(p.z - scz) - shz);
              in[u] = take && (o < n_ord) && (m <= rs) && !(m <= rs_old);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const unsigned long long mask = __ballot(in[u]);
              const unsigned long long hits = (mask >> (mysub * kSubQ)) & ((1ull << kSubQ) - 1ull);
              const int at = slen + __popcll(hits & ((1ull << sublane) - 1ull));
              if (in[u] && at < kSubCap) sub[mysub][at] = (uint16_t)((base + u * kSubQ + sublane) << 4);  // byte offset of the candidate in cand[]
              slen += __popcll(hits);  // may run past kSubCap
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          phase(3);
          // a FIRST list that does not fit is rebuilt with half the radius (the growth rounds bring back what a lane needs);
          // a list that overflows while growing hands its open lanes over
          const bool over = slen > kSubCap;
          redo = over && sdone == 0 && rs > rmin;
          if (redo) {
            slen = 0;
            rs = fmaxf(0.5f * rs, rmin);
          } else if (over) {
            if constexpr (STATS) st_fb[1] += __popcll(__ballot(open && !fb));
            fb = fb || open;
            slen = kSubCap;
          }
          // ---- rank: every lane walks the new part of ITS sub-tile's list
          const int from = sdone, to = slen;
          sdone = to;
          const int trip = wave_max_all_i(to - from);
          if constexpr (STATS) st_rank += trip;
          const uint16_t* mine = sub[mysub] + from;
          const int cnt = to - from;
          // Eight candidates per step: their eight LDS reads go out together, with the list entries of the NEXT step behind them, and
          // only then do the 8 x 31 VALU instructions of the step run - one exposed LDS latency per eight candidates.  (Left to
          // itself the compiler issued every LDS read right in front of its use and waited: five exposed latencies per FOUR
          // candidates, about as long as the arithmetic - round 3's phase clocks had the ranking at twice its instruction count.)  Rows are padded: reading up to 15 entries past `to` stays inside the
          // array (of the row of a sub-tile that has entries left); what is found there is ignored.
#if KNN_RANK_BATCH
          auto entries8 = [&](int t, uint32_t* o) {
#pragma unroll
            for (int u = 0; u < 8; u++) o[u] = (uint32_t)mine[t + u];
          };
          uint32_t o_nx[8];
          entries8(0, o_nx);
          for (int t = 0; t < trip; t += 8) {
            uint32_t o[8];
            float4 pp[8];
#pragma unroll
            for (int u = 0; u < 8; u++) o[u] = o_nx[u];
#pragma unroll
            for (int u = 0; u < 8; u++) pp[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cand) + o[u]);
            entries8(t + 8, o_nx);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const uint32_t d = __float_as_uint(dist2f(q.x, q.y, q.z, pp[u].x, pp[u].y, pp[u].z));
              const uint32_t key = (t + u < cnt) ? ((d & ~kOrdMask) | (o[u] >> 4)) : SmallestKeys<K + 2>::kEmpty;
#ifdef KNN_ABL_NO_RANK
              keys.v[u & 1] = min(keys.v[u & 1], key);
#else
              keys.insert(key);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#else  // (A/B aid: the loop of round 3)
#pragma unroll 4
          for (int t = 0; t < trip; t++) {
            const uint32_t off = (uint32_t)mine[t];
            const float4 p = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cand) + off);
            const uint32_t d = __float_as_uint(dist2f(q.x, q.y, q.z, p.x, p.y, p.z));
            keys.insert((t < cnt) ? ((d & ~kOrdMask) | (off >> 4)) : SmallestKeys<K + 2>::kEmpty);
          }
#endif
        }
        phase(4);
        // ---- which lanes are done?  every point within rsafe of a query of the sub-tile has been seen by its lanes
        const float rsafe = redo ? 0.f : rs * 0.9999f - eps;
        // (k == K - the reference's k = 20 - reads the register: a runtime index is K + 2 selects per use, ~150 instructions per tile;
        // 145.8 -> 145.0 ms per 4 096-pair step over four alternations in one box.  Folding k into the covariance epilogue as well
        // - the "t < k" guards, the division by k - costs 29 VGPRs and gains nothing: not kept.)
        const uint32_t kk = (k == K) ? keys.v[K - 1] : keys.at(k - 1);
        const bool full = kk != SmallestKeys<K + 2>::kEmpty;
        const float ub = __uint_as_float(kk | kOrdMask);  // >= the exact k-th squared distance
        if (act && full) hint = sqrtf(ub) * 1.001f + 2.f * eps;
        open = act && !fb && !(full && ub <= rsafe * rsafe);
        const unsigned long long om = __ballot(open);
        if (!om) break;
        if (round >= 6) {
          if constexpr (STATS) st_fb[2] += __popcll(__ballot(open && !fb));
          fb = fb || open;
          break;
        }
        // ---- next radius of every sub-tile that still has open lanes: what they ask for; lanes asking for much more wait
        // for a later round
        take = ((om >> (mysub * kSubQ)) & ((1ull << kSubQ) - 1ull)) != 0ull;
        if (take && !redo) {
          const float need = open ? (full ? sqrtf(ub) * 1.001f + 2.f * eps : 2.f * rs) : 0.f;
          const float near = group_max((open && need <= 2.f * rs) ? need : 0.f);
          const float far = group_min((open && need > 2.f * rs) ? need : INFINITY);
          float want = (near > 0.f) ? near : far;
          want = fminf(fmaxf(want, 1.02f * rs), rcap);
          if (!(want > rs)) {  // at the radius cap: the open lanes of this sub-tile are handed over
            if constexpr (STATS) st_fb[3] += __popcll(__ballot(open && !fb));
            fb = fb || open;
            take = false;
          } else {
            rs_old = rs;
            rs = want;
          }
        }
        if (!__any(take)) {  // only lanes that have just been handed over were open
          open = open && !fb;
          break;
        }
      }
      phase(5);
      if (pf == 1) {
        if (i_nx < n) q_nx = g.pts[i_nx];
        pf = 2;
      }
      if constexpr (STATS) { if (!split) t_cands += n_ord; }
      if (!split) {
        // closed lanes: exact unless truncation could have swapped the k-th and (k+1)-th candidates (or they tie exactly)
        const uint32_t kk = (k == K) ? keys.v[K - 1] : keys.at(k - 1), kn = (k == K) ? keys.v[K] : keys.at(k);
        const uint32_t tk = kk >> kOrdBits;
        const bool closed = act && !open && !fb;
        const bool clear = tk < (kn >> kOrdBits);
        // exactly two candidates share the k-th truncated distance: rank the pair exactly (ties: lower caller index)
        const uint32_t kn2 = (k == K) ? keys.v[K + 1] : keys.at(k + 1), kp = (k == K) ? keys.v[K >= 2 ? K - 2 : 0] : keys.at(k - 2);
        const bool pair = closed && !clear && tk < (kn2 >> kOrdBits) && (k < 2 || (kp >> kOrdBits) < tk);
        if (pair) {
          const float4 a = cand[kk & kOrdMask], b = cand[kn & kOrdMask];
          const float da = dist2f(q.x, q.y, q.z, a.x, a.y, a.z), db = dist2f(q.x, q.y, q.z, b.x, b.y, b.z);
          if (db < da || (db == da && g.orig[__float_as_int(b.w)] < g.orig[__float_as_int(a.w)])) {
#pragma unroll
            for (int t = 0; t < K; t++)
              if (t == k - 1) keys.v[t] = kn;
          }
        }
        const bool exact = closed && (clear || pair);
        if constexpr (STATS) st_fb[4] += __popcll(__ballot((act && !exact) && !fb));
        fb = fb || (act && !exact);
        if constexpr (SOR) {
          if (exact) mean_dist[(size_t)c * cap + g.orig[i]] = sor_mean_distance<K>(k, q, [&](int t) { return cand[keys.v[t] & kOrdMask]; });
        } else {
#ifdef KNN_ABL_NO_COV
          if (exact) cov[(size_t)c * cap + i] = cov6_pack(__uint_as_float(keys.v[0]), __uint_as_float(keys.v[1]), __uint_as_float(keys.v[K - 1]), 1.f, 0.f, 1.f);
#else
          if constexpr (K > 32 || !KNN_COV_REGS) {
            if (exact) covariance_of<K>(k, method, cov + ((size_t)c * cap + i), [&](int t) { return cand[keys.v[t] & kOrdMask]; });
          } else if (exact) {
            // the k neighbours into registers first, ALL their LDS reads in flight together (the two passes of the covariance then run
            // on registers: 2 k reads, each waited for in front of its use, were ~4 k cycles of a tile's ~130 k)
            float nx[K], ny[K], nz[K];
#pragma unroll
            for (int t = 0; t < K; t++) {
              const float4 pp = cand[keys.v[t < k ? t : 0] & kOrdMask];
              nx[t] = pp.x; ny[t] = pp.y; nz[t] = pp.z;
            }
            covariance_of<K>(k, method, cov + ((size_t)c * cap + i), [&](int t) { return make_float4(nx[t], ny[t], nz[t], 0.f); });
          }
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next segment overwrites the lists
        phase(6);
      }
    }
    if (split) {
      len >>= 1;
    } else {
      seg += len;
      len = seg & -seg;
    }
  }
  const unsigned long long fm = __ballot(fb);
  if (fm) {
    int base = 0;
    if (lane == 0) base = atomicAdd(&queue_n[c], __popcll(fm));
    base = __builtin_amdgcn_readfirstlane(base);
    if (fb) {
      queue[(size_t)c * cap + base + __popcll(fm & below)] = i;
      queue_hint[(size_t)c * cap + base + __popcll(fm & below)] = hint;
    }
  }
  phase(7);
  if constexpr (STATS) {
    if (stats && lane == 0) {
      atomicAdd(&stats[16 + min(t_segs, 7)], 1ull);                       // tiles by number of segments processed (incl. split attempts)
      if (t_segs > 1 && t_cands <= kTileMaxCand) atomicAdd(&stats[24], 1ull);  // ... split tiles whose segments' lists would fit one list together
      atomicAdd(&stats[25], (unsigned long long)t_fail_cells); atomicAdd(&stats[26], (unsigned long long)t_fail_list);
      if (t_segs > 1) atomicAdd(&stats[27], (unsigned long long)t_cands);
    }
  }
  }  // next tile
  if constexpr (STATS) {
    if (stats && lane == 0) {
      atomicAdd(&stats[0], (unsigned long long)st_tiles); atomicAdd(&stats[1], (unsigned long long)st_segs);
      atomicAdd(&stats[2], (unsigned long long)st_rounds); atomicAdd(&stats[3], (unsigned long long)st_gathers);
      atomicAdd(&stats[4], (unsigned long long)st_cands); atomicAdd(&stats[5], (unsigned long long)st_build);
      atomicAdd(&stats[6], (unsigned long long)st_rank); atomicAdd(&stats[7], (unsigned long long)st_direct);
      for (int w = 0; w < 8; w++) atomicAdd(&stats[8 + w], (unsigned long long)ph[w]);
      atomicAdd(&stats[32], (unsigned long long)ph[8]); atomicAdd(&stats[33], (unsigned long long)ph[9]);
      for (int w = 0; w < 4; w++) atomicAdd(&stats[28 + w], (unsigned long long)st_fb[w]);
      atomicAdd(&stats[15], 0ull);
    }
  }
}


// Fallback for the points the tile kernel hands over (wide or sparse neighbourhoods: far-field points whose k-th neighbour
// is metres away, boxes that outgrow the tile list, exact distance ties): ONE WAVE PER QUERY, the candidates spread over
// the lanes, exact (distance, caller index) order.  Per round of the growing box around the query:
//   A  every lane runs its share of the candidates through a branch-free (k+1)-deep network of FULL 32-bit distances
//      (v_med3 per position; the first version inserted into a sorted (distance, index) list under a branch: ~170
//      instructions whenever any lane wanted a candidate — 4.9 k instructions per query against 0.27 k in the tile kernel);
//   B  the 64 sorted lists, parked as columns in LDS, are merged by popping the smallest head (all equal heads at once)
//      until k values are out: that is the exact k-th distance tau, with the number of candidates below and at it;
//   C  a second pass over the same candidates collects everything below tau by ballot, and of the candidates AT tau as many as
//      are still needed, lowest caller index first (the oracle's tie rule).
// The covariance is fused: the k neighbours sit in k lanes, mean and covariance are wave reductions in fp64, and the raw
// matrices of up to 64 queries are parked in LDS so that the regularisation (the expensive part) runs lane-parallel.
template <int K, bool SOR>
// Occupancy: a query is a chain of dependent phases (cell lookups, two candidate walks, the LDS merge), so resident waves are what
// hides it: 4 per SIMD (128 VGPRs, 288 B of scratch) 1.36 ms per 1 024-cloud launch, the compiler's own 161 VGPRs (3 waves) 1.68 ms,
// 5 waves (460 B of scratch) 2.2 ms, 6: 1.67 ms.
__global__ __launch_bounds__(64, (K <= 32 ? 4 : 1)) void k_knn_hard(const float4* __restrict__ pts, const int* __restrict__ orig, const HashSlot* __restrict__ table,
                                                 const GridMeta* __restrict__ meta, Cov6* __restrict__ cov, float* __restrict__ mean_dist, const int* __restrict__ queue,
                                                 const float* __restrict__ queue_hint, const int* __restrict__ queue_n, size_t cap, size_t slots_cap,
                                                 int k, int method, int G, int B) {
  __shared__ uint32_t col[K + 1][kWave];  // per-lane sorted distance lists (column = lane)
  __shared__ double raw[kWave][6];        // raw covariances awaiting regularisation
  __shared__ int raw_i[kWave];
  __shared__ int tie_j[kWave];            // candidates at the k-th distance (tie resolution)
  __shared__ int out_j[kWave];            // the neighbours below the k-th distance, then the chosen ones at it
  int c, bx;
  if (!xcd_item_chunk(G, B, c, bx)) return;
  const GridView g = grid_view(pts, orig, table, meta, c, cap, slots_cap);
  const int n = g.m.n;
  const int lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  const int nq = queue_n[c];
  int n_raw = 0;
  auto flush = [&]() {  // regularise the parked covariances, one per lane
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < n_raw) {
      const Sym3<double> cv = {raw[lane][0], raw[lane][1], raw[lane][2], raw[lane][3], raw[lane][4], raw[lane][5]};
      const Sym3<double> r = regularize_covariance(cv, method);
      cov[(size_t)c * cap + raw_i[lane]] = cov6_pack((float)r.xx, (float)r.xy, (float)r.xz, (float)r.yy, (float)r.yz, (float)r.zz);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    n_raw = 0;
  };
  for (int t = bx; t < nq; t += G) {
    const int i = queue[(size_t)c * cap + t];
    const float4 q = g.pts[i];
    const TileBox qb{q.x, q.y, q.z, q.x, q.y, q.z};
    const float eps = tile_eps(qb, 0.f);
    // the tile kernel's upper bound of the k-th distance, when it got that far: one round then covers the neighbourhood
    const float hint = queue_hint[(size_t)c * cap + t];
    float r = (hint > 0.f) ? hint : 2.f * g.m.cell;
    int mine = -1;  // lane u < k ends up holding the u-th neighbour (sorted index); fewer than k points in the cloud: -1
    for (;;) {
      // candidates: the cells of the box q +- r, or the whole cloud once the box outgrows 64 top-level cells
      const TileCells cells = tile_cells(g, grow(qb, r), 0x7fffffff);
      const bool whole = !cells.ok || cells.total >= n;  // (a cloud smaller than k is scanned whole and yields what it has)
      if (!whole && cells.total < k) { r *= 2.f; continue; }
      const unsigned long long cellmask = __ballot(cells.ce > cells.cs);
      // Walk the candidates 64 at a time in lock step, kGatherDepth chunk loads in flight (the chunk loads do not depend on
      // each other; one load per cell at a time made a query take ~70 us): body(sorted index, point, is a candidate).
      auto walk = [&](auto&& body) {
        unsigned long long todo = whole ? 0ull : cellmask;
        int j = whole ? -kWave : 0, e = whole ? n : 0;
        auto advance = [&]() -> bool {
          j += kWave;
          if (j < e) return true;
          if (!todo) return false;
          const int cl = __builtin_ctzll(todo);
          todo &= todo - 1;
          j = __builtin_amdgcn_readlane(cells.cs, cl);
          e = __builtin_amdgcn_readlane(cells.ce, cl);
          return true;
        };
        for (;;) {
          int jb[kGatherDepth];
          bool have[kGatherDepth], in[kGatherDepth];
          float4 p[kGatherDepth];
#pragma unroll
          for (int u = 0; u < kGatherDepth; u++) {
            have[u] = advance();
            jb[u] = j;
            in[u] = have[u] && j + lane < e;
            p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in[u]) p[u] = g.pts[j + lane];
          }
          if (!have[0]) break;
#pragma unroll
          for (int u = 0; u < kGatherDepth; u++) {
            if (!have[u]) break;
            body(jb[u] + lane, p[u], in[u]);
          }
          if (!have[kGatherDepth - 1]) break;
        }
      };
      // ---- A: per-lane network of distances (positive floats order like unsigned integers)
      SmallestKeys<K + 1> nk;
      nk.init();
      walk([&](int, const float4& p, bool in) {
        nk.insert(in ? __float_as_uint(dist2f(q.x, q.y, q.z, p.x, p.y, p.z)) : SmallestKeys<K + 1>::kEmpty);
      });
#pragma unroll
      for (int u = 0; u < K + 1; u++) col[u][lane] = nk.v[u];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- B: pop the smallest head (every lane holding it pops) until k values are out
      int head = 0, n_lt = 0, n_eq = 0;
      uint32_t tau = SmallestKeys<K + 1>::kEmpty;
      while (n_lt + n_eq < k) {
        const uint32_t hd = (head <= K) ? col[min(head, K)][lane] : SmallestKeys<K + 1>::kEmpty;
        uint32_t best = wave_reduce64(hd, [](uint32_t a, uint32_t b) { return min(a, b); });
        if (best == SmallestKeys<K + 1>::kEmpty) break;  // fewer than k candidates
        // a lane may hold the value several times (distinct points at the same distance): pop them all
        int pops = 0;
        if (hd == best) {
          pops = 1;
          while (head + pops <= K && col[head + pops][lane] == best) pops++;
          head += pops;
        }
        n_lt += n_eq;
        n_eq = wave_sum_all_i(pops);
        tau = best;
      }
      const bool enough = n_lt + n_eq >= k;
      const float tau_f = __uint_as_float(tau);
      const float rs = r * 0.9999f - eps;
      if (!(whole || (enough && tau_f <= rs * rs))) {  // the box does not cover the k-th distance yet
        r = enough ? sqrtf(tau_f) * 1.001f + 2.f * eps : 2.f * r;
        continue;
      }
      // ---- C: collect.  Everything below tau is a neighbour; of the n_eq candidates AT tau, k - n_lt are needed
      const int want_eq = enough ? (k - n_lt) : n_eq;
      int n_out = 0, n_tie = 0;
      mine = -1;
      walk([&](int j, const float4& p, bool in) {
        const uint32_t key = __float_as_uint(dist2f(q.x, q.y, q.z, p.x, p.y, p.z));
        const bool lt = in && key < tau, eq = in && key == tau;
        const unsigned long long ml = __ballot(lt), me = __ballot(eq);
        if (ml) {  // output slots n_out ... in candidate order
          if (lt) out_j[n_out + __popcll(ml & below)] = j;  // n_lt < k <= 64 slots in all
          n_out += __popcll(ml);
        }
        if (me) {
          if (eq) {
            const int at = n_tie + __popcll(me & below);
            if (at < kWave) tie_j[at] = j;
          }
          n_tie += __popcll(me);
        }
      });
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ties at tau: the want_eq lowest caller indices (n_tie == n_eq; more than 64 candidates at one distance do not occur
      // in point clouds — the list is cut there)
      {
        const int nt = min(n_tie, kWave);
        const int tj = (lane < nt) ? tie_j[lane] : -1;
        const int to = (lane < nt) ? g.orig[tj] : 0x7fffffff;
        int rank = 0;  // number of tied candidates with a lower caller index
        for (int u = 0; u < nt; u++) rank += (__shfl(to, u, kWave) < to) ? 1 : 0;
        if (lane < nt && rank < want_eq && n_out + rank < kWave) out_j[n_out + rank] = tj;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int n_nb = min(n_out + min(want_eq, nt), k);
        mine = (lane < n_nb) ? out_j[lane] : -1;
      }
      break;
    }
    // ---- covariance: lanes 0..k-1 hold the neighbours
    {
      const bool has = lane < k;
      const float4 p = (has && mine >= 0) ? g.pts[mine] : make_float4(0.f, 0.f, 0.f, 0.f);
      auto wsum = [&](double v) { return wave_reduce64(v, [](double a, double b) { return a + b; }); };  // (the same bits in every lane)
      if constexpr (SOR) {  // the statistic of StatisticalOutlierRemoval instead (see sor_mean_distance)
        const double ds = wsum((has && mine >= 0) ? sqrt((double)dist2f(q.x, q.y, q.z, p.x, p.y, p.z)) : 0.0);
        if (lane == 0) mean_dist[(size_t)c * cap + g.orig[i]] = (float)(ds / (double)(k - 1));
        continue;
      }
      const double invk = 1.0 / (double)k;
      const double mx = wsum(has ? (double)p.x : 0.0) * invk, my = wsum(has ? (double)p.y : 0.0) * invk, mz = wsum(has ? (double)p.z : 0.0) * invk;
      const double x = has ? (double)p.x - mx : 0.0, y = has ? (double)p.y - my : 0.0, z = has ? (double)p.z - mz : 0.0;
      const double cxx = wsum(x * x) * invk, cxy = wsum(x * y) * invk, cxz = wsum(x * z) * invk, cyy = wsum(y * y) * invk, cyz = wsum(y * z) * invk,
                   czz = wsum(z * z) * invk;
      if (lane == 0) {
        raw[n_raw][0] = cxx; raw[n_raw][1] = cxy; raw[n_raw][2] = cxz; raw[n_raw][3] = cyy; raw[n_raw][4] = cyz; raw[n_raw][5] = czz;
        raw_i[n_raw] = i;
      }
      n_raw++;
      if (n_raw == kWave) flush();
    }
  }
  if (n_raw) flush();
}

// Step 2: neighbourhood covariance from the neighbour lists, one 24-byte record per point (all points, or the queue's).
__global__ __launch_bounds__(256) void k_cov_from_knn(const float4* __restrict__ pts, const GridMeta* __restrict__ meta, const int* __restrict__ nbr,
                                                      Cov6* __restrict__ cov, size_t cap, int k, int method, int G, int B,
                                                      const int* __restrict__ queue, const int* __restrict__ queue_n) {
  int c, bx;
  if (!xcd_item_chunk(G, B, c, bx)) return;
  const int n = meta[c].n;
  int i = bx * blockDim.x + threadIdx.x;
  if (queue) {
    if (i >= queue_n[c]) return;
    i = queue[(size_t)c * cap + i];
  }
  if (i >= n) return;
  const float4* p0 = pts + (size_t)c * cap;
  const int* nb = nbr + ((size_t)c * cap + i) * k;
  covariance_of<0>(k, method, cov + ((size_t)c * cap + i), [&](int t) {
    const int j = nb[t];
    return j >= 0 ? p0[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  });
}

// ---- the order in which the tile search takes the points (round 3) -----------------------------------------------------------
// A tile of k_knn_cov_tile is 64 points that should be ONE compact patch of a surface: the wave gathers the neighbourhood of
// their common box and every lane ranks what falls near its sub-tile.  64 consecutive points of the 3-D Morton order of the
// voxel hash are not that: the curve interleaves the height with the ground axes at every level, so a run of it is a
// CHUNK OF VOLUME — bits of the ground, of a wall and of what stands in front of it — and it jumps.  Measured on the bench's
// clouds, the longest sub-tile list of a tile (what sets the trip count of the ranking loop) is 322 candidates on average
// along the Morton order, 214 along a 3-D Hilbert curve, 141 along a 2-D Hilbert curve over the two long axes of the cloud,
// 119 for an exact kd partition into leaves of 64 (benchmarks/tile_order_sim.py); 16 % / 3 % / 0.6 % / 0.3 % of the tiles
// overflow the candidate list and are split.  A scan is a 2.5-D thing — ground, walls, poles: a small footprint on the
// plane of the two long axes holds one patch of one surface, whatever the height does.
// So the tiles are taken along a 2-D Hilbert curve (2^11 cells per axis) over the two longest axes of the cloud's bounding
// box: one more sort of (key, index) pairs per cloud, 3 radix passes.  The order changes which points share a wave, never a
// result: every point's neighbours are found exactly whatever its tile is.
__global__ __launch_bounds__(256) void k_tile_order_keys(const float4* __restrict__ pts, const GridMeta* __restrict__ meta, uint32_t* __restrict__ keys,
                                                         int* __restrict__ vals, size_t cap) {
  const int c = blockIdx.y;
  const GridMeta m = meta[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  keys[(size_t)c * cap + i] = tile_order_key(m, pts[(size_t)c * cap + i]);
  vals[(size_t)c * cap + i] = i;
}
// set.tile_order <- the sorted-array indices of every cloud along the 2-D Hilbert curve
void build_tile_order(hipStream_t s, Workspace& ws, CloudSet& set, bool single_cloud) {
  if (tuning().knn_tiles != 1 || tuning().tile_order < (single_cloud ? 2 : 1)) return;
  set.tile_order.ensure((size_t)set.B * set.cap);  // (like the workspace's value buffers: the two change hands below)
  if (!set.order_keys_in_ws) {  // (build_grid(..., tile_order_keys = true) on the same workspace has left them there already)
    const dim3 grid(div_up((long)set.cap, 256), set.B);
    hipLaunchKernelGGL(k_tile_order_keys, grid, dim3(256), 0, s, set.pts.p, set.meta.p, ws.keys_a.p, ws.vals_a.p, set.cap);
  }
  set.order_keys_in_ws = false;
  sort_pairs(s, ws, set.B, set.cap, set.count.p, 3);
  MI_LAUNCH_CHECK();
  // the sorted values ARE the order: the set takes the buffer, the workspace gets the set's old one (no copy pass; everything
  // that reads either is ordered behind this call on the stream)
  set.tile_order.swap(ws.vals_a);
  set.has_order = true;
}

template <int K, bool SOR = false>
static void launch_covariances(hipStream_t s, CloudSet& set, int k, int method, float* mean_dist = nullptr, const int* tile_order = nullptr) {
  const int G = div_up((long)set.cap, 256);
  const Tuning& tn = tuning();
  if (!tn.knn_tiles && !SOR) {  // per-lane search for every point
    hipLaunchKernelGGL(k_knn_self<K>, dim3(xcd_grid(G, set.B)), dim3(256), 0, s, set.pts.p, set.orig.p, set.table.p, set.meta.p, set.nbr.p, set.cap,
                       set.slots_cap, k, G, set.B, tn.knn_need_div, (const int*)nullptr, (const int*)nullptr);
    hipLaunchKernelGGL(k_cov_from_knn, dim3(xcd_grid(G, set.B)), dim3(256), 0, s, set.pts.p, set.meta.p, set.nbr.p, set.cov.p, set.cap, k, method, G,
                       set.B, (const int*)nullptr, (const int*)nullptr);
    return;
  }
  set.fbq.ensure((size_t)set.B * set.cap);
  set.fbhint.ensure((size_t)set.B * set.cap);
  set.fbn.ensure((size_t)2 * set.B);  // [0, B) fallback queue lengths, [B, 2B) tile counters
  MI_HIP(hipMemsetAsync(set.fbn.p, 0, sizeof(int) * 2 * set.B, s));
  // about two resident waves per SIMD over the whole batch, at most one wave per tile of a cloud, at least 16 per cloud (8 until
  // round 3; per 2 048-cloud launch 4 / 8 / 16 / 32 waves per cloud: 16.8 / 16.5 / 16.1 / 16.3 ms on the knn_only clouds - the waves
  // of a cloud finish together, so more of them shorten the tail each cloud leaves on its XCD).  MI355_TILE_WAVES < 0: that many per cloud.
  const int Gt = tn.tile_waves < 0 ? -tn.tile_waves : max(min(tn.tile_waves / max(set.B, 1), div_up((long)set.cap, kWave)), 16);
  if (tn.knn_stats && !SOR) {
    set.knn_stats.ensure(40);
    MI_HIP(hipMemsetAsync(set.knn_stats.p, 0, 40 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((k_knn_cov_tile<K, SOR, true>), dim3(xcd_grid(Gt, set.B)), dim3(kWave), 0, s, set.pts.p, set.orig.p, set.table.p, set.meta.p, set.cov.p,
                       mean_dist, set.fbq.p, set.fbhint.p, set.fbn.p, set.fbn.p + set.B, set.cap, set.slots_cap, k, method, Gt, set.B, tn.tile_rscale,
                       tile_order ? tn.tile_gather : 1.0f, tile_order, set.knn_stats.p);
    MI_HIP(hipStreamSynchronize(s));
    unsigned long long st[40];
    MI_HIP(hipMemcpy(st, set.knn_stats.p, sizeof(st), hipMemcpyDeviceToHost));
    const double tiles = (double)std::max(st[0], 1ull);
    fprintf(stderr, "[mi355] k-NN tiles by segments run (1..7+): %llu %llu %llu %llu %llu %llu %llu; split tiles whose lists fit one list together: %llu, candidates of split tiles %.0f each; "
            "gathers over budget: %llu by cells / loads, %llu by list length\n", st[17], st[18], st[19], st[20], st[21], st[22], st[23], st[24],
            (double)st[27] / std::max(1ull, st[18] + st[19] + st[20] + st[21] + st[22] + st[23]), st[25], st[26]);
    fprintf(stderr, "[mi355] k-NN lanes handed over by reason: gather over budget after the first round / a segment of 8 %llu, sub-list overflow while growing %llu, "
            "round limit %llu, radius cap %llu, (not exact: ties / truncation: the rest)\n", st[28], st[29], st[30], st[31]);
    {
      double all = 0;
      for (int w = 0; w < 8; w++) all += (double)st[8 + w];
      fprintf(stderr, "[mi355] k-NN tile phases, shader clocks per tile (share): fetch+kd %.0f (%.1f%%), segment setup %.0f (%.1f%%), gather %.0f (%.1f%%), build %.0f (%.1f%%), "
              "rank %.0f (%.1f%%), round end %.0f (%.1f%%), pair+covariance %.0f (%.1f%%), queue %.0f (%.1f%%)\n", st[8] / tiles, 100 * st[8] / all, st[9] / tiles,
              100 * st[9] / all, st[10] / tiles, 100 * st[10] / all, st[11] / tiles, 100 * st[11] / all, st[12] / tiles, 100 * st[12] / all, st[13] / tiles, 100 * st[13] / all,
              st[14] / tiles, 100 * st[14] / all, st[15] / tiles, 100 * st[15] / all);
    }
    fprintf(stderr, "[mi355] k-NN gather, shader clocks per tile: cell lookup (64 hash probes) %.0f, before it (radii, box) %.0f, the rest (flat sequence, loads, compaction) %.0f\n",
            st[32] / tiles, st[33] / tiles, st[10] / tiles);
    fprintf(stderr, "[mi355] k-NN sub-tile lists per tile: %.2f segments, %.2f rounds, %.2f gathers, %.0f candidates gathered, %.1f build trips (x 4 x 16 box tests), "
            "%.1f rank steps per lane (+ %.1f direct)\n", st[1] / tiles, st[2] / tiles, st[3] / tiles, st[4] / tiles, st[5] / tiles, st[6] / tiles, st[7] / tiles);
  } else {
    TimedLaunch tl(KF_KNN_COV_TILE, s);
    hipLaunchKernelGGL((k_knn_cov_tile<K, SOR>), dim3(xcd_grid(Gt, set.B)), dim3(kWave), 0, s, set.pts.p, set.orig.p, set.table.p, set.meta.p, set.cov.p,
                       mean_dist, set.fbq.p, set.fbhint.p, set.fbn.p, set.fbn.p + set.B, set.cap, set.slots_cap, k, method, Gt, set.B, tn.tile_rscale,
                       tile_order ? tn.tile_gather : 1.0f, tile_order);
  }
  // the points the tile kernel handed over: one wave per point, covariance included
  const int Gh = std::max(std::max(1, tn.hard_waves), std::min(4096 / std::max(set.B, 1), div_up((long)set.cap, kWave)));  // a lone big cloud gets thousands of waves
  TimedLaunch tl(KF_KNN_HARD, s);
  hipLaunchKernelGGL((k_knn_hard<K, SOR>), dim3(xcd_grid(Gh, set.B)), dim3(kWave), 0, s, set.pts.p, set.orig.p, set.table.p, set.meta.p, set.cov.p,
                     mean_dist, (const int*)set.fbq.p, (const float*)set.fbhint.p, (const int*)set.fbn.p, set.cap, set.slots_cap, k, method, Gh, set.B);
  if (tn.knn_stats) {  // tuning aid: how much of the work the tile kernel handed over
    std::vector<int> fbn(set.B), cnt(set.B);
    MI_HIP(hipStreamSynchronize(s));
    MI_HIP(hipMemcpy(fbn.data(), set.fbn.p, sizeof(int) * set.B, hipMemcpyDeviceToHost));
    MI_HIP(hipMemcpy(cnt.data(), set.count.p, sizeof(int) * set.B, hipMemcpyDeviceToHost));
    long nf = 0, np = 0;
    int worst = 0;
    for (int c = 0; c < set.B; c++) { nf += fbn[c]; np += cnt[c]; worst = std::max(worst, fbn[c]); }
    fprintf(stderr, "[mi355] k-NN tiles: %ld of %ld points handed over to k_knn_hard (%.3f %%), worst cloud %d\n", nf, np, 100.0 * nf / std::max(np, 1L), worst);
  }
}

void compute_covariances(hipStream_t s, CloudSet& set, int k, int method) {
  if (k < 1 || k > 64) throw ArgError(MI355_EINVAL, "k_correspondences must be in [1, 64]");
  set.cov.ensure((size_t)set.B * set.cap);
  if (!tuning().knn_tiles) set.nbr.ensure((size_t)set.B * set.cap * k);  // neighbour lists exist only on the per-lane path
  const int* order = set.has_order ? set.tile_order.p : nullptr;  // build_tile_order, if the caller ran it for this grid
  if (k == 20) launch_covariances<20>(s, set, k, method, nullptr, order);
  else if (k <= 8) launch_covariances<8>(s, set, k, method, nullptr, order);
  else if (k <= 16) launch_covariances<16>(s, set, k, method, nullptr, order);
  else if (k <= 32) launch_covariances<32>(s, set, k, method, nullptr, order);
  else launch_covariances<64>(s, set, k, method, nullptr, order);
  MI_LAUNCH_CHECK();
  set.has_cov = true;
}

// ---- stand-alone k-NN query (parity tests, SOR) -----------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_knn_query(const float4* __restrict__ pts, const int* __restrict__ orig, const HashSlot* __restrict__ table,
                                                   const GridMeta* __restrict__ meta, const float4* __restrict__ queries, int m, int k,
                                                   int* __restrict__ out_idx, float* __restrict__ out_d2, size_t cap, size_t slots_cap) {
  const GridView g = grid_view(pts, orig, table, meta, 0, cap, slots_cap);
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= m) return;
  const float4 q = queries[qi];
  NearestK<K> nk;
  nk.init(g.orig, k);
  grid_search(g, q.x, q.y, q.z, INFINITY, nk, knn_start_level(g, q.x, q.y, q.z, k), 1);
#pragma unroll
  for (int j = 0; j < K; j++)
    if (j < k) {
      out_idx[(size_t)qi * k + j] = nk.idx[j] >= 0 ? g.orig[nk.idx[j]] : -1;
      out_d2[(size_t)qi * k + j] = nk.d2[j];
    }
}

void knn_query(hipStream_t s, const CloudSet& set, const float4* d_queries, int m, int k, int* d_idx, float* d_d2) {
  if (k < 1 || k > 64) throw ArgError(MI355_EINVAL, "k must be in [1, 64]");
  const dim3 grid(div_up(m, 256)), blk(256);
#define MI_KNN(KK)                                                                                                                     \
  hipLaunchKernelGGL(k_knn_query<KK>, grid, blk, 0, s, set.pts.p, set.orig.p, set.table.p, set.meta.p, d_queries, m, k, d_idx, d_d2, set.cap, \
                     set.slots_cap)
  if (k == 1) MI_KNN(1);
  else if (k <= 8) MI_KNN(8);
  else if (k <= 16) MI_KNN(16);
  else if (k == 20) MI_KNN(20);
  else if (k <= 32) MI_KNN(32);
  else MI_KNN(64);
#undef MI_KNN
  MI_LAUNCH_CHECK();
}

// ---- A8/P3: pcl::transformPointCloud ----------------------------------------------------------------------------
struct Mat4f {
  float m[16];
};
__global__ __launch_bounds__(256) void k_transform(const float4* __restrict__ in, float4* __restrict__ out, size_t n, Mat4f T) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  out[i] = make_float4(transform_row(T.m, p.x, p.y, p.z), transform_row(T.m + 4, p.x, p.y, p.z), transform_row(T.m + 8, p.x, p.y, p.z), p.w);
}
void transform_points(hipStream_t s, const float4* in, float4* out, size_t n, const float* T16_host) {
  Mat4f T;
  memcpy(T.m, T16_host, sizeof(T.m));
  if (n == 0) return;
  hipLaunchKernelGGL(k_transform, dim3(div_up((long)n, 256)), dim3(256), 0, s, in, out, n, T);
  MI_LAUNCH_CHECK();
}

// ---- device-resident clouds (key-frame ring) ---------------------------------------------------------
// out[i] = T * in[i] (the fmaf order of k_transform: bit-equal to the pcl::transformPointCloud restatement) and, when the
// cloud carries covariances, cov_out[i] = R cov_in[i] R^T: a rigid motion rotates a neighbourhood covariance, and the
// PLANE-regularised matrix (I - 0.999 n n^T) with it, so a key frame's covariances are computed ONCE and follow the key frame
// into every sub-map it is part of.
__global__ __launch_bounds__(256) void k_transform_cov(const float4* __restrict__ in, const Cov6* __restrict__ cov_in, float4* __restrict__ out,
                                                       Cov6* __restrict__ cov_out, size_t n, Mat4f T) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  out[i] = make_float4(transform_row(T.m, p.x, p.y, p.z), transform_row(T.m + 4, p.x, p.y, p.z), transform_row(T.m + 8, p.x, p.y, p.z), p.w);
  if (cov_in) {
    const float R[9] = {T.m[0], T.m[1], T.m[2], T.m[4], T.m[5], T.m[6], T.m[8], T.m[9], T.m[10]};
    const Sym3<float> c = sym3_rotate(R, cov6_unpack(cov_in[i]));
    cov_out[i] = cov6_pack(c.xx, c.xy, c.xz, c.yy, c.yz, c.zz);
  }
}
void transform_points_cov(hipStream_t s, const float4* in, const Cov6* cov_in, float4* out, Cov6* cov_out, size_t n, const float* T16_host) {
  if (n == 0) return;
  Mat4f T;
  memcpy(T.m, T16_host, sizeof(T.m));
  hipLaunchKernelGGL(k_transform_cov, dim3(div_up((long)n, 256)), dim3(256), 0, s, in, cov_in, out, cov_out, n, T);
  MI_LAUNCH_CHECK();
}
// covariances between the caller's point order and the Morton order of a built single-cloud set
__global__ __launch_bounds__(256) void k_cov_permute(const Cov6* __restrict__ in, Cov6* __restrict__ out, const int* __restrict__ orig, int n, int to_caller) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (to_caller) out[orig[i]] = in[i];
  else out[i] = in[orig[i]];
}
void cov_to_caller_order(hipStream_t s, const CloudSet& set, size_t n, Cov6* out_caller) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_cov_permute, dim3(div_up((long)n, 256)), dim3(256), 0, s, (const Cov6*)set.cov.p, out_caller, (const int*)set.orig.p, (int)n, 1);
  MI_LAUNCH_CHECK();
}
void cov_from_caller_order(hipStream_t s, CloudSet& set, size_t n, const Cov6* in_caller) {
  set.cov.ensure((size_t)set.B * set.cap);
  if (n) hipLaunchKernelGGL(k_cov_permute, dim3(div_up((long)n, 256)), dim3(256), 0, s, in_caller, set.cov.p, (const int*)set.orig.p, (int)n, 0);
  MI_LAUNCH_CHECK();
  set.has_cov = true;
}

// ---- sensor_msgs/PointCloud2 <-> packed float4 -----------------------------------------------------------
__device__ __forceinline__ float load_field_as_float(const unsigned char* p, int datatype) {
  switch (datatype) {  // sensor_msgs/PointField datatypes; unaligned offsets are legal in a PointCloud2: assemble from bytes
    case 1: return (float)(signed char)p[0];
    case 2: return (float)p[0];
    case 3: { short v; memcpy(&v, p, 2); return (float)v; }
    case 4: { unsigned short v; memcpy(&v, p, 2); return (float)v; }
    case 5: { int v; memcpy(&v, p, 4); return (float)v; }
    case 6: { unsigned v; memcpy(&v, p, 4); return (float)v; }
    case 8: { double v; memcpy(&v, p, 8); return (float)v; }
    default: { float v; memcpy(&v, p, 4); return v; }
  }
}
__global__ __launch_bounds__(256) void k_pc2_unpack(const unsigned char* __restrict__ raw, float4* __restrict__ out, uint32_t width, uint32_t height,
                                                    uint32_t point_step, uint32_t row_step, int ox, int oy, int oz, int oi, int itype) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)width * height) return;
  const unsigned char* p = raw + (i / width) * (size_t)row_step + (i % width) * (size_t)point_step;
  float x, y, z;
  memcpy(&x, p + ox, 4); memcpy(&y, p + oy, 4); memcpy(&z, p + oz, 4);
  out[i] = make_float4(x, y, z, oi >= 0 ? load_field_as_float(p + oi, itype) : 0.f);
}
void pointcloud2_unpack(hipStream_t s, const unsigned char* d_raw, float4* out, uint32_t width, uint32_t height, uint32_t point_step,
                        uint32_t row_step, int ox, int oy, int oz, int oi, int itype) {
  const size_t n = (size_t)width * height;
  if (n == 0) return;
  hipLaunchKernelGGL(k_pc2_unpack, dim3(div_up((long)n, 256)), dim3(256), 0, s, d_raw, out, width, height, point_step, row_step, ox, oy, oz, oi, itype);
  MI_LAUNCH_CHECK();
}
// pcl::PointXYZI memory image: {x, y, z, 1.0f}, {intensity, 0, 0, 0}
__global__ __launch_bounds__(256) void k_pc2_pack(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  out[2 * i] = make_float4(p.x, p.y, p.z, 1.0f);
  out[2 * i + 1] = make_float4(p.w, 0.f, 0.f, 0.f);
}
void pointcloud2_pack(hipStream_t s, const float4* in, float4* out32, size_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pc2_pack, dim3(div_up((long)n, 256)), dim3(256), 0, s, in, out32, n);
  MI_LAUNCH_CHECK();
}

// ---- order-preserving compaction --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(const float4* __restrict__ in, const int* __restrict__ counts, const int* __restrict__ flags,
                                                 const int* __restrict__ local, const int* __restrict__ blockoff, const int* __restrict__ totals,
                                                 float4* __restrict__ out, int* __restrict__ cnt_out, size_t cap, int nblk) {
  const int c = blockIdx.y;
  const int n = counts[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) cnt_out[c] = (n > 0) ? totals[c] : 0;
  if (i >= n) return;
  const size_t o = (size_t)c * cap;
  if (flags[o + i]) out[o + local[o + i] + blockoff[(size_t)c * nblk + i / kScanTile]] = in[o + i];
}

// ---- A1: distance filter ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_distance_flags(const float4* __restrict__ in, const int* __restrict__ counts, int* __restrict__ flags,
                                                        size_t cap, double min_distance) {
  const int c = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counts[c]) return;
  const float4 p = in[(size_t)c * cap + i];
  flags[(size_t)c * cap + i] = passes_distance(p, min_distance) ? 1 : 0;
}

void distance_filter(hipStream_t s, Workspace& ws, const float4* in, const int* cnt_in, float4* out, int* cnt_out, int B, size_t cap,
                     double min_distance) {
  ws_reserve(ws, B, cap);
  const dim3 grid(div_up((long)cap, 256), B);
  hipLaunchKernelGGL(k_distance_flags, grid, dim3(256), 0, s, in, cnt_in, ws.flags.p, cap, min_distance);
  scan_flags(s, ws, B, cap, cnt_in);
  hipLaunchKernelGGL(k_compact, grid, dim3(256), 0, s, in, cnt_in, ws.flags.p, ws.scan_local.p, ws.scan_block.p, ws.totals.p, out, cnt_out, cap,
                     div_up((long)cap, kScanTile));
  MI_LAUNCH_CHECK();
}

// ---- P1: pcl::VoxelGrid -----------------------------------------------------------------------------------------
__global__ void k_voxel_meta(const uint32_t* __restrict__ bounds, const int* __restrict__ counts, GridMeta* __restrict__ vmeta, int B, float leaf,
                             int* __restrict__ sort_passes) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B) return;
  GridMeta m;
  m.n = counts[c];
  const float inv = 1.0f / leaf;  // inverse_leaf_size_
  float mn[3], mx[3];
  for (int d = 0; d < 3; d++) { mn[d] = ordered_to_float(bounds[c * 6 + d]); mx[d] = ordered_to_float(bounds[c * 6 + 3 + d]); }
  const bool any = m.n > 0 && (mn[0] <= mx[0]);
  if (!any) { mn[0] = mn[1] = mn[2] = 0.f; mx[0] = mx[1] = mx[2] = 0.f; }
  const long long ddx = (long long)((mx[0] - mn[0]) * inv) + 1, ddy = (long long)((mx[1] - mn[1]) * inv) + 1, ddz = (long long)((mx[2] - mn[2]) * inv) + 1;
  m.overflow = (ddx * ddy * ddz > 2147483647LL) ? 1 : 0;
  // min_b_ / div_b_ travel in (ox,oy,oz as int bits) and (dx,dy,dz)
  const int mb0 = (int)floorf(mn[0] * inv), mb1 = (int)floorf(mn[1] * inv), mb2 = (int)floorf(mn[2] * inv);
  m.ox = __int_as_float(mb0); m.oy = __int_as_float(mb1); m.oz = __int_as_float(mb2);
  m.dx = (int)floorf(mx[0] * inv) - mb0 + 1;
  m.dy = (int)floorf(mx[1] * inv) - mb1 + 1;
  m.dz = (int)floorf(mx[2] * inv) - mb2 + 1;
  m.inv_cell = inv;
  m.cell = leaf;
  m.hshift = 0; m.hmask = 0; m.pmask = 0;
  m.top_level = 0;
  m.n_cells = 0;
  for (int l = 0; l <= kMaxLevel; l++) m.level_cells[l] = 0;
  vmeta[c] = m;
  // voxel indices are below dx dy dz; the points the filter drops carry the all-ones key and must end up behind every real one,
  // so p passes are enough when dx dy dz <= 2^(8p) - 1 (a 100 m x 100 m x 10 m scan at 0.2 m: 12.5 M voxels, 3 passes)
  const long long nv = (long long)m.dx * m.dy * m.dz;
  int p = 4;
  if (!m.overflow) { p = 1; while (p < 4 && nv > (1LL << (8 * p)) - 1) p++; }
  sort_passes[c] = p;
}

__global__ __launch_bounds__(256) void k_voxel_keys(const float4* __restrict__ in, const GridMeta* __restrict__ vmeta, uint32_t* __restrict__ keys,
                                                    int* __restrict__ vals, size_t cap, double min_distance) {
  const int c = blockIdx.y;
  const GridMeta m = vmeta[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  const uint32_t key = voxel_index_key(m, in[(size_t)c * cap + i], min_distance);
  keys[(size_t)c * cap + i] = key;
  vals[(size_t)c * cap + i] = i;
}

// Is sorted element i the head of a voxel's run (or, when the index overflowed and VoxelGrid passes its input through - "Leaf size
// is too small" -, a point the fused distance filter keeps; all keys are empty then and the stable sort left the points in order)?
__device__ __forceinline__ int voxel_head(const GridMeta& m, const uint32_t* __restrict__ k, const float4* __restrict__ in, int i, uint32_t prev,
                                          double min_distance) {
  if (m.overflow) return passes_distance(in[i], min_distance) ? 1 : 0;
  const uint32_t kk = k[i];
  return (kk != kEmptyKey && (i == 0 || kk != prev)) ? 1 : 0;
}
// First stage of the scan of those head flags (sort.hip, k_scan_local) with the flags derived on the fly: writing them out and
// reading them back cost a kernel and 1 GB per 2 048-cloud launch.  Output as k_scan_local's: ws.scan_local, ws.scan_block.
__global__ __launch_bounds__(256) void k_voxel_scan_heads(const uint32_t* __restrict__ keys, const GridMeta* __restrict__ vmeta, const float4* __restrict__ in,
                                                          int* __restrict__ local, int* __restrict__ blocksum, size_t cap, int nblk, double min_distance) {
  __shared__ int wsum[4];
  const int c = blockIdx.y;
  const GridMeta m = vmeta[c];
  const int n = m.n;
  const int base = blockIdx.x * kScanTile;
  if (base >= n) {
    if (threadIdx.x == 0) blocksum[(size_t)c * nblk + blockIdx.x] = 0;
    return;
  }
  const size_t o = (size_t)c * cap;
  const uint32_t* k = keys + o;
  int* lo = local + o;
  const int i0 = base + threadIdx.x * 4;
  int v[4];
  uint32_t prev = (i0 > 0 && i0 - 1 < n) ? k[i0 - 1] : kEmptyKey;
  // (16-byte accesses need a 16-byte aligned address: a capacity that is not a multiple of four shifts the rows of the clouds behind the first)
  const bool vec = i0 + 3 < n && ((o + (size_t)i0) & 3) == 0;
  if (!m.overflow && vec) {
    // (the thread's four keys in ONE 16-byte load, its four positions in one store: four 4-byte accesses 16 bytes apart were four
    // quarter-used wave instructions each way - 0.56 ms per 1 024-cloud launch for 0.55 GB)
    const uint4 kq = *reinterpret_cast<const uint4*>(k + i0);
    const uint32_t kk[4] = {kq.x, kq.y, kq.z, kq.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      v[j] = (kk[j] != kEmptyKey && (i0 + j == 0 || kk[j] != prev)) ? 1 : 0;
      prev = kk[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      v[j] = (i0 + j < n) ? voxel_head(m, k, in + o, i0 + j, prev, min_distance) : 0;
      prev = (i0 + j < n) ? k[i0 + j] : kEmptyKey;
    }
  }
  const int s = v[0] + v[1] + v[2] + v[3];
  int incl = s;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int t = __shfl_up(incl, off, kWave);
    if (lane_id() >= off) incl += t;
  }
  if (lane_id() == kWave - 1) wsum[wave_id()] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave_id(); w++) wbase += wsum[w];
  int run = wbase + incl - s;
  if (vec) {
    *reinterpret_cast<int4*>(lo + i0) = make_int4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i0 + j < n) lo[i0 + j] = run;
      run += v[j];
    }
  }
  if (threadIdx.x == 255) blocksum[(size_t)c * nblk + blockIdx.x] = wbase + incl;
}

// The centroid of every voxel: the points of a run of equal keys added in ascending point-index order (the stable sort's order - the
// same as the oracle's, so the fp32 sums are bit-equal) and divided by their number.
// History, per 1 024-cloud launch: 1 024-point tiles staged through LDS behind a block barrier 5.7 ms; head threads walking their runs
// from memory 2.6 ms; one sorted element per lane, a run handed to its head lane by lane shifts, 1.7 ms (round 4: four groups of 64 per
// wave, their loads in flight together: 2.1 ms per 2 048 clouds).  That version was bound by its own instruction stream: ~1 660 VALU +
// 1 490 SALU instructions per 256 elements at 6.8 resident waves per SIMD (round 5 counters: 0.67 of the SIMDs' issue slots) - every lane
// of a group ran the shift loop for as many steps as the LONGEST run in the group is long (8 on average, 18 at the 90th percentile) while
// a run averages 3.5 points.
// Round 6: a LANE takes kVcLane = 4 CONSECUTIVE sorted elements (one 16-byte load each of keys, indices and output positions, four
// point gathers in flight), folds the runs that lie inside its four elements in registers, and only the one run that is still open at the
// end of its elements goes on through lane shifts - four elements of the next lane per step, so a wave of 256 elements takes
// (longest continuation) / 4 steps.  A block of one cloud that had a wave per 256 elements keeps that geometry.
constexpr int kVcLane = 4, kVcWaveElems = kWave * kVcLane;
__global__ __launch_bounds__(kPointBlock) void k_voxel_centroids(const float4* __restrict__ in, const uint32_t* __restrict__ keys,
                                                         const int* __restrict__ vals, const GridMeta* __restrict__ vmeta,
                                                         const int* __restrict__ local,
                                                         const int* __restrict__ blockoff, const int* __restrict__ totals,
                                                         float4* __restrict__ out, int* __restrict__ cnt_out, size_t cap, int nblk, int G, int B,
                                                         double min_distance, const GridMeta* __restrict__ next_meta, uint32_t* __restrict__ next_keys,
                                                         int* __restrict__ next_vals) {
  static_assert(kScanTile % kVcWaveElems == 0, "a wave's elements share one scan tile");
  int c, bx;
  if (!xcd_item_chunk(G, B, c, bx)) return;
  const GridMeta m = vmeta[c];
  const int n = m.n;
  const int lane = lane_id();
  const int w0 = bx * kVcWaveElems;
  if (w0 == 0 && lane == 0) cnt_out[c] = (n > 0) ? totals[c] : 0;
  if (w0 >= n) return;  // whole block past the end of the cloud
  const size_t o = (size_t)c * cap;
  // next_meta: the voxel hash that indexes this filter's output next (voxel_grid's `index_next`) - the output point's cell key is
  // written while the point is in a register, so build_grid needs neither a bounds pass nor a key pass over the output
  GridMeta nm;
  if (next_meta) nm = next_meta[c];
  auto emit = [&](int pos, const float4& pt) {
    out[o + pos] = pt;
    if (next_meta) {
      next_keys[o + pos] = cell_key(nm, pt.x, pt.y, pt.z);
      next_vals[o + pos] = pos;
    }
  };
  const int i0 = w0 + lane * kVcLane;
  const int boff = blockoff[(size_t)c * nblk + w0 / kScanTile];
  // ---- this lane's four elements: keys, indices, output positions (16-byte accesses where the row of the cloud is 16-byte aligned)
  uint32_t k[kVcLane];
  int v[kVcLane], pos[kVcLane];
  const bool vec = ((o & 3) == 0) && i0 + kVcLane <= n;
  if (vec) {
    const uint4 kq = *reinterpret_cast<const uint4*>(keys + o + i0);
    const int4 vq = *reinterpret_cast<const int4*>(vals + o + i0);
    const int4 lq = *reinterpret_cast<const int4*>(local + o + i0);
    k[0] = kq.x; k[1] = kq.y; k[2] = kq.z; k[3] = kq.w;
    v[0] = vq.x; v[1] = vq.y; v[2] = vq.z; v[3] = vq.w;
    pos[0] = lq.x; pos[1] = lq.y; pos[2] = lq.z; pos[3] = lq.w;
  } else {
#pragma unroll
    for (int e = 0; e < kVcLane; e++) {
      const bool valid = i0 + e < n;
      k[e] = valid ? keys[o + i0 + e] : 0xFFFFFFFEu;  // (no voxel index is that large: the overflow test bounds it by INT_MAX)
      v[e] = valid ? vals[o + i0 + e] : 0;
      pos[e] = valid ? local[o + i0 + e] : 0;
    }
  }
  const uint32_t kprev = (i0 > 0 && i0 <= n) ? keys[o + i0 - 1] : kEmptyKey;
  float4 p[kVcLane];
#pragma unroll
  for (int e = 0; e < kVcLane; e++) p[e] = (i0 + e < n) ? in[o + v[e]] : make_float4(0.f, 0.f, 0.f, 0.f);
  if (m.overflow) {  // "Leaf size is too small for the input dataset" -> output = input (what the fused distance filter keeps)
#pragma unroll
    for (int e = 0; e < kVcLane; e++)
      if (i0 + e < n && passes_distance(p[e], min_distance)) emit(pos[e] + boff, p[e]);  // (all keys are empty: the stable sort left the points in order)
    return;
  }
  // ---- heads, and the runs inside the lane: head e takes the elements behind it while their key is its own, in order
  bool head[kVcLane];
  int cnt[kVcLane];
#pragma unroll
  for (int e = 0; e < kVcLane; e++) {
    head[e] = i0 + e < n && k[e] != kEmptyKey && (i0 + e == 0 || k[e] != (e == 0 ? kprev : k[e - 1]));
    cnt[e] = 1;
  }
  float4 a[kVcLane];
#pragma unroll
  for (int e = 0; e < kVcLane; e++) a[e] = p[e];
#pragma unroll
  for (int e = 0; e + 1 < kVcLane; e++) {
    bool open = head[e];
#pragma unroll
    for (int f = e + 1; f < kVcLane; f++) {
      open = open && k[f] == k[e];
      if (open) { a[e].x += p[f].x; a[e].y += p[f].y; a[e].z += p[f].z; a[e].w += p[f].w; cnt[e]++; }
    }
  }
  // ---- the run that reaches the end of the lane's elements goes on in the next lanes: their elements arrive four per step
  // (one whole-wave DPP shift per register and step), and are added while their key is the run's
  int te = -1;  // which of the lane's heads owns that run
#pragma unroll
  for (int e = 0; e < kVcLane; e++)
    if (head[e] && k[kVcLane - 1] == k[e]) te = e;
  bool cont = te >= 0;
  const uint32_t kt = k[kVcLane - 1];
  float4 ta = make_float4(0.f, 0.f, 0.f, 0.f);
  int tc = 0;
#pragma unroll
  for (int e = 0; e < kVcLane; e++)
    if (e == te) { ta = a[e]; tc = cnt[e]; }
  if (__any(cont)) {
    uint32_t nk[kVcLane];
    float nx[kVcLane], ny[kVcLane], nz[kVcLane], nw[kVcLane];
#pragma unroll
    for (int e = 0; e < kVcLane; e++) { nk[e] = k[e]; nx[e] = p[e].x; ny[e] = p[e].y; nz[e] = p[e].z; nw[e] = p[e].w; }
    for (int d = 1; d < kWave; d++) {
#pragma unroll
      for (int e = 0; e < kVcLane; e++) {
        nk[e] = wave_shl1(nk[e]);
        nx[e] = wave_shl1(nx[e]); ny[e] = wave_shl1(ny[e]); nz[e] = wave_shl1(nz[e]); nw[e] = wave_shl1(nw[e]);
      }
      const bool in_wave = lane + d < kWave;
      bool take = cont && in_wave;
#pragma unroll
      for (int e = 0; e < kVcLane; e++) {
        take = take && nk[e] == kt;
        if (take) { ta.x += nx[e]; ta.y += ny[e]; ta.z += nz[e]; ta.w += nw[e]; tc++; }
      }
      cont = take || (cont && !in_wave);  // a run that reaches the end of the wave stays open for the memory walk below
      if (!__any(cont && in_wave)) break;
    }
    if (cont) {  // the run continues past this wave's 256 elements
      for (int j = w0 + kVcWaveElems; j < n; j += 4) {
        uint32_t kq[4];
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int jj = min(j + u, n - 1);
          kq[u] = (j + u < n) ? keys[o + jj] : ~kt;
          q[u] = in[o + vals[o + jj]];
        }
        bool more = true;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          more = more && (kq[u] == kt);
          if (more) { ta.x += q[u].x; ta.y += q[u].y; ta.z += q[u].z; ta.w += q[u].w; tc++; }
        }
        if (!more) break;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < kVcLane; e++) {
    if (!head[e]) continue;
    const float4 s = (e == te) ? ta : a[e];
    const float fc = (float)((e == te) ? tc : cnt[e]);
    emit(pos[e] + boff, make_float4(s.x / fc, s.y / fc, s.z / fc, s.w / fc));
  }
}

void voxel_grid(hipStream_t s, Workspace& ws, const float4* in, const int* cnt_in, float4* out, int* cnt_out, int B, size_t cap, float leaf,
                double min_distance, CloudSet* index_next, float index_cell) {
  ws_reserve(ws, B, cap);
  compute_bounds(s, ws, in, cnt_in, B, cap, min_distance);
  hipLaunchKernelGGL(k_voxel_meta, dim3(div_up(B, 64)), dim3(64), 0, s, ws.bounds.p, cnt_in, ws.vmeta.p, B, leaf, ws.sort_passes.p);
  const dim3 grid(div_up((long)cap, kPointBlock), B);
  if (sort_pairs_in_tiles(cap)) {
    // the tile sort computes a point's voxel index while it loads its tile: no key pass, no (key, index) arrays written and read back
    const VoxelKeySource vk{in, ws.vmeta.p, min_distance};
    sort_pairs(s, ws, B, cap, cnt_in, 4, ws.sort_passes.p, &vk);
  } else {
    hipLaunchKernelGGL(k_voxel_keys, grid, dim3(kPointBlock), 0, s, in, ws.vmeta.p, ws.keys_a.p, ws.vals_a.p, cap, min_distance);
    sort_pairs(s, ws, B, cap, cnt_in, 4, ws.sort_passes.p);
  }
  const int nblk = div_up((long)cap, kScanTile);
  ws.scan_local.ensure((size_t)B * cap);
  ws.scan_block.ensure((size_t)B * nblk);
  hipLaunchKernelGGL(k_voxel_scan_heads, dim3(nblk, B), dim3(256), 0, s, ws.keys_a.p, ws.vmeta.p, in, ws.scan_local.p, ws.scan_block.p, cap, nblk, min_distance);
  scan_blocks(s, ws, B, cap);
  if (index_next) {
    // The voxel hash of the OUTPUT gets its geometry now - from the bounds of what goes into the centroids (they lie inside) and the
    // output counts the scan has just produced - so that the centroid pass can write the cell keys of build_grid's sort.
    if (index_next->B != B || index_next->cap != cap) throw ArgError(MI355_EINVAL, "voxel_grid: index_next has another shape");
    hipLaunchKernelGGL(k_grid_meta, dim3(div_up(B, 64)), dim3(64), 0, s, ws.bounds.p, ws.totals.p, index_next->meta.p, B, index_cell, ws.sort_passes.p);
  }
  const int Gv = div_up((long)cap, kVcWaveElems);
  TimedLaunch tl(KF_VOXEL_CENTROIDS, s);
  hipLaunchKernelGGL(k_voxel_centroids, dim3(xcd_grid(Gv, B)), dim3(kPointBlock), 0, s, in, ws.keys_a.p, ws.vals_a.p, ws.vmeta.p, ws.scan_local.p,
                     ws.scan_block.p, ws.totals.p, out, cnt_out, cap, nblk, Gv, B, min_distance,
                     index_next ? (const GridMeta*)index_next->meta.p : (const GridMeta*)nullptr, ws.keys_b.p, ws.vals_b.p);
  MI_LAUNCH_CHECK();
  if (index_next) {  // sort_pairs reads the *_a buffers
    std::swap(ws.keys_a.p, ws.keys_b.p); std::swap(ws.keys_a.n, ws.keys_b.n);
    std::swap(ws.vals_a.p, ws.vals_b.p); std::swap(ws.vals_a.n, ws.vals_b.n);
    index_next->cell_keys_in_ws = true;
  }
}

// ---- P2: pcl::StatisticalOutlierRemoval -------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_sor_meandist(const float4* __restrict__ pts, const int* __restrict__ orig, const HashSlot* __restrict__ table,
                                                      const GridMeta* __restrict__ meta, float* __restrict__ mean_dist, size_t cap, size_t slots_cap,
                                                      int mean_k) {
  const int c = blockIdx.y;
  const GridView g = grid_view(pts, orig, table, meta, c, cap, slots_cap);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.m.n) return;
  const float4 q = g.pts[i];
  NearestKD<K> nk;  // distances only: the statistic needs no neighbour identities
  nk.init(mean_k + 1);
  grid_search(g, q.x, q.y, q.z, INFINITY, nk, knn_start_level(g, q.x, q.y, q.z, mean_k + 1), 1);
  double dist_sum = 0.0;
#pragma unroll
  for (int j = 1; j < K; j++)  // j = 0 is the query point itself
    if (j < mean_k + 1) dist_sum += sqrt((double)nk.d2[j]);
  mean_dist[(size_t)c * cap + g.orig[i]] = (float)(dist_sum / (double)mean_k);  // caller order
}

__global__ __launch_bounds__(1024) void k_sor_threshold(const float* __restrict__ mean_dist, const int* __restrict__ counts, double* __restrict__ thr,
                                                        size_t cap, double stddev_mul) {
  __shared__ double sm[2 * 16];
  const int c = blockIdx.x;
  const int n = counts[c];
  double acc[2] = {0.0, 0.0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double d = (double)mean_dist[(size_t)c * cap + i];
    acc[0] += d;
    acc[1] += d * d;
  }
  block_sum<2, 16>(acc, sm);
  if (threadIdx.x == 0) {
    const double valid = (double)n;
    const double mean = acc[0] / valid;
    const double variance = (acc[1] - acc[0] * acc[0] / valid) / (valid - 1.0);
    thr[c] = mean + stddev_mul * sqrt(variance);
  }
}

__global__ __launch_bounds__(256) void k_sor_flags(const float* __restrict__ mean_dist, const int* __restrict__ counts, const double* __restrict__ thr,
                                                   int* __restrict__ flags, uint8_t* __restrict__ keep, size_t cap) {
  const int c = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= counts[c]) return;
  const int f = !((double)mean_dist[(size_t)c * cap + i] > thr[c]);
  flags[(size_t)c * cap + i] = f;
  if (keep) keep[(size_t)c * cap + i] = (uint8_t)f;
}

void sor_filter(hipStream_t s, Workspace& ws, CloudSet& set, const float4* in, const int* cnt_in, float4* out, int* cnt_out, int B, size_t cap,
                int mean_k, double stddev_mul, float cell, uint8_t* d_keep, float* d_mean_dist) {
  if (mean_k < 1 || mean_k > 63) throw ArgError(MI355_EINVAL, "mean_k must be in [1, 63]");
  if (set.B != B || set.cap != cap) set.alloc(B, cap);
  FamilyScope family(KF_SOR);
  build_grid(s, ws, in, cnt_in, set, cell, B > 1);
  build_tile_order(s, ws, set, B == 1);  // batches: the tiles of the search along the Hilbert curve, like the covariances'
  const int* order = set.has_order ? set.tile_order.p : nullptr;
  ws.fscratch.ensure((size_t)B * cap);
  ws.dscratch.ensure(B);
  float* md = d_mean_dist ? d_mean_dist : ws.fscratch.p;
  const dim3 grid(div_up((long)cap, 256), B);
  const int kk = mean_k + 1;
  if (tuning().knn_tiles) {
    // the (mean_k + 1)-NN of every point is the covariance kernels' search with another epilogue: wave-cooperative tiles, the
    // wave-per-query kernel for what they hand over (round 2 walked the hash per lane here: 0.98 ms for ONE 64 k-ray cloud)
    if (kk <= 8) launch_covariances<8, true>(s, set, kk, 0, md, order);
    else if (kk <= 16) launch_covariances<16, true>(s, set, kk, 0, md, order);
    else if (kk <= 32) launch_covariances<32, true>(s, set, kk, 0, md, order);
    else launch_covariances<64, true>(s, set, kk, 0, md, order);
  } else {
#define MI_SOR(KK) hipLaunchKernelGGL(k_sor_meandist<KK>, grid, dim3(256), 0, s, set.pts.p, set.orig.p, set.table.p, set.meta.p, md, cap, set.slots_cap, mean_k)
    if (kk == 31) MI_SOR(31);
    else if (kk <= 8) MI_SOR(8);
    else if (kk <= 16) MI_SOR(16);
    else if (kk <= 32) MI_SOR(32);
    else MI_SOR(64);
#undef MI_SOR
  }
  TimedLaunch tl(KF_SOR, s);
  hipLaunchKernelGGL(k_sor_threshold, dim3(B), dim3(1024), 0, s, md, cnt_in, ws.dscratch.p, cap, stddev_mul);
  hipLaunchKernelGGL(k_sor_flags, grid, dim3(256), 0, s, md, cnt_in, ws.dscratch.p, ws.flags.p, d_keep, cap);
  scan_flags(s, ws, B, cap, cnt_in);
  hipLaunchKernelGGL(k_compact, grid, dim3(256), 0, s, in, cnt_in, ws.flags.p, ws.scan_local.p, ws.scan_block.p, ws.totals.p, out, cnt_out, cap,
                     div_up((long)cap, kScanTile));
  MI_LAUNCH_CHECK();
}

}  // namespace mi355
