// Wave-cooperative neighbour search over the Morton-ordered voxel hash (device side).
//
// grid_search (grid.hpp) gives every lane its own walk through hash cells: exact, but each candidate costs one dependent,
// divergent load and the wave advances at the pace of its slowest lane (PMC: 28 % lane utilisation, 59 % of the wave
// cycles waiting on memory in k_knn_self).  Here one wave serves a TILE of 64 queries that are close in space (64
// consecutive points of the Morton-sorted cloud, or their rigid transform) and works on the union of their
// neighbourhoods instead:
//   1. the tile's bounding box, grown by a radius r, is covered by <= 64 cells of the finest level that allows it;
//      lane c looks up cell c (one hash probe per lane, all in flight together);
//   2. the non-empty cells are contiguous ranges of the sorted point array, so the wave gathers them with coalesced
//      64-point loads, several in flight; a ballot keeps the points inside the grown box (and outside the box of the
//      previous round) and compacts them into the wave's candidate list in LDS;
//   3. every lane then walks the same list (LDS broadcast reads) and tests each candidate against its own query.
// All lanes do the same work on the same candidate, so the inner loop has no divergence and no global load.
//
// Exactness: every point within distance r of ANY query of the tile lies inside the grown box (the query is inside the
// tile box), and cell binning is monotone in each coordinate, so the cells of [lo - r, hi + r] hold all of them.  A lane
// whose result distance is below (r - eps)^2 therefore has the exact answer; the others ask for a larger r (a second
// round scans only the points of the new box that were not in the old one) or fall back to grid_search.
#pragma once
#include "grid.hpp"
#include "primitives.hpp"

namespace mi355 {

struct TileBox {
  float lx, ly, lz, hx, hy, hz;
};
__device__ __forceinline__ bool in_box(const TileBox& b, float x, float y, float z) {
  return x >= b.lx && x <= b.hx && y >= b.ly && y <= b.hy && z >= b.lz && z <= b.hz;
}
__device__ __forceinline__ TileBox grow(const TileBox& b, float r) { return TileBox{b.lx - r, b.ly - r, b.lz - r, b.hx + r, b.hy + r, b.hz + r}; }

#ifndef WAVE_REDUCE_DPP
#define WAVE_REDUCE_DPP 1  // 0: the ds_bpermute butterflies of round 3 (A/B builds)
#endif
#if WAVE_REDUCE_DPP
__device__ __forceinline__ float wave_min_all(float v) { return wave_reduce64(v, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max_all(float v) { return wave_reduce64(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ int wave_sum_all_i(int v) { return wave_reduce64(v, [](int a, int b) { return a + b; }); }
__device__ __forceinline__ int wave_max_all_i(int v) { return wave_reduce64(v, [](int a, int b) { return max(a, b); }); }
#else
__device__ __forceinline__ float wave_min_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ int wave_sum_all_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_all_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, kWave));
  return __builtin_amdgcn_readfirstlane(v);
}
#endif
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// Absolute slack that covers the fp32 rounding of `lo - r` / `hi + r` and of the squared distances at this coordinate scale.
__device__ __forceinline__ float tile_eps(const TileBox& b, float rmax) {
  const float a = fmaxf(fmaxf(fmaxf(fabsf(b.lx), fabsf(b.hx)), fmaxf(fabsf(b.ly), fabsf(b.hy))), fmaxf(fabsf(b.lz), fabsf(b.hz))) + rmax;
  return 1.0e-6f * a;
}

// Bounding box of the queries of the lanes with `act` set, the same value in every lane.
__device__ __forceinline__ TileBox tile_bounds(bool act, float x, float y, float z) {
  TileBox b;
  b.lx = uniform_f(wave_min_all(act ? x : INFINITY)); b.ly = uniform_f(wave_min_all(act ? y : INFINITY));
  b.lz = uniform_f(wave_min_all(act ? z : INFINITY));
  b.hx = uniform_f(wave_max_all(act ? x : -INFINITY)); b.hy = uniform_f(wave_max_all(act ? y : -INFINITY));
  b.hz = uniform_f(wave_max_all(act ? z : -INFINITY));
  return b;
}

// The cells covering a box: lane c holds the sorted-array range [cs, ce) of cell c (empty range if the cell is empty or
// c is past the cell count).  The level is the finest at which the box needs at most 64 cells.  ok == false: the box needs
// more than 64 cells even at the top level, or its cells hold more than `max_points` points.
struct TileCells {
  int cs, ce;
  int total;  // points in all cells (wave-uniform)
  bool ok;
};
__device__ __forceinline__ TileCells tile_cells(const GridView& g, const TileBox& nb, int max_points) {
  const GridMeta& m = g.m;
  const int lane = lane_id();
  auto base_cell = [&](float v, float o, int d) { return min(max((int)floorf(cell_coord(v, o, m.inv_cell)), 0), d - 1); };
  const int x0 = base_cell(nb.lx, m.ox, m.dx), x1 = base_cell(nb.hx, m.ox, m.dx);
  const int y0 = base_cell(nb.ly, m.oy, m.dy), y1 = base_cell(nb.hy, m.oy, m.dy);
  const int z0 = base_cell(nb.lz, m.oz, m.dz), z1 = base_cell(nb.hz, m.oz, m.dz);
  int L = 0;
  auto count = [&](int l) { return ((x1 >> l) - (x0 >> l) + 1) * ((y1 >> l) - (y0 >> l) + 1) * ((z1 >> l) - (z0 >> l) + 1); };
  while (L < m.top_level && count(L) > kWave) L++;
  const int cx0 = x0 >> L, cy0 = y0 >> L, cz0 = z0 >> L;
  const int sx = (x1 >> L) - cx0 + 1, sy = (y1 >> L) - cy0 + 1, sz = (z1 >> L) - cz0 + 1;
  const int nc = sx * sy * sz;
  TileCells t{0, 0, 0, nc <= kWave};
  if (!t.ok) return t;
  if (lane < nc) {
    const int ix = cx0 + lane % sx, iy = cy0 + (lane / sx) % sy, iz = cz0 + lane / (sx * sy);
    int s, e;
    if (cell_range(g, ((uint32_t)L << 27) | morton27((uint32_t)ix, (uint32_t)iy, (uint32_t)iz), s, e)) { t.cs = s; t.ce = e; }
  }
  t.total = wave_sum_all_i(t.ce - t.cs);
  t.ok = t.total <= max_points;
  return t;
}

// The points of a TileCells as ONE sequence of `total` positions (cells in lane order, a cell's points in array order): position f is
// sorted-array index index_of(f).  A base cell of a voxelised scan holds ~6 points, so walking the cells one 64-lane load per cell -
// what tile_gather did until round 4 - issued up to 64 loads with a tenth of their lanes active, eight at a time; the flat sequence
// takes total / 64 full loads.  scratch: 2 x 64 ints of LDS private to the wave (prefix sums, range starts); the position of a lane is
// found by a 6-step binary search of the prefix sums.
#ifndef TILE_FLAT
#define TILE_FLAT 1  // 0: a load per cell (A/B builds)
#endif
struct FlatCells {
  int total;
  __device__ __forceinline__ void init(const TileCells& c, int* scratch) {
    const int lane = lane_id();
    const int sz = max(c.ce - c.cs, 0);
    int incl = sz;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int t = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += t;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (earlier readers of the scratch are done)
    scratch[lane] = incl - sz;
    scratch[kWave + lane] = c.cs;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    total = __builtin_amdgcn_readlane(incl, kWave - 1);
  }
  // (f < total) the largest cell whose first position is <= f is the cell that holds f: empty cells share their successor's prefix
  __device__ __forceinline__ int index_of(int f, const int* scratch) const {
    int lo = 0;
#pragma unroll
    for (int step = kWave / 2; step > 0; step >>= 1)
      if (scratch[lo + step] <= f) lo += step;
    return scratch[kWave + lo] + (f - scratch[lo]);
  }
};

// Append the points of `cells` that lie inside `nb` and (if has_old) outside `ob` to the wave's LDS candidate list
// cand[n0 ...] as (x, y, z, sorted index).  Returns the new list length, or -1 if it would exceed `cap` (the list is then
// undefined past n0).  The chunk loads do not depend on each other, so kGatherDepth of them are in flight per wave.
constexpr int kGatherDepth = 8;
template <int DEPTH = kGatherDepth>
__device__ __forceinline__ int tile_gather(const GridView& g, const TileCells& cells_in, const TileBox& nb, const TileBox& ob, bool has_old,
                                           float4* cand, int n0, int cap, int* scratch) {
  const int lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
#if TILE_FLAT
  FlatCells fc;
  fc.init(cells_in, scratch);
  int n = n0;
  for (int f0 = 0; f0 < fc.total; f0 += kWave * DEPTH) {
    int idx[DEPTH];
    float4 p[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
      const int f = f0 + u * kWave + lane;
      idx[u] = 0;
      p[u] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
      if (f < fc.total) {
        idx[u] = fc.index_of(f, scratch);
        p[u] = g.pts[idx[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
      if (f0 + u * kWave >= fc.total) break;
      // a lane past the end of the sequence holds +inf coordinates and fails the box test
      const bool in = in_box(nb, p[u].x, p[u].y, p[u].z) && !(has_old && in_box(ob, p[u].x, p[u].y, p[u].z));
      const unsigned long long mask = __ballot(in);
      const int cnt = __popcll(mask);
      if (n + cnt > cap) return -1;
      if (in) cand[n + __popcll(mask & below)] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float(idx[u]));
      n += cnt;
    }
  }
  return n;
#else
  (void)scratch;
  unsigned long long cells = __ballot(cells_in.ce > cells_in.cs);
  int j = 0, e = 0;  // cursor of the chunk iterator (wave-uniform)
  auto advance = [&]() -> bool {
    j += kWave;
    if (j < e) return true;
    if (!cells) return false;
    const int cl = __builtin_ctzll(cells);
    cells &= cells - 1;
    j = __builtin_amdgcn_readlane(cells_in.cs, cl);
    e = __builtin_amdgcn_readlane(cells_in.ce, cl);
    return true;
  };
  int n = n0;
  for (;;) {
    int jb[DEPTH];
    bool have[DEPTH];
    float4 p[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
      have[u] = advance();
      jb[u] = j;
      p[u] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
      if (have[u] && j + lane < e) p[u] = g.pts[j + lane];
    }
    if (!have[0]) break;
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
      if (!have[u]) break;
      // a lane past the end of its range holds +inf coordinates and fails the box test
      const bool in = in_box(nb, p[u].x, p[u].y, p[u].z) && !(has_old && in_box(ob, p[u].x, p[u].y, p[u].z));
      const unsigned long long mask = __ballot(in);
      const int cnt = __popcll(mask);
      if (n + cnt > cap) return -1;
      if (in) cand[n + __popcll(mask & below)] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float(jb[u] + lane));
      n += cnt;
    }
    if (!have[DEPTH - 1]) break;
  }
  return n;
#endif
}

// The K smallest 32-bit keys seen so far, ascending, maintained by a branch-free median network (one v_med3_u32 per
// position and candidate).  Keys are squared distances (positive floats order like unsigned integers) whose low mantissa
// bits carry a payload, so the identity of a neighbour travels through the network with its distance.
template <int K>
struct SmallestKeys {
  static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
  uint32_t v[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < K; i++) v[i] = kEmpty;
  }
  __device__ __forceinline__ void insert(uint32_t key) {
#pragma unroll
    for (int i = K - 1; i > 0; i--) {
      const uint32_t a = v[i - 1], b = v[i];  // a <= b
      v[i] = max(min(a, b), min(max(a, b), key));  // median(a, b, key) -> v_med3_u32
    }
    v[0] = min(v[0], key);
  }
  __device__ __forceinline__ uint32_t at(int i) const {  // runtime index
    uint32_t r = kEmpty;
#pragma unroll
    for (int t = 0; t < K; t++)
      if (t == i) r = v[t];
    return r;
  }
};

}  // namespace mi355
