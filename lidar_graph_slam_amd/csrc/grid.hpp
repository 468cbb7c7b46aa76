// Multi-level voxel-hash nearest-neighbour search (device side).
//
// Replaces the FLANN kd-trees behind pcl::search::KdTree::nearestKSearch with a
// structure that is built by ONE sort and searched by coalesced cell-range scans:
//   * points are sorted by the Morton code of their base cell (9 bits per axis), so the cell a point falls in at
//     EVERY level l (edge = base * 2^l) is a contiguous range of the same array;
//   * one open-addressing hash maps (level, Morton >> 3l) -> [start, end) (16-byte slots);
//   * a query scans the Chebyshev shells r = 0, 1 of cells around its own cell at a fine level; if the k-th best
//     distance is not yet inside the radius those shells are guaranteed to cover, it ESCALATES to a coarser level
//     (instead of walking hundreds of fine cells) and only the top level keeps expanding shells.  The search stops
//     when the guaranteed radius covers the k-th distance, so the result is the EXACT k-NN (same answer as the
//     kd-tree; ties towards the lower point index), independent of cell size and level choice.
// Cells that cannot beat the current k-th distance are skipped before their hash probe.
#pragma once
#include "engine.hpp"

namespace mi355 {

// A1: the distance filter's test, and P1: pcl::VoxelGrid's voxel index of a point (k_voxel_meta keeps min_b_ in ox / oy / oz as int bits and
// div_b_ in dx / dy / dz) - kEmptyKey for a point the filter drops.  Shared by k_voxel_keys and by the tile sort that computes the keys
// while it loads its tile (sort.hip).
__device__ __forceinline__ bool passes_distance(const float4& p, double min_distance) {
  if (min_distance < 0.0) return true;
  const float nrm = sqrtf(p.x * p.x + (p.y * p.y + p.z * p.z));
  return min_distance < (double)nrm;
}
__device__ __forceinline__ uint32_t voxel_index_key(const GridMeta& m, const float4& p, double min_distance) {
  uint32_t key = kEmptyKey;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && !m.overflow && passes_distance(p, min_distance)) {
    const int mb0 = __float_as_int(m.ox), mb1 = __float_as_int(m.oy), mb2 = __float_as_int(m.oz);
    const int i0 = (int)(floorf(p.x * m.inv_cell) - (float)mb0);
    const int i1 = (int)(floorf(p.y * m.inv_cell) - (float)mb1);
    const int i2 = (int)(floorf(p.z * m.inv_cell) - (float)mb2);
    key = (uint32_t)(i0 + i1 * m.dx + i2 * m.dx * m.dy);
  }
  return key;
}

struct GridView {
  const float4* pts;      // this cloud's sorted points
  const int* orig;        // sorted -> caller index
  const HashSlot* table;  // this cloud's slots
  GridMeta m;
};

__device__ __forceinline__ GridView grid_view(const float4* pts, const int* orig, const HashSlot* table, const GridMeta* meta, int c, size_t cap,
                                              size_t slots_cap) {
  GridView g;
  g.pts = pts + (size_t)c * cap;
  g.orig = orig + (size_t)c * cap;
  g.table = table + (size_t)c * slots_cap;
  g.m = meta[c];
  return g;
}

__host__ __device__ __forceinline__ uint32_t morton_expand9(uint32_t x) {  // 9 bits -> every third bit
  x &= 0x1FFu;
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}
__host__ __device__ __forceinline__ uint32_t morton27(uint32_t ix, uint32_t iy, uint32_t iz) {
  return morton_expand9(ix) | (morton_expand9(iy) << 1) | (morton_expand9(iz) << 2);
}
// table key of the level-l cell holding a point whose base-cell Morton code is `m27`
__host__ __device__ __forceinline__ uint32_t level_key(uint32_t m27, int l) { return ((uint32_t)l << 27) | (m27 >> (3 * l)); }

__device__ __forceinline__ uint32_t hash_slot(uint32_t key, uint32_t hshift) { return (key * 0x9E3779B1u) >> hshift; }

// next slot of a probe sequence: linear, wrapping inside the probe block (engine.hpp kProbeBlock) of the first slot
__device__ __forceinline__ uint32_t next_probe(uint32_t h, uint32_t pmask) { return (h & ~pmask) | ((h + 1u) & pmask); }

__device__ __forceinline__ bool cell_range(const GridView& g, uint32_t key, int& s, int& e) {
  uint32_t h = hash_slot(key, g.m.hshift);
  for (;;) {
    const int4 sl = *reinterpret_cast<const int4*>(&g.table[h]);
    if ((uint32_t)sl.x == key) { s = sl.y; e = sl.z; return true; }
    if ((uint32_t)sl.x == kEmptyKey) return false;
    h = next_probe(h, g.m.pmask);
  }
}

// Continuous base-cell coordinate of a query along one axis, clamped so the int conversion cannot overflow.
__device__ __forceinline__ float cell_coord(float q, float o, float inv) { return fminf(fmaxf((q - o) * inv, -1.0e6f), 1.0e6f); }

// Number of points in the level-l cell that holds base-cell coordinates (bx,by,bz) (inside the grid).
__device__ __forceinline__ int own_cell_count(const GridView& g, int bx, int by, int bz, int l) {
  int s, e;
  if (!cell_range(g, level_key(morton27((uint32_t)bx, (uint32_t)by, (uint32_t)bz), l), s, e)) return 0;
  return e - s;
}

// Scan the points of one cell range, Visitor::kScanWidth loads in flight per step.  The search is latency-bound on these
// dependent loads (one wave-instruction per candidate under SIMT); issuing W of them back to back overlaps their latency
// and cuts the wave-level iteration count by W.
template <typename Visitor>
__device__ __forceinline__ void scan_range(const GridView& g, int s, int e, float qx, float qy, float qz, Visitor& vis) {
  constexpr int W = Visitor::kScanWidth;
  for (int j = s; j < e; j += W) {
    float4 p[W];
#pragma unroll
    for (int u = 0; u < W; u++) p[u] = g.pts[min(j + u, e - 1)];
#pragma unroll
    for (int u = 0; u < W; u++)
      if (j + u < e) vis.visit(j + u, dist2f(qx, qy, qz, p[u].x, p[u].y, p[u].z));
  }
}

// First probe of a cell: the slot the key hashes to (one 16-byte load, issued early by the callers).
__device__ __forceinline__ int4 first_slot(const GridView& g, uint32_t key) {
  return *reinterpret_cast<const int4*>(&g.table[hash_slot(key, g.m.hshift)]);
}
// Finish a lookup whose first slot has already been loaded.
__device__ __forceinline__ bool resolve_range(const GridView& g, uint32_t key, int4 sl, int& s, int& e) {
  uint32_t h = hash_slot(key, g.m.hshift);
  for (;;) {
    if ((uint32_t)sl.x == key) { s = sl.y; e = sl.z; return true; }
    if ((uint32_t)sl.x == kEmptyKey) return false;
    h = next_probe(h, g.m.pmask);
    sl = *reinterpret_cast<const int4*>(&g.table[h]);
  }
}

// Visitor interface:
//   float bound()            pruning squared distance: nothing farther can enter the result
//   float complete()         k-th best squared distance of the CURRENT scan (+inf while the list is not full)
//   void  visit(j, d2)       candidate (sorted index j)
//   void  restart()          called when the search escalates to a coarser level (the scan starts over)
// max_d2: nothing at or beyond this squared distance is wanted.  level0: level to start at; level_step: levels to
// climb per escalation.  Returns the squared radius inside which EVERY point has been seen (or pruned as unable to
// improve the result).
//
// SIMT shape: for the shells r <= 1 every lane first builds a 27-bit mask of the neighbour cells that can still matter
// (pure ALU), then walks only its own set bits — a wave iterates max-popcount times instead of 27, and no lane drags the
// others through cells only it needs.
// level_cap (optional): the search never works above that level — after the shells r <= 1 of level min(level_cap, top) it
// returns what it has explored even if that does not cover the result yet (the caller hands such a query to a wave-cooperative
// search: a lone lane walking the big cells of the coarse levels is what the slowest wave of a launch waits for).
template <typename Visitor>
__device__ __forceinline__ float grid_search(const GridView& g, float qx, float qy, float qz, float max_d2, Visitor& vis, int level0, int level_step,
                                             int level_cap = kMaxLevel + 1) {
  const GridMeta& m = g.m;
  if (m.n <= 0) return INFINITY;
  const float bx = cell_coord(qx, m.ox, m.inv_cell), by = cell_coord(qy, m.oy, m.inv_cell), bz = cell_coord(qz, m.oz, m.inv_cell);
  const float margin = 1.0e-3f;  // in cell units: covers the fp32 fuzz of the cell lattice
  int l = min(min(level0, level_cap), m.top_level);
  for (;;) {
    const bool top = l >= m.top_level;
    const float scale = 1.0f / (float)(1 << l);
    const float cell = m.cell * (float)(1 << l);
    const float tx = bx * scale, ty = by * scale, tz = bz * scale;
    const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
    const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
    const float fx = tx - flx, fy = ty - fly, fz = tz - flz;
    const float minfrac = fminf(fminf(fminf(fx, 1.f - fx), fminf(fy, 1.f - fy)), fminf(fz, 1.f - fz));
    const int dx = ((m.dx - 1) >> l) + 1, dy = ((m.dy - 1) >> l) + 1, dz = ((m.dz - 1) >> l) + 1;
    // shells below r0 hold no grid cell; shells above rmax hold none either
    const int r0 = max(0, max(max(max(-cx, cx - (dx - 1)), max(-cy, cy - (dy - 1))), max(-cz, cz - (dz - 1))));
    const int rmax = max(max(max(cx, dx - 1 - cx), max(cy, dy - 1 - cy)), max(cz, dz - 1 - cz));
    // a query farther outside the grid than max_d2 has nothing to find
    if (r0 >= 1) {
      const float gap = (float)(r0 - 1) * cell;
      if (gap * gap >= max_d2) return gap * gap;
    }
    float explored2 = 0.f;
    bool done = false;
    if (r0 <= 1) {
      // ---- shell 0: the query's own cell
      if (r0 == 0) {
        int s, e;
        if (cell_range(g, ((uint32_t)l << 27) | morton27((uint32_t)cx, (uint32_t)cy, (uint32_t)cz), s, e)) scan_range(g, s, e, qx, qy, qz, vis);
        const float gr = (minfrac - margin) * cell;
        if (gr > 0.f) {
          explored2 = gr * gr;
          vis.flush();
          if (vis.complete() < explored2 || explored2 >= max_d2) done = true;
        }
      }
      // ---- shell 1: mask of the 26 neighbours that can still hold something useful
      if (!done && rmax >= 1) {
        vis.flush();
        const float gxs[3] = {fx, 0.f, 1.f - fx}, gys[3] = {fy, 0.f, 1.f - fy}, gzs[3] = {fz, 0.f, 1.f - fz};
        const float b_units = fminf(vis.bound(), max_d2) / (cell * cell);
        uint32_t mask = 0;
#pragma unroll
        for (int c = 0; c < 27; c++) {
          if (c == 13) continue;
          const int a0 = c % 3, a1 = (c / 3) % 3, a2 = c / 9;
          const int ix = cx + a0 - 1, iy = cy + a1 - 1, iz = cz + a2 - 1;
          const float lbu = fmaxf(sqrtf(gxs[a0] * gxs[a0] + gys[a1] * gys[a1] + gzs[a2] * gzs[a2]) - margin, 0.f);
          const bool ok = ix >= 0 && ix < dx && iy >= 0 && iy < dy && iz >= 0 && iz < dz && !(lbu * lbu > b_units);
          mask |= ok ? (1u << c) : 0u;
        }
        // walk the set bits; the hash slot of the NEXT cell is fetched while the current cell's points are scanned
        auto cell_key = [&](int c) {
          const int a0 = c % 3, a1 = (c / 3) % 3, a2 = c / 9;
          return ((uint32_t)l << 27) | morton27((uint32_t)(cx + a0 - 1), (uint32_t)(cy + a1 - 1), (uint32_t)(cz + a2 - 1));
        };
        int c_cur = -1;
        uint32_t key_cur = 0;
        int4 slot_cur = make_int4(0, 0, 0, 0);
        if (mask) {
          c_cur = __ffs(mask) - 1;
          mask &= mask - 1;
          key_cur = cell_key(c_cur);
          slot_cur = first_slot(g, key_cur);
        }
        while (c_cur >= 0) {
          int c_nxt = -1;
          uint32_t key_nxt = 0;
          int4 slot_nxt = make_int4(0, 0, 0, 0);
          if (mask) {
            c_nxt = __ffs(mask) - 1;
            mask &= mask - 1;
            key_nxt = cell_key(c_nxt);
            slot_nxt = first_slot(g, key_nxt);
          }
          const int a0 = c_cur % 3, a1 = (c_cur / 3) % 3, a2 = c_cur / 9;
          const float gx = (a0 == 0) ? fx : ((a0 == 2) ? 1.f - fx : 0.f);
          const float gy = (a1 == 0) ? fy : ((a1 == 2) ? 1.f - fy : 0.f);
          const float gz = (a2 == 0) ? fz : ((a2 == 2) ? 1.f - fz : 0.f);
          const float lb = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz) - margin, 0.f) * cell;
          if (!(lb * lb > fminf(vis.bound(), max_d2))) {  // the bound may have shrunk since the mask was built
            int s, e;
            if (resolve_range(g, key_cur, slot_cur, s, e)) scan_range(g, s, e, qx, qy, qz, vis);
          }
          c_cur = c_nxt;
          key_cur = key_nxt;
          slot_cur = slot_nxt;
        }
        const float gr = (1.f + minfrac - margin) * cell;
        explored2 = gr * gr;
        vis.flush();
        if (vis.complete() < explored2 || explored2 >= max_d2) done = true;
      }
      if (!done && rmax <= 1 && top) return INFINITY;  // the whole grid fits in these shells
    }
    if (done) return explored2;
    if (l >= level_cap) return explored2;  // capped: the rest is the caller's
    if (top) {
      // ---- shells r >= 2 (or a query outside the grid): generic walk, top level only
      for (int r = max(r0, 2); r <= rmax; r++) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, dz - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, dy - 1);
        for (int iz = z0; iz <= z1; iz++) {
          const float gz = (iz > cz) ? ((float)iz - tz) : ((iz < cz) ? (tz - (float)(iz + 1)) : 0.f);
          const bool zface = (iz - cz == r) || (cz - iz == r);
          const uint32_t mz = morton_expand9((uint32_t)iz) << 2;
          for (int iy = y0; iy <= y1; iy++) {
            const float gy = (iy > cy) ? ((float)iy - ty) : ((iy < cy) ? (ty - (float)(iy + 1)) : 0.f);
            const bool face = zface || (iy - cy == r) || (cy - iy == r);
            const float gyz2 = gy * gy + gz * gz;
            const uint32_t myz = mz | (morton_expand9((uint32_t)iy) << 1);
            const int xa = cx - r, xb = cx + r;
            const int step = face ? 1 : 2 * r;  // face row: whole x range; otherwise the two x end caps
            for (int ix = xa; ix <= xb; ix += step) {
              if (ix < 0 || ix >= dx) continue;
              const float gx = (ix > cx) ? ((float)ix - tx) : ((ix < cx) ? (tx - (float)(ix + 1)) : 0.f);
              const float lb = fmaxf(sqrtf(gx * gx + gyz2) - margin, 0.f) * cell;
              if (lb * lb > fminf(vis.bound(), max_d2)) continue;
              int s, e;
              if (cell_range(g, ((uint32_t)l << 27) | myz | morton_expand9((uint32_t)ix), s, e)) scan_range(g, s, e, qx, qy, qz, vis);
            }
          }
        }
        const float gr = ((float)r + minfrac - margin) * cell;
        explored2 = gr * gr;
        vis.flush();
        if (vis.complete() < explored2 || explored2 >= max_d2) return explored2;
      }
      return INFINITY;  // the whole grid has been scanned
    }
    // next level; never coarser than the first level whose shell 1 is guaranteed to cover max_d2 (cell edge >= the radius):
    // beyond it the cells only get bigger, not the region that matters
    int l_next = min(min(l + level_step, level_cap), m.top_level);
    if (max_d2 < INFINITY) {
      const float rel = sqrtf(max_d2) * m.inv_cell;
      const int l_cover = (rel <= 1.f) ? 0 : min(m.top_level, (int)ceilf(log2f(rel)));
      l_next = min(l_next, max(l + 1, l_cover));
    }
    l = l_next;
    vis.flush();
    vis.restart();
  }
}

// Ball query: the visitor already holds a finite bound (squared radius) — from the previous iteration's correspondent,
// or from k real points near the query — so every point that can still improve the result lies inside the ball
// B(q, sqrt(bound)).  One pass over the cells that intersect that ball, at the level whose cell edge is about
// `cells_per_radius` radii: no shells, no escalation, and all lanes of a wave walk a box of similar (small) size.
// The result is exact for the same reason the kd-tree's is: nothing outside the ball can beat the bound, and the bound
// only shrinks while scanning.  Ties on the ball surface are visited (closed ball, lower caller index wins).
template <typename Visitor>
__device__ __forceinline__ void grid_ball_search(const GridView& g, float qx, float qy, float qz, Visitor& vis, float cells_per_radius) {
  const GridMeta& m = g.m;
  if (m.n <= 0) return;
  const float radius = sqrtf(vis.bound());
  const float rel = radius * m.inv_cell * cells_per_radius;  // wanted cell edge, in base cells
  const int l = (rel <= 1.f) ? 0 : min(m.top_level, (int)ceilf(log2f(rel)));
  const float scale = 1.0f / (float)(1 << l);
  const float cell = m.cell * (float)(1 << l);
  const float margin = 1.0e-3f;
  const float tx = cell_coord(qx, m.ox, m.inv_cell) * scale, ty = cell_coord(qy, m.oy, m.inv_cell) * scale, tz = cell_coord(qz, m.oz, m.inv_cell) * scale;
  const float rr = radius / cell + margin;
  const int dx = ((m.dx - 1) >> l) + 1, dy = ((m.dy - 1) >> l) + 1, dz = ((m.dz - 1) >> l) + 1;
  const int x0 = max((int)floorf(tx - rr), 0), x1 = min((int)floorf(tx + rr), dx - 1);
  const int y0 = max((int)floorf(ty - rr), 0), y1 = min((int)floorf(ty + rr), dy - 1);
  const int z0 = max((int)floorf(tz - rr), 0), z1 = min((int)floorf(tz + rr), dz - 1);
  const int cx = (int)floorf(tx), cy = (int)floorf(ty), cz = (int)floorf(tz);
  // own cell first: it fills the list / tightens the bound before the neighbours are tested
  if (cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1 && cz >= z0 && cz <= z1) {
    int s, e;
    if (cell_range(g, ((uint32_t)l << 27) | morton27((uint32_t)cx, (uint32_t)cy, (uint32_t)cz), s, e)) scan_range(g, s, e, qx, qy, qz, vis);
  }
  for (int iz = z0; iz <= z1; iz++) {
    const float gz = (iz > cz) ? ((float)iz - tz) : ((iz < cz) ? (tz - (float)(iz + 1)) : 0.f);
    const uint32_t mz = morton_expand9((uint32_t)iz) << 2;
    for (int iy = y0; iy <= y1; iy++) {
      const float gy = (iy > cy) ? ((float)iy - ty) : ((iy < cy) ? (ty - (float)(iy + 1)) : 0.f);
      const float gyz2 = gy * gy + gz * gz;
      const uint32_t myz = mz | (morton_expand9((uint32_t)iy) << 1);
      for (int ix = x0; ix <= x1; ix++) {
        if (ix == cx && iy == cy && iz == cz) continue;
        const float gx = (ix > cx) ? ((float)ix - tx) : ((ix < cx) ? (tx - (float)(ix + 1)) : 0.f);
        const float lb = (sqrtf(gx * gx + gyz2) - margin) * cell;
        if (lb > 0.f && lb * lb > vis.bound()) continue;
        int s, e;
        if (cell_range(g, ((uint32_t)l << 27) | myz | morton_expand9((uint32_t)ix), s, e)) scan_range(g, s, e, qx, qy, qz, vis);
      }
    }
  }
}

// ---- visitors -----------------------------------------------------------------------------------------------------
struct Nearest1 {
  static constexpr int kScanWidth = 4;
  const int* orig;
  float best_d2;
  int best_j;
  __device__ __forceinline__ float bound() const { return best_d2; }
  __device__ __forceinline__ float complete() const { return best_d2; }
  __device__ __forceinline__ void flush() {}
  __device__ __forceinline__ void restart() {}  // re-visiting a point cannot change a 1-NN result
  __device__ __forceinline__ void visit(int j, float d2) {
    if (d2 < best_d2 || (d2 == best_d2 && best_j >= 0 && j != best_j && orig[j] < orig[best_j])) { best_d2 = d2; best_j = j; }
  }
};

// Nearest + distance of the runner-up: what the incremental correspondence update needs (gicp.hip).  The search
// prunes with the runner-up distance, so both are exact inside the explored region; `limit` caps a ball query.
struct Nearest2 {
  static constexpr int kScanWidth = 4;
  const int* orig;
  float d1;
  int j1;
  float d2nd;
  float limit;
  __device__ __forceinline__ float bound() const { return fminf(d2nd, limit); }
  __device__ __forceinline__ float complete() const { return d1; }
  __device__ __forceinline__ void flush() {}
  __device__ __forceinline__ void restart() {}
  __device__ __forceinline__ void visit(int j, float dd) {
    if (j == j1) return;  // a coarser level re-visits points
    if (dd < d1 || (dd == d1 && j1 >= 0 && orig[j] < orig[j1]) || j1 < 0) {
      if (j1 >= 0) d2nd = fminf(d2nd, d1);
      d1 = dd;
      j1 = j;
    } else if (dd < d2nd) {
      d2nd = dd;
    }
  }
};

// Sorted (ascending by (d2, caller index)) list of the K best, fully in registers.
template <int K>
struct NearestK {
  static constexpr int kScanWidth = 1;
  const int* orig;
  int k;        // 1..K actually wanted
  float hint;   // k-th distance of the previous (finer) level: an upper bound of the true k-th distance
  float d2[K];
  int idx[K];
  __device__ __forceinline__ void init(const int* o, int kk) {
    orig = o;
    k = kk;
    hint = INFINITY;
    clear();
  }
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < K; i++) { d2[i] = INFINITY; idx[i] = -1; }
  }
  __device__ __forceinline__ float kth() const {
    if (k == K) return d2[K - 1];
    float b = INFINITY;
#pragma unroll
    for (int i = 0; i < K; i++)
      if (i == k - 1) b = d2[i];
    return b;
  }
  __device__ __forceinline__ float bound() const { return fminf(kth(), hint); }
  __device__ __forceinline__ float complete() const { return kth(); }
  __device__ __forceinline__ void flush() {}
  // squared distance to the farthest of k DISTINCT points of the cloud: an upper bound of the k-th NN distance
  __device__ __forceinline__ void set_hint(float d2max) { hint = d2max; }
  __device__ __forceinline__ void restart() {
    hint = fminf(hint, kth());
    clear();
  }
  __device__ __forceinline__ bool less(float da, int ja, float db, int jb) const {
    if (da < db) return true;
    if (da > db || jb < 0) return da < db;
    return orig[ja] < orig[jb];  // exact tie: lower caller index first
  }
  __device__ __forceinline__ void visit(int j, float dd) {
    if (dd > hint) return;  // cannot be among the k nearest (ties with the hint are kept)
    if (!less(dd, j, d2[K - 1], idx[K - 1])) return;
    bool placed = false;
#pragma unroll
    for (int i = K - 1; i > 0; i--) {
      const bool lt = less(dd, j, d2[i - 1], idx[i - 1]);
      const float nd = lt ? d2[i - 1] : dd;
      const int ni = lt ? idx[i - 1] : j;
      d2[i] = placed ? d2[i] : nd;
      idx[i] = placed ? idx[i] : ni;
      placed = placed || !lt;
    }
    if (!placed) { d2[0] = dd; idx[0] = j; }
  }
};

// Distances of the K nearest only (sorted ascending), maintained by a BRANCH-FREE insertion network: for a sorted list
// a <= b and a new value c, the new element at that position is median(a, b, c) — one v_med3_f32 per position, executed
// for every candidate whether it enters the list or not.  Under SIMT that beats a conditional insertion by a wide margin:
// the conditional version ran its ~170-instruction body whenever ANY of the 64 lanes wanted the candidate (21 % lane
// utilisation measured); this version has no data-dependent branch inside the candidate loop at all.  The identities of
// the neighbours are recovered afterwards by one ball query with the (exact) k-th distance — see k_knn_self.
template <int K>
struct NearestKD {
  static constexpr int kScanWidth = 8;
  int k;
  float hint;
  float d2[K];
  __device__ __forceinline__ void init(int kk) {
    k = kk;
    hint = INFINITY;
    clear();
  }
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < K; i++) d2[i] = INFINITY;
  }
  __device__ __forceinline__ float kth() const {
    if (k == K) return d2[K - 1];
    float b = INFINITY;
#pragma unroll
    for (int i = 0; i < K; i++)
      if (i == k - 1) b = d2[i];
    return b;
  }
  __device__ __forceinline__ float bound() const { return fminf(kth(), hint); }
  __device__ __forceinline__ float complete() const { return kth(); }
  __device__ __forceinline__ void flush() {}
  __device__ __forceinline__ void restart() {
    hint = fminf(hint, kth());
    clear();
  }
  __device__ __forceinline__ void visit(int, float dd) {
    dd = (dd > hint) ? INFINITY : dd;  // cannot be among the k nearest (ties with the hint are kept)
#pragma unroll
    for (int i = K - 1; i > 0; i--) d2[i] = __builtin_amdgcn_fmed3f(d2[i - 1], d2[i], dd);
    d2[0] = fminf(d2[0], dd);
  }
};

// Collects the indices of the points within a known k-th distance tau2: everything strictly closer, plus as many of the
// points AT tau2 as are needed to make k, lowest caller index first (the oracle's tie rule).
struct BallCollector {
  static constexpr int kScanWidth = 8;
  const int* orig;
  int* out;       // k slots
  int k;
  float tau2;
  int n_lt, n_eq; // strictly closer / exactly at tau2
  int eq_first;   // first point seen at tau2
  __device__ __forceinline__ float bound() const { return tau2; }
  __device__ __forceinline__ float complete() const { return tau2; }
  __device__ __forceinline__ void flush() {}
  __device__ __forceinline__ void restart() {}
  __device__ __forceinline__ void visit(int j, float dd) {
    if (dd < tau2) {
      if (n_lt < k) out[n_lt] = j;
      n_lt++;
    } else if (dd == tau2) {
      if (n_eq == 0) eq_first = j;
      n_eq++;
    }
  }
};

// Level at which a k-NN query of a point of the cloud itself should start: the finest level whose own cell already
// holds about k/3 points (then shells r <= 1 almost always suffice).  Only a speed heuristic — any level is exact.
__device__ __forceinline__ int knn_start_level(const GridView& g, float qx, float qy, float qz, int k, int need_div = 3) {
  const GridMeta& m = g.m;
  const int bx = min(max((int)floorf(cell_coord(qx, m.ox, m.inv_cell)), 0), m.dx - 1);
  const int by = min(max((int)floorf(cell_coord(qy, m.oy, m.inv_cell)), 0), m.dy - 1);
  const int bz = min(max((int)floorf(cell_coord(qz, m.oz, m.inv_cell)), 0), m.dz - 1);
  const int need = max(1, k / need_div);
  int l = 0;
  for (; l < m.top_level; l++)
    if (own_cell_count(g, bx, by, bz, l) >= need) break;
  return l;
}

}  // namespace mi355
