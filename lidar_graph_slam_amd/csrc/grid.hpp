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

__device__ __forceinline__ bool cell_range(const GridView& g, uint32_t key, int& s, int& e) {
  uint32_t h = hash_slot(key, g.m.hshift);
  for (;;) {
    const int4 sl = *reinterpret_cast<const int4*>(&g.table[h]);
    if ((uint32_t)sl.x == key) { s = sl.y; e = sl.z; return true; }
    if ((uint32_t)sl.x == kEmptyKey) return false;
    h = (h + 1) & g.m.hmask;
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

// Visitor interface:
//   float bound()            pruning squared distance: nothing farther can enter the result
//   float complete()         k-th best squared distance of the CURRENT scan (+inf while the list is not full)
//   void  visit(j, d2)       candidate (sorted index j)
//   void  restart()          called when the search escalates to a coarser level (the scan starts over)
// max_d2: nothing at or beyond this squared distance is wanted.  level0: level to start at; level_step: levels to
// climb per escalation.
template <typename Visitor>
__device__ __forceinline__ void grid_search(const GridView& g, float qx, float qy, float qz, float max_d2, Visitor& vis, int level0, int level_step) {
  const GridMeta& m = g.m;
  if (m.n <= 0) return;
  const float bx = cell_coord(qx, m.ox, m.inv_cell), by = cell_coord(qy, m.oy, m.inv_cell), bz = cell_coord(qz, m.oz, m.inv_cell);
  const float margin = 1.0e-3f;  // in cell units: covers the fp32 fuzz of the cell lattice
  int l = min(level0, m.top_level);
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
      const float gap = ((float)(r0 - 1) + 0.f) * cell;
      if (gap * gap >= max_d2) return;
    }
    const int rlast = top ? rmax : min(rmax, 1);
    bool done = false;
    for (int r = r0; r <= rlast; r++) {
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
          // on a face row visit the whole x range, otherwise only the two x end caps
          const int step = face ? 1 : max(2 * r, 1);
          for (int ix = xa; ix <= xb; ix += step) {
            if (ix < 0 || ix >= dx) continue;
            const float gx = (ix > cx) ? ((float)ix - tx) : ((ix < cx) ? (tx - (float)(ix + 1)) : 0.f);
            // lower bound (metres) of the distance from the query to anything inside this cell
            const float lb = (sqrtf(gx * gx + gyz2) - margin) * cell;
            if (lb > 0.f && lb * lb > fminf(vis.bound(), max_d2)) continue;
            const uint32_t key = ((uint32_t)l << 27) | myz | morton_expand9((uint32_t)ix);
            int s, e;
            if (!cell_range(g, key, s, e)) continue;
            for (int j = s; j < e; j++) {
              const float4 p = g.pts[j];
              vis.visit(j, dist2f(qx, qy, qz, p.x, p.y, p.z));
            }
          }
        }
      }
      // every point closer than gr has now been seen
      const float gr = ((float)r + minfrac - margin) * cell;
      if (gr > 0.f) {
        const float gr2 = gr * gr;
        if (vis.complete() < gr2 || gr2 >= max_d2) { done = true; break; }
      }
    }
    if (done || top) return;  // top level: the whole grid has been scanned when the shells run out
    l = min(l + level_step, m.top_level);
    vis.restart();
  }
}

// ---- visitors -----------------------------------------------------------------------------------------------------
struct Nearest1 {
  const int* orig;
  float best_d2;
  int best_j;
  __device__ __forceinline__ float bound() const { return best_d2; }
  __device__ __forceinline__ float complete() const { return best_d2; }
  __device__ __forceinline__ void restart() {}  // re-visiting a point cannot change a 1-NN result
  __device__ __forceinline__ void visit(int j, float d2) {
    if (d2 < best_d2 || (d2 == best_d2 && best_j >= 0 && j != best_j && orig[j] < orig[best_j])) { best_d2 = d2; best_j = j; }
  }
};

// Sorted (ascending by (d2, caller index)) list of the K best, fully in registers.
template <int K>
struct NearestK {
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

// Level at which a k-NN query of a point of the cloud itself should start: the finest level whose own cell already
// holds about k/3 points (then shells r <= 1 almost always suffice).  Only a speed heuristic — any level is exact.
__device__ __forceinline__ int knn_start_level(const GridView& g, float qx, float qy, float qz, int k) {
  const GridMeta& m = g.m;
  const int bx = min(max((int)floorf(cell_coord(qx, m.ox, m.inv_cell)), 0), m.dx - 1);
  const int by = min(max((int)floorf(cell_coord(qy, m.oy, m.inv_cell)), 0), m.dy - 1);
  const int bz = min(max((int)floorf(cell_coord(qz, m.oz, m.inv_cell)), 0), m.dz - 1);
  const int need = max(1, k / 3);
  int l = 0;
  for (; l < m.top_level; l++)
    if (own_cell_count(g, bx, by, bz, l) >= need) break;
  return l;
}

}  // namespace mi355
