// Voxel-hash nearest-neighbour search (device side).
//
// Replaces the FLANN kd-trees behind pcl::search::KdTree::nearestKSearch with a
// structure that is built by a sort and searched by coalesced cell-range scans:
//   * points are sorted by the linear index of their cubic cell (x fastest), so a cell is a contiguous range;
//   * an open-addressing hash maps cell key -> [start, end) (16-byte slots, one probe in the common case);
//   * a query scans Chebyshev shells r = 0, 1, 2, ... of cells around its own cell and stops as soon as the
//     k-th best distance is within the radius the scanned shells are guaranteed to cover — so the result is the
//     EXACT k-NN (same answer as the kd-tree; ties towards the lower point index), not an approximation.
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

// Continuous cell coordinate of a query along one axis, clamped so the int conversion cannot overflow.
__device__ __forceinline__ float cell_coord(float q, float o, float inv) { return fminf(fmaxf((q - o) * inv, -1.0e6f), 1.0e6f); }

// Generic shell search.  Visitor: float bound() — current pruning squared distance (k-th best; +inf while the
// list is not full); void visit(int j, float d2).  max_d2: nothing at or beyond this squared distance is wanted.
template <typename Visitor>
__device__ __forceinline__ void grid_search(const GridView& g, float qx, float qy, float qz, float max_d2, Visitor& vis) {
  const GridMeta& m = g.m;
  if (m.n <= 0) return;
  const float tx = cell_coord(qx, m.ox, m.inv_cell), ty = cell_coord(qy, m.oy, m.inv_cell), tz = cell_coord(qz, m.oz, m.inv_cell);
  const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
  const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
  const float fx = tx - flx, fy = ty - fly, fz = tz - flz;
  const float minfrac = fminf(fminf(fminf(fx, 1.f - fx), fminf(fy, 1.f - fy)), fminf(fz, 1.f - fz));
  // shells below r0 hold no grid cell; shells above rmax hold none either
  const int r0 = max(0, max(max(max(-cx, cx - (m.dx - 1)), max(-cy, cy - (m.dy - 1))), max(-cz, cz - (m.dz - 1))));
  const int rmax = max(max(max(cx, m.dx - 1 - cx), max(cy, m.dy - 1 - cy)), max(cz, m.dz - 1 - cz));
  const float cell = m.cell;
  const float margin = 1.0e-3f;  // in cell units: covers the fp32 fuzz of the cell lattice
  for (int r = r0; r <= rmax; r++) {
    const int z0 = max(cz - r, 0), z1 = min(cz + r, m.dz - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, m.dy - 1);
    for (int iz = z0; iz <= z1; iz++) {
      const float gz = (iz > cz) ? ((float)iz - tz) : ((iz < cz) ? (tz - (float)(iz + 1)) : 0.f);
      const bool zface = (iz - cz == r) || (cz - iz == r);
      for (int iy = y0; iy <= y1; iy++) {
        const float gy = (iy > cy) ? ((float)iy - ty) : ((iy < cy) ? (ty - (float)(iy + 1)) : 0.f);
        const bool face = zface || (iy - cy == r) || (cy - iy == r);
        const float gyz2 = gy * gy + gz * gz;
        const int xa = cx - r, xb = cx + r;
        // on a face row visit the whole x range, otherwise only the two x end caps
        const int step = face ? 1 : max(2 * r, 1);
        for (int ix = xa; ix <= xb; ix += step) {
          if (ix < 0 || ix >= m.dx) continue;
          const float gx = (ix > cx) ? ((float)ix - tx) : ((ix < cx) ? (tx - (float)(ix + 1)) : 0.f);
          // lower bound (metres) of the distance from the query to anything inside this cell
          const float lb = (sqrtf(gx * gx + gyz2) - margin) * cell;
          if (lb > 0.f && lb * lb > fminf(vis.bound(), max_d2)) continue;
          const uint32_t key = (uint32_t)ix + (uint32_t)m.dx * ((uint32_t)iy + (uint32_t)m.dy * (uint32_t)iz);
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
      if (vis.bound() <= gr2 || gr2 >= max_d2) break;
    }
  }
}

// ---- visitors -----------------------------------------------------------------------------------------------------
struct Nearest1 {
  const int* orig;
  float best_d2;
  int best_j;
  __device__ __forceinline__ float bound() const { return best_d2; }
  __device__ __forceinline__ void visit(int j, float d2) {
    if (d2 < best_d2 || (d2 == best_d2 && best_j >= 0 && orig[j] < orig[best_j])) { best_d2 = d2; best_j = j; }
  }
};

// Sorted (ascending by (d2, caller index)) list of the K best, fully in registers.
template <int K>
struct NearestK {
  const int* orig;
  int k;  // 1..K actually wanted
  float d2[K];
  int idx[K];
  __device__ __forceinline__ void init(const int* o, int kk) {
    orig = o;
    k = kk;
#pragma unroll
    for (int i = 0; i < K; i++) { d2[i] = INFINITY; idx[i] = -1; }
  }
  __device__ __forceinline__ float bound() const {
    if (k == K) return d2[K - 1];
    float b = INFINITY;
#pragma unroll
    for (int i = 0; i < K; i++)
      if (i == k - 1) b = d2[i];
    return b;
  }
  __device__ __forceinline__ bool less(float da, int ja, float db, int jb) const {
    if (da < db) return true;
    if (da > db || jb < 0) return da < db;
    return orig[ja] < orig[jb];  // exact tie: lower caller index first
  }
  __device__ __forceinline__ void visit(int j, float dd) {
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

}  // namespace mi355
