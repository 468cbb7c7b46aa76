// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  PARITY UNPINNED.
//
// Restatement of the PCL pieces behind the reference's callers.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "kdtree.hpp"
#include "oracle.h"

using namespace oracle;

extern "C" int orc_max_threads(void) { return omp_get_max_threads(); }

// A1 — PointsPreFiltering::distance_filter, /root/reference/points_prefiltering/src/points_prefiltering.cpp:102-112:
// keep iff min_distance_cloud_ < ||xyz|| (float norm promoted to double, strict compare); order preserved;
// max_distance_cloud_ is declared (:51) but never used.
extern "C" int orc_distance_filter(const float* pts, size_t n, double min_distance, float* out, size_t* out_n) {
  size_t o = 0;
  for (size_t i = 0; i < n; i++) {
    const float* p = pts + 4 * i;
    // Eigen's Vector3f::norm() = sqrt(cwiseAbs2().sum()); a fixed-size 3-vector is summed by redux_novec_unroller<.., 0, 3>
    // (Eigen/src/Core/Redux.h), which halves the range: func(x*x, func(y*y, z*z)) -> x*x + (y*y + z*z) in fp32.
    const float nrm = std::sqrt(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));
    const double distance = (double)nrm;
    if (min_distance < distance) {
      std::memcpy(out + 4 * o, p, 4 * sizeof(float));
      o++;
    }
  }
  *out_n = o;
  return 0;
}

// P1 — pcl::VoxelGrid<PointXYZI>::applyFilter as called from points_prefiltering.cpp:114-121 (leaf l,l,l;
// downsample_all_data_ = true so intensity is averaged too; min_points_per_voxel_ = 0) and from
// /root/reference/graph_based_slam/src/graph_based_slam.cpp:61,311-313 (0.5 m).  Output order = ascending
// linear voxel index.  Centroids accumulate in fp32 in index order within a voxel (CentroidPoint).
extern "C" int orc_voxel_grid(const float* pts, size_t n, float leaf, float* out, size_t* out_n, int* overflow) {
  *overflow = 0;
  *out_n = 0;
  if (n == 0) return 0;
  const float inv = 1.0f / leaf;  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  bool any = false;
  for (size_t i = 0; i < n; i++) {
    const float* p = pts + 4 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    any = true;
    for (int d = 0; d < 3; d++) { mn[d] = std::min(mn[d], p[d]); mx[d] = std::max(mx[d], p[d]); }
  }
  if (!any) return 0;
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
  const int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
  const int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)std::numeric_limits<int32_t>::max()) {
    // "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = input
    *overflow = 1;
    std::memcpy(out, pts, n * 4 * sizeof(float));
    *out_n = n;
    return 0;
  }
  int min_b[3], max_b[3], div_b[3];
  for (int d = 0; d < 3; d++) {
    min_b[d] = (int)std::floor(mn[d] * inv);
    max_b[d] = (int)std::floor(mx[d] * inv);
    div_b[d] = max_b[d] - min_b[d] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  struct IP { unsigned idx; unsigned pt; };
  std::vector<IP> iv;
  iv.reserve(n);
  for (size_t i = 0; i < n; i++) {
    const float* p = pts + 4 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    const int i0 = (int)(std::floor(p[0] * inv) - (float)min_b[0]);
    const int i1 = (int)(std::floor(p[1] * inv) - (float)min_b[1]);
    const int i2 = (int)(std::floor(p[2] * inv) - (float)min_b[2]);
    iv.push_back({(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
  }
  // PCL sorts on idx only (std::sort / spreadsort: order inside a voxel unspecified).  The oracle fixes it to
  // ascending point index (stable), the rule the HIP path follows too.
  std::stable_sort(iv.begin(), iv.end(), [](const IP& a, const IP& b) { return a.idx < b.idx; });
  size_t o = 0;
  for (size_t s = 0; s < iv.size();) {
    size_t e = s + 1;
    while (e < iv.size() && iv[e].idx == iv[s].idx) e++;
    float acc[4] = {0, 0, 0, 0};
    for (size_t k = s; k < e; k++) {
      const float* p = pts + 4 * (size_t)iv[k].pt;
      for (int d = 0; d < 4; d++) acc[d] += p[d];
    }
    const float cnt = (float)(e - s);
    for (int d = 0; d < 4; d++) out[4 * o + d] = acc[d] / cnt;
    o++;
    s = e;
  }
  *out_n = o;
  return 0;
}

extern "C" int orc_knn(const float* pts, size_t n, const float* queries, size_t m, int k, int* idx, float* d2,
                       int brute_force) {
  if (k <= 0 || (size_t)k > n) return -1;
  if (brute_force) {
#pragma omp parallel for schedule(dynamic, 16)
    for (long qi = 0; qi < (long)m; qi++) {
      std::vector<Neighbor> all(n);
      for (size_t j = 0; j < n; j++) all[j] = {dist2f(queries + 4 * qi, pts + 4 * j), (int)j};
      std::partial_sort(all.begin(), all.begin() + k, all.end());
      for (int j = 0; j < k; j++) { idx[(size_t)qi * k + j] = all[j].idx; d2[(size_t)qi * k + j] = all[j].d2; }
    }
    return 0;
  }
  KdTree tree;
  tree.build(pts, n, 4);
#pragma omp parallel for schedule(guided, 8)
  for (long qi = 0; qi < (long)m; qi++) tree.knn(queries + 4 * qi, k, idx + (size_t)qi * k, d2 + (size_t)qi * k);
  return 0;
}

// P2 — pcl::StatisticalOutlierRemoval::applyFilterIndices as called from points_prefiltering.cpp:132-140
// (mean_k 30, stddev 1.2): (mean_k+1)-NN, first hit (the query itself) skipped, mean of sqrt(d2) in double,
// threshold mean + mul * stddev with the (n-1) variance; keep iff distance <= threshold; order preserved.
extern "C" int orc_sor(const float* pts, size_t n, int mean_k, double stddev_mul, float* out, size_t* out_n,
                       uint8_t* keep_mask, float* mean_dist) {
  *out_n = 0;
  if (n == 0) return 0;
  if ((size_t)mean_k + 1 > n) return -1;
  KdTree tree;
  tree.build(pts, n, 4);
  std::vector<float> distances(n);
  const int kk = mean_k + 1;
#pragma omp parallel
  {
    std::vector<int> ni(kk);
    std::vector<float> nd(kk);
#pragma omp for schedule(guided, 8)
    for (long i = 0; i < (long)n; i++) {
      tree.knn(pts + 4 * i, kk, ni.data(), nd.data());
      double dist_sum = 0.0;
      for (int k = 1; k < kk; k++) dist_sum += std::sqrt((double)nd[k]);
      distances[i] = (float)(dist_sum / mean_k);
    }
  }
  double sum = 0, sq_sum = 0;
  for (size_t i = 0; i < n; i++) { sum += distances[i]; sq_sum += (double)distances[i] * (double)distances[i]; }
  const double valid = (double)n;
  const double mean = sum / valid;
  const double variance = (sq_sum - sum * sum / valid) / (valid - 1);
  const double stddev = std::sqrt(variance);
  const double thr = mean + stddev_mul * stddev;
  size_t o = 0;
  for (size_t i = 0; i < n; i++) {
    const bool keep = !((double)distances[i] > thr);
    if (keep_mask) keep_mask[i] = keep ? 1 : 0;
    if (mean_dist) mean_dist[i] = distances[i];
    if (keep) { std::memcpy(out + 4 * o, pts + 4 * i, 4 * sizeof(float)); o++; }
  }
  *out_n = o;
  return 0;
}

// P3 — pcl::transformPointCloud(in, out, Matrix4f) as called from
// /root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:282,291: xyz <- R*xyz + t in fp32, intensity copied.
// Specified evaluation order (shared with the HIP path): fmaf(r0,x, fmaf(r1,y, fmaf(r2,z, t))).
extern "C" int orc_transform(const float* pts, size_t n, const float* T, float* out) {
  for (size_t i = 0; i < n; i++) {
    const float x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
    for (int r = 0; r < 3; r++)
      out[4 * i + r] = std::fmaf(T[4 * r], x, std::fmaf(T[4 * r + 1], y, std::fmaf(T[4 * r + 2], z, T[4 * r + 3])));
    out[4 * i + 3] = pts[4 * i + 3];
  }
  return 0;
}

// P5 — pcl::Registration::getFitnessScore(max_range) as called from
// /root/reference/graph_based_slam/src/graph_based_slam.cpp:321: mean of SQUARED 1-NN distances of the
// transformed source into the target; PCL compares the squared distance with max_range as is.
extern "C" int orc_fitness(const float* src, size_t n, const float* tgt, size_t m, const float* T16, double max_range,
                           double* score) {
  if (m == 0) return -1;
  KdTree tree;
  tree.build(tgt, m, 4);
  std::vector<float> tp(4 * n);
  orc_transform(src, n, T16, tp.data());
  std::vector<float> dd(n);
#pragma omp parallel for schedule(guided, 8)
  for (long i = 0; i < (long)n; i++) {
    int j;
    tree.knn(&tp[4 * i], 1, &j, &dd[i]);
  }
  double fit = 0;
  long nr = 0;
  for (size_t i = 0; i < n; i++)
    if ((double)dd[i] <= max_range) { fit += (double)dd[i]; nr++; }
  *score = nr > 0 ? fit / (double)nr : std::numeric_limits<double>::max();
  return 0;
}
