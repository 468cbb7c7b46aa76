// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header).  PARITY UNPINNED.
//
// Exact k-nearest-neighbour search over float3 points: restates what the reference obtains from
// pcl::search::KdTree / pcl::KdTreeFLANN::nearestKSearch (FLANN single kd-tree index, L2_Simple in fp32,
// sorted results, eps = 0) — SURVEY.md §8(a) row P4 [UPSTREAM-RECALL]; call sites
// /root/reference/points_prefiltering/src/points_prefiltering.cpp:136-139 (SOR searcher) and the
// fast_gicp kd-trees behind /root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:149,162.
//
// Distance definition (shared, by specification, with the HIP path so that both pick the same neighbour):
//   d2 = fmaf(dz, dz, fmaf(dy, dy, dx*dx)),  dx = q.x - p.x  (all fp32)
// Ties are broken towards the LOWER point index (FLANN's tie order is traversal-dependent and therefore
// not a specification; lowest-index is the deterministic rule both implementations follow).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace oracle {

inline float dist2f(const float* q, const float* p) {
  const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
  return std::fmaf(dz, dz, std::fmaf(dy, dy, dx * dx));
}

struct Neighbor {
  float d2;
  int idx;
  bool operator<(const Neighbor& o) const { return d2 < o.d2 || (d2 == o.d2 && idx < o.idx); }
};

class KdTree {
 public:
  // pts: n points, `stride` floats apart, xyz at offset 0.
  void build(const float* pts, size_t n, size_t stride) {
    n_ = n;
    xyz_.resize(3 * n);
    idx_.resize(n);
    std::iota(idx_.begin(), idx_.end(), 0);
    src_ = pts;
    stride_ = stride;
    nodes_.clear();
    nodes_.reserve(2 * n / kLeaf + 16);
    if (n > 0) build_rec(0, (int)n);
    for (size_t i = 0; i < n; i++) {
      const float* p = pts + (size_t)idx_[i] * stride;
      xyz_[3 * i + 0] = p[0]; xyz_[3 * i + 1] = p[1]; xyz_[3 * i + 2] = p[2];
    }
  }
  size_t size() const { return n_; }

  // k nearest of q, ascending by (d2, idx).  Returns number found (min(k, n)).
  int knn(const float* q, int k, int* out_idx, float* out_d2) const {
    if (n_ == 0 || k <= 0) return 0;
    std::vector<Neighbor> heap;  // max-heap on (d2, idx)
    heap.reserve(k + 1);
    search(0, q, k, heap);
    std::sort_heap(heap.begin(), heap.end());
    const int m = (int)heap.size();
    for (int i = 0; i < m; i++) { out_idx[i] = heap[i].idx; out_d2[i] = heap[i].d2; }
    return m;
  }

 private:
  static constexpr int kLeaf = 16;
  struct Node {
    int lo, hi;      // point range [lo, hi)
    int left, right; // children (-1 => leaf)
    int dim;
    float split;
  };

  int build_rec(int lo, int hi) {
    const int me = (int)nodes_.size();
    nodes_.push_back({lo, hi, -1, -1, 0, 0.f});
    if (hi - lo <= kLeaf) return me;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; i++) {
      const float* p = src_ + (size_t)idx_[i] * stride_;
      for (int d = 0; d < 3; d++) { mn[d] = std::min(mn[d], p[d]); mx[d] = std::max(mx[d], p[d]); }
    }
    int dim = 0;
    for (int d = 1; d < 3; d++)
      if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
    if (!(mx[dim] > mn[dim])) return me;  // all identical: keep as (large) leaf
    const int mid = (lo + hi) / 2;
    std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi, [&](int a, int b) {
      const float va = src_[(size_t)a * stride_ + dim], vb = src_[(size_t)b * stride_ + dim];
      return va < vb || (va == vb && a < b);
    });
    const float split = src_[(size_t)idx_[mid] * stride_ + dim];
    const int l = build_rec(lo, mid);
    const int r = build_rec(mid, hi);
    nodes_[me].left = l; nodes_[me].right = r; nodes_[me].dim = dim; nodes_[me].split = split;
    return me;
  }

  void search(int ni, const float* q, int k, std::vector<Neighbor>& heap) const {
    const Node& nd = nodes_[ni];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; i++) {
        Neighbor c{dist2f(q, &xyz_[3 * i]), idx_[i]};
        if ((int)heap.size() < k) {
          heap.push_back(c);
          std::push_heap(heap.begin(), heap.end());
        } else if (c < heap.front()) {
          std::pop_heap(heap.begin(), heap.end());
          heap.back() = c;
          std::push_heap(heap.begin(), heap.end());
        }
      }
      return;
    }
    // left holds values <= split (ties by index may land either side: both sides satisfy |q-p| >= |q-split|
    // for the far child only when the plane really separates, so test the plane with <=).
    const float diff = q[nd.dim] - nd.split;
    const int near = diff < 0 ? nd.left : nd.right;
    const int far = diff < 0 ? nd.right : nd.left;
    search(near, q, k, heap);
    // Monotone fp32 rounding makes `diff*diff` a lower bound of every far-side computed d2 (see dist2f), so
    // strict '>' pruning never discards a candidate that would win or tie.
    if ((int)heap.size() < k || !(diff * diff > heap.front().d2)) search(far, q, k, heap);
  }

  size_t n_ = 0;
  const float* src_ = nullptr;
  size_t stride_ = 0;
  std::vector<float> xyz_;
  std::vector<int> idx_;
  std::vector<Node> nodes_;
};

}  // namespace oracle
