// Device-side building blocks for gfx950 (wave = 64 lanes): wave/block reductions in fp64, ordered-float
// atomics for bounding boxes.  Reductions use a fixed tree so results are bitwise reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace mi355 {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;  // valid in lane 0
}
// Sum NV doubles per thread across a block of NW waves.  Result valid in thread 0 (acc[] overwritten).
// `smem` must hold NW * NV doubles.
template <int NV, int NW>
__device__ __forceinline__ void block_sum(double* acc, double* smem) {
#pragma unroll
  for (int k = 0; k < NV; k++) acc[k] = wave_sum(acc[k]);
  const int w = wave_id(), l = lane_id();
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) smem[w * NV + k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double s = smem[k];
      for (int ww = 1; ww < NW; ww++) s += smem[ww * NV + k];
      acc[k] = s;
    }
  }
  __syncthreads();
}

// The same sums through an LDS transpose, for kernels that reduce MANY values per thread: the butterfly above costs 12 ds_bpermute_b32
// + 6 v_add_f64 per value and wave (738 instructions for the 41 sums of k_gicp_linearize - a quarter of what a block of that kernel
// executes in its whole life), this one ~10: sixteen values at a time every lane leaves its 16 in a row of `wave_buf`, lane
// (column c = lane & 15, quarter q = lane >> 4) adds column c over the 16 lanes of quarter q in lane order, and the four quarters meet
// in two butterfly steps.  Fixed order => bitwise reproducible.  Returns, in thread k < NV, the block's sum of value k (0 elsewhere).
// wave_buf: 64 x 17 doubles private to the calling wave; smem: NW * NV doubles.
constexpr int kBlockSumRow = 17;
template <int NV, int NW>
__device__ __forceinline__ double block_sum_columns(const double* acc, double* wave_buf, double* smem) {
  const int lane = lane_id(), w = wave_id();
  const int col = lane & 15, q = lane >> 4;
#pragma unroll
  for (int c0 = 0; c0 < NV; c0 += 16) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the rows of the previous round have been read)
#pragma unroll
    for (int k = 0; k < 16; k++)
      if (c0 + k < NV) wave_buf[lane * kBlockSumRow + k] = acc[c0 + k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += wave_buf[(q * 16 + r) * kBlockSumRow + col];
    s += __shfl_xor(s, 16, kWave);
    s += __shfl_xor(s, 32, kWave);
    if (lane < 16 && c0 + lane < NV) smem[w * NV + c0 + lane] = s;
  }
  __syncthreads();
  double total = 0.0;
  if ((int)threadIdx.x < NV) {
#pragma unroll
    for (int ww = 0; ww < NW; ww++) total += smem[ww * NV + threadIdx.x];
  }
  return total;
}

// The value of lane + 1 (0 in lane 63): one whole-wave DPP shift, no LDS round trip.
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ float wave_shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true)); }

// ---- wave / row all-reductions without LDS ------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 on gfx950: an LDS-crossbar round trip (~100 cycles) per step, and the six steps of a
// wave reduction depend on each other.  The tile kernels reduce a bounding box (6 values), group boxes, radii and counts per tile and
// per round - ~100 such steps per 64-point tile of k_knn_cov_tile, more per tile of k_gicp_correspond_tile.  DPP does the steps
// inside a row of 16 lanes in the VALU itself (quad permutes, then the two mirrors: every lane ends with its row's result), and the
// four rows are combined through v_readlane - the result is wave-uniform and sits in an SGPR.  Exact for min / max / integer sums
// whatever the order.  Callers are in wave-uniform control flow (all 64 lanes active).
template <typename T, typename Op>
__device__ __forceinline__ T row_reduce16(T v, Op op) {  // every lane of an aligned group of 16 gets the group's result
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
  auto dpp = [](T x, auto ctrl) {
    constexpr int c = decltype(ctrl)::value;
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &x, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; k++) w[k] = __builtin_amdgcn_update_dpp(w[k], w[k], c, 0xF, 0xF, false);
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
  };
  v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
  v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
  v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));  // row_mirror
  return v;
}
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce64(T v, Op op) {  // wave-uniform result
  v = row_reduce16(v, op);
  auto lane_of = [&](int l) {
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; k++) w[k] = __builtin_amdgcn_readlane(w[k], l);
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
  };
  return op(op(lane_of(0), lane_of(16)), op(lane_of(32), lane_of(48)));
}

// XCD-aware work mapping (perf only, never correctness).  MI355X deals workgroups round-robin over its 8 XCDs, each with
// a private 4 MB L2; consecutive block ids therefore land on DIFFERENT L2s.  A 1-D grid of ceil(S/8)*8*G blocks is
// remapped so that all G blocks of one item (scan pair / cloud) share blockIdx % 8, i.e. one XCD: the item's points,
// covariances and hash table then stay resident in that XCD's L2 while it is being worked on, instead of every XCD
// touching every item.
__device__ __forceinline__ bool xcd_item_chunk(int G, int S, int& item, int& chunk) {
  const int id = blockIdx.x;
  if (S < 8) {  // fewer items than XCDs: spread every item over the whole chip instead
    item = id / G;
    chunk = id % G;
    return item < S;
  }
  const int xcd = id & 7, slot = id >> 3;
  item = (slot / G) * 8 + xcd;
  chunk = slot % G;
  return item < S;
}
inline int xcd_grid(int G, int S) { return (S < 8) ? S * G : ((S + 7) / 8) * 8 * G; }

// float <-> order-preserving uint (for atomicMin/atomicMax bounding boxes)
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

}  // namespace mi355
