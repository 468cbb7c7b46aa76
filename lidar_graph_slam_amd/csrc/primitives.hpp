// Device-side building blocks for gfx950 (wave = 64 lanes): wave/block reductions in fp64, ordered-float
// atomics for bounding boxes.  Reductions use a fixed tree so results are bitwise reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
