// Batched stable LSD radix sort (8-bit digits) and batched exclusive scan for gfx950.
//
// Why hand-written: the voxel-grid downsample and the NN voxel-hash build both
// reduce to "sort B independent segments of (cell key, point index) pairs"; library segmented sorts launch per
// segment or fall back to slow paths at 10^4-10^6 elements per segment.  Here one launch covers every cloud of
// the batch (blockIdx.y = cloud), each 64-lane wave owns a 1 024-element sub-tile, ranks elements with wave
// ballots (no LDS atomics on the scatter path) and keeps its 256 digit counters in LDS.  Stability makes the
// order inside a voxel "ascending point index" — the deterministic rule the oracle follows too.
#include "engine.hpp"
#include "primitives.hpp"

namespace mi355 {

constexpr int kSortRounds = 16;                    // rounds of 64 elements per wave sub-tile
constexpr int kSubTile = kWave * kSortRounds;      // 1024
constexpr int kSortWaves = 4;
constexpr int kSortBlock = kSortWaves * kWave;     // 256 threads
constexpr int kBlockTile = kSubTile * kSortWaves;  // 4096 elements per block

// hist[c][digit][subtile]
__global__ __launch_bounds__(kSortBlock) void k_sort_hist(const uint32_t* __restrict__ keys, const int* __restrict__ counts, int* __restrict__ hist,
                                                          size_t cap, int nsub, int shift) {
  __shared__ int cnt[kSortWaves][256];
  const int c = blockIdx.y;
  const int n = counts[c];
  const int w = wave_id(), l = lane_id();
  const int sub = blockIdx.x * kSortWaves + w;
  if ((long)blockIdx.x * kBlockTile >= n) {
    // still publish zeros so the scan sees a defined histogram
    if (sub < nsub)
      for (int d = l; d < 256; d += kWave) hist[((size_t)c * 256 + d) * nsub + sub] = 0;
    return;
  }
  for (int d = l; d < 256; d += kWave) cnt[w][d] = 0;
  __builtin_amdgcn_wave_barrier();
  const uint32_t* k = keys + (size_t)c * cap;
  const int base = sub * kSubTile;
#pragma unroll 4
  for (int r = 0; r < kSortRounds; r++) {
    const int i = base + r * kWave + l;
    if (i < n) atomicAdd(&cnt[w][(k[i] >> shift) & 255u], 1);
  }
  __builtin_amdgcn_wave_barrier();
  if (sub < nsub)
    for (int d = l; d < 256; d += kWave) hist[((size_t)c * 256 + d) * nsub + sub] = cnt[w][d];
}

// Exclusive scan of the histogram of each cloud (digit-major), in place.  Only the sub-tiles the cloud actually fills take
// part (a 16 k-point cloud in a 64 k-point segment uses 17 of the 64 columns of every digit row).
__global__ __launch_bounds__(1024) void k_sort_scan(int* __restrict__ hist, const int* __restrict__ counts, int nsub) {
  __shared__ int wsum[16];
  int* h = hist + (size_t)blockIdx.x * 256 * nsub;
  const int used = min(nsub, max(1, (counts[blockIdx.x] + kSubTile - 1) / kSubTile));
  const int total = 256 * used;
  const int per = (total + 1023) / 1024;
  const int beg = min(threadIdx.x * per, total);
  const int end = min(beg + per, total);
  auto at = [&](int e) -> int& { return h[(e / used) * nsub + (e % used)]; };
  int s = 0;
  for (int i = beg; i < end; i++) s += at(i);
  // block exclusive scan of s
  int v = s;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int t = __shfl_up(v, off, kWave);
    if (lane_id() >= off) v += t;
  }
  if (lane_id() == kWave - 1) wsum[wave_id()] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 16; i++) { const int t = wsum[i]; wsum[i] = run; run += t; }
  }
  __syncthreads();
  int run = wsum[wave_id()] + v - s;
  for (int i = beg; i < end; i++) { const int t = at(i); at(i) = run; run += t; }
}

__global__ __launch_bounds__(kSortBlock) void k_sort_scatter(const uint32_t* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out, int* __restrict__ vals_out,
                                                             const int* __restrict__ counts, const int* __restrict__ hist, size_t cap, int nsub,
                                                             int shift) {
  __shared__ int off[kSortWaves][256];
  const int c = blockIdx.y;
  const int n = counts[c];
  if ((long)blockIdx.x * kBlockTile >= n) return;
  const int w = wave_id(), l = lane_id();
  const int sub = blockIdx.x * kSortWaves + w;
  if (sub >= nsub) return;
  for (int d = l; d < 256; d += kWave) off[w][d] = hist[((size_t)c * 256 + d) * nsub + sub];
  __builtin_amdgcn_wave_barrier();
  const uint32_t* ki = keys_in + (size_t)c * cap;
  const int* vi = vals_in + (size_t)c * cap;
  uint32_t* ko = keys_out + (size_t)c * cap;
  int* vo = vals_out + (size_t)c * cap;
  const int base = sub * kSubTile;
  const unsigned long long lt_mask = (1ull << l) - 1ull;
  for (int r = 0; r < kSortRounds; r++) {
    const int i = base + r * kWave + l;
    const bool valid = i < n;
    uint32_t key = 0;
    int val = 0;
    if (valid) { key = ki[i]; val = vi[i]; }
    const uint32_t d = (key >> shift) & 255u;
    // lanes of this round holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    int pos = 0;
    if (valid) pos = off[w][d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) off[w][d] += __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (valid) { ko[pos] = key; vo[pos] = val; }
  }
}

void sort_pairs(hipStream_t s, Workspace& ws, int B, size_t cap, const int* d_counts, int passes) {
  const int nsub = div_up((long)cap, kSubTile);
  ws.hist.ensure((size_t)B * 256 * nsub);
  const dim3 grid(div_up((long)cap, kBlockTile), B);
  uint32_t* ka = ws.keys_a.p; uint32_t* kb = ws.keys_b.p;
  int* va = ws.vals_a.p; int* vb = ws.vals_b.p;
  TimedLaunch tl(KF_SORT, s);
  for (int p = 0; p < passes; p++) {
    const int shift = 8 * p;
    hipLaunchKernelGGL(k_sort_hist, grid, dim3(kSortBlock), 0, s, ka, d_counts, ws.hist.p, cap, nsub, shift);
    hipLaunchKernelGGL(k_sort_scan, dim3(B), dim3(1024), 0, s, ws.hist.p, d_counts, nsub);
    hipLaunchKernelGGL(k_sort_scatter, grid, dim3(kSortBlock), 0, s, ka, va, kb, vb, d_counts, ws.hist.p, cap, nsub, shift);
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  MI_LAUNCH_CHECK();
  if (passes & 1) {  // result sits in the b buffers: swap the workspace handles so callers always read *_a
    std::swap(ws.keys_a.p, ws.keys_b.p);
    std::swap(ws.keys_a.n, ws.keys_b.n);
    std::swap(ws.vals_a.p, ws.vals_b.p);
    std::swap(ws.vals_a.n, ws.vals_b.n);
  }
}

// ---- batched exclusive scan of int flags ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan_local(const int* __restrict__ flags, const int* __restrict__ counts, int* __restrict__ local,
                                                    int* __restrict__ blocksum, size_t cap, int nblk) {
  __shared__ int wsum[4];
  const int c = blockIdx.y;
  const int n = counts[c];
  const int base = blockIdx.x * kScanTile;
  if (base >= n) {
    if (threadIdx.x == 0) blocksum[(size_t)c * nblk + blockIdx.x] = 0;
    return;
  }
  const int* f = flags + (size_t)c * cap;
  int* lo = local + (size_t)c * cap;
  const int i0 = base + threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) v[j] = (i0 + j < n) ? f[i0 + j] : 0;
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
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (i0 + j < n) lo[i0 + j] = run;
    run += v[j];
  }
  if (threadIdx.x == 255) blocksum[(size_t)c * nblk + blockIdx.x] = wbase + incl;
}

__global__ __launch_bounds__(256) void k_scan_blocks(int* __restrict__ blocksum, int* __restrict__ totals, int nblk) {
  // one block per cloud; serial-in-chunks exclusive scan over nblk block sums
  __shared__ int wsum[4];
  int* b = blocksum + (size_t)blockIdx.x * nblk;
  const int per = (nblk + 255) / 256;
  const int beg = threadIdx.x * per, end = min(beg + per, nblk);
  int s = 0;
  for (int i = beg; i < end; i++) s += b[i];
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
  for (int i = beg; i < end; i++) { const int t = b[i]; b[i] = run; run += t; }
  if (threadIdx.x == 255) totals[blockIdx.x] = wbase + incl;
}

void scan_flags(hipStream_t s, Workspace& ws, int B, size_t cap, const int* d_counts) {
  const int nblk = div_up((long)cap, kScanTile);
  ws.scan_local.ensure((size_t)B * cap);
  ws.scan_block.ensure((size_t)B * nblk);
  ws.totals.ensure(B);
  hipLaunchKernelGGL(k_scan_local, dim3(nblk, B), dim3(256), 0, s, ws.flags.p, d_counts, ws.scan_local.p, ws.scan_block.p, cap, nblk);
  hipLaunchKernelGGL(k_scan_blocks, dim3(B), dim3(256), 0, s, ws.scan_block.p, ws.totals.p, nblk);
  MI_LAUNCH_CHECK();
}

}  // namespace mi355
