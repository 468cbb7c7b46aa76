// Batched stable LSD radix sort (8-bit digits) and batched exclusive scan for gfx950.
//
// Why hand-written: the voxel-grid downsample and the NN voxel-hash build both
// reduce to "sort B independent segments of (cell key, point index) pairs"; library segmented sorts launch per
// segment or fall back to slow paths at 10^4-10^6 elements per segment.  Here one launch covers every cloud of
// the batch (blockIdx.y = cloud), each 64-lane wave owns a 1 024-element sub-tile, ranks elements with wave
// ballots (no LDS atomics on the scatter path) and keeps its 256 digit counters in LDS.  Stability makes the
// order inside a voxel "ascending point index" — the deterministic rule the oracle follows too.
#include "engine.hpp"
#include "grid.hpp"
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

// ---- in-LDS tile sort + merges: the path for batch-sized segments ------------------------------------------------------
// The global LSD sort above moves every (key, index) pair through HBM once per pass with scattered 4-byte writes (each wave
// spreads its 64 elements over up to 64 digit buckets) and needs 3 launches per pass; at 1 024 clouds x 64 k points it ran
// at 6 % of the HBM roofline (the k_sort_scan launches alone: 0.9 ms each, latency-bound).  Here a 1 024-thread block keeps a
// TILE of up to 24 576 elements in registers and LDS for ALL passes: per pass every wave ranks its 1 536 consecutive elements
// with wave ballots (as k_sort_scatter does), the 16 x 256 counters are scanned once, and the elements are permuted through
// LDS.  HBM sees one coalesced read and one coalesced write.  A voxelised scan (~16 k points) is one tile = one launch; a raw
// 64 k-point scan is 3 tiles followed by two rounds of pairwise merge-path merges.  Values are the element's index in its
// segment (all callers sort (key, i) pairs), so a tile carries 16-bit local indices only.
constexpr int kTsWaves = 16, kTsThreads = kTsWaves * kWave, kTsE = 24;
constexpr int kTsSpan = kWave * kTsE;               // 1 536 consecutive elements per wave
constexpr int kTsTile = kTsSpan * kTsWaves;         // 24 576 elements per tile
constexpr int kMergeChunk = 4096, kMergeThreads = 256, kMergeE = kMergeChunk / kMergeThreads;

// tile geometry of a cloud of n elements: nt tiles of tl elements (the last one shorter)
__device__ __forceinline__ void tile_geometry(int n, int& nt, int& tl) {
  nt = max(1, (n + kTsTile - 1) / kTsTile);
  tl = (((n + nt - 1) / nt) + kWave - 1) / kWave * kWave;
}

// A cloud of ONE tile is written straight to (keys_final, vals_final) — where the merge rounds would deliver it — and the merge
// kernels skip it: a voxelised scan costs one launch's worth of traffic, not three.
// The LAST tile of a cloud with an odd number of tiles (the third of a raw 64 k-point scan) has no partner in the first merge round,
// which used to copy it: it is written straight to (keys_r1, vals_r1), where that round delivers, and the round skips it (round 6: a third
// of the first round's traffic).  That may be the buffer the tile's own keys were read from: every thread holds its elements in
// registers by then, and no other block reads that range.
// VOXEL: the keys are the voxel indices of the points vpts (VoxelKeySource), computed here - keys_in is not read.
template <bool VOXEL>
__global__ __launch_bounds__(kTsThreads) void k_sort_tile(const uint32_t* keys_in, uint32_t* keys_out, int* vals_out,
                                                          uint32_t* __restrict__ keys_final, int* __restrict__ vals_final,
                                                          uint32_t* keys_r1, int* vals_r1,
                                                          const int* __restrict__ counts, size_t cap, int passes_max,
                                                          const int* __restrict__ seg_passes, const float4* __restrict__ vpts,
                                                          const GridMeta* __restrict__ vmeta, double min_distance) {
  __shared__ uint32_t keyL[kTsTile];
  __shared__ uint16_t idxL[kTsTile];
  __shared__ uint16_t cnt[kTsWaves][256];
  __shared__ int wtot[kTsWaves];
  const int c = blockIdx.y;
  const int n = counts[c];
  int nt, tl;
  tile_geometry(n, nt, tl);
  if ((int)blockIdx.x >= nt) return;
  if (nt == 1) { keys_out = keys_final; vals_out = vals_final; }
  else if ((nt & 1) && (int)blockIdx.x == nt - 1) { keys_out = keys_r1; vals_out = vals_r1; }
  const int passes = seg_passes ? min(passes_max, max(1, seg_passes[c])) : passes_max;
  const int base = blockIdx.x * tl;
  const int len = max(0, min(tl, n - base));
  const int w = wave_id(), l = lane_id();
  const size_t o = (size_t)c * cap + base;
  uint32_t key[kTsE];
  uint32_t il[kTsE];  // local index (high half) | rank inside the wave for the current pass (low half)
  // The tile's elements are dealt evenly to the 16 waves, `span` consecutive ones each (a multiple of 64): element p lives in wave
  // p / span, round (p % span) / 64, lane p % 64; padding sorts to the end (all-ones key, last in the stable order).  A voxelised scan
  // (13 - 17 k points: two of a cloud's three sorts) then takes 13 - 17 rounds per wave, not the 24 of a full tile of which seven waves
  // ranked nothing but padding.
  // (positions as p0 + a constant, bounds as a constant against lim: one register each instead of 24 live positions)
  const int span = ((len + kTsThreads - 1) / kTsThreads) * kWave;
  const int rounds = span / kWave;  // <= kTsE
  const int p0 = w * span + l;
  const int lim = len - p0;  // element r of this lane exists iff r * kWave < lim
  if constexpr (VOXEL) {
    const GridMeta vm = vmeta[c];
    const float4* pin = vpts + o + p0;
    constexpr int kLoadGroup = 6;  // points in flight per lane (24 float4 at once would not fit beside the 48 key / index registers)
    static_assert(kTsE % kLoadGroup == 0, "load groups");
#pragma unroll
    for (int r0 = 0; r0 < kTsE; r0 += kLoadGroup) {
      float4 pv[kLoadGroup];
#pragma unroll
      for (int u = 0; u < kLoadGroup; u++) {
        const int r = r0 + u;
        pv[u] = (r * kWave < lim && r < rounds) ? pin[r * kWave] : make_float4(NAN, NAN, NAN, 0.f);
      }
#pragma unroll
      for (int u = 0; u < kLoadGroup; u++) {
        const int r = r0 + u;
        key[r] = (r * kWave < lim && r < rounds) ? voxel_index_key(vm, pv[u], min_distance) : 0xFFFFFFFFu;
        il[r] = (uint32_t)(p0 + r * kWave) << 16;
      }
    }
  } else {
    const uint32_t* kin = keys_in + o + p0;
#pragma unroll
    for (int r = 0; r < kTsE; r++) {
      key[r] = (r * kWave < lim && r < rounds) ? kin[r * kWave] : 0xFFFFFFFFu;
      il[r] = (uint32_t)(p0 + r * kWave) << 16;
    }
  }
  // (fully unrolled: carried through a rolled loop the two register arrays become 32-wide vector values that the compiler
  // copies and spills wholesale — 4 KB of scratch per lane)
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
    if (pass >= passes) break;
    const int shift = 8 * pass;
    for (int d = l; d < 256; d += kWave) cnt[w][d] = 0;
    __builtin_amdgcn_wave_barrier();
    // ---- rank inside the wave: elements of earlier rounds first, then lower lanes
#pragma unroll
    for (int r = 0; r < kTsE; r++) {
      if (r < rounds) {  // (block-uniform; no `break`: the loop must unroll for key[] / il[] to stay in registers)
        const uint32_t d = (key[r] >> shift) & 255u;
        // the lanes holding the same digit: per bit of the digit one ballot, and peers &= ~(own bit ^ the ballot) - ONE v_bitop3_b32 per
        // half of the mask (gfx950: any three-input boolean; table 0x90 = a & ~(b ^ c)) where and / xor / not were three; the rank is
        // the peers below the lane (v_mbcnt).  The tile sort is bound by the VALU work of this loop (valu_issue_frac 0.81).
        uint32_t plo = ~0u, phi = ~0u;
#pragma unroll
        for (int b = 0; b < 8; b++) {
          uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)key[r], (unsigned)(shift + b), 1u);  // all ones iff bit b of the digit is set (one v_bfe_i32)
          asm volatile("" : "+v"(t));  // (the ballot tests THIS register: left alone the compiler shifts the key a second time for a sign test)
          const unsigned long long m = __ballot(t != 0u);
          plo = __builtin_amdgcn_bitop3_b32(plo, t, (uint32_t)m, 0x90);
          phi = __builtin_amdgcn_bitop3_b32(phi, t, (uint32_t)(m >> 32), 0x90);
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const int before = cnt[w][d];
        __builtin_amdgcn_wave_barrier();
        if (rank == 0) cnt[w][d] = (uint16_t)(before + __popc(plo) + __popc(phi));
        __builtin_amdgcn_wave_barrier();
        il[r] = (il[r] & 0xFFFF0000u) | (uint32_t)(before + rank);
        asm volatile("" : "+v"(il[r]));  // materialise the rank HERE: deferred, the 64-bit peer masks of all 24 elements stay live (and spill)
      }
    }
    __syncthreads();
    // ---- exclusive scan of the 256 x 16 counters, digit-major: thread t owns digit t / 4, waves 4 (t % 4) .. + 3
    {
      static_assert(kTsWaves == 16, "counter scan mapping");
      const int t = threadIdx.x, d = t >> 2, w0 = (t & 3) * 4;
      const int c0 = cnt[w0][d], c1 = cnt[w0 + 1][d], c2 = cnt[w0 + 2][d], c3 = cnt[w0 + 3][d];
      const int sum = c0 + c1 + c2 + c3;
      int incl = sum;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int u = __shfl_up(incl, off, kWave);
        if (l >= off) incl += u;
      }
      if (l == kWave - 1) wtot[w] = incl;
      __syncthreads();
      int wbase = 0;
      for (int ww = 0; ww < w; ww++) wbase += wtot[ww];
      const int ex = wbase + incl - sum;
      cnt[w0][d] = (uint16_t)ex;
      cnt[w0 + 1][d] = (uint16_t)(ex + c0);
      cnt[w0 + 2][d] = (uint16_t)(ex + c0 + c1);
      cnt[w0 + 3][d] = (uint16_t)(ex + c0 + c1 + c2);
    }
    __syncthreads();
    // ---- permute through LDS
    // (the keys pass through an empty asm: the compiler then recomputes the two-instruction counter address of every element here
    // instead of keeping the 24 addresses of the rank loop alive across the scan — those were spilled to scratch)
#pragma unroll
    for (int r = 0; r < kTsE; r++) asm volatile("" : "+v"(key[r]));
#pragma unroll
    for (int r = 0; r < kTsE; r++) {
      if (r < rounds) {
        const int dest = (int)cnt[w][(key[r] >> shift) & 255u] + (int)(il[r] & 0xFFFFu);
        keyL[dest] = key[r];
        idxL[dest] = (uint16_t)(il[r] >> 16);
      }
      if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // 8 elements in flight, not 24: the registers are full of elements
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTsE; r++) {
      if (r < rounds) {
        key[r] = keyL[p0 + r * kWave];
        il[r] = (uint32_t)idxL[p0 + r * kWave] << 16;
      }
      if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  uint32_t* kout = keys_out + o + p0;
  int* vout = vals_out + o + p0;
#pragma unroll
  for (int r = 0; r < kTsE; r++) {
    if (r * kWave < lim && r < rounds) {
      kout[r * kWave] = key[r];
      vout[r * kWave] = base + (int)(il[r] >> 16);
    }
  }
}

// One round of pairwise merges of sorted runs of `run` tiles each (run = 1, 2, 4, ...).  Stable: on equal keys the element of the lower
// run goes first.  A run without a partner is a pair with an empty second run (copied).
// The output of a PAIR is cut into chunks of kMergeChunk elements (the last one shorter: no chunk straddles two pairs); block x of a
// cloud takes the cloud's chunks x, x + gridDim.x, ... - at most kMergeMaxChunks of them (the host sizes the grid).
// Round 4: (a) the merge-path splits of ALL chunk boundaries of a block are searched at once, one boundary per thread - the first
// version searched the two boundaries of a chunk when it got there: two threads walking ~16 dependent global loads while 254 waited, once
// per chunk; (b) a thread keeps its kMergeE merged elements in registers, the block puts them back into the staging arrays and
// stores them with full coalesced rows - kMergeE consecutive 4-byte stores per thread were 16-way scattered wave stores.  Per round of
// 2 048 raw 64 k-point clouds: 1.3 -> ~0.55 ms (the sorts of a step 10.9 -> 7.9 ms).
constexpr int kMergeMaxChunks = 8;
__global__ __launch_bounds__(kMergeThreads) void k_sort_merge(const uint32_t* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                              uint32_t* __restrict__ keys_out, int* __restrict__ vals_out,
                                                              const int* __restrict__ counts, size_t cap, int run) {
  // (run == 1: a last tile without a partner is in its place already - k_sort_tile wrote it to this round's output)
  __shared__ uint32_t sk[kMergeChunk + 1];
  __shared__ int sv[kMergeChunk + 1];
  __shared__ int s_a0[kMergeMaxChunks], s_b0[kMergeMaxChunks], s_b1[kMergeMaxChunks], s_d[kMergeMaxChunks][2], s_i[kMergeMaxChunks][2];
  const int c = blockIdx.y;
  const int n = counts[c];
  int nt, tl;
  tile_geometry(n, nt, tl);
  if (nt == 1) return;  // already in its final place (k_sort_tile)
  const size_t o = (size_t)c * cap;
  const uint32_t* K = keys_in + o;
  const int* V = vals_in + o;
  const long L = (long)tl * run;  // elements per input run
  // ---- every boundary of this block's chunks, one per thread: which pair, which diagonal, the merge-path split there
  if (threadIdx.x < 2 * kMergeMaxChunks) {
    const int k = threadIdx.x >> 1, side = threadIdx.x & 1;
    long g = (long)blockIdx.x + (long)k * gridDim.x;  // chunk number in the cloud
    int a0 = 0, b0 = 0, b1 = 0;
    bool found = false;
    for (long pa = 0; pa < n; pa += 2 * L) {
      const int len = (int)min((long)n - pa, 2 * L);
      const long nch = (run == 1 && pa + L >= n) ? 0 : (len + kMergeChunk - 1) / kMergeChunk;
      if (g < nch) {
        a0 = (int)pa;
        b0 = (int)min((long)n, pa + L);
        b1 = (int)min((long)n, pa + 2 * L);
        found = true;
        break;
      }
      g -= nch;
    }
    int d = 0, split = 0;
    if (found) {
      const int la = b0 - a0, lb = b1 - b0;
      d = (int)min((long)(la + lb), (g + side) * (long)kMergeChunk);
      // merge-path split of diagonal d: number of A elements among the first d outputs
      int lo = max(0, d - lb), hi = min(d, la);
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (K[a0 + mid] <= K[b0 + d - 1 - mid]) lo = mid + 1; else hi = mid;
      }
      split = lo;
    }
    if (side == 0) { s_a0[k] = found ? a0 : -1; s_b0[k] = b0; s_b1[k] = b1; }
    s_d[k][side] = d;
    s_i[k][side] = split;
  }
  __syncthreads();
  for (int k = 0; k < kMergeMaxChunks; k++) {
    const int a0 = s_a0[k];
    if (a0 < 0) break;  // (chunk numbers grow with k: no later chunk exists either)
    const int b0 = s_b0[k];
    const int d0 = s_d[k][0], d1 = s_d[k][1];
    const int i0 = s_i[k][0], j0 = d0 - i0, na = s_i[k][1] - i0, nb = (d1 - s_i[k][1]) - j0;
    const int total = na + nb;
    // stage A[i0, i0 + na) then B[j0, j0 + nb)
    for (int t = threadIdx.x; t < total; t += kMergeThreads) {
      const int src = (t < na) ? (a0 + i0 + t) : (b0 + j0 + (t - na));
      sk[t] = K[src];
      sv[t] = V[src];
    }
    __syncthreads();
    // every thread merges kMergeE consecutive outputs of the chunk into registers
    const int q0 = min(total, (int)threadIdx.x * kMergeE), q1 = min(total, q0 + kMergeE);
    uint32_t rk[kMergeE];
    int rv[kMergeE];
    if (q0 < q1) {
      int lo = max(0, q0 - nb), hi = min(q0, na);
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sk[mid] <= sk[na + q0 - 1 - mid]) lo = mid + 1; else hi = mid;
      }
      int ia = lo, ib = q0 - lo;
#pragma unroll
      for (int u = 0; u < kMergeE; u++) {
        const bool live = q0 + u < q1;
        const bool take_a = live && ((ib >= nb) || (ia < na && sk[ia] <= sk[na + ib]));
        const int src = live ? (take_a ? ia : na + ib) : 0;
        rk[u] = sk[src];
        rv[u] = sv[src];
        ia += take_a ? 1 : 0;
        ib += (live && !take_a) ? 1 : 0;
      }
    }
    __syncthreads();  // everybody has read its inputs: the staging arrays take the merged chunk
    if (q0 < q1) {
#pragma unroll
      for (int u = 0; u < kMergeE; u++)
        if (q0 + u < q1) { sk[q0 + u] = rk[u]; sv[q0 + u] = rv[u]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < total; t += kMergeThreads) {
      keys_out[o + a0 + d0 + t] = sk[t];
      vals_out[o + a0 + d0 + t] = sv[t];
    }
    __syncthreads();
  }
}

// Sorts B segments of (key, index-in-segment) pairs by the low 8 * passes key bits, stably; the result ends up in
// (keys_a, vals_a) of the workspace.  vals_a on input must be the identity (vals[i] = i) — every caller's is.
bool sort_pairs_in_tiles(size_t cap) { return div_up((long)cap, kTsTile) <= 64; }

void sort_pairs(hipStream_t s, Workspace& ws, int B, size_t cap, const int* d_counts, int passes, const int* d_passes, const VoxelKeySource* voxel) {
  TimedLaunch tl(KF_SORT, s);
  const int max_tiles = div_up((long)cap, kTsTile);
  if (voxel && !sort_pairs_in_tiles(cap)) throw std::logic_error("sort_pairs: a voxel key source needs the tile path");
  if (max_tiles <= 64) {  // up to 1.5 M elements per segment: log2(tiles) merge rounds still move less than 4 global passes
    // tile sort a -> b, then ceil(log2(max_tiles)) merge rounds ping-ponging; an odd number of hops ends in b: swap the handles
    int hops = 1;
    for (int run = 1; run < max_tiles; run *= 2) hops++;
    // after `hops` hops (a -> b -> a ...) the result sits in b when hops is odd
    if (voxel)
      hipLaunchKernelGGL(k_sort_tile<true>, dim3(max_tiles, B), dim3(kTsThreads), 0, s, ws.keys_a.p, ws.keys_b.p, ws.vals_b.p,
                         (hops & 1) ? ws.keys_b.p : ws.keys_a.p, (hops & 1) ? ws.vals_b.p : ws.vals_a.p, ws.keys_a.p, ws.vals_a.p, d_counts, cap, passes, d_passes,
                         voxel->pts, voxel->vmeta, voxel->min_distance);
    else
      hipLaunchKernelGGL(k_sort_tile<false>, dim3(max_tiles, B), dim3(kTsThreads), 0, s, ws.keys_a.p, ws.keys_b.p, ws.vals_b.p,
                         (hops & 1) ? ws.keys_b.p : ws.keys_a.p, (hops & 1) ? ws.vals_b.p : ws.vals_a.p, ws.keys_a.p, ws.vals_a.p, d_counts, cap, passes, d_passes,
                         (const float4*)nullptr, (const GridMeta*)nullptr, -1.0);
    hops = 1;
    uint32_t* ki = ws.keys_b.p; uint32_t* ko = ws.keys_a.p;
    int* vi = ws.vals_b.p; int* vo = ws.vals_a.p;
    for (int run = 1; run < max_tiles; run *= 2) {
      // (enough blocks per segment to fill the chip when every segment merges: >= 8 k blocks over the batch, at most one per chunk)
      const int chunks = div_up((long)cap, kMergeChunk) + max_tiles;  // (a pair's last chunk is short: at most one more chunk per pair)
      const int gx = std::max(div_up(chunks, kMergeMaxChunks), std::min(chunks, std::max(4, div_up(8192, B))));
      hipLaunchKernelGGL(k_sort_merge, dim3(gx, B), dim3(kMergeThreads), 0, s, ki, vi, ko, vo, d_counts, cap, run);
      std::swap(ki, ko);
      std::swap(vi, vo);
      hops++;
    }
    MI_LAUNCH_CHECK();
    if (hops & 1) {
      std::swap(ws.keys_a.p, ws.keys_b.p);
      std::swap(ws.keys_a.n, ws.keys_b.n);
      std::swap(ws.vals_a.p, ws.vals_b.p);
      std::swap(ws.vals_a.n, ws.vals_b.n);
    }
    return;
  }
  // large segments (sub-maps of 10^5 - 10^6 points): global LSD passes
  const int nsub = div_up((long)cap, kSubTile);
  ws.hist.ensure((size_t)B * 256 * nsub);
  const dim3 grid(div_up((long)cap, kBlockTile), B);
  uint32_t* ka = ws.keys_a.p; uint32_t* kb = ws.keys_b.p;
  int* va = ws.vals_a.p; int* vb = ws.vals_b.p;
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

// Second stage alone: for a caller whose own kernel has produced ws.scan_local / ws.scan_block the way k_scan_local does.
void scan_blocks(hipStream_t s, Workspace& ws, int B, size_t cap) {
  const int nblk = div_up((long)cap, kScanTile);
  ws.totals.ensure(B);
  hipLaunchKernelGGL(k_scan_blocks, dim3(B), dim3(256), 0, s, ws.scan_block.p, ws.totals.p, nblk);
  MI_LAUNCH_CHECK();
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
