// Stable LSD radix sort of (u64 key, u32 index) pairs, 8 bits per pass — shared by the Hilbert BVH (K6, replaces
// std::sort at src/bvh.h:55-94) and the octree build.  Per pass: per-block digit histogram, per-digit row scan,
// stable scatter (wave match-any ranking; the 256-digit base scan is folded into the scatter kernel).
// Kernels are `static` so both translation units can include this header.
#pragma once
#include "common.hpp"

namespace nbody {

constexpr int kSortB = 256;
constexpr int kSortIPT  = 8;  // keys per lane
constexpr int kSortTile = kSortB * kSortIPT;  // 2048 keys per block; wave w owns keys [w*512, w*512+512)

static __global__ __launch_bounds__(kSortB) void radix_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, int shift,
                                                        uint32_t* __restrict__ hist, uint32_t nblk) {
  __shared__ uint32_t cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kSortIPT; ++q) {
    uint64_t i = uint64_t(blockIdx.x) * kSortTile + q * kSortB + threadIdx.x;
    if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[uint64_t(threadIdx.x) * nblk + blockIdx.x] = cnt[threadIdx.x];
}

// Exclusive scan of one digit's row hist[d][0..nblk) (one block per digit, coalesced) + the digit's total.
// The scan ACROSS digits (256 totals) is folded into the scatter kernel, so a pass has no serial kernel.
static __global__ __launch_bounds__(kSortB) void radix_scan_rows_kernel(uint32_t* __restrict__ hist, uint32_t nblk,
                                                             uint32_t* __restrict__ totals) {
  __shared__ uint32_t wsum[kSortB / 64];
  __shared__ uint32_t carry;
  uint32_t* row  = hist + uint64_t(blockIdx.x) * nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblk; base += kSortB) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblk ? row[i] : 0u;
    uint32_t inc     = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = carry;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (i < nblk) row[i] = pre + inc - v;
    __syncthreads();
    if (threadIdx.x == kSortB - 1) carry = pre + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// FUSED: the per-digit row scan over the blocks is done here, by every block for itself (thread d adds row d of the histogram up
// to its own block and to the end) — one launch less per pass.  For few blocks only (the rows are read once per block).
template <bool FUSED>
static __global__ __launch_bounds__(kSortB) void radix_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                                                           uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                                                           uint32_t n, int shift, const uint32_t* __restrict__ hist,
                                                           const uint32_t* __restrict__ totals, uint32_t nblk) {
  __shared__ uint32_t wcnt[kSortB / 64][256];
  __shared__ uint32_t dbase[256];  // exclusive scan of the 256 digit totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = threadIdx.x; q < (kSortB / 64) * 256; q += kSortB) (&wcnt[0][0])[q] = 0;
  __syncthreads();

  uint64_t key[kSortIPT];
  uint32_t pay[kSortIPT], rank[kSortIPT];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int s = 0; s < kSortIPT; ++s) {
    uint64_t i       = uint64_t(blockIdx.x) * kSortTile + wave * (64 * kSortIPT) + s * 64 + lane;
    const bool valid = i < n;
    key[s]           = valid ? keys_in[i] : 0ull;
    pay[s]           = valid ? (idx_in ? idx_in[i] : uint32_t(i)) : 0u;
    const uint32_t d = uint32_t(key[s] >> shift) & 255u;
    // lanes of this strip holding the same digit
    uint64_t same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit      = (d >> b) & 1u;
      const uint64_t vote = __ballot(valid && bit);
      same &= bit ? vote : ~vote;
    }
    const uint32_t before = __popcll(same & lt_mask);
    const uint32_t base   = wcnt[wave][d];
    rank[s]               = base + before;
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) wcnt[wave][d] = base + uint32_t(__popcll(same));
    __builtin_amdgcn_wave_barrier();
  }
  __shared__ uint32_t dtot[256];   // FUSED: the digit totals this block computed itself
  uint32_t before_me = 0;          // FUSED: keys with this digit in the blocks before this one
  {  // digit = threadIdx.x: exclusive scan of totals[0..255]; within a wave here, wave offsets after the barrier
    uint32_t v;
    if constexpr (FUSED) {
      const uint32_t* row = hist + uint64_t(threadIdx.x) * nblk;
      uint32_t tot = 0;
      for (uint32_t b = 0; b < nblk; ++b) {
        const uint32_t h = row[b];
        before_me += b < blockIdx.x ? h : 0u;
        tot += h;
      }
      dtot[threadIdx.x] = tot;
      v                 = tot;
    } else {
      v = totals[threadIdx.x];
    }
    uint32_t inc     = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    dbase[threadIdx.x] = inc - v;
  }
  __syncthreads();
  // digit = threadIdx.x: turn per-wave counts into global bases
  {
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w)  // totals of the earlier 64-digit groups
      woff += dbase[w * 64 + 63] + (FUSED ? dtot[w * 64 + 63] : totals[w * 64 + 63]);
    uint32_t run = woff + dbase[threadIdx.x] + (FUSED ? before_me : hist[uint64_t(threadIdx.x) * nblk + blockIdx.x]);
#pragma unroll
    for (int w = 0; w < kSortB / 64; ++w) {
      uint32_t cw          = wcnt[w][threadIdx.x];
      wcnt[w][threadIdx.x] = run;
      run += cw;
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < kSortIPT; ++s) {
    uint64_t i = uint64_t(blockIdx.x) * kSortTile + wave * (64 * kSortIPT) + s * 64 + lane;
    if (i < n) {
      const uint32_t d   = uint32_t(key[s] >> shift) & 255u;
      const uint32_t pos = wcnt[wave][d] + rank[s];
      keys_out[pos]      = key[s];
      idx_out[pos]       = pay[s];
    }
  }
}


// A sort that fits ONE block's tile (n <= 2048: the reference's default run is 1000 bodies) does all its passes in one launch: the
// block's own digit counts are the whole histogram, and a block barrier separates the passes (the pairs ping-pong through the
// same two global buffers; a block's stores are visible to its own later loads).  16 dependent launches were half of the octree
// step at that size.
static __global__ __launch_bounds__(kSortB) void radix_sort_one_block_kernel(uint64_t* __restrict__ k0, uint64_t* __restrict__ k1,
                                                                             uint32_t* __restrict__ i0, uint32_t* __restrict__ i1,
                                                                             uint32_t n, int key_bits) {
  __shared__ uint32_t wcnt[kSortB / 64][256];
  __shared__ uint32_t dbase[256], dtot[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int pass = 0;
  for (int shift = 0; shift < key_bits; shift += 8, ++pass) {
    const uint64_t* kin  = (pass & 1) ? k1 : k0;
    uint64_t* kout       = (pass & 1) ? k0 : k1;
    const uint32_t* iin  = pass == 0 ? nullptr : ((pass & 1) ? i1 : i0);  // first pass: payload = position
    uint32_t* iout       = (pass & 1) ? i0 : i1;
    for (int q = threadIdx.x; q < (kSortB / 64) * 256; q += kSortB) (&wcnt[0][0])[q] = 0;
    __syncthreads();
    uint64_t key[kSortIPT];
    uint32_t pay[kSortIPT], rank[kSortIPT];
#pragma unroll
    for (int s = 0; s < kSortIPT; ++s) {
      const uint32_t i = wave * (64 * kSortIPT) + s * 64 + lane;
      const bool valid = i < n;
      key[s]           = valid ? kin[i] : 0ull;
      pay[s]           = valid ? (iin ? iin[i] : i) : 0u;
      const uint32_t d = uint32_t(key[s] >> shift) & 255u;
      uint64_t same    = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit      = (d >> b) & 1u;
        const uint64_t vote = __ballot(valid && bit);
        same &= bit ? vote : ~vote;
      }
      const uint32_t before = __popcll(same & lt_mask);
      const uint32_t base   = wcnt[wave][d];
      rank[s]               = base + before;
      __builtin_amdgcn_wave_barrier();
      if (valid && before == 0) wcnt[wave][d] = base + uint32_t(__popcll(same));
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {  // digit = threadIdx.x: its total over the waves, then the exclusive scan over the digits
      uint32_t v = 0;
#pragma unroll
      for (int w = 0; w < kSortB / 64; ++w) v += wcnt[w][threadIdx.x];
      dtot[threadIdx.x] = v;
      uint32_t inc      = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
      }
      dbase[threadIdx.x] = inc - v;
    }
    __syncthreads();
    {
      uint32_t run = dbase[threadIdx.x];
      for (int w = 0; w < wave; ++w) run += dbase[w * 64 + 63] + dtot[w * 64 + 63];  // totals of the earlier 64-digit groups
#pragma unroll
      for (int w = 0; w < kSortB / 64; ++w) {
        uint32_t cw          = wcnt[w][threadIdx.x];
        wcnt[w][threadIdx.x] = run;
        run += cw;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSortIPT; ++s) {
      const uint32_t i = wave * (64 * kSortIPT) + s * 64 + lane;
      if (i < n) {
        const uint32_t d   = uint32_t(key[s] >> shift) & 255u;
        const uint32_t pos = wcnt[wave][d] + rank[s];
        kout[pos]          = key[s];
        iout[pos]          = pay[s];
      }
    }
    __threadfence_block();
    __syncthreads();  // the next pass reads what this one wrote
  }
}

// hist needs 256 * (nblk + 1) u32, nblk = radix_sort_blocks(n).  Sorts by key bits [0, key_bits): the pairs start in
// (keys[0], identity) and end in (keys[final], idx[final]); returns `final` (0 or 1) through *final_buf.
inline uint32_t radix_sort_blocks(uint32_t n) { return (n + kSortTile - 1) / kSortTile; }
// Tried and not kept (round 3): taking the NEXT pass's histogram inside the scatter — it knows the block every key lands in —
// with one global atomicAdd per key into three rotating histogram buffers (7 launches fewer per sort).  Bit-exact, but the
// octree step at N = 10^5 went from 0.445 to 0.525 ms: 10^5 device-scope atomics per pass cost more than the 4.8 us launch
// they replace, and (block, next digit) pairs of one block's keys are nearly all distinct, so LDS cannot pre-aggregate them.
// Also tried and not kept: 11-bit digits (6 passes instead of 8; 2048 bins, 40 KB of LDS in the scatter, 11 ballots per strip).
// Octree step, graph replay, before -> after: N = 10^4 0.214 -> 0.226 ms, 10^5 0.445 -> 0.437, 10^6 2.81 -> 2.92 — the two
// passes saved are paid back by the wider bins everywhere but at 10^5.
// Measured (profiles/r03/small_trees_kernel_stats.txt): at 49 blocks (N = 10^5) the fused scatter takes 13.2 us against 8 + 5 for
// scatter + scan — nothing gained; at 5 blocks (N = 10^4) the octree step goes from 0.324 to 0.278 ms.
constexpr uint32_t kSortFusedBlocks = 16;  // up to 32 768 keys: every block scans the 256 x nblk histogram itself

inline int radix_sort_pairs(uint64_t* keys[2], uint32_t* idx[2], uint32_t n, int key_bits, uint32_t* hist, hipStream_t st,
                            int* final_buf) {
  const uint32_t nblk    = radix_sort_blocks(n);
  if (nblk == 1) {
    hipLaunchKernelGGL(radix_sort_one_block_kernel, dim3(1), dim3(kSortB), 0, st, keys[0], keys[1], idx[0], idx[1], n, key_bits);
    NB_HIP(hipGetLastError());
    *final_buf = ((key_bits + 7) / 8) & 1;
    return NBODY_OK;
  }
  int cur                = 0;
  const uint32_t* idx_in = nullptr;  // first pass: payload = position
  for (int shift = 0; shift < key_bits; shift += 8) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(kSortB), 0, st, keys[cur], n, shift, hist, nblk);
    NB_HIP(hipGetLastError());
    if (nblk <= kSortFusedBlocks) {  // small sorts are bound by their dependent launches: 2 per pass instead of 3
      hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(nblk), dim3(kSortB), 0, st, keys[cur], idx_in, keys[cur ^ 1], idx[cur ^ 1],
                         n, shift, hist, hist + 256u * size_t(nblk), nblk);
      NB_HIP(hipGetLastError());
    } else {
      hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(256), dim3(kSortB), 0, st, hist, nblk, hist + 256u * size_t(nblk));
      NB_HIP(hipGetLastError());
      hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(nblk), dim3(kSortB), 0, st, keys[cur], idx_in, keys[cur ^ 1],
                         idx[cur ^ 1], n, shift, hist, hist + 256u * size_t(nblk), nblk);
      NB_HIP(hipGetLastError());
    }
    cur ^= 1;
    idx_in = idx[cur];
  }
  *final_buf = cur;
  return NBODY_OK;
}

}  // namespace nbody
