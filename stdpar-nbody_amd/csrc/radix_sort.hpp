// Sort of (u64 key, u32 index) pairs by (key, original position) — shared by the Hilbert BVH (K6, replaces std::sort at
// src/bvh.h:55-94) and the octree build.  The order is total (positions are distinct), so every correct sort produces the same
// permutation bit for bit; three forms, by size:
//   * n <= 2048: one block, the pairs held in registers: a network per wave, then merge rounds (sort_one_block_kernel, block_sort_regs);
//   * up to 1.5 M pairs: a SPLITTER sort in five launches (round 4; 24 before): B - 1 splitters from a regular sample of the input
//     sorted by one block, one counting pass + row scan + scatter into the B buckets (the "digit" of a pair is its bucket), then
//     every bucket sorted by one block (block_sort_regs).  Buckets follow the data's own quantiles, so clustered keys (a galaxy inside a box
//     inflated by escapers: most keys share their top 30 bits) cost nothing extra, and ties are broken by position, so equal keys
//     cannot overfill a bucket;
//   * beyond: stable LSD radix sort, 8 bits per pass (per-block digit histogram, per-digit row scan, stable scatter by wave
//     match-any ranking with the 256-digit base scan folded in): 24 launches.
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

// (Until round 4 a FUSED form did the per-digit row scan here, by every block for itself — one launch less per pass for sorts of
// up to 16 blocks.  Those sizes now take the one-block or the splitter sort; the form was unreachable and is gone.)
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
  {  // digit = threadIdx.x: exclusive scan of totals[0..255]; within a wave here, wave offsets after the barrier
    const uint32_t v = totals[threadIdx.x];
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
      woff += dbase[w * 64 + 63] + totals[w * 64 + 63];
    uint32_t run = woff + dbase[threadIdx.x] + hist[uint64_t(threadIdx.x) * nblk + blockIdx.x];
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


// ------------------------------------------------------------------------------------------------
// Splitter sort (2048 < n <= kSplitterMaxN).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kSplitMaxBuckets = 2048;               // B <= 2048: splitters fit LDS (24 KB)
constexpr uint32_t kSampleMax       = 4096;               // pairs the sample block sorts (48 KB of LDS for its merge rounds)
constexpr uint32_t kBucketCap       = 4096;               // pairs one block sorts (48 KB of LDS for its merge rounds); larger buckets: slow global path
constexpr uint32_t kSplitterMaxN    = 768u * kSplitMaxBuckets;  // average bucket <= 768: the largest stays far below the cap
constexpr int kSampleThreads = 1024, kBucketThreads = 512;

__host__ __device__ inline uint32_t splitter_sample(uint32_t B) { return 4u * B < kSampleMax ? 4u * B : kSampleMax; }  // power of two, >= 2 B
__host__ __device__ inline uint32_t splitter_buckets(uint32_t n) {  // power of two, average bucket 384 .. 768
  uint32_t b = 4;
  while (b < kSplitMaxBuckets && uint64_t(b) * 768u < n) b <<= 1;
  return b;
}

__device__ __forceinline__ bool pair_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) { return ka < kb || (ka == kb && ia < ib); }

// Bitonic sorting network in the all-ascending ("flip / disperse") form over p[0 .. P) IN GLOBAL MEMORY, P a power of two, of which
// only the first n hold pairs: the rest count as +infinity, and since every exchange leaves the smaller pair at the lower index an
// exchange whose upper index is >= n is a no-op and is skipped.  The slow path of a bucket that outgrows LDS (an input whose
// regular sample misrepresents it): a fence and a block barrier per stage.
template <int NT>
__device__ __forceinline__ void bitonic_sort_global(uint64_t* __restrict__ K, uint32_t* __restrict__ V, uint32_t P, uint32_t n) {
  auto exchange = [&](uint32_t i, uint32_t j) {
    if (j < n) {
      const uint64_t ki = K[i], kj = K[j];
      const uint32_t vi = V[i], vj = V[j];
      if (pair_less(kj, vj, ki, vi)) {
        K[i] = kj, V[i] = vj;
        K[j] = ki, V[j] = vi;
      }
    }
  };
  for (uint32_t k = 2; k <= P; k <<= 1) {
    const uint32_t half = k >> 1;
    __threadfence_block();
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < P / 2; t += NT) {  // flip: i against the mirror position in its k-block
      const uint32_t base = (t / half) * k, r = t % half;
      exchange(base + r, base + k - 1 - r);
    }
    for (uint32_t j = k >> 2; j >= 1; j >>= 1) {  // disperse
      __threadfence_block();
      __syncthreads();
      for (uint32_t t = threadIdx.x; t < P / 2; t += NT) {
        const uint32_t i = 2 * j * (t / j) + t % j;
        exchange(i, i + j);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
}

// The same order by ONE borrow chain over the 96-bit number key:position (three full-rate instructions and their wait
// states; the compiler's version of pair_less is three 64-bit compares and their selects).
__device__ __forceinline__ bool pair_less_chain(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
  uint64_t mask;
  uint32_t t;
  asm("v_sub_co_u32_e32 %1, vcc, %2, %3\n\t"
      "s_nop 1\n\t"  // (gfx940: a VALU that reads VCC needs two wait states behind the VALU that wrote it, carries included)
      "v_subb_co_u32_e32 %1, vcc, %4, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32_e32 %1, vcc, %6, %7, vcc\n\t"
      "s_mov_b64 %0, vcc"
      : "=s"(mask), "=&v"(t)
      : "v"(ia), "v"(ib), "v"(uint32_t(ka)), "v"(uint32_t(kb)), "v"(uint32_t(ka >> 32)), "v"(uint32_t(kb >> 32))
      : "vcc");
  return __builtin_amdgcn_inverse_ballot_w64(mask);
}

// (ka, ia) < (kb, ib), or <= where `or_equal`: the same chain started with a borrow (a - b - 1 < 0  <=>  a <= b; nothing overflows,
// the chain IS the extended subtraction).  `or_equal_mask` = the lanes that ask for <=, as a lane mask.
__device__ __forceinline__ bool pair_less_chain_or_equal(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib, uint64_t or_equal_mask) {
  uint64_t mask;
  uint32_t t;
  asm("s_mov_b64 vcc, %8\n\t"
      "v_subb_co_u32_e32 %1, vcc, %2, %3, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32_e32 %1, vcc, %4, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32_e32 %1, vcc, %6, %7, vcc\n\t"
      "s_mov_b64 %0, vcc"
      : "=s"(mask), "=&v"(t)
      : "v"(ia), "v"(ib), "v"(uint32_t(ka)), "v"(uint32_t(kb)), "v"(uint32_t(ka >> 32)), "v"(uint32_t(kb >> 32)), "s"(or_equal_mask)
      : "vcc");
  return __builtin_amdgcn_inverse_ballot_w64(mask);
}

// Sort of up to NT E pairs by one block, the pairs in REGISTERS: register e of thread t is element e NT + t of p[0 .. NT E), and the
// aligned block p[0 .. P) comes out ascending (P a power of two <= NT E; elements that are not there are (~0, ~0): greater than
// every pair).
//   1. Every wave sorts the 64 pairs each of its registers holds across the lanes with a bitonic network — 21 stages, the partner's
//      pair fetched with three ds_bpermute (no LDS storage, no barrier), the E networks of a thread side by side.
//   2. The sorted runs are merged pairwise, log2(P / 64) rounds.  One pair per thread: every pair finds its rank in the sibling run
//      by binary search in LDS and is written to its place in the merged run (log2(L) + 1 dependent LDS reads per round).  Several
//      pairs per thread: every thread finds by one binary search where its stretch of the merged run starts in the two runs
//      (the merge path) and merges its E pairs from there.  A pair of the left run goes before an equal one of the right run, so
//      the padding cannot collide.
// A sorting network over all P pairs is log P (log P + 1) / 2 dependent stages of ~0.2 us each whatever they move, so the stage
// count is the time; this form has 21 + log2(P / 64) rounds of log2(L) + 1 probes.  Measured, round 4 (LDS network with a barrier
// or a wave fence per stage / register network with LDS only across waves / this form): 1000 pairs of the one-block octree insert
// 14.9 / 12.8 / 7.5 us; the 1024 sample pairs at N = 10^5 16 / 16 / 8.6; 2048 pairs - / 22 / 16.5; the 4096 sample pairs at 10^6
// 53 / 43 / 38 — there the rounds are bound by LDS throughput (4096 x 12 probes of 12 bytes at random addresses per round); the
// bucket kernel 72 / 67 / 69 at 10^6 (its duration is its largest bucket's), 24.6 / 22.7 / 20.2 at 10^5.
// Ks / Vs: P pairs of LDS.  Every thread of the block must call (block barriers inside).
template <int NT, int E>
__device__ __forceinline__ void block_sort_regs(uint64_t (&k)[E], uint32_t (&v)[E], uint32_t P, uint64_t* Ks, uint32_t* Vs) {
  // P == NT E, or E == 1 and P <= NT (block_sort_dispatch below picks E): every register of every wave below P takes part
  const uint32_t t = threadIdx.x, lane = t & 63u;
  const bool holds = (t & ~63u) < P;  // (wave-uniform; false only where P < NT)
  if (holds) {
    for (uint32_t kk = 2; kk <= 64u; kk <<= 1) {
      const bool asc = (lane & kk) == 0;  // (the last level: everywhere)
      for (uint32_t j = kk >> 1; j >= 1u; j >>= 1) {
        const bool keep_min = ((lane & j) == 0) == asc;
        const int src       = int((lane ^ j) << 2);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const uint32_t pl = uint32_t(__builtin_amdgcn_ds_bpermute(src, int(uint32_t(k[e]))));
          const uint32_t ph = uint32_t(__builtin_amdgcn_ds_bpermute(src, int(uint32_t(k[e] >> 32))));
          const uint32_t pv = uint32_t(__builtin_amdgcn_ds_bpermute(src, int(v[e])));
          const uint64_t pk = (uint64_t(ph) << 32) | pl;
          const bool take   = pair_less_chain(pk, pv, k[e], v[e]) == keep_min;
          k[e] = take ? pk : k[e], v[e] = take ? pv : v[e];
        }
      }
    }
  }
  if (P <= 64u) return;
  if (holds) {
#pragma unroll
    for (int e = 0; e < E; ++e) Ks[e * NT + t] = k[e], Vs[e * NT + t] = v[e];
  }
  __syncthreads();
  if constexpr (E == 1) {
    for (uint32_t L = 64u; L < P; L <<= 1) {  // runs of L -> runs of 2 L: every pair ranks itself in the sibling run
      uint32_t c = 0;
      if (holds) {
        const uint32_t run   = t / L;  // (a wave's 64 pairs lie in one run)
        const uint32_t sib   = (run ^ 1u) * L;
        const uint64_t right = __builtin_amdgcn_ballot_w64((run & 1u) != 0);
        // the sibling's pairs that go before this one: those < it (it is of the left run), those <= it (of the right run)
        for (uint32_t step = L >> 1; step >= 1; step >>= 1)
          if (pair_less_chain_or_equal(Ks[sib + c + step - 1], Vs[sib + c + step - 1], k[0], v[0], right)) c += step;
        c += pair_less_chain_or_equal(Ks[sib + c], Vs[sib + c], k[0], v[0], right);
      }
      __syncthreads();
      if (holds) {
        const uint32_t dst = ((t / L) & ~1u) * L + t % L + c;
        Ks[dst] = k[0], Vs[dst] = v[0];
      }
      __syncthreads();
      if (holds) k[0] = Ks[t], v[0] = Vs[t];
    }
  } else {
    // Several pairs per thread (P == NT E, every thread takes part): thread t produces the E consecutive pairs t E ... t E + E - 1 of
    // the merged runs — it finds where that stretch starts in the two runs (the merge path: one binary search per THREAD, not per
    // pair) and merges E steps from there; log2(L) + 2 E dependent LDS reads per round and thread instead of E (log2(L) + 1), which
    // at 4096 pairs were the LDS's whole throughput.  A pair of the left run goes before an equal one of the right run.
    for (uint32_t L = 64u; L < P; L <<= 1) {
      const uint32_t o0 = t * E, pb = o0 & ~(2u * L - 1u), o = o0 - pb;  // (E divides 2 L: the stretch lies in one pair of runs)
      const uint64_t *ak = Ks + pb, *bk = Ks + pb + L;
      const uint32_t *av = Vs + pb, *bv = Vs + pb + L;
      uint32_t lo = o > L ? o - L : 0u, hi = o < L ? o : L;  // pairs of the left run among the first o of the merged pair
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pair_less_chain(bk[o - 1u - mid], bv[o - 1u - mid], ak[mid], av[mid])) hi = mid;  // left[mid] comes after right[o-1-mid]
        else lo = mid + 1u;
      }
      uint32_t i = lo, j = o - lo;
      uint64_t ka = ak[i < L ? i : L - 1u], kb = bk[j < L ? j : L - 1u];
      uint32_t va = av[i < L ? i : L - 1u], vb = bv[j < L ? j : L - 1u];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const bool take_a = i < L && (j >= L || !pair_less_chain(kb, vb, ka, va));
        k[e] = take_a ? ka : kb, v[e] = take_a ? va : vb;
        i += take_a, j += !take_a;
        if (e + 1 < E) {  // the next head of the run the pair came from
          const uint32_t nx = take_a ? (i < L ? i : L - 1u) : L + (j < L ? j : L - 1u);
          const uint64_t nk = ak[nx];
          const uint32_t nv = av[nx];
          ka = take_a ? nk : ka, va = take_a ? nv : va;
          kb = take_a ? kb : nk, vb = take_a ? vb : nv;
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) Ks[o0 + e] = k[e], Vs[o0 + e] = v[e];
      __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < E; ++e) k[e] = Ks[e * NT + t], v[e] = Vs[e * NT + t];
  }
}

// the same for any power of two P <= NT EMAX: with as few registers per thread as P needs (the others hold padding only)
template <int NT, int EMAX>
__device__ __forceinline__ void block_sort_dispatch(uint64_t (&k)[EMAX], uint32_t (&v)[EMAX], uint32_t P, uint64_t* Ks, uint32_t* Vs) {
  if constexpr (EMAX == 1) {
    block_sort_regs<NT, 1>(k, v, P, Ks, Vs);
  } else {
    if (P <= uint32_t(NT) * (EMAX / 2)) {
      block_sort_dispatch<NT, EMAX / 2>(reinterpret_cast<uint64_t(&)[EMAX / 2]>(k), reinterpret_cast<uint32_t(&)[EMAX / 2]>(v), P, Ks, Vs);
    } else {
      block_sort_regs<NT, EMAX>(k, v, P, Ks, Vs);
    }
  }
}

// A sort that fits ONE block (n <= 2048: the reference's default run is 1000 bodies): the block sort above over one or two pairs
// per thread, from (k0, position) into (k1, i1).  (Until round 4 all eight passes of the LSD radix sort in one launch: 44.6 us at
// n = 2048; 16 dependent launches before that.)
constexpr int kOneBlockThreads = 1024;
static __global__ __launch_bounds__(kOneBlockThreads) void sort_one_block_kernel(const uint64_t* __restrict__ k0, uint64_t* __restrict__ k1,
                                                                                         uint32_t* __restrict__ i1, uint32_t n) {
  constexpr int E = 2;
  __shared__ uint64_t K[E * kOneBlockThreads];
  __shared__ uint32_t V[E * kOneBlockThreads];
  uint64_t k[E];
  uint32_t v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t q = e * kOneBlockThreads + threadIdx.x;
    k[e]             = q < n ? k0[q] : ~0ull;
    v[e]             = q < n ? q : ~0u;
  }
  uint32_t P = 1;
  while (P < n) P <<= 1;
  block_sort_dispatch<kOneBlockThreads, E>(k, v, P, K, V);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t q = e * kOneBlockThreads + threadIdx.x;
    if (q < n) k1[q] = k[e], i1[q] = v[e];
  }
}

// S1: the splitters.  One block sorts m = splitter_sample(B) pairs taken at a regular stride from the input and keeps every
// (m / B)-th one.  (Measured and not kept, round 4: ranking every sample pair against all others from LDS broadcasts instead of
// sorting — no barriers, no dependent stages, any number of blocks — 28.8 us against 16 for m = 1024 and 115 against 53 for
// m = 4096: three compares, two mask operations and two LDS reads per comparison cost more than the network's 66 round trips.)
static __global__ __launch_bounds__(kSampleThreads) void splitter_sample_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t B,
                                                                                uint64_t* __restrict__ split_key,
                                                                                uint32_t* __restrict__ split_idx) {
  constexpr int E = kSampleMax / kSampleThreads;
  __shared__ uint64_t K[kSampleMax];
  __shared__ uint32_t V[kSampleMax];
  const uint32_t m = splitter_sample(B), every = m / B;
  uint64_t k[E];
  uint32_t v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t q = e * kSampleThreads + threadIdx.x;
    const uint32_t i = q < m ? uint32_t((uint64_t(q) * n + n / 2) / m) : 0u;  // < n
    k[e]             = q < m ? keys[i] : ~0ull;
    v[e]             = q < m ? i : ~0u;
  }
  block_sort_dispatch<kSampleThreads, E>(k, v, m, K, V);
#pragma unroll
  for (int e = 0; e < E; ++e) {  // bucket b holds the pairs p with splitter[b-1] <= p < splitter[b]
    const uint32_t q = e * kSampleThreads + threadIdx.x;
    if (q < m && q >= every && q % every == 0) {
      split_key[q / every - 1] = k[e];
      split_idx[q / every - 1] = v[e];
    }
  }
}

// lanes of the wave holding the same bucket as this one (match-any over the log2(B) bits of the bucket number)
__device__ __forceinline__ uint64_t same_bucket_lanes(uint32_t b, bool valid, int bits) {
  uint64_t same = __ballot(valid);
  for (int q = 0; q < bits; ++q) {
    const bool bit      = (b >> q) & 1u;
    const uint64_t vote = __ballot(valid && bit);
    same &= bit ? vote : ~vote;
  }
  return same;
}

// S2: per-(bucket, block) counts; the bucket of every pair is kept (u16) for the scatter.
template <int NT, int IPT>
static __global__ __launch_bounds__(NT) void splitter_count_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t B, int bits,
                                                                       const uint64_t* __restrict__ split_key,
                                                                       const uint32_t* __restrict__ split_idx,
                                                                       uint32_t* __restrict__ hist, uint32_t nblk,
                                                                       uint16_t* __restrict__ bucket_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* sk  = reinterpret_cast<uint64_t*>(smem);
  uint32_t* si  = reinterpret_cast<uint32_t*>(sk + B);
  uint32_t* cnt = si + B;
  for (uint32_t q = threadIdx.x; q < B; q += NT) {
    if (q + 1 < B) sk[q] = split_key[q], si[q] = split_idx[q];
    cnt[q] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // the thread's IPT searches advance together: each probe is a dependent LDS round trip (one after the other they were most of
  // the kernel: 29 us at N = 10^6)
  uint64_t key[IPT];
  uint32_t lo[IPT];
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const uint64_t i = uint64_t(blockIdx.x) * (NT * IPT) + q * NT + threadIdx.x;
    key[q]           = i < n ? keys[i] : 0ull;
    lo[q]            = 0;  // invariant: splitters [0, lo) are <= the pair, and the answer is < lo + step after each step
  }
  for (uint32_t step = B >> 1; step >= 1; step >>= 1) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const uint32_t i   = blockIdx.x * uint32_t(NT * IPT) + q * NT + threadIdx.x;
      const uint32_t mid = lo[q] + step;  // mid - 1 in [0, B - 2]
      if (!pair_less_chain(key[q], i, sk[mid - 1], si[mid - 1])) lo[q] = mid;
    }
  }
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const uint64_t i = uint64_t(blockIdx.x) * (NT * IPT) + q * NT + threadIdx.x;
    const bool valid = i < n;
    const uint32_t b = valid ? lo[q] : 0u;
    if (valid) bucket_out[i] = uint16_t(b);
    const uint64_t same = same_bucket_lanes(b, valid, bits);  // one LDS atomic per distinct bucket of the strip
    if (valid && (same & lt_mask) == 0ull) atomicAdd(&cnt[b], uint32_t(__popcll(same)));
  }
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < B; q += NT) hist[uint64_t(q) * nblk + blockIdx.x] = cnt[q];
}

// S4: scatter into the buckets (after radix_scan_rows_kernel over the B rows).  Order inside a bucket is whatever the LDS
// atomics give: the bucket is sorted afterwards, and the sorted order is unique.
template <int NT, int IPT>
static __global__ __launch_bounds__(NT) void splitter_scatter_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t B, int bits,
                                                                         const uint16_t* __restrict__ bucket_in,
                                                                         const uint32_t* __restrict__ hist,
                                                                         const uint32_t* __restrict__ totals, uint32_t nblk,
                                                                         uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                                                                         uint32_t* __restrict__ bucket_start) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* base = reinterpret_cast<uint32_t*>(smem);  // [B]: where this block's pairs of a bucket go next
  __shared__ uint32_t wsum[NT / 64];
  __shared__ uint32_t carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < B; b0 += NT) {  // exclusive scan of the bucket totals, NT at a time
    const uint32_t q = b0 + threadIdx.x;
    const uint32_t v = q < B ? totals[q] : 0u;
    uint32_t inc     = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = carry;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (q < B) {
      base[q] = pre + inc - v + hist[uint64_t(q) * nblk + blockIdx.x];
      if (blockIdx.x == 0) bucket_start[q] = pre + inc - v;
    }
    __syncthreads();
    if (threadIdx.x == NT - 1) carry = pre + inc;
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) bucket_start[B] = n;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const uint64_t i = uint64_t(blockIdx.x) * (NT * IPT) + q * NT + threadIdx.x;
    const bool valid = i < n;
    const uint32_t b = valid ? bucket_in[i] : 0u;
    const uint64_t same = same_bucket_lanes(b, valid, bits);
    uint32_t first = 0;
    if (valid && (same & lt_mask) == 0ull) first = atomicAdd(&base[b], uint32_t(__popcll(same)));
    first = __shfl(first, valid ? int(__builtin_ctzll(same)) : 0, 64);  // the strip's leader for this bucket
    if (valid) {
      const uint32_t pos = first + uint32_t(__popcll(same & lt_mask));
      keys_out[pos]      = keys[i];
      idx_out[pos]       = uint32_t(i);
    }
  }
}

// S5: one block per bucket sorts it — in registers up to kBucketCap pairs, in place in
// global memory beyond (a bucket that large takes an input whose regular sample misrepresents it; correct, slow) — and writes it
// to the output buffers at the same positions.  (One block per bucket, not a loop over buckets in fewer blocks: 256 ... 2048 blocks
// for 2048 buckets time the same.)
// (The rank-by-comparison form of S1's note was measured here too: 217 us against 22-36 at N = 10^5, 592 against 77-102 at 10^6.)
static __global__ __launch_bounds__(kBucketThreads) void splitter_bucket_sort_kernel(uint64_t* __restrict__ keys_in, uint32_t* __restrict__ idx_in,
                                                                                     uint64_t* __restrict__ keys_out,
                                                                                     uint32_t* __restrict__ idx_out,
                                                                                     const uint32_t* __restrict__ bucket_start) {
  __shared__ uint64_t K[kBucketCap];
  __shared__ uint32_t V[kBucketCap];
  const uint32_t start = bucket_start[blockIdx.x], cnt = bucket_start[blockIdx.x + 1] - start;
  if (cnt == 0) return;
  uint32_t P = 1;
  while (P < cnt) P <<= 1;
  if (cnt <= kBucketCap) {
    constexpr int E = kBucketCap / kBucketThreads;  // (the registers beyond the bucket's size take no part)
    uint64_t k[E];
    uint32_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t q = e * kBucketThreads + threadIdx.x;
      k[e]             = q < cnt ? keys_in[start + q] : ~0ull;
      v[e]             = q < cnt ? idx_in[start + q] : ~0u;
    }
    block_sort_dispatch<kBucketThreads, E>(k, v, P, K, V);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t q = e * kBucketThreads + threadIdx.x;
      if (q < cnt) keys_out[start + q] = k[e], idx_out[start + q] = v[e];
    }
  } else {
    bitonic_sort_global<kBucketThreads>(keys_in + start, idx_in + start, P, cnt);
    for (uint32_t q = threadIdx.x; q < cnt; q += kBucketThreads) keys_out[start + q] = keys_in[start + q], idx_out[start + q] = idx_in[start + q];
  }
}

// Scratch of a sort of n pairs, in u32 words: the count matrix (rows x blocks) + row totals, the bucket starts, the splitters
// and the per-pair bucket numbers.  hist must hold radix_sort_scratch_words(n) words.
// pairs per counting / scatter block: 512 up to 2^18 pairs (256 threads x 2: more, smaller blocks), 2048 beyond (1024 threads x 2).
// Measured at N = 10^6 (count / scatter, us): 256 threads x 8 pairs 29 / 35 (the thread's searches one after the other or together:
// the same), x 4 29 / 40, x 16 33 / 45; 1024 threads x 2 (shipped) 21 / 31 — the per-block fixed work (24 KB of splitters in, 2048
// counts out, the scan of the bucket totals) is what the larger block shortens.
constexpr uint32_t kSplitSmallN = 1u << 18;
constexpr int kSplitBigThreads  = 1024;
inline uint32_t splitter_blocks(uint32_t n) { return n <= kSplitSmallN ? (n + 511u) / 512u : (n + 2047u) / 2048u; }
inline size_t radix_sort_scratch_words(uint32_t n) {
  const bool split  = n > kSortTile && n <= kSplitterMaxN;
  const size_t nblk = split ? splitter_blocks(n) : (size_t(n) + kSortTile - 1) / kSortTile;
  const size_t rows = split ? splitter_buckets(n) : 256;
  return rows * (nblk + 1) + (rows + 1) + 3 * rows + (size_t(n) + 1) / 2 + 8;
}

// hist needs radix_sort_scratch_words(n) u32 (8-byte aligned).  Sorts by (key, position): the pairs start in (keys[0], identity)
// and end in (keys[final], idx[final]); returns `final` (0 or 1) through *final_buf.  Both buffers of keys / idx are overwritten.
// key_bits: bits [key_bits, 64) of every key MUST be zero (63 for the 3D curves, 64 in 2D) — the one-block and the splitter sort
// compare all 64 bits, the radix passes stop at key_bits, and under that condition they agree.  n == 0: nothing is launched.
inline uint32_t radix_sort_blocks(uint32_t n) { return (n + kSortTile - 1) / kSortTile; }
// Tried and not kept (round 3): taking the NEXT pass's histogram inside the scatter — it knows the block every key lands in —
// with one global atomicAdd per key into three rotating histogram buffers (7 launches fewer per sort).  Bit-exact, but the
// octree step at N = 10^5 went from 0.445 to 0.525 ms: 10^5 device-scope atomics per pass cost more than the 4.8 us launch
// they replace, and (block, next digit) pairs of one block's keys are nearly all distinct, so LDS cannot pre-aggregate them.
// Also tried and not kept: 11-bit digits (6 passes instead of 8; 2048 bins, 40 KB of LDS in the scatter, 11 ballots per strip).
// Octree step, graph replay, before -> after: N = 10^4 0.214 -> 0.226 ms, 10^5 0.445 -> 0.437, 10^6 2.81 -> 2.92 — the two
// passes saved are paid back by the wider bins everywhere but at 10^5.

inline int radix_sort_pairs(uint64_t* keys[2], uint32_t* idx[2], uint32_t n, int key_bits, uint32_t* hist, hipStream_t st,
                            int* final_buf) {
  const uint32_t nblk    = radix_sort_blocks(n);
  if (n == 0) {
    *final_buf = 0;
    return NBODY_OK;
  }
  if (nblk == 1) {
    hipLaunchKernelGGL(sort_one_block_kernel, dim3(1), dim3(kOneBlockThreads), 0, st, keys[0], keys[1], idx[1], n);
    NB_HIP(hipGetLastError());
    *final_buf = 1;
    return NBODY_OK;
  }
  if (n <= kSplitterMaxN) {  // splitter sort: pairs start in (keys[0], position), go through (keys[1], idx[1]) bucketed, end in (keys[0], idx[0])
    const uint32_t B = splitter_buckets(n), sblk = splitter_blocks(n);
    int bits         = 0;
    while ((1u << bits) < B) ++bits;
    uint32_t* totals       = hist + size_t(B) * sblk;
    uint32_t* bucket_start = totals + B;
    uint32_t* split_idx    = bucket_start + B + 1;
    uint64_t* split_key    = reinterpret_cast<uint64_t*>(split_idx + B + 1);  // 8-byte aligned: hist is, and B (sblk + 3) + 2 words is even
    uint16_t* bucket       = reinterpret_cast<uint16_t*>(reinterpret_cast<uint32_t*>(split_key) + 2 * size_t(B));
    hipLaunchKernelGGL(splitter_sample_kernel, dim3(1), dim3(kSampleThreads), 0, st, keys[0], n, B, split_key, split_idx);
    NB_HIP(hipGetLastError());
    if (n <= kSplitSmallN)
      hipLaunchKernelGGL((splitter_count_kernel<kSortB, 2>), dim3(sblk), dim3(kSortB), B * 16u, st, keys[0], n, B, bits, split_key, split_idx, hist, sblk, bucket);
    else
      hipLaunchKernelGGL((splitter_count_kernel<kSplitBigThreads, 2>), dim3(sblk), dim3(kSplitBigThreads), B * 16u, st, keys[0], n, B, bits, split_key, split_idx, hist, sblk, bucket);
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(B), dim3(kSortB), 0, st, hist, sblk, totals);
    NB_HIP(hipGetLastError());
    if (n <= kSplitSmallN)
      hipLaunchKernelGGL((splitter_scatter_kernel<kSortB, 2>), dim3(sblk), dim3(kSortB), B * 4u, st, keys[0], n, B, bits, bucket, hist, totals, sblk, keys[1], idx[1], bucket_start);
    else
      hipLaunchKernelGGL((splitter_scatter_kernel<kSplitBigThreads, 2>), dim3(sblk), dim3(kSplitBigThreads), B * 4u, st, keys[0], n, B, bits, bucket, hist, totals, sblk, keys[1], idx[1], bucket_start);
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL(splitter_bucket_sort_kernel, dim3(B), dim3(kBucketThreads), 0, st, keys[1], idx[1], keys[0], idx[0], bucket_start);
    NB_HIP(hipGetLastError());
    *final_buf = 0;
    return NBODY_OK;
  }
  int cur                = 0;
  const uint32_t* idx_in = nullptr;  // first pass: payload = position
  for (int shift = 0; shift < key_bits; shift += 8) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(kSortB), 0, st, keys[cur], n, shift, hist, nblk);
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(256), dim3(kSortB), 0, st, hist, nblk, hist + 256u * size_t(nblk));
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(kSortB), 0, st, keys[cur], idx_in, keys[cur ^ 1], idx[cur ^ 1], n, shift,
                       hist, hist + 256u * size_t(nblk), nblk);
    NB_HIP(hipGetLastError());
    cur ^= 1;
    idx_in = idx[cur];
  }
  *final_buf = cur;
  return NBODY_OK;
}

}  // namespace nbody
