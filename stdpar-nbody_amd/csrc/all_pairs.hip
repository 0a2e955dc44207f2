// K1 all-pairs force, K2 all-pairs-collapsed force, K3 leapfrog step — hand-written for gfx950.
//
// K1 (replaces src/all_pairs.h:14-27).  Bound: FP64/FP32 VALU issue (no MFMA: the inner body is
// sub/FMA/rsq, not a contraction).  Structure:
//   * one lane per target body (R targets per lane), targets of a block held in VGPRs;
//   * sources are packed (x, m) records visited in tiles of 512; every tile is cut into JS slices, one per wave of the
//     block that shares a target group, and the tile sequence into source chunks over grid.y — both from sz alone, so a
//     shard window (N/8 targets on one GPU) sums exactly as the whole system does and still fills the chip;
//   * slice partials are combined through LDS in wave order, chunk sums added in chunk order — c * (((s_0 + s_1) + s_2) + ...) — by
//     the chunks' blocks themselves: passed from block to block in `a` (a turn word per target group; a failure is loud: k1_status), or,
//     where the whole grid is resident at once (<= 2048 blocks), collected by whichever chunk arrives last.  Deterministic, the same
//     bits either way, and independent of how bodies are sharded over GPUs.
//   Two ways of bringing a source record to the 64 lanes that all need the same one (bitwise the same result):
//   - LDS tiles (all_pairs_force_kernel): tiles staged in LDS by all 256 lanes with a register prefetch of the
//     next tile; the inner loop reads each record as an LDS broadcast (ds_read_b128) into VGPRs;
//   - scalar stream (all_pairs_force_sgpr_kernel, default from 2048 bodies): the record is wave-uniform, so it belongs in
//     SGPRs: ONE pre-pass launch per call (k1_prepare_kernel) packs the records (32 B x N), sums the positions' moments for the
//     pair rule and resets the turn words; every wave streams its slice with s_load_dwordx16 two batches deep (inline asm) and the
//     VALU instructions take their source operands from SGPRs.  No staging loads, no LDS traffic, no barriers in the loop, half
//     the VGPRs.
//   Pair rule (common.hpp, pair_batch): f64 is reciprocal-free — 16 full-rate ops + v_rsq_f64 per pair on sparse systems
//   (the launch-level far mode: the positions' variances, common.hpp k1_rule, say that few batches can hold a pair closer than 2),
//   17 + 1 on dense ones; pairs below 2^-16 take the guarded reciprocal form.
//   Measured: N = 2^20 galaxy 602-633 ms per pass by the clock the box sustains = 44-46.5 % of the 78.6 TF FP64 vector peak
//   (profiles/r05/bench_n1.json; VALU 98 % busy: 97 % of what 16 + rsq allow).
//
// K2 (replaces src/all_pairs.h:29-50, intended semantics).  Lanes run along the SOURCE axis (one
// ordered pair per lane and step), each wave owns 64 targets whose positions it broadcasts with
// v_readlane; the partial sums of 16 (f32) / 8 (f64) targets stay in registers over a whole 2048- / 1024-record tile and are
// reduced together by a transposed DPP butterfly (transpose_reduce), lane t keeps target t's sum, and after its tiles each
// wave issues D coalesced atomic adds.  grid.y splits the source range so small N still fills the chip.
// Config 3 (f32, N = 262 144): 23.9 ms = 36.5 % of the FP32 vector peak.
//
// K3 (replaces src/system.h:52-60).  Pure HBM stream (7*D*sizeof(T) bytes/body), flat elementwise
// over count*D scalars, FP contraction off so it is bit-identical to the reference's x86 -O2 build.
#include "common.hpp"

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

namespace nbody {

constexpr int kBlock = 256;  // 4 waves
constexpr int kWaves = kBlock / 64;

struct ap_config {
  int split = 0;  // 0 = auto
  int tpt   = 0;  // targets per thread, 0 = auto
  int path  = 0;  // source path: 0 = auto (scalar stream), 1 = LDS tiles, 2 = scalar stream
};
// process-wide default in NBODY_TUNING encoding (0 = all auto); a view with a non-zero `tuning` overrides it
static std::atomic<uint32_t> g_ap_default{0};

static ap_config ap_config_of(const nbody_state* s) {
  const uint32_t t = s->tuning ? s->tuning : g_ap_default.load(std::memory_order_relaxed);
  ap_config c;
  c.split = int(t & 15u);
  c.tpt   = int((t >> 4) & 3u);
  c.path  = int((t >> 6) & 3u);
  return c;
}

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int R, int JS>
__global__ __launch_bounds__(kBlock) void all_pairs_force_kernel(const T* __restrict__ m, const T* __restrict__ x,
                                                                 T* __restrict__ a, T c, uint32_t sz, uint32_t first,
                                                                 uint32_t count, const k1_rule* __restrict__ rule) {
  using rec_t = src_rec<T, D>;
  constexpr int TG  = kWaves / JS;    // target groups per block
  constexpr int TB  = TG * 64 * R;    // targets per block
  constexpr int LPT = kTileJ / kBlock;  // source records each lane stages per tile
  constexpr int SUB = kTileJ / JS;    // sources of a tile handled by one wave

  __shared__ rec_t tile[kTileJ];
  __shared__ T partial[(JS > 1) ? (JS - 1) * TG * 64 * R * D : 1];

  const int lane   = threadIdx.x & 63;
  const int wave   = threadIdx.x >> 6;
  const int tgroup = wave / JS;
  const int jpart  = wave % JS;

  // targets of this lane
  T xi[R][D], acc[R][D];
  uint32_t ti[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t local = blockIdx.x * TB + tgroup * (64 * R) + r * 64 + lane;
    ti[r]          = local;
    uint64_t i     = uint64_t(first) + (local < count ? local : 0u);  // clamp: out-of-range lanes compute, never store
#pragma unroll
    for (int k = 0; k < D; ++k) {
      xi[r][k]  = x[i * D + k];
      acc[r][k] = T(0);
    }
  }

  const uint32_t ntiles = (sz + kTileJ - 1) / kTileJ;
  const pair_consts<T> pc;
  const bool ffar = ap_far_mode(rule);

  // register staging of one tile: LPT records per lane
  rec_t stage[LPT];
  auto stage_load = [&](uint32_t t) {
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      uint64_t j = uint64_t(t) * kTileJ + q * kBlock + threadIdx.x;
      if (j < sz) {
#pragma unroll
        for (int k = 0; k < D; ++k) stage[q].p[k] = x[j * D + k];
        stage[q].m = m[j];
      } else {  // padding: zero mass contributes exactly 0
#pragma unroll
        for (int k = 0; k < D; ++k) stage[q].p[k] = T(0);
        stage[q].m = T(0);
      }
      if (D == 2) stage[q].p[2] = T(0);
    }
  };

  stage_load(0);
  auto run = [&](auto ff) {  // the tile loop, once per pair rule (pair_batch)
    constexpr bool FF = decltype(ff)::value;
    for (uint32_t t = 0; t < ntiles; ++t) {
      __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
      for (int q = 0; q < LPT; ++q) tile[q * kBlock + threadIdx.x] = stage[q];
      __syncthreads();
      if (t + 1 < ntiles) stage_load(t + 1);  // in flight while this tile is consumed

      const rec_t* src = &tile[jpart * SUB];
      constexpr int U  = 64 / int(sizeof(rec_t));  // the scalar-stream form's batch: 2 records in f64, 4 in f32
#pragma unroll 2
      for (int jj = 0; jj < SUB; jj += U) {
        rec_t s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = src[jj + u];  // wave-uniform address: LDS broadcast
        pair_batch<T, D, R, U, FF>(acc, xi, s, pc);
      }
    }
  };
  if (ffar) run(std::true_type{});
  else run(std::false_type{});

  // combine the JS source-split partials in wave order, then a = c * sum
  if constexpr (JS > 1) {
    __syncthreads();
    if (jpart > 0) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) partial[((((jpart - 1) * TG + tgroup) * R + r) * D + k) * 64 + lane] = acc[r][k];
    }
    __syncthreads();
    if (jpart == 0) {
#pragma unroll
      for (int p = 1; p < JS; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int k = 0; k < D; ++k) acc[r][k] += partial[((((p - 1) * TG + tgroup) * R + r) * D + k) * 64 + lane];
    }
  }
  if (jpart == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (ti[r] < count) {
#pragma unroll
        for (int k = 0; k < D; ++k) a[uint64_t(ti[r]) * D + k] = c * acc[r][k];
      }
    }
  }
}

// Scalar-stream form: pre-pass that packs (x, m) into aligned records, zero-mass padding up to a whole tile.
// (Measured and not kept, round 4: laying the records out in the order a wave reads them — one contiguous run per (chunk, slice),
// so that the stream's next address is `+ 64 bytes` instead of ten scalar operations on the record number.  Bitwise the same
// results, half the scalar instructions in the loop, and SLOWER: f32 N = 262 144 21.8 -> 22.5 ms dense, 19.3 -> 19.7 sparse — the
// eight slices' streams of a block then start 128 KB apart and fall into the same sets of the 16 KB scalar cache.)
template <typename T, int D>
__global__ __launch_bounds__(kBlock) void pack_sources_kernel(const T* __restrict__ m, const T* __restrict__ x,
                                                              src_rec<T, D>* __restrict__ out, uint32_t sz, uint32_t padded) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= padded) return;
  src_rec<T, D> r;
#pragma unroll
  for (int k = 0; k < 3; ++k) r.p[k] = (k < D && j < sz) ? x[uint64_t(j) * D + (k < D ? k : 0)] : T(0);
  r.m    = j < sz ? m[j] : T(0);
  out[j] = r;
}

// What a K1 launch needs done first, in ONE launch (rounds 2-4: a memset, the extent kernel, the pack kernel and another memset —
// four dependent launches, 20 us of a 73 us call at n = 8192):
//   PACK     the (x, m) records of the scalar-stream form, as pack_sources_kernel above;
//   MOMENTS  the first and second moments of all sz positions, relative to body 0 (so that a system far from the origin does not
//            cancel), summed per block in a fixed order; the LAST block to deliver (a ticket) adds the blocks' sums in index order
//            and writes the rule (common.hpp: k1_rule) — the same bits whichever block comes last;
//   and the last block also hands the turn of every target group back to chunk 0 (the words K1's chunks pass around).
// One block per 256 bodies of the padded set; the ticket returns to 0, so a recorded launch can be replayed.
template <typename T, int D, bool PACK, bool MOMENTS>
__global__ __launch_bounds__(kBlock) void k1_prepare_kernel(const T* __restrict__ m, const T* __restrict__ x, src_rec<T, D>* __restrict__ out,
                                                            uint32_t sz, uint32_t padded, k1_rule* rule, double* partial,
                                                            uint32_t* turn, uint32_t turn_words) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  T p[D];
#pragma unroll
  for (int k = 0; k < D; ++k) p[k] = j < sz ? x[uint64_t(j) * D + k] : T(0);
  if constexpr (PACK) {
    if (j < padded) {
      src_rec<T, D> r;
#pragma unroll
      for (int k = 0; k < 3; ++k) r.p[k] = k < D ? p[k < D ? k : 0] : T(0);
      r.m    = j < sz ? m[j] : T(0);
      out[j] = r;
    }
  }
  __shared__ double red[2 * D][kWaves];
  __shared__ bool last_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (MOMENTS) {
    double s1[D], s2[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double d = j < sz ? double(p[k]) - double(x[k]) : 0.0;  // relative to body 0 (sz >= 1)
      s1[k] = d;
      s2[k] = d * d;
    }
#pragma unroll
    for (int k = 0; k < D; ++k)
      for (int off = 32; off > 0; off >>= 1) {  // butterfly: every lane ends with the same sum, formed in the same order
        s1[k] += __shfl_xor(s1[k], off, 64);
        s2[k] += __shfl_xor(s2[k], off, 64);
      }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        red[k][wave]     = s1[k];
        red[D + k][wave] = s2[k];
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
      double v = red[threadIdx.x][0];
      for (int w = 1; w < kWaves; ++w) v += red[threadIdx.x][w];
      __hip_atomic_store(partial + size_t(blockIdx.x) * (2 * D) + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // The ticket: the block that draws the last number knows that every other block's partial sums are out.  No __threadfence():
  // an agent-scope fence writes back and invalidates the XCD's whole L2 — which holds the records this kernel has just packed —
  // and 4096 of them made this launch 430 us at N = 2^20.  The partial sums are agent-scope stores (written through), the wave that
  // issued them waits for their acknowledgement (s_waitcnt 0) before its lane 0 draws the ticket, and the last block reads them
  // with agent-scope loads; everything else this kernel writes is read by the NEXT kernel only.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) last_s = __hip_atomic_fetch_add(&rule->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  __syncthreads();
  if (!last_s) return;
  if constexpr (MOMENTS) {
    // thread t: moment q = t % (2 D) of the blocks g, g + G, g + 2 G, ... in ascending order (g = t / (2 D), G groups); eight
    // loads in flight at a time (one after the other this loop was 50 us of latency at N = 262 144), added in index order
    double acc = 0.0;
    constexpr int G = kBlock / (2 * D), U = 8;
    const int q = threadIdx.x % (2 * D), g = threadIdx.x / (2 * D);
    if (g < G) {
      for (uint32_t b0 = uint32_t(g); b0 < gridDim.x; b0 += G * U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t b = b0 + uint32_t(u) * G;
          v[u] = b < gridDim.x ? __hip_atomic_load(partial + size_t(b) * (2 * D) + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];  // (+ 0.0 past the end: no change)
      }
    }
    __shared__ double fin[kBlock];
    fin[threadIdx.x] = g < G ? acc : 0.0;
    __syncthreads();
    if (threadIdx.x < 2 * D) {  // thread q: the G group sums of moment q in index order
      double tot = 0.0;
#pragma unroll 6
      for (int gg = 0; gg < G; ++gg) tot += fin[gg * (2 * D) + threadIdx.x];
      fin[threadIdx.x] = tot;   // (row 0 of fin: its values have been read by exactly this thread)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double vol = 1.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double mean = fin[k] / double(sz), var = fin[D + k] / double(sz) - mean * mean;
        vol *= var > 0.0 ? __builtin_sqrt(12.0 * var) : 0.0;
      }
      rule->volume = vol;
      rule->sparse = vol >= kFarMinVolume<D> ? 1u : 0u;
    }
  }
  for (uint32_t w = threadIdx.x; w < turn_words; w += kBlock) turn[w] = 0u;  // chunk 0 holds every turn
  if (threadIdx.x == 0) __hip_atomic_store(&rule->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int JS>
constexpr int kSgprWaves = JS > kWaves ? JS : kWaves;  // waves per block of the scalar-stream form

// Scalar-stream form.  grid = (target blocks, source chunks): block (b, y) sums chunk y's tiles for target block b — its JS slices'
// sums added in slice order through LDS — and adds that sum s_y to the target's running total IN CHUNK ORDER:
//     a = c * (((s_0 + s_1) + s_2) + ...),
// the total living in `a` itself.  Block (b, y) waits until turn[b] == y (block (b, y - 1) has added its sum), adds, and passes the
// turn on; the last chunk applies c.  No chunk-sum scratch (403 MB at N = 2^20 until round 3) and no combine launch.
//
// Progress rests on ONE assumption that is a property of the dispatcher, not of the ISA: a block waits only for blocks of smaller
// linear index, and the workgroups of a grid are started in linear index order (dealt to the XCDs round-robin, in order on each),
// so the unfinished block of smallest index is always running and never waits.  In practice nobody waits at all — block (b, y - 1)
// finished a whole round of blocks earlier (k1_status.waits counts the waves that did poll, .polls their polls).
//
// If the assumption ever fails, the failure is LOUD (k1_status, nbody_all_pairs_status, nbody_stream_sync, nbody_download):
//   * a wave that has polled h.spins times (kTurnSpins: minutes) gives up: it swaps kTurnPoison into the turn word, records
//     (block, group, chunk) in the stream's sticky status block and overwrites the group's total with NaN;
//   * every wave that finds kTurnPoison while polling overwrites the total with NaN and leaves;
//   * a turn is passed on with a compare-and-swap y -> y + 1; a holder that finds the poison instead writes NaN over what it has
//     just added.
//   EVERY wave that meets the poison writes NaN (round 6; the writes are idempotent), so the order in which they come does not
//   matter: the last store to the group's rows is a NaN in every interleaving — a holder that was in the middle of its add when
//   the poison arrived stores its finite sum, then fails its compare-and-swap and stores NaN behind it; a predecessor that had
//   not even begun to poll when its successors gave up (the case this protocol exists for: a block started late or out of order)
//   finds the poison at its first poll.  Until round 6 only the wave that lost the compare-and-swap wrote NaN, and that late
//   predecessor left a finite partial sum s_0 + ... + s_{y'-1} behind, reported by the status word alone.
//   So a poisoned group ends as NaN (never as a finite partial sum), and the host call that waits for the stream returns
//   NBODY_ERR_STATE naming the block and the chunk.
// The running total is read and written with agent-scope accesses (it changes hands between CUs and XCDs): the predecessor's
// stores are acknowledged (s_waitcnt 0) before its turn store is issued, the successor issues its loads after it has seen the turn.
// What fixes the ROUNDING is unchanged from rounds 2-3, so every shard window still sums exactly as the whole system does.
constexpr uint32_t kTurnSpins  = 1u << 26;
constexpr uint32_t kTurnPoison = 0xffffffffu;
struct k1_status {  // one per (device, stream), device memory, zeroed when allocated; sticky until nbody_all_pairs_status(..., clear)
  uint32_t err, block, group, chunk;
  unsigned long long polls, waits;
};
struct k1_handoff {
  void* sums;         // small launches (k1_collect): every chunk's sum of every target, [chunk][target block][group][r][k][lane]; else nullptr
  uint32_t* turn;     // one word per (target block, target group): the chunk whose sum is added next, or kTurnPoison
                      // (k1_collect: the number of chunks that have delivered their sums)
  k1_status* status;
  uint32_t spins;     // polls before a waiting wave gives up (kTurnSpins; lowered only by the experiments build's tests)
  uint32_t delay;     // s_sleep(127) rounds before a turn is passed on (0; the experiments build's tests make successors wait)
  uint32_t late;      // s_sleep(127) rounds before chunk 1's waves look at the turn word at all (0; experiments: a block started late)
};
template <typename T, int D, int R, int JS, int RULE = 0>
__global__ __launch_bounds__(64 * kSgprWaves<JS>) void all_pairs_force_sgpr_kernel(const src_rec<T, D>* __restrict__ packed,
                                                                                   const T* __restrict__ x, T* a, T c, uint32_t sz,
                                                                                   uint32_t first, uint32_t count,
                                                                                   uint32_t tiles_per_chunk, k1_handoff h,
                                                                                   const k1_rule* __restrict__ rule) {
  using rec_t = src_rec<T, D>;
  constexpr int TG  = kSgprWaves<JS> / JS;
  constexpr int TB  = TG * 64 * R;
  constexpr int SUB = kTileJ / JS;
  __shared__ T partial[(JS > 1) ? (JS - 1) * TG * 64 * R * D : 1];
  const int lane   = threadIdx.x & 63;
  const int wave   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tgroup = wave / JS;
  const int jpart  = wave % JS;
  T xi[R][D], acc[R][D];
  uint32_t ti[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t local = blockIdx.x * TB + tgroup * (64 * R) + r * 64 + lane;
    ti[r]          = local;
    uint64_t i     = uint64_t(first) + (local < count ? local : 0u);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      xi[r][k]  = x[i * D + k];
      acc[r][k] = T(0);
    }
  }
  // source chunk of this block (grid.y): tiles [t0, t1) of the padded source set; one chunk = everything when grid.y == 1
  const uint32_t ntiles = (sz + kTileJ - 1) / kTileJ;
  const uint32_t t0     = blockIdx.y * tiles_per_chunk;
  const uint32_t t1     = min(ntiles, t0 + tiles_per_chunk);
  const pair_consts<T> pc;
  const bool ffar       = ap_far_mode(rule);
  const uint32_t nsteps = (t1 - t0) * SUB;  // sources this wave visits: its SUB-record slice of every tile, in tile order
  constexpr int U = 64 / int(sizeof(rec_t));  // records per 64-byte batch (2 in f64, 4 in f32); SUB % (2 * U) == 0
  struct batch_t {
    rec_t r[U];
  };
  auto batch = [&](uint32_t k) { return packed + (uint64_t(t0 + k / SUB) * kTileJ + uint32_t(jpart) * SUB + (k % SUB)); };
  // Two SGPR buffers, each requested (s_load_dwordx16) one compute phase before it is consumed.  Written with inline
  // asm: hipcc folds a loop-carried load from read-only memory back into a load at the loop top and waits for it there.
  // SMEM returns out of order, so the only usable wait is lgkmcnt(0): wait for X, request Y, consume X.
  auto run = [&](auto ff) {  // the source stream, once per pair rule (pair_batch)
    constexpr bool FF = decltype(ff)::value;
    sgpr16 A = sload16(batch(0), xi[0][0]), B;
    for (uint32_t k = 0; k < nsteps; k += 2 * U) {
      swait(A, acc[0][0]);
      B = sload16(batch(k + U), xi[0][0]);
      {
        const batch_t ba = __builtin_bit_cast(batch_t, A);
        pair_batch<T, D, R, U, FF>(acc, xi, ba.r, pc);
      }
      swait(B, acc[0][0]);
      A = sload16(batch(k + 2 * U < nsteps ? k + 2 * U : k), xi[0][0]);  // the last iteration re-requests its own batch
      {
        const batch_t bb = __builtin_bit_cast(batch_t, B);
        pair_batch<T, D, R, U, FF>(acc, xi, bb.r, pc);
      }
    }
    swait(A, acc[0][0]);  // nothing in flight when the wave goes on
  };
  if constexpr (RULE == 1) run(std::false_type{});      // (experiments: one rule per instantiation, forced from the host)
  else if constexpr (RULE == 2) run(std::true_type{});
  else if (ffar) run(std::true_type{});  // two copies of the loop: inside ONE loop hipcc hoists the rules' common head above the branch
  else run(std::false_type{});
  if constexpr (JS > 1) {
    if (jpart > 0) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) partial[((((jpart - 1) * TG + tgroup) * R + r) * D + k) * 64 + lane] = acc[r][k];
    }
    __syncthreads();
    if (jpart == 0) {
#pragma unroll
      for (int p = 1; p < JS; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int k = 0; k < D; ++k) acc[r][k] += partial[((((p - 1) * TG + tgroup) * R + r) * D + k) * 64 + lane];
    }
  }
  if (jpart != 0) return;
  const uint32_t y = blockIdx.y, last = gridDim.y - 1u;
  uint32_t* const tw = h.turn + blockIdx.x * TG + tgroup;  // nullptr + ... when there is one chunk: never dereferenced (y == last == 0)
  if (h.sums != nullptr) {
    // Small launches (all blocks resident within a few rounds: the blocks of one target group's sixteen chunks finish TOGETHER, and
    // a chain of turns is fifteen dependent round trips through memory — 18 of 41 us at N = 4096, 42 of 76 us in float at 8192):
    // every chunk's wave stores its sum, then draws a ticket; whoever draws the last one — whichever chunk it is — adds the
    // sums IN CHUNK ORDER, ((s_0 + s_1) + s_2) + ..., applies c and writes `a`: the same bits as the turns give, no waiting, no
    // failure mode.  The sums are agent-scope stores acknowledged (s_waitcnt 0) before the ticket is drawn.
    const size_t group_scalars = size_t(R) * D * 64;
    T* const all  = static_cast<T*>(h.sums);
    T* const mine = all + ((size_t(y) * gridDim.x + blockIdx.x) * TG + tgroup) * group_scalars;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < D; ++k) __hip_atomic_store(mine + (r * D + k) * 64 + lane, acc[r][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    uint32_t drawn = 0;
    if (lane == 0) drawn = __hip_atomic_fetch_add(tw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (uint32_t(__builtin_amdgcn_readfirstlane(int(drawn))) != last) return;
    T tot[R][D];
    for (uint32_t yy = 0; yy <= last; ++yy) {
      const T* src = all + ((size_t(yy) * gridDim.x + blockIdx.x) * TG + tgroup) * group_scalars;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const T v = __hip_atomic_load(src + (r * D + k) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tot[r][k] = yy == 0 ? v : tot[r][k] + v;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (ti[r] < count) {
#pragma unroll
        for (int k = 0; k < D; ++k) a[uint64_t(ti[r]) * D + k] = c * tot[r][k];
      }
    return;
  }
  bool poisoned = false;
  if (y > 0 && h.turn != nullptr) {  // my turn?  (wave-uniform address: every lane reads the same value)
    uint32_t spins = 0, seen;
    if (y == 1)
      for (uint32_t d = 0; d < h.late; ++d) __builtin_amdgcn_s_sleep(127);  // 0 rounds, except in the hand-off tests of the experiments build
    while ((seen = __hip_atomic_load(tw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != y) {
      if (seen == kTurnPoison) {  // somebody gave up on this group: nobody will add to its total again, NaN goes over it
        poisoned = true;
        break;
      }
      if (++spins > h.spins) {  // give up (see above)
        if (lane == 0) {
          __hip_atomic_exchange(tw, kTurnPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (atomicCAS(&h.status->err, 0u, 1u) == 0u) {
            h.status->block = blockIdx.x;
            h.status->group = uint32_t(tgroup);
            h.status->chunk = y;
          }
        }
        poisoned = true;  // whatever the word held — this wave's own number (the turn came between the last poll and the swap), a
        break;            // predecessor's that has yet to come, or poison already — the group's total is overwritten with NaN
      }
      __builtin_amdgcn_s_sleep(8);
    }
    // The loads of the running total below must be ISSUED after the poll that saw the turn.  The accesses are relaxed agent-scope
    // atomics (sc1: they bypass this XCD's L2), so nothing has to be invalidated; what is needed is that the compiler keeps them
    // behind the loop — a wavefront-scope acquire fence says so and emits no instruction (tools/check_k1_handoff.py verifies in the
    // built code object that every load of the total carries sc1 and follows the last poll).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (spins && lane == 0) {
      atomicAdd(&h.status->polls, (unsigned long long)spins);
      atomicAdd(&h.status->waits, 1ull);
    }
  }
  // The total's R * D * 64 scalars of this target group are contiguous in `a`: through LDS (the slices' partials are spent) every
  // lane takes scalars e = q * 64 + lane, so each access is one full-width coalesced instruction — per component the lanes would
  // touch every line three times (measured: 3.2 GB of traffic per launch at N = 2^20 instead of 1.3).
  const uint32_t gbase = blockIdx.x * TB + tgroup * (64 * R);  // first target of the group
  auto add_sum = [&](bool nan) {  // nan: overwrite the group's total with NaN instead
    if constexpr (JS > 1) {
      T* stage = partial + size_t(tgroup) * R * D * 64;  // [(jpart - 1 = 0) * TG + tgroup] block of the partials: read above, free now
      if (!nan) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int k = 0; k < D; ++k) stage[(r * 64 + lane) * D + k] = acc[r][k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int q = 0; q < R * D; ++q) {
        const uint32_t e = uint32_t(q) * 64u + uint32_t(lane);
        if (gbase + e / D < count) {
          T* slot = a + uint64_t(gbase) * D + e;
          T t     = nan ? T(__builtin_nan("")) : stage[e];
          if (!nan && y > 0) t = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + t;  // ((s_0 + s_1) + ...) + s_y
          if (!nan && y == last) t = c * t;
          __hip_atomic_store(slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (ti[r] < count) {
#pragma unroll
          for (int k = 0; k < D; ++k) {
            T* slot = a + uint64_t(ti[r]) * D + k;
            T t     = nan ? T(__builtin_nan("")) : acc[r][k];
            if (!nan && y > 0) t = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + t;
            if (!nan && y == last) t = c * t;
            __hip_atomic_store(slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  };
  add_sum(poisoned);
  if (y < last && !poisoned && h.turn != nullptr) {  // pass the turn on once the stores above have been acknowledged
    for (uint32_t d = 0; d < h.delay; ++d) __builtin_amdgcn_s_sleep(127);  // 0 rounds, except in the hand-off tests of the experiments build
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's stores are at the agent's coherence point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // and the compiler keeps them above the hand-over (no instruction)
    __builtin_amdgcn_wave_barrier();
    uint32_t held = y;
    if (lane == 0) {
      __hip_atomic_compare_exchange_strong(tw, &held, y + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    held = __builtin_amdgcn_readfirstlane(held);  // what the word held: y, or kTurnPoison left by a successor that gave up
    if (held != y) add_sum(true);
  }
}

// Scratch (packed sources; per-chunk sums), one slot per (device, stream) that has called the scalar-stream form (grow-only).  A context
// reserves its buffer when it is created (nbody_create), so that a step recorded with nbody_graph_begin never has to
// allocate; other callers get theirs on the first call, which therefore must not be inside a capture.  A buffer that is
// outgrown is retired, not freed: a recorded graph or a launch another host thread is about to make may still name it.
// Retired buffers go when the slot is released (nbody_destroy) — at most log2(growth) of them, the largest last.
namespace {
struct scratch_buf {
  void* ptr  = nullptr;
  size_t cap = 0;
};
struct packed_slot {
  int device;
  hipStream_t stream;
  scratch_buf buf[6];  // 0: packed sources, 1: K1 turn words, 2: energies work area, 3: the pair rule + partial moments, 4: K1 hand-off status (k1_status), 5: K1 chunk sums of small launches
  std::vector<void*> retired;
  bool k1_dirty  = false;  // a K1 that passes turns (or a recorded step, which may hold one) has been queued since the status was last read
  bool k1_failed = false;  // the last read found the sticky error set
};
std::mutex g_packed_mu;
std::vector<packed_slot> g_packed_slots;
}  // namespace

int ap_scratch_get(hipStream_t st, int which, size_t bytes, void** out) {
  const int dev = stream_device(st);
  std::lock_guard<std::mutex> lock(g_packed_mu);
  packed_slot* slot = nullptr;
  for (auto& sl : g_packed_slots)
    if (sl.stream == st && sl.device == dev) slot = &sl;
  if (slot && slot->buf[which].cap >= bytes) {
    *out = slot->buf[which].ptr;
    return NBODY_OK;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (st != nullptr) (void)hipStreamIsCapturing(st, &cs);
  if (cs != hipStreamCaptureStatusNone) {
    set_error("all-pairs: the scratch buffers of this stream must exist before a step is recorded "
              "(call nbody_all_pairs_force once outside nbody_graph_begin/end, or use a context from nbody_create)");
    return NBODY_ERR_STATE;
  }
  if (!slot) {
    g_packed_slots.push_back({dev, st, {}, {}});
    slot = &g_packed_slots.back();
  }
  void* fresh = nullptr;
  NB_HIP(hipMalloc(&fresh, bytes));  // on the stream's device: the caller holds a device_guard
  if (which == 4 || which == 3) {    // the status block and the rule's ticket start at zero: ordered on the stream that will use
    // them (every launch that reads them is queued on `st` behind this call) — a synchronous memset would go through the legacy
    // stream, which has no defined order with a non-blocking `st` and would invalidate another thread's global-mode capture
    if (hipError_t e = hipMemsetAsync(fresh, 0, bytes, st); e != hipSuccess) {
      (void)hipFree(fresh);
      return hip_fail(e, "hipMemsetAsync(K1 scratch)", __FILE__, __LINE__);
    }
  }
  if (slot->buf[which].ptr) slot->retired.push_back(slot->buf[which].ptr);
  slot->buf[which].ptr = fresh;
  slot->buf[which].cap = bytes;
  *out                 = fresh;
  return NBODY_OK;
}

void ap_scratch_release(hipStream_t st) {
  const int dev = stream_device(st);
  std::lock_guard<std::mutex> lock(g_packed_mu);
  for (size_t i = 0; i < g_packed_slots.size(); ++i) {
    if (g_packed_slots[i].stream == st && g_packed_slots[i].device == dev) {
      for (auto& b : g_packed_slots[i].buf) (void)hipFree(b.ptr);
      for (void* p : g_packed_slots[i].retired) (void)hipFree(p);
      g_packed_slots.erase(g_packed_slots.begin() + long(i));
      return;
    }
  }
}

// ---- launch plan -----------------------------------------------------------------------------------------------
// What fixes the ROUNDING ORDER of a target's sum is derived from sz alone — never from first/count — so every shard
// of a multi-GPU run sums exactly as the single-GPU run does:
//   * split  (JS): the 512-record tile is cut into JS slices, one per wave of a target group; 8 slices (512-thread
//     blocks, scalar-stream form only) once sz >= 2048, else 4;
//   * chunks (Y):  the tile sequence is cut into Y runs, one per grid.y, each run summed by its own block and the Y
//     sums combined in run order by a second kernel.  Equal-sized blocks that start together finish together, so a
//     launch costs ceil(blocks / resident slots) block times: 1563 blocks on 1024 slots (N = 10^5) ran at 31 % of peak
//     against 39 % for N = 2^20.  Y makes the blocks short and many (>= 16 rounds for the whole system, >= 8 for a
//     1/8 shard) whatever N is; co-resident blocks then also share one 128 KB..1 MB run of source records in L2.
// What does NOT touch the order may depend on the window: targets per lane (R) and the LDS / scalar-stream choice.
struct k1_plan {
  bool scalar = false;
  int r = 1, js = 4;
  uint32_t chunks = 1, tiles_per_chunk = 0;
};

static int auto_split(uint32_t sz) { return sz >= 2048u ? 8 : 4; }

// Chunks from sz alone: 16, or more where 16 would leave a chunk above 65 536 records (up to 64), and never more than tiles.
//   * at least 16: every size runs >= 16 rounds of short blocks and small systems get the waves the scalar stream needs.
//     Measured (profiles/r02/k1_chunks_tuning.txt, k1_small_n_tuning.txt; f64, ms per pass, chunks 1 / 16): N = 4096: 0.070 /
//     0.023, 10^4: 0.166 / 0.083, 3*10^4: 0.66 / 0.57, 65 536: 2.64 / 2.50, 10^5: 7.45 / 5.81; more change nothing (N = 10^5:
//     64 chunks 5.87 ms).
//   * chunks of at most 65 536 records (2 MB in f64): blocks are dispatched x-fastest, so the blocks in flight on an XCD stream
//     the SAME chunk, and a chunk of half the XCD's 4 MiB L2 stays there.  N = 2^20 (rocprofv3 FETCH_SIZE per launch, kernel
//     time unchanged within 0.2 %): 4 chunks of 8 MB 8.4 GB, 8 chunks 4.7 GB, 16 chunks of 2 MB 0.46 GB.
void ap_auto_chunks(uint32_t sz, uint32_t* chunks, uint32_t* tiles_per_chunk) {
  const uint32_t ntiles = (sz + kTileJ - 1) / kTileJ;
  uint32_t y = 16;
  while (y < 64 && uint64_t(y) * 65536u < sz) y *= 2;
  if (y > ntiles) y = ntiles ? ntiles : 1;
  uint32_t tpc = ntiles ? (ntiles + y - 1) / y : 1;
  *tiles_per_chunk = tpc;
  *chunks          = ntiles ? (ntiles + tpc - 1) / tpc : 1;
}

template <typename T>
static int plan_all_pairs(const nbody_state* s, k1_plan* out) {
  const ap_config cfg = ap_config_of(s);
  k1_plan p;
  p.js = cfg.split ? cfg.split : auto_split(s->sz);
  p.r  = cfg.tpt;
  // Source path (bitwise identical results at equal split and chunks).  The scalar stream pays an SMEM round trip per
  // 64-byte batch, which needs several waves per SIMD to hide — the source chunks supply them at every size (N = 10^4,
  // f64: LDS tiles 0.19 ms, scalar stream unchunked 0.17-0.33 ms, with 16 chunks 0.083 ms), so the automatic choice is the
  // scalar stream wherever the 8-slice split applies; the LDS-tile form remains for tiny systems and on request.
  const uint64_t waves_r1 = (uint64_t(s->count) + 63) / 64 * uint64_t(p.js);
  p.scalar                = p.js == 8 || cfg.path == 2 || (cfg.path == 0 && waves_r1 >= 4096);
  if (p.js == 8 && cfg.path == 1) {
    set_error("all-pairs: the LDS-tile form has at most 4 source slices; sz = %u uses 8 (nbody_all_pairs_configure(4, ...) to force 4)",
              s->sz);
    return NBODY_ERR_ARG;
  }
  // source chunks: only where the split is the automatic one of large systems (so every explicit configuration keeps
  // its single-chunk order) and only in the scalar-stream form
  p.tiles_per_chunk = (s->sz + kTileJ - 1) / kTileJ;
  if (p.scalar && cfg.split == 0 && p.js == 8) ap_auto_chunks(s->sz, &p.chunks, &p.tiles_per_chunk);
  if (const char* e = experiment_env("NBODY_K1_CHUNKS"); e && p.scalar) {  // -DNBODY_EXPERIMENTS builds only (tools/tune_chunks.py): changes the rounding order
    const uint32_t ntiles = (s->sz + kTileJ - 1) / kTileJ;
    uint32_t y            = uint32_t(atoi(e));
    if (y >= 1 && y <= ntiles) {
      p.tiles_per_chunk = (ntiles + y - 1) / y;
      p.chunks          = (ntiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk;
    }
  }
  if (p.r == 0) {
    // R = 2 halves the scalar-cache traffic per pair and the block count; it pays once the grid is many rounds deep.
    const uint64_t blocks_r2 = (uint64_t(s->count) + 127) / 128 * p.chunks;
    if (p.scalar) p.r = (sizeof(T) == 8 && blocks_r2 >= 4096) ? 2 : 1;
    else p.r = blocks_r2 * uint64_t(p.js) >= 8192 ? 2 : 1;
  }
  *out = p;
  return NBODY_OK;
}

// packs (x, m) of all sz bodies into this stream's record buffer (stream-ordered); used by K1 and by the energies
int ap_pack_sources(const nbody_state* s, hipStream_t st, void** packed_out) {
  const uint32_t padded = (s->sz + kTileJ - 1) / kTileJ * kTileJ;
  const size_t tsz      = s->dtype == NBODY_F32 ? 4 : 8;
  void* scratch         = nullptr;
  if (int r = ap_scratch_get(st, 0, 4 * tsz * size_t(padded), &scratch)) return r;
  int rc = dispatch(s->dtype, s->dim, [&](auto tg) {
    using T         = typename decltype(tg)::type;
    constexpr int D = decltype(tg)::dim;
    hipLaunchKernelGGL((pack_sources_kernel<T, D>), dim3((padded + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                       static_cast<const T*>(s->m), static_cast<const T*>(s->x), static_cast<src_rec<T, D>*>(scratch), s->sz, padded);
    NB_HIP(hipGetLastError());
    return int(NBODY_OK);
  });
  *packed_out = scratch;
  return rc;
}

// Bytes of scratch buffer 3 for a system of sz bodies: the rule, then one row of 2 D partial sums per prepare block.
static size_t ap_rule_bytes(uint32_t sz, int dim) {
  const size_t padded = (size_t(sz) + kTileJ - 1) / kTileJ * kTileJ;
  return 32 + sizeof(double) * 2 * size_t(dim) * (padded / kBlock);
}

// Everything a K1 launch needs done first, queued as ONE launch (k1_prepare_kernel): the packed records (pack: the scalar-stream
// form), the pair rule of the whole system (sz >= kFarMinBodies; nullptr = "dense" below) and the turn words handed back to chunk 0.
template <typename T, int D>
static int ap_prepare(const nbody_state* s, hipStream_t st, bool pack, src_rec<T, D>** packed_out, const k1_rule** rule_out,
                      uint32_t* turn, size_t turn_words) {
  const uint32_t padded = (s->sz + kTileJ - 1) / kTileJ * kTileJ;
  const bool moments    = s->sz >= kFarMinBodies;
  void* scratch         = nullptr;
  if (pack)
    if (int r = ap_scratch_get(st, 0, 4 * sizeof(T) * size_t(padded), &scratch)) return r;
  void* q = nullptr;
  if (int r = ap_scratch_get(st, 3, ap_rule_bytes(s->sz, D), &q)) return r;  // (the ticket lives there even without moments)
  auto* rule    = static_cast<k1_rule*>(q);
  auto* partial = reinterpret_cast<double*>(static_cast<char*>(q) + 32);
  auto* out     = static_cast<src_rec<T, D>*>(scratch);
  const dim3 grid(padded / kBlock), block(kBlock);
  const T *m = static_cast<const T*>(s->m), *x = static_cast<const T*>(s->x);
  const uint32_t words = uint32_t(turn_words);
  if (pack && moments) hipLaunchKernelGGL((k1_prepare_kernel<T, D, true, true>), grid, block, 0, st, m, x, out, s->sz, padded, rule, partial, turn, words);
  else if (pack) hipLaunchKernelGGL((k1_prepare_kernel<T, D, true, false>), grid, block, 0, st, m, x, out, s->sz, padded, rule, partial, turn, words);
  else if (moments) hipLaunchKernelGGL((k1_prepare_kernel<T, D, false, true>), grid, block, 0, st, m, x, out, s->sz, padded, rule, partial, turn, words);
  NB_HIP(hipGetLastError());
  if (packed_out) *packed_out = out;
  *rule_out = moments ? rule : nullptr;
  return NBODY_OK;
}

// How the chunks' sums of a launch meet (bitwise the same either way): collected by the last chunk to arrive when the whole grid is
// resident within two rounds (<= 2048 blocks: up to 3 MB of sums), passed from chunk to chunk in `a` otherwise (no scratch).
// Measured (profiles/r05/k1_handoff_cost.txt; f64, us per launch, collected / turns / no protocol at all): N = 2048 16.2 / 23.5 / 15.1,
// 4096 23 / 41 / 20, 8192 55 / 70 / 51, f32 8192 34 / 76 / 29 — and at 4096 blocks (N = 16 384) 176 / 172 / 172: turns from there on.
constexpr uint32_t kCollectMaxBlocks = 2048;
static bool k1_collect(uint32_t target_blocks, uint32_t chunks) {
  uint32_t limit = kCollectMaxBlocks;
  if (const char* e = experiment_env("NBODY_K1_COLLECT_MAX")) limit = uint32_t(atoi(e));  // -DNBODY_EXPERIMENTS builds only (tools/tune_small_k1.py)
  return chunks > 1 && uint64_t(target_blocks) * chunks <= limit;
}
template <typename T, int D, int R, int JS>
static size_t k1_collect_bytes(uint32_t target_blocks, uint32_t chunks) {
  constexpr int TG = kSgprWaves<JS> / JS;
  return sizeof(T) * size_t(chunks) * target_blocks * TG * R * D * 64;
}

// words of the turn array for a launch of `blocks` target blocks (one per target group of a block)
template <int R, int JS>
static size_t sgpr_turn_words(uint32_t count) {
  constexpr int TG = kSgprWaves<JS> / JS, TB = TG * 64 * R;
  return size_t((count + TB - 1) / TB) * TG;
}

template <typename T, int D, int R, int JS>
static int launch_all_pairs_sgpr(const nbody_state* s, const k1_plan& plan, hipStream_t st) {
  constexpr int TB = (kSgprWaves<JS> / JS) * 64 * R;
  uint32_t blocks  = (s->count + TB - 1) / TB;
  if (blocks == 0) return NBODY_OK;
  k1_handoff h{nullptr, nullptr, nullptr, kTurnSpins, 0u, 0u};
  if (plan.chunks > 1) {  // before anything is queued: a failed reservation leaves the stream untouched
    void* q = nullptr;
    if (int r = ap_scratch_get(st, 1, sizeof(uint32_t) * sgpr_turn_words<R, JS>(s->count), &q)) return r;
    h.turn = static_cast<uint32_t*>(q);
    if (k1_collect(blocks, plan.chunks)) {
      if (int r = ap_scratch_get(st, 5, k1_collect_bytes<T, D, R, JS>(blocks, plan.chunks), &q)) return r;
      h.sums = q;
    }
    if (int r = ap_scratch_get(st, 4, sizeof(k1_status), &q)) return r;
    h.status = static_cast<k1_status*>(q);
    // -DNBODY_EXPERIMENTS builds only (tests/test_gpu_all_pairs.py: the hand-off made to wait, and made to fail)
    if (const char* e = experiment_env("NBODY_K1_TURN_SPINS")) h.spins = uint32_t(strtoul(e, nullptr, 10));
    if (const char* e = experiment_env("NBODY_K1_HANDOFF_DELAY")) h.delay = uint32_t(strtoul(e, nullptr, 10));
    if (const char* e = experiment_env("NBODY_K1_HANDOFF_LATE")) h.late = uint32_t(strtoul(e, nullptr, 10));
    // timing experiment only (WRONG sums): no turn words — every chunk's block adds to whatever `a` holds without waiting
    if (const char* e = experiment_env("NBODY_K1_NO_HANDOFF"); e && e[0] == '1') h.turn = nullptr, h.sums = nullptr;
    if (const char* e = experiment_env("NBODY_K1_COLLECT"); e && e[0] == '0') h.sums = nullptr;  // experiments: turns at every size
  }
  src_rec<T, D>* packed = nullptr;
  const k1_rule* rule   = nullptr;
  if (int r = ap_prepare<T, D>(s, st, true, &packed, &rule, h.turn, h.turn ? sgpr_turn_words<R, JS>(s->count) : 0)) return r;
#ifdef NBODY_EXPERIMENTS
  if (const char* e = getenv("NBODY_K1_RULE_FORCE"); e && JS == 8 && D == 3) {  // timing experiment: one rule per instantiation
    if (e[0] == '1')
      hipLaunchKernelGGL((all_pairs_force_sgpr_kernel<T, D, R, JS, 1>), dim3(blocks, plan.chunks), dim3(64 * kSgprWaves<JS>), 0, st,
                         packed, static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->sz, s->first, s->count,
                         plan.tiles_per_chunk, h, rule);
    else
      hipLaunchKernelGGL((all_pairs_force_sgpr_kernel<T, D, R, JS, 2>), dim3(blocks, plan.chunks), dim3(64 * kSgprWaves<JS>), 0, st,
                         packed, static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->sz, s->first, s->count,
                         plan.tiles_per_chunk, h, rule);
    NB_HIP(hipGetLastError());
    return NBODY_OK;
  }
#endif
  hipLaunchKernelGGL((all_pairs_force_sgpr_kernel<T, D, R, JS>), dim3(blocks, plan.chunks), dim3(64 * kSgprWaves<JS>), 0, st,
                     packed, static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->sz, s->first, s->count,
                     plan.tiles_per_chunk, h, rule);
  NB_HIP(hipGetLastError());
  if (h.turn != nullptr && h.sums == nullptr) ap_status_mark(st);  // turns were passed: the next wait for this stream reads the status
  return NBODY_OK;
}

template <typename T, int D, int R, int JS>
static int launch_all_pairs(const nbody_state* s, hipStream_t st) {
  constexpr int TB = (kWaves / JS) * 64 * R;
  uint32_t blocks  = (s->count + TB - 1) / TB;
  if (blocks == 0) return NBODY_OK;
  const k1_rule* rule = nullptr;  // the same per-pair rule as the scalar-stream form (bitwise the same result)
  if (int r = ap_prepare<T, D>(s, st, false, nullptr, &rule, nullptr, 0)) return r;
  hipLaunchKernelGGL((all_pairs_force_kernel<T, D, R, JS>), dim3(blocks), dim3(kBlock), 0, st, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->sz, s->first, s->count, rule);
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

// calls fn(R, JS) as integral constants for the instantiation a plan names; false if there is none
template <typename F>
static bool with_k1_instance(const k1_plan& p, F&& fn) {
#define NB_CASE(RR, JJ)                                                                       \
  if (p.r == RR && p.js == JJ) {                                                              \
    fn(std::integral_constant<int, RR>{}, std::integral_constant<int, JJ>{});                 \
    return true;                                                                              \
  }
  NB_CASE(1, 1);
  NB_CASE(1, 2);
  NB_CASE(1, 4);
  NB_CASE(2, 1);
  NB_CASE(2, 2);
  NB_CASE(2, 4);
  if (p.scalar) {
    NB_CASE(1, 8);
    NB_CASE(2, 8);
  }
#undef NB_CASE
  return false;
}

template <typename T, int D>
static int all_pairs_dispatch(const nbody_state* s, hipStream_t st) {
  k1_plan p;
  if (int rc = plan_all_pairs<T>(s, &p)) return rc;
  int rc = NBODY_OK;
  const bool found = with_k1_instance(p, [&](auto r, auto js) {
    constexpr int R = decltype(r)::value, JS = decltype(js)::value;
    rc = p.scalar ? launch_all_pairs_sgpr<T, D, R, JS>(s, p, st) : (JS <= kWaves ? launch_all_pairs<T, D, R, (JS <= kWaves ? JS : 1)>(s, st) : int(NBODY_ERR_ARG));
  });
  if (!found) {
    set_error("all-pairs: unsupported config split=%d targets_per_thread=%d", p.js, p.r);
    return NBODY_ERR_ARG;
  }
  return rc;
}

template <typename T, int D>
static int all_pairs_describe(const nbody_state* s, char* out, size_t len) {
  k1_plan p;
  if (int rc = plan_all_pairs<T>(s, &p)) return rc;
  const char* t = sizeof(T) == 8 ? "double" : "float";
  const char* pair = sizeof(T) == 4 ? (s->sz >= kFarMinBodies ? "rsq+rcp[m y^3 at r2 >= 4 if sparse]" : "rsq+rcp")
                                    : s->sz >= kFarMinBodies ? "far3[-eps if sparse]/near3" : "far3/near3";
  if (p.scalar) {
    bool collect = false;
    with_k1_instance(p, [&](auto r, auto js) {
      constexpr int TB = (kSgprWaves<decltype(js)::value> / decltype(js)::value) * 64 * decltype(r)::value;
      collect = k1_collect((s->count + TB - 1) / TB, p.chunks);
    });
    snprintf(out, len, "all_pairs_force_sgpr_kernel<%s,%d,R=%d,JS=%d> tile=%d chunks=%u%s pair=%s", t, D, p.r, p.js, kTileJ, p.chunks,
             p.chunks > 1 ? (collect ? "(summed in chunk order by the last to arrive)" : "(summed in turn into a)") : "", pair);
  }
  else
    snprintf(out, len, "all_pairs_force_kernel<%s,%d,R=%d,JS=%d> tile=%d chunks=1 pair=%s", t, D, p.r, p.js, kTileJ, pair);
  return NBODY_OK;
}

// Everything the K1 launch for this view needs from the stream's scratch slot, reserved ahead (nbody_create,
// nbody_ctx_set_shard, nbody_ctx_configure_all_pairs) so that a recorded step never allocates: the packed records, the extent
// keys and the turn words (one per target group: 32 KB at N = 2^20; this slot held 403 MB of chunk sums until round 3).
int ap_scratch_reserve(hipStream_t st, const nbody_state* s) {
  void* q             = nullptr;
  const size_t tsz    = s->dtype == NBODY_F32 ? 4 : 8;
  const size_t padded = (size_t(s->sz) + 2 * kTileJ - 1) / (2 * kTileJ) * (2 * kTileJ);  // K1 needs whole tiles, the streamed K2 pairs of them
  if (int r = ap_scratch_get(st, 0, 4 * tsz * padded, &q)) return r;
  if (int r = ap_scratch_get(st, 3, ap_rule_bytes(s->sz, s->dim), &q)) return r;
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using T = typename decltype(tg)::type;
    k1_plan p;
    if (int rc = plan_all_pairs<T>(s, &p)) return rc;
    if (!p.scalar || p.chunks <= 1) return int(NBODY_OK);
    size_t words = 0;
    with_k1_instance(p, [&](auto r, auto js) { words = sgpr_turn_words<decltype(r)::value, decltype(js)::value>(s->count); });
    if (!words) return int(NBODY_OK);
    if (int rc = ap_scratch_get(st, 1, sizeof(uint32_t) * words, &q)) return rc;
    size_t sums = 0;
    with_k1_instance(p, [&](auto r, auto js) {
      constexpr int RR = decltype(r)::value, JJ = decltype(js)::value;
      const uint32_t blocks = (s->count + (kSgprWaves<JJ> / JJ) * 64 * RR - 1) / ((kSgprWaves<JJ> / JJ) * 64 * RR);
      if (k1_collect(blocks, p.chunks)) sums = k1_collect_bytes<T, decltype(tg)::dim, RR, JJ>(blocks, p.chunks);
    });
    if (sums)
      if (int rc = ap_scratch_get(st, 5, sums, &q)) return rc;
    return ap_scratch_get(st, 4, sizeof(k1_status), &q);
  });
}

// The hand-off status of a stream's K1 launches (see all_pairs_force_sgpr_kernel).  Waits for the stream.  out (may be NULL):
// {err, block, group, chunk, polls, waits}.  Returns NBODY_ERR_STATE (and the message) while the sticky error word is set.
void ap_status_mark(hipStream_t st) {
  const int dev = stream_device(st);
  std::lock_guard<std::mutex> lock(g_packed_mu);
  for (auto& sl : g_packed_slots)
    if (sl.stream == st && sl.device == dev) sl.k1_dirty = true;
}

int ap_status_read(hipStream_t st, unsigned long long out[6], bool clear) {
  const int dev = stream_device(st);
  k1_status* dptr = nullptr;
  {
    // the 32-byte copy is made only when it can say something new: an explicit query, a turn-passing K1 or a recorded step queued
    // since the last read, or an error already seen (a tree run in --csv-detailed mode waits for the stream six times per step)
    std::lock_guard<std::mutex> lock(g_packed_mu);
    for (auto& sl : g_packed_slots)
      if (sl.stream == st && sl.device == dev && (out != nullptr || sl.k1_dirty || sl.k1_failed)) dptr = static_cast<k1_status*>(sl.buf[4].ptr);
  }
  k1_status h{};
  if (dptr) NB_HIP(hipMemcpyAsync(&h, dptr, sizeof h, hipMemcpyDeviceToHost, st));
  NB_HIP(hipStreamSynchronize(st));
  if (dptr && clear && (h.err || h.polls || h.waits)) {
    NB_HIP(hipMemsetAsync(dptr, 0, sizeof h, st));
    NB_HIP(hipStreamSynchronize(st));
  }
  if (out) {
    out[0] = h.err, out[1] = h.block, out[2] = h.group, out[3] = h.chunk;
    out[4] = h.polls, out[5] = h.waits;
  }
  if (dptr) {
    std::lock_guard<std::mutex> lock(g_packed_mu);
    for (auto& sl : g_packed_slots)
      if (sl.stream == st && sl.device == dev) sl.k1_dirty = false, sl.k1_failed = h.err != 0 && !clear;
  }
  if (h.err) {
    set_error("all-pairs: the chunk hand-off of target block %u (group %u) failed at source chunk %u: the block of chunk %u never passed "
              "the turn on within the poll budget (workgroups not started in index order?).  `a` of that launch is not valid (the "
              "group's rows are NaN).  The flag is sticky for this stream until nbody_all_pairs_status(stream, out, 1)",
              h.block, h.group, h.chunk, h.chunk - 1u);
    return NBODY_ERR_STATE;
  }
  return NBODY_OK;
}

// ------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------
// sources per LDS tile: 32 per lane in f32 (32 KB), 16 per lane in f64 (32 KB): the cross-lane reduction and the target
// broadcasts are paid once per (NT targets x tile), so the tile is as long as the LDS budget of 3-4 blocks per CU allows
template <typename T>
constexpr int kColTileJ = sizeof(T) == 4 ? 2048 : 1024;

template <typename T, int D>
__global__ __launch_bounds__(kBlock) void collapsed_reset_kernel(T* __restrict__ a, const T* __restrict__ ao, uint64_t n) {
  // diagonal pairs of the reference (src/all_pairs.h:35-40): a[i] -= ao[i]
  uint64_t e = uint64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e < n) a[e] = a[e] - ao[e];
}

// Wavefront sum over 64 lanes with DPP cross-lane adds (VALU data path; __shfl_xor compiles to ds_bpermute_b32,
// which goes through the LDS crossbar at ~21 cycles each when every SIMD does it).  Quad swaps and row mirrors give
// every lane of a 16-lane row the row sum; row_bcast:15 / row_bcast:31 (GFX9 DPP, present on gfx950) chain the four
// rows so that lane 63 holds the total, which is returned wave-uniform through v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);  // disabled/invalid lanes read 0
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, dpp_i32<CTRL, ROW_MASK>(__builtin_bit_cast(int, v)));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_get(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = dpp_i32<CTRL, ROW_MASK>(int(b)), hi = dpp_i32<CTRL, ROW_MASK>(int(b >> 32));
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

template <typename T>
__device__ __forceinline__ T lane_bcast(T v, int src) {  // src is wave-uniform
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  } else {
    long long b = __builtin_bit_cast(long long, v);
    int lo      = __builtin_amdgcn_readlane(int(b), src);
    int hi      = __builtin_amdgcn_readlane(int(b >> 32), src);
    return __builtin_bit_cast(T, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
  }
}

// Transposed wavefront reduction.  Every lane holds NT partial sums p[0..NT-1] (one per target of a group); on return
// lane l holds, in p[0], the sum over ALL 64 lanes of partial number (l % NT).  One halving stage pairs each lane with a
// partner on the other side of one lane-index bit (row_mirror: l^15, row_half_mirror: l^7, quad_perm: l^2, l^1 — all
// plain DPP row operations on the VALU data path); a lane passes the half of its partials the partner keeps and adds
// what it receives to the half it keeps, so a stage over 2m partials costs m DPP adds + 2m selects and the whole row
// part NT - 1 DPP adds — against 6 DPP adds + a v_readlane PER partial for one wave_sum each (the form this replaces:
// 96 vs 15 cross-lane adds per 16 targets and component).  The remaining factor 64 / NT is summed with full-wave
// exchanges (lane ^ 8 by row_ror:8, lane ^ 16 and ^ 32 through ds_bpermute: 2-3 per group).
template <typename T>
__device__ __forceinline__ T xor_lane_add(T v, int mask) {  // v + v of lane (l ^ mask), mask in {16, 32}
  return v + __shfl_xor(v, mask, 64);
}

template <typename T, int NT>
__device__ __forceinline__ T transpose_reduce(T (&p)[NT], int lane) {
  static_assert(NT == 16 || NT == 8, "groups of 16 (f32) or 8 (f64) targets");
  if constexpr (NT == 16) {
    const bool up = (lane & 8) != 0;  // row_mirror: l <-> 15 - l
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const T send = up ? p[i] : p[i + 8], keep = up ? p[i + 8] : p[i];
      p[i]         = keep + dpp_get<0x140>(send);
    }
  }
  {
    const bool up = (lane & 4) != 0;  // row_half_mirror: l <-> l ^ 7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const T send = up ? p[i] : p[i + 4], keep = up ? p[i + 4] : p[i];
      p[i]         = keep + dpp_get<0x141>(send);
    }
  }
  {
    const bool up = (lane & 2) != 0;  // quad_perm:[2,3,0,1]: l <-> l ^ 2
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T send = up ? p[i] : p[i + 2], keep = up ? p[i + 2] : p[i];
      p[i]         = keep + dpp_get<0x4E>(send);
    }
  }
  {
    const bool up = (lane & 1) != 0;  // quad_perm:[1,0,3,2]: l <-> l ^ 1
    const T send  = up ? p[0] : p[1], keep = up ? p[1] : p[0];
    p[0]          = keep + dpp_get<0xB1>(send);
  }
  T v = p[0];  // sum over the lane's NT-lane group of partial (lane % NT)
  if constexpr (NT == 8) v += dpp_get<0x128>(v);  // row_ror:8: l <-> l ^ 8
  v = xor_lane_add(v, 16);
  v = xor_lane_add(v, 32);
  return v;
}

// K2's float pair (3D), scheduled by hand (round 6).  hipcc emits the pair as ONE dependent chain — sub, fma, fma, fma, rsq,
// s_nop, mul, fma, rcp, s_nop, mul, fma x 3: every instruction waits for the one before it and each transcendental is followed
// by a wait state (239 s_nop per 128 pairs) — and its own interleavings of two or four chains put v_rsq_f32 and v_rcp_f32 back
// to back and lost (28.7 against 24.05 ms at config 3, rounds 2 and 5).  Here pair p's head (differences, r^2, rsq) is
// interleaved with pair p - 1's tail (r^2 y, + eps, rcp, x m, three accumulations): the same fourteen instructions, the same
// operations on the same operands in the same order per accumulator (bitwise what pair_math<float>::weight gives), no wait
// state, every transcendental at least three instructions ahead of its consumer and seven behind the other one.
// The carried state (the last pair's differences, r^2 and rsq) lives in VGPRs between the blocks.
struct k2_carry {
  float dx, dy, dz, r2, y;
};
__device__ __forceinline__ k2_carry k2_pair_head(const src_rec<float, 3>& s, const float (&xg)[3]) {
  k2_carry n;
  asm volatile(
      "v_subrev_f32 %[dx], %[X], %[sx]\n\t"
      "v_subrev_f32 %[dy], %[Y], %[sy]\n\t"
      "v_subrev_f32 %[dz], %[Z], %[sz]\n\t"
      "v_fmaak_f32 %[r2], %[dx], %[dx], 0x2081cea\n\t"   // pair_math<float>::tiny
      "v_fmac_f32 %[r2], %[dy], %[dy]\n\t"
      "v_fmac_f32 %[r2], %[dz], %[dz]\n\t"
      "v_rsq_f32 %[y], %[r2]\n\t"
      "s_nop 0"   // the consumer may be the very next instruction (the tail of a block of one pair)
      : [dx] "=&v"(n.dx), [dy] "=&v"(n.dy), [dz] "=&v"(n.dz), [r2] "=&v"(n.r2), [y] "=&v"(n.y)
      : [sx] "v"(s.p[0]), [sy] "v"(s.p[1]), [sz] "v"(s.p[2]), [X] "s"(xg[0]), [Y] "s"(xg[1]), [Z] "s"(xg[2]));
  return n;
}
// head of the pair (s, xg) + tail of the carried pair: its weight r2 -> m / (r2 * (r2 * y) + eps), accumulated into (ax, ay, az)
__device__ __forceinline__ k2_carry k2_pair_step(const src_rec<float, 3>& s, const float (&xg)[3], float mp, const k2_carry& c,
                                                 float& ax, float& ay, float& az) {
  k2_carry n;
  float t;
  asm volatile(
      "v_mul_f32 %[t], %[r2p], %[yp]\n\t"
      "v_subrev_f32 %[dx], %[X], %[sx]\n\t"
      "v_fmaak_f32 %[t], %[r2p], %[t], 0x34000000\n\t"   // FLT_EPSILON
      "v_subrev_f32 %[dy], %[Y], %[sy]\n\t"
      "v_subrev_f32 %[dz], %[Z], %[sz]\n\t"
      "v_rcp_f32 %[t], %[t]\n\t"
      "v_fmaak_f32 %[r2], %[dx], %[dx], 0x2081cea\n\t"
      "v_fmac_f32 %[r2], %[dy], %[dy]\n\t"
      "v_fmac_f32 %[r2], %[dz], %[dz]\n\t"
      "v_mul_f32 %[t], %[mp], %[t]\n\t"
      "v_rsq_f32 %[y], %[r2]\n\t"
      "v_fmac_f32 %[ax], %[t], %[dxp]\n\t"
      "v_fmac_f32 %[ay], %[t], %[dyp]\n\t"
      "v_fmac_f32 %[az], %[t], %[dzp]"
      : [dx] "=&v"(n.dx), [dy] "=&v"(n.dy), [dz] "=&v"(n.dz), [r2] "=&v"(n.r2), [y] "=&v"(n.y), [t] "=&v"(t), [ax] "+v"(ax),
        [ay] "+v"(ay), [az] "+v"(az)
      : [sx] "v"(s.p[0]), [sy] "v"(s.p[1]), [sz] "v"(s.p[2]), [X] "s"(xg[0]), [Y] "s"(xg[1]), [Z] "s"(xg[2]), [mp] "v"(mp),
        [r2p] "v"(c.r2), [yp] "v"(c.y), [dxp] "v"(c.dx), [dyp] "v"(c.dy), [dzp] "v"(c.dz));
  return n;
}
__device__ __forceinline__ void k2_pair_tail(float mp, const k2_carry& c, float& ax, float& ay, float& az) {
  float t;
  asm volatile(
      "v_mul_f32 %[t], %[r2p], %[yp]\n\t"
      "v_fmaak_f32 %[t], %[r2p], %[t], 0x34000000\n\t"
      "v_rcp_f32 %[t], %[t]\n\t"
      "s_nop 0\n\t"
      "v_mul_f32 %[t], %[mp], %[t]\n\t"
      "v_fmac_f32 %[ax], %[t], %[dxp]\n\t"
      "v_fmac_f32 %[ay], %[t], %[dyp]\n\t"
      "v_fmac_f32 %[az], %[t], %[dzp]"
      : [t] "=&v"(t), [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az)
      : [mp] "v"(mp), [r2p] "v"(c.r2), [yp] "v"(c.y), [dxp] "v"(c.dx), [dyp] "v"(c.dy), [dzp] "v"(c.dz));
}

// NT: targets reduced together (NT * D partial sums live per lane); KS: source records per lane held in registers at a
// time; NB: independent pair chains in flight
// (NB = 0: float 3D only — the pair scheduled by hand, k2_pair_step above; the form that ships for config 3)
template <typename T, int D, int NT, int KS, int NB>
__global__ __launch_bounds__(kBlock) void all_pairs_collapsed_kernel(const T* __restrict__ m, const T* __restrict__ x,
                                                                     T* __restrict__ a, T c, uint32_t sz,
                                                                     uint32_t tiles_per_block) {
  using rec_t = src_rec<T, D>;
  constexpr int TJ = kColTileJ<T>;
  constexpr int KJ = TJ / 64;
  __shared__ rec_t tile[TJ];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const pair_consts<T> pc;

  // 64 targets per wave: lane l owns target i0 + l
  const uint32_t i0   = (blockIdx.x * kWaves + wave) * 64;
  const uint32_t imy  = i0 + lane;
  const bool ivalid   = imy < sz;
  T xt[D], mine[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    xt[k]   = x[uint64_t(ivalid ? imy : 0u) * D + k];  // lanes past the end hold body 0's position: computed, never stored
    mine[k] = T(0);
  }
  const int ngroups = (i0 < sz) ? int((min(64u, sz - i0) + NT - 1) / NT) : 0;  // wave-uniform

  const uint32_t ntiles = (sz + TJ - 1) / TJ;
  const uint32_t t0     = blockIdx.y * tiles_per_block;
  const uint32_t t1     = min(ntiles, t0 + tiles_per_block);

  for (uint32_t t = t0; t < t1; ++t) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TJ / kBlock; ++q) {
      uint32_t slot = q * kBlock + threadIdx.x;
      uint64_t j    = uint64_t(t) * TJ + slot;
      rec_t r;
      if (j < sz) {
#pragma unroll
        for (int k = 0; k < D; ++k) r.p[k] = x[j * D + k];
        r.m = m[j];
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) r.p[k] = T(0);
        r.m = T(0);
      }
      if (D == 2) r.p[2] = T(0);
      tile[slot] = r;
    }
    __syncthreads();

    // Per group of NT targets: the lane's KJ sources of the tile pass through registers in SUBS batches of KS (re-read from
    // LDS per group: 64 VGPRs of sources at once left no room for independent pair chains), the NT * D partial sums
    // stay in registers over the whole tile and are reduced across the wavefront once per (group, tile).
    constexpr int SUBS = KJ / KS;
    for (int g = 0; g < ngroups; ++g) {
      T xg[NT][D];  // wave-uniform: the group's target positions, broadcast into SGPRs once per tile
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int k = 0; k < D; ++k) xg[tt][k] = lane_bcast(xt[k], g * NT + tt);
      T part[D][NT];
      // f64: first with the reciprocal-free weight and no near-pair handling at all; a (group, tile) block that held a pair
      // closer than 2^-8 — the tile with the group's own bodies, a very close pair — is recomputed with the guarded form
      bool guarded = sizeof(T) == 4;
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) part[k][tt] = T(0);
        uint32_t lowest = 0xffffffffu;
#pragma unroll 1
        for (int sub = 0; sub < SUBS; ++sub) {
          rec_t src[KS];
#pragma unroll
          for (int q = 0; q < KS; ++q) src[q] = tile[(sub * KS + q) * 64 + lane];
          if constexpr (NB == 0) {
            static_assert(NB != 0 || (sizeof(T) == 4 && D == 3), "the hand-scheduled pair is float 3D");
            // the NT * KS pairs of this batch as one software pipeline: pair p = (target p / KS, source p % KS)
            k2_carry cy = k2_pair_head(src[0], xg[0]);
#pragma unroll
            for (int p = 1; p < NT * KS; ++p) {
              const int tp = (p - 1) / KS;
              cy = k2_pair_step(src[p % KS], xg[p / KS], src[(p - 1) % KS].m, cy, part[0][tp], part[1][tp], part[2][tp]);
            }
            k2_pair_tail(src[KS - 1].m, cy, part[0][NT - 1], part[1][NT - 1], part[2][NT - 1]);
          } else
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            T pt[D];
#pragma unroll
            for (int k = 0; k < D; ++k) pt[k] = part[k][tt];
            if constexpr (sizeof(T) == 8) {
              if (guarded) {
#pragma unroll
                for (int q = 0; q < KS; q += NB) pair_accumulate_multi<T, D, NB>(pt, xg[tt], &src[q]);
              } else {
#pragma unroll
                for (int q = 0; q < KS; q += NB) pair_accumulate_far<D, NB>(pt, xg[tt], &src[q], lowest, pc.k15, pc.k1875);
              }
            } else {
#pragma unroll
              for (int q = 0; q < KS; q += NB) pair_accumulate_multi<T, D, NB>(pt, xg[tt], &src[q]);
            }
#pragma unroll
            for (int k = 0; k < D; ++k) part[k][tt] = pt[k];
          }
        }
        if (guarded || __builtin_amdgcn_ballot_w64(lowest < pair_math<double>::near_hi) == 0ull) break;
        guarded = true;  // wave-uniform
      }
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const T tot = transpose_reduce<T, NT>(part[k], lane);  // lane l: target g*NT + l % NT, summed over the wave
        mine[k] += (lane / NT == g) ? tot : T(0);
      }
    }
  }

  if (ivalid && t0 < t1) {
#pragma unroll
    for (int k = 0; k < D; ++k) atomicAdd(&a[uint64_t(imy) * D + k], c * mine[k]);
  }
}

// ---- K2, float 3D: the streamed form (round 6; what ships for config 3) ---------------------------------------------------
// The tile form above pays per (group of 16 targets, tile of 2048 sources) for 48 v_readlane (the targets into SGPRs), 48 zeroed
// partial sums and a transposed reduction (~150 instructions), i.e. 0.5 instructions per pair on top of the pair's fourteen,
// plus the staging of every tile through LDS behind two block barriers.  Here a WAVE owns one group of 16 targets for its whole
// source chunk: the targets go into SGPRs once, the 48 partial sums live in registers until the chunk is through, and the wave
// reduces and adds atomically once.  Sources are the packed (x, y, z, m) records (one coalesced global_load_dwordx4 per 64 sources
// and 16 pairs per lane, straight from L2 — the 4 MB of config 3 fit every XCD's L2 —: no LDS, no barrier), two batches of KS
// records per lane in registers, the next one in flight while this one is consumed; the pairs of the whole chunk form ONE
// software pipeline of k2_pair_step blocks (head of pair p with the tail of pair p - 1, above).  Lanes run along the sources and
// the wavefront reduction stays what north_star asks of the collapsed variant.
template <typename T, int D>
__global__ __launch_bounds__(kBlock) void collapsed_reset_pack_kernel(T* __restrict__ a, const T* __restrict__ ao, uint64_t nelem,
                                                                      const T* __restrict__ m, const T* __restrict__ x,
                                                                      src_rec<T, D>* __restrict__ packed, uint32_t sz, uint32_t padded) {
  const uint64_t e = uint64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e < nelem) a[e] = a[e] - ao[e];  // diagonal pairs of the reference (src/all_pairs.h:35-40): a[i] -= ao[i]
  if (e < padded) {                    // records past the end carry mass 0 at the origin: their pairs add exactly 0
    src_rec<T, D> r;
#pragma unroll
    for (int k = 0; k < 3; ++k) r.p[k] = (k < D && e < sz) ? x[e * D + k] : T(0);
    r.m       = e < sz ? m[e] : T(0);
    packed[e] = r;
  }
}

constexpr int kK2Group = 16;  // targets per wave of the streamed form
template <int KS>
__global__ __launch_bounds__(kBlock) void all_pairs_collapsed_stream_kernel(const src_rec<float, 3>* __restrict__ packed,
                                                                            const float* __restrict__ x, float* __restrict__ a, float c,
                                                                            uint32_t sz, uint32_t batches_per_chunk, uint32_t nbatches) {
  using rec_t       = src_rec<float, 3>;
  constexpr int NT  = kK2Group;
  const int lane    = threadIdx.x & 63;
  const uint32_t g  = (blockIdx.x * kWaves + (threadIdx.x >> 6));  // this wave's group of 16 targets
  const uint32_t i0 = g * NT;
  if (i0 >= sz) return;  // wave-uniform (no barrier in this kernel)
  // the group's positions: lane l < 16 loads target i0 + l (body 0's for targets past the end: computed, never stored), then SGPRs
  float xt[3];
  {
    const uint32_t i = (lane < NT && i0 + lane < sz) ? i0 + lane : 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) xt[k] = x[uint64_t(i) * 3 + k];
  }
  float xg[NT][3];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int k = 0; k < 3; ++k) xg[tt][k] = lane_bcast(xt[k], tt);
  const uint32_t b0 = blockIdx.y * batches_per_chunk, b1 = min(nbatches, b0 + batches_per_chunk);
  if (b0 >= b1) return;
  float part[3][NT];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) part[k][tt] = 0.0f;
  // batch b = records [b * 64 * KS, (b + 1) * 64 * KS): lane l holds records b * 64 * KS + q * 64 + l, q < KS
  auto fetch = [&](rec_t (&dst)[KS], uint32_t b) {
    const rec_t* base = packed + (uint64_t(b) * (64 * KS) + lane);
#pragma unroll
    for (int q = 0; q < KS; ++q) dst[q] = base[q * 64];
  };
  rec_t cur[KS], nxt[KS];
  fetch(cur, b0);
  k2_carry cy = k2_pair_head(cur[0], xg[0]);
  float mp    = cur[0].m;
  // one batch against the 16 targets; its first pair's head has been issued already (by the prologue or by the batch before).
  // (Macros, not lambdas over array references: a pointer to a buffer makes hipcc keep both buffers in scratch memory.)
#define K2S_CONSUME(SRC)                                                                                       \
  _Pragma("unroll") for (int p = 1; p < NT * KS; ++p) {                                                        \
    const int tp = (p - 1) / KS;                                                                               \
    cy           = k2_pair_step(SRC[p % KS], xg[p / KS], mp, cy, part[0][tp], part[1][tp], part[2][tp]);       \
    mp           = SRC[p % KS].m;                                                                              \
  }
  // the last pair's tail rides on the head of the next batch's first pair
#define K2S_BRIDGE(NEXT)                                                                                       \
  cy = k2_pair_step(NEXT[0], xg[0], mp, cy, part[0][NT - 1], part[1][NT - 1], part[2][NT - 1]);                \
  mp = NEXT[0].m;
  // two batches per trip so that the two register buffers swap without copies; a chunk is an even number of batches (the host
  // pads the record array with zero-mass records to that), and the last trip fetches its own second batch again instead of one
  // past the chunk: the head that rides on the last bridge belongs to no pair and is dropped with the wave
  for (uint32_t b = b0; b < b1; b += 2) {
    fetch(nxt, b + 1);
    K2S_CONSUME(cur)
    K2S_BRIDGE(nxt)
    fetch(cur, b + 2 < b1 ? b + 2 : b + 1);
    K2S_CONSUME(nxt)
    K2S_BRIDGE(cur)
  }
#undef K2S_CONSUME
#undef K2S_BRIDGE
  float tot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) tot[k] = transpose_reduce<float, NT>(part[k], lane);  // lane l: target i0 + l % 16, summed over the wave
  if (lane < NT && i0 + lane < sz) {
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(&a[uint64_t(i0 + lane) * 3 + k], c * tot[k]);
  }
}

// The same shape with the pair left to the compiler (K1's pair_batch: one target from SGPRs against U per-lane sources, the dense
// rule — K2 has no rule pre-pass): float 2D, and float 3D as the A/B of the hand-scheduled pair.
template <typename T, int D, int KS>
__global__ __launch_bounds__(kBlock) void all_pairs_collapsed_stream_cxx_kernel(const src_rec<T, D>* __restrict__ packed,
                                                                                const T* __restrict__ x, T* __restrict__ a, T c, uint32_t sz,
                                                                                uint32_t batches_per_chunk, uint32_t nbatches) {
  using rec_t      = src_rec<T, D>;
  constexpr int NT = sizeof(T) == 4 ? 16 : 8;
  constexpr int U  = 64 / int(sizeof(rec_t));  // records per pair_batch call: 4 in float, 2 in double
  static_assert(KS % U == 0, "whole pair batches");
  static_assert(sizeof(T) == 4, "float only: see collapsed_dispatch");
  const int lane    = threadIdx.x & 63;
  const uint32_t g  = (blockIdx.x * kWaves + (threadIdx.x >> 6));
  const uint32_t i0 = g * NT;
  if (i0 >= sz) return;  // wave-uniform (no barrier in this kernel)
  T xt[D];
  {
    const uint32_t i = (lane < NT && i0 + lane < sz) ? i0 + lane : 0u;
#pragma unroll
    for (int k = 0; k < D; ++k) xt[k] = x[uint64_t(i) * D + k];
  }
  T xg[NT][D];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int k = 0; k < D; ++k) xg[tt][k] = lane_bcast(xt[k], tt);
  const uint32_t b0 = blockIdx.y * batches_per_chunk, b1 = min(nbatches, b0 + batches_per_chunk);
  if (b0 >= b1) return;
  const pair_consts<T> pc;
  T part[D][NT];
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) part[k][tt] = T(0);
  rec_t cur[KS], nxt[KS];
#define K2S_FETCH(DST, B)                                                       \
  {                                                                             \
    const rec_t* base = packed + (uint64_t(B) * (64 * KS) + lane);              \
    _Pragma("unroll") for (int q = 0; q < KS; ++q) DST[q] = base[q * 64];       \
  }
#define K2S_CONSUME(SRC)                                                                              \
  _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) {                                                 \
    T acc1[1][D], xi1[1][D];                                                                          \
    _Pragma("unroll") for (int k = 0; k < D; ++k) acc1[0][k] = part[k][tt], xi1[0][k] = xg[tt][k];    \
    _Pragma("unroll") for (int q = 0; q < KS; q += U) {                                               \
      rec_t su[U];                                                                                    \
      _Pragma("unroll") for (int u = 0; u < U; ++u) su[u] = SRC[q + u];                               \
      pair_batch<T, D, 1, U, false>(acc1, xi1, su, pc);                                               \
    }                                                                                                 \
    _Pragma("unroll") for (int k = 0; k < D; ++k) part[k][tt] = acc1[0][k];                           \
  }
  K2S_FETCH(cur, b0)
  for (uint32_t b = b0; b < b1; b += 2) {  // an even number of batches per chunk (see the float 3D kernel)
    K2S_FETCH(nxt, b + 1)
    K2S_CONSUME(cur)
    K2S_FETCH(cur, (b + 2 < b1 ? b + 2 : b + 1))
    K2S_CONSUME(nxt)
  }
#undef K2S_FETCH
#undef K2S_CONSUME
  T tot[D];
#pragma unroll
  for (int k = 0; k < D; ++k) tot[k] = transpose_reduce<T, NT>(part[k], lane);  // lane l: target i0 + l % NT, summed over the wave
  if (lane < NT && i0 + lane < sz) {
#pragma unroll
    for (int k = 0; k < D; ++k) atomicAdd(&a[uint64_t(i0 + lane) * D + k], c * tot[k]);
  }
}

template <typename T, int D>
static int collapsed_stream_launch(const nbody_state* s, hipStream_t st, int form, uint32_t want_chunks) {
  // form: 20 = float 3D with the hand-scheduled pair, 8 records per lane and batch (21: 4); 22 / 23 / 24 = the compiler-scheduled
  // kernel with 8 / 4 / 2 records per lane and batch
  using rec_t = src_rec<T, D>;
  constexpr bool kHand = sizeof(T) == 4 && D == 3;
  constexpr int NT     = sizeof(T) == 4 ? 16 : 8;
  constexpr int U      = 64 / int(sizeof(rec_t));
  int ks = form == 20 || form == 22 ? 8 : (form == 24 ? 2 : 4);
  if (ks < U) ks = U;
  const uint32_t pair_of_batches = 2u * 64u * uint32_t(ks);  // at most 1024 records: what ap_scratch_reserve pads a context's buffer to
  const uint32_t padded = (s->sz + pair_of_batches - 1) / pair_of_batches * pair_of_batches;
  void* q = nullptr;
  if (int r = ap_scratch_get(st, 0, sizeof(rec_t) * size_t(padded), &q)) return r;
  rec_t* packed        = static_cast<rec_t*>(q);
  const uint64_t nelem = uint64_t(s->sz) * D;
  const uint64_t work  = nelem > padded ? nelem : padded;
  hipLaunchKernelGGL((collapsed_reset_pack_kernel<T, D>), dim3((work + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                     static_cast<T*>(s->a), static_cast<const T*>(s->ao), nelem, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->x), packed, s->sz, padded);
  NB_HIP(hipGetLastError());
  const uint32_t groups   = (s->sz + NT - 1) / NT;
  const uint32_t blocks   = (groups + kWaves - 1) / kWaves;
  const uint32_t nbatches = padded / uint32_t(64 * ks);
  // chunks: enough waves for >= 16 rounds of the chip's 4096 wave slots where the system allows, whole batches per chunk
  uint32_t chunks = want_chunks ? want_chunks : (16u * 4096u + groups - 1) / groups;
  chunks          = std::max(1u, std::min({chunks, nbatches, 65535u}));
  uint32_t bpc = (nbatches + chunks - 1) / chunks;
  bpc += bpc & 1u;  // an even number of batches per chunk (nbatches is even)
  chunks = (nbatches + bpc - 1) / bpc;
  const dim3 grid(blocks, chunks), block(kBlock);
  const T* x = static_cast<const T*>(s->x);
  T* a       = static_cast<T*>(s->a);
  const T c  = static_cast<T>(s->c);
  if constexpr (kHand) {
    if (form == 20) hipLaunchKernelGGL((all_pairs_collapsed_stream_kernel<8>), grid, block, 0, st, packed, x, a, c, s->sz, bpc, nbatches);
    if (form == 21) hipLaunchKernelGGL((all_pairs_collapsed_stream_kernel<4>), grid, block, 0, st, packed, x, a, c, s->sz, bpc, nbatches);
  }
  if (form >= 22 || !kHand) {
    if (ks == 8) hipLaunchKernelGGL((all_pairs_collapsed_stream_cxx_kernel<T, D, 8>), grid, block, 0, st, packed, x, a, c, s->sz, bpc, nbatches);
    else if (ks == 4) hipLaunchKernelGGL((all_pairs_collapsed_stream_cxx_kernel<T, D, 4>), grid, block, 0, st, packed, x, a, c, s->sz, bpc, nbatches);
    else hipLaunchKernelGGL((all_pairs_collapsed_stream_cxx_kernel<T, D, (U <= 2 ? 2 : 4)>), grid, block, 0, st, packed, x, a, c, s->sz, bpc, nbatches);
  }
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

template <typename T, int D>
static int collapsed_dispatch(const nbody_state* s, hipStream_t st) {
  NB_ARG(s->first == 0 && s->count == s->sz, "all-pairs-collapsed is single-GPU: needs first=0, count=sz");
  if (s->sz == 0) return NBODY_OK;
  {
    // float: the streamed form — 3D (config 3) with the hand-scheduled pair, 2D with the compiled one, 4 records per lane and batch
    // (N = 10^4 2D: 0.043 ms against the tile form's 0.060, profiles/r06/k2_streamed_form_2.txt); double: the tile form below.
    // Experiments select the others (tools/time_collapsed.py).
    int form        = sizeof(T) == 4 ? (D == 3 ? 20 : 23) : -1;
    uint32_t chunks = 0;
    if (const char* e = experiment_env("NBODY_K2_CFG")) form = atoi(e);  // -DNBODY_EXPERIMENTS builds only
    if (const char* e = experiment_env("NBODY_K2_CHUNKS")) chunks = uint32_t(atoi(e));
    // (float only: in double the per-batch near-pair branch of pair_batch under the full unroll over targets x sources makes hipcc keep
    // every chain's temporaries alive — 264 ... 512 VGPRs and scratch for <double, 3, 2 ... 8> —; double keeps the tile form below)
    if constexpr (sizeof(T) == 4) {
      if (form >= 22 && form <= 24) return collapsed_stream_launch<T, D>(s, st, form, chunks);
      if constexpr (D == 3)
        if (form == 20 || form == 21) return collapsed_stream_launch<T, D>(s, st, form, chunks);
    }
  }
  uint64_t nelem = uint64_t(s->sz) * D;
  hipLaunchKernelGGL((collapsed_reset_kernel<T, D>), dim3((nelem + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                     static_cast<T*>(s->a), static_cast<const T*>(s->ao), nelem);
  NB_HIP(hipGetLastError());
  uint32_t iblocks = (s->sz + kWaves * 64 - 1) / (kWaves * 64);
  uint32_t ntiles  = (s->sz + kColTileJ<T> - 1) / kColTileJ<T>;
  // aim for >= 4096 blocks (16 per CU) while keeping >= 1 tile per block
  uint32_t ysplit = (4096 + iblocks - 1) / iblocks;
  if (ysplit > ntiles) ysplit = ntiles;
  if (ysplit < 1) ysplit = 1;
  if (ysplit > 65535) ysplit = 65535;
  uint32_t tpb = (ntiles + ysplit - 1) / ysplit;
  ysplit       = (ntiles + tpb - 1) / tpb;
  // Measured at config 3 (f32, N = 262 144) and on the reference's matrix size (f64, N = 10^5), gpurun_out/r02/k2_times*.txt:
  // every variant with ONE pair chain per lane in flight runs at 24.05 ms (36.3 % of the FP32 vector peak) whatever NT, KS
  // and the occupancy (3 to 6 waves per SIMD); every variant that interleaves 2 or 4 chains runs at 28.7 ms — the interleaved
  // order puts v_rsq_f32 and v_rcp_f32 back to back.  Round 5, same box, cfg (16, 8, 1) 23.87 ms: the next tile's records prefetched
  // into registers while this one is consumed 23.85 (the staging is not what stalls); the target's eight chains software-pipelined with
  // sched_group_barrier (6 VALU, 1 transcendental, ...; one scheduling region per target: over the whole block of 128 pairs hipcc did
  // not finish in 30 minutes) 27.8 — independent chains side by side are SLOWER than one chain with its two wait states, again.
  // f64 (N = 10^5, ms): with the guarded rsq+rcp weight everywhere (8, 4, 4) was
  // the fastest at 7.47; with the reciprocal-free weight and a guarded second pass only for the blocks that held a near pair
  // (see the kernel) (8, 8, 1) 6.79, (8, 4, 1) 6.95, (8, 4, 2) 7.00, (8, 2, 2) 7.01, (8, 2, 1) 7.26, (16, 8, 1) 8.03, (8, 4, 4) 9.96.
  int cfg = sizeof(T) == 4 ? 0 : 5;
  if (const char* e = experiment_env("NBODY_K2_CFG")) cfg = atoi(e) >= 20 ? cfg : atoi(e);  // -DNBODY_EXPERIMENTS builds only (tools/time_collapsed.py)
#define NB_K2(NT, KS, NBC)                                                                                              \
  hipLaunchKernelGGL((all_pairs_collapsed_kernel<T, D, NT, KS, NBC>), dim3(iblocks, ysplit), dim3(kBlock), 0, st,        \
                     static_cast<const T*>(s->m), static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), \
                     s->sz, tpb)
  if constexpr (sizeof(T) == 4 && D == 3) {  // the pair scheduled by hand (k2_pair_step): what ships for float 3D (config 3)
    switch (cfg) {
      case 0: NB_K2(16, 8, 0); NB_HIP(hipGetLastError()); return NBODY_OK;
      case 10: NB_K2(16, 4, 0); NB_HIP(hipGetLastError()); return NBODY_OK;
      case 11: NB_K2(8, 8, 0); NB_HIP(hipGetLastError()); return NBODY_OK;
      case 12: NB_K2(8, 4, 0); NB_HIP(hipGetLastError()); return NBODY_OK;
      case 8: cfg = 0; break;  // experiments: the compiler-scheduled (16, 8, 1), the cross-check of the hand-scheduled form
      default: break;
    }
  }
  switch (cfg) {
    case 0: NB_K2(16, 8, 1); break;
    case 1: NB_K2(8, 4, 4); break;
    case 2: NB_K2(8, 4, 1); break;
    case 4: NB_K2(8, 4, 2); break;
    case 5: NB_K2(8, 8, 1); break;
    case 6: NB_K2(8, 2, 1); break;
    case 7: NB_K2(8, 2, 2); break;
    default: NB_K2(16, 8, 4); break;
  }
#undef NB_K2
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

// ------------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void accelerate_step_kernel(T* __restrict__ x, T* __restrict__ v, const T* __restrict__ a,
                                                                 T* __restrict__ ao, T dt, uint64_t n) {
#pragma clang fp contract(off)
  const T hdt   = T(0.5) * dt;
  const T hdtdt = hdt * dt;
  for (uint64_t e = uint64_t(blockIdx.x) * kBlock + threadIdx.x; e < n; e += uint64_t(gridDim.x) * kBlock) {
    T ae = a[e], aoe = ao[e], ve = v[e];
    T t1 = dt * ve;
    T t2 = hdtdt * aoe;
    x[e] = x[e] + (t1 + t2);
    v[e] = ve + hdt * (ae + aoe);
    ao[e] = ae;
  }
}

template <typename T, int D>
static int accelerate_dispatch(const nbody_state* s, hipStream_t st) {
  uint64_t n = uint64_t(s->count) * D;
  if (n == 0) return NBODY_OK;
  uint64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride the rest
  T* xloc = static_cast<T*>(s->x) + uint64_t(s->first) * D;
  hipLaunchKernelGGL((accelerate_step_kernel<T>), dim3(uint32_t(blocks)), dim3(kBlock), 0, st, xloc, static_cast<T*>(s->v),
                     static_cast<const T*>(s->a), static_cast<T*>(s->ao), static_cast<T>(s->dt), n);
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

}  // namespace nbody

// ---- C ABI ------------------------------------------------------------------------------------------
using namespace nbody;

extern "C" int nbody_all_pairs_configure(int split, int targets_per_thread) {
  const uint32_t cur = g_ap_default.load(std::memory_order_relaxed);
  if (int r = check_tuning(split, targets_per_thread, int((cur >> 6) & 3u))) return r;
  uint32_t want = (cur & (3u << 6)) | uint32_t(split & 15) | (uint32_t(targets_per_thread & 3) << 4);
  g_ap_default.store(want, std::memory_order_relaxed);
  return NBODY_OK;
}

extern "C" int nbody_all_pairs_source_path(int mode) {
  if (int r = check_tuning(0, 0, mode)) return r;
  uint32_t cur = g_ap_default.load(std::memory_order_relaxed);
  while (!g_ap_default.compare_exchange_weak(cur, (cur & ~(3u << 6)) | (uint32_t(mode) << 6), std::memory_order_relaxed)) {
  }
  return NBODY_OK;
}

extern "C" int nbody_all_pairs_describe(const nbody_state* s, char* out, size_t len) {
  NB_ARG(out != nullptr && len > 0, "out is NULL");
  if (int r = check_state(s)) return r;
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return all_pairs_describe<typename TG::type, TG::dim>(s, out, len);
  });
}

extern "C" int nbody_all_pairs_pair_rule(const nbody_state* s, void* stream, int* sparse_out, double* volume_out) {
  NB_ARG(sparse_out != nullptr, "sparse_out is NULL");
  if (int r = check_state(s)) return r;
  device_guard guard(stream_device(as_stream(stream)));
  *sparse_out = 0;
  if (volume_out) *volume_out = 0.0;
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using T         = typename decltype(tg)::type;
    constexpr int D = decltype(tg)::dim;
    const k1_rule* rule = nullptr;
    if (int r = ap_prepare<T, D>(s, as_stream(stream), false, nullptr, &rule, nullptr, 0)) return r;
    if (rule == nullptr) return int(NBODY_OK);  // below kFarMinBodies: the dense rule by definition
    k1_rule host;
    NB_HIP(hipMemcpyAsync(&host, rule, sizeof host, hipMemcpyDeviceToHost, as_stream(stream)));
    NB_HIP(hipStreamSynchronize(as_stream(stream)));
    *sparse_out = host.sparse ? 1 : 0;
    if (volume_out) *volume_out = host.volume;
    return int(NBODY_OK);
  });
}

extern "C" int nbody_all_pairs_force(const nbody_state* s, void* stream) {
  if (int r = check_state(s)) return r;
  device_guard guard(stream_device(as_stream(stream)));
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return all_pairs_dispatch<typename TG::type, TG::dim>(s, as_stream(stream));
  });
}

extern "C" int nbody_all_pairs_status(void* stream, uint64_t out[6], int clear) {
  device_guard guard(stream_device(as_stream(stream)));
  unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
  const int rc = ap_status_read(as_stream(stream), v, clear != 0);
  if (out)
    for (int i = 0; i < 6; ++i) out[i] = v[i];
  return rc;
}

extern "C" int nbody_all_pairs_collapsed_force(const nbody_state* s, void* stream) {
  if (int r = check_state(s)) return r;
  device_guard guard(stream_device(as_stream(stream)));
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return collapsed_dispatch<typename TG::type, TG::dim>(s, as_stream(stream));
  });
}

extern "C" int nbody_accelerate_step(const nbody_state* s, void* stream) {
  if (int r = check_state(s)) return r;
  device_guard guard(stream_device(as_stream(stream)));
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return accelerate_dispatch<typename TG::type, TG::dim>(s, as_stream(stream));
  });
}
