// K1 all-pairs force, K2 all-pairs-collapsed force, K3 leapfrog step — hand-written for gfx950.
//
// K1 (replaces src/all_pairs.h:14-27).  Bound: FP64/FP32 VALU issue (no MFMA: the inner body is
// sub/FMA/rsq/rcp, not a contraction).  Structure:
//   * one lane per target body (R targets per lane), targets of a block held in VGPRs;
//   * sources are packed (x, m) records visited in tiles of TJ; the source range of every tile is split over
//     the JS waves that share a target group, so a small shard (N/8 bodies on one GPU) still puts >= 4 waves
//     on every SIMD — one wave alone reaches only 75% of the FP64 issue rate
//     (profiles/r01_valu_rates_microbench.txt);
//   * wave partials are combined through LDS in a fixed order, so the result is deterministic and
//     independent of how bodies are sharded over GPUs.
//   Two ways of bringing a source record to the 64 lanes that all need the same one:
//   - LDS tiles (all_pairs_force_kernel): tiles staged in LDS by all 256 lanes with a register prefetch of the
//     next tile; the inner loop reads each record as an LDS broadcast (ds_read_b128) into VGPRs;
//   - scalar stream (all_pairs_force_sgpr_kernel, default): the record is wave-uniform, so it belongs in SGPRs:
//     a pre-pass packs the records once per call (32 B x N), every wave streams its slice with
//     s_load_dwordx16 two batches deep and the VALU instructions take their source operands from SGPRs.
//     No staging loads, no LDS traffic, no barriers in the loop, half the VGPRs (occupancy 8).  Same
//     arithmetic in the same order: bitwise the LDS kernel's result; 3 % faster at every size measured
//     (722 vs 746 ms at N = 2^20 on the same box) on a kernel that runs at the socket power cap.
//
// K2 (replaces src/all_pairs.h:29-50, intended semantics).  Lanes run along the SOURCE axis (one
// ordered pair per lane and step), each wave owns 64 targets whose positions it broadcasts with
// v_readlane; the per-target partial over 64*KJ sources is reduced across the wavefront with DPP
// cross-lane adds (the __shfl family compiles to ds_bpermute here), lane t keeps target t's sum, and
// after the tile each wave issues D coalesced atomic adds.  grid.y splits the source range so small N still fills the chip.
//
// K3 (replaces src/system.h:52-60).  Pure HBM stream (7*D*sizeof(T) bytes/body), flat elementwise
// over count*D scalars, FP contraction off so it is bit-identical to the reference's x86 -O2 build.
#include "common.hpp"

#include <mutex>
#include <vector>

namespace nbody {

constexpr int kBlock = 256;  // 4 waves
constexpr int kWaves = kBlock / 64;
constexpr int kTileJ = 512;  // sources per LDS tile (fixed: the rounding order depends on it)

struct ap_config {
  int split = 0;  // 0 = auto
  int tpt   = 0;  // targets per thread, 0 = auto
  int path  = 0;  // source path: 0 = auto (scalar stream), 1 = LDS tiles, 2 = scalar stream
};
static ap_config g_ap_config;

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int R, int JS>
__global__ __launch_bounds__(kBlock) void all_pairs_force_kernel(const T* __restrict__ m, const T* __restrict__ x,
                                                                 T* __restrict__ a, T c, uint32_t sz, uint32_t first,
                                                                 uint32_t count) {
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
  for (uint32_t t = 0; t < ntiles; ++t) {
    __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
    for (int q = 0; q < LPT; ++q) tile[q * kBlock + threadIdx.x] = stage[q];
    __syncthreads();
    if (t + 1 < ntiles) stage_load(t + 1);  // in flight while this tile is consumed

    const rec_t* src = &tile[jpart * SUB];
#pragma unroll 4
    for (int jj = 0; jj < SUB; ++jj) {
      rec_t s = src[jj];  // wave-uniform address: LDS broadcast
#pragma unroll
      for (int r = 0; r < R; ++r) pair_accumulate<T, D>(acc[r], xi[r], s);
    }
  }

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

typedef uint32_t sgpr16 __attribute__((ext_vector_type(16)));
// `tie` is a VGPR value the surrounding arithmetic reads (sload16) or produces (swait): the statements carry no
// instruction for it, it only pins them in program order relative to that arithmetic (inline asm is otherwise free to
// drift across pure FP code during instruction selection).
template <typename V>
__device__ __forceinline__ sgpr16 sload16(const void* p, V& tie) {  // p wave-uniform, 4-byte aligned
  sgpr16 r;
  asm volatile("s_load_dwordx16 %0, %2, 0x0" : "=s"(r), "+v"(tie) : "s"(p));
  return r;
}
template <typename V>
__device__ __forceinline__ void swait(sgpr16& v, V& tie) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+v"(tie));
}

// Scalar-stream form: pre-pass that packs (x, m) into aligned records, zero-mass padding up to a whole tile
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

template <int JS>
constexpr int kSgprWaves = JS > kWaves ? JS : kWaves;  // waves per block of the scalar-stream form

template <typename T, int D, int R, int JS>
__global__ __launch_bounds__(64 * kSgprWaves<JS>) void all_pairs_force_sgpr_kernel(const src_rec<T, D>* __restrict__ packed,
                                                                                   const T* __restrict__ x, T* __restrict__ a, T c,
                                                                                   uint32_t sz, uint32_t first, uint32_t count) {
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
  const uint32_t ntiles = (sz + kTileJ - 1) / kTileJ;
  const uint32_t nsteps = ntiles * SUB;  // sources this wave visits: its SUB-record slice of every tile, in tile order
  constexpr int U = 64 / int(sizeof(rec_t));  // records per 64-byte batch (2 in f64, 4 in f32); SUB % (2 * U) == 0
  struct batch_t {
    rec_t r[U];
  };
  auto batch = [&](uint32_t k) { return packed + (uint64_t(k / SUB) * kTileJ + uint32_t(jpart) * SUB + (k % SUB)); };
  // Two SGPR buffers, each requested (s_load_dwordx16) one compute phase before it is consumed.  Written with inline
  // asm: hipcc folds a loop-carried load from read-only memory back into a load at the loop top and waits for it there.
  // SMEM returns out of order, so the only usable wait is lgkmcnt(0): wait for X, request Y, consume X.
  sgpr16 A = sload16(batch(0), xi[0][0]), B;
  for (uint32_t k = 0; k < nsteps; k += 2 * U) {
    swait(A, acc[0][0]);
    B = sload16(batch(k + U), xi[0][0]);
    {
      const batch_t ba = __builtin_bit_cast(batch_t, A);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) pair_accumulate<T, D>(acc[r], xi[r], ba.r[u]);
    }
    swait(B, acc[0][0]);
    A = sload16(batch(k + 2 * U < nsteps ? k + 2 * U : k), xi[0][0]);  // the last iteration re-requests its own batch
    {
      const batch_t bb = __builtin_bit_cast(batch_t, B);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) pair_accumulate<T, D>(acc[r], xi[r], bb.r[u]);
    }
  }
  swait(A, acc[0][0]);  // nothing in flight when the wave goes on
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

// Packed-source scratch, one buffer per stream that has called the scalar-stream form (grow-only).  A context
// reserves its buffer when it is created (nbody_create), so that a step recorded with nbody_graph_begin never has to
// allocate; other callers get theirs on the first call, which therefore must not be inside a capture.
namespace {
struct packed_slot {
  hipStream_t stream;
  void* ptr;
  size_t cap;
};
std::mutex g_packed_mu;
std::vector<packed_slot> g_packed_slots;
}  // namespace

int ap_scratch_get(hipStream_t st, size_t bytes, void** out) {
  std::lock_guard<std::mutex> lock(g_packed_mu);
  packed_slot* slot = nullptr;
  for (auto& sl : g_packed_slots)
    if (sl.stream == st) slot = &sl;
  if (slot && slot->cap >= bytes) {
    *out = slot->ptr;
    return NBODY_OK;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (st != nullptr) (void)hipStreamIsCapturing(st, &cs);
  if (cs != hipStreamCaptureStatusNone) {
    set_error("all-pairs: the packed-source buffer of this stream must exist before a step is recorded "
              "(call nbody_all_pairs_force once outside nbody_graph_begin/end, or use a context from nbody_create)");
    return NBODY_ERR_STATE;
  }
  if (!slot) {
    g_packed_slots.push_back({st, nullptr, 0});
    slot = &g_packed_slots.back();
  }
  if (slot->ptr) NB_HIP(hipFree(slot->ptr));  // hipFree waits for the device: no launch still reads the old buffer
  slot->ptr = nullptr;
  slot->cap = 0;
  NB_HIP(hipMalloc(&slot->ptr, bytes));
  slot->cap = bytes;
  *out      = slot->ptr;
  return NBODY_OK;
}

int ap_scratch_reserve(hipStream_t st, int dtype, uint32_t n) {
  void* p             = nullptr;
  const size_t padded = (size_t(n) + kTileJ - 1) / kTileJ * kTileJ;
  return ap_scratch_get(st, (dtype == NBODY_F32 ? 16u : 32u) * padded, &p);
}

void ap_scratch_release(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_packed_mu);
  for (size_t i = 0; i < g_packed_slots.size(); ++i) {
    if (g_packed_slots[i].stream == st) {
      (void)hipFree(g_packed_slots[i].ptr);
      g_packed_slots.erase(g_packed_slots.begin() + long(i));
      return;
    }
  }
}

template <typename T, int D, int R, int JS>
static int launch_all_pairs_sgpr(const nbody_state* s, hipStream_t st) {
  constexpr int TB = (kSgprWaves<JS> / JS) * 64 * R;
  uint32_t blocks  = (s->count + TB - 1) / TB;
  if (blocks == 0) return NBODY_OK;
  const uint32_t padded = (s->sz + kTileJ - 1) / kTileJ * kTileJ;
  void* scratch         = nullptr;
  if (int r = ap_scratch_get(st, sizeof(src_rec<T, D>) * size_t(padded), &scratch)) return r;
  auto* packed = static_cast<src_rec<T, D>*>(scratch);
  hipLaunchKernelGGL((pack_sources_kernel<T, D>), dim3((padded + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                     static_cast<const T*>(s->m), static_cast<const T*>(s->x), packed, s->sz, padded);
  NB_HIP(hipGetLastError());
  hipLaunchKernelGGL((all_pairs_force_sgpr_kernel<T, D, R, JS>), dim3(blocks), dim3(64 * kSgprWaves<JS>), 0, st, packed,
                     static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->sz, s->first, s->count);
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

template <typename T, int D, int R, int JS>
static int launch_all_pairs(const nbody_state* s, hipStream_t st) {
  constexpr int TB = (kWaves / JS) * 64 * R;
  uint32_t blocks  = (s->count + TB - 1) / TB;
  if (blocks == 0) return NBODY_OK;
  hipLaunchKernelGGL((all_pairs_force_kernel<T, D, R, JS>), dim3(blocks), dim3(kBlock), 0, st, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->sz, s->first, s->count);
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

// Source split: chosen from sz ONLY (never from first/count) so that every shard of a multi-GPU run
// sums in the same order as the single-GPU run.  8 slices (512-thread blocks, scalar-stream form only) once the system
// is large enough for that form: twice the waves for the same work, which is what a 1/8 shard of N = 2^20 lacks
// (93.1 vs 97.6 ms; whole system 720 vs 730 ms; N = 65 536: 2.92 vs 3.08 ms).  Small systems keep 4.
static int auto_split(uint32_t sz) { return sz >= 65536u ? 8 : 4; }

template <typename T, int D>
static int all_pairs_dispatch(const nbody_state* s, hipStream_t st) {
  int js = g_ap_config.split ? g_ap_config.split : auto_split(s->sz);
  int r  = g_ap_config.tpt;
  // Source path (bitwise identical results, so the choice may depend on the shard size).  The scalar stream pays an SMEM
  // round trip per 64-byte batch, which needs several waves per SIMD to hide: measured f64, split 4, lds / sgpr:
  // 0.16 / 0.34 ms at N = 10^4, 0.78 / 0.98 ms at 3*10^4, 3.20 / 3.11 ms at 65 536, 8.62 / 8.32 ms at 10^5, 746 / 722 ms at 2^20.
  const uint64_t waves_r1 = (uint64_t(s->count) + 63) / 64 * js;
  const bool scalar       = js == 8 || g_ap_config.path == 2 || (g_ap_config.path == 0 && waves_r1 >= 4096);
  if (js == 8 && g_ap_config.path == 1) {
    set_error("all-pairs: the LDS-tile form has at most 4 source slices; sz = %u uses 8 (nbody_all_pairs_configure(4, ...) to force 4)",
              s->sz);
    return NBODY_ERR_ARG;
  }
  if (r == 0) {
    // R = 2 halves the record traffic per pair; it pays when its blocks still spread evenly over the 256 CUs.
    // Measured (scalar form, f64): N = 65 536 (512 blocks) 2.92 vs 2.95 ms, 10^5 (782 blocks) 9.40 vs 8.10 ms, 262 144
    // (2048) 45.8 vs 46.6 ms, 2^20 720 vs 723 ms, its 1/8 shard (1024) 93.1 vs 96.0 ms; never in f32 (24.8 vs 23.0 ms at
    // 262 144).  LDS form: >= 8 waves per SIMD.
    const uint64_t blocks_r2 = (uint64_t(s->count) + 127) / 128;
    if (scalar) r = (sizeof(T) == 8 && (blocks_r2 >= 2048 || (blocks_r2 >= 512 && blocks_r2 % 256 == 0))) ? 2 : 1;
    else r = blocks_r2 * js >= 8192 ? 2 : 1;
  }
#define NB_CASE(RR, JJ)                                                            \
  if (r == RR && js == JJ)                                                         \
  return scalar ? launch_all_pairs_sgpr<T, D, RR, JJ>(s, st) : launch_all_pairs<T, D, RR, JJ>(s, st)
  NB_CASE(1, 1);
  NB_CASE(1, 2);
  NB_CASE(1, 4);
  NB_CASE(2, 1);
  NB_CASE(2, 2);
  NB_CASE(2, 4);
#undef NB_CASE
  if (scalar && js == 8) {
    if (r == 1) return launch_all_pairs_sgpr<T, D, 1, 8>(s, st);
    if (r == 2) return launch_all_pairs_sgpr<T, D, 2, 8>(s, st);
  }
  set_error("all-pairs: unsupported config split=%d targets_per_thread=%d", js, r);
  return NBODY_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------
// sources per LDS tile: 16 per lane in f32, 8 per lane in f64 (64 VGPRs of source registers either way; 8 per lane in
// f32 was measured slower, 34 vs 27 ms at config 3: the per-target broadcast + reduction is amortised over fewer pairs)
template <typename T>
constexpr int kColTileJ = sizeof(T) == 4 ? 1024 : 512;

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
__device__ __forceinline__ T wave_sum(T v) {
  v += dpp_get<0xB1>(v);        // quad_perm:[1,0,3,2]
  v += dpp_get<0x4E>(v);        // quad_perm:[2,3,0,1]
  v += dpp_get<0x141>(v);       // row_half_mirror
  v += dpp_get<0x140>(v);       // row_mirror       -> every lane of a row holds its row's sum
  v += dpp_get<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_get<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  } else {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane(int(b), 63), hi = __builtin_amdgcn_readlane(int(b >> 32), 63);
    return __builtin_bit_cast(T, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
  }
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

template <typename T, int D>
__global__ __launch_bounds__(kBlock) void all_pairs_collapsed_kernel(const T* __restrict__ m, const T* __restrict__ x,
                                                                     T* __restrict__ a, T c, uint32_t sz,
                                                                     uint32_t tiles_per_block) {
  using rec_t = src_rec<T, D>;
  constexpr int TJ = kColTileJ<T>;
  constexpr int KJ = TJ / 64;
  __shared__ rec_t tile[TJ];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;

  // 64 targets per wave: lane l owns target i0 + l
  const uint32_t i0   = (blockIdx.x * kWaves + wave) * 64;
  const uint32_t imy  = i0 + lane;
  const bool ivalid   = imy < sz;
  T xt[D], mine[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    xt[k]   = x[uint64_t(ivalid ? imy : 0u) * D + k];
    mine[k] = T(0);
  }
  const int ntargets = (i0 < sz) ? int(min(64u, sz - i0)) : 0;  // wave-uniform

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

    // my KJ sources of this tile live in registers for all 64 targets of the wave
    rec_t src[KJ];
#pragma unroll
    for (int q = 0; q < KJ; ++q) src[q] = tile[q * 64 + lane];

    for (int tt = 0; tt < ntargets; ++tt) {
      T xi[D], part[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        xi[k]   = lane_bcast(xt[k], tt);
        part[k] = T(0);
      }
#pragma unroll
      for (int q = 0; q < KJ; ++q) pair_accumulate<T, D>(part, xi, src[q]);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        T tot = wave_sum(part[k]);
        if (lane == tt) mine[k] += tot;
      }
    }
  }

  if (ivalid && t0 < t1) {
#pragma unroll
    for (int k = 0; k < D; ++k) atomicAdd(&a[uint64_t(imy) * D + k], c * mine[k]);
  }
}

template <typename T, int D>
static int collapsed_dispatch(const nbody_state* s, hipStream_t st) {
  NB_ARG(s->first == 0 && s->count == s->sz, "all-pairs-collapsed is single-GPU: needs first=0, count=sz");
  if (s->sz == 0) return NBODY_OK;
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
  hipLaunchKernelGGL((all_pairs_collapsed_kernel<T, D>), dim3(iblocks, ysplit), dim3(kBlock), 0, st,
                     static_cast<const T*>(s->m), static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c),
                     s->sz, tpb);
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
  NB_ARG(split == 0 || split == 1 || split == 2 || split == 4 || split == 8, "split must be 0, 1, 2, 4 or 8 (got %d)", split);
  NB_ARG(targets_per_thread >= 0 && targets_per_thread <= 2, "targets_per_thread must be 0, 1 or 2 (got %d)",
         targets_per_thread);
  g_ap_config.split = split;
  g_ap_config.tpt   = targets_per_thread;
  return NBODY_OK;
}

extern "C" int nbody_all_pairs_source_path(int mode) {
  NB_ARG(mode >= 0 && mode <= 2, "source path must be 0 (auto), 1 (LDS tiles) or 2 (scalar stream), got %d", mode);
  g_ap_config.path = mode;
  return NBODY_OK;
}

extern "C" int nbody_all_pairs_force(const nbody_state* s, void* stream) {
  if (int r = check_state(s)) return r;
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return all_pairs_dispatch<typename TG::type, TG::dim>(s, as_stream(stream));
  });
}

extern "C" int nbody_all_pairs_collapsed_force(const nbody_state* s, void* stream) {
  if (int r = check_state(s)) return r;
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return collapsed_dispatch<typename TG::type, TG::dim>(s, as_stream(stream));
  });
}

extern "C" int nbody_accelerate_step(const nbody_state* s, void* stream) {
  if (int r = check_state(s)) return r;
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return accelerate_dispatch<typename TG::type, TG::dim>(s, as_stream(stream));
  });
}
