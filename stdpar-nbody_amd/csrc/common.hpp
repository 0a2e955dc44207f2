// Shared helpers for the gfx950 N-body backend: error plumbing, (dtype, dim) dispatch and the pair
// kernel math.  Device code here is written for CDNA4 only (wave64, no portability layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/nbody_hip.h"

namespace nbody {

// ---- error plumbing -----------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define NB_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) return ::nbody::hip_fail(e_, #call, __FILE__, __LINE__);     \
  } while (0)

#define NB_ARG(cond, ...)                \
  do {                                   \
    if (!(cond)) {                       \
      ::nbody::set_error(__VA_ARGS__);   \
      return NBODY_ERR_ARG;              \
    }                                    \
  } while (0)

// ---- devices --------------------------------------------------------------------------------------
// One host thread may drive several GPUs (the CLI's --gpus N, ncclCommInitAll): every entry point runs on the device its
// handle or stream belongs to and puts the caller's current device back when it returns.
struct device_guard {
  int prev     = -1;
  bool changed = false;
  explicit device_guard(int dev) {
    if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
  }
  ~device_guard() {
    if (changed) (void)hipSetDevice(prev);
  }
  device_guard(const device_guard&)            = delete;
  device_guard& operator=(const device_guard&) = delete;
};

inline int current_device() {
  int d = -1;
  return hipGetDevice(&d) == hipSuccess ? d : -1;
}

// 0 unless `st` is recording (nbody_graph_begin): then the capture's id
inline unsigned long long capture_id(hipStream_t st) {
  if (st == nullptr) return 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  unsigned long long id     = 0;
  if (hipStreamGetCaptureInfo(st, &cs, &id) != hipSuccess || cs != hipStreamCaptureStatusActive) return 0;
  return id ? id : ~0ull;
}

// device a stream was created on; the NULL stream belongs to whatever device is current
inline int stream_device(hipStream_t st) {
  if (st == nullptr) return current_device();
  hipDevice_t d = -1;
  return hipStreamGetDevice(st, &d) == hipSuccess ? int(d) : current_device();
}

// a handle (context, tree, communicator) is used on the device it was created on: its buffers live there
inline int check_same_device(int handle_device, hipStream_t st, const char* what) {
  const int sd = stream_device(st);
  NB_ARG(handle_device == sd, "%s was created on device %d but the call's stream belongs to device %d", what, handle_device, sd);
  return NBODY_OK;
}

// Experiment switches (tools/README.md) exist only in the build made with -DNBODY_EXPERIMENTS (`make experiments`:
// libnbody_hip_exp.so); the shipped library never reads the environment — the launch shape is what nbody_all_pairs_describe
// states and what the ABI setters select.
#ifdef NBODY_EXPERIMENTS
inline const char* experiment_env(const char* name) { return getenv(name); }
#else
constexpr const char* experiment_env(const char*) { return nullptr; }
#endif

inline int check_tuning(int split, int tpt, int path);

inline int check_state(const nbody_state* s) {
  NB_ARG(s != nullptr, "nbody_state is NULL");
  NB_ARG(s->dtype == NBODY_F32 || s->dtype == NBODY_F64, "bad dtype %d", s->dtype);
  NB_ARG(s->dim == 2 || s->dim == 3, "bad dim %d (must be 2 or 3)", s->dim);
  NB_ARG(s->m && s->x && s->v && s->a && s->ao, "nbody_state has a NULL array pointer");
  NB_ARG(uint64_t(s->first) + uint64_t(s->count) <= uint64_t(s->sz), "shard [%u, %u+%u) exceeds sz=%u", s->first, s->first,
         s->count, s->sz);
  // tuning is 0 or a value NBODY_TUNING() made: a hand-built state that was not zero-initialised is refused, not obeyed
  if (s->tuning != 0) {
    NB_ARG((s->tuning & ~0x1ffu) == 0 && (s->tuning & 0x100u) != 0, "nbody_state.tuning = 0x%x is not 0 or an NBODY_TUNING(...) value",
           s->tuning);
    if (int r = check_tuning(int(s->tuning & 15u), int((s->tuning >> 4) & 3u), int((s->tuning >> 6) & 3u))) return r;
  }
  return NBODY_OK;
}

// Runtime (dtype, dim) -> compile-time <T, D>.  F is a generic lambda taking (T tag, integral_constant D).
template <typename T, int D>
struct tag {
  using type = T;
  static constexpr int dim = D;
};

template <typename F>
inline int dispatch(int dtype, int dim, F&& f) {
  if (dtype == NBODY_F32 && dim == 2) return f(tag<float, 2>{});
  if (dtype == NBODY_F32 && dim == 3) return f(tag<float, 3>{});
  if (dtype == NBODY_F64 && dim == 2) return f(tag<double, 2>{});
  if (dtype == NBODY_F64 && dim == 3) return f(tag<double, 3>{});
  set_error("unsupported (dtype=%d, dim=%d)", dtype, dim);
  return NBODY_ERR_ARG;
}

// ---- pair math -----------------------------------------------------------------------------------
// The reference pair term (src/all_pairs.h:23, src/bvh.h:297,308; dist3 at src/vec.h:249-252) is
//     m_j * (x_j - x_i) / (pow(r2, 3/2) + eps)
// i.e. one pow, one add and D divisions per pair.  On gfx950 FP64 transcendentals run at quarter rate and do not overlap
// the FMA pipe (profiles/r01_valu_rates_microbench.txt), and an IEEE divide is a ~10-instruction sequence, so the kernels
// evaluate the weight   w = m_j / (r2*sqrt(r2) + eps)   once per pair and then D FMAs.  What is shipped (f64):
//   * r2 >= 4        weight_far<false>: w = (m y^3)(1 + e(3/2 + 15/8 e)), y = v_rsq_f64(r2), e = 1 - r2 y^2 — no reciprocal,
//                    and the `+ eps` dropped: eps/r^3 <= 2^-55 there, an eighth of an ulp.   7 ops + 1 transcendental
//   * 2^-16 <= r2 < 4  weight_far<true>: the same with the first-order term of the eps expansion, - eps y^3.   8 ops + 1
//   * r2 < 2^-16     weight<3>(): the guarded reciprocal form (the self pair, coincident and very close bodies), <= 2 ulp.
// Which form a pair takes depends on ITS OWN r2 alone (never on the other lanes of the wave), so a target's sum is bitwise
// independent of the shard window.  Every form is within 2.5 ulp of the exact term with no bias
// (tests/test_gpu_all_pairs.py::test_pair_term_accuracy, test_pair_term_two_body_ulps, test_pair_math_adversarial_separations).
// weight() with NBODY_PAIR_POLISH = 2 is what round 1 shipped everywhere and what the compiler-scheduled tree walks still use.
//
// r2 must be > 0: callers fold TINY into the first FMA of r2 (r2 = fma(dx,dx,TINY)), which makes the
// self term and coincident bodies evaluate to exactly 0 * (finite) = 0 — the value the reference
// gets from `(m*0)/eps` (self term skipped by `if (i == j) continue`) — with no branch
// and no select (a v_cmp + 2 v_cndmask select costs ~5 FMA slots on this chip).
template <typename T>
struct pair_math;

// NBODY_PAIR_POLISH: order of the polish applied to the two 2^-24 FP64 seeds of weight() as the tree walks (K9, octree)
// use it.  K1 uses weight() only for pairs closer than 2^-8 (see weight_far below) and takes the third-order form there.
//   2 (default): Newton steps  s = h + h*e/2,  w = zm + zm*e2        17 full-rate ops + 2 transcendentals per pair.
//      Per-term error bound: +3/8*e^2 (e <= 2^-23.2) - e2^2 (e2 <= 2^-24.4)  =>  within [-2e-15, +4.1e-15] relative.
//   3: third-order steps (+2 FMA): <= 2 ulp per term.
#ifndef NBODY_PAIR_POLISH
  #define NBODY_PAIR_POLISH 2
#endif

template <>
struct pair_math<double> {
  static constexpr double tiny = 1e-300;
  // returns mj / (r2 * sqrt(r2) + DBL_EPSILON); POLISH = order of the steps on the two seeds
  template <int POLISH = NBODY_PAIR_POLISH>
  __device__ static __forceinline__ double weight(double r2, double mj) {
    double y0 = __builtin_amdgcn_rsq(r2);     // ~2^-24 relative
    double h  = r2 * y0;                      // ~sqrt(r2)
    double e  = __builtin_fma(-h, y0, 1.0);   // 1 - r2*y0^2
    double s;
    if constexpr (POLISH >= 3) {
      double p = __builtin_fma(e, 0.375, 0.5);
      s        = __builtin_fma(h * e, p, h);   // sqrt(r2)*(1 + O(e^3))
    } else {
      s = __builtin_fma(h * 0.5, e, h);        // sqrt(r2)*(1 - 3/8 e^2)
    }
    double d3 = __builtin_fma(r2, s, DBL_EPSILON);
    double z0 = __builtin_amdgcn_rcp(d3);     // ~2^-24 relative
    double e2 = __builtin_fma(-d3, z0, 1.0);
    double zm = z0 * mj;
    if constexpr (POLISH >= 3) {
      double q = __builtin_fma(e2, e2, e2);   // e2 + e2^2
      return __builtin_fma(zm, q, zm);        // mj/d3 * (1 + O(e2^3))
    } else {
      return __builtin_fma(zm, e2, zm);       // mj/d3 * (1 - e2^2)
    }
  }

  // The same quantity without the reciprocal, for r2 >= 2^-16 (K1's fast path).  With u = r2^(-3/2):
  //     mj / (r2^(3/2) + eps) = mj * u / (1 + eps*u) = mj * u * (1 - eps*u + (eps*u)^2 - ...)
  // and eps*u <= 2^-52 * 2^24 = 2^-28 there, so dropping (eps*u)^2 <= 2^-56 is below half an ulp.  u comes from the
  // 2^-24 v_rsq_f64 seed y by one THIRD-order step on the cube:  y^3 * (1 - e)^(-3/2) = y^3 * (1 + 3/2 e + 15/8 e^2 + O(e^3)),
  // e = 1 - r2*y^2 <= 2^-23 (truncation 35/16 e^3 < 3e-21).  Both corrections share one FMA:
  //     w = (mj*y^3) * (1 + g),  g = e*(3/2 + 15/8 e) - eps*y^3          (the cross term of the two is < 2^-50 * 2^-28)
  // 8 full-rate ops + 1 transcendental = 48 issue cycles per wave-pair instead of 12 + 2 = 80 for weight(), and only
  // rounding error is left: a = fl(y*y) perturbs e by <= 2^-53 (x 3/2 in w), then y^3, mj*y^3 and the final FMA round once
  // each: <= 2.5 ulp per term, no bias (measured in tests/test_gpu_all_pairs.py::test_pair_term_accuracy).
  static constexpr uint32_t near_hi = 0x3EF00000u;  // high word of 2^-16: r2 below this takes weight()
  // r2 >= 4: eps*u <= 2^-52 / 8 = 2^-55, an eighth of an ulp of the weight — the term is dropped (weight_far<false>), which
  // saves the eps*y^3 multiply: 7 full-rate ops + 1 transcendental.  K1 takes that form when EVERY pair of a batch is that far
  // (one more compare on the minimum it already tracks); in a mixed batch each lane selects eps or 0 by its own r2 (the high
  // word of DBL_EPSILON is all there is to select: its low word is 0), and fma(p, e, -0) == p * e bit for bit, so a pair's
  // weight never depends on which other pairs share the batch.
  static constexpr uint32_t far_hi = 0x40100000u;   // high word of 4.0
  // k15 = 1.5 in a VGPR pair and k1875 = 1.875 in an SGPR pair, held by the caller (pair_consts): neither is an inline
  // constant, and left to itself hipcc feeds a VOP2 v_fmac with the literal and re-creates 1.5 in its accumulator
  // operand with a v_mov per pair.
  template <bool EPS = true>
  __device__ static __forceinline__ double weight_far(double r2, double mj, double k15, double k1875, double eps = DBL_EPSILON) {
    double y  = __builtin_amdgcn_rsq(r2);
    double a  = y * y;
    double e  = __builtin_fma(-r2, a, 1.0);
    double y3 = a * y;
    double p  = __builtin_fma(e, k1875, k15);
    double g;
    if constexpr (EPS) g = __builtin_fma(p, e, -(eps * y3));  // eps: DBL_EPSILON, or a lane's own choice of DBL_EPSILON / 0
    else g = p * e;
    double my = mj * y3;
    return __builtin_fma(my, g, my);
  }
};

template <>
struct pair_math<float> {
  static constexpr float tiny = 1e-37f;
  // returns mj / (r2 * sqrt(r2) + FLT_EPSILON).  v_rsq_f32 / v_rcp_f32 are 1-ulp instructions, so no polish:
  // the term is within ~4 ulp (5e-7) of the exact expression, below the spread between the reference's own
  // float builds (its powf/divide chain is not reproducible at this level) and far below the
  // float parity tolerance (2e-5).  3 full-rate ops + 2 transcendentals.
  __device__ static __forceinline__ float weight(float r2, float mj) {
    float y0 = __builtin_amdgcn_rsqf(r2);
    float d3 = __builtin_fmaf(r2, r2 * y0, FLT_EPSILON);
    return __builtin_amdgcn_rcpf(d3) * mj;
  }
  // Sparse systems (ap_far_mode): a pair at r2 >= 4 takes m y^3 — eps u <= 2^-26 there, an eighth of an ulp.  Measured with it
  // and not kept (round 3, profiles/r03/f32_pair_forms.txt): the reciprocal-free series m u (1 - q + q^2) for every far pair —
  // v_rcp_f32 costs ~3 FMA slots on this chip, the series 2-3 FMAs plus a near check: slower wherever the check is taken.
  static constexpr uint32_t far_bits = 0x40800000u;  // 4.0f
};

// LDS source record: (x[0..D-1], m) padded to a power-of-two size so one or two ds_read_b128 fetch it.
template <typename T, int D>
struct alignas(sizeof(T) * 4) src_rec {
  T p[3];  // D used
  T m;
};

// acc += w * (xj - xi) with r2 built by FMAs starting from TINY.
template <typename T, int D>
__device__ __forceinline__ void pair_accumulate(T (&acc)[D], const T (&xi)[D], const src_rec<T, D>& s) {
  T d[D];
#pragma unroll
  for (int k = 0; k < D; ++k) d[k] = s.p[k] - xi[k];
  T r2 = pair_math<T>::tiny;
#pragma unroll
  for (int k = 0; k < D; ++k) r2 = __builtin_elementwise_fma(d[k], d[k], r2);
  T w = pair_math<T>::weight(r2, s.m);
#pragma unroll
  for (int k = 0; k < D; ++k) acc[k] = __builtin_elementwise_fma(w, d[k], acc[k]);
}

// NB sources against one target, written stage by stage (all differences, all r2, all weights, then the accumulation)
// so that NB independent dependency chains are in flight: under register pressure hipcc otherwise emits one pair after
// the other through the same temporaries, each transcendental followed by a wait state (K2: 446 s_nop per 256 pairs).
template <typename T, int D, int NB>
__device__ __forceinline__ void pair_accumulate_multi(T (&acc)[D], const T (&xi)[D], const src_rec<T, D>* s) {
  T d[NB][D], r2[NB], w[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int k = 0; k < D; ++k) d[b][k] = s[b].p[k] - xi[k];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    r2[b] = pair_math<T>::tiny;
#pragma unroll
    for (int k = 0; k < D; ++k) r2[b] = __builtin_elementwise_fma(d[b][k], d[b][k], r2[b]);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) w[b] = pair_math<T>::weight(r2[b], s[b].m);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] = __builtin_elementwise_fma(w[b], d[b][k], acc[k]);
  }
}

// K2 in f64: the same pairs with K1's reciprocal-free weight and NO per-chain branch (a near-pair branch in every chain makes
// hipcc keep all chains' temporaries alive: 350 VGPRs).  The caller tracks the smallest high word of r2 it has seen and, if a
// block of pairs held one below 2^-16 (a self pair, a very close body), discards that block's sums and recomputes them with the
// guarded form (pair_accumulate_multi) — the far results of such a block may be inf/NaN and are never used.
template <int D, int NB>
__device__ __forceinline__ void pair_accumulate_far(double (&acc)[D], const double (&xi)[D], const src_rec<double, D>* s,
                                                    uint32_t& lowest, double k15, double k1875) {
  double d[NB][D], r2[NB], w[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int k = 0; k < D; ++k) d[b][k] = s[b].p[k] - xi[k];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    r2[b] = pair_math<double>::tiny;
#pragma unroll
    for (int k = 0; k < D; ++k) r2[b] = __builtin_elementwise_fma(d[b][k], d[b][k], r2[b]);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    w[b]              = pair_math<double>::weight_far(r2[b], s[b].m, k15, k1875);
    const uint32_t hi = uint32_t(__builtin_bit_cast(unsigned long long, r2[b]) >> 32);
    lowest            = hi < lowest ? hi : lowest;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] = __builtin_elementwise_fma(w[b], d[b][k], acc[k]);
  }
}

// Loop-invariant operands of pair_batch that must stay in registers (made opaque to the optimiser once, outside the loop).
template <typename T>
struct pair_consts {
  __device__ __forceinline__ pair_consts() {}
};
template <>
struct pair_consts<double> {
  double k15, k1875;
  __device__ __forceinline__ pair_consts() : k15(1.5), k1875(1.875) {
    asm volatile("" : "+v"(k15));
    asm volatile("" : "+s"(k1875));
  }
};

// K1's unit of work: U source records against the R targets of a lane.  acc[r] += w * (x_j - x_i) in source order.
// `ffar` is a property of the LAUNCH, compiled as two copies of the kernels' source loops — inside ONE loop hipcc hoists the
// common head of the two rules above the branch and interleaves the f32 pair chains, which costs 15 % (24.4 against 21.2 ms at
// config 3) — (ap_far_mode: the variances of ALL positions say that few batches can hold a pair closer than 2 — k1_rule below;
// every rank and every shard window of one system computes the same bits) and selects the per-pair rule:
//   dense (ffar false): f64 weight_far<true> for r2 >= 2^-16, weight<3>() below; f32 weight().
//   sparse (ffar true): f64 weight_far<false> for r2 >= 4, weight_far<true> in [2^-16, 4), weight<3>() below;
//                       f32 m y^3 for r2 >= 4, weight() below.
// The smallest r2 of the batch (one v_min3_u32 per two pairs on the high words) decides with one wave-uniform branch which code
// runs; in a mixed batch each lane keeps, pair by pair, what its own r2 asks for.  Under a given rule a pair's term depends on
// that pair alone, never on which other targets share the wave: results stay bitwise independent of the shard window.
// (In the dense regime the sparse rule would put every batch on the mixed path: measured 3.20 against 2.62 ms at config 2.)
template <typename T, int D, int R, int U, bool ffar>
__device__ __forceinline__ void pair_batch(T (&acc)[R][D], const T (&xi)[R][D], const src_rec<T, D> (&s)[U],
                                           const pair_consts<T>& pc) {
  if constexpr (sizeof(T) == 4) {
    if constexpr (!ffar) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) pair_accumulate<T, D>(acc[r], xi[r], s[u]);
    } else {
      // sparse system (ap_far_mode): a pair at r2 >= 4 takes m y^3 — eps u <= 2^-26 there, an eighth of an ulp — and every other
      // pair the guarded form above, by ITS OWN r2; a batch whose pairs are all that far never issues the reciprocal
      T d[U][R][D], r2[U][R], w[U][R];
      uint32_t lowest = 0xffffffffu;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
          for (int k = 0; k < D; ++k) d[u][r][k] = s[u].p[k] - xi[r][k];
          T q = pair_math<T>::tiny;
#pragma unroll
          for (int k = 0; k < D; ++k) q = __builtin_elementwise_fma(d[u][r][k], d[u][r][k], q);
          r2[u][r]            = q;
          const uint32_t bits = __builtin_bit_cast(uint32_t, q);
          lowest              = bits < lowest ? bits : lowest;
        }
      T y[U][R];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          y[u][r] = __builtin_amdgcn_rsqf(r2[u][r]);
          w[u][r] = s[u].m * ((y[u][r] * y[u][r]) * y[u][r]);
        }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(lowest < pair_math<T>::far_bits) != 0ull, 0)) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const T d3 = __builtin_fmaf(r2[u][r], r2[u][r] * y[u][r], FLT_EPSILON);  // == pair_math<float>::weight
            const T wn = __builtin_amdgcn_rcpf(d3) * s[u].m;
            w[u][r]    = __builtin_bit_cast(uint32_t, r2[u][r]) < pair_math<T>::far_bits ? wn : w[u][r];
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int k = 0; k < D; ++k) acc[r][k] = __builtin_elementwise_fma(w[u][r], d[u][r][k], acc[r][k]);
    }
  } else {
    T d[U][R][D], r2[U][R], w[U][R];
    uint32_t lowest = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < D; ++k) d[u][r][k] = s[u].p[k] - xi[r][k];
        T q = pair_math<T>::tiny;
#pragma unroll
        for (int k = 0; k < D; ++k) q = __builtin_elementwise_fma(d[u][r][k], d[u][r][k], q);
        r2[u][r] = q;
        const uint32_t hi = uint32_t(__builtin_bit_cast(unsigned long long, q) >> 32);
        lowest            = hi < lowest ? hi : lowest;
      }
    if constexpr (!ffar) {
      // dense system: every pair takes weight_far<true> unless some lane holds one below 2^-16
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) w[u][r] = pair_math<T>::template weight_far<true>(r2[u][r], s[u].m, pc.k15, pc.k1875);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(lowest < pair_math<T>::near_hi) != 0ull, 0)) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t hi = uint32_t(__builtin_bit_cast(unsigned long long, r2[u][r]) >> 32);
            const T wn        = pair_math<T>::template weight<3>(r2[u][r], s[u].m);  // rare: take the <= 2 ulp form
            w[u][r]           = hi < pair_math<T>::near_hi ? wn : w[u][r];
          }
      }
    } else if (__builtin_expect(__builtin_amdgcn_ballot_w64(lowest < pair_math<T>::far_hi) == 0ull, 1)) {
      // sparse system, every pair of the batch at r2 >= 4: no eps term at all
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) w[u][r] = pair_math<T>::template weight_far<false>(r2[u][r], s[u].m, pc.k15, pc.k1875);
    } else {
      // sparse system, some pair is closer: every lane keeps, pair by pair, what ITS r2 asks for — eps or 0 in the far form ...
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t hi  = uint32_t(__builtin_bit_cast(unsigned long long, r2[u][r]) >> 32);
          const uint32_t ehi = hi < pair_math<T>::far_hi ? 0x3CB00000u : 0u;  // high word of DBL_EPSILON (its low word is 0)
          const T eps        = __builtin_bit_cast(T, (unsigned long long)ehi << 32);
          w[u][r]            = pair_math<T>::template weight_far<true>(r2[u][r], s[u].m, pc.k15, pc.k1875, eps);
        }
      // ... and the guarded reciprocal form below 2^-16 (the self pair, coincident or very close bodies)
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(lowest < pair_math<T>::near_hi) != 0ull, 0)) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t hi = uint32_t(__builtin_bit_cast(unsigned long long, r2[u][r]) >> 32);
            const T wn        = pair_math<T>::template weight<3>(r2[u][r], s[u].m);
            w[u][r]           = hi < pair_math<T>::near_hi ? wn : w[u][r];
          }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < D; ++k) acc[r][k] = __builtin_elementwise_fma(w[u][r], d[u][r][k], acc[r][k]);
  }
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- which per-pair rule a launch takes: ap_far_mode ---------------------------------------------------------------------
// Sparse iff a batch of 256 pairs rarely holds one closer than 2.  For bodies spread with density rho (normalised) the chance that a
// random pair is that close is (32 pi / 3) * integral(rho^2); for a uniformly filled box that integral is 1 / V, and the rule asks
// for 256 * (4/3 pi 8) / V < 0.05 (256 * 4 pi / A in 2D).  V is NOT the bounding box (rounds 3-4): one escaper inflates that at will
// — config 2 as written (the uniform cube, dt = 0.1) ejects a few bodies at ten times the bulk's speed, after 35 steps the box says
// "sparse" while 98 % of the bodies sit in a cube of side 7 and EVERY batch holds a close pair, and the sparse rule's mixed path
// made the last 65 of its 100 steps 8.5 % slower (profiles/r05/config2_state_probe.txt).  V is the volume of the uniformly
// filled box with the positions' VARIANCES: side_k = sqrt(12 var_k) — the same number for a uniform box, 0.94 of the Gaussian's
// 1 / integral(rho^2), and a few escapers move it by their share of the mass, not by their distance.  Real systems are clumpier
// than their variances (the galaxy's two discs: 6 % of the batches hold a close pair at V = 3e6) — the rule only has to tell a
// cube from a galaxy.  The moments are summed in double in a FIXED order (per block of 256 bodies, then over the blocks in index
// order by the last block to finish), from all sz positions whatever the shard window: every rank, every window and every run
// computes the same bits, so the rule — which fixes the rounding of every pair — is a function of the state alone.
struct k1_rule {
  double volume;    // prod_k sqrt(12 var_k)
  uint32_t sparse;  // volume >= threshold
  uint32_t ticket;  // blocks of the running prepare launch that have delivered their partial sums (0 between launches)
};
constexpr uint32_t kFarMinBodies = 32768;  // below this the rule is "dense" by definition (from sz alone: nothing is measured)
template <int D>
constexpr double kFarMinVolume = D == 3 ? 1.7e5 : 6.4e4;
__device__ __forceinline__ bool ap_far_mode(const k1_rule* __restrict__ rule) {
  if (rule == nullptr) return false;
  return __builtin_amdgcn_readfirstlane(int(rule->sparse)) != 0;
}

// ---- scalar-stream helpers (K1's default form, the energies) ------------------------------------------------
constexpr int kTileJ = 512;  // source records per tile (fixed: the rounding order of K1 depends on it)

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

// K1 launch shape: validation shared by the process-wide default and the per-context setting
inline int check_tuning(int split, int tpt, int path) {
  NB_ARG(split == 0 || split == 1 || split == 2 || split == 4 || split == 8, "split must be 0, 1, 2, 4 or 8 (got %d)", split);
  NB_ARG(tpt >= 0 && tpt <= 2, "targets_per_thread must be 0, 1 or 2 (got %d)", tpt);
  NB_ARG(path >= 0 && path <= 2, "source path must be 0 (auto), 1 (LDS tiles) or 2 (scalar stream), got %d", path);
  return NBODY_OK;
}

// all_pairs.hip: per-(device, stream) packed-source scratch of the scalar-stream K1 (reserved by nbody_create, freed by nbody_destroy)
int ap_scratch_reserve(hipStream_t st, const nbody_state* view);
void ap_scratch_release(hipStream_t st);
int ap_scratch_get(hipStream_t st, int which, size_t bytes, void** out);  // which: 0 packed sources, 1 K1 turn words, 2 energies, 3 the pair rule (k1_rule + the blocks' partial moments), 4 K1 hand-off status, 5 K1 chunk sums of small launches
void ap_status_mark(hipStream_t st);  // a recorded step is being replayed on st: it may hold a turn-passing K1
int ap_status_read(hipStream_t st, unsigned long long out[6], bool clear);  // waits for the stream; NBODY_ERR_STATE while a K1 hand-off failure is recorded
int ap_pack_sources(const nbody_state* s, hipStream_t st, void** packed_out);
void ap_auto_chunks(uint32_t sz, uint32_t* chunks, uint32_t* tiles_per_chunk);

}  // namespace nbody
