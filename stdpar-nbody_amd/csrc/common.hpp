// Shared helpers for the gfx950 N-body backend: error plumbing, (dtype, dim) dispatch and the pair
// kernel math.  Device code here is written for CDNA4 only (wave64, no portability layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

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

inline int check_state(const nbody_state* s) {
  NB_ARG(s != nullptr, "nbody_state is NULL");
  NB_ARG(s->dtype == NBODY_F32 || s->dtype == NBODY_F64, "bad dtype %d", s->dtype);
  NB_ARG(s->dim == 2 || s->dim == 3, "bad dim %d (must be 2 or 3)", s->dim);
  NB_ARG(s->m && s->x && s->v && s->a && s->ao, "nbody_state has a NULL array pointer");
  NB_ARG(uint64_t(s->first) + uint64_t(s->count) <= uint64_t(s->sz), "shard [%u, %u+%u) exceeds sz=%u", s->first, s->first,
         s->count, s->sz);
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
// i.e. one pow, one add and D divisions per pair.  On gfx950 FP64 transcendentals run at quarter
// rate and do not overlap the FMA pipe (profiles/r01_valu_rates_microbench.txt), and an IEEE divide is
// a ~10-instruction sequence, so the kernel evaluates   w = m_j / (r2*sqrt(r2) + eps)   once per pair
// with two 2^-24 hardware seeds (v_rsq, v_rcp) each polished by one Newton step, then D FMAs.
// Error budget vs the exact expression: within [-2e-15, +4.1e-15] per term (see NBODY_PAIR_POLISH below;
// tests/test_gpu_all_pairs.py::test_pair_term_accuracy pins the measured size).
//
// r2 must be > 0: callers fold TINY into the first FMA of r2 (r2 = fma(dx,dx,TINY)), which makes the
// self term and coincident bodies evaluate to exactly 0 * (finite) = 0 — the value the reference
// gets from `(m*0)/eps` (self term skipped by `if (i == j) continue`) — with no branch
// and no select (a v_cmp + 2 v_cndmask select costs ~5 FMA slots on this chip).
template <typename T>
struct pair_math;

// NBODY_PAIR_POLISH: order of the polish applied to the two 2^-24 FP64 seeds.
//   2 (default): Newton steps  s = h + h*e/2,  w = zm + zm*e2        17 full-rate ops + 2 transcendentals per pair.
//      Per-term error bound: +3/8*e^2 (e <= 2^-23.2) - e2^2 (e2 <= 2^-24.4)  =>  within [-2e-15, +4.1e-15] relative.
//      Measured on accelerations against the oracle in the SAME summation order: max 1.9e-15, mean signed 4e-17 (no
//      bias) — the same size as the order-of-summation differences (2-3e-15) and 3 decades under the 1e-12 parity
//      tolerance.  758.6 ms per force pass at N=2^20 (36.9 % of FP64 peak).
//   3: third-order steps (+2 FMA): <= 2 ulp per term (max 3e-16 on accelerations), 814 ms (34.4 %).
#ifndef NBODY_PAIR_POLISH
  #define NBODY_PAIR_POLISH 2
#endif

template <>
struct pair_math<double> {
  static constexpr double tiny = 1e-300;
  // returns mj / (r2 * sqrt(r2) + DBL_EPSILON)
  __device__ static __forceinline__ double weight(double r2, double mj) {
    double y0 = __builtin_amdgcn_rsq(r2);     // ~2^-24 relative
    double h  = r2 * y0;                      // ~sqrt(r2)
    double e  = __builtin_fma(-h, y0, 1.0);   // 1 - r2*y0^2
#if NBODY_PAIR_POLISH >= 3
    double p  = __builtin_fma(e, 0.375, 0.5);
    double s  = __builtin_fma(h * e, p, h);   // sqrt(r2)*(1 + O(e^3))
#else
    double s  = __builtin_fma(h * 0.5, e, h); // sqrt(r2)*(1 - 3/8 e^2)
#endif
    double d3 = __builtin_fma(r2, s, DBL_EPSILON);
    double z0 = __builtin_amdgcn_rcp(d3);     // ~2^-24 relative
    double e2 = __builtin_fma(-d3, z0, 1.0);
    double zm = z0 * mj;
#if NBODY_PAIR_POLISH >= 3
    double q  = __builtin_fma(e2, e2, e2);    // e2 + e2^2
    return __builtin_fma(zm, q, zm);          // mj/d3 * (1 + O(e2^3))
#else
    return __builtin_fma(zm, e2, zm);         // mj/d3 * (1 - e2^2)
#endif
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

// Same, for lanes selected by `take`; the others add w = 0, i.e. exactly nothing (d is finite).  Used by the
// wave-cooperative traversal under a WAVE-UNIFORM branch: predicating the weight (2 v_cndmask) instead of the
// control flow keeps the accumulators in place — with a divergent `if` hipcc copies all of them at both ends of
// every loop iteration.
template <typename T, int D>
__device__ __forceinline__ void pair_accumulate_if(bool take, T (&acc)[D], const T (&xi)[D], const src_rec<T, D>& s) {
  T d[D];
#pragma unroll
  for (int k = 0; k < D; ++k) d[k] = s.p[k] - xi[k];
  T r2 = pair_math<T>::tiny;
#pragma unroll
  for (int k = 0; k < D; ++k) r2 = __builtin_elementwise_fma(d[k], d[k], r2);
  T w = pair_math<T>::weight(r2, s.m);
  w   = take ? w : T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) acc[k] = __builtin_elementwise_fma(w, d[k], acc[k]);
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// all_pairs.hip: per-stream packed-source scratch of the scalar-stream K1 (reserved by nbody_create, freed by nbody_destroy)
int ap_scratch_reserve(hipStream_t st, int dtype, uint32_t n);
void ap_scratch_release(hipStream_t st);

}  // namespace nbody
