// K10/K11 energies — replaces System::calc_energies (src/system.h:62-79):
//   kinetic   = 0.5 * sum_i m_i |v_i|^2
//   potential = -0.5 * c * sum_i m_i * sum_{j != i} m_j / (sqrt(|x_i - x_j|^2) + eps(T))
// The O(N^2) part runs on K1's machinery (all_pairs.hip): lane = target (R per lane), the packed (x, m) records are
// streamed through the scalar cache into SGPRs two batches deep, the 512-record tile is cut into 8 slices (one per
// wave of the block, combined through LDS in wave order) and the tile sequence into the same source chunks K1 uses
// (grid.y; combined in chunk order by the reduction kernel), so the sum is deterministic.
// Pair term  m_j / (s + eps),  s = sqrt(r2):
//   far  (r2 >= 2^-48 in f64, 2^-20 in f32):  m_j * y * (1 - eps*y),  y = 1/s from the v_rsq seed polished to third
//        order in f64 (y*(1 + e/2 + 3/8 e^2), e = 1 - r2*y^2), the 1-ulp v_rsq_f32 in f32; dropping (eps*y)^2 <= 2^-56
//        (2^-26) is below half an ulp.  7 full-rate ops + 1 transcendental per pair (K1's force term: 8 + 1).
//   near (anything closer, found with one v_min3_u32 per two pairs and one wave-uniform branch per batch): the
//        reference's expression with a polished reciprocal.  Unlike the force, the self pair is NOT zero here
//        (m_i^2 / eps): it always lands in the near path (r2 = 0), where it is removed by comparing the body indices.
//        Coincident DISTINCT bodies keep the reference's m_i m_j / eps.
// Against the reference's sequential sums the result agrees to rounding (tolerance parity).  Bound: FP64/FP32 VALU issue.
#include "common.hpp"

namespace nbody {

constexpr int kEB    = 256;
constexpr int kPotJS = 8;  // source slices = waves per block

template <typename T>
struct pot_math;

template <>
struct pot_math<double> {
  static constexpr uint32_t near_bits = 0x3CF00000u;  // high word of 2^-48
  __device__ static __forceinline__ uint32_t bits(double r2) { return uint32_t(__builtin_bit_cast(unsigned long long, r2) >> 32); }
  __device__ static __forceinline__ double far(double r2, double mj, double k0375) {
    double y  = __builtin_amdgcn_rsq(r2);
    double a  = y * y;
    double e  = __builtin_fma(-r2, a, 1.0);
    double p  = __builtin_fma(e, k0375, 0.5);
    double ey = DBL_EPSILON * y;
    double g  = __builtin_fma(p, e, -ey);
    double my = mj * y;
    return __builtin_fma(my, g, my);
  }
  __device__ static __forceinline__ double near(double r2, double mj) {  // mj / (sqrt(r2) + eps)
    double y0 = __builtin_amdgcn_rsq(r2);
    double h  = r2 * y0;
    double e  = __builtin_fma(-h, y0, 1.0);
    double p  = __builtin_fma(e, 0.375, 0.5);
    double s  = __builtin_fma(h * e, p, h);
    double d  = s + DBL_EPSILON;
    double z0 = __builtin_amdgcn_rcp(d);
    double e2 = __builtin_fma(-d, z0, 1.0);
    double q  = __builtin_fma(e2, e2, e2);
    double zm = z0 * mj;
    return __builtin_fma(zm, q, zm);
  }
};

template <>
struct pot_math<float> {
  static constexpr uint32_t near_bits = 0x35800000u;  // 2^-20
  __device__ static __forceinline__ uint32_t bits(float r2) { return __builtin_bit_cast(uint32_t, r2); }
  __device__ static __forceinline__ float far(float r2, float mj, float) {
    float y  = __builtin_amdgcn_rsqf(r2);
    float my = mj * y;
    return __builtin_fmaf(my, -FLT_EPSILON * y, my);
  }
  __device__ static __forceinline__ float near(float r2, float mj) {
    float d = __builtin_fmaf(r2, __builtin_amdgcn_rsqf(r2), FLT_EPSILON);
    return __builtin_amdgcn_rcpf(d) * mj;
  }
};

template <typename T>
struct pot_consts {
  T k0375;
  __device__ __forceinline__ pot_consts() : k0375(T(0.375)) { asm volatile("" : "+s"(k0375)); }  // not an inline constant
};

// U source records (global indices j0 ...) against the R targets of a lane (global indices tg[r]): acc[r] += term
template <typename T, int D, int R, int U>
__device__ __forceinline__ void pot_batch(T (&acc)[R], const T (&xi)[R][D], const uint32_t (&tg)[R], const src_rec<T, D> (&s)[U],
                                          uint32_t j0, const pot_consts<T>& pc) {
  T r2[U][R], w[U][R];
  uint32_t lowest = 0xffffffffu;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T q = pair_math<T>::tiny;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const T d = s[u].p[k] - xi[r][k];
        q         = __builtin_elementwise_fma(d, d, q);
      }
      r2[u][r]          = q;
      const uint32_t hi = pot_math<T>::bits(q);
      lowest            = hi < lowest ? hi : lowest;
    }
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < R; ++r) w[u][r] = pot_math<T>::far(r2[u][r], s[u].m, pc.k0375);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(lowest < pot_math<T>::near_bits) != 0ull, 0)) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T wn    = pot_math<T>::near(r2[u][r], s[u].m);
        wn      = (j0 + uint32_t(u) == tg[r]) ? T(0) : wn;  // the self pair
        w[u][r] = pot_math<T>::bits(r2[u][r]) < pot_math<T>::near_bits ? wn : w[u][r];
      }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] += w[u][r];
}

// sums[chunk][i] = sum over the chunk's sources j != i of m_j / (|x_i - x_j| + eps)
template <typename T, int D, int R>
__global__ __launch_bounds__(64 * kPotJS) void potential_sgpr_kernel(const src_rec<T, D>* __restrict__ packed, const T* __restrict__ x,
                                                                     T* __restrict__ sums, uint32_t sz, uint32_t tiles_per_chunk) {
  using rec_t       = src_rec<T, D>;
  constexpr int TB  = 64 * R;
  constexpr int SUB = kTileJ / kPotJS;
  __shared__ T partial[(kPotJS - 1) * 64 * R];
  const int lane  = threadIdx.x & 63;
  const int jpart = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T xi[R][D], acc[R];
  uint32_t tg[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    tg[r]            = blockIdx.x * TB + r * 64 + lane;
    const uint64_t i = tg[r] < sz ? tg[r] : 0u;  // clamp: out-of-range lanes compute, never store
#pragma unroll
    for (int k = 0; k < D; ++k) xi[r][k] = x[i * D + k];
    acc[r] = T(0);
  }
  const uint32_t ntiles = (sz + kTileJ - 1) / kTileJ;
  const uint32_t t0     = blockIdx.y * tiles_per_chunk;
  const uint32_t t1     = min(ntiles, t0 + tiles_per_chunk);
  const pot_consts<T> pc;
  const uint32_t nsteps = (t1 - t0) * SUB;
  constexpr int U       = 64 / int(sizeof(rec_t));
  struct batch_t {
    rec_t r[U];
  };
  auto index = [&](uint32_t k) { return (t0 + k / SUB) * uint32_t(kTileJ) + uint32_t(jpart) * SUB + (k % SUB); };
  auto batch = [&](uint32_t k) { return packed + uint64_t(index(k)); };
  // the same two-deep SMEM pipeline as all_pairs_force_sgpr_kernel (see there)
  sgpr16 A = sload16(batch(0), xi[0][0]), B;
  for (uint32_t k = 0; k < nsteps; k += 2 * U) {
    swait(A, acc[0]);
    B = sload16(batch(k + U), xi[0][0]);
    {
      const batch_t ba = __builtin_bit_cast(batch_t, A);
      pot_batch<T, D, R, U>(acc, xi, tg, ba.r, index(k), pc);
    }
    swait(B, acc[0]);
    A = sload16(batch(k + 2 * U < nsteps ? k + 2 * U : k), xi[0][0]);
    {
      const batch_t bb = __builtin_bit_cast(batch_t, B);
      pot_batch<T, D, R, U>(acc, xi, tg, bb.r, index(k + U), pc);
    }
  }
  swait(A, acc[0]);
  if (jpart > 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) partial[((jpart - 1) * R + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (jpart == 0) {
#pragma unroll
    for (int p = 1; p < kPotJS; ++p)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] += partial[((p - 1) * R + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (tg[r] < sz) sums[uint64_t(blockIdx.y) * sz + tg[r]] = acc[r];
  }
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red /* [kEB/64] */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T t = red[0];
  for (int w = 1; w < kEB / 64; ++w) t += red[w];
  return t;
}

// per block: sum_i m_i |v_i|^2 and sum_i m_i * (chunk sums of i in chunk order), bodies dealt to blocks grid-stride
template <typename T, int D>
__global__ __launch_bounds__(kEB) void energy_partial_kernel(const T* __restrict__ m, const T* __restrict__ v,
                                                             const T* __restrict__ sums, uint32_t nchunks, uint32_t sz,
                                                             T* __restrict__ partial) {
  __shared__ T red[kEB / 64];
  T ke = T(0), pe = T(0);
  for (uint64_t i = uint64_t(blockIdx.x) * kEB + threadIdx.x; i < sz; i += uint64_t(gridDim.x) * kEB) {
    T tot = sums[i];
    for (uint32_t y = 1; y < nchunks; ++y) tot += sums[uint64_t(y) * sz + i];
    T n2 = T(0);
#pragma unroll
    for (int k = 0; k < D; ++k) n2 = __builtin_elementwise_fma(v[i * D + k], v[i * D + k], n2);
    ke += m[i] * n2;
    pe += m[i] * tot;
  }
  ke = block_sum(ke, red);
  pe = block_sum(pe, red);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x + 0] = ke;
    partial[2 * blockIdx.x + 1] = pe;
  }
}

template <typename T>
__global__ __launch_bounds__(kEB) void energy_final_kernel(const T* __restrict__ partial, uint32_t nblk, T c, T* __restrict__ out) {
  __shared__ T red[kEB / 64];
  T ke = T(0), pe = T(0);
  for (uint32_t b = threadIdx.x; b < nblk; b += kEB) {
    ke += partial[2 * b + 0];
    pe += partial[2 * b + 1];
  }
  ke = block_sum(ke, red);
  pe = block_sum(pe, red);
  if (threadIdx.x == 0) {
    out[0] = T(0.5) * ke;
    out[1] = -T(0.5) * c * pe;
  }
}

template <typename T, int D>
static int energies_run(const nbody_state* s, void* ke_out, void* pe_out, hipStream_t st) {
  constexpr int R       = sizeof(T) == 8 ? 2 : 1;
  const uint32_t n      = s->sz;
  const uint32_t ntiles = (n + kTileJ - 1) / kTileJ;
  uint32_t chunks = 1, tpc = ntiles;
  ap_auto_chunks(n, &chunks, &tpc);
  uint32_t nblk = (n + kEB - 1) / kEB;
  if (nblk > 1024) nblk = 1024;
  // persistent work area of this stream: chunk sums [chunks][n], block partials [2 * nblk], result [2]
  void* w = nullptr;
  if (int r = ap_scratch_get(st, 2, sizeof(T) * (size_t(chunks) * n + 2 * size_t(nblk) + 2), &w)) return r;
  T* sums    = static_cast<T*>(w);
  T* partial = sums + size_t(chunks) * n;
  T* out     = partial + 2 * size_t(nblk);
  void* packed = nullptr;
  if (int r = ap_pack_sources(s, st, &packed)) return r;
  hipLaunchKernelGGL((potential_sgpr_kernel<T, D, R>), dim3((n + 64 * R - 1) / (64 * R), chunks), dim3(64 * kPotJS), 0, st,
                     static_cast<const src_rec<T, D>*>(packed), static_cast<const T*>(s->x), sums, n, tpc);
  NB_HIP(hipGetLastError());
  hipLaunchKernelGGL((energy_partial_kernel<T, D>), dim3(nblk), dim3(kEB), 0, st, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->v), sums, chunks, n, partial);
  NB_HIP(hipGetLastError());
  hipLaunchKernelGGL((energy_final_kernel<T>), dim3(1), dim3(kEB), 0, st, partial, nblk, static_cast<T>(s->c), out);
  NB_HIP(hipGetLastError());
  T host[2];
  NB_HIP(hipMemcpyAsync(host, out, sizeof host, hipMemcpyDeviceToHost, st));
  NB_HIP(hipStreamSynchronize(st));
  *static_cast<T*>(ke_out) = host[0];
  *static_cast<T*>(pe_out) = host[1];
  return NBODY_OK;
}

}  // namespace nbody

using namespace nbody;

extern "C" int nbody_calc_energies(const nbody_state* s, void* kinetic_out, void* potential_out, void* stream) {
  if (int r = check_state(s)) return r;
  NB_ARG(kinetic_out && potential_out, "NULL output pointer");
  NB_ARG(s->first == 0 && s->count == s->sz, "nbody_calc_energies needs the whole system (first=0, count=sz)");
  NB_ARG(s->sz >= 1, "empty system");
  device_guard guard(stream_device(as_stream(stream)));
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return energies_run<typename TG::type, TG::dim>(s, kinetic_out, potential_out, as_stream(stream));
  });
}
