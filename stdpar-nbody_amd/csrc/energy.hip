// K10/K11 energies — replaces System::calc_energies (src/system.h:62-79):
//   kinetic   = 0.5 * sum_i m_i |v_i|^2
//   potential = -0.5 * c * sum_i sum_{j != i} m_i m_j / (sqrt(|x_i - x_j|^2) + eps(T))
// The O(N^2) part reuses K1's structure (lane = target, sources through an LDS tile).  Unlike the force, the
// self pair is NOT zero here (m_i^2/eps), so it is masked with an integer compare on the body indices;
// coincident distinct bodies keep the reference's m_i m_j / eps.  Per-block partial sums are reduced in a fixed
// order by a second kernel, so the result is deterministic; against the reference's sequential sum it agrees to
// rounding (tolerance parity).  Bound: FP64 VALU, same seeds/polish as the force kernel.
#include "common.hpp"

namespace nbody {

constexpr int kEB = 256;

template <typename T>
__device__ __forceinline__ T inv_dist_times_mass(T r2, T mj);

template <>
__device__ __forceinline__ double inv_dist_times_mass<double>(double r2, double mj) {  // mj / (sqrt(r2) + eps)
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

template <>
__device__ __forceinline__ float inv_dist_times_mass<float>(float r2, float mj) {
  float d = __builtin_fmaf(r2, __builtin_amdgcn_rsqf(r2), FLT_EPSILON);
  return __builtin_amdgcn_rcpf(d) * mj;
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

template <typename T, int D>
__global__ __launch_bounds__(kEB) void energy_partial_kernel(const T* __restrict__ m, const T* __restrict__ x,
                                                             const T* __restrict__ v, uint32_t sz, T* __restrict__ partial) {
  using rec_t = src_rec<T, D>;
  __shared__ rec_t tile[kEB];
  __shared__ T red[kEB / 64];
  const uint32_t i  = blockIdx.x * kEB + threadIdx.x;
  const bool valid  = i < sz;
  const uint64_t ic = valid ? i : 0u;
  T xi[D], tot = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) xi[k] = x[ic * D + k];
  const uint32_t ntiles = (sz + kEB - 1) / kEB;
  for (uint32_t t = 0; t < ntiles; ++t) {
    const uint64_t j = uint64_t(t) * kEB + threadIdx.x;
    rec_t r;
#pragma unroll
    for (int k = 0; k < 3; ++k) r.p[k] = T(0);
    r.m = T(0);
    if (j < sz) {
#pragma unroll
      for (int k = 0; k < D; ++k) r.p[k] = x[j * D + k];
      r.m = m[j];
    }
    __syncthreads();
    tile[threadIdx.x] = r;
    __syncthreads();
    const uint32_t j0 = t * kEB;
#pragma unroll 4
    for (int jj = 0; jj < kEB; ++jj) {
      const rec_t s = tile[jj];
      T r2 = pair_math<T>::tiny;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        T d = s.p[k] - xi[k];
        r2  = __builtin_elementwise_fma(d, d, r2);
      }
      T term = inv_dist_times_mass<T>(r2, s.m);  // padding records have mass 0
      tot += (j0 + uint32_t(jj) != i) ? term : T(0);
    }
  }
  T pe = valid ? m[ic] * tot : T(0);
  T ke = T(0);
  if (valid) {
    T n2 = T(0);
#pragma unroll
    for (int k = 0; k < D; ++k) n2 = __builtin_elementwise_fma(v[ic * D + k], v[ic * D + k], n2);
    ke = m[ic] * n2;
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
  const uint32_t nblk = (s->sz + kEB - 1) / kEB;
  T* work             = nullptr;
  NB_HIP(hipMalloc(reinterpret_cast<void**>(&work), sizeof(T) * (2 * size_t(nblk) + 2)));
  T* out = work + 2 * size_t(nblk);
  hipLaunchKernelGGL((energy_partial_kernel<T, D>), dim3(nblk), dim3(kEB), 0, st, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->x), static_cast<const T*>(s->v), s->sz, work);
  hipLaunchKernelGGL((energy_final_kernel<T>), dim3(1), dim3(kEB), 0, st, work, nblk, static_cast<T>(s->c), out);
  T host[2];
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(host, out, sizeof host, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(work);
  if (e != hipSuccess) return hip_fail(e, "nbody_calc_energies", __FILE__, __LINE__);
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
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return energies_run<typename TG::type, TG::dim>(s, kinetic_out, potential_out, as_stream(stream));
  });
}
