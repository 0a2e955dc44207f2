// Microbenchmark: per-SIMD issue cost (shader cycles per wave64 instruction) of the VALU ops the
// all-pairs inner loop is built from, plus the accuracy of the v_rsq_f64 / v_rcp_f64 seeds.
// Diagnostic tool only (not part of the product library).  Build:
//   hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 65536;

template <int OP>
__global__ void k_f64(double* out, unsigned long long* cyc, double b, double c) {
  double r0 = threadIdx.x * 1e-3 + 1.0, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,
         r7 = r0 + 7;
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
  unsigned long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (OP == 0)
      asm volatile(
       "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
       "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 1)
      asm volatile(
       "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
       "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 2)
      asm volatile(
       "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
       "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 3)
      asm volatile(
       "v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
       "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 4)
      asm volatile(
       "v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
       "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 5)
      asm volatile(
       "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
       "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 6)
      asm volatile(
       "v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n"
       "v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 7) {  // f64 -> f32 -> f64 round trip (2 instructions per chain element)
      float f0, f1, f2, f3;
      asm volatile(
       "v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3\n"
       "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3));
    }
    if constexpr (OP == 8) {  // compare + 2x cndmask (select on f64)
      asm volatile(
       "v_cmp_gt_f64 vcc, %0, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
       "v_cmp_gt_f64 vcc, %1, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
       "v_cmp_gt_f64 vcc, %2, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
       "v_cmp_gt_f64 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
       : "v"(b)
       : "vcc");
    }
    if constexpr (OP == 12)  // 8 independent f64 compares into 8 different SGPR pairs
      asm volatile(
       "v_cmp_gt_f64 s[20:21], %0, %8\n v_cmp_gt_f64 s[22:23], %1, %8\n v_cmp_gt_f64 s[24:25], %2, %8\n v_cmp_gt_f64 s[26:27], %3, %8\n"
       "v_cmp_gt_f64 s[28:29], %4, %8\n v_cmp_gt_f64 s[30:31], %5, %8\n v_cmp_gt_f64 s[32:33], %6, %8\n v_cmp_gt_f64 s[34:35], %7, %8\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c)
       : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    if constexpr (OP == 13)  // 8 independent 32-bit integer compares
      asm volatile(
       "v_cmp_gt_i32 s[20:21], %0, %4\n v_cmp_gt_i32 s[22:23], %1, %4\n v_cmp_gt_i32 s[24:25], %2, %4\n v_cmp_gt_i32 s[26:27], %3, %4\n"
       "v_cmp_gt_i32 s[28:29], %0, %4\n v_cmp_gt_i32 s[30:31], %1, %4\n v_cmp_gt_i32 s[32:33], %2, %4\n v_cmp_gt_i32 s[34:35], %3, %4\n"
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
       : "v"(threadIdx.x)
       : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    if constexpr (OP == 14)  // 8 v_cndmask_b32 on 4 registers, fixed condition
      asm volatile(
       "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
       "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
       : "v"(threadIdx.x)
       : "vcc");
    if constexpr (OP == 16) {  // 8 independent v_cndmask_b32 (distinct destinations), condition in vcc
      unsigned d0, d1, d2, d3, d4, d5, d6, d7;
      asm volatile(
       "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %9, %10, vcc\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %3, %11, %8, vcc\n"
       "v_cndmask_b32 %4, %8, %10, vcc\n v_cndmask_b32 %5, %9, %11, vcc\n v_cndmask_b32 %6, %10, %8, vcc\n v_cndmask_b32 %7, %11, %9, vcc\n"
       : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
       : "v"(u0), "v"(u1), "v"(u2), "v"(u3)
       : "vcc");
      u0 ^= d0 ^ d4; u1 ^= d1 ^ d5; u2 ^= d2 ^ d6; u3 ^= d3 ^ d7;
    }
    if constexpr (OP == 17) {  // same with the condition in an SGPR pair (VOP3 form)
      unsigned d0, d1, d2, d3, d4, d5, d6, d7;
      asm volatile(
       "v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %9, %10, s[20:21]\n v_cndmask_b32 %2, %10, %11, s[20:21]\n v_cndmask_b32 %3, %11, %8, s[20:21]\n"
       "v_cndmask_b32 %4, %8, %10, s[20:21]\n v_cndmask_b32 %5, %9, %11, s[20:21]\n v_cndmask_b32 %6, %10, %8, s[20:21]\n v_cndmask_b32 %7, %11, %9, s[20:21]\n"
       : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
       : "v"(u0), "v"(u1), "v"(u2), "v"(u3)
       : "s20", "s21");
      u0 ^= d0 ^ d4; u1 ^= d1 ^ d5; u2 ^= d2 ^ d6; u3 ^= d3 ^ d7;
    }
    if constexpr (OP == 18) {  // 8 independent v_bfi_b32 (select by a VGPR bit mask)
      unsigned d0, d1, d2, d3, d4, d5, d6, d7;
      asm volatile(
       "v_bfi_b32 %0, %8, %9, %10\n v_bfi_b32 %1, %9, %10, %11\n v_bfi_b32 %2, %10, %11, %8\n v_bfi_b32 %3, %11, %8, %9\n"
       "v_bfi_b32 %4, %8, %10, %11\n v_bfi_b32 %5, %9, %11, %8\n v_bfi_b32 %6, %10, %8, %9\n v_bfi_b32 %7, %11, %9, %10\n"
       : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
       : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
      u0 ^= d0 ^ d4; u1 ^= d1 ^ d5; u2 ^= d2 ^ d6; u3 ^= d3 ^ d7;
    }
    if constexpr (OP == 19) {  // 8 independent v_xor_b32 (baseline for a 32-bit op; the xor merges above cost the same)
      unsigned d0, d1, d2, d3, d4, d5, d6, d7;
      asm volatile(
       "v_xor_b32 %0, %8, %9\n v_xor_b32 %1, %9, %10\n v_xor_b32 %2, %10, %11\n v_xor_b32 %3, %11, %8\n"
       "v_xor_b32 %4, %8, %10\n v_xor_b32 %5, %9, %11\n v_xor_b32 %6, %10, %8\n v_xor_b32 %7, %11, %9\n"
       : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
       : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
      u0 ^= d0 ^ d4; u1 ^= d1 ^ d5; u2 ^= d2 ^ d6; u3 ^= d3 ^ d7;
    }
    if constexpr (OP == 15)  // v_ldexp_f64
      asm volatile(
       "v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n"
       "v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(u0 & 1u));
    if constexpr (OP == 9)  // dependent chain latency: one fma chain
      asm volatile(
       "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n"
       "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 10)  // mixed: 3 fma + 1 rsq (does the transcendental overlap the fma pipe?)
      asm volatile(
       "v_rsq_f64 %0, %0\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
       "v_rsq_f64 %4, %4\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 11)  // fma with an SGPR source operand
      asm volatile(
       "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
       "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "s"(b), "v"(c));
  }
  unsigned long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (u0 ^ u1 ^ u2 ^ u3);
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
__global__ void k_f32(float* out, unsigned long long* cyc, float b, float c) {
  float r0 = threadIdx.x * 1e-3f + 1.0f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,
        r7 = r0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, pb = {b, b}, pc = {c, c};
  unsigned long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (OP == 0)
      asm volatile(
       "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
       "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 1)
      asm volatile(
       "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
       "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
       : "v"(pb), "v"(pc));
    if constexpr (OP == 2)
      asm volatile(
       "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
       "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 3)
      asm volatile(
       "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
       "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
    if constexpr (OP == 4)  // mixed 3 fma : 1 rsq
      asm volatile(
       "v_rsq_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
       "v_rsq_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
       : "v"(b), "v"(c));
  }
  unsigned long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

__global__ void k_seed(const double* in, double* rsq, double* rcp, double* sq, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = in[i];
  rsq[i] = __builtin_amdgcn_rsq(x);
  rcp[i] = __builtin_amdgcn_rcp(x);
  sq[i]  = __builtin_amdgcn_sqrt(x);
}

template <typename K, typename T>
void run(const char* name, K kern, int waves_per_simd, int instr_per_iter, T b, T c) {
  int ncu = 256;
  int threads = 256 * waves_per_simd;  // 4 SIMDs x waves
  int blocks = ncu;
  if (threads > 1024) { blocks = ncu * (threads / 1024); threads = 1024; }
  size_t nthreads = size_t(blocks) * threads;
  T* out; unsigned long long* cyc;
  CK(hipMalloc(&out, nthreads * sizeof(T)));
  CK(hipMalloc(&cyc, (nthreads / 64) * sizeof(unsigned long long)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, b, c);  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, b, c);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(nthreads / 64);
  CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avg = 0; for (auto v : h) avg += double(v); avg /= h.size();
  double ninstr = double(ITERS) * instr_per_iter;
  // per-SIMD issue cost: wave-cycles * (1/waves_per_simd) / instructions
  double per_simd_instr = ninstr * waves_per_simd;
  printf("%-38s waves/SIMD=%d  memtime-ticks/instr/SIMD=%6.2f  wall=%8.3f ms  ns/instr/SIMD=%6.3f  => cycles@2.4GHz=%5.2f\n", name,
         waves_per_simd, avg / ninstr / waves_per_simd, ms, ms * 1e6 / per_simd_instr, ms * 1e6 / per_simd_instr * 2.4);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clockRate=%d kHz  wall_clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, 0);
  for (int w : {1, 2, 4, 8}) {
    run("v_fma_f64 (8 indep chains)", k_f64<0>, w, 8, 1.0000001, 1e-9);
    run("v_mul_f64", k_f64<1>, w, 8, 1.0000001, 0.0);
    run("v_add_f64", k_f64<2>, w, 8, 1e-9, 0.0);
    run("v_max_f64", k_f64<3>, w, 8, 1.5, 0.0);
    run("v_rsq_f64", k_f64<4>, w, 8, 1.0, 0.0);
    run("v_rcp_f64", k_f64<5>, w, 8, 1.0, 0.0);
    run("v_sqrt_f64", k_f64<6>, w, 8, 1.0, 0.0);
    run("cvt f64->f32->f64 (per cvt)", k_f64<7>, w, 8, 1.0, 0.0);
    run("cmp_f64 + 2 cndmask (per triple)", k_f64<8>, w, 4, 1.0, 0.0);
    run("v_cmp_gt_f64 (8 indep, SGPR dst)", k_f64<12>, w, 8, 1.0, 0.0);
    run("v_cmp_gt_i32 (8 indep, SGPR dst)", k_f64<13>, w, 8, 1.0, 0.0);
    run("v_cndmask_b32 (fixed vcc)", k_f64<14>, w, 8, 1.0, 0.0);
    run("v_ldexp_f64", k_f64<15>, w, 8, 1.0, 0.0);
    run("v_cndmask_b32 vcc, 8 indep (+8 xor merges; per 16 instr)", k_f64<16>, w, 16, 1.0, 0.0);
    run("v_cndmask_b32 sgpr, 8 indep (+8 xor; per 16 instr)", k_f64<17>, w, 16, 1.0, 0.0);
    run("v_bfi_b32, 8 indep (+8 xor; per 16 instr)", k_f64<18>, w, 16, 1.0, 0.0);
    run("v_xor_b32, 8 indep (+8 xor; per 16 instr)", k_f64<19>, w, 16, 1.0, 0.0);
    run("v_fma_f64 dependent chain", k_f64<9>, w, 8, 1.0000001, 1e-9);
    run("mix 3 fma_f64 : 1 rsq_f64 (per instr)", k_f64<10>, w, 8, 1.0000001, 1e-9);
    run("v_fma_f64 with SGPR operand", k_f64<11>, w, 8, 1.0000001, 1e-9);
    run("v_fma_f32", k_f32<0>, w, 8, 1.0000001f, 1e-9f);
    run("v_pk_fma_f32", k_f32<1>, w, 8, 1.0000001f, 1e-9f);
    run("v_rsq_f32", k_f32<2>, w, 8, 1.0f, 0.0f);
    run("v_rcp_f32", k_f32<3>, w, 8, 1.0f, 0.0f);
    run("mix 3 fma_f32 : 1 rsq_f32 (per instr)", k_f32<4>, w, 8, 1.0000001f, 1e-9f);
  }
  // Seed accuracy
  const int n = 1 << 20;
  std::vector<double> h(n), r1(n), r2(n), r3(n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> ue(-30.0, 30.0);
  for (int i = 0; i < n; ++i) h[i] = std::exp2(ue(g));
  double *d, *o1, *o2, *o3;
  CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&o1, n * 8)); CK(hipMalloc(&o2, n * 8)); CK(hipMalloc(&o3, n * 8));
  CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_seed, dim3(n / 256), dim3(256), 0, 0, d, o1, o2, o3, n);
  CK(hipMemcpy(r1.data(), o1, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r2.data(), o2, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r3.data(), o3, n * 8, hipMemcpyDeviceToHost));
  long double e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    long double x = h[i];
    e1 = fmaxl(e1, fabsl(r1[i] * sqrtl(x) - 1.0L));
    e2 = fmaxl(e2, fabsl(r2[i] * x - 1.0L));
    e3 = fmaxl(e3, fabsl(r3[i] / sqrtl(x) - 1.0L));
  }
  printf("seed accuracy (max rel err, log2): v_rsq_f64 %.2f  v_rcp_f64 %.2f  v_sqrt_f64 %.2f\n", (double)log2l(e1), (double)log2l(e2), (double)log2l(e3));
  return 0;
}
