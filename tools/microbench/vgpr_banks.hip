// Microbenchmark: does it matter which VGPR banks (register number mod 4) the operands of a wave64 v_fmac_f32 come from?
// Eight independent v_fmac_f32 per iteration on hard-coded registers; every variant issues the same instructions and differs only in
// the register numbers of the two multiplicands relative to the accumulator.  Full occupancy (8 waves per SIMD), cycles per
// instruction per SIMD from s_memtime.  Diagnostic tool only.  Build: hipcc -O3 --offload-arch=gfx950 vgpr_banks.hip -o vgpr_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 1 << 15;

// acc registers v8..v15; A operands and B operands given per variant
#define FM(acc, a, b) "v_fmac_f32_e32 v" #acc ", v" #a ", v" #b "\n\t"
template <int V>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc) {
  asm volatile("v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"
               "v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
               "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n"
               "v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n" ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
               "v20", "v21", "v22", "v23");
  unsigned long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (V == 0)  // all three operands in different banks: acc b0..b3, a = acc+1, b = acc+2 (mod 4)
      asm volatile(FM(8, 17, 22) FM(9, 18, 23) FM(10, 19, 20) FM(11, 16, 21) FM(12, 17, 22) FM(13, 18, 23) FM(14, 19, 20) FM(15, 16, 21) ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    if constexpr (V == 1)  // a and b in the same bank (different registers), acc elsewhere
      asm volatile(FM(8, 17, 21) FM(9, 18, 22) FM(10, 19, 23) FM(11, 16, 20) FM(12, 17, 21) FM(13, 18, 22) FM(14, 19, 23) FM(15, 16, 20) ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    if constexpr (V == 2)  // a in the accumulator's bank
      asm volatile(FM(8, 16, 21) FM(9, 17, 22) FM(10, 18, 23) FM(11, 19, 20) FM(12, 16, 21) FM(13, 17, 22) FM(14, 18, 23) FM(15, 19, 20) ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    if constexpr (V == 3)  // all three in one bank
      asm volatile(FM(8, 16, 20) FM(9, 17, 21) FM(10, 18, 22) FM(11, 19, 23) FM(12, 16, 20) FM(13, 17, 21) FM(14, 18, 22) FM(15, 19, 23) ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    if constexpr (V == 4)  // a == b (the square: one register read twice), acc in another bank
      asm volatile(FM(8, 17, 17) FM(9, 18, 18) FM(10, 19, 19) FM(11, 16, 16) FM(12, 17, 17) FM(13, 18, 18) FM(14, 19, 19) FM(15, 16, 16) ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
  }
  unsigned long long t1 = clock64();
  float r;
  asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int V>
void run(const char* name, float* out, unsigned long long* cyc) {
  hipLaunchKernelGGL(k<V>, dim3(256 * 4), dim3(512), 0, 0, out, cyc);  // 4 blocks of 8 waves per CU: 8 waves per SIMD
  CK(hipDeviceSynchronize());
  unsigned long long c;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  // clock64 = s_memtime at 100 MHz on gfx9: report relative numbers
  printf("%-58s %8.3f ticks per 1000 wave-instructions of one wave\n", name, double(c) / (double(ITERS) * 8) * 1000.0);
}

int main() {
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, 4 * 256 * 4 * 512));
  CK(hipMalloc(&cyc, 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("acc, a, b in three different banks", out, cyc);
    run<1>("a and b in one bank", out, cyc);
    run<2>("a in the accumulator's bank", out, cyc);
    run<3>("acc, a, b in one bank", out, cyc);
    run<4>("a == b (a square), acc in another bank", out, cyc);
  }
  return 0;
}
