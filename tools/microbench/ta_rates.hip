// Microbenchmark: cost of a vector load in the texture-address unit as a function of its width and of WHICH lanes are
// active (the octree walk's loads: 8 lanes per body read 8 consecutive 16-byte pieces of a 320-byte sibling group).
// Every wave issues LOADS independent loads per iteration from an L1/L2-resident table; the chip is filled (8 waves per
// SIMD), so kernel time / loads per CU = unit cycles per load instruction.  Diagnostic tool only.  Build:
//   hipcc -O3 --offload-arch=gfx950 ta_rates.hip -o ta_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int NG    = 96;  // sibling groups in the table (30 KB)

template <int WIDTH>  // dwords per lane: 1, 2, 4
__global__ __launch_bounds__(64) void k_load(const char* __restrict__ table, unsigned long long mask, float* __restrict__ out) {
  const unsigned lane = threadIdx.x;
  const bool on       = (mask >> lane) & 1ull;
  unsigned g          = (blockIdx.x * 8u + lane / 8u) % NG;
  float acc           = 0.f;
  if (on) {
    for (int it = 0; it < ITERS; ++it) {
      const char* p = table + g * 320u + (lane & 7u) * 16u;
      if constexpr (WIDTH == 4) {
        float4 a, b, c, d;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:128\n\t"
                     "global_load_dwordx4 %2, %5, off\n\tglobal_load_dwordx4 %3, %5, off offset:128\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p), "v"(p + 320) : "memory");
        acc += a.x + b.y + c.z + d.w;
      } else if constexpr (WIDTH == 2) {
        float2 a, b, c, d;
        asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:128\n\t"
                     "global_load_dwordx2 %2, %5, off\n\tglobal_load_dwordx2 %3, %5, off offset:128\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p), "v"(p + 320) : "memory");
        acc += a.x + b.y + c.x + d.y;
      } else {
        float a, b, c, d;
        asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:128\n\t"
                     "global_load_dword %2, %5, off\n\tglobal_load_dword %3, %5, off offset:128\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p), "v"(p + 320) : "memory");
        acc += a + b + c + d;
      }
      g = (g + 8u) % NG;
    }
  }
  if (acc == 12345.678f) out[blockIdx.x * 64 + lane] = acc;
}

template <int WIDTH>
static void run(const char* name, const char* table, unsigned long long mask, float* out, int cus, double ghz) {
  const int blocks = cus * 32;  // 8 waves per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_load<WIDTH>), dim3(blocks), dim3(64), 0, 0, table, mask, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_load<WIDTH>), dim3(blocks), dim3(64), 0, 0, table, mask, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double loads_per_cu = 32.0 * ITERS * 4.0;
  printf("%-44s x%d: %7.3f ms  %6.2f cycles per load instruction per CU (at %.2f GHz)\n", name, WIDTH, ms, ms * 1e-3 * ghz * 1e9 / loads_per_cu, ghz);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus    = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  char* table; float* out;
  CK(hipMalloc(&table, NG * 320 + 1024));
  CK(hipMemset(table, 0, NG * 320 + 1024));
  CK(hipMalloc(&out, size_t(cus) * 32 * 64 * 4));
  printf("%s, %d CUs, %.2f GHz nominal\n", prop.gcnArchName, cus, ghz);
  struct { const char* name; unsigned long long mask; } pats[] = {
      {"all 64 lanes", ~0ull},
      {"lanes 0-31", 0xffffffffull},
      {"first 4 of every 8 (one quad of two)", 0x0f0f0f0f0f0f0f0full},
      {"first 3 of every 8 (children compacted)", 0x0707070707070707ull},
      {"3 of every 8, spread (0,3,6)", 0x4949494949494949ull},
      {"2 of every 8, one per quad (1,6)", 0x4242424242424242ull},
      {"1 of every 8", 0x0101010101010101ull},
      {"1 of every 4 (every quad has one lane)", 0x1111111111111111ull},
  };
  for (auto& p : pats) {
    run<4>(p.name, table, p.mask, out, cus, ghz);
    run<2>(p.name, table, p.mask, out, cus, ghz);
    run<1>(p.name, table, p.mask, out, cus, ghz);
  }
  return 0;
}
