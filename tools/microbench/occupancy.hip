// How many one-wave workgroups run at once on the chip, by the highest SGPR the kernel touches (gfx950: 800 SGPRs per SIMD,
// allocated in blocks of 16 — but how many does a wave need beyond the ones it names?).  Every block spins ~200 us and records
// its start and end (s_memrealtime); the host counts the blocks alive at the middle of the launch.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/occupancy.hip -o tools/microbench/occupancy && tools/microbench/occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int TOP, int BS = 64>
__global__ __launch_bounds__(BS) void spin(unsigned long long* t, int iters) {
  unsigned long long t0 = wall_clock64();
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  if constexpr (TOP == 47) asm volatile("s_mov_b32 s47, 0" ::: "s47");
  if constexpr (TOP == 63) asm volatile("s_mov_b32 s63, 0" ::: "s63");
  if constexpr (TOP == 71) asm volatile("s_mov_b32 s71, 0" ::: "s71");
  if constexpr (TOP == 79) asm volatile("s_mov_b32 s79, 0" ::: "s79");
  if constexpr (TOP == 87) asm volatile("s_mov_b32 s87, 0" ::: "s87");
  if constexpr (TOP == 95) asm volatile("s_mov_b32 s95, 0" ::: "s95");
  if constexpr (TOP == 101) asm volatile("s_mov_b32 s101, 0" ::: "s101");
  if (threadIdx.x == 0) {
    t[2 * blockIdx.x]     = t0;
    t[2 * blockIdx.x + 1] = wall_clock64() + (v == 123.f);
  }
}

template <int TOP, int BS = 64>
void run(const char* label) {
  const int blocks = 16384;
  unsigned long long* d;
  hipMalloc(&d, 2 * blocks * sizeof(unsigned long long));
  hipLaunchKernelGGL((spin<TOP, BS>), dim3(blocks), dim3(BS), 0, 0, d, 400000);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((spin<TOP, BS>), dim3(blocks), dim3(BS), 0, 0, d, 400000);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int b = 0; b < blocks; ++b) lo = std::min(lo, h[2 * b]), hi = std::max(hi, h[2 * b + 1]);
  int best = 0;
  for (int s = 1; s < 40; ++s) {
    const unsigned long long x = lo + (hi - lo) * s / 40;
    int alive = 0;
    for (int b = 0; b < blocks; ++b) alive += h[2 * b] <= x && h[2 * b + 1] > x;
    best = std::max(best, alive);
  }
  std::vector<unsigned long long> durs(blocks);
  for (int b = 0; b < blocks; ++b) durs[b] = h[2 * b + 1] - h[2 * b];
  std::sort(durs.begin(), durs.end());
  std::vector<unsigned long long> starts(blocks);
  for (int b = 0; b < blocks; ++b) starts[b] = h[2 * b] - lo;
  std::sort(starts.begin(), starts.end());
  printf("    span %.1f us, block duration median %.1f us (min %.1f, max %.1f); block number 4096 / 8192 / 12288 started at %.1f / %.1f / %.1f us\n",
         (hi - lo) * 0.01, durs[blocks / 2] * 0.01, durs[0] * 0.01, durs[blocks - 1] * 0.01, starts[4096] * 0.01, starts[blocks / 2] * 0.01, starts[3 * blocks / 4] * 0.01);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spin<TOP, BS>));
  printf("%-34s VGPRs %3d  most blocks alive at once: %5d = %.2f waves per SIMD\n", label, fa.numRegs, best, best * (BS / 64) / 1024.0);
  hipFree(d);
}

int main() {
  run<0>("no SGPR named");
  run<47>("touches s47");
  run<63>("touches s63");
  run<71>("touches s71");
  run<79>("touches s79");
  run<87>("touches s87");
  run<95>("touches s95");
  run<101>("touches s101");
  run<0, 128>("128 threads, no SGPR named");
  run<87, 128>("128 threads, touches s87");
  run<0, 256>("256 threads, no SGPR named");
  run<87, 256>("256 threads, touches s87");
  run<0, 512>("512 threads, no SGPR named");
  return 0;
}
