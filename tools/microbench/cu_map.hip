// Which CUs, SIMDs and wave slots the one-wave blocks of a long launch land on (s_getreg HW_ID / XCC_ID), and how many of them are
// alive at once per CU.   hipcc --offload-arch=gfx950 -O2 -w tools/microbench/cu_map.hip -o tools/microbench/cu_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

template <int BS, int TOP = 0>
__global__ __launch_bounds__(BS) void spin(unsigned long long* t, unsigned* id, int iters) {
  unsigned long long t0 = wall_clock64();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  if constexpr (TOP == 71) asm volatile("s_mov_b32 s71, 0" ::: "s71");
  if constexpr (TOP == 72) asm volatile("s_mov_b32 s72, 0" ::: "s72");
  if constexpr (TOP == 73) asm volatile("s_mov_b32 s73, 0" ::: "s73");
  if constexpr (TOP == 74) asm volatile("s_mov_b32 s74, 0" ::: "s74");
  if constexpr (TOP == 79) asm volatile("s_mov_b32 s79, 0" ::: "s79");
  if constexpr (TOP == 80) asm volatile("s_mov_b32 s80, 0" ::: "s80");
  if constexpr (TOP == 87) asm volatile("s_mov_b32 s87, 0" ::: "s87");
  if constexpr (TOP == 88) asm volatile("s_mov_b32 s88, 0" ::: "s88");
  if constexpr (TOP == 95) asm volatile("s_mov_b32 s95, 0" ::: "s95");
  if constexpr (TOP == 101) asm volatile("s_mov_b32 s101, 0" ::: "s101");
  if (threadIdx.x == 0) {
    t[2 * blockIdx.x]     = t0;
    t[2 * blockIdx.x + 1] = wall_clock64() + (v == 123.f);
    id[2 * blockIdx.x]     = hw;
    id[2 * blockIdx.x + 1] = xcc;
  }
}

template <int BS, int TOP = 0>
void run() {
  const int blocks = 16384 * 64 / BS;
  unsigned long long* d;
  unsigned* di;
  hipMalloc(&d, 2 * blocks * sizeof(unsigned long long));
  hipMalloc(&di, 2 * blocks * sizeof(unsigned));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((spin<BS, TOP>), dim3(blocks), dim3(BS), 0, 0, d, di, 400000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(2 * blocks);
  std::vector<unsigned> id(2 * blocks);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  hipMemcpy(id.data(), di, id.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int b = 0; b < blocks; ++b) lo = std::min(lo, h[2 * b]), hi = std::max(hi, h[2 * b + 1]);
  // the sample time with the most blocks alive
  int best = 0;
  unsigned long long bestx = lo;
  for (int s = 1; s < 80; ++s) {
    const unsigned long long x = lo + (hi - lo) * s / 80;
    int alive = 0;
    for (int b = 0; b < blocks; ++b) alive += h[2 * b] <= x && h[2 * b + 1] > x;
    if (alive > best) best = alive, bestx = x;
  }
  std::map<unsigned, int> per_cu, per_xcc, wave_ids, per_simd;
  for (int b = 0; b < blocks; ++b) {
    if (!(h[2 * b] <= bestx && h[2 * b + 1] > bestx)) continue;
    const unsigned hw = id[2 * b], xcc = id[2 * b + 1] & 15u;
    const unsigned wave = hw & 15u, simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    per_simd[(xcc << 18) | (se << 10) | (sh << 6) | (cu << 2) | simd]++;
    per_xcc[xcc]++;
    wave_ids[wave]++;
  }
  std::map<int, int> hist, hist_simd;
  for (auto& kv : per_cu) hist[kv.second]++;
  for (auto& kv : per_simd) hist_simd[kv.second]++;
  printf("block size %d, highest SGPR named s%d: most blocks alive at once %d on %zu distinct CUs (%zu SIMDs)\n  blocks per CU: ", BS, TOP, best, per_cu.size(), per_simd.size());
  for (auto& kv : hist) printf("%d CUs hold %d;  ", kv.second, kv.first);
  printf("\n  blocks per SIMD: ");
  for (auto& kv : hist_simd) printf("%d SIMDs hold %d;  ", kv.second, kv.first);
  printf("\n  per XCC: ");
  for (auto& kv : per_xcc) printf("xcc %u: %d  ", kv.first, kv.second);
  printf("\n  wave slot ids in use: ");
  for (auto& kv : wave_ids) printf("%u: %d  ", kv.first, kv.second);
  printf("\n");
  hipFree(d);
  hipFree(di);
}

int main() {
  run<64>();
  run<64, 71>();
  run<64, 72>();
  run<64, 73>();
  run<64, 74>();
  run<64, 79>();
  run<64, 80>();
  run<64, 87>();
  run<64, 88>();
  run<64, 95>();
  run<64, 101>();
  run<512>();
  return 0;
}
