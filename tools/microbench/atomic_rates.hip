// Throughput of device-scope float atomic adds (no return) from every CU into a small array that lives in L2 / MALL: what a
// pair-symmetric K2 would need (one add per source, component and 16 pairs).  hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics
// tools/microbench/atomic_rates.hip -o tools/microbench/atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>

// every wave walks the array in steps of 64 bodies; lane l adds to component k of body (base + l): the AoS pattern a[j * 3 + k]
template <int K>
__global__ __launch_bounds__(256) void hammer(float* a, unsigned n, int iters, int fma_per_atomic) {
  const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  unsigned j = (wave * 2654435761u) % n;
  float v = threadIdx.x * 1e-9f, w = 1.0f;
  for (int it = 0; it < iters; ++it) {
    for (int f = 0; f < fma_per_atomic; ++f) w = __builtin_fmaf(w, 1.0000001f, v);  // the pairs' arithmetic between two flushes
    const unsigned body = (j + lane) % n;
#pragma unroll
    for (int k = 0; k < K; ++k) atomicAdd(&a[body * 3u + k], w * 1e-30f);
    j = (j + 64u * 977u) % n;
  }
}

int main() {
  const unsigned n = 262144;
  float* a;
  hipMalloc(&a, n * 3 * sizeof(float));
  hipMemset(a, 0, n * 3 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int fma : {0, 64, 256, 1024}) {
    const int blocks = 4096, iters = 2000;
    hipLaunchKernelGGL(hammer<3>, dim3(blocks), dim3(256), 0, 0, a, n, 10, fma);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(hammer<3>, dim3(blocks), dim3(256), 0, 0, a, n, iters, fma);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double adds = double(blocks) * 256 * iters * 3;
    printf("%5d FMAs between flushes: %.3f ms, %.3e float atomic adds/s (%.3e wave-level instructions/s)\n", fma, ms, adds / (ms * 1e-3),
           adds / 64 / (ms * 1e-3));
  }
  return 0;
}
