#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_rot(int* out) {
  int v = threadIdx.x;
  int r1 = __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, false);  // wave_ror:1
  int r2 = __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, false);  // wave_rol:1
  out[threadIdx.x] = r1; out[64 + threadIdx.x] = r2;
}
constexpr int ITERS = 65536;
template <int OP> __global__ void k_rate(double* out, double b) {
  double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2;
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (OP == 0) {  // 6 dpp movs (3 f64 rotated) + 3 dependent fma
      long long a0 = __builtin_bit_cast(long long, r0), a1 = __builtin_bit_cast(long long, r1), a2 = __builtin_bit_cast(long long, r2);
      int l0 = __builtin_amdgcn_update_dpp(0, (int)a0, 0x13C, 0xf, 0xf, false), h0 = __builtin_amdgcn_update_dpp(0, (int)(a0 >> 32), 0x13C, 0xf, 0xf, false);
      int l1 = __builtin_amdgcn_update_dpp(0, (int)a1, 0x13C, 0xf, 0xf, false), h1 = __builtin_amdgcn_update_dpp(0, (int)(a1 >> 32), 0x13C, 0xf, 0xf, false);
      int l2 = __builtin_amdgcn_update_dpp(0, (int)a2, 0x13C, 0xf, 0xf, false), h2 = __builtin_amdgcn_update_dpp(0, (int)(a2 >> 32), 0x13C, 0xf, 0xf, false);
      r0 = __builtin_bit_cast(double, ((long long)h0 << 32) | (unsigned)l0);
      r1 = __builtin_bit_cast(double, ((long long)h1 << 32) | (unsigned)l1);
      r2 = __builtin_bit_cast(double, ((long long)h2 << 32) | (unsigned)l2);
      r0 = __builtin_fma(r0, b, 1e-9); r1 = __builtin_fma(r1, b, 1e-9); r2 = __builtin_fma(r2, b, 1e-9);
    } else if constexpr (OP == 1) {  // 3 fma only
      r0 = __builtin_fma(r0, b, 1e-9); r1 = __builtin_fma(r1, b, 1e-9); r2 = __builtin_fma(r2, b, 1e-9);
    } else {  // 3 f64 rotated through ds_bpermute (6 of them) + 3 fma
      int src = (((threadIdx.x + 1) & 63)) << 2;
      long long a0 = __builtin_bit_cast(long long, r0), a1 = __builtin_bit_cast(long long, r1), a2 = __builtin_bit_cast(long long, r2);
      int l0 = __builtin_amdgcn_ds_bpermute(src, (int)a0), h0 = __builtin_amdgcn_ds_bpermute(src, (int)(a0 >> 32));
      int l1 = __builtin_amdgcn_ds_bpermute(src, (int)a1), h1 = __builtin_amdgcn_ds_bpermute(src, (int)(a1 >> 32));
      int l2 = __builtin_amdgcn_ds_bpermute(src, (int)a2), h2 = __builtin_amdgcn_ds_bpermute(src, (int)(a2 >> 32));
      r0 = __builtin_bit_cast(double, ((long long)h0 << 32) | (unsigned)l0);
      r1 = __builtin_bit_cast(double, ((long long)h1 << 32) | (unsigned)l1);
      r2 = __builtin_bit_cast(double, ((long long)h2 << 32) | (unsigned)l2);
      r0 = __builtin_fma(r0, b, 1e-9); r1 = __builtin_fma(r1, b, 1e-9); r2 = __builtin_fma(r2, b, 1e-9);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2;
}
template <int OP> void run(const char* name) {
  double* out; hipMalloc(&out, 256 * 1024 * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<OP>, dim3(256 * 2), dim3(1024), 0, 0, out, 0.999999);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<OP>, dim3(256 * 2), dim3(1024), 0, 0, out, 0.999999);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // 8 waves/SIMD; per SIMD iterations = ITERS*8
  printf("%-40s %.3f ms -> %.2f ns per iteration per SIMD (= %.1f cycles @2.3GHz)\n", name, ms, ms * 1e6 / (ITERS * 8.0), ms * 1e6 / (ITERS * 8.0) * 2.3);
}
int main() {
  int* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k_rot, dim3(1), dim3(64), 0, 0, d);
  std::vector<int> h(128); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  printf("wave_ror:1 lane0..3 = %d %d %d %d ... lane63 = %d\n", h[0], h[1], h[2], h[3], h[63]);
  printf("wave_rol:1 lane0..3 = %d %d %d %d ... lane63 = %d\n", h[64], h[65], h[66], h[67], h[127]);
  run<1>("3 fma_f64");
  run<0>("6 dpp wave_ror + 3 fma_f64");
  run<2>("6 ds_bpermute + 3 fma_f64");
}
