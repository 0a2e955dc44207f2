// Wave-uniform values that the vector unit holds, moved to SGPRs for the "s" operands of the hand-written ISA blocks.
//
// __builtin_amdgcn_readfirstlane is not used for this: the optimizer folds a readfirstlane of a value it can prove uniform back
// into the VGPR computation and then cannot satisfy the "s" constraint.  An asm v_readfirstlane_b32 is opaque to it — and also
// to hipcc's hazard recognizer, which pads its own instructions but does not look inside an asm block.  gfx940-class parts
// (gfx950 included) need, and the block therefore carries itself:
//   * 1 wait state between a VALU write of a VGPR and a v_readlane / v_readfirstlane of it (LLVM: VALUWriteVGPRReadlaneRead);
//     without it the lane read returns the register's PREVIOUS content when the two instructions are adjacent — measured in round 3:
//     ot_force_isa_f32_kernel<2, false> received theta in place of the root cell's side (wrong, deterministically), the double 2D
//     instantiation a stale word some of the time (3-7 flipped decisions per 200 000 bodies, different ones on every walk);
//   * 4 wait states between a VALU write of EXEC (v_cmpx) and a lane read (VALUWriteEXECRWLane) — hipcc rarely writes EXEC that
//     way, but the s_nop in front is sized for it;
//   * 2 wait states between a VALU write of an SGPR and a VALU read of it, 5 before a VMEM read of it (VALUWriteSGPRVALURead,
//     VALUWriteSGPRVMEMRead): the s_nop after the read covers whatever the compiler schedules next.
#pragma once
#include <cstdint>

namespace nbody {

__device__ __forceinline__ uint32_t to_sgpr(uint32_t v) {
  uint32_t r;
  asm volatile("s_nop 3\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ int to_sgpr(int v) { return int(to_sgpr(uint32_t(v))); }
__device__ __forceinline__ float to_sgpr(float v) { return __builtin_bit_cast(float, to_sgpr(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ double to_sgpr(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const uint32_t lo = to_sgpr(uint32_t(b)), hi = to_sgpr(uint32_t(b >> 32));
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}

}  // namespace nbody
