// Device helpers shared by the Winograd-form kernels (conv_wino.hip, conv_wino3.hip): buffer loads, packed fp32 adds, the
// 1-D input transform, one group of 16 matrix instructions, pinned accumulator reads.
#pragma once
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// packed fp32 add / subtract (the compiler selects scalar v_sub_f32 for vector subtraction; every vector-ALU instruction
// issued next to the matrix pipe costs ~9 cycles of a single-wave SIMD, so halving their number is worth the asm)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_add(a.xy, b.xy), hi = pk_add(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
// y transform of one z-combined row set
__device__ __forceinline__ void wino_yt(const f32x4 (&c)[4], f32x4 (&v)[4]) {
#ifdef W3_EXP_NOXF    // timing experiment (wrong results): no transforms
  return;
#endif
  v[0] = sub4(c[0], c[2]);
  v[1] = add4(c[1], c[2]);
  v[2] = sub4(c[2], c[1]);
  v[3] = sub4(c[1], c[3]);
}
// one group: xi_z fixed, 4 xi_y values, 4 channel pairs -> 16 MFMAs (A = U rows = couts, B = transformed input columns = x)
// ZERO: the accumulators are not read by their first matrix instruction (C = literal 0) -- the first group of each xi_z in an
// item's first stage starts the sums this way, so the 256 accumulation registers are never cleared by separate writes
template <bool ZERO>
__device__ __forceinline__ void wino_mfma16(const f32x4 (&v)[4], const f32x4 (&a)[4], f32x16 (&acc)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ZERO && j == 0) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e][j], v[e][j], z, 0, 0, 0);
      } else {
        acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e][j], v[e][j], acc[e], 0, 0, 0);
      }
    }
}
// accumulator element -> vector register, pinned in program order (volatile): the compiler otherwise reads all 256
// accumulation registers at the loop exit and spills what does not fit
__device__ __forceinline__ float acc_rd(float a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
// after every MFMA: room for two vector-ALU operations and one memory request of the group
#define WINO_SCHED_GROUP()                                 \
  _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);     \
  }

