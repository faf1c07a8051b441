// Storage-type traits and small helpers shared by the 16-bit translation units (lowp.hip, lowp_s1d.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

#define LP_F16 1
#define LP_BF16 2

// ---- storage-type traits: conversions are explicit, sums never happen in 16 bits ----
struct TF16 {
  typedef h16x8 frag;
  static __device__ __forceinline__ float ld(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
  static __device__ __forceinline__ unsigned short st(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // RNE
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
};
struct TBF16 {
  typedef b16x8 frag;
  static __device__ __forceinline__ float ld(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
  static __device__ __forceinline__ unsigned short st(float f) {   // round to nearest even: v_cvt_pk_bf16_f32 (NaN stays NaN)
    return __builtin_bit_cast(unsigned short, (__bf16)f);
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
  }
};
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b) {
  return (unsigned)T::st(a) | ((unsigned)T::st(b) << 16);
}
typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ unsigned pack2<TBF16>(float a, float b) {   // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{a, b}, b16x2));
}
template <typename T> __device__ __forceinline__ void unpack8(u32x4 v, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = T::ld((unsigned short)(v[i] & 0xffffu)); o[2 * i + 1] = T::ld((unsigned short)(v[i] >> 16)); }
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float (&o)[8]) {
  return u32x4{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7])};
}
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

// ---- weight packing: source addressing shared by every packed-image layout ----
struct LpPackParams {
  const float* w;
  unsigned short* wp;
  int ntaps, K, N, KS, NB;
  long sT, sK, sN;    // source strides of (tap, contraction index k, column n)
  int flip;           // tap t reads source tap ntaps-1-t (stride-1 data gradient)
  int cin_is_k;       // 1: the (possibly folded) input-channel axis is k (forward role), 0: it is n (data-gradient role)
  int shift, dup_start;
};
// source value of packed position (tap t, k, n) -- same conventions as the fp32 images (conv_igemm.hip: pack_src)
__device__ __forceinline__ float lp_pack_src(const LpPackParams& q, int t, int k, int n) {
  if (k >= q.K || n >= q.N) return 0.f;
  const int ts = q.flip ? (q.ntaps - 1 - t) : t;
  int kk = k, nn = n, k2 = -1, n2 = -1;
  if (q.cin_is_k == 1) {
    // slab channel c is reference channel c + shift; inside [dup_start, ...) it is ALSO reference channel c - dup_start
    // (encoder.py:83-87: [o_{j-1}, o_0 .. o_{j-1}] read once from the slab [o_0 .. o_{j-1}])
    if (k >= q.dup_start) k2 = k - q.dup_start;
    kk = k + q.shift;
    n2 = n;
  } else {
    if (n >= q.dup_start) n2 = n - q.dup_start;
    nn = n + q.shift;
    k2 = k;
  }
  float v = q.w[ts * q.sT + kk * q.sK + nn * q.sN];
  if (q.shift > 0 && k2 >= 0 && n2 >= 0 && k2 < (q.cin_is_k ? q.shift : q.K) && n2 < (q.cin_is_k ? q.N : q.shift))
    v += q.w[ts * q.sT + k2 * q.sK + n2 * q.sN];
  return v;
}

// element i of the LDS-DMA stage-order part of a K3S1 image: [cout group of CBW blocks][k-step][dz][dy*3+dx][k-half][cout in group][8 cin]
// (lowp_s1d.hip's pack kernel and lowp.hip's batched pack)
template <typename T> __device__ __forceinline__ void lp_s1d_pack_elem(const LpPackParams& p, int CBW, long i) {
  const int e = (int)(i & 7);
  long q = i >> 3;
  const int row = (int)(q % (32 * CBW)); q /= 32 * CBW;
  const int hh = (int)(q & 1); q >>= 1;
  const int t9 = (int)(q % 9); q /= 9;
  const int dz = (int)(q % 3); q /= 3;
  const int ks = (int)(q % p.KS);
  const int cg = (int)(q / p.KS);
  p.wp[i] = T::st(lp_pack_src(p, dz * 9 + t9, ks * 16 + hh * 8 + e, cg * CBW * 32 + row));
}


// 4 x 4 transpose of 16-byte pieces between the four lanes of a quad (lane bits 0-1) and four registers: afterwards register j of quad
// lane b holds what register b of quad lane j held.  Two butterfly stages of quad-permute DPP moves; its own inverse.
// What it is for: NDHWC rows of 64 channels are 128 bytes = 8 pieces.  The matrix instruction wants lane (voxel, k-half h) to hold piece
// 2 ks + h of ITS voxel in the register of k-step ks -- loaded that way, one instruction touches 32 bytes of each of 32 rows, four
// instructions per cache line, and the L1 request rate (not HBM) bounds the kernel at ~4 TB/s.  Loaded transposed -- instruction i:
// quad lane b fetches piece 2 b + h of voxel (quad base + i) -- an instruction covers 8 whole rows; the transpose then hands every lane
// its own voxel.  The same on the way out for 64-cout groups.
__device__ __forceinline__ void k1_quad_transpose(u32x4 (&r)[4], int b) {
  u32x4 s[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x4 t;
#pragma unroll
    for (int d = 0; d < 4; ++d) t[d] = (unsigned)__builtin_amdgcn_mov_dpp((int)r[j ^ 1][d], 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    const bool keep = ((b ^ j) & 1) == 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) s[j][d] = keep ? r[j][d] : t[d];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x4 t;
#pragma unroll
    for (int d = 0; d < 4; ++d) t[d] = (unsigned)__builtin_amdgcn_mov_dpp((int)s[j ^ 2][d], 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    const bool keep = ((b ^ j) & 2) == 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) r[j][d] = keep ? s[j][d] : t[d];
  }
}

// the 2 x 2 form: register j of pair lane b <-> register b of pair lane j (64-byte row chunks: two k-steps)
__device__ __forceinline__ void k1_pair_transpose(u32x4 (&r)[2], int b) {
  u32x4 s[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    u32x4 t;
#pragma unroll
    for (int d = 0; d < 4; ++d) t[d] = (unsigned)__builtin_amdgcn_mov_dpp((int)r[j ^ 1][d], 0xB1, 0xf, 0xf, true);
    const bool keep = ((b ^ j) & 1) == 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) s[j][d] = keep ? r[j][d] : t[d];
  }
  r[0] = s[0]; r[1] = s[1];
}

// Cache policy of the conv kernels' OUTPUT stores (the aux operand of buffer_store: bit 0 sc0, bit 1 nt, bit 4 sc1).  The outputs of a
// conv are not read again by the same launch, while its INPUT lines are (a k-step fetches 32 bytes of a 128-byte line, the other k-steps
// and the neighbouring tiles want the rest): stores that allocate in L2 could push those lines out.  Measured in round 5
// (profiles/r05_ab_e7_store_policy.txt; make variantf FILE=lowp_s1d NAME=st2 EXTRA=-DLP_OUT_STORE_AUX=2): non-temporal stores make the
// batch-8 step 4 % SLOWER (79.7 against 76.5 ms; sc0+nt and nt+sc1 the same or worse) and the inference forward 3 % -- the default stays 0.
#ifndef LP_OUT_STORE_AUX
#define LP_OUT_STORE_AUX 0
#endif

// GroupNorm-apply on the way IN: the conv reads the RAW output c of the previous conv and applies a = relu((c - mean) rstd gamma + beta)
// (group_norm.py:110-122 + the ReLU of resnet.py:133-136) to each input plane after it has landed in LDS -- inference only, where no
// backward needs the applied tensor: the 1 read + 1 write pass of bts_lp_gn_apply goes away.  Slab semantics with whole z planes per
// group (D % G == 0) and classes that tile a 16-byte slot (cg | 8): a slot's eight channels then have classes e mod cg whatever the slot.
// The arithmetic is bts_lp_gn_apply's, element for element (max(fmaf(v - mean, rstd * gamma, beta), 0), rounded to the storage type).
struct LpGnaFuse {
  const float* gamma;
  const float* beta;
  const float* mean;   // (N*G)
  const float* rstd;
  int G, cg;
};

// the arithmetic of the GNA transforms on one 16-byte slot: o = max(fmaf(v - mu, sc, be), 0) per element, rounded to the storage type --
// on the packed fp32 instructions (v_pk_add_f32 / v_pk_fma_f32: the same IEEE results as the scalar forms bts_lp_gn_apply uses)
template <typename T> __device__ __forceinline__ u32x4 lp_gna_slot(u32x4 raw, float mu, const float (&sc)[8], const float (&be)[8]) {
  const f32x2_ m2 = {mu, mu};
  u32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2_ v = {T::ld((unsigned short)(raw[k] & 0xffffu)), T::ld((unsigned short)(raw[k] >> 16))};
    const f32x2_ t = __builtin_elementwise_fma(v - m2, f32x2_{sc[2 * k], sc[2 * k + 1]}, f32x2_{be[2 * k], be[2 * k + 1]});
    r[k] = pack2<T>(fmaxf(t[0], 0.f), fmaxf(t[1], 0.f));
  }
  return r;
}

// GroupNorm-backward class sums from the epilogue of the data-gradient conv that PRODUCES the GroupNorm output's gradient (resnet.py:80-93
// under train.py:151: conv2^T(dc2) = da, then GN1 backward needs A_j = sum da_E * xh and B_j = sum da_E per (sample, group, class
// j = channel mod cg) before anything else -- lp_gn_bwd_reduce_kernel's pass over da and c1).  The epilogue holds da in registers; it
// reads the matching 16 bytes of the GroupNorm INPUT x per stored 16 bytes and leaves the class sums as per-(unit, slot) partial rows
// in lp_gn_bwd_finalize_kernel's layout [N*G][B][cg][2] (fp64).  Slab semantics (whole z planes per group: D % G == 0).
struct LpGnbFuse {
  const unsigned short* x;   // GroupNorm input, dense (N, D, H, W, C) with C = the conv's output channels here
  const float* gamma;
  const float* beta;
  const float* mean;         // [N*G]
  const float* rstd;
  double* part;              // [N*G][B][cg][2]
  int G, cg, relu;
  long B;
};
