// Shared device/host helpers for the gfx950 (MI355X, CDNA4) segmentation engine.
// Wave width is 64 everywhere; nothing here is portable to other targets on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Engine status codes returned through the C ABI (0 = ok, >0 = hipError_t).
#define BTS_OK 0
#define BTS_ERR_SHAPE (-1)
#define BTS_ERR_ALIGN (-2)
#define BTS_ERR_UNSUPPORTED (-3)
#define BTS_ERR_WORKSPACE (-4)

// Timing experiments that skip parts of a kernel (results are WRONG by design) exist only in builds made with
// -DBTS_TIMING_EXPERIMENTS; in the product library the switch is the literal 0 and the branches fold away.
#ifdef BTS_TIMING_EXPERIMENTS
#define BTS_DBG(p) ((p).dbg)
#else
#define BTS_DBG(p) 0
#endif

#define BTS_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int bts_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Sum over the 64 lanes of a wave, fixed order, the total returned in EVERY lane.  Cross-lane moves by DPP (quad permutes, row rotates,
// row_bcast15 / row_bcast31: register-to-register, a few cycles each) instead of __shfl_down's ds_bpermute round trips through the LDS
// crossbar (two per step for a double, ~100 cycles each, and the conv kernels' epilogues take four of these sums per item with the
// matrix pipe idle).
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_move_f64(double x) {      // lanes the control leaves out (masked rows) receive 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWS, 0xf, false);
  return __hiloint2double(hi, lo);
}
// PRECONDITION: all 64 lanes of the wave are active (EXEC = ~0) -- DPP leaves 0 where the source lane is inactive and the total is read
// from lane 63, so a partial wave would return a partial (or zero) sum in silence.  Every caller is wave-uniform (lanes that have nothing
// to add pass 0); -DBTS_DEBUG_FULL_WAVE turns the contract into a trap.  lane_group_sum_f32 below: whole groups active, likewise.
__device__ __forceinline__ void bts_assert_full_wave() {
#ifdef BTS_DEBUG_FULL_WAVE
  if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  bts_assert_full_wave();
#ifdef BTS_WAVE_SUM_SHFL      // A/B builds only (make alt NAME=shfl EXTRA=-DBTS_WAVE_SUM_SHFL): the ds_bpermute form
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return __shfl(v, 0, 64);
#endif
  v += dpp_move_f64<0xB1, 0xf>(v);       // quad_perm [1,0,3,2]
  v += dpp_move_f64<0x4E, 0xf>(v);       // quad_perm [2,3,0,1]: every lane holds its quad's sum
  v += dpp_move_f64<0x124, 0xf>(v);      // row_ror:4
  v += dpp_move_f64<0x128, 0xf>(v);      // row_ror:8: every lane holds its row's (16 lanes) sum
  v += dpp_move_f64<0x142, 0xa>(v);      // row_bcast15 into rows 1 and 3
  v += dpp_move_f64<0x143, 0xc>(v);      // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// Sum over aligned groups of n ADJACENT lanes (n a power of two <= 32), the result in every lane of the group -- the per-voxel dot
// products of the element-wise kernels (a voxel's C / 8 lanes sit side by side).  DPP moves for the steps inside a row of 16 (quad
// permutes, then row_half_mirror / row_mirror: after the quad steps every lane of a quad holds the same bits, so the mirrored partner
// is as good as the xor partner) instead of __shfl_xor's ds_bpermute round trips; bit-identical to the xor butterfly (fp add commutes).
template <int CTRL> __device__ __forceinline__ float dpp_move_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_group_sum_f32(float v, int n) {
#ifdef BTS_GROUP_SUM_SHFL      // A/B builds only
  for (int m = 1; m < n; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
#endif
  if (n >= 2) v += dpp_move_f32<0xB1>(v);
  if (n >= 4) v += dpp_move_f32<0x4E>(v);
  if (n >= 8) v += dpp_move_f32<0x141>(v);
  if (n >= 16) v += dpp_move_f32<0x140>(v);
  if (n >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Block-wide (256 threads = 4 waves) fixed-order sum; result valid in thread 0.
__device__ __forceinline__ double block_sum_f64(double v, double* sh /*>=4*/) {
  v = wave_sum_f64(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sh[i];
  }
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// counter-based uniform generator of the dropout mask / the VAE's eps (pointwise.hip) and of lowp.hip's fused dropout + cast: element i of a
// stream `seed` is a pure function of (seed, i), so every kernel that draws element i gets the same value
__device__ __forceinline__ uint32_t mix32(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float u01(uint64_t seed, uint64_t i) {  // [0,1)
  return (float)(mix32(seed * 0xD1342543DE82EF95ull + i) >> 8) * (1.0f / 16777216.0f);
}
