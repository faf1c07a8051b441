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

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
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
