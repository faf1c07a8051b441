// Keras (optimizer_v2) Adam as used by ScheduledOptim (util.py:60-84), fused over the flat parameter buffer:
//   m <- b1*m + (1-b1)*g ; v <- b2*v + (1-b2)*g^2 ; w <- w - lr_t * m / (sqrt(v) + eps)
//   lr_t = lr * sqrt(1-b2^t)/(1-b1^t) is computed on the host in fp64 (epsilon is NOT bias-corrected: this is the TF
//   form, which differs from torch.optim.Adam). One pass, 4 streams x 16 B per lane: HBM-bound.
#include "common.h"
#include "bts_internal.h"
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n4, long n, float lr_t, float b1, float b2,
                                                   float eps, float gmul) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * gmul;
      mm[e] = b1 * mm[e] + (1.f - b1) * ge;
      vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
      pp[e] = pp[e] - lr_t * mm[e] / (sqrtf(vv[e]) + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // tail
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float ge = g[i] * gmul;
      const float mm = b1 * m[i] + (1.f - b1) * ge;
      const float vv = b2 * v[i] + (1.f - b2) * ge * ge;
      m[i] = mm; v[i] = vv;
      p[i] = p[i] - lr_t * mm / (sqrtf(vv) + eps);
    }
}

// the same update, skipped ON THE DEVICE when *skip != 0 (the 16-bit trainer's overflow flag, bts_grad_nonfinite below): no host
// round trip between the gradient exchange and the optimiser; every workgroup reads the same word before it touches anything
__global__ __launch_bounds__(256) void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, long n4, long n, float lr_t, float b1, float b2,
                                                           float eps, float gmul, const int* __restrict__ skip) {
  if (*skip != 0) return;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * gmul;
      mm[e] = b1 * mm[e] + (1.f - b1) * ge;
      vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
      pp[e] = pp[e] - lr_t * mm[e] / (sqrtf(vv[e]) + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float ge = g[i] * gmul;
      const float mm = b1 * m[i] + (1.f - b1) * ge;
      const float vv = b2 * v[i] + (1.f - b2) * ge * ge;
      m[i] = mm; v[i] = vv;
      p[i] = p[i] - lr_t * mm / (sqrtf(vv) + eps);
    }
}

// flag[0] = 1 when any of g[0..n) is Inf or NaN, else 0 (integer OR: order-independent).  One read pass, HBM-bound.
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long n4, long n, int* __restrict__ flag) {
  int bad = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const u32x4 w = reinterpret_cast<const u32x4*>(g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) bad |= ((w[e] & 0x7f800000u) == 0x7f800000u) ? 1 : 0;     // exponent all ones: Inf or NaN
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) bad |= ((__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u) ? 1 : 0;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
extern "C" int bts_grad_nonfinite(const float* g, long n, int* flag, hipStream_t stream) {
  if (n <= 0 || flag == nullptr) return BTS_ERR_SHAPE;
  if (((uintptr_t)g) & 15) return BTS_ERR_ALIGN;
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), stream);
  if (e != hipSuccess) return (int)e;
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  (void)hipGetLastError(); hipLaunchKernelGGL(grad_nonfinite_kernel, dim3((int)blocks), dim3(256), 0, stream, g, n4, n, flag);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
extern "C" int bts_adam_tf_step_guarded(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                                        float eps, float gmul, const int* skip, hipStream_t stream) {
  if (n <= 0 || skip == nullptr) return BTS_ERR_SHAPE;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return BTS_ERR_ALIGN;
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError(); hipLaunchKernelGGL(adam_guarded_kernel, dim3((int)blocks), dim3(256), 0, stream, p, g, m, v, n4, n, lr_t, beta1, beta2, eps, gmul, skip);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// gmul scales the gradient first (1/world_size after a summing all-reduce; 1 otherwise)
extern "C" int bts_adam_tf_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                                float eps, float gmul, hipStream_t stream) {
  if (n <= 0) return BTS_ERR_SHAPE;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return BTS_ERR_ALIGN;
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError(); hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, stream, p, g, m, v, n4, n, lr_t, beta1, beta2, eps, gmul);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
