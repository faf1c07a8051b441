// GPU-side training augmentation (reference train.py:14-49, a tf.data map on the host).
// Per example: per-channel intensity shift/scale driven by the volume's per-channel standard deviation, a random
// crop_size window of the (x,y) pair, per-axis flips, and one-hot labels without the background channel.
// The random draws are arguments (the TF stream cannot be reproduced); given the draws the arithmetic is the
// reference's: x <- (x + shift*sqrt(var)) * scale in fp32, var = population variance over the whole volume.
#include "common.h"
#include "bts_internal.h"

#define AUG_MAXC 16
#define AUG_BLOCKS 512

// per-block fp64 partial sums of x and x^2 for every channel; combined in a fixed order by the finalize kernel
__global__ __launch_bounds__(256) void moments_partial_kernel(const float* __restrict__ x, double* part, long nvox, int C,
                                                              int ld) {
  __shared__ double sh[4];
  double s[AUG_MAXC], q[AUG_MAXC];
#pragma unroll
  for (int c = 0; c < AUG_MAXC; ++c) { s[c] = 0.0; q[c] = 0.0; }
  for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < nvox; v += (long)gridDim.x * blockDim.x) {
    const float* row = x + v * ld;
#pragma unroll
    for (int c = 0; c < AUG_MAXC; ++c)
      if (c < C) { const double t = (double)row[c]; s[c] += t; q[c] += t * t; }
  }
#pragma unroll
  for (int c = 0; c < AUG_MAXC; ++c) {
    if (c < C) {
      const double a = block_sum_f64(s[c], sh);
      const double b = block_sum_f64(q[c], sh);
      if (threadIdx.x == 0) { part[((long)blockIdx.x * C + c) * 2 + 0] = a; part[((long)blockIdx.x * C + c) * 2 + 1] = b; }
    }
  }
}
__global__ void moments_finalize_kernel(const double* part, float* mean, float* var, int nblocks, int C, long nvox) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < nblocks; ++b) { s += part[((long)b * C + c) * 2 + 0]; q += part[((long)b * C + c) * 2 + 1]; }
  const double m = s / (double)nvox;
  double vv = q / (double)nvox - m * m;  // population variance (tf.nn.moments)
  if (vv < 0.0) vv = 0.0;
  if (mean) mean[c] = (float)m;
  var[c] = (float)vv;
}

extern "C" long bts_channel_moments_workspace(int C) { return (long)AUG_BLOCKS * C * 2 * sizeof(double); }

extern "C" int bts_channel_moments(const float* x, float* mean, float* var, void* workspace, long workspace_bytes, long nvox,
                                   int C, int ld, hipStream_t stream) {
  if (nvox <= 0 || C <= 0 || C > AUG_MAXC || ld < C) return BTS_ERR_SHAPE;
  if (workspace == nullptr || workspace_bytes < bts_channel_moments_workspace(C)) return BTS_ERR_WORKSPACE;
  long blocks = (nvox + 255) / 256;
  if (blocks > AUG_BLOCKS) blocks = AUG_BLOCKS;
  double* part = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError(); hipLaunchKernelGGL(moments_partial_kernel, dim3((int)blocks), dim3(256), 0, stream, x, part, nvox, C, ld);
  BTS_LAUNCH_CHECK();
  (void)hipGetLastError(); hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(64), 0, stream, part, mean, var, (int)blocks, C, nvox);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

struct AugParams {
  const float* x;    // (S0,S1,S2,C) source volume
  const float* y;    // (S0,S1,S2) labels stored as floats
  const float* var;  // C: per-channel population variance of the source volume
  float* xo;         // (T0,T1,T2,C)
  float* yo;         // (T0,T1,T2,out_ch)
  int S0, S1, S2, C, T0, T1, T2, o0, o1, o2, flip, out_ch;
  float shift[AUG_MAXC], scale[AUG_MAXC];
};
// one thread per output voxel: crop window [o, o+T) then tf.reverse on the flagged axes (bit 4/2/1 = axis 0/1/2)
__global__ void augment_kernel(const AugParams p) {
  const long total = (long)p.T0 * p.T1 * p.T2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long v = i;
    const int t2 = (int)(v % p.T2); v /= p.T2;
    const int t1 = (int)(v % p.T1);
    const int t0 = (int)(v / p.T1);
    const int c0 = (p.flip & 4) ? p.T0 - 1 - t0 : t0, c1 = (p.flip & 2) ? p.T1 - 1 - t1 : t1, c2 = (p.flip & 1) ? p.T2 - 1 - t2 : t2;
    const long src = ((long)(p.o0 + c0) * p.S1 + (p.o1 + c1)) * p.S2 + (p.o2 + c2);
    for (int c = 0; c < p.C; ++c) {
      float val = p.x[src * p.C + c];
      val += p.shift[c] * sqrtf(p.var[c]);   // train.py:19-21: x += shift * sqrt(var)
      val *= p.scale[c];                     // train.py:22
      p.xo[i * p.C + c] = val;
    }
    const int lbl = (int)p.y[src];           // tf.cast(y, tf.int32) truncates (train.py:38)
    for (int k = 0; k < p.out_ch; ++k) p.yo[i * p.out_ch + k] = (lbl == k + 1) ? 1.f : 0.f;  // one_hot(out_ch+1) minus channel 0
  }
}

extern "C" int bts_augment_crop(const float* x, const float* y, const float* var, float* xo, float* yo, int S0, int S1, int S2,
                                int C, int T0, int T1, int T2, int o0, int o1, int o2, int flip_mask, const float* shift,
                                const float* scale, int out_ch, hipStream_t stream) {
  if (C <= 0 || C > AUG_MAXC || out_ch <= 0 || T0 <= 0 || T1 <= 0 || T2 <= 0 || (flip_mask & ~7)) return BTS_ERR_SHAPE;
  if (o0 < 0 || o1 < 0 || o2 < 0 || o0 + T0 > S0 || o1 + T1 > S1 || o2 + T2 > S2) return BTS_ERR_SHAPE;
  AugParams p;
  p.x = x; p.y = y; p.var = var; p.xo = xo; p.yo = yo;
  p.S0 = S0; p.S1 = S1; p.S2 = S2; p.C = C; p.T0 = T0; p.T1 = T1; p.T2 = T2; p.o0 = o0; p.o1 = o1; p.o2 = o2;
  p.flip = flip_mask; p.out_ch = out_ch;
  for (int c = 0; c < AUG_MAXC; ++c) { p.shift[c] = c < C ? shift[c] : 0.f; p.scale[c] = c < C ? scale[c] : 1.f; }  // host arrays
  const long total = (long)T0 * T1 * T2;
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError(); hipLaunchKernelGGL(augment_kernel, dim3((int)blocks), dim3(256), 0, stream, p);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
