// Bodies of two finalize kernels that exist on their own (groupnorm.hip, se.hip) and as the two roles of ONE launch in the fused block
// backward (block_bwd.hip: both only need the reduce pass's partials, and each costs a ~5 us launch on the main stream's chain).
#pragma once
#include "common.h"

// slab-mode GroupNorm backward finalize for group g: class sums over the blocks -> c1, c2 per sample; over the samples -> dgamma, dbeta.
// One 256-thread workgroup; sh: 512 doubles of LDS.  cg = C / G is a power of two <= 256.
__device__ __forceinline__ void gn_bwd_finalize_slab_body(const double* partial, const float* gamma, float* dgamma, float* dbeta, float* c1,
                                                          float* c2, int N, int G, int B, int cg, double L, int accum, int g, double* sh) {
  const int j = threadIdx.x % cg, sl = threadIdx.x / cg, S = 256 / cg;
  double ga = 0.0, gb = 0.0;
  for (int n = 0; n < N; ++n) {
    const long unit = (long)n * G + g;
    double sa = 0.0, sb = 0.0;
    for (int b = sl; b < B; b += S) {
      const double* o = partial + ((unit * B + b) * cg + j) * 2;
      sa += o[0]; sb += o[1];
    }
    __syncthreads();
    sh[threadIdx.x * 2] = sa; sh[threadIdx.x * 2 + 1] = sb;
    __syncthreads();
    if (sl == 0) {
      sa = 0.0; sb = 0.0;
      for (int s2 = 0; s2 < S; ++s2) { sa += sh[(s2 * cg + j) * 2]; sb += sh[(s2 * cg + j) * 2 + 1]; }
      ga += sa; gb += sb;
    }
    __syncthreads();
    if (sl == 0) {
      sh[j * 2] = (double)gamma[g * cg + j] * sb;       // -> c1
      sh[j * 2 + 1] = (double)gamma[g * cg + j] * sa;   // -> c2
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < cg; ++q) { s1 += sh[q * 2]; s2 += sh[q * 2 + 1]; }
      c1[unit] = (float)(s1 / L);
      c2[unit] = (float)(s2 / L);
    }
  }
  if (sl == 0) {
    const int idx = g * cg + j;
    if (dgamma) dgamma[idx] = accum ? dgamma[idx] + (float)ga : (float)ga;
    if (dbeta) dbeta[idx] = accum ? dbeta[idx] + (float)gb : (float)gb;
  }
}

// gate backward stage 2a for the four (n, c) pairs of workgroup `blk`: one wave per pair sums the per-block partials
__device__ __forceinline__ void se_bwd_partial_reduce_body(const double* partial, double* red, int N, int B, int F, int blk) {
  const int lane = threadIdx.x & 63;
  const int i = blk * 4 + (threadIdx.x >> 6);
  if (i >= N * F) return;
  const int n = i / F, c = i % F;
  double a = 0.0, b = 0.0;
  const double2* pp = reinterpret_cast<const double2*>(partial) + (long)n * B * F + c;      // pair (n, k, c) at pp[k * F]
  int k = lane;
  for (; k + 7 * 64 < B; k += 8 * 64) {      // eight loads in flight, added in index order (the fused block backward leaves 2048 rows per sample)
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = pp[(long)(k + u * 64) * F];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
  }
  for (; k < B; k += 64) {
    const double2 v = pp[(long)k * F];
    a += v.x;
    b += v.y;
  }
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  if (lane == 0) { red[i * 2] = a; red[i * 2 + 1] = b; }
}

// gate backward, the sums over the samples (dW2, dW1, dw_sp) for workgroup `blk` of ceil(R * F / 256): se.hip's se_mlp_bwd_param_kernel, and one
// role of the 16-bit block backward's tail launch (lowp.hip).  red: [N][F][2] summed partials; scratch: dz2 [N][F] | dz1 [N][R] of the per-sample pass.
__device__ __forceinline__ void se_mlp_bwd_param_body(const double* red, const float* gap, const float* hbuf, float* dw1, float* dw2, float* dwsp,
                                                      const double* scratch, int N, int F, int R, int accum, int blk) {
  const double* dz2 = scratch;
  const double* dz1 = scratch + (long)N * F;
  const int i = blk * 256 + threadIdx.x;
  if (i < R * F) {
    {
      const int k = i / F, c = i % F;
      double s = 0.0;
      for (int n = 0; n < N; ++n) s += (double)hbuf[n * R + k] * dz2[n * F + c];
      dw2[i] = accum ? dw2[i] + (float)s : (float)s;
    }
    {
      const int c = i / R, k = i % R;
      double s = 0.0;
      for (int n = 0; n < N; ++n) s += (double)gap[n * F + c] * dz1[n * R + k];
      dw1[i] = accum ? dw1[i] + (float)s : (float)s;
    }
  }
  if (i < F) {
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += red[((long)n * F + i) * 2 + 1];
    dwsp[i] = accum ? dwsp[i] + (float)s : (float)s;
  }
}
