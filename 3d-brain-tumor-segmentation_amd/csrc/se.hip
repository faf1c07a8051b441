// Squeeze-excitation gate of the ResNet block and the block epilogue, forward and backward (gfx950).
// Reference: layers/resnet.py:116-138
//   res  = conv1x1(inputs)                                   (:118, computed by bts_conv3d_fwd K1)
//   chse = sigmoid(W2^T relu(W1^T GAP(res)))                 (:121-124)  GAP = mean over D,H,W per (n,c)
//   spse = sigmoid(res . w_sp)                               (:127)
//   out  = res*(spse + chse) + relu(GN2(conv2(...)))         (:130,:133-137)   (gate applied to the SHORTCUT; sum, not product)
// Backward math: SURVEY Appendix A' "ResnetBlock gate". All cross-workgroup sums go through fp64 partials in a
// caller-supplied workspace and are combined in a fixed order (bitwise reproducible).
#include "common.h"
#include "bts_internal.h"
#include "finalize_parts.h"

#define SE_BLOCKS_MAX 512

// ---------------------------------------------------------------------------------------------
// column sums over voxels: out[n][c] = scale * sum_rows x[n,row,c]   (GAP, conv-transpose bias gradient)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, double* partial, long rows, int C, int ld,
                                                     long rspan, int vec) {
  __shared__ double sh[256 * 4];
  const int C4 = (C + 3) / 4;
  const int TC = C4 < 256 ? C4 : 256;
  const int TR = 256 / TC;
  const int tx = threadIdx.x % TC, ty = threadIdx.x / TC;
  const long r0 = (long)blockIdx.x * rspan;
  long r1 = r0 + rspan;
  if (r1 > rows) r1 = rows;
  const float* base = x + (long)blockIdx.y * rows * ld;
  for (int cb = 0; cb < C4; cb += TC) {
    const int c = (cb + tx) * 4;
    double s[4] = {0, 0, 0, 0};
    if (ty < TR && c < C) {
      long r = r0 + ty;
      if (vec) {  // four rows in flight; the sums are still taken in row order
        for (; r + 3L * TR < r1; r += 4L * TR) {
          f32x4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (r + (long)u * TR) * ld + c);
#pragma unroll
          for (int u = 0; u < 4; ++u) { s[0] += v[u][0]; s[1] += v[u][1]; s[2] += v[u][2]; s[3] += v[u][3]; }
        }
      }
      for (; r < r1; r += TR) {
        const float* src = base + r * ld + c;
        if (vec) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        } else {
          for (int e = 0; e < 4; ++e)
            if (c + e < C) s[e] += src[e];
        }
      }
    }
    __syncthreads();
    for (int e = 0; e < 4; ++e) sh[threadIdx.x * 4 + e] = s[e];
    __syncthreads();
    // thread t handles column cb*4 + t (t < TC*4)
    for (int t = threadIdx.x; t < TC * 4; t += 256) {
      const int col = cb * 4 + t;
      if (col < C) {
        double a = 0.0;
        for (int k = 0; k < TR; ++k) a += sh[(k * TC + (t >> 2)) * 4 + (t & 3)];
        partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * C + col] = a;
      }
    }
  }
}

// one wave per output element: lanes split the partial blocks, fixed-order shuffle tree
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const double* partial, float* out, int N, int B, int C,
                                                              double scale, int accum, int sum_over_n) {
  const int lane = threadIdx.x & 63;
  const int total = sum_over_n ? C : N * C;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= total) return;
  double s = 0.0;
  if (sum_over_n) {
    for (int k = lane; k < N * B; k += 64) s += partial[(long)k * C + i];
  } else {
    const int n = i / C, c = i % C;
    for (int b = lane; b < B; b += 64) s += partial[((long)n * B + b) * C + c];
  }
  s = wave_sum_f64(s);
  if (lane == 0) {
    const float v = (float)(s * scale);
    out[i] = accum ? out[i] + v : v;
  }
}

static int colsum_blocks(long rows, int N) {
  long B = (1024 + N - 1) / N;
  if (B > SE_BLOCKS_MAX) B = SE_BLOCKS_MAX;
  long span = (rows + B - 1) / B;
  if (span < 64) span = 64;
  return (int)((rows + span - 1) / span);
}

extern "C" long bts_colsum_workspace(int N, long rows, int C) { return (long)N * colsum_blocks(rows, N) * C * 8 + 64; }

// out[n][c] (sum_over_n=0) or out[c] (sum_over_n=1) = scale * column sums of x (N, rows, C) with pixel stride ld
extern "C" int bts_colsum(const float* x, float* out, void* workspace, long workspace_bytes, int N, long rows, int C,
                          int ld, float scale, int sum_over_n, int accumulate, hipStream_t stream) {
  if (N <= 0 || rows <= 0 || C <= 0 || ld < C || C > 4096) return BTS_ERR_SHAPE;
  if (workspace_bytes < bts_colsum_workspace(N, rows, C)) return BTS_ERR_WORKSPACE;
  const int B = colsum_blocks(rows, N);
  const long rspan = (rows + B - 1) / B;
  const int vec = (C % 4 == 0) && (ld % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
  double* partial = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError(); hipLaunchKernelGGL(colsum_kernel, dim3(B, N), dim3(256), 0, stream, x, partial, rows, C, ld, rspan, vec);
  BTS_LAUNCH_CHECK();
  const int total = sum_over_n ? C : N * C;
  (void)hipGetLastError(); hipLaunchKernelGGL(colsum_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, stream, partial, out, N, B, C,
                     (double)scale, accumulate, sum_over_n);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// channel-SE MLP forward: one block per sample (resnet.py:47-58,122-124; Dense kernels are (in,out), no bias)
// ---------------------------------------------------------------------------------------------
// (the first layer's F-long sums are split over 256 / R slices of the block and combined in fixed order: with one thread per hidden
// unit the kernel was a chain of F dependent loads -- 11 us at F = 256, on the critical path of every block)
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* gap, const float* w1, const float* w2, float* hbuf, float* ch, int F, int R) {
  extern __shared__ float shf[];  // gap[F] + h[R] + partial[256]
  float* hs = shf + F;
  float* part = hs + R;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < F; c += 256) shf[c] = gap[n * F + c];
  __syncthreads();
  if (R <= 256) {
    const int P = 256 / R;                  // slices
    const int k = threadIdx.x % R, sl = threadIdx.x / R;
    float s = 0.f;
    if (sl < P)
      for (int c = sl; c < F; c += P) s = fmaf(shf[c], w1[c * R + k], s);
    part[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < R) {
      float t = 0.f;
      for (int q = 0; q < P; ++q) t += part[q * R + threadIdx.x];
      t = fmaxf(t, 0.f);
      hs[threadIdx.x] = t;
      hbuf[n * R + threadIdx.x] = t;
    }
  } else {
    for (int k = threadIdx.x; k < R; k += 256) {
      float s = 0.f;
      for (int c = 0; c < F; ++c) s = fmaf(shf[c], w1[c * R + k], s);
      s = fmaxf(s, 0.f);
      hs[k] = s;
      hbuf[n * R + k] = s;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < F; c += 256) {
    float s = 0.f;
    for (int k = 0; k < R; ++k) s = fmaf(hs[k], w2[k * F + c], s);
    ch[n * F + c] = sigmoidf_(s);
  }
}

extern "C" int bts_se_mlp_fwd(const float* gap, const float* w1, const float* w2, float* h, float* ch, int N, int F, int R,
                              hipStream_t stream) {
  if (N <= 0 || F <= 0 || R <= 0) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(N), dim3(256), (F + R + 256) * sizeof(float), stream, gap, w1, w2, h, ch, F, R);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// block epilogue forward: out = res*(sigmoid(res.wsp) + ch) + relu(GN2(c2))   (GN2 optional: c2==nullptr -> no conv branch)
// F/4 lanes per voxel (float4 each), dot reduced with xor shuffles inside the lane group.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block_epilogue_kernel(
    const float* __restrict__ res, const float* __restrict__ c2, float* __restrict__ out, float* __restrict__ sp_out,
    const float* __restrict__ wsp, const float* __restrict__ ch, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd, long NV, long V, int F,
    int ldo, int G, int cg, long L, int mode) {
  const int F4 = F >> 2;  // lanes per voxel (power of two, <= 64)
  const int vpb = 256 / F4;
  const int lg = threadIdx.x % F4, vl = threadIdx.x / F4;
  const int c = lg * 4;
  const f32x4 w = *reinterpret_cast<const f32x4*>(wsp + c);
  for (long v = (long)blockIdx.x * vpb + vl; v < NV; v += (long)gridDim.x * vpb) {
    const long n = v / V;
    const f32x4 r = *reinterpret_cast<const f32x4*>(res + v * F + c);
    float d = (r[0] * w[0] + r[1] * w[1]) + (r[2] * w[2] + r[3] * w[3]);
    for (int o = F4 >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    const float sp = sigmoidf_(d);
    if (lg == 0) sp_out[v] = sp;
    const f32x4 cc = *reinterpret_cast<const f32x4*>(ch + n * F + c);
    f32x4 o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = r[e] * (sp + cc[e]);
    if (c2) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(c2 + v * F + c);
      const long rin = (v - n * V) * F + c;  // element index inside the sample
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = (mode == BTS_GN_SLAB) ? (int)((rin + e) / L) : (c + e) / cg;
        const int idx = (mode == BTS_GN_SLAB) ? g * cg + ((c + e) % cg) : (c + e);
        const float y = (x[e] - mean[n * G + g]) * rstd[n * G + g] * gamma[idx] + beta[idx];
        o4[e] += fmaxf(y, 0.f);
      }
    }
    *reinterpret_cast<f32x4*>(out + v * ldo + c) = o4;
  }
}

extern "C" int bts_block_epilogue_fwd(const float* res, const float* c2, float* out, float* sp, const float* wsp,
                                      const float* ch, const float* gamma, const float* beta, const float* mean,
                                      const float* rstd, int N, long V, int F, int ldo, int G, int mode,
                                      hipStream_t stream) {
  if (N <= 0 || V <= 0 || F < 4 || (F & (F - 1)) || F > 256 || ldo < F || ldo % 4) return BTS_ERR_SHAPE;
  if (c2 && (F % G != 0)) return BTS_ERR_SHAPE;
  if ((((uintptr_t)res) & 15) || (((uintptr_t)out) & 15) || (c2 && (((uintptr_t)c2) & 15))) return BTS_ERR_ALIGN;
  const long NV = (long)N * V;
  const int vpb = 256 / (F / 4);
  long blocks = (NV + vpb - 1) / vpb;
  if (blocks > 8192) blocks = 8192;
  const int cg = c2 ? F / G : 1;
  const long L = c2 ? V * F / G : 1;
  (void)hipGetLastError(); hipLaunchKernelGGL(block_epilogue_kernel, dim3((int)blocks), dim3(256), 0, stream, res, c2, out, sp, wsp, ch, gamma,
                     beta, mean, rstd, NV, V, F, ldo, G, cg, L, mode);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// gate backward, stage 1: per voxel t = sum_c dout*res ; ds = t*sp*(1-sp) -> ds_out ;
// per-block partial column sums: Pch[n][c] = sum_v dout*res , Pw[c] = sum_v ds*res
// grid (B, N); partial[((n*B + b)*F + c)*2 + {ch, w}]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ res,
                                                            const float* __restrict__ sp, float* __restrict__ ds_out,
                                                            double* partial, long V, int F, int lddo, long vspan) {
  __shared__ double sh[256 * 8];
  const int F4 = F >> 2;
  const int vpb = 256 / F4;
  const int lg = threadIdx.x % F4, vl = threadIdx.x / F4;
  const int c = lg * 4;
  const long n = blockIdx.y;
  const long v0 = (long)blockIdx.x * vspan;
  long v1 = v0 + vspan;
  if (v1 > V) v1 = V;
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  auto one = [&](long v, const f32x4 r, const f32x4 d, float s) {
    float t = (d[0] * r[0] + d[1] * r[1]) + (d[2] * r[2] + d[3] * r[3]);
    for (int o = F4 >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const float ds = t * s * (1.f - s);
    if (lg == 0) ds_out[v] = ds;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] += (double)(d[e] * r[e]);
      b[e] += (double)(ds * r[e]);
    }
  };
  long vv = v0 + vl;
  for (; vv + 3L * vpb < v1; vv += 4L * vpb) {  // four voxels per thread in flight: the shuffle chain of one hides the loads of the next
    f32x4 r[4], d[4];
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long v = n * V + vv + (long)u * vpb;
      r[u] = *reinterpret_cast<const f32x4*>(res + v * F + c);
      d[u] = *reinterpret_cast<const f32x4*>(dout + v * lddo + c);
      s[u] = sp[v];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(n * V + vv + (long)u * vpb, r[u], d[u], s[u]);
  }
  for (; vv < v1; vv += vpb) {
    const long v = n * V + vv;
    const f32x4 r = *reinterpret_cast<const f32x4*>(res + v * F + c);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dout + v * lddo + c);
    one(v, r, d, sp[v]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { sh[threadIdx.x * 8 + e] = a[e]; sh[threadIdx.x * 8 + 4 + e] = b[e]; }
  __syncthreads();
  for (int col = threadIdx.x; col < F; col += 256) {
    const int e = col & 3, l = col >> 2;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < vpb; ++k) { sa += sh[(k * F4 + l) * 8 + e]; sb += sh[(k * F4 + l) * 8 + 4 + e]; }
    const long o = (((long)n * gridDim.x + blockIdx.x) * F + col) * 2;
    partial[o] = sa;
    partial[o + 1] = sb;
  }
}

// stage 2a: one wave per (n,c): sum the per-block partials -> red[(n*F+c)*2 + {ch, w}]
__global__ __launch_bounds__(256) void se_bwd_partial_reduce_kernel(const double* partial, double* red, int N, int B, int F) {
  se_bwd_partial_reduce_body(partial, red, N, B, F, blockIdx.x);
}

// stage 2 (one block): finish the sums, SE-MLP backward, emit dgap (already divided by V) for stage 3
__global__ void se_mlp_bwd_kernel(const double* partial, const float* gap, const float* hbuf, const float* ch,
                                  const float* w1, const float* w2, float* dw1, float* dw2, float* dwsp, float* dgap,
                                  double* scratch /* N*F dz2 + N*R dz1 */, int N, int B, int F, int R, double invV, int accum) {
  double* dz2 = scratch;
  double* dz1 = scratch + (long)N * F;
  for (int i = threadIdx.x; i < N * F; i += blockDim.x) {
    const int n = i / F, c = i % F;
    const double s = partial[((long)n * F + c) * 2];
    const double cc = (double)ch[i];
    dz2[i] = s * cc * (1.0 - cc);
  }
  for (int c = threadIdx.x; c < F; c += blockDim.x) {
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += partial[((long)n * F + c) * 2 + 1];
    dwsp[c] = accum ? dwsp[c] + (float)s : (float)s;
  }
  __syncthreads();
  {  // dz1[n][k] = relu'(h) * sum_c W2[k][c] dz2[n][c]: one wave per output (the F-long sum was the kernel's critical path)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int i = wave; i < N * R; i += nwave) {
      const int n = i / R, k = i % R;
      double s = 0.0;
      for (int c = lane; c < F; c += 64) s += (double)w2[k * F + c] * dz2[n * F + c];
      s = wave_sum_f64(s);
      if (lane == 0) dz1[i] = (hbuf[i] > 0.f) ? s : 0.0;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R * F; i += blockDim.x) {  // dW2[k][c] = sum_n h[n][k]*dz2[n][c]
    const int k = i / F, c = i % F;
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += (double)hbuf[n * R + k] * dz2[n * F + c];
    dw2[i] = accum ? dw2[i] + (float)s : (float)s;
  }
  for (int i = threadIdx.x; i < F * R; i += blockDim.x) {  // dW1[c][k] = sum_n gap[n][c]*dz1[n][k]
    const int c = i / R, k = i % R;
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += (double)gap[n * F + c] * dz1[n * R + k];
    dw1[i] = accum ? dw1[i] + (float)s : (float)s;
  }
  for (int i = threadIdx.x; i < N * F; i += blockDim.x) {  // dgap[n][c] = sum_k W1[c][k]*dz1[n][k], / V
    const int n = i / F, c = i % F;
    double s = 0.0;
    for (int k = 0; k < R; ++k) s += (double)w1[c * R + k] * dz1[n * R + k];
    dgap[i] = (float)(s * invV);
  }
}

// The same arithmetic for batches (N >= 2) in two launches with many workgroups: the one-block kernel above walks N*R dot products of
// length F with four waves (64 us at N = 8, F = 256: 1.35 ms of the batch-8 16-bit step).  Per-sample part: one workgroup per sample.
__global__ __launch_bounds__(256) void se_mlp_bwd_sample_kernel(const double* partial, const float* hbuf, const float* ch, const float* w1,
                                                                const float* w2, float* dgap, double* scratch, int N, int F, int R, double invV) {
  const int n = blockIdx.x;
  double* dz2 = scratch + (long)n * F;
  double* dz1 = scratch + (long)N * F + (long)n * R;
  for (int c = threadIdx.x; c < F; c += 256) {
    const double cc = (double)ch[n * F + c];
    dz2[c] = partial[((long)n * F + c) * 2] * cc * (1.0 - cc);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = wave; k < R; k += 4) {
    double s = 0.0;
    for (int c = lane; c < F; c += 64) s += (double)w2[k * F + c] * dz2[c];
    s = wave_sum_f64(s);
    if (lane == 0) dz1[k] = (hbuf[n * R + k] > 0.f) ? s : 0.0;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < F; c += 256) {
    double s = 0.0;
    for (int k = 0; k < R; ++k) s += (double)w1[c * R + k] * dz1[k];
    dgap[n * F + c] = (float)(s * invV);
  }
}
// sums over the samples: dW2, dW1, dw_sp
__global__ __launch_bounds__(256) void se_mlp_bwd_param_kernel(const double* partial, const float* gap, const float* hbuf, float* dw1, float* dw2,
                                                               float* dwsp, const double* scratch, int N, int F, int R, int accum) {
  se_mlp_bwd_param_body(partial, gap, hbuf, dw1, dw2, dwsp, scratch, N, F, R, accum, blockIdx.x);
}
// the per-sample pass alone (dz2, dz1 into scratch, dgap): callers that run the sums over the samples themselves (lowp.hip's tail launch)
int bts_se_mlp_bwd_sample_(const double* red, double* scratch, const float* h, const float* ch, const float* w1, const float* w2, float* dgap, int N,
                           long V, int F, int R, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(se_mlp_bwd_sample_kernel, dim3(N), dim3(256), 0, stream, red, h, ch, w1, w2, dgap, scratch, N, F, R, 1.0 / (double)V);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
static int se_mlp_bwd_launch(const double* red, const float* gap, const float* h, const float* ch, const float* w1, const float* w2, float* dw1,
                             float* dw2, float* dwsp, float* dgap, double* scratch, int N, int B, long V, int F, int R, int accumulate_params,
                             hipStream_t stream) {
  (void)hipGetLastError();
  // (also for one sample of a wide block: the one-block kernel takes 20 / 41 us at F = 128 / 256 -- R*F products on 256 threads -- on the
  // critical path of the block's backward; the pair below 13 us)
  if (N >= 2 || (long)F * R >= 2048) {
    hipLaunchKernelGGL(se_mlp_bwd_sample_kernel, dim3(N), dim3(256), 0, stream, red, h, ch, w1, w2, dgap, scratch, N, F, R, 1.0 / (double)V);
    BTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(se_mlp_bwd_param_kernel, dim3((R * F + 255) / 256), dim3(256), 0, stream, red, gap, h, dw1, dw2, dwsp, scratch, N, F, R,
                       accumulate_params);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
  }
  hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(1), dim3(256), 0, stream, red, gap, h, ch, w1, w2, dw1, dw2, dwsp, dgap, scratch, N, B, F, R,
                     1.0 / (double)V, accumulate_params);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// stage 3: dres = dout*(sp + ch) + ds*wsp + dgap/V
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ sp,
                                                           const float* __restrict__ ds, const float* __restrict__ ch,
                                                           const float* __restrict__ wsp, const float* __restrict__ dgap,
                                                           float* __restrict__ dres, long NV, long V, int F, int lddo) {
  const int F4 = F >> 2;
  const long total = NV * F4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / F4;
    const int c = (int)(i - v * F4) * 4;
    const long n = v / V;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dout + v * lddo + c);
    const f32x4 cc = *reinterpret_cast<const f32x4*>(ch + n * F + c);
    const f32x4 w = *reinterpret_cast<const f32x4*>(wsp + c);
    const f32x4 dg = *reinterpret_cast<const f32x4*>(dgap + n * F + c);
    const float s = sp[v], dsv = ds[v];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = d[e] * (s + cc[e]) + dsv * w[e] + dg[e];
    *reinterpret_cast<f32x4*>(dres + v * F + c) = o;
  }
}

// stage 2 alone (block_bwd.hip sums the partials in its own middle launch)
int bts_se_mlp_bwd_(const double* red, double* scratch, const float* gap, const float* h, const float* ch, const float* w1, const float* w2,
                    float* dw1, float* dw2, float* dwsp, float* dgap, int N, int B, long V, int F, int R, int accumulate_params,
                    hipStream_t stream) {
  return se_mlp_bwd_launch(red, gap, h, ch, w1, w2, dw1, dw2, dwsp, dgap, scratch, N, B, V, F, R, accumulate_params, stream);
}
// stages 2a + 2 for callers that produced the per-block partials themselves (lowp.hip: 16-bit dout / res), same layouts as above
int bts_se_bwd_middle_(double* partial, double* red, double* scratch, const float* gap, const float* h, const float* ch, const float* w1,
                       const float* w2, float* dw1, float* dw2, float* dwsp, float* dgap, int N, int B, long V, int F, int R,
                       int accumulate_params, hipStream_t stream) {
  (void)hipGetLastError(); hipLaunchKernelGGL(se_bwd_partial_reduce_kernel, dim3((N * F + 3) / 4), dim3(256), 0, stream, partial, red, N, B, F);
  BTS_LAUNCH_CHECK();
  return se_mlp_bwd_launch(red, gap, h, ch, w1, w2, dw1, dw2, dwsp, dgap, scratch, N, B, V, F, R, accumulate_params, stream);
}

static int se_bwd_blocks(long V, int N, int F, long* vspan) {
  const int vpb = 256 / (F / 4);
  long B = (1024 + N - 1) / N;
  if (B > SE_BLOCKS_MAX) B = SE_BLOCKS_MAX;
  long span = (V + B - 1) / B;
  span = (span + vpb - 1) / vpb * vpb;
  *vspan = span;
  return (int)((V + span - 1) / span);
}

extern "C" long bts_se_bwd_workspace(int N, long V, int F, int R) {
  long vspan;
  const int B = se_bwd_blocks(V, N, F, &vspan);
  return (long)N * B * F * 2 * 8 + ((long)N * F * 3 + (long)N * R) * 8 + 128;
}

// Gate backward. Outputs: dres (dense N,V,F), ds (N*V scratch), dgap (N*F scratch), parameter grads dw1,dw2,dwsp.
extern "C" int bts_se_bwd(const float* dout, const float* res, const float* sp, const float* gap, const float* h,
                          const float* ch, const float* w1, const float* w2, const float* wsp, float* dres, float* ds,
                          float* dgap, float* dw1, float* dw2, float* dwsp, void* workspace, long workspace_bytes, int N,
                          long V, int F, int R, int lddo, int accumulate_params, hipStream_t stream) {
  if (N <= 0 || V <= 0 || F < 4 || (F & (F - 1)) || F > 256 || lddo < F || lddo % 4) return BTS_ERR_SHAPE;
  if (workspace_bytes < bts_se_bwd_workspace(N, V, F, R)) return BTS_ERR_WORKSPACE;
  if (((uintptr_t)workspace) & 15) return BTS_ERR_ALIGN;     // (the partial reduce reads double2)
  long vspan;
  const int B = se_bwd_blocks(V, N, F, &vspan);
  double* partial = reinterpret_cast<double*>(workspace);
  double* red = partial + (long)N * B * F * 2;
  double* scratch = red + (long)N * F * 2;
  (void)hipGetLastError(); hipLaunchKernelGGL(se_bwd_reduce_kernel, dim3(B, N), dim3(256), 0, stream, dout, res, sp, ds, partial, V, F, lddo, vspan);
  BTS_LAUNCH_CHECK();
  (void)hipGetLastError(); hipLaunchKernelGGL(se_bwd_partial_reduce_kernel, dim3((N * F + 3) / 4), dim3(256), 0, stream, partial, red, N, B, F);
  BTS_LAUNCH_CHECK();
  {
    const int r = se_mlp_bwd_launch(red, gap, h, ch, w1, w2, dw1, dw2, dwsp, dgap, scratch, N, B, V, F, R, accumulate_params, stream);
    if (r != BTS_OK) return r;
  }
  const long total = (long)N * V * (F / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  (void)hipGetLastError(); hipLaunchKernelGGL(se_bwd_apply_kernel, dim3((int)blocks), dim3(256), 0, stream, dout, sp, ds, ch, wsp, dgap, dres,
                     (long)N * V, V, F, lddo);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
