// Loss, metric and regulariser of the training step (gfx950, single streaming pass each).
//   DiceVAELoss       util.py:13-24   loss = mean_c(1-(2I_c+1)/(P_c+T_c+1)) + 0.1*mean((x-y_vae)^2) + 0.1*mean(mu^2+exp(lv)-lv-1)
//                                      I,P,T are summed over batch AND space (util.py:11,18-20: SURVEY F9) -> in data-parallel
//                                      runs the raw sums are all-reduced between bts_loss_sums and bts_loss_value.
//   DiceCoefficient   util.py:35-57   mask = max_c>0.5, one-hot(argmax_c)*mask, macro over (last spatial axis, C) cells for
//                                      channels_last (util.py:36: SURVEY F8) or over C for channels_first; micro = SI/(SP+ST)
//   L2 regularisers   train.py:146    sum_p l2_p * sum p^2 over the masked parameter set
#include "common.h"
#include "bts_internal.h"

#define LOSS_MAXC 8
#define LOSS_BLOCKS 1024

// sums layout (double): [0..C) I, [C..2C) P, [2C..3C) T, [3C] sum (x-y_vae)^2, [3C+1] KL sum, [3C+2] numel_x, [3C+3] numel_z
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ ypred, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ yvae,
                                                           double* partial, long NV, int C, int ldp, int ldy, int Cx,
                                                           int ldx, int ldv) {
  __shared__ double sh[8];
  double acc[3 * LOSS_MAXC + 1];
#pragma unroll
  for (int i = 0; i < 3 * LOSS_MAXC + 1; ++i) acc[i] = 0.0;
  for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < NV; v += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < LOSS_MAXC; ++c)
      if (c < C) {
        const float p = ypred[v * ldp + c], t = y[v * ldy + c];
        acc[c] += (double)(p * t);
        acc[LOSS_MAXC + c] += (double)(p * p);
        acc[2 * LOSS_MAXC + c] += (double)(t * t);
      }
    if (x) {
      float m = 0.f;
      for (int c = 0; c < Cx; ++c) {
        const float d = x[v * ldx + c] - yvae[v * ldv + c];
        m = fmaf(d, d, m);
      }
      acc[3 * LOSS_MAXC] += (double)m;
    }
  }
  for (int i = 0; i < 3 * LOSS_MAXC + 1; ++i) {
    const double r = block_sum_f64(acc[i], sh);
    if (threadIdx.x == 0) partial[(long)blockIdx.x * (3 * LOSS_MAXC + 1) + i] = r;
  }
}

__global__ void loss_sums_finalize_kernel(const double* partial, const float* proj, double* sums, int blocks, int C, int N,
                                          int Lz, double numel_x) {
  __shared__ double sh[8];
  for (int i = 0; i < 3 * C + 1; ++i) {
    const int src = (i < 3 * C) ? (i / C) * LOSS_MAXC + (i % C) : 3 * LOSS_MAXC;
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += blockDim.x) s += partial[(long)b * (3 * LOSS_MAXC + 1) + src];
    const double r = block_sum_f64(s, sh);
    if (threadIdx.x == 0) sums[i] = r;
  }
  double k = 0.0;
  if (proj)
    for (int i = threadIdx.x; i < N * Lz; i += blockDim.x) {
      const int n = i / Lz, j = i % Lz;
      const float mu = proj[n * 2 * Lz + j], lv = proj[n * 2 * Lz + Lz + j];
      k += (double)(mu * mu + expf(lv) - lv - 1.0f);
    }
  const double rk = block_sum_f64(k, sh);
  if (threadIdx.x == 0) {
    sums[3 * C + 1] = rk;
    sums[3 * C + 2] = numel_x;
    sums[3 * C + 3] = (double)N * Lz;
  }
}

extern "C" long bts_loss_workspace(void) { return (long)LOSS_BLOCKS * (3 * LOSS_MAXC + 1) * 8 + 64; }

// Raw sums of this rank's batch shard -> sums[3C+4] (double, device). x/y_vae/proj may be null (no VAE terms).
extern "C" int bts_loss_sums(const float* y_pred, const float* y, const float* x, const float* y_vae, const float* proj,
                             double* sums, void* workspace, long workspace_bytes, int N, long V, int C, int ldp, int ldy,
                             int Cx, int ldx, int ldv, int Lz, hipStream_t stream) {
  if (N <= 0 || V <= 0 || C <= 0 || C > LOSS_MAXC) return BTS_ERR_SHAPE;
  if (workspace_bytes < bts_loss_workspace()) return BTS_ERR_WORKSPACE;
  const long NV = (long)N * V;
  long blocks = (NV + 255) / 256;
  if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
  double* partial = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError(); hipLaunchKernelGGL(loss_partial_kernel, dim3((int)blocks), dim3(256), 0, stream, y_pred, y, x, y_vae, partial, NV, C, ldp,
                     ldy, Cx, ldx, ldv);
  BTS_LAUNCH_CHECK();
  (void)hipGetLastError(); hipLaunchKernelGGL(loss_sums_finalize_kernel, dim3(1), dim3(256), 0, stream, partial, proj, sums, (int)blocks, C, N, Lz,
                     (double)NV * Cx);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// loss (1 float) and its three components from (possibly all-reduced) sums
__global__ void loss_value_kernel(const double* sums, float* loss, float* parts, int C, int has_vae) {
  if (threadIdx.x || blockIdx.x) return;
  double d = 0.0;
  for (int c = 0; c < C; ++c) d += 1.0 - (2.0 * sums[c] + 1.0) / (sums[C + c] + sums[2 * C + c] + 1.0);
  d /= C;
  double l2 = 0.0, kl = 0.0;
  if (has_vae) {
    l2 = sums[3 * C] / sums[3 * C + 2];
    kl = sums[3 * C + 1] / sums[3 * C + 3];
  }
  loss[0] = (float)(d + 0.1 * l2 + 0.1 * kl);
  if (parts) { parts[0] = (float)d; parts[1] = (float)l2; parts[2] = (float)kl; }
}
extern "C" int bts_loss_value(const double* sums, float* loss, float* parts, int C, int has_vae, hipStream_t stream) {
  (void)hipGetLastError(); hipLaunchKernelGGL(loss_value_kernel, dim3(1), dim3(64), 0, stream, sums, loss, parts, C, has_vae);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// Gradients (SURVEY Appendix A' "Loss"), scaled by the upstream scalar *gscale (device) :
//   dlogit = -(1/C)*[2y/D_c - 2p(2I_c+1)/D_c^2] * p(1-p)   (through the fused output sigmoid, decoder.py:60)
//   dyvae  = 0.2*(y_vae - x)/numel_x ; dproj[:, :L] = 0.2*mu/numel_z ; dproj[:, L:] = 0.1*(exp(lv)-1)/numel_z
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ ypred, const float* __restrict__ y,
                                                       const float* __restrict__ x, const float* __restrict__ yvae,
                                                       const float* __restrict__ proj, const double* __restrict__ sums,
                                                       const float* __restrict__ gscale, float* __restrict__ dlogit,
                                                       float* __restrict__ dyvae, float* __restrict__ dproj, long NV, int C,
                                                       int ldp, int ldy, int Cx, int ldx, int ldv, int N, int Lz,
                                                       int through_sigmoid) {
  const float gs = gscale ? gscale[0] : 1.f;
  float k1[LOSS_MAXC], k2[LOSS_MAXC];
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) {
    k1[c] = k2[c] = 0.f;
    if (c < C) {
      const double D = sums[C + c] + sums[2 * C + c] + 1.0;
      k1[c] = (float)(-2.0 / (C * D) * gs);
      k2[c] = (float)(2.0 * (2.0 * sums[c] + 1.0) / (C * D * D) * gs);
    }
  }
  const float kx = x ? (float)(0.2 / sums[3 * C + 2]) * gs : 0.f;
  for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < NV; v += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < LOSS_MAXC; ++c)
      if (c < C) {
        const float p = ypred[v * ldp + c], t = y[v * ldy + c];
        float g = k1[c] * t + k2[c] * p;
        if (through_sigmoid) g *= p * (1.f - p);
        dlogit[v * C + c] = g;
      }
    if (x)
      for (int c = 0; c < Cx; ++c) dyvae[v * Cx + c] = kx * (yvae[v * ldv + c] - x[v * ldx + c]);
  }
  if (proj && blockIdx.x == 0) {
    const float kz = (float)(1.0 / sums[3 * C + 3]) * gs;
    for (int i = threadIdx.x; i < N * Lz; i += blockDim.x) {
      const int n = i / Lz, j = i % Lz;
      dproj[n * 2 * Lz + j] = 0.2f * kz * proj[n * 2 * Lz + j];
      dproj[n * 2 * Lz + Lz + j] = 0.1f * kz * (expf(proj[n * 2 * Lz + Lz + j]) - 1.f);
    }
  }
}
extern "C" int bts_loss_bwd(const float* y_pred, const float* y, const float* x, const float* y_vae, const float* proj,
                            const double* sums, const float* gscale, float* dlogit, float* dyvae, float* dproj, int N, long V,
                            int C, int ldp, int ldy, int Cx, int ldx, int ldv, int Lz, int through_sigmoid,
                            hipStream_t stream) {
  if (N <= 0 || V <= 0 || C <= 0 || C > LOSS_MAXC) return BTS_ERR_SHAPE;
  const long NV = (long)N * V;
  long blocks = (NV + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError(); hipLaunchKernelGGL(loss_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, y_pred, y, x, y_vae, proj, sums, gscale, dlogit,
                     dyvae, dproj, NV, C, ldp, ldy, Cx, ldx, ldv, N, Lz, through_sigmoid);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Dice metric. table[(cell*C + c)*3 + {I,P,T}] (double, zeroed here), cells = W (channels_last quirk) or 1.
// labels[v] = argmax_c y_pred (first max, like tf.argmax) + 1, or 0 where max <= 0.5  (uint8)
// ---------------------------------------------------------------------------------------------
__global__ void zero_f64_kernel(double* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.0;
}
__global__ __launch_bounds__(256) void dice_metric_kernel(const float* __restrict__ ytrue, const float* __restrict__ ypred,
                                                          uint8_t* __restrict__ labels, double* table, long NV, int C,
                                                          int ldt, int ldp, int cells, int W) {
  extern __shared__ double tab[];  // cells*C*3
  const int tsz = cells * C * 3;
  for (int i = threadIdx.x; i < tsz; i += blockDim.x) tab[i] = 0.0;
  __syncthreads();
  for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < NV; v += (long)gridDim.x * blockDim.x) {
    float best = ypred[v * ldp];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float p = ypred[v * ldp + c];
      if (p > best) { best = p; arg = c; }
    }
    const bool on = best > 0.5f;
    if (labels) labels[v] = on ? (uint8_t)(arg + 1) : 0;
    const int cell = (cells > 1) ? (int)(v % W) : 0;
    for (int c = 0; c < C; ++c) {
      const float t = ytrue[v * ldt + c];
      const float ph = (on && c == arg) ? 1.f : 0.f;
      double* e = tab + (cell * C + c) * 3;
      if (ph != 0.f) {
        atomicAdd(e + 1, 1.0);
        if (t != 0.f) atomicAdd(e, (double)t);
      }
      if (t != 0.f) atomicAdd(e + 2, (double)t);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tsz; i += blockDim.x)
    if (tab[i] != 0.0) atomicAdd(table + i, tab[i]);
}
extern "C" int bts_dice_metric_sums(const float* y_true, const float* y_pred, uint8_t* labels, double* table, int N, long V,
                                    int W, int C, int ldt, int ldp, int channels_last_axes, hipStream_t stream) {
  if (N <= 0 || V <= 0 || C <= 0 || W <= 0 || V % W) return BTS_ERR_SHAPE;
  const int cells = channels_last_axes ? W : 1;
  const int tsz = cells * C * 3;
  if ((size_t)tsz * 8 > 60 * 1024) return BTS_ERR_UNSUPPORTED;
  (void)hipGetLastError(); hipLaunchKernelGGL(zero_f64_kernel, dim3((tsz + 255) / 256), dim3(256), 0, stream, table, tsz);
  BTS_LAUNCH_CHECK();
  const long NV = (long)N * V;
  long blocks = (NV + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  (void)hipGetLastError(); hipLaunchKernelGGL(dice_metric_kernel, dim3((int)blocks), dim3(256), tsz * sizeof(double), stream, y_true, y_pred, labels,
                     table, NV, C, ldt, ldp, cells, W);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
// out[0] = macro, out[1] = micro   (micro has no smoothing: NaN when both empty, as in the reference)
__global__ void dice_metric_value_kernel(const double* table, float* out, int cells, int C) {
  // one wave: lane l takes the entries l, l + 64, ... (independent loads in flight: the single-thread loop spent 65 us on 384 dependent
  // L2 round trips), then a fixed-order tree over the lanes
  if (blockIdx.x || threadIdx.x >= 64) return;
  double macro = 0.0, si = 0.0, sp = 0.0, st = 0.0;
  for (int i = threadIdx.x; i < cells * C; i += 64) {
    const double I = table[i * 3], P = table[i * 3 + 1], T = table[i * 3 + 2];
    macro += (2.0 * I + 1.0) / (P + T + 1.0);
    si += I; sp += P; st += T;
  }
  macro = wave_sum_f64(macro); si = wave_sum_f64(si); sp = wave_sum_f64(sp); st = wave_sum_f64(st);
  if (threadIdx.x == 0) {
    out[0] = (float)(macro / (cells * C));
    out[1] = (float)(si / (sp + st));
  }
}
extern "C" int bts_dice_metric_value(const double* table, float* out, int W, int C, int channels_last_axes,
                                     hipStream_t stream) {
  (void)hipGetLastError(); hipLaunchKernelGGL(dice_metric_value_kernel, dim3(1), dim3(64), 0, stream, table, out, channels_last_axes ? W : 1, C);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// L2 regulariser over ranges of the flat parameter buffer: value = sum_r coef_r * sum(p[r]^2); grad += 2*coef_r*p
// ---------------------------------------------------------------------------------------------
#define L2_MAXR 128   // one range per (layer group, coefficient): the flat buffer is laid out in backward-completion order (model.py)
struct L2Ranges {
  long off[L2_MAXR], len[L2_MAXR];
  float coef[L2_MAXR];
  int nr;
};
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ p, double* partial, L2Ranges rg) {
  __shared__ double sh[8];
  double acc = 0.0;
  for (int r = 0; r < rg.nr; ++r) {
    const float* q = p + rg.off[r];
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rg.len[r]; i += (long)gridDim.x * blockDim.x)
      s += (double)(q[i] * q[i]);
    acc += (double)rg.coef[r] * s;
  }
  const double t = block_sum_f64(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ void l2_finalize_kernel(const double* partial, float* out, int blocks) {
  __shared__ double sh[8];
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks; b += blockDim.x) s += partial[b];
  const double r = block_sum_f64(s, sh);
  if (threadIdx.x == 0) out[0] = (float)r;
}
__global__ void l2_grad_kernel(const float* __restrict__ p, float* __restrict__ g, const float* gscale, L2Ranges rg) {
  const float gs = gscale ? gscale[0] : 1.f;
  for (int r = 0; r < rg.nr; ++r) {
    const float k = 2.f * rg.coef[r] * gs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rg.len[r]; i += (long)gridDim.x * blockDim.x)
      g[rg.off[r] + i] = fmaf(k, p[rg.off[r] + i], g[rg.off[r] + i]);
  }
}
static int l2_make(L2Ranges& rg, const long* off, const long* len, const float* coef, int nr) {
  if (nr < 0 || nr > L2_MAXR) return BTS_ERR_SHAPE;
  rg.nr = nr;
  for (int i = 0; i < nr; ++i) { rg.off[i] = off[i]; rg.len[i] = len[i]; rg.coef[i] = coef[i]; }
  return BTS_OK;
}
extern "C" long bts_l2_workspace(void) { return 1024 * 8 + 64; }
extern "C" int bts_l2_reg_fwd(const float* params, const long* off, const long* len, const float* coef, int nranges,
                              float* out, void* workspace, long workspace_bytes, hipStream_t stream) {
  L2Ranges rg;
  int r = l2_make(rg, off, len, coef, nranges);
  if (r != BTS_OK) return r;
  if (workspace_bytes < bts_l2_workspace()) return BTS_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError(); hipLaunchKernelGGL(l2_partial_kernel, dim3(1024), dim3(256), 0, stream, params, partial, rg);
  BTS_LAUNCH_CHECK();
  (void)hipGetLastError(); hipLaunchKernelGGL(l2_finalize_kernel, dim3(1), dim3(256), 0, stream, partial, out, 1024);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
extern "C" int bts_l2_reg_bwd(const float* params, float* grads, const long* off, const long* len, const float* coef,
                              int nranges, const float* gscale, hipStream_t stream) {
  L2Ranges rg;
  int r = l2_make(rg, off, len, coef, nranges);
  if (r != BTS_OK) return r;
  (void)hipGetLastError(); hipLaunchKernelGGL(l2_grad_kernel, dim3(2048), dim3(256), 0, stream, params, grads, gscale, rg);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
