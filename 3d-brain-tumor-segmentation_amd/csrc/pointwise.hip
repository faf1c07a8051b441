// Element-wise pieces of the hot path (all HBM-bound, 16 B per lane, grid-stride):
//   dropout on the input volume           layers/encoder.py:39,71   (y = x*mask/(1-rate), mask injectable)
//   counter-based uniform / normal draws  (tf.random.normal in layers/vae.py:12; Dropout's mask)
//   VAE reparameterisation z = mu + exp(0.5*logvar)*eps   layers/vae.py:9-13,123-125  (+ backward)
//   small utilities (fill, axpy, strided add/copy, scalar add) used by the gradient tape.
#include "common.h"
#include "bts_internal.h"

static inline int ew_blocks(long n, int per_thread = 1) {
  long b = (n / per_thread + 255) / 256;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  return (int)b;
}

// (mix32 / u01: common.h -- the 16-bit engine's fused dropout + cast draws from the same generator)

__global__ void dropout_mask_kernel(uint8_t* mask, long n, float rate, uint64_t seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    mask[i] = u01(seed, (uint64_t)i) >= rate ? 1 : 0;
}
extern "C" int bts_dropout_mask(uint8_t* mask, long n, float rate, uint64_t seed, hipStream_t stream) {
  if (n <= 0) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, mask, n, rate, seed);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

__global__ void dropout_apply_kernel(const float* x, const uint8_t* mask, float* y, long n, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = mask[i] ? x[i] * scale : 0.f;
}
// y = x * mask / (1 - rate)
extern "C" int bts_dropout_apply(const float* x, const uint8_t* mask, float* y, long n, float rate, hipStream_t stream) {
  if (n <= 0 || rate < 0.f || rate >= 1.f) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(dropout_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, x, mask, y, n, 1.0f / (1.0f - rate));
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

__global__ void normal_kernel(float* out, long n, uint64_t seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u1 = fmaxf(u01(seed, 2 * (uint64_t)i), 5.9604645e-8f);
    const float u2 = u01(seed, 2 * (uint64_t)i + 1);
    out[i] = sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
  }
}
extern "C" int bts_normal(float* out, long n, uint64_t seed, hipStream_t stream) {
  if (n <= 0) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(normal_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, out, n, seed);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// proj: (N, 2L) = [z_mean | z_logvar]; z: (N, L)
__global__ void vae_sample_fwd_kernel(const float* proj, const float* eps, float* z, int N, int L) {
  const int total = N * L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / L, k = i % L;
    z[i] = proj[n * 2 * L + k] + expf(0.5f * proj[n * 2 * L + L + k]) * eps[i];
  }
}
extern "C" int bts_vae_sample_fwd(const float* proj, const float* eps, float* z, int N, int L, hipStream_t stream) {
  if (N <= 0 || L <= 0) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(vae_sample_fwd_kernel, dim3(ew_blocks((long)N * L)), dim3(256), 0, stream, proj, eps, z, N, L);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
// dproj[:, :L] += dz ; dproj[:, L:] += dz * 0.5*exp(0.5*logvar)*eps
__global__ void vae_sample_bwd_kernel(const float* proj, const float* eps, const float* dz, float* dproj, int N, int L) {
  const int total = N * L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / L, k = i % L;
    dproj[n * 2 * L + k] += dz[i];
    dproj[n * 2 * L + L + k] += dz[i] * 0.5f * expf(0.5f * proj[n * 2 * L + L + k]) * eps[i];
  }
}
extern "C" int bts_vae_sample_bwd(const float* proj, const float* eps, const float* dz, float* dproj, int N, int L,
                                  hipStream_t stream) {
  if (N <= 0 || L <= 0) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(vae_sample_bwd_kernel, dim3(ew_blocks((long)N * L)), dim3(256), 0, stream, proj, eps, dz, dproj, N, L);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

__global__ void fill_kernel(float* p, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int bts_fill(float* p, long n, float v, hipStream_t stream) {
  if (n <= 0) return BTS_OK;
  (void)hipGetLastError(); hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, p, n, v);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

__global__ void axpy_kernel(float* y, const float* x, long n, float a) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = fmaf(a, x[i], y[i]);
}
extern "C" int bts_axpy(float* y, const float* x, long n, float a, hipStream_t stream) {
  if (n <= 0) return BTS_OK;
  (void)hipGetLastError(); hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, y, x, n, a);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// dst[r*ldd + c] (+)= src[r*lds + c]   (channel-slice copy / accumulate; virtual-concat bookkeeping)
__global__ void add_strided_kernel(float* dst, const float* src, long rows, int C, int ldd, int lds_, int accum) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float v = src[r * lds_ + c];
    float* d = dst + r * ldd + c;
    *d = accum ? *d + v : v;
  }
}
extern "C" int bts_add_strided(float* dst, const float* src, long rows, int C, int ldd, int lds_, int accumulate,
                               hipStream_t stream) {
  if (rows <= 0 || C <= 0 || ldd < C || lds_ < C) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(add_strided_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, stream, dst, src, rows, C, ldd, lds_, accumulate);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

__global__ void scalar_lincomb_kernel(float* out, const float* a, const float* b, float ca, float cb) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = ca * a[0] + (b ? cb * b[0] : 0.f);
}
// out = ca*a + cb*b on 1-element device buffers (loss + sum(model.losses), train.py:145-146)
extern "C" int bts_scalar_lincomb(float* out, const float* a, const float* b, float ca, float cb, hipStream_t stream) {
  (void)hipGetLastError(); hipLaunchKernelGGL(scalar_lincomb_kernel, dim3(1), dim3(64), 0, stream, out, a, b, ca, cb);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// dx = dy * [y > 0]   (stand-alone ReLU gradient, Dense activation='relu' in layers/vae.py:105-109)
__global__ void relu_bwd_kernel(const float* y, const float* dy, float* dx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
extern "C" int bts_relu_bwd(const float* y, const float* dy, float* dx, long n, hipStream_t stream) {
  if (n <= 0) return BTS_OK;
  (void)hipGetLastError(); hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, y, dy, dx, n);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// dx = dy * y*(1-y)   (gradient through the fused activation='sigmoid' of the output conv, decoder.py:60)
__global__ void sigmoid_bwd_kernel(const float* y, const float* dy, float* dx, long rows, int C, int ldy, int lddy) {
  const long n = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float v = y[r * ldy + c];
    dx[i] = dy[r * lddy + c] * v * (1.f - v);
  }
}
extern "C" int bts_sigmoid_bwd(const float* y, const float* dy, float* dx, long rows, int C, int ldy, int lddy,
                               hipStream_t stream) {
  if (rows <= 0 || C <= 0) return BTS_ERR_SHAPE;
  (void)hipGetLastError(); hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, stream, y, dy, dx, rows, C, ldy, lddy);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- full-volume inference helpers (test.py:95-151) ----------------------------------------------------
// dst[n,d,h,w,c] (+)= scale * t(src[n, fd(d), fh(h), fw(w), c]) with t(v) = (v - mean[c]) / std[c] when mean != NULL
// (input normalisation, test.py:111), identity otherwise; flip bits: 4 = D, 2 = H, 1 = W (tf.reverse on those axes,
// test.py:139,142).  One float4 per thread when C % 4 == 0, dense NDHWC tensors.
__global__ void flip_affine_kernel(const float* __restrict__ src, float* dst, const float* __restrict__ mean,
                                   const float* __restrict__ stdv, long nvox, int D, int H, int W, int C, int flip, float scale,
                                   int accum) {
  const long total = nvox * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long v = i / C;
    const int w = (int)(v % W); v /= W;
    const int h = (int)(v % H); v /= H;
    const int d = (int)(v % D);
    const long n = v / D;
    const int sd = (flip & 4) ? D - 1 - d : d, sh = (flip & 2) ? H - 1 - h : h, sw = (flip & 1) ? W - 1 - w : w;
    float val = src[(((n * D + sd) * H + sh) * W + sw) * C + c];
    if (mean) val = (val - mean[c]) / stdv[c];
    val *= scale;
    dst[i] = accum ? dst[i] + val : val;
  }
}

extern "C" int bts_flip_affine(const float* src, float* dst, const float* mean, const float* stdv, int N, int D, int H, int W,
                               int C, int flip_mask, float scale, int accumulate, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (flip_mask & ~7)) return BTS_ERR_SHAPE;
  if ((mean == nullptr) != (stdv == nullptr)) return BTS_ERR_SHAPE;
  if (src == dst && flip_mask != 0) return BTS_ERR_UNSUPPORTED;  // an in-place flip would read overwritten voxels
  const long total = (long)N * D * H * W * C;
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError(); hipLaunchKernelGGL(flip_affine_kernel, dim3((int)blocks), dim3(256), 0, stream, src, dst, mean, stdv,
                     (long)N * D * H * W, D, H, W, C, flip_mask, scale, accumulate);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// y = prob * bmask (test.py:154-155) and the label map the script intends (its argmax is commented out, :157-158):
// label = argmax_c + 1, values >= 3 become 4 (test.py:259-261), 0 where the voxel is masked out or no class reaches
// `threshold`.  First maximum wins ties (tf.argmax / np.argmax convention).
__global__ void tta_finish_kernel(const float* __restrict__ prob, const float* __restrict__ bmask, float* y, uint8_t* labels,
                                  long nvox, int C, float threshold) {
  for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < nvox; v += (long)gridDim.x * blockDim.x) {
    const float m = bmask[v];
    float best = -1.f;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
      const float pv = prob[v * C + c] * m;
      if (y) y[v * C + c] = pv;
      if (pv > best) { best = pv; arg = c; }
    }
    if (labels) {
      int lbl = arg + 1;
      if (lbl >= 3) lbl = 4;
      if (m == 0.f || best < threshold) lbl = 0;
      labels[v] = (uint8_t)lbl;
    }
  }
}

extern "C" int bts_tta_finish(const float* prob, const float* bmask, float* y, uint8_t* labels, long nvox, int C,
                              float threshold, hipStream_t stream) {
  if (nvox <= 0 || C <= 0 || C > 250) return BTS_ERR_SHAPE;
  long blocks = (nvox + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError(); hipLaunchKernelGGL(tta_finish_kernel, dim3((int)blocks), dim3(256), 0, stream, prob, bmask, y, labels, nvox, C,
                     threshold);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- non-default samplers (downsample.py:51-70 MaxPooling3D 2/2; upsample.py:49-79 UpSampling3D 2 = nearest) ----------
// x: (N,D,H,W,C) stride ldx (D,H,W even), y: (N,D/2,H/2,W/2,C) stride ldy; idx: window position 0..7 (dz*4+dy*2+dx) of the
// FIRST maximum (strict > in window order), one byte per output element, dense.
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* y, uint8_t* idx, long nout, int Do, int Ho, int Wo, int C,
                                    int ldx, int ldy) {
  const long total = nout * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long v = i / C;
    const int wo = (int)(v % Wo); v /= Wo;
    const int ho = (int)(v % Ho); v /= Ho;
    const int dz = (int)(v % Do);
    const long n = v / Do;
    float best = 0.f;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long src = (((n * (2 * Do) + 2 * dz + (k >> 2)) * (2 * Ho) + 2 * ho + ((k >> 1) & 1)) * (2 * Wo) + 2 * wo + (k & 1));
      const float val = x[src * ldx + c];
      if (k == 0 || val > best) { best = val; arg = k; }
    }
    y[(i / C) * ldy + c] = best;
    idx[i] = (uint8_t)arg;
  }
}
// dx[n, 2d+a, 2h+b, 2w+e, c] (+)= dy[n,d,h,w,c] if idx == a*4+b*2+e else 0  (every input voxel belongs to one window)
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* dx, long nin, int D, int H,
                                    int W, int C, int lddy, int lddx, int accum) {
  const long total = nin * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long v = i / C;
    const int w = (int)(v % W); v /= W;
    const int h = (int)(v % H); v /= H;
    const int d = (int)(v % D);
    const long n = v / D;
    const long o = ((n * (D / 2) + d / 2) * (H / 2) + h / 2) * (W / 2) + w / 2;
    const int pos = (d & 1) * 4 + (h & 1) * 2 + (w & 1);
    const float g = (idx[o * C + c] == pos) ? dy[o * lddy + c] : 0.f;
    float* dst = dx + (i / C) * lddx + c;
    *dst = accum ? *dst + g : g;
  }
}
extern "C" int bts_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int N, int D, int H, int W, int C, int ldx, int ldy,
                                hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || ((D | H | W) & 1) || ldx < C || ldy < C) return BTS_ERR_SHAPE;
  const long nout = (long)N * (D / 2) * (H / 2) * (W / 2);
  long blocks = (nout * C + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError(); hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((int)blocks), dim3(256), 0, stream, x, y, idx, nout, D / 2, H / 2, W / 2, C,
                     ldx, ldy);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
extern "C" int bts_maxpool2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int D, int H, int W, int C, int lddy,
                                int lddx, int accumulate, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || ((D | H | W) & 1) || lddx < C || lddy < C) return BTS_ERR_SHAPE;
  const long nin = (long)N * D * H * W;
  long blocks = (nin * C + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError(); hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, dy, idx, dx, nin, D, H, W, C, lddy, lddx,
                     accumulate);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// nearest-neighbour x2: y[n,2d+a,2h+b,2w+e,c] = x[n,d,h,w,c]; backward sums the 8 children
__global__ void upsample2_fwd_kernel(const float* __restrict__ x, float* y, long nfine, int D2, int H2, int W2, int C, int ldx,
                                     int ldy) {
  const long total = nfine * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long v = i / C;
    const int w = (int)(v % W2); v /= W2;
    const int h = (int)(v % H2); v /= H2;
    const int d = (int)(v % D2);
    const long n = v / D2;
    const long src = ((n * (D2 / 2) + d / 2) * (H2 / 2) + h / 2) * (W2 / 2) + w / 2;
    y[(i / C) * ldy + c] = x[src * ldx + c];
  }
}
__global__ void upsample2_bwd_kernel(const float* __restrict__ dy, float* dx, long ncoarse, int D, int H, int W, int C, int lddy,
                                     int lddx, int accum) {
  const long total = ncoarse * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long v = i / C;
    const int w = (int)(v % W); v /= W;
    const int h = (int)(v % H); v /= H;
    const int d = (int)(v % D);
    const long n = v / D;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // fixed order
      const long f = ((n * (2 * D) + 2 * d + (k >> 2)) * (2 * H) + 2 * h + ((k >> 1) & 1)) * (2 * W) + 2 * w + (k & 1);
      s += dy[f * lddy + c];
    }
    float* dst = dx + (i / C) * lddx + c;
    *dst = accum ? *dst + s : s;
  }
}
extern "C" int bts_upsample2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, int ldx, int ldy,
                                 hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || ldx < C || ldy < C) return BTS_ERR_SHAPE;
  const long nfine = (long)N * D * H * W * 8;
  long blocks = (nfine * C + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError(); hipLaunchKernelGGL(upsample2_fwd_kernel, dim3((int)blocks), dim3(256), 0, stream, x, y, nfine, 2 * D, 2 * H, 2 * W, C, ldx,
                     ldy);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
extern "C" int bts_upsample2_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, int lddy, int lddx, int accumulate,
                                 hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || lddx < C || lddy < C) return BTS_ERR_SHAPE;
  const long nc = (long)N * D * H * W;
  long blocks = (nc * C + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError(); hipLaunchKernelGGL(upsample2_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, dy, dx, nc, D, H, W, C, lddy, lddx,
                     accumulate);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
