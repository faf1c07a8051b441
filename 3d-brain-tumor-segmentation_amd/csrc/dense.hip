// Dense layers of the VAE head (layers/vae.py:61-64 proj 8192->256, :105-109 unproj 128->512 ReLU).
// Batch is tiny (N = 1..8 per GPU) so these are weight-streaming GEMVs, HBM-bound on the (in,out) kernel:
// forward splits the input rows over workgroups (coalesced 4 B/lane rows of W), fp32 partials combined in a fixed
// order by a finalize kernel; backward is one thread per weight for dW and one wave per input row for dx.
#include "common.h"
#include "bts_internal.h"

#define DENSE_NB 8  // samples handled per pass

// grid (chunks, ceil(out/256), ceil(N/8)); partial[(chunk*N + n)*out + u]
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* partial, int N, int in, int out, int rows_per_chunk) {
  __shared__ float xs[DENSE_NB * 256];
  const int u = blockIdx.y * 256 + threadIdx.x;
  const int n0 = blockIdx.z * DENSE_NB;
  const int nb = (N - n0) < DENSE_NB ? (N - n0) : DENSE_NB;
  const int i0 = blockIdx.x * rows_per_chunk;
  int i1 = i0 + rows_per_chunk;
  if (i1 > in) i1 = in;
  float acc[DENSE_NB];
#pragma unroll
  for (int n = 0; n < DENSE_NB; ++n) acc[n] = 0.f;
  for (int ib = i0; ib < i1; ib += 256) {
    __syncthreads();
    for (int n = 0; n < nb; ++n) {
      const int i = ib + threadIdx.x;
      xs[n * 256 + threadIdx.x] = (i < i1) ? x[(long)(n0 + n) * in + i] : 0.f;
    }
    __syncthreads();
    const int lim = (i1 - ib) < 256 ? (i1 - ib) : 256;
    if (u < out) {
      for (int k = 0; k < lim; ++k) {
        const float wv = w[(long)(ib + k) * out + u];
#pragma unroll
        for (int n = 0; n < DENSE_NB; ++n) acc[n] = fmaf(xs[n * 256 + k], wv, acc[n]);
      }
    }
  }
  if (u < out)
    for (int n = 0; n < nb; ++n) partial[((long)blockIdx.x * N + n0 + n) * out + u] = acc[n];
}

__global__ void dense_fwd_finalize_kernel(const float* partial, const float* bias, float* y, int N, int out, int chunks, int relu) {
  const int total = N * out;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += partial[(long)c * total + i];
    if (bias) s += bias[i % out];
    if (relu) s = fmaxf(s, 0.f);
    y[i] = s;
  }
}

static int dense_chunks(int in, int* rows_per_chunk) {
  int chunks = (in + 127) / 128;
  if (chunks > 128) chunks = 128;
  int rpc = (in + chunks - 1) / chunks;
  *rows_per_chunk = rpc;
  return (in + rpc - 1) / rpc;
}

extern "C" long bts_dense_workspace(int N, int in, int out) {
  int rpc;
  const int chunks = dense_chunks(in, &rpc);
  return (long)chunks * N * out * 4 + 64;
}

// y (N,out) = act(x (N,in) @ w (in,out) + b)
extern "C" int bts_dense_fwd(const float* x, const float* w, const float* bias, float* y, void* workspace,
                             long workspace_bytes, int N, int in, int out, int relu, hipStream_t stream) {
  if (N <= 0 || in <= 0 || out <= 0) return BTS_ERR_SHAPE;
  if (workspace_bytes < bts_dense_workspace(N, in, out)) return BTS_ERR_WORKSPACE;
  int rpc;
  const int chunks = dense_chunks(in, &rpc);
  float* partial = reinterpret_cast<float*>(workspace);
  (void)hipGetLastError(); hipLaunchKernelGGL(dense_fwd_kernel, dim3(chunks, (out + 255) / 256, (N + DENSE_NB - 1) / DENSE_NB), dim3(256), 0, stream,
                     x, w, partial, N, in, out, rpc);
  BTS_LAUNCH_CHECK();
  (void)hipGetLastError(); hipLaunchKernelGGL(dense_fwd_finalize_kernel, dim3((N * out + 255) / 256), dim3(256), 0, stream, partial, bias, y, N, out,
                     chunks, relu);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// dW[i][u] (+)= sum_n x[n][i]*g[n][u]; g = dy (already through the activation derivative)
__global__ void dense_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ g, float* dw, float* db, int N,
                                   int in, int out, int accum) {
  const long total = (long)in * out;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int i = (int)(idx / out), u = (int)(idx % out);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(x[(long)n * in + i], g[(long)n * out + u], s);
    dw[idx] = accum ? dw[idx] + s : s;
    if (db && i == 0) {
      float b = 0.f;
      for (int n = 0; n < N; ++n) b += g[(long)n * out + u];
      db[u] = accum ? db[u] + b : b;
    }
  }
}

// dx[n][i] (+)= sum_u w[i][u]*g[n][u]; one wave per input row i
__global__ __launch_bounds__(256) void dense_bwd_x_kernel(const float* __restrict__ w, const float* __restrict__ g, float* dx,
                                                          int N, int in, int out, int accum) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= in) return;
  for (int n = 0; n < N; ++n) {
    float s = 0.f;
    for (int u = lane; u < out; u += 64) s = fmaf(w[(long)i * out + u], g[(long)n * out + u], s);
    s = wave_sum_f32(s);
    if (lane == 0) dx[(long)n * in + i] = accum ? dx[(long)n * in + i] + s : s;
  }
}

extern "C" int bts_dense_bwd(const float* x, const float* w, const float* g, float* dx, float* dw, float* db, int N, int in,
                             int out, int accumulate_dx, int accumulate_params, hipStream_t stream) {
  if (N <= 0 || in <= 0 || out <= 0) return BTS_ERR_SHAPE;
  long total = (long)in * out;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  (void)hipGetLastError(); hipLaunchKernelGGL(dense_bwd_w_kernel, dim3((int)blocks), dim3(256), 0, stream, x, g, dw, db, N, in, out, accumulate_params);
  BTS_LAUNCH_CHECK();
  if (dx) {
    (void)hipGetLastError(); hipLaunchKernelGGL(dense_bwd_x_kernel, dim3((in + 3) / 4), dim3(256), 0, stream, w, g, dx, N, in, out, accumulate_dx);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}
