// A ResnetBlock's gate backward and GroupNorm-2 backward in ONE pair of passes, fp32 engine (resnet.py:121-137 under TF autodiff,
// train.py:142-151).  Both read the gradient of the block output.  The separate routes (bts_se_bwd on the gate stream, bts_gn_bwd on the
// main one) stream nine tensor-sized passes -- reduce: (dout, res) and (dout, c2); apply: dout -> dres and (dout, c2) -> dc2 -- and launch
// nine kernels; here the reduce pass reads dout, res, c2 once (GroupNorm class sums + gate sums + the per-voxel spatial-gate gradient)
// and the apply pass reads dout, c2 once and writes dres and dc2: seven passes, five kernels (reduce, one middle launch for both finalizes, SE-MLP backward, apply).  The 16-bit engine's bts_lp_block_bwd is the
// same idea on 16-bit tensors (lowp.hip).
//
//   gate (SURVEY Appendix A'):  g = dout * res per element;  t_v = sum_c g;  ds_v = t_v sp_v (1 - sp_v)
//                               Pch[n][c] = sum_v g;  Pw[c] = sum_v ds_v res;   dres = dout (sp + ch) + ds w_sp + dgap / V
//   GroupNorm-2 (slab units):   xh = (c2 - mean) rstd;  dE = dout [gamma xh + beta > 0];  A_j = sum dE xh;  B_j = sum dE
//                               dc2 = (dE gamma - c1 - xh c2') rstd   with c1 = sum_j gamma_j B_j / L, c2' = sum_j gamma_j A_j / L
//
// Thread mapping of the GroupNorm kernels (groupnorm.hip): a workgroup owns a span of one (sample, slab) unit in 1024-element chunks,
// a thread four consecutive channels of a voxel, the C/4 lanes of a voxel are neighbours (so the per-voxel sum is a shuffle tree).
#include "common.h"
#include "bts_internal.h"
#include "finalize_parts.h"

__global__ __launch_bounds__(256) void blk_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ res, const float* __restrict__ sp,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             double* partial, double* se_partial, float* __restrict__ ds_out, long E, long L,
                                                             long span, int C, int G, int cg, int lddy) {
  __shared__ double sh[256 * 8];
  const int unit = blockIdx.y, n = unit / G, gs = unit % G;
  const long unitBase = (long)gs * L;
  const long lo = (long)blockIdx.x * span;
  long hi = lo + span;
  if (hi > L) hi = L;
  const int F4 = C >> 2;
  const int cph = (int)((unitBase + lo + threadIdx.x * 4) % C);      // fixed per thread: 1024 % C == 0, span % 1024 == 0
  float gam[4], bet[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = gs * cg + ((cph + e) % cg);
    gam[e] = gamma[idx];
    bet[e] = beta[idx];
  }
  const float m_s = mean[unit], rs_s = rstd[unit];
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, pa[4] = {0, 0, 0, 0}, pb[4] = {0, 0, 0, 0};
  const float* xb = x + (long)n * E + unitBase;
  const float* rb = res + (long)n * E + unitBase;
  const long pix0 = ((long)n * E + unitBase + lo + threadIdx.x * 4) / C;      // voxel index over the batch
  const long pstep = 1024 / C;
  const bool head = (threadIdx.x & (F4 - 1)) == 0;
  auto one = [&](const f32x4 v, const f32x4 r, const f32x4 d, float s, long pix) {
    float t = (d[0] * r[0] + d[1] * r[1]) + (d[2] * r[2] + d[3] * r[3]);
    for (int o = F4 >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const float dsv = t * s * (1.f - s);
    if (head) ds_out[pix] = dsv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - m_s) * rs_s;
      float de = d[e];
      if (!(xh * gam[e] + bet[e] > 0.f)) de = 0.f;
      a[e] += (double)(de * xh);
      b[e] += (double)de;
      pa[e] += (double)(d[e] * r[e]);
      pb[e] += (double)(dsv * r[e]);
    }
  };
  // (hi - lo is a multiple of 1024: the lanes of a voxel leave the loop together.)  Two chunks per trip, their six loads issued first.
  const long K = (hi - lo) / 1024;
  const long i0 = lo + threadIdx.x * 4;
  long k = 0;
  for (; k + 2 <= K; k += 2) {
    f32x4 v[2], r[2], d[2];
    float s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = i0 + (k + u) * 1024, pix = pix0 + (k + u) * pstep;
      v[u] = *reinterpret_cast<const f32x4*>(xb + i);
      r[u] = *reinterpret_cast<const f32x4*>(rb + i);
      d[u] = *reinterpret_cast<const f32x4*>(dy + pix * lddy + cph);
      s[u] = sp[pix];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) one(v[u], r[u], d[u], s[u], pix0 + (k + u) * pstep);
  }
  for (; k < K; ++k) {
    const long i = i0 + k * 1024, pix = pix0 + k * pstep;
    one(*reinterpret_cast<const f32x4*>(xb + i), *reinterpret_cast<const f32x4*>(rb + i), *reinterpret_cast<const f32x4*>(dy + pix * lddy + cph), sp[pix],
        pix);
  }
  // GroupNorm class sums, exactly as gn_bwd_reduce_kernel (groupnorm.hip) leaves them
#pragma unroll
  for (int e = 0; e < 4; ++e) { sh[threadIdx.x * 8 + e] = a[e]; sh[threadIdx.x * 8 + 4 + e] = b[e]; }
  __syncthreads();
  for (int j = threadIdx.x; j < cg; j += 256) {
    double sa = 0.0, sb = 0.0;
    if (cg >= 4) {
      const int e = j & 3, rr = j >> 2, P4 = cg >> 2;
      for (int q = rr; q < 256; q += P4) { sa += sh[q * 8 + e]; sb += sh[q * 8 + 4 + e]; }
    } else {
      for (int q = 0; q < 256; ++q)
        for (int e = j; e < 4; e += cg) { sa += sh[q * 8 + e]; sb += sh[q * 8 + 4 + e]; }
    }
    const long o = (((long)unit * gridDim.x + blockIdx.x) * cg + j) * 2;
    partial[o] = sa;
    partial[o + 1] = sb;
  }
  __syncthreads();
  // gate sums, in se_bwd_reduce_kernel's layout (se.hip): block index inside the sample = gs * B + b
#pragma unroll
  for (int e = 0; e < 4; ++e) { sh[threadIdx.x * 8 + e] = pa[e]; sh[threadIdx.x * 8 + 4 + e] = pb[e]; }
  __syncthreads();
  const int vpb = 256 / F4;
  for (int col = threadIdx.x; col < C; col += 256) {
    const int e = col & 3, l = col >> 2;
    double sa = 0.0, sb = 0.0;
    for (int q = 0; q < vpb; ++q) { sa += sh[(q * F4 + l) * 8 + e]; sb += sh[(q * F4 + l) * 8 + 4 + e]; }
    const long o = ((((long)n * G + gs) * gridDim.x + blockIdx.x) * C + col) * 2;
    se_partial[o] = sa;
    se_partial[o + 1] = sb;
  }
}

__global__ __launch_bounds__(256) void blk_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ dres, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ c1,
                                                            const float* __restrict__ c2, const float* __restrict__ sp,
                                                            const float* __restrict__ ds, const float* __restrict__ ch,
                                                            const float* __restrict__ wsp, const float* __restrict__ dgap, long chunks_per_unit,
                                                            int cpb, int C, int cg, int G, int lddy) {
  const long chunk0 = (long)blockIdx.x * cpb;
  const long unit = chunk0 / chunks_per_unit;
  const int g = (int)(unit % G), n = (int)(unit / G);
  const float m = mean[unit], rs = rstd[unit], k1 = c1[unit], k2 = c2[unit];
  const int c = (threadIdx.x * 4) % C;
  float ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = g * cg + ((c + e) % cg);
    ga[e] = gamma[idx];
    be[e] = beta[idx];
  }
  const f32x4 chv = *reinterpret_cast<const f32x4*>(ch + (long)n * C + c);
  const f32x4 wsv = {wsp[c], wsp[c + 1], wsp[c + 2], wsp[c + 3]};      // (a parameter view: no 16-byte alignment promised)
  const f32x4 dgv = *reinterpret_cast<const f32x4*>(dgap + (long)n * C + c);
  const long off = chunk0 * 1024 + threadIdx.x * 4;
  const long vpc = 1024 / C;
  const long pix = chunk0 * vpc + (threadIdx.x * 4) / C;
  const long doff = pix * lddy + c;
  const long dstep = vpc * lddy;
  auto one = [&](const f32x4 v, const f32x4 d, float s, float dsv, long o_) {
    f32x4 o, q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - m) * rs;
      float de = d[e];
      if (!(xh * ga[e] + be[e] > 0.f)) de = 0.f;
      o[e] = (de * ga[e] - k1 - xh * k2) * rs;
      q[e] = d[e] * (s + chv[e]) + dsv * wsv[e] + dgv[e];
    }
    *reinterpret_cast<f32x4*>(dx + o_) = o;
    *reinterpret_cast<f32x4*>(dres + o_) = q;
  };
  int k = 0;
  for (; k + 1 < cpb; k += 2) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + off + (long)k * 1024);
    const f32x4 d0 = *reinterpret_cast<const f32x4*>(dy + doff + (long)k * dstep);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + off + (long)(k + 1) * 1024);
    const f32x4 d1 = *reinterpret_cast<const f32x4*>(dy + doff + (long)(k + 1) * dstep);
    const float s0 = sp[pix + (long)k * vpc], s1 = sp[pix + (long)(k + 1) * vpc];
    const float t0 = ds[pix + (long)k * vpc], t1 = ds[pix + (long)(k + 1) * vpc];
    one(v0, d0, s0, t0, off + (long)k * 1024);
    one(v1, d1, s1, t1, off + (long)(k + 1) * 1024);
  }
  for (; k < cpb; ++k) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + off + (long)k * 1024);
    const f32x4 d0 = *reinterpret_cast<const f32x4*>(dy + doff + (long)k * dstep);
    one(v0, d0, sp[pix + (long)k * vpc], ds[pix + (long)k * vpc], off + (long)k * 1024);
  }
}

// Between the two passes, one launch for the two finalizes that only need the reduce pass's partials: workgroups [0, G) turn the
// GroupNorm class sums into dgamma / dbeta / c1 / c2 (gn_bwd_finalize_slab_kernel's work), the rest sum the gate partials per (n, c)
// (se_bwd_partial_reduce_kernel's).
__global__ __launch_bounds__(256) void blk_bwd_middle_kernel(const double* partial, const float* gamma, float* dgamma, float* dbeta, float* c1,
                                                             float* c2, const double* se_partial, double* red, int N, int G, int B, int cg,
                                                             double L, int accum, int F) {
  __shared__ double sh[256 * 2];
  if ((int)blockIdx.x < G) gn_bwd_finalize_slab_body(partial, gamma, dgamma, dbeta, c1, c2, N, G, B, cg, L, accum, blockIdx.x, sh);
  else se_bwd_partial_reduce_body(se_partial, red, N, G * B, F, (int)blockIdx.x - G);
}

// the shapes the pair of kernels takes: slab units of whole 1024-element chunks with C | 1024, classes that tile a workgroup
static bool blk_bwd_plan(int N, long V, int F, int R, int G, int* B, long* span) {
  if (N <= 0 || V <= 0 || R <= 0 || G <= 0 || F < 4 || F > 256 || (F & (F - 1)) != 0 || F % G != 0) return false;
  const long E = V * F;
  if (E % G != 0) return false;
  const long L = E / G;
  const int cg = F / G;
  if (L % 1024 != 0 || 1024 % F != 0 || 256 % cg != 0) return false;
  if (!bts_gn_slab_blocks_(N, V, F, G, B, span)) return false;
  return (long)N * G * (L / 1024) <= 0x7fffffffL;
}
extern "C" long bts_block_bwd_workspace(int N, long V, int F, int R, int G) {
  int B;
  long span;
  if (!blk_bwd_plan(N, V, F, R, G, &B, &span)) return -1;
  const int cg = F / G;
  return (long)N * G * B * cg * 2 * 8 + (long)N * G * 2 * 4 + 64      // GroupNorm partials, c1 / c2
         + (long)N * G * B * F * 2 * 8 + ((long)N * F * 3 + (long)N * R) * 8 + 128;      // gate partials, red, scratch
}
// dout (N,V,F) rows of lddo; res, c2 dense; dres, dc2 dense outputs; ds (N*V) and dgap (N,F) scratch outputs.  accumulate_gate_params: dw1,
// dw2, dwsp += (else =); accumulate_norm_params: dgamma, dbeta likewise.  BTS_ERR_UNSUPPORTED outside the kernels' tiling (the workspace
// query returns -1 there): the caller runs bts_gn_bwd and bts_se_bwd.
extern "C" int bts_block_bwd(const float* dout, int lddo, const float* res, const float* c2x, const float* sp, const float* gap, const float* h,
                             const float* ch, const float* w1, const float* w2, const float* wsp, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, float* dres, float* dc2, float* ds, float* dgap, float* dw1, float* dw2,
                             float* dwsp, float* dgamma, float* dbeta, void* workspace, long workspace_bytes, int N, long V, int F, int R, int G,
                             int accumulate_gate_params, int accumulate_norm_params, hipStream_t stream) {
  int B;
  long span;
  if (!blk_bwd_plan(N, V, F, R, G, &B, &span)) return BTS_ERR_UNSUPPORTED;
  if (lddo < F || lddo % 4 != 0) return BTS_ERR_ALIGN;
  if ((((uintptr_t)dout) & 15) || (((uintptr_t)res) & 15) || (((uintptr_t)c2x) & 15) || (((uintptr_t)dres) & 15) || (((uintptr_t)dc2) & 15) ||
      (((uintptr_t)ch) & 15) || (((uintptr_t)dgap) & 15))
    return BTS_ERR_ALIGN;
  if (workspace == nullptr || (((uintptr_t)workspace) & 15) || workspace_bytes < bts_block_bwd_workspace(N, V, F, R, G)) return BTS_ERR_WORKSPACE;
  const long E = V * F, L = E / G;
  const int cg = F / G;
  double* partial = reinterpret_cast<double*>(workspace);
  float* c1 = reinterpret_cast<float*>(partial + (long)N * G * B * cg * 2);
  float* c2 = c1 + (long)N * G;
  double* sep = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(c2 + (long)N * G) + 63) & ~(uintptr_t)63);
  double* red = sep + (long)N * G * B * F * 2;
  double* scratch = red + (long)N * F * 2;
  (void)hipGetLastError();
  hipLaunchKernelGGL(blk_bwd_reduce_kernel, dim3(B, N * G), dim3(256), 0, stream, c2x, dout, res, sp, gamma, beta, mean, rstd, partial, sep, ds, E, L,
                     span, F, G, cg, lddo);
  BTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(blk_bwd_middle_kernel, dim3(G + (N * F + 3) / 4), dim3(256), 0, stream, partial, gamma, dgamma, dbeta, c1, c2, sep, red, N, G, B, cg,
                     (double)L, accumulate_norm_params, F);
  BTS_LAUNCH_CHECK();
  const int r = bts_se_mlp_bwd_(red, scratch, gap, h, ch, w1, w2, dw1, dw2, dwsp, dgap, N, G * B, V, F, R, accumulate_gate_params, stream);
  if (r != BTS_OK) return r;
  const long cpu = L / 1024;
  int cpb = 8;
  while (cpb > 1 && cpu % cpb != 0) cpb >>= 1;
  const long nblk = (long)N * G * cpu / cpb;
  hipLaunchKernelGGL(blk_bwd_apply_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, c2x, dout, dc2, dres, gamma, beta, mean, rstd, c1, c2, sp, ds, ch,
                     wsp, dgap, cpu, cpb, F, cg, G, lddo);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
