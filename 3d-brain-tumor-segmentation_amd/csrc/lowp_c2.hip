// The FIRST ResnetBlock's two convolutions of the raw 2-channel volume on 16-bit storage, forward without a backward (round 5):
//   c1  = conv3x3x3(x) + b3   (resnet.py:80-87, conv1 of encoder level 0: 2 -> F channels)   + GroupNorm-1's slab statistics of c1
//   res = conv1x1x1(x) + b1   (resnet.py:30-37,118, the block's shortcut)                     + the per-channel mean of x (for the squeeze)
// from ONE pass over the fp32 input (model.py:63-68 with in_ch = 2, args.py:105-143; test.py:181-270 drives it).
//
// Why its own kernel: the generic 16-bit kernels contract over whole 16-channel matrix steps, so the 2-channel volume had to be cast into
// a zero-padded 16-channel tensor (lp_cast_pad16: 157 MB written at 160 x 192 x 160) which the shortcut kernel and the z-marching conv then
// each read back -- 14 of every 16 bytes zeros, 27 matrix instructions per 32 x 32 block for 54 products.  Here the contraction index is
// (tap, channel): k = 2 * tap + c, 54 of 64 slots live, FOUR v_mfma_f32_32x32x16 per 32 positions x 32 couts instead of 27; the input is
// read as fp32 (8 bytes per voxel) and rounded to the storage type on the way into LDS -- the same values the cast produced, so the
// results are those of the three-kernel route up to fp32 summation order.
//
// Work decomposition: 256 threads = 4 waves; a workgroup owns a 32 (x) x 4 (y) x 4 (z) output tile (halo 34 x 6 x 6 voxels x 4 bytes in
// LDS), wave w its z plane w: four rows of 32 positions, for each 4 + 1 matrix instructions (c1's four k-steps, the shortcut's one).  A B
// operand (lane = position, k-half h) is four ds_read_b32 -- the (2-channel) voxels of taps 8 s + 4 h + {0..3}; the A operands (lane =
// cout) are formed once per workgroup from the fp32 weights.  Outputs leave as 16-byte stores of 8 consecutive couts (lowp_s1d.hip's
// v_permlane32_swap exchange).  GroupNorm partial sums: one fp64 (sum, sumsq) pair per (plane, tile column) in bts_gn_finalize_partials_'s
// layout.  The squeeze: GAP(res)[co] = sum_c mean_v(x[v][c]) W1[c][co] + b1[co] is linear in x, so the kernel only leaves the per-tile sums
// of the (rounded) input channels; lp_c2_gap_kernel finishes it.  HBM-bound: 8 bytes in, 4 F bytes out per voxel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "bts_internal.h"
#include "lowp_common.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

struct LpC2Params {
  const float* x;        // (N, D, H, W, 2) fp32, dense
  const float* w3;       // (3, 3, 3, 2, F) fp32
  const float* b3;       // (F)
  const float* w1;       // (1, 1, 1, 2, F)
  const float* b1;       // (F)
  unsigned short* c1;    // (N, D, H, W, F) storage type, dense
  unsigned short* res;
  double* gnp;           // [N * G][B][2]
  double* xsum;          // [N][tiles per sample][2]
  int N, D, H, W, F, G, zt;
  int ntx, nty, ntz;
  long gn_B;
};

#define C2_TX 32
#define C2_TY 4
#define C2_TZ 4
#define C2_SX (C2_TX + 2)
#define C2_SY (C2_TY + 2)
#define C2_SZ (C2_TZ + 2)
#define C2_NVOX (C2_SX * C2_SY * C2_SZ)      // 1224

// (first version, one tile per workgroup, all four rows' accumulators live: 328 us at 160 x 192 x 160 -- 2 workgroups per CU, each a serial
// chain of fetch -> LDS -> 20 matrix instructions -> 32 stores; the three kernels it replaces took 291.  Now: PERSISTENT workgroups, four per
// CU (<= 128 registers: one row's accumulators at a time), the next tile's voxels in flight in registers while the current tile multiplies,
// the operand fragments and biases formed once per workgroup.)
#define C2_NLD ((C2_NVOX + 255) / 256)      // halo voxels a thread stages per tile (5)

template <typename T>
__global__ __launch_bounds__(256, 4) void lp_c2_kernel(const LpC2Params p, long total_tiles) {
  __shared__ unsigned tile[2][C2_NVOX];         // one voxel = its two channels in the storage type; double buffered
  __shared__ double red[2][4][2];
  __shared__ float shb[2][32];                  // biases of conv1 / the shortcut
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  if (tid < 64) shb[tid >> 5][tid & 31] = ((tid & 31) < p.F) ? ((tid >> 5) ? (p.b1 ? p.b1[tid & 31] : 0.f) : (p.b3 ? p.b3[tid & 31] : 0.f)) : 0.f;
  // ---- A operands (lane = cout l32, k-half h): k = 16 s + 8 h + j <-> tap 8 s + 4 h + (j >> 1), channel j & 1 ----
  u32x4 a3[4], a1;
  {
    const bool co_ok = l32 < p.F;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tap = 8 * s + 4 * h + i;
        float w0 = 0.f, w1v = 0.f;
        if (co_ok && tap < 27) { w0 = p.w3[(tap * 2 + 0) * p.F + l32]; w1v = p.w3[(tap * 2 + 1) * p.F + l32]; }
        a3[s][i] = pack2<T>(w0, w1v);
      }
    }
    a1 = u32x4{0u, 0u, 0u, 0u};
    if (co_ok && h == 0) a1[0] = pack2<T>(p.w1[l32], p.w1[p.F + l32]);      // the shortcut's two products sit in k = 0, 1
  }
  // halo voxels this thread stages: index inside the tile, (vz, vy, vx), interior flag
  int hv[C2_NLD];
#pragma unroll
  for (int i = 0; i < C2_NLD; ++i) {
    const int v = tid + 256 * i;
    const int vz = v / (C2_SY * C2_SX), r = v - vz * (C2_SY * C2_SX), vy = r / C2_SX, vx = r - vy * C2_SX;
    const bool inner = vz >= 1 && vz <= C2_TZ && vy >= 1 && vy <= C2_TY && vx >= 1 && vx <= C2_TX;
    hv[i] = v < C2_NVOX ? (vx | (vy << 8) | (vz << 16) | (inner ? 1 << 24 : 0)) : -1;
  }
  // this lane's tap offsets (voxels of the halo tile) per k-step; taps >= 27 multiply zero weights: any valid address
  int toff[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int tap = 8 * s + 4 * h + i;
      if (tap > 26) tap = 26;
      toff[s][i] = ((tap / 9) * C2_SY + (tap / 3) % 3) * C2_SX + tap % 3;
    }
  struct Org { int n, tx, ty, tz; };
  auto origin = [&](long t) {
    Org o;
    o.tx = (int)(t % p.ntx); t /= p.ntx;
    o.ty = (int)(t % p.nty); t /= p.nty;
    o.tz = (int)(t % p.ntz);
    o.n = (int)(t / p.ntz);
    return o;
  };
  float2 pre[C2_NLD];
  auto prefetch = [&](long t) {      // the fp32 voxels of tile t into registers (zeros outside the volume: 'same' padding)
    const Org o = origin(t);
    const int oz0 = o.tz * C2_TZ - 1, oy0 = o.ty * C2_TY - 1, ox0 = o.tx * C2_TX - 1;
#pragma unroll
    for (int i = 0; i < C2_NLD; ++i) {
      pre[i] = float2{0.f, 0.f};
      if (hv[i] >= 0) {
        const int z = oz0 + ((hv[i] >> 16) & 0xff), y = oy0 + ((hv[i] >> 8) & 0xff), xx = ox0 + (hv[i] & 0xff);
        if ((unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
          pre[i] = *reinterpret_cast<const float2*>(p.x + ((((long)o.n * p.D + z) * p.H + y) * p.W + xx) * 2);
      }
    }
  };
  long t = blockIdx.x;
  if (t >= total_tiles) return;
  prefetch(t);
  int buf = 0;
  for (; t < total_tiles; t += gridDim.x, buf ^= 1) {
    const Org o = origin(t);
    // ---- the prefetched voxels -> storage type -> LDS; the tile's own voxels also feed the squeeze ----
    double sx0 = 0.0, sx1 = 0.0;
#pragma unroll
    for (int i = 0; i < C2_NLD; ++i) {
      if (hv[i] >= 0) {
        const unsigned pk = pack2<T>(pre[i].x, pre[i].y);
        tile[buf][tid + 256 * i] = pk;
        if (hv[i] & (1 << 24)) { sx0 += (double)T::ld((unsigned short)(pk & 0xffffu)); sx1 += (double)T::ld((unsigned short)(pk >> 16)); }
      }
    }
    {
      const double w0 = wave_sum_f64(sx0), w1s = wave_sum_f64(sx1);
      if (lane == 0) { red[buf][wave][0] = w0; red[buf][wave][1] = w1s; }
    }
    __syncthreads();      // (the one barrier per tile: whoever passes it has finished reading the OTHER buffer, which the next tile overwrites)
    if (t + gridDim.x < total_tiles) prefetch(t + gridDim.x);      // in flight under this tile's arithmetic and stores
    if (tid == 0) {
      double* xs = p.xsum + ((long)o.n * (p.ntx * p.nty * p.ntz) + ((long)o.tz * p.nty + o.ty) * p.ntx + o.tx) * 2;
      xs[0] = (red[buf][0][0] + red[buf][1][0]) + (red[buf][2][0] + red[buf][3][0]);
      xs[1] = (red[buf][0][1] + red[buf][1][1]) + (red[buf][2][1] + red[buf][3][1]);
    }
    // ---- plane `wave` of the tile, one row of 32 positions at a time: 4 + 1 matrix instructions, then its stores ----
    const int oz = o.tz * C2_TZ + wave, ox = o.tx * C2_TX + l32;
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.c1 + (long)o.n * p.D * p.H * p.W * (long)p.F), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + (long)o.n * p.D * p.H * p.W * (long)p.F), 0, 0x7fffffff, 0x00020000);
    float gn_s = 0.f, gn_q = 0.f;
#pragma unroll 1
    for (int r = 0; r < C2_TY; ++r) {
      f32x16 acc3, acc1;      // register 4 q + j belongs to cout 8 q + 4 h + j: start at the bias
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b3q = *reinterpret_cast<const f32x4*>(&shb[0][8 * q + 4 * h]), b1q = *reinterpret_cast<const f32x4*>(&shb[1][8 * q + 4 * h]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc3[4 * q + j] = b3q[j]; acc1[4 * q + j] = b1q[j]; }
      }
      const int base = (wave * C2_SY + r) * C2_SX + l32;      // halo voxel of tap (0, 0, 0) of output (x = l32, y = r, z = wave)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4 bf = {tile[buf][base + toff[s][0]], tile[buf][base + toff[s][1]], tile[buf][base + toff[s][2]], tile[buf][base + toff[s][3]]};
        acc3 = T::mfma(a3[s], bf, acc3);
      }
      const unsigned ctr = tile[buf][base + (C2_SY + 1) * C2_SX + 1];     // the centre tap
      acc1 = T::mfma(a1, u32x4{h == 0 ? ctr : 0u, 0u, 0u, 0u}, acc1);
      const int oy = o.ty * C2_TY + r;
      const bool vox_ok = oz < p.D && oy < p.H && ox < p.W;
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int co = 16 * qp + 8 * h;
        const bool ok = co < p.F && vox_ok;
        const unsigned off = ok ? (unsigned)((((oz * p.H + oy) * p.W + ox) * p.F + co) * 2) : 0x80000000u;
        float f[4], g2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = acc3[8 * qp + j]; g2[j] = acc3[8 * qp + 4 + j]; }
        if (vox_ok) {       // (own values, before the exchange: couts 8 q + 4 h + j of q = 2 qp and 2 qp + 1)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (16 * qp + 4 * h + j < p.F) { gn_s += f[j]; gn_q = fmaf(f[j], f[j], gn_q); }
            if (16 * qp + 8 + 4 * h + j < p.F) { gn_s += g2[j]; gn_q = fmaf(g2[j], g2[j], gn_q); }
          }
        }
        unsigned d0 = pack2<T>(f[0], f[1]), d1 = pack2<T>(f[2], f[3]), d2 = pack2<T>(g2[0], g2[1]), d3 = pack2<T>(g2[2], g2[3]);
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{d0, d1, d2, d3}, cr, off, 0, 0);
        unsigned e0 = pack2<T>(acc1[8 * qp], acc1[8 * qp + 1]), e1 = pack2<T>(acc1[8 * qp + 2], acc1[8 * qp + 3]),
                 e2 = pack2<T>(acc1[8 * qp + 4], acc1[8 * qp + 5]), e3 = pack2<T>(acc1[8 * qp + 6], acc1[8 * qp + 7]);
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{e0, e1, e2, e3}, rr, off, 0, 0);
      }
    }
    {   // GroupNorm-1 partial pair of this plane and tile column
      const double ds = wave_sum_f64((double)gn_s), dq = wave_sum_f64((double)gn_q);
      if (lane == 0 && oz < p.D) {
        const int gg = oz / p.zt;
        const long slot = ((long)(oz - gg * p.zt) * p.nty + o.ty) * p.ntx + o.tx;
        double* dst = p.gnp + (((long)o.n * p.G + gg) * p.gn_B + slot) * 2;
        dst[0] = ds;
        dst[1] = dq;
      }
    }
  }
}

// gap[n][co] = mean_v(res[n][v][co]) = sum_c mean_v(x_c) * W1[c][co] + b1[co] with x and W1 as the conv saw them (rounded to the storage type)
template <typename T>
__global__ __launch_bounds__(1024) void lp_c2_gap_kernel(const double* xsum, const float* w1, const float* b1, float* gap, int tiles, int F, double inv_v) {
  // (one workgroup per sample; 9600 tile rows at 160 x 192 x 160: a single wave walking them took 55 us -- 16 waves, fixed order)
  __shared__ double sh[16][2];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s0 = 0.0, s1 = 0.0;
  for (int t = tid; t < tiles; t += 1024) { s0 += xsum[((long)n * tiles + t) * 2]; s1 += xsum[((long)n * tiles + t) * 2 + 1]; }
  s0 = wave_sum_f64(s0); s1 = wave_sum_f64(s1);
  if (lane == 0) { sh[wave][0] = s0; sh[wave][1] = s1; }
  __syncthreads();
  if (tid < F) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { t0 += sh[k][0]; t1 += sh[k][1]; }
    const double w0 = (double)T::ld(T::st(w1[tid])), w1v = (double)T::ld(T::st(w1[F + tid]));
    gap[(long)n * F + tid] = (float)(t0 * inv_v * w0 + t1 * inv_v * w1v + (b1 ? (double)b1[tid] : 0.0));
  }
}

static bool c2_takes(int N, int D, int H, int W, int F, int G) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || F < 8 || F > 32 || F % 8 != 0 || G <= 0 || D % G != 0 || F % G != 0) return false;
  if ((long)D * H * W * (long)F * 2 >= 0x7fffffffL) return false;      // 31-bit byte offsets inside one sample
  const char* e = getenv("BTS_LP_C2");      // =0: the cast + shortcut + conv route (A/B; read per call)
  return !(e && atoi(e) == 0);
}
static long c2_gn_B(int D, int H, int W, int G) { return (long)(D / G) * ((H + C2_TY - 1) / C2_TY) * ((W + C2_TX - 1) / C2_TX); }
static long c2_tiles(int D, int H, int W) { return (long)((D + C2_TZ - 1) / C2_TZ) * ((H + C2_TY - 1) / C2_TY) * ((W + C2_TX - 1) / C2_TX); }

extern "C" long bts_lp_first_block_workspace(int N, int D, int H, int W, int F, int G) {
  if (!c2_takes(N, D, H, W, F, G)) return -1;
  return (long)N * G * c2_gn_B(D, H, W, G) * 16 + (long)N * c2_tiles(D, H, W) * 16 + 64;
}
extern "C" int bts_lp_first_block_fwd(int dtype, const float* x, const float* w3, const float* b3, const float* w1, const float* b1, void* c1,
                                      void* res, float* mean, float* rstd, float* gap, void* workspace, long workspace_bytes, int N, int D,
                                      int H, int W, int F, int G, float eps, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  const long need = bts_lp_first_block_workspace(N, D, H, W, F, G);
  if (need < 0) return BTS_ERR_UNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < need || (((uintptr_t)workspace) & 15)) return BTS_ERR_WORKSPACE;
  if ((((uintptr_t)x) & 7) || (((uintptr_t)c1) & 15) || (((uintptr_t)res) & 15)) return BTS_ERR_ALIGN;
  LpC2Params p;
  p.x = x; p.w3 = w3; p.b3 = b3; p.w1 = w1; p.b1 = b1; p.c1 = (unsigned short*)c1; p.res = (unsigned short*)res;
  p.N = N; p.D = D; p.H = H; p.W = W; p.F = F; p.G = G; p.zt = D / G;
  p.ntx = (W + C2_TX - 1) / C2_TX; p.nty = (H + C2_TY - 1) / C2_TY; p.ntz = (D + C2_TZ - 1) / C2_TZ;
  p.gn_B = c2_gn_B(D, H, W, G);
  p.gnp = reinterpret_cast<double*>(workspace);
  p.xsum = p.gnp + (long)N * G * p.gn_B * 2;
  const long tiles = c2_tiles(D, H, W);
  const long blocks = (long)N * tiles;
  if (blocks > 0x7fffffffL) return BTS_ERR_SHAPE;
  const long V = (long)D * H * W;
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(40, 2.0 * 28.0 * 2.0 * F * (double)N * V, stream);
  (void)hipGetLastError();
  const long grid = blocks < 1024 ? blocks : 1024;      // four persistent workgroups per CU
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_c2_kernel<TF16>, dim3((unsigned)grid), dim3(256), 0, stream, p, blocks);
  else hipLaunchKernelGGL(lp_c2_kernel<TBF16>, dim3((unsigned)grid), dim3(256), 0, stream, p, blocks);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_c2_gap_kernel<TF16>, dim3(N), dim3(1024), 0, stream, p.xsum, w1, b1, gap, (int)tiles, F, 1.0 / (double)V);
  else hipLaunchKernelGGL(lp_c2_gap_kernel<TBF16>, dim3(N), dim3(1024), 0, stream, p.xsum, w1, b1, gap, (int)tiles, F, 1.0 / (double)V);
  BTS_LAUNCH_CHECK();
  return bts_gn_finalize_partials_(p.gnp, mean, rstd, N * G, p.gn_B, (double)(V * F / G), eps, stream);
}
