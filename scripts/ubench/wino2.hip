// Feasibility bench: 3x3x3 stride-1 convolution with the (z,y) plane in Winograd F(2x2,3x3) form and the x axis direct,
// on the exact-fp32 matrix pipe.  12 MFMAs per output voxel and (cin pair) instead of 27.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wino wino.hip && ./wino [C] [K] [D]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

struct WinoParams {
  const float* x;
  const float* up;
  const float* bias;
  float* y;
  int N, D, H, W, Cin, ldx, Cout, ldy, Npad, KG;
  int ntz, nty, ntx, tpw, nzc;
  int dbg;
  long long* stamps;
};

#define WS 12
#define WIX 34
#define WIY 6
#define WIZ 6
#define WVOX (WIX * WIY * WIZ)
#define WBUF (WVOX * WS)
#define WNSLOT 10

#define LDSOFF(i, j, dx) ((((i) * WIY + (j)) * WIX + (dx)) * WS)

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_add(a.xy, b.xy), hi = pk_add(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x2 s, f32x4 c) {  // a*s + c
  const f32x2 lo = pk_fma(a.xy, s, c.xy), hi = pk_fma(a.zw, s, c.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void wino_yt(const f32x4 (&c)[4], f32x4 (&v)[4]) {
  v[0] = sub4(c[0], c[2]);
  v[1] = add4(c[1], c[2]);
  v[2] = sub4(c[2], c[1]);
  v[3] = sub4(c[1], c[3]);
}
__device__ __forceinline__ void wino_mfma16(const f32x4 (&v)[4], const f32x4 (&a)[4], f32x16 (&acc)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e][j], v[e][j], acc[e], 0, 0, 0);
}
__device__ __forceinline__ float acc_rd(float a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

// Two waves per SIMD: 512 threads, the 16 transform points of a 2x2 patch split over a wave PAIR:
//   wave A (even): xi_z = 0 (d0 - d2), 1 (d1 + d2)      wave B (odd): xi_z = 3 (d1 - d3), 2 (d2 - d1)
// 8 accumulators = 128 registers per wave.  Both waves run the same code on rows (X, Y, Z) = (0, 2, 1) / (3, 1, 2):
//   group 0: c = X - Y   (B: the negative of xi_z = 3, undone in the output transform)     group 1: c = Z + s*Y, s = +1 / -1
// LDS tile as two channel-quad planes [h][voxel][4] filled by global->LDS DMA (no staging registers, no padding).
#define W2IZ 4
#define W2VOX (WIX * WIY * W2IZ)   /* 816 */
#define W2PLANE (W2VOX * 4)
#define W2BUF (28 * 256)   /* dwords: 2 planes (6528) rounded up to the 7 x 4 DMA chunks of 1 KB */
#define W2SLOT 7
#define CELL(j, dx) (((j) * WIX + (dx)) * 4)

template <int DBG>
__global__ __launch_bounds__(256, 2) void wino2_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  long long ts0 = clock64(), ts1 = 0, ts2 = 0;
  const long long wc0 = wall_clock64();
  int b = blockIdx.x;
  const int tx = b % p.ntx; b /= p.ntx;
  const int ty = b % p.nty; b /= p.nty;
  const int tz = b % p.ntz;
  const int n = b / p.ntz;
  const int oz0 = tz * 2, oy0 = ty * 4, ox0 = tx * 32;
  const int iz0 = oz0 - 1, iy0 = oy0 - 1, ix0 = ox0 - 1;
  const float* xorg = p.x + ((((long)n * p.D + iz0) * p.H + iy0) * p.W + ix0) * (long)p.ldx;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xorg, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.up + (long)blockIdx.y * p.KG * (3 * 16 * 256)), 0, 0x7fffffff, 0x00020000);
  unsigned goff[W2SLOT];
#pragma unroll
  for (int i = 0; i < W2SLOT; ++i) {
    const int e = (i * 4 + wave) * 64 + lane;
    goff[i] = 0x80000000u;
    if (e < W2VOX * 2) {
      const int q = e >= W2VOX ? 1 : 0;
      const int vox = e - q * W2VOX;
      const int vz = vox / (WIY * WIX);
      const int r = vox - vz * (WIY * WIX);
      const int vy = r / WIX;
      const int vx = r - vy * WIX;
      if ((unsigned)(iz0 + vz) < (unsigned)p.D && (unsigned)(iy0 + vy) < (unsigned)p.H && (unsigned)(ix0 + vx) < (unsigned)p.W)
        goff[i] = (unsigned)(((vz * p.H + vy) * p.W + vx) * p.ldx + q * 4) * 4u;
    }
  }
  auto fetch = [&](int st, float* buf) {
#pragma unroll
    for (int i = 0; i < W2SLOT; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(buf + (i * 4 + wave) * 256), 16, goff[i],
                                               (unsigned)st * 32u, 0, 0);
  };
  const int pair = wave >> 1, wb = wave & 1;
  const int tz2 = 0, ty2 = pair;
  const int pb = h * W2PLANE + (((2 * tz2) * WIY + 2 * ty2) * WIX + l32) * 4;
  const int offX = pb + (wb ? 3 : 0) * (WIY * WIX * 4), offY = pb + (wb ? 1 : 2) * (WIY * WIX * 4), offZ = pb + (wb ? 2 : 1) * (WIY * WIX * 4);
  const int xz0 = wb ? 3 : 0, xz1 = wb ? 2 : 1;
  const float sgn = wb ? -1.f : 1.f;
  const f32x2 s2 = {sgn, sgn};
  const unsigned wlane = (unsigned)((h * 32 + l32) * 16);

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  fetch(0, lds);
  f32x4 aw[2][4];
  auto wload = [&](f32x4 (&dst)[4], int st, int dx, int xz) {
    const unsigned so = (unsigned)(((st * 3 + dx) * 16 + xz * 4) * 1024);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e] = bufload(wr, wlane + e * 1024, so);
  };
  wload(aw[0], 0, 0, xz0);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the tile's DMA was issued before the 4 weight loads
  __builtin_amdgcn_s_barrier();
  ts1 = clock64();

  const int nst = p.KG;
  for (int st = 0; st < nst; ++st) {
    const float* cur = lds + (st & 1) * W2BUF;
    float* nxt = lds + ((st + 1) & 1) * W2BUF;
    const bool more = (st + 1) < nst;
    if (more && DBG != 2) fetch(st + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);
    const int stn = more ? st + 1 : st;
    const float* lX = cur + offX;
    const float* lY = cur + offY;
    const float* lZ = cur + offZ;
    if (DBG != 1) {
      f32x4 rx[4], ry[4], v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ry[j] = *reinterpret_cast<const f32x4*>(lY + CELL(j, 0));
        rx[j] = *reinterpret_cast<const f32x4*>(lX + CELL(j, 0));
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        {  // group 0: c = X - Y (in place), then Z requested into X's registers
          wload(aw[1], st, dx, xz1);
#pragma unroll
          for (int j = 0; j < 4; ++j) rx[j] = sub4(rx[j], ry[j]);
          wino_yt(rx, v);
          asm volatile("s_nop 3" ::: "memory");  // hand-written VALU -> MFMA operand: the compiler's hazard tracking does not see it
#pragma unroll
          for (int j = 0; j < 4; ++j) rx[j] = *reinterpret_cast<const f32x4*>(lZ + CELL(j, dx));
          wino_mfma16(v, aw[0], acc[0]);
          __builtin_amdgcn_sched_barrier(0);
        }
        {  // group 1: c = Z + s*Y (in place), then the next x tap's Y, X requested
          if (dx < 2) wload(aw[0], st, dx + 1, xz0); else wload(aw[0], stn, 0, xz0);
#pragma unroll
          for (int j = 0; j < 4; ++j) rx[j] = fma4(ry[j], s2, rx[j]);
          wino_yt(rx, v);
          asm volatile("s_nop 3" ::: "memory");  // hand-written VALU -> MFMA operand: the compiler's hazard tracking does not see it
          if (dx < 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ry[j] = *reinterpret_cast<const f32x4*>(lY + CELL(j, dx + 1));
              rx[j] = *reinterpret_cast<const f32x4*>(lX + CELL(j, dx + 1));
            }
          }
          wino_mfma16(v, aw[1], acc[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (and the one prefetched weight group)
    __builtin_amdgcn_s_barrier();
  }
  ts2 = clock64();

  // ---- output transform; the pair exchanges one partial through LDS: A finishes plane oz = 0, B plane oz = 1 ----
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float* xbuf = lds;  // staging buffers are idle
  f32x4 keep[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 t0[2], t1[2];
    {
      f32x4 q[4];
#pragma unroll
      for (int xy = 0; xy < 4; ++xy)
        q[xy] = f32x4{acc_rd(acc[0][xy][4 * g]), acc_rd(acc[0][xy][4 * g + 1]), acc_rd(acc[0][xy][4 * g + 2]), acc_rd(acc[0][xy][4 * g + 3])};
      t0[0] = q[0] + q[1] + q[2];
      t0[1] = q[1] - q[2] - q[3];
#pragma unroll
      for (int xy = 0; xy < 4; ++xy)
        q[xy] = f32x4{acc_rd(acc[1][xy][4 * g]), acc_rd(acc[1][xy][4 * g + 1]), acc_rd(acc[1][xy][4 * g + 2]), acc_rd(acc[1][xy][4 * g + 3])};
      t1[0] = q[0] + q[1] + q[2];
      t1[1] = q[1] - q[2] - q[3];
    }
    // A: t0 = T0, t1 = T1 -> keep T0 + T1 (plane 0), send T1.   B: t0 = -T3, t1 = T2 -> keep -T3 - T2 (plane 1), send T2
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      keep[g][oy] = wb ? (t0[oy] - t1[oy]) : (t0[oy] + t1[oy]);
      *reinterpret_cast<f32x4*>(xbuf + ((wave * 8 + g * 2 + oy) * 64 + lane) * 4) = t1[oy];
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  const int oxx = ox0 + l32;
  const int zb = oz0 + 2 * tz2 + wb, yb = oy0 + 2 * ty2;
  float* ybase = p.y + ((((long)n * p.D + zb) * p.H + yb) * p.W + oxx) * (long)p.ldy + blockIdx.y * 32 + 4 * h;
  const long ysY = (long)p.W * p.ldy;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int co = blockIdx.y * 32 + 8 * g + 4 * h;
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && co < p.Cout) bq = f32x4{p.bias[co], p.bias[co + 1], p.bias[co + 2], p.bias[co + 3]};
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      const f32x4 other = *reinterpret_cast<const f32x4*>(xbuf + (((wave ^ 1) * 8 + g * 2 + oy) * 64 + lane) * 4);
      const f32x4 o = keep[g][oy] + other + bq;   // A: T0+T1+T2   B: -T3-T2+T1
      if (oxx < p.W && yb + oy < p.H && zb < p.D && co < p.Cout) *reinterpret_cast<f32x4*>(ybase + oy * ysY + 8 * g) = o;
    }
  }
  if (tid == 0) {
    const long long ts3 = clock64();
    long long* o = p.stamps + (long)blockIdx.x * 4;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
    if (blockIdx.x == 777) { p.stamps[(long)gridDim.x * 4] = ts3 - ts0; p.stamps[(long)gridDim.x * 4 + 1] = wall_clock64() - wc0; }
  }
}

__global__ void ref_kernel(const float* x, const float* w, const float* bias, float* y, int D, int H, int W, int C, int K,
                           int z0, int nz) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long total = (long)nz * H * W * K;
  if (i >= total) return;
  const int k = i % K;
  long v = i / K;
  const int xx = v % W; v /= W;
  const int yy = v % H; v /= H;
  const int zz = z0 + (int)v;
  float s = bias ? bias[k] : 0.f;
  for (int kz = 0; kz < 3; ++kz)
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int iz = zz + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
        if ((unsigned)iz >= (unsigned)D || (unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        const float* xr = x + (((long)iz * H + iy) * W + ix) * C;
        const float* wr = w + (long)((kz * 3 + ky) * 3 + kx) * C * K + k;
        for (int c = 0; c < C; ++c) s = fmaf(xr[c], wr[(long)c * K], s);
      }
  y[(((long)zz * H + yy) * W + xx) * K + k] = s;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 32, K = argc > 2 ? atoi(argv[2]) : 32, D = argc > 3 ? atoi(argv[3]) : 128;
  const int dbg = argc > 4 ? atoi(argv[4]) : 0;
  const int H = D, W = D;
  const long nvox = (long)D * H * W;
  std::vector<float> hx(nvox * C), hw(27L * C * K), hb(K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.2f;
  for (auto& v : hb) v = rnd();
  const int Npad = (K + 31) / 32 * 32, KG = C / 8;
  // U = G g G^T over (kz, ky) per (kx, c, k)
  const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  std::vector<float> hup((long)KG * 3 * 16 * 2 * Npad * 4, 0.f);
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < K; ++k)
      for (int dx = 0; dx < 3; ++dx)
        for (int xz = 0; xz < 4; ++xz)
          for (int xy = 0; xy < 4; ++xy) {
            double u = 0;
            for (int kz = 0; kz < 3; ++kz)
              for (int ky = 0; ky < 3; ++ky) u += G[xz][kz] * G[xy][ky] * hw[((long)((kz * 3 + ky) * 3 + dx) * C + c) * K + k];
            const int kg = c / 8, hh = (c % 8) / 4, e = c % 4;
            hup[(((((long)(k / 32) * KG + kg) * 3 + dx) * 16 + xz * 4 + xy) * 2 + hh) * 128 + (k % 32) * 4 + e] = (float)u;
          }
  float *dx_, *dw, *db, *dy, *dref, *dup;
  hipMalloc(&dx_, hx.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&db, hb.size() * 4);
  hipMalloc(&dy, nvox * K * 4); hipMalloc(&dup, hup.size() * 4 + 65536);
  const int nzref = 6;
  hipMalloc(&dref, (long)nzref * H * W * K * 4 * 2);
  hipMemcpy(dx_, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dup, hup.data(), hup.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dy, 0, nvox * K * 4);
  WinoParams p{};
  p.x = dx_; p.up = dup; p.bias = getenv("NOBIAS") ? nullptr : db; p.y = dy; p.N = 1; p.D = D; p.H = H; p.W = W; p.Cin = C; p.ldx = C; p.Cout = K; p.ldy = K;
  p.Npad = Npad; p.KG = KG; p.ntz = (D + 1) / 2; p.nty = (H + 3) / 4; p.ntx = (W + 31) / 32; p.dbg = dbg;
  p.tpw = argc > 5 ? atoi(argv[5]) : 8; p.nzc = (p.ntz + p.tpw - 1) / p.tpw;
  const int ldsb = 2 * W2BUF * 4;
  void (*kern)(const WinoParams) = dbg == 1 ? wino2_kernel<1> : dbg == 2 ? wino2_kernel<2> : wino2_kernel<0>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  dim3 grid(p.ntz * p.nty * p.ntx, Npad / 32);
  hipMalloc(&p.stamps, ((long)grid.x * 4 + 2) * 8);
  hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, 0, p);
  hipError_t e = hipDeviceSynchronize();
  printf("launch: %s  grid %d x %d lds %d\n", hipGetErrorString(e), grid.x, grid.y, ldsb);
  // check first and last nzref/2 planes + a middle band
  double maxerr = 0, maxref = 0;
  std::vector<float> hy((long)nzref * H * W * K), hr((long)nzref * H * W * K);
  const int z0s[3] = {0, D / 2 - 3, D - nzref};
  for (int q = 0; q < 3; ++q) {
    const int z0 = z0s[q];
    float* dr = dref;
    const long tot = (long)nzref * H * W * K;
    hipLaunchKernelGGL(ref_kernel, dim3((tot + 255) / 256), dim3(256), 0, 0, dx_, dw, db, dr - (long)z0 * H * W * K, D, H, W, C, K, z0, nzref);
    hipMemcpy(hr.data(), dr, tot * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hy.data(), dy + (long)z0 * H * W * K, tot * 4, hipMemcpyDeviceToHost);
    for (long i = 0; i < tot; ++i) {
      const double d = fabs((double)hy[i] - hr[i]);
      if (d > maxerr) maxerr = d;
      if (fabs(hr[i]) > maxref) maxref = fabs(hr[i]);
    }
  }
  printf("max abs err %.3e  (max |ref| %.3f, rel %.2e)\n", maxerr, maxref, maxerr / maxref);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, 0, p);
  hipEventRecord(e0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, 0, p);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  const double fl = 2.0 * 27 * C * K * nvox;
  printf("C=%d K=%d D=%d: %.3f ms  direct-equivalent %.1f TF  (matrix pipe %.1f TF)\n", C, K, D, ms, fl / ms * 1e-9, fl * 12 / 27 / ms * 1e-9);
  {
    std::vector<long long> st((long)grid.x * 4 + 2);
    hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b2 = 0, c = 0;
    for (unsigned i = 0; i < grid.x; ++i) { a += st[i * 4 + 1] - st[i * 4]; b2 += st[i * 4 + 2] - st[i * 4 + 1]; c += st[i * 4 + 3] - st[i * 4 + 2]; }
    printf("shader clock in WG 777: %.0f MHz (clock64 %lld ticks in %lld x 10 ns)\n", 100.0 * st[(long)grid.x * 4] / st[(long)grid.x * 4 + 1], st[(long)grid.x * 4], st[(long)grid.x * 4 + 1]);
    printf("clock64 ticks per WG: prologue %.0f  loop %.0f  epilogue %.0f\n", a / grid.x, b2 / grid.x, c / grid.x);
  }
  return 0;
}
