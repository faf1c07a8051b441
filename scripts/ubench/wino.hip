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

// y transform of one z-combined row set: B^T rows (1,0,-1,0) (0,1,1,0) (0,-1,1,0) (0,1,0,-1)
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
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_add(a.xy, b.xy), hi = pk_add(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void wino_yt(const f32x4 (&c)[4], f32x4 (&v)[4]) {
  v[0] = sub4(c[0], c[2]);
  v[1] = add4(c[1], c[2]);
  v[2] = sub4(c[2], c[1]);
  v[3] = sub4(c[1], c[3]);
}
// one group: xi_z fixed, 4 xi_y values, 4 channel pairs -> 16 MFMAs
__device__ __forceinline__ void wino_mfma16(const f32x4 (&v)[4], const f32x4 (&a)[4], f32x16 (&acc)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e][j], v[e][j], acc[e], 0, 0, 0);
}
// interleave: after every MFMA two vector-ALU operations (the NEXT group's transform) and the group's memory requests
#define WINO_SCHED_GROUP()                                 \
  _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);     \
  }

// DBG: 0 normal, 1 no sweep, 2 no re-staging; clock64 stamps (start / first barrier / loop end / exit) always written
template <int DBG>
__global__ __launch_bounds__(256, 1) void wino_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  long long ts0 = 0, ts1 = 0, ts2 = 0;
  ts0 = clock64();
  int b = blockIdx.x;
  const int tx = b % p.ntx; b /= p.ntx;
  const int ty = b % p.nty; b /= p.nty;
  const int tz = b % p.ntz;
  const int n = b / p.ntz;
  const int oz0 = tz * 4, oy0 = ty * 4, ox0 = tx * 32;
  const int iz0 = oz0 - 1, iy0 = oy0 - 1, ix0 = ox0 - 1;

  // halo origin of this tile; slots outside the image get a 2 GB offset = outside the descriptor -> the load returns zeros
  const float* xorg = p.x + ((((long)n * p.D + iz0) * p.H + iy0) * p.W + ix0) * (long)p.ldx;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xorg, 0, 0x7fffffff, 0x00020000);
  // weights of this cout block: [k-group][dx][xi][h][32][4] = 1 KB per (dx, xi) image
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.up + (long)blockIdx.y * p.KG * (3 * 16 * 256)), 0, 0x7fffffff, 0x00020000);
  unsigned goff[WNSLOT];
#pragma unroll
  for (int i = 0; i < WNSLOT; ++i) {
    const int e = tid + i * 256;
    goff[i] = 0x80000000u;
    if (e < WVOX * 2) {
      const int vox = e >> 1, q = e & 1;
      const int vz = vox / (WIY * WIX);
      const int r = vox - vz * (WIY * WIX);
      const int vy = r / WIX;
      const int vx = r - vy * WIX;
      if ((unsigned)(iz0 + vz) < (unsigned)p.D && (unsigned)(iy0 + vy) < (unsigned)p.H && (unsigned)(ix0 + vx) < (unsigned)p.W)
        goff[i] = (unsigned)(((vz * p.H + vy) * p.W + vx) * p.ldx + q * 4) * 4u;
    }
  }
  const int tz2 = wave >> 1, ty2 = wave & 1;
  const int bbase = ((2 * tz2 * WIY + 2 * ty2) * WIX + l32) * WS + h * 4;
  const unsigned wlane = (unsigned)((h * 32 + l32) * 16);

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  f32x4 pre[WNSLOT];
  auto fetch = [&](int st) {
#pragma unroll
    for (int i = 0; i < WNSLOT; ++i) pre[i] = bufload(xr, goff[i], (unsigned)st * 32u);
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < WNSLOT; ++i) {
      const int e = tid + i * 256;
      if (e < WVOX * 2) *reinterpret_cast<f32x4*>(buf + (e >> 1) * WS + (e & 1) * 4) = pre[i];
    }
  };

  fetch(0);
  // weight fragments of group G = st*12 + g live in aw[G % 3]; two groups are always in flight
  f32x4 aw[3][4];
  constexpr int zorder[4] = {1, 2, 0, 3};
  auto wload = [&](f32x4 (&dst)[4], int st, int g) {
    const int dx = g >> 2, xz = zorder[g & 3];
    const unsigned so = (unsigned)(((st * 3 + dx) * 16 + xz * 4) * 1024);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e] = bufload(wr, wlane + e * 1024, so);
  };
  wload(aw[0], 0, 0);
  wload(aw[1], 0, 1);
  commit(lds);
  __syncthreads();
  ts1 = clock64();

  const int nst = p.KG;
  for (int st = 0; st < nst; ++st) {
    const float* cur = lds + (DBG == 2 ? 0 : (st & 1) * WBUF);
    float* nxt = lds + ((st + 1) & 1) * WBUF;
    const bool more = (st + 1) < nst;
    if (more && DBG != 2) fetch(st + 1);
    const float* lb = cur + bbase;
    const int stn = more ? st + 1 : st;  // the last stage re-requests its own first groups instead of running past the image
    if (DBG != 1) {
      // Software pipeline: while the 16 MFMAs of group G run on v[G&1], the vector ALU forms v[(G+1)&1] -- a matrix
      // instruction never waits for an operand written just before it.
      f32x4 r1[4], r2[4], rt[4], c[4], v[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r1[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(1, j, 0));
        r2[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(2, j, 0));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = add4(r1[j], r2[j]);
      wino_yt(c, v[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        {  // MFMA xi_z = 1 ; form xi_z = 2 : d2 - d1 ; request row 0
          const int G = dx * 4 + 0;
          wload(aw[(G + 2) % 3], (G + 2 < 12) ? st : stn, (G + 2) % 12);
#pragma unroll
          for (int j = 0; j < 4; ++j) rt[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(0, j, dx));
#pragma unroll
          for (int j = 0; j < 4; ++j) c[j] = sub4(r2[j], r1[j]);
          wino_yt(c, v[1]);
          wino_mfma16(v[0], aw[G % 3], acc[1]);
          WINO_SCHED_GROUP();
          __builtin_amdgcn_sched_barrier(0);
        }
        {  // MFMA xi_z = 2 ; form xi_z = 0 : d0 - d2 ; then request row 3
          const int G = dx * 4 + 1;
          wload(aw[(G + 2) % 3], (G + 2 < 12) ? st : stn, (G + 2) % 12);
#pragma unroll
          for (int j = 0; j < 4; ++j) c[j] = sub4(rt[j], r2[j]);
          wino_yt(c, v[0]);
#pragma unroll
          for (int j = 0; j < 4; ++j) rt[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(3, j, dx));
          wino_mfma16(v[1], aw[G % 3], acc[2]);
          WINO_SCHED_GROUP();
          __builtin_amdgcn_sched_barrier(0);
        }
        {  // MFMA xi_z = 0 ; form xi_z = 3 : d1 - d3 ; then request rows 1, 2 of the next x tap
          const int G = dx * 4 + 2;
          wload(aw[(G + 2) % 3], (G + 2 < 12) ? st : stn, (G + 2) % 12);
#pragma unroll
          for (int j = 0; j < 4; ++j) c[j] = sub4(r1[j], rt[j]);
          wino_yt(c, v[1]);
          if (dx < 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              r1[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(1, j, dx + 1));
              r2[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(2, j, dx + 1));
            }
          }
          wino_mfma16(v[0], aw[G % 3], acc[0]);
          WINO_SCHED_GROUP();
          __builtin_amdgcn_sched_barrier(0);
        }
        {  // MFMA xi_z = 3 ; form xi_z = 1 of the next x tap : d1 + d2
          const int G = dx * 4 + 3;
          wload(aw[(G + 2) % 3], (G + 2 < 12) ? st : stn, (G + 2) % 12);
          if (dx < 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = add4(r1[j], r2[j]);
            wino_yt(c, v[0]);
          }
          wino_mfma16(v[1], aw[G % 3], acc[3]);
          WINO_SCHED_GROUP();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (more && DBG != 2) commit(nxt);
    __syncthreads();
  }
  ts2 = clock64();

  // ---- output transform (A^T . A over (z,y)) and store; the bias quads are requested first and arrive under the adds ----
  // bias of this lane's 16 couts (4 quads), 4 quads
  f32x4 bq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bq[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bq[g] = *reinterpret_cast<const f32x4*>(p.bias + blockIdx.y * 32 + 8 * g + 4 * h);
  }
  const int oxx = ox0 + l32;
  const int zb = oz0 + 2 * tz2, yb = oy0 + 2 * ty2;
  float* ybase = p.y + ((((long)n * p.D + zb) * p.H + yb) * p.W + oxx) * (long)p.ldy + blockIdx.y * 32 + 4 * h;
  const long ysY = (long)p.W * p.ldy, ysZ = (long)p.H * p.W * p.ldy;
  const bool inx = oxx < p.W;
  // one register quad (4 couts) of all 16 accumulators at a time: 64 + 32 live values
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 tq[4][2];
#pragma unroll
    for (int xz = 0; xz < 4; ++xz) {
      f32x4 q[4];
#pragma unroll
      for (int xy = 0; xy < 4; ++xy) q[xy] = f32x4{acc[xz][xy][4 * g], acc[xz][xy][4 * g + 1], acc[xz][xy][4 * g + 2], acc[xz][xy][4 * g + 3]};
      tq[xz][0] = q[0] + q[1] + q[2];
      tq[xz][1] = q[1] - q[2] - q[3];
    }
    const bool cok = inx && (blockIdx.y * 32 + 8 * g + 4 * h < p.Cout);
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      const f32x4 o0 = tq[0][oy] + tq[1][oy] + tq[2][oy] + bq[g];
      const f32x4 o1 = tq[1][oy] - tq[2][oy] - tq[3][oy] + bq[g];
      if (cok && yb + oy < p.H) {
        if (zb < p.D) *reinterpret_cast<f32x4*>(ybase + oy * ysY + 8 * g) = o0;
        if (zb + 1 < p.D) *reinterpret_cast<f32x4*>(ybase + ysZ + oy * ysY + 8 * g) = o1;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (tid == 0) {
    const long long ts3 = clock64();
    long long* o = p.stamps + (long)blockIdx.x * 4;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
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
  p.Npad = Npad; p.KG = KG; p.ntz = (D + 3) / 4; p.nty = (H + 3) / 4; p.ntx = (W + 31) / 32; p.dbg = dbg;
  p.tpw = argc > 5 ? atoi(argv[5]) : 8; p.nzc = (p.ntz + p.tpw - 1) / p.tpw;
  const int ldsb = 2 * WBUF * 4;
  void (*kern)(const WinoParams) = dbg == 1 ? wino_kernel<1> : dbg == 2 ? wino_kernel<2> : wino_kernel<0>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  dim3 grid(p.ntz * p.nty * p.ntx, Npad / 32);
  hipMalloc(&p.stamps, (long)grid.x * 4 * 8);
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
    std::vector<long long> st((long)grid.x * 4);
    hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b2 = 0, c = 0;
    for (unsigned i = 0; i < grid.x; ++i) { a += st[i * 4 + 1] - st[i * 4]; b2 += st[i * 4 + 2] - st[i * 4 + 1]; c += st[i * 4 + 3] - st[i * 4 + 2]; }
    printf("clock64 ticks per WG: prologue %.0f  loop %.0f  epilogue %.0f\n", a / grid.x, b2 / grid.x, c / grid.x);
  }
  return 0;
}
