// Feasibility bench: weight gradient of the 3x3x3 stride-1 convolution in Winograd form (F(2x2,3x3) over (z,y), x direct):
//   dU[dx][xi] = sum_v V[xi](c; x') * T[xi](k; x' - dx + 1),  V = B^T d B (input transform of P),  T = A dY A^T (of Q = dy),
//   dg[kz][ky][dx] = G^T dU[dx] G.   48 accumulator tiles fed 1/4 as often as the 27 of the direct form: 12/27 of the MFMAs.
// Contraction over voxels: channels on the lanes, 4 consecutive x per lane (LDS tiles are x-fastest, transposed while
// staging) so the transforms are packed.  4 waves = the 4 xi_z, each 3 x taps x 4 xi_y = 12 accumulators (192 registers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wgw wgw.hip && ./wgw [D] [nwg]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgwParams {
  const float* p;
  const float* q;
  float* part;
  int N, D, H, W, ldp, ldq;
  int npz, npy, ntx, ntiles, per;
  long long* stamps;
};

#define QSTR 28                        /* floats per (row, channel) of the Q tile: x = -1..16 at entries 3..20 */
#define PTILE (24 * 32 * 16)           /* 4 z-rows x 6 y-rows, 16 x per (row, channel), quads XOR-swizzled by (c >> 2) & 3 */
#define QTILE (8 * 32 * QSTR)          /* 2 z-rows x 4 y-rows */
#define WBUF (PTILE + QTILE)           /* 19456 floats = 76 KB */
#define NPS 12
#define NQS 5

__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) { const f32x2 lo = pk_add(a.xy, b.xy), hi = pk_add(a.zw, b.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) { const f32x2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x2 s, f32x4 c) { const f32x2 lo = pk_fma(a.xy, s, c.xy), hi = pk_fma(a.zw, s, c.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ float acc_rd(float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }

// tile = 16 x * 4 y * 2 z of Q (two 2x2 (z,y) patches side by side in y); part: [wg][27 taps][32 c][32 k]
__global__ __launch_bounds__(256, 1) void wgw_kernel(const WgwParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = xi_z
  const int ch = lane & 31, hh = lane >> 5;
  const long long ts0 = clock64();

  // ---- staging maps: slot id = tid + 256*i ; P: x = id&15, channel quad = (id>>4)&7, row = id>>7 = (tid>>7) + 2i ----
  const int sxx = tid & 15, scq = (tid >> 4) & 7, srow0 = tid >> 7;
  const int p_lds0 = (srow0 * 32 + scq * 4) * 16 + ((((sxx >> 2) ^ (scq & 3)) << 2) | (sxx & 3));
  const int p_g0 = sxx * p.ldp + scq * 4;
  int q_lds[NQS], q_goff[NQS], q_row[NQS], q_x[NQS];
#pragma unroll
  for (int i = 0; i < NQS; ++i) {
    const int id = tid + 256 * i;
    q_row[i] = -1; q_lds[i] = 0; q_goff[i] = 0; q_x[i] = 0;
    if (id < 1152) {
      const int xx = id % 18, kq = (id / 18) & 7, row = id / 144;
      q_row[i] = row;
      q_x[i] = xx - 1;
      q_lds[i] = PTILE + (row * 32 + kq * 4) * QSTR + xx + 3;
      q_goff[i] = (xx - 1) * p.ldq + kq * 4;
    }
  }
  f32x4 pre[NPS + NQS];
  auto fetch = [&](int tile) {
    int t = tile;
    const int tx = t % p.ntx; t /= p.ntx;
    const int py = t % p.npy; t /= p.npy;
    const int pz = t % p.npz;
    const int n = t / p.npz;
    const int z0 = 2 * pz, y0 = 4 * py, x0 = 16 * tx;
    const float* pb = p.p + ((((long)n * p.D + z0 - 1) * p.H + y0 - 1) * p.W + x0) * p.ldp + p_g0;
#pragma unroll
    for (int i = 0; i < NPS; ++i) {
      const int row = srow0 + 2 * i;
      const int zr = row / 6, yr = row - zr * 6;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)(z0 - 1 + zr) < (unsigned)p.D && (unsigned)(y0 - 1 + yr) < (unsigned)p.H)
        v = *reinterpret_cast<const f32x4*>(pb + ((long)zr * p.H + yr) * p.W * p.ldp);
      pre[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NQS; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q_row[i] >= 0) {
        const int z = z0 + (q_row[i] >> 2), y = y0 + (q_row[i] & 3), x = x0 + q_x[i];
        if ((unsigned)x < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(p.q + ((((long)n * p.D + z) * p.H + y) * p.W + x0) * p.ldq + q_goff[i]);
      }
      pre[NPS + i] = v;
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < NPS; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) buf[p_lds0 + i * (2 * 32 * 16) + e * 16] = pre[i][e];
    }
#pragma unroll
    for (int i = 0; i < NQS; ++i) {
      if (q_row[i] >= 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) buf[q_lds[i] + e * QSTR] = pre[NPS + i][e];
      }
    }
  };

  // ---- wave roles: xi_z: 0: d0 - d2 | 1: d1 + d2 | 2: d2 - d1 | 3: d1 - d3 ;  c = P[ra] + s * P[rb] ----
  const int ra = (wave == 0) ? 0 : (wave == 2) ? 2 : 1;
  const int rb = (wave == 0) ? 2 : (wave == 1) ? 2 : (wave == 2) ? 1 : 3;
  const float sp = (wave == 1) ? 1.f : -1.f;
  const f32x2 sp2 = {sp, sp};
  // T z-part: 0: q0 | 1: q0 + q1 | 2: q0 - q1 | 3: +q1 (the true -q1 is undone in the tap combine)
  const float sq = (wave == 2) ? -1.f : 1.f;
  const f32x2 sq2 = {sq, sq};
  const int pswz = (ch >> 2) & 3;
  const int pA = (ra * 6 * 32 + ch) * 16, pB = (rb * 6 * 32 + ch) * 16;

  f32x16 acc[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  const int t0 = blockIdx.x * p.per;
  int t1 = t0 + p.per;
  if (t1 > p.ntiles) t1 = p.ntiles;
  if (t0 < t1) {
    fetch(t0);
    commit(lds);
  }
  __syncthreads();
  const long long ts1 = clock64();
  for (int t = t0; t < t1; ++t) {
    const float* cur = lds + ((t - t0) & 1) * WBUF;
    float* nxt = lds + ((t - t0 + 1) & 1) * WBUF;
    const bool more = (t + 1) < t1;
    if (more) fetch(t + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int py2 = 0; py2 < 2; ++py2)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int xq = 2 * s + hh;               // logical x quad of this lane
        const int xo = 4 * xq;
        const int pq = (xq ^ pswz) << 2;          // physical quad offset inside the 16-float row
        // ---- V: z-combine of two rows, then the y transform ----
        f32x4 c[4], v[4];
#pragma unroll
        for (int yr = 0; yr < 4; ++yr) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(cur + pA + (2 * py2 + yr) * 512 + pq);
          const f32x4 b = *reinterpret_cast<const f32x4*>(cur + pB + (2 * py2 + yr) * 512 + pq);
          c[yr] = fma4(b, sp2, a);
        }
        v[0] = sub4(c[0], c[2]);
        v[1] = add4(c[1], c[2]);
        v[2] = sub4(c[2], c[1]);
        v[3] = sub4(c[1], c[3]);
        // ---- T: window of 12 x values (entries xo .. xo+11 of the Q rows), z part then y part ----
        f32x4 a0[3], a1[3], u1[3], u2[3];
        const float* qb = cur + PTILE + ((2 * py2) * 32 + ch) * QSTR + xo;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const f32x4 q00 = *reinterpret_cast<const f32x4*>(qb + (0 * 32) * QSTR + 4 * w);   // (oz 0, oy 0)
          const f32x4 q01 = *reinterpret_cast<const f32x4*>(qb + (1 * 32) * QSTR + 4 * w);   // (oz 0, oy 1)
          const f32x4 q10 = *reinterpret_cast<const f32x4*>(qb + (4 * 32) * QSTR + 4 * w);   // (oz 1, oy 0)
          const f32x4 q11 = *reinterpret_cast<const f32x4*>(qb + (5 * 32) * QSTR + 4 * w);   // (oz 1, oy 1)
          if (wave == 0) { a0[w] = q00; a1[w] = q01; }
          else if (wave == 3) { a0[w] = q10; a1[w] = q11; }
          else { a0[w] = fma4(q10, sq2, q00); a1[w] = fma4(q11, sq2, q01); }
          u1[w] = add4(a0[w], a1[w]);
          u2[w] = sub4(a0[w], a1[w]);
        }
        __builtin_amdgcn_sched_barrier(0);   // the hand-written VALU must not sink between the matrix instructions:
        asm volatile("s_nop 3" ::: "memory");  // the compiler does not track their write -> MFMA-read hazard
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int w = 5 + j - dx;
            acc[dx][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0][j], a0[w >> 2][w & 3], acc[dx][0], 0, 0, 0);
            acc[dx][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1][j], u1[w >> 2][w & 3], acc[dx][1], 0, 0, 0);
            acc[dx][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2][j], u2[w >> 2][w & 3], acc[dx][2], 0, 0, 0);
            acc[dx][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3][j], a1[w >> 2][w & 3], acc[dx][3], 0, 0, 0);   // true t3 = -a1
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    if (more) commit(nxt);
    __syncthreads();
  }
  const long long ts2 = clock64();

  // ---- G^T dU G: y part in the wave (u3 carries the opposite sign) -> LDS [xi_z][dx*3+ky][c][k]; z part across the waves ----
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float* xl = lds + wave * 9 * 1024;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float u0 = acc_rd(acc[dx][0][r]), u1 = acc_rd(acc[dx][1][r]), u2 = acc_rd(acc[dx][2][r]), u3 = acc_rd(acc[dx][3][r]);
      const float hs = 0.5f * (u1 + u2), hd = 0.5f * (u1 - u2);
      const int crow = 8 * (r >> 2) + 4 * hh + (r & 3);
      float* o = xl + (dx * 3) * 1024 + crow * 32 + ch;
      o[0] = u0 + hs;
      o[1024] = hd;
      o[2048] = hs - u3;
    }
  __syncthreads();
  // taps (kz, ky, dx): kz 0: L0 + .5 L1 + .5 L2 | kz 1: .5 L1 - .5 L2 | kz 2: .5 L1 + .5 L2 - L3   (L3 accumulated with +q1)
  float* out = p.part + (long)blockIdx.x * 27 * 1024;
  for (int e = tid; e < 9 * 1024; e += 256) {
    const int g = e >> 10, ck = e & 1023;       // g = dx*3 + ky
    const int dx = g / 3, ky = g - dx * 3;
    const float l0 = lds[(0 * 9 + g) * 1024 + ck], l1 = lds[(1 * 9 + g) * 1024 + ck], l2 = lds[(2 * 9 + g) * 1024 + ck],
                l3 = lds[(3 * 9 + g) * 1024 + ck];
    const float hs = 0.5f * (l1 + l2), hd = 0.5f * (l1 - l2);
    out[((0 * 3 + ky) * 3 + dx) * 1024 + ck] = l0 + hs;
    out[((1 * 3 + ky) * 3 + dx) * 1024 + ck] = hd;
    out[((2 * 3 + ky) * 3 + dx) * 1024 + ck] = hs - l3;
  }
  if (tid == 0) {
    long long* o = p.stamps + (long)blockIdx.x * 4;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = clock64();
  }
}

// dW[tap][c][k] = sum_wg part[wg][tap][c][k]
__global__ void wgw_finalize(const float* part, float* dw, int nwg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 27 * 1024) return;
  double s = 0.0;
  for (int w = 0; w < nwg; ++w) s += part[(long)w * 27 * 1024 + i];
  dw[i] = (float)s;
}

__global__ void ref_wgrad(const float* p, const float* q, double* dw, int D, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 27 * 1024) return;
  const int k = i & 31, c = (i >> 5) & 31, t = i >> 10;
  const int kx = t % 3, ky = (t / 3) % 3, kz = t / 9;
  double s = 0.0;
  for (int z = 0; z < D; ++z)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const int iz = z + kz - 1, iy = y + ky - 1, ix = x + kx - 1;
        if ((unsigned)iz >= (unsigned)D || (unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        s += (double)p[(((long)iz * H + iy) * W + ix) * 32 + c] * q[(((long)z * H + y) * W + x) * 32 + k];
      }
  dw[i] = s;
}

int main(int argc, char** argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 128;
  int nwg = argc > 2 ? atoi(argv[2]) : 512;
  const int H = D, W = D;
  const long nvox = (long)D * H * W;
  std::vector<float> hp(nvox * 32), hq(nvox * 32);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hp) v = rnd();
  for (auto& v : hq) v = rnd();
  if (getenv("DELTA")) {
    const int xs = atoi(getenv("DELTA"));
    for (auto& v : hp) v = 0.f;
    for (auto& v : hq) v = 0.f;
    hp[(((long)5 * H + 5) * W + xs) * 32 + 3] = 1.f;
    if (xs + 1 < W) hq[(((long)5 * H + 5) * W + xs + 1) * 32 + 7] = 1.f;   // expected: tap 12 (kx = 0), c 3, k 7 = 1
    hq[(((long)5 * H + 5) * W + xs) * 32 + 9] = 1.f;                        // and tap 13 (kx = 1), c 3, k 9 = 1
  }
  float *dp, *dq, *dpart, *ddw;
  double* dref;
  hipMalloc(&dp, hp.size() * 4); hipMalloc(&dq, hq.size() * 4);
  hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
  WgwParams p{};
  p.p = dp; p.q = dq; p.N = 1; p.D = D; p.H = H; p.W = W; p.ldp = 32; p.ldq = 32;
  p.npz = D / 2; p.npy = H / 4; p.ntx = W / 16; p.ntiles = p.npz * p.npy * p.ntx;
  if (nwg > p.ntiles) nwg = p.ntiles;
  p.per = (p.ntiles + nwg - 1) / nwg;
  nwg = (p.ntiles + p.per - 1) / p.per;
  hipMalloc(&dpart, (long)nwg * 27 * 1024 * 4); hipMalloc(&ddw, 27 * 1024 * 4); hipMalloc(&dref, 27 * 1024 * 8);
  hipMalloc(&p.stamps, (long)nwg * 4 * 8);
  p.part = dpart;
  const int ldsb = 2 * WBUF * 4;
  hipFuncSetAttribute((const void*)wgw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL(wgw_kernel, dim3(nwg), dim3(256), ldsb, 0, p);
  hipLaunchKernelGGL(wgw_finalize, dim3(108), dim3(256), 0, 0, dpart, ddw, nwg);
  hipError_t e = hipDeviceSynchronize();
  printf("launch: %s  tiles %d  wgs %d x %d tiles  lds %d\n", hipGetErrorString(e), p.ntiles, nwg, p.per, ldsb);
  if (D <= 48) {
    hipLaunchKernelGGL(ref_wgrad, dim3(108), dim3(256), 0, 0, dp, dq, dref, D, H, W);
    std::vector<float> hw(27 * 1024);
    std::vector<double> hr(27 * 1024);
    hipMemcpy(hw.data(), ddw, hw.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), dref, hr.size() * 8, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    int worst = 0;
    for (int i = 0; i < 27 * 1024; ++i) {
      const double d = fabs(hw[i] - hr[i]);
      if (d > maxerr) { maxerr = d; worst = i; }
      if (fabs(hr[i]) > maxref) maxref = fabs(hr[i]);
    }
    for (int t = 0; t < 27; ++t) { double m = 0, r = 0; for (int i = 0; i < 1024; ++i) { m = fmax(m, fabs(hw[t * 1024 + i] - hr[t * 1024 + i])); r = fmax(r, fabs(hr[t * 1024 + i])); } printf("tap %2d (kz %d ky %d kx %d): max err %.3e  max ref %.1f\n", t, t / 9, (t / 3) % 3, t % 3, m, r); }
    if (getenv("DELTA")) for (int i = 0; i < 27 * 1024; ++i) if (fabs(hw[i]) > 1e-3 || fabs(hr[i]) > 1e-3) printf("  tap %d c %d k %d: got %.3f ref %.3f\n", i >> 10, (i >> 5) & 31, i & 31, hw[i], hr[i]);
    printf("max abs err %.3e at tap %d (c %d, k %d): got %.5f ref %.5f   max |ref| %.3f\n", maxerr, worst >> 10, (worst >> 5) & 31, worst & 31,
           hw[worst], hr[worst], maxref);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(wgw_kernel, dim3(nwg), dim3(256), ldsb, 0, p); hipLaunchKernelGGL(wgw_finalize, dim3(108), dim3(256), 0, 0, dpart, ddw, nwg); }
  hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; ++i) { hipLaunchKernelGGL(wgw_kernel, dim3(nwg), dim3(256), ldsb, 0, p); hipLaunchKernelGGL(wgw_finalize, dim3(108), dim3(256), 0, 0, dpart, ddw, nwg); }
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  const double fl = 2.0 * 27 * 32 * 32 * nvox;
  printf("D=%d: %.3f ms (kernel + finalize)  direct-equivalent %.1f TF\n", D, ms, fl / ms * 1e-9);
  std::vector<long long> st((long)nwg * 4);
  hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, b = 0, c = 0;
  for (int i = 0; i < nwg; ++i) { a += st[i * 4 + 1] - st[i * 4]; b += st[i * 4 + 2] - st[i * 4 + 1]; c += st[i * 4 + 3] - st[i * 4 + 2]; }
  printf("clock64 ticks per WG: prologue %.0f  loop %.0f (%.0f per tile, %d MFMA cycles)  epilogue %.0f\n", a / nwg, b / nwg, b / nwg / p.per, 192 * 64, c / nwg);
  return 0;
}
