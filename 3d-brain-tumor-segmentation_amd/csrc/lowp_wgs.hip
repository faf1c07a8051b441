// Weight gradient of the STRIDED 3x3x3 convolutions on 16-bit operands (round 3): what TF autodiff derives for the Conv3D kernel of
// ConvDownsample (downsample.py:28-35) and the Conv3DTranspose kernel of ConvUpsample (upsample.py:28-33) under train.py:142-151.
// Both are   dW[t][cp][cq] = sum over coarse voxels o of  P[2o + t][cp] * Q[o][cq],   t in {0,1,2}^3, fine index 2o + t past the end = 0:
//   stride-2 conv        P = its input x (fine grid, Cin),     Q = dy (coarse grid, Cout)   -> dW (kd,kh,kw,Cin,Cout)
//   transposed conv      P = dy (fine grid, Cout),             Q = its input x (coarse, Cin) -> dW (kd,kh,kw,Cout,Cin)
// Round 2 ran these on the fp32 kernels over widened copies (bts_lp_uncast): 15 ms of a 120 ms batch-8 step plus 4 ms of casts.
//
// The contraction runs over VOXELS while memory is channel-fastest, so both matrix operands need a transpose.  gfx950's
// ds_read_b64_tr_b16 does it on the way out of LDS: the tiles are staged as they lie in memory ([voxel][32 channels], by
// buffer_load ... lds, no registers), and a 16-lane group that points at a [4 voxels][16 channels] block receives, per lane, one
// channel's four voxels -- two such reads are the 8 consecutive k of a lane's matrix operand.  The read takes a per-lane address, so
// the stride-2 walk through the fine tile (voxel 2o + t) costs nothing.  With 16-bit matrix instructions these layers are bound by
// reading P once (8 fine voxels per coarse one, 27/16 matrix instructions per coarse voxel and 32x32 channel block): the kernel is
// built to stream -- small double-buffered tiles (coarse 16x2x2), all 27 taps dealt to the 8 waves (3-4 accumulators each), persistent
// workgroups, fp32 partials in the layout of lowp.hip's fixed-order finalize (shared with the stride-1 kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.h"
#include "bts_internal.h"
#include "lowp_common.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct LpWgsParams {
  const unsigned short* p;   // fine tensor  (N, 2*Dc.., Cp), voxel stride ldp; fine dims Df, Hf, Wf (may be 2*Dc or 2*Dc - 1 ... any: bounds-checked)
  const unsigned short* q;   // coarse tensor (N, Dc, Hc, Wc, Cq), voxel stride ldq
  float* part;               // [workgroup][cp block][cq group][27 taps][32][32*NQ]
  int N, Df, Hf, Wf, Dc, Hc, Wc, Cp, ldp, Cq, ldq;
  int ntx, nty, ntz, ncp, ncqg;
  long ntiles;
};
#define WGS_TX 16
#define WGS_TY 2
#define WGS_TZ 2
#define WGS_FX (2 * WGS_TX + 1)
#define WGS_FY (2 * WGS_TY + 1)
#define WGS_FZ (2 * WGS_TZ + 1)
#define WGS_PCH 56                       // 1 KB chunks (16 fine voxels x 64 B) of a P tile: 825 voxels -> 52, padded to the 8 waves
#define WGS_PBYTES (WGS_PCH * 1024)
#define WGS_QBYTES 8192                  // 64 coarse voxels x up to 128 B

template <typename T, int NQ>
__global__ __launch_bounds__(512, 2) void lp_wgs_kernel(const LpWgsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (see lp_s1d_kernel: the host pass drops the launch stub of this template otherwise)
  constexpr int FX = WGS_FX, FY = WGS_FY, FZ = WGS_FZ, NPV = FX * FY * FZ;
  constexpr int BUF = WGS_PBYTES + WGS_QBYTES, OFF_SCR = 2 * BUF;
  constexpr int QROW = 64 * NQ;          // bytes per coarse voxel in the Q tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cpt = blockIdx.y / p.ncqg, cqg = blockIdx.y % p.ncqg;
  const int cp0 = cpt * 32, cq0 = cqg * 32 * NQ;

  // ---- DMA side ----
  // P request r of a wave = chunk r*8 + wave: 16 fine voxels, lanes (4i .. 4i+3) the four 16-byte pieces of voxel i's 32 channels
  constexpr int NPR = WGS_PCH / 8;
  unsigned prel[NPR], pcrd[NPR], poff[NPR];
#pragma unroll
  for (int r = 0; r < NPR; ++r) {
    const int vox = (r * 8 + wave) * 16 + (lane >> 2), piece = lane & 3;
    const int fz = vox / (FY * FX), rem = vox - fz * (FY * FX);
    const int fy = rem / FX, fx = rem - fy * FX;
    const bool geo = vox < NPV && cp0 + piece * 8 < p.Cp;
    prel[r] = (unsigned)((((fz * p.Hf + fy) * p.Wf + fx) * p.ldp + cp0 + piece * 8) * 2);
    pcrd[r] = (unsigned)(fx | (fy << 8)) | (geo ? (unsigned)fz << 16 : 0xffff0000u);
  }
  // Q request (one per wave): NQ = 1: chunk `wave` (< 4) = 16 coarse voxels x 4 pieces; NQ = 2: chunk `wave` = 8 voxels x 8 pieces
  unsigned qrel, qcrd, qoff;
  {
    const int vox = NQ == 1 ? wave * 16 + (lane >> 2) : wave * 8 + (lane >> 3);
    const int piece = NQ == 1 ? (lane & 3) : (lane & 7);
    const int cz = vox / (WGS_TY * WGS_TX), rem = vox - cz * (WGS_TY * WGS_TX);
    const int cy = rem / WGS_TX, cx = rem - cy * WGS_TX;
    const bool geo = vox < WGS_TX * WGS_TY * WGS_TZ && cq0 + piece * 8 < p.Cq;
    qrel = (unsigned)((((cz * p.Hc + cy) * p.Wc + cx) * p.ldq + cq0 + piece * 8) * 2);
    qcrd = (unsigned)(cx | (cy << 8)) | (geo ? (unsigned)cz << 16 : 0xffff0000u);
  }
  __amdgpu_buffer_rsrc_t pr, qr;
  auto setup = [&](long tile) {
    long b = tile;
    const int tx = (int)(b % p.ntx); b /= p.ntx;
    const int ty = (int)(b % p.nty); b /= p.nty;
    const int tz = (int)(b % p.ntz);
    const int n = (int)(b / p.ntz);
    const int ox0 = tx * WGS_TX, oy0 = ty * WGS_TY, oz0 = tz * WGS_TZ;
    pr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.p + ((((long)n * p.Df + 2 * oz0) * p.Hf + 2 * oy0) * p.Wf + 2 * ox0) * (long)p.ldp), 0,
                                           0x7fffffff, 0x00020000);
    qr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q + ((((long)n * p.Dc + oz0) * p.Hc + oy0) * p.Wc + ox0) * (long)p.ldq), 0, 0x7fffffff,
                                           0x00020000);
#pragma unroll
    for (int r = 0; r < NPR; ++r) {
      const int fx = pcrd[r] & 0xff, fy = (pcrd[r] >> 8) & 0xff, fz = pcrd[r] >> 16;
      const bool ok = (unsigned)(2 * oz0 + fz) < (unsigned)p.Df && (unsigned)(2 * oy0 + fy) < (unsigned)p.Hf && (unsigned)(2 * ox0 + fx) < (unsigned)p.Wf;
      poff[r] = ok ? prel[r] : 0x80000000u;
    }
    {
      const int cx = qcrd & 0xff, cy = (qcrd >> 8) & 0xff, cz = qcrd >> 16;
      const bool ok = (unsigned)(oz0 + cz) < (unsigned)p.Dc && (unsigned)(oy0 + cy) < (unsigned)p.Hc && (unsigned)(ox0 + cx) < (unsigned)p.Wc;
      qoff = ok ? qrel : 0x80000000u;
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int r = 0; r < NPR; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (lds_ptr_t)(lds + buf * BUF + (r * 8 + wave) * 1024), 16, poff[r], 0, 0, 0);
    const bool qlive = NQ == 2 || wave < 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(qr, (lds_ptr_t)(lds + (qlive ? buf * BUF + WGS_PBYTES + wave * 1024 : OFF_SCR)), 16,
                                             qlive ? qoff : 0x80000000u, 0, 0, 0);
  };

  // ---- compute side ----
  // transposing reads: lane l of 16-lane group g = l >> 4 points at 4 channels (16 (g & 1) + 4 (l & 3) ...) of voxel 8 (g >> 1) + ((l & 15) >> 2)
  // (+ 4 for the second read) and receives channel 16 (g & 1) + (l & 15)'s four voxels: the matrix operand of row / column l & 31, k half l >> 5
  const int g4 = lane >> 4, rr = (lane & 15) >> 2, c4 = lane & 3;
  const int c0 = 16 * (g4 & 1) + 4 * c4, kx = 8 * (g4 >> 1) + rr;
  const unsigned lbA = (unsigned)((2 * kx) * 64 + c0 * 2);
  const unsigned lbB = (unsigned)(WGS_PBYTES + kx * QROW + c0 * 2);
  // this wave's taps: wave, wave + 8, wave + 16, wave + 24 (< 27)
  unsigned tbase[4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const int t = wave + 8 * ti;
    const int tz = t / 9, ty = (t / 3) % 3, tx = t % 3;
    tbase[ti] = lbA + (unsigned)(((tz * FY + ty) * FX + tx) * 64);
  }
  f32x16 acc[4][NQ];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int c = 0; c < NQ; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][c][r] = 0.f;
  auto trd = [&](const unsigned char* base) -> u32x2 {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
  };

  long tile = blockIdx.x;
  if (tile >= p.ntiles) {      // (never: the launcher sizes the grid by the tile count; partials of an idle workgroup would be garbage)
    return;
  }
  setup(tile);
  issue(0);
  int buf = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const long nxt = tile + gridDim.x;
    if (nxt < p.ntiles) {
      setup(nxt);
      issue(buf ^ 1);
    }
    const unsigned char* lb = lds + buf * BUF;
#pragma unroll
    for (int zc = 0; zc < WGS_TZ; ++zc)
#pragma unroll
      for (int yc = 0; yc < WGS_TY; ++yc) {
        u32x4 bq[NQ];
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
          const u32x2 lo = trd(lb + lbB + ((zc * WGS_TY + yc) * WGS_TX) * QROW + c * 64);
          const u32x2 hi = trd(lb + lbB + ((zc * WGS_TY + yc) * WGS_TX + 4) * QROW + c * 64);
          bq[c] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          if (wave + 8 * ti < 27) {
            const unsigned char* pa = lb + tbase[ti] + ((2 * zc * FY + 2 * yc) * FX) * 64;
            const u32x2 lo = trd(pa), hi = trd(pa + 8 * 64);
            const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int c = 0; c < NQ; ++c) acc[ti][c] = T::mfma(a, bq[c], acc[ti][c]);
          }
        }
      }
    buf ^= 1;
  }
  // ---- partial sums: [workgroup][cp block][cq group][tap][32 rows = cp][32*NQ columns = cq] ----
  float* pb = p.part + (((long)blockIdx.x * p.ncp + cpt) * p.ncqg + cqg) * (27L * 32 * 32 * NQ);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const int t = wave + 8 * ti;
    if (t < 27) {
#pragma unroll
      for (int c = 0; c < NQ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          pb[((long)t * 32 + row) * (32 * NQ) + c * 32 + (lane & 31)] = acc[ti][c][r];
        }
    }
  }
#endif
}

// =====================================================================================================================
// plan + launch (the finalize is lowp.hip's)
// =====================================================================================================================
int bts_lp_wgrad_finalize_(const float* part, float* dw, int nwg, int ncp, int ncqg, int nslot, int ntaps, int NQ, int Cp, int Cq, int Cin_ref,
                           int dup_start, int dup_shift, int accum, hipStream_t stream);

static void wgs_plan(int N, int Dc, int Hc, int Wc, int Cp, int Cq, int& nq, int& nwg, long& ntiles, int& ncp, int& ncqg, int& ntx, int& nty,
                     int& ntz) {
  nq = Cq >= 64 ? 2 : 1;
  ncp = (Cp + 31) / 32;
  ncqg = (Cq + 32 * nq - 1) / (32 * nq);
  ntx = (Wc + WGS_TX - 1) / WGS_TX; nty = (Hc + WGS_TY - 1) / WGS_TY; ntz = (Dc + WGS_TZ - 1) / WGS_TZ;
  ntiles = (long)N * ntz * nty * ntx;
  long cap = 256 / ((long)ncp * ncqg);          // one 512-thread workgroup per CU over the whole launch
  if (cap < 1) cap = 1;
  nwg = (int)(ntiles < cap ? ntiles : cap);
}
long bts_lp_wgs_workspace_(int N, int Dc, int Hc, int Wc, int Cp, int Cq) {
  int nq, nwg, ncp, ncqg, ntx, nty, ntz;
  long ntiles;
  wgs_plan(N, Dc, Hc, Wc, Cp, Cq, nq, nwg, ntiles, ncp, ncqg, ntx, nty, ntz);
  return (long)nwg * ncp * ncqg * 27 * 32 * 32 * nq * 4;
}
// dw[t][cp][cq] (+)= sum_o P[2o + t][cp] Q[o][cq].  P: (N, Df, Hf, Wf, Cp) stride ldp; Q: (N, Dc, Hc, Wc, Cq) stride ldq.
int bts_lp_wgs_launch_(int dtype, const void* P, const void* Q, float* dw, void* ws, long ws_bytes, int N, int Df, int Hf, int Wf, int Dc, int Hc,
                       int Wc, int Cp, int ldp, int Cq, int ldq, int accum, hipStream_t stream) {
  if (Cp % 8 != 0 || Cq % 8 != 0 || ldp % 8 != 0 || ldq % 8 != 0) return BTS_ERR_SHAPE;
  if ((((uintptr_t)P) & 15) || (((uintptr_t)Q) & 15) || (((uintptr_t)ws) & 15)) return BTS_ERR_ALIGN;
  if ((long)(WGS_FZ + 1) * Hf * Wf * (long)ldp * 2 >= 0x7fffffffL || (long)(WGS_TZ + 1) * Hc * Wc * (long)ldq * 2 >= 0x7fffffffL) return BTS_ERR_SHAPE;
  LpWgsParams p;
  int nq, nwg;
  wgs_plan(N, Dc, Hc, Wc, Cp, Cq, nq, nwg, p.ntiles, p.ncp, p.ncqg, p.ntx, p.nty, p.ntz);
  if (ws_bytes < (long)nwg * p.ncp * p.ncqg * 27 * 32 * 32 * nq * 4) return BTS_ERR_WORKSPACE;
  p.p = (const unsigned short*)P; p.q = (const unsigned short*)Q; p.part = reinterpret_cast<float*>(ws);
  p.N = N; p.Df = Df; p.Hf = Hf; p.Wf = Wf; p.Dc = Dc; p.Hc = Hc; p.Wc = Wc; p.Cp = Cp; p.ldp = ldp; p.Cq = Cq; p.ldq = ldq;
  const size_t shmem = 2 * (WGS_PBYTES + WGS_QBYTES) + 1024;
  (void)hipGetLastError();
#define WGS_LAUNCH(TT, NQ_)                                                                                                  \
  do {                                                                                                                       \
    auto kern = lp_wgs_kernel<TT, NQ_>;                                                                                      \
    static bool done = false;                                                                                                \
    if (!done) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      if (e != hipSuccess) return (int)e;                                                                                    \
      done = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL(kern, dim3(nwg, p.ncp * p.ncqg), dim3(512), shmem, stream, p);                                        \
  } while (0)
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(36, 2.0 * 27.0 * (double)Cp * Cq * (double)N * Dc * Hc * Wc, stream);
  if (dtype == LP_F16) { if (nq == 2) WGS_LAUNCH(TF16, 2); else WGS_LAUNCH(TF16, 1); }
  else { if (nq == 2) WGS_LAUNCH(TBF16, 2); else WGS_LAUNCH(TBF16, 1); }
#undef WGS_LAUNCH
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return bts_lp_wgrad_finalize_(p.part, dw, nwg, p.ncp, p.ncqg, 27, 27, nq, Cp, Cq, Cp, 0, 0, accum, stream);
}
