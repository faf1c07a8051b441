// Transposed-form 3x3x3 stride-2 convolution on 16-bit storage, all eight output-parity classes in one pass (round 3):
//   ConvUpsample forward                 (Conv3DTranspose k3 s2 'same', upsample.py:28-33):  y[2i + k] += x[i] W[k], cropped to [0, 2n)
//   ConvDownsample data gradient         (of Conv3D k3 s2 'same', downsample.py:28-35; train.py:142-151 under TF autodiff)
// Both are the gather  out[2g + p] = bias + sum over the taps of class p of  W[k]^T in[g + off]  on the coarse grid g, where per axis
// k = 0 -> (p = 0, off = 0), k = 1 -> (p = 1, off = 0), k = 2 -> (p = 0, off = -1): each of the 27 taps belongs to exactly one
// (input offset in {-1,0}^3, output class in {0,1}^3).
//
// Round 2 ran this as eight launches of the general gather kernel, one per class: the coarse tensor was read eight times, every
// launch wrote every other voxel of the fine tensor (half lines), and each 512-position workgroup paid a full memory round trip before
// its first matrix instruction: 116 TF on 64 -> 32 channels @ 64^3 -> 128^3.  Here (the 16-bit sibling of the fp32 engine's upm_kernel,
// on the staging scheme of lowp_s1d.hip):
//   * a wave owns 32 coarse positions (one x row) and ALL eight classes: 8 accumulators; per k-step of 16 input channels it reads the
//     8 input fragments at offsets {-1,0}^3 and the 27 weight fragments from LDS and issues 27 matrix instructions;
//   * the coarse halo tile (low-side halo of one voxel) and the k-step's weights reach LDS by buffer_load ... lds, RD k-steps ahead
//     (ring), one `s_waitcnt vmcnt(N)` + s_barrier per k-step; the image part is the stride-1 DMA kernel's ([cout group][k-step][27
//     taps][k-half][cout][8 cin]: bts_lp_s1d_pack_ writes it for these kinds too);
//   * every fine voxel is written once, 16 bytes per lane (8 consecutive couts after v_permlane32_swap), the two x classes of a lane
//     to neighbouring voxels: whole rows of the fine tensor leave a wave together.
// Declines (the per-class gather runs): coarse W < 12, Cout % 8 != 0, odd offsets beyond 31 bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "bts_internal.h"
#include "lowp_common.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct LpUpParams {
  const unsigned short* x;    // coarse tensor (N, D, H, W, Cin), voxel stride ldx
  const unsigned short* wp;   // DMA part of the image: [cout group][k-step][27 taps][k-half][cout in group][8 cin]
  const float* bias;
  unsigned short* y;          // fine tensor (N, 2D, 2H, 2W, Cout), voxel stride ldy
  int N, D, H, W, ldx, ldy, Cout, KS, NB;
  int ntx, nty, ntz, ncg;
  long nitems;
  int accum;
  // fused GroupNorm partial sums of the fine tensor (slab semantics: whole fine planes per group) [N*G][gn_B][2] fp64, or NULL.
  // One (sum, sumsq) pair per (fine z plane, tile row, tile column, cout group, wave sharing that plane): fixed order
  double* gnp;
  int gn_G, gn_zt;
  long gn_B;
};

template <int MODE, int TXL>
struct UpGeo {
  static constexpr int TX = 1 << TXL, ZP = 32 / TX;          // coarse x extent of a fragment, z planes per fragment
  static constexpr int CBW = MODE ? 2 : 1;                   // cout blocks of 32 per item
  static constexpr int RZ = MODE ? 1 : 2;                    // 8 waves = CBW x 4 y rows x RZ z units
  static constexpr int TY = 4, TZ = RZ * ZP;
  static constexpr int SX = TX + 1, SY = TY + 1, SZ = TZ + 1;
  static constexpr int PS = (ZP == 1) ? SY * SX : ((SY * SX + 15) / 16) * 16;
  static constexpr int NVOX = SZ * PS;
  static constexpr int NCH = ((NVOX + 255) / 256) * 8;       // 1 KB chunks (32 voxels x 32 bytes), whole rounds of the 8 waves
  static constexpr int HBUF = NCH * 1024, NH = NCH / 8;
  static constexpr int WTAP = CBW * 1024, WSTEP = 27 * WTAP;
  static constexpr int NWC = 27 * CBW, NW = (NWC + 7) / 8;
  static constexpr int RD = MODE ? 2 : 3;                    // ring depth (k-steps in LDS)
  static constexpr int OFF_W = RD * HBUF, OFF_SCR = OFF_W + RD * WSTEP, OFF_BIAS = OFF_SCR + 1024;
  static constexpr int LDS_BYTES = OFF_BIAS + 2 * 256;
  static constexpr int NREQ = NH + NW + 1;                   // requests per wave and k-step (halo, weights, bias)
  static constexpr int NST = 16;                             // stores per wave and item
};

template <int N> __device__ __forceinline__ void up_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int MODE, int TXL>
__global__ __launch_bounds__(512, 2) void lp_up_kernel(const LpUpParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (see lp_s1d_kernel: the host pass drops the launch stub of this template otherwise)
  typedef UpGeo<MODE, TXL> G;
  constexpr int TX = G::TX, ZP = G::ZP, CBW = G::CBW, SX = G::SX, SY = G::SY, PS = G::PS, NVOX = G::NVOX;
  constexpr int NH = G::NH, NW = G::NW, RD = G::RD;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  const int lx = l32 & (TX - 1), lz = l32 >> TXL;
  const int wy = wave & 3;                         // y row of this wave inside the tile
  const int wz = MODE ? 0 : (wave >> 2);           // z unit
  const int cbw = MODE ? (wave >> 2) : 0;          // cout block inside the item

  // ---- item walk (cout group fastest, then x, y, z tiles, samples); XCD k walks its own contiguous eighth ----
  unsigned it, it_end, it_step;
  {
    const unsigned Gx = gridDim.x, b = blockIdx.x, ni_ = (unsigned)p.nitems;
    if (Gx >= 8) {
      const unsigned xcd = b & 7, slot = b >> 3;
      const unsigned q = ni_ / 8, r = ni_ % 8;
      const unsigned start = xcd * q + (xcd < r ? xcd : r);
      it_end = start + q + (xcd < r ? 1 : 0);
      it_step = (Gx - xcd + 7) >> 3;
      it = start + slot;
    } else {
      it = b; it_end = ni_; it_step = Gx;
    }
  }
  if (it >= it_end) return;
  struct Item { int cg, n, gx0, gy0, gz0; };
  auto decode = [&](unsigned i) {
    Item t;
    unsigned q = i / (unsigned)p.ncg;
    t.cg = (int)(i - q * p.ncg); i = q; q = i / (unsigned)p.ntx;
    t.gx0 = (int)(i - q * p.ntx) * TX; i = q; q = i / (unsigned)p.nty;
    t.gy0 = (int)(i - q * p.nty) * G::TY; i = q; q = i / (unsigned)p.ntz;
    t.gz0 = (int)(i - q * p.ntz) * G::TZ;
    t.n = (int)q;
    return t;
  };

  // ---- DMA side: chunk r*8 + wave of a k-step's halo tile = 32 voxels, lanes (2i, 2i+1) the two 16-byte slots of voxel i ----
  unsigned hrel[NH], hcrd[NH], hoff[NH];
#pragma unroll
  for (int r = 0; r < NH; ++r) {
    const int vox = (r * 8 + wave) * 32 + (lane >> 1);
    const int vz = vox / PS, rem = vox - vz * PS;
    const int vy = rem / SX, vx = rem - vy * SX;
    const int hp = (lane & 1) ^ ((vx >> 3) & 1);
    const bool geo = vox < NVOX && rem < SY * SX;
    hrel[r] = (unsigned)(((vz * p.H + vy) * p.W + vx) * p.ldx * 2 + hp * 16);
    hcrd[r] = (unsigned)(vx | (vy << 8)) | (geo ? (unsigned)vz << 16 : 0xffff0000u);
  }
  __amdgpu_buffer_rsrc_t xr;
  auto dma_item = [&](const Item& t, bool live) {
    const unsigned short* xorg = p.x + ((((long)t.n * p.D + (t.gz0 - 1)) * p.H + (t.gy0 - 1)) * p.W + (t.gx0 - 1)) * (long)p.ldx;
    xr = __builtin_amdgcn_make_buffer_rsrc((void*)xorg, 0, 0x7fffffff, 0x00020000);
    const int zb = live ? t.gz0 - 1 : 0x100000;
#pragma unroll
    for (int r = 0; r < NH; ++r) {
      const int vx = hcrd[r] & 0xff, vy = (hcrd[r] >> 8) & 0xff, vz = hcrd[r] >> 16;
      const bool ok = (unsigned)(zb + vz) < (unsigned)p.D && (unsigned)(t.gy0 - 1 + vy) < (unsigned)p.H &&
                      (unsigned)(t.gx0 - 1 + vx) < (unsigned)p.W;
      hoff[r] = ok ? hrel[r] : 0x80000000u;
    }
  };
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  unsigned wvo[NW];
#pragma unroll
  for (int r = 0; r < NW; ++r) wvo[r] = (r * 8 + wave) < G::NWC ? (unsigned)((r * 8 + wave) * 1024 + lane * 16) : 0x80000000u;
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
  // request j (0 .. NREQ-1) of the k-step that goes to ring slot `slot`
  auto issue1 = [&](int j, int slot, int ks, int cg, bool live, int bslot) {
    if (j < NH) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(lds + slot * G::HBUF + (j * 8 + wave) * 1024), 16, hoff[j], (unsigned)ks * 32u, 0, 0);
    } else if (j < NH + NW) {
      const int r = j - NH, c = r * 8 + wave;
      const unsigned soff = live ? (unsigned)((cg * p.KS + ks) * G::WSTEP) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(lds + (c < G::NWC ? G::OFF_W + slot * G::WSTEP + c * 1024 : G::OFF_SCR)), 16, wvo[r], soff, 0, 0);
    } else {
      const int co = cg * CBW * 32 + lane;
      const unsigned off = (lane < 32 * CBW && co < p.Cout) ? (unsigned)co * 4u : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (lds_ptr_t)(lds + G::OFF_BIAS + bslot * 256), 4, off, 0, 0, 0);
    }
  };

  // ---- compute side ----
  // B-operand bases of the 8 input offsets: voxel (1 + z + oz, 1 + wy + oy, 1 + lx + ox) of the halo tile, physical slot of k-half h
  unsigned hbB[2];     // per ox (the half swap depends on the halo column only)
#pragma unroll
  for (int ox = 0; ox < 2; ++ox) {
    const int vx = lx + ox;          // ox = 0 <-> offset -1
    hbB[ox] = (unsigned)((((wz * ZP + lz) * PS) + wy * SX + vx) * 32 + ((h ^ ((vx >> 3) & 1)) * 16));
  }
  const unsigned wbA = (unsigned)(G::OFF_W + h * (CBW * 512) + cbw * 512 + l32 * 16);
  f32x16 acc[8];       // class (pz, py, px) -> acc[pz*4 + py*2 + px]

  // the k-step stream: (item, ks) pairs; requests run RD-1 k-steps ahead of the compute
  Item ci = decode(it);
  const int KS = p.KS;
  // state of the request side
  unsigned rit = it;            // item of the next k-step to request
  Item ri = ci;
  int rks = 0, rslot = 0, rpar = 0;
  bool rlive = true;
  dma_item(ri, true);
  auto advance_req = [&]() {    // move the request cursor one k-step on
    if (++rks == KS) {
      rks = 0;
      rit += it_step;
      rlive = rit < it_end;
      if (rlive) ri = decode(rit);
      rpar ^= 1;
      dma_item(ri, rlive);
    }
    rslot = (rslot + 1 == RD) ? 0 : rslot + 1;
  };
  // prologue: RD-1 k-steps in flight
#pragma unroll
  for (int d = 0; d < RD - 1; ++d) {
#pragma unroll
    for (int j = 0; j < G::NREQ; ++j) issue1(j, rslot, rks, ri.cg, rlive, rpar);
    advance_req();
  }
  int cslot = 0, ipar = 0;
  bool after_out = false;
  for (;;) {
    for (int ks = 0; ks < KS; ++ks) {
      if (after_out) up_wait<(RD - 2) * G::NREQ + G::NST>(); else up_wait<(RD - 2) * G::NREQ>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      after_out = false;
      if (ks == 0) {     // the accumulators start at the bias
        const float* bsh = reinterpret_cast<const float*>(lds + G::OFF_BIAS + ipar * 256) + cbw * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bsh + 8 * q);
#pragma unroll
          for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][4 * q + j] = bq[j];
        }
      }
      // the 8 input fragments of this k-step
      const unsigned char* hb = lds + cslot * G::HBUF;
      u32x4 bf[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int oz = o >> 2, oy = (o >> 1) & 1, ox = o & 1;      // 0 <-> offset -1, 1 <-> offset 0
        bf[o] = *reinterpret_cast<const u32x4*>(hb + hbB[ox] + (oz * PS + oy * SX) * 32);
      }
      const unsigned char* wb = lds + wbA + cslot * G::WSTEP;
      // 27 taps: k -> (offset, class); the requests of the k-step RD-1 ahead are dealt out between the matrix instructions
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
        const int o = ((kz == 2 ? 0 : 1) << 2) | ((ky == 2 ? 0 : 1) << 1) | (kx == 2 ? 0 : 1);
        const int cls = ((kz == 1 ? 1 : 0) << 2) | ((ky == 1 ? 1 : 0) << 1) | (kx == 1 ? 1 : 0);
        const u32x4 a = *reinterpret_cast<const u32x4*>(wb + t * G::WTAP);
        acc[cls] = T::mfma(a, bf[o], acc[cls]);
        if (t % 2 == 1 && t / 2 < G::NREQ) issue1(t / 2, rslot, rks, ri.cg, rlive, rpar);
      }
      static_assert(G::NREQ <= 13, "a k-step has 13 request slots");
      advance_req();
      cslot = (cslot + 1 == RD) ? 0 : cslot + 1;
    }
    // ---- output side: class (pz, py, px) of coarse (gz, gy, gx) -> fine voxel (2gz + pz, 2gy + py, 2gx + px) ----
    {
      const int gz = ci.gz0 + wz * ZP + lz, gy = ci.gy0 + wy, gx = ci.gx0 + lx;
      const int cb = ci.cg * CBW + cbw;
      const int Ho = 2 * p.H, Wo = 2 * p.W;
      const __amdgpu_buffer_rsrc_t yr =
          __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long)ci.n * 8L * p.D * p.H * p.W * (long)p.ldy), 0, 0x7fffffff, 0x00020000);
      const bool inb = gz < p.D && gy < p.H && gx < p.W;
      const bool gn_on = p.gnp != nullptr;
      float gn_s[2] = {0.f, 0.f}, gn_q[2] = {0.f, 0.f};      // per fine plane 2 gz + pz of this lane
#pragma unroll
      for (int cls = 0; cls < 8; ++cls) {
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
        const unsigned vbase = (unsigned)((((2 * gz + pz) * Ho + 2 * gy + py) * Wo + 2 * gx + px) * p.ldy);
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int co = cb * 32 + 16 * qp + 8 * h;
          const bool ok = inb && co < p.Cout;
          const unsigned off = ok ? (vbase + co) * 2u : 0x80000000u;
          float f[4], g2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { f[j] = acc[cls][8 * qp + j]; g2[j] = acc[cls][8 * qp + 4 + j]; }
          if (p.accum) {
            u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(yr, off, 0, 0);
            asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                         : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
            float old[8];
            unpack8<T>(e, old);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] += old[j]; g2[j] += old[4 + j]; }
          }
          // (own values, before the exchange: the gate is the voxel's validity, not the stored piece's column range -- see lowp_s1d.hip)
          if (gn_on && inb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              gn_s[pz] += f[j] + g2[j];
              gn_q[pz] = fmaf(f[j], f[j], fmaf(g2[j], g2[j], gn_q[pz]));
            }
          }
          unsigned d0 = pack2<T>(f[0], f[1]), d1 = pack2<T>(f[2], f[3]), d2 = pack2<T>(g2[0], g2[1]), d3 = pack2<T>(g2[2], g2[3]);
          asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{d0, d1, d2, d3}, yr, off, 0, 0);
        }
      }
      if (gn_on) {
        constexpr int WPZ = 8 / G::RZ;                       // waves of an item that share a coarse z unit
        const int wsub = MODE ? wave : (wave & 3);           // this wave's index among them
        const int ty = ci.gy0 / G::TY, tx = ci.gx0 / TX;
#pragma unroll
        for (int zl = 0; zl < ZP; ++zl)
#pragma unroll
          for (int pz = 0; pz < 2; ++pz) {
            const bool mine = (ZP == 1) || lz == zl;
            const double ds = wave_sum_f64(mine ? (double)gn_s[pz] : 0.0), dq = wave_sum_f64(mine ? (double)gn_q[pz] : 0.0);
            const int gzc = ci.gz0 + wz * ZP + zl;          // coarse plane
            const int zf = 2 * gzc + pz;                    // fine plane
            if (lane == 0 && gzc < p.D) {
              const int gg = zf / p.gn_zt;
              const long slot = (((((long)(zf - gg * p.gn_zt) * p.nty + ty) * p.ntx + tx) * p.ncg + ci.cg) * WPZ) + wsub;
              double* dst = p.gnp + (((long)ci.n * p.gn_G + gg) * p.gn_B + slot) * 2;
              dst[0] = ds;
              dst[1] = dq;
            }
          }
      }
    }
    after_out = true;
    it += it_step;
    if (it >= it_end) break;
    ci = decode(it);
    ipar ^= 1;
  }
  up_wait<0>();    // look-ahead requests past the last item (zeros into LDS) must not outlive the workgroup's LDS allocation
#endif
}

// =====================================================================================================================
// plan + launch
// =====================================================================================================================
static bool up_enabled() {   // BTS_LP_UP=0: the transposed form back on eight launches of the gather kernel (A/B; read per call)
  const char* e = getenv("BTS_LP_UP");
  return !(e && atoi(e) == 0);
}
struct UpPlan { int mode, txl, ntx, nty, ntz, ncg; long nitems; };
static bool up_plan(int N, int D, int H, int W, int Cin, int Cout, UpPlan& pl) {      // (D,H,W) = the COARSE grid
  if (!up_enabled() || Cin % 16 != 0 || Cout % 8 != 0 || W < 12) return false;
  if ((long)N * D * H * W < 2048) return false;
  const int NB = (Cout + 31) / 32;
  pl.mode = NB >= 2 ? 1 : 0;
  pl.txl = W >= 24 ? 5 : 4;
  const int TX = 1 << pl.txl, ZP = 32 / TX, TZ = (pl.mode ? 1 : 2) * ZP;
  pl.ntx = (W + TX - 1) / TX; pl.nty = (H + 3) / 4; pl.ntz = (D + TZ - 1) / TZ;
  pl.ncg = pl.mode ? (NB + 1) / 2 : 1;
  pl.nitems = (long)N * pl.ntz * pl.nty * pl.ntx * pl.ncg;
  return pl.nitems <= 0x7fffffffL;
}
bool bts_lp_up_takes_(int N, int D, int H, int W, int Cin, int Cout) {
  UpPlan pl;
  return up_plan(N, D, H, W, Cin, Cout, pl);
}

template <typename T, int MODE, int TXL>
static int up_launch_t(const LpUpParams& p, hipStream_t stream) {
  typedef UpGeo<MODE, TXL> G;
  auto kern = lp_up_kernel<T, MODE, TXL>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const long gx = p.nitems < 256 ? p.nitems : 256;
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), G::LDS_BYTES, stream, p);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// GroupNorm-partial slots per (n, group) of the FINE tensor when the kernel takes the shape and can emit them (whole fine planes per
// group); 0 otherwise
long bts_lp_up_gn_B_(int N, int D, int H, int W, int Cin, int Cout, int Gn) {
  UpPlan pl;
  if (Gn <= 0 || (2 * D) % Gn != 0 || !up_plan(N, D, H, W, Cin, Cout, pl)) return 0;
  return (long)(2 * D / Gn) * pl.nty * pl.ntx * pl.ncg * (pl.mode ? 8 : 4);
}
// BTS_OK = ran, 1 = declined.  x: coarse (N,D,H,W,Cin); y: fine (N,2D,2H,2W,Cout); wp_dma: the DMA part of the image.
// gnp (may be NULL): fused GroupNorm partial sums of y, [N*gn_G][bts_lp_up_gn_B_][2]
int bts_lp_up_launch_(int dtype, const void* x, const void* wp_dma, const float* bias, void* y, int N, int D, int H, int W, int Cin, int ldx,
                      int Cout, int ldy, int accum, hipStream_t stream, double* gnp, int gn_G) {
  UpPlan pl;
  if (!up_plan(N, D, H, W, Cin, Cout, pl)) return 1;
  if (gnp != nullptr && (gn_G <= 0 || (2 * D) % gn_G != 0)) return 1;
  if (ldx % 8 != 0 || ldy % 8 != 0 || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 15) || (((uintptr_t)wp_dma) & 15)) return 1;
  if (((long)(D + 2) * H * W + 64) * (long)ldx * 2 >= 0x7fffffffL) return 1;
  if ((8L * D * H * W + 64) * (long)ldy * 2 >= 0x7fffff00L) return 1;
  LpUpParams p;
  p.x = (const unsigned short*)x; p.wp = (const unsigned short*)wp_dma; p.bias = bias; p.y = (unsigned short*)y;
  p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldy; p.Cout = Cout; p.KS = Cin / 16; p.NB = (Cout + 31) / 32;
  p.ntx = pl.ntx; p.nty = pl.nty; p.ntz = pl.ntz; p.ncg = pl.ncg; p.nitems = pl.nitems; p.accum = accum;
  p.gnp = gnp; p.gn_G = gn_G; p.gn_zt = gn_G > 0 ? 2 * D / gn_G : 1;
  p.gn_B = gn_G > 0 ? (long)p.gn_zt * pl.nty * pl.ntx * pl.ncg * (pl.mode ? 8 : 4) : 0;
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(35, 2.0 * 27.0 * Cin * (double)Cout * (double)N * D * H * W, stream);
  int r;
#define UP_CASE(M_, X_)                                                                                               \
  if (pl.mode == M_ && pl.txl == X_)                                                                                  \
    r = dtype == LP_F16 ? up_launch_t<TF16, M_, X_>(p, stream) : up_launch_t<TBF16, M_, X_>(p, stream);
  UP_CASE(0, 5) else UP_CASE(1, 5) else UP_CASE(0, 4) else UP_CASE(1, 4) else r = BTS_ERR_UNSUPPORTED;
#undef UP_CASE
  if (prof) bts_prof_end(stream);
  return r;
}
