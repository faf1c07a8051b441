// 1x1x1 convolution on 16-bit storage as a streaming kernel (round 3): the shortcut conv of a ResnetBlock (resnet.py:96-103,118-121),
// the decoder / VAE projections, and -- on the role-swapped image -- their data gradients (train.py:142-151 under TF autodiff).
//
// These layers read and write every voxel once and multiply it by a tiny matrix: at 128^3 they are bound by HBM, and what the general
// gather kernel of lowp.hip lacked for that was memory-level parallelism (one 512-position tile per workgroup, its first loads a
// full round trip away from its first matrix instruction) and wide stores.  Here
//   * the output grid is a flat list of positions (no spatial arithmetic at all: y[pos] = W^T x[pos] + b);
//   * a wave owns 64 positions x all of the item's couts and keeps a ring of RD k-steps of operands in flight, three waves per SIMD
//     (launch bound 168 registers): >= 70 KB requested per CU at any time;
//   * the B operand of the matrix instruction IS the 16-byte global load of a voxel's 8 channels (NDHWC is the operand layout), the
//     weights (one 1 KB fragment per k-step and cout block) come from L2;
//   * results leave as 16-byte stores of 8 consecutive couts (v_permlane32_swap between the two lanes of a voxel, as lowp_s1d.hip),
//     accumulation reads the old values the same way; the bias initialises the accumulators;
//   * the fused global-average-pool partial sums of bts_lp_conv1_gap (column sums per 256-position block) leave from the same place.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.h"
#include "bts_internal.h"
#include "lowp_common.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

struct LpK1Params {
  const unsigned short* x;
  const unsigned short* wp;   // first part of a K1 image: [k-step][cout block][k-half][32 couts][8 cin]
  const float* bias;
  unsigned short* y;
  long npos;
  int ldx, ldy, Cout, KS, NB, accum;
  int ncg;                    // cout groups (of CB blocks of 32)
  int nit;                    // position blocks of 256 one workgroup walks (its column sums leave as ONE partial row)
  double* gap_part;           // [workgroup][Cout] column sums of the unrounded outputs, or NULL
};
#define LPK1_POS 256   // positions per workgroup: 4 waves x 2 fragments x 32

// Output side of a wave's VB fragments: (+ old values), column sums for the fused pool, rounding, stores.  32-cout items: 16-byte
// stores of 8 consecutive couts (v_permlane32_swap between the two lanes of a voxel).  64-cout items: the four 16-byte pieces of a
// lane are quad-transposed first, so that a store instruction writes whole 128-byte rows (and reads them whole when accumulating).
template <typename T, int CB, int VB>
__device__ __forceinline__ void k1_output(const LpK1Params& p, f32x16 (&acc)[VB][CB], const long (&pos)[VB], const bool (&live)[VB], int cg, int h,
                                          int l32, bool gap_on, float (&csum)[CB][16]) {
  if constexpr (CB == 2) {
    const int b = l32 & 3;
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      // row j of this store set: voxel (quad base + j), this lane's piece 2 b + h of the 64-cout group
      const long qpos = pos[v] - b;
      const int co_t = cg * 64 + (2 * b + h) * 8;
      bool okt[4];
      unsigned short* dstt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        okt[j] = (qpos + j) < p.npos && co_t < p.Cout;
        dstt[j] = p.y + (qpos + j) * (long)p.ldy + co_t;
      }
      u32x4 e[4];
      if (p.accum) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          e[j] = u32x4{0u, 0u, 0u, 0u};
          if (okt[j]) e[j] = *reinterpret_cast<const u32x4*>(dstt[j]);
        }
        k1_quad_transpose(e, b);      // -> own voxel, piece 2 j + h (the exchanged layout the 32-cout path loads directly)
      }
      u32x4 dq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {      // j = 2 c + qp
        const int c = j >> 1, qp = j & 1;
        float f[4], g2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = acc[v][c][8 * qp + i]; g2[i] = acc[v][c][8 * qp + 4 + i]; }
        if (p.accum) {
          u32x4 o = e[j];
          asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                       : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
          float old[8];
          unpack8<T>(o, old);
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[i] += old[i]; g2[i] += old[4 + i]; }
        }
        if (gap_on && live[v]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { csum[c][8 * qp + i] += f[i]; csum[c][8 * qp + 4 + i] += g2[i]; }
        }
        unsigned d0 = pack2<T>(f[0], f[1]), d1 = pack2<T>(f[2], f[3]), d2 = pack2<T>(g2[0], g2[1]), d3 = pack2<T>(g2[2], g2[3]);
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        dq[j] = u32x4{d0, d1, d2, d3};
      }
      k1_quad_transpose(dq, b);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (okt[j]) *reinterpret_cast<u32x4*>(dstt[j]) = dq[j];
    }
  } else {
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int cb = cg * CB + c;
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int co = cb * 32 + 16 * qp + 8 * h;
#pragma unroll
        for (int v = 0; v < VB; ++v) {
          const bool ok = live[v] && cb < p.NB && co < p.Cout;
          unsigned short* dst = p.y + pos[v] * (long)p.ldy + co;
          float f[4], g2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { f[j] = acc[v][c][8 * qp + j]; g2[j] = acc[v][c][8 * qp + 4 + j]; }
          if (p.accum) {     // old values arrive in the exchanged layout: the exchange is its own inverse
            u32x4 e = {0u, 0u, 0u, 0u};
            if (ok) e = *reinterpret_cast<const u32x4*>(dst);
            asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                         : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
            float old[8];
            unpack8<T>(e, old);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] += old[j]; g2[j] += old[4 + j]; }
          }
          if (gap_on && live[v]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { csum[c][8 * qp + j] += f[j]; csum[c][8 * qp + 4 + j] += g2[j]; }
          }
          unsigned d0 = pack2<T>(f[0], f[1]), d1 = pack2<T>(f[2], f[3]), d2 = pack2<T>(g2[0], g2[1]), d3 = pack2<T>(g2[2], g2[3]);
          asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
          if (ok) *reinterpret_cast<u32x4*>(dst) = u32x4{d0, d1, d2, d3};
        }
      }
    }
  }
}

template <typename T, int CB, bool GAP>
__global__ __launch_bounds__(256, (CB == 2 && GAP) ? 2 : 3) void lp_k1_kernel(const LpK1Params p) {
  constexpr int VB = 2, RD = 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  // cout group fastest: the groups of one position block run side by side and find its voxels in L2 (as the slow grid dimension the
  // groups were whole passes apart: 64 -> 192 channels read its input three times from HBM)
  const long blk = blockIdx.x / p.ncg;
  const int cg = (int)(blockIdx.x - blk * p.ncg);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  const unsigned wlane = (unsigned)(lane * 16);
  constexpr bool gap_on = GAP;          // (p.gap_part != nullptr; a template parameter so that the 16-32 sum registers exist only there)
  float csum[CB][16];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) csum[c][r] = 0.f;
  for (int it = 0; it < p.nit; ++it) {
    long pos[VB];
    bool live[VB];
    const unsigned short* xb[VB];
  #pragma unroll
    for (int v = 0; v < VB; ++v) {
      pos[v] = (blk * p.nit + it) * LPK1_POS + (wave * VB + v) * 32 + l32;
      live[v] = pos[v] < p.npos;
      xb[v] = p.x + (live[v] ? pos[v] : p.npos - 1) * (long)p.ldx + h * 8;     // (masked lanes re-read the last voxel: no traffic of their own)
    }
    f32x16 acc[VB][CB];
  #pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int cb = cg * CB + c;
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = cb * 32 + 8 * q + 4 * h;
        float bq[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && co + 3 < p.Cout) {       // (the bias is a view into the flat parameter buffer: 4-byte aligned only)
  #pragma unroll
          for (int j = 0; j < 4; ++j) bq[j] = p.bias[co + j];
        }
  #pragma unroll
        for (int v = 0; v < VB; ++v)
  #pragma unroll
          for (int j = 0; j < 4; ++j) acc[v][c][4 * q + j] = bq[j];
      }
    }
    u32x4 ar[RD][CB], br[RD][VB];
    auto issue = [&](int ks, u32x4 (&a)[CB], u32x4 (&b)[VB]) {
  #pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int cb = cg * CB + c;
        a[c] = bload16(wr, wlane, (unsigned)((ks * p.NB + (cb < p.NB ? cb : 0)) * 1024));
      }
  #pragma unroll
      for (int v = 0; v < VB; ++v) b[v] = *reinterpret_cast<const u32x4*>(xb[v] + ks * 16);
    };
  #pragma unroll
    for (int j = 0; j < RD - 1; ++j)
      if (j < p.KS) issue(j, ar[j], br[j]);
    for (int k0 = 0; k0 < p.KS; k0 += RD) {
  #pragma unroll
      for (int j = 0; j < RD; ++j) {
        if (k0 + j < p.KS) {
          if (k0 + j + RD - 1 < p.KS) issue(k0 + j + RD - 1, ar[(j + RD - 1) % RD], br[(j + RD - 1) % RD]);
  #pragma unroll
          for (int v = 0; v < VB; ++v)
  #pragma unroll
            for (int c = 0; c < CB; ++c) acc[v][c] = T::mfma(ar[j][c], br[j][v], acc[v][c]);
        }
      }
    }
    // ---- output side ----
    k1_output<T, CB, VB>(p, acc, pos, live, cg, h, l32, gap_on, csum);
  }
  // Fused global average pool (resnet.py:121: the squeeze of the shortcut output): column sums of this block's 256 positions --
  // lanes by xor shuffles over the 32 positions of a fragment, the 4 waves through LDS in fixed order, one fp64 partial per
  // (position block, cout); the caller's finalize adds the blocks of a sample
  if (gap_on) {   // (launch-uniform)
    __shared__ float csh[4][CB * 32];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = csum[c][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (l32 == 0) csh[wave][c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = s;
      }
    __syncthreads();
    if (tid < CB * 32) {
      const int co = cg * CB * 32 + tid;
      if (co < p.Cout) p.gap_part[blk * p.Cout + co] = ((double)csh[0][tid] + (double)csh[1][tid]) + ((double)csh[2][tid] + (double)csh[3][tid]);
    }
  }
}

// Whole-row form for inputs of 64 x m channels: the k-steps go in groups of four (one 128-byte row piece per voxel and group), each
// group loaded transposed (see k1_quad_transpose) and handed to the matrix instructions after the quad exchange.  One register set
// (a second one for the next group spilled at three waves per SIMD): the overlap comes from the other waves and workgroups of the CU.
template <typename T, int CB, bool GAP>
__global__ __launch_bounds__(256, (CB == 2 && GAP) ? 2 : 3) void lp_k1f_kernel(const LpK1Params p) {
  constexpr int VB = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31, b = l32 & 3;
  const long blk = blockIdx.x / p.ncg;
  const int cg = (int)(blockIdx.x - blk * p.ncg);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  const unsigned wlane = (unsigned)(lane * 16);
  constexpr bool gap_on = GAP;          // (p.gap_part != nullptr; a template parameter so that the 16-32 sum registers exist only there)
  const int NG = p.KS >> 2;
  float csum[CB][16];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) csum[c][r] = 0.f;
  for (int it = 0; it < p.nit; ++it) {
    long pos[VB];
    bool live[VB];
    const unsigned short* xq[VB][4];     // row of quad voxel i, at this lane's piece 2 b + h
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      pos[v] = (blk * p.nit + it) * LPK1_POS + (wave * VB + v) * 32 + l32;
      live[v] = pos[v] < p.npos;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        long q = pos[v] - b + i;
        if (q >= p.npos) q = p.npos - 1;                  // (masked voxels re-read the last one: no traffic of their own)
        xq[v][i] = p.x + q * (long)p.ldx + (2 * b + h) * 8;
      }
    }
    f32x16 acc[VB][CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int cb = cg * CB + c;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = cb * 32 + 8 * q + 4 * h;
        float bq[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && co + 3 < p.Cout) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bq[j] = p.bias[co + j];
        }
#pragma unroll
        for (int v = 0; v < VB; ++v)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[v][c][4 * q + j] = bq[j];
      }
    }
    for (int g = 0; g < NG; ++g) {
      u32x4 a[4][CB], bb[VB][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          const int cb = cg * CB + c;
          a[j][c] = bload16(wr, wlane, (unsigned)(((4 * g + j) * p.NB + (cb < p.NB ? cb : 0)) * 1024));
        }
#pragma unroll
      for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) bb[v][i] = *reinterpret_cast<const u32x4*>(xq[v][i] + g * 64);
#pragma unroll
      for (int v = 0; v < VB; ++v) k1_quad_transpose(bb[v], b);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < VB; ++v)
#pragma unroll
          for (int c = 0; c < CB; ++c) acc[v][c] = T::mfma(a[j][c], bb[v][j], acc[v][c]);
    }
    k1_output<T, CB, VB>(p, acc, pos, live, cg, h, l32, gap_on, csum);
  }
  if (gap_on) {   // (launch-uniform; as lp_k1_kernel)
    __shared__ float csh[4][CB * 32];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = csum[c][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (l32 == 0) csh[wave][c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = s;
      }
    __syncthreads();
    if (tid < CB * 32) {
      const int co = cg * CB * 32 + tid;
      if (co < p.Cout) p.gap_part[blk * p.Cout + co] = ((double)csh[0][tid] + (double)csh[1][tid]) + ((double)csh[2][tid] + (double)csh[3][tid]);
    }
  }
}

static bool k1_enabled() {   // BTS_LP_K1=0: 1x1x1 convs back on the general gather kernel (A/B; read per call)
  const char* e = getenv("BTS_LP_K1");
  return !(e && atoi(e) == 0);
}
// positions per gap partial row of the streaming kernel for samples of V positions (a row never spans two samples: the largest of 2048
// .. 256 that divides V; fewer, longer rows keep the finalize short), or 0 when a call with these dimensions is declined
int bts_lp_k1_gap_block_(long npos, long V, int Cin, int Cout) {
  if (!k1_enabled() || Cin % 16 != 0 || Cout % 8 != 0 || npos < 4096) return 0;
  // (shorter rows = more workgroups were tried for the 160x192x160 volume, 2400 workgroups at 2048 positions per row: within the
  // run-to-run noise at full resolution, 10-25 % slower at the 80x96x80 level where a workgroup's fixed costs stop amortising)
  for (int kb = 2048; kb >= LPK1_POS; kb >>= 1)
    if (V % kb == 0) return kb;
  return 0;
}
// BTS_OK = ran, 1 = declined.  Views: x rows of ldx elements, y rows of ldy, both 16-byte aligned with ld % 8 == 0.
int bts_lp_k1_launch_(int dtype, const void* x, const void* wp, const float* bias, void* y, long npos, int Cin, int ldx, int Cout, int ldy,
                      int accum, double* gap_part, int gap_block, hipStream_t stream) {
  if (bts_lp_k1_gap_block_(npos, LPK1_POS, Cin, Cout) == 0) return 1;
  // (npos % gap_block != 0: the caller vouches that the ragged last row belongs to the only sample)
  if (gap_part != nullptr && (gap_block < LPK1_POS || gap_block % LPK1_POS != 0)) return 1;
  if (ldx % 8 != 0 || ldy % 8 != 0 || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 15)) return 1;
  LpK1Params p;
  p.x = (const unsigned short*)x; p.wp = (const unsigned short*)wp; p.bias = bias; p.y = (unsigned short*)y;
  p.npos = npos; p.ldx = ldx; p.ldy = ldy; p.Cout = Cout; p.KS = Cin / 16; p.NB = (Cout + 31) / 32; p.accum = accum; p.gap_part = gap_part; p.nit = gap_part != nullptr ? gap_block / LPK1_POS : 1;
  const int cb = p.NB >= 2 ? 2 : 1;
  const long blocks = (npos + (long)LPK1_POS * p.nit - 1) / ((long)LPK1_POS * p.nit);
  if (blocks > 0x7fffffffL) return 1;
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(34, 2.0 * Cin * (double)Cout * (double)npos, stream);
  (void)hipGetLastError();
  p.ncg = (p.NB + cb - 1) / cb;
  if (blocks * p.ncg > 0x7fffffffL) return 1;
  const dim3 grid((unsigned)(blocks * p.ncg));
  static const bool lf_on = !(getenv("BTS_LP_K1F") && atoi(getenv("BTS_LP_K1F")) == 0);
  const bool lf = lf_on && Cin % 64 == 0;      // whole 128-byte row pieces per load instruction
  const bool gp = gap_part != nullptr;
#define K1_GO(KERN, T_) do {                                                                                                          \
    if (cb == 2) { if (gp) hipLaunchKernelGGL((KERN<T_, 2, true>), grid, dim3(256), 0, stream, p); else hipLaunchKernelGGL((KERN<T_, 2, false>), grid, dim3(256), 0, stream, p); } \
    else { if (gp) hipLaunchKernelGGL((KERN<T_, 1, true>), grid, dim3(256), 0, stream, p); else hipLaunchKernelGGL((KERN<T_, 1, false>), grid, dim3(256), 0, stream, p); }        \
  } while (0)
  if (lf) { if (dtype == LP_F16) K1_GO(lp_k1f_kernel, TF16); else K1_GO(lp_k1f_kernel, TBF16); }
  else { if (dtype == LP_F16) K1_GO(lp_k1_kernel, TF16); else K1_GO(lp_k1_kernel, TBF16); }
#undef K1_GO
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
