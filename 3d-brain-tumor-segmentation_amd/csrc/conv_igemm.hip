// Implicit-GEMM 3-D convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, 157 TF peak = VALU peak, far easier to keep busy).
//
// One kernel serves every "gather" form the U-Net+VAE hot path needs (reference call sites:
// layers/resnet.py:30-37,80-87,96-103  layers/downsample.py:28-35  layers/upsample.py:28-33
// layers/decoder.py:55-63  layers/vae.py:92-99) and their data-gradients:
//   out[n, o*os+oo, co] (+)= act( bias[co] + sum_t sum_ci in[n, o*s + off_t, ci] * Wp[t][ci][co] )
//     K1    : 1 tap, s=1                    (1x1x1 conv, also its data-gradient with transposed Wp)
//     K3S1  : 27 taps off=t-1, s=1          (TF 'same' pad 1; data-gradient = same form, taps flipped in Wp)
//     DOWN  : 27 taps off=t,   s=2          (TF 'same' on even sizes pads (0,1); also d/dx of Conv3DTranspose)
//     UP(p) : per output-parity class p, taps {k=0:0, k=2:-1} (even) / {k=1:0} (odd), s=1, os=2, oo=p
//             (Conv3DTranspose k3 s2 'same' in gather form: every output voxel written once, no atomics;
//              also d/dx of the stride-2 conv)
//
// Data layout: activations NDHWC fp32 with an explicit pixel stride (ld) so a tensor can be a channel
// slice of a wider "level slab" (virtual concatenation, encoder.py:83-91 / decoder.py:75).
// Weights are pre-packed (bts_conv_pack) as Wp[tap][kgroup(8 ch)][half][Npad][4] so that one
// 16-byte load per lane feeds four MFMAs: lane (half h, column n) holds channels kg*8+h*4+{0..3}.
//
// Work decomposition: 256-thread workgroup = 4 waves; output tile = (32*MS*WM voxels) x (32*NS*WN couts);
// input halo tile for KGS*8 channels staged global->regs->LDS, double buffered (one barrier per stage);
// weight fragments come straight from L2/L1 (shared by the 4 waves), input fragments from LDS via
// ds_read_b128 with a 16B-odd voxel stride (S = KGS*8+4 dwords) so 16-lane groups hit distinct slots.
//
// Also in this translation unit (they share the packed-weight layout, the flag bits and the split-K reduce kernel):
//   upm_kernel  -- UP form with all 8 output-parity classes per workgroup (Conv3DTranspose forward, stride-2 data gradient)
//   k1s_kernel  -- LDS-free streaming 1x1x1 conv for big grids
//   dsc_kernel  -- vector-ALU direct 3x3x3 conv for <= 4 output channels
//   c2_kernel   -- 3x3x3 conv of a 2-channel input (the MFMA's K = 2 is the channel pair)
// The K3S1 weight image carries a second, Winograd-domain part; stride-1 3x3x3 launches are offered to conv_wino.hip first.
// and the fused entry points: shortcut pair (FUSE2), GroupNorm statistics in the epilogue, data-gradient pair (second input).
#include <stdlib.h>
#include "common.h"
#include "bts_internal.h"

#define MAXSLOT 8

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

struct IgemmParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int N, Di, Hi, Wi, Cin, ldx;
  int Do, Ho, Wo;  // iteration (class) grid
  int Cout, ldy, Npad, KG;
  int ODa, OHa, OWa;  // actual output tensor dims
  int os, ooz, ooy, oox;
  int s, loz, loy, lox;
  int IZ, IY, IX;
  int lgTX, lgTY, TZ;
  int ntz, nty, ntx;
  int ntaps, flags;
  int tap_sz, tap_sy, tap_sx;  // LDS dword strides of the (z,y,x) tap axes (27-tap geometries)
  int tap_lds[27];
  int tap_w[27];
  // fused second output (FUSE2): y2 = bias2 + 1x1x1 conv of the same input with wp2 (K1 packing), evaluated at the centre
  // tap of the 27-tap sweep -- the ResNet block's shortcut conv shares conv1's input tile (resnet.py:118,134)
  const float* wp2;
  const float* bias2;
  float* y2;
  int ldy2;
  // fused second INPUT (27-tap kernels, never together with FUSE2): out += 1x1x1 conv of x2 with wp2 (K1 packing), evaluated at
  // the centre tap -- the ResNet block's data gradient dx = bwd(conv1)(dc1) + bwd(shortcut)(dres) in one pass
  // (resnet.py:118,134); the x2 fragment is a 16-byte global load of the voxel's own row (no halo, no LDS)
  const float* x2;
  int ldx2;
  // split-K: blockIdx.z handles k-groups [z*kg_per, ...); raw partials go to part[z][voxel][Npad] (no bias/act)
  int ksplit, kg_per;
  float* part;
  long ws_bytes, ws_need;  // host-side planning only
  int plan_only;
  int dbg;  // profiling aid (BTS_IGEMM_DBG): 1 = skip the MFMA sweep, 2 = skip re-staging after the first stage
  // fused GroupNorm statistics of the output (slab semantics: group = z-slab of D/G planes, whole tiles per group):
  // every workgroup writes (sum, sum of squares) of its tile to gnp[((n*G+g)*gn_B + b)*2], b = tile index inside the group
  double* gnp;
  int gn_G, gn_zt;  // groups; z-tiles per group
  int gn_gridy;     // host side: grid.y of the launch (partials per group = gn_zt*nty*ntx*gn_gridy)
  // merged launch of the 8 output-parity classes of the UP geometry (blockIdx.z = class)
  int ncls;
  int cls_nt[8];
  int cls_tl[8][8];
  int cls_tw[8][8];
};

#define IG_FLAG_BIAS 1
#define IG_FLAG_ACCUM 2
#define IG_FLAG_SIGMOID 4
#define IG_FLAG_VECIN 8
#define IG_FLAG_VECOUT 16

// One stage of the implicit GEMM: nkg k-groups (8 input channels each) x NT taps, fully unrolled over the taps.
template <int NT, int MS, int NS, bool F2 = false, bool FIXG = false>
__device__ __forceinline__ void stage_taps(const IgemmParams& p, const int* tlp, const int* twp, const float* cur,
                                           const int (&bbase)[MS], const int (&lane_woff)[NS], int kg0, int nkg,
                                           f32x16 (&acc)[MS][NS], f32x16 (*acc2)[NS], const float* const (&x2row)[MS],
                                           bool have_x2) {
  const int wstepKG = 2 * p.Npad * 4;       // floats between consecutive k-groups of one tap
  const int wstepTap = p.KG * wstepKG;      // floats between consecutive taps
  // 27-tap geometries (k3s1, DOWN): LDS offset and weight tap index are arithmetic in the compile-time tap number, so
  // nothing is loaded inside the tap sequence (an s_load there shares lgkmcnt with the ds_read prefetch and drains it);
  // the UP parity classes (<= 8 taps) keep their small tables in SGPRs.
  int tl[NT], tw[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (NT == 27) {
      // FIXG: the 32x4xTZ tile of the big-grid configs has a compile-time halo geometry (IX=34, IY=6, S=12), so every
      // tap offset folds into the ds_read immediate: no per-tap address VALU, no address registers
      tl[t] = FIXG ? (((t / 9) * 6 + (t / 3) % 3) * 34 + (t % 3)) * 12
                   : (t / 9) * p.tap_sz + ((t / 3) % 3) * p.tap_sy + (t % 3) * p.tap_sx;
      tw[t] = t * wstepTap;
    } else {
      tl[t] = tlp[t];
      tw[t] = twp[t] * wstepTap;
    }
  }
  for (int kgl = 0; kgl < nkg; ++kgl) {
    const int wk = (kg0 + kgl) * wstepKG;
    const float* lb = cur + kgl * 8;
    // prefetch depths: weights (L2, ~500-900 cycles) AD taps ahead, input fragments (LDS, ~130 cycles) BD taps ahead;
    // deep enough that a single wave per SIMD keeps the matrix pipe fed while its partner workgroup is in its prologue
    constexpr int AD = 2, BD = 1;  // deeper (3/2) measured neutral on MI355X and costs registers
    f32x4 a[AD + 1][NS], b[BD + 1][MS];
    f32x4 a2s[NS];
    f32x4 b2in[MS];
    const bool fin = (NT == 27) && !F2 && have_x2;
    if (fin) {  // second input: this voxel's 8 channels of the k-group, requested a whole stage ahead of their use
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) b2in[ms] = *reinterpret_cast<const f32x4*>(x2row[ms] + (kg0 + kgl) * 8);
    }
#pragma unroll
    for (int q = 0; q < AD; ++q)
      if (q < NT) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          a[q][ns] = *reinterpret_cast<const f32x4*>((p.wp + (wk + tw[q < NT ? q : 0])) + lane_woff[ns]);
      }
#pragma unroll
    for (int q = 0; q < BD; ++q)
      if (q < NT) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) b[q][ms] = *reinterpret_cast<const f32x4*>(lb + bbase[ms] + tl[q < NT ? q : 0]);
      }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + AD < NT) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          a[(t + AD) % (AD + 1)][ns] =
              *reinterpret_cast<const f32x4*>((p.wp + (wk + tw[t + AD < NT ? t + AD : 0])) + lane_woff[ns]);  // scalar base + lane offset
      }
      if (t + BD < NT) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
          b[(t + BD) % (BD + 1)][ms] = *reinterpret_cast<const f32x4*>(lb + bbase[ms] + tl[t + BD < NT ? t + BD : 0]);
      }
      if (F2 && NT == 27 && t == 11) {  // shortcut-conv weights, two taps ahead of the centre tap
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) a2s[ns] = *reinterpret_cast<const f32x4*>((p.wp2 + wk) + lane_woff[ns]);
      }
      if (!F2 && NT == 27 && t == 11) {
        if (fin) {
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) a2s[ns] = *reinterpret_cast<const f32x4*>((p.wp2 + wk) + lane_woff[ns]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // prefetches are issued before this tap's MFMAs, not sunk behind them
      if (F2 && NT == 27 && t == 13) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
              acc2[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2s[ns][j], b[t % (BD + 1)][ms][j], acc2[ms][ns], 0, 0, 0);
      }
      if (!F2 && NT == 27 && t == 13) {
        if (fin) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ms = 0; ms < MS; ++ms)
#pragma unroll
              for (int ns = 0; ns < NS; ++ns)
                acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2s[ns][j], b2in[ms][j], acc[ms][ns], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t % (AD + 1)][ns][j], b[t % (BD + 1)][ms][j], acc[ms][ns], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the hand-written 2-deep pipeline: no hoisting of later taps' loads
    }
  }
}

// TRI: three resident workgroups per CU instead of two (single-buffered halo tile, <= 168 VGPRs): whenever one
// workgroup is in its prologue / refill / epilogue the SIMD still holds two MFMA-issuing waves.
#define IG_TRI(MS, NS, KGS, FUSE2, FIXG) ((MS) == 2 && (NS) == 1 && (KGS) == 1 && !(FUSE2) && (FIXG))
template <int MS, int NS, int WM, int WN, int KGS, bool FUSE2 = false, bool FIXG = false, bool T27 = false>
__global__ __launch_bounds__(256, IG_TRI(MS, NS, KGS, FUSE2, FIXG) ? 3 : 2) void igemm_kernel(const IgemmParams p) {
  // MS == 4 (512-voxel tile, 16 MFMAs per tap for 32-cout layers): its 59 KB halo tile is single-buffered so that two
  // workgroups still fit a CU; the partner workgroup covers the (short) LDS refill between stages
  constexpr int NSLOT = (MS == 4) ? 10 : MAXSLOT;
  constexpr bool SINGLE = (MS == 4) || IG_TRI(MS, NS, KGS, FUSE2, FIXG);
  constexpr int S = KGS * 8 + 4;  // dwords per staged voxel (pad 4: 16B-odd stride)
  constexpr int QPV = KGS * 2;    // float4 slots per voxel
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l32 = lane & 31;

  int b = blockIdx.x;
  const int tx = b % p.ntx; b /= p.ntx;
  const int ty = b % p.nty; b /= p.nty;
  const int tz = b % p.ntz;
  const int n = b / p.ntz;
  const int TX = 1 << p.lgTX, TY = 1 << p.lgTY;
  int ntaps = p.ntaps, ooz = p.ooz, ooy = p.ooy, oox = p.oox;
  const int* tlp = p.tap_lds;
  const int* twp = p.tap_w;
  if (p.ncls > 1) {
    const int cls = blockIdx.z;
    ntaps = p.cls_nt[cls]; tlp = p.cls_tl[cls]; twp = p.cls_tw[cls];
    ooz = (cls >> 2) & 1; ooy = (cls >> 1) & 1; oox = cls & 1;
  }
  const int oz0 = tz * p.TZ, oy0 = ty * TY, ox0 = tx * TX;
  const int iz0 = oz0 * p.s + p.loz, iy0 = oy0 * p.s + p.loy, ix0 = ox0 * p.s + p.lox;
  const int tileVox = p.IZ * p.IY * p.IX;
  const int bufDw = tileVox * S;

  // ---- staging map: slot e = tid + i*256 -> (voxel, quad). Only a 32-bit element offset relative to the tile origin
  //      is kept per slot (-1 = zero padding); LDS offset and channel quad are recomputed from e (QPV is a constant).
  const int nslots = tileVox * QPV;
  const float* xbase = p.x + ((((long)n * p.Di + iz0) * p.Hi + iy0) * p.Wi + ix0) * (long)p.ldx;  // wave-uniform
  int goff[NSLOT];
  const int IYX = p.IY * p.IX;
  const float invIX = 1.0f / (float)p.IX, invIYX = 1.0f / (float)IYX;
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int e = tid + i * 256;
    goff[i] = -1;
    if (e < nslots) {
      const int vox = e / QPV, q = e - vox * QPV;
      // exact small-integer division by float reciprocal (operands < 2^16): the prologue is on every workgroup's
      // critical path, integer division would cost ~50 VALU instructions per slot
      const int vz = (int)(((float)vox + 0.5f) * invIYX);
      const int r = vox - vz * IYX;
      const int vy = (int)(((float)r + 0.5f) * invIX);
      const int vx = r - vy * p.IX;
      if ((unsigned)(iz0 + vz) < (unsigned)p.Di && (unsigned)(iy0 + vy) < (unsigned)p.Hi && (unsigned)(ix0 + vx) < (unsigned)p.Wi)
        goff[i] = ((vz * p.Hi + vy) * p.Wi + vx) * p.ldx + q * 4;
    }
  }

  // ---- fragment bases ----
  int bbase[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int m = (wm * MS + ms) * 32 + l32;
    const int mx = m & (TX - 1);
    const int my = (m >> p.lgTX) & (TY - 1);
    const int mz = m >> (p.lgTX + p.lgTY);
    bbase[ms] = ((mz * p.s * p.IY + my * p.s) * p.IX + mx * p.s) * S + h * 4;
  }
  int lane_woff[NS];  // float offset of this lane's weight quad inside one (tap, k-group) image
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    int ncol = ((blockIdx.y * WN + wn) * NS + ns) * 32 + l32;
    if (ncol >= p.Npad) ncol = p.Npad - 1;  // tile wider than the padded cout range: results are discarded
    lane_woff[ns] = (h * p.Npad + ncol) * 4;
  }

  const float* x2row[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    x2row[ms] = nullptr;
    if (p.x2) {
      const int m = (wm * MS + ms) * 32 + l32;
      int oz = oz0 + (m >> (p.lgTX + p.lgTY)), oy = oy0 + ((m >> p.lgTX) & (TY - 1)), ox = ox0 + (m & (TX - 1));
      if (oz >= p.Do) oz = p.Do - 1;   // voxels past the edge are computed and discarded: keep their loads in bounds
      if (oy >= p.Ho) oy = p.Ho - 1;
      if (ox >= p.Wo) ox = p.Wo - 1;
      x2row[ms] = p.x2 + ((((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * (long)p.ldx2 + h * 4;
    }
  }
  const bool have_x2 = p.x2 != nullptr;
  f32x16 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;
  f32x16 acc2[FUSE2 ? MS : 1][NS];
  if (FUSE2) {
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[ms][ns][r] = 0.f;
  }

  int kgBeg = 0, kgEnd = p.KG;
  if (p.ksplit > 1) {
    kgBeg = blockIdx.z * p.kg_per;
    kgEnd = kgBeg + p.kg_per;
    if (kgEnd > p.KG) kgEnd = p.KG;
  }
  const int stBeg = kgBeg / KGS;  // kg_per is a multiple of KGS
  const int nstages = (kgEnd - kgBeg + KGS - 1) / KGS;
  const bool vecin = (p.flags & IG_FLAG_VECIN) != 0;

  f32x4 pre[NSLOT];
  auto fetch = [&](int st) {
    const int c0 = (stBeg + st) * KGS * 8;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (goff[i] >= 0) {
        const int c = c0 + ((tid + i * 256) % QPV) * 4;
        const float* src = xbase + goff[i] + c0;
        if (vecin) {
          if (c < p.Cin) v = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (c + 0 < p.Cin) v[0] = src[0];
          if (c + 1 < p.Cin) v[1] = src[1];
          if (c + 2 < p.Cin) v[2] = src[2];
          if (c + 3 < p.Cin) v[3] = src[3];
        }
      }
      pre[i] = v;
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = tid + i * 256;
      if (e < nslots) *reinterpret_cast<f32x4*>(buf + (e / QPV) * S + (e % QPV) * 4) = pre[i];
    }
  };

  fetch(0);
  commit(lds);
  __syncthreads();

  for (int st = 0; st < nstages; ++st) {
    const float* cur = SINGLE ? lds : lds + (st & 1) * bufDw;
    float* nxt = SINGLE ? lds : lds + ((st + 1) & 1) * bufDw;
    const bool more = (st + 1) < nstages;
    if (more && BTS_DBG(p) != 2) fetch(st + 1);

    // k-groups of this stage x taps: fully unrolled tap sequence (tables hoisted to SGPRs), weight fragments
    // prefetched two taps ahead and input fragments one tap ahead so their latencies sit under the MFMAs
    const int kg0 = (stBeg + st) * KGS;
    int nkg = kgEnd - kg0;
    if (nkg > KGS) nkg = KGS;
    if (BTS_DBG(p) == 1) {
    } else if constexpr (FIXG || T27 || FUSE2) {  // 27-tap form known at compile time (fixed geometry, the fused pair, or
      // the launcher's T27 instantiation for k3s1 / stride-2 convs): no runtime tap-count dispatch, so the accumulators
      // are not shuffled between register sets around a switch
      stage_taps<27, MS, NS, FUSE2, FIXG>(p, tlp, twp, cur, bbase, lane_woff, kg0, nkg, acc, acc2, x2row, have_x2);
    } else if constexpr (KGS == 1) {
      switch (ntaps) {
        case 27: stage_taps<27, MS, NS, FUSE2, FIXG>(p, tlp, twp, cur, bbase, lane_woff, kg0, nkg, acc, acc2, x2row, have_x2); break;
        case 8: stage_taps<8, MS, NS>(p, tlp, twp, cur, bbase, lane_woff, kg0, nkg, acc, nullptr, x2row, false); break;
        case 4: stage_taps<4, MS, NS>(p, tlp, twp, cur, bbase, lane_woff, kg0, nkg, acc, nullptr, x2row, false); break;
        case 2: stage_taps<2, MS, NS>(p, tlp, twp, cur, bbase, lane_woff, kg0, nkg, acc, nullptr, x2row, false); break;
        default: stage_taps<1, MS, NS>(p, tlp, twp, cur, bbase, lane_woff, kg0, nkg, acc, nullptr, x2row, false); break;
      }
    } else {  // the 4-k-group staging variant only serves the 1x1x1 convolutions
      stage_taps<1, MS, NS>(p, tlp, twp, cur, bbase, lane_woff, kg0, nkg, acc, nullptr, x2row, false);
    }
    if (SINGLE) __syncthreads();  // every wave has finished reading the tile before it is overwritten
    if (more && BTS_DBG(p) != 2) commit(nxt);
    __syncthreads();
  }

  // ---- split-K: raw partial tile to the workspace, finished by igemm_reduce_kernel ----
  if (p.ksplit > 1) {
    const long nvox = (long)p.N * p.Do * p.Ho * p.Wo;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      const int m = (wm * MS + ms) * 32 + l32;
      const int oz = oz0 + (m >> (p.lgTX + p.lgTY)), oy = oy0 + ((m >> p.lgTX) & (TY - 1)), ox = ox0 + (m & (TX - 1));
      if (oz >= p.Do || oy >= p.Ho || ox >= p.Wo) continue;
      const long v = (((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
      float* row = p.part + ((long)blockIdx.z * nvox + v) * p.Npad;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int nb = ((blockIdx.y * WN + wn) * NS + ns) * 32;
        if (nb >= p.Npad) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v4 = {acc[ms][ns][4 * g + 0], acc[ms][ns][4 * g + 1], acc[ms][ns][4 * g + 2], acc[ms][ns][4 * g + 3]};
          *reinterpret_cast<f32x4*>(row + nb + 8 * g + 4 * h) = v4;
        }
      }
    }
    return;
  }

  // ---- epilogue: D rows = couts (4 consecutive per register quad), cols = voxels ----
  const bool vecout = (p.flags & IG_FLAG_VECOUT) != 0;
  float gn_s = 0.f, gn_q = 0.f;  // this lane's share of the tile's GroupNorm sums (<= 64 values)
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int m = (wm * MS + ms) * 32 + l32;
    const int mx = m & (TX - 1);
    const int my = (m >> p.lgTX) & (TY - 1);
    const int mz = m >> (p.lgTX + p.lgTY);
    const int oz = oz0 + mz, oy = oy0 + my, ox = ox0 + mx;
    if (oz >= p.Do || oy >= p.Ho || ox >= p.Wo) continue;
    const long pix = (((long)n * p.ODa + (oz * p.os + ooz)) * p.OHa + (oy * p.os + ooy)) * p.OWa + (ox * p.os + oox);
    float* yrow = p.y + pix * p.ldy;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int nb = ((blockIdx.y * WN + wn) * NS + ns) * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = nb + 8 * g + 4 * h;
        if (co >= p.Cout) continue;
        f32x4 v = {acc[ms][ns][4 * g + 0], acc[ms][ns][4 * g + 1], acc[ms][ns][4 * g + 2], acc[ms][ns][4 * g + 3]};
        if (p.flags & IG_FLAG_BIAS) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Cout) v[j] += p.bias[co + j];
        }
        if (p.gnp) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Cout) { gn_s += v[j]; gn_q = fmaf(v[j], v[j], gn_q); }
        }
        if (p.flags & IG_FLAG_SIGMOID) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = sigmoidf_(v[j]);
        }
        if (vecout) {
          f32x4* dst = reinterpret_cast<f32x4*>(yrow + co);
          if (p.flags & IG_FLAG_ACCUM) {
            f32x4 o = *dst;
            v += o;
          }
          *dst = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Cout) {
              float r = v[j];
              if (p.flags & IG_FLAG_ACCUM) r += yrow[co + j];
              yrow[co + j] = r;
            }
        }
        if (FUSE2) {  // shortcut output: bias only, dense or strided rows, same cout tiling
          float* y2row = p.y2 + pix * p.ldy2;
          f32x4 v2 = {acc2[FUSE2 ? ms : 0][ns][4 * g + 0], acc2[FUSE2 ? ms : 0][ns][4 * g + 1],
                      acc2[FUSE2 ? ms : 0][ns][4 * g + 2], acc2[FUSE2 ? ms : 0][ns][4 * g + 3]};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (p.bias2 && co + j < p.Cout) v2[j] += p.bias2[co + j];
          if (vecout && (p.ldy2 % 4 == 0)) *reinterpret_cast<f32x4*>(y2row + co) = v2;
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (co + j < p.Cout) y2row[co + j] = v2[j];
          }
        }
      }
    }
  }
  if (p.gnp) {  // fixed-order combine: lanes (shuffle tree) -> 4 waves (LDS) -> one (sum, sumsq) pair per workgroup
    double ds = wave_sum_f64((double)gn_s), dq = wave_sum_f64((double)gn_q);
    double* sh = reinterpret_cast<double*>(lds);  // the staging buffers are idle (last barrier passed)
    if (lane == 0) { sh[wave * 2] = ds; sh[wave * 2 + 1] = dq; }
    __syncthreads();
    if (tid == 0) {
      const int g = tz / p.gn_zt;
      const long B = (long)p.gn_zt * p.nty * p.ntx * gridDim.y;
      const long b_ = (((long)(tz - g * p.gn_zt) * p.nty + ty) * p.ntx + tx) * gridDim.y + blockIdx.y;
      double* o = p.gnp + (((long)n * p.gn_G + g) * B + b_) * 2;
      o[0] = sh[0] + sh[2] + sh[4] + sh[6];
      o[1] = sh[1] + sh[3] + sh[5] + sh[7];
    }
  }
}

// split-K finish: y[v][co] (+)= act(bias[co] + sum_z part[z][v][co]); fixed summation order (deterministic)
__global__ __launch_bounds__(256) void igemm_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                           float* __restrict__ y, long nvox, int Cout, int Npad, int ldy,
                                                           int ksplit, int flags) {
  const long total = nvox * Cout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / Cout;
    const int co = (int)(i - v * Cout);
    float s = 0.f;
    for (int z = 0; z < ksplit; ++z) s += part[((long)z * nvox + v) * Npad + co];
    if (flags & IG_FLAG_BIAS) s += bias[co];
    if (flags & IG_FLAG_SIGMOID) s = sigmoidf_(s);
    float* d = y + v * ldy + co;
    *d = (flags & IG_FLAG_ACCUM) ? *d + s : s;
  }
}

int bts_igemm_reduce_(const float* part, const float* bias, float* y, long nvox, int Cout, int Npad, int ldy, int ksplit,
                      int with_bias, int accum, hipStream_t stream) {
  long blocks = (nvox * Cout + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError();
  hipLaunchKernelGGL(igemm_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, part, bias, y, nvox, Cout, Npad, ldy, ksplit,
                     (with_bias ? IG_FLAG_BIAS : 0) | (accum ? IG_FLAG_ACCUM : 0));
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight packing (role = forward or data-gradient). Source layouts are the reference's Keras layouts:
// Conv3D (kd,kh,kw,Cin,Cout) -- resnet.py:30-37,80-87; Conv3DTranspose (kd,kh,kw,Cout,Cin) -- upsample.py:28-33.
// The encoder's dense connections feed block j the list [o_{j-1}, o_0..o_{j-1}] (encoder.py:83-87): the
// duplicated slice is folded here (Wp = W[first copy] + W[second copy]) so the kernel reads the level slab
// [o_0..o_{j-1}] once.
// ---------------------------------------------------------------------------------------------
struct PackParams {
  const float* w;
  float* wp;
  int ntaps, K, N, KG, Npad;
  long sT, sK, sN;  // source strides for (tap, k, n)
  int flip;         // tap t reads source tap ntaps-1-t
  int cin_is_k;     // 1: the (possibly folded) input-channel axis is k, 0: it is n, -1: no fold
  int shift, dup_start;
  long wino_off;    // K3S1 images: element index where the Winograd part starts; otherwise beyond the image
  long w3_item0;    // K3S1 images: first work item of the 3-D Winograd part (conv_wino3.hip); its floats start at w3_off
  long w3_off;
  long total;       // floats of the whole image
  long items;       // work items (pack_item): implicit-GEMM floats + Winograd positions (16 floats each), of the parts in `parts` only
  unsigned parts;   // K3S1 images: which of the three parts this pack writes (bit 0 implicit-GEMM, 1 F(2x2,3x3) x direct, 2 F(2x2x2,3x3x3)); 7 = all
};
// index into the items of the enabled parts -> work item of the whole image
__device__ __forceinline__ long pack_map(const PackParams& q, long j) {
  if (q.parts == 7u) return j;
  const long n0 = (q.parts & 1u) ? q.wino_off : 0;
  if (j < n0) return j;
  j -= n0;
  const long n1 = (q.parts & 2u) ? (q.w3_item0 - q.wino_off) : 0;
  if (j < n1) return q.wino_off + j;
  return q.w3_item0 + (j - n1);
}

// source value of packed position (tap t, contraction index k, column n), with the tap flip of the data-gradient role and
// the encoder's duplicated input slice folded in
__device__ __forceinline__ float pack_src(const PackParams& q, int t, int k, int n) {
  if (k >= q.K || n >= q.N) return 0.f;
  const int ts = q.flip ? (q.ntaps - 1 - t) : t;
  int kk = k, nn = n, k2 = -1, n2 = -1;
  if (q.cin_is_k == 1) {
    if (k >= q.dup_start) k2 = k - q.dup_start;
    kk = k + q.shift;
    n2 = n;
  } else if (q.cin_is_k == 0) {
    if (n >= q.dup_start) n2 = n - q.dup_start;
    nn = n + q.shift;
    k2 = k;
  }
  float v = q.w[ts * q.sT + kk * q.sK + nn * q.sN];
  if (q.shift > 0 && k2 >= 0 && n2 >= 0) v += q.w[ts * q.sT + k2 * q.sK + n2 * q.sN];
  return v;
}

// One work item of an image: index i < wino_off writes one float of the implicit-GEMM part; an index above it writes the
// 16 transform points of one (cout block, k-group, x tap, half, cout, cin) position of the Winograd part (conv_wino.hip):
// U = G g G^T over the (z, y) taps, layout [cout block of 32][k-group][x tap][xi_z*4 + xi_y][half][32][4]; the 9 source
// taps are read once.  rows of G: (1 0 0) (.5 .5 .5) (.5 -.5 .5) (0 0 1)
__device__ __forceinline__ float pack_g_(int r, float g0, float g1, float g2) {   // row r of G applied to three taps
  return r == 0 ? g0 : r == 1 ? fmaf(0.5f, g2, fmaf(0.5f, g1, 0.5f * g0)) : r == 2 ? fmaf(0.5f, g2, fmaf(-0.5f, g1, 0.5f * g0)) : g2;
}
__device__ __forceinline__ void pack_item(const PackParams& q, long i) {
  if (i >= q.w3_item0) {
    // third part: U = (G x G x G) g over all 27 taps, 64 transform points xi = (xi_z*4 + xi_y)*4 + xi_x per (cin, cout) pair,
    // layout [cout block of 32][k-group of 8 cin][xi][half][32 couts][4 cin] (conv_wino3.hip)
    long r = i - q.w3_item0;
    const int j = (int)(r & 3);
    const int n32 = (int)((r >> 2) & 31);
    const int hh = (int)((r >> 7) & 1);
    r >>= 8;
    const int kg = (int)(r % q.KG);
    const int cb = (int)(r / q.KG);
    const int k = kg * 8 + hh * 4 + j, n = cb * 32 + n32;
    float t[4][3][3], u[4][4][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float g0 = pack_src(q, (0 * 3 + ky) * 3 + kx, k, n), g1 = pack_src(q, (1 * 3 + ky) * 3 + kx, k, n),
                    g2 = pack_src(q, (2 * 3 + ky) * 3 + kx, k, n);
#pragma unroll
        for (int a = 0; a < 4; ++a) t[a][ky][kx] = pack_g_(a, g0, g1, g2);
      }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int b = 0; b < 4; ++b) u[a][b][kx] = pack_g_(b, t[a][0][kx], t[a][1][kx], t[a][2][kx]);
    float* o = q.wp + q.w3_off + (((long)cb * q.KG + kg) * 64) * 256 + hh * 128 + n32 * 4 + j;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) o[((a * 4 + b) * 4 + c) * 256] = pack_g_(c, u[a][b][0], u[a][b][1], u[a][b][2]);
    return;
  }
  if (i >= q.wino_off) {
    long r = i - q.wino_off;
    const int j = (int)(r & 3);
    const int n32 = (int)((r >> 2) & 31);
    const int hh = (int)((r >> 7) & 1);
    r >>= 8;
    const int dx = (int)(r % 3); r /= 3;
    const int kg = (int)(r % q.KG);
    const int cb = (int)(r / q.KG);
    const int k = kg * 8 + hh * 4 + j, n = cb * 32 + n32;
    float g[3][3], t[4][3];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) g[kz][ky] = pack_src(q, (kz * 3 + ky) * 3 + dx, k, n);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {  // G along z
      t[0][ky] = g[0][ky];
      t[1][ky] = fmaf(0.5f, g[2][ky], fmaf(0.5f, g[1][ky], 0.5f * g[0][ky]));
      t[2][ky] = fmaf(0.5f, g[2][ky], fmaf(-0.5f, g[1][ky], 0.5f * g[0][ky]));
      t[3][ky] = g[2][ky];
    }
    float* o = q.wp + q.wino_off + ((((long)cb * q.KG + kg) * 3 + dx) * 16) * 256 + hh * 128 + n32 * 4 + j;
#pragma unroll
    for (int xz = 0; xz < 4; ++xz) {  // G along y
      o[(xz * 4 + 0) * 256] = t[xz][0];
      o[(xz * 4 + 1) * 256] = fmaf(0.5f, t[xz][2], fmaf(0.5f, t[xz][1], 0.5f * t[xz][0]));
      o[(xz * 4 + 2) * 256] = fmaf(0.5f, t[xz][2], fmaf(-0.5f, t[xz][1], 0.5f * t[xz][0]));
      o[(xz * 4 + 3) * 256] = t[xz][2];
    }
    return;
  }
  const int j = (int)(i & 3);
  long r = i >> 2;
  const int n = (int)(r % q.Npad); r /= q.Npad;
  const int hh = (int)(r & 1); r >>= 1;
  const int kg = (int)(r % q.KG);
  const int t = (int)(r / q.KG);
  q.wp[i] = pack_src(q, t, kg * 8 + hh * 4 + j, n);
}

__global__ void pack_kernel(const PackParams q) {
  const long total = q.items;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    pack_item(q, pack_map(q, i));
}

// All layers' weight images in one launch (the optimiser step invalidates every image at once; 120 separate 5 us
// launches per training step otherwise).  Descriptor table in device memory, block -> entry by binary search.
#define PACK_BLOCK_ELEMS 2048
struct PackDesc {
  PackParams q;
  long first_block, total;
};
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackDesc* __restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  const long b = blockIdx.x;
  while (lo < hi) {  // last entry with first_block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  const long i0 = (b - d.first_block) * PACK_BLOCK_ELEMS;
#pragma unroll
  for (int u = 0; u < PACK_BLOCK_ELEMS / 256; ++u) {
    const long i = i0 + threadIdx.x + u * 256;
    if (i < d.total) pack_item(d.q, pack_map(d.q, i));
  }
}

static inline int npad32(int n) { return (n + 31) / 32 * 32; }

extern "C" long bts_conv_packed_floats(int kind, int role, int Cin, int Cout) {
  const int ntaps = (kind == BTS_CONV_K1) ? 1 : 27;
  const int K = (role == BTS_ROLE_FWD) ? Cin : Cout;
  const int N = (role == BTS_ROLE_FWD) ? Cout : Cin;
  // K3S1 images carry two Winograd-domain copies: 16 transform points x 3 x taps (conv_wino.hip), 64 points (conv_wino3.hip)
  return (long)(ntaps + (kind == BTS_CONV_K3S1 ? 48 + 64 : 0)) * ((K + 7) / 8) * 2 * npad32(N) * 4;
}

static int pack_params(PackParams& q, int kind, int role, const float* w, float* wp, int Cin_ref, int Cout, int Cin_slab,
                       int dup_start, int dup_shift, unsigned parts = 7u) {
  if (kind < 0 || kind > 3 || role < 0 || role > 1) return BTS_ERR_UNSUPPORTED;
  if (dup_shift > 0 && (kind == BTS_CONV_K3S2 || kind == BTS_CONV_K3S2T)) return BTS_ERR_UNSUPPORTED;
  if (Cin_slab + dup_shift != Cin_ref) return BTS_ERR_SHAPE;
  q.w = w;
  q.wp = wp;
  q.ntaps = (kind == BTS_CONV_K1) ? 1 : 27;
  q.shift = dup_shift;
  q.dup_start = dup_shift > 0 ? dup_start : (1 << 30);
  long sCin, sCout;
  if (kind == BTS_CONV_K3S2T) { sCout = Cin_ref; sCin = 1; }  // (t, Cout, Cin)
  else { sCin = Cout; sCout = 1; }                            // (t, Cin, Cout)
  q.sT = (long)Cin_ref * Cout;
  if (role == BTS_ROLE_FWD) {
    q.K = Cin_slab; q.N = Cout; q.sK = sCin; q.sN = sCout; q.flip = 0; q.cin_is_k = 1;
  } else {
    q.K = Cout; q.N = Cin_slab; q.sK = sCout; q.sN = sCin; q.cin_is_k = 0;
    q.flip = (kind == BTS_CONV_K3S1) ? 1 : 0;
  }
  q.KG = (q.K + 7) / 8;
  q.Npad = npad32(q.N);
  const long ig = (long)q.ntaps * q.KG * 2 * q.Npad * 4;
  q.wino_off = (kind == BTS_CONV_K3S1) ? ig : (1L << 62);
  const long pairs = (long)q.KG * 2 * q.Npad * 4;   // (contraction index, column) pairs of the padded image
  q.total = ig + ((kind == BTS_CONV_K3S1) ? (48L + 64L) * pairs : 0L);
  q.w3_item0 = (kind == BTS_CONV_K3S1) ? ig + 3L * pairs : (1L << 62);
  q.w3_off = ig + 48L * pairs;
  q.parts = 7u;
  q.items = ig + ((kind == BTS_CONV_K3S1) ? (3L + 1L) * pairs : 0L);
  if (kind == BTS_CONV_K3S1 && (parts & 7u) != 7u && (parts & 7u) != 0u) {
    q.parts = parts & 7u;
    q.items = ((q.parts & 1u) ? ig : 0L) + ((q.parts & 2u) ? 3L * pairs : 0L) + ((q.parts & 4u) ? pairs : 0L);
  }
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Which parts of a K3S1 image are READ.  An image carries three forms of the same weights (139 floats per (cin, cout) pair: 27 + 48 +
// 64) and a layer uses ONE of them at a given geometry -- re-packing all three after every optimiser step wrote ~1.6 GB per fp32
// training step (pack_batch_kernel 0.67-0.77 ms, round-4 review).  The library therefore keeps, per image (keyed by its base address),
// the parts the dispatcher has picked so far (`used`, recorded right before each launch), and descriptor tables are built for those
// parts only (`table_mask`).  A launch that picks a part no table covers packs it on the spot, on its own stream, from the recorded
// source (`fresh_extra` remembers that until the next batch pack, i.e. the next weight change), and bumps the generation counter so that
// the host rebuilds its table (ops.PackTable) at the next re-pack.  bts_conv_pack (single image, all parts) is how images are born.
// ---------------------------------------------------------------------------------------------
#include <mutex>
#include <unordered_map>
struct ImgState {
  int kind, role, Cin_ref, Cout, Cin_slab, dup_start, dup_shift;
  const float* w;
  unsigned used, table_mask, fresh_extra;
};
static std::mutex g_img_mu;
static std::unordered_map<const void*, ImgState> g_img;
static long g_img_gen = 0;
static bool img_masks_enabled() {   // BTS_PACK_USED=0: every pack writes all three parts (A/B; read per call)
  const char* e = getenv("BTS_PACK_USED");
  return !(e && atoi(e) == 0);
}
// (re)register an image; returns the parts a table built now should describe.  table_mask is what the table LAST RUN on the image
// wrote (set by bts_conv_pack_batch from the host copy of that table), never what some table merely describes: with two tables for one
// image, or a cached narrow table behind a bts_conv_pack of the whole image, the parts a batch did NOT write must go stale (round-5
// advisor finding: a re-registration through bts_conv_pack used to mark all three parts as table-covered).
static unsigned img_register(const float* wp, int kind, int role, const float* w, int Cin_ref, int Cout, int Cin_slab, int dup_start,
                             int dup_shift, bool all_parts) {
  std::lock_guard<std::mutex> lk(g_img_mu);
  ImgState& e = g_img[wp];
  const bool same = e.w == w && e.kind == kind && e.role == role && e.Cin_ref == Cin_ref && e.Cout == Cout && e.Cin_slab == Cin_slab &&
                    e.dup_start == dup_start && e.dup_shift == dup_shift;
  if (!same) { e = ImgState{kind, role, Cin_ref, Cout, Cin_slab, dup_start, dup_shift, w, 0u, 0u, 0u}; }      // (another image at this address: nothing is known)
  if (all_parts) {      // bts_conv_pack writes every part: all fresh until the next batch re-pack over this image
    e.fresh_extra = 7u;
    return 7u;
  }
  unsigned m = 7u;
  if (kind == BTS_CONV_K3S1 && e.used != 0u && img_masks_enabled()) m = e.used;
  return m;
}
extern "C" long bts_conv_pack_generation(void) {
  std::lock_guard<std::mutex> lk(g_img_mu);
  return g_img_gen;
}
extern "C" int bts_conv_pack_forget(const float* wp) {      // the image's memory is being released: drop its registry entry
  std::lock_guard<std::mutex> lk(g_img_mu);
  return g_img.erase(wp) ? 1 : 0;
}
__global__ void pack_kernel(const PackParams q);
// Called right before a kernel reads part `bit` of the K3S1 image at `base`: packs the part on `stream` from the recorded source when the
// last re-pack left it out.  The part counts as fresh only after the pack launch was accepted; an error is the caller's to return.
int bts_img_ensure_(const float* base, unsigned bit, hipStream_t stream) {
  PackParams q;
  {
    std::lock_guard<std::mutex> lk(g_img_mu);
    auto it = g_img.find(base);
    if (it == g_img.end() || it->second.kind != BTS_CONV_K3S1) return BTS_OK;      // (an image this registry never saw: packed whole by its owner)
    const ImgState& e = it->second;
    if ((e.table_mask | e.fresh_extra) & bit) return BTS_OK;
    const int r = pack_params(q, e.kind, e.role, e.w, const_cast<float*>(base), e.Cin_ref, e.Cout, e.Cin_slab, e.dup_start, e.dup_shift, bit);
    if (r != BTS_OK) return r;
  }
  long blocks = (q.items + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError();
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, q);
  BTS_LAUNCH_CHECK();
  std::lock_guard<std::mutex> lk(g_img_mu);
  auto it = g_img.find(base);
  if (it != g_img.end()) it->second.fresh_extra |= bit;
  return BTS_OK;
}
// a launcher has ACCEPTED a launch that reads part `bit`: descriptor tables built from now on describe it (generation moves once per new form)
void bts_img_mark_used_(const float* base, unsigned bit) {
  std::lock_guard<std::mutex> lk(g_img_mu);
  auto it = g_img.find(base);
  if (it == g_img.end() || it->second.kind != BTS_CONV_K3S1) return;
  if (!(it->second.used & bit)) { it->second.used |= bit; ++g_img_gen; }
}
int bts_img_note_use_(const float* base, unsigned bit, hipStream_t stream) {      // both, for launchers that have already accepted
  const int r = bts_img_ensure_(base, bit, stream);
  if (r == BTS_OK) bts_img_mark_used_(base, bit);
  return r;
}

extern "C" int bts_conv_pack(int kind, int role, const float* w, float* wp, int Cin_ref, int Cout, int Cin_slab,
                             int dup_start, int dup_shift, hipStream_t stream) {
  PackParams q;
  const int r = pack_params(q, kind, role, w, wp, Cin_ref, Cout, Cin_slab, dup_start, dup_shift);
  if (r != BTS_OK) return r;
  if (w != nullptr && wp != nullptr) img_register(wp, kind, role, w, Cin_ref, Cout, Cin_slab, dup_start, dup_shift, true);
  const long total = q.items;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError(); hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, stream, q);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

extern "C" long bts_conv_pack_desc_bytes(void) { return (long)sizeof(PackDesc); }

// Fill descriptor #index of a HOST table (bts_conv_pack_desc_bytes() bytes per entry); first_block = sum of the block
// counts returned for the entries before it.  Returns this entry's block count (> 0) or a negative engine code.
extern "C" long bts_conv_pack_desc(void* host_table, int index, long first_block, int kind, int role, const float* w, float* wp,
                                   int Cin_ref, int Cout, int Cin_slab, int dup_start, int dup_shift) {
  PackDesc d;
  int r = pack_params(d.q, kind, role, w, wp, Cin_ref, Cout, Cin_slab, dup_start, dup_shift);
  if (r != BTS_OK) return r;
  const unsigned parts = img_register(wp, kind, role, w, Cin_ref, Cout, Cin_slab, dup_start, dup_shift, false);   // (the parts read so far; all when none yet)
  if (parts != 7u) r = pack_params(d.q, kind, role, w, wp, Cin_ref, Cout, Cin_slab, dup_start, dup_shift, parts);
  if (r != BTS_OK) return r;
  d.total = d.q.items;
  d.first_block = first_block;
  reinterpret_cast<PackDesc*>(host_table)[index] = d;
  return (d.total + PACK_BLOCK_ELEMS - 1) / PACK_BLOCK_ELEMS;
}

// table_dev: the host table copied to device memory by the caller; table_host: that host table (still readable; may be NULL);
// total_blocks = sum of all block counts.  A batch pack follows a weight change: for every image IN THE TABLE the parts this table
// writes are fresh afterwards and every other part is stale (packed on demand at its next use).  Without the host copy the library cannot
// know which images / parts the table covers: every registered image then counts as stale in every part (correct, but each K3S1 launch
// re-packs its part once per weight change).
extern "C" int bts_conv_pack_batch(const void* table_dev, const void* table_host, int n, long total_blocks, hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffL) return BTS_ERR_SHAPE;
  {
    std::lock_guard<std::mutex> lk(g_img_mu);
    if (table_host != nullptr) {
      const PackDesc* t = reinterpret_cast<const PackDesc*>(table_host);
      for (int i = 0; i < n; ++i) {
        auto it = g_img.find(t[i].q.wp);
        if (it == g_img.end()) continue;
        it->second.table_mask = (it->second.kind == BTS_CONV_K3S1) ? (t[i].q.parts & 7u) : 7u;
        it->second.fresh_extra = 0u;
      }
    } else {
      for (auto& kv : g_img) { kv.second.table_mask = 0u; kv.second.fresh_extra = 0u; }
    }
  }
  (void)hipGetLastError(); hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream,
                     reinterpret_cast<const PackDesc*>(table_dev), n);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Transposed-conv gather form with all 8 output-parity classes in ONE workgroup (Conv3DTranspose k3 s2 forward,
// upsample.py:28-33, and the data gradient of the stride-2 conv, downsample.py:28-35).
// out[2c+p] = sum over the taps of parity class p of in[c+d] * W[k]  with per axis  p=0: (k,d) in {(0,0),(2,-1)}, p=1: (1,0).
// The per-class launches of igemm_kernel give the 1-,2- and 4-tap classes 8..32 MFMAs per staged 8-channel slab, far too
// little to cover staging and barriers (51-62 TF).  Here a wave owns 32 coarse voxels and ALL 8 classes: 8 accumulators,
// the 27 taps of a slab need only the 8 input fragments at offsets {-1,0}^3 (8 ds_read_b128 for 108 MFMAs), and the
// tap -> (class, offset) map is compile-time.
// ---------------------------------------------------------------------------------------------
#define UPM_NSLOT 3
static int ilog2(int v);
struct UpmParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int N, Di, Hi, Wi, Cin, ldx;  // coarse input grid
  int Cout, ldy, Npad, KG;
  int lgTX, lgTY, TZ, ntx, nty, ntz;
  int IX, IY, IZ;  // halo tile = coarse tile + 1 on the low side of every axis
  int flags;
  float* part;     // split-K partials [ksplit][fine voxel][Npad] (igemm_reduce_kernel finishes them)
  int ksplit, kg_per;
};
__host__ __device__ constexpr int upm_cls(int t) { return ((t / 9 == 1) ? 4 : 0) | (((t / 3) % 3 == 1) ? 2 : 0) | ((t % 3 == 1) ? 1 : 0); }
// halo coordinate of the input voxel per axis: k=2 reads c-1 (index 0), k=0/1 read c (index 1)
__host__ __device__ constexpr int upm_off(int t) { return ((t / 9 == 2) ? 0 : 4) | (((t / 3) % 3 == 2) ? 0 : 2) | ((t % 3 == 2) ? 0 : 1); }

__global__ __launch_bounds__(256, 2) void upm_kernel(const UpmParams p) {
  constexpr int S = 12;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  int b = blockIdx.x;
  const int tx = b % p.ntx; b /= p.ntx;
  const int ty = b % p.nty; b /= p.nty;
  const int tz = b % p.ntz;
  const int n = b / p.ntz;
  const int TX = 1 << p.lgTX, TY = 1 << p.lgTY;
  const int cz0 = tz * p.TZ, cy0 = ty * TY, cx0 = tx * TX;
  const int iz0 = cz0 - 1, iy0 = cy0 - 1, ix0 = cx0 - 1;
  const int tileVox = p.IZ * p.IY * p.IX;
  const int bufDw = tileVox * S;
  const int nslots = tileVox * 2;
  const float* xbase = p.x + ((((long)n * p.Di + iz0) * p.Hi + iy0) * p.Wi + ix0) * (long)p.ldx;
  int goff[UPM_NSLOT];
  const int IYX = p.IY * p.IX;
  const float invIX = 1.0f / (float)p.IX, invIYX = 1.0f / (float)IYX;
#pragma unroll
  for (int i = 0; i < UPM_NSLOT; ++i) {
    const int e = tid + i * 256;
    goff[i] = -1;
    if (e < nslots) {
      const int vox = e >> 1, q = e & 1;
      const int vz = (int)(((float)vox + 0.5f) * invIYX);
      const int r = vox - vz * IYX;
      const int vy = (int)(((float)r + 0.5f) * invIX);
      const int vx = r - vy * p.IX;
      if ((unsigned)(iz0 + vz) < (unsigned)p.Di && (unsigned)(iy0 + vy) < (unsigned)p.Hi && (unsigned)(ix0 + vx) < (unsigned)p.Wi)
        goff[i] = ((vz * p.Hi + vy) * p.Wi + vx) * p.ldx + q * 4;
    }
  }
  const int m = wave * 32 + l32;
  const int mx = m & (TX - 1), my = (m >> p.lgTX) & (TY - 1), mz = m >> (p.lgTX + p.lgTY);
  const int bbase = ((mz * p.IY + my) * p.IX + mx) * S + h * 4;
  int off8[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) off8[o] = ((((o >> 2) & 1) * p.IY + ((o >> 1) & 1)) * p.IX + (o & 1)) * S;
  int ncol = blockIdx.y * 32 + l32;
  if (ncol >= p.Npad) ncol = p.Npad - 1;
  const int lane_woff = (h * p.Npad + ncol) * 4;
  const int wstepKG = 2 * p.Npad * 4, wstepTap = p.KG * wstepKG;

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  f32x4 pre[UPM_NSLOT];
  auto fetch = [&](int st) {
#pragma unroll
    for (int i = 0; i < UPM_NSLOT; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (goff[i] >= 0) v = *reinterpret_cast<const f32x4*>(xbase + goff[i] + st * 8);
      pre[i] = v;
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < UPM_NSLOT; ++i) {
      const int e = tid + i * 256;
      if (e < nslots) *reinterpret_cast<f32x4*>(buf + (e >> 1) * S + (e & 1) * 4) = pre[i];
    }
  };
  int kgBeg = 0, kgEnd = p.KG;
  if (p.ksplit > 1) {
    kgBeg = blockIdx.z * p.kg_per;
    kgEnd = kgBeg + p.kg_per;
    if (kgEnd > p.KG) kgEnd = p.KG;
  }
  fetch(kgBeg);
  commit(lds);
  __syncthreads();
  for (int st = kgBeg; st < kgEnd; ++st) {
    const float* cur = lds + ((st - kgBeg) & 1) * bufDw;
    const bool more = (st + 1) < kgEnd;
    if (more) fetch(st + 1);
    const float* wk = p.wp + st * wstepKG;  // wave-uniform
    constexpr int AD = 2;
    f32x4 a[AD + 1], bf[8];
#pragma unroll
    for (int q = 0; q < AD; ++q) a[q] = *reinterpret_cast<const f32x4*>((wk + q * wstepTap) + lane_woff);
#pragma unroll
    for (int o = 0; o < 8; ++o) bf[o] = *reinterpret_cast<const f32x4*>(cur + bbase + off8[o]);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      if (t + AD < 27) a[(t + AD) % (AD + 1)] = *reinterpret_cast<const f32x4*>((wk + (t + AD) * wstepTap) + lane_woff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[upm_cls(t)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t % (AD + 1)][j], bf[upm_off(t)][j], acc[upm_cls(t)], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) commit(lds + ((st + 1 - kgBeg) & 1) * bufDw);
    __syncthreads();
  }

  // ---- epilogue: class c = (pz,py,px) writes fine voxel 2*coarse + parity ----
  const int cz = cz0 + mz, cy = cy0 + my, cx = cx0 + mx;
  if (cz >= p.Di || cy >= p.Hi || cx >= p.Wi) return;
  const int nb = blockIdx.y * 32;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int oz = 2 * cz + ((c >> 2) & 1), oy = 2 * cy + ((c >> 1) & 1), ox = 2 * cx + (c & 1);
    const long vfine = (((long)n * (2 * p.Di) + oz) * (2 * p.Hi) + oy) * (2 * p.Wi) + ox;
    if (p.ksplit > 1) {  // raw partial, finished (bias / activation / accumulate) by igemm_reduce_kernel in a fixed order
      float* row = p.part + ((long)blockIdx.z * ((long)p.N * 8 * p.Di * p.Hi * p.Wi) + vfine) * p.Npad;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v4 = {acc[c][4 * g + 0], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3]};
        *reinterpret_cast<f32x4*>(row + nb + 8 * g + 4 * h) = v4;
      }
      continue;
    }
    float* yrow = p.y + vfine * (long)p.ldy;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = nb + 8 * g + 4 * h;
      if (co >= p.Cout) continue;
      f32x4 v = {acc[c][4 * g + 0], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3]};
      if (p.flags & IG_FLAG_BIAS) {   // element loads: the bias is a view into the flat parameter buffer, 4-byte aligned only
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += p.bias[co + j];
      }
      if (p.flags & IG_FLAG_SIGMOID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sigmoidf_(v[j]);
      }
      f32x4* dst = reinterpret_cast<f32x4*>(yrow + co);
      if (p.flags & IG_FLAG_ACCUM) v += *dst;
      *dst = v;
    }
  }
}

// Shape-only part of the decision (shared by the workspace query): tile geometry and split-K factor.
// returns 0 when the merged-class kernel does not fit the shape.
static int plan_upm(UpmParams& p, int N, int Di, int Hi, int Wi, int Cin, int Cout, long* need) {
  *need = 0;
  if ((Cin & 7) || (Cout & 3) || Wi < 8 || Hi < 4 || Di < 2) return 0;
  p.N = N; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout;
  p.Npad = npad32(Cout); p.KG = Cin / 8;
  int TX = 32;
  while (TX > 8 && TX / 2 >= Wi) TX /= 2;
  int TY = (TX == 32) ? 2 : 4;
  int TZ = 128 / (TX * TY);
  p.lgTX = ilog2(TX); p.lgTY = ilog2(TY); p.TZ = TZ;
  p.ntx = (Wi + TX - 1) / TX; p.nty = (Hi + TY - 1) / TY; p.ntz = (Di + TZ - 1) / TZ;
  p.IX = TX + 1; p.IY = TY + 1; p.IZ = TZ + 1;
  if (p.IZ * p.IY * p.IX * 2 > 256 * UPM_NSLOT) return 0;
  const long wgs = (long)N * p.ntz * p.nty * p.ntx * (p.Npad / 32);
  const long min_wgs = getenv("BTS_IGEMM_UPM_MIN") ? atol(getenv("BTS_IGEMM_UPM_MIN")) : 256;  // (tests force 1)
  p.ksplit = 1; p.kg_per = p.KG;
  if (wgs < min_wgs) {
    // too few workgroups to fill the chip: split the contraction (deterministic two-stage reduction) when it is long
    if (p.KG < 8 || getenv("BTS_IGEMM_UPM_NOSPLIT") != nullptr) return 0;
    int ks = (int)((512 + wgs - 1) / wgs);
    if (ks > p.KG / 4) ks = p.KG / 4;
    if (ks > 16) ks = 16;
    if (ks < 2) return 0;
    p.kg_per = (p.KG + ks - 1) / ks;
    p.ksplit = (p.KG + p.kg_per - 1) / p.kg_per;
    if (p.ksplit < 2 || wgs * p.ksplit < min_wgs / 2) return 0;
    *need = (long)p.ksplit * N * 8 * Di * Hi * Wi * p.Npad * 4;
  }
  return 1;
}

// returns BTS_OK when the merged-class kernel took the launch, 1 when the shape is left to the per-class path
static int launch_upm(const float* x, const float* wp, const float* bias, float* y, int N, int Di, int Hi, int Wi, int Cin,
                      int ldx, int Cout, int ldy, int flags, void* ws, long ws_bytes, hipStream_t stream) {
  if (getenv("BTS_IGEMM_NOUPM") != nullptr) return 1;
  if ((ldx & 3) || (ldy & 3) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 15)) return 1;
  UpmParams p;
  long need = 0;
  if (!plan_upm(p, N, Di, Hi, Wi, Cin, Cout, &need)) return 1;
  if (need > 0 && (ws == nullptr || ws_bytes < need)) return 1;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.ldx = ldx; p.ldy = ldy; p.flags = flags;
  p.part = reinterpret_cast<float*>(ws);
  const long tiles = (long)N * p.ntz * p.nty * p.ntx;
  const int ny = p.Npad / 32;
  const int tileVox = p.IZ * p.IY * p.IX;
  const size_t shmem = (size_t)2 * tileVox * 12 * sizeof(float);
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(20, 2.0 * 27.0 * Cin * Cout * (double)N * Di * Hi * Wi, stream);
  (void)hipGetLastError(); hipLaunchKernelGGL(upm_kernel, dim3((unsigned)tiles, ny, p.ksplit), dim3(256), shmem, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  if (p.ksplit > 1) {
    const long nvox = (long)N * 8 * Di * Hi * Wi;
    long blocks = (nvox * Cout + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    (void)hipGetLastError(); hipLaunchKernelGGL(igemm_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, p.part, bias, y, nvox, Cout, p.Npad, ldy,
                       p.ksplit, flags);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// 1x1x1 convolution as a streaming GEMM (resnet.py:96-103 shortcut, decoder.py:55-63 output head, vae.py pointwise convs,
// and their data gradients).  These layers are HBM-bound (<= 64 FLOP/B), so nothing is staged: the B fragment of the
// MFMA (voxel l32, channels kg*8 + 4h .. +3) IS a 16-byte global load of that voxel's row, the A fragment the matching
// quad of the packed weights (L1/L2 resident).  A wave owns 64 voxels x 32*NS couts; latency is covered by occupancy
// (~100 VGPRs -> 4 waves per SIMD), not by a software pipeline.
// ---------------------------------------------------------------------------------------------
struct K1sParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  long nvox;
  int Cin, ldx, Cout, ldy, Npad, KG, flags;
};
template <int NS>
__global__ __launch_bounds__(256, 4) void k1s_kernel(const K1sParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  const long v0 = (long)blockIdx.x * 256 + wave * 64;
  const float* xrow[2];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) {
    long v = v0 + ms * 32 + l32;
    if (v >= p.nvox) v = p.nvox - 1;  // clamped loads, masked stores
    xrow[ms] = p.x + v * p.ldx + h * 4;
  }
  int woff[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    int ncol = (blockIdx.y * NS + ns) * 32 + l32;
    if (ncol >= p.Npad) ncol = p.Npad - 1;
    woff[ns] = (h * p.Npad + ncol) * 4;
  }
  const int wstepKG = 2 * p.Npad * 4;
  f32x16 acc[2][NS];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;
  // U k-groups per step (fewer for the wider tile: 128-VGPR budget), next step's operands requested before this
  // step's MFMAs
  constexpr int U = (NS == 1) ? 2 : 1;
  f32x4 b[2][U][2], a[2][U][NS];
  auto load = [&](int buf, int kg) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (kg + u < p.KG) ? kg + u : p.KG - 1;
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) b[buf][u][ms] = *reinterpret_cast<const f32x4*>(xrow[ms] + k * 8);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) a[buf][u][ns] = *reinterpret_cast<const f32x4*>((p.wp + k * wstepKG) + woff[ns]);
    }
  };
  auto mma = [&](int buf, int kg) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kg + u < p.KG) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
              acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][u][ns][j], b[buf][u][ms][j], acc[ms][ns], 0, 0, 0);
      }
    }
  };
  load(0, 0);
  for (int kg = 0; kg < p.KG; kg += 2 * U) {
    if (kg + U < p.KG) load(1, kg + U);
    mma(0, kg);
    if (kg + U < p.KG) {
      if (kg + 2 * U < p.KG) load(0, kg + 2 * U);
      mma(1, kg + U);
    }
  }
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) {
    const long v = v0 + ms * 32 + l32;
    if (v >= p.nvox) continue;
    float* yrow = p.y + v * p.ldy;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int nb = (blockIdx.y * NS + ns) * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = nb + 8 * g + 4 * h;
        if (co >= p.Cout) continue;
        f32x4 val = {acc[ms][ns][4 * g + 0], acc[ms][ns][4 * g + 1], acc[ms][ns][4 * g + 2], acc[ms][ns][4 * g + 3]};
        if (p.flags & IG_FLAG_BIAS) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Cout) val[j] += p.bias[co + j];
        }
        if (p.flags & IG_FLAG_SIGMOID) {
#pragma unroll
          for (int j = 0; j < 4; ++j) val[j] = sigmoidf_(val[j]);
        }
        if ((p.flags & IG_FLAG_VECOUT) && co + 3 < p.Cout) {
          f32x4* dst = reinterpret_cast<f32x4*>(yrow + co);
          if (p.flags & IG_FLAG_ACCUM) val += *dst;
          *dst = val;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Cout) yrow[co + j] = (p.flags & IG_FLAG_ACCUM) ? yrow[co + j] + val[j] : val[j];
        }
      }
    }
  }
}

// returns BTS_OK when taken, 1 when the shape is left to the staged igemm path
static int launch_k1s(const float* x, const float* wp, const float* bias, float* y, long nvox, int Cin, int ldx, int Cout, int ldy,
                      int flags, hipStream_t stream) {
  if (getenv("BTS_IGEMM_NOK1S") != nullptr) return 1;
  if ((Cin & 7) || (ldx & 3) || (((uintptr_t)x) & 15)) return 1;  // whole k-groups, 16-byte row quads
  const long min_vox = getenv("BTS_IGEMM_K1S_MIN") ? atol(getenv("BTS_IGEMM_K1S_MIN")) : 256 * 256;  // (tests force 1)
  if (nvox < min_vox) return 1;  // small grids (deep levels): too few workgroups, keep the split-K capable path
  K1sParams p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.nvox = nvox;
  p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy; p.Npad = npad32(Cout); p.KG = Cin / 8;
  p.flags = flags & (IG_FLAG_BIAS | IG_FLAG_ACCUM | IG_FLAG_SIGMOID);
  if ((ldy % 4 == 0) && (((uintptr_t)y) % 16 == 0)) p.flags |= IG_FLAG_VECOUT;
  const long gx = (nvox + 255) / 256;
  if (gx > 0x7fffffffL) return 1;
  const bool prof = bts_prof_on();
  const int ns = (p.Npad >= 64) ? 2 : 1;
  if (prof) bts_prof_begin(21, 2.0 * Cin * (double)Cout * (double)nvox, stream);
  (void)hipGetLastError();
  if (ns == 2) hipLaunchKernelGGL(k1s_kernel<2>, dim3((unsigned)gx, (p.Npad + 63) / 64), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(k1s_kernel<1>, dim3((unsigned)gx, 1), dim3(256), 0, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// 3x3x3 stride-1 conv with 1..4 output channels (VAE reconstruction head 32 -> 2, vae.py:92-99 / model.py:72) on the
// vector ALU: padding 2 couts to an MFMA N of 32 wastes 94 % of the matrix pipe (0.99 ms at 128^3).  One output voxel
// per lane; the halo tile is staged 8 channels at a time in the igemm layout (conflict-free ds_read_b128), the
// wave-uniform weights come straight out of the packed image through scalar loads (v_fma with an SGPR operand), so the
// LDS serves only the 2 x 27 activation quads per voxel and slab.  Latency is covered by occupancy (4 workgroups / CU).
// ---------------------------------------------------------------------------------------------
struct DscParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int N, D, H, W, Cin, ldx, Cout, ldy, Npad, KG, ntx, nty, ntz, flags;
};
template <int CO>
__global__ __launch_bounds__(256, 4) void dsc_kernel(const DscParams p) {
  constexpr int S = 12, TX = 32, TY = 4, TZ = 2, IX = TX + 2, IY = TY + 2, IZ = TZ + 2;
  constexpr int tileVox = IZ * IY * IX, nslots = tileVox * 2, NSL = (nslots + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[tileVox * S];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx = b % p.ntx; b /= p.ntx;
  const int ty = b % p.nty; b /= p.nty;
  const int tz = b % p.ntz;
  const int n = b / p.ntz;
  const int iz0 = tz * TZ - 1, iy0 = ty * TY - 1, ix0 = tx * TX - 1;
  const float* xbase = p.x + ((((long)n * p.D + iz0) * p.H + iy0) * p.W + ix0) * (long)p.ldx;
  int goff[NSL];
#pragma unroll
  for (int i = 0; i < NSL; ++i) {
    const int e = tid + i * 256;
    goff[i] = -1;
    if (e < nslots) {
      const int vox = e >> 1, q = e & 1;
      const int vz = vox / (IY * IX), r = vox - vz * (IY * IX), vy = r / IX, vx = r - vy * IX;
      if ((unsigned)(iz0 + vz) < (unsigned)p.D && (unsigned)(iy0 + vy) < (unsigned)p.H && (unsigned)(ix0 + vx) < (unsigned)p.W)
        goff[i] = ((vz * p.H + vy) * p.W + vx) * p.ldx + q * 4;
    }
  }
  const int mx = tid & (TX - 1), my = (tid >> 5) & (TY - 1), mz = tid >> 7;
  const float* lb = lds + ((mz * IY + my) * IX + mx) * S;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const int wstepKG = 2 * p.Npad * 4, wstepTap = p.KG * wstepKG;
  f32x4 pre[NSL];
  auto fetch = [&](int kg) {
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (goff[i] >= 0) v = *reinterpret_cast<const f32x4*>(xbase + goff[i] + kg * 8);
      pre[i] = v;
    }
  };
  fetch(0);
  for (int kg = 0; kg < p.KG; ++kg) {
    __syncthreads();  // previous slab fully consumed
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int e = tid + i * 256;
      if (e < nslots) *reinterpret_cast<f32x4*>(lds + (e >> 1) * S + (e & 1) * 4) = pre[i];
    }
    __syncthreads();
    if (kg + 1 < p.KG) fetch(kg + 1);  // next slab's global loads fly under this slab's arithmetic
    const float* wk = p.wp + kg * wstepKG;  // wave-uniform: scalar loads
    // one x-row of taps (3 taps x 2 halves x CO quads) per scalar-load batch: SMEM returns out of order, so every use
    // of a scalar load drains lgkmcnt to 0 -- batching makes that 9 drains per slab instead of 54
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      float w[3][2][CO * 4];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int q = 0; q < CO * 4; ++q) w[dx][hh][q] = wk[(r * 3 + dx) * wstepTap + hh * p.Npad * 4 + q];
      f32x4 xv[3][2];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float* src = lb + (((r / 3) * IY + r % 3) * IX + dx) * S;
        xv[dx][0] = *reinterpret_cast<const f32x4*>(src);
        xv[dx][1] = *reinterpret_cast<const f32x4*>(src + 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int c = 0; c < CO; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c] = fmaf(xv[dx][hh][j], w[dx][hh][c * 4 + j], acc[c]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int oz = tz * TZ + mz, oy = ty * TY + my, ox = tx * TX + mx;
  if (oz >= p.D || oy >= p.H || ox >= p.W) return;
  float* yrow = p.y + ((((long)n * p.D + oz) * p.H + oy) * p.W + ox) * (long)p.ldy;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    if (c < p.Cout) {
      float v = acc[c];
      if (p.flags & IG_FLAG_BIAS) v += p.bias[c];
      if (p.flags & IG_FLAG_SIGMOID) v = sigmoidf_(v);
      if (p.flags & IG_FLAG_ACCUM) v += yrow[c];
      yrow[c] = v;
    }
  }
}

// returns BTS_OK when taken, 1 when the shape is left to the MFMA path
static int launch_dsc(const float* x, const float* wp, const float* bias, float* y, int N, int D, int H, int W, int Cin, int ldx,
                      int Cout, int ldy, int flags, hipStream_t stream) {
  if (getenv("BTS_IGEMM_NODSC") != nullptr) return 1;
  if (Cout > 4 || (Cin & 7) || (ldx & 3) || (((uintptr_t)x) & 15)) return 1;
  const long tiles = (long)N * ((D + 1) / 2) * ((H + 3) / 4) * ((W + 31) / 32);
  const long min_tiles = getenv("BTS_IGEMM_DSC_MIN") ? atol(getenv("BTS_IGEMM_DSC_MIN")) : 1024;  // (tests force 1)
  if (tiles < min_tiles || tiles > 0x7fffffffL) return 1;
  DscParams p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy;
  p.Npad = npad32(Cout); p.KG = Cin / 8;
  p.ntx = (W + 31) / 32; p.nty = (H + 3) / 4; p.ntz = (D + 1) / 2;
  p.flags = flags & (IG_FLAG_BIAS | IG_FLAG_ACCUM | IG_FLAG_SIGMOID);
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(22, 2.0 * 27.0 * Cin * (double)Cout * (double)N * D * H * W, stream);
  (void)hipGetLastError();
  if (Cout <= 2) hipLaunchKernelGGL(dsc_kernel<2>, dim3((unsigned)tiles), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(dsc_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// 3x3x3 stride-1 conv of a TWO-channel input (the network's first convolutions, model.py / resnet.py:30-37 on the 2ch x 128^3
// volume) into <= 32 output channels.  The tiled kernel pads the 2 channels to a k-group of 8 and runs at a quarter of the
// useful rate; here the contraction index of the MFMA (K = 2) IS the channel pair: one matrix instruction per tap, its B
// operand a single ds_read_b32 of the planar LDS halo tile [channel][z][y][x] (lane = (channel, x)), its A operand the
// tap's 2 x 32 weights held in a register for the whole workgroup.  A wave owns one z plane of the 32 x 4 x 4 tile: the
// 18 input fragments of a kz slab feed the 36 matrix instructions of its four output rows.  Optional second output
// (the 1x1x1 shortcut conv of the same input: one more instruction on the centre fragment), bias, GroupNorm partials.
// ---------------------------------------------------------------------------------------------
struct C2Params {
  const float* x;
  const float* wp;    // K3S1 forward image (KG = 1): W[t][c][n] = wp[(t*2*Npad + n)*4 + c]
  const float* bias;
  float* y;
  const float* wp2;   // K1 forward image of the shortcut (or null): W2[c][n] = wp2[n*4 + c]
  const float* bias2;
  float* y2;
  int N, D, H, W, ldx, Cout, ldy, ldy2, Npad;
  int ntz, nty, ntx;
  double* gnp;
  int gn_G, gn_zt;
};

template <bool F2>
__global__ __launch_bounds__(256, 2) void c2_kernel(const C2Params p) {
  __shared__ float lds[2 * 6 * 6 * 34 + 16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  int b = blockIdx.x;
  const int tx = b % p.ntx; b /= p.ntx;
  const int ty = b % p.nty; b /= p.nty;
  const int tz = b % p.ntz;
  const int n = b / p.ntz;
  const int oz0 = tz * 4, oy0 = ty * 4, ox0 = tx * 32;
  // ---- halo tile (34 x 6 x 6 voxels x 2 channels) -> planar LDS ----
  for (int e = tid; e < 6 * 6 * 34; e += 256) {
    const int vz = e / (6 * 34), r = e - vz * (6 * 34);
    const int vy = r / 34, vx = r - vy * 34;
    const int z = oz0 - 1 + vz, yy = oy0 - 1 + vy, xx = ox0 - 1 + vx;
    float v0 = 0.f, v1 = 0.f;
    if ((unsigned)z < (unsigned)p.D && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
      const float* s = p.x + ((((long)n * p.D + z) * p.H + yy) * p.W + xx) * (long)p.ldx;
      v0 = s[0]; v1 = s[1];
    }
    lds[e] = v0;
    lds[6 * 6 * 34 + e] = v1;
  }
  // ---- weights: lane (channel h, cout l32) ----
  float wt[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) wt[t] = p.wp[((long)t * 2 * p.Npad + l32) * 4 + h];
  float w2 = 0.f;
  if (F2) w2 = p.wp2[l32 * 4 + h];
  __syncthreads();
  f32x16 acc[4], acc2[F2 ? 4 : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; if (F2) acc2[F2 ? i : 0][r] = 0.f; }
  const float* lb = lds + h * (6 * 6 * 34) + (wave * 6) * 34 + l32;   // (channel h, z = wave + kz, y, x = l32 + kx)
#pragma unroll
  for (int kz = 0; kz < 3; ++kz) {
    float f[6][3];
#pragma unroll
    for (int yr = 0; yr < 6; ++yr)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) f[yr][kx] = lb[(kz * 6 + yr) * 34 + kx];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int oy = 0; oy < 4; ++oy)
          acc[oy] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[(kz * 3 + ky) * 3 + kx], f[oy + ky][kx], acc[oy], 0, 0, 0);
    if (F2 && kz == 1) {
#pragma unroll
      for (int oy = 0; oy < 4; ++oy) acc2[F2 ? oy : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2, f[oy + 1][1], acc2[F2 ? oy : 0], 0, 0, 0);
    }
  }
  // ---- epilogue: D rows = couts (4 per register quad), cols = x ----
  const int oz = oz0 + wave, ox = ox0 + l32;
  float gn_s = 0.f, gn_q = 0.f;
  if (oz < p.D && ox < p.W) {
#pragma unroll
    for (int oy = 0; oy < 4; ++oy) {
      const int y = oy0 + oy;
      if (y >= p.H) continue;
      const long pix = (((long)n * p.D + oz) * p.H + y) * p.W + ox;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = 8 * g + 4 * h;
        if (co >= p.Cout) continue;
        f32x4 v = {acc[oy][4 * g], acc[oy][4 * g + 1], acc[oy][4 * g + 2], acc[oy][4 * g + 3]};
        if (p.bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += p.bias[co + j];
        }
        if (p.gnp) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { gn_s += v[j]; gn_q = fmaf(v[j], v[j], gn_q); }
        }
        *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + co) = v;
        if (F2) {
          f32x4 v2 = {acc2[F2 ? oy : 0][4 * g], acc2[F2 ? oy : 0][4 * g + 1], acc2[F2 ? oy : 0][4 * g + 2], acc2[F2 ? oy : 0][4 * g + 3]};
          if (p.bias2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v2[j] += p.bias2[co + j];
          }
          *reinterpret_cast<f32x4*>(p.y2 + pix * p.ldy2 + co) = v2;
        }
      }
    }
  }
  if (p.gnp) {  // fixed-order combine, layout as igemm_kernel's
    const double ds = wave_sum_f64((double)gn_s), dq = wave_sum_f64((double)gn_q);
    __syncthreads();
    double* sh = reinterpret_cast<double*>(lds);
    if (lane == 0) { sh[wave * 2] = ds; sh[wave * 2 + 1] = dq; }
    __syncthreads();
    if (tid == 0) {
      const int g = tz / p.gn_zt;
      const long B = (long)p.gn_zt * p.nty * p.ntx;
      const long b_ = ((long)(tz - g * p.gn_zt) * p.nty + ty) * p.ntx + tx;
      double* o = p.gnp + (((long)n * p.gn_G + g) * B + b_) * 2;
      o[0] = sh[0] + sh[2] + sh[4] + sh[6];
      o[1] = sh[1] + sh[3] + sh[5] + sh[7];
    }
  }
}

// returns BTS_OK when taken, 1 when the shape is left to the tiled kernel
static int launch_c2(const float* x, const float* wp, const float* bias, float* y, int N, int D, int H, int W, int Cin, int ldx,
                     int Cout, int ldy, int flags, const float* wp2, const float* bias2, float* y2, int ldy2, double* gnp, int gnG,
                     long* gn_B, hipStream_t stream) {
  if (getenv("BTS_IGEMM_NOC2") != nullptr) return 1;
  if (Cin != 2 || Cout > 32 || Cout % 4 != 0 || ldy % 4 != 0 || (((uintptr_t)y) & 15)) return 1;
  if (flags & (IG_FLAG_ACCUM | IG_FLAG_SIGMOID)) return 1;
  if (y2 != nullptr && (ldy2 % 4 != 0 || (((uintptr_t)y2) & 15))) return 1;
  C2Params p;
  p.x = x; p.wp = wp; p.bias = (flags & IG_FLAG_BIAS) ? bias : nullptr; p.y = y;
  p.wp2 = wp2; p.bias2 = bias2; p.y2 = y2;
  p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy; p.ldy2 = ldy2; p.Npad = npad32(Cout);
  p.ntz = (D + 3) / 4; p.nty = (H + 3) / 4; p.ntx = (W + 31) / 32;
  const long tiles = (long)N * p.ntz * p.nty * p.ntx;
  const long min_tiles = getenv("BTS_IGEMM_C2_MIN") ? atol(getenv("BTS_IGEMM_C2_MIN")) : 512;  // (tests force 1)
  if (tiles < min_tiles || tiles > 0x7fffffffL) return 1;
  p.gnp = nullptr; p.gn_G = 0; p.gn_zt = 1;
  if (gnp != nullptr && gnG > 0 && D % gnG == 0 && (D / gnG) % 4 == 0 && getenv("BTS_IGEMM_NOGNFUSE") == nullptr) {
    p.gnp = gnp; p.gn_G = gnG; p.gn_zt = (D / gnG) / 4;
  }
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(25, 2.0 * (27.0 + (y2 ? 1.0 : 0.0)) * Cin * (double)Cout * (double)N * D * H * W, stream);
  (void)hipGetLastError();
  if (y2 != nullptr) hipLaunchKernelGGL(c2_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(c2_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  if (gn_B && p.gnp != nullptr) *gn_B = (long)p.gn_zt * p.nty * p.ntx;
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Launch logic
// ---------------------------------------------------------------------------------------------
enum Geo { GEO_K1 = 0, GEO_S1 = 1, GEO_DOWN = 2, GEO_UP = 3 };

template <int MS, int NS, int WM, int WN, int KGS, bool FUSE2 = false, bool FIXG = false, bool T27 = false>
static int launch_cfg(IgemmParams& p, hipStream_t stream) {
  constexpr int S = KGS * 8 + 4;
  const int tileVox = p.IZ * p.IY * p.IX;
  if (tileVox * KGS * 2 > 256 * ((MS == 4) ? 10 : MAXSLOT)) return BTS_ERR_SHAPE;
  // a single stage (all channels fit one staging pass, e.g. the 1x1x1 convs with Cin <= 32) needs no second buffer:
  // half the LDS -> twice the resident workgroups to hide the (then un-overlapped) staging latency
  const int nstages_all = (p.KG + KGS - 1) / KGS;
  const size_t shmem = (size_t)((nstages_all > 1 && MS != 4 && !IG_TRI(MS, NS, KGS, FUSE2, FIXG)) ? 2 : 1) * tileVox * S * sizeof(float);
  auto kern = igemm_kernel<MS, NS, WM, WN, KGS, FUSE2, FIXG, T27>;
  static bool attr_done = false;
  if (!attr_done && !p.plan_only) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int NT = 32 * NS * WN;
  dim3 grid(p.N * p.ntz * p.nty * p.ntx, (p.Npad + NT - 1) / NT, p.ncls > 1 ? p.ncls : 1);
  // split-K when the tile grid cannot fill the chip and the contraction is long (tiny spatial grids, wide channels)
  p.ksplit = 1;
  p.kg_per = p.KG;
  p.ws_need = 0;
  const long wgs = (long)grid.x * grid.y;
  // two workgroups fit a CU: below 512 tiles the chip is not full and a single wave per SIMD cannot keep the matrix
  // pipe busy, so the contraction is split (measured: 16^3 256->256 0.157 -> 0.137 ms, 16^3 768->256 0.46 -> 0.38 ms);
  // the 1x1x1 staging variant has too little work per k-group for that and keeps the old, lower threshold
  const long sk_below = (KGS == 1) ? 512 : 192, sk_target = (KGS == 1) ? 512 : 384;
  if (!FUSE2 && p.ncls <= 1 && wgs < sk_below && p.KG >= 4 * KGS) {
    int ks = (int)((sk_target + wgs - 1) / wgs);
    const int maxks = p.KG / (2 * KGS);
    if (ks > maxks) ks = maxks;
    if (ks > 64) ks = 64;
    if (ks > 1) {
      int per = (p.KG + ks - 1) / ks;
      per = (per + KGS - 1) / KGS * KGS;
      ks = (p.KG + per - 1) / per;
      const long need = (long)ks * p.N * p.Do * p.Ho * p.Wo * p.Npad * 4;
      if (ks > 1 && (p.plan_only || (p.part != nullptr && p.ws_bytes >= need))) {
        p.ksplit = ks;
        p.kg_per = per;
        p.ws_need = need;
        grid.z = ks;
      }
    }
  }
  if (p.ksplit > 1) p.gnp = nullptr;  // split-K tiles are finished by the reduce kernel: no fused statistics
  p.gn_gridy = (int)grid.y;
  if (p.plan_only) return BTS_OK;
  const bool prof = bts_prof_on();
  if (prof) {
    constexpr int cfgid = (MS == 4) ? 5 : (MS == 2 && NS == 1) ? 0 : (MS == 2 && NS == 2) ? 1 : (MS == 1 && NS == 2) ? 2 : (WN == 2) ? 3 : 4;
    const double taps = (p.ncls > 1) ? 27.0 : (double)p.ntaps;
    bts_prof_begin(cfgid + (KGS == 4 ? 8 : 0), 2.0 * taps * p.Cin * p.Cout * (double)p.N * p.Do * p.Ho * p.Wo, stream);
  }
  (void)hipGetLastError(); hipLaunchKernelGGL(kern, grid, dim3(256), shmem, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  if (p.ksplit > 1) {
    const long nvox = (long)p.N * p.Do * p.Ho * p.Wo;
    long blocks = (nvox * p.Cout + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    (void)hipGetLastError(); hipLaunchKernelGGL(igemm_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, p.part, p.bias, p.y, nvox, p.Cout,
                       p.Npad, p.ldy, p.ksplit, p.flags);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// cfg ids: 0:(2,1,4,1) M256 N32 | 1:(2,2,4,1) M256 N64 | 2:(1,2,2,2) M64 N128 | 3:(1,1,2,2) M64 N64 | 4:(1,1,4,1) M128 N32
static int choose_cfg(int geo, int N, int Do, int Ho, int Wo, int Npad, int* Mout) {
  const long vox = (long)Do * Ho * Wo;
  int M, cfg;
  if (geo == GEO_DOWN) {
    M = 64;
    cfg = (Npad >= 128) ? 2 : 3;
  } else {
    const long wg256 = (long)N * ((vox + 255) / 256) * ((Npad + 63) / 64);
    if (Npad <= 32) {
      if (geo == GEO_S1 && Wo >= 32 && Ho >= 4 && Do >= 4 && (long)N * ((vox + 511) / 512) >= 1024 &&
          getenv("BTS_IGEMM_M512") != nullptr) { M = 512; cfg = 5; }  // measured equal to M=256 on MI355X: opt-in only
      else if ((long)N * ((vox + 255) / 256) >= 512) { M = 256; cfg = 0; }
      else { M = 128; cfg = 4; }
    } else if (wg256 >= 512) { M = 256; cfg = 1; }
    else {
      M = 64;
      const long wg64_128 = (long)N * ((vox + 63) / 64) * ((Npad + 127) / 128);
      cfg = (Npad >= 128 && wg64_128 >= 384) ? 2 : 3;
    }
  }
  *Mout = M;
  return cfg;
}

// geometry + config selection for one gather-conv launch
static int launch_igemm(int geo, const float* x, const float* wp, const float* bias, float* y, int N, int Di, int Hi,
                        int Wi, int Cin, int ldx, int Do, int Ho, int Wo, int Cout, int ldy, int ODa, int OHa, int OWa,
                        int pz, int py, int px, int flags, hipStream_t stream, void* ws = nullptr, long ws_bytes = 0,
                        long* need_out = nullptr, const float* wp2 = nullptr, const float* bias2 = nullptr,
                        float* y2 = nullptr, int ldy2 = 0, double* gnp = nullptr, int gnG = 0, long* gn_B = nullptr,
                        const float* x2 = nullptr, int ldx2 = 0) {
  if (gn_B) *gn_B = 0;  // stays 0 unless the tiled kernel took the launch and emitted the GroupNorm partials
  if (geo == GEO_S1 && wp2 == nullptr && need_out == nullptr && Cout <= 4) {
    { const int e = bts_img_ensure_(wp, 1u, stream); if (e != BTS_OK) return e; }      // (a declined launch costs at most one early pack)
    const int r = launch_dsc(x, wp, bias, y, N, Di, Hi, Wi, Cin, ldx, Cout, ldy, flags, stream);
    if (r != 1) { bts_img_mark_used_(wp, 1u); return r; }
  }
  if (geo == GEO_S1 && need_out == nullptr && Cin == 2 && x2 == nullptr && (wp2 == nullptr) == (y2 == nullptr)) {
    { const int e = bts_img_ensure_(wp, 1u, stream); if (e != BTS_OK) return e; }
    const int r = launch_c2(x, wp, bias, y, N, Di, Hi, Wi, Cin, ldx, Cout, ldy, flags, wp2, bias2, y2, ldy2, gnp, gnG, gn_B, stream);
    if (r != 1) { bts_img_mark_used_(wp, 1u); return r; }
  }
  if (geo == GEO_S1 && need_out == nullptr && !(flags & IG_FLAG_SIGMOID)) {
    // Winograd form (conv_wino.hip) on the second part of the K3S1 image; the fused shortcut output / second input of the
    // pair entry points then run as their own 1x1x1 launches
    const long pairs = (long)((Cin + 7) / 8) * 2 * npad32(Cout) * 4;
    const float* up = wp + 27L * pairs;
    int r = bts_w3_launch_(x, up + 48L * pairs, (flags & IG_FLAG_BIAS) ? bias : nullptr, y, N, Di, Hi, Wi, Cin, ldx, Cout, ldy,
                           (flags & IG_FLAG_ACCUM) ? 1 : 0, gnp, gnG, gn_B, ws, ws_bytes, stream);
    if (r == 1)
      r = bts_wino_launch_(x, up, (flags & IG_FLAG_BIAS) ? bias : nullptr, y, N, Di, Hi, Wi, Cin, ldx, Cout, ldy,
                           (flags & IG_FLAG_ACCUM) ? 1 : 0, gnp, gnG, gn_B, ws, ws_bytes, stream);
    if (r == BTS_OK) {
      if (y2 != nullptr)
        return launch_igemm(GEO_K1, x, wp2, bias2, y2, N, Di, Hi, Wi, Cin, ldx, Di, Hi, Wi, Cout, ldy2, Di, Hi, Wi, 0, 0, 0,
                            bias2 ? IG_FLAG_BIAS : 0, stream);
      if (x2 != nullptr)
        return launch_igemm(GEO_K1, x2, wp2, nullptr, y, N, Di, Hi, Wi, Cin, ldx2, Di, Hi, Wi, Cout, ldy, Di, Hi, Wi, 0, 0, 0,
                            IG_FLAG_ACCUM, stream);
      return BTS_OK;
    }
    if (r != 1) return r;
  }
  if (geo == GEO_K1 && wp2 == nullptr && need_out == nullptr) {
    const int r = launch_k1s(x, wp, bias, y, (long)N * Di * Hi * Wi, Cin, ldx, Cout, ldy, flags, stream);
    if (r != 1) return r;
  }
  if (geo == GEO_UP && pz < 0 && wp2 == nullptr && !(flags & ~(IG_FLAG_BIAS | IG_FLAG_ACCUM | IG_FLAG_SIGMOID | IG_FLAG_VECIN | IG_FLAG_VECOUT))) {
    // merged parity classes (no workspace needed); shapes it declines fall through to the per-class launch
    if (need_out != nullptr) {
      // planning call: whether the merged kernel takes the launch also depends on pointer alignment we do not have here,
      // so report its split-K requirement as an upper bound (the per-class path needs none)
      UpmParams pp;
      long need = 0;
      if (plan_upm(pp, N, Di, Hi, Wi, Cin, Cout, &need) && need > 0) { *need_out = need; return BTS_OK; }
    } else {
      const int r = launch_upm(x, wp, bias, y, N, Di, Hi, Wi, Cin, ldx, Cout, ldy, flags, ws, ws_bytes, stream);
      if (r != 1) return r;
    }
  }
  if (geo == GEO_S1 && need_out == nullptr) {      // the implicit-GEMM form reads the first part
    const int e = bts_img_note_use_(wp, 1u, stream);
    if (e != BTS_OK) return e;
  }
  IgemmParams p;
  p.wp2 = wp2; p.bias2 = bias2; p.y2 = y2; p.ldy2 = ldy2;
  p.x2 = x2; p.ldx2 = ldx2;   // second input (x2 + wp2 without y2): see IgemmParams
  p.part = reinterpret_cast<float*>(ws);
  p.ws_bytes = ws_bytes;
  p.plan_only = need_out != nullptr;
#ifdef BTS_TIMING_EXPERIMENTS
  { const char* e = getenv("BTS_IGEMM_DBG"); p.dbg = e ? atoi(e) : 0; }
#else
  p.dbg = 0;
#endif
  p.ws_need = 0;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y;
  p.N = N; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.ldx = ldx;
  p.Do = Do; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.ldy = ldy;
  p.Npad = npad32(Cout); p.KG = (Cin + 7) / 8;
  p.ODa = ODa; p.OHa = OHa; p.OWa = OWa;
  p.os = 1; p.ooz = p.ooy = p.oox = 0;
  p.s = 1;
  p.flags = flags;
  if ((ldx % 4 == 0) && (Cin % 4 == 0) && (((uintptr_t)x) % 16 == 0)) p.flags |= IG_FLAG_VECIN;
  if ((ldy % 4 == 0) && (Cout % 4 == 0) && (((uintptr_t)y) % 16 == 0)) p.flags |= IG_FLAG_VECOUT;

  // tap table: offsets relative to o*s
  int offz[27], offy[27], offx[27], tw[27], nt = 0;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  if (geo == GEO_K1) {
    offz[0] = offy[0] = offx[0] = 0; tw[0] = 0; nt = 1;
  } else if (geo == GEO_S1) {
    for (int t = 0; t < 27; ++t) { offz[t] = t / 9 - 1; offy[t] = (t / 3) % 3 - 1; offx[t] = t % 3 - 1; tw[t] = t; }
    nt = 27;
    lo[0] = lo[1] = lo[2] = -1; hi[0] = hi[1] = hi[2] = 1;
  } else if (geo == GEO_DOWN) {
    for (int t = 0; t < 27; ++t) { offz[t] = t / 9; offy[t] = (t / 3) % 3; offx[t] = t % 3; tw[t] = t; }
    nt = 27;
    hi[0] = hi[1] = hi[2] = 2;
    p.s = 2;
  } else if (pz < 0) {  // GEO_UP, all 8 parity classes in one launch: common halo (offset -1 on every axis)
    lo[0] = lo[1] = lo[2] = -1;
    p.os = 2;
    nt = 8;  // placeholder; per-class tables are filled below once the tile geometry is known
  } else {  // GEO_UP, single parity class (pz,py,px)
    const int par[3] = {pz, py, px};
    int kz[2], ky[2], kx[2], dz[2], dy[2], dx[2], nz, ny, nx;
    auto axis = [](int pp, int* k, int* d) {
      if (pp == 0) { k[0] = 0; d[0] = 0; k[1] = 2; d[1] = -1; return 2; }
      k[0] = 1; d[0] = 0; return 1;
    };
    nz = axis(par[0], kz, dz); ny = axis(par[1], ky, dy); nx = axis(par[2], kx, dx);
    for (int a = 0; a < nz; ++a)
      for (int bb = 0; bb < ny; ++bb)
        for (int c = 0; c < nx; ++c) {
          offz[nt] = dz[a]; offy[nt] = dy[bb]; offx[nt] = dx[c];
          tw[nt] = (kz[a] * 3 + ky[bb]) * 3 + kx[c];
          ++nt;
        }
    lo[0] = par[0] == 0 ? -1 : 0; lo[1] = par[1] == 0 ? -1 : 0; lo[2] = par[2] == 0 ? -1 : 0;
    p.os = 2; p.ooz = pz; p.ooy = py; p.oox = px;
  }
  p.ntaps = nt;
  p.ncls = 1;
  p.loz = lo[0]; p.loy = lo[1]; p.lox = lo[2];

  // ---- config selection ----
  const int k1 = (geo == GEO_K1);
  int M;  // voxels per workgroup tile
  int cfg = choose_cfg(geo, (geo == GEO_UP && pz < 0) ? 8 * N : N, Do, Ho, Wo, p.Npad, &M);
  if (p.y2 != nullptr && cfg == 5) { cfg = 0; M = 256; }  // the fused pair needs the registers of the 256-voxel tiling
  // tile dims (powers of two in x,y)
  int TX = 32;
  while (TX > 4 && TX / 2 >= Wo) TX /= 2;  // smallest pow2 >= Wo, capped at 32
  if (geo == GEO_DOWN && TX > 8) TX = 8;
  if (TX > M) TX = M;
  int TY = 4;
  while (TY > 1 && TY / 2 >= Ho) TY /= 2;
  if (TX * TY > M) TY = M / TX;
  int TZ = M / (TX * TY);
  // prefer shrinking z extent into y when the volume is shallow in z
  while (TZ > 1 && TZ / 2 >= Do && TY * 2 <= 64) { TZ /= 2; TY *= 2; }
  p.lgTX = ilog2(TX); p.lgTY = ilog2(TY); p.TZ = TZ;
  p.ntx = (Wo + TX - 1) / TX; p.nty = (Ho + TY - 1) / TY; p.ntz = (Do + TZ - 1) / TZ;
  p.gnp = nullptr; p.gn_G = 0; p.gn_zt = 1; p.gn_gridy = 1;
  if (gnp != nullptr && gnG > 0 && geo != GEO_UP && Do % gnG == 0 && (Do / gnG) % TZ == 0 && getenv("BTS_IGEMM_NOGNFUSE") == nullptr) {
    p.gnp = gnp; p.gn_G = gnG; p.gn_zt = (Do / gnG) / TZ;   // whole tiles per z-slab group
  }
  p.IX = (TX - 1) * p.s + (hi[2] - lo[2] + 1);
  p.IY = (TY - 1) * p.s + (hi[1] - lo[1] + 1);
  p.IZ = (TZ - 1) * p.s + (hi[0] - lo[0] + 1);
  const int KGS = k1 ? 4 : 1;
  const int S = KGS * 8 + 4;
  for (int t = 0; t < nt; ++t) {
    p.tap_lds[t] = (((offz[t] - lo[0]) * p.IY + (offy[t] - lo[1])) * p.IX + (offx[t] - lo[2])) * S;
    p.tap_w[t] = tw[t];
  }
  for (int t = nt; t < 27; ++t) { p.tap_lds[t] = 0; p.tap_w[t] = 0; }
  p.tap_sz = p.IY * p.IX * S; p.tap_sy = p.IX * S; p.tap_sx = S;
  if (geo == GEO_UP && pz < 0) {
    p.ncls = 8;
    for (int c = 0; c < 8; ++c) {
      const int par[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
      int k[3][2], d[3][2], n[3];
      for (int a = 0; a < 3; ++a) {
        if (par[a] == 0) { k[a][0] = 0; d[a][0] = 0; k[a][1] = 2; d[a][1] = -1; n[a] = 2; }
        else { k[a][0] = 1; d[a][0] = 0; n[a] = 1; }
      }
      int ct = 0;
      for (int a = 0; a < n[0]; ++a)
        for (int bb = 0; bb < n[1]; ++bb)
          for (int cc = 0; cc < n[2]; ++cc) {
            p.cls_tl[c][ct] = (((d[0][a] + 1) * p.IY + (d[1][bb] + 1)) * p.IX + (d[2][cc] + 1)) * S;
            p.cls_tw[c][ct] = (k[0][a] * 3 + k[1][bb]) * 3 + k[2][cc];
            ++ct;
          }
      p.cls_nt[c] = ct;
      for (; ct < 8; ++ct) { p.cls_tl[c][ct] = 0; p.cls_tw[c][ct] = 0; }
    }
  }

  int rc;
  const bool fixg = (geo == GEO_S1) && p.IX == 34 && p.IY == 6;
  if (fixg && !k1 && (cfg == 0 || cfg == 1 || cfg == 5)) {
    if (p.y2 != nullptr) {
      if (cfg == 0) rc = launch_cfg<2, 1, 4, 1, 1, true, true>(p, stream);
      else rc = BTS_ERR_UNSUPPORTED;
    } else {
      if (cfg == 0) rc = launch_cfg<2, 1, 4, 1, 1, false, true>(p, stream);
      else if (cfg == 5) rc = launch_cfg<4, 1, 4, 1, 1, false, true>(p, stream);
      else rc = launch_cfg<2, 2, 4, 1, 1, false, true>(p, stream);
    }
  } else if (k1) {
    switch (cfg) {
      case 0: rc = launch_cfg<2, 1, 4, 1, 4>(p, stream); break;
      case 1: rc = launch_cfg<2, 2, 4, 1, 4>(p, stream); break;
      case 2: rc = launch_cfg<1, 2, 2, 2, 4>(p, stream); break;
      case 3: rc = launch_cfg<1, 1, 2, 2, 4>(p, stream); break;
      default: rc = launch_cfg<1, 1, 4, 1, 4>(p, stream); break;
    }
  } else if (p.y2 != nullptr) {  // fused shortcut conv: every tiling but the 64-accumulator one has the registers
    switch (cfg) {
      case 0: rc = launch_cfg<2, 1, 4, 1, 1, true>(p, stream); break;
      case 5: rc = BTS_ERR_UNSUPPORTED; break;  // M=512 tiling exists only with the compile-time halo geometry
      case 2: rc = launch_cfg<1, 2, 2, 2, 1, true>(p, stream); break;
      case 3: rc = launch_cfg<1, 1, 2, 2, 1, true>(p, stream); break;
      case 4: rc = launch_cfg<1, 1, 4, 1, 1, true>(p, stream); break;
      default: rc = BTS_ERR_UNSUPPORTED; break;
    }
  } else if (p.ntaps == 27 && p.ncls <= 1) {
    switch (cfg) {
      case 0: rc = launch_cfg<2, 1, 4, 1, 1, false, false, true>(p, stream); break;
      case 5: rc = BTS_ERR_UNSUPPORTED; break;
      case 1: rc = launch_cfg<2, 2, 4, 1, 1, false, false, true>(p, stream); break;
      case 2: rc = launch_cfg<1, 2, 2, 2, 1, false, false, true>(p, stream); break;
      case 3: rc = launch_cfg<1, 1, 2, 2, 1, false, false, true>(p, stream); break;
      default: rc = launch_cfg<1, 1, 4, 1, 1, false, false, true>(p, stream); break;
    }
  } else {
    switch (cfg) {
      case 0: rc = launch_cfg<2, 1, 4, 1, 1>(p, stream); break;
      case 5: rc = BTS_ERR_UNSUPPORTED; break;
      case 1: rc = launch_cfg<2, 2, 4, 1, 1>(p, stream); break;
      case 2: rc = launch_cfg<1, 2, 2, 2, 1>(p, stream); break;
      case 3: rc = launch_cfg<1, 1, 2, 2, 1>(p, stream); break;
      default: rc = launch_cfg<1, 1, 4, 1, 1>(p, stream); break;
    }
  }
  if (need_out) {
    *need_out = p.ws_need;
    if (geo == GEO_S1) {  // the Winograd form may split the contraction where the implicit GEMM does not (and vice versa)
      const long wn = bts_wino_workspace_(N, Di, Hi, Wi, Cin, Cout);
      if (wn > *need_out) *need_out = wn;
      const long w3 = bts_w3_workspace_(N, Di, Hi, Wi, Cin, Cout);
      if (w3 > *need_out) *need_out = w3;
    }
  }
  if (gn_B && rc == BTS_OK && p.gnp != nullptr) *gn_B = (long)p.gn_zt * p.nty * p.ntx * p.gn_gridy;
  return rc;
}

static int geo_of_kind_fwd(int kind) {
  switch (kind) {
    case BTS_CONV_K1: return GEO_K1;
    case BTS_CONV_K3S1: return GEO_S1;
    case BTS_CONV_K3S2: return GEO_DOWN;
    default: return GEO_UP;
  }
}

static int conv_fwd_impl(int kind, const float* x, const float* wp_fwd, const float* bias, float* y, void* ws, long ws_bytes,
                         long* need_out, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, int flags,
                         hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldy < Cout) return BTS_ERR_SHAPE;
  int f = 0;
  if (bias) f |= IG_FLAG_BIAS;
  if (flags & BTS_CONV_FLAG_SIGMOID) f |= IG_FLAG_SIGMOID;
  if (flags & BTS_CONV_FLAG_ACCUM) f |= IG_FLAG_ACCUM;
  const int geo = geo_of_kind_fwd(kind);
  if (geo == GEO_K1 || geo == GEO_S1)
    return launch_igemm(geo, x, wp_fwd, bias, y, N, D, H, W, Cin, ldx, D, H, W, Cout, ldy, D, H, W, 0, 0, 0, f, stream, ws,
                        ws_bytes, need_out);
  if (geo == GEO_DOWN) {
    if ((D | H | W) & 1) return BTS_ERR_SHAPE;  // TF 'same' pads (0,1) only for even sizes
    return launch_igemm(geo, x, wp_fwd, bias, y, N, D, H, W, Cin, ldx, D / 2, H / 2, W / 2, Cout, ldy, D / 2, H / 2,
                        W / 2, 0, 0, 0, f, stream, ws, ws_bytes, need_out);
  }
  return launch_igemm(GEO_UP, x, wp_fwd, bias, y, N, D, H, W, Cin, ldx, D, H, W, Cout, ldy, 2 * D, 2 * H, 2 * W, -1, -1, -1,
                      f, stream, ws, ws_bytes, need_out);
}

// Forward. x:(N,D,H,W,Cin) ld=ldx ; y:(N,Do,Ho,Wo,Cout) ld=ldy with (Do,Ho,Wo) = (D,H,W) | (D/2,..) | (2D,..).
// workspace (may be NULL) enables split-K for grids too small to fill the chip; size from bts_conv3d_fwd_workspace.
extern "C" int bts_conv3d_fwd(int kind, const float* x, const float* wp_fwd, const float* bias, float* y, void* workspace,
                              long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy,
                              int flags, hipStream_t stream) {
  return conv_fwd_impl(kind, x, wp_fwd, bias, y, workspace, workspace_bytes, nullptr, N, D, H, W, Cin, ldx, Cout, ldy, flags,
                       stream);
}
extern "C" long bts_conv3d_fwd_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  long need = 0;
  if (conv_fwd_impl(kind, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &need, N, D, H, W, Cin, Cin, Cout, Cout, 0,
                    nullptr) != BTS_OK)
    return -1;
  return need;
}

static int conv_bwd_impl(int kind, const float* dy, const float* wp_bwd, float* dx, void* ws, long ws_bytes, long* need_out,
                         int N, int D, int H, int W, int Cin, int lddx, int Cout, int lddy, int flags, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || lddx < Cin || lddy < Cout) return BTS_ERR_SHAPE;
  int f = 0;
  if (flags & BTS_CONV_FLAG_ACCUM) f |= IG_FLAG_ACCUM;
  if (kind == BTS_CONV_K1 || kind == BTS_CONV_K3S1)
    return launch_igemm(kind == BTS_CONV_K1 ? GEO_K1 : GEO_S1, dy, wp_bwd, nullptr, dx, N, D, H, W, Cout, lddy, D, H, W,
                        Cin, lddx, D, H, W, 0, 0, 0, f, stream, ws, ws_bytes, need_out);
  if (kind == BTS_CONV_K3S2) {  // gather form over the fine grid's parity classes
    if ((D | H | W) & 1) return BTS_ERR_SHAPE;
    return launch_igemm(GEO_UP, dy, wp_bwd, nullptr, dx, N, D / 2, H / 2, W / 2, Cout, lddy, D / 2, H / 2, W / 2, Cin, lddx,
                        D, H, W, -1, -1, -1, f, stream, ws, ws_bytes, need_out);
  }
  // transposed conv: d/dx is the stride-2 'same' conv of dy (fine grid 2D x 2H x 2W) -> coarse grid
  return launch_igemm(GEO_DOWN, dy, wp_bwd, nullptr, dx, N, 2 * D, 2 * H, 2 * W, Cout, lddy, D, H, W, Cin, lddx, D, H, W,
                      0, 0, 0, f, stream, ws, ws_bytes, need_out);
}

// Data gradient. dy has the forward output's shape, dx the forward input's shape (N,D,H,W,Cin).
// wp_bwd is the BTS_ROLE_BWD_DATA packing. flags: BTS_CONV_FLAG_ACCUM adds into dx.
extern "C" int bts_conv3d_bwd_data(int kind, const float* dy, const float* wp_bwd, float* dx, void* workspace,
                                   long workspace_bytes, int N, int D, int H, int W, int Cin, int lddx, int Cout, int lddy,
                                   int flags, hipStream_t stream) {
  return conv_bwd_impl(kind, dy, wp_bwd, dx, workspace, workspace_bytes, nullptr, N, D, H, W, Cin, lddx, Cout, lddy, flags,
                       stream);
}
extern "C" long bts_conv3d_bwd_data_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  long need = 0;
  if (conv_bwd_impl(kind, nullptr, nullptr, nullptr, nullptr, 0, &need, N, D, H, W, Cin, Cin, Cout, Cout, 0, nullptr) != BTS_OK)
    return -1;
  return need;
}

// Fused pair of the ResNet block (resnet.py:118 and :133-134): y = conv3x3x3(x) + bias, y2 = conv1x1x1(x) + bias2 from ONE
// pass over x. wp_fwd: K3S1 forward packing, wp2: K1 forward packing of the shortcut kernel (same Cin/Cout).
// Returns BTS_ERR_UNSUPPORTED when the selected tiling has no register room for the second accumulator set
// (bts_conv3d_fwd_can_fuse tells beforehand): the caller then issues the two convolutions separately.
extern "C" int bts_conv3d_fwd_can_fuse(int N, int D, int H, int W, int Cin, int Cout) {
  int M;
  return choose_cfg(GEO_S1, N, D, H, W, npad32(Cout), &M) != 1;
}
extern "C" int bts_conv3d_fwd_fused2(const float* x, const float* wp_fwd, const float* bias, float* y, const float* wp2,
                                     const float* bias2, float* y2, int N, int D, int H, int W, int Cin, int ldx, int Cout,
                                     int ldy, int ldy2, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldy < Cout || ldy2 < Cout) return BTS_ERR_SHAPE;
  if (!wp2 || !y2) return BTS_ERR_SHAPE;
  return launch_igemm(GEO_S1, x, wp_fwd, bias, y, N, D, H, W, Cin, ldx, D, H, W, Cout, ldy, D, H, W, 0, 0, 0,
                      bias ? IG_FLAG_BIAS : 0, stream, nullptr, 0, nullptr, wp2, bias2, y2, ldy2);
}

// ---- convolution + GroupNorm statistics of its output (resnet.py:80-93, downsample.py:41-43: conv -> GroupNormalization) ----
extern "C" long bts_gn_workspace(int N, long V, int C, int G, int mode);
extern "C" int bts_gn_stats(const float* x, float* mean, float* rstd, void* workspace, long workspace_bytes, int N, long V, int C,
                            int G, int mode, float eps, hipStream_t stream);
static long gnfuse_partial_bytes(int N, int Do, int Ho, int Wo, int Cout) {
  // upper bound of N*G*B*2 doubles: one pair per 64-voxel tile and 32-cout tile
  const long tiles = (long)N * ((Do + 0) ) * ((Ho + 3) / 4) * ((Wo + 7) / 8);
  return tiles * ((npad32(Cout) + 31) / 32) * 2 * (long)sizeof(double) + 64;
}
extern "C" long bts_conv3d_fwd_gn_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout, int G) {
  const long cw = bts_conv3d_fwd_workspace(kind, N, D, H, W, Cin, Cout);
  if (cw < 0) return -1;
  int Do = D, Ho = H, Wo = W;
  if (kind == BTS_CONV_K3S2) { Do = D / 2; Ho = H / 2; Wo = W / 2; }
  if (kind == BTS_CONV_K3S2T) { Do = 2 * D; Ho = 2 * H; Wo = 2 * W; }
  const long gw = bts_gn_workspace(N, (long)Do * Ho * Wo, Cout, G, BTS_GN_SLAB);
  if (gw < 0) return -1;
  const long fused = ((cw + 63) & ~63L) + gnfuse_partial_bytes(N, Do, Ho, Wo, Cout);
  return fused > gw ? fused : gw;
}
// y = conv(x) + bias (y dense: ldy == Cout) and (mean, rstd) = slab-mode GroupNorm statistics of y.  The statistics come out
// of the conv epilogue when the tiled kernel takes the launch without split-K and the z-slab groups hold whole tiles;
// otherwise bts_gn_stats runs on y afterwards (same result up to the summation order, both deterministic).
static int conv_fwd_gn_impl(int kind, const float* x, const float* wp, const float* bias, float* y, const float* wp2,
                            const float* bias2, float* y2, int ldy2, void* ws, long ws_bytes, int N, int D, int H, int W, int Cin,
                            int ldx, int Cout, int G, float eps, float* mean, float* rstd, hipStream_t stream) {
  if (G <= 0 || Cout % G != 0) return BTS_ERR_SHAPE;
  if (ws == nullptr || ws_bytes < bts_conv3d_fwd_gn_workspace(kind, N, D, H, W, Cin, Cout, G)) return BTS_ERR_WORKSPACE;
  int Do = D, Ho = H, Wo = W;
  if (kind == BTS_CONV_K3S2) { Do = D / 2; Ho = H / 2; Wo = W / 2; }
  if (kind == BTS_CONV_K3S2T) { Do = 2 * D; Ho = 2 * H; Wo = 2 * W; }
  const long cw = (bts_conv3d_fwd_workspace(kind, N, D, H, W, Cin, Cout) + 63) & ~63L;
  double* gnp = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + cw);
  long gnB = 0;
  int r;
  if (wp2 != nullptr) {
    // the workspace serves the Winograd form's split-K on small grids (the fused implicit-GEMM pair never splits)
    r = launch_igemm(GEO_S1, x, wp, bias, y, N, D, H, W, Cin, ldx, D, H, W, Cout, Cout, D, H, W, 0, 0, 0, bias ? IG_FLAG_BIAS : 0,
                     stream, ws, cw, nullptr, wp2, bias2, y2, ldy2, gnp, G, &gnB);
  } else {
    const int geo = geo_of_kind_fwd(kind);
    const int f = bias ? IG_FLAG_BIAS : 0;
    if (geo == GEO_K1 || geo == GEO_S1)
      r = launch_igemm(geo, x, wp, bias, y, N, D, H, W, Cin, ldx, D, H, W, Cout, Cout, D, H, W, 0, 0, 0, f, stream, ws, cw, nullptr,
                       nullptr, nullptr, nullptr, 0, gnp, G, &gnB);
    else if (geo == GEO_DOWN) {
      if ((D | H | W) & 1) return BTS_ERR_SHAPE;
      r = launch_igemm(geo, x, wp, bias, y, N, D, H, W, Cin, ldx, Do, Ho, Wo, Cout, Cout, Do, Ho, Wo, 0, 0, 0, f, stream, ws, cw,
                       nullptr, nullptr, nullptr, nullptr, 0, gnp, G, &gnB);
    } else
      r = launch_igemm(GEO_UP, x, wp, bias, y, N, D, H, W, Cin, ldx, D, H, W, Cout, Cout, Do, Ho, Wo, -1, -1, -1, f, stream, ws, cw);
  }
  if (r != BTS_OK) return r;
  const long V = (long)Do * Ho * Wo;
  if (gnB > 0) return bts_gn_finalize_partials_(gnp, mean, rstd, N * G, gnB, (double)V * Cout / G, eps, stream);
  return bts_gn_stats(y, mean, rstd, ws, ws_bytes, N, V, Cout, G, BTS_GN_SLAB, eps, stream);
}
extern "C" int bts_conv3d_fwd_gn(int kind, const float* x, const float* wp_fwd, const float* bias, float* y, void* workspace,
                                 long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G, float eps,
                                 float* mean, float* rstd, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin) return BTS_ERR_SHAPE;
  if (((uintptr_t)workspace) & 15) return BTS_ERR_ALIGN;     // (the GroupNorm partials in it are read as double2: refuse before y is written)
  return conv_fwd_gn_impl(kind, x, wp_fwd, bias, y, nullptr, nullptr, nullptr, 0, workspace, workspace_bytes, N, D, H, W, Cin, ldx,
                          Cout, G, eps, mean, rstd, stream);
}
extern "C" int bts_conv3d_fwd_fused2_gn(const float* x, const float* wp_fwd, const float* bias, float* y, const float* wp2,
                                        const float* bias2, float* y2, void* workspace, long workspace_bytes, int N, int D, int H,
                                        int W, int Cin, int ldx, int Cout, int ldy2, int G, float eps, float* mean, float* rstd,
                                        hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldy2 < Cout || !wp2 || !y2) return BTS_ERR_SHAPE;
  if (((uintptr_t)workspace) & 15) return BTS_ERR_ALIGN;
  return conv_fwd_gn_impl(BTS_CONV_K3S1, x, wp_fwd, bias, y, wp2, bias2, y2, ldy2, workspace, workspace_bytes, N, D, H, W, Cin, ldx,
                          Cout, G, eps, mean, rstd, stream);
}

// dx (+)= data gradient of a 3x3x3 stride-1 conv (dy, wp_bwd) + data gradient of a 1x1x1 conv of the SAME input (dy2, wp2_bwd):
// the two gradient paths of a ResNet block's input (resnet.py:118,134) in one pass over dx.  Falls back to two launches when
// the tiled kernel would split the contraction or dy2 cannot be read with 16-byte loads.
extern "C" long bts_conv3d_bwd_data_pair_workspace(int N, int D, int H, int W, int Cin, int Cout) {
  const long a = bts_conv3d_bwd_data_workspace(BTS_CONV_K3S1, N, D, H, W, Cin, Cout);
  const long b = bts_conv3d_bwd_data_workspace(BTS_CONV_K1, N, D, H, W, Cin, Cout);
  if (a < 0 || b < 0) return -1;
  return a > b ? a : b;
}
extern "C" int bts_conv3d_bwd_data_pair(const float* dy, const float* wp_bwd, const float* dy2, const float* wp2_bwd, float* dx,
                                        void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int lddx,
                                        int Cout, int lddy, int lddy2, int flags, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || lddx < Cin || lddy < Cout || lddy2 < Cout) return BTS_ERR_SHAPE;
  long need = 0;
  int r = conv_bwd_impl(BTS_CONV_K3S1, nullptr, nullptr, nullptr, nullptr, 0, &need, N, D, H, W, Cin, Cin, Cout, Cout, 0, nullptr);
  if (r != BTS_OK) return r;
  const bool fuse = need == 0 && (Cout % 8 == 0) && (lddy2 % 4 == 0) && ((((uintptr_t)dy2) & 15) == 0) && Cin > 4 &&
                    getenv("BTS_IGEMM_NOPAIR") == nullptr;
  if (fuse) {
    int f = 0;
    if (flags & BTS_CONV_FLAG_ACCUM) f |= IG_FLAG_ACCUM;
    return launch_igemm(GEO_S1, dy, wp_bwd, nullptr, dx, N, D, H, W, Cout, lddy, D, H, W, Cin, lddx, D, H, W, 0, 0, 0, f, stream,
                        workspace, workspace_bytes, nullptr, wp2_bwd, nullptr, nullptr, 0, nullptr, 0, nullptr, dy2, lddy2);
  }
  r = conv_bwd_impl(BTS_CONV_K3S1, dy, wp_bwd, dx, workspace, workspace_bytes, nullptr, N, D, H, W, Cin, lddx, Cout, lddy, flags, stream);
  if (r != BTS_OK) return r;
  return conv_bwd_impl(BTS_CONV_K1, dy2, wp2_bwd, dx, workspace, workspace_bytes, nullptr, N, D, H, W, Cin, lddx, Cout, lddy2,
                       flags | BTS_CONV_FLAG_ACCUM, stream);
}

// Which igemm_kernel<...> instantiation a call resolves to: returns cfg + 8*(KGS==4); cfg ids as in choose_cfg.
// Lets the host attribute measured launch times to kernel symbols (bench.py roofline).
extern "C" int bts_conv3d_fwd_config(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  int M;
  const int geo = geo_of_kind_fwd(kind);
  int Do = D, Ho = H, Wo = W;
  if (geo == GEO_DOWN) { Do = D / 2; Ho = H / 2; Wo = W / 2; }
  return choose_cfg(geo, geo == GEO_UP ? 8 * N : N, Do, Ho, Wo, npad32(Cout), &M) + (geo == GEO_K1 ? 8 : 0);
}
extern "C" int bts_conv3d_bwd_data_config(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  int M;
  int geo = GEO_S1, Do = D, Ho = H, Wo = W;
  if (kind == BTS_CONV_K1) geo = GEO_K1;
  else if (kind == BTS_CONV_K3S2) { geo = GEO_UP; Do = D / 2; Ho = H / 2; Wo = W / 2; }
  else if (kind == BTS_CONV_K3S2T) geo = GEO_DOWN;
  return choose_cfg(geo, geo == GEO_UP ? 8 * N : N, Do, Ho, Wo, npad32(Cin), &M) + (geo == GEO_K1 ? 8 : 0);
}
