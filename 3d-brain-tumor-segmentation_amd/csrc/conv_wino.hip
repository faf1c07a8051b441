// 3x3x3 stride-1 'same' convolution with the (z, y) plane in Winograd F(2x2, 3x3) form and the x axis direct, on the
// exact-fp32 matrix pipe of gfx950 (v_mfma_f32_32x32x2_f32).  Serves the same call sites as the 27-tap implicit GEMM
// (reference: layers/resnet.py:30-37,80-87,96-103  layers/decoder.py:55-63  layers/vae.py:92-99, and their data
// gradients = the same form on flipped, transposed weights): 12 matrix instructions per output voxel and channel pair
// instead of 27.
//
//   Y = A^T [ (G g G^T) o (B^T d B) ] A   per 2x2 (z, y) output patch, summed over the three x taps and the input channels
//   B^T = (1 0 -1 0 | 0 1 1 0 | 0 -1 1 0 | 0 1 0 -1)   G = (1 0 0 | .5 .5 .5 | .5 -.5 .5 | 0 0 1)   A^T = (1 1 1 0 | 0 1 -1 -1)
//
// U = G g G^T is formed at weight-packing time (conv_igemm.hip: second part of the K3S1 image, layout
// [cout block of 32][k-group of 8 cin][x tap][xi = xi_z*4 + xi_y][half h][32 couts][4 cin]).
//
// Work decomposition: 256 threads = 4 waves; workgroup tile = 32 (x) x 4 (y) x 4 (z) output voxels x 32 couts; each wave
// owns one 2x2 (z, y) patch row of 32 x positions and ALL 16 transform points: 16 accumulators of 32 voxels x 32 couts =
// 256 accumulation registers, one wave per SIMD.  The 34 x 6 x 6 halo tile of 8 input channels is staged
// global -> registers -> LDS (double buffered, voxel stride 12 dwords: conflict-free ds_read_b128, as the implicit GEMM).
// Lane (h, x) reads the 4 x 4 patch of its x position for channels h*4..h*4+3, forms the transform with packed fp32 adds
// one group (one xi_z, four xi_y) AHEAD of the matrix instructions that consume it, and the U fragments stream from L2
// two groups ahead.  Out-of-image halo voxels are fetched with an out-of-range buffer offset (the load returns zeros).
//
// Rounding: the transforms add fp32 values before the multiply, so results differ from the direct form in the last bits
// (measured max |err| 5e-6 at |y| ~ 3, K = 864); the summation order is fixed (deterministic).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "bts_internal.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

#ifdef BTS_WINO_STAMPS   // experiment builds only (scripts/wino_timeline.py): per-item clock stamps of wave 0
__device__ long long g_wino_stamps[1 << 20];
extern "C" int bts_wino_stamps_copy_(long long* dst, long n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wino_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#define WSTAMP(slot)                                                                              \
  do {                                                                                            \
    if (tid == 0) g_wino_stamps[((long)blockIdx.x * p.T + it) * 16 + (slot)] = wall_clock64();    \
  } while (0)
#else
#define WSTAMP(slot) do { } while (0)
#endif

#include "wino_util.h"

struct WinoParams {
  const float* x;
  const float* up;
  const float* bias;
  float* y;
  int N, D, H, W, ldx, Cout, ldy, KG;
  int ntz, nty, ntx;
  int accum;
  double* gnp;  // fused GroupNorm partial sums (slab semantics), layout as igemm_kernel's
  int gn_G, gn_zt;
  // split-K: blockIdx.z handles k-groups [z*kg_per, ...); raw partial outputs go to part[z][voxel][Npad] and are finished
  // (bias, accumulate) by the implicit GEMM's reduce kernel
  int ksplit, kg_per, Npad;
  float* part;
  int nb, ntiles, tiles_per_xcd;   // cout blocks of 32; tiles = N * ntz * nty * ntx
  int T;                           // (tile, cout block) items one workgroup walks back to back
};

#define WS 12      // dwords per staged voxel: 8 channels + 4 pad (16-byte-odd stride, conflict-free ds_read_b128)
#define WNSLOT 10  // staging slots (16 bytes) per thread and stage
#ifndef WPF
#define WPF 2      // groups of 16 matrix instructions a U fragment is requested ahead of its use
#endif
// Tile geometry.  XW = 32: a wave's 32 matrix columns are 32 x positions of one 2x2 (z, y) patch row; tile 32 x 4 x 4.
// XW = 16 (grids narrower than 32): the columns are 16 x positions of TWO y patches; tile 16 x 8 x 4.  The LDS row
// stride LX is padded to 24 voxels there: two patch rows are then a multiple of 64 dwords apart and the two halves of a
// ds_read_b128 lane group fall on complementary banks.
template <int XW>
struct WinoGeo {
  static constexpr int SX = XW + 2;                 // staged halo columns
  static constexpr int LX = (XW == 32) ? 34 : 24;   // LDS row stride (voxels)
  static constexpr int IY = (XW == 32) ? 6 : 10, IZ = 6;
  static constexpr int TY = (XW == 32) ? 4 : 8;     // output rows per tile
  static constexpr int NSTAGE = SX * IY * IZ;       // staged voxels (1224 / 1080)
  static constexpr int BUF = LX * IY * IZ * WS;     // dwords per LDS buffer
  static constexpr int off(int i, int j, int dx) { return ((i * IY + j) * LX + dx) * WS; }
};
#define LDSOFF(i, j, dx) (G_::off(i, j, dx))

template <int XW>
__global__ __launch_bounds__(256, 1) void wino_kernel(const WinoParams p) {
  typedef WinoGeo<XW> G_;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  // 1-D grid over (tile, cout block) ITEMS.  Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2): XCD k
  // walks its own contiguous eighth of the tiles, the cout blocks of a tile back to back -- x / y / z neighbours (a third of
  // every halo tile each) and the second cout block find their data in the XCD's L2 instead of behind the fabric.  One
  // workgroup takes p.T consecutive items of its XCD's sequence: the first halo tile and weight fragments of item i+1 are
  // requested BEFORE the output transform / stores of item i, so only a workgroup's first item pays the fetch latency with
  // the matrix pipe idle (one wave per SIMD: nothing else on the CU could hide it).
  const int nb_ = p.nb;
  const int xcd_ = blockIdx.x & 7;
  const int seq0 = (blockIdx.x >> 3) * p.T;

  // per-slot constants of the staging pattern (tile independent): voxel coordinates inside the halo tile and the byte
  // offset from the tile's halo origin
  unsigned gbase[WNSLOT], gpos[WNSLOT];
#pragma unroll
  for (int i = 0; i < WNSLOT; ++i) {
    const int e = tid + i * 256;
    gbase[i] = 0x80000000u;
    gpos[i] = 0x00ffffffu;   // (255, 255, 255): never inside an image
    if (e < G_::NSTAGE * 2) {
      const int vox = e >> 1, q = e & 1;
      const int vz = vox / (G_::IY * G_::SX);
      const int r = vox - vz * (G_::IY * G_::SX);
      const int vy = r / G_::SX;
      const int vx = r - vy * G_::SX;
      gbase[i] = (unsigned)(((vz * p.H + vy) * p.W + vx) * p.ldx + q * 4) * 4u;
      gpos[i] = (unsigned)((vz << 16) | (vy << 8) | vx);
    }
  }
  const int tz2 = wave >> 1;
  const int ty2 = (XW == 32) ? (wave & 1) : ((wave & 1) * 2 + (l32 >> 4));  // y patch of this lane
  const int xl = (XW == 32) ? l32 : (l32 & 15);                              // x position inside the tile
  const int bbase = ((2 * tz2 * G_::IY + 2 * ty2) * G_::LX + xl) * WS + h * 4;
  const unsigned wlane = (unsigned)((h * 32 + l32) * 16);

  int st0 = 0, st1 = p.KG;
  if (p.ksplit > 1) {
    st0 = blockIdx.z * p.kg_per;
    st1 = st0 + p.kg_per;
    if (st1 > p.KG) st1 = p.KG;
  }

  // item coordinates (wave-uniform)
  struct Item { int n, tz, ty, tx, cb; };
  auto decode = [&](int seq) {
    Item it;
    it.cb = seq % nb_;
    int b = xcd_ * p.tiles_per_xcd + seq / nb_;
    it.tx = b % p.ntx; b /= p.ntx;
    it.ty = b % p.nty; b /= p.nty;
    it.tz = b % p.ntz;
    it.n = b / p.ntz;
    return it;
  };
  // the item after `it` in this XCD's sequence (no divisions: cout block, then x, y, z, sample); false past the last one
  int tiles_left = p.ntiles - xcd_ * p.tiles_per_xcd;          // tiles of this XCD's eighth that exist
  if (tiles_left > p.tiles_per_xcd) tiles_left = p.tiles_per_xcd;
  const int items_here = tiles_left * nb_;
  if (seq0 >= items_here) return;   // (before any barrier: whole workgroups leave)
  auto advance = [&](Item& it) {
    if (++it.cb < nb_) return;
    it.cb = 0;
    if (++it.tx < p.ntx) return;
    it.tx = 0;
    if (++it.ty < p.nty) return;
    it.ty = 0;
    if (++it.tz < p.ntz) return;
    it.tz = 0;
    ++it.n;
  };

  __amdgpu_buffer_rsrc_t xr, wr, wr_n;
  unsigned goff[WNSLOT];
  // input descriptor + per-slot offsets of an item; slots outside the image get a 2 GB offset = outside the descriptor -> zeros
  auto setup_x = [&](const Item& it) {
    const int iz0 = it.tz * 4 - 1, iy0 = it.ty * G_::TY - 1, ix0 = it.tx * XW - 1;
    const float* xorg = p.x + ((((long)it.n * p.D + iz0) * p.H + iy0) * p.W + ix0) * (long)p.ldx;
    xr = __builtin_amdgcn_make_buffer_rsrc((void*)xorg, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < WNSLOT; ++i) {
      const int vz = (int)(gpos[i] >> 16), vy = (int)((gpos[i] >> 8) & 255), vx = (int)(gpos[i] & 255);
      const bool in = (unsigned)(iz0 + vz) < (unsigned)p.D && (unsigned)(iy0 + vy) < (unsigned)p.H && (unsigned)(ix0 + vx) < (unsigned)p.W;
      goff[i] = in ? gbase[i] : 0x80000000u;
    }
  };
  // U of an item's cout block: [k-group][x tap][xi][h][32][4] = 1 KB per (x tap, xi) image
  auto wdesc = [&](const Item& it) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.up + (long)it.cb * p.KG * (3 * 16 * 256)), 0, 0x7fffffff, 0x00020000);
  };

  f32x16 acc[4][4];
  f32x4 pre[WNSLOT];
  // soff = 0x80000000: nothing left to fetch -- every slot is then out of range (zeros, no traffic).  The requests are issued
  // all the same: with one request count on every path the compiler's vmcnt waits stay exact, and a wait for a weight
  // fragment never drains the stores of the previous item's output (vector memory returns in order)
  auto fetch = [&](unsigned soff) {
#pragma unroll
    for (int i = 0; i < WNSLOT; ++i) pre[i] = bufload(xr, goff[i], soff);
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < WNSLOT; ++i) {
      const int e = tid + i * 256;
      if (e < G_::NSTAGE * 2) {
        const int vox = e >> 1;
        const int row = vox / G_::SX;   // (vz*IY + vy); SX == LX for the 32-wide tile
        const int lo = (G_::SX == G_::LX) ? vox * WS : (row * G_::LX + (vox - row * G_::SX)) * WS;
        *reinterpret_cast<f32x4*>(buf + lo + (e & 1) * 4) = pre[i];
      }
    }
  };
  // U fragments of group G = st*12 + g live in aw[G % (WPF + 1)]; two groups are always in flight
  f32x4 aw[WPF + 1][4];
  constexpr int zorder[4] = {1, 2, 0, 3};
  auto wload = [&](f32x4 (&dst)[4], const __amdgpu_buffer_rsrc_t d, int st, int g) {
    const int dx = g >> 2, xz = zorder[g & 3];
    const unsigned so = (unsigned)(((st * 3 + dx) * 16 + xz * 4) * 1024);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e] = bufload(d, wlane + e * 1024, so);
  };
  double* const gsh = reinterpret_cast<double*>(lds + 2 * G_::BUF);   // GroupNorm partial exchange: past the staging buffers
  float* const bsh = lds + 2 * G_::BUF + 16;                           // 32 bias values of the current / next item
  float bias_v = 0.f;
  auto bias_of = [&](const Item& it) {
    const int co = it.cb * 32 + tid;
    return (p.bias && p.ksplit <= 1 && tid < 32 && co < p.Cout) ? p.bias[co] : 0.f;
  };

  // One k-group stage: 12 groups of 16 matrix instructions on LDS buffer `par`.  FIRST: the item's first stage (accumulators
  // start from zero).  more: the item has another stage -- its halo tile is fetched now and committed to the other buffer at
  // the end.  chain (last stage of an item that has a successor in this workgroup; xr / goff / wr_n / bias_v already belong to
  // the successor): the successor's FIRST halo tile and weight fragments take the place of the next stage's, so an item
  // boundary costs what a stage boundary costs and the output transform below runs with everything it needs next in LDS.
  auto stage = [&](auto first_tag, int st, int par, bool more, bool chain, int bias_slot) {
    constexpr bool FIRST = decltype(first_tag)::value;
    fetch(more ? (unsigned)(st + 1) * 32u : (chain ? (unsigned)st0 * 32u : 0x80000000u));
    const float* lb = lds + par * G_::BUF + bbase;
    const int stn = more ? st + 1 : st0;              // (neither more nor chain: a harmless re-request of this cout block's U)
    const __amdgpu_buffer_rsrc_t wt = chain ? wr_n : wr;
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
    asm volatile("s_nop 3" ::: "memory");
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      {  // MFMA xi_z = 1 ; form xi_z = 2 : d2 - d1 ; request row 0
        const int G = dx * 4 + 0;
        wload(aw[(G + WPF) % (WPF + 1)], (G + WPF < 12) ? wr : wt, (G + WPF < 12) ? st : stn, (G + WPF) % 12);
#pragma unroll
        for (int j = 0; j < 4; ++j) rt[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(0, j, dx));
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = sub4(r2[j], r1[j]);
        wino_yt(c, v[1]);
        if (FIRST && dx == 0) wino_mfma16<true>(v[0], aw[G % (WPF + 1)], acc[1]); else wino_mfma16<false>(v[0], aw[G % (WPF + 1)], acc[1]);
        WINO_SCHED_GROUP();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");  // hand-written transform VALU -> next group's MFMA operand: hazard not tracked by the compiler
      }
      {  // MFMA xi_z = 2 ; form xi_z = 0 : d0 - d2 ; then request row 3
        const int G = dx * 4 + 1;
        wload(aw[(G + WPF) % (WPF + 1)], (G + WPF < 12) ? wr : wt, (G + WPF < 12) ? st : stn, (G + WPF) % 12);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = sub4(rt[j], r2[j]);
        wino_yt(c, v[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) rt[j] = *reinterpret_cast<const f32x4*>(lb + LDSOFF(3, j, dx));
        if (FIRST && dx == 0) wino_mfma16<true>(v[1], aw[G % (WPF + 1)], acc[2]); else wino_mfma16<false>(v[1], aw[G % (WPF + 1)], acc[2]);
        WINO_SCHED_GROUP();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
      }
      {  // MFMA xi_z = 0 ; form xi_z = 3 : d1 - d3 ; then request rows 1, 2 of the next x tap
        const int G = dx * 4 + 2;
        wload(aw[(G + WPF) % (WPF + 1)], (G + WPF < 12) ? wr : wt, (G + WPF < 12) ? st : stn, (G + WPF) % 12);
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
        if (FIRST && dx == 0) wino_mfma16<true>(v[0], aw[G % (WPF + 1)], acc[0]); else wino_mfma16<false>(v[0], aw[G % (WPF + 1)], acc[0]);
        WINO_SCHED_GROUP();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
      }
      {  // MFMA xi_z = 3 ; form xi_z = 1 of the next x tap : d1 + d2
        const int G = dx * 4 + 3;
        wload(aw[(G + WPF) % (WPF + 1)], (G + WPF < 12) ? wr : wt, (G + WPF < 12) ? st : stn, (G + WPF) % 12);
        if (dx < 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) c[j] = add4(r1[j], r2[j]);
          wino_yt(c, v[0]);
        }
        if (FIRST && dx == 0) wino_mfma16<true>(v[1], aw[G % (WPF + 1)], acc[3]); else wino_mfma16<false>(v[1], aw[G % (WPF + 1)], acc[3]);
        WINO_SCHED_GROUP();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
      }
    }
    if (more || chain) commit(lds + (par ^ 1) * G_::BUF);
    if (chain && tid < 32) bsh[bias_slot * 32 + tid] = bias_v;
    __syncthreads();
  };

  // output side: buffer stores with an out-of-range offset for masked lanes -- the store count is the same on every path
  // (see fetch) and there are no branches around the stores
  auto store4 = [&](const __amdgpu_buffer_rsrc_t d, unsigned voff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), d, voff, 0, 0);
  };

  // ---- output transform (A^T . A over (z, y)), bias, optional accumulate, store; D rows = couts (4 per register quad) ----
  auto finish = [&](const Item& o, int slot) {
    const bool raw = p.ksplit > 1;
    const int oxx = o.tx * XW + xl;
    const int zb = o.tz * 4 + 2 * tz2, yb = o.ty * G_::TY + 2 * ty2;
    const int ld = raw ? p.Npad : p.ldy;
    const float* obase = raw ? p.part + ((long)blockIdx.z * p.N + o.n) * p.D * p.H * p.W * (long)p.Npad
                             : p.y + (long)o.n * p.D * p.H * p.W * (long)p.ldy;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, 0x7fffffff, 0x00020000);
    const int clim = raw ? p.Npad : p.Cout;
    unsigned yo[2][2];   // [oz][oy] byte offsets of this lane's first cout quad; masked positions out of range
#pragma unroll
    for (int oz = 0; oz < 2; ++oz)
#pragma unroll
      for (int oy = 0; oy < 2; ++oy) {
        const bool ok = oxx < p.W && (yb + oy) < p.H && (zb + oz) < p.D;
        yo[oz][oy] = ok ? (unsigned)(((((zb + oz) * p.H + (yb + oy)) * p.W + oxx) * ld + o.cb * 32 + 4 * h) * 4) : 0x80000000u;
      }
    f32x4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(bsh + slot * 32 + 8 * g + 4 * h);
    // the accumulators are read with hand-written v_accvgpr_read (acc_rd): the compiler's hazard tracking does not see
    // them, so the 16-pass latency of the last matrix instructions is covered explicitly (18 wait states required, 80 given)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float gn_s = 0.f, gn_q = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // one register quad (4 couts) of all 16 accumulators at a time
      f32x4 tq[4][2];
#pragma unroll
      for (int xz = 0; xz < 4; ++xz) {
        f32x4 q[4];
#pragma unroll
        for (int xy = 0; xy < 4; ++xy)
          q[xy] = f32x4{acc_rd(acc[xz][xy][4 * g]), acc_rd(acc[xz][xy][4 * g + 1]), acc_rd(acc[xz][xy][4 * g + 2]), acc_rd(acc[xz][xy][4 * g + 3])};
        tq[xz][0] = q[0] + q[1] + q[2];
        tq[xz][1] = q[1] - q[2] - q[3];
      }
      const bool cok = o.cb * 32 + 8 * g + 4 * h < clim;
#pragma unroll
      for (int oy = 0; oy < 2; ++oy) {
        f32x4 ov[2];
        ov[0] = tq[0][oy] + tq[1][oy] + tq[2][oy] + bq[g];
        ov[1] = tq[1][oy] - tq[2][oy] - tq[3][oy] + bq[g];
#pragma unroll
        for (int oz = 0; oz < 2; ++oz) {
          const unsigned off = cok ? yo[oz][oy] : 0x80000000u;
          const bool live = off != 0x80000000u;
          f32x4 v = ov[oz];
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // masked positions contribute nothing to the GroupNorm sums
            const float t = live ? v[j] : 0.f;
            gn_s += t;
            gn_q = fmaf(t, t, gn_q);
          }
          if (p.accum && !raw) v += bufload(yr, off + 32u * g, 0);
          store4(yr, off + 32u * g, v);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // GroupNorm partials: fixed-order combine, lanes (shuffle tree) -> 4 waves (LDS) -> one (sum, sumsq) pair per item.  The
    // exchange runs (and thread 0 stores, out of range when there is no consumer) whether or not p.gnp is set: same request
    // count on every path
    {
      const double ds = wave_sum_f64((double)gn_s), dq = wave_sum_f64((double)gn_q);
      if (lane == 0) { gsh[wave * 2] = ds; gsh[wave * 2 + 1] = dq; }
      __syncthreads();
      const int gn_g = o.tz / p.gn_zt;
      const long B = (long)p.gn_zt * p.nty * p.ntx * nb_;
      const long gn_slot = (((long)(o.tz - gn_g * p.gn_zt) * p.nty + o.ty) * p.ntx + o.tx) * nb_ + o.cb;
      const double* dst = p.gnp ? p.gnp + (((long)o.n * p.gn_G + gn_g) * B + gn_slot) * 2 : nullptr;
      const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, p.gnp ? 16 : 0, 0x00020000);
      const double s0 = gsh[0] + gsh[2] + gsh[4] + gsh[6], s1 = gsh[1] + gsh[3] + gsh[5] + gsh[7];
      struct D2 { double a, b; } d2{s0, s1};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d2), gr, tid == 0 ? 0u : 0x80000000u, 0, 0);
      // (gsh is rewritten only after the next item's stage barriers)
    }
  };

  Item cur = decode(seq0);
  setup_x(cur);
  wr = wdesc(cur);
  wr_n = wr;
  bias_v = bias_of(cur);
  fetch((unsigned)st0 * 32u);
#pragma unroll
  for (int g = 0; g < WPF; ++g) wload(aw[g], wr, st0, g);
  commit(lds);
  if (tid < 32) bsh[tid] = bias_v;
  __syncthreads();
  {  // as many (dropped) stores as an item's output side issues: the first item's waits then match the later items'
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 17; ++i) store4(none, 0x80000000u + 16u * i, f32x4{0.f, 0.f, 0.f, 0.f});   // (distinct: not dead stores)
  }

  int par = 0;
  for (int it = 0; it < p.T; ++it) {
    const bool have_next = (it + 1 < p.T) && (seq0 + it + 1 < items_here);
    const Item out = cur;
    WSTAMP(0);
#ifdef BTS_WINO_STAMPS
    if (tid == 0) {
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_wino_stamps[((long)blockIdx.x * p.T + it) * 16 + 6] = ((long long)xcc << 32) | hwid;
      g_wino_stamps[((long)blockIdx.x * p.T + it) * 16 + 7] = clock64();
    }
#endif
    WSTAMP(1);
    // from an item's last stage on, the input-side state is the successor's (the item's own last halo tile is in LDS by then)
    auto enter = [&](int st) {
      const bool chain = (st + 1 == st1) && have_next;
      if (chain) {
        advance(cur);
        setup_x(cur);
        wr_n = wdesc(cur);
        bias_v = bias_of(cur);
      }
      return chain;
    };
    {
      const bool chain = enter(st0);
      stage(std::true_type{}, st0, par, st0 + 1 < st1, chain, (it + 1) & 1);
      par ^= 1;
    }
    WSTAMP(2);
    for (int st = st0 + 1; st < st1; ++st) {
      const bool chain = enter(st);
      stage(std::false_type{}, st, par, st + 1 < st1, chain, (it + 1) & 1);
      par ^= 1;
    }
    WSTAMP(3);
    if (have_next) wr = wr_n;
    WSTAMP(4);
    finish(out, it & 1);
    WSTAMP(5);
    if (!have_next) break;
  }
}

static int wino_enabled() {  // BTS_WINO=0: every 3x3x3 conv on the implicit GEMM (read per call: tests and A/B runs toggle it)
  const char* e = getenv("BTS_WINO");
  return e ? atoi(e) : 1;
}

struct WinoPlan {
  int xw, ntz, nty, ntx, nb, ksplit, kg_per;
  long wgs, need;
};
// tile geometry and k-split of one call; false = not a Winograd shape
static bool wino_plan(WinoPlan& q, int N, int D, int H, int W, int Cin, int Cout) {
  if (Cin % 8 != 0 || Cout % 4 != 0 || Cout < 16) return false;
  if (W < 12 || H < 4 || D < 4) return false;  // a narrower grid leaves most matrix columns of a wave empty
  // 16-wide tiles when they waste fewer columns (W <= 16, or W = 33..48 etc.)
  const int pad32 = (W + 31) / 32 * 32, pad16 = (W + 15) / 16 * 16;
  q.xw = (pad16 < pad32) ? 16 : 32;
  q.ntz = (D + 3) / 4;
  q.nty = (q.xw == 32) ? (H + 3) / 4 : (H + 7) / 8;
  q.ntx = (W + q.xw - 1) / q.xw;
  q.nb = (Cout + 31) / 32;
  q.wgs = (long)N * q.ntz * q.nty * q.ntx * q.nb;
  if (q.wgs > 0x7fffffffL / q.nb) return false;
  // one workgroup per CU at a time: a grid below ~one wave of workgroups is split along the input channels
  const int KG = Cin / 8;
  q.ksplit = 1;
  q.kg_per = KG;
  q.need = 0;
  if (q.wgs < 192 && KG >= 8) {
    int ks = (int)((256 + q.wgs - 1) / q.wgs);
    if (ks > KG / 4) ks = KG / 4;   // at least 4 stages per workgroup
    if (ks > 16) ks = 16;
    if (ks > 1) {
      const int per = (KG + ks - 1) / ks;
      ks = (KG + per - 1) / per;
      if (ks > 1) {
        q.ksplit = ks;
        q.kg_per = per;
        q.need = (long)ks * N * D * H * W * (q.nb * 32) * 4;
      }
    }
  }
  return true;
}
// workspace the Winograd form wants for a call (0 = none / not taken): lets the planner size the buffer for both forms
long bts_wino_workspace_(int N, int D, int H, int W, int Cin, int Cout) {
  WinoPlan q;
  if (!wino_enabled() || !wino_plan(q, N, D, H, W, Cin, Cout)) return 0;
  return q.need;
}

template <int XW>
static int wino_launch_cfg(const WinoParams& p, const WinoPlan& q, double flops, hipStream_t stream) {
  auto kern = wino_kernel<XW>;
  static bool attr_done = false;
  const size_t shmem = 2 * WinoGeo<XW>::BUF * sizeof(float) + 64 + 256;   // + GroupNorm partial exchange (8 doubles) + 2 x 32 bias values
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  // (p.up = image base + 27 x the padded (cin, cout) pairs: conv_igemm.hip's image layout; p.KG = Cin / 8, q.nb = cout blocks of 32)
  { const int e = bts_img_note_use_(p.up - 27L * ((long)p.KG * 2 * (q.nb * 32) * 4), 2u, stream); if (e != BTS_OK) return e; }
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(23, flops, stream);
  (void)hipGetLastError();
  const long wgs_per_xcd = ((long)p.tiles_per_xcd * q.nb + p.T - 1) / p.T;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8L * wgs_per_xcd), 1, q.ksplit), dim3(256), shmem, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// Returns BTS_OK when the launch was taken, 1 when declined (the caller runs the implicit GEMM), another code on error.
// up: the Winograd part of the K3S1 packed image.  gn_B: as launch_igemm's.  ws: split-K workspace (may be NULL).
int bts_wino_launch_(const float* x, const float* up, const float* bias, float* y, int N, int D, int H, int W, int Cin, int ldx,
                     int Cout, int ldy, int accum, double* gnp, int gnG, long* gn_B, void* ws, long ws_bytes, hipStream_t stream) {
  if (!wino_enabled()) return 1;
  if (ldx % 4 != 0 || ldy % 4 != 0) return 1;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 15)) return 1;
  if (((long)(D + 2) * H * W + 64) * (long)ldx * 4 >= 0x7fffffffL) return 1;  // 31-bit byte offsets inside one volume
  WinoPlan q;
  if (!wino_plan(q, N, D, H, W, Cin, Cout)) return 1;
  {  // the output side forms 31-bit byte offsets too: voxel index * (ldy, or the padded split-K row) * 4, 0x80000000 = masked lane
    const long orow = (long)ldy > (long)q.nb * 32 ? (long)ldy : (long)q.nb * 32;
    if (((long)D * H * W + 64) * orow * 4 >= 0x7fffffffL) return 1;
  }
  if (q.ksplit > 1 && (ws == nullptr || ws_bytes < q.need || (((uintptr_t)ws) & 15))) { q.ksplit = 1; q.kg_per = Cin / 8; }
  int min_wgs = 192;
  { const char* e = getenv("BTS_WINO_MIN_WGS"); if (e) min_wgs = atoi(e); }
  if (getenv("BTS_WINO_LOG"))
    fprintf(stderr, "wino %s N=%d D=%d H=%d W=%d Cin=%d Cout=%d ldx=%d ldy=%d accum=%d gn=%d xw=%d wgs=%ld ksplit=%d ws=%ld\n",
            q.wgs * q.ksplit < min_wgs ? "declined" : "taken", N, D, H, W, Cin, Cout, ldx, ldy, accum, gnp != nullptr, q.xw, q.wgs,
            q.ksplit, ws_bytes);
  if (q.wgs * q.ksplit < min_wgs) return 1;
  WinoParams p;
  p.x = x; p.up = up; p.bias = bias; p.y = y;
  p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy; p.KG = Cin / 8;
  p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx;
  p.accum = accum;
  p.ksplit = q.ksplit; p.kg_per = q.kg_per; p.Npad = q.nb * 32; p.part = reinterpret_cast<float*>(ws);
  p.nb = q.nb; p.ntiles = N * q.ntz * q.nty * q.ntx; p.tiles_per_xcd = (p.ntiles + 7) / 8;
  // items per workgroup: as many as leave every CU (32 per XCD, one workgroup each at a time) at least two workgroups
  {
    const long per_cu = ((long)p.tiles_per_xcd * q.nb) / 32;
    p.T = q.ksplit > 1 ? 1 : (per_cu >= 8 ? 4 : per_cu >= 4 ? 2 : 1);
    const char* e = getenv("BTS_WINO_T");   // A/B aid
    if (e && q.ksplit == 1) p.T = atoi(e) > 0 ? atoi(e) : 1;
  }
  p.gnp = nullptr; p.gn_G = 0; p.gn_zt = 1;
  if (q.ksplit == 1 && gnp != nullptr && gnG > 0 && D % gnG == 0 && (D / gnG) % 4 == 0 && getenv("BTS_IGEMM_NOGNFUSE") == nullptr) {
    p.gnp = gnp; p.gn_G = gnG; p.gn_zt = (D / gnG) / 4;
  }
  const double flops = 2.0 * 27 * Cin * Cout * (double)N * D * H * W;
  const int r = (q.xw == 32) ? wino_launch_cfg<32>(p, q, flops, stream) : wino_launch_cfg<16>(p, q, flops, stream);
  if (r != BTS_OK) return r;
  if (q.ksplit > 1) {
    const int rr = bts_igemm_reduce_(p.part, bias, y, (long)N * D * H * W, Cout, p.Npad, ldy, q.ksplit, bias != nullptr, accum, stream);
    if (rr != BTS_OK) return rr;
  }
  if (gn_B && p.gnp != nullptr) *gn_B = (long)p.gn_zt * q.nty * q.ntx * q.nb;
  return BTS_OK;
}
