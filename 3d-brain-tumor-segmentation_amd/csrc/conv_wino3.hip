// 3x3x3 stride-1 'same' convolution in Winograd F(2x2x2, 3x3x3) form -- all three axes transformed -- on the exact-fp32 matrix
// pipe of gfx950 (v_mfma_f32_32x32x2_f32): 8 matrix multiply-adds per output voxel and channel pair instead of 27 (the
// F(2x2, 3x3) x direct form of conv_wino.hip needs 12).  Same call sites as conv_wino.hip (reference: layers/resnet.py:30-37,
// 80-87,96-103  layers/decoder.py:55-63  layers/vae.py:92-99, and their data gradients on flipped, transposed weights).
//
//   Y = (A^T x A^T x A^T) [ ((G x G x G) g) o ((B^T x B^T x B^T) d) ]   per 2x2x2 output patch, summed over the input channels
//   B^T = (1 0 -1 0 | 0 1 1 0 | 0 -1 1 0 | 0 1 0 -1)   G = (1 0 0 | .5 .5 .5 | .5 -.5 .5 | 0 0 1)   A^T = (1 1 1 0 | 0 1 -1 -1)
//
// U = (G x G x G) g is formed at weight-packing time (conv_igemm.hip: third part of the K3S1 image, layout
// [cout block of 32][k-group of 8 cin][xi = (xi_z*4 + xi_y)*4 + xi_x][half h][32 couts][4 cin]).
//
// Work decomposition: 256 threads = 4 waves; workgroup tile = 16 (x) x 4 (y) x 4 (z) output voxels x 32 couts = 32 patches of
// 2x2x2 = the 32 matrix columns of every wave.  Wave a owns the 16 transform points with xi_z = a: 16 accumulators of
// 32 patches x 32 couts = 256 accumulation registers, one wave per SIMD.  The 18 x 6 x 6 halo tile of 8 input channels is staged
// global -> registers -> LDS (double buffered) with the z transform applied ON THE WAY: a staging thread holds the six z planes
// of its (y, x) column and writes the eight z-combined planes (2 z patches x 4 xi_z) -- every consumer lane then reads ONE
// plane and no wave repeats the z combination.  LDS: voxel stride 12 dwords; even and odd x columns in separate half rows so
// that lanes of neighbouring patches are 12 dwords apart; row stride 20 voxels puts the two y patches 32 banks apart:
// conflict-free ds_read_b128.  Lane (h, patch) reads the 4 x 4 voxels of its patch in plane (z patch, xi_z), channels
// h*4..h*4+3, and forms the (y, x) transform one group (one xi_y, four xi_x) ahead of the matrix instructions that consume it;
// U fragments stream from L2 two groups ahead.  The stage barrier sits before the stage's LAST group: the next
// stage's first operands are formed under that group's matrix instructions.
// Output: each wave transforms its 16 accumulators over (y, x), the four z partial results meet in LDS, wave w finishes and
// stores the outputs of (oy, ox) = w.
//
// Rounding: as conv_wino.hip (transforms add fp32 values before the multiply; fixed summation order, deterministic);
// measured error 1.35x the F(2x2,3x3) x direct form's.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "bts_internal.h"
#include "wino_util.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

#ifdef BTS_WINO_STAMPS   // experiment builds only (scripts/w3_timeline.py): clock stamps of wave 0, one row per workgroup
__device__ long long g_w3_stamps[1 << 20];
extern "C" int bts_w3_stamps_copy_(long long* dst, long n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_w3_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#define W3STAMP(slot)                                                                     \
  do {                                                                                    \
    if (tid == 0 && blockIdx.z == 0) g_w3_stamps[(long)blockIdx.x * 16 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define W3STAMP(slot) do { } while (0)
#endif

struct W3Params {
  const float* x;
  const float* up;
  const float* bias;
  float* y;
  int N, D, H, W, ldx, Cout, ldy, KG;
  int ntz, nty, ntx;
  int accum;
  double* gnp;  // fused GroupNorm partial sums (slab semantics), layout as igemm_kernel's
  int gn_G, gn_zt;
  int ksplit, kg_per, Npad;   // split-K as conv_wino.hip
  float* part;
  int nb, ntiles, tiles_per_xcd;
  int T;   // (tile, cout block) items one workgroup walks back to back
};

#define W3S 12                    // dwords per staged voxel: 8 channels + 4 pad
#define W3SX 18                   // staged halo columns
#define W3LX 20                   // LDS row stride (voxels): 10 even + 10 odd columns
#define W3NCOL (W3SX * 6 * 2)     // staging threads in use: one per (y, x) column and channel half (216)
#define W3PL (6 * W3LX * W3S)     // dwords per z-combined plane (1440)
#define W3BUF (8 * W3PL)          // dwords per staging buffer: 8 planes = (z patch, xi_z) (11520)
#define W3NSLOT 6                 // staging slots (16 bytes) per thread and stage = the six z planes of the thread's column
#define W3EX (4 * 4 * 4 * 64 * 4) // dwords of the output exchange [xi_z][cout quad][oy*2+ox][lane] x 4
// patch-relative LDS offset (dwords) of row j, column k of a plane
#define W3OFF(j, k) ((((j) * W3LX) + ((k) & 1) * 10 + ((k) >> 1)) * W3S)

__global__ __launch_bounds__(256, 1) void w3_kernel(const W3Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = xi_z of this wave
  const int h = lane >> 5, l32 = lane & 31;
  const int ptx = l32 & 7, pty = (l32 >> 3) & 1, ptz = l32 >> 4;   // this lane's 2x2x2 patch inside the tile
  // 1-D grid over (tile, cout block) ITEMS, dealt as conv_wino.hip deals them: workgroup ids round-robin over the 8 XCDs, XCD k
  // walks its own contiguous eighth of the tiles with the cout blocks of a tile back to back, one workgroup takes p.T
  // consecutive items.  The stages of consecutive items form ONE stream: halo tiles are requested two stages ahead and weight
  // fragments two groups ahead across item boundaries, so only a workgroup's first item waits for memory with the pipe idle.
  const int nb_ = p.nb;
  const int xcd_ = blockIdx.x & 7;
  const int seq0 = (blockIdx.x >> 3) * p.T;
  int tiles_left = p.ntiles - xcd_ * p.tiles_per_xcd;
  if (tiles_left > p.tiles_per_xcd) tiles_left = p.tiles_per_xcd;
  const int items_here = tiles_left * nb_;
  if (seq0 >= items_here) return;   // (before any barrier: whole workgroups leave)

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

  // staging pattern (tile independent): thread -> (y, x) column and channel half; slot i = z plane i of the halo
  const int s_q = tid & 1, s_vy = (tid >> 1) / W3SX, s_vx = (tid >> 1) - s_vy * W3SX;
  const bool s_on = tid < W3NCOL;
  const unsigned gcol = (unsigned)((s_vy * p.W + s_vx) * p.ldx + s_q * 4) * 4u;
  const unsigned gplane = (unsigned)(p.H * p.W * p.ldx) * 4u;
  const int lcol = (s_vy * W3LX + (s_vx & 1) * 10 + (s_vx >> 1)) * W3S + s_q * 4;
  __amdgpu_buffer_rsrc_t xr, wr, wr_n;
  unsigned goff[W3NSLOT];
  auto setup_x = [&](const Item& it) {
    const int iz0 = it.tz * 4 - 1, iy0 = it.ty * 4 - 1, ix0 = it.tx * 16 - 1;
    const float* xorg = p.x + ((((long)it.n * p.D + iz0) * p.H + iy0) * p.W + ix0) * (long)p.ldx;
    xr = __builtin_amdgcn_make_buffer_rsrc((void*)xorg, 0, 0x7fffffff, 0x00020000);
    const bool col_in = s_on && (unsigned)(iy0 + s_vy) < (unsigned)p.H && (unsigned)(ix0 + s_vx) < (unsigned)p.W;
#pragma unroll
    for (int i = 0; i < W3NSLOT; ++i) goff[i] = (col_in && (unsigned)(iz0 + i) < (unsigned)p.D) ? gcol + (unsigned)i * gplane : 0x80000000u;
  };
  auto wdesc = [&](const Item& it) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.up + (long)it.cb * p.KG * (64 * 256)), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t br =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (p.bias && p.ksplit <= 1) ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
  auto bias_of = [&](const Item& it) {   // (one request on every path; out of range -> 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, tid < 32 ? (unsigned)(it.cb * 32 + tid) * 4u : 0x80000000u, 0, 0));
  };

  int st0 = 0, st1 = p.KG;
  if (p.ksplit > 1) {
    st0 = blockIdx.z * p.kg_per;
    st1 = st0 + p.kg_per;
    if (st1 > p.KG) st1 = p.KG;
  }

  // this lane's operand plane (z patch, xi_z = wave) and patch origin inside it
  const int offP = (ptz * 4 + wave) * W3PL + (2 * pty * W3LX + ptx) * W3S + h * 4;
  const unsigned wlane = (unsigned)((h * 32 + l32) * 16);
  const unsigned wwave = (unsigned)(wave * 16 * 1024);

  f32x16 acc[4][4];   // [xi_y][xi_x]
  f32x4 pre[W3NSLOT];
  // soff = 0x80000000: nothing left to fetch -- every slot is then out of range (zeros, no traffic); the requests are issued all
  // the same, so that the request count is the same on every path and the compiler's vmcnt waits stay exact
  auto fetch = [&](unsigned soff) {
#pragma unroll
    for (int i = 0; i < W3NSLOT; ++i) pre[i] = bufload(xr, goff[i], soff);
  };
  // z transform on the way to LDS: plane (zp, xi_z) = xi_z 0: d0 - d2 | 1: d1 + d2 | 2: d2 - d1 | 3: d1 - d3 of the raw planes 2 zp + (0..3)
  auto commit = [&](float* buf) {
    if (s_on) {
      float* o = buf + lcol;
#pragma unroll
      for (int zp = 0; zp < 2; ++zp) {
        const f32x4 d0 = pre[2 * zp], d1 = pre[2 * zp + 1], d2 = pre[2 * zp + 2], d3 = pre[2 * zp + 3];
        *reinterpret_cast<f32x4*>(o + (zp * 4 + 0) * W3PL) = sub4(d0, d2);
        *reinterpret_cast<f32x4*>(o + (zp * 4 + 1) * W3PL) = add4(d1, d2);
        *reinterpret_cast<f32x4*>(o + (zp * 4 + 2) * W3PL) = sub4(d2, d1);
        *reinterpret_cast<f32x4*>(o + (zp * 4 + 3) * W3PL) = sub4(d1, d3);
      }
    }
  };
  // U fragments of group G = 4*stage + gi live in aw[gi]; the groups of a stage run xi_y = 1, 2, 0, 3
  f32x4 aw[4][4];
  auto wload = [&](f32x4 (&dst)[4], const __amdgpu_buffer_rsrc_t d, int st, int b) {
#ifdef W3_EXP_SAMEU   // timing experiment (wrong results): every fragment request hits the same 16 KB
    const unsigned so = (unsigned)(b * 4096) + wwave + 0u * st;
#else
    const unsigned so = (unsigned)(st * (64 * 1024) + b * 4096) + wwave;
#endif
#ifdef W3_EXP_NOW     // timing experiment (wrong results): no weight fragment requests
    return;
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e] = bufload(d, wlane + e * 1024, so);
  };
  f32x4 c[4][4], v[2][4], t[4];
  auto rd_row = [&](const float* lb, int j) {   // row j of the lane's patch: 4 voxels x 4 channels, already z-combined
#ifdef W3_EXP_NOLDS   // timing experiment (wrong results): no operand reads from LDS
    return;
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) c[j][k] = *reinterpret_cast<const f32x4*>(lb + offP + W3OFF(j, k));
  };
  float* const ex = lds + 2 * W3BUF;
  float* const bsh = ex + W3EX;                                  // 2 x 32 bias values (current / next item)
  double* const gsh = reinterpret_cast<double*>(bsh + 64);       // GroupNorm partial exchange
  float bias_v = 0.f;

  // One k-group stage on LDS buffer `par`.  On entry c[1], c[2] hold this stage's rows 1, 2 and v[0] the operands of
  // xi_y = 1.  FIRST: the item's first stage (accumulators start from zero; the item's bias goes to LDS slot `bslot`).
  // f_soff: the halo tile two stages ahead in the workgroup's stage stream (xr / goff already belong to its item).
  // wt / stn: weight descriptor and k-group of the NEXT stage of the stream.
  auto stage = [&](auto first_tag, int st, int par, unsigned f_soff, const __amdgpu_buffer_rsrc_t wt, int stn, int bslot) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const float* lb = lds + par * W3BUF;
    const float* ln = lds + (par ^ 1) * W3BUF;
    {  // MFMA xi_y = 1 ; form xi_y = 2 : c2 - c1 ; request row 0
      wload(aw[2], wr, st, 0);
      rd_row(lb, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = sub4(c[2][k], c[1][k]);
      wino_yt(t, v[1]);
      wino_mfma16<FIRST>(v[0], aw[0], acc[1]);
      WINO_SCHED_GROUP();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1" ::: "memory");
    }
    {  // MFMA xi_y = 2 ; form xi_y = 0 : c0 - c2 ; request row 3
      wload(aw[3], wr, st, 3);
      rd_row(lb, 3);
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = sub4(c[0][k], c[2][k]);
      wino_yt(t, v[0]);
      wino_mfma16<FIRST>(v[1], aw[1], acc[2]);
      WINO_SCHED_GROUP();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1" ::: "memory");
    }
    {  // MFMA xi_y = 0 ; form xi_y = 3 : c1 - c3 ; hand the next halo tile to LDS (z transform on the way)
      wload(aw[0], wt, stn, 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = sub4(c[1][k], c[3][k]);
      wino_yt(t, v[1]);
      wino_mfma16<FIRST>(v[0], aw[2], acc[0]);
      WINO_SCHED_GROUP();
      __builtin_amdgcn_sched_barrier(0);
      commit(lds + (par ^ 1) * W3BUF);
      if (FIRST && tid < 32) bsh[bslot * 32 + tid] = bias_v;
#ifndef W3_EXP_NOBAR   // timing experiment (wrong results): no stage barrier
      __syncthreads();
#endif
#ifdef W3_EXP_NOFETCH  // timing experiment (wrong results): no halo traffic after the first tile
      fetch(0x80000000u);
#else
      fetch(f_soff);
#endif
    }
    {  // MFMA xi_y = 3 ; read rows 1, 2 of the NEXT stage and form its xi_y = 1 : c1 + c2
      wload(aw[1], wt, stn, 2);
      rd_row(ln, 1);
      rd_row(ln, 2);
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = add4(c[1][k], c[2][k]);
      wino_yt(t, v[0]);
      wino_mfma16<FIRST>(v[1], aw[3], acc[3]);
      WINO_SCHED_GROUP();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1" ::: "memory");
    }
  };

  // ---- output: (y, x) transform per wave, z combination through LDS, bias, optional accumulate, store ----
  const bool raw = p.ksplit > 1;
  const int ld = raw ? p.Npad : p.ldy;
  const int clim = raw ? p.Npad : p.Cout;
  const bool acc_in = p.accum && !raw;
  auto finish = [&](const Item& o, int bslot) {
    const float* obase = raw ? p.part + ((long)blockIdx.z * p.N + o.n) * p.D * p.H * p.W * (long)p.Npad
                             : p.y + (long)o.n * p.D * p.H * p.W * (long)p.ldy;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, 0x7fffffff, 0x00020000);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // one register quad (4 couts) of all 16 accumulators at a time
      f32x4 tq[4][2];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          q[k] = f32x4{acc_rd(acc[b][k][4 * g]), acc_rd(acc[b][k][4 * g + 1]), acc_rd(acc[b][k][4 * g + 2]), acc_rd(acc[b][k][4 * g + 3])};
        tq[b][0] = add4(add4(q[0], q[1]), q[2]);
        tq[b][1] = sub4(sub4(q[1], q[2]), q[3]);
      }
#pragma unroll
      for (int ox = 0; ox < 2; ++ox) {
        const f32x4 m0 = add4(add4(tq[0][ox], tq[1][ox]), tq[2][ox]);
        const f32x4 m1 = sub4(sub4(tq[1][ox], tq[2][ox]), tq[3][ox]);
        *reinterpret_cast<f32x4*>(ex + (((wave * 4 + g) * 4 + 0 + ox) * 64 + lane) * 4) = m0;
        *reinterpret_cast<f32x4*>(ex + (((wave * 4 + g) * 4 + 2 + ox) * 64 + lane) * 4) = m1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    W3STAMP(4);
    f32x2 gn_s2 = {0.f, 0.f}, gn_q2 = {0.f, 0.f};
    const bool gn_on = p.gnp != nullptr;
    {
      const int oy = wave >> 1, ox = wave & 1;
      const int zb = o.tz * 4 + 2 * ptz, yy = o.ty * 4 + 2 * pty + oy, xx = o.tx * 16 + 2 * ptx + ox;
      unsigned yo[2];
#pragma unroll
      for (int oz = 0; oz < 2; ++oz) {
        const bool ok = xx < p.W && yy < p.H && (zb + oz) < p.D;
        yo[oz] = ok ? (unsigned)(((((zb + oz) * p.H + yy) * p.W + xx) * ld + o.cb * 32 + 4 * h) * 4) : 0x80000000u;
      }
      // all reads of the old output (accumulate; out of range -> zeros: one request on every path) BEFORE the first store:
      // vector memory returns in order, a read behind a store would wait for the store to complete
      f32x4 old[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int oz = 0; oz < 2; ++oz) {
          const bool cok = o.cb * 32 + 8 * g + 4 * h < clim;
          old[g][oz] = bufload(yr, (acc_in && cok) ? yo[oz] + 32u * g : 0x80000000u, 0);
        }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 m[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) m[a] = *reinterpret_cast<const f32x4*>(ex + (((a * 4 + g) * 4 + wave) * 64 + lane) * 4);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bsh + bslot * 32 + 8 * g + 4 * h);
        f32x4 ov[2];
        ov[0] = add4(add4(add4(m[0], m[1]), m[2]), bq);
        ov[1] = add4(sub4(sub4(m[1], m[2]), m[3]), bq);
        const bool cok = o.cb * 32 + 8 * g + 4 * h < clim;
#pragma unroll
        for (int oz = 0; oz < 2; ++oz) {
          const unsigned off = cok ? yo[oz] : 0x80000000u;
          const bool live = off != 0x80000000u;
          f32x4 w = ov[oz];
          if (gn_on) {   // (wave-uniform) masked positions contribute nothing to the GroupNorm sums
            const f32x4 tt = live ? w : f32x4{0.f, 0.f, 0.f, 0.f};
            gn_s2 = pk_add(pk_add(gn_s2, tt.xy), tt.zw);
            gn_q2 = pk_fma(tt.zw, tt.zw, pk_fma(tt.xy, tt.xy, gn_q2));
          }
          w = add4(w, old[g][oz]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), yr, off + 32u * g, 0, 0);
        }
      }
    }
    {  // GroupNorm partials: lanes (shuffle tree) -> 4 waves (LDS) -> one (sum, sumsq) pair per item, fixed order; thread 0
       // stores (the store is issued on every path -- out of range when there is no consumer -- the arithmetic is not)
      double s0 = 0.0, s1 = 0.0;
      const double* dst = nullptr;
      if (gn_on) {   // (launch-uniform)
        const double ds = wave_sum_f64((double)gn_s2.x + (double)gn_s2.y), dq = wave_sum_f64((double)gn_q2.x + (double)gn_q2.y);
        if (lane == 0) { gsh[wave * 2] = ds; gsh[wave * 2 + 1] = dq; }
        __syncthreads();
        const int gn_g = o.tz / p.gn_zt;
        const long B = (long)p.gn_zt * p.nty * p.ntx * nb_;
        const long gn_slot = (((long)(o.tz - gn_g * p.gn_zt) * p.nty + o.ty) * p.ntx + o.tx) * nb_ + o.cb;
        dst = p.gnp + (((long)o.n * p.gn_G + gn_g) * B + gn_slot) * 2;
        s0 = gsh[0] + gsh[2] + gsh[4] + gsh[6];
        s1 = gsh[1] + gsh[3] + gsh[5] + gsh[7];
      }
      const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, gn_on ? 16 : 0, 0x00020000);
      struct D2 { double a, b; } d2{s0, s1};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d2), gr, (gn_on && tid == 0) ? 0u : 0x80000000u, 0, 0);
    }
  };

  W3STAMP(0);
#ifdef BTS_WINO_STAMPS
  if (tid == 0 && blockIdx.z == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_w3_stamps[(long)blockIdx.x * 16 + 6] = ((long long)xcc << 32) | hwid;
    g_w3_stamps[(long)blockIdx.x * 16 + 7] = clock64();
  }
#endif
  // ---- prologue: first halo tile, first operands of the workgroup's first item ----
  Item cur = decode(seq0), nxt = cur;
  setup_x(cur);
  wr = wdesc(cur);
  wr_n = wr;
  fetch((unsigned)st0 * 32u);
  wload(aw[0], wr, st0, 1);
  wload(aw[1], wr, st0, 2);
  bias_v = bias_of(cur);
  {  // as many (dropped) requests as an item's output side issues: the first item's waits then match the later items'
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0, 0x00020000);
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) z += bufload(none, 0x80000000u + 16u * i, 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, z), none, 0x80000000u + 16u * i, 0, 0);
    const float dummy = bias_of(cur);   // (stands for the successor's bias request of a later item)
    asm volatile("" ::"v"(dummy));
  }
  commit(lds);
  __syncthreads();
  fetch(st0 + 1 < st1 ? (unsigned)(st0 + 1) * 32u : 0x80000000u);
  rd_row(lds, 1);
  rd_row(lds, 2);
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = add4(c[1][k], c[2][k]);
  wino_yt(t, v[0]);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 3" ::: "memory");
  W3STAMP(1);

  int par = 0;
  for (int it = 0; it < p.T; ++it) {
    const bool have_next = (it + 1 < p.T) && (seq0 + it + 1 < items_here);
    const Item out = cur;
    float bias_n = 0.f;
    if (have_next) {
      advance(nxt);
      wr_n = wdesc(nxt);
    }
    bias_n = bias_of(nxt);
    // stage s of the item requests the halo tile of stream position s + 2 and the weights of position s + 1
    auto run = [&](auto first_tag, int st) {
      if (st + 2 == st1 && have_next) setup_x(nxt);   // the halo requests cross into the successor here
      const int f = st + 2;
      const unsigned f_soff = f < st1 ? (unsigned)f * 32u : (have_next ? (unsigned)(st0 + f - st1) * 32u : 0x80000000u);
      const bool last = st + 1 == st1;
      stage(first_tag, st, par, f_soff, last ? wr_n : wr, last ? st0 : st + 1, it & 1);
      par ^= 1;
    };
    run(std::true_type{}, st0);
    if (it == 0) W3STAMP(2);
    for (int st = st0 + 1; st < st1; ++st) run(std::false_type{}, st);
    if (it == 0) W3STAMP(3);
    finish(out, it & 1);
    if (!have_next) break;
    cur = nxt;
    wr = wr_n;
    bias_v = bias_n;
  }
  W3STAMP(5);
#ifdef BTS_WINO_STAMPS
  if (tid == 0 && blockIdx.z == 0) g_w3_stamps[(long)blockIdx.x * 16 + 8] = clock64();
#endif
}

static int w3_enabled() {  // BTS_WINO=0: no Winograd form at all; BTS_W3=0: this one off, conv_wino.hip's F(2x2,3x3) x direct form stays
  const char* e = getenv("BTS_WINO");   // (read per call: tests and A/B runs toggle them)
  if (e && atoi(e) == 0) return 0;
  e = getenv("BTS_W3");
  return e ? atoi(e) : 1;
}

struct W3Plan {
  int ntz, nty, ntx, nb, ksplit, kg_per;
  long wgs, need;
};
static bool w3_plan(W3Plan& q, int N, int D, int H, int W, int Cin, int Cout) {
  if (Cin % 8 != 0 || Cout % 4 != 0 || Cout < 16) return false;
  if (W < 12 || H < 4 || D < 4) return false;
  q.ntz = (D + 3) / 4;
  q.nty = (H + 3) / 4;
  q.ntx = (W + 15) / 16;
  q.nb = (Cout + 31) / 32;
  q.wgs = (long)N * q.ntz * q.nty * q.ntx * q.nb;
  if (q.wgs > 0x7fffffffL / 16) return false;
  const int KG = Cin / 8;
  q.ksplit = 1;
  q.kg_per = KG;
  q.need = 0;
  if (q.wgs < 384 && KG >= 8) {
    int ks = (int)((512 + q.wgs - 1) / q.wgs);
    if (ks > KG / 4) ks = KG / 4;
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
long bts_w3_workspace_(int N, int D, int H, int W, int Cin, int Cout) {
  W3Plan q;
  if (!w3_enabled() || !w3_plan(q, N, D, H, W, Cin, Cout)) return 0;
  return q.need;
}

// Returns BTS_OK when the launch was taken, 1 when declined (the caller offers conv_wino.hip, then the implicit GEMM).
// up3: the third part of the K3S1 packed image.
int bts_w3_launch_(const float* x, const float* up3, const float* bias, float* y, int N, int D, int H, int W, int Cin, int ldx,
                   int Cout, int ldy, int accum, double* gnp, int gnG, long* gn_B, void* ws, long ws_bytes, hipStream_t stream) {
  if (!w3_enabled()) return 1;
  if (ldx % 4 != 0 || ldy % 4 != 0) return 1;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 15)) return 1;
  if (((long)(D + 2) * H * W + 64) * (long)ldx * 4 >= 0x7fffffffL) return 1;
  W3Plan q;
  if (!w3_plan(q, N, D, H, W, Cin, Cout)) return 1;
  {  // the output side forms 31-bit byte offsets too: voxel index * (ldy, or the padded split-K row) * 4, 0x80000000 = masked lane
    const long orow = (long)ldy > (long)q.nb * 32 ? (long)ldy : (long)q.nb * 32;
    if (((long)D * H * W + 64) * orow * 4 >= 0x7fffffffL) return 1;
  }
  if (q.ksplit > 1 && (ws == nullptr || ws_bytes < q.need || (((uintptr_t)ws) & 15))) { q.ksplit = 1; q.kg_per = Cin / 8; }
  int min_wgs = 192;
  { const char* e = getenv("BTS_WINO_MIN_WGS"); if (e) min_wgs = atoi(e); }
  if (q.wgs * q.ksplit < min_wgs) return 1;
  W3Params p;
  p.x = x; p.up = up3; p.bias = bias; p.y = y;
  p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy; p.KG = Cin / 8;
  p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx;
  p.accum = accum;
  p.ksplit = q.ksplit; p.kg_per = q.kg_per; p.Npad = q.nb * 32; p.part = reinterpret_cast<float*>(ws);
  p.nb = q.nb; p.ntiles = N * q.ntz * q.nty * q.ntx; p.tiles_per_xcd = (p.ntiles + 7) / 8;
  p.gnp = nullptr; p.gn_G = 0; p.gn_zt = 1;
  if (q.ksplit == 1 && gnp != nullptr && gnG > 0 && D % gnG == 0 && (D / gnG) % 4 == 0 && getenv("BTS_IGEMM_NOGNFUSE") == nullptr) {
    p.gnp = gnp; p.gn_G = gnG; p.gn_zt = (D / gnG) / 4;
  }
  static bool attr_done = false;
  const size_t shmem = (2 * W3BUF + W3EX + 64 + 16) * sizeof(float);
  // items per workgroup (chaining needs two stages per item): each XCD runs 32 workgroups at a time (one per CU), so a launch
  // takes ceil(workgroups per XCD / 32) rounds of T items; the T <= 8 with the fewest item-times wins, ties go to the larger T
  // (fewer un-overlapped first fetches)
  {
    const long items = (long)p.tiles_per_xcd * q.nb;
    p.T = 1;
    if (q.ksplit == 1 && p.KG >= 2) {
      long best = -1;
      for (int t = 1; t <= 8; ++t) {
        const long rounds = ((items + t - 1) / t + 31) / 32;
        const long cost = rounds * (8L * t + 1);   // (+1: the first item of a round waits for its fetch)
        if (best < 0 || cost <= best) { best = cost; p.T = t; }
      }
      const char* e = getenv("BTS_W3_T");   // A/B aid
      if (e && atoi(e) > 0) p.T = atoi(e);
    }
  }
  const long wgs_per_xcd = ((long)p.tiles_per_xcd * q.nb + p.T - 1) / p.T;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(w3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  // (up3 = image base + (27 + 48) x the padded (cin, cout) pairs: conv_igemm.hip's image layout)
  { const int e = bts_img_note_use_(up3 - 75L * ((long)((Cin + 7) / 8) * 2 * (((Cout + 31) / 32) * 32) * 4), 4u, stream); if (e != BTS_OK) return e; }
  const double flops = 2.0 * 27 * Cin * Cout * (double)N * D * H * W;
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(27, flops, stream);
  (void)hipGetLastError();
  hipLaunchKernelGGL(w3_kernel, dim3((unsigned)(8L * wgs_per_xcd), 1, q.ksplit), dim3(256), shmem, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  if (q.ksplit > 1) {
    const int rr = bts_igemm_reduce_(p.part, bias, y, (long)N * D * H * W, Cout, p.Npad, ldy, q.ksplit, bias != nullptr, accum, stream);
    if (rr != BTS_OK) return rr;
  }
  if (gn_B && p.gnp != nullptr) *gn_B = (long)p.gn_zt * q.nty * q.ntx * q.nb;
  return BTS_OK;
}
