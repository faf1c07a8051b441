// Stride-1 3x3x3 convolution on 16-bit storage for the layers with FEW channels (Cin 16 or 32, Cout <= 32: the 128^3 level of the CLI
// model -- resnet.py:80-87, vae.py:92-99 and their data gradients under train.py:142-151) as a z-marching streaming kernel.
//
// At 32 -> 32 channels the layer moves 128 operand bytes per 55 k flops: 432 flop/B against a machine balance of ~500 -- it is bound by
// reading the input and writing the output ONCE.  lowp_s1d.hip's tiled kernel re-reads the input 2.0x (34 x 10 x 6 halo of a 32 x 8 x 4
// tile), re-streams the weights per tile and drains its pipeline after two k-steps: 0.46 PFLOP/s on these launches inside the batch-8
// step.  Here
//   * a workgroup owns a 32 (x) x 16 (y) column and marches along z: ONE input plane (34 x 18 voxels, all channels) per stage -- the
//     x/y halo (1.2x) is the only re-read;
//   * all 27 x Cin/16 weight fragments stay in LDS for the whole launch (27 or 54 KB);
//   * an input plane feeds the three output planes z-1, z, z+1: three accumulator sets per wave (2 rows x 32 positions x 32 couts each)
//     change roles every stage; the completed plane leaves after its third input plane (bias-initialised accumulators, 16-byte stores
//     of 8 consecutive couts after a v_permlane32_swap, optional accumulate, fused GroupNorm partial sums), then restarts at the bias;
//   * planes go global -> LDS by buffer_load ... lds one stage ahead (inline assembly: see lowp_wgd.hip for why), in lowp_s1d.hip's
//     conflict-free [voxel][2 x 8 ch] layout; fragments are plain ds_read_b128 at immediate offsets.
// The weight image is the DMA part of the K3S1 image (lowp_s1d.hip's pack: [k-step][dz][dy*3+dx][k-half][32 couts][8 cin]).
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
template <int V> using S1zIC = std::integral_constant<int, V>;

struct LpS1zParams {
  const unsigned short* x;
  const unsigned short* wp;
  const float* bias;
  unsigned short* y;
  int N, D, H, W, ldx, ldy, Cout;
  int ntx, nty, nzc, ZC, nitems, ipw, xcd_order, accum;
  double* gnp;       // fused GroupNorm partial sums (slab semantics) [N*G][gn_B][2], or NULL
  int gn_G, gn_zt;
  int gn_run;        // 1: a wave's sums run over all planes of a (group, item) before they leave (z chunks nest with the groups); 0: one pair per plane
  long gn_B;
  LpGnbFuse gb;      // GNB kernels: GroupNorm-backward class sums of the stored output against the GroupNorm input gb.x (lowp_common.h)
  int gb_zt;         // planes per group
  LpGnaFuse ga;      // GNA kernels: GroupNorm (+ReLU) applied to the input planes in LDS (lowp_common.h)
  int ga_zt;         // planes per group of the input's GroupNorm
  // SC kernels: a second contraction at the CENTRE tap only -- y += x2 (N,D,H,W,Cin) . w2, the 1x1x1 image [k-step][k-half][32][8] (first
  // part of a K1 image with one cout block): the shortcut conv's data gradient riding on conv1's (resnet.py:96-103 / 80-87 under train.py:151)
  const unsigned short* x2;
  const unsigned short* wp2;
  int ldx2;
  // FS kernels (forward): a second set of OUTPUT columns at the centre tap only -- y2 = x . w2 + bias2, the block's 1x1x1 shortcut conv
  // (resnet.py:96-103,118: it reads the same `inputs` as conv1, resnet.py:134) from the input planes this kernel already holds in LDS, and
  // the column sums of its unrounded output (the gate's squeeze, resnet.py:121) as fp64 partial rows [N][fs_B][Cout2].  wp2 as above
  // (first part of the K1 forward image, one cout block).
  unsigned short* y2;
  const float* bias2;
  double* gap_part;          // (may be NULL: the first of two passes over the halves of a 64-channel input leaves no sums)
  int ldy2, Cout2, fs_B;
  int accum2;                // y2 += (second pass: the column sums are those of the final values)
};
#define S1Z_TX 32
#define S1Z_TY 16
#define S1Z_SX 34
#define S1Z_SY 18
#define S1Z_NCHK 20          // 1 KB chunks (32 voxels x 32 B) of one k-step of a plane: 612 voxels -> 20

__device__ __forceinline__ void s1z_dma16(u32x4 rsrc, unsigned lds_byte, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff) : "memory");   // (m0 is a reserved register: the compiler sets it right before each of its own uses, never across statements)
}
__device__ __forceinline__ u32x4 s1z_rsrc(const void* base) {
  const unsigned long a = (unsigned long)base;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
}

template <typename T, int KS, bool GNB = false, bool GNA = false, int XC = 0>      // XC: 0 plain, 1 SC (extra K-segment), 2 FS (extra output columns)
__global__ __launch_bounds__(512, 1) void lp_s1z_kernel(const LpS1zParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool SC = XC == 1, FS = XC == 2;
  constexpr int SX = S1Z_SX, SY = S1Z_SY, NVOX = SX * SY, NCHK = S1Z_NCHK;
  constexpr int WB = 27 * KS * 1024, PLB = NCHK * 1024 * KS;
  constexpr int OFF_P = WB, OFF_BIAS = WB + 2 * PLB, OFF_SCR = OFF_BIAS + 256;
  constexpr int OFF_BIAS2 = OFF_BIAS + 128;   // FS: the shortcut's bias (32 floats)
  constexpr int OFF_GB = OFF_SCR + 1024;      // GNB: gamma'[32] | beta'[32] | mean[32] | rstd[32] of the current sample (floats); GNA: gamma | beta | mean | rstd
#ifdef S1Z_EXP_2ISSUE   // timing experiment: two waves issue all plane requests of a stage (is request back-pressure what stalls the others?)
  constexpr int NREQ = NCHK * KS, NR = NREQ / 2;
#define S1Z_ID(j) ((wave < 2) ? (wave * NR + (j)) : 0x7fff)
#else
  constexpr int NREQ = NCHK * KS, NR = (NREQ + 7) / 8;       // plane requests per stage / per wave
#define S1Z_ID(j) ((j) * 8 + wave)
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;

  // ---- DMA side ----
  u32x4 xr;
  unsigned voff[NR];
  unsigned xplane;       // bytes per input plane
  int zlo = 0, zhi = 0, cn = 0, cx0 = 0, cy0 = 0;
  auto setup = [&](int item) {
    int b = item;
    const int zc = b % p.nzc; b /= p.nzc;
    const int tx = b % p.ntx; b /= p.ntx;
    const int ty = b % p.nty;
    cn = b / p.nty;
    cx0 = tx * S1Z_TX; cy0 = ty * S1Z_TY;
    zlo = zc * p.ZC;
    zhi = zlo + p.ZC;
    if (zhi > p.D) zhi = p.D;
    const long xvox = ((long)cn * p.D * p.H + (cy0 - 1)) * p.W + (cx0 - 1);     // (may lie before the sample: such lanes are masked)
    xr = s1z_rsrc(p.x + xvox * p.ldx);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int id = S1Z_ID(j);
      unsigned v = 0x80000000u;
      if (id < NREQ) {
        const int ks = id / NCHK, chunk = id - ks * NCHK;
        const int vox = chunk * 32 + (lane >> 1);
        const int vy = vox / SX, vx = vox - vy * SX;
        const int hp = (lane & 1) ^ ((vx >> 3) & 1);
        if (vox < NVOX && (unsigned)(cx0 - 1 + vx) < (unsigned)p.W && (unsigned)(cy0 - 1 + vy) < (unsigned)p.H)
          v = (unsigned)(((vy * p.W + vx) * p.ldx + ks * 16 + hp * 8) * 2);
      }
      voff[j] = v;
    }
  };
  // GNA: the 16-byte slots of a plane buffer this thread transforms (slot tid + 512 i) and which of them hold a voxel of the image (the
  // others were zero-filled by the DMA and must stay zero: 'same' padding is applied AFTER the normalisation)
  constexpr int NSLOT = KS * NCHK * 64, NSL = (NSLOT + 511) / 512;
  unsigned ga_ok = 0;
  int ga_g = -1;                 // group whose parameters sit in the registers below
  float ga_mu = 0.f, ga_sc[8], ga_be[8];
  auto ga_setup = [&]() {
    ga_ok = 0;
    ga_g = -1;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int id = tid + 512 * i;
      const int chunk = (id >> 6) % NCHK, vox = chunk * 32 + ((id & 63) >> 1);
      const int vy = vox / SX, vx = vox - vy * SX;
      if (id < NSLOT && vox < NVOX && (unsigned)(cx0 - 1 + vx) < (unsigned)p.W && (unsigned)(cy0 - 1 + vy) < (unsigned)p.H) ga_ok |= 1u << i;
    }
  };
  auto ga_apply = [&](int buf, int z) {      // plane z has landed in buffer buf and nobody reads it yet
    if (z < 0 || z >= p.D) return;           // (a zero plane outside the volume)
    const int gq = z / p.ga_zt;
    if (gq != ga_g) {                        // (uniform: the march entered another group)
      const float* gsh = reinterpret_cast<const float*>(lds + OFF_GB);
      const int cm = p.ga.cg - 1;
      ga_g = gq;
      ga_mu = gsh[64 + gq];
      const float rs = gsh[96 + gq];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ga_sc[e] = rs * gsh[gq * p.ga.cg + (e & cm)];
        ga_be[e] = gsh[32 + gq * p.ga.cg + (e & cm)];
      }
    }
    const float mu = ga_mu;
    const float (&sc)[8] = ga_sc;
    const float (&be)[8] = ga_be;
    // all loads first (one LDS latency), the arithmetic of bts_lp_gn_apply, slots outside the image written back as the zeros they were
    u32x4 raw[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i)
      if (tid + 512 * i < NSLOT) raw[i] = *reinterpret_cast<const u32x4*>(lds + OFF_P + buf * PLB + (tid + 512 * i) * 16);
#ifdef S1Z_EXP_GNA_NOLDS      // timing experiment (wrong results): the transform without its arithmetic and stores
    return;
#endif
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      if (tid + 512 * i < NSLOT) {
        u32x4 r = lp_gna_slot<T>(raw[i], mu, sc, be);
        const bool ok = (ga_ok >> i) & 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = ok ? r[k] : 0u;
        *reinterpret_cast<u32x4*>(lds + OFF_P + buf * PLB + (tid + 512 * i) * 16) = r;
      }
    }
  };
  auto issue1 = [&](int j, int zp, int buf) {      // request j of input plane zp -> plane buffer buf (out of the volume: zeros)
    const int id = S1Z_ID(j);
    const bool pok = zp >= 0 && zp < p.D;
    if (id < NREQ) s1z_dma16(xr, lds0 + (unsigned)(OFF_P + buf * PLB + id * 1024), pok ? voff[j] : 0x80000000u, pok ? (unsigned)zp * xplane : 0u);
    else s1z_dma16(xr, lds0 + (unsigned)OFF_SCR, 0x80000000u, 0u);
  };
  auto issue_filler = [&]() { s1z_dma16(xr, lds0 + (unsigned)OFF_SCR, 0x80000000u, 0u); };      // (keeps the request counts of all stages equal)

  // ---- compute side ----
  const int r0 = 2 * wave;      // this wave's two output rows inside the column
  unsigned hbB[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) hbB[dx] = (unsigned)(OFF_P + (r0 * SX + l32 + dx) * 32 + ((h ^ (((l32 + dx) >> 3) & 1)) * 16));
  const unsigned wbA = (unsigned)(h * 512 + l32 * 16);
  f32x16 acc[3][2];
  const float* bsh = reinterpret_cast<const float*>(lds + OFF_BIAS) + 4 * h;
  auto init_set = [&](int s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(bsh + 8 * q);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[s][r][4 * q + j] = bq[j];
    }
  };
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7fffffff, 0x00020000);
  // SC: the operand pair of the centre tap, global -> registers (a lane's B fragment is its own voxel's 16 bytes; the weights stay in
  // registers for the whole launch).  The rows of plane z + 1 are requested inside stage z (after its plane requests) and multiplied at
  // the head of stage z + 1, into the accumulator set of that stage's centre tap.
  u32x4 A2[SC ? KS : 1], B2[SC ? KS : 1][2];
  __amdgpu_buffer_rsrc_t x2r = yr;
  unsigned sc_off[2] = {0x80000000u, 0x80000000u}, x2plane = 0;
  auto sc_request = [&](int z) {
    if constexpr (SC) {
      const bool pok = z >= zlo && z < zhi;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          B2[ks][r] = __builtin_amdgcn_raw_buffer_load_b128(x2r, pok ? sc_off[r] : 0x80000000u, pok ? (unsigned)z * x2plane + (unsigned)ks * 32u : 0u, 0);
    }
  };
  // FS: the shortcut's accumulators of the CENTRE output plane (two rows), its weights (registers, whole launch) and the per-lane column
  // sums of the item (lane (h, voxel l32) holds couts 8 q + 4 h + j at index 4 q + j)
  f32x16 acc2[FS ? 2 : 1];
  u32x4 Apt[FS ? KS : 1];
  float gsum[FS ? 16 : 1];
  __amdgpu_buffer_rsrc_t y2r = yr;
  auto init_res = [&]() {
    if constexpr (FS) {
      const float* b2 = reinterpret_cast<const float*>(lds + OFF_BIAS2) + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(b2 + 8 * q);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc2[r][4 * q + j] = bq[j];
      }
    }
  };
  // plane z of the shortcut output leaves (16-byte stores of 8 consecutive couts, as store_set); the unrounded values join the column sums
  auto store_res = [&](int z) {
    if constexpr (FS) {
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int co = 16 * qp + 8 * h;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const unsigned off = co < p.Cout2 ? (unsigned)((((z * p.H + cy0 + r0 + r) * p.W + cx0 + l32) * p.ldy2 + co) * 2) : 0x80000000u;
          float f[4], g2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { f[j] = acc2[r][8 * qp + j]; g2[j] = acc2[r][8 * qp + 4 + j]; }
          if (p.accum2) {     // old values arrive in the exchanged layout: the exchange is its own inverse
            u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(y2r, off, 0, 0);
            asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                         : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
            float old[8];
            unpack8<T>(e, old);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] += old[j]; g2[j] += old[4 + j]; }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { gsum[8 * qp + j] += f[j]; gsum[8 * qp + 4 + j] += g2[j]; }
          unsigned d0 = pack2<T>(f[0], f[1]), d1 = pack2<T>(f[2], f[3]), d2 = pack2<T>(g2[0], g2[1]), d3 = pack2<T>(g2[2], g2[3]);
          asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{d0, d1, d2, d3}, y2r, off, 0, LP_OUT_STORE_AUX);
        }
      }
      init_res();
    }
  };
  // the item's column sums: over the 32 voxels of a lane half, then one fp64 partial row per (item, wave)
  auto gap_flush = [&](int item) {
    if constexpr (FS) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = gsum[i];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        gsum[i] = v;
      }
      if (l32 == 0 && p.gap_part != nullptr) {
        const int per = p.fs_B / 8;                       // items per sample
        const int li = item - cn * per;
        double* dst = p.gap_part + ((long)cn * p.fs_B + (long)li * 8 + wave) * p.Cout2;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = 8 * (i >> 2) + 4 * h + (i & 3);
          if (co < p.Cout2) dst[co] = (double)gsum[i];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) gsum[i] = 0.f;
    }
  };
  // GNB (GroupNorm-backward class sums, LpGnbFuse).  Element e of a lane's stored 8 couts is channel co + e with co a multiple of 8, so
  // its class (channel mod cg, cg | 4) is e mod cg for every lane.  With t = c * g' + b' (g' = rstd * gamma_j, b' = beta_j - mean * g':
  // per sample and channel, in LDS) the masked gradient is dE = [t > 0] da and the class sums are B_j = S0_j, A_j = rstd * (S1_j - mean *
  // S0_j) with S0 = sum dE, S1 = sum dE * c: five vector instructions per element, run right after each 16-byte store.  The
  // GroupNorm-input rows of the plane a stage completes are requested from INSIDE that stage (after its plane requests), so they have
  // arrived when it ends: requested at the store they cost the HBM latency once per plane with the matrix pipe idle (0.60 -> 0.95 ms
  // per launch; 0.83 ms this way).  What is left is the arithmetic itself, eight waves between two stages (~1.5 us per plane).  Moving
  // it into the next stage's matrix-instruction shadow needs the four stored rows + the four input rows live across the stage: 32
  // registers the kernel does not have (256 with scratch spills of 0.9-2 KB per lane; built, measured in the compiler's report, dropped).
  // Per-lane fp32 sums are flushed (wave sum in fp64, one partial row per wave) where the march leaves a group or the item.
  __amdgpu_buffer_rsrc_t cr = yr;
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  f32x2v gs0[2], gs1[2];          // per-lane sums of the element pairs (0,1) and (2,3) (cg <= 4: elements e and e + 4 of a row share a class)
#pragma unroll
  for (int e = 0; e < 2; ++e) gs0[e] = gs1[e] = f32x2v{0.f, 0.f};
  double gn_ds = 0.0, gn_dq = 0.0;      // (gn_run) this lane's sums over the planes of the current group
  u32x4 cxs[2][2], dst_;
  int zst = -1;                    // plane of the row in dst_
  auto gnb_request = [&](int z) {
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int co = 16 * qp + 8 * h;
        const unsigned offc = (co < p.Cout && z >= zlo) ? (unsigned)((((z * p.H + cy0 + r0 + r) * p.W + cx0 + l32) * p.Cout + co) * 2) : 0x80000000u;
        cxs[qp][r] = __builtin_amdgcn_raw_buffer_load_b128(cr, offc, 0, 0);
      }
  };
  f32x2v gpr[2], bpr[2];           // g', b' of the element pairs of the plane being stored (one batch of LDS reads per plane: read where
                                   // they are used, every one of them cost its own lgkmcnt wait -- 75 per plane)
  auto gnb_params = [&](int z) {
    const float* gsh = reinterpret_cast<const float*>(lds + OFF_GB) + (z / p.gb_zt) * p.gb.cg;
    const int cmask = p.gb.cg - 1;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      gpr[e] = f32x2v{gsh[(2 * e) & cmask], gsh[(2 * e + 1) & cmask]};
      bpr[e] = f32x2v{gsh[32 + ((2 * e) & cmask)], gsh[32 + ((2 * e + 1) & cmask)]};
    }
  };
  // the row just stored (dst_) against its GroupNorm-input row, two elements (one dword of each) at a time on the packed fp32
  // instructions: t = c g' + b', dE = [t > 0] d, S1 += dE c, S0 += dE -- 3 packed + 2 selects + 4 unpacks per pair
  auto gnb_math = [&](const u32x4& crow, bool live) {
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned cw = crow[i], dw = dst_[i];
        const f32x2v c = {T::ld((unsigned short)(cw & 0xffffu)), T::ld((unsigned short)(cw >> 16))};
        const f32x2v d = {T::ld((unsigned short)(dw & 0xffffu)), T::ld((unsigned short)(dw >> 16))};
        const f32x2v t = c * gpr[i & 1] + bpr[i & 1];
        const f32x2v de = {(!p.gb.relu || t[0] > 0.f) ? d[0] : 0.f, (!p.gb.relu || t[1] > 0.f) ? d[1] : 0.f};
        gs1[i & 1] = de * c + gs1[i & 1];
        gs0[i & 1] = gs0[i & 1] + de;
      }
    }
  };
  auto gnb_flush = [&]() {          // after the last row of plane zst: the run of planes of one (sample, group) may end here
    if ((zst + 1) % p.gb_zt == 0 || zst == zhi - 1) {
      const float* gsh = reinterpret_cast<const float*>(lds + OFF_GB);
      const int gg = zst / p.gb_zt;
      const double gm = (double)gsh[64 + gg], grs = (double)gsh[96 + gg];
      double ra[4], rb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { ra[e] = wave_sum_f64((double)gs1[e >> 1][e & 1]); rb[e] = wave_sum_f64((double)gs0[e >> 1][e & 1]); }
#pragma unroll
      for (int e = 0; e < 2; ++e) gs0[e] = gs1[e] = f32x2v{0.f, 0.f};
      if (lane == 0) {
        const int run = p.ZC < p.gb_zt ? (zlo - gg * p.gb_zt) / p.ZC : 0;
        const long slot = ((long)run * (p.nty * p.ntx) + (cy0 / S1Z_TY) * p.ntx + cx0 / S1Z_TX) * 8 + wave;
        double* dst = p.gb.part + (((long)cn * p.gb.G + gg) * p.gb.B + slot) * (p.gb.cg * 2);
        for (int j = 0; j < p.gb.cg; ++j) {
          double s1 = 0.0, s0 = 0.0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((e & (p.gb.cg - 1)) == j) { s1 += ra[e]; s0 += rb[e]; }
          dst[2 * j] = grs * (s1 - gm * s0);
          dst[2 * j + 1] = s0;
        }
      }
    }
  };
  // the completed output plane z of accumulator set s leaves (and the set restarts at the bias)
  auto store_set = [&](auto sc, int z) {
    constexpr int s = decltype(sc)::value;
    const bool gn_on = p.gnp != nullptr;
    float gn_s = 0.f, gn_q = 0.f;
    if constexpr (GNB) gnb_params(z);
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      const int co = 16 * qp + 8 * h;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool ok = co < p.Cout;
        const unsigned off = ok ? (unsigned)((((z * p.H + cy0 + r0 + r) * p.W + cx0 + l32) * p.ldy + co) * 2) : 0x80000000u;
        float f[4], g2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = acc[s][r][8 * qp + j]; g2[j] = acc[s][r][8 * qp + 4 + j]; }
        if (p.accum) {     // old values arrive in the exchanged layout: the exchange is its own inverse
          u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(yr, off, 0, 0);
          asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                       : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
          float old[8];
          unpack8<T>(e, old);
#pragma unroll
          for (int j = 0; j < 4; ++j) { f[j] += old[j]; g2[j] += old[4 + j]; }
        }
        // (own values, BEFORE the exchange: this lane's couts 16 qp + 4 h + j and 16 qp + 8 + 4 h + j, not the stored piece's 16 qp + 8 h ..
        // + 7 -- gating the sums on the STORED piece's `ok` dropped real columns where Cout % 16 == 8 (round-6 finding: statistics of a
        // 24-cout conv off by 50 % of the mean).  Columns >= Cout carry zero weights and a zero bias: they add exact zeros)
        if (gn_on) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            gn_s += f[j] + g2[j];
            gn_q = fmaf(f[j], f[j], fmaf(g2[j], g2[j], gn_q));
          }
        }
        unsigned d0 = pack2<T>(f[0], f[1]), d1 = pack2<T>(f[2], f[3]), d2 = pack2<T>(g2[0], g2[1]), d3 = pack2<T>(g2[2], g2[3]);
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#ifdef S1Z_EXP_NOSTORE   // timing experiment (wrong results)
        if (z == -12345)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{d0, d1, d2, d3}, yr, off, 0, LP_OUT_STORE_AUX);
        if constexpr (GNB) {      // (after the exchange: the STORED couts co .. co + 7 of voxel l32, as a reduce pass would read them)
          dst_ = u32x4{d0, d1, d2, d3};
          zst = z;
          gnb_math(cxs[qp][r], 16 * qp + 8 * h < p.Cout);
        }
      }
    }
    if constexpr (GNB) gnb_flush();
    if (gn_on && p.gn_run) {
      // the planes of a group this item covers, one after the other: per-lane fp64 running sums of the per-plane fp32 sums; ONE pair per
      // (group, item, column, wave) leaves where the march leaves the group or the item (a pair per plane made the finalize walk 9600
      // pairs per unit at 160 x 192 x 160: 38 us per launch on the critical path of every top-level GroupNorm)
      gn_ds += (double)gn_s;
      gn_dq += (double)gn_q;
      if ((z + 1) % p.gn_zt == 0 || z == zhi - 1) {
        const double ds = wave_sum_f64(gn_ds), dq = wave_sum_f64(gn_dq);
        gn_ds = gn_dq = 0.0;
        if (lane == 0) {
          const int gg = z / p.gn_zt;
          const int run = p.ZC < p.gn_zt ? (zlo - gg * p.gn_zt) / p.ZC : 0;
          const long slot = ((long)run * (p.nty * p.ntx) + (cy0 / S1Z_TY) * p.ntx + cx0 / S1Z_TX) * 8 + wave;
          double* dst = p.gnp + (((long)cn * p.gn_G + gg) * p.gn_B + slot) * 2;
          dst[0] = ds;
          dst[1] = dq;
        }
      }
    } else if (gn_on) {   // one fp64 (sum, sumsq) pair per (z plane, column, wave): fixed order
      const double ds = wave_sum_f64((double)gn_s), dq = wave_sum_f64((double)gn_q);
      if (lane == 0) {
        const int gg = z / p.gn_zt;
        const long slot = ((long)(z - gg * p.gn_zt) * (p.nty * p.ntx) + (cy0 / S1Z_TY) * p.ntx + cx0 / S1Z_TX) * 8 + wave;
        double* dst = p.gnp + (((long)cn * p.gn_G + gg) * p.gn_B + slot) * 2;
        dst[0] = ds;
        dst[1] = dq;
      }
    }
    init_set(s);
  };
  // One stage: input plane zp (buffer buf) into the three accumulator sets.  Set of output plane z: (z - zlo + 1) % 3, i.e. with
  // R = (zp - (zlo - 1)) % 3 the tap kz (output plane zp + 1 - kz) accumulates into set (R + 4 - kz) % 3.
  // 9 * KS groups (k-step, x tap, z tap) of six matrix instructions: three weight fragments (y taps) x the wave's two rows.  The fragments
  // of group g + 1 are read while group g multiplies (two register sets, fixed by sched_barriers: left alone the compiler reads every
  // fragment right before its first use and waits out the LDS latency 54 times per stage: 0.93 PFLOP/s instead of ...); the input rows
  // change every third group.  One plane request per group while there are any.
  auto stage = [&](auto rc, int zp, int buf) {
    constexpr int R = decltype(rc)::value;
    constexpr int NG = 9 * KS;
    const unsigned pbo = (unsigned)(buf * PLB);
    const bool more = zp < zhi;
    u32x4 A[2][3], B[2][4];
    auto ldA = [&](u32x4 (&a)[3], int g) {
      const int ks = g / 9, dx = (g / 3) % 3, kz = g % 3;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) a[ky] = *reinterpret_cast<const u32x4*>(lds + wbA + ((ks * 3 + kz) * 9 + ky * 3 + dx) * 1024);
    };
    auto ldB = [&](u32x4 (&bb)[4], int g) {
      const int ks = g / 9, dx = (g / 3) % 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) bb[i] = *reinterpret_cast<const u32x4*>(lds + hbB[dx] + pbo + ks * (NCHK * 1024) + i * (SX * 32));
    };
    ldB(B[0], 0);
    ldA(A[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SC) {      // centre tap (kz = 1) of output plane zp: set R
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        acc[R][0] = T::mfma(A2[ks], B2[ks][0], acc[R][0]);
        acc[R][1] = T::mfma(A2[ks], B2[ks][1], acc[R][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int kz = g % 3;
      const int bs = (g / 3) & 1;
      if (g + 1 < NG) {
        ldA(A[(g + 1) & 1], g + 1);
        if ((g + 1) % 3 == 0) ldB(B[bs ^ 1], g + 1);
      }
#ifdef S1Z_EXP_NODMA     // timing experiment (wrong results)
      if (g < NR) issue_filler();
#else
#ifdef S1Z_EXP_2ISSUE
      if (g == 0 && wave < 2) {
        for (int j = 0; j < NR; ++j) { if (more) issue1(j, zp + 1, buf ^ 1); else issue_filler(); }
      }
#else
      if (g < NR) { if (more) issue1(g, zp + 1, buf ^ 1); else issue_filler(); }
#endif
#endif
      if constexpr (GNB) {
        if (g == (NR < NG ? NR : NG - 1)) gnb_request(zp - 1);       // (after this stage's plane requests)
      }
      if constexpr (SC) {
        if (g == (NR < NG ? NR : NG - 1)) sc_request(zp + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int s = (R + 4 - kz) % 3;
      if constexpr (FS) {      // centre tap of the 1x1x1 conv: x tap 1, rows r0 / r0 + 1 = fragments 1 and 2 of this x tap
        if ((g / 3) % 3 == 1 && kz == 1) {
          acc2[0] = T::mfma(Apt[g / 9], B[bs][1], acc2[0]);
          acc2[1] = T::mfma(Apt[g / 9], B[bs][2], acc2[1]);
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#ifdef S1Z_EXP_NOMFMA   // timing experiment (wrong results)
        if (ky) { asm volatile("" ::"v"(A[g & 1][ky]), "v"(B[bs][ky + 1])); continue; }
#endif
        acc[s][0] = T::mfma(A[g & 1][ky], B[bs][ky], acc[s][0]);
        acc[s][1] = T::mfma(A[g & 1][ky], B[bs][ky + 1], acc[s][1]);
#ifdef S1Z_EXP_MFMA2     // timing experiment (wrong results): twice the matrix work per stage
        acc[s][0] = T::mfma(A[g & 1][ky], B[bs][ky], acc[s][0]);
        acc[s][1] = T::mfma(A[g & 1][ky], B[bs][ky + 1], acc[s][1]);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef S1Z_EXP_2ISSUE
    static_assert(NR <= NG, "a stage has one request slot per group");
#endif
  };

  xplane = (unsigned)(p.H * p.W * p.ldx * 2);
  // weights and bias: once per workgroup
  {
    const u32x4 wr = s1z_rsrc(p.wp);
    for (int c = wave; c < 27 * KS; c += 8) s1z_dma16(wr, lds0 + (unsigned)(c * 1024), (unsigned)(lane * 16), (unsigned)(c * 1024));
    if (tid < 32) reinterpret_cast<float*>(lds + OFF_BIAS)[tid] = (p.bias != nullptr && tid < p.Cout) ? p.bias[tid] : 0.f;
    if constexpr (SC) {
      const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp2, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) A2[ks] = __builtin_amdgcn_raw_buffer_load_b128(w2r, (unsigned)(h * 512 + l32 * 16), (unsigned)ks * 1024u, 0);
      x2plane = (unsigned)(p.H * p.W * p.ldx2 * 2);
    }
    if constexpr (FS) {
      const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp2, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) Apt[ks] = __builtin_amdgcn_raw_buffer_load_b128(w2r, (unsigned)(h * 512 + l32 * 16), (unsigned)ks * 1024u, 0);
      if (tid >= 32 && tid < 64) reinterpret_cast<float*>(lds + OFF_BIAS2)[tid - 32] = (p.bias2 != nullptr && tid - 32 < p.Cout2) ? p.bias2[tid - 32] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) gsum[i] = 0.f;
    }
  }
  const int w = blockIdx.x;
  int it0 = w * p.ipw;
  if (p.xcd_order) it0 = ((w & 7) * (gridDim.x >> 3) + (w >> 3)) * p.ipw;
  int it1 = it0 + p.ipw;
  if (it1 > p.nitems) it1 = p.nitems;
  for (int item = it0; item < it1; ++item) {
    setup(item);
    yr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long)cn * p.D * p.H * p.W * (long)p.ldy), 0, 0x7fffffff, 0x00020000);
    if constexpr (FS) y2r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y2 + (long)cn * p.D * p.H * p.W * (long)p.ldy2), 0, 0x7fffffff, 0x00020000);
    if constexpr (SC) {
      x2r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 + (long)cn * p.D * p.H * p.W * (long)p.ldx2), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int r = 0; r < 2; ++r) sc_off[r] = (unsigned)((((cy0 + r0 + r) * p.W + cx0 + l32) * p.ldx2 + 8 * h) * 2);
      sc_request(zlo - 1);      // (outside the chunk: zeros)
    }
    if constexpr (GNB) {
      cr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gb.x + (long)cn * p.D * p.H * p.W * (long)p.Cout), 0, 0x7fffffff, 0x00020000);
      // (the previous item's last reads lie behind its closing barrier; the barrier below publishes these)
      float* gsh = reinterpret_cast<float*>(lds + OFF_GB);
      if (tid < p.gb.G) {
        gsh[64 + tid] = p.gb.mean[cn * p.gb.G + tid];
        gsh[96 + tid] = p.gb.rstd[cn * p.gb.G + tid];
      }
      if (tid >= 64 && tid < 64 + p.Cout) {      // channel c = g * cg + j of the affine parameters: g' = rstd gamma, b' = beta - mean g'
        const int c = tid - 64, gq = c / p.gb.cg;
        const float gp = p.gb.rstd[cn * p.gb.G + gq] * p.gb.gamma[c];
        gsh[c] = gp;
        gsh[32 + c] = p.gb.beta[c] - p.gb.mean[cn * p.gb.G + gq] * gp;
      }
    }
    if constexpr (GNA) {
      ga_setup();
      float* gsh = reinterpret_cast<float*>(lds + OFF_GB);
      const int C = 16 * KS;
      if (tid < C) { gsh[tid] = p.ga.gamma[tid]; gsh[32 + tid] = p.ga.beta[tid]; }
      if (tid >= 64 && tid < 64 + p.ga.G) {
        gsh[64 + tid - 64] = p.ga.mean[cn * p.ga.G + tid - 64];
        gsh[96 + tid - 64] = p.ga.rstd[cn * p.ga.G + tid - 64];
      }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) issue1(j, zlo - 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (GNA) {
      ga_apply(0, zlo - 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    init_set(0); init_set(1); init_set(2);
    init_res();
    int zp = zlo - 1, buf = 0;
    // stage zp completes output plane zp - 1 (set of tap kz = 2)
#define S1Z_STAGE(RR)                                                                        \
    stage(S1zIC<RR>{}, zp, buf);                                                             \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_s_barrier();                                                            \
    asm volatile("" ::: "memory");                                                           \
    if constexpr (GNA) {      /* the next plane has landed: normalise it in place; visible to every wave before the next stage reads it */ \
      if (zp < zhi) ga_apply(buf ^ 1, zp + 1);                                               \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
      __builtin_amdgcn_s_barrier();                                                          \
      asm volatile("" ::: "memory");                                                         \
    }                                                                                        \
    if (zp - 1 >= zlo) store_set(S1zIC<(RR + 2) % 3>{}, zp - 1); else init_set((RR + 2) % 3);  \
    if constexpr (FS) { if (zp >= zlo && zp < zhi) store_res(zp); else init_res(); }         \
    buf ^= 1;                                                                                \
    if (++zp > zhi) break;
    for (;;) {
      S1Z_STAGE(0)
      S1Z_STAGE(1)
      S1Z_STAGE(2)
    }
#undef S1Z_STAGE
    gap_flush(item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last plane's stores and fillers) before the buffers change hands
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#endif
}

// =====================================================================================================================
// plan + launch
// =====================================================================================================================
struct S1zPlan { int ntx, nty, nzc, ZC, nitems, ipw, nwg, xcd; };
static bool s1z_enabled() {   // BTS_LP_S1Z=0: these layers back on lowp_s1d.hip's tiled kernel (A/B; read per call)
  const char* e = getenv("BTS_LP_S1Z");
  return !(e && atoi(e) == 0);
}
// Cin = 64 (the decoder's top block reading [skip, up-sampled]: 64 -> 32 at 128^3, decoder.py:55-63 / resnet.py:80-87): TWO passes of the
// 32-channel kernel over the two channel halves of the input slab, the second accumulating into the first's output.  The tiled kernel
// (lowp_s1d.hip) it replaces there re-reads its input 2.7x (34 x 10 x 6 halo of a 32 x 8 x 4 tile, a block's footprint beyond one XCD's
// L2) and is HBM-bound at 0.35 of the matrix peak; the two marches read the input 1.2x and pay one extra read of the 32-channel output.
// Measured (round 5, profiles/r05_ab_e1.txt): the batch-8 training step gains 0.3 ms (8 x 128^3 voxels: 2 x ~0.72 ms against 2.09 ms), the
// single 160 x 192 x 160 inference volume LOSES 0.07 ms (2 x 285 us against 506 us: the tiled kernel's re-reads still fit that grid's
// caches) -- so the pair takes launches of at least 8 M voxels only.  BTS_LP_S1Z_PAIR=0: never, =1: always (A/B; read per call).
static bool s1z_pair_takes(long voxels) {
  const char* e = getenv("BTS_LP_S1Z_PAIR");
  if (e) return atoi(e) != 0;
  return voxels >= (8L << 20);
}
static bool s1z_plan(S1zPlan& pl, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy) {
  // (the geometry of both passes: the plan does not depend on the channel half.  A voxel stride below 64 says the halves are two dense
  // tensors -- the operands of a concat -- which no other kernel reads as one: the pair at any size)
  if (Cin == 64 && (ldx < 64 || s1z_pair_takes((long)N * D * H * W))) Cin = 32;
  if (!s1z_enabled() || (Cin != 16 && Cin != 32) || Cout > 32 || Cout % 8 != 0 || W % S1Z_TX != 0 || H % S1Z_TY != 0 || D < 8) return false;
  if (ldx % 8 != 0 || ldy % 8 != 0) return false;
  if ((long)D * H * W * (long)ldx * 2 >= 0x7fffffffL || (long)D * H * W * (long)ldy * 2 >= 0x7fffffffL) return false;
  pl.ntx = W / S1Z_TX; pl.nty = H / S1Z_TY;
  const long ncol = (long)N * pl.ntx * pl.nty;
  if (ncol * D < 96) return false;      // (small volumes: the tiled kernel)
  int nzc = 1;
  while (ncol * nzc < 224 && (D + 2 * nzc - 1) / (2 * nzc) >= 16) nzc *= 2;
  pl.ZC = (D + nzc - 1) / nzc;
  pl.nzc = (D + pl.ZC - 1) / pl.ZC;
  const long items = ncol * pl.nzc;
  if (items > 0x3fffffffL) return false;
  pl.nitems = (int)items;
  const int nwg = pl.nitems < 256 ? pl.nitems : 256;
  pl.ipw = (pl.nitems + nwg - 1) / nwg;
  pl.nwg = (pl.nitems + pl.ipw - 1) / pl.ipw;
  pl.xcd = (pl.nwg % 8 == 0 && pl.nwg * pl.ipw == pl.nitems) ? 1 : 0;
  return true;
}
// GroupNorm-partial slots per (n, group) when the kernel takes the shape and can emit them (whole planes per group); 0 otherwise
static bool s1z_gn_runs(const S1zPlan& pl, int zt) { return pl.ZC % zt == 0 || zt % pl.ZC == 0; }      // z chunks nest with the groups
static long s1z_gn_B(const S1zPlan& pl, int zt) {
  if (s1z_gn_runs(pl, zt)) return (long)(pl.ZC < zt ? zt / pl.ZC : 1) * pl.nty * pl.ntx * 8;      // a pair per (run, column, wave)
  return (long)zt * pl.nty * pl.ntx * 8;                                                             // a pair per (plane, column, wave)
}
long bts_lp_s1z_gn_B_(int N, int D, int H, int W, int Cin, int Cout, int Gn) {
  S1zPlan pl;
  if (Gn <= 0 || D % Gn != 0 || !s1z_plan(pl, N, D, H, W, Cin, Cin, Cout, Cout)) return 0;
  return s1z_gn_B(pl, D / Gn);
}
// partial rows per (n, group) of the fused GroupNorm-BACKWARD class sums (LpGnbFuse) when the kernel takes the shape and can emit them:
// whole planes per group, z chunks that nest with the groups, classes that divide a lane's 8 couts; 0 otherwise
long bts_lp_s1z_gnb_B_(int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, int Gn) {
  S1zPlan pl;
  if (Cin == 64) return 0;      // (two-pass shape: the epilogue forms belong to single launches)
  if (Gn <= 0 || D % Gn != 0 || Cout % Gn != 0 || !s1z_plan(pl, N, D, H, W, Cin, ldx, Cout, ldy)) return 0;
  const int cg = Cout / Gn, zt = D / Gn;
  if (cg > 4 || (4 % cg) != 0 || Gn > 32 || Cout > 32) return 0;
  if (pl.ZC % zt != 0 && zt % pl.ZC != 0) return 0;
  return (long)(pl.ZC < zt ? zt / pl.ZC : 1) * pl.nty * pl.ntx * 8;
}
// FS form: partial rows per sample of the shortcut's column sums (8 per item), or 0 where the kernel does not take the shape
long bts_lp_s1z_fs_B_(int N, int D, int H, int W, int Cin, int ldx, int Cout, int Cout2) {
  S1zPlan pl;
  if (Cout2 <= 0 || Cout2 > 32 || Cout2 % 8 != 0 || !s1z_plan(pl, N, D, H, W, Cin, ldx, Cout, Cout)) return 0;      // (Cin = 64: the two-pass form, where it is taken)
  return (long)pl.ntx * pl.nty * pl.nzc * 8;
}
// BTS_OK = ran, 1 = declined.  wp = the DMA part of the K3S1 image.  gb (may be NULL): see LpGnbFuse; its B must be bts_lp_s1z_gnb_B_'s
// does the kernel take the shape with GroupNorm `in_G` applied to its input planes (LpGnaFuse)?
bool bts_lp_s1z_gna_ok_(int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, int in_G) {
  S1zPlan pl;
  if (Cin == 64) return false;
  if (in_G <= 0 || in_G > 32 || D % in_G != 0 || Cin % in_G != 0 || ldx != Cin || !s1z_plan(pl, N, D, H, W, Cin, ldx, Cout, ldy)) return false;
  const int cg = Cin / in_G;
  return cg <= 8 && 8 % cg == 0;
}
int bts_lp_s1z_launch_(int dtype, const void* x, const void* wp, const float* bias, void* y, int N, int D, int H, int W, int Cin, int ldx,
                       int Cout, int ldy, int accum, double* gn_part, int gn_G, hipStream_t stream, const LpGnbFuse* gb, const LpGnaFuse* ga,
                       const void* x2, const void* wp2, int ldx2, void* y2, const float* bias2, double* gap_part, int ldy2, int Cout2, int accum2,
                       const void* xb, int ldxb) {
  S1zPlan pl;
  if (!s1z_plan(pl, N, D, H, W, Cin, ldx, Cout, ldy)) return 1;
  const bool sc = x2 != nullptr;
  const bool fs = y2 != nullptr;
  // (FS: accum / accum2 / a NULL gap_part are the two-pass form's own business -- callers ask for a plain launch)
  if (fs && (sc || gb != nullptr || ga != nullptr || wp2 == nullptr || Cout2 <= 0 || Cout2 > 32 ||
             Cout2 % 8 != 0 || ldy2 < Cout2 || ldy2 % 8 != 0 || (((uintptr_t)y2) & 15) || (((uintptr_t)wp2) & 15) ||
             (long)D * H * W * (long)ldy2 * 2 >= 0x7fffffffL))
    return 1;
  if (xb != nullptr && (Cin != 64 || ldxb < 32 || ldxb % 8 != 0 || (((uintptr_t)xb) & 15) || (long)D * H * W * (long)ldxb * 2 >= 0x7fffffffL)) return 1;
  if (sc && (Cin == 64 || gb != nullptr || ga != nullptr || gn_part != nullptr || wp2 == nullptr || ldx2 < Cin || ldx2 % 8 != 0 ||
             (((uintptr_t)x2) & 15) || (((uintptr_t)wp2) & 15) || (long)D * H * W * (long)ldx2 * 2 >= 0x7fffffffL))
    return 1;
  if (Cin == 64) {      // two 32-channel passes (see s1z_plan): channels [0, 32) write (or accumulate, as asked), [32, 64) accumulate and count
    if (gb != nullptr || ga != nullptr || sc) return 1;
    // everything either pass checks, BEFORE the first one touches y: a decline after pass 1 would leave a half-summed output behind (and
    // the caller's fallback would add the first half twice when accumulating)
    if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 15) || (((uintptr_t)wp) & 15)) return 1;      // (x + 32 channels = + 64 bytes, wp + 54 KB: aligned with them)
    if (gn_part != nullptr && (gn_G <= 0 || D % gn_G != 0)) return 1;
    if (fs && (accum || accum2)) return 1;
    // (FS: the shortcut's two halves likewise -- the first pass writes res = x[0:32] . W[0:32] + bias, the second adds x[32:64] . W[32:64] and
    // leaves the column sums of the result; the 1x1x1 image is [k-step][k-half][32][8]: the second half starts two k-steps = 2 KB in)
    const int r = bts_lp_s1z_launch_(dtype, x, wp, bias, y, N, D, H, W, 32, ldx, Cout, ldy, accum, nullptr, 0, stream, nullptr, nullptr, nullptr,
                                     fs ? wp2 : nullptr, 0, fs ? y2 : nullptr, bias2, nullptr, ldy2, Cout2, 0, nullptr, 0);
    if (r != BTS_OK) return r;
    // (image: [k-step][dz][tap][k-half][32 couts][8 cin], 27 KB per k-step: the second half starts two k-steps in.  The first half's sum
    // passes through the storage type once before the second is added: one extra rounding, carried by the test bounds.  xb: the second
    // half as a tensor of its own -- the two operands of a concat, decoder.py:75 -- instead of channels 32..63 of x)
    const void* x2nd = xb != nullptr ? xb : static_cast<const void*>(reinterpret_cast<const unsigned short*>(x) + 32);
    const int r2 = bts_lp_s1z_launch_(dtype, x2nd, reinterpret_cast<const char*>(wp) + 2 * 27 * 1024, nullptr, y, N, D, H, W, 32,
                                      xb != nullptr ? ldxb : ldx, Cout, ldy, 1, gn_part, gn_G, stream, nullptr, nullptr, nullptr,
                                      fs ? reinterpret_cast<const char*>(wp2) + 2 * 1024 : nullptr, 0, fs ? y2 : nullptr, nullptr, fs ? gap_part : nullptr,
                                      ldy2, Cout2, fs ? 1 : 0, nullptr, 0);
    return r2 == 1 ? BTS_ERR_UNSUPPORTED : r2;      // (y has been written: "declined" is no longer an answer)
  }
  if (ga != nullptr && (gb != nullptr || !bts_lp_s1z_gna_ok_(N, D, H, W, Cin, ldx, Cout, ldy, ga->G) || ga->cg != Cin / ga->G)) return 1;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 15) || (((uintptr_t)wp) & 15)) return 1;
  if (gn_part != nullptr && (gn_G <= 0 || D % gn_G != 0)) return 1;
  if (gb != nullptr && (gn_part != nullptr || gb->B != bts_lp_s1z_gnb_B_(N, D, H, W, Cin, ldx, Cout, ldy, gb->G) || gb->B <= 0 ||
                        (((uintptr_t)gb->x) & 15)))
    return 1;
  LpS1zParams p;
  p.x = (const unsigned short*)x; p.wp = (const unsigned short*)wp; p.bias = bias; p.y = (unsigned short*)y;
  p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldy; p.Cout = Cout;
  p.ntx = pl.ntx; p.nty = pl.nty; p.nzc = pl.nzc; p.ZC = pl.ZC; p.nitems = pl.nitems; p.ipw = pl.ipw; p.xcd_order = pl.xcd; p.accum = accum;
  p.gnp = gn_part; p.gn_G = gn_G; p.gn_zt = gn_G > 0 ? D / gn_G : 1;
  p.gn_run = (gn_G > 0 && s1z_gn_runs(pl, p.gn_zt)) ? 1 : 0;
  p.gn_B = gn_G > 0 ? s1z_gn_B(pl, p.gn_zt) : 0;
  if (gb != nullptr) { p.gb = *gb; p.gb_zt = D / gb->G; } else { p.gb = LpGnbFuse{}; p.gb_zt = 1; }
  if (ga != nullptr) { p.ga = *ga; p.ga_zt = D / ga->G; } else { p.ga = LpGnaFuse{}; p.ga_zt = 1; }
  p.x2 = (const unsigned short*)x2; p.wp2 = (const unsigned short*)wp2; p.ldx2 = ldx2;
  p.y2 = (unsigned short*)y2; p.bias2 = bias2; p.gap_part = gap_part; p.ldy2 = ldy2; p.Cout2 = Cout2; p.accum2 = accum2;
  p.fs_B = pl.ntx * pl.nty * pl.nzc * 8;
  const int KS = Cin / 16;
  const size_t shmem = (size_t)(27 * KS * 1024 + 2 * S1Z_NCHK * 1024 * KS + 256 + 1024 + 512);
  (void)hipGetLastError();
#define S1Z_LAUNCH(TT, KS_) do { if (sc) S1Z_LAUNCH_(TT, KS_, false, false, 1); else if (fs) S1Z_LAUNCH_(TT, KS_, false, false, 2); else if (gb != nullptr) S1Z_LAUNCH_(TT, KS_, true, false, 0); else if (ga != nullptr) S1Z_LAUNCH_(TT, KS_, false, true, 0); else S1Z_LAUNCH_(TT, KS_, false, false, 0); } while (0)
#define S1Z_LAUNCH_(TT, KS_, GB_, GA_, SC_)                                                                                  \
  do {                                                                                                                       \
    auto kern = lp_s1z_kernel<TT, KS_, GB_, GA_, SC_>;                                                                          \
    static bool done = false;                                                                                                \
    if (!done) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      if (e != hipSuccess) return (int)e;                                                                                    \
      done = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL(kern, dim3(pl.nwg), dim3(512), shmem, stream, p);                                                     \
  } while (0)
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(38, 2.0 * ((double)(sc ? 28 : 27) * Cout + (fs ? Cout2 : 0)) * (double)Cin * (double)N * D * H * W, stream);
  if (dtype == LP_F16) { if (KS == 2) S1Z_LAUNCH(TF16, 2); else S1Z_LAUNCH(TF16, 1); }
  else { if (KS == 2) S1Z_LAUNCH(TBF16, 2); else S1Z_LAUNCH(TBF16, 1); }
#undef S1Z_LAUNCH
#undef S1Z_LAUNCH_
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
