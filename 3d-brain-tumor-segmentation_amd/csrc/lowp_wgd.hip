// Weight gradient of the stride-1 3x3x3 convolutions with FEW output channels (Cout <= 32: the 128^3 level of the CLI model) on
// 16-bit operands -- what TF autodiff derives for the Conv3D kernels of resnet.py:80-87,96-103 and vae.py:92-99 under
// train.py:142-151:   dW[t][c][k] = sum_v P[v + off_t][c] * Q[v][k],   P = the conv's input, Q = the gradient of its output.
//
// At 32 x 32 channels the arithmetic intensity of this contraction (27 taps x 2 x 32 x 32 flops per 128 operand bytes = 432 flop/B)
// sits just below the machine balance (2.5 PFLOP/s : ~5 TB/s): the layer is bound by reading P and Q ONCE.  lowp.hip's general
// kernel (register staging with a 2-byte interleave, 16 x 8 x 4 tiles whose 18 x 10 x 6 halo re-reads P 2.1 times, two idle waves)
// ran it at 0.79 PFLOP/s = 0.43 of that bound.  This kernel is built to stream:
//   * a workgroup owns a 32 (x) x 8 (y) column and marches along z: every stage brings ONE plane of P (34 x 10 voxels: the x/y halo is
//     the only re-read, 1.33x) and one plane of Q; the three z taps come from a ring of Q planes that stay in LDS.  Planes are
//     requested TWO stages ahead (3 P buffers + 5 Q slots = 155 KB of LDS: ~76 KB per CU in flight -- with one stage ahead the
//     kernel ran at 2.2 TB/s, bound by the bytes in flight over the memory latency);
//   * the planes go global -> LDS by buffer_load ... lds (no registers, no staging instructions), as they lie in memory: [voxel][32
//     channels]; out-of-range voxels ('same' padding, planes outside the item) are requests with an out-of-range offset = zeros;
//   * the contraction runs over voxels, so both matrix operands need a transpose: ds_read_b64_tr_b16 does it on the way out of LDS (a
//     16-lane group pointing at [4 voxels][16 channels] receives, per lane, one channel's four voxels).  The two 16-byte pieces of a
//     voxel's channel half are swapped on every other 8-voxel block (chosen by the DMA's per-lane global address), which makes the
//     32 lanes served per LDS cycle cover the 64 banks exactly once;
//   * v_mfma_f32_16x16x32: a wave owns ONE 16 x 16 (cin x cout) block of all 27 taps (108 accumulation registers) for half of the
//     column's rows -- 8 waves = 2 cin halves x 2 cout halves x 2 row halves, every SIMD equally loaded.  A P fragment (one row, one
//     x tap) meets the 3 x 3 (z, y) neighbourhood of Q fragments: 60 transposing reads per 108 matrix instructions;
//   * the two row halves meet in LDS at the end: one fp32 partial slab per workgroup in the layout of lowp.hip's fixed-order finalize.
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
int bts_lp_wgrad_finalize_(const float* part, float* dw, int nwg, int ncp, int ncqg, int nslot, int ntaps, int NQ, int Cp, int Cq, int Cin_ref,
                           int dup_start, int dup_shift, int accum, hipStream_t stream);

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int V> using WgdIC = std::integral_constant<int, V>;

struct LpWgdParams {
  const unsigned short* p;   // (N, D, H, W, Cp) voxel stride ldp
  const unsigned short* q;   // (N, D, H, W, Cq) voxel stride ldq
  float* part;               // [workgroup][cp block][cq block][27 taps][32][32]
  int N, D, H, W, Cp, ldp, Cq, ldq;
  int ntx, nty, nzc, ZC;     // columns per sample (x, y), z chunks per column, planes per chunk
  int nitems, ipw, ncp, ncq, xcd_order;
  LpGnaFuse ga;              // GNA kernels: P is the RAW GroupNorm input; relu(GroupNorm(P)) is formed on the planes in LDS (lowp_common.h)
  int ga_zt;                 // planes per group
  // K1F kernels: the 1x1x1 weight gradient of a SECOND convolution that reads the same input (resnet.py:118 shortcut next to resnet.py:134
  // conv1): dW1[c][k] = sum_v P[v][c] * Q2[v][k], Q2 = the gradient of the shortcut's output -- one more accumulator per wave, fed by the P
  // fragments of the centre tap this kernel reads anyway: the 1x1x1 weight-gradient launch and its own pass over P go away
  const unsigned short* q2;  // (N, D, H, W, Cq) voxel stride ldq2
  float* part2;              // [workgroup][cp block][cq block][32][32]
  int ldq2;
  // P as a LIST of 32-channel tensors (the operands of a concat, decoder.py:75): cp block b reads p + b * psplit with voxel stride ldp
  // (0: one (N, D, H, W, Cp) tensor) -- a workgroup owns one 32-channel block of P anyway
  long psplit;
};
#define WGD_TX 32
#define WGD_TY 8
#define WGD_PROW 2560                       // bytes of a P row in LDS: 40 voxel slots x 64 B (34 in use)
#define WGD_PPL (10 * WGD_PROW)             // a P plane: rows y0 - 1 .. y0 + 8 = 400 slots = 25 requests of 16
#define WGD_QROW 2048
#define WGD_QPL (WGD_TY * WGD_QROW)
#define WGD_NPB 3                           // P planes in LDS: the stage's and the next two in flight
#define WGD_NQS 5                           // Q planes in LDS: the stage's three and the next two in flight
#define WGD_QBASE (WGD_NPB * WGD_PPL)
#define WGD_SCR (WGD_QBASE + WGD_NQS * WGD_QPL)   // 1 KB that swallows the filler requests
#define WGD_GA (WGD_SCR + 1024)             // GNA: gamma[32] | beta[32] | mean[32] | rstd[32] of the item's sample (floats)
#define WGD_LDS (WGD_GA + 512)              // 160256 bytes

// One LDS-DMA request (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes -> 1 KB of LDS at M0) as inline assembly.  Through the builtin the
// compiler orders every later LDS read behind the request with s_waitcnt vmcnt(0) (it cannot tell the read from the request's
// destination apart): the requests of a stage, meant to land two stages later, were each waited for on the spot -- transfer and
// arithmetic ran one after the other (1.12 ms per 32->32 @128^3 x8 launch, 0.38 + 0.57 apart).  Waits are counted by hand below.
__device__ __forceinline__ void wgd_dma16(u32x4 rsrc, unsigned lds_byte, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff) : "memory");   // (m0 is a reserved register: the compiler sets it right before each of its own uses, never across statements)
}
__device__ __forceinline__ u32x4 wgd_rsrc(const void* base) {
  const unsigned long a = (unsigned long)base;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
}

template <typename T> struct Mfma16;
template <> struct Mfma16<TF16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<TBF16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
  }
};

template <typename T, bool GNA = false, bool K1F = false>
__global__ __launch_bounds__(512, 1) void lp_wgd_kernel(const LpWgdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (see lp_s1d_kernel: the host pass drops the launch stub of this template otherwise)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave & 1, wb = (wave >> 1) & 1, hv = wave >> 2;     // cin half, cout half, row half of this wave
  const int cpt = blockIdx.y, cp0 = cpt * 32;
  const int cqt = blockIdx.z, cq0 = cqt * 32;

  // ---- DMA side: every wave issues SIX requests per stage (a counted s_waitcnt needs equal counts): id = j * 8 + wave; ids 0..24 = the 25
  // pieces of 16 slots of a P plane, 25..40 = Q rows x 2 segments, 41..47 = fillers (out of range: no traffic, 1 KB of scratch) ----
  const int vi = lane >> 2, pos = lane & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;      // LDS byte address of the dynamic segment
  u32x4 pr, qr, q2r;
  unsigned voff[6];               // this lane's byte offset inside a plane for request j of the current item (bit 31: masked = zeros)
  unsigned voff2[2];              // K1F: the same for this wave's two requests of a Q2 plane (16 = rows x 2 segments: id j * 8 + wave)
  unsigned pplane, qplane, q2plane = 0;        // bytes per plane
  int zlo = 0, zhi = 0;
  unsigned ga_ok = 0;
  int ga_n = 0, ga_g = -1;
  float ga_mu = 0.f, ga_sc[8], ga_be[8];
  auto kind_of = [&](int j) { const int id = j * 8 + wave; return id < 25 ? 0 : id < 41 ? 1 : 2; };   // P, Q, filler (wave-uniform)
  auto dst_of = [&](int j) {      // LDS byte offset of request j inside its plane
    const int id = j * 8 + wave;
    if (id < 25) return id * 1024;
    const int qi = id - 25;
    return (qi >> 1) * WGD_QROW + (qi & 1) * 1024;
  };
  auto setup = [&](int item) {
    int b = item;
    const int zc = b % p.nzc; b /= p.nzc;
    const int tx = b % p.ntx; b /= p.ntx;
    const int ty = b % p.nty;
    const int n = b / p.nty;
    const int x0 = tx * WGD_TX, y0 = ty * WGD_TY;
    zlo = zc * p.ZC;
    zhi = zlo + p.ZC;
    if (zhi > p.D) zhi = p.D;
    // origins: P at (x0 - 1, y0 - 1) of plane 0 -- possibly before the sample's first voxel: every lane that would reach there is masked
    const long pvox = ((long)n * p.D * p.H + (y0 - 1)) * p.W + (x0 - 1);
    const long qvox = ((long)n * p.D * p.H + y0) * p.W + x0;
    pr = wgd_rsrc(p.p + pvox * p.ldp + (p.psplit ? (long)cpt * p.psplit : 0L));
    qr = wgd_rsrc(p.q + qvox * p.ldq);
    if constexpr (K1F) {
      q2r = wgd_rsrc(p.q2 + qvox * p.ldq2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int qi = j * 8 + wave;
        const int row = qi >> 1, xl = (qi & 1) * 16 + vi;
        const int oct = pos ^ (2 * ((xl >> 3) & 1));
        voff2[j] = (cq0 + oct * 8 < p.Cq && x0 + xl < p.W && y0 + row < p.H) ? (unsigned)(((row * p.W + xl) * p.ldq2 + cq0 + oct * 8) * 2) : 0x80000000u;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int id = j * 8 + wave;
      unsigned v = 0x80000000u;
      if (id < 25) {
        const int slot = id * 16 + vi;                    // slot of the plane: row * 40 + xl
        const int row = slot / 40, xl = slot - row * 40;
        const int oct = pos ^ (2 * ((slot >> 3) & 1));
        if (xl < 34 && cp0 + oct * 8 < p.Cp && (unsigned)(x0 - 1 + xl) < (unsigned)p.W && (unsigned)(y0 - 1 + row) < (unsigned)p.H)
          v = (unsigned)(((row * p.W + xl) * p.ldp + (p.psplit ? 0 : cp0) + oct * 8) * 2);
      } else if (id < 41) {
        const int qi = id - 25;
        const int row = qi >> 1, xl = (qi & 1) * 16 + vi;
        const int oct = pos ^ (2 * ((xl >> 3) & 1));
        if (cq0 + oct * 8 < p.Cq && x0 + xl < p.W && y0 + row < p.H) v = (unsigned)(((row * p.W + xl) * p.ldq + cq0 + oct * 8) * 2);
      }
      voff[j] = v;
    }
    if constexpr (GNA) {      // which of this thread's 16-byte pieces of a P plane (piece tid + 512 i of 25 x 64) hold image voxels
      ga_n = n;
      ga_ok = 0;
      ga_g = -1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = tid + 512 * i;
        const int slot = (id >> 6) * 16 + ((id & 63) >> 2), ps = id & 3;
        const int row = slot / 40, xl = slot - row * 40;
        const int oct = ps ^ (2 * ((slot >> 3) & 1));
        if (id < 1600 && xl < 34 && cp0 + oct * 8 < p.Cp && (unsigned)(x0 - 1 + xl) < (unsigned)p.W && (unsigned)(y0 - 1 + row) < (unsigned)p.H)
          ga_ok |= 1u << i;
      }
    }
  };
  // GNA: P plane z has landed in buffer pbuf and nobody reads it yet: a = relu(fmaf(v - mean, rstd * gamma, beta)) in place, the arithmetic
  // of bts_lp_gn_apply element for element; pieces outside the image stay the zeros the DMA wrote ('same' padding applies to a)
  auto ga_apply = [&](int pbuf, int z) {
    if (z < 0 || z >= p.D) return;
    const int gq = z / p.ga_zt;
    if (gq != ga_g) {
      const float* gsh = reinterpret_cast<const float*>(lds + WGD_GA);
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
    u32x4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (tid + 512 * i < 1600) raw[i] = *reinterpret_cast<const u32x4*>(lds + pbuf * WGD_PPL + (tid + 512 * i) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (tid + 512 * i < 1600) {
        u32x4 r = lp_gna_slot<T>(raw[i], ga_mu, ga_sc, ga_be);
        const bool ok = (ga_ok >> i) & 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = ok ? r[k] : 0u;
        *reinterpret_cast<u32x4*>(lds + pbuf * WGD_PPL + (tid + 512 * i) * 16) = r;
      }
    }
  };
  auto ga_fence = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // request j of a stage's six: Q plane zq -> ring slot qslot, P plane zp -> buffer pbuf (want_p false: fillers instead).  The stage
  // deals its six between its matrix instructions (a request costs the wave ~100 issue cycles)
  struct StageReq { bool qok, pok, want_p; unsigned qso, pso; int qslot, pbuf; };
  auto make_req = [&](int zq, int qslot, int zp, int pbuf, bool want_p) {
    StageReq r;
    r.qok = zq >= zlo && zq < zhi;
    r.want_p = want_p;
    r.pok = want_p && zp >= 0 && zp < p.D;
    r.qso = r.qok ? (unsigned)zq * qplane : 0u;
    r.pso = r.pok ? (unsigned)zp * pplane : 0u;
    r.qslot = qslot; r.pbuf = pbuf;
    return r;
  };
  auto issue1 = [&](const StageReq& r, int j) {
#ifdef WGD_EXP_NODMA    // timing experiment (wrong results): no requests after the prologue
    if (r.qslot >= 0) return;
#endif
    if (kind_of(j) == 1) {
      wgd_dma16(qr, lds0 + (unsigned)(WGD_QBASE + r.qslot * WGD_QPL + dst_of(j)), r.qok ? voff[j] : 0x80000000u, r.qso);
    } else {
      const bool live = kind_of(j) == 0 && r.want_p;
      wgd_dma16(pr, lds0 + (unsigned)(live ? r.pbuf * WGD_PPL + dst_of(j) : WGD_SCR), (live && r.pok) ? voff[j] : 0x80000000u, r.pso);
    }
  };
  // K1F: Q2 plane z2 -> Q-ring slot `slot` (the slot of Q plane z2 - 1, dead since the stage before last: the Q fragments of a plane live in
  // registers after their first stage).  ONE stage ahead, so these two requests open a stage: the counted wait at its end leaves exactly the
  // stage's six two-ahead requests in flight and has these behind it.
  auto issue_q2 = [&](int z2, int slot) {
    if constexpr (K1F) {
      const bool ok = z2 >= zlo && z2 < zhi;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int qi = j * 8 + wave;
        wgd_dma16(q2r, lds0 + (unsigned)(WGD_QBASE + slot * WGD_QPL + (qi >> 1) * WGD_QROW + (qi & 1) * 1024), ok ? voff2[j] : 0x80000000u,
                  ok ? (unsigned)z2 * q2plane : 0u);
      }
    }
  };
  auto issue = [&](int zq, int qslot, int zp, int pbuf, bool want_p) {
    const StageReq r = make_req(zq, qslot, zp, pbuf, want_p);
#pragma unroll
    for (int j = 0; j < 6; ++j) issue1(r, j);
  };

  // ---- compute side ----
  const int g = lane >> 4, rr = (lane & 15) >> 2, c4 = lane & 3;
  unsigned poff[3][2], qoff[2];      // [x tap][k half] for even rows; odd rows: ^ 32 (row * 40 slots shifts the 8-slot block parity)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xl = kx + 8 * g + 4 * h + rr;
      const int ps = (2 * wa + (c4 >> 1)) ^ (2 * ((xl >> 3) & 1));
      poff[kx][h] = (unsigned)((4 * hv) * WGD_PROW + xl * 64 + ps * 16 + (c4 & 1) * 8);
    }
    const int xl = 8 * g + 4 * h + rr;
    const int ps = (2 * wb + (c4 >> 1)) ^ (2 * ((xl >> 3) & 1));
    qoff[h] = (unsigned)(WGD_QBASE + (4 * hv) * WGD_QROW + xl * 64 + ps * 16 + (c4 & 1) * 8);
  }
  auto trd = [&](const unsigned char* a) -> u32x2 {
#ifdef WGD_EXP_NOREAD   // timing experiment (wrong results): no operand reads from LDS
    return u32x2{(unsigned)(uintptr_t)a, 1u};
#endif
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a)));
  };
  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};      // K1F: the 1x1x1 tap

  pplane = (unsigned)(p.H * p.W * p.ldp * 2);
  qplane = (unsigned)(p.H * p.W * p.ldq * 2);
  if constexpr (K1F) q2plane = (unsigned)(p.H * p.W * p.ldq2 * 2);
  const int w = blockIdx.x;
  int it0 = w * p.ipw;
  if (p.xcd_order) it0 = ((w & 7) * (gridDim.x >> 3) + (w >> 3)) * p.ipw;      // consecutive item ranges stay on one XCD (its L2 holds the halos)
  int it1 = it0 + p.ipw;
  if (it1 > p.nitems) it1 = p.nitems;
  // Q fragments [plane role kz][row]: the planes zp, zp - 1 of a stage are the planes zp + 1, zp of the stage before -- their fragments
  // stay in registers and change ROLE (physical index (kz + rot) % 3, rot steps 0 -> 2 -> 1 -> 0); only the new plane's four are read
  u32x4 qfp[3][4];
  u32x4 pf[3];
  // One stage, straight-line: 18 steps (P row, x tap) of 3..9 matrix instructions; P fragments are read two steps ahead, the new Q
  // plane's rows just before their first use, the six requests of the stage after next are dealt over the steps.
  auto stage = [&](auto rotc, const unsigned char* pb, const unsigned char* qnew, const StageReq& rq, const unsigned char* q2p) {
    constexpr int R = decltype(rotc)::value;
    // K1F: row qrow of the Q2 plane of THIS stage's P plane (same layout and offsets as a Q plane)
    auto rdq2 = [&](int qrow) -> u32x4 {
      const u32x2 lo = trd(q2p + qoff[0] + qrow * WGD_QROW), hi = trd(q2p + qoff[1] + qrow * WGD_QROW);
      return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    u32x4 q2f = u32x4{0u, 0u, 0u, 0u};
    auto rdp = [&](int st) -> u32x4 {
      const int prow = st / 3, kx = st - prow * 3;
      const unsigned sw = (prow & 1) ? 32u : 0u;
      const u32x2 lo = trd(pb + (poff[kx][0] ^ sw) + prow * WGD_PROW), hi = trd(pb + (poff[kx][1] ^ sw) + prow * WGD_PROW);
      return u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    auto rdq = [&](int qrow) {
      const u32x2 lo = trd(qnew + qoff[0] + qrow * WGD_QROW), hi = trd(qnew + qoff[1] + qrow * WGD_QROW);
      qfp[R % 3][qrow] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    };
    rdq(0);
    pf[0] = rdp(0);
    pf[1] = rdp(1);
#ifdef WGD_EXP_DMA_FIRST   // timing experiment: all six requests at the start of the stage
#pragma unroll
    for (int j = 0; j < 6; ++j) issue1(rq, j);
#else
    issue1(rq, 0);
#endif
    rdq(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int prow = st / 3, kx = st - prow * 3;
      if (st + 2 < 18) pf[(st + 2) % 3] = rdp(st + 2);
      if (st == 2) rdq(2);
      if (st == 5) rdq(3);
#ifdef WGD_EXP_DMA_STAGGER   // timing experiment: the two row halves (SIMD partners) issue in different thirds of the stage
      if (hv == 0 ? (st == 1 || st == 2 || st == 3 || st == 4 || st == 5) : (st == 9 || st == 10 || st == 11 || st == 12 || st == 13))
        issue1(rq, hv == 0 ? st : st - 8);
#elif !defined(WGD_EXP_DMA_FIRST)
      if (st == 1 || st == 4 || st == 7 || st == 10 || st == 13) issue1(rq, (st + 2) / 3);
#endif
      const u32x4 a = pf[st % 3];
      if constexpr (K1F) {      // centre tap (ky = 1, kx = 1) of the 1x1x1 conv: P row prow against Q2 row prow - 1 of the same plane
        if (kx == 0 && prow >= 1 && prow <= 4) q2f = rdq2(prow - 1);        // (read one step ahead of its use)
        if (kx == 1 && prow >= 1 && prow <= 4) acc1 = Mfma16<T>::run(a, q2f, acc1);
      }
      // resident planes first: the new plane's fragments (kz 0) have had the longest to arrive by the time they are used
#pragma unroll
      for (int kz = 2; kz >= 0; --kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int qrow = prow - ky;
          if (qrow < 0 || qrow > 3) continue;
#ifdef WGD_EXP_NOMFMA   // timing experiment (wrong results): one matrix instruction per step keeps the reads alive
          if (kz != 0 || ky != (prow > 3 ? prow - 3 : 0)) continue;
#endif
          acc[(kz * 3 + ky) * 3 + kx] = Mfma16<T>::run(a, qfp[(kz + R) % 3][qrow], acc[(kz * 3 + ky) * 3 + kx]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int item = it0; item < it1; ++item) {
    setup(item);
    // ring slot of Q plane z: (z - zlo + 2) % 5; P plane z: (z - zlo + 1) % 3
    issue(zlo - 2, 0, 0, 0, false);
    issue(zlo - 1, 1, 0, 0, false);
    issue(zlo, 2, zlo - 1, 0, true);
    issue(zlo + 1, 3, zlo, 1, true);
    if constexpr (GNA) {      // (the previous item's last table reads lie behind its closing barrier; the barrier below publishes these)
      float* gsh = reinterpret_cast<float*>(lds + WGD_GA);
      if (tid < p.Cp && tid < 32) { gsh[tid] = p.ga.gamma[tid]; gsh[32 + tid] = p.ga.beta[tid]; }
      if (tid >= 64 && tid < 64 + p.ga.G) {
        gsh[tid] = p.ga.mean[ga_n * p.ga.G + tid - 64];
        gsh[32 + tid] = p.ga.rstd[ga_n * p.ga.G + tid - 64];
      }
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // everything but the last stage's worth has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (GNA) { ga_apply(0, zlo - 1); ga_fence(); }
    // the first stage's resident planes (zlo - 1, zlo - 2: outside the item = zeros): roles kz 1, 2 at rot 0
#pragma unroll
    for (int qrow = 0; qrow < 4; ++qrow) qfp[1][qrow] = qfp[2][qrow] = u32x4{0u, 0u, 0u, 0u};
    int pb_i = 0, qs = 0, rot = 0;                         // P buffer of plane zp, ring slot of Q plane zp - 1, register rotation
    for (int zp = zlo - 1; zp <= zhi; ++zp) {      // stage: P plane zp against Q planes zp + 1 (kz 0), zp (kz 1), zp - 1 (kz 2)
      int pn = pb_i + 2; if (pn >= WGD_NPB) pn -= WGD_NPB;
      int qn = qs + 4; if (qn >= WGD_NQS) qn -= WGD_NQS;
      const StageReq rq = make_req(zp + 3, qn, zp + 2, pn, zp + 2 <= zhi);     // two stages ahead (past the item's end: zeros / fillers)
      const unsigned char* pb = lds + pb_i * WGD_PPL;
      int sl = qs + 2; if (sl >= WGD_NQS) sl -= WGD_NQS;
      const unsigned char* qnew = lds + sl * WGD_QPL;
      // K1F: this stage's Q2 plane zp sits in the slot of Q plane zp - 1 (qs); plane zp + 1 is requested NOW into the slot of Q plane zp
      // (dead since the barrier that ended the stage before: Q planes are read from LDS once, in the stage they are new)
      if constexpr (K1F) { int q2n = qs + 1; if (q2n >= WGD_NQS) q2n -= WGD_NQS; issue_q2(zp + 1, q2n); }
      stage(WgdIC<0>{}, pb, qnew, rq, lds + qs * WGD_QPL);
#pragma unroll
      for (int qrow = 0; qrow < 4; ++qrow) { qfp[2][qrow] = qfp[1][qrow]; qfp[1][qrow] = qfp[0][qrow]; }
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // the next stage's planes have landed; this stage's requests stay in flight
#ifndef WGD_EXP_NOBAR   // timing experiment (wrong results): no stage barrier
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      if constexpr (GNA) {      // the next stage's P plane (zp + 1) has landed: normalise it before anyone reads it
        if (zp < zhi) { ga_apply(pb_i + 1 == WGD_NPB ? 0 : pb_i + 1, zp + 1); }
        ga_fence();
      }
      if (++pb_i == WGD_NPB) pb_i = 0;
      if (++qs == WGD_NQS) qs = 0;
      rot = rot == 0 ? 2 : rot - 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (fillers of the last stage: nothing may land after the buffers change hands)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // ---- the two row halves meet in LDS; partial slab [tap][32 cin][32 cout] ----
  float* xl = reinterpret_cast<float*>(lds);
  const int row0 = wa * 16 + 4 * g, col = wb * 16 + (lane & 15);
  if (hv == 1) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xl[(t * 32 + row0 + r) * 32 + col] = acc[t][r];
  }
  __syncthreads();
  if (hv == 0) {
    float* pbw = p.part + (((long)blockIdx.x * p.ncp + cpt) * p.ncq + cqt) * (27L * 1024);
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) pbw[(t * 32 + row0 + r) * 32 + col] = acc[t][r] + xl[(t * 32 + row0 + r) * 32 + col];
  }
  if constexpr (K1F) {      // the 1x1x1 tap: the same meeting of the two row halves, one more [32][32] slab per (workgroup, cp, cq)
    __syncthreads();
    if (hv == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xl[(row0 + r) * 32 + col] = acc1[r];
    }
    __syncthreads();
    if (hv == 0) {
      float* pb1 = p.part2 + (((long)blockIdx.x * p.ncp + cpt) * p.ncq + cqt) * 1024L;
#pragma unroll
      for (int r = 0; r < 4; ++r) pb1[(row0 + r) * 32 + col] = acc1[r] + xl[(row0 + r) * 32 + col];
    }
  }
#endif
}

// =====================================================================================================================
// plan + launch (the finalize is lowp.hip's)
// =====================================================================================================================
struct WgdPlan { int ntx, nty, nzc, ZC, nitems, ipw, nwg, ncp, ncq, xcd; };
static bool wgd_enabled() {   // BTS_LP_WGD=0: these layers back on lowp.hip's general weight-gradient kernel (A/B; read per call)
  const char* e = getenv("BTS_LP_WGD");
  return !(e && atoi(e) == 0);
}
static bool wgd_plan(WgdPlan& pl, int N, int D, int H, int W, int Cp, int ldp, int Cq, int ldq) {
  if (!wgd_enabled() || Cq % 8 != 0 || Cp % 8 != 0 || W % WGD_TX != 0 || H % WGD_TY != 0 || D < 4) return false;
  if ((long)D * H * W * (long)ldp * 2 >= 0x7fffffffL || (long)D * H * W * (long)ldq * 2 >= 0x7fffffffL) return false;
  pl.ncp = (Cp + 31) / 32;
  pl.ncq = (Cq + 31) / 32;
  pl.ntx = W / WGD_TX; pl.nty = H / WGD_TY;
  const long ncol = (long)N * pl.ntx * pl.nty;
  if (ncol * D < 64) return false;      // (fewer than 64 plane stages: nothing to stream)
  int cus = 256 / (pl.ncp * pl.ncq);
  if (cus < 8) cus = 8;
  int nzc = 1;
  while (ncol * nzc < 2L * cus && (D + 2 * nzc - 1) / (2 * nzc) >= 8) nzc *= 2;
  pl.nzc = nzc;
  pl.ZC = (D + nzc - 1) / nzc;
  pl.nzc = (D + pl.ZC - 1) / pl.ZC;
  const long items = ncol * pl.nzc;
  if (items > 0x3fffffffL) return false;
  pl.nitems = (int)items;
  int nwg = pl.nitems < cus ? pl.nitems : cus;
  pl.ipw = (pl.nitems + nwg - 1) / nwg;
  pl.nwg = (pl.nitems + pl.ipw - 1) / pl.ipw;
  pl.xcd = (pl.nwg % 8 == 0 && pl.nwg * pl.ipw == pl.nitems) ? 1 : 0;
  return true;
}
// bytes of partial slabs a call with these dimensions needs, or 0 when it is declined
long bts_lp_wgd_workspace_(int N, int D, int H, int W, int Cp, int Cq) {
  WgdPlan pl;
  if (!wgd_plan(pl, N, D, H, W, Cp, Cp, Cq, Cq)) return 0;
  return (long)pl.nwg * pl.ncp * pl.ncq * 27 * 1024 * 4;
}
// does the kernel take the shape with GroupNorm `in_G` (+ReLU) applied to its P planes (LpGnaFuse)?
bool bts_lp_wgd_gna_ok_(int N, int D, int H, int W, int Cp, int Cq, int in_G) {
  WgdPlan pl;
  if (in_G <= 0 || in_G > 32 || D % in_G != 0 || Cp % in_G != 0 || Cp > 32 || !wgd_plan(pl, N, D, H, W, Cp, Cp, Cq, Cq)) return false;
  const int cg = Cp / in_G;
  return cg <= 8 && 8 % cg == 0;
}
// BTS_OK = ran (dw written by the shared finalize), 1 = declined
// dy2 / dw1 / lddy2 (may be NULL / 0): the K1F form -- dw1 (Keras layout (1,1,1,Cin_ref,Cout)) (+)= the 1x1x1 weight gradient of a second conv
// on the same input, from dy2 (N,D,H,W,Cq); its partial slabs sit behind the 27-tap ones in `ws` (bts_lp_wgd_workspace_ x 28 / 27)
int bts_lp_wgd_launch_(int dtype, const void* x, const void* dy, float* dw, void* ws, long ws_bytes, int N, int D, int H, int W, int Cp, int ldp,
                       int Cq, int ldq, int dup_start, int dup_shift, int accum, hipStream_t stream, const LpGnaFuse* ga, const void* dy2,
                       float* dw1, int lddy2, long psplit) {
  WgdPlan pl;
  if (!wgd_plan(pl, N, D, H, W, Cp, ldp, Cq, ldq)) return 1;
  if (psplit != 0 && (psplit < 0 || ga != nullptr || Cp % 32 != 0 || ldp < 32 || psplit % 8 != 0)) return 1;
  if (ga != nullptr && (ldp != Cp || dup_shift != 0 || !bts_lp_wgd_gna_ok_(N, D, H, W, Cp, Cq, ga->G) || ga->cg != Cp / ga->G)) return 1;
  const bool k1f = dy2 != nullptr;
  if (k1f && (ga != nullptr || dw1 == nullptr || lddy2 < Cq || lddy2 % 8 != 0 || (((uintptr_t)dy2) & 15) ||
              (long)D * H * W * (long)lddy2 * 2 >= 0x7fffffffL))
    return 1;
  if (ws_bytes < (long)pl.nwg * pl.ncp * pl.ncq * (k1f ? 28 : 27) * 1024 * 4) return 1;
  LpWgdParams p;
  p.p = (const unsigned short*)x; p.q = (const unsigned short*)dy; p.part = reinterpret_cast<float*>(ws);
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cp = Cp; p.ldp = ldp; p.Cq = Cq; p.ldq = ldq;
  p.ntx = pl.ntx; p.nty = pl.nty; p.nzc = pl.nzc; p.ZC = pl.ZC; p.nitems = pl.nitems; p.ipw = pl.ipw; p.ncp = pl.ncp; p.ncq = pl.ncq; p.xcd_order = pl.xcd;
  if (ga != nullptr) { p.ga = *ga; p.ga_zt = D / ga->G; } else { p.ga = LpGnaFuse{}; p.ga_zt = 1; }
  p.q2 = (const unsigned short*)dy2; p.ldq2 = lddy2; p.psplit = psplit;
  p.part2 = p.part + (long)pl.nwg * pl.ncp * pl.ncq * 27 * 1024;
  (void)hipGetLastError();
#define WGD_LAUNCH(TT) do { if (ga != nullptr) WGD_LAUNCH_(TT, true, false); else if (k1f) WGD_LAUNCH_(TT, false, true); else WGD_LAUNCH_(TT, false, false); } while (0)
#define WGD_LAUNCH_(TT, GA_, K1_)                                                                                            \
  do {                                                                                                                       \
    auto kern = lp_wgd_kernel<TT, GA_, K1_>;                                                                                   \
    static bool done = false;                                                                                                \
    if (!done) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WGD_LDS); \
      if (e != hipSuccess) return (int)e;                                                                                    \
      done = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL(kern, dim3(pl.nwg, pl.ncp, pl.ncq), dim3(512), WGD_LDS, stream, p);                                           \
  } while (0)
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(37, 2.0 * (k1f ? 28.0 : 27.0) * (double)Cp * Cq * (double)N * D * H * W, stream);
  if (dtype == LP_F16) WGD_LAUNCH(TF16); else WGD_LAUNCH(TBF16);
#undef WGD_LAUNCH
#undef WGD_LAUNCH_
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  const int r = bts_lp_wgrad_finalize_(p.part, dw, pl.nwg, pl.ncp, pl.ncq, 27, 27, 1, Cp, Cq, Cp + dup_shift, dup_start, dup_shift, accum, stream);
  if (r != BTS_OK || !k1f) return r;
  return bts_lp_wgrad_finalize_(p.part2, dw1, pl.nwg, pl.ncp, pl.ncq, 1, 1, 1, Cp, Cq, Cp + dup_shift, dup_start, dup_shift, accum, stream);
}
