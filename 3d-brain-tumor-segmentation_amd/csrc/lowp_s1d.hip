// 3x3x3 stride-1 'same' convolution on 16-bit storage, LDS-DMA staged (round 3).  Replaces, for every shape it takes, the
// register-staged lp_conv_s1_kernel of lowp.hip: the Conv3D of resnet.py:80-87 (both convs of a ResnetBlock), decoder.py / vae.py
// blocks, and -- on role-swapped weight images -- their data gradients (train.py:142-151 under TF autodiff).
//
// Why a second kernel: the first one fetched every weight fragment per wave straight from L2 (27 KB per wave and k-step, 2.9 scalar
// + 3.8 vector instructions per matrix instruction of run-time address arithmetic, halo tile global -> registers -> LDS) and sat at
// 0.25-0.27 of the 2.5 PF dense peak with its waves parked 64 % of the time (profiles/r02b_pmc_lp_s1.txt).  Here
//   * BOTH operands reach LDS by buffer_load_dwordx4 ... lds (no staging registers, no ds_write pass): the halo tile of one k-step
//     (16 input channels) as [voxel][2 x 8 channels] with the two 16-byte halves of a voxel swapped where bit 3 of its x is set --
//     two neighbouring lanes fetch one voxel's 32 contiguous bytes (one L2 request per line; k-half planes fetched by separate
//     requests doubled the L2 request count, which is what bounds the 128^3 layers), and a B fragment (32 consecutive x of one
//     k-half) is ONE conflict-free ds_read_b128 at a compile-time offset from one of three per-lane bases (one per x tap: the swap
//     depends on x only, rows and planes are whole multiples of 32 bytes); out-of-image voxels use an offset outside the buffer
//     descriptor (the DMA then writes zeros: 'same' padding without a branch);
//   * the weights of a STAGE (k-step, dz: 9 taps x 16 cin x 32|64 couts = 9|18 KB) are copied verbatim from a packed image laid out
//     as the LDS image [tap][k-half][cout][8 cin] and shared by the 8 waves (A fragment = one ds_read_b128 at an immediate offset);
//   * 512 threads: wave = (z plane, g) with g = y half of a 32x8x4 tile (32-cout items) or cout block of a 32x4x4 tile (64-cout
//     items); per (dz, dx) a wave reads 6 input rows + 3 weight fragments for 12 matrix instructions (rows serve the three dy taps);
//   * a three-slot weight ring and a double-buffered halo tile are filled two stages / one k-step ahead; a stage boundary is
//     `s_waitcnt vmcnt(N)` with N counted (never 0) + one s_barrier; a workgroup walks its items (tile x cout group, XCD-aware
//     order) as ONE stage stream, the next item's operands in flight under the current item's output side.
// Declines (caller runs the old kernel): W < 12, fewer than 4096 voxels, Cout % 8 != 0, offsets beyond 31 bits.
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
long bts_lp_splitk_gn_B_(int N, long V, int Cout, int G);
int bts_lp_splitk_reduce_gn_(int dtype, const float* part, const float* bias, void* y, double* gn_partial, int N, long V, int Cout, int G,
                             int ksplit, hipStream_t stream);
int bts_lp_splitk_reduce_(int dtype, const float* part, const float* bias, void* y, long nvox, int Cout, int Npad, int ldy, int ksplit,
                          int accum, hipStream_t stream);

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct LpS1dParams {
  const unsigned short* x;
  const unsigned short* wp;   // the DMA part of the K3S1 image: [cout group][k-step][dz][dy*3+dx][k-half][cout in group][8 cin]
  const float* bias;
  unsigned short* y;
  int N, D, H, W, ldx, ldy, Cout, KS, NB;
  int ntx, nty, ntz, ncg;
  long nitems;
  // item order: cout group fastest, then the tiles of a block of bx x by x bz tiles (x fastest), then the blocks (x fastest), then
  // the samples -- a block is what the 32 workgroups of an XCD hold at a time, so the halo voxels its tiles share are fetched from
  // HBM once and found in that XCD's L2 by the neighbours.  Divisions by run-time constants as multiply-high + shift.
  int dbg;          // timing experiments (builds with -DBTS_TIMING_EXPERIMENTS only): 1 no output stores, 2 no halo traffic, 4 no matrix instructions,
                    // 8 no fragment reads from LDS, 16 no weight traffic
  int bx, by, bz_;
  unsigned bvol_, nbx_, nby_, nbz_;
  unsigned dv_mul[7], dv_sh[7];    // divisors: ncg, bx*by*bz, bx, by, ntx/bx, nty/by, ntz/bz
  int ksplit, ks_per;
  int accum;
  float* part;       // split-K: fp32 partial sums [split][voxel][NB*32]
  double* gnp;       // fused GroupNorm partial sums (slab semantics) [N*G][gn_B][2], or NULL
  int gn_G, gn_zt;
  long gn_B;
  // SC kernels: a second contraction at the CENTRE tap only -- y += x2 (N,D,H,W,K) . w2, the 1x1x1 image [k-step][cout block][k-half][32][8]
  // (lowp.hip's first part of a K1 image) -- the shortcut conv's data gradient riding on conv1's (resnet.py:96-103 / 80-87 under train.py:151)
  const unsigned short* x2;
  const unsigned short* wp2;
  int ldx2;
  // split output (0 = off): the columns of cout block cb go to y + cb * ysplit (elements) as a tensor of its own with voxel stride ldy --
  // the 64-wide gradient of the decoder's [skip | up-sampled] concat (decoder.py:75) leaves as two DENSE 32-channel tensors, so that its
  // readers (the skip level's block backward, the up-sampler's GroupNorm backward) do not fetch 64-byte halves of 128-byte lines
  long ysplit;        // elements between the tensors of consecutive cout blocks (0: one tensor)
  int ycol;           // columns a cout block advances inside its tensor: 32 (one tensor) | 0 (split)
};

template <int MODE, int TXL>
struct S1dGeo {
  static constexpr int TX = 1 << TXL, ZP = 32 / TX;          // x extent of a fragment, z planes per fragment
  static constexpr int CBW = MODE ? 2 : 1;                   // cout blocks of 32 per item
  static constexpr int YG = MODE ? 1 : 2;                    // groups of 4 output rows per tile
  static constexpr int TY = 4 * YG, TZ = 4 * ZP;
  static constexpr int SX = TX + 2, SY = TY + 2, SZ = TZ + 2;
  // plane stride in voxels: two-plane fragments (TX = 16) need it a multiple of 16 voxels = 256 bytes, so that the two 256-byte
  // runs of a 16-byte-per-lane read land on the same bank phase (MI355X LDS: ds_read_b128 is served in 16-lane groups)
  static constexpr int PS = (ZP == 1) ? SY * SX : ((SY * SX + 15) / 16) * 16;
  static constexpr int NVOX = SZ * PS;
  static constexpr int NCH = ((NVOX + 255) / 256) * 8;       // 1 KB DMA chunks (32 voxels x 32 bytes) per k-step, whole rounds of the 8 waves
  static constexpr int HBUF = NCH * 1024;
  static constexpr int NH = NCH / 8;                         // halo requests per wave and k-step
  static constexpr int NHA = (NH + 1) / 2, NHB = NH - NHA;   // issued in stage 0 / stage 1
  static constexpr int WTAP = CBW * 1024, WSTAGE = 9 * WTAP;
  static constexpr int NWC = 9 * CBW, NW = (NWC + 7) / 8;    // weight chunks per stage / per wave
  static constexpr int OFF_W = 2 * HBUF, OFF_SCR = OFF_W + 3 * WSTAGE, OFF_BIAS = OFF_SCR + 1024;
  static constexpr int LDS_BYTES = OFF_BIAS + 2 * 256;       // two bias slots of 64 floats (a 4-byte DMA writes all 64 lanes)
  static constexpr int NST = 8;                              // output-side store instructions per wave and item (16 bytes per lane)
};

template <int N> __device__ __forceinline__ void s1d_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void s1d_barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <typename T, int MODE, int TXL, bool SC = false>
__global__ __launch_bounds__(512, 2) void lp_s1d_kernel(const LpS1dParams p) {
  // (the host pass of hipcc 7.2 silently drops this template's launch stub when it instantiates the body -- the array-indexed
  // LDS-DMA offsets trigger it -- so the body exists in the device pass only; there is no other code path)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef S1dGeo<MODE, TXL> G;
  constexpr int TX = G::TX, ZP = G::ZP, CBW = G::CBW, SX = G::SX, SY = G::SY, PS = G::PS, NVOX = G::NVOX;
  constexpr int NH = G::NH, NHA = G::NHA, NHB = G::NHB, NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  const int lx = l32 & (TX - 1), lz = l32 >> TXL;
  const int zz = wave & 3, g = wave >> 2;
  const int y0 = MODE ? 0 : 4 * g;        // first output row of this wave inside the tile
  const int cbw = MODE ? g : 0;           // cout block of this wave inside the item

  // ---- item walk: (n, tz, ty, tx, cg), cg fastest; XCD k walks its own contiguous eighth, its workgroups interleaved ----
  unsigned it, it_end, it_step;      // (32-bit: the decode below runs once per item on the scalar unit)
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
  struct Item { int cg, n, ox0, oy0, oz0; };
  auto fdiv = [&](unsigned n, int k, unsigned d, unsigned& rem) -> unsigned {      // n < 2^31
    const unsigned q = p.dv_mul[k] ? (__umulhi(n, p.dv_mul[k]) >> p.dv_sh[k]) : n;
    rem = n - q * d;
    return q;
  };
  auto decode = [&](unsigned i) {
    Item t;
    unsigned r, lx_, ly_, lz_, Bx, By, Bz;
    unsigned q = fdiv(i, 0, (unsigned)p.ncg, r);
    t.cg = (int)r;
    unsigned blk = fdiv(q, 1, p.bvol_, r);             // r = tile inside the block
    unsigned rz = fdiv(r, 2, (unsigned)p.bx, lx_);     // rz = ly + by * lz
    lz_ = fdiv(rz, 3, (unsigned)p.by, ly_);
    unsigned b2 = fdiv(blk, 4, p.nbx_, Bx);
    unsigned b3 = fdiv(b2, 5, p.nby_, By);
    t.n = (int)fdiv(b3, 6, p.nbz_, Bz);
    t.ox0 = (int)(Bx * p.bx + lx_) * TX;
    t.oy0 = (int)(By * p.by + ly_) * G::TY;
    t.oz0 = (int)(Bz * p.bz_ + lz_) * G::TZ;
    return t;
  };
  int ks0 = 0, ks1 = p.KS;
  if (p.ksplit > 1) {
    ks0 = blockIdx.y * p.ks_per;
    ks1 = ks0 + p.ks_per;
    if (ks1 > p.KS) ks1 = p.KS;
  }

  // ---- DMA side ----
  // request r of a wave fills the 1 KB chunk r*8 + wave of the halo buffer: 32 voxels, lanes (2i, 2i+1) the two 16-byte slots of
  // voxel i; slot c of a voxel at halo column x holds k-half c ^ bit3(x)
  unsigned hrel[NH], hcrd[NH];
#pragma unroll
  for (int r = 0; r < NH; ++r) {
    const int vox = (r * 8 + wave) * 32 + (lane >> 1);
    const int vz = vox / PS, rem = vox - vz * PS;
    const int vy = rem / SX, vx = rem - vy * SX;
    const int hp = (lane & 1) ^ ((vx >> 3) & 1);
    const bool geo = vox < NVOX && rem < SY * SX;
    hrel[r] = (unsigned)(((vz * p.H + vy) * p.W + vx) * p.ldx * 2 + hp * 16);
    hcrd[r] = (unsigned)(vx | (vy << 8)) | (geo ? (unsigned)vz << 16 : 0xffff0000u);   // padding voxels: a z no image reaches
  }
  unsigned hoff[NH];
  __amdgpu_buffer_rsrc_t xr;
  auto dma_item = [&](const Item& t, bool live) {   // halo origin + per-chunk offsets of the item the NEXT k-step belongs to
    const unsigned short* xorg = p.x + ((((long)t.n * p.D + (t.oz0 - 1)) * p.H + (t.oy0 - 1)) * p.W + (t.ox0 - 1)) * (long)p.ldx;
    xr = __builtin_amdgcn_make_buffer_rsrc((void*)xorg, 0, 0x7fffffff, 0x00020000);
    const int zb = (live && !(BTS_DBG(p) & 2)) ? t.oz0 - 1 : 0x100000;       // no next item: every voxel out of range (zeros, no traffic)
#pragma unroll
    for (int r = 0; r < NH; ++r) {
      const int vx = hcrd[r] & 0xff, vy = (hcrd[r] >> 8) & 0xff, vz = hcrd[r] >> 16;
      const bool ok = (unsigned)(zb + vz) < (unsigned)p.D && (unsigned)(t.oy0 - 1 + vy) < (unsigned)p.H &&
                      (unsigned)(t.ox0 - 1 + vx) < (unsigned)p.W;
      hoff[r] = ok ? hrel[r] : 0x80000000u;
    }
  };
  // one request each (the k-step body below deals them out between its matrix instructions)
  auto issue_halo1 = [&](int r, int ks, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(lds + buf * G::HBUF + (r * 8 + wave) * 1024), 16, hoff[r],
                                             (unsigned)ks * 32u, 0, 0);
  };
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  // weight chunk c = r*8 + wave of stage (cg, ks, dz) -> ring slot dz; chunks past the stage's 9*CBW go to the scratch KB with an
  // out-of-range offset (every wave issues the same number of requests: the vmcnt counts below are compile-time constants)
  unsigned wvo[NW];
  int wdst[NW];
#pragma unroll
  for (int r = 0; r < NW; ++r) {
    const int c = r * 8 + wave;
    wvo[r] = c < G::NWC ? (unsigned)(c * 1024 + lane * 16) : 0x80000000u;
    wdst[r] = c < G::NWC ? G::OFF_W + c * 1024 : G::OFF_SCR;
  }
  auto w_soff = [&](int cg, int ks, int dz, bool live) -> unsigned {
    if (BTS_DBG(p) & 16) return 0x80000000u;
    return live ? (unsigned)((((cg * p.KS + ks) * 3) + dz) * G::WSTAGE) : 0x80000000u;    // (no next item: nothing to fetch)
  };
  auto issue_w1 = [&](int r, unsigned soff, int dz) {
    const int c = r * 8 + wave;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(lds + wdst[r] + (c < G::NWC ? dz * G::WSTAGE : 0)), 16, wvo[r], soff, 0, 0);
  };
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
  auto issue_bias = [&](int cg, int slot) {   // every wave writes the same 32*CBW values (benign duplicates, uniform request counts)
    const int co = cg * CBW * 32 + lane;
    const unsigned off = (lane < 32 * CBW && co < p.Cout) ? (unsigned)co * 4u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (lds_ptr_t)(lds + G::OFF_BIAS + slot * 256), 4, off, 0, 0, 0);
  };

  // ---- SC: the centre-tap operand pair of a k-step, global -> registers (no LDS: a lane's B fragment is its own voxel's 16 bytes) ----
  // B2[v]: voxel (z, y0 + v, lx) of the tile, channels 16 ks + 8 h .. + 7 of x2; A2: couts 32 cb + l32, the same 8 channels of w2.
  // Requested in stage 1 of the k-step BEFORE, right after that k-step's own pair has been multiplied (one register set), used after the
  // centre group of stage 1.
  u32x4 A2, B2[4];
  unsigned sc_off[4], sc_aoff = 0x80000000u;
  __amdgpu_buffer_rsrc_t x2r, w2r;
  auto sc_item = [&](const Item& t, bool live) {
    if constexpr (SC) {
      x2r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 + (long)t.n * p.D * p.H * p.W * (long)p.ldx2), 0, 0x7fffffff, 0x00020000);
      const int oz = t.oz0 + zz * ZP + lz, ox = t.ox0 + lx;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int oy = t.oy0 + y0 + v;
        const bool ok = live && oz < p.D && oy < p.H && ox < p.W;
        sc_off[v] = ok ? (unsigned)((((oz * p.H + oy) * p.W + ox) * p.ldx2 + 8 * h) * 2) : 0x80000000u;
      }
      const int cb = t.cg * CBW + cbw;
      sc_aoff = (live && cb < p.NB) ? (unsigned)(cb * 1024 + h * 512 + l32 * 16) : 0x80000000u;
    }
  };
  auto sc_issue = [&](int ks) {
    if constexpr (SC) {
#pragma unroll
      for (int v = 0; v < 4; ++v) B2[v] = __builtin_amdgcn_raw_buffer_load_b128(x2r, sc_off[v], (unsigned)ks * 32u, 0);
      A2 = __builtin_amdgcn_raw_buffer_load_b128(w2r, sc_aoff, (unsigned)(ks * p.NB) * 1024u, 0);
    }
  };
  if constexpr (SC) w2r = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp2, 0, 0x7fffffff, 0x00020000);

  // ---- compute side ----
  // A k-step is nine groups (dz, dx) of 12 matrix instructions: 6 input rows (they serve the three dy taps of the wave's four output
  // rows) + 3 weight fragments.  The fragments of group g+1 are read from LDS while group g multiplies (two register sets; the
  // compiler, left alone, reads each fragment right before its first use and exposes the LDS latency 27 times per k-step), the rows
  // of a stage's first group already before the stage barrier (the halo tile is complete since stage 0; only the weights are new).
  unsigned hbB[3];      // this lane's B-operand base for x tap dx: voxel (z, y0, lx + dx), physical slot of k-half h
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) hbB[dx] = (unsigned)((((zz * ZP + lz) * PS) + y0 * SX + lx + dx) * 32 + ((h ^ (((lx + dx) >> 3) & 1)) * 16));
  const unsigned wbA = (unsigned)(G::OFF_W + h * (CBW * 512) + cbw * 512 + l32 * 16);
  f32x16 acc[4];
  u32x4 Bc[6], Ac[3], Bn[6], An[3];
  auto ldB = [&](u32x4 (&B)[6], const unsigned char* hb, auto dzc, auto dxc) {     // hb = the halo buffer of this k-step
    constexpr int DZ = decltype(dzc)::value, DX = decltype(dxc)::value;
    if (BTS_DBG(p) & 8) { asm volatile("" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5])); return; }
#pragma unroll
    for (int j = 0; j < 6; ++j) B[j] = *reinterpret_cast<const u32x4*>(hb + hbB[DX] + ((DZ * PS) + j * SX) * 32);
  };
  auto ldA = [&](u32x4 (&A)[3], auto dzc, auto dxc) {
    constexpr int DZ = decltype(dzc)::value, DX = decltype(dxc)::value;
    if (BTS_DBG(p) & 8) { asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2])); return; }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) A[dy] = *reinterpret_cast<const u32x4*>(lds + wbA + DZ * G::WSTAGE + (dy * 3 + DX) * G::WTAP);
  };
  // 12 matrix instructions; hook(0..2) runs after each dy's four (one LDS-DMA request each: issued between matrix instructions a
  // request costs its issue slot, issued in a block at the head of a stage it cost the matrix pipe ~100 cycles apiece)
  auto mm = [&](const u32x4 (&A)[3], const u32x4 (&B)[6], auto&& hook) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (!(BTS_DBG(p) & 4)) {
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = T::mfma(A[dy], B[v + dy], acc[v]);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) asm volatile("" ::"v"(A[dy]), "v"(B[v + dy]));
      }
      hook(dy);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  Item ci = decode(it);
  dma_item(ci, true);
  int ipar = 0, buf = 0;
  // prologue: the request order of a steady-state k-step's tail (stage 0 below waits for everything but the last weight stage)
#pragma unroll
  for (int r = 0; r < NHA; ++r) issue_halo1(r, ks0, 0);
#pragma unroll
  for (int r = 0; r < NW; ++r) issue_w1(r, w_soff(ci.cg, ks0, 0, true), 0);
#pragma unroll
  for (int r = NHA; r < NH; ++r) issue_halo1(r, ks0, 0);
  issue_bias(ci.cg, 0);
  sc_item(ci, true);
  sc_issue(ks0);
#pragma unroll
  for (int r = 0; r < NW; ++r) issue_w1(r, w_soff(ci.cg, ks0, 1, true), 1);
  bool after_out = false;
  for (;;) {
    const unsigned nit = it + it_step;
    const bool have_next = nit < it_end;
    const Item ni = have_next ? decode(nit) : ci;
    for (int ks = ks0; ks < ks1; ++ks) {
      const bool last = ks + 1 == ks1;
      const bool nlive = !last || have_next;
      const int ncg_ = last ? ni.cg : ci.cg, nks = last ? ks0 : ks + 1;
      const unsigned char* hb = lds + (buf ? G::HBUF : 0);
      const int nbuf = buf ^ 1;
      // ---- stage 0 (dz = 0): requests W(ks, 2) then the first half of the next k-step's halo tile ----
      // (SC: the centre-tap pair of THIS k-step, requested after the second group of the stage 1 before, may still be in flight here -- it
      // sits behind every request this stage needs and ahead of W(ks, 1), so the stage-1 wait below covers it)
#ifdef S1D_SC_LATE      // A/B build (make variantf FILE=lowp_s1d NAME=sclate EXTRA=-DS1D_SC_LATE): the pair requested in stage 2, as first built
      constexpr int NSC = 0;
#else
      constexpr int NSC = SC ? 5 : 0;
#endif
      if (after_out) s1d_wait<NW + NSC + G::NST>(); else s1d_wait<NW + NSC>();
      s1d_barrier();
      after_out = false;
      ldB(Bc, hb, I0(), I0()); ldA(Ac, I0(), I0());
      if (ks == ks0) {     // the accumulators start at the bias (split-K partial sums: at zero; the reduce kernel adds it)
        const float* bsh = reinterpret_cast<const float*>(lds + G::OFF_BIAS + ipar * 256) + cbw * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 bq = *reinterpret_cast<const f32x4*>(bsh + 8 * q);
          if (p.ksplit > 1) bq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[v][4 * q + j] = bq[j];
        }
      }
      if (last) dma_item(ni, have_next);
      const unsigned so0 = w_soff(ci.cg, ks, 2, true);
      ldB(Bn, hb, I0(), I1()); ldA(An, I0(), I1());
      __builtin_amdgcn_sched_barrier(0);
      mm(Ac, Bc, [&](int i) { if (i < NW) issue_w1(i, so0, 2); else if (i - NW < NHA) issue_halo1(i - NW, nks, nbuf); });
      ldB(Bc, hb, I0(), I2()); ldA(Ac, I0(), I2());
      __builtin_amdgcn_sched_barrier(0);
      mm(An, Bn, [&](int i) { const int j = i + 3; if (j < NW) issue_w1(j, so0, 2); else if (j - NW < NHA) issue_halo1(j - NW, nks, nbuf); });
      ldB(Bn, hb, I1(), I0());
      __builtin_amdgcn_sched_barrier(0);
      mm(Ac, Bc, [&](int i) { const int j = i + 6; if (j < NW) issue_w1(j, so0, 2); else if (j - NW < NHA) issue_halo1(j - NW, nks, nbuf); });
      static_assert(NW + NHA <= 9 && NW + NHB + 1 <= 9, "a stage has nine request slots");
      // ---- stage 1: W(next k-step, 0), the second half of its halo tile, its bias ----
      s1d_wait<NW + NHA>();
      s1d_barrier();
      ldA(An, I1(), I0());
      const unsigned so1 = w_soff(ncg_, nks, 0, nlive);
      const int bslot = last ? (ipar ^ 1) : ipar;
      auto hook1 = [&](int j) {
        if (j < NW) issue_w1(j, so1, 0);
        else if (j - NW < NHB) issue_halo1(NHA + j - NW, nks, nbuf);
        else if (j - NW == NHB) issue_bias(ncg_, bslot);
      };
      ldB(Bc, hb, I1(), I1()); ldA(Ac, I1(), I1());
      __builtin_amdgcn_sched_barrier(0);
      mm(An, Bn, [&](int i) { hook1(i); });
      ldB(Bn, hb, I1(), I2()); ldA(An, I1(), I2());
      __builtin_amdgcn_sched_barrier(0);
      mm(Ac, Bc, [&](int i) { hook1(i + 3); });
      if constexpr (SC) {      // the shortcut's k-step: centre tap only
        if (!(BTS_DBG(p) & 4)) {
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = T::mfma(A2, B2[v], acc[v]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ... and the NEXT k-step's pair right away, into the registers just read: behind all of this stage's requests (they went out with
        // the first two groups), a stage ahead of W(next, 1) -- two and a third stages of flight time before the matrix pipe asks for it
        // (requested in stage 2, ahead of a wait one third of a k-step later, the pair cost the 128^3 launch 17 %: HBM latency exposed)
        static_assert(NW + NHB + 1 <= 6, "the stage's requests leave with its first two groups");
#ifndef S1D_SC_LATE
        if (last) sc_item(ni, have_next);
        sc_issue(nks);
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      ldB(Bc, hb, I2(), I0());
      __builtin_amdgcn_sched_barrier(0);
      mm(An, Bn, [&](int i) { hook1(i + 6); });
      // ---- stage 2: W(next k-step, 1) ----
      s1d_wait<NHA + NW + NHB + 1 + NSC>();
      s1d_barrier();
      ldA(Ac, I2(), I0());
      const unsigned so2 = w_soff(ncg_, nks, 1, nlive);
#ifdef S1D_SC_LATE
      if constexpr (SC) {
        if (last) sc_item(ni, have_next);
        sc_issue(nks);
      }
#endif
      ldB(Bn, hb, I2(), I1()); ldA(An, I2(), I1());
      __builtin_amdgcn_sched_barrier(0);
      mm(Ac, Bc, [&](int i) { if (i < NW) issue_w1(i, so2, 1); });
      ldB(Bc, hb, I2(), I2()); ldA(Ac, I2(), I2());
      __builtin_amdgcn_sched_barrier(0);
      mm(An, Bn, [&](int i) { if (i + 3 < NW) issue_w1(i + 3, so2, 1); });
      mm(Ac, Bc, [&](int) {});
      buf ^= 1;
    }
    // ---- output side of `ci` ----
    {
      const int oy = ci.oy0 + y0, ox = ci.ox0 + lx, oz = ci.oz0 + zz * ZP + lz;
      const int cb = ci.cg * CBW + cbw;
      if (p.ksplit > 1) {   // raw fp32 partial sums [split][voxel][NB*32]; bias / rounding happen in the reduce kernel
        const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.part + (((long)blockIdx.y * p.N + ci.n) * p.D * p.H * p.W) * (long)(p.NB * 32)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const bool ok = cb < p.NB && oz < p.D && oy + v < p.H && ox < p.W;
            const unsigned off = ok ? (unsigned)((((oz * p.H + oy + v) * p.W + ox) * (p.NB * 32) + cb * 32 + 8 * q + 4 * h) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(u32x4, f32x4{acc[v][4 * q], acc[v][4 * q + 1], acc[v][4 * q + 2], acc[v][4 * q + 3]}), pr, off, 0, 0);
          }
      } else {
        // (SC kernels: a split output moves the BASE of the wave's cout block -- a run-time scalar offset operand on these loads / stores
        // made the fp16 variants produce garbage, in the block the compiler already mis-handled once, see below)
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.y + (long)ci.n * p.D * p.H * p.W * (long)p.ldy + (SC ? (long)cb * p.ysplit : 0L)), 0, 0x7fffffff, 0x00020000);
        const bool gn_on = p.gnp != nullptr;
        float gn_s = 0.f, gn_q = 0.f;
        // The matrix instruction leaves a lane (voxel, h) with couts 8q + 4h + {0..3}: four 8-byte pieces of the voxel's 64-byte
        // row.  v_permlane32_swap between the two lanes of a voxel (q even of the upper lane <-> q odd of the lower one) gives each
        // lane EIGHT consecutive couts = one 16-byte store, the pair of lanes 32 contiguous bytes: half the store instructions and
        // half the L2 write requests of the 8-byte form (the 128^3 layers are bound by L2 requests, not bytes).
        const int cb_col = SC ? cb * p.ycol : cb * 32;
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int co = cb * 32 + 16 * qp + 8 * h;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const bool oks = oz < p.D && oy + v < p.H && ox < p.W;
            const bool ok = co < p.Cout && oks;
            const unsigned off = (ok && !(BTS_DBG(p) & 1)) ? (unsigned)((((oz * p.H + oy + v) * p.W + ox) * p.ldy + cb_col + 16 * qp + 8 * h) * 2) : 0x80000000u;
            float f[4], g2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] = acc[v][8 * qp + j]; g2[j] = acc[v][8 * qp + 4 + j]; }
            if (p.accum) {     // old values arrive in the exchanged layout: the exchange is its own inverse
              u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(yr, off, 0, 0);
              asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                           : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
              float old[8];
              unpack8<T>(e, old);
#pragma unroll
              for (int j = 0; j < 4; ++j) { f[j] += old[j]; g2[j] += old[4 + j]; }
            }
            // (own values, before the exchange: this lane's couts 16 qp + 4 h + j / 16 qp + 8 + 4 h + j, NOT the stored piece's -- so the
            // gate is the voxel's validity only; columns >= Cout carry zero weights and a zero bias and add exact zeros.  Gating on the
            // stored piece's `ok` dropped real columns where Cout % 16 == 8: round-6 finding)
            if (gn_on && oks) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                gn_s += f[j] + g2[j];
                gn_q = fmaf(f[j], f[j], fmaf(g2[j], g2[j], gn_q));
              }
            }
            // (inline asm, pad inside the string: with the two-result builtin hipcc 7.2 dropped the second result in one of the
            // loop-unswitched copies of this block -- ragged launches came out wrong at couts 8k + 4..7)
            unsigned d0 = pack2<T>(f[0], f[1]), d1 = pack2<T>(f[2], f[3]), d2 = pack2<T>(g2[0], g2[1]), d3 = pack2<T>(g2[2], g2[3]);
            asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_nop"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{d0, d1, d2, d3}, yr, off, 0, LP_OUT_STORE_AUX);
          }
        }
        if (gn_on) {   // one fp64 (sum, sumsq) pair per (z plane, tile column, cout group, wave of that plane): fixed order
          const int ty = ci.oy0 / G::TY, tx = ci.ox0 / TX;
#pragma unroll
          for (int zl = 0; zl < ZP; ++zl) {
            const bool mine = (ZP == 1) || lz == zl;
            const double ds = wave_sum_f64(mine ? (double)gn_s : 0.0), dq = wave_sum_f64(mine ? (double)gn_q : 0.0);
            const int ozp = ci.oz0 + zz * ZP + zl;
            if (lane == 0 && ozp < p.D) {
              const int gg = ozp / p.gn_zt;
              const long slot = (((((long)(ozp - gg * p.gn_zt) * p.nty + ty) * p.ntx + tx) * p.ncg + ci.cg) * 2) + g;
              double* dst = p.gnp + (((long)ci.n * p.gn_G + gg) * p.gn_B + slot) * 2;
              dst[0] = ds;
              dst[1] = dq;
            }
          }
        }
      }
    }
    after_out = true;
    if (!have_next) break;
    ci = ni;
    it = nit;
    ipar ^= 1;
  }
  s1d_wait<0>();   // the last k-step's look-ahead requests (zeros into LDS) must not outlive the workgroup's LDS allocation
#endif
}

// =====================================================================================================================
// packed image of the DMA kernel: [cout group of CBW blocks][k-step][dz][dy*3+dx][k-half][cout in group][8 cin]
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void lp_s1d_pack_kernel(const LpPackParams p, int CBW, int NCG) {
  const long per_stage = 9L * CBW * 512;   // elements
  const long total = (long)NCG * p.KS * 3 * per_stage;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) lp_s1d_pack_elem<T>(p, CBW, i);
}

static int s1d_cbw(int NB) { return NB >= 2 ? 2 : 1; }
// bytes of the DMA part of a K3S1 image with K contraction channels and N output columns
long bts_lp_s1d_image_bytes_(int K, int N) {
  const int KS = (K + 15) / 16, NB = (N + 31) / 32, cbw = s1d_cbw(NB), ncg = (NB + cbw - 1) / cbw;
  return (long)ncg * KS * 3 * 9 * cbw * 1024;
}
int bts_lp_s1d_pack_(int dtype, const LpPackParams& p0, void* dst, hipStream_t stream) {
  LpPackParams p = p0;
  p.wp = reinterpret_cast<unsigned short*>(dst);
  const int cbw = s1d_cbw(p.NB), ncg = (p.NB + cbw - 1) / cbw;
  const long total = (long)ncg * p.KS * 3 * 9 * cbw * 512;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_s1d_pack_kernel<TF16>, dim3(blocks), dim3(256), 0, stream, p, cbw, ncg);
  else hipLaunchKernelGGL(lp_s1d_pack_kernel<TBF16>, dim3(blocks), dim3(256), 0, stream, p, cbw, ncg);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// =====================================================================================================================
// plan + launch
// =====================================================================================================================
struct S1dPlan {
  int mode, txl, ntx, nty, ntz, ncg, ksplit, ks_per;
  int bx, by, bz;
  long nitems;
};
// q = n / d for n < 2^31 as (n * mul) >> (32 + sh); mul == 0 means d == 1
static void s1d_fastdiv(unsigned d, unsigned& mul, unsigned& sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  int l = 0;
  while ((1u << l) < d) ++l;                                          // l = ceil(log2 d) >= 1
  const unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;     // ceil(2^(31+l) / d) < 2^32
  mul = (unsigned)m; sh = (unsigned)(l - 1);
}
// block of tiles one XCD's workgroups hold at a time: divisors of the tile grid (no partial blocks), at most `target` tiles, the
// largest shared-halo fraction (smallest unique volume per tile)
static void s1d_block(int ntx, int nty, int ntz, int tx, int ty, int tz, int target, int& bx, int& by, int& bz) {
  double best = 1e30;
  bx = by = bz = 1;
  for (int x = 1; x <= ntx; ++x) {
    if (ntx % x) continue;
    for (int y = 1; y <= nty; ++y) {
      if (nty % y) continue;
      for (int z = 1; z <= ntz; ++z) {
        if (ntz % z || x * y * z > target) continue;
        const double u = (double)(x * tx + 2) * (y * ty + 2) * (z * tz + 2) / ((double)x * y * z);   // unique halo voxels per tile
        if (u < best - 1e-9) { best = u; bx = x; by = y; bz = z; }
      }
    }
  }
}
static bool s1d_enabled() {   // BTS_LP_S1D=0: every stride-1 conv on the register-staged kernel (A/B; read per call so one process can do both)
  const char* e = getenv("BTS_LP_S1D");
  return !(e && atoi(e) == 0);
}
// tile grid + split-K of one tile shape (TX = 1 << txl); returns the modelled cost in k-step times: rounds of the 256 CUs x (k-steps
// per workgroup + 1.5 for an item's fill and output side)
static double s1d_plan_shape(int N, int D, int H, int W, int Cin, int Cout, int txl, S1dPlan& pl) {
  const int NB = (Cout + 31) / 32, KS = Cin / 16;
  pl.mode = NB >= 2 ? 1 : 0;
  pl.txl = txl;
  const int TX = 1 << pl.txl, ZP = 32 / TX, TY = pl.mode ? 4 : 8, TZ = 4 * ZP;
  pl.ntx = (W + TX - 1) / TX; pl.nty = (H + TY - 1) / TY; pl.ntz = (D + TZ - 1) / TZ;
  pl.ncg = pl.mode ? (NB + 1) / 2 : 1;
  pl.nitems = (long)N * pl.ntz * pl.nty * pl.ntx * pl.ncg;
  if (pl.nitems > 0x7fffffffL) return -1.0;
  s1d_block(pl.ntx, pl.nty, pl.ntz, TX, TY, TZ, pl.ncg >= 32 ? 1 : 32 / pl.ncg, pl.bx, pl.by, pl.bz);
  // split-K: grids that cannot give most CUs an item split the input channels (>= 2 k-steps per workgroup); the split with the
  // fewest (rounds x k-steps) wins, ties to the smaller split (less partial-sum traffic)
  // (+ a tenth of the fractional round count: of two shapes with the same whole rounds the one with fewer items runs 5 % faster)
  auto cost = [&](int per, int split) { return ((double)((pl.nitems * split + 255) / 256) + 0.1 * (double)(pl.nitems * split) / 256.0) * (per + 1.5); };
  pl.ksplit = 1; pl.ks_per = KS;
  double best = cost(KS, 1);
  { const char* e = getenv("BTS_LP_S1D_SPLIT_ITEMS"); if (pl.nitems < (e ? atol(e) : 160) && KS >= 4) {
    for (int ks = 2; ks <= KS / 2 && ks <= 16; ++ks) {
      const int per = (KS + ks - 1) / ks, split = (KS + per - 1) / per;
      const double c = cost(per, split) + 0.5 * split;      // (+ the reduce pass grows with the split)
      if (c < best - 1e-9) { best = c; pl.ks_per = per; pl.ksplit = split; }
    }
  } }
  return best;
}
static bool s1d_plan(int N, int D, int H, int W, int Cin, int Cout, S1dPlan& pl) {
  if (!s1d_enabled() || Cin % 16 != 0 || Cout % 8 != 0 || W < 12) return false;
  { const char* fl = getenv("BTS_LP_S1D_FLOOR"); if ((long)N * D * H * W < (fl ? atol(fl) : 4096)) return false; }
  // x extent of a tile: 32, or 16 with two z planes per fragment -- whichever wastes less of the 256 CUs on this grid (20x24x20, the
  // deepest level of the full inference volume: 32 wide gives 120 items = one round at split 2, 16 wide 144 items = two rounds;
  // 80x96x80: 16 wide tiles it exactly, 1200 items against 1440, 118 us against 136)
  const char* e = getenv("BTS_LP_S1D_TXL");
  if (e) return s1d_plan_shape(N, D, H, W, Cin, Cout, atoi(e) == 4 ? 4 : 5, pl) >= 0.0;
  S1dPlan a, b;
  const double ca = s1d_plan_shape(N, D, H, W, Cin, Cout, 5, a), cb = s1d_plan_shape(N, D, H, W, Cin, Cout, 4, b);
  if (ca < 0.0 && cb < 0.0) return false;
  const bool wide = cb < 0.0 || (ca >= 0.0 && ca <= cb);     // (ties -- every 128^3-derived grid -- go 32 wide, as measured in round 2)
  pl = wide ? a : b;
  return true;
}
long bts_lp_s1d_workspace_(int N, int D, int H, int W, int Cin, int Cout) {
  S1dPlan pl;
  if (!s1d_plan(N, D, H, W, Cin, Cout, pl)) return -1;
  return pl.ksplit > 1 ? (long)pl.ksplit * N * D * H * W * ((Cout + 31) / 32) * 32 * 4 : 0;
}
// GroupNorm-partial slots per (n, group) when the conv can emit them (no split-K, whole planes per group); 0 otherwise
long bts_lp_s1d_gn_B_(int N, int D, int H, int W, int Cin, int Cout, int Gn) {
  S1dPlan pl;
  if (Gn <= 0 || !s1d_plan(N, D, H, W, Cin, Cout, pl)) return 0;
  // (the split-K finish leaves the partials, in its own layout: slabs need not be whole planes there)
  if (pl.ksplit > 1) return bts_lp_splitk_gn_B_(N, (long)D * H * W, Cout, Gn);
  if (D % Gn != 0) return 0;
  return (long)(D / Gn) * pl.nty * pl.ntx * pl.ncg * 2;
}

// does the SC form with a split output take this shape?  (K contraction channels, Ncols output columns)
bool bts_lp_s1d_sc_split_ok_(int N, int D, int H, int W, int K, int Ncols) {
  S1dPlan pl;
  return s1d_plan(N, D, H, W, K, Ncols, pl) && pl.mode == 1 && pl.ksplit == 1 && Ncols % 32 == 0;
}

template <typename T, int MODE, int TXL, bool SC = false>
static int s1d_launch_t(const LpS1dParams& p, hipStream_t stream) {
  typedef S1dGeo<MODE, TXL> G;
  auto kern = lp_s1d_kernel<T, MODE, TXL, SC>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  long gx = p.nitems < 256 ? p.nitems : 256;
  if (p.ksplit > 1) { gx = (256 + p.ksplit - 1) / p.ksplit; if (gx > p.nitems) gx = p.nitems; if (gx < 1) gx = 1; }
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, p.ksplit), dim3(512), G::LDS_BYTES, stream, p);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// BTS_OK = ran, 1 = declined.  wp_dma = the DMA part of the K3S1 image.  gn_B (out, may be NULL): partial slots per (n, group) written.
// x2 / wp2 / ldx2 (may be NULL): the SC form -- y += x2 . w2 at the centre tap, x2 (N,D,H,W,Cin) with voxel stride ldx2, wp2 the first
// part of a K1 image with the same K = Cin and N = Cout (64-cout items only: 32-cout items have no registers left for the operand pair)
int bts_lp_s1d_launch_(int dtype, const void* x, const void* wp_dma, const float* bias, void* y, void* ws, long ws_bytes, int N, int D, int H,
                       int W, int Cin, int ldx, int Cout, int ldy, int accum, double* gnp, int gn_G, hipStream_t stream, const void* x2,
                       const void* wp2, int ldx2, long ysplit) {
  S1dPlan pl;
  if (!s1d_plan(N, D, H, W, Cin, Cout, pl)) return 1;
  // split output: whole 32-column blocks, no split-K (its finish writes one tensor), no fused statistics, 32-bit scalar offsets
  if (ysplit != 0 && (x2 == nullptr || ysplit < 0 || Cout % 32 != 0 || ldy < 32 || pl.ksplit > 1 || gnp != nullptr)) return 1;
  const bool sc = x2 != nullptr;
  if (sc && (pl.mode != 1 || wp2 == nullptr || ldx2 < Cin || ldx2 % 8 != 0 || (((uintptr_t)x2) & 15) || (((uintptr_t)wp2) & 15) ||
             (long)D * H * W * (long)ldx2 * 2 >= 0x7fffff00L))
    return 1;
  if (((long)(D + 2) * H * W + 64) * (long)ldx * 2 >= 0x7fffffffL) return 1;
  const long omax = (long)ldy > (long)((Cout + 31) / 32) * 32 * 2 ? ldy : (long)((Cout + 31) / 32) * 32 * 2;
  if ((long)D * H * W * omax * 2 >= 0x7fffff00L) return 1;
  LpS1dParams p;
  p.x = (const unsigned short*)x; p.wp = (const unsigned short*)wp_dma; p.bias = bias; p.y = (unsigned short*)y;
  p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldy; p.Cout = Cout; p.KS = Cin / 16; p.NB = (Cout + 31) / 32;
  p.ntx = pl.ntx; p.nty = pl.nty; p.ntz = pl.ntz; p.ncg = pl.ncg; p.nitems = pl.nitems;
  p.bx = pl.bx; p.by = pl.by; p.bz_ = pl.bz;
  p.bvol_ = (unsigned)(pl.bx * pl.by * pl.bz); p.nbx_ = (unsigned)(pl.ntx / pl.bx); p.nby_ = (unsigned)(pl.nty / pl.by); p.nbz_ = (unsigned)(pl.ntz / pl.bz);
  {
    const unsigned dv[7] = {(unsigned)pl.ncg, p.bvol_, (unsigned)pl.bx, (unsigned)pl.by, p.nbx_, p.nby_, p.nbz_};
    for (int k = 0; k < 7; ++k) s1d_fastdiv(dv[k], p.dv_mul[k], p.dv_sh[k]);
  }
  p.ksplit = pl.ksplit; p.ks_per = pl.ks_per; p.accum = accum;
  p.part = reinterpret_cast<float*>(ws);
  const long nvox = (long)N * D * H * W;
  if (p.ksplit > 1 && (ws == nullptr || ws_bytes < (long)p.ksplit * nvox * p.NB * 32 * 4 || (((uintptr_t)ws) & 15))) {
    if (gnp != nullptr) return BTS_ERR_WORKSPACE;      // (the caller sized the partial array for the split plan)
    p.ksplit = 1; p.ks_per = p.KS;
  }
  p.dbg = 0;
#ifdef BTS_TIMING_EXPERIMENTS
  { const char* e = getenv("BTS_S1D_DBG"); if (e) p.dbg = atoi(e); }
#endif
  p.x2 = (const unsigned short*)x2; p.wp2 = (const unsigned short*)wp2; p.ldx2 = ldx2;
  p.ysplit = ysplit; p.ycol = ysplit ? 0 : 32;
  p.gnp = gnp; p.gn_G = gn_G; p.gn_zt = gn_G > 0 ? D / gn_G : 1;
  p.gn_B = gn_G > 0 ? (long)p.gn_zt * pl.nty * pl.ntx * pl.ncg * 2 : 0;
  const bool split_gn = gnp != nullptr && p.ksplit > 1;      // statistics from the split-K finish (dense y, whole 32-cout blocks)
  if (split_gn) {
    if (accum || ldy != Cout || bts_lp_splitk_gn_B_(N, (long)D * H * W, Cout, gn_G) <= 0) return BTS_ERR_UNSUPPORTED;
    p.gnp = nullptr; p.gn_G = 0; p.gn_zt = 1; p.gn_B = 0;
  }
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(33 | ((1 + pl.mode * 2 + (pl.txl == 4 ? 1 : 0)) << 16), 2.0 * (sc ? 28.0 : 27.0) * Cin * (double)Cout * (double)nvox, stream);   // (bits 16+: the variant, for bench.py's per-variant table)
  int r;
  if (sc) {
    if (pl.txl == 5) r = dtype == LP_F16 ? s1d_launch_t<TF16, 1, 5, true>(p, stream) : s1d_launch_t<TBF16, 1, 5, true>(p, stream);
    else r = dtype == LP_F16 ? s1d_launch_t<TF16, 1, 4, true>(p, stream) : s1d_launch_t<TBF16, 1, 4, true>(p, stream);
  } else
#define S1D_CASE(M_, X_)                                                                                                  \
  if (pl.mode == M_ && pl.txl == X_)                                                                                      \
    r = dtype == LP_F16 ? s1d_launch_t<TF16, M_, X_>(p, stream) : s1d_launch_t<TBF16, M_, X_>(p, stream);
  S1D_CASE(0, 5) else S1D_CASE(1, 5) else S1D_CASE(0, 4) else S1D_CASE(1, 4) else r = BTS_ERR_UNSUPPORTED;
#undef S1D_CASE
  if (prof) bts_prof_end(stream);
  if (r != BTS_OK) return r;
  if (split_gn) return bts_lp_splitk_reduce_gn_(dtype, p.part, bias, y, gnp, N, (long)D * H * W, Cout, gn_G, p.ksplit, stream);
  if (p.ksplit > 1) return bts_lp_splitk_reduce_(dtype, p.part, bias, y, nvox, Cout, p.NB * 32, ldy, p.ksplit, accum, stream);
  return BTS_OK;
}
