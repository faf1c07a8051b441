// Reduced-precision STORAGE path of the forward pass (BASELINE configs[4]: full-volume inference in fp16, VAE off; also the
// forward half of configs[2], bf16): activations and packed weights are 16-bit (fp16 or bf16), every sum is fp32 -- conv
// contractions on v_mfma_f32_32x32x16_{f16,bf16}, GroupNorm statistics / SE gates / sigmoid in fp32 from the stored values.
// The fp32 engine (conv_igemm / conv_wino / groupnorm / se) is the parity reference of this path; the reference itself has
// no reduced-precision mode.  Call sites replaced: the same as the fp32 kernels' (layers/resnet.py:116-138,
// group_norm.py:83-124, downsample.py:28-45, upsample.py:28-43, decoder.py:55-63,65-83, encoder.py:69-101, model.py:58-68).
//
// Matrix instruction layout (32x32x16, K = 16 channels): A = weights (row = cout, lane>>5 selects k 0..7 / 8..15: one 16-byte
// load of 8 consecutive input channels), B = activations (column = voxel, same k split: one 16-byte read of 8 consecutive
// channels of the voxel -- NDHWC memory IS the operand layout), D: lane holds voxel lane&31 and couts 8*(r>>2) + 4*(lane>>5)
// + (r&3): four consecutive couts per register quad = one 8-byte store.
//
// Two conv kernels:
//   lp_conv_s1_kernel      3x3x3 stride-1 'same' (90 % of the FLOPs): halo tile of 16 channels staged global -> registers ->
//                          LDS (voxel stride 48 bytes: conflict-free 16-byte reads), wave = 32 x-columns x VB rows of one z
//                          plane x CB cout blocks, weights straight from L2 (one fragment feeds VB matrix instructions)
//   lp_conv_gather_kernel  1x1x1, stride-2 and transposed convs (gather form out[o*os+oo] = sum_t in[o*s+off_t] W[t], the
//                          transposed conv as 8 output-parity classes): operands straight from global memory -- every input
//                          voxel is needed by at most 8 outputs, an LDS tile would buy nothing
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.h"
#include "bts_internal.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

#include "lowp_common.h"
#include "finalize_parts.h"

// lowp_s1d.hip: LDS-DMA staged stride-1 3x3x3 kernel (offered first; 1 = declined) and its part of the K3S1 image
long bts_lp_s1d_image_bytes_(int K, int N);
int bts_lp_s1d_pack_(int dtype, const LpPackParams& p, void* dst, hipStream_t stream);
long bts_lp_s1d_workspace_(int N, int D, int H, int W, int Cin, int Cout);
long bts_lp_s1d_gn_B_(int N, int D, int H, int W, int Cin, int Cout, int Gn);
// lowp_s1z.hip: z-marching stride-1 3x3x3 kernel for few channels (offered first; 1 = declined)
long bts_lp_s1z_gn_B_(int N, int D, int H, int W, int Cin, int Cout, int Gn);
int bts_lp_s1z_launch_(int dtype, const void* x, const void* wp, const float* bias, void* y, int N, int D, int H, int W, int Cin, int ldx,
                       int Cout, int ldy, int accum, double* gn_part, int gn_G, hipStream_t stream, const LpGnbFuse* gb = nullptr,
                       const LpGnaFuse* ga = nullptr, const void* x2 = nullptr, const void* wp2 = nullptr, int ldx2 = 0, void* y2 = nullptr,
                       const float* bias2 = nullptr, double* gap_part = nullptr, int ldy2 = 0, int Cout2 = 0, int accum2 = 0,
                       const void* xb = nullptr, int ldxb = 0);
long bts_lp_s1z_fs_B_(int N, int D, int H, int W, int Cin, int ldx, int Cout, int Cout2);
bool bts_lp_s1z_gna_ok_(int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, int in_G);
long bts_lp_s1z_gnb_B_(int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, int Gn);
// lowp_k1.hip: streaming 1x1x1 kernel (offered first; 1 = declined); gap partials per block of bts_lp_k1_gap_block_ positions
int bts_lp_k1_gap_block_(long npos, long V, int Cin, int Cout);
int bts_lp_s2t_launch_(int dtype, const void* x, const void* wp, const float* bias, void* y, int N, int D, int H, int W, int Cin, int ldx,
                       int Cout, int ldy, int accum, hipStream_t stream);
int bts_lp_k1_launch_(int dtype, const void* x, const void* wp, const float* bias, void* y, long npos, int Cin, int ldx, int Cout, int ldy,
                      int accum, double* gap_part, int gap_block, hipStream_t stream);
// lowp_up.hip: transposed form, all eight output classes in one pass (offered first; 1 = declined)
int bts_lp_up_launch_(int dtype, const void* x, const void* wp_dma, const float* bias, void* y, int N, int D, int H, int W, int Cin, int ldx,
                      int Cout, int ldy, int accum, hipStream_t stream, double* gnp = nullptr, int gn_G = 0);
long bts_lp_up_gn_B_(int N, int D, int H, int W, int Cin, int Cout, int Gn);
int bts_lp_s1d_launch_(int dtype, const void* x, const void* wp_dma, const float* bias, void* y, void* ws, long ws_bytes, int N, int D, int H,
                       int W, int Cin, int ldx, int Cout, int ldy, int accum, double* gnp, int gn_G, hipStream_t stream,
                       const void* x2 = nullptr, const void* wp2 = nullptr, int ldx2 = 0, long ysplit = 0);
bool bts_lp_s1d_sc_split_ok_(int N, int D, int H, int W, int K, int Ncols);

// =====================================================================================================================
// weight packing: Keras layout fp32 -> [tap][k-step of 16 cin][cout block of 32][h][32 couts][8 cin] 16-bit
// =====================================================================================================================
template <typename T>
__device__ __forceinline__ void lp_pack_elem(const LpPackParams& p, long i) {
  const int e = (int)(i & 7), r = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
  long q = i >> 9;
  const int cb = (int)(q % p.NB); q /= p.NB;
  const int ks = (int)(q % p.KS);
  const int t = (int)(q / p.KS);
  p.wp[i] = T::st(lp_pack_src(p, t, ks * 16 + h * 8 + e, cb * 32 + r));
}
template <typename T>
__global__ __launch_bounds__(256) void lp_pack_kernel(const LpPackParams p) {
  const long total = (long)p.ntaps * p.KS * p.NB * 512;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) lp_pack_elem<T>(p, i);
}

static int lp_ntaps(int kind) { return kind == BTS_CONV_K1 ? 1 : 27; }

static bool lp_has_dma_part(int kind, int role) {
  return kind == BTS_CONV_K3S1 || (kind == BTS_CONV_K3S2T && role == BTS_ROLE_FWD) || (kind == BTS_CONV_K3S2 && role == BTS_ROLE_BWD_DATA);
}
extern "C" long bts_lp_packed_bytes(int kind, int role, int Cin_slab, int Cout) {
  if (kind < 0 || kind > 3 || role < 0 || role > 1 || Cin_slab <= 0 || Cout <= 0) return -1;
  const int K = role == BTS_ROLE_FWD ? Cin_slab : Cout, N = role == BTS_ROLE_FWD ? Cout : Cin_slab;
  const long first = (long)lp_ntaps(kind) * ((K + 15) / 16) * ((N + 31) / 32) * 512 * 2;
  // stride-1 3x3x3 images -- and the two images the transposed form runs on -- carry a second part, the same weights in the stage
  // order of the LDS-DMA kernels (lowp_s1d.hip, lowp_up.hip)
  return lp_has_dma_part(kind, role) ? first + bts_lp_s1d_image_bytes_(K, N) : first;
}
// byte offset of the DMA part inside a K3S1 image with K contraction channels and N output columns
static long lp_s1d_part_offset(int K, int N) { return 27L * ((K + 15) / 16) * ((N + 31) / 32) * 1024; }
static int lp_pack_params(LpPackParams& p, int kind, int role, const float* w, void* wp, int Cin_ref, int Cout, int Cin_slab, int dup_start,
                          int dup_shift) {
  if (kind < 0 || kind > 3 || role < 0 || role > 1) return BTS_ERR_UNSUPPORTED;
  if (Cin_slab + dup_shift != Cin_ref || dup_shift < 0 || dup_start < 0 || dup_start + dup_shift > Cin_slab) return BTS_ERR_SHAPE;
  p.w = w; p.wp = reinterpret_cast<unsigned short*>(wp);
  p.ntaps = lp_ntaps(kind);
  p.shift = dup_shift;
  p.dup_start = dup_shift > 0 ? dup_start : (1 << 30);
  long sCin, sCout;
  if (kind == BTS_CONV_K3S2T) { sCout = Cin_ref; sCin = 1; }   // (t, Cout, Cin)
  else { sCin = Cout; sCout = 1; }                             // (t, Cin, Cout)
  p.sT = (long)Cin_ref * Cout;
  if (role == BTS_ROLE_FWD) {
    p.K = Cin_slab; p.N = Cout; p.sK = sCin; p.sN = sCout; p.flip = 0; p.cin_is_k = 1;
  } else {
    p.K = Cout; p.N = Cin_slab; p.sK = sCout; p.sN = sCin; p.cin_is_k = 0;
    p.flip = (kind == BTS_CONV_K3S1) ? 1 : 0;
  }
  p.KS = (p.K + 15) / 16; p.NB = (p.N + 31) / 32;
  return BTS_OK;
}
extern "C" int bts_lp_pack(int kind, int role, int dtype, const float* w, void* wp, int Cin_ref, int Cout, int Cin_slab, int dup_start,
                           int dup_shift, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  LpPackParams p;
  const int r = lp_pack_params(p, kind, role, w, wp, Cin_ref, Cout, Cin_slab, dup_start, dup_shift);
  if (r != BTS_OK) return r;
  const long total = (long)p.ntaps * p.KS * p.NB * 512;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_pack_kernel<TF16>, dim3(blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(lp_pack_kernel<TBF16>, dim3(blocks), dim3(256), 0, stream, p);
  BTS_LAUNCH_CHECK();
  if (lp_has_dma_part(kind, role)) return bts_lp_s1d_pack_(dtype, p, reinterpret_cast<char*>(wp) + lp_s1d_part_offset(p.K, p.N), stream);
  return BTS_OK;
}
// All 16-bit weight images in one launch (every image goes stale together at the optimiser step, train.py:152; the batch-8 step spent
// 1.2 ms in 194 launches of 6 us).  Host table of descriptors as bts_conv_pack_desc / bts_conv_pack_batch of the fp32 engine.
struct LpPackDesc {
  LpPackParams p;          // first part
  unsigned short* wp_dma;  // second part (LDS-DMA stage order), or NULL
  int cbw, ncg;
  long total_main, total;  // elements of the first part / of both
  long first_block;
};
#define LP_PACK_BLOCK_ELEMS 2048
template <typename T>
__global__ __launch_bounds__(256) void lp_pack_batch_kernel(const LpPackDesc* tab, int n) {
  const long blk = blockIdx.x;
  int lo = 0, hi = n - 1;       // last descriptor with first_block <= blk
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  const LpPackDesc d = tab[lo];
  const long e0 = (blk - d.first_block) * LP_PACK_BLOCK_ELEMS;
  for (int k = 0; k < LP_PACK_BLOCK_ELEMS / 256; ++k) {
    const long i = e0 + k * 256 + threadIdx.x;
    if (i < d.total_main) lp_pack_elem<T>(d.p, i);
    else if (i < d.total) {
      LpPackParams q = d.p;
      q.wp = d.wp_dma;
      lp_s1d_pack_elem<T>(q, d.cbw, i - d.total_main);
    }
  }
}
extern "C" long bts_lp_pack_desc_bytes(void) { return (long)sizeof(LpPackDesc); }
extern "C" long bts_lp_pack_desc(void* host_table, int index, long first_block, int kind, int role, const float* w, void* wp, int Cin_ref,
                                 int Cout, int Cin_slab, int dup_start, int dup_shift) {
  LpPackDesc d;
  const int r = lp_pack_params(d.p, kind, role, w, wp, Cin_ref, Cout, Cin_slab, dup_start, dup_shift);
  if (r != BTS_OK) return r;
  d.total_main = (long)d.p.ntaps * d.p.KS * d.p.NB * 512;
  d.total = d.total_main;
  d.wp_dma = nullptr; d.cbw = 1; d.ncg = 1;
  if (lp_has_dma_part(kind, role)) {
    d.wp_dma = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(wp) + lp_s1d_part_offset(d.p.K, d.p.N));
    d.cbw = d.p.NB >= 2 ? 2 : 1;
    d.ncg = (d.p.NB + d.cbw - 1) / d.cbw;
    d.total += bts_lp_s1d_image_bytes_(d.p.K, d.p.N) / 2;
  }
  d.first_block = first_block;
  reinterpret_cast<LpPackDesc*>(host_table)[index] = d;
  return (d.total + LP_PACK_BLOCK_ELEMS - 1) / LP_PACK_BLOCK_ELEMS;
}
extern "C" int bts_lp_pack_batch(int dtype, const void* table_dev, int n, long total_blocks, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffL) return BTS_ERR_SHAPE;
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_pack_batch_kernel<TF16>, dim3((unsigned)total_blocks), dim3(256), 0, stream, reinterpret_cast<const LpPackDesc*>(table_dev), n);
  else hipLaunchKernelGGL(lp_pack_batch_kernel<TBF16>, dim3((unsigned)total_blocks), dim3(256), 0, stream, reinterpret_cast<const LpPackDesc*>(table_dev), n);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// one register quad (4 consecutive couts) of a result: bias added by the caller; optional read-modify-write accumulation
template <typename T>
__device__ __forceinline__ void lp_store_quad(unsigned short* dst, float o0, float o1, float o2, float o3, int nleft, int accum) {
  if (nleft >= 4) {
    if (accum) {
      const u32x2 old = *reinterpret_cast<const u32x2*>(dst);
      o0 += T::ld((unsigned short)(old[0] & 0xffffu)); o1 += T::ld((unsigned short)(old[0] >> 16));
      o2 += T::ld((unsigned short)(old[1] & 0xffffu)); o3 += T::ld((unsigned short)(old[1] >> 16));
    }
    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<T>(o0, o1), pack2<T>(o2, o3)};
  } else {
    const float o[3] = {o0, o1, o2};
    for (int j = 0; j < nleft; ++j) dst[j] = T::st(accum ? o[j] + T::ld(dst[j]) : o[j]);
  }
}

// =====================================================================================================================
// 3x3x3 stride-1 convolution
// =====================================================================================================================
struct LpS1Params {
  const unsigned short* x;
  const unsigned short* wp;
  const float* bias;
  unsigned short* y;
  int N, D, H, W, ldx, ldy, Cout, KS, NB;   // KS = k-steps of 16 input channels, NB = cout blocks of 32
  int ntx, nty, ntz, ncg;                    // tiles per axis, cout groups of CB blocks
  int ksplit, ks_per;                        // split-K over blockIdx.y (small grids)
  long ntiles;
  int accum;                                 // y += result (gradient accumulation into a slab), else y = result
  float* part;
  double* gnp;                               // fused GroupNorm partial sums of y (slab semantics), [N*G][B][2], B = gn_zt*nty*ntx*ncg; or NULL
  int gn_G, gn_zt;                           // groups; z planes per slab group (D / G)
};
#define LPS 24   // halves per staged voxel: 16 channels + 8 pad (48-byte stride: conflict-free 16-byte reads)

__device__ __forceinline__ u32x2 bload8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
}
__device__ __forceinline__ void bstore8(__amdgpu_buffer_rsrc_t r, unsigned voff, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 0);
}

template <typename T, int VB, int CB, int TXL>
__global__ __launch_bounds__(256, 2) void lp_conv_s1_kernel(const LpS1Params p) {
  constexpr int TX = 1 << TXL, R = 32 / TX, TY = VB * R, TZ = 4;
  constexpr int SX = TX + 2, SY = TY + 2, SZ = TZ + 2;
  constexpr int NVOX = SX * SY * SZ;
  constexpr int NSLOT = (NVOX * 2 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  float* const bsh = reinterpret_cast<float*>(lds + NVOX * LPS);     // the tile's 32 * CB bias values (behind the halo tile)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  const int lx = l32 & (TX - 1), ly = l32 >> TXL;
  // Persistent over tiles (blockIdx.x, + gridDim.x, ...), and a tile boundary is a k-step boundary: during a tile's LAST k-step
  // the first halo slab and weight groups of the workgroup's next tile are requested, so the output side below runs with its
  // successor's operands in flight (with 32 input channels a tile has only two k-steps: a fresh global round trip per tile was
  // most of its time).  Everything on the vector-memory queue is issued unconditionally -- masked lanes get an offset outside the
  // buffer descriptor -- so that the compiler's vmcnt waits are exact and the wait for a halo slab never drains the stores issued
  // after it.
  struct Tile { int cg, n, ox0, oy0, oz0; };
  auto coords = [&](long tile) {
    Tile t;
    long b = tile;
    t.cg = (int)(b % p.ncg); b /= p.ncg;
    t.ox0 = (int)(b % p.ntx) * TX; b /= p.ntx;
    t.oy0 = (int)(b % p.nty) * TY; b /= p.nty;
    t.oz0 = (int)(b % p.ntz) * TZ;
    t.n = (int)(b / p.ntz);
    return t;
  };
  __amdgpu_buffer_rsrc_t xr;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
  unsigned goff[NSLOT];
  // halo origin + per-slot offsets of a tile; slots outside the image get an offset outside the descriptor -> zeros ('same' padding)
  auto setup = [&](const Tile& t) {
    const unsigned short* xorg = p.x + ((((long)t.n * p.D + (t.oz0 - 1)) * p.H + (t.oy0 - 1)) * p.W + (t.ox0 - 1)) * (long)p.ldx;
    xr = __builtin_amdgcn_make_buffer_rsrc((void*)xorg, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = tid + i * 256;
      goff[i] = 0x80000000u;
      if (e < NVOX * 2) {
        const int vox = e >> 1, q = e & 1;
        const int vz = vox / (SY * SX);
        const int r = vox - vz * (SY * SX);
        const int vy = r / SX, vx = r - vy * SX;
        if ((unsigned)(t.oz0 - 1 + vz) < (unsigned)p.D && (unsigned)(t.oy0 - 1 + vy) < (unsigned)p.H && (unsigned)(t.ox0 - 1 + vx) < (unsigned)p.W)
          goff[i] = (unsigned)(((vz * p.H + vy) * p.W + vx) * p.ldx + q * 8) * 2u;
      }
    }
  };
  u32x4 pre[NSLOT];
  auto fetch = [&](unsigned soff) {     // soff = k-step * 32 bytes, or 0x80000000: nothing to fetch (zeros, no traffic)
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) pre[i] = bload16(xr, goff[i], soff);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = tid + i * 256;
      if (e < NVOX * 2) *reinterpret_cast<u32x4*>(lds + (e >> 1) * LPS + (e & 1) * 8) = pre[i];
    }
  };
  auto bias_of = [&](const Tile& t) {   // thread i < 32*CB holds bias[cout block base + i]; out-of-range -> 0 (one request on every path)
    const int co = t.cg * CB * 32 + tid;
    const unsigned off = (tid < 32 * CB && co < p.Cout) ? (unsigned)co * 4u : 0x80000000u;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, off, 0, 0));
  };
  // this lane's B-operand base inside the tile: voxel (z = wave, y = ly, x = lx), channel half h
  const int bbase = ((wave * SY + ly) * SX + lx) * LPS + h * 8;
  const unsigned wlane = (unsigned)((h * 32 + l32) * 16);
  // k-steps of this workgroup (split-K: blockIdx.y takes k-steps [ks0, ks1))
  int ks0 = 0, ks1 = p.KS;
  if (p.ksplit > 1) {
    ks0 = blockIdx.y * p.ks_per;
    ks1 = ks0 + p.ks_per;
    if (ks1 > p.KS) ks1 = p.KS;
  }
  auto wfrag = [&](int t, int ks, int c, int cg) {
    const int cb = cg * CB + c;
    return bload16(wr, wlane, (unsigned)((((t * p.KS + ks) * p.NB) + (cb < p.NB ? cb : 0)) * 1024));
  };
  constexpr bool ROWS = (R == 1);   // 32-wide tiles: input rows shared by the three dy taps
  // weights: the tap consumed u-th in a k-step (ROWS: (dz, dx) groups of three dy, else plain tap order) lives in a[u % 9] and is
  // requested WT taps ahead (27 % 9 == 0: the slot of a tap is the same in every k-step)
  constexpr int WT = ROWS ? ((CB == 1) ? 6 : 3) : 2;
  u32x4 a[9][CB];
  auto tap_of = [](int u) { return ROWS ? (((u / 9) * 3 + (u % 3)) * 3 + (u / 3) % 3) : u; };   // u -> (dz, dy, dx) tap index
  auto wtap = [&](int u, int ks, int cg) {
#pragma unroll
    for (int c = 0; c < CB; ++c) a[u % 9][c] = wfrag(tap_of(u), ks, c, cg);
  };

  f32x16 acc[VB][CB];
  long tile = blockIdx.x;
  if (tile >= p.ntiles) return;
  Tile cur = coords(tile);
  setup(cur);
  float bias_v = bias_of(cur);
  fetch((unsigned)ks0 * 32u);
#pragma unroll
  for (int u = 0; u < WT; ++u) wtap(u, ks0, cur.cg);
  for (; tile < p.ntiles; tile += gridDim.x) {
    const bool have_next = tile + gridDim.x < p.ntiles;
    const Tile nxt = have_next ? coords(tile + gridDim.x) : cur;
#pragma unroll
    for (int v = 0; v < VB; ++v)
#pragma unroll
      for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[v][c][r] = 0.f;
    for (int ks = ks0; ks < ks1; ++ks) {
      __syncthreads();        // every wave is done reading the previous k-step's tile (and the previous tile's bias)
      commit();
      if (ks == ks0 && tid < 32 * CB) bsh[tid] = bias_v;
      __syncthreads();
      const bool last = ks + 1 == ks1;
      if (last && have_next) {   // from here on the input-side state is the next tile's
        setup(nxt);
        bias_v = bias_of(nxt);
      } else {
        bias_v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, 0x80000000u, 0, 0));   // (same request count)
      }
      fetch(last ? (have_next ? (unsigned)ks0 * 32u : 0x80000000u) : (unsigned)(ks + 1) * 32u);
      const int ksn = last ? ks0 : ks + 1;
      const int cgn = last ? nxt.cg : cur.cg;
      if constexpr (ROWS) {
        // the input rows of group g+1 are read while group g computes -- where the registers allow it (4 x 2 tiles: 128 accumulators
        // + weight ring + halo prefetch leave room for one row set only)
        constexpr int NB_ = (VB * CB >= 8) ? 1 : 2;
        u32x4 bj[NB_][VB + 2];
        auto rows = [&](u32x4 (&dst)[VB + 2], int g) {
          const int dz = g / 3, dx = g % 3;
#pragma unroll
          for (int j = 0; j < VB + 2; ++j) dst[j] = *reinterpret_cast<const u32x4*>(lds + bbase + ((dz * SY + j) * SX + dx) * LPS);
        };
        if (NB_ == 2) rows(bj[0], 0);
#pragma unroll
        for (int u = 0; u < 27; ++u) {
          const int g = u / 3, dy = u % 3;
          if (u + WT < 27) wtap(u + WT, ks, cur.cg); else wtap(u + WT - 27, ksn, cgn);
          if (dy == 0) { if (NB_ == 2) { if (g + 1 < 9) rows(bj[(g + 1) % NB_], g + 1); } else rows(bj[0], g); }
#pragma unroll
          for (int v = 0; v < VB; ++v)
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[v][c] = T::mfma(a[u % 9][c], bj[g % NB_][v + dy], acc[v][c]);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
          const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
          if (t + WT < 27) wtap(t + WT, ks, cur.cg); else wtap(t + WT - 27, ksn, cgn);
#pragma unroll
          for (int v = 0; v < VB; ++v) {
            const u32x4 bv = *reinterpret_cast<const u32x4*>(lds + bbase + ((dz * SY + (v * R + dy)) * SX + dx) * LPS);
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[v][c] = T::mfma(a[t % 9][c], bv, acc[v][c]);
          }
        }
      }
    }
    // ---- output side of `cur` ----
    const int oz = cur.oz0 + wave, ox = cur.ox0 + lx;
    const bool gn_on = p.gnp != nullptr;
    float gn_s = 0.f, gn_q = 0.f;
    if (p.ksplit > 1) {   // raw fp32 partial sums [split][voxel][NB*32]; bias / rounding happen in the reduce kernel
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int cb = cur.cg * CB + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int v = 0; v < VB; ++v) {
            const int oy = cur.oy0 + v * R + ly;
            if (cb < p.NB && oz < p.D && oy < p.H && ox < p.W) {
              float* dst = p.part + ((((long)blockIdx.y * p.N + cur.n) * p.D + oz) * p.H + oy) * (long)p.W * (p.NB * 32) +
                           (long)ox * (p.NB * 32) + cb * 32 + 8 * q + 4 * h;
              *reinterpret_cast<f32x4*>(dst) = f32x4{acc[v][c][4 * q], acc[v][c][4 * q + 1], acc[v][c][4 * q + 2], acc[v][c][4 * q + 3]};
            }
          }
        }
      }
    } else if (p.Cout % 4 == 0) {
      // bias from LDS, convert, one 8-byte buffer store per register quad (masked positions out of range)
      const __amdgpu_buffer_rsrc_t yr =
          __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long)cur.n * p.D * p.H * p.W * (long)p.ldy), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int cb = cur.cg * CB + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cb * 32 + 8 * q + 4 * h;
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bsh + c * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int v = 0; v < VB; ++v) {
            const int oy = cur.oy0 + v * R + ly;
            const bool ok = cb < p.NB && co < p.Cout && oz < p.D && oy < p.H && ox < p.W;
            const unsigned off = ok ? (unsigned)((((oz * p.H + oy) * p.W + ox) * p.ldy + co) * 2) : 0x80000000u;
            float o0 = acc[v][c][4 * q] + bq[0], o1 = acc[v][c][4 * q + 1] + bq[1], o2 = acc[v][c][4 * q + 2] + bq[2],
                  o3 = acc[v][c][4 * q + 3] + bq[3];
            if (p.accum) {
              const u32x2 old = bload8(yr, off);
              o0 += T::ld((unsigned short)(old[0] & 0xffffu)); o1 += T::ld((unsigned short)(old[0] >> 16));
              o2 += T::ld((unsigned short)(old[1] & 0xffffu)); o3 += T::ld((unsigned short)(old[1] >> 16));
            }
            bstore8(yr, off, u32x2{pack2<T>(o0, o1), pack2<T>(o2, o3)});
            if (gn_on && ok) {   // (gn_on is launch-uniform) the unrounded outputs: what the stored values estimate
              gn_s += (o0 + o1) + (o2 + o3);
              gn_q = fmaf(o0, o0, fmaf(o1, o1, fmaf(o2, o2, fmaf(o3, o3, gn_q))));
            }
          }
        }
      }
      if (gn_on) {   // one fp64 (sum, sumsq) pair per (z plane = wave, tile column, cout group): lanes by shuffle tree, fixed order
        const double ds = wave_sum_f64((double)gn_s), dq = wave_sum_f64((double)gn_q);
        if (lane == 0 && oz < p.D) {
          const int ty = cur.oy0 / TY, tx = cur.ox0 / TX;
          const int gg = oz / p.gn_zt;      // gn_zt = planes per z-slab group
          const long B = (long)p.gn_zt * p.nty * p.ntx * p.ncg;
          const long slot = (((long)(oz - gg * p.gn_zt) * p.nty + ty) * p.ntx + tx) * p.ncg + cur.cg;
          double* dst = p.gnp + (((long)cur.n * p.gn_G + gg) * B + slot) * 2;
          dst[0] = ds;
          dst[1] = dq;
        }
      }
    } else {
      // heads with fewer than four output channels per quad: element-wise stores
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int cb = cur.cg * CB + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cb * 32 + 8 * q + 4 * h;
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bsh + c * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int v = 0; v < VB; ++v) {
            const int oy = cur.oy0 + v * R + ly;
            if (cb < p.NB && oz < p.D && oy < p.H && ox < p.W && co < p.Cout) {
              unsigned short* dst = p.y + ((((long)cur.n * p.D + oz) * p.H + oy) * p.W + ox) * (long)p.ldy + co;
              lp_store_quad<T>(dst, acc[v][c][4 * q] + bq[0], acc[v][c][4 * q + 1] + bq[1], acc[v][c][4 * q + 2] + bq[2],
                               acc[v][c][4 * q + 3] + bq[3], p.Cout - co, p.accum);
            }
          }
        }
      }
    }
    cur = nxt;
  }
}

// finish of a split-K launch: y = round(bias + sum_z part[z]) in fixed order
template <typename T>
__global__ __launch_bounds__(256) void lp_splitk_reduce_kernel(const float* part, const float* bias, unsigned short* y, long nvox, int Cout,
                                                               int Npad, int ldy, int ksplit, int accum) {
  const int q4 = Npad / 4;
  const long total = nvox * q4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long v = i / q4;
    const int c = (int)(i - v * q4) * 4;
    if (c >= Cout) continue;
    f32x4 s = *reinterpret_cast<const f32x4*>(part + v * Npad + c);
    for (int z = 1; z < ksplit; ++z) s += *reinterpret_cast<const f32x4*>(part + ((long)z * nvox + v) * Npad + c);
    if (bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (c + j < Cout) s[j] += bias[c + j];
    }
    lp_store_quad<T>(y + v * ldy + c, s[0], s[1], s[2], s[3], Cout - c, accum);
  }
}

// The same finish for a conv whose output goes into a slab-mode GroupNorm (dense y, Cout % 32 == 0): the workgroups walk the (sample, slab)
// units the way lp_gn_stats_kernel does and leave (sum, sumsq) partials of the ROUNDED values, like that kernel -- the statistics pass over
// the stored tensor and its launch go away (the deepest level of the inference volume runs eight such layers).
template <typename T>
__global__ __launch_bounds__(256) void lp_splitk_reduce_gn_kernel(const float* part, const float* bias, unsigned short* y, double* gn_partial,
                                                                  long nvox, long E, long L, int C, int G, int ksplit, int B) {
  __shared__ double sh[8];
  const int unit = blockIdx.y, n = unit / G, g = unit % G;
  const long lo = (long)n * E + (long)g * L;
  const long per = ((L / 8 + B - 1) / B) * 8;
  const long a = lo + (long)blockIdx.x * per, bnd = (a + per < lo + L) ? a + per : lo + L;
  const long tot = nvox * C;
  double s = 0.0, q = 0.0;
  float fs = 0.f, fq = 0.f;
  long cnt = 0;
  for (long i = a + threadIdx.x * 8L; i < bnd; i += 256 * 8) {
    const int c = (int)(i % C);
    f32x4 u0 = *reinterpret_cast<const f32x4*>(part + i), u1 = *reinterpret_cast<const f32x4*>(part + i + 4);
    for (int z = 1; z < ksplit; ++z) {
      u0 += *reinterpret_cast<const f32x4*>(part + (long)z * tot + i);
      u1 += *reinterpret_cast<const f32x4*>(part + (long)z * tot + i + 4);
    }
    float o[8], v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = u0[e]; o[4 + e] = u1[e]; }
    if (bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += bias[c + e];
    }
    const u32x4 r = pack8<T>(o);
    *reinterpret_cast<u32x4*>(y + i) = r;
    unpack8<T>(r, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { fs += v[e]; fq = fmaf(v[e], v[e], fq); }
    if (++cnt == 64) { s += fs; q += fq; fs = fq = 0.f; cnt = 0; }
  }
  s += fs; q += fq;
  s = block_sum_f64(s, sh);
  q = block_sum_f64(q, sh + 4);
  if (threadIdx.x == 0) {
    double* o = gn_partial + ((long)unit * B + blockIdx.x) * 2;
    o[0] = s; o[1] = q;
  }
}
// partial slots per (n, group) the kernel above leaves (0 = it does not take the shape)
long bts_lp_splitk_gn_B_(int N, long V, int Cout, int G) {
  if (G <= 0 || Cout % 32 != 0 || Cout % G != 0 || (V * Cout) % G != 0 || ((V * Cout) / G) % 8 != 0) return 0;
  static const bool off = [] { const char* e = getenv("BTS_LP_SPLIT_GN"); return e && atoi(e) == 0; }();      // A/B: the separate statistics pass
  if (off) return 0;
  // (many short workgroups -- one or two 2048-element steps each: the tensor is small and the pass is latency-bound; < 512 partials per
  // unit keeps the finalize on its one-wave-per-unit form)
  long b = (V * Cout / G + 2047) / 2048;
  if (b > 448) b = 448;
  return b;
}
int bts_lp_splitk_reduce_gn_(int dtype, const float* part, const float* bias, void* y, double* gn_partial, int N, long V, int Cout, int G,
                             int ksplit, hipStream_t stream) {
  const long B = bts_lp_splitk_gn_B_(N, V, Cout, G);
  if (B <= 0 || (((uintptr_t)y) & 15)) return BTS_ERR_UNSUPPORTED;
  const long E = V * Cout, L = E / G;
  (void)hipGetLastError();
  if (dtype == LP_F16)
    hipLaunchKernelGGL(lp_splitk_reduce_gn_kernel<TF16>, dim3((unsigned)B, (unsigned)(N * G)), dim3(256), 0, stream, part, bias, (unsigned short*)y,
                       gn_partial, (long)N * V, E, L, Cout, G, ksplit, (int)B);
  else
    hipLaunchKernelGGL(lp_splitk_reduce_gn_kernel<TBF16>, dim3((unsigned)B, (unsigned)(N * G)), dim3(256), 0, stream, part, bias, (unsigned short*)y,
                       gn_partial, (long)N * V, E, L, Cout, G, ksplit, (int)B);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

int bts_lp_splitk_reduce_(int dtype, const float* part, const float* bias, void* y, long nvox, int Cout, int Npad, int ldy, int ksplit,
                          int accum, hipStream_t stream) {
  long blocks = (nvox * (Npad / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  (void)hipGetLastError();
  if (dtype == LP_F16)
    hipLaunchKernelGGL(lp_splitk_reduce_kernel<TF16>, dim3((unsigned)blocks), dim3(256), 0, stream, part, bias, (unsigned short*)y, nvox, Cout, Npad, ldy,
                       ksplit, accum);
  else
    hipLaunchKernelGGL(lp_splitk_reduce_kernel<TBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, part, bias, (unsigned short*)y, nvox, Cout, Npad, ldy,
                       ksplit, accum);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// split-K plan of the stride-1 kernel: grids that cannot give every CU a workgroup split the input channels
static int lp_s1_ksplit(long wgs, int KS) {
  if (wgs >= 192 || KS < 8) return 1;
  int ks = (int)((384 + wgs - 1) / wgs);
  if (ks > KS / 4) ks = KS / 4;    // at least 4 k-steps per workgroup
  if (ks > 16) ks = 16;
  return ks < 1 ? 1 : ks;
}

template <typename T, int VB, int CB, int TXL>
static int lp_s1_launch(LpS1Params p, void* ws, long ws_bytes, hipStream_t stream) {
  constexpr int TX = 1 << TXL, R = 32 / TX, TY = VB * R, TZ = 4;
  p.ntx = (p.W + TX - 1) / TX; p.nty = (p.H + TY - 1) / TY; p.ntz = (p.D + TZ - 1) / TZ;
  p.ncg = (p.NB + CB - 1) / CB;
  const long wgs = (long)p.N * p.ntz * p.nty * p.ntx * p.ncg;
  if (wgs > 0x7fffffffL) return BTS_ERR_SHAPE;
  p.ksplit = lp_s1_ksplit(wgs, p.KS);
  p.ks_per = p.KS;
  p.part = reinterpret_cast<float*>(ws);
  const long nvox = (long)p.N * p.D * p.H * p.W;
  if (p.ksplit > 1) {
    p.ks_per = (p.KS + p.ksplit - 1) / p.ksplit;
    p.ksplit = (p.KS + p.ks_per - 1) / p.ks_per;
    if (ws == nullptr || ws_bytes < (long)p.ksplit * nvox * p.NB * 32 * 4 || (((uintptr_t)ws) & 15)) { p.ksplit = 1; p.ks_per = p.KS; }
  }
  if (p.gnp != nullptr && p.ksplit > 1) return BTS_ERR_UNSUPPORTED;   // (bts_lp_conv3d_fwd_gn plans with the same rule: not reached)
  const size_t shmem = (size_t)(TX + 2) * (TY + 2) * (TZ + 2) * LPS * 2 + 32 * CB * 4;   // halo tile + the tile's bias values
  if ((long)p.D * p.H * p.W * (long)p.ldy * 2 >= 0x7fffff00L) return BTS_ERR_SHAPE;   // 31-bit output offsets inside one sample
  auto kern = lp_conv_s1_kernel<T, VB, CB, TXL>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  (void)hipGetLastError();
  p.ntiles = wgs;
  const long grid = wgs < 1024 ? wgs : 1024;   // <= 4 workgroups per CU in flight or queued
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(30, 2.0 * 27.0 * 16.0 * p.KS * p.Cout * (double)nvox, stream);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, p.ksplit), dim3(256), shmem, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  if (p.ksplit > 1) {
    long blocks = (nvox * (p.NB * 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lp_splitk_reduce_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, p.part, p.bias, p.y, nvox, p.Cout, p.NB * 32,
                       p.ldy, p.ksplit, p.accum);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}

// (VB, CB, TXL) of a stride-1 call; shared by the launcher and the workspace query
static void lp_s1_shape(int N, int D, int H, int W, int NB, int& vb, int& cb, int& txl) {
  // x extent of a wave's 32 columns: the widest power of two that wastes no more columns than a narrower one would
  txl = (W >= 24) ? 5 : (W >= 12 ? 4 : 3);
  // rows per wave (VB) and cout blocks per wave (CB): 4 x 1 for 32-cout layers (one weight fragment feeds 4 matrix
  // instructions), 4 x 2 from 64 couts on.  Small grids keep the 4 rows (weight reuse) and split the input channels instead
  cb = NB >= 2 ? 2 : 1;
  const long vox = (long)N * D * H * W;
  vb = vox >= 4096 ? 4 : (vox >= 1024 ? 2 : 1);
  if (vb == 4 && txl == 5) cb = 1;   // 4 x 2 tiles (128 accumulators + weight ring + halo prefetch) spill at two waves per SIMD
}
static long lp_s1_wgs(int N, int D, int H, int W, int NB, int vb, int cb, int txl) {
  const int TX = 1 << txl, TY = vb * (32 / TX);
  return (long)N * ((D + 3) / 4) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX) * ((NB + cb - 1) / cb);
}

template <typename T>
static int lp_s1_dispatch(const LpS1Params& p, void* ws, long ws_bytes, hipStream_t stream) {
  int vb, cb, txl;
  lp_s1_shape(p.N, p.D, p.H, p.W, p.NB, vb, cb, txl);
#define LP_S1_CASE(VB_, CB_, TXL_) if (vb == VB_ && cb == CB_ && txl == TXL_) return lp_s1_launch<T, VB_, CB_, TXL_>(p, ws, ws_bytes, stream);
  LP_S1_CASE(4, 1, 5) LP_S1_CASE(4, 2, 5) LP_S1_CASE(2, 1, 5) LP_S1_CASE(2, 2, 5) LP_S1_CASE(1, 1, 5) LP_S1_CASE(1, 2, 5)
  LP_S1_CASE(4, 1, 4) LP_S1_CASE(4, 2, 4) LP_S1_CASE(2, 1, 4) LP_S1_CASE(2, 2, 4) LP_S1_CASE(1, 1, 4) LP_S1_CASE(1, 2, 4)
  LP_S1_CASE(4, 1, 3) LP_S1_CASE(4, 2, 3) LP_S1_CASE(2, 1, 3) LP_S1_CASE(2, 2, 3) LP_S1_CASE(1, 1, 3) LP_S1_CASE(1, 2, 3)
#undef LP_S1_CASE
  return BTS_ERR_UNSUPPORTED;
}

// =====================================================================================================================
// gather-form convolution (1x1x1, stride 2, transposed): operands straight from global memory
// =====================================================================================================================
struct LpTap { short dz, dy, dx, w; };   // input offset of the tap, index of its weight slab
struct LpGatherParams {
  const unsigned short* x;
  const unsigned short* wp;
  const float* bias;
  unsigned short* y;
  int N, Di, Hi, Wi, ldx;           // input grid
  int Dg, Hg, Wg;                   // grid of this launch (output positions of one class)
  int Do, Ho, Wo, ldy, Cout;        // output tensor
  int s, os, ooz, ooy, oox;         // in = g*s + off_t ; out = g*os + oo
  int KS, NB, ncg, ntaps, accum;
  long npos;                        // N*Dg*Hg*Wg
  double* gap_part;                 // fused global-average-pool partials [position block][Cout] (1x1x1 launches only), else NULL
  LpTap taps[27];
};

template <typename T, int VB, int CB>
__global__ __launch_bounds__(256, 2) void lp_conv_gather_kernel(const LpGatherParams p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int cg = blockIdx.x % p.ncg;
  const long blk = blockIdx.x / p.ncg;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  const unsigned wlane = (unsigned)((h * 32 + l32) * 16);
  // this lane's VB grid positions
  int gz[VB], gy[VB], gx[VB], gn[VB];
  bool live[VB];
#pragma unroll
  for (int v = 0; v < VB; ++v) {
    long pos = ((blk * 4 + wave) * VB + v) * 32 + l32;
    live[v] = pos < p.npos;
    if (!live[v]) pos = 0;
    gx[v] = (int)(pos % p.Wg); pos /= p.Wg;
    gy[v] = (int)(pos % p.Hg); pos /= p.Hg;
    gz[v] = (int)(pos % p.Dg);
    gn[v] = (int)(pos / p.Dg);
  }
  f32x16 acc[VB][CB];
#pragma unroll
  for (int v = 0; v < VB; ++v)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[v][c][r] = 0.f;
  // Software pipeline over the flattened (tap, k-step) sequence: the operands of step i+1 are requested before the matrix
  // instructions of step i run (both operands come straight from global memory: a step without its successor in flight
  // would wait a full memory round trip for every 4-8 matrix instructions)
  const unsigned short* src[VB];   // of the tap being REQUESTED
  bool ok[VB];
  int wtap = 0;
  auto tap_setup = [&](int t) {
    const LpTap tp = p.taps[t];
    wtap = tp.w;
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      const int iz = gz[v] * p.s + tp.dz, iy = gy[v] * p.s + tp.dy, ix = gx[v] * p.s + tp.dx;
      ok[v] = live[v] && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      src[v] = p.x + ((((long)gn[v] * p.Di + iz) * p.Hi + iy) * p.Wi + ix) * (long)p.ldx + h * 8;
    }
  };
  auto request = [&](int ks, u32x4 (&a)[CB], u32x4 (&b)[VB]) {
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int cb = cg * CB + c;
      a[c] = bload16(wr, wlane, (unsigned)((((wtap * p.KS + ks) * p.NB) + (cb < p.NB ? cb : 0)) * 1024));
    }
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      b[v] = u32x4{0u, 0u, 0u, 0u};
      if (ok[v]) b[v] = *reinterpret_cast<const u32x4*>(src[v] + ks * 16);
    }
  };
  // ring of GD + 1 operand sets: step i computes on set i % (GD + 1) while the requests of steps i+1 .. i+GD are in flight
  constexpr int GD = (VB + CB >= 6) ? 2 : 3;   // (4 x 2 tiles: a third set in flight would spill)
  u32x4 ar[GD + 1][CB], br[GD + 1][VB];
  const int total = p.ntaps * p.KS;
  int rt = 0, rks = 0, issued = 0;
  tap_setup(0);
  auto issue = [&](u32x4 (&a)[CB], u32x4 (&b)[VB]) {   // request the operands of step `issued` (no-op past the end)
    if (issued < total) {
      request(rks, a, b);
      ++issued;
      if (++rks == p.KS) { rks = 0; if (++rt < p.ntaps) tap_setup(rt); }
    }
  };
#pragma unroll
  for (int j = 0; j < GD; ++j) issue(ar[j], br[j]);
  for (int i0 = 0; i0 < total; i0 += GD + 1) {
#pragma unroll
    for (int j = 0; j <= GD; ++j) {
      if (i0 + j < total) {
        issue(ar[(j + GD) % (GD + 1)], br[(j + GD) % (GD + 1)]);
#pragma unroll
        for (int v = 0; v < VB; ++v)
#pragma unroll
          for (int c = 0; c < CB; ++c) acc[v][c] = T::mfma(ar[j][c], br[j][v], acc[v][c]);
      }
    }
  }
  float csum[CB][16];   // column sums of what this lane stores (dead code unless p.gap_part)
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) csum[c][r] = 0.f;
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int cb = cg * CB + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = cb * 32 + 8 * q + 4 * h;
      float bq[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && cb < p.NB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (co + j < p.Cout) bq[j] = p.bias[co + j];
      }
#pragma unroll
      for (int v = 0; v < VB; ++v) {
        if (live[v] && cb < p.NB && co < p.Cout) {
          const int oz = gz[v] * p.os + p.ooz, oy = gy[v] * p.os + p.ooy, ox = gx[v] * p.os + p.oox;
          if (oz < p.Do && oy < p.Ho && ox < p.Wo) {
            unsigned short* dst = p.y + ((((long)gn[v] * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * (long)p.ldy + co;
            const float o0 = acc[v][c][4 * q] + bq[0], o1 = acc[v][c][4 * q + 1] + bq[1], o2 = acc[v][c][4 * q + 2] + bq[2],
                        o3 = acc[v][c][4 * q + 3] + bq[3];
            lp_store_quad<T>(dst, o0, o1, o2, o3, p.Cout - co, p.accum);
            csum[c][4 * q] += o0; csum[c][4 * q + 1] += o1; csum[c][4 * q + 2] += o2; csum[c][4 * q + 3] += o3;
          }
        }
      }
    }
  }
  // Fused global average pool of the output (resnet.py:121: the squeeze of the block's 1x1x1 shortcut output): column sums of
  // this block's 128 * VB positions -- lanes (xor shuffles over the 32 positions of a wave), the 4 waves through LDS in fixed
  // order, one fp64 partial per (position block, cout); bts_lp_conv1_gap's finalize adds the blocks of a sample
  if (p.gap_part != nullptr) {   // (launch-uniform)
    __shared__ float csh[4][CB * 32];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = csum[c][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (l32 == 0) csh[wave][c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = v;
      }
    __syncthreads();
    if (tid < CB * 32) {
      const int co = cg * CB * 32 + tid;
      if (co < p.Cout) p.gap_part[blk * p.Cout + co] = ((double)csh[0][tid] + (double)csh[1][tid]) + ((double)csh[2][tid] + (double)csh[3][tid]);
    }
  }
}

// Stride-2 gather with whole-row loads.  In the kernel above a B operand is 16 bytes of each of 32 voxels that sit two rows apart:
// an instruction touches 32 cache lines and uses 32 bytes of each, every line comes back for the other k-steps, and the L2 -> L1 fill
// rate bounds the launch (0.09 of the matrix peak at 32 channels).  Here the k-steps of a tap go in groups of GK = 2 | 4 (64 | 128
// bytes of a voxel's row): load instruction i has the GK lanes of a quad fetch the GK * 32 contiguous bytes of output voxel
// (quad base + i), the quad transpose (lowp_common.h) hands every lane its own voxel's pieces, one per k-step.  Two register sets:
// the next group's rows and weight fragments are in flight while the current group multiplies.  Needs Wg % GK == 0 (a quad never
// leaves its output row).
// VB = 4 (round 5): four position groups per wave share every weight fragment -- a wave's weight re-streaming from L2 (one fragment per
// two matrix instructions at VB = 2, as much traffic as the activations) halves; GK = 2 only (register budget: 128 accumulators + two
// operand sets).
template <typename T, int CB, int GK, int VB = 2>
__global__ __launch_bounds__(256, VB == 4 ? 1 : 2) void lp_conv_gatherq_kernel(const LpGatherParams p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31, b = l32 & (GK - 1);
  const int cg = blockIdx.x % p.ncg;
  const long blk = blockIdx.x / p.ncg;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  const unsigned wlane = (unsigned)((h * 32 + l32) * 16);
  int gz[VB], gy[VB], gx[VB], gn[VB];
  bool live[VB];
  const unsigned short* base[VB];     // own voxel's quad base (n, s gz, s gy, s (gx - b)), this lane's piece of a chunk
#pragma unroll
  for (int v = 0; v < VB; ++v) {
    long pos = ((blk * 4 + wave) * VB + v) * 32 + l32;
    live[v] = pos < p.npos;           // (npos is a multiple of Wg, Wg of GK: the lanes of a quad are live together)
    if (!live[v]) pos = 0;
    gx[v] = (int)(pos % p.Wg); pos /= p.Wg;
    gy[v] = (int)(pos % p.Hg); pos /= p.Hg;
    gz[v] = (int)(pos % p.Dg);
    gn[v] = (int)(pos / p.Dg);
    base[v] = p.x + ((((long)gn[v] * p.Di + gz[v] * p.s) * p.Hi + gy[v] * p.s) * p.Wi + (long)(gx[v] - b) * p.s) * (long)p.ldx + (2 * b + h) * 8;
  }
  f32x16 acc[VB][CB];
#pragma unroll
  for (int v = 0; v < VB; ++v)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[v][c][r] = 0.f;
  const int NQ = p.KS / GK;                 // chunks per tap
  const int total = p.ntaps * NQ;
  int rt = 0, rq = 0, issued = 0;
  u32x4 a0[GK][CB], a1[GK][CB], b0[VB][GK], b1[VB][GK];
  auto issue = [&](u32x4 (&a)[GK][CB], u32x4 (&bb)[VB][GK]) {
    if (issued >= total) return;
    const LpTap tp = p.taps[rt];
#pragma unroll
    for (int j = 0; j < GK; ++j)
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int cb = cg * CB + c;
        a[j][c] = bload16(wr, wlane, (unsigned)((((tp.w * p.KS + rq * GK + j) * p.NB) + (cb < p.NB ? cb : 0)) * 1024));
      }
    const long off = (((long)tp.dz * p.Hi + tp.dy) * p.Wi + tp.dx) * (long)p.ldx + rq * (GK * 16);
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      const int iz = gz[v] * p.s + tp.dz, iy = gy[v] * p.s + tp.dy;
      const bool okzy = live[v] && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi;
#pragma unroll
      for (int i = 0; i < GK; ++i) {
        const int ix = (gx[v] - b + i) * p.s + tp.dx;
        bb[v][i] = u32x4{0u, 0u, 0u, 0u};
        if (okzy && (unsigned)ix < (unsigned)p.Wi) bb[v][i] = *reinterpret_cast<const u32x4*>(base[v] + off + (long)i * p.s * p.ldx);
      }
    }
    ++issued;
    if (++rq == NQ) { rq = 0; ++rt; }
  };
  auto compute = [&](u32x4 (&a)[GK][CB], u32x4 (&bb)[VB][GK]) {
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      if constexpr (GK == 4) k1_quad_transpose(bb[v], b); else k1_pair_transpose(bb[v], b);
    }
#pragma unroll
    for (int j = 0; j < GK; ++j)
#pragma unroll
      for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[v][c] = T::mfma(a[j][c], bb[v][j], acc[v][c]);
  };
  issue(a0, b0);
  for (int g = 0; g < total; g += 2) {
    issue(a1, b1);
    compute(a0, b0);
    if (g + 1 < total) {
      issue(a0, b0);
      compute(a1, b1);
    }
  }
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int cb = cg * CB + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = cb * 32 + 8 * q + 4 * h;
      float bq[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && cb < p.NB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (co + j < p.Cout) bq[j] = p.bias[co + j];
      }
#pragma unroll
      for (int v = 0; v < VB; ++v) {
        if (live[v] && cb < p.NB && co < p.Cout) {
          const int oz = gz[v] * p.os + p.ooz, oy = gy[v] * p.os + p.ooy, ox = gx[v] * p.os + p.oox;
          if (oz < p.Do && oy < p.Ho && ox < p.Wo) {
            unsigned short* dst = p.y + ((((long)gn[v] * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * (long)p.ldy + co;
            lp_store_quad<T>(dst, acc[v][c][4 * q] + bq[0], acc[v][c][4 * q + 1] + bq[1], acc[v][c][4 * q + 2] + bq[2], acc[v][c][4 * q + 3] + bq[3],
                             p.Cout - co, p.accum);
          }
        }
      }
    }
  }
}

// positions per workgroup of a gather launch (4 waves x VB x 32)
static int lp_gather_vb(long npos, int NB) {
  const int cb = NB >= 2 ? 2 : 1;
  const long wg4 = ((npos + 511) / 512) * ((NB + cb - 1) / cb);
  return wg4 >= 512 ? 4 : (wg4 >= 128 ? 2 : 1);
}
template <typename T>
static int lp_gather_launch(LpGatherParams p, hipStream_t stream) {
  p.npos = (long)p.N * p.Dg * p.Hg * p.Wg;
  const int cb = p.NB >= 2 ? 2 : 1;
  p.ncg = (p.NB + cb - 1) / cb;
  int vb = lp_gather_vb(p.npos, p.NB);
  // whole-row loads for the stride-2 forms on grids that fill the chip (BTS_LP_GATHERQ=0: the plain kernel, for A/B)
  int gk = 0;
  {
    const char* e = getenv("BTS_LP_GATHERQ");
    if (!(e && atoi(e) == 0) && p.s == 2 && p.gap_part == nullptr && vb >= 2 && p.KS % 2 == 0) {
      if (p.KS % 4 == 0 && p.Wg % 4 == 0) gk = 4;
      else if (p.Wg % 2 == 0) gk = 2;
    }
    // (below ~300 workgroups the plain kernel's smaller tiles win: 128 -> 128 at 8 x 32^3, 256 workgroups, 77 against 90 us)
    { const char* m = getenv("BTS_LP_GATHERQ_MIN"); if (gk && ((p.npos + 255) / 256) * p.ncg < (m ? atol(m) : 288)) gk = 0; }
    if (gk) {
      // four position groups per wave: OFF by default -- measured in round 5 (profiles/r05_ab_e6_gatherq_vb4.txt): the 128 accumulators
      // + two operand sets need 442 registers, i.e. one wave per SIMD, and the batch-8 step loses 1.7 ms (76.6 against 74.9), the
      // inference forward nothing / 0.1 ms.  BTS_LP_GATHERQ_VB4=<n> (n > 1) takes grids of at least n double-size workgroups (tests, A/B)
      const char* v4 = getenv("BTS_LP_GATHERQ_VB4");
      const bool vb4 = v4 && atoi(v4) > 1 && cb == 2 && p.Wg % 2 == 0 && ((p.npos + 511) / 512) * p.ncg >= atol(v4);
      vb = vb4 ? 4 : 2;
      if (vb4) gk = 2;
    }
  }
  const long blocks = ((p.npos + 128L * vb - 1) / (128L * vb)) * p.ncg;
  if (blocks > 0x7fffffffL) return BTS_ERR_SHAPE;
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(31, 2.0 * p.ntaps * 16.0 * p.KS * p.Cout * (double)p.npos, stream);
  (void)hipGetLastError();
  if (gk) {
#define LP_GQ(CB_, GK_) hipLaunchKernelGGL((lp_conv_gatherq_kernel<T, CB_, GK_>), dim3((unsigned)blocks), dim3(256), 0, stream, p)
    if (vb == 4) hipLaunchKernelGGL((lp_conv_gatherq_kernel<T, 2, 2, 4>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else if (cb == 2) { if (gk == 4) LP_GQ(2, 4); else LP_GQ(2, 2); }
    else { if (gk == 4) LP_GQ(1, 4); else LP_GQ(1, 2); }
#undef LP_GQ
    if (prof) bts_prof_end(stream);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
  }
#define LP_G_CASE(VB_, CB_) if (vb == VB_ && cb == CB_) hipLaunchKernelGGL((lp_conv_gather_kernel<T, VB_, CB_>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
  LP_G_CASE(4, 1) LP_G_CASE(4, 2) LP_G_CASE(2, 1) LP_G_CASE(2, 2) LP_G_CASE(1, 1) LP_G_CASE(1, 2)
#undef LP_G_CASE
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// geometry-driven core of both entry points below.  geo: 0 = 1x1x1, 1 = 3x3x3 stride 1, 2 = stride-2 gather (out = ceil(in/2),
// in = 2o + k - pad), 3 = 8 output-parity classes of the transposed form (out = 2 in; even outputs take (i, k=0) and (i-1, k=2),
// odd ones (i, k=1)).  (D,H,W) are the dims of `x`, the tensor the taps read; Cin its channels (the contraction).
static int lp_conv_run(int geo, int dtype, const void* x, const void* wp, const float* bias, void* y, void* workspace, long workspace_bytes,
                       int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, int accum, hipStream_t stream,
                       double* gap_part = nullptr, double* gn_part = nullptr, int gn_G = 0, const LpGnbFuse* gnb = nullptr,
                       const LpGnaFuse* gna = nullptr) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return BTS_ERR_SHAPE;
  // (results are stored four couts = 8 bytes at a time; a head with fewer than four output channels stores them one by one)
  const bool vec_out = Cout >= 4;
  if (Cin % 16 != 0 || ldx % 8 != 0 || (vec_out && ldy % 4 != 0) || ldx < Cin || ldy < Cout) return BTS_ERR_ALIGN;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & (vec_out ? 7 : 1)) || (((uintptr_t)wp) & 15)) return BTS_ERR_ALIGN;
  const int KS = Cin / 16, NB = (Cout + 31) / 32;
  if (geo == 1) {
    if (((long)(D + 2) * H * W + 64) * (long)ldx * 2 >= 0x7fffffffL) return BTS_ERR_SHAPE;   // 31-bit offsets inside one volume
    {   // few channels on a big volume: the z-marching streaming kernel (lowp_s1z.hip); same image part as the tiled DMA kernel
      const int r = bts_lp_s1z_launch_(dtype, x, reinterpret_cast<const char*>(wp) + lp_s1d_part_offset(Cin, Cout), bias, y, N, D, H, W, Cin, ldx,
                                       Cout, ldy, accum, gn_part, gn_G, stream, gnb, gna);
      if (r != 1) return r;
    }
    if (gnb != nullptr || gna != nullptr) return 1;      // (only the streaming kernel has these epilogue / prologue forms: nothing was launched)
    {
      const int r = bts_lp_s1d_launch_(dtype, x, reinterpret_cast<const char*>(wp) + lp_s1d_part_offset(Cin, Cout), bias, y, workspace,
                                       workspace_bytes, N, D, H, W, Cin, ldx, Cout, ldy, accum, gn_part, gn_G, stream);
      if (r != 1) return r;
    }
    LpS1Params p;
    p.x = (const unsigned short*)x; p.wp = (const unsigned short*)wp; p.bias = bias; p.y = (unsigned short*)y;
    p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldy; p.Cout = Cout; p.KS = KS; p.NB = NB; p.accum = accum;
    p.gnp = gn_part; p.gn_G = gn_G; p.gn_zt = gn_G > 0 ? D / gn_G : 1;
    return dtype == LP_F16 ? lp_s1_dispatch<TF16>(p, workspace, workspace_bytes, stream) : lp_s1_dispatch<TBF16>(p, workspace, workspace_bytes, stream);
  }
  LpGatherParams g;
  g.x = (const unsigned short*)x; g.wp = (const unsigned short*)wp; g.bias = bias; g.y = (unsigned short*)y;
  g.N = N; g.Di = D; g.Hi = H; g.Wi = W; g.ldx = ldx; g.ldy = ldy; g.Cout = Cout; g.KS = KS; g.NB = NB; g.accum = accum;
  g.gap_part = (geo == 0) ? gap_part : nullptr;
  auto run = [&](const LpGatherParams& q) { return dtype == LP_F16 ? lp_gather_launch<TF16>(q, stream) : lp_gather_launch<TBF16>(q, stream); };
  if (geo == 0 && gap_part == nullptr) {     // (the fused-pool form is offered the streaming kernel by bts_lp_conv1_gap itself)
    const int r = bts_lp_k1_launch_(dtype, x, wp, bias, y, (long)N * D * H * W, Cin, ldx, Cout, ldy, accum, nullptr, 0, stream);
    if (r != 1) return r;
  }
  if (geo == 0) {
    g.Dg = g.Do = D; g.Hg = g.Ho = H; g.Wg = g.Wo = W; g.s = 1; g.os = 1; g.ooz = g.ooy = g.oox = 0; g.ntaps = 1;
    g.taps[0] = LpTap{0, 0, 0, 0};
    return run(g);
  }
  if (geo == 2) {   // TF 'same', stride 2: out = ceil(in/2), pad_before = max((out-1)*2+3-in, 0) / 2
    {   // 32 -> <= 32 channels on a big even grid: the LDS-tiled kernel (lowp_s2t.hip)
      const int r = bts_lp_s2t_launch_(dtype, x, wp, bias, y, N, D, H, W, Cin, ldx, Cout, ldy, accum, stream);
      if (r != 1) return r;
    }
    g.Do = (D + 1) / 2; g.Ho = (H + 1) / 2; g.Wo = (W + 1) / 2;
    g.Dg = g.Do; g.Hg = g.Ho; g.Wg = g.Wo; g.s = 2; g.os = 1; g.ooz = g.ooy = g.oox = 0; g.ntaps = 27;
    auto padb = [](int in, int out) { const int t = (out - 1) * 2 + 3 - in; return t > 0 ? t / 2 : 0; };
    const int pz = padb(D, g.Do), py = padb(H, g.Ho), px = padb(W, g.Wo);
    for (int t = 0; t < 27; ++t) g.taps[t] = LpTap{(short)(t / 9 - pz), (short)((t / 3) % 3 - py), (short)(t % 3 - px), (short)t};
    return run(g);
  }
  if (geo == 3) {  // y[2i+k] += x[i] w[k], cropped to [0, 2n): 8 output-parity classes, every output written once
    {
      const int r = bts_lp_up_launch_(dtype, x, reinterpret_cast<const char*>(wp) + lp_s1d_part_offset(Cin, Cout), bias, y, N, D, H, W, Cin, ldx,
                                      Cout, ldy, accum, stream, gn_part, gn_G);
      if (r != 1) return r;
    }
    if (gn_part != nullptr) return 1;      // (only the merged kernel emits the statistics: nothing was launched)
    g.Do = 2 * D; g.Ho = 2 * H; g.Wo = 2 * W; g.Dg = D; g.Hg = H; g.Wg = W; g.s = 1; g.os = 2;
    for (int cls = 0; cls < 8; ++cls) {
      const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
      int ozs[2], kzs[2], oys[2], kys[2], oxs[2], kxs[2];
      auto fill = [](int par, int* off, int* k) { if (par) { off[0] = 0; k[0] = 1; return 1; } off[0] = 0; k[0] = 0; off[1] = -1; k[1] = 2; return 2; };
      const int noz = fill(pz, ozs, kzs), noy = fill(py, oys, kys), nox = fill(px, oxs, kxs);
      int nt = 0;
      for (int a = 0; a < noz; ++a)
        for (int b2 = 0; b2 < noy; ++b2)
          for (int c = 0; c < nox; ++c)
            g.taps[nt++] = LpTap{(short)ozs[a], (short)oys[b2], (short)oxs[c], (short)((kzs[a] * 3 + kys[b2]) * 3 + kxs[c])};
      g.ntaps = nt; g.ooz = pz; g.ooy = py; g.oox = px;
      const int r = run(g);
      if (r != BTS_OK) return r;
    }
    return BTS_OK;
  }
  return BTS_ERR_UNSUPPORTED;
}

static long lp_s1_workspace(int N, int D, int H, int W, int Cin, int Cout) {
  if (Cin % 16 != 0) return 0;
  {
    const long d = bts_lp_s1d_workspace_(N, D, H, W, Cin, Cout);
    if (d >= 0) return d;
  }
  const int NB = (Cout + 31) / 32;
  int vb, cb, txl;
  lp_s1_shape(N, D, H, W, NB, vb, cb, txl);
  const int ks = lp_s1_ksplit(lp_s1_wgs(N, D, H, W, NB, vb, cb, txl), Cin / 16);
  return ks > 1 ? (long)ks * N * D * H * W * NB * 32 * 4 : 0;
}
extern "C" long bts_lp_conv3d_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  return kind == BTS_CONV_K3S1 ? lp_s1_workspace(N, D, H, W, Cin, Cout) : 0;
}
// y = conv(x) + bias in the storage type.  x (N,D,H,W,Cin) stride ldx (elements); y (N,D',H',W',Cout) stride ldy; D' = D | D/2
// (TF 'same', stride 2) | 2D (transposed).  Cin must be a multiple of 16 and ldx / ldy / the views' first channel multiples of 8.
extern "C" int bts_lp_conv3d_fwd(int kind, int dtype, const void* x, const void* wp, const float* bias, void* y, void* workspace,
                                 long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, hipStream_t stream) {
  if (kind < 0 || kind > 3) return BTS_ERR_UNSUPPORTED;
  return lp_conv_run(kind, dtype, x, wp, bias, y, workspace, workspace_bytes, N, D, H, W, Cin, ldx, Cout, ldy, 0, stream);
}
// y = conv3x3x3(x) + bias in the storage type (dense) AND the slab-mode GroupNorm statistics of y (group_norm.py:83-108 as the
// reference runs it on channels_last data: resnet.py:80-93 conv -> GroupNormalization) in one pass: (sum, sumsq) partials leave the
// conv's epilogue per (z plane, tile column), bts_gn_finalize_partials_ turns them into mean / rstd.  Grids that split the input
// channels, z-slabs that are not whole planes (D % G != 0) and heads with Cout % 4 != 0 run the conv and bts_lp_gn_stats on the stored y.
static bool lp_s1_gn_plan(int N, int D, int H, int W, int Cin, int Cout, int G, long* B) {
  if (Cin % 16 != 0 || Cout % 4 != 0 || G <= 0) return false;
  if (D % G != 0) {      // slabs that are not whole planes: only the split-K finish of the DMA kernel counts them
    const bool s1z_takes_it = bts_lp_s1z_gn_B_(N, D, H, W, Cin, Cout, 1) > 0;      // (offered first by lp_conv_run)
    const bool splits = bts_lp_s1d_workspace_(N, D, H, W, Cin, Cout) > 0;
    *B = (!s1z_takes_it && splits) ? bts_lp_s1d_gn_B_(N, D, H, W, Cin, Cout, G) : 0;
    return *B > 0;
  }
  *B = bts_lp_s1z_gn_B_(N, D, H, W, Cin, Cout, G);          // the streaming kernel takes this shape: its partial layout
  if (*B > 0) return true;
  if (bts_lp_s1d_workspace_(N, D, H, W, Cin, Cout) >= 0) {      // the DMA kernel takes this shape: its partial layout
    *B = bts_lp_s1d_gn_B_(N, D, H, W, Cin, Cout, G);
    return *B > 0;
  }
  const int NB = (Cout + 31) / 32;
  int vb, cb, txl;
  lp_s1_shape(N, D, H, W, NB, vb, cb, txl);
  if (lp_s1_ksplit(lp_s1_wgs(N, D, H, W, NB, vb, cb, txl), Cin / 16) > 1) return false;
  const int TX = 1 << txl, TY = vb * (32 / TX);
  *B = (long)(D / G) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX) * ((NB + cb - 1) / cb);
  return true;
}
extern "C" long bts_lp_conv3d_fwd_gn_workspace(int N, int D, int H, int W, int Cin, int Cout, int G) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || Cout % G != 0) return -1;
  long B = 0;
  const long fused = lp_s1_gn_plan(N, D, H, W, Cin, Cout, G, &B) ? (long)N * G * B * 16 + 64 : 0;
  const long conv = ((lp_s1_workspace(N, D, H, W, Cin, Cout) + 63) / 64) * 64;
  const long stats = bts_lp_gn_workspace(N, (long)D * H * W, Cout, G);
  return conv + (fused > stats ? fused : stats) + 64;
}
static int lp_conv3d_fwd_gn_impl(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd,
                                 void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G,
                                 float eps, int accum, hipStream_t stream);
extern "C" int bts_lp_conv3d_fwd_gn(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd,
                                    void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G,
                                    float eps, hipStream_t stream) {
  return lp_conv3d_fwd_gn_impl(dtype, x, wp, bias, y, mean, rstd, workspace, workspace_bytes, N, D, H, W, Cin, ldx, Cout, G, eps, 0, stream);
}
// The same, ADDING the convolution to what y already holds (accumulate != 0), the statistics taken of the final sums: a contraction
// split over its input channels whose parts become available at different times -- the decoder's conv1 over [skip | up-sampled]
// (decoder.py:75, resnet.py:134): the skip part (with the bias) can run as soon as the encoder level is done, the up-sampled part adds to
// it later (pass bias = NULL then).  The partial sum passes through the storage type once (as in the two-pass form of lowp_s1z.hip).
extern "C" int bts_lp_conv3d_fwd_gn_acc(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd,
                                        void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G,
                                        float eps, int accumulate, hipStream_t stream) {
  return lp_conv3d_fwd_gn_impl(dtype, x, wp, bias, y, mean, rstd, workspace, workspace_bytes, N, D, H, W, Cin, ldx, Cout, G, eps, accumulate ? 1 : 0,
                               stream);
}
static int lp_conv3d_fwd_gn_impl(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd,
                                 void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G,
                                 float eps, int accum, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || Cout % G != 0) return BTS_ERR_SHAPE;
  if (workspace == nullptr || workspace_bytes < bts_lp_conv3d_fwd_gn_workspace(N, D, H, W, Cin, Cout, G) || (((uintptr_t)workspace) & 15))
    return BTS_ERR_WORKSPACE;
  const long conv_ws = ((lp_s1_workspace(N, D, H, W, Cin, Cout) + 63) / 64) * 64;
  char* tail = reinterpret_cast<char*>(workspace) + conv_ws;
  const long V = (long)D * H * W;
  long B = 0;
  // (accumulating: the split-K finish writes y from its partial sums and cannot add to it with statistics -- such grids take the
  // statistics from the stored result)
  const bool split = bts_lp_s1z_gn_B_(N, D, H, W, Cin, Cout, 1) <= 0 && bts_lp_s1d_workspace_(N, D, H, W, Cin, Cout) > 0;
  if (!(accum && split) && lp_s1_gn_plan(N, D, H, W, Cin, Cout, G, &B)) {
    double* part = reinterpret_cast<double*>(tail);
    const int r = lp_conv_run(1, dtype, x, wp, bias, y, workspace, conv_ws, N, D, H, W, Cin, ldx, Cout, Cout, accum, stream, nullptr, part, G);
    if (r != BTS_OK) return r;
    return bts_gn_finalize_partials_(part, mean, rstd, N * G, B, (double)(V * Cout / G), eps, stream);
  }
  const int r = lp_conv_run(1, dtype, x, wp, bias, y, workspace, conv_ws, N, D, H, W, Cin, ldx, Cout, Cout, accum, stream);
  if (r != BTS_OK) return r;
  return bts_lp_gn_stats(dtype, y, mean, rstd, tail, workspace_bytes - conv_ws, N, V, Cout, G, BTS_GN_SLAB, eps, stream);
}
// conv1 AND the shortcut of a ResnetBlock from ONE pass over the block input (resnet.py:118 `res = conv3d_ptwise(inputs)` and resnet.py:134
// `x = conv3d_1(inputs)` read the same tensor): y = conv3x3x3(x) + bias with GroupNorm `G`'s statistics of y (as bts_lp_conv3d_fwd_gn), res =
// conv1x1x1(x) + bias_pt and gap[n][c] = mean over the voxels of the unrounded res (as bts_lp_conv1_gap) -- the shortcut is a second set of
// output columns at the centre tap of the z-marching kernel's input planes (lowp_s1z.hip, FS form): x is read once, the 1x1x1 launch and
// its read of x go away.  wp = bts_lp_pack(K3S1, FWD), wp_pt = bts_lp_pack(K1, FWD) with the same Cin_slab / fold; y and res dense
// (N,D,H,W,Cout).  The workspace query returns -1 and the call 1 (nothing launched) where the streaming kernel does not take the shape:
// the caller runs bts_lp_conv1_gap + bts_lp_conv3d_fwd_gn.  BTS_LP_FS=0 in the environment: never (A/B aid).
__global__ __launch_bounds__(256) void lp_colsum_finalize_kernel(const double* partial, float* out, int N, int C, int B, double scale);
extern "C" long bts_lp_conv3d_fwd_gn_shortcut_workspace(int N, int D, int H, int W, int Cin, int ldx, int Cout, int G) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || Cout % G != 0 || D % G != 0 || Cin % 16 != 0) return -1;
  static const bool off = [] { const char* e = getenv("BTS_LP_FS"); return e && atoi(e) == 0; }();
  static const bool off64 = [] { const char* e = getenv("BTS_LP_FS_PAIR"); return e && atoi(e) == 0; }();      // BTS_LP_FS_PAIR=0: not on the two-pass 64-channel form (A/B)
  if (off || (off64 && Cin == 64)) return -1;
  // (two dense 32-channel operands -- voxel stride 32 under 64 channels: the two-pass form at any size, planned like one of its passes)
  const long Bg = bts_lp_s1z_gn_B_(N, D, H, W, (Cin == 64 && ldx < 64) ? 32 : Cin, Cout, G), Bf = bts_lp_s1z_fs_B_(N, D, H, W, Cin, ldx, Cout, Cout);
  if (Bg <= 0 || Bf <= 0) return -1;
  return (long)N * G * Bg * 16 + 64 + (long)N * Bf * Cout * 8 + 64;
}
extern "C" int bts_lp_conv3d_fwd_gn_shortcut(int dtype, const void* x, long x_split, const void* wp, const float* bias, void* y, float* mean,
                                             float* rstd, const void* wp_pt, const float* bias_pt, void* res, float* gap, void* workspace,
                                             long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G, float eps,
                                             hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  const long need = bts_lp_conv3d_fwd_gn_shortcut_workspace(N, D, H, W, Cin, ldx, Cout, G);
  if (need < 0) return 1;
  if (workspace == nullptr || workspace_bytes < need || (((uintptr_t)workspace) & 15)) return BTS_ERR_WORKSPACE;
  if (x == nullptr || wp == nullptr || wp_pt == nullptr || y == nullptr || res == nullptr || gap == nullptr || mean == nullptr || rstd == nullptr)
    return BTS_ERR_ALIGN;
  if (ldx % 8 != 0 || ldx < (x_split ? 32 : Cin) || (((uintptr_t)x) & 15) || (((uintptr_t)wp) & 15)) return BTS_ERR_ALIGN;
  if (x_split != 0 && (Cin != 64 || x_split < 0 || x_split % 8 != 0)) return BTS_ERR_SHAPE;      // (two 32-channel operands: the two-pass form)
  const long Bg = bts_lp_s1z_gn_B_(N, D, H, W, (Cin == 64 && ldx < 64) ? 32 : Cin, Cout, G), Bf = bts_lp_s1z_fs_B_(N, D, H, W, Cin, ldx, Cout, Cout);
  double* gpart = reinterpret_cast<double*>(workspace);
  double* fpart = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + (((long)N * G * Bg * 16 + 63) / 64) * 64);
  const long V = (long)D * H * W;
  const int r = bts_lp_s1z_launch_(dtype, x, reinterpret_cast<const char*>(wp) + lp_s1d_part_offset(Cin, Cout), bias, y, N, D, H, W, Cin, ldx, Cout,
                                   Cout, 0, gpart, G, stream, nullptr, nullptr, nullptr, wp_pt, 0, res, bias_pt, fpart, Cout, Cout, 0,
                                   x_split ? static_cast<const void*>(reinterpret_cast<const unsigned short*>(x) + x_split) : nullptr, x_split ? ldx : 0);
  if (r != BTS_OK) return r;
  const int r2 = bts_gn_finalize_partials_(gpart, mean, rstd, N * G, Bg, (double)(V * Cout / G), eps, stream);
  if (r2 != BTS_OK) return r2;
  (void)hipGetLastError();
  hipLaunchKernelGGL(lp_colsum_finalize_kernel, dim3((N * Cout + 3) / 4), dim3(256), 0, stream, fpart, gap, N, Cout, (int)Bf, 1.0 / (double)V);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
// The same with GroupNorm + ReLU of the INPUT applied on the way in (in_relu must be 1): y = conv3x3x3(relu(GN_in(x))) + bias and the statistics of y --
// conv2 of a ResnetBlock reading conv1's raw output (resnet.py:133-136: conv -> GroupNormalization -> relu -> conv) where no backward
// needs the normalised tensor (inference, test.py:128-151 via Model.call(inference=True)).  The workspace query returns -1 where the
// streaming kernel does not take the shape in this form (the caller runs bts_lp_gn_apply + bts_lp_conv3d_fwd_gn).
extern "C" long bts_lp_conv3d_gnin_fwd_gn_workspace(int N, int D, int H, int W, int Cin, int Cout, int in_G, int G) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || Cout % G != 0 || in_G <= 0 || Cin % in_G != 0) return -1;
  static const bool off = [] { const char* e = getenv("BTS_LP_GNA"); return e && atoi(e) == 0; }();      // A/B: the separate apply pass
  if (off || !bts_lp_s1z_gna_ok_(N, D, H, W, Cin, Cin, Cout, Cout, in_G)) return -1;
  const long B = bts_lp_s1z_gn_B_(N, D, H, W, Cin, Cout, G);
  if (B <= 0) return -1;
  return (long)N * G * B * 16 + 128;
}
extern "C" int bts_lp_conv3d_gnin_fwd_gn(int dtype, const void* x, const float* in_gamma, const float* in_beta, const float* in_mean,
                                         const float* in_rstd, int in_G, int in_relu, const void* wp, const float* bias, void* y, float* mean,
                                         float* rstd, void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int Cout, int G,
                                         float eps, hipStream_t stream) {
  const long need = bts_lp_conv3d_gnin_fwd_gn_workspace(N, D, H, W, Cin, Cout, in_G, G);
  if (need < 0 || !in_relu) return BTS_ERR_UNSUPPORTED;      // (the kernel form that exists applies GroupNorm + ReLU)
  if (workspace == nullptr || workspace_bytes < need || (((uintptr_t)workspace) & 15)) return BTS_ERR_WORKSPACE;
  const long B = bts_lp_s1z_gn_B_(N, D, H, W, Cin, Cout, G);
  double* part = reinterpret_cast<double*>(workspace);
  LpGnaFuse ga{in_gamma, in_beta, in_mean, in_rstd, in_G, Cin / in_G};
  const int r = lp_conv_run(1, dtype, x, wp, bias, y, nullptr, 0, N, D, H, W, Cin, Cin, Cout, Cout, 0, stream, nullptr, part, G, nullptr, &ga);
  if (r == 1) return BTS_ERR_UNSUPPORTED;
  if (r != BTS_OK) return r;
  return bts_gn_finalize_partials_(part, mean, rstd, N * G, B, (double)((long)D * H * W * Cout / G), eps, stream);
}
// y = Conv3DTranspose(k3, s2, 'same')(x) + bias (dense fine tensor, storage type) AND the slab-mode GroupNorm statistics of y -- ConvUpsample
// (upsample.py:28-43: conv -> GroupNormalization) without the statistics pass over the fine tensor: (sum, sumsq) partials leave the
// merged transposed-conv kernel's epilogue per fine plane.  (D,H,W) = the COARSE grid.  Shapes that kernel declines, or fine z-slabs
// that are not whole planes, run the conv and bts_lp_gn_stats on the stored y.
extern "C" long bts_lp_convT3d_fwd_gn_workspace(int N, int D, int H, int W, int Cin, int Cout, int G) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || Cout % G != 0) return -1;
  const long B = bts_lp_up_gn_B_(N, D, H, W, Cin, Cout, G);
  const long fused = B > 0 ? (long)N * G * B * 16 + 64 : 0;
  const long stats = bts_lp_gn_workspace(N, 8L * D * H * W, Cout, G);
  return (fused > stats ? fused : stats) + 64;
}
extern "C" int bts_lp_convT3d_fwd_gn(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd,
                                     void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G,
                                     float eps, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || Cout % G != 0) return BTS_ERR_SHAPE;
  if (workspace == nullptr || workspace_bytes < bts_lp_convT3d_fwd_gn_workspace(N, D, H, W, Cin, Cout, G) || (((uintptr_t)workspace) & 15))
    return BTS_ERR_WORKSPACE;
  const long Vf = 8L * D * H * W;
  const long B = (Cin % 16 == 0) ? bts_lp_up_gn_B_(N, D, H, W, Cin, Cout, G) : 0;
  if (B > 0) {
    double* part = reinterpret_cast<double*>(workspace);
    const int r = lp_conv_run(3, dtype, x, wp, bias, y, nullptr, 0, N, D, H, W, Cin, ldx, Cout, Cout, 0, stream, nullptr, part, G);
    if (r == BTS_OK) return bts_gn_finalize_partials_(part, mean, rstd, N * G, B, (double)(Vf * Cout / G), eps, stream);
    if (r != 1) return r;
  }
  const int r = lp_conv_run(3, dtype, x, wp, bias, y, nullptr, 0, N, D, H, W, Cin, ldx, Cout, Cout, 0, stream);
  if (r != BTS_OK) return r;
  return bts_lp_gn_stats(dtype, y, mean, rstd, workspace, workspace_bytes, N, Vf, Cout, G, BTS_GN_SLAB, eps, stream);
}
__global__ __launch_bounds__(256) void lp_colsum_finalize_kernel(const double* partial, float* out, int N, int C, int B, double scale);
// res = conv1x1x1(x) + bias in the storage type AND gap[n][c] = mean over the voxels of (the unrounded) res -- the block's shortcut
// and the squeeze of its gate (resnet.py:118-121) in one pass: the column sums leave the conv's epilogue as per-block partials, a
// small finalize adds them.  Sample volumes that are not whole position blocks run the conv and bts_lp_colsum on the stored res.
extern "C" long bts_lp_conv1_gap_workspace(int N, long V, int Cout) {
  if (N <= 0 || V <= 0 || Cout <= 0) return -1;
  const long a = ((long)N * V / 128 + 1) * Cout * 8 + 64;      // (VB >= 1: at most N*V/128 position blocks)
  const long b = bts_lp_colsum_workspace(N, V, Cout);
  return a > b ? a : b;
}
static int lp_conv1_gap_impl(int dtype, const void* x, const void* wp, const float* bias, void* res, float* gap, void* workspace,
                             long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldres, int accum, hipStream_t stream);
extern "C" int bts_lp_conv1_gap(int dtype, const void* x, const void* wp, const float* bias, void* res, float* gap, void* workspace,
                                long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldres, hipStream_t stream) {
  return lp_conv1_gap_impl(dtype, x, wp, bias, res, gap, workspace, workspace_bytes, N, D, H, W, Cin, ldx, Cout, ldres, 0, stream);
}
// The same, ADDING to what res already holds (the shortcut over [skip | up-sampled], the skip part computed earlier: see
// bts_lp_conv3d_fwd_gn_acc); gap = mean of the final sums
extern "C" int bts_lp_conv1_gap_acc(int dtype, const void* x, const void* wp, const float* bias, void* res, float* gap, void* workspace,
                                    long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldres, int accumulate,
                                    hipStream_t stream) {
  return lp_conv1_gap_impl(dtype, x, wp, bias, res, gap, workspace, workspace_bytes, N, D, H, W, Cin, ldx, Cout, ldres, accumulate ? 1 : 0, stream);
}
static int lp_conv1_gap_impl(int dtype, const void* x, const void* wp, const float* bias, void* res, float* gap, void* workspace,
                             long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldres, int accum, hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0) return BTS_ERR_SHAPE;
  const long V = (long)D * H * W;
  if (workspace == nullptr || workspace_bytes < bts_lp_conv1_gap_workspace(N, V, Cout) || (((uintptr_t)workspace) & 15)) return BTS_ERR_WORKSPACE;
  const int NB = (Cout + 31) / 32;
  {   // streaming kernel: column sums per block of 256 positions
    // (one sample: a partial row can never span two samples, so the last position block may be ragged -- 20x24x20 = 9600 positions,
    // the deepest level of the full inference volume, is not a multiple of 256)
    int kb = bts_lp_k1_gap_block_((long)N * V, V, Cin, Cout);
    if (kb == 0 && N == 1 && bts_lp_k1_gap_block_(V, 256, Cin, Cout) > 0) kb = 256;
    if (kb > 0 && (V % kb == 0 || N == 1) && ldres == Cout) {
      double* part = reinterpret_cast<double*>(workspace);
      const int r = bts_lp_k1_launch_(dtype, x, wp, bias, res, (long)N * V, Cin, ldx, Cout, ldres, accum, part, kb, stream);
      if (r == BTS_OK) {
        hipLaunchKernelGGL(lp_colsum_finalize_kernel, dim3((N * Cout + 3) / 4), dim3(256), 0, stream, part, gap, N, Cout, (int)((V + kb - 1) / kb), 1.0 / (double)V);
        BTS_LAUNCH_CHECK();
        return BTS_OK;
      }
      if (r != 1) return r;
    }
  }
  const long ppb = 128L * lp_gather_vb((long)N * V, NB);     // positions per block
  if (accum || V % ppb != 0 || ldres != Cout) {      // (the gather kernel's fused column sums do not see the old values: sums of the stored result)
    const int r = lp_conv_run(0, dtype, x, wp, bias, res, nullptr, 0, N, D, H, W, Cin, ldx, Cout, ldres, accum, stream);
    if (r != BTS_OK) return r;
    if (ldres != Cout) return BTS_ERR_UNSUPPORTED;
    return bts_lp_colsum(dtype, res, gap, workspace, workspace_bytes, N, V, Cout, (float)(1.0 / (double)V), stream);
  }
  double* part = reinterpret_cast<double*>(workspace);
  const int r = lp_conv_run(0, dtype, x, wp, bias, res, nullptr, 0, N, D, H, W, Cin, ldx, Cout, ldres, accum, stream, part);
  if (r != BTS_OK) return r;
  hipLaunchKernelGGL(lp_colsum_finalize_kernel, dim3((N * Cout + 3) / 4), dim3(256), 0, stream, part, gap, N, Cout, (int)(V / ppb),
                     1.0 / (double)V);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
// dx (+)= conv^T(dy) (replaces tf.GradientTape for these ops, train.py:142-151).  (D,H,W) are the forward INPUT dims, Cin / Cout
// the forward channel counts; wp_bwd = bts_lp_pack(kind, BTS_ROLE_BWD_DATA, ...).  The data gradient of each kind is one of the
// forward geometries on the role-swapped image: stride 1 -> stride 1 with flipped taps, 1x1x1 -> 1x1x1, stride 2 -> the
// transposed form's parity classes (each input voxel receives from (o, k=i-2o)), transposed -> the stride-2 gather.
extern "C" long bts_lp_conv3d_bwd_data_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  return kind == BTS_CONV_K3S1 ? lp_s1_workspace(N, D, H, W, Cout, Cin) : 0;
}
extern "C" int bts_lp_conv3d_bwd_data(int kind, int dtype, const void* dy, const void* wp_bwd, void* dx, void* workspace,
                                      long workspace_bytes, int N, int D, int H, int W, int Cin, int lddx, int Cout, int lddy, int accum,
                                      hipStream_t stream) {
  switch (kind) {
    case BTS_CONV_K1:
    case BTS_CONV_K3S1:
      return lp_conv_run(kind, dtype, dy, wp_bwd, nullptr, dx, workspace, workspace_bytes, N, D, H, W, Cout, lddy, Cin, lddx, accum, stream);
    case BTS_CONV_K3S2:   // dy lives on the half grid; TF 'same' with even sizes pads (0,1) only, which is what the class tables assume
      if ((D | H | W) & 1) return BTS_ERR_UNSUPPORTED;
      return lp_conv_run(3, dtype, dy, wp_bwd, nullptr, dx, workspace, workspace_bytes, N, D / 2, H / 2, W / 2, Cout, lddy, Cin, lddx, accum, stream);
    case BTS_CONV_K3S2T:  // dy lives on the doubled grid: dx[i] = sum_k dy[2i+k] W[k]
      return lp_conv_run(2, dtype, dy, wp_bwd, nullptr, dx, workspace, workspace_bytes, N, 2 * D, 2 * H, 2 * W, Cout, lddy, Cin, lddx, accum, stream);
  }
  return BTS_ERR_UNSUPPORTED;
}
// dx (+)= conv3x3x3^T(dy) + conv1x1x1^T(dy2) in ONE launch: the data gradients of the two convolutions that read a ResnetBlock's input
// (resnet.py:80-87 conv1 and resnet.py:96-103 the shortcut, both applied to `inputs`: resnet.py:118,134) meet in dx under train.py:151.
// The shortcut's contraction rides on conv1's as an extra K-segment at the CENTRE tap (K more matrix steps per 27 K: 3.7 %), its operand
// dy2 read straight from global memory into the matrix instruction -- instead of a second launch that read-modify-writes the Cin-wide dx.
// dy, dy2: (N,D,H,W,Cout) with voxel strides lddy / lddy2; wp_bwd / wp2_bwd: bts_lp_pack(K3S1 / K1, BTS_ROLE_BWD_DATA, ...) of the two
// kernels with the same Cin_slab / fold.  Shapes the fused kernels do not take run as the two launches they replace (same result up to the
// one extra rounding of the stored intermediate): the call always completes.  *fused (may be NULL): 1 if the one-launch form ran.
// dx_split (elements; 0 = dx is one (N,D,H,W,Cin) view): columns [32 b, 32 b + 32) of the result go to dx + b * dx_split, each block a
// tensor of its own with voxel stride lddx -- the gradient of a concat of 32-channel tensors (decoder.py:75) leaves as dense tensors whose
// readers fetch whole lines.  Only the fused tiled kernel writes that form: ask bts_lp_conv3d_bwd_data_sc_split_ok first.
extern "C" long bts_lp_conv3d_bwd_data_sc_workspace(int N, int D, int H, int W, int Cin, int Cout) {
  return lp_s1_workspace(N, D, H, W, Cout, Cin);
}
extern "C" int bts_lp_conv3d_bwd_data_sc_split_ok(int N, int D, int H, int W, int Cin, int Cout) {
  static const bool off = [] { const char* e = getenv("BTS_LP_SC"); return e && atoi(e) == 0; }();
  static const bool off2 = [] { const char* e = getenv("BTS_LP_SC_SPLIT"); return e && atoi(e) == 0; }();      // BTS_LP_SC_SPLIT=0: never (A/B)
  if (off || off2 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cout % 16 != 0) return 0;
  return bts_lp_s1d_sc_split_ok_(N, D, H, W, Cout, Cin) ? 1 : 0;
}
extern "C" int bts_lp_conv3d_bwd_data_sc(int dtype, const void* dy, const void* wp_bwd, const void* dy2, const void* wp2_bwd, void* dx,
                                         long dx_split, void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int lddx,
                                         int Cout, int lddy, int lddy2, int accum, int* fused, hipStream_t stream) {
  if (fused) *fused = 0;
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return BTS_ERR_SHAPE;
  if (dy == nullptr || dy2 == nullptr || wp_bwd == nullptr || wp2_bwd == nullptr || dx == nullptr) return BTS_ERR_ALIGN;
  // (role-swapped: the contraction runs over the forward's Cout, the columns are the forward's Cin)
  const int K = Cout, Nc = Cin;
  static const bool off = [] { const char* e = getenv("BTS_LP_SC"); return e && atoi(e) == 0; }();      // BTS_LP_SC=0: always the two launches (A/B)
  if (!off && K % 16 == 0 && lddy % 8 == 0 && lddy2 % 8 == 0 && lddy >= K && lddy2 >= K && lddx >= (dx_split ? 32 : Nc) && lddx % 4 == 0 &&
      !(((uintptr_t)dy) & 15) && !(((uintptr_t)dy2) & 15) && !(((uintptr_t)dx) & 7) && dx_split % 4 == 0 && !(((uintptr_t)wp_bwd) & 15) && !(((uintptr_t)wp2_bwd) & 15) &&
      ((long)(D + 2) * H * W + 64) * (long)lddy * 2 < 0x7fffffffL) {
    const char* dma = reinterpret_cast<const char*>(wp_bwd) + lp_s1d_part_offset(K, Nc);
    int r = 1;
    if (dx_split == 0)
      r = bts_lp_s1z_launch_(dtype, dy, dma, nullptr, dx, N, D, H, W, K, lddy, Nc, lddx, accum, nullptr, 0, stream, nullptr, nullptr, dy2, wp2_bwd, lddy2);
    if (r == 1) r = bts_lp_s1d_launch_(dtype, dy, dma, nullptr, dx, workspace, workspace_bytes, N, D, H, W, K, lddy, Nc, lddx, accum, nullptr, 0, stream,
                                       dy2, wp2_bwd, lddy2, dx_split);
    if (r != 1) {
      if (r == BTS_OK && fused) *fused = 1;
      return r;
    }
  }
  if (dx_split != 0) return BTS_ERR_UNSUPPORTED;      // (the two-launch route writes one tensor)
  const int r = lp_conv_run(1, dtype, dy, wp_bwd, nullptr, dx, workspace, workspace_bytes, N, D, H, W, K, lddy, Nc, lddx, accum, stream);
  if (r != BTS_OK) return r;
  return lp_conv_run(0, dtype, dy2, wp2_bwd, nullptr, dx, nullptr, 0, N, D, H, W, K, lddy2, Nc, lddx, 1, stream);
}

// =====================================================================================================================
// element-wise passes of the forward (16-bit in / out, fp32 arithmetic)
// =====================================================================================================================
// fp32 -> storage type, rows of C elements with row strides (elements)
template <typename T>
__global__ __launch_bounds__(256) void lp_cast_kernel(const float* src, long lds_, unsigned short* dst, long ldd, long rows, int C) {
  const long total = rows * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    dst[r * ldd + c] = T::st(src[r * lds_ + c]);
  }
}
// rows of a few channels (the 2-channel volume into its 16-channel matrix step, the 1- and 2-channel VAE tensors): a thread per row --
// the element form spends a 64-bit division per 2 bytes
template <typename T, int C>
__global__ __launch_bounds__(256) void lp_cast_rows_kernel(const float* __restrict__ src, long lds_, unsigned short* __restrict__ dst, long ldd, long rows) {
  for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = src[r * lds_ + c];
    if constexpr (C % 2 == 0) {
#pragma unroll
      for (int c = 0; c < C; c += 2) *reinterpret_cast<unsigned*>(dst + r * ldd + c) = pack2<T>(v[c], v[c + 1]);      // (ldd and the view's first channel even)
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) dst[r * ldd + c] = T::st(v[c]);
    }
  }
}
extern "C" int bts_lp_cast(int dtype, const float* src, long ld_src, void* dst, long ld_dst, long rows, int C, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (rows <= 0 || C <= 0) return BTS_ERR_SHAPE;
  (void)hipGetLastError();
  if (C <= 4 && (C % 2 != 0 || (ld_dst % 2 == 0 && (((uintptr_t)dst) & 3) == 0))) {
    long rb = (rows + 255) / 256;
    if (rb > 65536) rb = 65536;
#define LP_CR(T_, C_) hipLaunchKernelGGL((lp_cast_rows_kernel<T_, C_>), dim3((unsigned)rb), dim3(256), 0, stream, src, ld_src, (unsigned short*)dst, ld_dst, rows)
    if (dtype == LP_F16) { if (C == 1) LP_CR(TF16, 1); else if (C == 2) LP_CR(TF16, 2); else if (C == 3) LP_CR(TF16, 3); else LP_CR(TF16, 4); }
    else { if (C == 1) LP_CR(TBF16, 1); else if (C == 2) LP_CR(TBF16, 2); else if (C == 3) LP_CR(TBF16, 3); else LP_CR(TBF16, 4); }
#undef LP_CR
    BTS_LAUNCH_CHECK();
    return BTS_OK;
  }
  long blocks = (rows * C + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_cast_kernel<TF16>, dim3((unsigned)blocks), dim3(256), 0, stream, src, ld_src, (unsigned short*)dst, ld_dst, rows, C);
  else hipLaunchKernelGGL(lp_cast_kernel<TBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, src, ld_src, (unsigned short*)dst, ld_dst, rows, C);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
template <typename T>
__global__ __launch_bounds__(256) void lp_uncast_kernel(const unsigned short* src, long lds_, float* dst, long ldd, long rows, int C) {
  const long total = rows * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    dst[r * ldd + c] = T::ld(src[r * lds_ + c]);
  }
}
// fp32 rows of C (<= 4) channels -> storage-type rows of 16 channels, the tail zero: the 2-channel volume / VAE-output gradient and
// the 1-channel VAE tensor as whole matrix steps in ONE pass (a zero fill of the 16-channel tensor + bts_lp_cast into its first
// channels wrote every row twice: 0.28 ms per 8 x 128^3 tensor)
template <typename T, int C>
__global__ __launch_bounds__(256) void lp_cast_pad16_kernel(const float* __restrict__ src, long lds_, unsigned short* __restrict__ dst, long rows) {
  for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = src[r * lds_ + c];
    u32x4* o = reinterpret_cast<u32x4*>(dst + r * 16);
    o[0] = u32x4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), 0u, 0u};
    o[1] = u32x4{0u, 0u, 0u, 0u};
  }
}
extern "C" int bts_lp_cast_pad16(int dtype, const float* src, long ld_src, void* dst, long rows, int C, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (rows <= 0 || C <= 0 || C > 4 || ld_src < C) return BTS_ERR_SHAPE;
  if (((uintptr_t)dst) & 15) return BTS_ERR_ALIGN;
  long blocks = (rows + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  (void)hipGetLastError();
#define LP_CP(TT, C_) hipLaunchKernelGGL((lp_cast_pad16_kernel<TT, C_>), dim3((unsigned)blocks), dim3(256), 0, stream, src, ld_src, (unsigned short*)dst, rows)
  if (dtype == LP_F16) { if (C == 1) LP_CP(TF16, 1); else if (C == 2) LP_CP(TF16, 2); else if (C == 3) LP_CP(TF16, 3); else LP_CP(TF16, 4); }
  else { if (C == 1) LP_CP(TBF16, 1); else if (C == 2) LP_CP(TBF16, 2); else if (C == 3) LP_CP(TBF16, 3); else LP_CP(TBF16, 4); }
#undef LP_CP
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
// The input dropout of the encoder (encoder.py:39,71: Dropout(0.2) on the 2-channel volume) AND the cast into the zero-padded 16-channel
// matrix step in ONE pass: element i of the dense (rows, C) fp32 volume is kept (and scaled by 1 / (1 - rate)) where the counter-based
// generator of bts_dropout_mask says so -- the SAME draw, element for element (u01(seed, i) >= rate) -- so the result equals
// bts_dropout_mask + bts_dropout_apply + bts_lp_cast_pad16 bit for bit, without the mask and the dropped fp32 volume ever being written
// (three launches, 0.30 ms per 8 x 128^3 step -> one).  The first block's input has no gradient: nothing downstream needs the mask.
template <typename T, int C>
__global__ __launch_bounds__(256) void lp_dropout_cast_pad16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long rows, float rate,
                                                                    float scale, uint64_t seed) {
  for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      // (the product must be rounded to fp32 BEFORE the cast, as the three-pass route stores it: left to the compiler -- __fmul_rn
      // included -- multiply and fp16 conversion become one mixed-precision instruction, rounded once: 1 element in 30,000 differs)
      float pr = src[r * C + c] * scale;
      asm volatile("" : "+v"(pr));
      v[c] = u01(seed, (uint64_t)(r * C + c)) >= rate ? pr : 0.f;
    }
    u32x4* o = reinterpret_cast<u32x4*>(dst + r * 16);
    o[0] = u32x4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), 0u, 0u};
    o[1] = u32x4{0u, 0u, 0u, 0u};
  }
}
extern "C" int bts_lp_dropout_cast_pad16(int dtype, const float* src, void* dst, long rows, int C, float rate, uint64_t seed, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (rows <= 0 || C <= 0 || C > 4 || rate < 0.f || rate >= 1.f) return BTS_ERR_SHAPE;
  if (((uintptr_t)dst) & 15) return BTS_ERR_ALIGN;
  long blocks = (rows + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  const float scale = 1.0f / (1.0f - rate);
  (void)hipGetLastError();
#define LP_DCP(TT, C_) hipLaunchKernelGGL((lp_dropout_cast_pad16_kernel<TT, C_>), dim3((unsigned)blocks), dim3(256), 0, stream, src, (unsigned short*)dst, rows, rate, scale, seed)
  if (dtype == LP_F16) { if (C == 1) LP_DCP(TF16, 1); else if (C == 2) LP_DCP(TF16, 2); else if (C == 3) LP_DCP(TF16, 3); else LP_DCP(TF16, 4); }
  else { if (C == 1) LP_DCP(TBF16, 1); else if (C == 2) LP_DCP(TBF16, 2); else if (C == 3) LP_DCP(TBF16, 3); else LP_DCP(TBF16, 4); }
#undef LP_DCP
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
extern "C" int bts_lp_uncast(int dtype, const void* src, long ld_src, float* dst, long ld_dst, long rows, int C, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (rows <= 0 || C <= 0) return BTS_ERR_SHAPE;
  long blocks = (rows * C + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_uncast_kernel<TF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)src, ld_src, dst, ld_dst, rows, C);
  else hipLaunchKernelGGL(lp_uncast_kernel<TBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)src, ld_src, dst, ld_dst, rows, C);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- GroupNormalization statistics (group_norm.py:100-107): per (n, group) sum and sum of squares from the STORED values,
// fp64 partials per block, fixed-order combine (bts_gn_finalize_partials_).  Dense tensors (ld == C).
//   slab mode: group g of sample n = elements [g*L, (g+1)*L) of the sample's flattened (D,H,W,C) memory
//   channel mode: group g = channels [g*cg, (g+1)*cg)
template <typename T>
__global__ __launch_bounds__(256) void lp_gn_stats_kernel(const unsigned short* x, double* partial, long E, long L, int C, int G, int cg,
                                                          int mode, int B) {
  __shared__ double sh[8];
  const int unit = blockIdx.y;          // n*G + g
  const int n = unit / G, g = unit % G;
  const unsigned short* base = x + (long)n * E;
  double s = 0.0, q = 0.0;
  if (mode == BTS_GN_SLAB) {
    const long lo = (long)g * L;
    const long per = ((L / 8 + B - 1) / B) * 8;
    const long a = lo + (long)blockIdx.x * per, bnd = (a + per < lo + L) ? a + per : lo + L;
    float fs = 0.f, fq = 0.f;
    long cnt = 0;
    for (long i = a + threadIdx.x * 8L; i < bnd; i += 256 * 8) {
      float v[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(base + i), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { fs += v[e]; fq = fmaf(v[e], v[e], fq); }
      if (++cnt == 64) { s += fs; q += fq; fs = fq = 0.f; cnt = 0; }   // bounded fp32 run lengths (512 values per lane)
    }
    s += fs; q += fq;
  } else {
    const long V = E / C;
    const long per = (V + B - 1) / B;
    const long a = (long)blockIdx.x * per, bnd = (a + per < V) ? a + per : V;
    const int c0 = g * cg;
    for (long v = a + threadIdx.x; v < bnd; v += 256) {
      float fs = 0.f, fq = 0.f;
      for (int c = 0; c < cg; ++c) { const float t = T::ld(base[v * C + c0 + c]); fs += t; fq = fmaf(t, t, fq); }
      s += fs; q += fq;
    }
  }
  s = block_sum_f64(s, sh);
  q = block_sum_f64(q, sh + 4);
  if (threadIdx.x == 0) {
    double* o = partial + ((long)unit * B + blockIdx.x) * 2;
    o[0] = s; o[1] = q;
  }
}
static int lp_gn_blocks(long L) {
  long b = L / (256 * 8 * 8);
  if (b < 1) b = 1;
  if (b > 256) b = 256;
  return (int)b;
}
extern "C" long bts_lp_gn_workspace(int N, long V, int C, int G) {
  if (N <= 0 || V <= 0 || C <= 0 || G <= 0 || C % G != 0) return -1;
  return (long)N * G * lp_gn_blocks(V * C / G) * 2 * 8 + 64;
}
extern "C" int bts_lp_gn_stats(int dtype, const void* x, float* mean, float* rstd, void* workspace, long workspace_bytes, int N, long V,
                               int C, int G, int mode, float eps, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || C < G || C % G != 0) return BTS_ERR_SHAPE;
  const long E = V * C, L = E / G;
  if (mode == BTS_GN_SLAB && (E % G != 0 || L % 8 != 0)) return BTS_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)workspace)) & 15) return BTS_ERR_ALIGN;
  const int B = lp_gn_blocks(L);
  if (workspace_bytes < bts_lp_gn_workspace(N, V, C, G)) return BTS_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_gn_stats_kernel<TF16>, dim3(B, N * G), dim3(256), 0, stream, (const unsigned short*)x, partial, E, L, C, G, C / G, mode, B);
  else hipLaunchKernelGGL(lp_gn_stats_kernel<TBF16>, dim3(B, N * G), dim3(256), 0, stream, (const unsigned short*)x, partial, E, L, C, G, C / G, mode, B);
  BTS_LAUNCH_CHECK();
  return bts_gn_finalize_partials_(partial, mean, rstd, N * G, B, (double)L, eps, stream);
}

// ---- y = [relu]((x - mean) * rstd * gamma[idx] + beta[idx])  (group_norm.py:110-122); x dense, y rows of stride ldy
template <typename T>
__global__ __launch_bounds__(256) void lp_gn_apply_kernel(const unsigned short* x, unsigned short* y, const float* gamma, const float* beta,
                                                          const float* mean, const float* rstd, long total8, long E, long L, int C, int G,
                                                          int cg, int ldy, int mode, int relu) {
  for (long f = blockIdx.x * 256L + threadIdx.x; f < total8; f += (long)gridDim.x * 256) {
    const long i = f * 8;
    const long n = i / E;
    const long r = i - n * E;
    const int c = (int)(r % C);
    const long pix = i / C;
    float v[8], o[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(x + i), v);
    const int gsl = (int)(r / L);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (mode == BTS_GN_SLAB) ? gsl : (c + e) / cg;
      const int idx = (mode == BTS_GN_SLAB) ? g * cg + ((c + e) % cg) : (c + e);
      float t = (v[e] - mean[n * G + g]) * rstd[n * G + g] * gamma[idx] + beta[idx];
      if (relu) t = fmaxf(t, 0.f);
      o[e] = t;
    }
    *reinterpret_cast<u32x4*>(y + pix * ldy + c) = pack8<T>(o);
  }
}
// The same pass for tensors whose (sample, slab) units are whole multiples of 2048 elements with C | 2048 (every layer of the model):
// a workgroup streams one contiguous chunk of one unit, so a thread's eight channels -- and with them its statistics, scales and
// shifts -- are fixed before the loop: no division and no parameter load per 16 bytes, four loads in flight per thread.  (The
// grid-stride form above spent 4 64-bit divisions + 16 small ones per 16 bytes and streamed 3.1 TB/s; this one is bound by HBM.)
// Chunks of one unit: enough workgroups for ~16 per CU, at least 4 iterations each where the unit is that long.
static long lp_chunk_per(long Lu, long units, int* B, long target = 4096) {
  const long steps = Lu / 2048;                 // 2048-element steps of a unit
  long b = target / units;
  if (b > steps / 4) b = steps / 4;
  if (b < 1) b = 1;
  const long per = (steps + b - 1) / b;
  *B = (int)((steps + per - 1) / per);
  return per * 2048;
}
template <typename T, int RELU>
__global__ __launch_bounds__(256) void lp_gn_apply_chunk_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd, long Lu,
                                                                long per, int C, int G, int cg, int ldy, int slab, int upn) {
  const int unit = blockIdx.y, n = unit / upn, u = unit - n * upn;
  const long lo = (long)unit * Lu;
  const long a = lo + (long)blockIdx.x * per;
  const long bnd = (a + per < lo + Lu) ? a + per : lo + Lu;
  const int t8 = threadIdx.x * 8, c = t8 % C;
  float mu[8], sc[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = slab ? u : (c + e) / cg;
    const int idx = slab ? g * cg + ((c + e) % cg) : (c + e);
    mu[e] = mean[n * G + g];
    sc[e] = rstd[n * G + g] * gamma[idx];
    be[e] = beta[idx];
  }
  const long pstep = 2048 / C;
  long pix = a / C + t8 / C;
  const unsigned short* src = x + a + t8;
  const long K = (bnd - a) / 2048;
  auto one = [&](const u32x4 raw, long px) {
    float v[8], o[8];
    unpack8<T>(raw, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = fmaf(v[e] - mu[e], sc[e], be[e]);
      o[e] = RELU ? fmaxf(t, 0.f) : t;
    }
    *reinterpret_cast<u32x4*>(y + px * ldy + c) = pack8<T>(o);
  };
  long k = 0;
  for (; k + 4 <= K; k += 4) {
    u32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const u32x4*>(src + (k + j) * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) one(r[j], pix + (k + j) * pstep);
  }
  for (; k < K; ++k) one(*reinterpret_cast<const u32x4*>(src + k * 2048), pix + k * pstep);
}
extern "C" int bts_lp_gn_apply(int dtype, const void* x, void* y, const float* gamma, const float* beta, const float* mean,
                               const float* rstd, int N, long V, int C, int ldy, int G, int mode, int relu, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || C < G || C % G != 0 || C % 8 != 0 || ldy % 8 != 0 || ldy < C) return BTS_ERR_SHAPE;
  const long E = V * C, L = E / G;
  if (mode == BTS_GN_SLAB && L % 8 != 0) return BTS_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 15)) return BTS_ERR_ALIGN;
  (void)hipGetLastError();
  {
    const int slab = mode == BTS_GN_SLAB;
    const long Lu = slab ? L : E;
    if (2048 % C == 0 && Lu % 2048 == 0 && !getenv("BTS_LP_ELEM_OLD")) {
      const int upn = slab ? G : 1;
      int B;
      const long per = lp_chunk_per(Lu, (long)N * upn, &B);
      const dim3 grid((unsigned)B, (unsigned)(N * upn));
#define LP_GA(T_, R_) hipLaunchKernelGGL((lp_gn_apply_chunk_kernel<T_, R_>), grid, dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, gamma, beta, mean, rstd, Lu, per, C, G, C / G, ldy, slab, upn)
      if (dtype == LP_F16) { if (relu) LP_GA(TF16, 1); else LP_GA(TF16, 0); }
      else { if (relu) LP_GA(TBF16, 1); else LP_GA(TBF16, 0); }
#undef LP_GA
      BTS_LAUNCH_CHECK();
      return BTS_OK;
    }
  }
  const long total8 = (long)N * E / 8;
  long blocks = (total8 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_gn_apply_kernel<TF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, gamma, beta, mean, rstd, total8, E, L, C, G, C / G, ldy, mode, relu);
  else hipLaunchKernelGGL(lp_gn_apply_kernel<TBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, gamma, beta, mean, rstd, total8, E, L, C, G, C / G, ldy, mode, relu);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- global average pool of the shortcut (resnet.py:45-46,121): per (n, channel) mean over the voxels, fp64 partials
template <typename T>
__global__ __launch_bounds__(256) void lp_colsum_kernel(const unsigned short* x, double* partial, long V, int C, int B) {
  // block (b, n): voxels [b*per, ...) ; thread t handles channel octet (t % (C/8)) of every (256 / (C/8))-th voxel
  const int n = blockIdx.y;
  const int oct = C / 8;
  const int co = threadIdx.x % oct, vr = threadIdx.x / oct, vstep = 256 / oct;
  const long per = (V + B - 1) / B;
  const long a = (long)blockIdx.x * per, bnd = (a + per < V) ? a + per : V;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float fs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int cnt = 0;
  if (vr < vstep) {
    for (long v = a + vr; v < bnd; v += vstep) {
      float t[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(x + ((long)n * V + v) * C + co * 8), t);
#pragma unroll
      for (int e = 0; e < 8; ++e) fs[e] += t[e];
      if (++cnt == 256) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += fs[e]; fs[e] = 0.f; }
        cnt = 0;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] += fs[e];
  // combine the vstep voxel-rows through LDS in fixed order
  __shared__ double sh[256 * 8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = (vr < vstep) ? s[e] : 0.0;
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x, o = c / 8, e = c % 8;
    double tot = 0.0;
    for (int r = 0; r < vstep; ++r) tot += sh[(r * oct + o) * 8 + e];
    partial[((long)n * B + blockIdx.x) * C + c] = tot;
  }
}
__global__ __launch_bounds__(256) void lp_colsum_finalize_kernel(const double* partial, float* out, int N, int C, int B, double scale) {
  // one wave per (n, c): lanes split the blocks, fixed-order shuffle tree
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  double s = 0.0;
  for (int b = lane; b < B; b += 64) s += partial[((long)n * B + b) * C + c];
  s = wave_sum_f64(s);
  if (lane == 0) out[i] = (float)(s * scale);
}
static int lp_colsum_blocks(long V) {
  long b = V / 2048;
  if (b < 1) b = 1;
  if (b > 512) b = 512;
  return (int)b;
}
extern "C" long bts_lp_colsum_workspace(int N, long V, int C) {
  if (N <= 0 || V <= 0 || C <= 0) return -1;
  return (long)N * lp_colsum_blocks(V) * C * 8 + 64;
}
extern "C" int bts_lp_colsum(int dtype, const void* x, float* out, void* workspace, long workspace_bytes, int N, long V, int C, float scale,
                             hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || C % 8 != 0 || C > 256 || 256 % (C / 8) != 0) return BTS_ERR_SHAPE;
  if (((uintptr_t)x) & 15) return BTS_ERR_ALIGN;
  const int B = lp_colsum_blocks(V);
  if (workspace_bytes < bts_lp_colsum_workspace(N, V, C)) return BTS_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_colsum_kernel<TF16>, dim3(B, N), dim3(256), 0, stream, (const unsigned short*)x, partial, V, C, B);
  else hipLaunchKernelGGL(lp_colsum_kernel<TBF16>, dim3(B, N), dim3(256), 0, stream, (const unsigned short*)x, partial, V, C, B);
  BTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(lp_colsum_finalize_kernel, dim3((N * C + 3) / 4), dim3(256), 0, stream, partial, out, N, C, B, (double)scale);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- ResNet block epilogue (resnet.py:127-137): out = res * (sigmoid(res . w_sp) + ch[n]) + relu(GN2(c2))
// one voxel per group of C/8 lanes: the spatial dot product is reduced across those lanes with xor shuffles
template <typename T>
__global__ __launch_bounds__(256) void lp_block_epilogue_kernel(const unsigned short* res, const unsigned short* c2, unsigned short* out,
                                                                float* sp_out, const float* wsp, const float* ch, const float* gamma, const float* beta,
                                                                const float* mean, const float* rstd, long total8, long E, long L, int C,
                                                                int G, int cg, int ldo, int mode) {
  const int oct = C / 8;
  for (long f = blockIdx.x * 256L + threadIdx.x; f < total8; f += (long)gridDim.x * 256) {
    const long i = f * 8;
    const long n = i / E;
    const long r = i - n * E;
    const int c = (int)(r % C);
    const long pix = i / C;
    float a[8], b[8], o[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(res + i), a);
    unpack8<T>(*reinterpret_cast<const u32x4*>(c2 + i), b);
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot = fmaf(a[e], wsp[c + e], dot);
    dot = lane_group_sum_f32(dot, oct);   // (oct is a power of two <= 32; lanes of a voxel are adjacent)
    const float sp = 1.f / (1.f + __expf(-dot));
    if (sp_out != nullptr && c == 0) sp_out[pix] = sp;   // the spatial gate, kept for the backward pass (resnet.py:127)
    const int gsl = (int)(r / L);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (mode == BTS_GN_SLAB) ? gsl : (c + e) / cg;
      const int idx = (mode == BTS_GN_SLAB) ? g * cg + ((c + e) % cg) : (c + e);
      const float t = fmaxf((b[e] - mean[n * G + g]) * rstd[n * G + g] * gamma[idx] + beta[idx], 0.f);
      o[e] = fmaf(a[e], sp + ch[n * C + c + e], t);
    }
    *reinterpret_cast<u32x4*>(out + pix * ldo + c) = pack8<T>(o);
  }
}
// chunked form (see lp_gn_apply_chunk_kernel): per-thread channels fixed, gate weights / statistics / scales in registers
template <typename T>
__global__ __launch_bounds__(256) void lp_block_epilogue_chunk_kernel(const unsigned short* __restrict__ res, const unsigned short* __restrict__ c2,
                                                                      unsigned short* __restrict__ out, float* __restrict__ sp_out,
                                                                      const float* __restrict__ wsp, const float* __restrict__ ch,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const float* __restrict__ mean, const float* __restrict__ rstd, long Lu,
                                                                      long per, int C, int G, int cg, int ldo, int slab, int upn) {
  const int unit = blockIdx.y, n = unit / upn, u = unit - n * upn;
  const long lo = (long)unit * Lu;
  const long a = lo + (long)blockIdx.x * per;
  const long bnd = (a + per < lo + Lu) ? a + per : lo + Lu;
  const int t8 = threadIdx.x * 8, c = t8 % C, oct = C / 8;
  float mu[8], sc[8], be[8], ws[8], cw[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = slab ? u : (c + e) / cg;
    const int idx = slab ? g * cg + ((c + e) % cg) : (c + e);
    mu[e] = mean[n * G + g];
    sc[e] = rstd[n * G + g] * gamma[idx];
    be[e] = beta[idx];
    ws[e] = wsp[c + e];
    cw[e] = ch[n * C + c + e];
  }
  const long pstep = 2048 / C;
  const long pix = a / C + t8 / C;
  const unsigned short* ra = res + a + t8;
  const unsigned short* rb = c2 + a + t8;
  const long K = (bnd - a) / 2048;
  auto one = [&](const u32x4 r0, const u32x4 r1, long px) {
    float va[8], vb[8], o[8];
    unpack8<T>(r0, va);
    unpack8<T>(r1, vb);
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot = fmaf(va[e], ws[e], dot);
    dot = lane_group_sum_f32(dot, oct);
    const float sp = 1.f / (1.f + __expf(-dot));
    if (sp_out != nullptr && c == 0) sp_out[px] = sp;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = fmaxf(fmaf(vb[e] - mu[e], sc[e], be[e]), 0.f);
      o[e] = fmaf(va[e], sp + cw[e], t);
    }
    *reinterpret_cast<u32x4*>(out + px * ldo + c) = pack8<T>(o);
  };
  long k = 0;
  for (; k + 2 <= K; k += 2) {
    u32x4 r0[2], r1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      r0[j] = *reinterpret_cast<const u32x4*>(ra + (k + j) * 2048);
      r1[j] = *reinterpret_cast<const u32x4*>(rb + (k + j) * 2048);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) one(r0[j], r1[j], pix + (k + j) * pstep);
  }
  for (; k < K; ++k) one(*reinterpret_cast<const u32x4*>(ra + k * 2048), *reinterpret_cast<const u32x4*>(rb + k * 2048), pix + k * pstep);
}
extern "C" int bts_lp_block_epilogue(int dtype, const void* res, const void* c2, void* out, float* sp_out, const float* wsp, const float* ch,
                                     const float* gamma, const float* beta, const float* mean, const float* rstd, int N, long V, int C,
                                     int ldo, int G, int mode, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || C % 8 != 0 || C > 256 || ((C / 8) & (C / 8 - 1)) != 0 || C % G != 0 || ldo % 8 != 0 || ldo < C) return BTS_ERR_SHAPE;
  const long E = V * C, L = E / G;
  if (mode == BTS_GN_SLAB && L % 8 != 0) return BTS_ERR_UNSUPPORTED;
  if ((((uintptr_t)res) & 15) || (((uintptr_t)c2) & 15) || (((uintptr_t)out) & 15)) return BTS_ERR_ALIGN;
  (void)hipGetLastError();
  {
    const int slab = mode == BTS_GN_SLAB;
    const long Lu = slab ? L : E;
    if (2048 % C == 0 && Lu % 2048 == 0 && !getenv("BTS_LP_ELEM_OLD")) {
      const int upn = slab ? G : 1;
      int B;
      const long per = lp_chunk_per(Lu, (long)N * upn, &B);
      const dim3 grid((unsigned)B, (unsigned)(N * upn));
      if (dtype == LP_F16) hipLaunchKernelGGL(lp_block_epilogue_chunk_kernel<TF16>, grid, dim3(256), 0, stream, (const unsigned short*)res, (const unsigned short*)c2, (unsigned short*)out, sp_out, wsp, ch, gamma, beta, mean, rstd, Lu, per, C, G, C / G, ldo, slab, upn);
      else hipLaunchKernelGGL(lp_block_epilogue_chunk_kernel<TBF16>, grid, dim3(256), 0, stream, (const unsigned short*)res, (const unsigned short*)c2, (unsigned short*)out, sp_out, wsp, ch, gamma, beta, mean, rstd, Lu, per, C, G, C / G, ldo, slab, upn);
      BTS_LAUNCH_CHECK();
      return BTS_OK;
    }
  }
  const long total8 = (long)N * E / 8;
  // (the C/8 lanes of a voxel are adjacent and aligned inside a wave, so they enter and leave the loop together)
  long blocks = (total8 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_block_epilogue_kernel<TF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)res, (const unsigned short*)c2, (unsigned short*)out, sp_out, wsp, ch, gamma, beta, mean, rstd, total8, E, L, C, G, C / G, ldo, mode);
  else hipLaunchKernelGGL(lp_block_epilogue_kernel<TBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)res, (const unsigned short*)c2, (unsigned short*)out, sp_out, wsp, ch, gamma, beta, mean, rstd, total8, E, L, C, G, C / G, ldo, mode);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- the LAST block's epilogue with the output head in it (inference: decoder.py:55-63 after the top decoder block, model.py:63-68) ----
// y_head[v][k] = sigmoid( sum_c out[v][c] * W[c][k] + b[k] ),  out = res * (sigmoid(res . w_sp) + ch) + relu(GN2(c2)) as above -- `out` has ONE
// reader in a forward without a backward, the 1x1x1 head (32 -> 3 at the CLI defaults), so it is never written: the separate pair costs a
// write and a read of the 32-channel tensor plus a launch (160 x 192 x 160: 177 + 101 us; fused: one pass over res and c2).  The head sees the
// fp32 values of `out`, not their 16-bit roundings.  Chunked form only (its conditions as in bts_lp_block_epilogue); K <= 4.
template <typename T, int K>
__global__ __launch_bounds__(256) void lp_block_epilogue_head_kernel(const unsigned short* __restrict__ res, const unsigned short* __restrict__ c2,
                                                                     float* __restrict__ yh, const float* __restrict__ wsp, const float* __restrict__ ch,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* __restrict__ hw, const float* __restrict__ hb, long Lu, long per,
                                                                     int C, int G, int cg, int slab, int upn, int sigmoid) {
  const int unit = blockIdx.y, n = unit / upn, u = unit - n * upn;
  const long lo = (long)unit * Lu;
  const long a = lo + (long)blockIdx.x * per;
  const long bnd = (a + per < lo + Lu) ? a + per : lo + Lu;
  const int t8 = threadIdx.x * 8, c = t8 % C, oct = C / 8;
  float mu[8], sc[8], be[8], ws[8], cw[8], wr[8][K], hbk[K];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = slab ? u : (c + e) / cg;
    const int idx = slab ? g * cg + ((c + e) % cg) : (c + e);
    mu[e] = mean[n * G + g];
    sc[e] = rstd[n * G + g] * gamma[idx];
    be[e] = beta[idx];
    ws[e] = wsp[c + e];
    cw[e] = ch[n * C + c + e];
#pragma unroll
    for (int k = 0; k < K; ++k) wr[e][k] = hw[(c + e) * K + k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) hbk[k] = hb ? hb[k] : 0.f;
  const long pstep = 2048 / C;
  const long pix = a / C + t8 / C;
  const unsigned short* ra = res + a + t8;
  const unsigned short* rb = c2 + a + t8;
  const long Kc = (bnd - a) / 2048;
  auto one = [&](const u32x4 r0, const u32x4 r1, long px) {
    float va[8], vb[8];
    unpack8<T>(r0, va);
    unpack8<T>(r1, vb);
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot = fmaf(va[e], ws[e], dot);
    dot = lane_group_sum_f32(dot, oct);
    const float sp = 1.f / (1.f + __expf(-dot));
    float hk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) hk[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = fmaxf(fmaf(vb[e] - mu[e], sc[e], be[e]), 0.f);
      const float o = fmaf(va[e], sp + cw[e], t);
#pragma unroll
      for (int k = 0; k < K; ++k) hk[k] = fmaf(o, wr[e][k], hk[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      hk[k] = lane_group_sum_f32(hk[k], oct);
    if (c == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float r = hk[k] + hbk[k];
        if (sigmoid) r = 1.f / (1.f + __expf(-r));
        yh[px * K + k] = r;
      }
    }
  };
  long k = 0;
  for (; k + 2 <= Kc; k += 2) {
    u32x4 r0[2], r1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      r0[j] = *reinterpret_cast<const u32x4*>(ra + (k + j) * 2048);
      r1[j] = *reinterpret_cast<const u32x4*>(rb + (k + j) * 2048);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) one(r0[j], r1[j], pix + (k + j) * pstep);
  }
  for (; k < Kc; ++k) one(*reinterpret_cast<const u32x4*>(ra + k * 2048), *reinterpret_cast<const u32x4*>(rb + k * 2048), pix + k * pstep);
}
extern "C" int bts_lp_block_epilogue_head(int dtype, const void* res, const void* c2, float* y_head, const float* wsp, const float* ch,
                                          const float* gamma, const float* beta, const float* mean, const float* rstd, const float* head_w,
                                          const float* head_b, int N, long V, int C, int G, int mode, int K, int sigmoid, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || C % 8 != 0 || C > 256 || ((C / 8) & (C / 8 - 1)) != 0 || C % G != 0 || K < 1) return BTS_ERR_SHAPE;
  const long E = V * C, L = E / G;
  const int slab = mode == BTS_GN_SLAB;
  const long Lu = slab ? L : E;
  // (per-thread state: 8 x K head weights on top of the epilogue's 40 registers -- wide blocks and heads stay on the two-kernel route)
  if (K > 4 || C > 64 || (slab && L % 8 != 0) || 2048 % C != 0 || Lu % 2048 != 0) return BTS_ERR_UNSUPPORTED;
  if ((((uintptr_t)res) & 15) || (((uintptr_t)c2) & 15) || (((uintptr_t)y_head) & 3)) return BTS_ERR_ALIGN;
  const int upn = slab ? G : 1;
  int B;
  const long per = lp_chunk_per(Lu, (long)N * upn, &B);
  const dim3 grid((unsigned)B, (unsigned)(N * upn));
  (void)hipGetLastError();
#define LP_EH(TT, K_) hipLaunchKernelGGL((lp_block_epilogue_head_kernel<TT, K_>), grid, dim3(256), 0, stream, (const unsigned short*)res, (const unsigned short*)c2, y_head, wsp, ch, gamma, beta, mean, rstd, head_w, head_b, Lu, per, C, G, C / G, slab, upn, sigmoid)
#define LP_EH_K(TT) do { if (K == 1) LP_EH(TT, 1); else if (K == 2) LP_EH(TT, 2); else if (K == 3) LP_EH(TT, 3); else LP_EH(TT, 4); } while (0)
  if (dtype == LP_F16) LP_EH_K(TF16); else LP_EH_K(TBF16);
#undef LP_EH_K
#undef LP_EH
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- the non-default samplers on 16-bit tensors (args.py:136-141): MaxPooling3D(2) (downsample.py:51-70) and
// UpSampling3D(2) = nearest-neighbour repeat (upsample.py:69), with their gradients (train.py:142-151 under TF autodiff).  Eight
// channels (one 16-byte piece) per thread; the pool keeps the window position of the FIRST maximum (scan order dz, dy, dx) like the
// fp32 engine's bts_maxpool2_fwd, so that the gradient routing is the same; sums of the repeat's gradient run in fp32.
template <typename T>
__global__ __launch_bounds__(256) void lp_maxpool2_fwd_kernel(const unsigned short* x, unsigned short* y, unsigned char* idx, long total8, int Do,
                                                              int Ho, int Wo, int C, int ldx, int ldy) {
  const int C8 = C >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    long v = i / C8;
    const int ox = (int)(v % Wo); v /= Wo;
    const int oy = (int)(v % Ho); v /= Ho;
    const int oz = (int)(v % Do);
    const long n = v / Do;
    float best[8];
    unsigned char bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const long src = (((n * (2 * Do) + 2 * oz + (t >> 2)) * (2 * Ho) + 2 * oy + ((t >> 1) & 1)) * (2L * Wo) + 2 * ox + (t & 1));
      float f[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(x + src * ldx + c), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) if (f[e] > best[e]) { best[e] = f[e]; bi[e] = (unsigned char)t; }
    }
    const long dst = ((n * Do + oz) * Ho + oy) * (long)Wo + ox;
    *reinterpret_cast<u32x4*>(y + dst * ldy + c) = pack8<T>(best);
    if (idx != nullptr) {
      unsigned lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), hi = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *reinterpret_cast<u32x2*>(idx + dst * C + c) = u32x2{lo, hi};
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void lp_maxpool2_bwd_kernel(const unsigned short* dy, const unsigned char* idx, unsigned short* dx, long total8,
                                                              int Do, int Ho, int Wo, int C, int lddy, int lddx, int accum) {
  const int C8 = C >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    long v = i / C8;
    const int ox = (int)(v % Wo); v /= Wo;
    const int oy = (int)(v % Ho); v /= Ho;
    const int oz = (int)(v % Do);
    const long n = v / Do;
    const long src = ((n * Do + oz) * Ho + oy) * (long)Wo + ox;
    float g[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(dy + src * lddy + c), g);
    const u32x2 pk = *reinterpret_cast<const u32x2*>(idx + src * C + c);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const long dst = (((n * (2 * Do) + 2 * oz + (t >> 2)) * (2 * Ho) + 2 * oy + ((t >> 1) & 1)) * (2L * Wo) + 2 * ox + (t & 1));
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (((e < 4 ? pk[0] >> (8 * e) : pk[1] >> (8 * (e - 4))) & 0xffu) == (unsigned)t) ? g[e] : 0.f;
      if (accum) {
        float old[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(dx + dst * lddx + c), old);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += old[e];
      }
      *reinterpret_cast<u32x4*>(dx + dst * lddx + c) = pack8<T>(o);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void lp_upsample2_fwd_kernel(const unsigned short* x, unsigned short* y, long total8, int D, int H, int W, int C,
                                                               int ldx, int ldy) {
  const int C8 = C >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {      // over the FINE grid
    const int c = (int)(i % C8) * 8;
    long v = i / C8;
    const int fx = (int)(v % (2 * W)); v /= 2 * W;
    const int fy = (int)(v % (2 * H)); v /= 2 * H;
    const int fz = (int)(v % (2 * D));
    const long n = v / (2 * D);
    const long src = ((n * D + (fz >> 1)) * H + (fy >> 1)) * (long)W + (fx >> 1);
    const long dst = ((n * (2 * D) + fz) * (2 * H) + fy) * (2L * W) + fx;
    *reinterpret_cast<u32x4*>(y + dst * ldy + c) = *reinterpret_cast<const u32x4*>(x + src * ldx + c);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void lp_upsample2_bwd_kernel(const unsigned short* dy, unsigned short* dx, long total8, int D, int H, int W, int C,
                                                               int lddy, int lddx, int accum) {
  const int C8 = C >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {      // over the COARSE grid
    const int c = (int)(i % C8) * 8;
    long v = i / C8;
    const int ox = (int)(v % W); v /= W;
    const int oy = (int)(v % H); v /= H;
    const int oz = (int)(v % D);
    const long n = v / D;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const long src = (((n * (2 * D) + 2 * oz + (t >> 2)) * (2 * H) + 2 * oy + ((t >> 1) & 1)) * (2L * W) + 2 * ox + (t & 1));
      float f[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(dy + src * lddy + c), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
    const long dst = ((n * D + oz) * H + oy) * (long)W + ox;
    if (accum) {
      float old[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(dx + dst * lddx + c), old);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += old[e];
    }
    *reinterpret_cast<u32x4*>(dx + dst * lddx + c) = pack8<T>(s);
  }
}
static int lp_sampler_check(int dtype, const void* a, const void* b, int N, int D, int H, int W, int C, int lda, int ldb) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 || lda < C || ldb < C) return BTS_ERR_SHAPE;
  if ((((uintptr_t)a) & 15) || (((uintptr_t)b) & 15)) return BTS_ERR_ALIGN;
  return BTS_OK;
}
static unsigned lp_sampler_blocks(long total8) { long b = (total8 + 255) / 256; return (unsigned)(b > 16384 ? 16384 : b); }
// (D,H,W) = INPUT dims (even); y (N,D/2,H/2,W/2,C); idx (may be NULL: inference) dense uint8 (N,D/2,H/2,W/2,C)
extern "C" int bts_lp_maxpool2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int D, int H, int W, int C, int ldx, int ldy,
                                   hipStream_t stream) {
  const int r = lp_sampler_check(dtype, x, y, N, D, H, W, C, ldx, ldy);
  if (r != BTS_OK) return r;
  if ((D | H | W) & 1) return BTS_ERR_SHAPE;
  const long total8 = (long)N * (D / 2) * (H / 2) * (W / 2) * (C / 8);
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_maxpool2_fwd_kernel<TF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, idx, total8, D / 2, H / 2, W / 2, C, ldx, ldy);
  else hipLaunchKernelGGL(lp_maxpool2_fwd_kernel<TBF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, idx, total8, D / 2, H / 2, W / 2, C, ldx, ldy);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
extern "C" int bts_lp_maxpool2_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int D, int H, int W, int C, int lddy, int lddx,
                                   int accumulate, hipStream_t stream) {
  const int r = lp_sampler_check(dtype, dy, dx, N, D, H, W, C, lddy, lddx);
  if (r != BTS_OK) return r;
  if (((D | H | W) & 1) || idx == nullptr) return BTS_ERR_SHAPE;
  const long total8 = (long)N * (D / 2) * (H / 2) * (W / 2) * (C / 8);
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_maxpool2_bwd_kernel<TF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)dy, idx, (unsigned short*)dx, total8, D / 2, H / 2, W / 2, C, lddy, lddx, accumulate);
  else hipLaunchKernelGGL(lp_maxpool2_bwd_kernel<TBF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)dy, idx, (unsigned short*)dx, total8, D / 2, H / 2, W / 2, C, lddy, lddx, accumulate);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
// (D,H,W) = COARSE dims in both calls
extern "C" int bts_lp_upsample2_fwd(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, int ldx, int ldy, hipStream_t stream) {
  const int r = lp_sampler_check(dtype, x, y, N, D, H, W, C, ldx, ldy);
  if (r != BTS_OK) return r;
  const long total8 = 8L * N * D * H * W * (C / 8);
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_upsample2_fwd_kernel<TF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, total8, D, H, W, C, ldx, ldy);
  else hipLaunchKernelGGL(lp_upsample2_fwd_kernel<TBF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, total8, D, H, W, C, ldx, ldy);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
extern "C" int bts_lp_upsample2_bwd(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, int lddy, int lddx, int accumulate,
                                    hipStream_t stream) {
  const int r = lp_sampler_check(dtype, dy, dx, N, D, H, W, C, lddy, lddx);
  if (r != BTS_OK) return r;
  const long total8 = (long)N * D * H * W * (C / 8);
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_upsample2_bwd_kernel<TF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)dy, (unsigned short*)dx, total8, D, H, W, C, lddy, lddx, accumulate);
  else hipLaunchKernelGGL(lp_upsample2_bwd_kernel<TBF16>, dim3(lp_sampler_blocks(total8)), dim3(256), 0, stream, (const unsigned short*)dy, (unsigned short*)dx, total8, D, H, W, C, lddy, lddx, accumulate);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

// ---- output head (decoder.py:55-63): y = sigmoid(x . W + b), 1x1x1 conv to out_ch <= 4 channels, fp32 result (the label map
// is taken from it, so it is never rounded to 16 bits).  One voxel per lane, weights in registers via scalar loads.
template <typename T>
__global__ __launch_bounds__(256) void lp_head_kernel(const unsigned short* x, const float* w, const float* bias, float* y, long nvox, int C,
                                                      int ldx, int K, int sigmoid) {
  for (long v = blockIdx.x * 256L + threadIdx.x; v < nvox; v += (long)gridDim.x * 256) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += 8) {
      float t[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(x + v * ldx + c0), t);
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < K) acc[k] = fmaf(t[e], w[(c0 + e) * K + k], acc[k]);
    }
    for (int k = 0; k < K; ++k) {
      float o = acc[k] + (bias ? bias[k] : 0.f);
      if (sigmoid) o = 1.f / (1.f + __expf(-o));
      y[v * K + k] = o;
    }
  }
}
// The same with the C/8 lanes of a voxel side by side (C/8 a power of two <= 32): every load instruction of a wave reads 1 KB of
// consecutive bytes (one voxel per lane made each of its four loads touch 64 different lines: 0.65 ms for the 128^3 x 8 head, 2 TB/s);
// the lanes of a voxel add their partial dot products with xor-shuffles, lane 0 of the group writes.
template <typename T, int K>
__global__ __launch_bounds__(256) void lp_head_oct_kernel(const unsigned short* x, const float* w, const float* bias, float* y, long nvox, int C8,
                                                          int ldx, int sigmoid) {
  const int o = threadIdx.x % C8;
  float wr[8][K];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < K; ++k) wr[e][k] = w[(o * 8 + e) * K + k];
  const long total = nvox * C8;
  const long stride = (long)gridDim.x * 256;
  // (the host launches this kernel with 256 % C8 == 0: a thread's voxel advances by whole steps, no division per 16 bytes)
  const long vpb = 256 / C8, vstride = (long)gridDim.x * vpb;
  long v = blockIdx.x * vpb + threadIdx.x / C8;
  auto one = [&](const u32x4 raw, long vv, bool live) {
    float t[8], acc[K];
    unpack8<T>(raw, t);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(t[e], wr[e][k], s);
      s = lane_group_sum_f32(s, C8);
      acc[k] = s;
    }
    if (live && o == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float r = acc[k] + (bias ? bias[k] : 0.f);
        if (sigmoid) r = 1.f / (1.f + __expf(-r));
        y[vv * K + k] = r;
      }
    }
  };
  // (whole waves stay in the loop: the shuffles need every lane.)  Four steps per trip, their loads issued first.
  for (long i0 = blockIdx.x * 256L; i0 < total; i0 += 4 * stride, v += 4 * vstride) {
    u32x4 raw[4];
    bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long vv = v + j * vstride;
      live[j] = vv < nvox;
      raw[j] = live[j] ? *reinterpret_cast<const u32x4*>(x + vv * ldx + o * 8) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) one(raw[j], v + j * vstride, live[j]);
  }
}
extern "C" int bts_lp_head(int dtype, const void* x, const float* w, const float* bias, float* y, long nvox, int C, int ldx, int K, int sigmoid,
                           hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (nvox <= 0 || C % 8 != 0 || K < 1 || K > 4 || ldx % 8 != 0) return BTS_ERR_SHAPE;
  if (((uintptr_t)x) & 15) return BTS_ERR_ALIGN;
  long blocks = (nvox + 255) / 256;
  if (blocks > 32768) blocks = 32768;
  (void)hipGetLastError();
  const int C8 = C / 8;
  if (C8 >= 2 && C8 <= 32 && (C8 & (C8 - 1)) == 0 && nvox >= 4096) {
    long ob = ((nvox * C8 + 255) / 256 + 3) / 4;      // (four steps per trip of the kernel's loop)
    if (ob > 32768) ob = 32768;
    if (ob < 1) ob = 1;
#define LP_HO(TT, K_) hipLaunchKernelGGL((lp_head_oct_kernel<TT, K_>), dim3((unsigned)ob), dim3(256), 0, stream, (const unsigned short*)x, w, bias, y, nvox, C8, ldx, sigmoid)
#define LP_HO_K(TT) do { if (K == 1) LP_HO(TT, 1); else if (K == 2) LP_HO(TT, 2); else if (K == 3) LP_HO(TT, 3); else LP_HO(TT, 4); } while (0)
    if (dtype == LP_F16) LP_HO_K(TF16); else LP_HO_K(TBF16);
#undef LP_HO_K
#undef LP_HO
    BTS_LAUNCH_CHECK();
    return BTS_OK;
  }
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_head_kernel<TF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)x, w, bias, y, nvox, C, ldx, K, sigmoid);
  else hipLaunchKernelGGL(lp_head_kernel<TBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)x, w, bias, y, nvox, C, ldx, K, sigmoid);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}

__global__ __launch_bounds__(256) void lp_dbias_finalize_kernel(const double* part, float* db, int nblocks, int C, int accum);
// ---- output head backward (decoder.py:55-63 under TF autodiff, train.py:142-151): with dpre = dL/d(x . W + b) per voxel (K <= 4 values,
// the sigmoid's gradient already applied: bts_sigmoid_bwd), ONE pass over the 16-bit activations gives all three gradients:
//   dx[v][c] = sum_k dpre[v][k] W[c][k]  (storage type),   dW[c][k] (+)= sum_v x[v][c] dpre[v][k],   db[k] (+)= sum_v dpre[v][k].
// Round 2 widened x to fp32, ran the fp32 1x1x1 weight- and data-gradient kernels and narrowed dx again: five passes over a 128^3 x 32
// tensor.  One voxel per lane; a lane keeps the C*K + K running sums of its voxels, waves reduce by shuffles, blocks through LDS, one
// fp64 row per block; lp_dbias_finalize_kernel adds the rows in block order.
template <typename T, int C, int K>
__global__ __launch_bounds__(256) void lp_head_bwd_kernel(const unsigned short* x, const float* dpre, const float* w, unsigned short* dx, double* part,
                                                          long nvox, int ldx, int lddx) {
  // lane = (voxel, octet of its C channels): coalesced 16-byte loads / stores; a lane keeps the 8*K sums of its octet (+ K bias sums on
  // octet 0); lanes of equal octet meet by shuffles (stride C/8), waves through LDS, one fp64 row [C*K + K] per block
  constexpr int C8 = C / 8, NS = C * K + K;
  __shared__ float sh[4][NS];
  const int o = threadIdx.x % C8;
  float wr[8][K], acc[8][K], accb[K];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < K; ++k) { wr[e][k] = w[(o * 8 + e) * K + k]; acc[e][k] = 0.f; }
#pragma unroll
  for (int k = 0; k < K; ++k) accb[k] = 0.f;
  const long total = nvox * C8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {      // (the stride is a multiple of C8: o is fixed)
    const long v = i / C8;
    float d[K], t[8], ov[8];
#pragma unroll
    for (int k = 0; k < K; ++k) d[k] = dpre[v * K + k];
    if (o == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) accb[k] += d[k];
    }
    unpack8<T>(*reinterpret_cast<const u32x4*>(x + v * ldx + o * 8), t);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        s = fmaf(d[k], wr[e][k], s);
        acc[e][k] = fmaf(t[e], d[k], acc[e][k]);
      }
      ov[e] = s;
    }
    *reinterpret_cast<u32x4*>(dx + v * lddx + o * 8) = pack8<T>(ov);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float s = acc[e][k];
      for (int m = C8; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
      if (lane < C8) sh[wave][(lane * 8 + e) * K + k] = s;
    }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float s = accb[k];
    for (int m = C8; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) sh[wave][C * K + k] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NS; i += 256) part[(long)blockIdx.x * NS + i] = ((double)sh[0][i] + (double)sh[1][i]) + ((double)sh[2][i] + (double)sh[3][i]);
}
extern "C" long bts_lp_head_bwd_workspace(int C, int K) { return (C <= 0 || K <= 0) ? -1 : 2048L * (C * K + K) * 8 + 64; }
// x (nvox, C) 16-bit rows of ldx; dpre (nvox, K) fp32 dense; w (C, K) fp32; dx (nvox, C) 16-bit rows of lddx; dw (C, K), db (K) fp32 (+= when accumulate).
// C in {16, 32, 64}, K in {1, 2, 3, 4}: the heads of this model (decoder.py:55-63: C = base_filters, K = out_ch)
extern "C" int bts_lp_head_bwd(int dtype, const void* x, const float* dpre, const float* w, void* dx, float* dw, float* db, void* workspace,
                               long workspace_bytes, long nvox, int C, int ldx, int lddx, int K, int accumulate, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (nvox <= 0 || (C != 16 && C != 32 && C != 64) || K < 1 || K > 4) return BTS_ERR_UNSUPPORTED;
  if (ldx % 8 != 0 || lddx % 8 != 0 || ldx < C || lddx < C) return BTS_ERR_SHAPE;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dx) & 15) || (((uintptr_t)workspace) & 15)) return BTS_ERR_ALIGN;
  if (workspace_bytes < bts_lp_head_bwd_workspace(C, K)) return BTS_ERR_WORKSPACE;
  long blocks = (nvox * (C / 8) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  double* part = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError();
#define LP_HB(TT, C_, K_) hipLaunchKernelGGL((lp_head_bwd_kernel<TT, C_, K_>), dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)x, dpre, w, (unsigned short*)dx, part, nvox, ldx, lddx)
#define LP_HB_K(TT, C_) do { if (K == 1) LP_HB(TT, C_, 1); else if (K == 2) LP_HB(TT, C_, 2); else if (K == 3) LP_HB(TT, C_, 3); else LP_HB(TT, C_, 4); } while (0)
#define LP_HB_C(TT) do { if (C == 16) LP_HB_K(TT, 16); else if (C == 32) LP_HB_K(TT, 32); else LP_HB_K(TT, 64); } while (0)
  if (dtype == LP_F16) LP_HB_C(TF16); else LP_HB_C(TBF16);
#undef LP_HB_C
#undef LP_HB_K
#undef LP_HB
  BTS_LAUNCH_CHECK();
  const int NS = C * K + K;
  float* tmp = nullptr; (void)tmp;
  // rows -> dw (C*K values) and db (K values): two calls of the fixed-order row adder on the same partial rows
  hipLaunchKernelGGL(lp_dbias_finalize_kernel, dim3(C * K), dim3(256), 0, stream, part, dw, (int)blocks, NS, accumulate);
  BTS_LAUNCH_CHECK();
  if (db != nullptr) {
    hipLaunchKernelGGL(lp_dbias_finalize_kernel, dim3(K), dim3(256), 0, stream, part + C * K, db, (int)blocks, NS, accumulate);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}

// Bias gradient of the conv that produced the tensor a backward apply pass writes (its dy): db[c] = sum over voxels and samples.  The
// apply kernels walk octets with a stride that is a multiple of the channel count, so a thread's channel octet never changes: it keeps
// eight running sums, the block adds the threads of equal octet in LDS (fixed order) and leaves one fp64 partial row per block;
// lp_dbias_finalize_kernel adds the rows in block order.  (Round 2 read every such dy a second time: bts_lp_colsum, 4.5 ms per step.)
__device__ __forceinline__ void lp_dbias_block(const float (&cs)[8], int oct, int noct, double* part_row, float* sh /* [256][8] */) {
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = cs[e];
  __syncthreads();
  if ((int)threadIdx.x < noct * 8) {
    const int o = threadIdx.x >> 3, e = threadIdx.x & 7;
    double s = 0.0;
    for (int t = o; t < 256; t += noct) s += (double)sh[t * 8 + e];     // threads t = o (mod noct) hold octet o
    part_row[o * 8 + e] = s;
  }
  (void)oct;
}
// one workgroup per channel c: threads split the block rows, fixed-order combine (sh: 4 doubles of LDS)
__device__ __forceinline__ void lp_dbias_finalize_body(const double* part, float* db, int nblocks, int C, int accum, int c, double* sh) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(long)b * C + c];
  s = block_sum_f64(s, sh);
  if (threadIdx.x == 0) db[c] = accum ? db[c] + (float)s : (float)s;
}
__global__ __launch_bounds__(256) void lp_dbias_finalize_kernel(const double* part, float* db, int nblocks, int C, int accum) {
  __shared__ double sh[4];
  lp_dbias_finalize_body(part, db, nblocks, C, accum, blockIdx.x, sh);
}
// =====================================================================================================================
// GroupNormalization backward on 16-bit tensors (channels_last = slab semantics; group_norm.py:83-124 under TF autodiff).
// With xh = (x - mean) * rstd, y = gamma_i xh + beta_i (i = g*cg + c mod cg), dyE = dy * [y > 0] (fused ReLU):
//   A_j = sum dyE * xh, B_j = sum dyE per (n, g, j = c mod cg)   ->  dgamma_i (+)= sum_n A_j, dbeta_i (+)= sum_n B_j
//   c1 = sum_j gamma_j B_j / L,  c2 = sum_j gamma_j A_j / L       ->  dx = (dyE gamma_i - c1 - xh c2) * rstd
// Sums in fp32 runs flushed to fp64, fixed order; dx is written in the storage type (for the data-gradient convs) AND, when
// asked, in fp32 (the weight-gradient kernels still run on the fp32 matrix pipe): one pass instead of three.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void lp_gn_bwd_reduce_kernel(const unsigned short* x, const unsigned short* dy, const float* gamma,
                                                               const float* beta, const float* mean, const float* rstd, double* partial,
                                                               long E, long L, long span, int C, int G, int cg, int lddy, int relu) {
  __shared__ double sh[4 * 4 * 16];
  const int unit = blockIdx.y, n = unit / G, g = unit % G;
  const long lo = (long)blockIdx.x * span;
  long hi = lo + span;
  if (hi > L) hi = L;
  const long ubase = (long)g * L;
  const int cph = (int)((ubase + lo + threadIdx.x * 8L) % C);   // constant over the loop: the stride 2048 is a multiple of C
  float gam[8], bet[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int idx = g * cg + ((cph + e) % cg); gam[e] = gamma[idx]; bet[e] = beta[idx]; }
  const float m = mean[unit], rs = rstd[unit];
  double a[8], b[8];
  float fa[8], fb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = b[e] = 0.0; fa[e] = fb[e] = 0.f; }
  const unsigned short* xb = x + (long)n * E + ubase;
  const long pstep = 2048 / C, pix0 = ((long)n * E + ubase + lo + threadIdx.x * 8L) / C;
  int cnt = 0;
  for (long i = lo + threadIdx.x * 8L; i < hi; i += 2048) {
    float v[8], d[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(xb + i), v);
    const long pix = pix0 + (i - lo) / 2048 * pstep;      // (voxel of the possibly strided dy; 2048 / C voxels per step)
    unpack8<T>(*reinterpret_cast<const u32x4*>(dy + pix * lddy + cph), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - m) * rs;
      float de = d[e];
      if (relu && !(xh * gam[e] + bet[e] > 0.f)) de = 0.f;
      fa[e] = fmaf(de, xh, fa[e]);
      fb[e] += de;
    }
    if (++cnt == 32) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] += fa[e]; b[e] += fb[e]; fa[e] = fb[e] = 0.f; }
      cnt = 0;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] += fa[e]; b[e] += fb[e]; }
  // lanes whose 8 channels fall on the same classes: every p-th lane, p = cg / 8 (1 when cg <= 8)
  const int p = cg > 8 ? cg / 8 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 32; off >= p; off >>= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] += __shfl_xor(a[e], off, 64); b[e] += __shfl_xor(b[e], off, 64); }
  }
  if (lane < p) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[((wave * 4 + lane) * 8 + e) * 2] = a[e]; sh[((wave * 4 + lane) * 8 + e) * 2 + 1] = b[e]; }
  }
  __syncthreads();
  if (threadIdx.x < cg) {
    const int j = threadIdx.x;
    const int cph0 = (int)((ubase + lo) % C);          // lane q of any wave has channel phase cph0 + 8q (+ wave * 512: a multiple of C)
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < 4; ++w)
      for (int q = 0; q < p; ++q)
        for (int e = 0; e < 8; ++e)
          if (((cph0 + 8 * q + e) % cg) == j) { sa += sh[((w * 4 + q) * 8 + e) * 2]; sb += sh[((w * 4 + q) * 8 + e) * 2 + 1]; }
    double* o = partial + (((long)unit * gridDim.x + blockIdx.x) * cg + j) * 2;
    o[0] = sa; o[1] = sb;
  }
}
// one block per group g: for every sample the class sums over the blocks -> c1, c2; the sums over the samples -> dgamma, dbeta.
// Up to 256 / cg samples side by side (thread = (slice of the blocks, sample, class)): the loop over the samples of a batch of 8 was
// eight dependent rounds, 15 us on the critical path between the reduce and the apply pass of every GroupNorm backward.
__device__ __forceinline__ void lp_gn_bwd_finalize_body(const double* partial, const float* gamma, float* dgamma, float* dbeta, float* c1,
                                                        float* c2, int N, int G, int B, int cg, double L, int accum, int g, double* sh,
                                                        double* tot /* [sample slot][class] totals of the current round */) {
  const int t = threadIdx.x;
  int Np = 1;
  while (Np * 2 <= N && cg * Np * 2 <= 256) Np *= 2;
  const int S = 256 / (cg * Np);
  const int j = t % cg, nl = (t / cg) % Np, sl = t / (cg * Np);
  double ga = 0.0, gb = 0.0;     // running dgamma / dbeta of class t (threads t < cg)
  for (int n0 = 0; n0 < N; n0 += Np) {
    const int n = n0 + nl;
    double sa = 0.0, sb = 0.0;
    if (n < N) {
      const long unit = (long)n * G + g;
      for (int b = sl; b < B; b += S) {
        const double* o = partial + ((unit * B + b) * cg + j) * 2;
        sa += o[0]; sb += o[1];
      }
    }
    __syncthreads();
    sh[t * 2] = sa; sh[t * 2 + 1] = sb;
    __syncthreads();
    if (sl == 0) {      // slices in fixed order
      sa = 0.0; sb = 0.0;
      for (int s2 = 0; s2 < S; ++s2) { sa += sh[((s2 * Np + nl) * cg + j) * 2]; sb += sh[((s2 * Np + nl) * cg + j) * 2 + 1]; }
      tot[(nl * cg + j) * 2] = sa; tot[(nl * cg + j) * 2 + 1] = sb;
    }
    __syncthreads();
    if (t < Np && n0 + t < N) {      // c1, c2 of sample n0 + t
      double s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < cg; ++q) {
        s1 += (double)gamma[g * cg + q] * tot[(t * cg + q) * 2 + 1];
        s2 += (double)gamma[g * cg + q] * tot[(t * cg + q) * 2];
      }
      const long unit = (long)(n0 + t) * G + g;
      c1[unit] = (float)(s1 / L);
      c2[unit] = (float)(s2 / L);
    }
    if (t < cg) {                    // samples in fixed order
      for (int q = 0; q < Np && n0 + q < N; ++q) { ga += tot[(q * cg + t) * 2]; gb += tot[(q * cg + t) * 2 + 1]; }
    }
  }
  if (t < cg) {
    const int idx = g * cg + t;
    dgamma[idx] = accum ? dgamma[idx] + (float)ga : (float)ga;
    dbeta[idx] = accum ? dbeta[idx] + (float)gb : (float)gb;
  }
}
__global__ __launch_bounds__(256) void lp_gn_bwd_finalize_kernel(const double* partial, const float* gamma, float* dgamma, float* dbeta, float* c1,
                                                                 float* c2, int N, int G, int B, int cg, double L, int accum) {
  __shared__ double sh[256 * 2];
  __shared__ double tot[256 * 2];
  lp_gn_bwd_finalize_body(partial, gamma, dgamma, dbeta, c1, c2, N, G, B, cg, L, accum, blockIdx.x, sh, tot);
}
// apply pass, chunked like lp_gn_apply_chunk_kernel: a workgroup streams one contiguous piece of one (n, group) unit, a thread's
// channels / statistics / c1, c2 are loop constants (the grid-stride form did 5 64-bit divisions per 16 bytes: 3.9 TB/s)
template <typename T>
__global__ __launch_bounds__(256) void lp_gn_bwd_apply_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                              unsigned short* __restrict__ dx, float* __restrict__ dx32,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ c1, const float* __restrict__ c2, long L, long per, int C,
                                                              int G, int cg, int lddy, int relu, double* dbias_part) {
  __shared__ float dbsh[256 * 8];
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int unit = blockIdx.y, g = unit % G;
  const long lo = (long)unit * L;                      // (= n * E + g * L)
  const long a = lo + (long)blockIdx.x * per;
  const long bnd = (a + per < lo + L) ? a + per : lo + L;
  const int t8 = threadIdx.x * 8, c = t8 % C;
  float gam[8], bet[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int idx = g * cg + ((c + e) % cg); gam[e] = gamma[idx]; bet[e] = beta[idx]; }
  const float m = mean[unit], rs = rstd[unit], k1 = c1[unit], k2 = c2[unit];
  const long pstep = 2048 / C, pix = a / C + t8 / C;
  const long K = (bnd - a) / 2048;
  auto one = [&](const u32x4 rx, const u32x4 rd, long i) {
    float v[8], d[8], o[8];
    unpack8<T>(rx, v);
    unpack8<T>(rd, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - m) * rs;
      float de = d[e];
      if (relu && !(xh * gam[e] + bet[e] > 0.f)) de = 0.f;
      o[e] = (de * gam[e] - k1 - xh * k2) * rs;
      cs[e] += o[e];
    }
    *reinterpret_cast<u32x4*>(dx + i) = pack8<T>(o);
    if (dx32 != nullptr) {
      *reinterpret_cast<f32x4*>(dx32 + i) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(dx32 + i + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
  };
  long k = 0;
  for (; k + 2 <= K; k += 2) {
    u32x4 rx[2], rd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      rx[j] = *reinterpret_cast<const u32x4*>(x + a + t8 + (k + j) * 2048);
      rd[j] = *reinterpret_cast<const u32x4*>(dy + (pix + (k + j) * pstep) * lddy + c);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) one(rx[j], rd[j], a + t8 + (k + j) * 2048);
  }
  for (; k < K; ++k)
    one(*reinterpret_cast<const u32x4*>(x + a + t8 + k * 2048), *reinterpret_cast<const u32x4*>(dy + (pix + k * pstep) * lddy + c), a + t8 + k * 2048);
  if (dbias_part != nullptr)     // (launch-uniform; 2048 % C == 0: thread t always holds octet t mod C/8)
    lp_dbias_block(cs, 0, C / 8, dbias_part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * C, dbsh);
}
static int lp_gnb_blocks(long L) {
  long b = L / (2048 * 16);
  if (b < 1) b = 1;
  if (b > 256) b = 256;
  return (int)b;
}
extern "C" long bts_lp_gn_bwd_workspace(int N, long V, int C, int G) {
  if (N <= 0 || V <= 0 || C <= 0 || G <= 0 || C % G != 0) return -1;
  const long L = V * C / G;
  return (long)N * G * lp_gnb_blocks(L) * (C / G) * 2 * 8 + (long)N * G * 2 * 4 + 64 + 16384L * C * 8 + 64;   // (+ bias-gradient rows)
}
// x dense (N,V,C) in the storage type; dy rows of stride lddy; dx dense in the storage type, dx32 (may be NULL) the same values in fp32
// finalize + apply (+ bias-gradient finalize) on class-sum partial rows [N*G][B][cg][2] -- lp_gn_bwd_reduce_kernel's, or the ones a
// data-gradient conv left from its epilogue (LpGnbFuse).  tail: c1, c2 ([N*G] floats each), then the 64-byte aligned bias rows
static int lp_gn_bwd_tail(int dtype, const void* x, const void* dy, void* dx, float* dx32, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, float* dgamma, float* dbeta, const double* partial, int B, float* tail, int N, long L, int C, int lddy,
                          int G, int relu, int accumulate_params, float* dbias, hipStream_t stream) {
  const int cg = C / G;
  float* c1 = tail;
  float* c2 = c1 + (long)N * G;
  double* dbp = dbias ? reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(c2 + (long)N * G) + 63) & ~(uintptr_t)63) : nullptr;
  hipLaunchKernelGGL(lp_gn_bwd_finalize_kernel, dim3(G), dim3(256), 0, stream, partial, gamma, dgamma, dbeta, c1, c2, N, G, B, cg, (double)L, accumulate_params);
  BTS_LAUNCH_CHECK();
  int Ba;
  const long per = lp_chunk_per(L, (long)N * G, &Ba, dbias ? 2048 : 4096);   // (bias rows: one per block -- 2048 blocks of 256 fill the chip eight waves deep)
  const long blocks = (long)N * G * Ba;
  if (blocks > 16384) return BTS_ERR_SHAPE;       // (the bias rows of the workspace)
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_gn_bwd_apply_kernel<TF16>, dim3((unsigned)Ba, (unsigned)(N * G)), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)dy, (unsigned short*)dx, dx32, gamma, beta, mean, rstd, c1, c2, L, per, C, G, cg, lddy, relu, dbp);
  else hipLaunchKernelGGL(lp_gn_bwd_apply_kernel<TBF16>, dim3((unsigned)Ba, (unsigned)(N * G)), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)dy, (unsigned short*)dx, dx32, gamma, beta, mean, rstd, c1, c2, L, per, C, G, cg, lddy, relu, dbp);
  BTS_LAUNCH_CHECK();
  if (dbias != nullptr) {      // bias gradient of the conv whose output this GroupNorm normalised (dx IS that conv's dy)
    hipLaunchKernelGGL(lp_dbias_finalize_kernel, dim3(C), dim3(256), 0, stream, dbp, dbias, (int)blocks, C, accumulate_params);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}
static int lp_gn_bwd_check(int dtype, const void* x, const void* dy, const void* dx, const float* dx32, int N, long V, int C, int lddy, int G) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || C < G || C % G != 0 || C % 8 != 0 || C > 256 || (C & (C - 1)) != 0 || lddy % 8 != 0 || lddy < C) return BTS_ERR_SHAPE;
  const long E = V * C, L = E / G;
  const int cg = C / G;
  if (E % G != 0 || L % 2048 != 0 || cg > 32 || 256 % cg != 0) return BTS_ERR_UNSUPPORTED;   // (the caller falls back to the fp32 kernels)
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dy) & 15) || (((uintptr_t)dx) & 15) || (dx32 && (((uintptr_t)dx32) & 15))) return BTS_ERR_ALIGN;
  return BTS_OK;
}
extern "C" int bts_lp_gn_bwd(int dtype, const void* x, const void* dy, void* dx, float* dx32, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, float* dgamma, float* dbeta, void* workspace, long workspace_bytes, int N,
                             long V, int C, int lddy, int G, int relu, int accumulate_params, float* dbias, hipStream_t stream) {
  const int chk = lp_gn_bwd_check(dtype, x, dy, dx, dx32, N, V, C, lddy, G);
  if (chk != BTS_OK) return chk;
  const long E = V * C, L = E / G;
  const int cg = C / G;
  if (workspace_bytes < bts_lp_gn_bwd_workspace(N, V, C, G)) return BTS_ERR_WORKSPACE;
  const int B = lp_gnb_blocks(L);
  const long span = ((L / 2048 + B - 1) / B) * 2048;
  double* partial = reinterpret_cast<double*>(workspace);
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_gn_bwd_reduce_kernel<TF16>, dim3(B, N * G), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)dy, gamma, beta, mean, rstd, partial, E, L, span, C, G, cg, lddy, relu);
  else hipLaunchKernelGGL(lp_gn_bwd_reduce_kernel<TBF16>, dim3(B, N * G), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)dy, gamma, beta, mean, rstd, partial, E, L, span, C, G, cg, lddy, relu);
  BTS_LAUNCH_CHECK();
  return lp_gn_bwd_tail(dtype, x, dy, dx, dx32, gamma, beta, mean, rstd, dgamma, dbeta, partial, B, reinterpret_cast<float*>(partial + (long)N * G * B * cg * 2),
                        N, L, C, lddy, G, relu, accumulate_params, dbias, stream);
}

// da = conv3x3x3^T(dy) (stride 1; stored dense in the storage type) and the backward of the GroupNormalization (+ReLU) that da is the
// output gradient of -- resnet.py:80-93 in reverse under train.py:151: conv2's data gradient, then norm1 -- as ONE entry point, so that
// the class sums GroupNorm's backward needs first (A_j, B_j above) leave the conv's epilogue instead of costing a reduce pass over da
// and c (2 of the 5 tensor passes of a GroupNorm backward).  Fused where the z-marching kernel takes the layer (lowp_s1z.hip: the
// data-gradient contraction over 16 | 32 channels, <= 32 GroupNorm channels, whole planes per group); otherwise the conv and
// bts_lp_gn_bwd run back to back: same results up to the order of the fp32 class sums.  (D,H,W): the grid; Cg = GroupNorm channels =
// the forward conv's INPUT channels, Cdy = dy's channels (row stride lddy); wp_bwd = bts_lp_pack(BTS_CONV_K3S1, BTS_ROLE_BWD_DATA, ...).
// *fused_out (may be NULL) <- 1 when the epilogue form ran, 0 otherwise.
extern "C" long bts_lp_conv3d_bwd_data_gn_bwd_workspace(int N, int D, int H, int W, int Cg, int Cdy, int G) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cg <= 0 || Cdy <= 0 || G <= 0 || Cg % G != 0) return -1;
  const long V = (long)D * H * W;
  const long conv = ((lp_s1_workspace(N, D, H, W, Cdy, Cg) + 63) / 64) * 64;
  const long plain = bts_lp_gn_bwd_workspace(N, V, Cg, G);
  const long B = bts_lp_s1z_gnb_B_(N, D, H, W, Cdy, Cdy, Cg, Cg, G);
  const long fused = B > 0 ? (long)N * G * B * (Cg / G) * 2 * 8 + (long)N * G * 2 * 4 + 64 + 16384L * Cg * 8 + 64 : 0;
  return conv + (fused > plain ? fused : plain) + 64;
}
extern "C" int bts_lp_conv3d_bwd_data_gn_bwd(int dtype, const void* dy, const void* wp_bwd, void* da, const void* c, void* dc, float* dc32,
                                             const float* gamma, const float* beta, const float* mean, const float* rstd, float* dgamma,
                                             float* dbeta, void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cg, int Cdy,
                                             int lddy, int G, int relu, int accumulate_params, float* dbias, int* fused_out,
                                             hipStream_t stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cg <= 0 || Cdy <= 0 || G <= 0 || Cg % G != 0) return BTS_ERR_SHAPE;
  const long V = (long)D * H * W;
  const int chk = lp_gn_bwd_check(dtype, c, da, dc, dc32, N, V, Cg, Cg, G);
  if (chk != BTS_OK) return chk;
  if (workspace == nullptr || workspace_bytes < bts_lp_conv3d_bwd_data_gn_bwd_workspace(N, D, H, W, Cg, Cdy, G) || (((uintptr_t)workspace) & 15))
    return BTS_ERR_WORKSPACE;
  const long conv_ws = ((lp_s1_workspace(N, D, H, W, Cdy, Cg) + 63) / 64) * 64;
  char* tail = reinterpret_cast<char*>(workspace) + conv_ws;
  const long L = V * Cg / G;
  const int cg = Cg / G;
  if (fused_out) *fused_out = 0;
  const long B = bts_lp_s1z_gnb_B_(N, D, H, W, Cdy, lddy, Cg, Cg, G);
  if (B > 0 && B <= 0x7fffffffL && B == bts_lp_s1z_gnb_B_(N, D, H, W, Cdy, Cdy, Cg, Cg, G)) {     // (the workspace was sized for dense dy)
    LpGnbFuse f;
    f.x = (const unsigned short*)c; f.gamma = gamma; f.beta = beta; f.mean = mean; f.rstd = rstd;
    f.part = reinterpret_cast<double*>(tail); f.G = G; f.cg = cg; f.relu = relu; f.B = B;
    const int r = lp_conv_run(1, dtype, dy, wp_bwd, nullptr, da, nullptr, 0, N, D, H, W, Cdy, lddy, Cg, Cg, 0, stream, nullptr, nullptr, 0, &f);
    if (r == BTS_OK) {
      if (fused_out) *fused_out = 1;
      return lp_gn_bwd_tail(dtype, c, da, dc, dc32, gamma, beta, mean, rstd, dgamma, dbeta, f.part, (int)B,
                            reinterpret_cast<float*>(f.part + (long)N * G * B * cg * 2), N, L, Cg, Cg, G, relu, accumulate_params, dbias, stream);
    }
    if (r != 1) return r;
  }
  const int r = lp_conv_run(1, dtype, dy, wp_bwd, nullptr, da, workspace, conv_ws, N, D, H, W, Cdy, lddy, Cg, Cg, 0, stream);
  if (r != BTS_OK) return r;
  return bts_lp_gn_bwd(dtype, c, da, dc, dc32, gamma, beta, mean, rstd, dgamma, dbeta, tail, workspace_bytes - conv_ws, N, V, Cg, Cg, G, relu,
                       accumulate_params, dbias, stream);
}

// =====================================================================================================================
// Weight gradient of the stride-1 3x3x3 and 1x1x1 convolutions on 16-bit operands (what TF autodiff derives for the Conv3D
// kernels of resnet.py:30-37,80-87,96-103; train.py:151):  dW[t][c][k] = sum_v P[v + off_t][c] * Q[v][k],  P = the conv's input,
// Q = the gradient of its output.  The contraction runs over VOXELS, so both matrix operands want 8 consecutive voxels of one
// channel per lane (v_mfma_f32_32x32x16: A row = input channel, B column = output channel, K = 16 voxels along x) while memory is
// channel-fastest: the (halo) tiles are staged voxel-major in LDS as they come and every fragment is gathered with eight
// 2-byte LDS reads -- taps, which shift the 8-voxel window by single voxels, cost nothing extra that way.  A Q fragment serves all
// of a wave's taps, a P fragment all of its cout blocks.  8 waves: the 27 taps are dealt round-robin (1x1x1: the 32 x-rows of the
// tile are), persistent workgroups accumulate over their tiles and leave fp32 partials for a fixed-order finalize that also folds
// the encoder's duplicated slice back onto both copies of the weight (encoder.py:83-87) and adds into the gradient buffer.
// =====================================================================================================================
struct LpWgParams {
  const unsigned short* p;
  const unsigned short* q;
  float* part;
  int N, D, H, W, Cp, ldp, Cq, ldq, ntaps;
  int ntx, nty, ntz;
  long ntiles;
  int ncp, ncqg;
};
#define LPW_TX 16
#define LPW_TY 8
#define LPW_TZ 4

template <typename T, int NQ, bool K3>
__global__ __launch_bounds__(512, 1) void lp_wgrad_kernel(const LpWgParams p) {
  constexpr int TX = LPW_TX, TY = LPW_TY, TZ = LPW_TZ;
  constexpr int PS = 36, QS = 32 * NQ + 4;          // halves per staged voxel (8-byte aligned rows, bank-skewed)
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  constexpr bool k3 = K3;
  constexpr int halo = K3 ? 1 : 0;
  constexpr int SX = TX + 2 * halo, SY = TY + 2 * halo, SZ = TZ + 2 * halo;
  constexpr int nvp = SX * SY * SZ;
  unsigned short* ldsP = lds;
  unsigned short* ldsQ = lds + (TX + 2) * (TY + 2) * (TZ + 2) * PS;
  const int cpt = blockIdx.y / p.ncqg, cqg = blockIdx.y % p.ncqg;
  const int cp0 = cpt * 32, cq0 = cqg * 32 * NQ;
  f32x16 acc[4][NQ];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < NQ; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;
  auto tile_origin = [&](long tile, int& n, int& x0, int& y0, int& z0) {
    long b = tile;
    const int tx = (int)(b % p.ntx); b /= p.ntx;
    const int ty = (int)(b % p.nty); b /= p.nty;
    const int tz = (int)(b % p.ntz);
    n = (int)(b / p.ntz);
    x0 = tx * TX; y0 = ty * TY; z0 = tz * TZ;
  };

  if constexpr (K3) {
    // ---- 3x3x3: LDS tiles with PAIRS of x-neighbours interleaved per channel, [row (z, y)][x pair][channel] dwords (low half = the
    // even slot).  A matrix operand (8 consecutive voxels of one channel) is then 4 consecutive pair-dwords of a lane's channel: 4 or
    // 5 ds_read_b32 + 4 v_alignbit (a tap's window starts at an odd slot for kx = 0, 2) instead of eight 2-byte reads and their
    // packing.  Staging interleaves two voxels' 8-channel chunks with v_perm and writes 32 contiguous bytes.
    constexpr int NPP = (SX + 2) / 2;            // pairs per P row: slots x = -2 .. SX - 1
    constexpr int NPQ = TX / 2;
    constexpr int PU = SZ * SY * NPP * 4, QU = TZ * TY * NPQ * 4 * NQ;    // staging units: (row, pair, channel octet)
    constexpr int PUS = (PU + 511) / 512, QUS = (QU + 511) / 512;
    unsigned* const ldsP32 = reinterpret_cast<unsigned*>(lds);
    unsigned* const ldsQ32 = ldsP32 + SZ * SY * NPP * 32;
    u32x4 preP[PUS][2], preQ[QUS][2];
    auto fetch = [&](long tile) {
      int n, x0, y0, z0;
      tile_origin(tile, n, x0, y0, z0);
#pragma unroll
      for (int i = 0; i < PUS; ++i) {
        const int e = tid + i * 512;
        preP[i][0] = preP[i][1] = u32x4{0u, 0u, 0u, 0u};
        if (e < PU) {
          const int oct = e & 3, xp = (e >> 2) % NPP, row = (e >> 2) / NPP;
          const int gz = z0 - 1 + row / SY, gy = y0 - 1 + row % SY;
          if ((unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && cp0 + oct * 8 < p.Cp) {
            const unsigned short* rowp = p.p + (((long)n * p.D + gz) * p.H + gy) * (long)p.W * p.ldp + cp0 + oct * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const int slot = 2 * xp + s2, gx = x0 - 2 + slot;   // slots 1 .. SX are the halo tile's columns -1 .. TX
              if (slot >= 1 && slot <= SX && (unsigned)gx < (unsigned)p.W) preP[i][s2] = *reinterpret_cast<const u32x4*>(rowp + (long)gx * p.ldp);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < QUS; ++i) {
        const int e = tid + i * 512;
        preQ[i][0] = preQ[i][1] = u32x4{0u, 0u, 0u, 0u};
        if (e < QU) {
          const int oct = e % (4 * NQ), xp = (e / (4 * NQ)) % NPQ, row = e / (4 * NQ * NPQ);
          const int gz = z0 + row / TY, gy = y0 + row % TY;
          if (gz < p.D && gy < p.H && cq0 + oct * 8 < p.Cq) {
            const unsigned short* rowq = p.q + (((long)n * p.D + gz) * p.H + gy) * (long)p.W * p.ldq + cq0 + oct * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const int gx = x0 + 2 * xp + s2;
              if (gx < p.W) preQ[i][s2] = *reinterpret_cast<const u32x4*>(rowq + (long)gx * p.ldq);
            }
          }
        }
      }
    };
    // (even voxel's channels c, c+1 | odd voxel's c, c+1) x 4 -> channel c: (even, odd), channel c+1: (even, odd)
    auto weave = [&](const u32x4 ev, const u32x4 od, unsigned* dst) {
      u32x4 lo, hi;
      lo[0] = __builtin_amdgcn_perm(od[0], ev[0], 0x05040100u); lo[1] = __builtin_amdgcn_perm(od[0], ev[0], 0x07060302u);
      lo[2] = __builtin_amdgcn_perm(od[1], ev[1], 0x05040100u); lo[3] = __builtin_amdgcn_perm(od[1], ev[1], 0x07060302u);
      hi[0] = __builtin_amdgcn_perm(od[2], ev[2], 0x05040100u); hi[1] = __builtin_amdgcn_perm(od[2], ev[2], 0x07060302u);
      hi[2] = __builtin_amdgcn_perm(od[3], ev[3], 0x05040100u); hi[3] = __builtin_amdgcn_perm(od[3], ev[3], 0x07060302u);
      reinterpret_cast<u32x4*>(dst)[0] = lo;
      reinterpret_cast<u32x4*>(dst)[1] = hi;
    };
    auto commit = [&]() {
#pragma unroll
      for (int i = 0; i < PUS; ++i) {
        const int e = tid + i * 512;
        if (e < PU) weave(preP[i][0], preP[i][1], ldsP32 + (e >> 2) * 32 + (e & 3) * 8);
      }
#pragma unroll
      for (int i = 0; i < QUS; ++i) {
        const int e = tid + i * 512;
        if (e < QU) weave(preQ[i][0], preQ[i][1], ldsQ32 + (e / (4 * NQ)) * (32 * NQ) + (e % (4 * NQ)) * 8);
      }
    };
    if constexpr (NQ == 1) {
      // Tap dealing: six compute waves = (dy, half of the tile's y rows); a wave walks the P rows (z', y' = y + dy) of its four y
      // and uses each row's three x windows (kx = 0, 1, 2: window starts at slots 1, 2, 3 -- six pair reads, five alignbits) for
      // ALL three dz with the Q rows z = z' - dz, which roll through three fragment registers: 10 LDS reads + 5 vector-ALU
      // instructions per 9 matrix instructions at full depth.  Nine accumulators (dz, kx) per wave; the two halves of a dy meet in
      // LDS at the end.  Waves 6, 7 only help staging.
      const bool cw = wave < 6;
      const int dy = wave % 3, half = (wave / 3) & 1;
      const unsigned* pbase = ldsP32 + 4 * h * 32 + l32;
      const unsigned* qbase = ldsQ32 + 4 * h * 32 + l32;
      f32x16 a9[3][3];
  #pragma unroll
      for (int dz = 0; dz < 3; ++dz)
  #pragma unroll
        for (int kx = 0; kx < 3; ++kx)
  #pragma unroll
          for (int r = 0; r < 16; ++r) a9[dz][kx][r] = 0.f;
      long tile = blockIdx.x;
      if (tile < p.ntiles) fetch(tile);
      for (; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();      // every wave is done with the previous tile
        commit();
        __syncthreads();
        if (tile + gridDim.x < p.ntiles) fetch(tile + gridDim.x);
        if (cw) {
  #pragma unroll 2
          for (int yi = 0; yi < TY / 2; ++yi) {
            const int yq = half * (TY / 2) + yi;
            const unsigned* prow = pbase + ((yq + dy) * NPP) * 32;
            const unsigned* qrow = qbase + (yq * NPQ) * 32;
            u32x4 qf[3];
  #pragma unroll
            for (int zp = 0; zp < SZ; ++zp) {
              if (zp < TZ) {
                const unsigned* qr = qrow + (zp * TY * NPQ) * 32;
                qf[zp % 3] = u32x4{qr[0], qr[32], qr[64], qr[96]};
              }
              const unsigned* pr = prow + (zp * SY * NPP) * 32;
              unsigned pp[6], sh[5];
  #pragma unroll
              for (int k = 0; k < 6; ++k) pp[k] = pr[k * 32];
  #pragma unroll
              for (int k = 0; k < 5; ++k) sh[k] = __builtin_amdgcn_alignbit(pp[k + 1], pp[k], 16);
              const u32x4 w0 = {sh[0], sh[1], sh[2], sh[3]}, w1 = {pp[1], pp[2], pp[3], pp[4]}, w2 = {sh[1], sh[2], sh[3], sh[4]};
  #pragma unroll
              for (int dz = 0; dz < 3; ++dz) {
                const int z = zp - dz;
                if (z >= 0 && z < TZ) {
                  a9[dz][0] = T::mfma(w0, qf[z % 3], a9[dz][0]);
                  a9[dz][1] = T::mfma(w1, qf[z % 3], a9[dz][1]);
                  a9[dz][2] = T::mfma(w2, qf[z % 3], a9[dz][2]);
                }
              }
            }
          }
        }
      }
      // the two halves of a dy: waves 3..5 hand their nine accumulators over through LDS, waves 0..2 add and write the partials
      __syncthreads();
      float* xch = reinterpret_cast<float*>(lds);
      if (wave >= 3 && wave < 6) {
  #pragma unroll
        for (int dz = 0; dz < 3; ++dz)
  #pragma unroll
          for (int kx = 0; kx < 3; ++kx)
  #pragma unroll
            for (int r = 0; r < 16; ++r) xch[((((wave - 3) * 9 + dz * 3 + kx) * 16) + r) * 64 + lane] = a9[dz][kx][r];
      }
      __syncthreads();
      if (wave < 3) {
        float* pb = p.part + (((long)blockIdx.x * p.ncp + cpt) * p.ncqg + cqg) * (long)27 * 32 * 32;
  #pragma unroll
        for (int dz = 0; dz < 3; ++dz)
  #pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int slot = (dz * 3 + dy) * 3 + kx;
  #pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
              pb[((long)slot * 32 + row) * 32 + l32] = a9[dz][kx][r] + xch[(((wave * 9 + dz * 3 + kx) * 16) + r) * 64 + lane];
            }
          }
      }
      return;
    } else {
      // wave w owns taps w, w+8, w+16 (and w+24 for w < 3) and walks all 32 x-rows of the tile; a tap's window starts at slot
      // kx + 1 (+ 8 for the lanes of the upper K half): pair offset and alignbit shift are wave constants
      int toff[4], tsh[4];
  #pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const int t = wave + 8 * ti;
        const int tt = t < 27 ? t : 0;
        const int kx = tt % 3;
        toff[ti] = ((((tt / 9) * SY + (tt / 3) % 3) * NPP) + ((kx + 1) >> 1)) * 32;
        tsh[ti] = ((kx + 1) & 1) ? 16 : 0;
      }
      const bool t3 = wave + 24 < 27;
      const unsigned* pbase = ldsP32 + 4 * h * 32 + l32;
      const unsigned* qbase = ldsQ32 + 4 * h * (32 * NQ) + l32;
      auto pfrag = [&](const unsigned* prow, int off, int sh) {
        unsigned pp[5];
  #pragma unroll
        for (int k = 0; k < 5; ++k) pp[k] = prow[off + k * 32];
        return u32x4{__builtin_amdgcn_alignbit(pp[1], pp[0], sh), __builtin_amdgcn_alignbit(pp[2], pp[1], sh),
                     __builtin_amdgcn_alignbit(pp[3], pp[2], sh), __builtin_amdgcn_alignbit(pp[4], pp[3], sh)};
      };
      long tile = blockIdx.x;
      if (tile < p.ntiles) fetch(tile);
      for (; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();      // every wave is done with the previous tile
        commit();
        __syncthreads();
        if (tile + gridDim.x < p.ntiles) fetch(tile + gridDim.x);
  #pragma unroll 4
        for (int kb = 0; kb < TY * TZ; ++kb) {
          const int z = kb / TY, y = kb % TY;
          u32x4 bq[NQ];
  #pragma unroll
          for (int c = 0; c < NQ; ++c) {
            const unsigned* qr = qbase + (kb * NPQ) * (32 * NQ) + c * 32;
            bq[c] = u32x4{qr[0], qr[32 * NQ], qr[2 * 32 * NQ], qr[3 * 32 * NQ]};
          }
          const unsigned* prow = pbase + ((z * SY + y) * NPP) * 32;
  #pragma unroll
          for (int ti = 0; ti < 3; ++ti) {
            const u32x4 a = pfrag(prow, toff[ti], tsh[ti]);
  #pragma unroll
            for (int c = 0; c < NQ; ++c) acc[ti][c] = T::mfma(a, bq[c], acc[ti][c]);
          }
          if (t3) {
            const u32x4 a = pfrag(prow, toff[3], tsh[3]);
  #pragma unroll
            for (int c = 0; c < NQ; ++c) acc[3][c] = T::mfma(a, bq[c], acc[3][c]);
          }
        }
      }
    }
  } else {
    // ---- 1x1x1: voxel-major tiles as they come, every fragment gathered with eight 2-byte LDS reads; wave w takes x-rows w, w+8,
    // w+16, w+24 (one accumulator per wave; the finalize adds the eight)
    constexpr int PSLOT = (nvp * 4 + 511) / 512;   // 16-byte chunks of the P tile per thread
    constexpr int QSLOT = (TX * TY * TZ * 4 * NQ + 511) / 512;
    u32x4 preP[PSLOT], preQ[QSLOT];
    auto fetch = [&](long tile) {
      int n, x0, y0, z0;
      tile_origin(tile, n, x0, y0, z0);
#pragma unroll
      for (int i = 0; i < PSLOT; ++i) {
        const int e = tid + i * 512;
        preP[i] = u32x4{0u, 0u, 0u, 0u};
        if (e < nvp * 4) {
          const int vox = e >> 2, cq = e & 3;
          const int vz = vox / (SY * SX), r = vox - vz * (SY * SX), vy = r / SX, vx = r - vy * SX;
          const int gz = z0 - halo + vz, gy = y0 - halo + vy, gx = x0 - halo + vx;
          if ((unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && cp0 + cq * 8 < p.Cp)
            preP[i] = *reinterpret_cast<const u32x4*>(p.p + ((((long)n * p.D + gz) * p.H + gy) * p.W + gx) * (long)p.ldp + cp0 + cq * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < QSLOT; ++i) {
        const int e = tid + i * 512;
        preQ[i] = u32x4{0u, 0u, 0u, 0u};
        if (e < TX * TY * TZ * 4 * NQ) {
          const int vox = e / (4 * NQ), cq = e % (4 * NQ);
          const int vz = vox / (TY * TX), r = vox - vz * (TY * TX), vy = r / TX, vx = r - vy * TX;
          const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
          if (gz < p.D && gy < p.H && gx < p.W && cq0 + cq * 8 < p.Cq)
            preQ[i] = *reinterpret_cast<const u32x4*>(p.q + ((((long)n * p.D + gz) * p.H + gy) * p.W + gx) * (long)p.ldq + cq0 + cq * 8);
        }
      }
    };
    auto commit = [&]() {
#pragma unroll
      for (int i = 0; i < PSLOT; ++i) {
        const int e = tid + i * 512;
        if (e < nvp * 4) {
          u32x2* d = reinterpret_cast<u32x2*>(ldsP + (e >> 2) * PS + (e & 3) * 8);
          d[0] = u32x2{preP[i][0], preP[i][1]};
          d[1] = u32x2{preP[i][2], preP[i][3]};
        }
      }
#pragma unroll
      for (int i = 0; i < QSLOT; ++i) {
        const int e = tid + i * 512;
        if (e < TX * TY * TZ * 4 * NQ) {
          u32x2* d = reinterpret_cast<u32x2*>(ldsQ + (e / (4 * NQ)) * QS + (e % (4 * NQ)) * 8);
          d[0] = u32x2{preQ[i][0], preQ[i][1]};
          d[1] = u32x2{preQ[i][2], preQ[i][3]};
        }
      }
    };
    // eight 2-byte reads -> one 16-byte matrix operand (K = 8 consecutive voxels along x of one channel)
    auto gather = [&](const unsigned short* base, int stride) {
      unsigned v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = base[j * stride];
      return u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    };
    long tile = blockIdx.x;
    if (tile < p.ntiles) fetch(tile);
    for (; tile < p.ntiles; tile += gridDim.x) {
      __syncthreads();      // every wave is done with the previous tile
      commit();
      __syncthreads();
      if (tile + gridDim.x < p.ntiles) fetch(tile + gridDim.x);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const int kb = wave + 8 * ti;
        const int z = kb / TY, y = kb % TY;
        const u32x4 a = gather(ldsP + ((z * SY + y) * SX + 8 * h) * PS + l32, PS);
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
          const u32x4 b = gather(ldsQ + ((z * TY + y) * TX + 8 * h) * QS + c * 32 + l32, QS);
          acc[0][c] = T::mfma(a, b, acc[0][c]);
        }
      }
    }
  }
  // partial sums: part[wg][cp tile][cq group][slot][32 cin][32*NQ cout]; slot = tap (3x3x3) or wave (1x1x1)
  const int nslot = k3 ? 27 : 8;
  float* pb = p.part + (((long)blockIdx.x * p.ncp + cpt) * p.ncqg + cqg) * (long)nslot * 32 * (32 * NQ);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const int slot = k3 ? wave + 8 * ti : wave;
    if ((k3 && slot < 27) || (!k3 && ti == 0)) {
#pragma unroll
      for (int c = 0; c < NQ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          pb[((long)slot * 32 + row) * (32 * NQ) + c * 32 + l32] = acc[ti][c][r];
        }
    }
  }
}

struct LpWfParams {
  const float* part;
  float* dw;
  int nwg, ncp, ncqg, nslot, ntaps, NQ, Cp, Cq, Cin_ref, dup_start, dup_shift, accum;
};
// dW[t][c_ref][k] (+)= sum over workgroups (and, for 1x1x1, waves) of the partials, fixed order; a slab channel c is reference
// channel c + shift and, inside the duplicated slice, ALSO reference channel c - dup_start (both copies get the gradient)
__global__ __launch_bounds__(256) void lp_wgrad_finalize_kernel(const LpWfParams f) {
  // 32 consecutive elements x 8 slices of the workgroup list per block: coalesced partial reads, slices combined in fixed order
  __shared__ double sh[8][32];
  const long total = (long)f.ntaps * f.Cp * f.Cq;
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (long i0 = blockIdx.x * 32L; i0 < total; i0 += (long)gridDim.x * 32) {
    const long i = i0 + el;
    double s = 0.0;
    int k = 0, c = 0, t = 0;
    if (i < total) {
      k = (int)(i % f.Cq);
      const long r = i / f.Cq;
      c = (int)(r % f.Cp);
      t = (int)(r / f.Cp);
      const int cpt = c / 32, row = c % 32, cqg = k / (32 * f.NQ), col = k % (32 * f.NQ);
      const int s0 = f.ntaps == 27 ? t : 0, s1 = f.ntaps == 27 ? t + 1 : f.nslot;      // (1x1x1: the general kernel's 8 wave slots, the streaming kernel's one)
      for (int wg = sl; wg < f.nwg; wg += 8) {
        const float* pb = f.part + (((long)wg * f.ncp + cpt) * f.ncqg + cqg) * (long)f.nslot * 32 * (32 * f.NQ);
        for (int q = s0; q < s1; ++q) s += pb[((long)q * 32 + row) * (32 * f.NQ) + col];
      }
    }
    __syncthreads();
    sh[sl][el] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) tot += sh[q][el];
      const float v = (float)tot;
      float* d0 = f.dw + ((long)t * f.Cin_ref + c + f.dup_shift) * f.Cq + k;
      *d0 = f.accum ? *d0 + v : v;
      if (f.dup_shift > 0 && c >= f.dup_start && c < f.dup_start + f.dup_shift) {
        float* d1 = f.dw + ((long)t * f.Cin_ref + (c - f.dup_start)) * f.Cq + k;
        *d1 = f.accum ? *d1 + v : v;
      }
    }
  }
}
// The same for the 3x3x3 launches (one slot per tap) with Cq % 4 == 0: a thread owns FOUR consecutive columns (16-byte partial reads:
// the 4-byte version above read 28 MB in 29 us = 1 TB/s, 1.7 ms of the batch-8 step), 32 threads x 4 = 128 elements x 8 slices per block
__global__ __launch_bounds__(256) void lp_wgrad_finalize4_kernel(const LpWfParams f) {
  __shared__ double sh[8][32][4];
  const long total4 = (long)f.ntaps * f.Cp * f.Cq / 4;
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int W = 32 * f.NQ;
  for (long i0 = blockIdx.x * 32L; i0 < total4; i0 += (long)gridDim.x * 32) {
    const long i4 = i0 + el;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int k = 0, c = 0, t = 0;
    if (i4 < total4) {
      const long i = i4 * 4;
      k = (int)(i % f.Cq);
      const long r = i / f.Cq;
      c = (int)(r % f.Cp);
      t = (int)(r / f.Cp);
      const int cpt = c / 32, row = c % 32, cqg = k / W, col = k % W;
      for (int wg = sl; wg < f.nwg; wg += 8) {
        const float* pb = f.part + (((long)wg * f.ncp + cpt) * f.ncqg + cqg) * (long)f.nslot * 32 * W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(pb + ((long)t * 32 + row) * W + col);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[sl][el][j] = s[j];
    __syncthreads();
    if (sl == 0 && i4 < total4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += sh[q][el][j];
        const float v = (float)tot;
        float* d0 = f.dw + ((long)t * f.Cin_ref + c + f.dup_shift) * f.Cq + k + j;
        *d0 = f.accum ? *d0 + v : v;
        if (f.dup_shift > 0 && c >= f.dup_start && c < f.dup_start + f.dup_shift) {
          float* d1 = f.dw + ((long)t * f.Cin_ref + (c - f.dup_start)) * f.Cq + k + j;
          *d1 = f.accum ? *d1 + v : v;
        }
      }
    }
  }
}
static void lp_wgrad_finalize_launch(const LpWfParams& f, hipStream_t stream) {
  const long total = (long)f.ntaps * f.Cp * f.Cq;
  if (f.ntaps == 27 && f.nslot == 27 && f.Cq % 4 == 0 && (((uintptr_t)f.part) & 15) == 0) {
    long blocks = (total / 4 + 31) / 32;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lp_wgrad_finalize4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f);
    return;
  }
  long blocks = (total + 31) / 32;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(lp_wgrad_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f);
}
int bts_lp_wgrad_finalize_(const float* part, float* dw, int nwg, int ncp, int ncqg, int nslot, int ntaps, int NQ, int Cp, int Cq, int Cin_ref,
                           int dup_start, int dup_shift, int accum, hipStream_t stream) {
  LpWfParams f;
  f.part = part; f.dw = dw; f.nwg = nwg; f.ncp = ncp; f.ncqg = ncqg; f.nslot = nslot; f.ntaps = ntaps; f.NQ = NQ;
  f.Cp = Cp; f.Cq = Cq; f.Cin_ref = Cin_ref; f.dup_start = dup_start; f.dup_shift = dup_shift; f.accum = accum;
  (void)hipGetLastError();
  lp_wgrad_finalize_launch(f, stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
// lowp_wgs.hip: weight gradient of the strided convolutions (P on the fine grid, Q on the coarse one)
long bts_lp_wgs_workspace_(int N, int Dc, int Hc, int Wc, int Cp, int Cq);
int bts_lp_wgs_launch_(int dtype, const void* P, const void* Q, float* dw, void* ws, long ws_bytes, int N, int Df, int Hf, int Wf, int Dc, int Hc,
                       int Wc, int Cp, int ldp, int Cq, int ldq, int accum, hipStream_t stream);
// lowp_wgd.hip: the streaming weight-gradient kernel of the stride-1 3x3x3 convolutions with Cout <= 32
long bts_lp_wgd_workspace_(int N, int D, int H, int W, int Cp, int Cq);
int bts_lp_wgd_launch_(int dtype, const void* x, const void* dy, float* dw, void* ws, long ws_bytes, int N, int D, int H, int W, int Cp, int ldp,
                       int Cq, int ldq, int dup_start, int dup_shift, int accum, hipStream_t stream, const LpGnaFuse* ga = nullptr,
                       const void* dy2 = nullptr, float* dw1 = nullptr, int lddy2 = 0, long psplit = 0);
bool bts_lp_wgd_gna_ok_(int N, int D, int H, int W, int Cp, int Cq, int in_G);
// db[k] (+)= sum_n colsum[n][k]
__global__ void lp_bias_grad_kernel(const float* cs, float* db, int N, int C, int accum) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= C) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += cs[n * C + k];
  db[k] = accum ? db[k] + s : s;
}

static void lp_wg_plan(int kind, int N, int D, int H, int W, int Cp, int Cq, int& nq, int& nwg, long& ntiles, int& ncp, int& ncqg) {
  nq = Cq >= 64 ? 2 : 1;
  ncp = (Cp + 31) / 32;
  ncqg = (Cq + 32 * nq - 1) / (32 * nq);
  ntiles = (long)N * ((D + LPW_TZ - 1) / LPW_TZ) * ((H + LPW_TY - 1) / LPW_TY) * ((W + LPW_TX - 1) / LPW_TX);
  long cap = 256 / ((long)ncp * ncqg);          // one 512-thread workgroup per CU over the whole launch
  if (cap < 1) cap = 1;
  nwg = (int)(ntiles < cap ? ntiles : cap);
}
// strided kinds: (P channels, Q channels, coarse dims, voxels dy lives on) of a call with forward-input dims (D,H,W)
static void lp_wgs_roles(int kind, int D, int H, int W, int Cin, int Cout, int& Cp, int& Cq, int& Dc, int& Hc, int& Wc, long& Vdy) {
  if (kind == BTS_CONV_K3S2) { Cp = Cin; Cq = Cout; Dc = D / 2; Hc = H / 2; Wc = W / 2; Vdy = (long)Dc * Hc * Wc; }
  else { Cp = Cout; Cq = Cin; Dc = D; Hc = H; Wc = W; Vdy = 8L * D * H * W; }
}
extern "C" long bts_lp_conv3d_bwd_weight_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  if (kind == BTS_CONV_K3S2 || kind == BTS_CONV_K3S2T) {
    int Cp, Cq, Dc, Hc, Wc; long Vdy;
    lp_wgs_roles(kind, D, H, W, Cin, Cout, Cp, Cq, Dc, Hc, Wc, Vdy);
    const long part = ((bts_lp_wgs_workspace_(N, Dc, Hc, Wc, Cp, Cq) + 255) & ~255L);
    return part + (long)N * ((Cout + 7) / 8 * 8) * 4 + 256 + bts_lp_colsum_workspace(N, Vdy, (Cout + 7) / 8 * 8) + 256;
  }
  if (kind != BTS_CONV_K3S1 && kind != BTS_CONV_K1) return -1;
  int nq, nwg, ncp, ncqg; long ntiles;
  lp_wg_plan(kind, N, D, H, W, Cin, Cout, nq, nwg, ntiles, ncp, ncqg);
  const int nslot = kind == BTS_CONV_K3S1 ? 27 : 8;
  long part = (long)nwg * ncp * ncqg * nslot * 32 * 32 * nq * 4;
  if (kind == BTS_CONV_K3S1) {          // (the streaming kernel's slabs sit in the same place; one per workgroup too)
    const long alt = bts_lp_wgd_workspace_(N, D, H, W, Cin, Cout);
    if (alt > part) part = alt;
  }
  return part + (long)N * ((Cout + 7) / 8 * 8) * 4 + bts_lp_colsum_workspace(N, (long)D * H * W, (Cout + 7) / 8 * 8) + 256;
}
// dw (Keras layout (kd,kh,kw,Cin_ref,Cout), fp32) (+)= the weight gradient; db (may be NULL) (+)= sum of dy over voxels and samples.
// x (N,D,H,W,Cin) stride ldx, dy (N,D,H,W,Cout) DENSE when db is wanted; Cin, Cout multiples of 8.  K3S1 and K1 only (-3 otherwise).
extern "C" int bts_lp_conv3d_bwd_weight(int kind, int dtype, const void* x, const void* dy, float* dw, float* db, void* workspace,
                                        long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int lddy, int dup_start,
                                        int dup_shift, int accumulate, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (kind == BTS_CONV_K3S2 || kind == BTS_CONV_K3S2T) {
    // strided kinds (lowp_wgs.hip): x lives on the forward-input grid (D,H,W), dy on the half grid (stride-2 conv; TF 'same' with even
    // sizes pads (0,1): input 2o + t) or the doubled grid (transposed conv: output 2i + k)
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin % 8 != 0 || Cout % 8 != 0 || ldx % 8 != 0 || lddy % 8 != 0) return BTS_ERR_SHAPE;
    if (kind == BTS_CONV_K3S2 && ((D | H | W) & 1)) return BTS_ERR_UNSUPPORTED;
    if (dup_shift != 0) return BTS_ERR_UNSUPPORTED;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)dy) & 15) || (((uintptr_t)workspace) & 15)) return BTS_ERR_ALIGN;
    if (workspace_bytes < bts_lp_conv3d_bwd_weight_workspace(kind, N, D, H, W, Cin, Cout)) return BTS_ERR_WORKSPACE;
    int Cp, Cq, Dc, Hc, Wc; long Vdy;
    lp_wgs_roles(kind, D, H, W, Cin, Cout, Cp, Cq, Dc, Hc, Wc, Vdy);
    const long part = ((bts_lp_wgs_workspace_(N, Dc, Hc, Wc, Cp, Cq) + 255) & ~255L);
    int r;
    if (kind == BTS_CONV_K3S2)
      r = bts_lp_wgs_launch_(dtype, x, dy, dw, workspace, part, N, D, H, W, Dc, Hc, Wc, Cp, ldx, Cq, lddy, accumulate, stream);
    else
      r = bts_lp_wgs_launch_(dtype, dy, x, dw, workspace, part, N, 2 * D, 2 * H, 2 * W, Dc, Hc, Wc, Cp, lddy, Cq, ldx, accumulate, stream);
    if (r != BTS_OK) return r;
    if (db != nullptr) {
      if (lddy != Cout) return BTS_ERR_UNSUPPORTED;
      char* wsb = reinterpret_cast<char*>(workspace) + part;
      float* cs = reinterpret_cast<float*>(wsb);
      void* cws = wsb + (((long)N * Cout * 4 + 255) & ~255L);
      r = bts_lp_colsum(dtype, dy, cs, cws, bts_lp_colsum_workspace(N, Vdy, Cout), N, Vdy, Cout, 1.0f, stream);
      if (r != BTS_OK) return r;
      hipLaunchKernelGGL(lp_bias_grad_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, cs, db, N, Cout, accumulate);
      BTS_LAUNCH_CHECK();
    }
    return BTS_OK;
  }
  if (kind != BTS_CONV_K3S1 && kind != BTS_CONV_K1) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin % 8 != 0 || Cout % 8 != 0 || ldx % 8 != 0 || lddy % 8 != 0) return BTS_ERR_SHAPE;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dy) & 15) || (((uintptr_t)workspace) & 15)) return BTS_ERR_ALIGN;
  if (dup_shift < 0 || (dup_shift > 0 && dup_start + dup_shift > Cin)) return BTS_ERR_SHAPE;
  if (workspace_bytes < bts_lp_conv3d_bwd_weight_workspace(kind, N, D, H, W, Cin, Cout)) return BTS_ERR_WORKSPACE;
  LpWgParams p;
  int nq, nwg;
  lp_wg_plan(kind, N, D, H, W, Cin, Cout, nq, nwg, p.ntiles, p.ncp, p.ncqg);
  long part_bytes = (long)nwg * p.ncp * p.ncqg * (kind == BTS_CONV_K3S1 ? 27 : 8) * 32 * 32 * nq * 4;
  bool streamed = false;
  if (kind == BTS_CONV_K3S1) {          // few output channels at a large volume: the streaming kernel (lowp_wgd.hip), else the general one
    const long alt = bts_lp_wgd_workspace_(N, D, H, W, Cin, Cout);
    if (alt > 0) {
      const int r = bts_lp_wgd_launch_(dtype, x, dy, dw, workspace, alt > part_bytes ? alt : part_bytes, N, D, H, W, Cin, ldx, Cout, lddy, dup_start,
                                       dup_shift, accumulate, stream);
      if (r < 0) return r;
      streamed = r == BTS_OK;
      if (alt > part_bytes) part_bytes = alt;
    }
  }
  if (!streamed) {
  p.p = (const unsigned short*)x; p.q = (const unsigned short*)dy; p.part = reinterpret_cast<float*>(workspace);
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cp = Cin; p.ldp = ldx; p.Cq = Cout; p.ldq = lddy; p.ntaps = kind == BTS_CONV_K3S1 ? 27 : 1;
  p.ntx = (W + LPW_TX - 1) / LPW_TX; p.nty = (H + LPW_TY - 1) / LPW_TY; p.ntz = (D + LPW_TZ - 1) / LPW_TZ;
  const size_t shmem = ((size_t)(LPW_TX + 2) * (LPW_TY + 2) * (LPW_TZ + 2) * 36 + (size_t)LPW_TX * LPW_TY * LPW_TZ * (32 * nq + 4)) * 2;
  (void)hipGetLastError();
#define LPW_LAUNCH(TT, NQ_, K3_)                                                                                                \
  do {                                                                                                                        \
    auto kern = lp_wgrad_kernel<TT, NQ_, K3_>;                                                                                \
    static bool done = false;                                                                                                 \
    if (!done) {                                                                                                              \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return (int)e;                                                                                     \
      done = true;                                                                                                            \
    }                                                                                                                         \
    hipLaunchKernelGGL(kern, dim3(nwg, p.ncp * p.ncqg), dim3(512), shmem, stream, p);                                         \
  } while (0)
  const bool k3 = kind == BTS_CONV_K3S1;
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(32 | ((k3 ? 1 : 2) << 16), 2.0 * p.ntaps * (double)Cin * Cout * (double)N * D * H * W, stream);   // (variant 2: 1x1x1 -- 2 FLOP per operand byte, HBM-bound)
  if (dtype == LP_F16) {
    if (k3) { if (nq == 2) LPW_LAUNCH(TF16, 2, true); else LPW_LAUNCH(TF16, 1, true); }
    else { if (nq == 2) LPW_LAUNCH(TF16, 2, false); else LPW_LAUNCH(TF16, 1, false); }
  } else {
    if (k3) { if (nq == 2) LPW_LAUNCH(TBF16, 2, true); else LPW_LAUNCH(TBF16, 1, true); }
    else { if (nq == 2) LPW_LAUNCH(TBF16, 2, false); else LPW_LAUNCH(TBF16, 1, false); }
  }
#undef LPW_LAUNCH
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  LpWfParams f;
  f.part = p.part; f.dw = dw; f.nwg = nwg; f.ncp = p.ncp; f.ncqg = p.ncqg; f.nslot = kind == BTS_CONV_K3S1 ? 27 : 8; f.ntaps = p.ntaps; f.NQ = nq;
  f.Cp = Cin; f.Cq = Cout; f.Cin_ref = Cin + dup_shift; f.dup_start = dup_start; f.dup_shift = dup_shift; f.accum = accumulate;
  lp_wgrad_finalize_launch(f, stream);
  BTS_LAUNCH_CHECK();
  }
  if (db != nullptr) {
    if (lddy != Cout) return BTS_ERR_UNSUPPORTED;
    char* wsb = reinterpret_cast<char*>(workspace) + part_bytes;
    float* cs = reinterpret_cast<float*>(wsb);
    void* cws = wsb + (((long)N * Cout * 4 + 255) & ~255L);
    const int r = bts_lp_colsum(dtype, dy, cs, cws, bts_lp_colsum_workspace(N, (long)D * H * W, Cout), N, (long)D * H * W, Cout, 1.0f, stream);
    if (r != BTS_OK) return r;
    hipLaunchKernelGGL(lp_bias_grad_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, cs, db, N, Cout, accumulate);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}
// The weight gradients of the TWO convolutions that read a ResnetBlock's input (resnet.py:134 conv1, 3x3x3; resnet.py:118 shortcut, 1x1x1)
// from ONE pass over that input: dw3[t][c][k] (+)= sum_v x[v + off_t][c] dy3[v][k] and dw1[c][k] (+)= sum_v x[v][c] dy1[v][k].  The streaming
// weight-gradient kernel (lowp_wgd.hip) holds a P fragment of the centre tap anyway; the shortcut's gradient is one more accumulator per
// wave fed by planes of dy1 that ride in the Q-ring slots the 3x3x3 contraction no longer needs (its older planes live in registers).
// The 1x1x1 weight-gradient launch -- HBM-bound, its own read of the Cin-wide x -- goes away.  db3 (may be NULL; dy3 dense then) (+)= sum dy3.
// Same conventions as bts_lp_conv3d_bwd_weight (dup_start / dup_shift fold both kernels alike).  The workspace query returns -1 and the
// call 1 (nothing launched) where the streaming kernel does not take the shape: run bts_lp_conv3d_bwd_weight twice.  BTS_LP_WPAIR=0: never.
extern "C" long bts_lp_conv3d_bwd_weight_pair_workspace(int N, int D, int H, int W, int Cin, int Cout) {
  static const bool off = [] { const char* e = getenv("BTS_LP_WPAIR"); return e && atoi(e) == 0; }();
  if (off || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin % 8 != 0 || Cout % 8 != 0) return -1;
  const long alt = bts_lp_wgd_workspace_(N, D, H, W, Cin, Cout);
  if (alt <= 0) return -1;
  const long part = ((alt / 27 * 28 + 255) & ~255L);
  return part + (long)N * ((Cout + 7) / 8 * 8) * 4 + bts_lp_colsum_workspace(N, (long)D * H * W, (Cout + 7) / 8 * 8) + 512;
}
extern "C" int bts_lp_conv3d_bwd_weight_pair(int dtype, const void* x, long x_split, const void* dy3, const void* dy1, float* dw3, float* dw1,
                                             float* db3, void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx,
                                             int Cout, int lddy3, int lddy1, int dup_start, int dup_shift, int accumulate, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  const long need = bts_lp_conv3d_bwd_weight_pair_workspace(N, D, H, W, Cin, Cout);
  if (need < 0) return 1;
  if (x == nullptr || dy3 == nullptr || dy1 == nullptr || dw3 == nullptr || dw1 == nullptr) return BTS_ERR_ALIGN;
  if (ldx % 8 != 0 || lddy3 % 8 != 0 || lddy1 % 8 != 0 || ldx < (x_split ? 32 : Cin) || lddy3 < Cout || lddy1 < Cout) return BTS_ERR_SHAPE;
  if (x_split != 0 && (x_split < 0 || x_split % 8 != 0 || Cin % 32 != 0 || dup_shift != 0)) return BTS_ERR_SHAPE;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dy3) & 15) || (((uintptr_t)dy1) & 15) || (((uintptr_t)workspace) & 15)) return BTS_ERR_ALIGN;
  if (dup_shift < 0 || (dup_shift > 0 && dup_start + dup_shift > Cin)) return BTS_ERR_SHAPE;
  if (workspace == nullptr || workspace_bytes < need) return BTS_ERR_WORKSPACE;
  if (db3 != nullptr && lddy3 != Cout) return BTS_ERR_UNSUPPORTED;
  const long part_bytes = ((bts_lp_wgd_workspace_(N, D, H, W, Cin, Cout) / 27 * 28 + 255) & ~255L);
  const int r = bts_lp_wgd_launch_(dtype, x, dy3, dw3, workspace, part_bytes, N, D, H, W, Cin, ldx, Cout, lddy3, dup_start, dup_shift, accumulate, stream,
                                   nullptr, dy1, dw1, lddy1, x_split);
  if (r != BTS_OK) return r;      // (1: declined, nothing launched)
  if (db3 != nullptr) {
    char* wsb = reinterpret_cast<char*>(workspace) + part_bytes;
    float* cs = reinterpret_cast<float*>(wsb);
    void* cws = wsb + (((long)N * Cout * 4 + 255) & ~255L);
    const int r2 = bts_lp_colsum(dtype, dy3, cs, cws, bts_lp_colsum_workspace(N, (long)D * H * W, Cout), N, (long)D * H * W, Cout, 1.0f, stream);
    if (r2 != BTS_OK) return r2;
    hipLaunchKernelGGL(lp_bias_grad_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, cs, db3, N, Cout, accumulate);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}
// conv2 of a ResnetBlock in TRAINING without the normalised tensor: conv2's forward reads conv1's raw output through
// bts_lp_conv3d_gnin_fwd_gn, and its weight gradient dW[t][c][k] = sum_v a[v + off_t][c] dy[v][k] with a = relu(GN1(x)) is taken from the
// raw x the same way -- the streaming weight-gradient kernel normalises its P planes in LDS (lowp_wgd.hip, GNA).  a = relu(GN1(c1))
// (resnet.py:133-134) is then never written: one 1 read + 1 write pass and one activation-sized tensor per block less.
// bts_lp_conv3d_gnin_train_ok: 1 when BOTH kernels take the shape in this form (ask before the forward), else 0.
extern "C" int bts_lp_conv3d_gnin_train_ok(int N, int D, int H, int W, int Cin, int Cout, int in_G, int G) {
  if (bts_lp_conv3d_gnin_fwd_gn_workspace(N, D, H, W, Cin, Cout, in_G, G) < 0) return 0;
  return bts_lp_wgd_gna_ok_(N, D, H, W, Cin, Cout, in_G) ? 1 : 0;
}
// x: the RAW GroupNorm input, dense (N,D,H,W,Cin); in_*: that GroupNorm's parameters and statistics (slab mode; ReLU follows it);
// dy (N,D,H,W,Cout) rows of lddy; dw (3,3,3,Cin,Cout) fp32 (+)=; db (may be NULL) (+)= column sums of dy (dense dy then).  Workspace:
// bts_lp_conv3d_bwd_weight_workspace(BTS_CONV_K3S1, ...).  BTS_ERR_UNSUPPORTED where bts_lp_conv3d_gnin_train_ok says 0.
extern "C" int bts_lp_conv3d_gnin_bwd_weight(int dtype, const void* x, const float* in_gamma, const float* in_beta, const float* in_mean,
                                             const float* in_rstd, int in_G, const void* dy, float* dw, float* db, void* workspace,
                                             long workspace_bytes, int N, int D, int H, int W, int Cin, int Cout, int lddy, int accumulate,
                                             hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin % 8 != 0 || Cout % 8 != 0 || lddy % 8 != 0 || lddy < Cout) return BTS_ERR_SHAPE;
  if (!bts_lp_wgd_gna_ok_(N, D, H, W, Cin, Cout, in_G)) return BTS_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)dy) & 15) || (((uintptr_t)workspace) & 15)) return BTS_ERR_ALIGN;
  if (workspace_bytes < bts_lp_conv3d_bwd_weight_workspace(BTS_CONV_K3S1, N, D, H, W, Cin, Cout)) return BTS_ERR_WORKSPACE;
  const long part_bytes = bts_lp_wgd_workspace_(N, D, H, W, Cin, Cout);
  LpGnaFuse ga{in_gamma, in_beta, in_mean, in_rstd, in_G, Cin / in_G};
  const int r = bts_lp_wgd_launch_(dtype, x, dy, dw, workspace, part_bytes, N, D, H, W, Cin, Cin, Cout, lddy, 0, 0, accumulate, stream, &ga);
  if (r == 1) return BTS_ERR_UNSUPPORTED;
  if (r != BTS_OK) return r;
  if (db != nullptr) {
    if (lddy != Cout) return BTS_ERR_UNSUPPORTED;
    char* wsb = reinterpret_cast<char*>(workspace) + ((part_bytes + 255) & ~255L);
    float* cs = reinterpret_cast<float*>(wsb);
    void* cws = wsb + (((long)N * Cout * 4 + 255) & ~255L);
    const int r2 = bts_lp_colsum(dtype, dy, cs, cws, bts_lp_colsum_workspace(N, (long)D * H * W, Cout), N, (long)D * H * W, Cout, 1.0f, stream);
    if (r2 != BTS_OK) return r2;
    hipLaunchKernelGGL(lp_bias_grad_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, cs, db, N, Cout, accumulate);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}

// =====================================================================================================================
// Gate (squeeze-excitation) backward on 16-bit tensors (resnet.py:121-130 under TF autodiff): with g = dout * res per element,
//   t_v = sum_c g, ds_v = t_v sp_v (1 - sp_v)            (spatial gate, fp32 per voxel)
//   Pch[n][c] = sum_v g, Pw[c] = sum_v ds_v res          (per-block partials, fp64, layout of se.hip; stages 2a / 2 are se.hip's)
//   dres = dout (sp + ch) + ds w_sp + dgap / V           (written in the storage type)
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void lp_se_bwd_reduce_kernel(const unsigned short* dout, const unsigned short* res, const float* sp,
                                                               float* ds_out, double* partial, long V, int F, int lddo, long vspan) {
  __shared__ double sh[256 * 16];
  const int F8 = F >> 3;
  const int vpb = 256 / F8;
  const int lg = threadIdx.x % F8, vl = threadIdx.x / F8;
  const int c = lg * 8;
  const long n = blockIdx.y;
  const long v0 = (long)blockIdx.x * vspan;
  long v1 = v0 + vspan;
  if (v1 > V) v1 = V;
  double a[8], b[8];
  float fa[8], fb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = b[e] = 0.0; fa[e] = fb[e] = 0.f; }
  int cnt = 0;
  for (long vv = v0 + vl; vv < v1; vv += vpb) {
    const long v = n * V + vv;
    float r[8], d[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(res + v * F + c), r);
    unpack8<T>(*reinterpret_cast<const u32x4*>(dout + v * lddo + c), d);
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t = fmaf(d[e], r[e], t);
    for (int o = F8 >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const float s = sp[v];
    const float dsv = t * s * (1.f - s);
    if (lg == 0) ds_out[v] = dsv;
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa[e] = fmaf(d[e], r[e], fa[e]); fb[e] = fmaf(dsv, r[e], fb[e]); }
    if (++cnt == 64) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] += fa[e]; b[e] += fb[e]; fa[e] = fb[e] = 0.f; }
      cnt = 0;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { sh[threadIdx.x * 16 + e] = a[e] + fa[e]; sh[threadIdx.x * 16 + 8 + e] = b[e] + fb[e]; }
  __syncthreads();
  for (int col = threadIdx.x; col < F; col += 256) {
    const int e = col & 7, l = col >> 3;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < vpb; ++k) { sa += sh[(k * F8 + l) * 16 + e]; sb += sh[(k * F8 + l) * 16 + 8 + e]; }
    const long o = (((long)n * gridDim.x + blockIdx.x) * F + col) * 2;
    partial[o] = sa;
    partial[o + 1] = sb;
  }
}
// =====================================================================================================================
// A ResnetBlock's gate backward and GroupNorm-2 backward in ONE pair of passes (resnet.py:121-137 under TF autodiff).  Both read the
// gradient of the block output: the separate routes (bts_lp_se_bwd, bts_lp_gn_bwd) read it four times and launch ten kernels; here
// the reduce pass reads dout, res, c2 once (GroupNorm class sums + gate sums + the per-voxel spatial-gate gradient) and the apply pass
// reads dout, c2 once and writes dres and dc2 (+ both bias-gradient rows).  Thread mapping of the GroupNorm kernels: a workgroup owns a
// span of one (n, group) unit, a thread 8 consecutive channels of a voxel, the C/8 lanes of a voxel are neighbours.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void lp_blk_bwd_reduce_kernel(const unsigned short* x, const unsigned short* dy, const unsigned short* res,
                                                                const float* sp, const float* gamma, const float* beta, const float* mean,
                                                                const float* rstd, double* partial, double* se_partial, float* ds_out, long E,
                                                                long L, long span, int C, int G, int cg, int lddy) {
  __shared__ double sh[4 * 4 * 16];
  __shared__ double sh2[256 * 16];
  const int unit = blockIdx.y, n = unit / G, g = unit % G;
  const long lo = (long)blockIdx.x * span;
  long hi = lo + span;
  if (hi > L) hi = L;
  const long ubase = (long)g * L;
  const int cph = (int)((ubase + lo + threadIdx.x * 8L) % C);
  const int F8 = C >> 3;
  float gam[8], bet[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int idx = g * cg + ((cph + e) % cg); gam[e] = gamma[idx]; bet[e] = beta[idx]; }
  const float m = mean[unit], rs = rstd[unit];
  double a[8], b[8], pa[8], pb[8];
  float fa[8], fb[8], qa[8], qb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = b[e] = pa[e] = pb[e] = 0.0; fa[e] = fb[e] = qa[e] = qb[e] = 0.f; }
  const unsigned short* xb = x + (long)n * E + ubase;
  const unsigned short* rb = res + (long)n * E + ubase;
  const long pstep = 2048 / C, pix0 = ((long)n * E + ubase + lo + threadIdx.x * 8L) / C;
  int cnt = 0;
  // (hi - lo is a multiple of 2048: the lanes of a voxel leave together.)  Two steps per trip, their six loads issued first.
  auto one = [&](const u32x4 rv, const u32x4 rr, const u32x4 rd, float s, long pix) {
    float v[8], d[8], r[8];
    unpack8<T>(rv, v);
    unpack8<T>(rr, r);
    unpack8<T>(rd, d);
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t = fmaf(d[e], r[e], t);
    for (int o = F8 >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const float dsv = t * s * (1.f - s);
    if ((threadIdx.x & (F8 - 1)) == 0) ds_out[pix] = dsv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - m) * rs;
      float de = d[e];
      if (!(xh * gam[e] + bet[e] > 0.f)) de = 0.f;
      fa[e] = fmaf(de, xh, fa[e]);
      fb[e] += de;
      qa[e] = fmaf(d[e], r[e], qa[e]);
      qb[e] = fmaf(dsv, r[e], qb[e]);
    }
    if (++cnt == 32) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] += fa[e]; b[e] += fb[e]; pa[e] += qa[e]; pb[e] += qb[e]; fa[e] = fb[e] = qa[e] = qb[e] = 0.f; }
      cnt = 0;
    }
  };
  {
    const long K = (hi - lo) / 2048;
    const long i0 = lo + threadIdx.x * 8L;
    long k = 0;
    for (; k + 2 <= K; k += 2) {
      u32x4 rv[2], rr[2], rd[2];
      float s[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long i = i0 + (k + j) * 2048, pix = pix0 + (k + j) * pstep;
        rv[j] = *reinterpret_cast<const u32x4*>(xb + i);
        rr[j] = *reinterpret_cast<const u32x4*>(rb + i);
        rd[j] = *reinterpret_cast<const u32x4*>(dy + pix * lddy + cph);
        s[j] = sp[pix];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) one(rv[j], rr[j], rd[j], s[j], pix0 + (k + j) * pstep);
    }
    for (; k < K; ++k) {
      const long i = i0 + k * 2048, pix = pix0 + k * pstep;
      one(*reinterpret_cast<const u32x4*>(xb + i), *reinterpret_cast<const u32x4*>(rb + i), *reinterpret_cast<const u32x4*>(dy + pix * lddy + cph), sp[pix], pix);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] += fa[e]; b[e] += fb[e]; pa[e] += qa[e]; pb[e] += qb[e]; }
  // gate sums first (their LDS array is separate): threads of equal octet (t mod F8) in thread order
#pragma unroll
  for (int e = 0; e < 8; ++e) { sh2[threadIdx.x * 16 + e] = pa[e]; sh2[threadIdx.x * 16 + 8 + e] = pb[e]; }
  // GroupNorm class sums exactly as lp_gn_bwd_reduce_kernel
  const int p = cg > 8 ? cg / 8 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 32; off >= p; off >>= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] += __shfl_xor(a[e], off, 64); b[e] += __shfl_xor(b[e], off, 64); }
  }
  if (lane < p) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[((wave * 4 + lane) * 8 + e) * 2] = a[e]; sh[((wave * 4 + lane) * 8 + e) * 2 + 1] = b[e]; }
  }
  __syncthreads();
  if (threadIdx.x < cg) {
    const int j = threadIdx.x;
    const int cph0 = (int)((ubase + lo) % C);
    double sa = 0.0, sb = 0.0;
    for (int w = 0; w < 4; ++w)
      for (int q = 0; q < p; ++q)
        for (int e = 0; e < 8; ++e)
          if (((cph0 + 8 * q + e) % cg) == j) { sa += sh[((w * 4 + q) * 8 + e) * 2]; sb += sh[((w * 4 + q) * 8 + e) * 2 + 1]; }
    double* o = partial + (((long)unit * gridDim.x + blockIdx.x) * cg + j) * 2;
    o[0] = sa; o[1] = sb;
  }
  const int vpb = 256 / F8;
  for (int col = threadIdx.x; col < C; col += 256) {
    const int e = col & 7, l = col >> 3;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < vpb; ++k) { sa += sh2[(k * F8 + l) * 16 + e]; sb += sh2[(k * F8 + l) * 16 + 8 + e]; }
    const long o = ((((long)n * G + g) * gridDim.x + blockIdx.x) * C + col) * 2;      // block index inside the sample: g * B + b
    se_partial[o] = sa;
    se_partial[o + 1] = sb;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void lp_blk_bwd_apply_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                               unsigned short* __restrict__ dx, unsigned short* __restrict__ dres,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ c1, const float* __restrict__ c2,
                                                               const float* __restrict__ sp, const float* __restrict__ ds,
                                                               const float* __restrict__ ch, const float* __restrict__ wsp,
                                                               const float* __restrict__ dgap, long L, long per, int C, int G, int cg, int lddy,
                                                               double* dbias_c2, double* dbias_pt) {
  __shared__ float dbsh[256 * 8];
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int unit = blockIdx.y, n = unit / G, g = unit - n * G;
  const long lo = (long)unit * L;
  const long a = lo + (long)blockIdx.x * per;
  const long bnd = (a + per < lo + L) ? a + per : lo + L;
  const int t8 = threadIdx.x * 8, c = t8 % C;
  float gam[8], bet[8], chv[8], wsv[8], dgv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int idx = g * cg + ((c + e) % cg);
    gam[e] = gamma[idx]; bet[e] = beta[idx];
    chv[e] = ch[n * C + c + e]; wsv[e] = wsp[c + e]; dgv[e] = dgap[n * C + c + e];
  }
  const float m = mean[unit], rs = rstd[unit], k1 = c1[unit], k2 = c2[unit];
  const long pstep = 2048 / C, pix = a / C + t8 / C;
  const long K = (bnd - a) / 2048;
  auto one = [&](const u32x4 rx, const u32x4 rd, float s, float dsv, long i) {
    float v[8], d[8], o[8], q[8];
    unpack8<T>(rx, v);
    unpack8<T>(rd, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - m) * rs;
      float de = d[e];
      if (!(xh * gam[e] + bet[e] > 0.f)) de = 0.f;
      o[e] = (de * gam[e] - k1 - xh * k2) * rs;
      cs[e] += o[e];
      q[e] = fmaf(d[e], s + chv[e], fmaf(dsv, wsv[e], dgv[e]));
      cr[e] += q[e];
    }
    *reinterpret_cast<u32x4*>(dx + i) = pack8<T>(o);
    *reinterpret_cast<u32x4*>(dres + i) = pack8<T>(q);
  };
  long k = 0;
  for (; k + 2 <= K; k += 2) {
    u32x4 rx[2], rd[2];
    float s[2], dv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long px = pix + (k + j) * pstep;
      rx[j] = *reinterpret_cast<const u32x4*>(x + a + t8 + (k + j) * 2048);
      rd[j] = *reinterpret_cast<const u32x4*>(dy + px * lddy + c);
      s[j] = sp[px]; dv[j] = ds[px];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) one(rx[j], rd[j], s[j], dv[j], a + t8 + (k + j) * 2048);
  }
  for (; k < K; ++k) {
    const long px = pix + k * pstep;
    one(*reinterpret_cast<const u32x4*>(x + a + t8 + k * 2048), *reinterpret_cast<const u32x4*>(dy + px * lddy + c), sp[px], ds[px], a + t8 + k * 2048);
  }
  const long row = (long)blockIdx.y * gridDim.x + blockIdx.x;
  if (dbias_c2 != nullptr) lp_dbias_block(cs, 0, C / 8, dbias_c2 + row * C, dbsh);     // (launch-uniform)
  if (dbias_pt != nullptr) lp_dbias_block(cr, 0, C / 8, dbias_pt + row * C, dbsh);
}
int bts_se_bwd_middle_(double* partial, double* red, double* scratch, const float* gap, const float* h, const float* ch, const float* w1,
                       const float* w2, float* dw1, float* dw2, float* dwsp, float* dgap, int N, int B, long V, int F, int R,
                       int accumulate_params, hipStream_t stream);
static int lp_gnb_blocks(long L);
static int lp_se_blocks(long V, int N, int F, long* vspan);
extern "C" long bts_lp_block_bwd_workspace(int N, long V, int F, int R, int G) {
  if (N <= 0 || V <= 0 || F < 8 || R <= 0 || G <= 0 || F % G != 0) return -1;
  const long L = V * F / G;
  const long B = lp_gnb_blocks(L);
  long vspan;
  long Bse = lp_se_blocks(V, N, F, &vspan);      // (the A/B route with separate reduce passes uses the gate kernel's own block count)
  if (Bse < G * B) Bse = G * B;
  return (long)N * G * B * (F / G) * 2 * 8 + (long)N * G * 2 * 4 + 64      // GroupNorm partials, c1 / c2
         + (long)N * Bse * F * 2 * 8 + ((long)N * F * 3 + (long)N * R) * 8 + 128      // gate partials, red, scratch
         + 2 * ((N * (long)G > 2048 ? N * (long)G : 2048L) * F * 8 + 64);      // two sets of bias-gradient rows (one per apply workgroup)
}
// The two small launches of the fused block backward (see bts_lp_block_bwd).  Middle: workgroups [0, G) finish GroupNorm-2's class sums
// (lp_gn_bwd_finalize_kernel's work), the rest sum the gate partials per (n, c) (se_bwd_partial_reduce_kernel's).  Tail: workgroups [0, np)
// form the SE-MLP's parameter gradients from the per-sample pass's scratch (se_mlp_bwd_param_kernel's work), the next F finish conv2's bias
// gradient, the last F the shortcut conv's (lp_dbias_finalize_kernel's).  Same bodies, same summation order: bit-identical to the six launches.
__global__ __launch_bounds__(256) void lp_blk_bwd_middle_kernel(const double* partial, const float* gamma, float* dgamma, float* dbeta, float* c1,
                                                                float* c2, const double* se_partial, double* red, int N, int G, int B, int cg,
                                                                double L, int Bse, int F) {
  __shared__ double sh[256 * 2];
  __shared__ double tot[256 * 2];
  if ((int)blockIdx.x < G) lp_gn_bwd_finalize_body(partial, gamma, dgamma, dbeta, c1, c2, N, G, B, cg, L, 1, blockIdx.x, sh, tot);
  else se_bwd_partial_reduce_body(se_partial, red, N, Bse, F, (int)blockIdx.x - G);
}
__global__ __launch_bounds__(256) void lp_blk_bwd_tail_kernel(const double* red, const float* gap, const float* hbuf, float* dw1, float* dw2, float* dwsp,
                                                              const double* scratch, int N, int F, int R, int np, const double* dbp1, float* db1,
                                                              const double* dbp2, float* db2, int nblocks) {
  __shared__ double sh[4];
  int b = blockIdx.x;
  if (b < np) { se_mlp_bwd_param_body(red, gap, hbuf, dw1, dw2, dwsp, scratch, N, F, R, 1, b); return; }
  b -= np;
  if (db1 != nullptr) {
    if (b < F) { lp_dbias_finalize_body(dbp1, db1, nblocks, F, 1, b, sh); return; }
    b -= F;
  }
  lp_dbias_finalize_body(dbp2, db2, nblocks, F, 1, b, sh);
}
// dout (N,V,F) rows of lddo; res, c2 dense; dres, dc2 dense outputs in the storage type; ds (N*V) and dgap (N,F) fp32 scratch outputs;
// parameter gradients accumulate (+=); dbias_pt / dbias_c2 (may be NULL): the shortcut conv's / conv2's bias gradients (+=).
// BTS_ERR_UNSUPPORTED outside the kernels' tiling: the caller runs bts_lp_gn_bwd and bts_lp_se_bwd.
extern "C" int bts_lp_block_bwd(int dtype, const void* dout, int lddo, const void* res, const void* c2x, const float* sp, const float* gap,
                                const float* h, const float* ch, const float* w1, const float* w2, const float* wsp, const float* gamma,
                                const float* beta, const float* mean, const float* rstd, void* dres, void* dc2, float* ds, float* dgap, float* dw1,
                                float* dw2, float* dwsp, float* dgamma, float* dbeta, float* dbias_pt, float* dbias_c2, void* workspace,
                                long workspace_bytes, int N, long V, int F, int R, int G, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || F < 8 || F < G || F % G != 0 || (F & (F - 1)) != 0 || F > 256 || lddo % 8 != 0 || lddo < F || R <= 0) return BTS_ERR_SHAPE;
  const long E = V * F, L = E / G;
  const int cg = F / G;
  if (E % G != 0 || L % 2048 != 0 || cg > 32 || 256 % cg != 0) return BTS_ERR_UNSUPPORTED;
  if ((((uintptr_t)dout) & 15) || (((uintptr_t)res) & 15) || (((uintptr_t)c2x) & 15) || (((uintptr_t)dres) & 15) || (((uintptr_t)dc2) & 15)) return BTS_ERR_ALIGN;
  if (workspace == nullptr || (((uintptr_t)workspace) & 15) || workspace_bytes < bts_lp_block_bwd_workspace(N, V, F, R, G)) return BTS_ERR_WORKSPACE;
  const int B = lp_gnb_blocks(L);
  const long span = ((L / 2048 + B - 1) / B) * 2048;
  double* partial = reinterpret_cast<double*>(workspace);
  float* c1 = reinterpret_cast<float*>(partial + (long)N * G * B * cg * 2);
  float* c2 = c1 + (long)N * G;
  double* sep = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(c2 + (long)N * G) + 63) & ~(uintptr_t)63);
  long vspan_se;
  long Bmax = lp_se_blocks(V, N, F, &vspan_se);
  if (Bmax < (long)G * B) Bmax = (long)G * B;
  double* red = sep + (long)N * Bmax * F * 2;
  double* scratch = red + (long)N * F * 2;
  double* dbp1 = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(scratch + (long)N * F + (long)N * R) + 63) & ~(uintptr_t)63);
  double* dbp2 = dbp1 + (N * (long)G > 2048 ? N * (long)G : 2048L) * F + 8;
  (void)hipGetLastError();
  // BTS_LP_FUSE_BLOCK_BWD_REDUCE=0 (A/B): the two light reduce passes of the separate routes, then the fused apply pass
  const char* fr = getenv("BTS_LP_FUSE_BLOCK_BWD_REDUCE");
  int Bse = G * B;
  if (fr && atoi(fr) == 0) {
    long vspan;
    Bse = lp_se_blocks(V, N, F, &vspan);
    if (dtype == LP_F16) {
      hipLaunchKernelGGL(lp_gn_bwd_reduce_kernel<TF16>, dim3(B, N * G), dim3(256), 0, stream, (const unsigned short*)c2x, (const unsigned short*)dout, gamma, beta, mean, rstd, partial, E, L, span, F, G, cg, lddo, 1);
      hipLaunchKernelGGL(lp_se_bwd_reduce_kernel<TF16>, dim3(Bse, N), dim3(256), 0, stream, (const unsigned short*)dout, (const unsigned short*)res, sp, ds, sep, V, F, lddo, vspan);
    } else {
      hipLaunchKernelGGL(lp_gn_bwd_reduce_kernel<TBF16>, dim3(B, N * G), dim3(256), 0, stream, (const unsigned short*)c2x, (const unsigned short*)dout, gamma, beta, mean, rstd, partial, E, L, span, F, G, cg, lddo, 1);
      hipLaunchKernelGGL(lp_se_bwd_reduce_kernel<TBF16>, dim3(Bse, N), dim3(256), 0, stream, (const unsigned short*)dout, (const unsigned short*)res, sp, ds, sep, V, F, lddo, vspan);
    }
  } else {
#define LP_BB_R(TT) hipLaunchKernelGGL(lp_blk_bwd_reduce_kernel<TT>, dim3(B, N * G), dim3(256), 0, stream, (const unsigned short*)c2x, (const unsigned short*)dout, (const unsigned short*)res, sp, gamma, beta, mean, rstd, partial, sep, ds, E, L, span, F, G, cg, lddo)
    if (dtype == LP_F16) LP_BB_R(TF16); else LP_BB_R(TBF16);
#undef LP_BB_R
  }
  BTS_LAUNCH_CHECK();
  // Between the reduce and the apply pass (review: ~100 launches of 5-10 us on this chain per step): ONE launch for the two finalizes that
  // only need the reduce pass's partials (GroupNorm class sums -> dgamma / dbeta / c1 / c2; gate partials -> per-(n, c) sums), then the
  // per-sample SE-MLP backward (dgap: the apply pass needs it).  The sums over the samples (dW1, dW2, dw_sp) and the two bias-gradient
  // finalizes wait for the tail launch behind the apply pass: nothing on the chain reads them.  BTS_LP_BLK_BWD_MERGE=0: the six launches (A/B).
  static const bool merge = [] { const char* e = getenv("BTS_LP_BLK_BWD_MERGE"); return !(e && atoi(e) == 0); }();
  if (merge) {
    hipLaunchKernelGGL(lp_blk_bwd_middle_kernel, dim3(G + (N * F + 3) / 4), dim3(256), 0, stream, partial, gamma, dgamma, dbeta, c1, c2, sep, red, N, G, B, cg,
                       (double)L, Bse, F);
    BTS_LAUNCH_CHECK();
    const int r = bts_se_mlp_bwd_sample_(red, scratch, h, ch, w1, w2, dgap, N, V, F, R, stream);
    if (r != BTS_OK) return r;
  } else {
    hipLaunchKernelGGL(lp_gn_bwd_finalize_kernel, dim3(G), dim3(256), 0, stream, partial, gamma, dgamma, dbeta, c1, c2, N, G, B, cg, (double)L, 1);
    BTS_LAUNCH_CHECK();
    const int r = bts_se_bwd_middle_(sep, red, scratch, gap, h, ch, w1, w2, dw1, dw2, dwsp, dgap, N, Bse, V, F, R, 1, stream);
    if (r != BTS_OK) return r;
  }
  int Ba;
  const long per = lp_chunk_per(L, (long)N * G, &Ba, 2048);
  const long blocks = (long)N * G * Ba;
#define LP_BB_A(TT) hipLaunchKernelGGL(lp_blk_bwd_apply_kernel<TT>, dim3((unsigned)Ba, (unsigned)(N * G)), dim3(256), 0, stream, (const unsigned short*)c2x, (const unsigned short*)dout, (unsigned short*)dc2, (unsigned short*)dres, gamma, beta, mean, rstd, c1, c2, sp, ds, ch, wsp, dgap, L, per, F, G, cg, lddo, dbias_c2 ? dbp1 : (double*)nullptr, dbias_pt ? dbp2 : (double*)nullptr)
  if (dtype == LP_F16) LP_BB_A(TF16); else LP_BB_A(TBF16);
#undef LP_BB_A
  BTS_LAUNCH_CHECK();
  if (merge) {
    const int np = (R * F + 255) / 256;
    hipLaunchKernelGGL(lp_blk_bwd_tail_kernel, dim3(np + (dbias_c2 ? F : 0) + (dbias_pt ? F : 0)), dim3(256), 0, stream, red, gap, h, dw1, dw2, dwsp, scratch,
                       N, F, R, np, dbias_c2 ? dbp1 : (const double*)nullptr, dbias_c2, dbias_pt ? dbp2 : (const double*)nullptr, dbias_pt, (int)blocks);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
  }
  if (dbias_c2 != nullptr) {
    hipLaunchKernelGGL(lp_dbias_finalize_kernel, dim3(F), dim3(256), 0, stream, dbp1, dbias_c2, (int)blocks, F, 1);
    BTS_LAUNCH_CHECK();
  }
  if (dbias_pt != nullptr) {
    hipLaunchKernelGGL(lp_dbias_finalize_kernel, dim3(F), dim3(256), 0, stream, dbp2, dbias_pt, (int)blocks, F, 1);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void lp_se_bwd_apply_kernel(const unsigned short* dout, const float* sp, const float* ds, const float* ch,
                                                              const float* wsp, const float* dgap, unsigned short* dres, long NV, long V, int F,
                                                              int lddo, double* dbias_part) {
  __shared__ float dbsh[256 * 8];
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int F8 = F >> 3;
  const long total = NV * F8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long v = i / F8;
    const int c = (int)(i - v * F8) * 8;
    const long n = v / V;
    float d[8], o[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(dout + v * lddo + c), d);
    const float s = sp[v], dsv = ds[v];
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = fmaf(d[e], s + ch[n * F + c + e], fmaf(dsv, wsp[c + e], dgap[n * F + c + e])); cs[e] += o[e]; }
    *reinterpret_cast<u32x4*>(dres + v * F + c) = pack8<T>(o);
  }
  if (dbias_part != nullptr)     // (launch-uniform; 256 % (F/8) == 0: thread t always holds octet t mod F/8)
    lp_dbias_block(cs, 0, F8, dbias_part + (long)blockIdx.x * F, dbsh);
}
static int lp_se_blocks(long V, int N, int F, long* vspan) {
  const int vpb = 256 / (F / 8);
  long B = (1024 + N - 1) / N;
  if (B > 512) B = 512;
  long span = (V + B - 1) / B;
  span = (span + vpb - 1) / vpb * vpb;
  *vspan = span;
  return (int)((V + span - 1) / span);
}
extern "C" long bts_lp_se_bwd_workspace(int N, long V, int F, int R) {
  if (N <= 0 || V <= 0 || F < 8 || R <= 0) return -1;
  long vspan;
  const int B = lp_se_blocks(V, N, F, &vspan);
  return (long)N * B * F * 2 * 8 + ((long)N * F * 3 + (long)N * R) * 8 + 128 + 16384L * F * 8 + 64;   // (+ bias-gradient rows)
}
// dout rows of stride lddo, res dense, both in the storage type; sp fp32 [N*V] (bts_lp_block_epilogue's sp_out); gap / h / ch from the forward.
// Outputs: dres dense in the storage type, ds [N*V] and dgap [N*F] fp32 scratch, parameter gradients fp32 (+= when accumulate_params).
extern "C" int bts_lp_se_bwd(int dtype, const void* dout, const void* res, const float* sp, const float* gap, const float* h, const float* ch,
                             const float* w1, const float* w2, const float* wsp, void* dres, float* ds, float* dgap, float* dw1, float* dw2,
                             float* dwsp, void* workspace, long workspace_bytes, int N, long V, int F, int R, int lddo, int accumulate_params,
                             float* dbias, hipStream_t stream) {
  if (dtype != LP_F16 && dtype != LP_BF16) return BTS_ERR_UNSUPPORTED;
  if (N <= 0 || V <= 0 || F < 8 || (F & (F - 1)) || F > 256 || lddo < F || lddo % 8) return BTS_ERR_SHAPE;
  if ((((uintptr_t)dout) & 15) || (((uintptr_t)res) & 15) || (((uintptr_t)dres) & 15)) return BTS_ERR_ALIGN;
  if (workspace_bytes < bts_lp_se_bwd_workspace(N, V, F, R)) return BTS_ERR_WORKSPACE;
  long vspan;
  const int B = lp_se_blocks(V, N, F, &vspan);
  double* partial = reinterpret_cast<double*>(workspace);
  double* red = partial + (long)N * B * F * 2;
  double* scratch = red + (long)N * F * 2;
  double* dbp = dbias ? reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(scratch + (long)N * F + (long)N * R) + 63) & ~(uintptr_t)63) : nullptr;
  (void)hipGetLastError();
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_se_bwd_reduce_kernel<TF16>, dim3(B, N), dim3(256), 0, stream, (const unsigned short*)dout, (const unsigned short*)res, sp, ds, partial, V, F, lddo, vspan);
  else hipLaunchKernelGGL(lp_se_bwd_reduce_kernel<TBF16>, dim3(B, N), dim3(256), 0, stream, (const unsigned short*)dout, (const unsigned short*)res, sp, ds, partial, V, F, lddo, vspan);
  BTS_LAUNCH_CHECK();
  const int r = bts_se_bwd_middle_(partial, red, scratch, gap, h, ch, w1, w2, dw1, dw2, dwsp, dgap, N, B, V, F, R, accumulate_params, stream);
  if (r != BTS_OK) return r;
  const long total = (long)N * V * (F / 8);
  long blocks = (total + 255) / 256;
  const long cap = dbias ? 2048 : 16384;
  if (blocks > cap) blocks = cap;
  if (dtype == LP_F16) hipLaunchKernelGGL(lp_se_bwd_apply_kernel<TF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)dout, sp, ds, ch, wsp, dgap, (unsigned short*)dres, (long)N * V, V, F, lddo, dbp);
  else hipLaunchKernelGGL(lp_se_bwd_apply_kernel<TBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)dout, sp, ds, ch, wsp, dgap, (unsigned short*)dres, (long)N * V, V, F, lddo, dbp);
  BTS_LAUNCH_CHECK();
  if (dbias != nullptr) {      // bias gradient of the block's 1x1x1 shortcut conv (dres IS its dy)
    hipLaunchKernelGGL(lp_dbias_finalize_kernel, dim3(F), dim3(256), 0, stream, dbp, dbias, (int)blocks, F, accumulate_params);
    BTS_LAUNCH_CHECK();
  }
  return BTS_OK;
}
