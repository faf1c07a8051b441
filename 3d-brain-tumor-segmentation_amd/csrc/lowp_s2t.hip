// 3x3x3 stride-2 'same' convolution with 32 input and <= 32 output channels on 16-bit storage, LDS-tiled (round 3): the Conv3D of
// downsample.py:30-48 at the top level -- 2ch x 128^3 x batch (or 160x192x160) -> half the grid, the single most expensive launch of the
// gather kernels (lowp.hip: 1.03 ms of the batch-8 step, 0.34 ms of the inference forward, 1.2 TB/s of input).
//
// Why the gather form is slow there: every output voxel reads 27 input rows, neighbouring outputs share 2/3 of them, and nothing holds
// the shared rows but the 32 KB L1 -- each row comes from L2 27/8 times, in 32- or 64-byte pieces of 128-byte lines (the input is a
// 32-channel view of a 64-channel slab).  What an LDS tile needs is 8x the output voxels (stride 2 in three axes): far too much for the
// wide layers, but with 32 channels a 16x4x2 output tile reads 33x9x5 = 1485 voxels x 64 bytes = 95 KB, and the WHOLE weight set
// (27 taps x 32 x 32 x 2 bytes = 54 KB) fits next to it.  So:
//   * persistent workgroups (one per CU, 512 threads), the weight image copied into LDS once -- it is the gather kernel's image
//     [tap][k-step][k-half][32 couts][8 cin], an A fragment is one ds_read_b128 at (tap * 2 + ks) * 1024 + lane * 16;
//   * per tile every input row is fetched ONCE, as whole 64-byte rows (four lanes per voxel), twelve 16-byte loads per thread in
//     flight at a time; the next tile's loads are issued before the current tile multiplies;
//   * LDS layout [plane][row][x slot][4 pieces]: a row keeps its 17 even-x voxels first, then the 16 odd ones, so that the 16 output
//     voxels of a fragment row read consecutive slots for every x tap; piece p of slot s sits at p ^ ((s >> 2) & 3) -- 16 consecutive
//     slots x one piece cover all 64 banks once;
//   * wave = (fragment of 16 x 2 outputs, k-step): 27 matrix instructions each, the two k-steps of a fragment summed through LDS.
// Where it stands: 8 x 128^3 from a 64-channel slab in 0.44 ms (0.30 from a dense 32-channel input), 160x192x160 in 0.14 ms -- 2.3x the
// gather kernel.  Counters (profiles/r03_pmc_lp_s2t.txt): HBM reads 1.24 GB = 1.16x the input (the halo rows are L2 hits), so it is
// NOT at the HBM roofline (2.8 TB/s); matrix pipe busy 0.10; 15 vector + 9 scalar instructions per matrix instruction before the
// per-thread offsets / per-lane tap addresses / bias were hoisted out of the tile loop (0.50 -> 0.44 ms); LDS bank conflicts 40 % of the
// LDS-active cycles (the swizzle below is conflict-free for 16 CONSECUTIVE lanes; ds_read_b128's lane groups are not consecutive) at 6 %
// LDS utilisation.  Tried without effect: loads two tiles ahead (254 VGPRs, same time), an XCD-aware tile order (5 % slower).  What is
// left is the serial chain per tile -- store, barrier, 27 dependent matrix instructions, barrier, reduce, barrier -- of one workgroup per
// CU with nothing to overlap it with.
// Declines everything else (the caller keeps the gather kernels): Cin != 32, Cout > 32 or not a multiple of 4, odd input extents.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.h"
#include "bts_internal.h"
#include "lowp_common.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

struct LpS2tParams {
  const unsigned short* x;
  const unsigned short* wp;    // K3S2 forward image, NB = 1: [tap][k-step][k-half][32 couts][8 cin]
  const float* bias;
  unsigned short* y;
  int N, D, H, W, ldx;         // input grid (even extents)
  int Do, Ho, Wo, ldy, Cout;
  int ntx, nty, ntz;
  long ntiles;
  int accum;
};

constexpr int S2T_TX = 16, S2T_TY = 4, S2T_TZ = 2;
constexpr int S2T_SX = 2 * S2T_TX + 1, S2T_SY = 2 * S2T_TY + 1, S2T_SZ = 2 * S2T_TZ + 1;     // 33 x 9 x 5
constexpr int S2T_NVOX = S2T_SX * S2T_SY * S2T_SZ;                                          // 1485
constexpr int S2T_NPIECE = S2T_NVOX * 4;                                                    // 16-byte pieces
constexpr int S2T_ROUNDS = (S2T_NPIECE + 511) / 512;                                        // 12
constexpr int S2T_TILE_BYTES = S2T_NVOX * 64;                                               // 95040
constexpr int S2T_W_BYTES = 27 * 2 * 1024;                                                  // 55296
constexpr int S2T_OFF_W = ((S2T_TILE_BYTES + 1023) / 1024) * 1024;                          // 95232
constexpr int S2T_LDS = S2T_OFF_W + S2T_W_BYTES;                                            // 150528

template <typename T>
__global__ __launch_bounds__(512, 1) void lp_s2t_kernel(const LpS2tParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int frag = wave & 3, ks = wave >> 2;
  // ---- weights: verbatim copy of the image ----
  for (int i = tid; i < S2T_W_BYTES / 16; i += 512)
    *reinterpret_cast<u32x4*>(lds + S2T_OFF_W + i * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.wp) + i * 16);
  // ---- this thread's pieces of a tile: (voxel, piece) -> LDS byte address, input offset relative to the tile origin ----
  int ldst[S2T_ROUNDS];           // LDS byte address, or -1 past the tile
  int vzyx[S2T_ROUNDS];           // vz | vy << 8 | vx << 16
#pragma unroll
  for (int r = 0; r < S2T_ROUNDS; ++r) {
    const int q = r * 512 + tid;
    const int v = q >> 2, gp = q & 3;
    const int vz = v / (S2T_SY * S2T_SX), rem = v - vz * (S2T_SY * S2T_SX);
    const int vy = rem / S2T_SX, vx = rem - vy * S2T_SX;
    const int xslot = (vx & 1) ? (S2T_TX + 1) + (vx >> 1) : (vx >> 1);
    const int slot = (vz * S2T_SY + vy) * S2T_SX + xslot;
    ldst[r] = q < S2T_NPIECE ? slot * 64 + ((gp ^ ((slot >> 2) & 3)) * 16) : -1;
    vzyx[r] = vz | (vy << 8) | (vx << 16) | (gp << 24);
  }
  // ---- this lane's operand addressing: output voxel (oxl, oyl) of fragment `frag` ----
  const int oxl = l32 & 15, oyl = ((frag & 1) << 1) + (l32 >> 4), ozl = frag >> 1;
  const int base_slot = ((2 * ozl) * S2T_SY + 2 * oyl) * S2T_SX + oxl;
  const int piece = 2 * ks + h;
  const unsigned char* wbase = lds + S2T_OFF_W + ks * 1024 + lane * 16;

  auto tile_origin = [&](long t, int& n, int& oz0, int& oy0, int& ox0) {
    const int tx = (int)(t % p.ntx); t /= p.ntx;
    const int ty = (int)(t % p.nty); t /= p.nty;
    const int tz = (int)(t % p.ntz);
    n = (int)(t / p.ntz);
    ox0 = tx * S2T_TX; oy0 = ty * S2T_TY; oz0 = tz * S2T_TZ;
  };
  // element offset of every piece relative to the tile's first input voxel (fixed per thread; < 2^31 inside one sample)
  int goff[S2T_ROUNDS];
#pragma unroll
  for (int r = 0; r < S2T_ROUNDS; ++r) {
    const int vz = vzyx[r] & 0xff, vy = (vzyx[r] >> 8) & 0xff, vx = (vzyx[r] >> 16) & 0xff, gp = vzyx[r] >> 24;
    goff[r] = ((vz * p.H + vy) * p.W + vx) * p.ldx + gp * 8;
  }
  u32x4 pre[S2T_ROUNDS];
  auto prefetch = [&](long t) {
    int n, oz0, oy0, ox0;
    tile_origin(t, n, oz0, oy0, ox0);
    const unsigned short* org = p.x + ((((long)n * p.D + 2 * oz0) * p.H + 2 * oy0) * p.W + 2 * ox0) * (long)p.ldx;
    // (even extents: TF 'same' pads at the far end only.)  A tile whose 33x9x5 input box lies inside the volume -- all but the last
    // tile of each axis -- needs no per-piece test: one add per load
    const bool inside = 2 * oz0 + S2T_SZ <= p.D && 2 * oy0 + S2T_SY <= p.H && 2 * ox0 + S2T_SX <= p.W;      // (workgroup-uniform)
    if (inside) {
#pragma unroll
      for (int r = 0; r < S2T_ROUNDS; ++r) {
        pre[r] = u32x4{0u, 0u, 0u, 0u};
        if (ldst[r] >= 0) pre[r] = *reinterpret_cast<const u32x4*>(org + goff[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < S2T_ROUNDS; ++r) {
        const int vz = vzyx[r] & 0xff, vy = (vzyx[r] >> 8) & 0xff, vx = (vzyx[r] >> 16) & 0xff;
        pre[r] = u32x4{0u, 0u, 0u, 0u};
        if (ldst[r] >= 0 && 2 * oz0 + vz < p.D && 2 * oy0 + vy < p.H && 2 * ox0 + vx < p.W) pre[r] = *reinterpret_cast<const u32x4*>(org + goff[r]);
      }
    }
  };
  // operand addresses of the 27 taps and the bias: fixed per lane
  int boff[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    const int slot = base_slot + (dz * S2T_SY + dy) * S2T_SX + (dx == 1 ? S2T_TX + 1 : (dx >> 1));
    boff[tap] = slot * 64 + ((piece ^ ((slot >> 2) & 3)) * 16);
  }
  f32x16 acc0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = 8 * q + 4 * h + j;
      acc0[4 * q + j] = (ks == 0 && p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
    }
  long t = blockIdx.x;
  if (t >= p.ntiles) return;
  prefetch(t);
  for (; t < p.ntiles; t += gridDim.x) {
    __syncthreads();                       // the previous tile's partial sums have been read
#pragma unroll
    for (int r = 0; r < S2T_ROUNDS; ++r)
      if (ldst[r] >= 0) *reinterpret_cast<u32x4*>(lds + ldst[r]) = pre[r];
    __syncthreads();
    const long tn = t + gridDim.x;
    if (tn < p.ntiles) prefetch(tn);       // in flight while this tile multiplies
    f32x16 acc = acc0;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const u32x4 b = *reinterpret_cast<const u32x4*>(lds + boff[tap]);
      const u32x4 a = *reinterpret_cast<const u32x4*>(wbase + tap * 2048);
      acc = T::mfma(a, b, acc);
    }
    __syncthreads();                       // every wave is done with the tile: its memory carries the k-step-1 partial sums now
    float* part = reinterpret_cast<float*>(lds) + (frag * 64 + lane) * 16;
    if (ks == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(part + 4 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    }
    __syncthreads();
    if (ks == 0) {
      int n, oz0, oy0, ox0;
      tile_origin(t, n, oz0, oy0, ox0);
      const int oz = oz0 + ozl, oy = oy0 + oyl, ox = ox0 + oxl;
      const bool in = oz < p.Do && oy < p.Ho && ox < p.Wo;
      unsigned short* row = p.y + ((((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * (long)p.ldy;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(part + 4 * q);
        const int co = 8 * q + 4 * h;
        if (in && co < p.Cout) {       // (Cout % 4 == 0)
          float v0 = acc[4 * q] + o[0], v1 = acc[4 * q + 1] + o[1], v2 = acc[4 * q + 2] + o[2], v3 = acc[4 * q + 3] + o[3];
          unsigned short* dst = row + co;
          if (p.accum) {
            const u32x2 old = *reinterpret_cast<const u32x2*>(dst);
            v0 += T::ld((unsigned short)(old[0] & 0xffffu)); v1 += T::ld((unsigned short)(old[0] >> 16));
            v2 += T::ld((unsigned short)(old[1] & 0xffffu)); v3 += T::ld((unsigned short)(old[1] >> 16));
          }
          *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<T>(v0, v1), pack2<T>(v2, v3)};
        }
      }
    }
  }
}

static bool s2t_enabled() {   // BTS_LP_S2T=0: the layer back on the gather kernels (A/B; read per call)
  const char* e = getenv("BTS_LP_S2T");
  return !(e && atoi(e) == 0);
}
// BTS_OK = ran, 1 = declined.  Forward geometry only (TF 'same', stride 2, even input extents: no padding in front).
int bts_lp_s2t_launch_(int dtype, const void* x, const void* wp, const float* bias, void* y, int N, int D, int H, int W, int Cin, int ldx,
                       int Cout, int ldy, int accum, hipStream_t stream) {
  if (!s2t_enabled() || Cin != 32 || Cout > 32 || Cout % 4 != 0 || (D & 1) || (H & 1) || (W & 1)) return 1;
  if (ldx % 8 != 0 || ldy % 4 != 0 || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 7) || (((uintptr_t)wp) & 15)) return 1;
  LpS2tParams p;
  p.x = (const unsigned short*)x; p.wp = (const unsigned short*)wp; p.bias = bias; p.y = (unsigned short*)y;
  p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.Do = D / 2; p.Ho = H / 2; p.Wo = W / 2; p.ldy = ldy; p.Cout = Cout; p.accum = accum;
  p.ntx = (p.Wo + S2T_TX - 1) / S2T_TX; p.nty = (p.Ho + S2T_TY - 1) / S2T_TY; p.ntz = (p.Do + S2T_TZ - 1) / S2T_TZ;
  p.ntiles = (long)N * p.ntz * p.nty * p.ntx;
  if (p.ntiles < 512) return 1;          // (small grids: the gather kernel's finer split fills the chip better)
  const bool prof = bts_prof_on();
  if (prof) bts_prof_begin(39, 2.0 * 27.0 * Cin * (double)Cout * (double)N * p.Do * p.Ho * p.Wo, stream);
  const int grid = p.ntiles < 256 ? (int)p.ntiles : 256;
  (void)hipGetLastError();
  static bool attr_done[2] = {false, false};
  if (dtype == LP_F16) {
    auto kern = lp_s2t_kernel<TF16>;
    if (!attr_done[0]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S2T_LDS) != hipSuccess) return 1; attr_done[0] = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S2T_LDS, stream, p);
  } else {
    auto kern = lp_s2t_kernel<TBF16>;
    if (!attr_done[1]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S2T_LDS) != hipSuccess) return 1; attr_done[1] = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S2T_LDS, stream, p);
  }
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
