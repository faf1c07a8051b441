// Weight (and bias) gradients of the 3-D convolutions on the exact-fp32 matrix pipe.
// The reference obtains these from TF autodiff (train.py:142-151); the math is SURVEY Appendix A':
//   dW[t][pc][qc] = sum_{n,v} P[n, v*s + off_t, pc] * Q[n, v, qc]        db[qc] = sum_{n,v} Q[n, v, qc]
//     Conv3D k1 / k3s1 / k3s2 : P = layer input x, Q = dy  -> dW in (t, Cin, Cout)
//     Conv3DTranspose k3s2    : P = dy (fine grid, s=2), Q = x (coarse grid) -> dW in (t, Cout, Cin)
//     "swapped" k3s1          : P = dy, Q = x with negated tap offsets (u = v + off_t): used when Cout <= 8 so the
//                               tiny channel count sits on the packable P side
// GEMM view: M = 32 rows (P channels of one tap, or -- when P has <= 16 channels -- packed (tap, channel) pairs),
// N = 32 Q-channels, K = voxels (v_mfma_f32_32x32x2_f32 consumes 2 voxels per instruction).
// A 512-thread workgroup (8 waves) walks a run of spatial sub-tiles; the waves split the row-tiles (27 taps) or, when
// there are at most 4 row-tiles, the voxel pairs.  P halo tile and Q tile live in LDS as [voxel][channels] so both
// fragment reads are conflict-free ds_read_b32; the next sub-tile is staged by direct global->LDS DMA
// (global_load_lds_dwordx4) while the current one is multiplied (double-buffered LDS, one barrier per sub-tile).
// Kernel variants <MODE, FIXG>: <2,1>/<2,2> straight-line fixed-geometry sweeps with interleaved staging for every big
// 3x3x3 layer (stride 1 / 2), <1,0> general LDS-DMA staging, <0,0> register staging for odd strides / channel counts.
// Per-workgroup partials go to a workspace and are combined in a fixed order by the finalize kernel (bitwise
// reproducible; no float atomics).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "bts_internal.h"

int bts_prof_on();
void bts_prof_begin(int sym, double flops, hipStream_t stream);
void bts_prof_end(hipStream_t stream);

#define WG_THREADS 512
#define WG_WAVES 8
#define WG_MAXT 4
#define WG_PSLOTS 8
#define WG_QSLOTS 4

struct WgradParams {
  const float* p;
  const float* q;
  float* partial;     // [nsp][npct][nqct][ntiles][32][32]
  double* partial_b;  // [nsp][nqct][32]
  const float* zeros;  // >= 16 zero bytes (source for padding lanes of the LDS-DMA staging)
  int N, Dp, Hp, Wp, Cp, ldp;
  int Dq, Hq, Wq, Cq, ldq;
  int s, loz, loy, lox;
  int IZ, IY, IX;
  int lgTX, lgTY, TZ;
  int ntz, nty, ntx;
  int ntaps, ntiles, cpad, lgSP;
  int nsub, sub_per_wg;
  int want_bias;
  int fixg;  // 1/2: the sub-tile is the unclamped 16x4x2 (stride 1) / 8x4x1 (stride 2) 27-tap geometry -> fixed sweep
  int fastf;  // fixed geometry AND whole tiles / whole 32-channel groups: per-lane staging offsets are precomputed once
  int dbg;  // profiling aid (BTS_WGRAD_DBG), see the sub-tile loop
  int tap_vox[27];  // voxel offset of each tap inside the P halo tile
};

__device__ __forceinline__ int fast_div(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// MFMA sweep over the voxel pairs ("steps") of one staged sub-tile for a wave that owns T row-tiles.
// Software pipelined by hand: the LDS reads of step k+1 are issued before the T MFMAs of step k, so LDS latency and the
// address arithmetic sit in the shadow of the matrix pipe.  Step indices are wave-uniform (scalar address part); the
// per-lane part (row offset of the tile, half-wave voxel, channel) is folded into laneoff[] once.
// The fragment reads are inline-asm ds_read_b32 with hand-counted s_waitcnt lgkmcnt(N): hipcc cannot count LDS
// operations across the loop back-edge and would otherwise wait lgkmcnt(0) right after issuing the prefetch.
template <int T>
struct WgFrag {
  float q, a[T];
};
__device__ __forceinline__ float lds_read_b32_async(unsigned byte_addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int T>
__device__ __forceinline__ void wgrad_steps(const WgradParams& p, const float* bp, const float* bq, const int (&rowoff)[WG_MAXT],
                                            f32x16 (&acc)[WG_MAXT], int st0, int stinc, int nsteps, int h, int l32) {
  const int TXm = (1 << p.lgTX) - 1, TYm = (1 << p.lgTY) - 1, lgXY = p.lgTX + p.lgTY;
  const unsigned bpb = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)bp;  // LDS byte addresses
  const unsigned bqb = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)bq;
  unsigned laneoff[T];
#pragma unroll
  for (int i = 0; i < T; ++i) laneoff[i] = bpb + 4u * (unsigned)(rowoff[i] + ((h * p.s) << p.lgSP));
  const unsigned qlane = bqb + 4u * (unsigned)(h * 32 + l32);
  const int nK = (nsteps - st0 + stinc - 1) / stinc;
  auto load = [&](WgFrag<T>& f, int k) {
    const int m = 2 * (st0 + k * stinc);  // wave-uniform
    const int pvu = (((m >> lgXY) * p.s * p.IY + ((m >> p.lgTX) & TYm) * p.s) * p.IX + (m & TXm) * p.s) << p.lgSP;
    f.q = lds_read_b32_async(qlane + (unsigned)(m * 128));
#pragma unroll
    for (int i = 0; i < T; ++i) f.a[i] = lds_read_b32_async(laneoff[i] + (unsigned)(pvu * 4));
    __builtin_amdgcn_sched_barrier(0);
  };
  // One step = T MFMAs on the fragment fetched one step earlier.  The next fragment's address arithmetic and LDS reads
  // are issued right after the first MFMA so they retire in the shadow of the matrix pipe (a wave blocks on MFMA issue
  // while the pipe is busy; anything placed before the burst would instead serialise with it), and the burst runs at
  // raised priority so the two waves of a SIMD alternate whole bursts instead of dragging each other through them
  // (scripts/ubench/mfma_lds_feed.hip: 142 -> 154 TFLOP/s at 2 waves/SIMD).
  auto step = [&](const WgFrag<T>& cur, WgFrag<T>& nxt, int knext) {
    wait_lgkm<0>();
    __builtin_amdgcn_s_setprio(1);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0], cur.q, acc[0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load(nxt, knext);
#pragma unroll
    for (int i = 1; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[i], cur.q, acc[i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  WgFrag<T> A, B;
  if (nK <= 0) return;
  load(A, 0);
  const int last = nK - 1;
  for (int k = 0; k < nK; k += 2) {
    step(A, B, k + 1 < last ? k + 1 : last);  // the final prefetch re-reads the last fragment (never consumed)
    if (k + 1 >= nK) break;
    step(B, A, k + 2 < last ? k + 2 : last);
  }
  wait_lgkm<0>();
}

// ---- fixed-geometry sweep (3x3x3 taps, 32-channel P voxels, unclamped sub-tile) ---------------------------------------
// Measured cost model (scripts/ubench/wgrad_sweep.hip): next to the matrix pipe EVERY issued instruction costs ~4 cycles
// of SIMD time, scalar ones included, so the general step above (13 SALU + 6 VALU + T+1 LDS reads per T MFMAs) tops out
// at ~105 TFLOP/s.  Here the sub-tile geometry is a template parameter and the whole sub-tile is one straight-line
// sequence: every voxel offset is a DS immediate off T+1 per-wave base registers (tile base + tap offset of row-tile i),
// a step is T+1 ds_read_b32 + T MFMAs and nothing else.  The staging of the NEXT sub-tile is not issued up front by all
// waves at once but handed in as `stage(IC<K>)`, a few instructions at a time in the MFMA shadow of chosen steps.
template <int I>
struct IC { static constexpr int value = I; };
template <int S, int TX, int TY, int TZ>
struct WgGeo {
  static constexpr int IX = (TX - 1) * S + 3, IY = (TY - 1) * S + 3;
  static constexpr int JR = TX / 2;       // voxel-pair steps per x-row
  static constexpr int NS = JR * TY * TZ;  // steps per sub-tile
  // P byte offset (before the tap offset) of the first voxel of step K; Q is simply [m][32]
  static constexpr int p_b(int K) {
    const int r = K / JR, j = K % JR, z = r / TY, y = r % TY;
    return ((z * S * IY + y * S) * IX + 2 * j * S) * 128;
  }
};
template <int T, typename G, int K>
__device__ __forceinline__ void wgx_load(WgFrag<T>& f, const unsigned (&vp)[WG_MAXT], unsigned vq) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.q) : "v"(vq), "n"(K * 256));
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a[0]) : "v"(vp[0]), "n"(G::p_b(K)));
  if constexpr (T > 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a[1]) : "v"(vp[1]), "n"(G::p_b(K)));
  if constexpr (T > 2) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a[2]) : "v"(vp[2]), "n"(G::p_b(K)));
  if constexpr (T > 3) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a[3]) : "v"(vp[3]), "n"(G::p_b(K)));
  __builtin_amdgcn_sched_barrier(0);
}
// Three rotating fragments: the reads of step K+2 are issued in step K, so a fragment has two full MFMA bursts to land.
// lgkmcnt(T+1): LDS operations retire in order, so at most the T+1 youngest (fragment K+1) are still in flight.
template <int T, typename G, int K, typename F>
__device__ __forceinline__ void wgx_steps(WgFrag<T>& f0, WgFrag<T>& f1, WgFrag<T>& f2, f32x16 (&acc)[WG_MAXT],
                                          const unsigned (&vp)[WG_MAXT], unsigned vq, F& stage) {
  if constexpr (K + 1 < G::NS) wait_lgkm<T + 1>();
  else wait_lgkm<0>();
  __builtin_amdgcn_s_setprio(1);
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a[0], f0.q, acc[0], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K + 2 < G::NS) wgx_load<T, G, K + 2>(f2, vp, vq);
  stage(IC<K>{});
  if constexpr (K + 2 >= G::NS && K + 1 < G::NS) wait_lgkm<T + 1>();  // staging writes issued here must not hide fragment K+1
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 1; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a[i], f0.q, acc[i], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K + 1 < G::NS) wgx_steps<T, G, K + 1>(f1, f2, f0, acc, vp, vq, stage);
}
template <int T, typename G, typename F>
__device__ __forceinline__ void wgrad_sweep_fixed(const unsigned (&vp)[WG_MAXT], unsigned vq, f32x16 (&acc)[WG_MAXT], F& stage) {
  WgFrag<T> A, B, C;
  wgx_load<T, G, 0>(A, vp, vq);
  wgx_load<T, G, 1>(B, vp, vq);
  wgx_steps<T, G, 0>(A, B, C, acc, vp, vq, stage);
  wait_lgkm<0>();
}

// GLDS = true: tiles are staged with direct global->LDS DMA (global_load_lds_dwordx4: no staging registers, the copy of
// the next sub-tile is fully asynchronous under the MFMA sweep).  The LDS image is lane-linear by construction (slot e
// lives at byte 16*e), padding lanes read 16 zero bytes from p.zeros.  GLDS = false: register-staged fallback for
// channel counts / strides that are not 16-byte granular.
// MODE 0: register staging, 1: LDS-DMA staging, 2: LDS-DMA staging with precomputed slot offsets (p.fastf).
// FIXG 0: general sweep, 1/2: fixed-geometry sweep (p.fixg).  Separate kernels so each keeps its own register budget.
template <int MODE, int FIXG>
__global__ __launch_bounds__(WG_THREADS, 2) void wgrad_kernel(const WgradParams p) {
  constexpr bool GLDS = MODE >= 1, FAST = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  const int pct = blockIdx.y, qct = blockIdx.z;
  const int TX = 1 << p.lgTX, TY = 1 << p.lgTY;
  const int M = TX * TY * p.TZ;
  const int SP = 1 << p.lgSP;
  const int tileVoxP = p.IZ * p.IY * p.IX;
  const int pRegion = ((tileVoxP * (SP >> 2) + 63) & ~63) * 4;  // dwords, padded to whole wave-loads (64 x 16 B)
  const int qRegion = ((M * 8 + 63) & ~63) * 4;
  const int bufDw = pRegion + qRegion;
  const bool ksplit = p.ntiles <= WG_MAXT;  // few row-tiles: every wave takes all of them, waves split the voxel pairs

  // ---- per-lane row offsets (dwords inside the P tile) for this wave's row-tiles ----
  int rowoff[WG_MAXT];
#pragma unroll
  for (int i = 0; i < WG_MAXT; ++i) {
    const int tile = ksplit ? i : wave + WG_WAVES * i;
    int off = 0;
    if (tile < p.ntiles) {
      if (p.cpad == 32) off = (p.tap_vox[tile] << p.lgSP) + l32;
      else {
        const int ci = tile * 32 + l32;
        int tap = ci / p.cpad;
        const int ch = ci - tap * p.cpad;
        if (tap >= p.ntaps) tap = 0;  // junk row, never read back
        off = (p.tap_vox[tap] << p.lgSP) + ch;
      }
    }
    rowoff[i] = off;
  }

  int ntw = 0;  // row-tiles owned by this wave
#pragma unroll
  for (int i = 0; i < WG_MAXT; ++i) {
    const int tile = ksplit ? i : wave + WG_WAVES * i;
    if (tile < p.ntiles) ntw = i + 1;
  }

  const int qpvP = SP >> 2;  // float4 per P voxel
  const float invIX = 1.0f / (float)p.IX, invIYX = 1.0f / (float)(p.IY * p.IX);
  // Staging slots: slot e of a tile covers float4 #e of its LDS image ([voxel][SP] resp. [m][32]), so the LDS offset is
  // simply 4*e; the voxel decomposition is recomputed per sub-tile (a dozen VALU ops) rather than kept in registers:
  // spilled slot tables would be reloaded through the VM counter and serialise the prefetch loads.
  const int pc0 = pct * 32;
  const int nPslots = tileVoxP * qpvP, nQslots = M * 8;
  const int IYX = p.IY * p.IX;
  const int vecP = (p.ldp % 4 == 0) && (p.Cp % 4 == 0) && ((((uintptr_t)p.p) & 15) == 0);
  const int vecQ = (p.ldq % 4 == 0) && (p.Cq % 4 == 0) && ((((uintptr_t)p.q) & 15) == 0);

  f32x4 preP[WG_PSLOTS], preQ[WG_QSLOTS];
  auto fetch = [&](int sub) {
    int b = sub;
    const int tx = b % p.ntx; b /= p.ntx;
    const int ty = b % p.nty; b /= p.nty;
    const int tz = b % p.ntz;
    const int n = b / p.ntz;
    const int oz0 = tz * p.TZ, oy0 = ty * TY, ox0 = tx * TX;
    const int iz0 = oz0 * p.s + p.loz, iy0 = oy0 * p.s + p.loy, ix0 = ox0 * p.s + p.lox;
    // wave-uniform tile origins (may be negative for the halo): pointer arithmetic stays scalar
    const float* pbase = p.p + ((((long)n * p.Dp + iz0) * p.Hp + iy0) * p.Wp + ix0) * (long)p.ldp;
    const float* qbase = p.q + ((((long)n * p.Dq + oz0) * p.Hq + oy0) * p.Wq + ox0) * (long)p.ldq;
#pragma unroll
    for (int i = 0; i < WG_PSLOTS; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int e = tid + i * WG_THREADS;
      if (e < nPslots) {
        const int vox = e >> (p.lgSP - 2), qd = e & (qpvP - 1);
        const int vz = fast_div(vox, invIYX);
        const int r = vox - vz * IYX;
        const int vy = fast_div(r, invIX);
        const int vx = r - vy * p.IX;
        const int c = pc0 + qd * 4;
        if ((unsigned)(iz0 + vz) < (unsigned)p.Dp && (unsigned)(iy0 + vy) < (unsigned)p.Hp &&
            (unsigned)(ix0 + vx) < (unsigned)p.Wp && c < p.Cp) {
          const float* src = pbase + (((vz * p.Hp + vy) * p.Wp + vx) * p.ldp + c);
          if (vecP) v = *reinterpret_cast<const f32x4*>(src);
          else {
            v[0] = src[0];
            if (c + 1 < p.Cp) v[1] = src[1];
            if (c + 2 < p.Cp) v[2] = src[2];
            if (c + 3 < p.Cp) v[3] = src[3];
          }
        }
      }
      preP[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WG_QSLOTS; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int e = tid + i * WG_THREADS;
      if (e < nQslots) {
        const int m = e >> 3, qd = e & 7;
        const int mz = m >> (p.lgTX + p.lgTY), my = (m >> p.lgTX) & (TY - 1), mx = m & (TX - 1);
        const int c = qct * 32 + qd * 4;
        if (oz0 + mz < p.Dq && oy0 + my < p.Hq && ox0 + mx < p.Wq && c < p.Cq) {
          const float* src = qbase + (((mz * p.Hq + my) * p.Wq + mx) * p.ldq + c);
          if (vecQ) v = *reinterpret_cast<const f32x4*>(src);
          else {
            v[0] = src[0];
            if (c + 1 < p.Cq) v[1] = src[1];
            if (c + 2 < p.Cq) v[2] = src[2];
            if (c + 3 < p.Cq) v[3] = src[3];
          }
        }
      }
      preQ[i] = v;
    }
  };
  // direct-to-LDS staging of one sub-tile into buffer `buf` (every lane of every issuing wave is active)
  auto fetch_glds = [&](int sub, float* buf) {
    int b = sub;
    const int tx = b % p.ntx; b /= p.ntx;
    const int ty = b % p.nty; b /= p.nty;
    const int tz = b % p.ntz;
    const int n = b / p.ntz;
    const int oz0 = tz * p.TZ, oy0 = ty * TY, ox0 = tx * TX;
    const int iz0 = oz0 * p.s + p.loz, iy0 = oy0 * p.s + p.loy, ix0 = ox0 * p.s + p.lox;
    const float* pbase = p.p + ((((long)n * p.Dp + iz0) * p.Hp + iy0) * p.Wp + ix0) * (long)p.ldp;
    const float* qbase = p.q + ((((long)n * p.Dq + oz0) * p.Hq + oy0) * p.Wq + ox0) * (long)p.ldq;
#pragma unroll
    for (int i = 0; i < WG_PSLOTS; ++i) {
      const int e0 = wave * 64 + i * WG_THREADS;  // wave-uniform first slot of this wave-load
      if (e0 * 4 < pRegion) {
        const int e = e0 + lane;
        const float* src = p.zeros;
        if (e < nPslots) {
          const int vox = e >> (p.lgSP - 2), qd = e & (qpvP - 1);
          const int vz = fast_div(vox, invIYX);
          const int r = vox - vz * IYX;
          const int vy = fast_div(r, invIX);
          const int vx = r - vy * p.IX;
          const int c = pc0 + qd * 4;
          if ((unsigned)(iz0 + vz) < (unsigned)p.Dp && (unsigned)(iy0 + vy) < (unsigned)p.Hp &&
              (unsigned)(ix0 + vx) < (unsigned)p.Wp && c < p.Cp)
            src = pbase + (((vz * p.Hp + vy) * p.Wp + vx) * p.ldp + c);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + e0 * 4), 16, 0, 0);
      }
    }
    float* bqd = buf + pRegion;
#pragma unroll
    for (int i = 0; i < WG_QSLOTS; ++i) {
      const int e0 = wave * 64 + i * WG_THREADS;
      if (e0 * 4 < qRegion) {
        const int e = e0 + lane;
        const float* src = p.zeros;
        if (e < nQslots) {
          const int m = e >> 3, qd = e & 7;
          const int mz = m >> (p.lgTX + p.lgTY), my = (m >> p.lgTX) & (TY - 1), mx = m & (TX - 1);
          const int c = qct * 32 + qd * 4;
          if (oz0 + mz < p.Dq && oy0 + my < p.Hq && ox0 + mx < p.Wq && c < p.Cq)
            src = qbase + (((mz * p.Hq + my) * p.Wq + mx) * p.ldq + c);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(bqd + e0 * 4), 16, 0, 0);
      }
    }
  };
  // Fast staging (p.fastf): slot -> (voxel, channel quad) never changes, so the byte offset of every slot relative to
  // the tile origin and its halo-face membership are computed ONCE; per sub-tile an interior tile costs one
  // global_load_lds (scalar base + 32-bit lane offset) per slot and nothing else -- next to the matrix pipe every
  // issued instruction costs ~4 cycles of SIMD time.  Face slots of boundary tiles are zero-filled with ds_write.
  unsigned relP[WG_PSLOTS], relQ[WG_QSLOTS], faceP[2] = {0u, 0u};
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < WG_PSLOTS; ++i) {
      const int e = wave * 64 + lane + i * WG_THREADS;
      int vz = -p.loz, vy = -p.loy, vx = -p.lox, qd = 0;  // padding lanes re-read an always-valid voxel (never consumed)
      unsigned face = 0;
      if (e < nPslots) {
        const int vox = e >> 3;
        qd = e & 7;
        vz = fast_div(vox, invIYX);
        const int r = vox - vz * IYX;
        vy = fast_div(r, invIX);
        vx = r - vy * p.IX;
        face = (vz == 0 ? 1u : 0u) | (vz == p.IZ - 1 ? 2u : 0u) | (vy == 0 ? 4u : 0u) | (vy == p.IY - 1 ? 8u : 0u) |
               (vx == 0 ? 16u : 0u) | (vx == p.IX - 1 ? 32u : 0u);
      }
      relP[i] = (unsigned)((((vz * p.Hp + vy) * p.Wp + vx) * p.ldp + pc0 + qd * 4) * 4);
      faceP[i >> 2] |= face << (8 * (i & 3));
    }
#pragma unroll
    for (int i = 0; i < WG_QSLOTS; ++i) {
      const int e = wave * 64 + lane + i * WG_THREADS;
      int mz = 0, my = 0, mx = 0, qd = 0;
      if (e < nQslots) {
        const int m = e >> 3;
        qd = e & 7;
        mz = m >> (p.lgTX + p.lgTY); my = (m >> p.lgTX) & (TY - 1); mx = m & (TX - 1);
      }
      relQ[i] = (unsigned)((((mz * p.Hq + my) * p.Wq + mx) * p.ldq + qct * 32 + qd * 4) * 4);
    }
  }
  // per-sub-tile staging state (wave-uniform), set by fast_prep and consumed by the per-slot issue functions
  const char* f_pbase = nullptr;
  const char* f_qbase = nullptr;
  unsigned f_tmask = 0;
  float* f_buf = nullptr;
  auto fast_prep = [&](int sub, float* buf) {
    int b = sub;
    const int tx = b % p.ntx; b /= p.ntx;
    const int ty = b % p.nty; b /= p.nty;
    const int tz = b % p.ntz;
    const int n = b / p.ntz;
    const int oz0 = tz * p.TZ, oy0 = ty * TY, ox0 = tx * TX;
    const int iz0 = oz0 * p.s + p.loz, iy0 = oy0 * p.s + p.loy, ix0 = ox0 * p.s + p.lox;
    f_pbase = reinterpret_cast<const char*>(p.p + ((((long)n * p.Dp + iz0) * p.Hp + iy0) * p.Wp + ix0) * (long)p.ldp);
    f_qbase = reinterpret_cast<const char*>(p.q + ((((long)n * p.Dq + oz0) * p.Hq + oy0) * p.Wq + ox0) * (long)p.ldq);
    // faces of the halo tile that lie outside the image (stride 1: one-voxel halo on both sides; stride 2: high side only)
    const unsigned lowf = p.loz < 0 ? 1u : 0u;
    f_tmask = ((tz == 0) ? lowf : 0u) | ((tz == p.ntz - 1) ? 2u : 0u) | ((ty == 0) ? 4u * lowf : 0u) |
              ((ty == p.nty - 1) ? 8u : 0u) | ((tx == 0) ? 16u * lowf : 0u) | ((tx == p.ntx - 1) ? 32u : 0u);
    f_buf = buf;
  };
  auto fast_slot_p = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const int e0 = wave * 64 + i * WG_THREADS;
    if (e0 * 4 < pRegion) {
      if (f_tmask == 0 || ((faceP[i >> 2] >> (8 * (i & 3))) & f_tmask) == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(f_pbase + relP[i]),
                                         (__attribute__((address_space(3))) void*)(f_buf + e0 * 4), 16, 0, 0);
      else {
        int ln = lane;
        asm volatile("" : "+v"(ln));  // keep the 8 per-slot zero-fill addresses out of the sub-tile loop's live registers
        *reinterpret_cast<f32x4*>(f_buf + (e0 + ln) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto fast_slot_q = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const int e0 = wave * 64 + i * WG_THREADS;
    if (e0 * 4 < qRegion)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(f_qbase + relQ[i]),
                                       (__attribute__((address_space(3))) void*)(f_buf + pRegion + e0 * 4), 16, 0, 0);
  };
  auto fetch_fast = [&](int sub, float* buf) {  // everything at once (prologue only)
    fast_prep(sub, buf);
    fast_slot_p(IC<0>{}); fast_slot_p(IC<1>{}); fast_slot_p(IC<2>{}); fast_slot_p(IC<3>{});
    fast_slot_p(IC<4>{}); fast_slot_p(IC<5>{}); fast_slot_p(IC<6>{}); fast_slot_p(IC<7>{});
    fast_slot_q(IC<0>{}); fast_slot_q(IC<1>{}); fast_slot_q(IC<2>{}); fast_slot_q(IC<3>{});
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < WG_PSLOTS; ++i) {
      const int e = tid + i * WG_THREADS;
      if (e < nPslots) *reinterpret_cast<f32x4*>(buf + e * 4) = preP[i];
    }
    float* bq = buf + pRegion;
#pragma unroll
    for (int i = 0; i < WG_QSLOTS; ++i) {
      const int e = tid + i * WG_THREADS;
      if (e < nQslots) *reinterpret_cast<f32x4*>(bq + e * 4) = preQ[i];
    }
  };

  f32x16 acc[WG_MAXT];
#pragma unroll
  for (int i = 0; i < WG_MAXT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // fixed-geometry kernels: partial accumulators of the three left-over row-tiles 24..26 (this wave's slice of K)
  f32x16 accx[WG_MAXT];
  int rowoffx[3] = {0, 0, 0};
  if constexpr (FIXG != 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[i][r] = 0.f;
      rowoffx[i] = (p.tap_vox[24 + i] << p.lgSP) + l32;
    }
  }
  double bsum = 0.0;  // bias column sum: thread (c = tid&31, part = tid>>5)

  const int sub0 = blockIdx.x * p.sub_per_wg;
  int sub1 = sub0 + p.sub_per_wg;
  if (sub1 > p.nsub) sub1 = p.nsub;
  const int nsteps = M >> 1;

  if (sub0 < sub1) {
    if constexpr (GLDS) {
      if constexpr (FAST) fetch_fast(sub0, lds);
      else fetch_glds(sub0, lds);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      fetch(sub0);
      commit(lds);
    }
  }
  __syncthreads();
  int cur = 0;
  for (int sub = sub0; sub < sub1; ++sub) {
    const float* bp = lds + cur * bufDw;
    const float* bq = bp + pRegion;
    const bool more = (sub + 1) < sub1;
    // BTS_WGRAD_DBG (timing only, results are wrong): 1 skip the sweep, 2 no re-staging, 3 = 2 + no barriers,
    // 4 = 2 + at most 3 row-tiles per wave, 5 stage but keep reading buffer 0, 7 = always stage the first sub-tile
    const bool dofetch = more && (BTS_DBG(p) < 2 || BTS_DBG(p) >= 5);
    if constexpr (FIXG == 0) {
      if (dofetch) {
        if constexpr (GLDS) fetch_glds(sub + 1, lds + (cur ^ 1) * bufDw);
        else fetch(sub + 1);
      }
    }

    if (p.want_bias && pct == 0) {
      const int c = tid & 31, part = tid >> 5;
      float s = 0.f;
      for (int m = part; m < M; m += 16) s += bq[m * 32 + c];
      bsum += (double)s;
    }
    if constexpr (FIXG != 0) {
      using G = typename std::conditional<FIXG == 1, WgGeo<1, 16, 4, 2>, WgGeo<2, 8, 4, 1>>::type;
      // staging schedule: tile decode at step 0, then one slot every few steps (8 P + 4 Q slots per wave at most)
      constexpr int P0 = 1, PD = (G::NS >= 64) ? 5 : 1, Q0 = P0 + 8 * PD, QD = (G::NS >= 64) ? 5 : 1;
      static_assert(Q0 + 3 * QD < G::NS, "staging schedule does not fit the sub-tile");
      auto stage = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (K == 0) {
          if (dofetch) fast_prep(BTS_DBG(p) == 7 ? sub0 : sub + 1, lds + (cur ^ 1) * bufDw);
        } else if constexpr (K >= P0 && K < Q0 && (K - P0) % PD == 0) {
          if (dofetch) fast_slot_p(IC<(K - P0) / PD>{});
        } else if constexpr (K >= Q0 && K <= Q0 + 3 * QD && (K - Q0) % QD == 0) {
          if (dofetch) fast_slot_q(IC<(K - Q0) / QD>{});
        }
      };
      const unsigned bpb = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)bp;
      const unsigned vq = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)bq + 4u * (unsigned)(h * 32 + l32);
      unsigned vp[WG_MAXT];
#pragma unroll
      for (int i = 0; i < WG_MAXT; ++i) vp[i] = bpb + 4u * (unsigned)(rowoff[i] + ((h * p.s) << 5));
      if (BTS_DBG(p) == 1) {
        if (dofetch) fetch_fast(sub + 1, lds + (cur ^ 1) * bufDw);
      } else {
        // 27 row-tiles over 8 waves: every wave owns tiles w, w+8, w+16 for the whole sub-tile and 1/8 of the voxel pairs
        // of the three left-over tiles 24..26 (its own x-row for stride 1, half a row for stride 2) in a second, short
        // sweep -> 216 MFMAs per wave and sub-tile on every wave instead of 256 / 192 (the barrier waits for the slowest)
        using GX = typename std::conditional<FIXG == 1, WgGeo<1, 16, 1, 1>, WgGeo<2, 4, 1, 1>>::type;   // NS = 8 / 2 steps
        static_assert(GX::NS * WG_WAVES == G::NS, "the left-over tiles' steps must split evenly over the waves");
        const int k0 = wave * GX::NS;                       // first step of this wave's slice (wave-uniform)
        const int r0 = k0 / G::JR, j0 = k0 % G::JR;         // its x-row (z-major) and position inside the row
        const int pb0 = (((r0 >> 2) * p.s * p.IY + (r0 & 3) * p.s) * p.IX + 2 * j0 * p.s) * 128;   // = G::p_b(k0), TY = 4
        unsigned vpx[WG_MAXT];
#pragma unroll
        for (int i = 0; i < 3; ++i) vpx[i] = bpb + 4u * (unsigned)(rowoffx[i] + ((h * p.s) << 5)) + (unsigned)pb0;
        vpx[3] = vpx[2];
        auto nostage = [](auto) {};
        wgrad_sweep_fixed<3, GX>(vpx, vq + (unsigned)(k0 * 256), accx, nostage);
        wgrad_sweep_fixed<3, G>(vp, vq, acc, stage);
      }
    } else {
      const int st0 = ksplit ? wave : 0, stinc = ksplit ? WG_WAVES : 1;
      if (BTS_DBG(p) != 1) switch ((BTS_DBG(p) == 4 && ntw > 3) ? 3 : ntw) {  // wave-uniform: row-tiles this wave owns -> branch-free MFMA bodies
        case 4: wgrad_steps<4>(p, bp, bq, rowoff, acc, st0, stinc, nsteps, h, l32); break;
        case 3: wgrad_steps<3>(p, bp, bq, rowoff, acc, st0, stinc, nsteps, h, l32); break;
        case 2: wgrad_steps<2>(p, bp, bq, rowoff, acc, st0, stinc, nsteps, h, l32); break;
        case 1: wgrad_steps<1>(p, bp, bq, rowoff, acc, st0, stinc, nsteps, h, l32); break;
        default: break;
      }
    }
    if (dofetch) {
      if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else commit(lds + (cur ^ 1) * bufDw);
    }
    if (BTS_DBG(p) != 3) __syncthreads();
    if (BTS_DBG(p) < 2) cur ^= 1;
  }

  // ---- write partials ----
  const long tileBase = ((((long)blockIdx.x * gridDim.y + pct) * gridDim.z + qct) * p.ntiles) * 1024;
  if constexpr (FIXG != 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // the wave's own tiles w, w+8, w+16
      float* dst = p.partial + tileBase + (long)(wave + WG_WAVES * i) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l32] = acc[i][r];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // tiles 24..26: fixed-order sum of the 8 waves' K-slices through LDS
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l32] = accx[i][r];
      __syncthreads();
      float* dst = p.partial + tileBase + (long)(24 + i) * 1024;
      for (int e = tid; e < 1024; e += WG_THREADS) {
        float s = lds[e];
#pragma unroll
        for (int w = 1; w < WG_WAVES; ++w) s += lds[w * 1024 + e];
        dst[e] = s;
      }
    }
  } else if (!ksplit) {
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
      const int tile = wave + WG_WAVES * i;
      if (tile < p.ntiles) {
        float* dst = p.partial + tileBase + (long)tile * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l32] = acc[i][r];
      }
    }
  } else {
    // cross-wave fixed-order reduction through LDS (staging buffers are free now)
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
      if (i < p.ntiles) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l32] = acc[i][r];
        __syncthreads();
        float* dst = p.partial + tileBase + (long)i * 1024;
        for (int e = tid; e < 1024; e += WG_THREADS) {
          float s = lds[e];
#pragma unroll
          for (int w = 1; w < WG_WAVES; ++w) s += lds[w * 1024 + e];
          dst[e] = s;
        }
      }
    }
  }
  if (p.want_bias && pct == 0) {
    __syncthreads();
    double* shd = reinterpret_cast<double*>(lds);
    shd[tid] = bsum;
    __syncthreads();
    if (tid < 32) {
      double s = 0.0;
      for (int part = 0; part < 16; ++part) s += shd[part * 32 + tid];
      p.partial_b[((long)blockIdx.x * gridDim.z + qct) * 32 + tid] = s;
    }
  }
}

struct WfinParams {
  const float* partial;
  const double* partial_b;
  float* dw;
  float* db;
  int nsp, npct, nqct, ntaps, ntiles, cpad, Cp, Cq;
  long sT, sP, sQ;
  int fold_axis;  // 0 none, 1 the P-channel axis is the (possibly folded) reference Cin axis, 2 the Q axis is
  int shift, dup_start;
  int accum;
};

// dW[t][pc][qc] (+)= sum_wg partial.  One block per row (32 consecutive qc of one (t, pc, qct)): 256 threads = 8 float4
// columns x 32 partial-lanes, every lane streams its share of the nsp partials (4 independent 16-byte loads in flight),
// fp64 accumulate, fixed-order combine through LDS (bitwise reproducible).  The pass is pure HBM streaming.
// Folded slab channel c maps to reference channel c+shift and, if c >= dup_start, also to c-dup_start (both copies of
// the duplicated slice see the same input: encoder.py:83-87).
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const WfinParams f) {
  __shared__ double sh[32][33];
  const int q4 = threadIdx.x & 7, wl = threadIdx.x >> 3;
  const long rows = (long)f.ntaps * f.Cp * f.nqct;
  const long wgStride = (long)f.npct * f.nqct * f.ntiles * 1024;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int qct = (int)(row % f.nqct);
    long r = row / f.nqct;
    const int pc = (int)(r % f.Cp);
    const int t = (int)(r / f.Cp);
    long off;
    if (f.cpad == 32) off = ((((long)(pc >> 5)) * f.nqct + qct) * f.ntiles + t) * 1024 + (pc & 31) * 32;
    else {
      const int ci = t * f.cpad + pc;
      off = ((long)qct * f.ntiles + (ci >> 5)) * 1024 + (ci & 31) * 32;
    }
    const float* src = f.partial + off + q4 * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // fp64 combine of the per-workgroup fp32 partials
    int w = wl;
    for (; w + 96 < f.nsp; w += 128) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + (long)w * wgStride);
      const f32x4 b = *reinterpret_cast<const f32x4*>(src + (long)(w + 32) * wgStride);
      const f32x4 c = *reinterpret_cast<const f32x4*>(src + (long)(w + 64) * wgStride);
      const f32x4 d = *reinterpret_cast<const f32x4*>(src + (long)(w + 96) * wgStride);
      s0 += (double)a[0]; s1 += (double)a[1]; s2 += (double)a[2]; s3 += (double)a[3];
      s0 += (double)b[0]; s1 += (double)b[1]; s2 += (double)b[2]; s3 += (double)b[3];
      s0 += (double)c[0]; s1 += (double)c[1]; s2 += (double)c[2]; s3 += (double)c[3];
      s0 += (double)d[0]; s1 += (double)d[1]; s2 += (double)d[2]; s3 += (double)d[3];
    }
    for (; w < f.nsp; w += 32) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + (long)w * wgStride);
      s0 += (double)a[0]; s1 += (double)a[1]; s2 += (double)a[2]; s3 += (double)a[3];
    }
    __syncthreads();
    sh[wl][q4 * 4 + 0] = s0; sh[wl][q4 * 4 + 1] = s1; sh[wl][q4 * 4 + 2] = s2; sh[wl][q4 * 4 + 3] = s3;
    __syncthreads();
    if (threadIdx.x < 32) {
      const int e = threadIdx.x;
      double totd = sh[0][e];
#pragma unroll
      for (int k = 1; k < 32; ++k) totd += sh[k][e];
      const float tot = (float)totd;
      const int qc = qct * 32 + e;
      if (qc < f.Cq) {
        const int pr = (f.fold_axis == 1) ? pc + f.shift : pc;
        const int qr = (f.fold_axis == 2) ? qc + f.shift : qc;
        float* d1 = f.dw + t * f.sT + pr * f.sP + qr * f.sQ;
        *d1 = f.accum ? (*d1 + tot) : tot;
        if (f.shift > 0) {
          if (f.fold_axis == 1 && pc >= f.dup_start) {
            float* d2 = f.dw + t * f.sT + (pc - f.dup_start) * f.sP + qr * f.sQ;
            *d2 = f.accum ? (*d2 + tot) : tot;
          } else if (f.fold_axis == 2 && qc >= f.dup_start) {
            float* d2 = f.dw + t * f.sT + pr * f.sP + (qc - f.dup_start) * f.sQ;
            *d2 = f.accum ? (*d2 + tot) : tot;
          }
        }
      }
    }
  }
  // bias: block qct combines the per-workgroup fp64 column sums of its 32 channels (32 channels x 8 partial-lanes)
  if (f.db && (int)blockIdx.x < f.nqct) {
    const int e = threadIdx.x & 31, l8 = threadIdx.x >> 5, qct = blockIdx.x;
    double s = 0.0;
    for (int w = l8; w < f.nsp; w += 8) s += f.partial_b[((long)w * f.nqct + qct) * 32 + e];
    __syncthreads();
    sh[l8][e] = s;
    __syncthreads();
    if (l8 == 0 && qct * 32 + e < f.Cq) {
      double tot = sh[0][e];
#pragma unroll
      for (int k = 1; k < 8; ++k) tot += sh[k][e];
      const int i = qct * 32 + e;
      f.db[i] = f.accum ? (f.db[i] + (float)tot) : (float)tot;
    }
  }
}

// Same combine for few partials (nsp <= 16): one thread per output element, partials read coalesced along qc.
__global__ __launch_bounds__(256) void wgrad_finalize_flat_kernel(const WfinParams f) {
  const int nq = f.nqct * 32;
  const long total = (long)f.ntaps * f.Cp * nq;
  const long wgStride = (long)f.npct * f.nqct * f.ntiles * 1024;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int qc = (int)(i % nq);
    long r = i / nq;
    const int pc = (int)(r % f.Cp);
    const int t = (int)(r / f.Cp);
    if (qc >= f.Cq) continue;
    const int qct = qc >> 5, e = qc & 31;
    long off;
    if (f.cpad == 32) off = ((((long)(pc >> 5)) * f.nqct + qct) * f.ntiles + t) * 1024 + (pc & 31) * 32 + e;
    else {
      const int ci = t * f.cpad + pc;
      off = ((long)qct * f.ntiles + (ci >> 5)) * 1024 + (ci & 31) * 32 + e;
    }
    double totd = 0.0;
    for (int w = 0; w < f.nsp; ++w) totd += (double)f.partial[w * wgStride + off];
    const float tot = (float)totd;
    const int pr = (f.fold_axis == 1) ? pc + f.shift : pc;
    const int qr = (f.fold_axis == 2) ? qc + f.shift : qc;
    float* d1 = f.dw + t * f.sT + pr * f.sP + qr * f.sQ;
    *d1 = f.accum ? (*d1 + tot) : tot;
    if (f.shift > 0) {
      if (f.fold_axis == 1 && pc >= f.dup_start) {
        float* d2 = f.dw + t * f.sT + (pc - f.dup_start) * f.sP + qr * f.sQ;
        *d2 = f.accum ? (*d2 + tot) : tot;
      } else if (f.fold_axis == 2 && qc >= f.dup_start) {
        float* d2 = f.dw + t * f.sT + pr * f.sP + (qc - f.dup_start) * f.sQ;
        *d2 = f.accum ? (*d2 + tot) : tot;
      }
    }
  }
  if (f.db && blockIdx.x == 0) {
    for (int i = threadIdx.x; i < f.Cq; i += blockDim.x) {
      double s = 0.0;
      for (int w = 0; w < f.nsp; ++w) s += f.partial_b[((long)w * f.nqct + (i >> 5)) * 32 + (i & 31)];
      f.db[i] = f.accum ? (f.db[i] + (float)s) : (float)s;
    }
  }
}

static int ilog2w(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

struct WgradPlan {
  WgradParams p;
  int nsp, npct, nqct;
  size_t shmem;
  long partial_floats;
  long partial_b_doubles;
};

// neg: tap offsets are negated (swapped-role form, s == 1 only)
static int plan_wgrad(WgradPlan& pl, int ntaps, int s, int neg, int N, int Dp, int Hp, int Wp_, int Cp, int Dq, int Hq,
                      int Wq, int Cq) {
  WgradParams& p = pl.p;
  p.N = N; p.Dp = Dp; p.Hp = Hp; p.Wp = Wp_; p.Cp = Cp;
  p.Dq = Dq; p.Hq = Hq; p.Wq = Wq; p.Cq = Cq;
  p.s = s; p.ntaps = ntaps;
  int lo = 0, span = 1;
  if (ntaps == 27 && s == 1) { lo = -1; span = 3; }
  if (ntaps == 27 && s == 2) { lo = 0; span = 3; }
  p.loz = p.loy = p.lox = lo;
  // (tap, channel) row packing when P has few channels
  int cpad = 32;
  if (Cp <= 16 && ntaps == 27) { cpad = 2; while (cpad < Cp) cpad *= 2; }
  p.cpad = cpad;
  const int SP = cpad == 32 ? 32 : (cpad < 4 ? 4 : cpad);
  p.lgSP = ilog2w(SP);
  p.ntiles = (cpad == 32) ? ntaps : (ntaps * cpad + 31) / 32;
  // sub-tile: 128 voxels for s=1 (16x4x2), 32 voxels for s=2 (8x4x1); 1x1x1: 256 voxels (32x4x2)
  int TX, TY, TZ;
  if (ntaps == 1) { TX = 32; TY = 4; TZ = 2; }
  else if (s == 1) { TX = 16; TY = 4; TZ = 2; }
  else { TX = 8; TY = 4; TZ = 1; }
  while (TX > 2 && TX / 2 >= Wq) { TX /= 2; if (TY * 2 <= 8) TY *= 2; else TZ *= 2; }
  while (TY > 1 && TY / 2 >= Hq) { TY /= 2; TZ *= 2; }
  while (TZ > 1 && TZ / 2 >= Dq && TX * TY * (TZ / 2) >= 2) TZ /= 2;
  p.lgTX = ilog2w(TX); p.lgTY = ilog2w(TY); p.TZ = TZ;
  p.ntx = (Wq + TX - 1) / TX; p.nty = (Hq + TY - 1) / TY; p.ntz = (Dq + TZ - 1) / TZ;
  p.IX = (TX - 1) * s + span; p.IY = (TY - 1) * s + span; p.IZ = (TZ - 1) * s + span;
  const int tileVoxP = p.IZ * p.IY * p.IX;
  const int M = TX * TY * TZ;
  if (tileVoxP * (SP / 4) > WG_THREADS * WG_PSLOTS || M * 8 > WG_THREADS * WG_QSLOTS) return BTS_ERR_SHAPE;
  pl.shmem = (size_t)2 * ((((tileVoxP * (SP / 4) + 63) & ~63) + ((M * 8 + 63) & ~63)) * 4) * sizeof(float);
  if (pl.shmem < (size_t)WG_WAVES * 1024 * sizeof(float)) pl.shmem = (size_t)WG_WAVES * 1024 * sizeof(float);
  if (pl.shmem > 160 * 1024) return BTS_ERR_SHAPE;
  for (int t = 0; t < 27; ++t) p.tap_vox[t] = 0;
  if (ntaps == 27)
    for (int t = 0; t < 27; ++t) {
      int oz = t / 9 + (s == 1 ? -1 : 0), oy = (t / 3) % 3 + (s == 1 ? -1 : 0), ox = t % 3 + (s == 1 ? -1 : 0);
      if (neg) { oz = -oz; oy = -oy; ox = -ox; }
      p.tap_vox[t] = ((oz - lo) * p.IY + (oy - lo)) * p.IX + (ox - lo);
    }
  p.fixg = 0;
  if (ntaps == 27 && cpad == 32 && !neg && getenv("BTS_WGRAD_NOFIX") == nullptr) {
    if (s == 1 && TX == 16 && TY == 4 && TZ == 2) p.fixg = 1;
    if (s == 2 && TX == 8 && TY == 4 && TZ == 1) p.fixg = 2;
  }
  p.fastf = 0;
  if (p.fixg && Wq % TX == 0 && Hq % TY == 0 && Dq % TZ == 0 && Cp % 32 == 0 && Cq % 32 == 0 &&
      getenv("BTS_WGRAD_NOFASTF") == nullptr)
    p.fastf = 1;  // (the 32-bit staging offsets are range-checked where the leading dimensions are known)
  if (!p.fastf) p.fixg = 0;  // the fixed sweep only ships together with the fast staging
  p.nsub = N * p.ntz * p.nty * p.ntx;
  pl.npct = (cpad == 32) ? (Cp + 31) / 32 : 1;
  pl.nqct = (Cq + 31) / 32;
  // K-split count nsp (workgroups per (P,Q) channel-tile pair).  One workgroup is resident per CU, so the launch runs in
  // rounds of 256; cost model in units of one sub-tile sweep: rounds x (sub-tiles per WG + 0.5 fixed) + the partials'
  // write/combine traffic (~0.0023 per workgroup).  Smallest cost wins, ties go to fewer partials.
  {
    const long pairs = (long)pl.npct * pl.nqct;
    const char* ov = getenv("BTS_WGRAD_WGS");  // experiment override: target workgroup count
    long best = 1;
    if (ov) {
      best = atol(ov) / pairs;
    } else {
      double bestc = 1e30;
      long maxn = 2048 / pairs;
      if (maxn < 1) maxn = 1;
      if (maxn > p.nsub) maxn = p.nsub;
      for (long n = 1; n <= maxn; ++n) {
        const long spw = (p.nsub + n - 1) / n;
        const long ne = (p.nsub + spw - 1) / spw;
        if (ne != n) continue;
        const long rounds = (ne * pairs + 255) / 256;
        const double c = (double)rounds * ((double)spw + 0.5) + 0.0023 * (double)(ne * pairs);
        if (c < bestc - 1e-9) { bestc = c; best = n; }
      }
    }
    if (best < 1) best = 1;
    if (best > p.nsub) best = p.nsub;
    p.sub_per_wg = (int)((p.nsub + best - 1) / best);
    pl.nsp = (p.nsub + p.sub_per_wg - 1) / p.sub_per_wg;
  }
  pl.partial_floats = (long)pl.nsp * pl.npct * pl.nqct * p.ntiles * 1024;
  pl.partial_b_doubles = (long)pl.nsp * pl.nqct * 32;
  return BTS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3x3 stride-1 convolution in Winograd form (F(2x2,3x3) over (z,y), x direct) -- the counterpart of
// conv_wino.hip for  dW[kz,ky,kx,c,k] = sum_v P[v + (kz-1,ky-1,kx-1)][c] * Q[v][k]:
//   dU[dx][xi] = sum over 2x2 (z,y) patches and x' of  V[xi](c; x') * T[xi](k; x' - dx + 1),
//   V = B^T d B (the forward's input transform, of P),  T = A dY A^T (of Q),  dg[.,.,dx] = G^T dU[dx] G.
// 48 accumulator tiles (3 x taps x 16 transform points) instead of 27, each fed a quarter as often: 12/27 of the matrix
// instructions.  The contraction runs over voxels, so channels sit on the lanes; for packed transforms every lane handles 4
// consecutive x of its channel: the LDS tiles are x-fastest ([row][channel][x], transposed while staging; the P rows have
// their 16-byte cells XOR-swizzled by (c >> 2) & 3 instead of padding).  Lanes 0-31 / 32-63 take two neighbouring x quads =
// the two voxels of an MFMA's K = 2, so one ds_read_b128 feeds four matrix instructions per (x tap, point).  The x tap only
// selects WHICH register of a 12-wide T window is the B operand (window value 5 + j - dx): V is formed once for all three taps.
// 4 waves = the 4 xi_z, each 3 taps x 4 xi_y = 12 accumulators (192 registers), one wave per SIMD; sub-tile 16 x 4 x 2 as
// the direct kernel's, so plan, partial layout ([27 taps][32][32] per workgroup, written after an in-LDS combine of the
// waves) and finalize kernels are shared.  Bias partials: wave 1 holds all four Q rows of a patch anyway.
// ---------------------------------------------------------------------------------------------------------------------
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wg_f32x2 wgw_pk_add(wg_f32x2 a, wg_f32x2 b) { wg_f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wg_f32x2 wgw_pk_sub(wg_f32x2 a, wg_f32x2 b) { wg_f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wg_f32x2 wgw_pk_fma(wg_f32x2 a, wg_f32x2 b, wg_f32x2 c) { wg_f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x4 wgw_add4(f32x4 a, f32x4 b) { const wg_f32x2 lo = wgw_pk_add(a.xy, b.xy), hi = wgw_pk_add(a.zw, b.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ f32x4 wgw_sub4(f32x4 a, f32x4 b) { const wg_f32x2 lo = wgw_pk_sub(a.xy, b.xy), hi = wgw_pk_sub(a.zw, b.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ f32x4 wgw_fma4(f32x4 a, wg_f32x2 s, f32x4 c) { const wg_f32x2 lo = wgw_pk_fma(a.xy, s, c.xy), hi = wgw_pk_fma(a.zw, s, c.zw); return f32x4{lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ float wgw_acc_rd(float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }

#define WGW_QSTR 28                          /* floats per (row, channel) of the Q tile: x = -1..16 at entries 3..20 */
#define WGW_PTILE (24 * 32 * 16)             /* 4 z-rows x 6 y-rows */
#define WGW_QTILE (8 * 32 * WGW_QSTR)        /* 2 z-rows x 4 y-rows */
#define WGW_BUF (WGW_PTILE + WGW_QTILE)      /* 19456 floats = 76 KB, double buffered */
#define WGW_NPS 12
#define WGW_NQS 5
#define WGW_THREADS 256

__global__ __launch_bounds__(WGW_THREADS, 1) void wgw_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = xi_z
  const int ch = lane & 31, hh = lane >> 5;
  const int pct = blockIdx.y, qct = blockIdx.z;
  const float* pP = p.p + pct * 32;
  const float* pQ = p.q + qct * 32;

  // ---- staging maps: slot id = tid + 256*i ; P: x = id & 15, channel quad = (id >> 4) & 7, row = id >> 7 = (tid >> 7) + 2i.
  //      Byte offsets relative to the tile origin are computed once; a slot outside the image gets a 2 GB offset = outside
  //      the buffer descriptor, the load returns zeros (border tiles only: 4 flag bits per slot against 4 per tile) ----
  const int sxx = tid & 15, scq = (tid >> 4) & 7, srow0 = tid >> 7;
  const int p_lds0 = (srow0 * 32 + scq * 4) * 16 + ((((sxx >> 2) ^ (scq & 3)) << 2) | (sxx & 3));
  unsigned poff[WGW_NPS], pfl0 = 0, pfl1 = 0;
#pragma unroll
  for (int i = 0; i < WGW_NPS; ++i) {
    const int row = srow0 + 2 * i;
    const int zr = row / 6, yr = row - zr * 6;
    poff[i] = (unsigned)(((zr * p.Hp + yr) * p.Wp + sxx) * p.ldp + scq * 4) * 4u;
    const unsigned fl = (zr == 0 ? 1u : 0u) | (zr == 3 ? 2u : 0u) | (yr == 0 ? 4u : 0u) | (yr == 5 ? 8u : 0u);
    if (i < 8) pfl0 |= fl << (4 * i); else pfl1 |= fl << (4 * (i - 8));
  }
  int q_lds[WGW_NQS];
  unsigned qoff[WGW_NQS], qfl = 0;
#pragma unroll
  for (int i = 0; i < WGW_NQS; ++i) {
    const int id = tid + WGW_THREADS * i;
    q_lds[i] = -1; qoff[i] = 0x80000000u;
    if (id < 1152) {
      const int xx = id % 18, kq = (id / 18) & 7, row = id / 144;   // x = x0 - 1 + xx ; row = oz*4 + oy
      q_lds[i] = WGW_PTILE + (row * 32 + kq * 4) * WGW_QSTR + xx + 3;
      qoff[i] = (unsigned)((((row >> 2) * p.Hq + (row & 3)) * p.Wq + xx) * p.ldq + kq * 4) * 4u;
      qfl |= ((xx == 0 ? 1u : 0u) | (xx == 17 ? 2u : 0u)) << (2 * i);
    }
  }
  // coordinates of the NEXT sub-tile to fetch (sub-tiles of a workgroup are consecutive: incremental, no division in the loop)
  int ftx, fty, ftz, fn;
  {
    int t = blockIdx.x * p.sub_per_wg;
    ftx = t % p.ntx; t /= p.ntx;
    fty = t % p.nty; t /= p.nty;
    ftz = t % p.ntz;
    fn = t / p.ntz;
  }
  // The 17 slots of the next sub-tile are fetched in four quarters (P 0-2 + Q 0-1 | P 3-5 + Q 2 | P 6-8 + Q 3 | P 9-11 + Q 4),
  // quarter k issued at step k of the current sub-tile and written to the OTHER LDS buffer one step later: the staging
  // registers of a quarter live for one step only and the ds_writes spread over the matrix instructions.
  f32x4 pre[WGW_NPS + WGW_NQS];
  __amdgpu_buffer_rsrc_t f_pr, f_qr;
  unsigned f_tm = 0, f_qm = 0;
  auto fetch_begin = [&]() {
    const int z0 = 2 * ftz, y0 = 4 * fty, x0 = 16 * ftx;
    const float* pb = pP + ((((long)fn * p.Dp + z0 - 1) * p.Hp + y0 - 1) * p.Wp + x0) * p.ldp;
    const float* qb = pQ + ((((long)fn * p.Dq + z0) * p.Hq + y0) * p.Wq + x0 - 1) * p.ldq;
    f_pr = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, 0x7fffffff, 0x00020000);
    f_qr = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, 0x7fffffff, 0x00020000);
    f_tm = (ftz == 0 ? 1u : 0u) | (ftz == p.ntz - 1 ? 2u : 0u) | (fty == 0 ? 4u : 0u) | (fty == p.nty - 1 ? 8u : 0u);
    f_qm = (ftx == 0 ? 1u : 0u) | (ftx == p.ntx - 1 ? 2u : 0u);
    if (++ftx == p.ntx) { ftx = 0; if (++fty == p.nty) { fty = 0; if (++ftz == p.ntz) { ftz = 0; ++fn; } } }
  };
  auto fetch_p = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int sh = 4 * (i & 7);
    const unsigned fl = (((i < 8 ? pfl0 : pfl1) >> sh) & 15u) & f_tm;
    pre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_pr, fl ? 0x80000000u : poff[i], 0, 0));
  };
  auto fetch_q = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const unsigned fl = ((qfl >> (2 * i)) & 3u) & f_qm;
    pre[WGW_NPS + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_qr, fl ? 0x80000000u : qoff[i], 0, 0));
  };
  auto commit_p = [&](float* buf, auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) buf[p_lds0 + i * (2 * 32 * 16) + e * 16] = pre[i][e];
  };
  auto commit_q = [&](float* buf, auto ic) {
    constexpr int i = decltype(ic)::value;
    if (q_lds[i] >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) buf[q_lds[i] + e * WGW_QSTR] = pre[WGW_NPS + i][e];
    }
  };
  auto fetch_quarter = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    fetch_p(IC<3 * k>{}); fetch_p(IC<3 * k + 1>{}); fetch_p(IC<3 * k + 2>{});
    if constexpr (k == 0) { fetch_q(IC<0>{}); fetch_q(IC<1>{}); } else fetch_q(IC<k + 1>{});
  };
  auto commit_quarter = [&](float* buf, auto kc) {
    constexpr int k = decltype(kc)::value;
    commit_p(buf, IC<3 * k>{}); commit_p(buf, IC<3 * k + 1>{}); commit_p(buf, IC<3 * k + 2>{});
    if constexpr (k == 0) { commit_q(buf, IC<0>{}); commit_q(buf, IC<1>{}); } else commit_q(buf, IC<k + 1>{});
  };

  // ---- wave roles: xi_z: 0: d0 - d2 | 1: d1 + d2 | 2: d2 - d1 | 3: d1 - d3 ;  c = P[ra] + s * P[rb] ----
  const int ra = (wave == 0) ? 0 : (wave == 2) ? 2 : 1;
  const int rb = (wave == 0) ? 2 : (wave == 1) ? 2 : (wave == 2) ? 1 : 3;
  const float sp = (wave == 1) ? 1.f : -1.f;
  const wg_f32x2 sp2 = {sp, sp};
  // T z-part: a = X + sq * Y over the two z-rows of Q:  0: q0 | 1: q0 + q1 | 2: q0 - q1 | 3: +q1 (the true -q1 is undone in the
  // tap combine) -- X = row oz 0 (wave 3: oz 1), Y = row oz 1, sq = 0 / +1 / -1 / 0: the same instructions for every wave
  const float sq = (wave == 1) ? 1.f : (wave == 2) ? -1.f : 0.f;
  const wg_f32x2 sq2 = {sq, sq};
  const int qX = (wave == 3) ? 4 * 32 * WGW_QSTR : 0;
  const int pswz = (ch >> 2) & 3;
  const int pA = (ra * 6 * 32 + ch) * 16, pB = (rb * 6 * 32 + ch) * 16;
  const bool bias = p.want_bias && pct == 0 && wave == 1;
  double bsum = 0.0;

  f32x16 acc[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  // operands of one step (patch py2, x-quad pair s): V = 4 quads (A operand), T = the middle quad of the 12-wide window in
  // full (window 4..7) and its two neighbours (window 3 and 8) as scalars
  struct WgwOps {
    f32x4 v[4];
    f32x4 a0, a1, u1, u2;
    float a0s[2], a1s[2], u1s[2], u2s[2];
  };
  auto form = [&](const float* cur, auto stc, WgwOps& o) {
    constexpr int st = decltype(stc)::value;
    constexpr int py2 = st >> 1, sx = st & 1;
    const int xq = 2 * sx + hh;               // logical x quad of this lane
    const int xo = 4 * xq;
    const int pq = (xq ^ pswz) << 2;          // physical cell inside the 16-float row
    f32x4 c[4];
#pragma unroll
    for (int yr = 0; yr < 4; ++yr) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(cur + pA + (2 * py2 + yr) * 512 + pq);
      const f32x4 b = *reinterpret_cast<const f32x4*>(cur + pB + (2 * py2 + yr) * 512 + pq);
      c[yr] = wgw_fma4(b, sp2, a);
    }
    o.v[0] = wgw_sub4(c[0], c[2]);
    o.v[1] = wgw_add4(c[1], c[2]);
    o.v[2] = wgw_sub4(c[2], c[1]);
    o.v[3] = wgw_sub4(c[1], c[3]);
    const float* qb = cur + WGW_PTILE + ((2 * py2) * 32 + ch) * WGW_QSTR + xo;
    {  // middle quad, packed
      const f32x4 x0q = *reinterpret_cast<const f32x4*>(qb + qX + 4);
      const f32x4 x1q = *reinterpret_cast<const f32x4*>(qb + qX + 32 * WGW_QSTR + 4);
      const f32x4 y0q = *reinterpret_cast<const f32x4*>(qb + (4 * 32) * WGW_QSTR + 4);
      const f32x4 y1q = *reinterpret_cast<const f32x4*>(qb + (5 * 32) * WGW_QSTR + 4);
      o.a0 = wgw_fma4(y0q, sq2, x0q);
      o.a1 = wgw_fma4(y1q, sq2, x1q);
      o.u1 = wgw_add4(o.a0, o.a1);
      o.u2 = wgw_sub4(o.a0, o.a1);
      if (bias) bsum += (double)((o.u1[0] + o.u1[1]) + (o.u1[2] + o.u1[3]));   // wave 1: the four Q rows of this lane's own 4 x
    }
    // window 3 and 8 (the x neighbours of the quad).  Thirty-two lanes reading ONE column of 32 channel rows with 4-byte reads hit
    // 8 banks (the row stride is a multiple of 4 floats; 4-byte reads bank on 32): those 4-way conflicts on eight reads per step
    // were 45 % of this kernel's LDS cycles (profiles/r02e_pmc_wino.txt).  Read the neighbouring CELLS with 16-byte reads instead
    // (conflict-free like the middle quad's: 16 lanes x 16 bytes cover the 64 banks once) and use one element of each.
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int off = e == 0 ? 0 : 8, el = e == 0 ? 3 : 0;
      f32x4 xs0 = *reinterpret_cast<const f32x4*>(qb + qX + off);
      f32x4 xs1 = *reinterpret_cast<const f32x4*>(qb + qX + 32 * WGW_QSTR + off);
      f32x4 ys0 = *reinterpret_cast<const f32x4*>(qb + (4 * 32) * WGW_QSTR + off);
      f32x4 ys1 = *reinterpret_cast<const f32x4*>(qb + (5 * 32) * WGW_QSTR + off);
      asm volatile("" : "+v"(xs0), "+v"(xs1), "+v"(ys0), "+v"(ys1));     // (keeps the four reads 16 bytes wide)
      o.a0s[e] = fmaf(ys0[el], sq, xs0[el]);
      o.a1s[e] = fmaf(ys1[el], sq, xs1[el]);
      o.u1s[e] = o.a0s[e] + o.a1s[e];
      o.u2s[e] = o.a0s[e] - o.a1s[e];
    }
  };
  auto mfma48 = [&](const WgwOps& o) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = 5 + j - dx;   // window value paired with this lane's x quad element j
        const float b0 = w == 3 ? o.a0s[0] : w == 8 ? o.a0s[1] : o.a0[(w - 4) & 3];
        const float b1 = w == 3 ? o.u1s[0] : w == 8 ? o.u1s[1] : o.u1[(w - 4) & 3];
        const float b2 = w == 3 ? o.u2s[0] : w == 8 ? o.u2s[1] : o.u2[(w - 4) & 3];
        const float b3 = w == 3 ? o.a1s[0] : w == 8 ? o.a1s[1] : o.a1[(w - 4) & 3];   // true t3 = -a1
        acc[dx][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.v[0][j], b0, acc[dx][0], 0, 0, 0);
        acc[dx][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.v[1][j], b1, acc[dx][1], 0, 0, 0);
        acc[dx][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.v[2][j], b2, acc[dx][2], 0, 0, 0);
        acc[dx][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.v[3][j], b3, acc[dx][3], 0, 0, 0);
      }
  };
  // one slot of the schedule per matrix instruction: the MFMA, then up to two vector-ALU operations and one memory operation
  // of the work that shares the step (next step's operands, the staging quarter)
#define WGW_SCHED_STEP()                                   \
  _Pragma("unroll") for (int q_ = 0; q_ < 48; ++q_) {      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x320, 1, 0);     \
  }

  const int t0 = blockIdx.x * p.sub_per_wg;
  int t1 = t0 + p.sub_per_wg;
  if (t1 > p.nsub) t1 = p.nsub;
  if (t0 < t1) {
    fetch_begin();
    fetch_quarter(IC<0>{}); fetch_quarter(IC<1>{}); fetch_quarter(IC<2>{}); fetch_quarter(IC<3>{});
    commit_quarter(lds, IC<0>{}); commit_quarter(lds, IC<1>{}); commit_quarter(lds, IC<2>{}); commit_quarter(lds, IC<3>{});
  }
  __syncthreads();
  WgwOps opA, opB;
  for (int t = t0; t < t1; ++t) {
    const float* cur = lds + ((t - t0) & 1) * WGW_BUF;
    float* nxt = lds + ((t - t0 + 1) & 1) * WGW_BUF;
    const bool more = (t + 1) < t1;
    form(cur, IC<0>{}, opA);
    if (more) fetch_begin();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3" ::: "memory");   // hand-written VALU -> MFMA operand: the compiler does not track that hazard
    // step 0: MFMA(opA) | form step 1 | fetch quarter 0
    if (more) fetch_quarter(IC<0>{});
    form(cur, IC<1>{}, opB);
    mfma48(opA);
    WGW_SCHED_STEP();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    // step 1: MFMA(opB) | form step 2 | commit quarter 0, fetch quarter 1
    if (more) { commit_quarter(nxt, IC<0>{}); fetch_quarter(IC<1>{}); }
    form(cur, IC<2>{}, opA);
    mfma48(opB);
    WGW_SCHED_STEP();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    // step 2: MFMA(opA) | form step 3 | commit quarter 1, fetch quarter 2
    if (more) { commit_quarter(nxt, IC<1>{}); fetch_quarter(IC<2>{}); }
    form(cur, IC<3>{}, opB);
    mfma48(opA);
    WGW_SCHED_STEP();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    // step 3: MFMA(opB) | commit quarter 2, fetch quarter 3
    if (more) { commit_quarter(nxt, IC<2>{}); fetch_quarter(IC<3>{}); }
    mfma48(opB);
    WGW_SCHED_STEP();
    __builtin_amdgcn_sched_barrier(0);
    if (more) commit_quarter(nxt, IC<3>{});
    __syncthreads();
  }

  // ---- G^T dU G: y part in the wave (u3 carries the opposite sign) -> LDS [xi_z][dx*3+ky][c][k]; z part across the waves ----
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float* xl = lds + wave * 9 * 1024;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float u0 = wgw_acc_rd(acc[dx][0][r]), u1 = wgw_acc_rd(acc[dx][1][r]), u2 = wgw_acc_rd(acc[dx][2][r]),
                  u3 = wgw_acc_rd(acc[dx][3][r]);
      const float hs = 0.5f * (u1 + u2), hd = 0.5f * (u1 - u2);
      const int crow = 8 * (r >> 2) + 4 * hh + (r & 3);
      float* o = xl + (dx * 3) * 1024 + crow * 32 + ch;
      o[0] = u0 + hs;
      o[1024] = hd;
      o[2048] = hs - u3;
    }
  __syncthreads();
  // taps (kz, ky, dx): kz 0: L0 + .5 L1 + .5 L2 | kz 1: .5 L1 - .5 L2 | kz 2: .5 L1 + .5 L2 - L3   (L3 accumulated with +q1)
  float* out = p.partial + ((((long)blockIdx.x * gridDim.y + pct) * gridDim.z + qct) * 27) * 1024;
  for (int e = tid; e < 9 * 1024; e += WGW_THREADS) {
    const int g = e >> 10, ck = e & 1023;       // g = dx*3 + ky
    const int dx = g / 3, ky = g - dx * 3;
    const float l0 = lds[(0 * 9 + g) * 1024 + ck], l1 = lds[(1 * 9 + g) * 1024 + ck], l2 = lds[(2 * 9 + g) * 1024 + ck],
                l3 = lds[(3 * 9 + g) * 1024 + ck];
    const float hs = 0.5f * (l1 + l2), hd = 0.5f * (l1 - l2);
    out[((0 * 3 + ky) * 3 + dx) * 1024 + ck] = l0 + hs;
    out[((1 * 3 + ky) * 3 + dx) * 1024 + ck] = hd;
    out[((2 * 3 + ky) * 3 + dx) * 1024 + ck] = hs - l3;
  }
  if (bias) {
    const double o = __shfl_xor(bsum, 32, 64);   // the two x-quad halves of the wave
    if (lane < 32) p.partial_b[((long)blockIdx.x * gridDim.z + qct) * 32 + ch] = bsum + o;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 1x1x1 convs (the ResNet blocks' shortcut convs, resnet.py:96-103, and the decoder / VAE projections):
// dW[c][k] = sum_v P[v][c] * Q[v][k] -- a 32x32 tile per (P, Q) channel-group pair with K = all voxels: 2 FLOP per loaded
// byte, HBM-bound.  The staged kernel (LDS DMA, a barrier per 256-voxel sub-tile, 8 matrix instructions between barriers) ran
// these at ~2 TB/s.  Here nothing is staged: the MFMA operand layout (lane = channel, register = voxel of the K pair) is
// exactly a coalesced 128-byte row read, so each lane streams its channel of consecutive voxels with 16 independent 4-byte
// loads in flight and feeds them to the matrix pipe directly; 8 waves per workgroup split the workgroup's voxel range and
// are summed in LDS in fixed order.  Bias partials (column sums of Q) fall out of the B operands.
// ---------------------------------------------------------------------------------------------------------------------
#define K1W_THREADS 512
#define K1W_U 8
__global__ __launch_bounds__(K1W_THREADS, 2) void k1w_kernel(const WgradParams p, long NV, long chunk) {
  __shared__ float lds[8 * 1024];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = lane & 31, hh = lane >> 5;
  const int pct = blockIdx.y, qct = blockIdx.z;
  const int pc = pct * 32 + ch, qc = qct * 32 + ch;
  const bool pok = pc < p.Cp, qok = qc < p.Cq;
  const long w0 = (long)blockIdx.x * chunk;
  long w1 = w0 + chunk;
  if (w1 > NV) w1 = NV;
  // this wave's part of the workgroup range, whole K pairs
  long per = ((w1 - w0 + 7) / 8 + 1) & ~1L;
  long v = w0 + wave * per;
  long ve = v + per;
  if (ve > w1) ve = w1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  double bsum = 0.0;
  const bool bias = p.want_bias && pct == 0;
  const float* pp = p.p + pc;
  const float* qq = p.q + qc;
  for (; v < ve; v += 2 * K1W_U) {
    float a[K1W_U], b[K1W_U];
#pragma unroll
    for (int u = 0; u < K1W_U; ++u) {
      const long vv = v + 2 * u + hh;
      const bool in = vv < ve;
      a[u] = (in && pok) ? pp[vv * p.ldp] : 0.f;
      b[u] = (in && qok) ? qq[vv * p.ldq] : 0.f;
    }
    float bs = 0.f;
#pragma unroll
    for (int u = 0; u < K1W_U; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
      bs += b[u];
    }
    if (bias) bsum += (double)bs;
  }
  // fixed-order sum of the 8 waves' tiles
#pragma unroll
  for (int r = 0; r < 16; ++r) lds[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + ch] = acc[r];
  __syncthreads();
  float* dst = p.partial + ((((long)blockIdx.x * gridDim.y + pct) * gridDim.z + qct) * p.ntiles) * 1024;
  for (int e = tid; e < 1024; e += K1W_THREADS) {
    float sacc = lds[e];
#pragma unroll
    for (int w = 1; w < 8; ++w) sacc += lds[w * 1024 + e];
    dst[e] = sacc;
  }
  if (bias) {
    __syncthreads();
    double* shd = reinterpret_cast<double*>(lds);
    shd[tid] = bsum;   // [wave][hh][ch]
    __syncthreads();
    if (tid < 32) {
      double t = 0.0;
      for (int k = 0; k < 16; ++k) t += shd[k * 32 + tid];
      p.partial_b[((long)blockIdx.x * gridDim.z + qct) * 32 + tid] = t;
    }
  }
}

struct WgradRoles {
  int ntaps, s, neg, swapped, transposed;
  int Dp, Hp, Wp, Cp, Dq, Hq, Wq, Cq;
};

static int wgrad_roles(WgradRoles& r, int kind, int D, int H, int W, int Cin, int Cout) {
  r.ntaps = (kind == BTS_CONV_K1) ? 1 : 27;
  r.s = 1; r.neg = 0; r.swapped = 0; r.transposed = 0;
  r.Dp = D; r.Hp = H; r.Wp = W; r.Cp = Cin; r.Dq = D; r.Hq = H; r.Wq = W; r.Cq = Cout;
  if (kind == BTS_CONV_K3S2) {
    if ((D | H | W) & 1) return BTS_ERR_SHAPE;
    r.s = 2; r.Dq = D / 2; r.Hq = H / 2; r.Wq = W / 2;
  } else if (kind == BTS_CONV_K3S2T) {  // P = dy on the fine grid (2D), Q = x on the coarse grid
    r.transposed = 1;
    r.s = 2; r.Dp = 2 * D; r.Hp = 2 * H; r.Wp = 2 * W; r.Cp = Cout; r.Cq = Cin;
  } else if (kind == BTS_CONV_K3S1 && Cout <= 8 && Cin > Cout) {
    r.swapped = 1; r.neg = 1; r.Cp = Cout; r.Cq = Cin;  // P = dy (tiny), Q = x
  }
  return BTS_OK;
}

extern "C" int bts_colsum(const float* x, float* out, void* workspace, long workspace_bytes, int N, long rows, int C,
                          int ld, float scale, int sum_over_n, int accumulate, hipStream_t stream);
extern "C" long bts_colsum_workspace(int N, long rows, int C);

// workspace bytes for bts_conv3d_bwd_weight; (D,H,W) are the forward INPUT dims, Cin the slab (folded) count
extern "C" long bts_conv3d_bwd_weight_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  WgradRoles r;
  if (wgrad_roles(r, kind, D, H, W, Cin, Cout) != BTS_OK) return -1;
  WgradPlan pl;
  if (plan_wgrad(pl, r.ntaps, r.s, r.neg, N, r.Dp, r.Hp, r.Wp, r.Cp, r.Dq, r.Hq, r.Wq, r.Cq) != BTS_OK) return -1;
  const long base = ((pl.partial_floats * 4 + pl.partial_b_doubles * 8 + 256 + 15) & ~15L);
  return base + (r.swapped ? bts_colsum_workspace(N, (long)D * H * W, Cout) : 0) + 128;
}

// dw is in the reference layout with Cin_ref = Cin + dup_shift input channels; db may be null.
extern "C" int bts_conv3d_bwd_weight(int kind, const float* x, const float* dy, float* dw, float* db, void* workspace,
                                     long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout,
                                     int lddy, int dup_start, int dup_shift, int accumulate, hipStream_t stream) {
  WgradRoles ro;
  int r = wgrad_roles(ro, kind, D, H, W, Cin, Cout);
  if (r != BTS_OK) return r;
  if (dup_shift > 0 && (kind == BTS_CONV_K3S2 || kind == BTS_CONV_K3S2T)) return BTS_ERR_UNSUPPORTED;
  WgradPlan pl;
  r = plan_wgrad(pl, ro.ntaps, ro.s, ro.neg, N, ro.Dp, ro.Hp, ro.Wp, ro.Cp, ro.Dq, ro.Hq, ro.Wq, ro.Cq);
  if (r != BTS_OK) return r;
  const long need = bts_conv3d_bwd_weight_workspace(kind, N, D, H, W, Cin, Cout);
  if (workspace_bytes < need || workspace == nullptr) return BTS_ERR_WORKSPACE;
  WgradParams& p = pl.p;
  const bool pIsDy = ro.transposed || ro.swapped;
  p.p = pIsDy ? dy : x;
  p.q = pIsDy ? x : dy;
  p.ldp = pIsDy ? lddy : ldx;
  p.ldq = pIsDy ? ldx : lddy;
  p.partial = reinterpret_cast<float*>(workspace);
  uintptr_t pb = (uintptr_t)(p.partial + pl.partial_floats);
  pb = (pb + 15) & ~(uintptr_t)15;
  p.partial_b = reinterpret_cast<double*>(pb);
  // bias gradient = column sums of dy: dy is Q in the plain form only
  p.want_bias = (db != nullptr && !pIsDy) ? 1 : 0;
#ifdef BTS_TIMING_EXPERIMENTS
  { const char* e = getenv("BTS_WGRAD_DBG"); p.dbg = e ? atoi(e) : 0; }
#else
  p.dbg = 0;
#endif
  typedef void (*WgKernel)(const WgradParams);
  static const WgKernel kernels[4] = {wgrad_kernel<0, 0>, wgrad_kernel<1, 0>, wgrad_kernel<2, 1>, wgrad_kernel<2, 2>};
  static bool attr_done = false;
  if (!attr_done) {
    for (int i = 0; i < 4; ++i) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    attr_done = true;
  }
  // 64 zero bytes at the tail of the workspace feed the padding lanes of the LDS-DMA staging
  char* ztail = reinterpret_cast<char*>(workspace) + ((need - 64) & ~15L);
  p.zeros = reinterpret_cast<const float*>(ztail);
  const bool glds = (p.ldp % 4 == 0) && (p.Cp % 4 == 0) && ((((uintptr_t)p.p) & 15) == 0) && (p.ldq % 4 == 0) &&
                    (p.Cq % 4 == 0) && ((((uintptr_t)p.q) & 15) == 0) && getenv("BTS_WGRAD_NOGLDS") == nullptr;
  if (p.fastf && ((double)p.IZ * p.Hp * p.Wp * p.ldp * 4.0 >= 4.0e9 || (double)p.TZ * p.Hq * p.Wq * p.ldq * 4.0 >= 4.0e9 || !glds))
    p.fastf = p.fixg = 0;
  const bool prof = bts_prof_on();
  if (glds && !p.fastf) {  // only the general LDS-DMA staging reads the zero source
    hipError_t e = hipMemsetAsync(ztail, 0, 64, stream);
    if (e != hipSuccess) return (int)e;
  }
  const int kidx = !glds ? 0 : (p.fastf ? 1 + p.fixg : 1);
  // stride-1 whole-tile layers: Winograd form (BTS_WGW=0: the direct fixed-geometry kernel)
  bool use_wgw = false;
  if (kidx == 2 && ro.ntaps == 27 && ro.s == 1) {
    const char* e = getenv("BTS_WGW");
    use_wgw = !(e && atoi(e) == 0) && 5.0 * p.Hp * p.Wp * p.ldp * 4.0 < 2.0e9 && 3.0 * p.Hq * p.Wq * p.ldq * 4.0 < 2.0e9;  // 31-bit tile offsets
  }
  // 1x1x1 convs on big grids: streaming kernel (BTS_K1W=0: the staged kernel)
  bool use_k1w = false;
  const long NVall = (long)N * ro.Dq * ro.Hq * ro.Wq;
  if (ro.ntaps == 1 && !pIsDy && p.ntiles == 1 && NVall >= 32768) {
    const char* e = getenv("BTS_K1W");
    use_k1w = !(e && atoi(e) == 0);
  }
  if (use_k1w) {
    const long chunk = (((NVall + pl.nsp - 1) / pl.nsp) + 15) & ~15L;
    if (prof) bts_prof_begin(26, 2.0 * (double)ro.Cp * ro.Cq * (double)NVall, stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k1w_kernel, dim3(pl.nsp, pl.npct, pl.nqct), dim3(K1W_THREADS), 0, stream, p, NVall, chunk);
    if (prof) bts_prof_end(stream);
    BTS_LAUNCH_CHECK();
  } else if (use_wgw) {
    static bool wattr = false;
    if (!wattr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      wattr = true;
    }
    if (prof) bts_prof_begin(24, 2.0 * ro.ntaps * (double)ro.Cp * ro.Cq * (double)N * ro.Dq * ro.Hq * ro.Wq, stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(wgw_kernel, dim3(pl.nsp, pl.npct, pl.nqct), dim3(WGW_THREADS), (size_t)2 * WGW_BUF * sizeof(float), stream, p);
    if (prof) bts_prof_end(stream);
    BTS_LAUNCH_CHECK();
  } else {
  static const int ksym[4] = {100, 104, 109, 110};  // 100 + (MODE << 2 | FIXG)
  if (prof) bts_prof_begin(ksym[kidx], 2.0 * ro.ntaps * (double)ro.Cp * ro.Cq * (double)N * ro.Dq * ro.Hq * ro.Wq, stream);
  (void)hipGetLastError(); hipLaunchKernelGGL(kernels[kidx], dim3(pl.nsp, pl.npct, pl.nqct), dim3(WG_THREADS), pl.shmem, stream, p);
  if (prof) bts_prof_end(stream);
  BTS_LAUNCH_CHECK();
  }
  WfinParams f;
  f.partial = p.partial; f.partial_b = p.partial_b; f.dw = dw; f.db = p.want_bias ? db : nullptr;
  f.nsp = pl.nsp; f.npct = pl.npct; f.nqct = pl.nqct; f.ntaps = ro.ntaps; f.ntiles = p.ntiles; f.cpad = p.cpad;
  f.Cp = ro.Cp; f.Cq = ro.Cq;
  const int Cin_ref = Cin + dup_shift;
  f.sT = (long)Cin_ref * Cout;
  if (ro.transposed) { f.sP = Cin_ref; f.sQ = 1; f.fold_axis = 0; }   // (t, Cout, Cin): P = cout, Q = cin
  else if (ro.swapped) { f.sP = 1; f.sQ = Cout; f.fold_axis = 2; }     // (t, Cin, Cout): P = cout, Q = cin
  else { f.sP = Cout; f.sQ = 1; f.fold_axis = 1; }                     // (t, Cin, Cout): P = cin, Q = cout
  f.shift = dup_shift;
  f.dup_start = dup_shift > 0 ? dup_start : (1 << 30);
  f.accum = accumulate;
  long rows = (long)ro.ntaps * ro.Cp * pl.nqct;
  if (pl.nsp <= 16) {
    long blocks = (rows * 32 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    (void)hipGetLastError(); hipLaunchKernelGGL(wgrad_finalize_flat_kernel, dim3((int)blocks), dim3(256), 0, stream, f);
  } else {
    int blocks = (int)(rows < 4096 ? rows : 4096);
    (void)hipGetLastError(); hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, stream, f);
  }
  BTS_LAUNCH_CHECK();
  if (db != nullptr && ro.swapped) {  // bias gradient of the swapped form: plain column sum of dy
    char* cw = reinterpret_cast<char*>(workspace) + ((pl.partial_floats * 4 + pl.partial_b_doubles * 8 + 256 + 15) & ~15L);
    const long rowsv = (long)D * H * W;
    return bts_colsum(dy, db, cw, bts_colsum_workspace(N, rowsv, Cout), N, rowsv, Cout, lddy, 1.0f, 1, accumulate, stream);
  }
  return BTS_OK;
}
