// Weight (and bias) gradients of the 3-D convolutions on the exact-fp32 matrix pipe.
// The reference obtains these from TF autodiff (train.py:142-151); the math is SURVEY Appendix A':
//   dW[t][pc][qc] = sum_{n,v} P[n, v*s + off_t, pc] * Q[n, v, qc]        db[qc] = sum_{n,v} Q[n, v, qc]
//     Conv3D k1 / k3s1 / k3s2 : P = layer input x, Q = dy  -> dW in (t, Cin, Cout)
//     Conv3DTranspose k3s2    : P = dy (fine grid, s=2), Q = x (coarse grid) -> dW in (t, Cout, Cin)
// GEMM view per tap: M = 32 P-channels, N = 32 Q-channels, K = voxels (v_mfma_f32_32x32x2_f32, 2 voxels per
// instruction). A workgroup walks a run of spatial sub-tiles; the 4 waves split the 27 taps (or, for the 1x1x1
// conv, split the voxels). P halo tile and Q tile are staged in LDS as [voxel][32 ch] so both fragment reads are
// conflict-free ds_read_b32. Per-workgroup partial sums go to a workspace and are combined in a fixed order by
// the finalize kernel (bitwise reproducible; no float atomics).
#include "common.h"
#include "bts_internal.h"

#define WG_MAXT 7

struct WgradParams {
  const float* p;
  const float* q;
  float* partial;     // [nsp][pct][qct][ntaps][32][32]
  double* partial_b;  // [nsp][qct][32]  (bias partials; only pct==0 workgroups write)
  int N, Dp, Hp, Wp, Cp, ldp;
  int Dq, Hq, Wq, Cq, ldq;
  int s, loz, loy, lox;
  int IZ, IY, IX;
  int lgTX, lgTY, TZ;
  int ntz, nty, ntx;
  int ntaps;
  int nsub, sub_per_wg;
  int want_bias;
  int tap_lds[27];
};

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  const int pct = blockIdx.y, qct = blockIdx.z;
  const int TX = 1 << p.lgTX, TY = 1 << p.lgTY;
  const int M = TX * TY * p.TZ;
  const int tileVoxP = p.IZ * p.IY * p.IX;
  float* ldsP = lds;
  float* ldsQ = lds + tileVoxP * 32;
  const bool ksplit = p.ntaps < 4;

  f32x16 acc[WG_MAXT];
#pragma unroll
  for (int i = 0; i < WG_MAXT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  double bsum = 0.0;  // bias column sum: thread (c = tid&31, part = tid>>5)

  const int sub0 = blockIdx.x * p.sub_per_wg;
  int sub1 = sub0 + p.sub_per_wg;
  if (sub1 > p.nsub) sub1 = p.nsub;
  const int vecP = (p.ldp % 4 == 0) && (p.Cp % 4 == 0) && ((((uintptr_t)p.p) & 15) == 0);
  const int vecQ = (p.ldq % 4 == 0) && (p.Cq % 4 == 0) && ((((uintptr_t)p.q) & 15) == 0);

  for (int sub = sub0; sub < sub1; ++sub) {
    int b = sub;
    const int tx = b % p.ntx; b /= p.ntx;
    const int ty = b % p.nty; b /= p.nty;
    const int tz = b % p.ntz;
    const int n = b / p.ntz;
    const int oz0 = tz * p.TZ, oy0 = ty * TY, ox0 = tx * TX;
    const int iz0 = oz0 * p.s + p.loz, iy0 = oy0 * p.s + p.loy, ix0 = ox0 * p.s + p.lox;

    __syncthreads();  // previous sub-tile fully consumed
    // ---- stage P halo tile: [voxel][32] ----
    for (int e = tid; e < tileVoxP * 8; e += 256) {
      const int vox = e >> 3, qd = e & 7;
      const int vx = vox % p.IX;
      const int r = vox / p.IX;
      const int vy = r % p.IY, vz = r / p.IY;
      const int gz = iz0 + vz, gy = iy0 + vy, gx = ix0 + vx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int c = pct * 32 + qd * 4;
      if (gz >= 0 && gz < p.Dp && gy >= 0 && gy < p.Hp && gx >= 0 && gx < p.Wp && c < p.Cp) {
        const float* src = p.p + ((((long)n * p.Dp + gz) * p.Hp + gy) * p.Wp + gx) * (long)p.ldp + c;
        if (vecP) v = *reinterpret_cast<const f32x4*>(src);
        else {
          v[0] = src[0];
          if (c + 1 < p.Cp) v[1] = src[1];
          if (c + 2 < p.Cp) v[2] = src[2];
          if (c + 3 < p.Cp) v[3] = src[3];
        }
      }
      *reinterpret_cast<f32x4*>(ldsP + vox * 32 + qd * 4) = v;
    }
    // ---- stage Q tile: [m][32]; voxels outside the grid are zero so they contribute nothing ----
    for (int e = tid; e < M * 8; e += 256) {
      const int m = e >> 3, qd = e & 7;
      const int mx = m & (TX - 1);
      const int my = (m >> p.lgTX) & (TY - 1);
      const int mz = m >> (p.lgTX + p.lgTY);
      const int gz = oz0 + mz, gy = oy0 + my, gx = ox0 + mx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int c = qct * 32 + qd * 4;
      if (gz < p.Dq && gy < p.Hq && gx < p.Wq && c < p.Cq) {
        const float* src = p.q + ((((long)n * p.Dq + gz) * p.Hq + gy) * p.Wq + gx) * (long)p.ldq + c;
        if (vecQ) v = *reinterpret_cast<const f32x4*>(src);
        else {
          v[0] = src[0];
          if (c + 1 < p.Cq) v[1] = src[1];
          if (c + 2 < p.Cq) v[2] = src[2];
          if (c + 3 < p.Cq) v[3] = src[3];
        }
      }
      *reinterpret_cast<f32x4*>(ldsQ + m * 32 + qd * 4) = v;
    }
    __syncthreads();

    if (p.want_bias && pct == 0) {
      const int c = tid & 31, part = tid >> 5;
      float s = 0.f;
      for (int m = part; m < M; m += 8) s += ldsQ[m * 32 + c];
      bsum += (double)s;
    }

    const int nsteps = M >> 1;
    if (!ksplit) {
      for (int st = 0; st < nsteps; ++st) {
        const int m = 2 * st + h;
        const int mx = m & (TX - 1);
        const int my = (m >> p.lgTX) & (TY - 1);
        const int mz = m >> (p.lgTX + p.lgTY);
        const int poff = ((mz * p.s * p.IY + my * p.s) * p.IX + mx * p.s) * 32 + l32;
        const float qv = ldsQ[m * 32 + l32];
#pragma unroll
        for (int i = 0; i < WG_MAXT; ++i) {
          const int t = wave + 4 * i;
          if (t < p.ntaps) {
            const float pv = ldsP[poff + p.tap_lds[t]];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, qv, acc[i], 0, 0, 0);
          }
        }
      }
    } else {
      // 1x1x1 conv (ntaps==1..3): waves split the voxel pairs; every wave handles all taps
      for (int st = wave; st < nsteps; st += 4) {
        const int m = 2 * st + h;
        const int mx = m & (TX - 1);
        const int my = (m >> p.lgTX) & (TY - 1);
        const int mz = m >> (p.lgTX + p.lgTY);
        const int poff = ((mz * p.s * p.IY + my * p.s) * p.IX + mx * p.s) * 32 + l32;
        const float qv = ldsQ[m * 32 + l32];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i < p.ntaps) {
            const float pv = ldsP[poff + p.tap_lds[i]];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, qv, acc[i], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- write partials ----
  const long tileBase = ((((long)blockIdx.x * gridDim.y + pct) * gridDim.z + qct) * p.ntaps) * 1024;
  if (!ksplit) {
#pragma unroll
    for (int i = 0; i < WG_MAXT; ++i) {
      const int t = wave + 4 * i;
      if (t < p.ntaps) {
        float* dst = p.partial + tileBase + (long)t * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // P channel
          dst[row * 32 + l32] = acc[i][r];
        }
      }
    }
  } else {
    // cross-wave fixed-order reduction through LDS (reuse the staging buffers)
    for (int i = 0; i < p.ntaps; ++i) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = 0.f;
        // static register selection
        if (i == 0) v = acc[0][r];
        else if (i == 1) v = acc[1][r];
        else v = acc[2][r];
        lds[wave * 1024 + row * 32 + l32] = v;
      }
      __syncthreads();
      float* dst = p.partial + tileBase + (long)i * 1024;
      for (int e = tid; e < 1024; e += 256) dst[e] = ((lds[e] + lds[1024 + e]) + lds[2048 + e]) + lds[3072 + e];
    }
  }
  if (p.want_bias && pct == 0) {
    __syncthreads();
    double* shd = reinterpret_cast<double*>(lds);
    shd[tid] = bsum;
    __syncthreads();
    if (tid < 32) {
      double s = 0.0;
      for (int part = 0; part < 8; ++part) s += shd[part * 32 + tid];
      p.partial_b[((long)blockIdx.x * gridDim.z + qct) * 32 + tid] = s;
    }
  }
}

struct WfinParams {
  const float* partial;
  const double* partial_b;
  float* dw;
  float* db;
  int nsp, npct, nqct, ntaps, Cp, Cq;
  long sT, sP, sQ;
  int fold_on_p;  // 1: the P-channel axis is the (possibly folded) reference Cin axis
  int shift, dup_start;
  int accum;
};

// dW[t][r(pc)][qc] (+)= sum_wg partial ; folded slab channel pc maps to reference channel pc+shift and, if
// pc >= dup_start, also to pc-dup_start (both copies of the duplicated slice see the same input: encoder.py:83-87).
__global__ void wgrad_finalize_kernel(const WfinParams f) {
  const long total = (long)f.ntaps * f.Cp * f.Cq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int qc = (int)(i % f.Cq);
    long r = i / f.Cq;
    const int pc = (int)(r % f.Cp);
    const int t = (int)(r / f.Cp);
    const int pct = pc >> 5, qct = qc >> 5;
    const long off = (((long)pct * f.nqct + qct) * f.ntaps + t) * 1024 + (pc & 31) * 32 + (qc & 31);
    const long wgStride = (long)f.npct * f.nqct * f.ntaps * 1024;
    float s = 0.f;
    for (int w = 0; w < f.nsp; ++w) s += f.partial[w * wgStride + off];
    const int pr = f.fold_on_p ? pc + f.shift : pc;
    float* d1 = f.dw + t * f.sT + pr * f.sP + qc * f.sQ;
    *d1 = f.accum ? (*d1 + s) : s;
    if (f.fold_on_p && f.shift > 0 && pc >= f.dup_start) {
      float* d2 = f.dw + t * f.sT + (pc - f.dup_start) * f.sP + qc * f.sQ;
      *d2 = f.accum ? (*d2 + s) : s;
    }
  }
  if (f.db) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < f.Cq; i += (long)gridDim.x * blockDim.x) {
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

static int plan_wgrad(WgradPlan& pl, int ntaps, int s, int N, int Dp, int Hp, int Wp_, int Cp, int Dq, int Hq, int Wq,
                      int Cq) {
  WgradParams& p = pl.p;
  p.N = N; p.Dp = Dp; p.Hp = Hp; p.Wp = Wp_; p.Cp = Cp;
  p.Dq = Dq; p.Hq = Hq; p.Wq = Wq; p.Cq = Cq;
  p.s = s; p.ntaps = ntaps;
  int lo = 0, span = 1;
  if (ntaps == 27 && s == 1) { lo = -1; span = 3; }
  if (ntaps == 27 && s == 2) { lo = 0; span = 3; }
  p.loz = p.loy = p.lox = lo;
  // sub-tile: 128 voxels for s=1 (16x4x2), 32 voxels for s=2 (8x4x1); 1x1x1: 256 voxels (32x4x2)
  int TX, TY, TZ;
  if (ntaps == 1) { TX = 32; TY = 4; TZ = 2; }
  else if (s == 1) { TX = 16; TY = 4; TZ = 2; }
  else { TX = 8; TY = 4; TZ = 1; }
  while (TX > 2 && TX / 2 >= Wq) { TX /= 2; if (TY * 2 <= 8) TY *= 2; else TZ *= 2; }
  while (TY > 1 && TY / 2 >= Hq) { TY /= 2; TZ *= 2; }
  p.lgTX = ilog2w(TX); p.lgTY = ilog2w(TY); p.TZ = TZ;
  p.ntx = (Wq + TX - 1) / TX; p.nty = (Hq + TY - 1) / TY; p.ntz = (Dq + TZ - 1) / TZ;
  p.IX = (TX - 1) * s + span; p.IY = (TY - 1) * s + span; p.IZ = (TZ - 1) * s + span;
  const int tileVoxP = p.IZ * p.IY * p.IX;
  const int M = TX * TY * TZ;
  pl.shmem = (size_t)(tileVoxP + M) * 32 * sizeof(float);
  if (pl.shmem < 4 * 1024 * sizeof(float)) pl.shmem = 4 * 1024 * sizeof(float);
  if (pl.shmem > 160 * 1024) return BTS_ERR_SHAPE;
  for (int t = 0; t < 27; ++t) p.tap_lds[t] = 0;
  if (ntaps == 27)
    for (int t = 0; t < 27; ++t) {
      const int oz = t / 9 + (s == 1 ? -1 : 0), oy = (t / 3) % 3 + (s == 1 ? -1 : 0), ox = t % 3 + (s == 1 ? -1 : 0);
      p.tap_lds[t] = (((oz - lo) * p.IY + (oy - lo)) * p.IX + (ox - lo)) * 32;
    }
  p.nsub = N * p.ntz * p.nty * p.ntx;
  pl.npct = (Cp + 31) / 32;
  pl.nqct = (Cq + 31) / 32;
  long want = 1024 / ((long)pl.npct * pl.nqct);
  if (want < 1) want = 1;
  if (want > p.nsub) want = p.nsub;
  p.sub_per_wg = (int)((p.nsub + want - 1) / want);
  pl.nsp = (p.nsub + p.sub_per_wg - 1) / p.sub_per_wg;
  pl.partial_floats = (long)pl.nsp * pl.npct * pl.nqct * ntaps * 1024;
  pl.partial_b_doubles = (long)pl.nsp * pl.nqct * 32;
  return BTS_OK;
}

static int wgrad_dims(int kind, int D, int H, int W, int Cin, int Cout, int& ntaps, int& s, int& Dp, int& Hp, int& Wp_,
                      int& Cp, int& Dq, int& Hq, int& Wq, int& Cq) {
  ntaps = (kind == BTS_CONV_K1) ? 1 : 27;
  s = 1;
  Dp = D; Hp = H; Wp_ = W; Cp = Cin; Dq = D; Hq = H; Wq = W; Cq = Cout;
  if (kind == BTS_CONV_K3S2) {
    if ((D | H | W) & 1) return BTS_ERR_SHAPE;
    s = 2; Dq = D / 2; Hq = H / 2; Wq = W / 2;
  } else if (kind == BTS_CONV_K3S2T) {  // P = dy on the fine grid (2D), Q = x on the coarse grid
    s = 2; Dp = 2 * D; Hp = 2 * H; Wp_ = 2 * W; Cp = Cout; Cq = Cin;
  }
  return BTS_OK;
}

// workspace bytes for bts_conv3d_bwd_weight; (D,H,W) are the forward INPUT dims, Cin the slab (folded) count
extern "C" long bts_conv3d_bwd_weight_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout) {
  int ntaps, s, Dp, Hp, Wp_, Cp, Dq, Hq, Wq, Cq;
  if (wgrad_dims(kind, D, H, W, Cin, Cout, ntaps, s, Dp, Hp, Wp_, Cp, Dq, Hq, Wq, Cq) != BTS_OK) return -1;
  WgradPlan pl;
  if (plan_wgrad(pl, ntaps, s, N, Dp, Hp, Wp_, Cp, Dq, Hq, Wq, Cq) != BTS_OK) return -1;
  return pl.partial_floats * 4 + pl.partial_b_doubles * 8 + 256;
}

// dw is in the reference layout with Cin_ref = Cin + dup_shift input channels; db may be null.
extern "C" int bts_conv3d_bwd_weight(int kind, const float* x, const float* dy, float* dw, float* db, void* workspace,
                                     long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout,
                                     int lddy, int dup_start, int dup_shift, int accumulate, hipStream_t stream) {
  int ntaps, s, Dp, Hp, Wp_, Cp, Dq, Hq, Wq, Cq;
  int r = wgrad_dims(kind, D, H, W, Cin, Cout, ntaps, s, Dp, Hp, Wp_, Cp, Dq, Hq, Wq, Cq);
  if (r != BTS_OK) return r;
  if (dup_shift > 0 && (kind == BTS_CONV_K3S2 || kind == BTS_CONV_K3S2T)) return BTS_ERR_UNSUPPORTED;
  WgradPlan pl;
  r = plan_wgrad(pl, ntaps, s, N, Dp, Hp, Wp_, Cp, Dq, Hq, Wq, Cq);
  if (r != BTS_OK) return r;
  const long need = pl.partial_floats * 4 + pl.partial_b_doubles * 8 + 256;
  if (workspace_bytes < need || workspace == nullptr) return BTS_ERR_WORKSPACE;
  WgradParams& p = pl.p;
  const bool transposed = (kind == BTS_CONV_K3S2T);
  p.p = transposed ? dy : x;
  p.q = transposed ? x : dy;
  p.ldp = transposed ? lddy : ldx;
  p.ldq = transposed ? ldx : lddy;
  p.partial = reinterpret_cast<float*>(workspace);
  uintptr_t pb = (uintptr_t)(p.partial + pl.partial_floats);
  pb = (pb + 15) & ~(uintptr_t)15;
  p.partial_b = reinterpret_cast<double*>(pb);
  // bias gradient = column sums of dy: dy is Q except for the transposed conv, where db is computed elsewhere
  p.want_bias = (db != nullptr && !transposed) ? 1 : 0;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  (void)hipGetLastError(); hipLaunchKernelGGL(wgrad_kernel, dim3(pl.nsp, pl.npct, pl.nqct), dim3(256), pl.shmem, stream, p);
  BTS_LAUNCH_CHECK();
  WfinParams f;
  f.partial = p.partial; f.partial_b = p.partial_b; f.dw = dw; f.db = p.want_bias ? db : nullptr;
  f.nsp = pl.nsp; f.npct = pl.npct; f.nqct = pl.nqct; f.ntaps = ntaps; f.Cp = Cp; f.Cq = Cq;
  const int Cin_ref = Cin + dup_shift;
  f.sT = (long)Cin_ref * Cout;
  if (transposed) { f.sP = Cin_ref; f.sQ = 1; f.fold_on_p = 0; }  // (t, Cout, Cin): P = cout, Q = cin
  else { f.sP = Cout; f.sQ = 1; f.fold_on_p = 1; }                // (t, Cin, Cout)
  f.shift = dup_shift;
  f.dup_start = dup_shift > 0 ? dup_start : (1 << 30);
  f.accum = accumulate;
  const long total = (long)ntaps * Cp * Cq;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  (void)hipGetLastError(); hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, stream, f);
  BTS_LAUNCH_CHECK();
  return BTS_OK;
}
