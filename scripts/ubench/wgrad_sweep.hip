// The wgrad MFMA sweep in isolation (LDS pre-filled once, no staging): which part of the step costs matrix-pipe time?
// VAR 0: the production step (address arithmetic per step, 5 ds_read_b32, 4/3 MFMAs, setprio)
// VAR 1: same, address arithmetic replaced by a running offset
// VAR 2: VAR 1 + all waves T=4 (32 tiles)    VAR 3: VAR 0 without setprio
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct P { int lgTX, lgTY, IY, IX, s, lgSP; int tap_vox[32]; };
template <int T> struct Frag { float q, a[T]; };
__device__ __forceinline__ float ldsr(unsigned a) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ void wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

template <int T, int VAR>
__device__ __forceinline__ void sweep(const P& p, unsigned bpb, unsigned bqb, const int (&rowoff)[4], f32x16 (&acc)[4], int nsteps, int h, int l32) {
  const int TXm = (1 << p.lgTX) - 1, TYm = (1 << p.lgTY) - 1, lgXY = p.lgTX + p.lgTY;
  unsigned laneoff[T];
#pragma unroll
  for (int i = 0; i < T; ++i) laneoff[i] = bpb + 4u * (unsigned)(rowoff[i] + ((h * p.s) << p.lgSP));
  const unsigned qlane = bqb + 4u * (unsigned)(h * 32 + l32);
  unsigned run = 0, junk = 0, sjunk = 0;
  auto load = [&](Frag<T>& f, int k) {
    const int m = 2 * k;
    int pvu;
    if (VAR & 1) { pvu = run; run = (run + 64) & 8191; }
    else pvu = (((m >> lgXY) * p.s * p.IY + ((m >> p.lgTX) & TYm) * p.s) * p.IX + (m & TXm) * p.s) << p.lgSP;
    f.q = ldsr(qlane + (unsigned)(m * 128));
    if (VAR & 8) {
      const unsigned b = laneoff[0] + (unsigned)(pvu * 4);
      asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(f.a[0]) : "v"(b));
      if (T > 1) asm volatile("ds_read_b32 %0, %1 offset:13824" : "=v"(f.a[1 % T]) : "v"(b));
      if (T > 2) asm volatile("ds_read_b32 %0, %1 offset:27648" : "=v"(f.a[2 % T]) : "v"(b));
      if (T > 3) asm volatile("ds_read_b32 %0, %1 offset:41472" : "=v"(f.a[3 % T]) : "v"(b));
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) f.a[i] = ldsr(laneoff[i] + (unsigned)(pvu * 4));
    }
    if (VAR & 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, 1" : "+v"(junk));
    }
    if (VAR & 32) {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sjunk));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step = [&](const Frag<T>& cur, Frag<T>& nxt, int kn) {
    if (VAR & 64) load(nxt, kn);
    if (VAR & 64) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(T + 1) : "memory"); else wait0();
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 4)) __builtin_amdgcn_s_setprio(1);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0], cur.q, acc[0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 64)) load(nxt, kn);
#pragma unroll
    for (int i = 1; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[i], cur.q, acc[i], 0, 0, 0);
    if (!(VAR & 4)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  Frag<T> A, B;
  load(A, 0);
  const int last = nsteps - 1;
  for (int k = 0; k < nsteps; k += 2) {
    step(A, B, k + 1 < last ? k + 1 : last);
    if (k + 1 >= nsteps) break;
    step(B, A, k + 2 < last ? k + 2 : last);
  }
  wait0();
  if (junk == 12345u || sjunk == 12345u) acc[0][0] += 1.f;
}

template <int VAR, int STG = 0>
__global__ __launch_bounds__(512, 2) void k(const P p, float* out, int nsub, const float* src = nullptr) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 35 * 1024; i += blockDim.x) lds[i] = (float)((i * 7) & 255) * 0.01f - 1.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l32 = lane & 31;
  int rowoff[4];
  for (int i = 0; i < 4; ++i) { const int tile = wave + 8 * i; rowoff[i] = (p.tap_vox[tile & 31] << p.lgSP) + l32; }
  const int ntw = (VAR & 2) ? 4 : (wave < 3 ? 4 : 3);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned bpb = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds;
  const unsigned bqb = bpb + 432 * 32 * 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 stg[9];
  for (int i = 0; i < 9; ++i) stg[i] = f4{0.f, 0.f, 0.f, 0.f};
  float* stage_dst = lds + 36 * 1024;  // a second 70 KB region, never read by the sweep
  for (int sub = 0; sub < nsub; ++sub) {
    // staging of the "next sub-tile": 9 wave-loads of 1 KB per wave (70 KB per workgroup), issued up front
    const float* g = src + ((size_t)(blockIdx.x * 37 + sub) % 4096) * 18432 + (threadIdx.x >> 6) * 64 * 4 + (threadIdx.x & 63) * 4;
    if (STG == 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 2048),
                                         (__attribute__((address_space(3))) void*)(stage_dst + (threadIdx.x >> 6) * 256 + i * 2048), 16, 0, 0);
    } else if (STG == 2 || STG == 3) {
#pragma unroll
      for (int i = 0; i < 9; ++i) stg[i] = *reinterpret_cast<const f4*>(g + i * 2048);
    }
    if (ntw == 4) sweep<4, VAR>(p, bpb, bqb, rowoff, acc, 64, h, l32);
    else sweep<3, VAR>(p, bpb, bqb, rowoff, acc, 64, h, l32);
    if (STG == 2 || STG == 4) {
#pragma unroll
      for (int i = 0; i < 9; ++i) *reinterpret_cast<f4*>(stage_dst + threadIdx.x * 4 + i * 2048) = stg[i];
    }
    if (STG == 3) {
#pragma unroll
      for (int i = 0; i < 9; ++i) asm volatile("" ::"v"(stg[i]));
    }
    if (STG == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float* g_src = nullptr;
template <typename K>
static void run(const char* name, K kern, int tiles) {
  if (!g_src) { (void)hipMalloc(&g_src, (size_t)4096 * 18432 * 4 + (1 << 20)); (void)hipMemset(g_src, 0, (size_t)4096 * 18432 * 4 + (1 << 20)); }
  P p; p.lgTX = 4; p.lgTY = 2; p.IY = 6; p.IX = 18; p.s = 1; p.lgSP = 5;
  for (int t = 0; t < 32; ++t) { int tt = t % 27; p.tap_vox[t] = ((tt / 9) * 6 + (tt / 3) % 3) * 18 + tt % 3; }
  float* out; (void)hipMalloc(&out, 512 * 512 * 4);
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nsub = 32;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(512), dim3(512), 150 * 1024, 0, p, out, nsub, g_src);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(512), dim3(512), 150 * 1024, 0, p, out, nsub, g_src);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = 512.0 * nsub * 64 * tiles * 4096.0;
  printf("%-56s %8.3f ms %7.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
  (void)hipFree(out);
}
int main() {
  for (int rep = 0; rep < 3; ++rep) {  // clocks ramp up over the first milliseconds: read the last repetition
  run("imm-offset + setprio, no staging", k<1|2|8, 0>, 32);
  run("  + LDS-DMA dwordx4 staging (9 KB/wave/sub-tile)", k<1|2|8, 1>, 32);
  run("  + global->VGPR then ds_write_b128", k<1|2|8, 2>, 32);
  run("  + global->VGPR only", k<1|2|8, 3>, 32);
  run("  + ds_write_b128 only", k<1|2|8, 4>, 32);
  }
  return 0;
}
