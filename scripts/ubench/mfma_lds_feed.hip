// What limits v_mfma_f32_32x32x2_f32 when its A/B operands arrive from LDS (the wgrad / igemm inner-loop pattern)?
// Variants isolate: LDS traffic alone, operand dependency on LDS results, VALU-written operands, wide LDS reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: const operands, no LDS.  1: LDS reads issued, results unused.  2: LDS b32 feed, ping-pong registers (no moves).
// 3: operands rewritten by VALU each step.  4: ds_read_b128 feed, ping-pong.
template <int T, int MODE, int PRIO = 0>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  f32x16 acc[T];
  for (int i = 0; i < T; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float q[2], a[2][T];
  int base = lane;
  for (int s = 0; s < 2; ++s) { q[s] = lds[base + s]; for (int i = 0; i < T; ++i) a[s][i] = lds[base + s + 64 * (i + 1)]; }
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (MODE == 5) {  // true software pipeline: issue next step's reads into the other buffer, then this step's MFMAs
        base = (base + 1024) & 8191;
        q[s ^ 1] = lds[base];
#pragma unroll
        for (int i = 0; i < T; ++i) a[s ^ 1][i] = lds[base + 64 * (i + 1)];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], q[s], acc[i], 0, 0, 0);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 5) base = (base + 1024) & 8191;
      if (MODE == 1) {
        float t0 = lds[base], t1 = lds[base + 64], t2 = lds[base + 128], t3 = lds[base + 192], t4 = lds[base + 256];
        asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4));
      } else if (MODE == 2) {
        q[s] = lds[base];
#pragma unroll
        for (int i = 0; i < T; ++i) a[s][i] = lds[base + 64 * (i + 1)];
      } else if (MODE == 3) {
        q[s] = q[s] + 1.0f;
#pragma unroll
        for (int i = 0; i < T; ++i) a[s][i] = a[s][i] + 0.5f;
      } else if (MODE == 4) {
        f4 v = *reinterpret_cast<f4*>(lds + ((base * 4) & 8191));
        f4 w = *reinterpret_cast<f4*>(lds + ((base * 4 + 256) & 8191));
        q[s] = w[0];
#pragma unroll
        for (int i = 0; i < T; ++i) a[s][i] = v[i & 3];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sum = 0.f;
  for (int i = 0; i < T; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <typename K>
static void run(const char* name, K kern, int threads, int T, int iters) {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 140 * 1024, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 140 * 1024, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = 256.0 * (threads / 64) * iters * T * 4096.0;
  printf("%-56s %8.3f ms %7.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
  (void)hipFree(out);
}
int main() {
  run("8w T=4 const", k<4, 0>, 512, 4, 20000);
  run("8w T=4 const, setprio", k<4, 0, 1>, 512, 4, 20000);
  run("8w T=8 const", k<8, 0>, 512, 8, 10000);
  run("8w T=2 const", k<2, 0>, 512, 2, 40000);
  run("4w T=4 const", k<4, 0>, 256, 4, 20000);
  run("8w T=4 LDS feed, reads after MFMAs", k<4, 2>, 512, 4, 20000);
  run("8w T=4 LDS feed, reads first", k<4, 5>, 512, 4, 20000);
  run("8w T=4 LDS feed, reads first, setprio", k<4, 5, 1>, 512, 4, 20000);
  run("4w T=4 LDS feed, reads first", k<4, 5>, 256, 4, 20000);
  run("4w T=8 LDS feed, reads first", k<8, 5>, 256, 8, 10000);
  run("8w T=8 LDS feed, reads first", k<8, 5>, 512, 8, 10000);
  run("16w T=4 LDS feed, reads first", k<4, 5>, 1024, 4, 20000);
  return 0;
}
