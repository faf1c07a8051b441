// Practical ceiling of v_mfma_f32_32x32x2_f32 / 16x16x4 on gfx950 for the occupancies this engine uses.
// build: hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o mfma_f32_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int EXTRA>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  int dummy = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < EXTRA; ++e) dummy = dummy * 3 + it;   // VALU filler between MFMA groups
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = (float)dummy * 1e-30f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
static void run(const char* name, K kern, int blocks, int iters, double flop_per_iter_per_wave, size_t shmem) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shmem, 0, out, iters, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shmem, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * flop_per_iter_per_wave;
  printf("%-44s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  const int it = 20000;
  const double f32 = 8 * 4096.0, f16 = 16 * 2048.0;
  // 1 WG/CU (1 wave/SIMD): force with 96 KB dynamic LDS; 2 WG/CU: 64 KB; 4 WG/CU: none
  hipFuncSetAttribute((const void*)k32<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k32<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k32<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k32<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k32<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k16<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  run("32x32x2 NACC=2 1wave/SIMD", k32<2, 0>, 256, it, f32, 96 * 1024);
  run("32x32x2 NACC=4 1wave/SIMD", k32<4, 0>, 256, it, f32, 96 * 1024);
  run("32x32x2 NACC=8 1wave/SIMD", k32<8, 0>, 256, it, f32, 96 * 1024);
  run("32x32x2 NACC=2 2waves/SIMD", k32<2, 0>, 512, it, f32, 64 * 1024);
  run("32x32x2 NACC=4 2waves/SIMD", k32<4, 0>, 512, it, f32, 64 * 1024);
  run("32x32x2 NACC=2 2waves/SIMD +8 VALU/8 MFMA", k32<2, 8>, 512, it, f32, 64 * 1024);
  run("32x32x2 NACC=4 2waves/SIMD +8 VALU/8 MFMA", k32<4, 8>, 512, it, f32, 64 * 1024);
  run("32x32x2 NACC=2 4waves/SIMD", k32<2, 0>, 1024, it, f32, 0);
  run("16x16x4 NACC=4 1wave/SIMD", k16<4>, 256, it, f16, 96 * 1024);
  run("16x16x4 NACC=8 2waves/SIMD", k16<8>, 512, it, f16, 64 * 1024);
  return 0;
}
