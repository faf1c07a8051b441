// LDS-DMA throughput per CU (microbenchmark): every wave of a 512-thread workgroup issues `reps` x 6 pieces (buffer_load_dwordx4 ... lds,
// 1 KB per wave-instruction) from its own stream of a big buffer, waits for them with vmcnt(0) every 6.  Variants:
//   mode 0: lanes read 64 x 16 contiguous bytes               mode 1: 16-byte pieces of every second 8-voxel block swapped (lowp_wgd.hip)
//   mode 2: all lanes out of range (zero fill, no traffic)      mode 3: plain global_load_dwordx4 to registers + ds_write_b128
//   mode 4: dword (4-byte) DMA form, 256 B per instruction
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o dma_rate.so dma_rate.hip ; driven by dma_rate.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void dma_kernel(const unsigned char* src, long bytes_per_wg, int reps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned char* base = src + (long)blockIdx.x * bytes_per_wg;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  unsigned lo = lane * 16;
  if (MODE == 1) { const int v = lane >> 2, pos = lane & 3; lo = v * 64 + ((pos ^ (2 * ((v >> 3) & 1))) * 16); }
  if (MODE == 2) lo = 0x80000000u;
  if (MODE == 4) lo = lane * 4;
  unsigned acc = 0;
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const unsigned so = (unsigned)(((i * 6 + j) * 8 + wave) * (MODE == 4 ? 256 : 1024));
      if (MODE == 3) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lo, so, 0);
        *reinterpret_cast<u32x4*>(lds + (j * 8 + wave) * 1024 + lane * 16) = v;
      } else if (MODE == 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + (j * 8 + wave) * 1024), 4, lo, so, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + (j * 8 + wave) * 1024), 16, lo, so, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  acc = *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}
extern "C" int dma_run(int mode, const void* src, long bytes_per_wg, int reps, int nwg, void* sink, hipStream_t s) {
  const unsigned char* p = (const unsigned char*)src;
  switch (mode) {
    case 0: hipLaunchKernelGGL(dma_kernel<0>, dim3(nwg), dim3(512), 49152, s, p, bytes_per_wg, reps, (unsigned*)sink); break;
    case 1: hipLaunchKernelGGL(dma_kernel<1>, dim3(nwg), dim3(512), 49152, s, p, bytes_per_wg, reps, (unsigned*)sink); break;
    case 2: hipLaunchKernelGGL(dma_kernel<2>, dim3(nwg), dim3(512), 49152, s, p, bytes_per_wg, reps, (unsigned*)sink); break;
    case 3: hipLaunchKernelGGL(dma_kernel<3>, dim3(nwg), dim3(512), 49152, s, p, bytes_per_wg, reps, (unsigned*)sink); break;
    default: hipLaunchKernelGGL(dma_kernel<4>, dim3(nwg), dim3(512), 49152, s, p, bytes_per_wg, reps, (unsigned*)sink); break;
  }
  return (int)hipGetLastError();
}
