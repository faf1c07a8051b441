// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds element index i at 16-bit slot i; every lane supplies the byte address addr[lane];
// out[lane][0..3] = the four 16-bit values the lane received.  Build: hipcc --offload-arch=gfx950 -shared -fPIC tr_probe.hip -o tr_probe.so
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr_probe_kernel(const unsigned* addr, unsigned short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 16384; i += 64) reinterpret_cast<unsigned short*>(lds)[i] = (unsigned short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
extern "C" int tr_probe(const unsigned* addr, unsigned short* out, hipStream_t s) {
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 32768, s, addr, out);
  return (int)hipGetLastError();
}
