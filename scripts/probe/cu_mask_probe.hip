// Which compute units does a stream created with hipExtStreamCreateWithCUMask use on this part?  (build: hipcc --offload-arch=gfx950 -O2
// cu_mask_probe.hip -o cu_mask_probe; run on the GPU box)  Every workgroup records (XCC_ID, HW_ID) of its first wave; the host counts the
// distinct (xcc, se, sh, cu) tuples per mask and times a fixed amount of spinning work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc | (a == 123.f ? 1u << 31 : 0u); }
}

static void run(const char* label, hipStream_t s, unsigned* d, int nblk) {
  std::vector<unsigned> h(2 * nblk);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, s, d, 20000);
  hipEventRecord(e0, s);
  hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, s, d, 20000);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::set<unsigned> cus, xccs;
  for (int b = 0; b < nblk; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
    xccs.insert(xcc);
  }
  printf("%-28s distinct CUs %3zu on %zu XCCs, %.3f ms for %d workgroups\n", label, cus.size(), xccs.size(), ms, nblk);
}

int main() {
  const int nblk = 8192;
  unsigned* d; hipMalloc(&d, 2 * nblk * 4);
  hipStream_t s0; hipStreamCreate(&s0);
  run("no mask", s0, d, nblk);
  const int sizes[] = {32, 64, 128, 192, 256};
  for (int k : sizes) {
    // (a) the first k bits
    unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < k; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("first %d bits: %s\n", k, hipGetErrorString(e)); continue; }
    char lab[64]; snprintf(lab, sizeof lab, "first %d bits", k);
    run(lab, s, d, nblk);
    hipStreamDestroy(s);
  }
  {  // (b) every other bit of 256
    unsigned mask[8];
    for (int i = 0; i < 8; ++i) mask[i] = 0x55555555u;
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) == hipSuccess) { run("even bits of 256", s, d, nblk); hipStreamDestroy(s); }
  }
  return 0;
}
