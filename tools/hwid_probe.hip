// Which workgroups share a CU?  Launch 1024 workgroups of 256 threads with 75 KB of LDS (two fit on a CU) and record HW_ID / XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned *out)
{
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    lds[0] = (unsigned char)hw;
  }
  for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(127);                   // stay resident ~ 7 ms so that the first 512 overlap
}
int main()
{
  unsigned *d; hipMalloc(&d, 1024 * 8);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 76800);
  hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 76800, 0, d);
  std::vector<unsigned> h(2048); hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost);
  for (int b = 0; b < 1024; b += (b < 520 ? 1 : 37)) {
    const unsigned hw = h[2 * b], x = h[2 * b + 1];
    printf("blk %4d hw %08x wave %u simd %u pipe %u cu %u sh %u se %u xcc %u\n", b, hw, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, x & 15);
  }
  return 0;
}
