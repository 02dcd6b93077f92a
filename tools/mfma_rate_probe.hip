// Issue rate of the two f16 MFMA shapes on gfx950, one wave per SIMD, four independent accumulators: cycles per instruction of v_mfma_f32_16x16x32_f16 (8 halves a lane per operand)
// against v_mfma_f32_16x16x16_f16 (4 halves).  Question behind it: would the hi x lo term of the 5x5 layers be cheaper as a K = 16 product on packed hi halves?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int K> __global__ void k(float *out, long long *cyc, int iters)
{
  h8 a8, b8; h4 a4, b4;
  for (int j = 0; j < 8; j++) { a8[j] = (_Float16)(threadIdx.x * 0.001f + j); b8[j] = (_Float16)(j * 0.5f); }
  for (int j = 0; j < 4; j++) { a4[j] = a8[j]; b4[j] = b8[j]; }
  f4 c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (K == 32) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c3, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main()
{
  float *out; long long *cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int rep = 0; rep < 2; rep++) for (int K = 32; K >= 16; K -= 16) {
    for (int waves = 1; waves <= 1; waves++) {          // waves per SIMD
      long long h = 0;
      if (K == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, iters); else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, iters);
      hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("K=%d  %d wave(s) per SIMD: %.2f cycles per MFMA per wave (%.2f per SIMD slot)\n", K, waves, (double)h / (4.0 * iters), (double)h / (4.0 * iters) / waves);
    }
  }
  return 0;
}
