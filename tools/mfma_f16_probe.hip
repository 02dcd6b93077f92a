// Operand layout of v_mfma_f32_16x16x32_f16 on gfx950, checked on the hardware: with the ASSUMED mapping
//   A[i][k]: lane l, element j (0..7) -> i = l & 15, k = 8 * (l >> 4) + j      B[k][n]: lane l, element j -> n = l & 15, k = 8 * (l >> 4) + j
//   D[i][n]: lane l, register r -> n = l & 15, i = 4 * (l >> 4) + r
// C = A x B of small integers (exact in f16 / f32) must equal the host product.   hipcc --offload-arch=gfx950 tools/mfma_f16_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D)
{
  const int l = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = (_Float16)B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f4 c = { 0, 0, 0, 0 };
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main()
{
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  for (int i = 0; i < 16; i++) for (int kk = 0; kk < 32; kk++) hA[i * 32 + kk] = (float)((i * 7 + kk * 3) % 11 - 5);
  for (int kk = 0; kk < 32; kk++) for (int n = 0; n < 16; n++) hB[kk * 16 + n] = (float)((kk * 5 + n * 13) % 7 - 3);
  for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) { float s = 0; for (int kk = 0; kk < 32; kk++) s += hA[i * 32 + kk] * hB[kk * 16 + n]; ref[i * 16 + n] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
  printf("mfma_f32_16x16x32_f16 layout probe: %d of 256 elements differ\n", bad);
  return bad != 0;
}
