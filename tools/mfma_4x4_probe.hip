// probe of v_mfma_f32_4x4x1_16b_f32 operand / result layout (run once on the GPU box; prints PASS when the
// hypothesis  D[lane = 4*blk + j][vgpr i] = A[lane 4*blk + i] * B[lane 4*blk + j]  holds)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
  const int l = threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), (float)(100 + 3 * l), acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = acc[r];
}
int main() {
  float *d, h[256]; hipMalloc(&d, sizeof h); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) { const float e = (float)(1 + 4 * (l / 4) + r) * (float)(100 + 3 * l); if (h[l * 4 + r] != e) bad++; }
  printf("%s (%d mismatches) lane5: %g %g %g %g\n", bad ? "FAIL" : "PASS", bad, h[20], h[21], h[22], h[23]);
  return 0;
}
