// Instruction latencies that bound the serial parts of the decision kernel on gfx950 (one wave alone on a CU, and eight waves per CU):
//   dependent v_add_f64 chain, dependent v_add_u32 chain, LDS read -> use round trip (pointer chase), v_readlane -> VALU use, ds_bpermute round trip
// hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 256
__global__ __launch_bounds__(512) void probe(unsigned long long *out, double seed, int active)
{
  __shared__ int chase[512 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) chase[i] = (i + 17) & 1023;
  __syncthreads();
  if (wave >= active) return;
  unsigned long long r[6];
  { // dependent f64 adds
    double a = seed + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(seed));
    r[0] = __builtin_readcyclecounter() - t0;
    if (a == 12345.0) out[1000] = 1;
  }
  { // dependent u32 adds
    unsigned a = (unsigned)lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(lane));
    r[1] = __builtin_readcyclecounter() - t0;
    if (a == 12345u) out[1000] = 1;
  }
  { // LDS pointer chase
    int p = lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < N; i++) p = chase[p];
    r[2] = __builtin_readcyclecounter() - t0;
    if (p == 12345) out[1000] = 1;
  }
  { // readlane -> valu -> readlane chain
    int v = lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; i++) { const int s = __builtin_amdgcn_readlane(v, 3); v = v + s; }
    r[3] = __builtin_readcyclecounter() - t0;
    if (v == 12345) out[1000] = 1;
  }
  { // ds_bpermute chain
    int v = lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < N; i++) v = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, v) + 1;
    r[4] = __builtin_readcyclecounter() - t0;
    if (v == 12345) out[1000] = 1;
  }
  { // dependent f64 fma (mul) chain
    double a = seed + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(seed));
    r[5] = __builtin_readcyclecounter() - t0;
    if (a == 12345.0) out[1000] = 1;
  }
  if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 6; i++) out[wave * 6 + i] = r[i];
}
int main()
{
  unsigned long long *out; hipMalloc(&out, 1024 * 8 + 8);
  const char *names[6] = { "v_add_f64 dependent", "v_add_u32 dependent", "LDS read dependent", "readlane+add dependent", "ds_bpermute dependent", "v_mul_f64 dependent" };
  for (int active = 1; active <= 8; active *= 2) {
    hipMemset(out, 0, 1024 * 8);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, 1.000001, active);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(48); hipMemcpy(h.data(), out, 48 * 8, hipMemcpyDeviceToHost);
    printf("%d wave(s) per CU: cycles per op (wave 0):", active);
    for (int i = 0; i < 6; i++) printf("  %s %.1f", names[i], (double)h[i] / N);
    printf("\n");
  }
  return 0;
}
