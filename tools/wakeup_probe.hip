// tools/wakeup_probe.hip -- does s_wakeup end another wave's s_sleep on gfx950, and how long does a sleeping wave take to notice an LDS flag with / without it?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/wakeup_probe.hip -o /tmp/wakeup_probe && /tmp/wakeup_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long *out, int use_wakeup, int sleep_arg)
{
  __shared__ int flag; __shared__ unsigned long long t_set;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { flag = 0; t_set = 0; }
  __syncthreads();
  for (int rep = 0; rep < 64; rep++) {
    if (wave == 0) {
      for (int i = 0; i < 37 + rep; i++) __builtin_amdgcn_s_sleep(13);            // let the others fall asleep, at a varying phase
      if (lane == 0) { t_set = __builtin_readcyclecounter(); __hip_atomic_store(&flag, rep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (use_wakeup) asm volatile("s_wakeup" ::: "memory");
    } else {
      while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != rep + 1) {
        if (sleep_arg == 127) __builtin_amdgcn_s_sleep(127); else if (sleep_arg == 32) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(2);
      }
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) out[(wave - 1) * 64 + rep] = t - t_set;
    }
    __syncthreads();
  }
}
int main()
{
  unsigned long long *d; hipMalloc(&d, 7 * 64 * 8);
  static unsigned long long h[7 * 64];
  for (int sl : { 2, 32, 127 }) for (int wk = 0; wk < 2; wk++) {
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d, wk, sl);
    hipDeviceSynchronize(); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; unsigned long long mx = 0;
    for (int i = 0; i < 7 * 64; i++) { sum += (double)h[i]; if (h[i] > mx) mx = h[i]; }
    printf("s_sleep %3d  s_wakeup %d : flag -> seen by a sleeping wave: mean %.0f cycles, max %llu\n", sl, wk, sum / (7 * 64), mx);
  }
  return 0;
}
