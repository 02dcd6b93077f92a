// Cost of agent-scope release / acquire fences on gfx950 while the other workgroups of the XCDs keep their L2s full of dirty lines
// (what a hand-over of work between workgroups on different XCDs pays): hipcc --offload-arch=gfx950 -O3 tools/fence_probe.hip -o /tmp/fence_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *buf, size_t words_per_wg, unsigned long long *out, int iters, int mode)
{
  const int wg = blockIdx.x, lane = threadIdx.x;
  unsigned *mine = buf + (size_t)wg * words_per_wg;
  if (wg % 32 == 0 && threadIdx.x < 64) { // the measuring wave: one per XCD-ish stride
    unsigned long long t_rel = 0, t_acq = 0;
    for (int i = 0; i < iters; i++) {
      for (int j = lane; j < 4096; j += 64) mine[j] = i + j;           // some own dirty data (16 KB)
      unsigned long long t0 = __builtin_readcyclecounter();
      if (mode & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      unsigned long long t1 = __builtin_readcyclecounter();
      if (mode & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      unsigned long long t2 = __builtin_readcyclecounter();
      t_rel += t1 - t0; t_acq += t2 - t1;
      __builtin_amdgcn_s_sleep(64);
    }
    if (lane == 0) { out[2 * (wg / 32)] = t_rel / iters; out[2 * (wg / 32) + 1] = t_acq / iters; }
  } else { // background: keep writing a private 256 KB region (dirty lines in this XCD's L2), like task slots / register saves
    for (int i = 0; i < iters * 4; i++)
      for (size_t j = threadIdx.x; j < words_per_wg; j += blockDim.x) mine[j] = (unsigned)(i + j);
  }
}
int main()
{
  const int G = 256; const size_t words = 65536;   // 256 KB per workgroup: 8 MB per XCD of 32 workgroups > 4 MB of L2
  unsigned *buf; unsigned long long *out;
  hipMalloc(&buf, (size_t)G * words * 4); hipMalloc(&out, 64 * 8); hipMemset(out, 0, 64 * 8);
  for (int mode = 1; mode <= 3; mode++) {
    hipLaunchKernelGGL(probe, dim3(G), dim3(512), 0, 0, buf, words, out, 200, mode);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16); hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost);
    printf("mode %d (1 release, 2 acquire, 3 both): cycles per fence (release, acquire) on 8 measuring waves:", mode);
    for (int i = 0; i < 8; i++) printf("  (%llu, %llu)", h[2 * i], h[2 * i + 1]);
    printf("\n");
  }
  return 0;
}
