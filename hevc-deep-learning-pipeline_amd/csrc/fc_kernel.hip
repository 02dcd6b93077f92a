// fc_kernel.hip -- the fully connected head of the depth CNN (use_model.py:44-58: fc1 2048 -> 256, fc2 -> 64, fc3 -> 16, each with ReLU
// but the last) and the label post-processing (use_model.py:101-119 + the boundary clamp), batched over CTUs for gfx950 (MI355X).
//
// cnn_kernel.hip leaves the flattened conv3 output of every (CTU, quadrant) in HBM: A[row = 4 * ctu + quadrant][2048].  Here a workgroup
// takes 64 rows (16 CTUs) and runs the three layers as GEMMs on v_mfma_f32_16x16x4_f32:
//   fc1  D[64][256] = A[64][2048] x W1[2048][256]   wave w owns output columns [64w, 64w + 64): 4 M-tiles x 4 N-tiles = 16 accumulators;
//        the 2 MB of fc1 weights are read once per 16 CTUs (once per CTU when the head lived in the per-CTU kernel: L2 bound there)
//   fc2  D[64][64]  = H1[64][256] x W2[256][64]     wave w owns N-tile w, 4 M-tiles; H1 from LDS
//   fc3  D[64][16]  = H2[64][64]  x W3[64][16]      wave w owns M-tile w; H2 from LDS
// Operand layout of the instruction: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; result register r of lane l
// is D[(l >> 4) * 4 + r][l & 15].  Weights are packed [k][n] by the host (pack_fc), bias behind them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hevcdl_dev.h"

namespace {
#define GLB __attribute__((address_space(1)))
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma4(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
constexpr int H1_ROW = 260, H2_ROW = 68;     // LDS pitches: + 4 floats keep the 16 rows of an A fragment on different banks
struct FcSmem { float h1[64 * H1_ROW]; float h2[64 * H2_ROW]; float lg[64][16]; };
}

extern "C" __global__ __launch_bounds__(256)
void hevcdl_fc_kernel(hevcdl_fc_params p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem_raw[];
  FcSmem &sm = *(FcSmem *)fc_smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;
  const int n_rows = p.n_ctus * 4, row0 = blockIdx.x * 64;
  const float GLB *W = (const float GLB *)p.weights;

  { // ---- fc1 ----------------------------------------------------------------------------------------------------
    const float GLB *W1 = W + HEVCDL_W_FC1 + (size_t)g4 * 256 + wave * 64 + i16;     // B[k0 + g4][64 * wave + 16 * nt + i16]
    const float GLB *a_ptr[4]; bool a_ok[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      const int r = row0 + mt * 16 + i16;
      a_ok[mt] = r < n_rows;
      a_ptr[mt] = (const float GLB *)p.a3 + (size_t)(a_ok[mt] ? r : 0) * 2048 + g4;
    }
    v4f acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (v4f){ 0.f, 0.f, 0.f, 0.f };
    // two register sets of 4 k-steps (A: 4 M-tiles, B: 4 N-tiles): the loads of the next set are in flight under 64 MFMAs
    float a0[4][4], b0[4][4], a1[4][4], b1[4][4];
    auto load_set = [&](float (&a)[4][4], float (&b)[4][4], int k0) {
#pragma unroll
      for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) a[s][mt] = a_ptr[mt][k0 + 4 * s];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) b[s][nt] = W1[(size_t)(k0 + 4 * s) * 256 + nt * 16];
      }
    };
    auto mac_set = [&](float (&a)[4][4], float (&b)[4][4]) {
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
          for (int nt = 0; nt < 4; nt++) acc[mt][nt] = mfma4(a[s][mt], b[s][nt], acc[mt][nt]);
    };
    load_set(a0, b0, 0);
#pragma unroll 1
    for (int k0 = 0; k0 < 2048; k0 += 32) {
      load_set(a1, b1, k0 + 16);
      __builtin_amdgcn_sched_barrier(0);
      mac_set(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_set(a0, b0, k0 + 32 < 2048 ? k0 + 32 : 0);
      __builtin_amdgcn_sched_barrier(0);
      mac_set(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float GLB *bias = W + HEVCDL_W_FC1 + 2048 * 256 + wave * 64 + i16;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      const float b = bias[nt * 16];
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        const v4f r = acc[mt][nt];
        float *d = sm.h1 + (mt * 16 + g4 * 4) * H1_ROW + wave * 64 + nt * 16 + i16;
        d[0] = fmaxf(r.x + b, 0.f); d[H1_ROW] = fmaxf(r.y + b, 0.f); d[2 * H1_ROW] = fmaxf(r.z + b, 0.f); d[3 * H1_ROW] = fmaxf(r.w + b, 0.f);
      }
    }
  }
  __syncthreads();
  { // ---- fc2: wave w -> output columns [16w, 16w + 16) ---------------------------------------------------------------
    const float GLB *W2 = W + HEVCDL_W_FC2 + (size_t)g4 * 64 + wave * 16 + i16;
    v4f acc[4] = { {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0} };
#pragma unroll 4
    for (int k0 = 0; k0 < 256; k0 += 4) {
      const float b = W2[(size_t)k0 * 64];
#pragma unroll
      for (int mt = 0; mt < 4; mt++) acc[mt] = mfma4(sm.h1[(mt * 16 + i16) * H1_ROW + k0 + g4], b, acc[mt]);
    }
    const float b = W[HEVCDL_W_FC2 + 256 * 64 + wave * 16 + i16];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      float *d = sm.h2 + (mt * 16 + g4 * 4) * H2_ROW + wave * 16 + i16;
      d[0] = fmaxf(acc[mt].x + b, 0.f); d[H2_ROW] = fmaxf(acc[mt].y + b, 0.f); d[2 * H2_ROW] = fmaxf(acc[mt].z + b, 0.f); d[3 * H2_ROW] = fmaxf(acc[mt].w + b, 0.f);
    }
  }
  __syncthreads();
  { // ---- fc3: wave w -> rows [16w, 16w + 16), 16 logits -----------------------------------------------------------------
    const float GLB *W3 = W + HEVCDL_W_FC3 + (size_t)g4 * 16 + i16;
    v4f acc = { 0, 0, 0, 0 };
#pragma unroll
    for (int k0 = 0; k0 < 64; k0 += 4) acc = mfma4(sm.h2[(wave * 16 + i16) * H2_ROW + k0 + g4], W3[(size_t)k0 * 16], acc);
    const float b = W[HEVCDL_W_FC3 + 64 * 16 + i16];
    const float v[4] = { acc.x + b, acc.y + b, acc.z + b, acc.w + b };
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = wave * 16 + g4 * 4 + r;
      sm.lg[row][i16] = v[r];
      if (p.logits && row0 + row < n_rows) ((float GLB *)p.logits)[(size_t)(row0 + row) * 16 + i16] = v[r];
    }
  }
  __syncthreads();

  // ---- labels of the 16 CTUs: 4x argmax + fix-ups (use_model.py:101-119), then the boundary clamp; one thread per CTU ----
  if (tid < 16 && blockIdx.x * 16 + tid < p.n_ctus) {
    const int gctu = p.ctu_base + blockIdx.x * 16 + tid;
    const int addr = gctu % p.ctus_per_frame, x0 = (addr % p.ctus_x) * 64, y0 = (addr / p.ctus_x) * 64;
    const float (*lg)[16] = &sm.lg[tid * 4];
    uint8_t lab[16];
    const int quads[4][4] = { {0, 1, 4, 5}, {2, 3, 6, 7}, {8, 9, 12, 13}, {10, 11, 14, 15} };
    for (int q = 0; q < 4; q++) {
      int d[4]; bool any0 = false, all0 = true, any1 = false, all1 = true;
      for (int k = 0; k < 4; k++) {
        int best = 0; float bv = lg[q][4 * k];
        for (int j = 1; j < 4; j++) if (lg[q][4 * k + j] > bv) { bv = lg[q][4 * k + j]; best = j; }   // first maximum wins
        d[k] = best;
      }
      for (int k = 0; k < 4; k++) { any0 |= d[k] == 0; all0 &= d[k] == 0; }
      if (any0 && !all0) for (int k = 0; k < 4; k++) if (d[k] == 0) d[k] = 1;
      for (int k = 0; k < 4; k++) { any1 |= d[k] == 1; all1 &= d[k] == 1; }
      if (any1 && !all1) for (int k = 0; k < 4; k++) if (d[k] == 1) d[k] = 2;
      bool zero = (d[0] | d[1] | d[2] | d[3]) == 0;
      if (q == 1 && zero && lab[0] != 0) d[0] = d[1] = d[2] = d[3] = 1;
      if (q == 2 && zero && lab[2] != 0) d[0] = d[1] = d[2] = d[3] = 1;
      if (q == 3 && zero && lab[8] != 0) d[0] = d[1] = d[2] = d[3] = 1;
      for (int k = 0; k < 4; k++) lab[quads[q][k]] = (uint8_t)d[k];
    }
    if (p.clamp) hevcdl_clamp_ctu_labels(lab, x0, y0, p.width, p.height);
    for (int c = 0; c < 16; c++) ((uint8_t GLB *)p.labels)[(size_t)(blockIdx.x * 16 + tid) * 16 + c] = lab[c];
  }
}

extern "C" size_t hevcdl_fc_smem_bytes(void) { return sizeof(FcSmem); }
