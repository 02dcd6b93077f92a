// fc_kernel.hip -- the fully connected head of the depth CNN (use_model.py:44-58: fc1 2048 -> 256, fc2 -> 64, fc3 -> 16, each with ReLU
// but the last) and the label post-processing (use_model.py:101-119 + the boundary clamp), batched over CTUs for gfx950 (MI355X).
//
// cnn_kernel.hip leaves the flattened conv3 output of every (CTU, quadrant) in HBM: A[row = 4 * ctu + quadrant][2048].  Here a workgroup
// takes 64 rows (16 CTUs) and runs the three layers as GEMMs on the matrix cores:
//   fc1  D[64][256] = A[64][2048] x W1[2048][256]   wave w owns output columns [64w, 64w + 64): 4 M-tiles x 4 N-tiles = 16 accumulators;
//        the 2 MB of fc1 weights are read once per 16 CTUs (once per CTU when the head lived in the per-CTU kernel: L2 bound there).
//        97 % of the head's arithmetic: on v_mfma_f32_16x16x32_f16 with SPLIT operands like the convolutions (cnn_kernel.hip: hi = f16(v), lo = f16(v - hi), a product
//        is hi*hi + hi*lo + lo*hi in f32) -- the conv kernel leaves its outputs as such pairs (one 32-bit word each, so the row layout of A is unchanged) and the host
//        packs W1 as pairs in B-operand lane order in exactly the bytes the f32 matrix took (hevcdl_api.hip pack_fc1).  fc2 / fc3 stay on v_mfma_f32_16x16x4_f32.
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
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
// 8 split values (k = 0..7 of a lane, one word each: low half hi, high half lo) -> the hi and the lo operand of the f16 MFMA
__device__ __forceinline__ void gather_hl(const u4 &x, const u4 &y, h8 &hi, h8 &lo)
{
  u4 a, b;
  a[0] = __builtin_amdgcn_perm(x[1], x[0], 0x05040100u); b[0] = __builtin_amdgcn_perm(x[1], x[0], 0x07060302u);
  a[1] = __builtin_amdgcn_perm(x[3], x[2], 0x05040100u); b[1] = __builtin_amdgcn_perm(x[3], x[2], 0x07060302u);
  a[2] = __builtin_amdgcn_perm(y[1], y[0], 0x05040100u); b[2] = __builtin_amdgcn_perm(y[1], y[0], 0x07060302u);
  a[3] = __builtin_amdgcn_perm(y[3], y[2], 0x05040100u); b[3] = __builtin_amdgcn_perm(y[3], y[2], 0x07060302u);
  hi = __builtin_bit_cast(h8, a); lo = __builtin_bit_cast(h8, b);
}
__device__ __forceinline__ v4f mfma3(const h8 &ah, const h8 &al, const h8 &bh, const h8 &bl, v4f c)
{ // acc += (ah + al) * (bh + bl) without al * bl
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
}
__device__ __forceinline__ v4f mfma4(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
constexpr int H1_ROW = 260, H2_ROW = 68;     // LDS pitches: + 4 floats keep the 16 rows of an A fragment on different banks
// h2 and the logits share h1's storage (fc2's results wait in registers behind a barrier until every wave has read h1): 65 KB, TWO workgroups per CU -- with one, a
// SIMD held a single wave and nothing covered its load latencies
struct FcSmem { union { float h1[64 * H1_ROW]; struct { float h2[64 * H2_ROW]; float lg[64][16]; } s; }; };
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

  if (p.logits_in) {       // label stage alone (hevcdl_labels_from_logits): the caller's logits take the place of fc3's
    if (row0 + (tid >> 2) < n_rows)
#pragma unroll
      for (int j = 0; j < 4; j++) sm.s.lg[tid >> 2][4 * (tid & 3) + j] = ((const float GLB *)p.logits_in)[(size_t)(row0 + (tid >> 2)) * 16 + 4 * (tid & 3) + j];
  } else {
  { // ---- fc1 ----------------------------------------------------------------------------------------------------
    // B: [k-step of 32][N-tile of 16][hi | lo][64 lanes] x 16 bytes (8 halves: k = 32 * step + 8 * (lane >> 4) + j, n = 16 * tile + (lane & 15))
    const u4 GLB *W1 = (const u4 GLB *)(W + HEVCDL_W_FC1) + lane + (size_t)(4 * wave) * 128;
    const u4 GLB *a_ptr[4]; bool a_ok[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      const int r = row0 + mt * 16 + i16;
      a_ok[mt] = r < n_rows;
      a_ptr[mt] = (const u4 GLB *)((const float GLB *)p.a3 + (size_t)(a_ok[mt] ? r : 0) * 2048 + 8 * g4);     // A[i16][k = 32 * step + 8 * g4 + j]: two 16-byte loads per k-step
    }
    v4f acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (v4f){ 0.f, 0.f, 0.f, 0.f };
    // two register sets of one k-step (A: 4 M-tiles x 2 loads, B: 4 N-tiles x hi / lo): the loads of the next step are in flight under 48 MFMAs
    u4 a0[4][2], b0[4][2], a1[4][2], b1[4][2];
    auto load_set = [&](u4 (&a)[4][2], u4 (&b)[4][2], int ks) {
#pragma unroll
      for (int mt = 0; mt < 4; mt++) { a[mt][0] = a_ptr[mt][8 * ks]; a[mt][1] = a_ptr[mt][8 * ks + 1]; }
#pragma unroll
      for (int nt = 0; nt < 4; nt++) { b[nt][0] = W1[((size_t)ks * 16 + nt) * 128]; b[nt][1] = W1[((size_t)ks * 16 + nt) * 128 + 64]; }
    };
    auto mac_set = [&](u4 (&a)[4][2], u4 (&b)[4][2]) {
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        h8 ah, al;
        gather_hl(a[mt][0], a[mt][1], ah, al);
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = mfma3(ah, al, __builtin_bit_cast(h8, b[nt][0]), __builtin_bit_cast(h8, b[nt][1]), acc[mt][nt]);
      }
    };
    load_set(a0, b0, 0);
#pragma unroll 1
    for (int ks = 0; ks < 64; ks += 2) {
      load_set(a1, b1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mac_set(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_set(a0, b0, ks + 2 < 64 ? ks + 2 : 0);
      __builtin_amdgcn_sched_barrier(0);
      mac_set(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float GLB *bias = W + HEVCDL_W_FC1 + 2048 * 256 + wave * 64 + i16;
    const float inv = W[HEVCDL_W_SCALES + 4];        // 1 / (weight scale * activation scale): a power of two, the product below is exact (hevcdl_dev.h HEVCDL_ACT_SCALE)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      const float b = bias[nt * 16];
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        const v4f r = acc[mt][nt];
        float *d = sm.h1 + (mt * 16 + g4 * 4) * H1_ROW + wave * 64 + nt * 16 + i16;
        d[0] = fmaxf(r.x * inv + b, 0.f); d[H1_ROW] = fmaxf(r.y * inv + b, 0.f); d[2 * H1_ROW] = fmaxf(r.z * inv + b, 0.f); d[3 * H1_ROW] = fmaxf(r.w * inv + b, 0.f);
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
    __syncthreads();                          // (h2 overlays h1)
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      float *d = sm.s.h2 + (mt * 16 + g4 * 4) * H2_ROW + wave * 16 + i16;
      d[0] = fmaxf(acc[mt].x + b, 0.f); d[H2_ROW] = fmaxf(acc[mt].y + b, 0.f); d[2 * H2_ROW] = fmaxf(acc[mt].z + b, 0.f); d[3 * H2_ROW] = fmaxf(acc[mt].w + b, 0.f);
    }
  }
  __syncthreads();
  { // ---- fc3: wave w -> rows [16w, 16w + 16), 16 logits -----------------------------------------------------------------
    const float GLB *W3 = W + HEVCDL_W_FC3 + (size_t)g4 * 16 + i16;
    v4f acc = { 0, 0, 0, 0 };
#pragma unroll
    for (int k0 = 0; k0 < 64; k0 += 4) acc = mfma4(sm.s.h2[(wave * 16 + i16) * H2_ROW + k0 + g4], W3[(size_t)k0 * 16], acc);
    const float b = W[HEVCDL_W_FC3 + 64 * 16 + i16];
    const float v[4] = { acc.x + b, acc.y + b, acc.z + b, acc.w + b };
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = wave * 16 + g4 * 4 + r;
      sm.s.lg[row][i16] = v[r];
      if (p.logits && row0 + row < n_rows) ((float GLB *)p.logits)[(size_t)(row0 + row) * 16 + i16] = v[r];
    }
  }
  }
  __syncthreads();

  // ---- labels of the 16 CTUs: 4x argmax + fix-ups (use_model.py:101-119), then the boundary clamp; one thread per CTU ----
  if (tid < 16 && blockIdx.x * 16 + tid < p.n_ctus) {
    const int gctu = p.ctu_base + blockIdx.x * 16 + tid;
    const int addr = gctu % p.ctus_per_frame, x0 = (addr % p.ctus_x) * 64, y0 = (addr / p.ctus_x) * 64;
    const float (*lg)[16] = &sm.s.lg[tid * 4];
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
