// cnn_kernel.hip -- on-device CNN depth predictor for gfx950 (MI355X), one 256-thread workgroup per CTU.
//
// Replaces the reference's PyTorch sidecar (/root/reference/use_model.py):
//   ConvNet2.forward                      use_model.py:16-58   (BatchNorm in TRAINING mode, :61-63)
//   CTU / quadrant tiling, zero padding   use_model.py:80-95
//   4x argmax + label fix-ups             use_model.py:101-119
//   label file IPC                        use_model.py:121-125 <-> TEncCu.cpp:244-253   (labels stay in HBM)
// plus the YUV->RGB input transform and the boundary clamp defined by this project.
//
// Whole network for the 4 quadrants of one CTU runs inside one workgroup with all activations in LDS
// (~122 KB): conv+BN+ReLU+pool are fused (BN statistics are per sample = per workgroup, so no global
// reduction exists); the conv64 branch is evaluated once and shared by the 4 quadrants (identical input,
// identical per-sample statistics).  Weights are pre-packed [k][oc] so that the oc run of one tap is
// contiguous: taps are wave-uniform and come through the scalar cache (s_load), activations come from LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hevcdl_dev.h"

namespace {

// weights are read-only and tap-uniform: constant address space -> scalar loads (s_load_dwordx*)
typedef __attribute__((address_space(4))) const float cfloat;

constexpr int A_STRIDE = 18;               // 16x16 map + 1-pixel zero halo
constexpr int A2_STRIDE = 10;              // 8x8 map + halo

struct CnnSmem {
  uint8_t in[3][64][64];                   // RGB planes of the CTU (zero past the picture edge)
  float lut[256];                          // u8 -> u8/255 (ToTensor, use_model.py:94-95)
  float a64[16][A_STRIDE * A_STRIDE];      // conv64 branch output (shared by the 4 quadrants)
  float a1[16][A_STRIDE * A_STRIDE];       // conv1 output of the current quadrant
  float a2[64][A2_STRIDE * A2_STRIDE];     // conv2 output of the current quadrant
  float a3[2048][4];                       // conv3 outputs, [flatten index][quadrant]
  double red[2][4][64];                    // per-wave partial sums / sums of squares
  float alpha[128], beta[128];             // BN folded to y = x*alpha + beta
  float h1[4][256], h2[4][64], lg[4][16];
};

__device__ __forceinline__ double shfl_xor_d(double v, int m)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}

// transposing butterfly: every lane holds N per-channel values; afterwards lane l holds in v[0] the
// wave-wide sum of channel (l % N).  N = 64 or 32, wave = 64 lanes.
template <int N>
__device__ __forceinline__ void wave_channel_sums(double (&v)[N], int lane)
{
  if (N == 32) {
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] += shfl_xor_d(v[i], 32);
  }
  int count = N;
#pragma unroll
  for (int mask = N / 2; mask >= 1; mask >>= 1) {
    const int half = count / 2;
    const bool up = (lane & mask) != 0;
#pragma unroll
    for (int i = 0; i < half; i++) {
      double keep = up ? v[i + half] : v[i];
      double send = up ? v[i] : v[i + half];
      v[i] = keep + shfl_xor_d(send, mask);
    }
    count = half;
  }
}

// BN in training mode folded to an affine map (nn.BatchNorm2d defaults: biased variance, eps 1e-5)
__device__ __forceinline__ void bn_fold(double s, double ss, double n, float gamma, float beta, float &alpha, float &betap)
{
  double mean = s / n;
  double var = ss / n - mean * mean;
  if (var < 0) var = 0;
  double inv = 1.0 / sqrt(var + 1e-5);
  alpha = (float)(inv * (double)gamma);
  betap = (float)((double)beta - mean * inv * (double)gamma);
}

// 5x5 conv (3 -> 16, pad 2 relative to the REGION) + BN(train) + ReLU + POOLxPOOL max pool -> 16 x 16x16.
// Region = size x size square of the CTU at (rx, ry); zero padding outside the region (the reference crops the
// 32x32 quadrant first, use_model.py:92, so the padding is not the neighbouring CTU pixels).
template <int POOL>
__device__ __noinline__ void conv5_block(CnnSmem &sm, cfloat *w, cfloat *bias, cfloat *gamma, cfloat *betaw,
                            int rx, int ry, float (*out)[A_STRIDE * A_STRIDE], int tid)
{
  constexpr int SIZE = 16 * POOL;
  const int py = tid >> 4, px = tid & 15;
  float mx[16], mn[16]; double s[16], ss[16];
#pragma unroll
  for (int o = 0; o < 16; o++) { mx[o] = -3.4e38f; mn[o] = 3.4e38f; s[o] = 0; ss[o] = 0; }
#pragma unroll 1
  for (int wy = 0; wy < POOL; wy++)
#pragma unroll 1
    for (int wx = 0; wx < POOL; wx++) {
      const int y = py * POOL + wy, x = px * POOL + wx;
      float acc[16];
#pragma unroll
      for (int o = 0; o < 16; o++) acc[o] = bias[o];
#pragma unroll 1
      for (int c = 0; c < 3; c++)
#pragma unroll 1
        for (int ky = 0; ky < 5; ky++) {
          const int yy = y + ky - 2;
          const bool yok = (yy >= 0) && (yy < SIZE);
#pragma unroll
          for (int kx = 0; kx < 5; kx++) {
            const int xx = x + kx - 2;
            float v = 0.f;
            if (yok && xx >= 0 && xx < SIZE) v = sm.lut[sm.in[c][ry + yy][rx + xx]];
            cfloat *wp = w + ((c * 5 + ky) * 5 + kx) * 16;
#pragma unroll
            for (int o = 0; o < 16; o++) acc[o] = fmaf(v, wp[o], acc[o]);
          }
        }
#pragma unroll
      for (int o = 0; o < 16; o++) {
        mx[o] = fmaxf(mx[o], acc[o]); mn[o] = fminf(mn[o], acc[o]);
        s[o] += (double)acc[o]; ss[o] += (double)acc[o] * (double)acc[o];
      }
    }
  // block-wide per-channel sums
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int o = 0; o < 16; o++) {
    double a = s[o], b = ss[o];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a += shfl_xor_d(a, m); b += shfl_xor_d(b, m); }
    if (lane == 0) { sm.red[0][wave][o] = a; sm.red[1][wave][o] = b; }
  }
  __syncthreads();
  if (tid < 16) {
    double a = 0, b = 0;
    for (int k = 0; k < 4; k++) { a += sm.red[0][k][tid]; b += sm.red[1][k][tid]; }
    bn_fold(a, b, (double)(SIZE * SIZE), gamma[tid], betaw[tid], sm.alpha[tid], sm.beta[tid]);
  }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < 16; o++) {
    const float al = sm.alpha[o], be = sm.beta[o];
    const float v = (al >= 0.f) ? mx[o] : mn[o];     // x -> relu(x*al+be) is monotone: pool before the affine map
    out[o][(py + 1) * A_STRIDE + px + 1] = fmaxf(v * al + be, 0.f);
  }
  __syncthreads();
}

} // namespace

extern "C" __global__ __launch_bounds__(256)
void hevcdl_cnn_ctu_kernel(hevcdl_cnn_params p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  CnnSmem &sm = *reinterpret_cast<CnnSmem *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gctu = blockIdx.x;                       // global CTU index over all frames
  const int frame = gctu / p.ctus_per_frame, addr = gctu - frame * p.ctus_per_frame;
  const int x0 = (addr % p.ctus_x) * 64, y0 = (addr / p.ctus_x) * 64;
  cfloat *W = (cfloat *)p.weights;

  // ---- stage 0: CTU input -> LDS (coalesced rows of the planes), LUT, zero the halo'd maps -------------
  sm.lut[tid] = (float)tid / 255.0f;
  if (p.input_mode == HEVCDL_DEV_INPUT_RGB_CTU) {
    const uint8_t *src = p.input + (size_t)gctu * (64 * 64 * 3);
    for (int i = tid; i < 64 * 64 * 3; i += 256) { int pix = i / 3, c = i - pix * 3; sm.in[c][pix >> 6][pix & 63] = src[i]; }
  } else {
    const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
    const uint8_t *Y = p.input + (size_t)frame * fsz, *U = Y + ysz, *V = U + (ysz >> 2);
    const int cw = p.width >> 1;
    for (int i = tid; i < 64 * 64; i += 256) {
      const int y = i >> 6, x = i & 63, gx = x0 + x, gy = y0 + y;
      int r = 0, g = 0, b = 0;
      if (gx < p.width && gy < p.height) {
        const int yv = Y[(size_t)gy * p.width + gx];
        if (p.input_mode == HEVCDL_DEV_INPUT_LUMA) { r = g = b = yv; }
        else {
          const int d = (int)U[(size_t)(gy >> 1) * cw + (gx >> 1)] - 128, e = (int)V[(size_t)(gy >> 1) * cw + (gx >> 1)] - 128, c = yv - 16;
          r = (298 * c + 409 * e + 128) >> 8; g = (298 * c - 100 * d - 208 * e + 128) >> 8; b = (298 * c + 516 * d + 128) >> 8;
          r = r < 0 ? 0 : (r > 255 ? 255 : r); g = g < 0 ? 0 : (g > 255 ? 255 : g); b = b < 0 ? 0 : (b > 255 ? 255 : b);
        }
      }
      sm.in[0][y][x] = (uint8_t)r; sm.in[1][y][x] = (uint8_t)g; sm.in[2][y][x] = (uint8_t)b;
    }
  }
  for (int i = tid; i < 16 * A_STRIDE * A_STRIDE; i += 256) { (&sm.a64[0][0])[i] = 0.f; (&sm.a1[0][0])[i] = 0.f; }
  for (int i = tid; i < 64 * A2_STRIDE * A2_STRIDE; i += 256) (&sm.a2[0][0])[i] = 0.f;
  __syncthreads();

  // ---- conv64 branch, once per CTU (use_model.py:38-43) --------------------------------------------
  conv5_block<4>(sm, W + HEVCDL_W_C64, W + HEVCDL_W_C64 + 1200, W + HEVCDL_W_C64 + 1216, W + HEVCDL_W_C64 + 1232, 0, 0, sm.a64, tid);

#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    // ---- conv1 on the 32x32 quadrant (use_model.py:20-25) ------------------------------------------
    conv5_block<2>(sm, W + HEVCDL_W_C1, W + HEVCDL_W_C1 + 1200, W + HEVCDL_W_C1 + 1216, W + HEVCDL_W_C1 + 1232,
                   (q & 1) * 32, (q >> 1) * 32, sm.a1, tid);

    // ---- conv2: 32 -> 64, 3x3, on cat(conv1, conv64) (use_model.py:26-31, 50) ----------------------
    // wave w owns output channels [16w, 16w+16); lane = one 2x2 pool window (its 4 positions), so the
    // pool is register-local and the BN statistics of a channel live inside one wave.
    {
      const int wy = lane >> 3, wx = lane & 7;
      cfloat *w2 = W + HEVCDL_W_C2 + wave * 16, *b2 = W + HEVCDL_W_C2 + 18432;
      float acc[4][16];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int o = 0; o < 16; o++) acc[i][o] = b2[wave * 16 + o];
#pragma unroll 1
      for (int ic = 0; ic < 32; ic++) {
        const float *src = ((ic < 16) ? sm.a1[ic] : sm.a64[ic - 16]) + (2 * wy) * A_STRIDE + 2 * wx;
        float patch[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int c = 0; c < 4; c++) patch[r][c] = src[r * A_STRIDE + c];
#pragma unroll
        for (int k = 0; k < 9; k++) {
          cfloat *wp = w2 + (ic * 9 + k) * 64;
#pragma unroll
          for (int o = 0; o < 16; o++) {
            const float wv = wp[o];
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][o] = fmaf(patch[(i >> 1) + k / 3][(i & 1) + k % 3], wv, acc[i][o]);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 16; o++) {
        double a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { a += (double)acc[i][o]; b += (double)acc[i][o] * (double)acc[i][o]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { a += shfl_xor_d(a, m); b += shfl_xor_d(b, m); }
        const int ch = wave * 16 + o;
        float al, be;
        bn_fold(a, b, 256.0, b2[64 + ch], b2[128 + ch], al, be);
        float v = fmaxf(fmaxf(acc[0][o] * al + be, acc[1][o] * al + be), fmaxf(acc[2][o] * al + be, acc[3][o] * al + be));
        sm.a2[ch][(wy + 1) * A2_STRIDE + wx + 1] = fmaxf(v, 0.f);
      }
      __syncthreads();
    }
    // ---- conv3: 64 -> 128, 3x3 (use_model.py:32-37); wave w owns output channels [32w, 32w+32) -----
    {
      const int widx = lane >> 2;
      const int y = 2 * (widx >> 2) + ((lane >> 1) & 1), x = 2 * (widx & 3) + (lane & 1);
      cfloat *w3 = W + HEVCDL_W_C3 + wave * 32, *b3 = W + HEVCDL_W_C3 + 73728;
      float acc[32];
#pragma unroll
      for (int o = 0; o < 32; o++) acc[o] = b3[wave * 32 + o];
#pragma unroll 1
      for (int ic = 0; ic < 64; ic++) {
#pragma unroll 3
        for (int k = 0; k < 9; k++) {
          const float v = sm.a2[ic][(y + k / 3) * A2_STRIDE + x + (k % 3)];
          cfloat *wp = w3 + (ic * 9 + k) * 128;
#pragma unroll
          for (int o = 0; o < 32; o++) acc[o] = fmaf(v, wp[o], acc[o]);
        }
      }
      double d[32];
#pragma unroll
      for (int o = 0; o < 32; o++) d[o] = (double)acc[o];
      wave_channel_sums<32>(d, lane);
      const double s = d[0];
#pragma unroll
      for (int o = 0; o < 32; o++) d[o] = (double)acc[o] * (double)acc[o];
      wave_channel_sums<32>(d, lane);
      if (lane < 32) {
        const int ch = wave * 32 + lane;
        bn_fold(s, d[0], 64.0, b3[128 + ch], b3[256 + ch], sm.alpha[ch], sm.beta[ch]);
      }
      __syncthreads();
#pragma unroll
      for (int o = 0; o < 32; o++) {
        const int ch = wave * 32 + o;
        float v = fmaxf(acc[o] * sm.alpha[ch] + sm.beta[ch], 0.f);
        v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2));
        if ((lane & 3) == 0) sm.a3[ch * 16 + widx][q] = v;       // flatten order (C,H,W), use_model.py:53
      }
      __syncthreads();
    }
  }

  // ---- fc1 (2048 -> 256) for the 4 quadrants at once; weights pre-transposed [k][j] ---------------------
  {
    const float *f1 = p.weights + HEVCDL_W_FC1;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
    for (int k = 0; k < 2048; k++) {
      const float w = f1[(size_t)k * 256 + tid];
      const float4 xv = *reinterpret_cast<const float4 *>(sm.a3[k]);
      a0 = fmaf(w, xv.x, a0); a1 = fmaf(w, xv.y, a1); a2 = fmaf(w, xv.z, a2); a3 = fmaf(w, xv.w, a3);
    }
    const float b = f1[2048 * 256 + tid];
    sm.h1[0][tid] = fmaxf(a0 + b, 0.f); sm.h1[1][tid] = fmaxf(a1 + b, 0.f);
    sm.h1[2][tid] = fmaxf(a2 + b, 0.f); sm.h1[3][tid] = fmaxf(a3 + b, 0.f);
  }
  __syncthreads();
  {
    const float *f2 = p.weights + HEVCDL_W_FC2; const int q = tid >> 6, j = tid & 63;
    float a = 0;
    for (int k = 0; k < 256; k++) a = fmaf(sm.h1[q][k], f2[k * 64 + j], a);
    sm.h2[q][j] = fmaxf(a + f2[256 * 64 + j], 0.f);
  }
  __syncthreads();
  if (tid < 64) {
    const float *f3 = p.weights + HEVCDL_W_FC3; const int q = tid >> 4, j = tid & 15;
    float a = 0;
    for (int k = 0; k < 64; k++) a = fmaf(sm.h2[q][k], f3[k * 16 + j], a);
    a += f3[64 * 16 + j];
    sm.lg[q][j] = a;
    if (p.logits) p.logits[(size_t)gctu * 64 + q * 16 + j] = a;
  }
  __syncthreads();

  // ---- labels: 4x argmax + fix-ups (use_model.py:101-119), then the boundary clamp ---------------------
  if (tid == 0) {
    uint8_t lab[16];
    const int quads[4][4] = { {0, 1, 4, 5}, {2, 3, 6, 7}, {8, 9, 12, 13}, {10, 11, 14, 15} };
    for (int q = 0; q < 4; q++) {
      int d[4]; bool any0 = false, all0 = true, any1 = false, all1 = true;
      for (int k = 0; k < 4; k++) {
        int best = 0; float bv = sm.lg[q][4 * k];
        for (int j = 1; j < 4; j++) if (sm.lg[q][4 * k + j] > bv) { bv = sm.lg[q][4 * k + j]; best = j; }   // first maximum wins
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
    if (p.clamp) {
      int mxl = 0;
      for (int c = 0; c < 16; c++) {
        const int px = x0 + (c & 3) * 16, py = y0 + (c >> 2) * 16;
        int md = 0;
        if (px < p.width && py < p.height) {
          while (md < 3) { const int s = 64 >> md; if ((px / s) * s + s <= p.width && (py / s) * s + s <= p.height) break; md++; }
        }
        if (lab[c] < md) lab[c] = (uint8_t)md;
        if (lab[c] > mxl) mxl = lab[c];
      }
      if (mxl > 0) for (int c = 0; c < 16; c++) if (lab[c] < 1) lab[c] = 1;
      for (int q = 0; q < 4; q++) {
        int m = 0; for (int k = 0; k < 4; k++) if (lab[quads[q][k]] > m) m = lab[quads[q][k]];
        if (m >= 2) for (int k = 0; k < 4; k++) if (lab[quads[q][k]] < 2) lab[quads[q][k]] = 2;
      }
    }
    for (int c = 0; c < 16; c++) p.labels[(size_t)gctu * 16 + c] = lab[c];
  }
}

extern "C" size_t hevcdl_cnn_smem_bytes(void) { return sizeof(CnnSmem); }
