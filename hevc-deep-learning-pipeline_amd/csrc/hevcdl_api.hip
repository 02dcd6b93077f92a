// hevcdl_api.hip -- host side of the C ABI declared in include/hevcdl.h (compiled by hipcc for gfx950).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "hevcdl.h"
#include "hevcdl_dev.h"

extern "C" __global__ void hevcdl_cnn_ctu_kernel(hevcdl_cnn_params p);
extern "C" __global__ void hevcdl_fc_kernel(hevcdl_fc_params p);          // fc_kernel.hip: fully connected head + labels, 16 CTUs per workgroup
extern "C" __global__ void hevcdl_rd_frame_kernel(hevcdl_rd_params p);
extern "C" __global__ void hevcdl_rd_frame_kernel_bd10(hevcdl_rd_params p);       // rd_kernel_bd10.hip: the same kernel for uint16 samples
extern "C" size_t hevcdl_rd_smem_bytes_bd10(void);
extern "C" size_t hevcdl_rd_scratch_bytes_bd10(void);
extern "C" int hevcdl_rd_waves_per_group(void);
extern "C" __global__ void hevcdl_rd_frame_kernel_wide(hevcdl_rd_params p);       // rd_kernel_wide.hip: the 8-bit kernel with more wavefronts per workgroup
extern "C" size_t hevcdl_rd_smem_bytes_wide(void);
extern "C" size_t hevcdl_rd_scratch_bytes_wide(void);
extern "C" int hevcdl_rd_waves_per_group_wide(void);
extern "C" __global__ void hevcdl_rd_frame_kernel_tools(hevcdl_rd_params p);      // rd_kernel_tools.hip: the 8-bit kernel with the cfg's tool switches read at run time
extern "C" size_t hevcdl_rd_smem_bytes_tools(void);
extern "C" size_t hevcdl_rd_scratch_bytes_tools(void);

// 10-bit samples -> the 8-bit planes the CNN stage reads (the reference's label producer works on 8-bit frames: gen_frames.py)
__global__ void hevcdl_narrow_samples_kernel(const uint16_t *src, uint8_t *dst, size_t n, int shift)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint8_t)(src[i] >> shift);
}

// Boundary policy HEVCDL_BOUNDARY_CLAMP applied to caller-supplied labels (the label files of the reference's use_model.py are unclamped):
// every cell is raised to the smallest depth whose CU lies inside the picture, then the quadtree is made consistent again (a split CTU has
// no depth-0 cell, a split 32x32 quadrant no depth-1 cell) -- the same rule the label stage of fc_kernel.hip applies to its own output.
// Labels above 3 are a caller error: *bad is set.  One thread per CTU.
__global__ void hevcdl_clamp_labels_kernel(uint8_t *labels, int n_ctus, int ctus_per_frame, int ctus_x, int width, int height, int *bad)
{
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_ctus) return;
  const int addr = g % ctus_per_frame, x0 = (addr % ctus_x) * 64, y0 = (addr / ctus_x) * 64;
  uint8_t *lab = labels + (size_t)g * 16;
  uint8_t l[16];
  for (int c = 0; c < 16; c++) { l[c] = lab[c]; if (l[c] > 3) { *bad = 1; l[c] = 3; } }
  hevcdl_clamp_ctu_labels(l, x0, y0, width, height);
  for (int c = 0; c < 16; c++) lab[c] = (uint8_t)l[c];
}

struct hevcdl_ctx {
  char last_rd[160] = { 0 };           // which build and launch form the last decision launch took (hevcdl_last_rd_launch)
  hevcdl_config cfg;
  int ctus_x, ctus_y, ctus, n_cus;
  int col_bd[21], row_bd[23];    // tile boundaries in CTUs
  size_t frame_bytes;
  float *d_weights;
  unsigned char *d_scratch;      // decision kernel workspace: one block per wave of every workgroup of a launch; allocated by the first launch that needs it, grown when a later one needs more (ensure_scratch)
  size_t scratch_bytes;
  size_t scratch_per_wave; int rd_groups, remote_groups;   // workgroups of a launch: one per CU, fewer when the context cannot hold that many units
  // staging buffers for the host-pointer entry points
  uint8_t *d_yuv, *d_labels, *d_recon; unsigned char *d_records, *d_stats; float *d_logits; uint8_t *d_rgb; size_t rgb_cap;
  uint8_t *d_yuv8;               // 8-bit copy of 10-bit input for the CNN stage
  uint8_t *d_picture;            // final pictures of hevcdl_encode_pictures (SAO output)
  unsigned char *h_chunk[2]; size_t h_chunk_bytes; hipStream_t copy_stream; hipEvent_t copy_ev[2];   // hevcdl_encode_pictures_chunked: two page-locked chunk buffers
  float *d_a3; size_t a3_ctus;   // conv3 outputs of one chunk of CTUs (32 KB per CTU): the hand-over from the conv kernel to the head kernel
  hipStream_t stream;
  // per-CTU session (hevcdl_begin_frames / hevcdl_compress_ctu): coder state after the last CTU of every frame, next CTU expected
  unsigned char *d_cabac; std::vector<int> next_ctu; int session_frames;
  unsigned char *d_sao_stats, *d_sao_recon, *d_sao_params, *d_sao_cand;
  unsigned char *d_wide;                 // 16-bit staging of hevcdl_*_planes with sample_bytes 2 on an 8-bit context   // SAO workspace
  int *d_flag;                   // device-side error flag of the label check
  int wpp_ring; size_t wpp_state_bytes;
  unsigned char *d_wpp;          // WaveFrontSynchro: [max_frames][ctus_y] 256 bytes (the contexts behind a row's second CTU, the row's finished-CTU count: rd_kernel.hip)
  unsigned char *d_sched;        // decision kernel: hand-over of units between workgroups (finished counter, per-workgroup unit counts, mailboxes)
  bool profile;
  std::vector<hipEvent_t> ev_cnn, ev_rd, ev_conv;       // start/stop pairs (ev_conv: the convolution kernel alone, one pair per chunk of CTUs)
  char err[256];
};

static const int CHROMA_SCALE_420[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
  29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };   // TComRom.cpp:536
static const int QUANT_SCALES[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };   // TComRom.cpp:354-357
static const int INV_QUANT_SCALES[6] = { 40, 45, 51, 57, 64, 72 };                  // TComRom.cpp:359-362

static hevcdl_status fail(hevcdl_ctx *c, hevcdl_status s, const char *what, hipError_t e = hipSuccess)
{
  if (c) snprintf(c->err, sizeof c->err, "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
  return s;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ctx, HEVCDL_ERR_HIP, #call, e_); } while (0)

extern "C" int hevcdl_ctus_per_frame(int w, int h) { return ((w + 63) >> 6) * ((h + 63) >> 6); }
extern "C" size_t hevcdl_frame_bytes(int w, int h) { return (size_t)w * h * 3 / 2; }
extern "C" size_t hevcdl_frame_bytes_bd(int w, int h, int bit_depth) { return (size_t)w * h * 3 / 2 * (bit_depth > 8 ? 2 : 1); }

extern "C" hevcdl_status hevcdl_config_default(hevcdl_config *cfg, int width, int height, int qp)
{
  return hevcdl_config_default_bd(cfg, width, height, qp, 8);
}

extern "C" hevcdl_status hevcdl_config_default_bd(hevcdl_config *cfg, int width, int height, int qp, int bit_depth)
{
  if (bit_depth != 8 && bit_depth != 10) return HEVCDL_ERR_UNSUPPORTED;
  if (!cfg || width <= 0 || height <= 0 || (width & 7) || (height & 7) || qp < 0 || qp > 51) return HEVCDL_ERR_INVALID_ARG;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = sizeof *cfg;
  cfg->width = width; cfg->height = height; cfg->bit_depth = bit_depth; cfg->chroma_format = 420; cfg->qp = qp;
  cfg->ctu_size = 64; cfg->max_partition_depth = 4; cfg->tu_log2_min = 2; cfg->tu_log2_max = 5; cfg->tu_max_depth_intra = 3;
  cfg->tools = HEVCDL_TOOLS_REFERENCE; cfg->bn_mode = HEVCDL_BN_REFERENCE; cfg->boundary_policy = HEVCDL_BOUNDARY_CLAMP;
  cfg->cnn_input = HEVCDL_CNN_INPUT_RGB601; cfg->device = 0; cfg->max_frames = 1; cfg->tile_columns = 1; cfg->tile_rows = 1; cfg->tile_uniform_spacing = 1; cfg->lf_across_tiles = 1;
  // TEncSlice::calculateLambda (TEncSlice.cpp:433-527) for an all-intra GOP of 1, then setUpLambda (:112-140)
  cfg->lambda = 0.57 * 1.0 * pow(2.0, (qp - 12) / 3.0);
  cfg->sqrt_lambda = sqrt(cfg->lambda);                         // TComRdCost::setLambda TComRdCost.cpp:109-122
  cfg->qp_chroma = CHROMA_SCALE_420[qp];
  cfg->chroma_weight = pow(2.0, (qp - cfg->qp_chroma) / 3.0);
  cfg->lambda_chroma = cfg->lambda / cfg->chroma_weight;
  for (int ch = 0; ch < 2; ch++) {
    // quantiser QP = QP + qpBdOffset (6 per extra bit, TComTrQuant.cpp:71-100); distortion is kept at 8-bit scale (FULL_NBIT 0:
    // DISTORTION_PRECISION_ADJUSTMENT, TypeDef.h:170), hence the 1 << 2(bd-8) divisors
    const int q = (ch ? cfg->qp_chroma : qp) + 6 * (bit_depth - 8), rem = q % 6, per = q / 6, dadj = 2 * (bit_depth - 8);
    for (int l = 0; l < 4; l++) {                               // TComTrQuant::setErrScaleCoeff TComTrQuant.cpp:3096-3126
      const int tshift = 15 - bit_depth - (l + 2);
      double s = (double)(1 << 15);
      s = s * pow(2.0, -2.0 * tshift);
      cfg->err_scale[ch][l] = s / QUANT_SCALES[rem] / QUANT_SCALES[rem] / (1 << dadj);
    }
    const double inv = (double)INV_QUANT_SCALES[rem], lam = ch ? cfg->lambda_chroma : cfg->lambda;
    cfg->sbh_rd_factor[ch] = (int64_t)(inv * inv * (1 << (2 * per)) / lam / 16 / (1 << dadj) + 0.5);   // TComTrQuant.cpp:2532-2535
  }
  return HEVCDL_OK;
}

// state_dict order of the reference checkpoint (fp32 tensors): offsets in floats
enum { B_C1W = 0, B_C1B = 1200, B_C1G = 1216, B_C1BE = 1232, B_C2W = 1280, B_C2B = 19712, B_C2G = 19776, B_C2BE = 19840,
       B_C3W = 20032, B_C3B = 93760, B_C3G = 93888, B_C3BE = 94016, B_F1W = 94400, B_F1B = 618688, B_F2W = 618944, B_F2B = 635328,
       B_F3W = 635392, B_F3B = 636416, B_C64W = 636432, B_C64B = 637632, B_C64G = 637648, B_C64BE = 637664 };

// f32 -> IEEE half, round to nearest even (weights are small: no overflow handling beyond saturation to the largest finite half)
static uint16_t f32_to_f16(float f)
{
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);
  if (x < 0x38800000u) { // subnormal half (or zero)
    if (x < 0x33000000u) return (uint16_t)sign;
    const int shift = 113 - (int)(x >> 23);
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const uint32_t r = m >> (shift + 13), rem = m & ((1u << (shift + 13)) - 1u), half = 1u << (shift + 12);
    return (uint16_t)(sign | (r + ((rem > half || (rem == half && (r & 1u))) ? 1u : 0u)));
  }
  const uint32_t r = ((x - 0x38000000u) >> 13), rem = x & 0x1fffu;
  return (uint16_t)(sign | (r + ((rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ? 1u : 0u)));
}
static float f16_to_f32(uint16_t h)
{
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) { if (!m) x = sign; else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; sh++; } x = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13); } }
  else x = sign | ((e + 112u) << 23) | (m << 13);
  float f; memcpy(&f, &x, 4); return f;
}
// conv2 / conv3 on v_mfma_f32_16x16x32_f16 with SPLIT operands (cnn_kernel.hip): a weight w is the pair hi = half(w), lo = half(w - hi).  B-operand packing:
// lane l supplies B[k = 8 * (l >> 4) + j][n = l & 15], j = 0..7, and k-value (l >> 4, j) of k-step s stands for half j & 1 of the channel PAIR 16 s + 4 (j >> 1) + (l >> 4)
// of the kernel's activation maps (a word of a map holds the hi -- or the lo -- halves of two channels: four words are an operand as they lie in LDS).  Layout: [oc/16 N-tiles][9 taps][ic/32 k-steps][hi | lo][64 lanes][8 halves],
// then bias, gamma, beta as floats -- the same number of bytes as the f32 packing it replaces.
// the 5x5 convolutions (3 input channels, 16 output channels = one N-tile): the 75 taps (c * 5 + ky) * 5 + kx in the order of hevcdl_conv5_slot_tap, 5 k-steps of 16.
// The A operand of a step is four raw split words (hi, lo, hi, lo, ...): B1 carries the weight's hi half against both halves, B2 its lo half against the hi half only.
// weight scale of a layer (hevcdl_dev.h HEVCDL_ACT_SCALE): the largest power of two that keeps max |w| * scale below 2^15 (f16 holds 65504), at most 2^24
static float layer_scale(const float *w, size_t n)
{
  float m = 0.f;
  for (size_t i = 0; i < n; i++) { const float a = fabsf(w[i]); if (a > m && a <= 3.0e38f) m = a; }
  float sc = 1.f;
  if (m > 0.f) while (sc < 16777216.f && m * sc * 2.f < 32768.f) sc *= 2.f;
  return sc;
}
// sc: the layer's weight scale; the bias enters the accumulator, which holds sc * HEVCDL_ACT_SCALE times the layer's output
static void pack_conv5(const float *w, const float *b, const float *g, const float *be, float *dst, float sc)
{
  uint16_t *d16 = (uint16_t *)dst;
  for (int s = 0; s < 5; s++) for (int l = 0; l < 64; l++) for (int e = 0; e < 8; e++) {
    const int k = hevcdl_conv5_slot_tap(16 * s + 4 * (l >> 4) + (e >> 1)), oc = l & 15;
    const float v = (k >= 0 ? w[oc * 75 + k] : 0.f) * sc;
    const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
    d16[(size_t)(s * 2) * 512 + (size_t)l * 8 + e] = hi; d16[(size_t)(s * 2 + 1) * 512 + (size_t)l * 8 + e] = (e & 1) ? (uint16_t)0 : lo;
  }
  for (int c = 0; c < 16; c++) dst[HEVCDL_W_C5 + c] = b[c] * sc * HEVCDL_ACT_SCALE;
  memcpy(dst + HEVCDL_W_C5 + 16, g, 16 * sizeof(float)); memcpy(dst + HEVCDL_W_C5 + 32, be, 16 * sizeof(float));
}
static void pack_conv3(const float *w, const float *b, const float *g, const float *be, int oc, int ic, float *dst, float sc)
{
  uint16_t *d16 = (uint16_t *)dst;
  const int ks = ic / 32;
  for (int nt = 0; nt < oc / 16; nt++) for (int tap = 0; tap < 9; tap++) for (int s = 0; s < ks; s++) for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) {
    // element j of lane group g = l >> 4: half j & 1 of the map's channel PAIR m = 16 s + 4 (j >> 1) + g (cnn_kernel.hip: the operand is four words of a plane of pairs).
    // conv2's input (32 channels): pair m = channels 2m | 2m + 1;  conv3's input (64): pairs 0..15 = channels m | m + 16, pairs 16..31 = 16 + m | 32 + m
    const int m = 16 * s + 4 * (j >> 1) + (l >> 4), hf = j & 1;
    const int c = ic == 32 ? 2 * m + hf : (m < 16 ? m + 16 * hf : 16 + m + 16 * hf), o = nt * 16 + (l & 15);
    const float v = w[((size_t)o * ic + c) * 9 + tap] * sc;
    const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
    const size_t base = ((((size_t)nt * 9 + tap) * ks + s) * 2) * 512;        // halves: 64 lanes x 8 per operand
    d16[base + (size_t)l * 8 + j] = hi; d16[base + 512 + (size_t)l * 8 + j] = lo;
  }
  float *t = dst + (size_t)9 * ic * oc;
  for (int c = 0; c < oc; c++) t[c] = b[c] * sc * HEVCDL_ACT_SCALE;
  memcpy(t + oc, g, oc * sizeof(float)); memcpy(t + 2 * oc, be, oc * sizeof(float));
}
// HEVCDL_BN_EVAL: BatchNorm with the checkpoint's running statistics (model.eval()) is the fixed affine map y = x * alpha + beta' with
// alpha = gamma / sqrt(running_var + eps), beta' = beta - running_mean * alpha; the kernels find the pair in the gamma / beta slots of the
// packed layer (state_dict order: weight, bias, running_mean, running_var follow each other, `oc` floats each).
// The map is applied to the accumulator (in_sc = weight scale * HEVCDL_ACT_SCALE times the layer's output) and writes activations scaled by HEVCDL_ACT_SCALE.
static void fold_bn_eval(const float *g, const float *be, int oc, float *dst_g, float *dst_be, float in_sc)
{
  const float *rm = be + oc, *rv = be + 2 * oc;
  for (int c = 0; c < oc; c++) {
    const double a = (double)g[c] / sqrt((double)rv[c] + 1e-5);
    dst_g[c] = (float)(a / (double)in_sc * (double)HEVCDL_ACT_SCALE); dst_be[c] = (float)(((double)be[c] - (double)rm[c] * a) * (double)HEVCDL_ACT_SCALE);
  }
}
// fc1 (2048 -> 256) for the split-f16 MFMA of fc_kernel.hip: [k-step of 32][N-tile of 16][hi | lo][64 lanes][8 halves], lane l element j <-> k = 32 * step + 8 * (l >> 4) + j,
// n = 16 * tile + (l & 15); the pairs take exactly the bytes of the f32 matrix, the bias follows as before
static void pack_fc1(const float *w, const float *b, float *dst, float sc)
{
  uint16_t *d16 = (uint16_t *)dst;
  for (int ks = 0; ks < 64; ks++) for (int nt = 0; nt < 16; nt++) for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) {
    const int k = 32 * ks + 8 * (l >> 4) + j, n = nt * 16 + (l & 15);
    const float v = w[(size_t)n * 2048 + k] * sc;
    const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
    const size_t base = (((size_t)ks * 16 + nt) * 2) * 512;
    d16[base + (size_t)l * 8 + j] = hi; d16[base + 512 + (size_t)l * 8 + j] = lo;
  }
  memcpy(dst + (size_t)2048 * 256, b, 256 * sizeof(float));
}
static void pack_fc(const float *w, const float *b, int out, int in, float *dst)
{ for (int j = 0; j < out; j++) for (int k = 0; k < in; k++) dst[(size_t)k * out + j] = w[(size_t)j * in + k]; memcpy(dst + (size_t)in * out, b, out * sizeof(float)); }

extern "C" hevcdl_status hevcdl_create(const hevcdl_config *cfg, const float *weights, size_t n_floats, hevcdl_ctx **out)
{
  if (!cfg || !out || !weights || cfg->struct_size != sizeof(hevcdl_config)) return HEVCDL_ERR_INVALID_ARG;
  if (n_floats != HEVCDL_WEIGHT_FLOATS) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7) || cfg->qp < 0 || cfg->qp > 51 || cfg->max_frames < 1) return HEVCDL_ERR_INVALID_ARG;
  // keys that would change the path are rejected, not ignored
  if ((cfg->bit_depth != 8 && cfg->bit_depth != 10) || cfg->chroma_format != 420 || cfg->ctu_size != 64 || cfg->max_partition_depth != 4 || cfg->tu_log2_min != 2 ||
      cfg->tu_log2_max != 5 || cfg->tu_max_depth_intra != 3 || !HEVCDL_TOOLS_SUPPORTED(cfg->tools) || cfg->lf_beta_offset_div2 < -6 || cfg->lf_beta_offset_div2 > 6 || cfg->lf_tc_offset_div2 < -6 || cfg->lf_tc_offset_div2 > 6 || (cfg->bn_mode != HEVCDL_BN_REFERENCE && cfg->bn_mode != HEVCDL_BN_EVAL) ||
      cfg->boundary_policy != HEVCDL_BOUNDARY_CLAMP || (cfg->cnn_input != HEVCDL_CNN_INPUT_RGB601 && cfg->cnn_input != HEVCDL_CNN_INPUT_LUMA) ||
      (cfg->exec_flags & ~(HEVCDL_EXEC_NO_UNIT_HANDOVER | HEVCDL_EXEC_RD_WIDE | HEVCDL_EXEC_RD_NARROW)) ||
      ((cfg->exec_flags & HEVCDL_EXEC_RD_WIDE) && (cfg->exec_flags & HEVCDL_EXEC_RD_NARROW)))
    return HEVCDL_ERR_UNSUPPORTED;
  { // tiles: uniform spacing, every column at least 4 CTUs wide and every row 1 CTU high (TComPicSym.cpp:380-392), at most 20 x 22 (level 6.2)
    const int cx = (cfg->width + 63) >> 6, cy = (cfg->height + 63) >> 6;
    int cb[21], rb[23];
    if (cfg->tile_columns < 1 || cfg->tile_rows < 1 || cfg->tile_columns > 20 || cfg->tile_rows > 22) return HEVCDL_ERR_INVALID_ARG;
    const int tiled = cfg->tile_columns > 1 || cfg->tile_rows > 1;
    if (hevcdl_tile_bounds(cx, cfg->tile_columns, cfg->tile_uniform_spacing, cfg->tile_column_width, tiled ? 4 : 1, cb) ||
        hevcdl_tile_bounds(cy, cfg->tile_rows, cfg->tile_uniform_spacing, cfg->tile_row_height, 1, rb)) return HEVCDL_ERR_INVALID_ARG;
  }
  if (cfg->wavefront != 0 && cfg->wavefront != 1) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->wavefront && cfg->tile_columns * cfg->tile_rows > 1) return HEVCDL_ERR_UNSUPPORTED;      // as the reference (TAppEncCfg.cpp xCheckParameter: only the high-throughput profile has both)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) return HEVCDL_ERR_NO_DEVICE;
  hevcdl_ctx *ctx = new (std::nothrow) hevcdl_ctx();
  if (!ctx) return HEVCDL_ERR_OOM;
  ctx->cfg = *cfg; ctx->err[0] = 0; ctx->profile = false;
  ctx->ctus_x = (cfg->width + 63) >> 6; ctx->ctus_y = (cfg->height + 63) >> 6; ctx->ctus = ctx->ctus_x * ctx->ctus_y;
  ctx->frame_bytes = hevcdl_frame_bytes_bd(cfg->width, cfg->height, cfg->bit_depth);
  hevcdl_tile_bounds(ctx->ctus_x, cfg->tile_columns, cfg->tile_uniform_spacing, cfg->tile_column_width, 1, ctx->col_bd);
  hevcdl_tile_bounds(ctx->ctus_y, cfg->tile_rows, cfg->tile_uniform_spacing, cfg->tile_row_height, 1, ctx->row_bd);
  ctx->d_weights = nullptr; ctx->d_scratch = nullptr; ctx->scratch_bytes = 0; ctx->d_yuv = ctx->d_labels = ctx->d_recon = nullptr; ctx->d_records = ctx->d_stats = nullptr;
  ctx->d_logits = nullptr; ctx->d_yuv8 = nullptr; ctx->d_a3 = nullptr; ctx->a3_ctus = 0; ctx->d_picture = nullptr; ctx->h_chunk[0] = ctx->h_chunk[1] = nullptr; ctx->h_chunk_bytes = 0; ctx->copy_stream = nullptr; ctx->copy_ev[0] = ctx->copy_ev[1] = nullptr; ctx->d_rgb = nullptr; ctx->rgb_cap = 0; ctx->stream = nullptr; ctx->d_cabac = nullptr; ctx->session_frames = 0; ctx->d_sao_stats = ctx->d_sao_recon = ctx->d_sao_params = ctx->d_sao_cand = nullptr; ctx->d_wide = nullptr; ctx->d_flag = nullptr; ctx->d_sched = nullptr;
  hipError_t e;
#define CK(call) if ((e = (call)) != hipSuccess) { hevcdl_status s_ = (e == hipErrorOutOfMemory) ? HEVCDL_ERR_OOM : HEVCDL_ERR_HIP; hevcdl_destroy(ctx); return s_; }
  CK(hipSetDevice(cfg->device));
  { hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, cfg->device)); ctx->n_cus = prop.multiProcessorCount; }
  std::vector<float> pk(HEVCDL_W_TOTAL);
  // operand scales (hevcdl_dev.h HEVCDL_ACT_SCALE): weights of a layer by a power of two, activations by HEVCDL_ACT_SCALE; the kernels find 1 / (their product) behind the weights
  const float sc1 = layer_scale(weights + B_C1W, 1200), sc64 = layer_scale(weights + B_C64W, 1200), sc2 = layer_scale(weights + B_C2W, 9 * 32 * 64),
              sc3 = layer_scale(weights + B_C3W, 9 * 64 * 128), scf = layer_scale(weights + B_F1W, 2048 * 256);
  { const float scs[5] = { sc1, sc64, sc2, sc3, scf }; for (int i = 0; i < 8; i++) pk[HEVCDL_W_SCALES + i] = i < 5 ? 1.0f / (scs[i] * HEVCDL_ACT_SCALE) : 0.f; }
  pack_conv5(weights + B_C1W, weights + B_C1B, weights + B_C1G, weights + B_C1BE, pk.data() + HEVCDL_W_C1, sc1);
  pack_conv5(weights + B_C64W, weights + B_C64B, weights + B_C64G, weights + B_C64BE, pk.data() + HEVCDL_W_C64, sc64);
  pack_conv3(weights + B_C2W, weights + B_C2B, weights + B_C2G, weights + B_C2BE, 64, 32, pk.data() + HEVCDL_W_C2, sc2);
  pack_conv3(weights + B_C3W, weights + B_C3B, weights + B_C3G, weights + B_C3BE, 128, 64, pk.data() + HEVCDL_W_C3, sc3);
  if (cfg->bn_mode == HEVCDL_BN_EVAL) {
    fold_bn_eval(weights + B_C1G, weights + B_C1BE, 16, pk.data() + HEVCDL_W_C1 + HEVCDL_W_C5 + 16, pk.data() + HEVCDL_W_C1 + HEVCDL_W_C5 + 32, sc1 * HEVCDL_ACT_SCALE);
    fold_bn_eval(weights + B_C64G, weights + B_C64BE, 16, pk.data() + HEVCDL_W_C64 + HEVCDL_W_C5 + 16, pk.data() + HEVCDL_W_C64 + HEVCDL_W_C5 + 32, sc64 * HEVCDL_ACT_SCALE);
    fold_bn_eval(weights + B_C2G, weights + B_C2BE, 64, pk.data() + HEVCDL_W_C2 + 9 * 32 * 64 + 64, pk.data() + HEVCDL_W_C2 + 9 * 32 * 64 + 128, sc2 * HEVCDL_ACT_SCALE);
    fold_bn_eval(weights + B_C3G, weights + B_C3BE, 128, pk.data() + HEVCDL_W_C3 + 9 * 64 * 128 + 128, pk.data() + HEVCDL_W_C3 + 9 * 64 * 128 + 256, sc3 * HEVCDL_ACT_SCALE);
  }
  pack_fc1(weights + B_F1W, weights + B_F1B, pk.data() + HEVCDL_W_FC1, scf);
  pack_fc(weights + B_F2W, weights + B_F2B, 64, 256, pk.data() + HEVCDL_W_FC2);
  pack_fc(weights + B_F3W, weights + B_F3B, 16, 64, pk.data() + HEVCDL_W_FC3);
  CK(hipMalloc(&ctx->d_flag, sizeof(int)));
  CK(hipMalloc(&ctx->d_sched, 8192 + 1024 * 192));
  ctx->d_wpp = nullptr;
  if (cfg->wavefront) { // per (frame, row) 256 bytes, then the queue of claimable rows: 1024 bytes of counters + a ring of one int per frame (rounded up to a power of two)
    ctx->wpp_ring = 64; while (ctx->wpp_ring < cfg->max_frames) ctx->wpp_ring <<= 1;
    ctx->wpp_state_bytes = (size_t)256 * ctx->ctus_y * cfg->max_frames;
    CK(hipMalloc(&ctx->d_wpp, ctx->wpp_state_bytes + 1024 + (size_t)4 * ctx->wpp_ring));
  }
  if (cfg->bit_depth > 8) CK(hipMalloc(&ctx->d_yuv8, hevcdl_frame_bytes(cfg->width, cfg->height) * (size_t)cfg->max_frames));    // the CNN stage's 8-bit copy
  CK(hipMalloc(&ctx->d_weights, sizeof(float) * HEVCDL_W_TOTAL));
  CK(hipMemcpy(ctx->d_weights, pk.data(), sizeof(float) * HEVCDL_W_TOTAL, hipMemcpyHostToDevice));
  ctx->scratch_per_wave = cfg->bit_depth == 8 ? std::max(std::max(hevcdl_rd_scratch_bytes(), hevcdl_rd_scratch_bytes_wide()), hevcdl_rd_scratch_bytes_tools()) : hevcdl_rd_scratch_bytes_bd10();
  ctx->rd_groups = (int)std::min<long long>(ctx->n_cus, (long long)cfg->max_frames * (cfg->wavefront ? ctx->ctus_y : cfg->tile_columns * cfg->tile_rows));
  // (launches of few units run on every CU: the workgroups without a unit take second luma passes from the others, launch_rd -- they need a workspace too)
  ctx->remote_groups = (cfg->bit_depth == 8 && !(cfg->exec_flags & HEVCDL_EXEC_NO_UNIT_HANDOVER) && ctx->n_cus >= 8 && ctx->n_cus <= 1024) ? ctx->n_cus : 0;
  // (the workspace itself -- 1.6 MB per wave -- is sized by the launches: a context that only ever codes a frame or two in the independent form holds a few MB, not the
  // several GB a launch on every CU needs)
  CK(hipFuncSetAttribute((const void *)hevcdl_cnn_ctu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hevcdl_cnn_smem_bytes()));
  CK(hipFuncSetAttribute((const void *)hevcdl_fc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hevcdl_fc_smem_bytes()));
  { // conv -> head hand-over buffer for one chunk of CTUs, allocated up front so that no step pays for it
    const size_t chunk = (size_t)std::min<long long>((long long)cfg->max_frames * ctx->ctus, 131072);
    CK(hipMalloc(&ctx->d_a3, chunk * 4 * 2048 * sizeof(float))); ctx->a3_ctus = chunk;
  }
  CK(hipFuncSetAttribute((const void *)hevcdl_rd_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hevcdl_rd_smem_bytes()));
  CK(hipFuncSetAttribute((const void *)hevcdl_rd_frame_kernel_bd10, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hevcdl_rd_smem_bytes_bd10()));
  CK(hipFuncSetAttribute((const void *)hevcdl_rd_frame_kernel_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hevcdl_rd_smem_bytes_wide()));
  CK(hipFuncSetAttribute((const void *)hevcdl_rd_frame_kernel_tools, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hevcdl_rd_smem_bytes_tools()));
#undef CK
  *out = ctx;
  return HEVCDL_OK;
}

extern "C" void hevcdl_destroy(hevcdl_ctx *ctx)
{
  if (!ctx) return;
  hipSetDevice(ctx->cfg.device);
  hipDeviceSynchronize();
  for (hipEvent_t e : ctx->ev_cnn) hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_rd) hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_conv) hipEventDestroy(e);
  for (int i = 0; i < 2; i++) { if (ctx->h_chunk[i]) hipHostFree(ctx->h_chunk[i]); if (ctx->copy_ev[i]) hipEventDestroy(ctx->copy_ev[i]); }
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  hipFree(ctx->d_weights); hipFree(ctx->d_scratch); hipFree(ctx->d_yuv); hipFree(ctx->d_labels); hipFree(ctx->d_recon);
  hipFree(ctx->d_wpp); hipFree(ctx->d_records); hipFree(ctx->d_stats); hipFree(ctx->d_logits); hipFree(ctx->d_yuv8); hipFree(ctx->d_a3); hipFree(ctx->d_picture); hipFree(ctx->d_rgb); hipFree(ctx->d_cabac); hipFree(ctx->d_sao_stats); hipFree(ctx->d_sao_recon); hipFree(ctx->d_sao_params); hipFree(ctx->d_sao_cand); hipFree(ctx->d_wide); hipFree(ctx->d_flag); hipFree(ctx->d_sched);
  delete ctx;
}

extern "C" const char *hevcdl_last_error(const hevcdl_ctx *ctx) { return ctx ? ctx->err : "null ctx"; }

// page-locked host memory for the buffers of the host-pointer entry points (optional: any host memory works, pinned memory copies faster)
extern "C" hevcdl_status hevcdl_device_memory(int device, size_t *free_bytes, size_t *total_bytes)
{
  if (!free_bytes || !total_bytes) return HEVCDL_ERR_INVALID_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return HEVCDL_ERR_NO_DEVICE; }
  if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(free_bytes, total_bytes) != hipSuccess) { (void)hipGetLastError(); return HEVCDL_ERR_HIP; }
  return HEVCDL_OK;
}
extern "C" void *hevcdl_host_alloc(size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
extern "C" void hevcdl_host_free(void *p) { if (p) hipHostFree(p); }

static hevcdl_status ensure_staging(hevcdl_ctx *ctx)
{
  if (ctx->d_yuv) return HEVCDL_OK;
  const size_t nf = (size_t)ctx->cfg.max_frames;
  // all or nothing: a partial set would let later calls run on null buffers (records alone are 63 GB at 2048 frames of 2160p)
  void **bufs[6] = { (void **)&ctx->d_recon, (void **)&ctx->d_labels, (void **)&ctx->d_logits, (void **)&ctx->d_records, (void **)&ctx->d_stats, (void **)&ctx->d_yuv };
  const size_t sizes[6] = { ctx->frame_bytes * nf, (size_t)ctx->ctus * 16 * nf, (size_t)ctx->ctus * 64 * sizeof(float) * nf, (size_t)ctx->ctus * sizeof(hevcdl_ctu_record) * nf,
                            sizeof(hevcdl_frame_stats) * nf, ctx->frame_bytes * nf };
  for (int i = 0; i < 6; i++) {
    const hipError_t e = hipMalloc(bufs[i], sizes[i]);
    if (e != hipSuccess) {
      for (int j = 0; j <= i; j++) { hipFree(*bufs[j]); *bufs[j] = nullptr; }
      (void)hipGetLastError();
      return fail(ctx, e == hipErrorOutOfMemory ? HEVCDL_ERR_OOM : HEVCDL_ERR_HIP, "staging buffers (cfg.max_frames)", e);
    }
  }
  return HEVCDL_OK;
}

static void prof_begin(hevcdl_ctx *ctx, std::vector<hipEvent_t> &v, hipStream_t s)
{
  if (!ctx->profile) return;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, s); v.push_back(a); v.push_back(b);
}
static void prof_end(hevcdl_ctx *ctx, std::vector<hipEvent_t> &v, hipStream_t s) { if (ctx->profile) hipEventRecord(v.back(), s); }

static hevcdl_status launch_cnn(hevcdl_ctx *ctx, const void *d_in, int mode, int n_ctus, int clamp, void *d_labels, void *d_logits, hipStream_t s)
{
  hevcdl_cnn_params p;
  p.input = (const uint8_t *)d_in; p.weights = ctx->d_weights; p.labels = (uint8_t *)d_labels; p.logits = (float *)d_logits;
  p.input_mode = mode; p.width = ctx->cfg.width; p.height = ctx->cfg.height; p.ctus_x = ctx->ctus_x; p.ctus_per_frame = ctx->ctus; p.clamp = clamp;
  if (mode == HEVCDL_DEV_INPUT_RGB_CTU) { p.ctus_per_frame = n_ctus > 0 ? n_ctus : 1; p.ctus_x = p.ctus_per_frame; }
  // convolutions per CTU (one workgroup each) -> 32 KB of conv3 output per CTU in HBM -> the fully connected head batched over 16 CTUs
  // per workgroup.  The hand-over buffer holds one chunk of CTUs (<= 4 GiB); larger batches go through it chunk by chunk.
  const size_t chunk = (size_t)std::min<long long>(n_ctus, 131072);
  if (ctx->a3_ctus < chunk) { hipFree(ctx->d_a3); ctx->d_a3 = nullptr; ctx->a3_ctus = 0; HIPCHK(hipMalloc(&ctx->d_a3, chunk * 4 * 2048 * sizeof(float))); ctx->a3_ctus = chunk; }
  p.a3 = ctx->d_a3; p.n_cus = ctx->n_cus; p.bn_eval = ctx->cfg.bn_mode == HEVCDL_BN_EVAL;
  hevcdl_fc_params f;
  f.logits_in = nullptr;
  f.a3 = ctx->d_a3; f.weights = ctx->d_weights; f.width = p.width; f.height = p.height; f.ctus_x = p.ctus_x; f.ctus_per_frame = p.ctus_per_frame; f.clamp = clamp;
  prof_begin(ctx, ctx->ev_cnn, s);
  for (size_t base = 0; base < (size_t)n_ctus; base += chunk) {
    const int n = (int)std::min<size_t>(chunk, (size_t)n_ctus - base);
    p.ctu_base = (int)base;
    prof_begin(ctx, ctx->ev_conv, s);
    hipLaunchKernelGGL(hevcdl_cnn_ctu_kernel, dim3(n), dim3(256), hevcdl_cnn_smem_bytes(), s, p);
    prof_end(ctx, ctx->ev_conv, s);
    f.n_ctus = n; f.ctu_base = (int)base; f.labels = (uint8_t *)d_labels + base * 16;
    f.logits = d_logits ? (float *)d_logits + base * 64 : nullptr;
#ifdef HEVCDL_CNN_PROF
    f.logits = nullptr;              // the profiling build returns the conv kernel's phase timers through the logits buffer
#endif
    hipLaunchKernelGGL(hevcdl_fc_kernel, dim3((n + 15) / 16), dim3(256), hevcdl_fc_smem_bytes(), s, f);
  }
  prof_end(ctx, ctx->ev_cnn, s);
  HIPCHK(hipGetLastError());
  return HEVCDL_OK;
}

#ifdef HEVCDL_STAGE_TRACE
static unsigned int *g_stage_log = nullptr;
static const size_t STAGE_LOG_WORDS = (12u << 20) + 8;       // rd_kernel.hip STAGE_CAP + header
// words of the log of the last launch (header excluded), copied to dst up to cap_words; returns the number of words the kernel wanted to write
extern "C" size_t hevcdl_stage_trace_fetch(unsigned int *dst, size_t cap_words)
{
  if (!g_stage_log) return 0;
  unsigned int used = 0;
  hipDeviceSynchronize();
  hipMemcpy(&used, g_stage_log, 4, hipMemcpyDeviceToHost);
  size_t n = used; if (n > (size_t)(12u << 20)) n = (size_t)(12u << 20); if (n > cap_words) n = cap_words;
  if (n) hipMemcpy(dst, g_stage_log + 2, n * 4, hipMemcpyDeviceToHost);
  return used;
}
#endif

static hevcdl_status launch_rd(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, const void *d_labels, void *d_records, void *d_recon, void *d_stats, hipStream_t s,
                               int ctu_begin = 0, int ctu_end = -1, const void *d_cabac_in = nullptr, void *d_cabac_out = nullptr,
                               int tile_begin = 0, int tile_count = -1, int session_frame = -1)
{
  hevcdl_rd_params p;
  memset(&p, 0, sizeof p);
  p.ctu_begin = ctu_begin; p.ctu_end = ctu_end < 0 ? ctx->ctus : ctu_end; p.cabac_in = (const unsigned char *)d_cabac_in; p.cabac_out = (unsigned char *)d_cabac_out;
  p.yuv = (const uint8_t *)d_yuv; p.labels = (const uint8_t *)d_labels; p.records = (unsigned char *)d_records; p.recon = (uint8_t *)d_recon;
  p.stats = (unsigned char *)d_stats; p.scratch = ctx->d_scratch; p.scratch_per_wave = ctx->scratch_per_wave;
  p.width = ctx->cfg.width; p.height = ctx->cfg.height; p.ctus_x = ctx->ctus_x; p.ctus_y = ctx->ctus_y; p.n_frames = n_frames;
  p.tile_cols = ctx->cfg.tile_columns; p.tile_rows = ctx->cfg.tile_rows;
  memcpy(p.col_bd, ctx->col_bd, sizeof p.col_bd); memcpy(p.row_bd, ctx->row_bd, sizeof p.row_bd);
  p.tile_begin = tile_begin; p.tile_count = tile_count < 0 ? p.tile_cols * p.tile_rows : tile_count;
  // WaveFrontSynchro: a unit is one CTU row of a frame (rows of a frame run two CTUs apart on different waves, which wait for each other: a cooperative launch), or -- where
  // the grid cannot be co-resident, or the caller shares the device (HEVCDL_EXEC_NO_UNIT_HANDOVER) -- a whole frame whose rows one wave walks in order
  const bool whole_launch = !d_cabac_in && !d_cabac_out && ctu_begin == 0 && p.ctu_end == ctx->ctus && tile_begin == 0 && tile_count < 0;
  if (ctx->cfg.wavefront) {
    if (session_frame >= 0 && n_frames == 1 && tile_begin == 0 && tile_count < 0) {
      // the per-CTU session (hevcdl_compress_ctu): one CTU a launch in the one-wave form; the contexts behind the second CTU of every row stay in the session frame's part of
      // d_wpp from call to call (hevcdl_begin_frames cleared it), so a row's first CTU finds what the row above left -- whatever state the caller hands in for that CTU
      p.wpp = 2; p.tile_count = 1; p.wpp_state = ctx->d_wpp + (size_t)256 * ctx->ctus_y * session_frame; p.wpp_queue = ctx->d_wpp + ctx->wpp_state_bytes; p.wpp_ring = ctx->wpp_ring;
    } else {
      if (!whole_launch) return fail(ctx, HEVCDL_ERR_UNSUPPORTED, "WaveFrontSynchro: whole-frame launches or the per-CTU session (no tile range, no CTU range of several frames)");
      p.wpp = (ctx->cfg.exec_flags & HEVCDL_EXEC_NO_UNIT_HANDOVER) ? 2 : 1; p.wpp_state = ctx->d_wpp; p.tile_count = p.wpp == 1 ? ctx->ctus_y : 1;
      p.wpp_queue = ctx->d_wpp + ctx->wpp_state_bytes; p.wpp_ring = ctx->wpp_ring;
      HIPCHK(hipMemsetAsync(ctx->d_wpp, 0, (size_t)256 * ctx->ctus_y * n_frames, s));
      HIPCHK(hipMemsetAsync(p.wpp_queue, 0, 1024 + (size_t)4 * ctx->wpp_ring, s));
    }
  }
  if (d_stats) HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(hevcdl_frame_stats) * (size_t)n_frames, s));     // the tile waves of a frame add into its entry
  p.k.lambda = ctx->cfg.lambda; p.k.sqrt_lambda = ctx->cfg.sqrt_lambda; p.k.chroma_weight = ctx->cfg.chroma_weight; p.k.lambda_chroma = ctx->cfg.lambda_chroma;
  memcpy(p.k.err_scale, ctx->cfg.err_scale, sizeof p.k.err_scale);
  p.k.sbh_rd_factor[0] = ctx->cfg.sbh_rd_factor[0]; p.k.sbh_rd_factor[1] = ctx->cfg.sbh_rd_factor[1];
  p.k.qp = ctx->cfg.qp; p.k.qp_chroma = ctx->cfg.qp_chroma; p.k.tools = (int)ctx->cfg.tools;
#ifdef HEVCDL_STAGE_TRACE
  // stage-trace build (tests/test_rd_gpu.py, lib/libhevcdl_hip_trace.so): the kernel logs its search events here; hevcdl_stage_trace_fetch reads them
  if (!g_stage_log) HIPCHK(hipMalloc(&g_stage_log, STAGE_LOG_WORDS * 4));
  HIPCHK(hipMemsetAsync(g_stage_log, 0, 8, s)); p.dbgbuf = g_stage_log;
#endif
#if defined(HEVCDL_KERNEL_PROF) || defined(HEVCDL_KERNEL_DEBUG)
  unsigned int *d_dbg = nullptr;               // instrumented builds only (tools/phase_profile.py): in-kernel timers / traces come back through this buffer
  HIPCHK(hipMalloc(&d_dbg, 8004 * 4)); HIPCHK(hipMemset(d_dbg, 0, 8004 * 4)); p.dbgbuf = d_dbg;
#endif
  // One workgroup (hevcdl_rd_waves_per_group() wavefronts, the whole LDS) per CU; the units (frame x tile) are dealt round-robin to the
  // workgroups, a wave per unit; waves left without a unit help the others (rd_kernel.hip).
  const int n_units = n_frames * p.tile_count;
  // WaveFrontSynchro, rows on waves of their own: rows are claimed as they become startable, so what sizes the launch is not the number of rows but how many of them can be
  // under way at once -- a row starts two CTUs behind the row above, i.e. at most (ctus_x + 1) / 2 rows of a frame (+ 1: one being claimed) -- `walkers`
  const long long walkers = p.wpp == 1 ? (long long)n_frames * std::min(ctx->ctus_y, (ctx->ctus_x + 1) / 2 + 1) : (long long)n_units;
  const int groups = (int)std::min<long long>(walkers, ctx->rd_groups);
  p.master_groups = (int)std::min<long long>(walkers, 1 << 30);
  // Uneven dealing (e.g. 600 frames on 256 workgroups): the surplus units travel round the ring of workgroups so that every workgroup --
  // and every frame -- is crowded for the same share of the time (rd_kernel.hip, process_unit).  Only for whole-unit launches of a few units
  // per workgroup.
  p.sched = ctx->d_sched;
  p.migrate = (!p.wpp && n_units > groups && n_units % groups != 0 && n_units / groups <= 3 && groups <= 1024 && groups >= 8 && !d_cabac_in && !d_cabac_out &&
               ctu_begin == 0 && p.ctu_end == ctx->ctus && !(ctx->cfg.exec_flags & HEVCDL_EXEC_NO_UNIT_HANDOVER)) ? 1 : 0;
  if (p.migrate) {
    std::vector<int> init(16 + groups, 0);
    const int base = n_units / groups, extra = n_units % groups;
    for (int g = 0; g < groups; g++) { const int e = (g * extra + groups - 1) / groups; init[16 + g] = base + ((e < extra && (e * groups) / extra == g) ? 1 : 0); }    // the kernel's dealing
    HIPCHK(hipMemsetAsync(ctx->d_sched, 0, 8192 + (size_t)groups * 192, s));
    HIPCHK(hipMemcpyAsync(ctx->d_sched, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice, s));      // pageable source: the copy is staged before the call returns
  }
  // Few units (one frame, ten, a GPU's share of a sharded job): a frame is bound by the work of its CU's eight waves while most CUs have nothing to do -> the
  // kernel runs on ALL CUs and the workgroups without a unit take the second luma passes the others post (rd_kernel.hip, remote_post / remote_serve)
  p.remote = (!p.migrate && p.wpp != 2 && ctx->remote_groups && 3 * walkers <= 2 * ctx->remote_groups && !d_cabac_in && !d_cabac_out && ctu_begin == 0 && p.ctu_end == ctx->ctus) ? 1 : 0;
  // more units than takers (up to two units per taker; measured on 256 CUs: 150 frames 3.89 -> 3.79 s, 200 frames 3.92 -> 4.00 s, hence the limit): a pass is
  // posted only while a taker is free (rd_kernel.hip, remote_room)
  if (p.remote && 2 * walkers > ctx->remote_groups) p.remote = 3;
  // very few units: enough idle workgroups for the chroma modes of every master as well (eleven jobs per unit in flight).  Measured on 256 CUs, chroma jobs posted /
  // kept in the unit's own workgroup: 1 frame 2.84 / 3.10 s, 10 frames 2.95 / 3.17 s, 16 frames 3.38 / 3.18 s -- hence a twenty-second of the CUs, not a sixteenth
  if (p.remote && 22 * walkers <= ctx->remote_groups) p.remote = 2;
  // the ring of posted jobs has 2048 entries (rd_kernel.hip RQ_SIZE): a unit has at most two passes and the ten component jobs of its chroma modes posted
  if (p.remote && 12 * walkers > 2048) p.remote = 0;
  if (p.remote && p.wpp == 1) { const char *e = getenv("HEVCDL_WPP_REMOTE"); if (e && atoi(e) >= 1 && atoi(e) <= 3) p.remote = atoi(e); }      // (measurement knob)
  if (p.remote) HIPCHK(hipMemsetAsync(ctx->d_sched, 0, 4096 + 2048 * 8, s));      // finished counter, queue head / tail, the ring (2048 pointers behind byte 4096)
  // Which build of the 8-bit kernel.  Measured at 2160p on 256 CUs, eight- / ten-wave build (rd_kernel_wide.hip: 168 registers per lane instead of 256, no look-ahead
  // region): 200 frames 3.53 / 4.21 s, 300 frames 4.16 / 4.59 s, 450 frames 5.17 / 5.21 s, 600 frames 6.24 / 6.11 s, 1024 frames 10.33 / 9.50 s, 2048 frames
  // 16.08 / 15.66 s, 2560 frames 21.47 / 18.14 s -- but the ten-wave build moves three times the bytes (600 frames: 5.8 MB per CTU through the L2s against 1.85 MB:
  // register spills at 168 registers, the CU walk's snapshots in HBM).  It is chosen where it clearly pays: from three units per workgroup on (round 5; four before).  Launches in the
  // few-units form keep the eight-wave build.
  // (a context with other tool switches than the reference cfg's runs the build that reads them: rd_kernel_tools.hip, eight waves)
  const bool rt_tools = ctx->cfg.bit_depth == 8 && ctx->cfg.tools != HEVCDL_TOOLS_REFERENCE;
  bool wide = false;
  if (ctx->cfg.bit_depth == 8 && !rt_tools && !(ctx->cfg.exec_flags & HEVCDL_EXEC_RD_NARROW))
    wide = (ctx->cfg.exec_flags & HEVCDL_EXEC_RD_WIDE) || (!p.remote && walkers >= 3LL * groups);      // round 5, eight- / ten-wave build on 256 CUs: 450 frames 4.80 / 4.87 s, 600 frames 5.76 / 5.70 s, 768 frames 7.00 / 6.87 s, 1024 frames 9.44 / 8.78 s
  if (wide) p.remote = 0;      // (the ten-wave build together with the hand-over of units: launches of more than three and fewer than four units per workgroup, or exec_flags; tests/test_rd_gpu.py::test_units_handed_over_between_workgroups_give_the_same_result runs the pair)
  const int waves = ctx->cfg.bit_depth != 8 ? hevcdl_rd_waves_per_group() : (wide ? hevcdl_rd_waves_per_group_wide() : hevcdl_rd_waves_per_group());
  const int threads = 64 * waves;
  // Waves of a workgroup that claim rows (the rest help them).  A frame's rows form a chain of ctus_x + 2 (ctus_y - 1) CTU steps; a step takes a claimer ~1.5 ms with seven
  // helpers and ~8 ms without any (t(m) ~ 0.75 + 0.75 m ms at m claimers per workgroup, measured on 600-frame launches: profiles/r06d_wavefront_masters.txt), while the
  // chip's rate grows with m (216 k CTU/s at 2 ... 313 k at 10).  A launch is bound by that chain until the work per claimer exceeds it: m = 0.7 x (CTUs of the launch) /
  // (workgroups x chain), at least 1, at most every wave.  Measured at 2160p, 75 frames: m = 3 0.79 s, m = 10 0.98 s; 150 frames: flat from 6 up (1.32 - 1.34 s); 600 frames:
  // m = 10 3.91 s, m = 6 4.54 s, m = 3 5.06 s.
  if (p.wpp == 1) { const long long chain = ctx->ctus_x + 2LL * (ctx->ctus_y - 1);
                    p.wpp_masters = p.remote ? 1 : (int)std::max<long long>(1, std::min<long long>(waves, (7LL * n_frames * ctx->ctus) / (10LL * groups * chain))); }
  if (p.wpp == 1 && !p.remote) { const char *e = getenv("HEVCDL_WPP_MASTERS"); if (e && atoi(e) > 0) p.wpp_masters = std::min(waves, atoi(e)); }      // (measurement knob: tools/time_rd.py sweeps)
  const void *kern = ctx->cfg.bit_depth == 8 ? (wide ? (const void *)hevcdl_rd_frame_kernel_wide : (rt_tools ? (const void *)hevcdl_rd_frame_kernel_tools : (const void *)hevcdl_rd_frame_kernel)) : (const void *)hevcdl_rd_frame_kernel_bd10;
  const size_t smem = ctx->cfg.bit_depth == 8 ? (wide ? hevcdl_rd_smem_bytes_wide() : (rt_tools ? hevcdl_rd_smem_bytes_tools() : hevcdl_rd_smem_bytes())) : hevcdl_rd_smem_bytes_bd10();
  { // the workspace: one block per wave of every workgroup of this launch
    const size_t need = ctx->scratch_per_wave * (size_t)(p.remote ? ctx->remote_groups : groups) * (size_t)waves;
    if (need > ctx->scratch_bytes) {
      if (ctx->d_scratch) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipFree(ctx->d_scratch)); ctx->d_scratch = nullptr; ctx->scratch_bytes = 0; }
      { const hipError_t e_ = hipMalloc(&ctx->d_scratch, need);
        if (e_ != hipSuccess) { (void)hipGetLastError(); ctx->d_scratch = nullptr; return fail(ctx, HEVCDL_ERR_OOM, "decision-kernel workspace (1.6 MB per wave of the launch)", e_); } }
      ctx->scratch_bytes = need;
    }
    p.scratch = ctx->d_scratch;
  }
  prof_begin(ctx, ctx->ev_rd, s);
  bool launched = false;
  if (p.migrate || p.remote || p.wpp == 1) { // workgroups that wait for each other: a cooperative launch, which the runtime only accepts when the whole grid can be resident at once
    void *args[] = { &p };
    if (hipLaunchCooperativeKernel(kern, dim3(p.remote ? ctx->remote_groups : groups), dim3(threads), args, smem, s) == hipSuccess) launched = true;
    else {
      (void)hipGetLastError(); p.migrate = 0; p.remote = 0;
      if (p.wpp == 1) { prof_end(ctx, ctx->ev_rd, s); ctx->cfg.exec_flags |= HEVCDL_EXEC_NO_UNIT_HANDOVER;      // rows on waves of their own need co-residency: this context walks a frame's rows on one wave from now on
                        return launch_rd(ctx, d_yuv, n_frames, d_labels, d_records, d_recon, d_stats, s, ctu_begin, ctu_end, d_cabac_in, d_cabac_out, tile_begin, tile_count); }
    }
  }
  if (!launched) {
    if (ctx->cfg.bit_depth != 8) hipLaunchKernelGGL(hevcdl_rd_frame_kernel_bd10, dim3(groups), dim3(threads), smem, s, p);
    else if (wide) hipLaunchKernelGGL(hevcdl_rd_frame_kernel_wide, dim3(groups), dim3(threads), smem, s, p);
    else if (rt_tools) hipLaunchKernelGGL(hevcdl_rd_frame_kernel_tools, dim3(groups), dim3(threads), smem, s, p);
    else hipLaunchKernelGGL(hevcdl_rd_frame_kernel, dim3(groups), dim3(threads), smem, s, p);
  }
  prof_end(ctx, ctx->ev_rd, s);
  HIPCHK(hipGetLastError());
  snprintf(ctx->last_rd, sizeof ctx->last_rd, "%s form=%s workgroups=%d waves=%d units=%d", ctx->cfg.bit_depth != 8 ? "hevcdl_rd_frame_kernel_bd10" : (wide ? "hevcdl_rd_frame_kernel_wide" : (rt_tools ? "hevcdl_rd_frame_kernel_tools" : "hevcdl_rd_frame_kernel")),
           p.wpp == 1 ? (p.remote ? "wavefront-rows+few-units" : "wavefront-rows") : p.wpp == 2 ? "wavefront(one wave per frame)" : p.migrate ? "unit-handover" : (p.remote == 1 ? "few-units(passes)" : (p.remote == 2 ? "few-units(passes+chroma)" : (p.remote == 3 ? "few-units(passes while takers idle)" : "independent"))),
           p.remote ? ctx->remote_groups : groups, waves, n_units);
#if defined(HEVCDL_KERNEL_PROF) || defined(HEVCDL_KERNEL_DEBUG)
  {
    std::vector<unsigned int> hb(8004); hipDeviceSynchronize(); hipMemcpy(hb.data(), d_dbg, 8004 * 4, hipMemcpyDeviceToHost);
#ifdef HEVCDL_KERNEL_PROF
    for (unsigned i = 0; i < hb[0] && i < 64; i++) { // accumulators of the kernel's timers: cycles in the low 40 bits, calls above -> kilocycles, calls
      const unsigned long long v = (unsigned long long)hb[2 + 2 * i] | ((unsigned long long)hb[3 + 2 * i] << 32);
      printf("DBGV %u %u\n", (unsigned)((v & 0xffffffffffull) >> 10), (unsigned)(v >> 40));
    }
#else
    for (unsigned i = 0; i < hb[0] && i < 3990; i++) printf("DBGV %u %u\n", hb[1 + 2 * i], hb[2 + 2 * i]);
#endif
    fflush(stdout); hipFree(d_dbg);
  }
#endif
  return HEVCDL_OK;
}

static hevcdl_status check_frames(hevcdl_ctx *ctx, int n_frames)
{
  if (!ctx) return HEVCDL_ERR_INVALID_ARG;
  if (n_frames < 0 || n_frames > ctx->cfg.max_frames) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "n_frames out of range (cfg.max_frames)");
  hipError_t e = hipSetDevice(ctx->cfg.device);
  if (e != hipSuccess) return fail(ctx, HEVCDL_ERR_HIP, "hipSetDevice", e);
  return HEVCDL_OK;
}

// caller-supplied labels (device copy): boundary clamp + quadtree consistency in place; labels above 3 -> HEVCDL_ERR_INVALID_ARG
extern "C" hevcdl_status hevcdl_clamp_labels_dev(hevcdl_ctx *ctx, void *d_labels, int n_frames, void *stream)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!d_labels) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null device pointer");
  const int n = n_frames * ctx->ctus;
  HIPCHK(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), (hipStream_t)stream));
  hipLaunchKernelGGL(hevcdl_clamp_labels_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint8_t *)d_labels, n, ctx->ctus, ctx->ctus_x, ctx->cfg.width, ctx->cfg.height, ctx->d_flag);
  int bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  if (bad) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "labels: depth above 3");
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_predict_depth_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, void *d_labels, void *d_logits_opt, void *stream)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!d_yuv || !d_labels) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null device pointer");
  if (ctx->cfg.bit_depth > 8) { // the CNN stage is defined on 8-bit pictures: the top 8 bits of every sample
    if (!ctx->d_yuv8) HIPCHK(hipMalloc(&ctx->d_yuv8, hevcdl_frame_bytes(ctx->cfg.width, ctx->cfg.height) * (size_t)ctx->cfg.max_frames));
    const size_t n = hevcdl_frame_bytes(ctx->cfg.width, ctx->cfg.height) * (size_t)n_frames;
    hipLaunchKernelGGL(hevcdl_narrow_samples_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)d_yuv, ctx->d_yuv8, n, ctx->cfg.bit_depth - 8);
    d_yuv = ctx->d_yuv8;
  }
  return launch_cnn(ctx, d_yuv, ctx->cfg.cnn_input == HEVCDL_CNN_INPUT_LUMA ? HEVCDL_DEV_INPUT_LUMA : HEVCDL_DEV_INPUT_RGB601,
                    n_frames * ctx->ctus, 1, d_labels, d_logits_opt, (hipStream_t)stream);
}

extern "C" hevcdl_status hevcdl_compress_frames_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, const void *d_labels,
                                                    void *d_records, void *d_recon, void *d_stats, void *stream)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!d_yuv || !d_labels || !d_records || !d_recon) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null device pointer");
  return launch_rd(ctx, d_yuv, n_frames, d_labels, d_records, d_recon, d_stats, (hipStream_t)stream);
}

extern "C" hevcdl_status hevcdl_compress_tiles_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, const void *d_labels, void *d_records, void *d_recon, void *d_stats,
                                                   int tile_begin, int tile_count, void *stream)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (tile_begin < 0 || tile_count < 0 || tile_begin + tile_count > ctx->cfg.tile_columns * ctx->cfg.tile_rows) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "tile range outside the tile grid");
  if (n_frames == 0 || tile_count == 0) return HEVCDL_OK;
  if (!d_yuv || !d_labels || !d_records || !d_recon) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null device pointer");
  return launch_rd(ctx, d_yuv, n_frames, d_labels, d_records, d_recon, d_stats, (hipStream_t)stream, 0, -1, nullptr, nullptr, tile_begin, tile_count);
}

extern "C" hevcdl_status hevcdl_encode_frames_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, void *d_labels,
                                                  void *d_records, void *d_recon, void *d_stats, void *stream)
{
  hevcdl_status st = hevcdl_predict_depth_dev(ctx, d_yuv, n_frames, d_labels, nullptr, stream); if (st) return st;
  return hevcdl_compress_frames_dev(ctx, d_yuv, n_frames, d_labels, d_records, d_recon, d_stats, stream);
}

// hevcdl_planes -> the packed planar frames of the staging buffer (row by row: hipMemcpy2D; 8-bit samples in 16-bit containers are narrowed on the device)
static hevcdl_status upload_planes(hevcdl_ctx *ctx, const hevcdl_planes *src, int n_frames)
{
  if (!src || !src->plane[0] || !src->plane[1] || !src->plane[2]) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null plane pointer");
  const int native = ctx->cfg.bit_depth > 8 ? 2 : 1, sb = src->sample_bytes;
  if (sb != native && !(sb == 2 && native == 1)) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "sample_bytes does not fit the context's bit depth");
  const int w = ctx->cfg.width, h = ctx->cfg.height;
  for (int c = 0; c < 3; c++) if (src->row_stride[c] < (size_t)(c ? w / 2 : w) * sb) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "row stride smaller than a row");
  unsigned char *dst = (unsigned char *)ctx->d_yuv;
  if (sb != native) {
    if (!ctx->d_wide) HIPCHK(hipMalloc(&ctx->d_wide, 2 * ctx->frame_bytes * (size_t)ctx->cfg.max_frames));
    dst = ctx->d_wide;
  }
  const size_t fb = ctx->frame_bytes / native * sb, off[3] = { 0, (size_t)w * h * sb, (size_t)w * h * sb + (size_t)(w / 2) * (h / 2) * sb };
  for (int f = 0; f < n_frames; f++) for (int c = 0; c < 3; c++) {
    const size_t rw = (size_t)(c ? w / 2 : w) * sb; const int rows = c ? h / 2 : h;
    HIPCHK(hipMemcpy2D(dst + (size_t)f * fb + off[c], rw, (const unsigned char *)src->plane[c] + (size_t)f * src->frame_stride[c], src->row_stride[c], rw, rows, hipMemcpyHostToDevice));
  }
  if (sb != native) {
    hipLaunchKernelGGL(hevcdl_narrow_samples_kernel, dim3(2048), dim3(256), 0, (hipStream_t)nullptr, (const uint16_t *)ctx->d_wide, (uint8_t *)ctx->d_yuv, ctx->frame_bytes * (size_t)n_frames, 0);
    HIPCHK(hipGetLastError());
  }
  return HEVCDL_OK;
}

static hevcdl_status predict_depth_staged(hevcdl_ctx *ctx, int n_frames, uint8_t *labels, float *logits_opt);
extern "C" hevcdl_status hevcdl_predict_depth(hevcdl_ctx *ctx, const uint8_t *yuv, int n_frames, uint8_t *labels, float *logits_opt)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!yuv || !labels) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  st = ensure_staging(ctx); if (st) return st;
  HIPCHK(hipMemcpy(ctx->d_yuv, yuv, ctx->frame_bytes * n_frames, hipMemcpyHostToDevice));
  return predict_depth_staged(ctx, n_frames, labels, logits_opt);
}
extern "C" hevcdl_status hevcdl_predict_depth_planes(hevcdl_ctx *ctx, const hevcdl_planes *src, int n_frames, uint8_t *labels, float *logits_opt)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!labels) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  st = ensure_staging(ctx); if (st) return st;
  st = upload_planes(ctx, src, n_frames); if (st) return st;
  return predict_depth_staged(ctx, n_frames, labels, logits_opt);
}
static hevcdl_status predict_depth_staged(hevcdl_ctx *ctx, int n_frames, uint8_t *labels, float *logits_opt)
{
  hevcdl_status st;
  st = hevcdl_predict_depth_dev(ctx, ctx->d_yuv, n_frames, ctx->d_labels, logits_opt ? ctx->d_logits : nullptr, nullptr); if (st) return st;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(labels, ctx->d_labels, (size_t)ctx->ctus * 16 * n_frames, hipMemcpyDeviceToHost));
  if (logits_opt) HIPCHK(hipMemcpy(logits_opt, ctx->d_logits, (size_t)ctx->ctus * 64 * sizeof(float) * n_frames, hipMemcpyDeviceToHost));
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_predict_depth_rgb(hevcdl_ctx *ctx, const uint8_t *ctu_rgb, int n_ctus, uint8_t *labels, float *logits_opt)
{
  if (!ctx) return HEVCDL_ERR_INVALID_ARG;
  if (n_ctus < 0) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "n_ctus < 0");
  if (n_ctus == 0) return HEVCDL_OK;
  if (!ctu_rgb || !labels) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  HIPCHK(hipSetDevice(ctx->cfg.device));
  uint8_t *d_in = nullptr, *d_lab = nullptr; float *d_lg = nullptr;
  hipError_t e0 = hipMalloc(&d_in, (size_t)n_ctus * 12288);
  if (e0 == hipSuccess) e0 = hipMalloc(&d_lab, (size_t)n_ctus * 16);
  if (e0 == hipSuccess) e0 = hipMalloc(&d_lg, (size_t)n_ctus * 64 * sizeof(float));
  if (e0 == hipSuccess) e0 = hipMemcpy(d_in, ctu_rgb, (size_t)n_ctus * 12288, hipMemcpyHostToDevice);
  if (e0 != hipSuccess) { hipFree(d_in); hipFree(d_lab); hipFree(d_lg); return fail(ctx, e0 == hipErrorOutOfMemory ? HEVCDL_ERR_OOM : HEVCDL_ERR_HIP, "predict_depth_rgb buffers", e0); }
  hevcdl_status st = launch_cnn(ctx, d_in, HEVCDL_DEV_INPUT_RGB_CTU, n_ctus, 0, d_lab, d_lg, nullptr);
  if (st == HEVCDL_OK) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) st = fail(ctx, HEVCDL_ERR_HIP, "cnn kernel", e);
  }
  if (st == HEVCDL_OK) {
    hipMemcpy(labels, d_lab, (size_t)n_ctus * 16, hipMemcpyDeviceToHost);
    if (logits_opt) hipMemcpy(logits_opt, d_lg, (size_t)n_ctus * 64 * sizeof(float), hipMemcpyDeviceToHost);
  }
  hipFree(d_in); hipFree(d_lab); hipFree(d_lg);
  return st;
}

extern "C" const char *hevcdl_last_rd_launch(const hevcdl_ctx *ctx) { return ctx ? ctx->last_rd : ""; }

// The decision kernel's workspace (1.6 MB per wave of a launch) is allocated by the first launch that needs it and grown when a later one needs more: a context that
// only ever codes a frame in the independent form holds megabytes, a launch on every CU 3.4 GB (4.3 GB for the ten-wave build).  A caller that wants to learn
// about a lack of device memory BEFORE its first pictures are in flight reserves the largest workspace any launch of this context can ask for -- launches of 1 ..
// max_frames frames: every CU's workgroup (the few-units form runs on all of them), ten waves where the ten-wave build can be chosen -- and gets HEVCDL_ERR_OOM here.
extern "C" hevcdl_status hevcdl_reserve_workspace(hevcdl_ctx *ctx)
{
  if (!ctx) return HEVCDL_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(ctx->cfg.device));
  const long long max_units = (long long)ctx->cfg.max_frames * (ctx->cfg.wavefront ? ctx->ctus_y : ctx->cfg.tile_columns * ctx->cfg.tile_rows);
  const int groups = std::max(ctx->rd_groups, ctx->remote_groups);
  int waves = ctx->cfg.bit_depth != 8 ? hevcdl_rd_waves_per_group() : hevcdl_rd_waves_per_group();
  const bool rt_tools = ctx->cfg.bit_depth == 8 && ctx->cfg.tools != HEVCDL_TOOLS_REFERENCE;      // launch_rd's rule: such a context never runs the ten-wave build
  if (ctx->cfg.bit_depth == 8 && !rt_tools && !(ctx->cfg.exec_flags & HEVCDL_EXEC_RD_NARROW) && ((ctx->cfg.exec_flags & HEVCDL_EXEC_RD_WIDE) || max_units >= 3LL * ctx->rd_groups))
    waves = std::max(waves, hevcdl_rd_waves_per_group_wide());
  const size_t need = ctx->scratch_per_wave * (size_t)groups * (size_t)waves;
  if (need <= ctx->scratch_bytes) return HEVCDL_OK;
  HIPCHK(hipDeviceSynchronize());
  if (ctx->d_scratch) { hipFree(ctx->d_scratch); ctx->d_scratch = nullptr; ctx->scratch_bytes = 0; }
  const hipError_t e = hipMalloc(&ctx->d_scratch, need);
  if (e != hipSuccess) { (void)hipGetLastError(); ctx->d_scratch = nullptr; return fail(ctx, HEVCDL_ERR_OOM, "decision-kernel workspace (1.6 MB per wave of the largest launch)", e); }
  ctx->scratch_bytes = need;
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_labels_from_logits(hevcdl_ctx *ctx, const float *logits, int n_ctus, int clamp, uint8_t *labels)
{
  if (!ctx) return HEVCDL_ERR_INVALID_ARG;
  if (n_ctus < 0) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "n_ctus < 0");
  if (n_ctus == 0) return HEVCDL_OK;
  if (!logits || !labels) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  HIPCHK(hipSetDevice(ctx->cfg.device));
  uint8_t *d_lab = nullptr; float *d_lg = nullptr;
  hipError_t e0 = hipMalloc(&d_lab, (size_t)n_ctus * 16);
  if (e0 == hipSuccess) e0 = hipMalloc(&d_lg, (size_t)n_ctus * 64 * sizeof(float));
  if (e0 == hipSuccess) e0 = hipMemcpy(d_lg, logits, (size_t)n_ctus * 64 * sizeof(float), hipMemcpyHostToDevice);
  if (e0 != hipSuccess) { hipFree(d_lab); hipFree(d_lg); return fail(ctx, e0 == hipErrorOutOfMemory ? HEVCDL_ERR_OOM : HEVCDL_ERR_HIP, "labels_from_logits buffers", e0); }
  hevcdl_fc_params f;
  f.a3 = nullptr; f.weights = ctx->d_weights; f.labels = d_lab; f.logits = nullptr; f.logits_in = d_lg;
  f.n_ctus = n_ctus; f.ctu_base = 0; f.width = ctx->cfg.width; f.height = ctx->cfg.height; f.ctus_x = ctx->ctus_x; f.ctus_per_frame = ctx->ctus; f.clamp = clamp ? 1 : 0;
  hipLaunchKernelGGL(hevcdl_fc_kernel, dim3((n_ctus + 15) / 16), dim3(256), hevcdl_fc_smem_bytes(), nullptr, f);
  hipError_t e = hipDeviceSynchronize();
  hevcdl_status st = e == hipSuccess ? HEVCDL_OK : fail(ctx, HEVCDL_ERR_HIP, "label kernel", e);
  if (st == HEVCDL_OK) hipMemcpy(labels, d_lab, (size_t)n_ctus * 16, hipMemcpyDeviceToHost);
  hipFree(d_lab); hipFree(d_lg);
  return st;
}

static hevcdl_status compress_frames_staged(hevcdl_ctx *ctx, int n_frames, const uint8_t *labels_opt, hevcdl_ctu_record *records, uint8_t *recon_opt, hevcdl_frame_stats *stats_opt);
extern "C" hevcdl_status hevcdl_compress_frames(hevcdl_ctx *ctx, const uint8_t *yuv, int n_frames, const uint8_t *labels_opt,
                                                hevcdl_ctu_record *records, uint8_t *recon_opt, hevcdl_frame_stats *stats_opt)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!yuv || !records) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  st = ensure_staging(ctx); if (st) return st;
  HIPCHK(hipMemcpy(ctx->d_yuv, yuv, ctx->frame_bytes * n_frames, hipMemcpyHostToDevice));
  return compress_frames_staged(ctx, n_frames, labels_opt, records, recon_opt, stats_opt);
}
extern "C" hevcdl_status hevcdl_compress_frames_planes(hevcdl_ctx *ctx, const hevcdl_planes *src, int n_frames, const uint8_t *labels_opt,
                                                       hevcdl_ctu_record *records, uint8_t *recon_opt, hevcdl_frame_stats *stats_opt)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!records) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  st = ensure_staging(ctx); if (st) return st;
  st = upload_planes(ctx, src, n_frames); if (st) return st;
  return compress_frames_staged(ctx, n_frames, labels_opt, records, recon_opt, stats_opt);
}
static hevcdl_status compress_frames_staged(hevcdl_ctx *ctx, int n_frames, const uint8_t *labels_opt, hevcdl_ctu_record *records, uint8_t *recon_opt, hevcdl_frame_stats *stats_opt)
{
  hevcdl_status st;
  if (labels_opt) { HIPCHK(hipMemcpy(ctx->d_labels, labels_opt, (size_t)ctx->ctus * 16 * n_frames, hipMemcpyHostToDevice)); st = hevcdl_clamp_labels_dev(ctx, ctx->d_labels, n_frames, nullptr); if (st) return st; }
  else { st = hevcdl_predict_depth_dev(ctx, ctx->d_yuv, n_frames, ctx->d_labels, nullptr, nullptr); if (st) return st; }
  st = hevcdl_compress_frames_dev(ctx, ctx->d_yuv, n_frames, ctx->d_labels, ctx->d_records, ctx->d_recon, ctx->d_stats, nullptr); if (st) return st;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(ctx, HEVCDL_ERR_HIP, "rd kernel", e);
  HIPCHK(hipMemcpy(records, ctx->d_records, (size_t)ctx->ctus * sizeof(hevcdl_ctu_record) * n_frames, hipMemcpyDeviceToHost));
  if (recon_opt) HIPCHK(hipMemcpy(recon_opt, ctx->d_recon, ctx->frame_bytes * n_frames, hipMemcpyDeviceToHost));
  if (stats_opt) HIPCHK(hipMemcpy(stats_opt, ctx->d_stats, sizeof(hevcdl_frame_stats) * n_frames, hipMemcpyDeviceToHost));
  return HEVCDL_OK;
}

// ---- deblocking (row f-2, first half): TComLoopFilter::loopFilterPic, TComLoopFilter.cpp:130, called at TEncGOP.cpp:1742 ----
static const unsigned char DBK_TC[54] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24 };   // :59-62
static const unsigned char DBK_BETA[52] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };   // :64-67

extern "C" hevcdl_status hevcdl_deblock_frames_dev(hevcdl_ctx *ctx, const void *d_recon, int n_frames, const void *d_records, void *d_out, void *stream)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!d_recon || !d_records || !d_out) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null device pointer");
  hevcdl_dbk_params p;
  p.in = (const uint8_t *)d_recon; p.out = (uint8_t *)d_out; p.records = (const unsigned char *)d_records;
  p.width = ctx->cfg.width; p.height = ctx->cfg.height; p.ctus_x = ctx->ctus_x; p.ctus_per_frame = ctx->ctus; p.n_frames = n_frames;
  const int qp = ctx->cfg.qp, qpc = CHROMA_SCALE_420[qp < 0 ? 0 : (qp > 57 ? 57 : qp)];      // TComLoopFilter.cpp:782-797, cQpOffset 0
  const int bd_scale = 1 << (ctx->cfg.bit_depth - 8);                                       // iBitdepthScale, TComLoopFilter.cpp:596, 770
  const int tco = 2 * ctx->cfg.lf_tc_offset_div2, bo = 2 * ctx->cfg.lf_beta_offset_div2;   // slice_tc_offset_div2 / slice_beta_offset_div2 << 1 (TComLoopFilter.cpp:623-624, 804), Bs 2
  auto clipi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
  p.tc = DBK_TC[clipi(qp + 2 + tco, 53)] * bd_scale; p.beta = DBK_BETA[clipi(qp + bo, 51)] * bd_scale; p.tc_c = DBK_TC[clipi(qpc + 2 + tco, 53)] * bd_scale;
  p.pel_max = (1 << ctx->cfg.bit_depth) - 1;
  p.lf_across_tiles = ctx->cfg.lf_across_tiles != 0; p.tile_cols = ctx->cfg.tile_columns; p.tile_rows = ctx->cfg.tile_rows;
  memcpy(p.col_bd, ctx->col_bd, sizeof p.col_bd); memcpy(p.row_bd, ctx->row_bd, sizeof p.row_bd);
  hevcdl_launch_deblock(&p, stream);
  HIPCHK(hipGetLastError());
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_deblock_frames(hevcdl_ctx *ctx, const uint8_t *recon, int n_frames, const hevcdl_ctu_record *records, uint8_t *out)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!recon || !records || !out) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  st = ensure_staging(ctx); if (st) return st;
  HIPCHK(hipMemcpy(ctx->d_recon, recon, ctx->frame_bytes * n_frames, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_records, records, (size_t)ctx->ctus * sizeof(hevcdl_ctu_record) * n_frames, hipMemcpyHostToDevice));
  st = hevcdl_deblock_frames_dev(ctx, ctx->d_recon, n_frames, ctx->d_records, ctx->d_yuv, nullptr); if (st) return st;   // d_yuv: free staging plane set
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(ctx, HEVCDL_ERR_HIP, "deblock kernels", e);
  HIPCHK(hipMemcpy(out, ctx->d_yuv, ctx->frame_bytes * n_frames, hipMemcpyDeviceToHost));
  return HEVCDL_OK;
}

// ---- SAO (row f-2, second half): TEncSampleAdaptiveOffset::SAOProcess, TEncSampleAdaptiveOffset.cpp:244, called at TEncGOP.cpp:1797 ----
extern "C" hevcdl_status hevcdl_sao_frames_dev(hevcdl_ctx *ctx, const void *d_org, const void *d_deblocked, int n_frames, void *d_params, void *d_out, void *stream)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!d_org || !d_deblocked || !d_params || !d_out) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null device pointer");
  const size_t nf = (size_t)ctx->cfg.max_frames;
  if (!ctx->d_sao_stats) HIPCHK(hipMalloc(&ctx->d_sao_stats, (size_t)ctx->ctus * 3 * 5 * 256 * nf));
  if (!ctx->d_sao_recon) HIPCHK(hipMalloc(&ctx->d_sao_recon, (size_t)ctx->ctus * sizeof(hevcdl_sao_blk) * nf));
  if (!ctx->d_sao_cand) HIPCHK(hipMalloc(&ctx->d_sao_cand, (size_t)ctx->ctus * (3 * 5 * 16 + 32) * nf));      // candidates, then the chain's records
  hevcdl_sao_params p;
  p.org = (const uint8_t *)d_org; p.deblocked = (const uint8_t *)d_deblocked; p.out = (uint8_t *)d_out;
  p.stats = ctx->d_sao_stats; p.params = (unsigned char *)d_params; p.recon_params = ctx->d_sao_recon; p.cand = ctx->d_sao_cand; p.crec = ctx->d_sao_cand + (size_t)ctx->ctus * 3 * 5 * 16 * nf;
  p.width = ctx->cfg.width; p.height = ctx->cfg.height; p.ctus_x = ctx->ctus_x; p.ctus_per_frame = ctx->ctus; p.n_frames = n_frames; p.qp = ctx->cfg.qp;
  p.lambda = ctx->cfg.lambda; p.lambda_chroma = ctx->cfg.lambda_chroma;          // slice lambdas per component, TEncSlice.cpp:112-140
  p.tile_cols = ctx->cfg.tile_columns; p.tile_rows = ctx->cfg.tile_rows; p.bit_depth = ctx->cfg.bit_depth; p.lf_across_tiles = ctx->cfg.lf_across_tiles != 0;
  memcpy(p.col_bd, ctx->col_bd, sizeof p.col_bd); memcpy(p.row_bd, ctx->row_bd, sizeof p.row_bd);
  hevcdl_launch_sao(&p, stream);
  HIPCHK(hipGetLastError());
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_sao_frames(hevcdl_ctx *ctx, const uint8_t *org, const uint8_t *deblocked, int n_frames, hevcdl_sao_blk *params, uint8_t *out)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!org || !deblocked || !params || !out) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  st = ensure_staging(ctx); if (st) return st;
  if (!ctx->d_sao_params) HIPCHK(hipMalloc(&ctx->d_sao_params, (size_t)ctx->ctus * sizeof(hevcdl_sao_blk) * ctx->cfg.max_frames));
  HIPCHK(hipMemcpy(ctx->d_yuv, org, ctx->frame_bytes * n_frames, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_recon, deblocked, ctx->frame_bytes * n_frames, hipMemcpyHostToDevice));
  // the records staging area is free here and large enough for one more picture set (15120 B per CTU >= 6144)
  st = hevcdl_sao_frames_dev(ctx, ctx->d_yuv, ctx->d_recon, n_frames, ctx->d_sao_params, ctx->d_records, nullptr); if (st) return st;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(ctx, HEVCDL_ERR_HIP, "sao kernels", e);
  HIPCHK(hipMemcpy(params, ctx->d_sao_params, (size_t)ctx->ctus * sizeof(hevcdl_sao_blk) * n_frames, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(out, ctx->d_records, ctx->frame_bytes * n_frames, hipMemcpyDeviceToHost));
  return HEVCDL_OK;
}

// ---- whole picture pipeline for host buffers: the stages of TEncGOP::compressGOP between reading a picture and writing its NAL units, with
// the picture staying in HBM in between (one upload of the originals, one download of records / final picture / SAO parameters) ----------
// device side of hevcdl_encode_pictures*: upload, labels, decisions, in-loop filters; the results stay in HBM (*d_final: the output pictures)
static hevcdl_status encode_pictures_device(hevcdl_ctx *ctx, const void *yuv, int n_frames, const uint8_t *labels_opt, int deblock, int want_sao, uint8_t **d_final)
{
  hevcdl_status st = ensure_staging(ctx); if (st) return st;
  if (want_sao && !ctx->d_sao_params) HIPCHK(hipMalloc(&ctx->d_sao_params, (size_t)ctx->ctus * sizeof(hevcdl_sao_blk) * ctx->cfg.max_frames));
  if (want_sao && !ctx->d_picture) HIPCHK(hipMalloc(&ctx->d_picture, ctx->frame_bytes * (size_t)ctx->cfg.max_frames));
  HIPCHK(hipMemcpy(ctx->d_yuv, yuv, ctx->frame_bytes * n_frames, hipMemcpyHostToDevice));
  if (labels_opt) { HIPCHK(hipMemcpy(ctx->d_labels, labels_opt, (size_t)ctx->ctus * 16 * n_frames, hipMemcpyHostToDevice)); st = hevcdl_clamp_labels_dev(ctx, ctx->d_labels, n_frames, nullptr); if (st) return st; }
  else { st = hevcdl_predict_depth_dev(ctx, ctx->d_yuv, n_frames, ctx->d_labels, nullptr, nullptr); if (st) return st; }
  st = hevcdl_compress_frames_dev(ctx, ctx->d_yuv, n_frames, ctx->d_labels, ctx->d_records, ctx->d_recon, ctx->d_stats, nullptr); if (st) return st;
  *d_final = ctx->d_recon;
  if (deblock) { st = hevcdl_deblock_frames_dev(ctx, ctx->d_recon, n_frames, ctx->d_records, ctx->d_recon, nullptr); if (st) return st; }      // in place
  if (want_sao) { st = hevcdl_sao_frames_dev(ctx, ctx->d_yuv, ctx->d_recon, n_frames, ctx->d_sao_params, ctx->d_picture, nullptr); if (st) return st; *d_final = ctx->d_picture; }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(ctx, HEVCDL_ERR_HIP, "picture pipeline", e);
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_encode_pictures(hevcdl_ctx *ctx, const void *yuv, int n_frames, const uint8_t *labels_opt, int deblock, hevcdl_ctu_record *records,
                                                void *picture_out, hevcdl_sao_blk *sao_opt, hevcdl_frame_stats *stats_opt)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!yuv || !records || !picture_out) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  uint8_t *d_final = nullptr;
  st = encode_pictures_device(ctx, yuv, n_frames, labels_opt, deblock, sao_opt != nullptr, &d_final); if (st) return st;
  HIPCHK(hipMemcpy(records, ctx->d_records, (size_t)ctx->ctus * sizeof(hevcdl_ctu_record) * n_frames, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(picture_out, d_final, ctx->frame_bytes * n_frames, hipMemcpyDeviceToHost));
  if (sao_opt) HIPCHK(hipMemcpy(sao_opt, ctx->d_sao_params, (size_t)ctx->ctus * sizeof(hevcdl_sao_blk) * n_frames, hipMemcpyDeviceToHost));
  if (stats_opt) HIPCHK(hipMemcpy(stats_opt, ctx->d_stats, sizeof(hevcdl_frame_stats) * n_frames, hipMemcpyDeviceToHost));
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_encode_pictures_chunked(hevcdl_ctx *ctx, const void *yuv, int n_frames, const uint8_t *labels_opt, int deblock, int want_sao,
                                                        int chunk_frames, hevcdl_chunk_fn fn, void *user)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0) return HEVCDL_OK;
  if (!yuv || !fn) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "null pointer");
  const int chunk = std::min(n_frames, chunk_frames > 0 ? chunk_frames : 64);
  const size_t rec_b = (size_t)ctx->ctus * sizeof(hevcdl_ctu_record), sao_b = (size_t)ctx->ctus * sizeof(hevcdl_sao_blk), pic_b = ctx->frame_bytes, stat_b = sizeof(hevcdl_frame_stats);
  auto up64 = [](size_t v) { return (v + 63) & ~(size_t)63; };      // every section of a chunk buffer starts on a 64-byte boundary (the callback gets pointers to structs with 64-bit members)
  const size_t o_pic = up64(rec_b * chunk), o_sao = up64(o_pic + pic_b * chunk), o_stat = up64(o_sao + sao_b * chunk), need = o_stat + stat_b * chunk;      // layout of a chunk buffer
  if (ctx->h_chunk_bytes < need) {
    for (int i = 0; i < 2; i++) { if (ctx->h_chunk[i]) hipHostFree(ctx->h_chunk[i]); ctx->h_chunk[i] = nullptr; }
    ctx->h_chunk_bytes = 0;
    for (int i = 0; i < 2; i++) HIPCHK(hipHostMalloc((void **)&ctx->h_chunk[i], need, hipHostMallocDefault));
    ctx->h_chunk_bytes = need;
  }
  if (!ctx->copy_stream) { HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)); for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&ctx->copy_ev[i], hipEventDisableTiming)); }
  uint8_t *d_final = nullptr;
  st = encode_pictures_device(ctx, yuv, n_frames, labels_opt, deblock, want_sao, &d_final); if (st) return st;
  auto fetch = [&](int ci) -> hipError_t { // chunk ci -> buffer ci & 1, asynchronously
    const int first = ci * chunk, cnt = std::min(chunk, n_frames - first);
    unsigned char *h = ctx->h_chunk[ci & 1];
    hipError_t e = hipMemcpyAsync(h, ctx->d_records + rec_b * first, rec_b * cnt, hipMemcpyDeviceToHost, ctx->copy_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h + o_pic, d_final + pic_b * first, pic_b * cnt, hipMemcpyDeviceToHost, ctx->copy_stream);
    if (e == hipSuccess && want_sao) e = hipMemcpyAsync(h + o_sao, ctx->d_sao_params + sao_b * first, sao_b * cnt, hipMemcpyDeviceToHost, ctx->copy_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h + o_stat, ctx->d_stats + stat_b * first, stat_b * cnt, hipMemcpyDeviceToHost, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->copy_ev[ci & 1], ctx->copy_stream);
    return e;
  };
  const int n_chunks = (n_frames + chunk - 1) / chunk;
  HIPCHK(fetch(0));
  for (int ci = 0; ci < n_chunks; ci++) {
    { hipError_t e_ = hipEventSynchronize(ctx->copy_ev[ci & 1]);
      if (e_ == hipSuccess && ci + 1 < n_chunks) e_ = fetch(ci + 1);                   // into the other buffer, behind the caller's work on this one
      if (e_ != hipSuccess) { (void)hipStreamSynchronize(ctx->copy_stream); return fail(ctx, HEVCDL_ERR_HIP, "chunk copy", e_); } }     // no copy into the library's buffers is left in flight
    const int first = ci * chunk, cnt = std::min(chunk, n_frames - first);
    unsigned char *h = ctx->h_chunk[ci & 1];
    if (fn(user, first, cnt, (const hevcdl_ctu_record *)h, h + o_pic, want_sao ? (const hevcdl_sao_blk *)(h + o_sao) : nullptr, (const hevcdl_frame_stats *)(h + o_stat)) != 0) {
      hipStreamSynchronize(ctx->copy_stream);
      return fail(ctx, HEVCDL_ERR_INVALID_ARG, "the chunk callback asked to stop");
    }
  }
  return HEVCDL_OK;
}

// ---- per-CTU session: the semantic drop-in for TEncCu::compressCtu + encodeCtu (TEncSlice.cpp:879,893) ----------------
extern "C" hevcdl_status hevcdl_begin_frames(hevcdl_ctx *ctx, const uint8_t *yuv, int n_frames, const uint8_t *labels_opt, uint8_t *labels_out_opt)
{
  hevcdl_status st = check_frames(ctx, n_frames); if (st) return st;
  if (n_frames == 0 || !yuv) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "begin_frames: no frames");
  st = ensure_staging(ctx); if (st) return st;
  if (!ctx->d_cabac) HIPCHK(hipMalloc(&ctx->d_cabac, (size_t)168 * ctx->cfg.max_frames));
  HIPCHK(hipMemcpy(ctx->d_yuv, yuv, ctx->frame_bytes * n_frames, hipMemcpyHostToDevice));
  if (labels_opt) { HIPCHK(hipMemcpy(ctx->d_labels, labels_opt, (size_t)ctx->ctus * 16 * n_frames, hipMemcpyHostToDevice)); st = hevcdl_clamp_labels_dev(ctx, ctx->d_labels, n_frames, nullptr); if (st) return st; }
  else { st = hevcdl_predict_depth_dev(ctx, ctx->d_yuv, n_frames, ctx->d_labels, nullptr, nullptr); if (st) return st; }
  HIPCHK(hipMemset(ctx->d_recon, 0, ctx->frame_bytes * n_frames));
  HIPCHK(hipMemset(ctx->d_records, 0, (size_t)ctx->ctus * sizeof(hevcdl_ctu_record) * n_frames));
  if (ctx->d_wpp) HIPCHK(hipMemset(ctx->d_wpp, 0, (size_t)256 * ctx->ctus_y * n_frames));      // WaveFrontSynchro: the rows' synchronisation contexts of this session
  HIPCHK(hipDeviceSynchronize());
  if (labels_out_opt) HIPCHK(hipMemcpy(labels_out_opt, ctx->d_labels, (size_t)ctx->ctus * 16 * n_frames, hipMemcpyDeviceToHost));
  ctx->next_ctu.assign(n_frames, 0); ctx->session_frames = n_frames;
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_compress_ctu(hevcdl_ctx *ctx, int frame, int ctu_addr, const hevcdl_cabac_state *state_in_opt,
                                             hevcdl_ctu_record *record, hevcdl_cabac_state *state_out_opt)
{
  if (!ctx) return HEVCDL_ERR_INVALID_ARG;
  if (ctx->cfg.tile_columns * ctx->cfg.tile_rows != 1) return fail(ctx, HEVCDL_ERR_UNSUPPORTED, "compress_ctu: the per-CTU session walks the untiled raster scan; with tiles use hevcdl_compress_frames");
  if (frame < 0 || frame >= ctx->session_frames) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "compress_ctu: frame outside the session (hevcdl_begin_frames)");
  if (ctu_addr < 0 || ctu_addr >= ctx->ctus || !record) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "compress_ctu: bad CTU address / null record");
  // the CTUs of a slice are a chain: neighbours' reconstruction and records must exist
  if (ctu_addr > ctx->next_ctu[frame]) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "compress_ctu: CTUs must be submitted in coding order");
  if (ctu_addr < ctx->next_ctu[frame] && !state_in_opt) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "compress_ctu: re-coding an earlier CTU needs its entry state");
  HIPCHK(hipSetDevice(ctx->cfg.device));
  unsigned char *cab = ctx->d_cabac + (size_t)168 * frame;
  if (state_in_opt) HIPCHK(hipMemcpy(cab, state_in_opt, 168, hipMemcpyHostToDevice));
  const void *cab_in = (state_in_opt || ctu_addr > 0) ? cab : nullptr;      // NULL: slice-start contexts from the QP
  hevcdl_status st = launch_rd(ctx, ctx->d_yuv + ctx->frame_bytes * frame, 1, ctx->d_labels + (size_t)ctx->ctus * 16 * frame,
                               ctx->d_records + (size_t)ctx->ctus * sizeof(hevcdl_ctu_record) * frame, ctx->d_recon + ctx->frame_bytes * frame, nullptr, nullptr,
                               ctu_addr, ctu_addr + 1, cab_in, cab, 0, -1, frame);
  if (st) return st;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(ctx, HEVCDL_ERR_HIP, "rd kernel", e);
  HIPCHK(hipMemcpy(record, ctx->d_records + ((size_t)ctx->ctus * frame + ctu_addr) * sizeof(hevcdl_ctu_record), sizeof(hevcdl_ctu_record), hipMemcpyDeviceToHost));
  if (state_out_opt) HIPCHK(hipMemcpy(state_out_opt, cab, 168, hipMemcpyDeviceToHost));
  if (ctu_addr == ctx->next_ctu[frame]) ctx->next_ctu[frame] = ctu_addr + 1;
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_get_recon(hevcdl_ctx *ctx, int frame, uint8_t *recon)
{
  if (!ctx) return HEVCDL_ERR_INVALID_ARG;
  if (frame < 0 || frame >= ctx->session_frames || !recon) return fail(ctx, HEVCDL_ERR_INVALID_ARG, "get_recon: frame outside the session / null pointer");
  HIPCHK(hipSetDevice(ctx->cfg.device));
  HIPCHK(hipMemcpy(recon, ctx->d_recon + ctx->frame_bytes * frame, ctx->frame_bytes, hipMemcpyDeviceToHost));
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_profile_enable(hevcdl_ctx *ctx, int enable)
{
  if (!ctx) return HEVCDL_ERR_INVALID_ARG;
  ctx->profile = enable != 0;
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_profile_get(hevcdl_ctx *ctx, hevcdl_profile *out)
{
  if (!ctx || !out) return HEVCDL_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(ctx->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  memset(out, 0, sizeof *out);
  for (int which = 0; which < 2; which++) {
    std::vector<hipEvent_t> &v = which ? ctx->ev_rd : ctx->ev_cnn;
    double ms = 0; unsigned n = 0;
    for (size_t i = 0; i + 1 < v.size(); i += 2) { float t = 0; if (hipEventElapsedTime(&t, v[i], v[i + 1]) == hipSuccess) { ms += t; n++; } hipEventDestroy(v[i]); hipEventDestroy(v[i + 1]); }
    v.clear();
    if (which) { out->rd_ms = ms; out->rd_launches = n; } else { out->cnn_ms = ms; out->cnn_launches = n; }
  }
  { std::vector<hipEvent_t> &v = ctx->ev_conv; double ms = 0;
    for (size_t i = 0; i + 1 < v.size(); i += 2) { float t = 0; if (hipEventElapsedTime(&t, v[i], v[i + 1]) == hipSuccess) ms += t; hipEventDestroy(v[i]); hipEventDestroy(v[i + 1]); }
    v.clear(); out->cnn_conv_ms = ms; }
  return HEVCDL_OK;
}
