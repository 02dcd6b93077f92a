// sao_kernel.hip -- sample adaptive offset for gfx950 (MI355X): second half of row f-2 of SURVEY.md section 8.
//
// Replaces TEncSampleAdaptiveOffset::SAOProcess (HM_dl/source/Lib/TLibEncoder/TEncSampleAdaptiveOffset.cpp:244-273, called at
// TEncGOP.cpp:1797) for the configuration of the hot path (8- or 10-bit 4:2:0, one slice, all-intra => SAO on at picture
// level, SAOLcuBoundary 0, offset step 1):
//   1. hevcdl_sao_stats_kernel   getStatistics / getBlkStats :295-341, 943-1335     one workgroup per (CTU, component); HBM bound
//   2. hevcdl_sao_decide_kernel  decideBlkParams / deriveModeNewRDO / deriveModeMergeRDO / deriveOffsets :421-941 with the
//                                counter coder of TEncSbac.cpp:1543-1720; the CTUs of a picture are a serial chain (merge
//                                candidates + adaptive contexts), pictures are independent: one lane per picture
//   3. hevcdl_sao_apply_kernel   TComSampleAdaptiveOffset::offsetCTU / offsetBlock  TComSampleAdaptiveOffset.cpp:316-620; HBM bound
// Statistics: per (type, class) sum of (org - deblocked) and count over the samples of the CTU whose neighbours exist and
// that lie outside the margin the reference leaves out next to a right / lower CTU (5/4 luma, 3/2 chroma columns/rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hevcdl.h"
#include "hevcdl_dev.h"

namespace {

#define GLB __attribute__((address_space(1)))
enum { EO_0 = 0, EO_90, EO_135, EO_45, BO, NTYPES, MODE_OFF = 0, MODE_NEW, MODE_MERGE, MERGE_LEFT = 0, MERGE_ABOVE };
constexpr double MAX_DOUBLE = 1.7e+308;

__constant__ int32_t s_entropy_bits[128] = { // ContextModel.cpp:103-112 (FAST_BIT_EST)
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };
__constant__ uint8_t s_next_mps[128] = { // ContextModel.cpp:68-101
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127 };
__constant__ uint8_t s_next_lps[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };

struct Stat { int32_t diff[32], count[32]; };                   // one (type) of one (CTU, component): 256 bytes
struct Sbac { uint8_t merge_ctx, type_ctx; unsigned long long frac; };

__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int clipbd(int v, int mx) { return v < 0 ? 0 : (v > mx ? mx : v); }      // ClipBD

// Which neighbouring CTUs of CTU a exist for SAO.  Offsets (for_stats 0): inside the picture and, with LFCrossTileBoundaryFlag 0, inside the
// same tile (TComPicSym::deriveLoopFilterBoundaryAvailibility; tiles are rectangles, so a tile border acts exactly like the picture
// border).  Statistics (for_stats 1): left / above likewise, but right / below only look at the picture
// (TEncSampleAdaptiveOffset::getStatistics :322-330 overrides them).
__device__ __forceinline__ void sao_neighbours(const hevcdl_sao_params &p, int a, int for_stats, int &left, int &right, int &above, int &below)
{
  const int cx = p.ctus_x, x = a % cx, y = a / cx, x0 = x * 64, y0 = y * 64;
  left = x0 > 0; above = y0 > 0; right = x0 + 64 < p.width; below = y0 + 64 < p.height;
  if (!p.lf_across_tiles) {
    for (int t = 1; t < p.tile_cols; t++) { if (p.col_bd[t] == x) left = 0; if (!for_stats && p.col_bd[t] == x + 1) right = 0; }
    for (int t = 1; t < p.tile_rows; t++) { if (p.row_bd[t] == y) above = 0; if (!for_stats && p.row_bd[t] == y + 1) below = 0; }
  }
}

// edge / band class of sample p for SAO type t
template <typename PEL> __device__ __forceinline__ int sao_class(int t, const PEL GLB *p, int stride, int band_shift)
{
  const int c = p[0];
  switch (t) {
    case EO_0:   return 2 + sgn(c - p[-1]) + sgn(c - p[1]);
    case EO_90:  return 2 + sgn(c - p[-stride]) + sgn(c - p[stride]);
    case EO_135: return 2 + sgn(c - p[-stride - 1]) + sgn(c - p[stride + 1]);
    case EO_45:  return 2 + sgn(c - p[-stride + 1]) + sgn(c - p[stride - 1]);
    default:     return c >> band_shift;          // bitDepth - 5
  }
}

// ---- decision (serial per picture, one lane) --------------------------------------------------------------------------
__device__ void sb_bin(Sbac &c, uint8_t &ctx, int bin)
{
  const uint8_t s = ctx;
  c.frac += (unsigned long long)s_entropy_bits[s ^ bin];
  ctx = bin == (s & 1) ? s_next_mps[s] : s_next_lps[s];
}
__device__ void sb_ep(Sbac &c, int n) { c.frac += 32768ull * (unsigned long long)n; }
__device__ uint32_t sb_bits(const Sbac &c) { return (uint32_t)(c.frac >> 15); }
__device__ void sb_reset(Sbac &c) { c.frac &= 32767ull; }

// bit-depth dependent constants of the decision: offset range (getMaxOffsetQVal: 7 / 31), distortion shift 2 (bitDepth - 8), rounding
struct Bd { int max_off, dist_shift, bits_above_8; };

__device__ void code_offset_param(Sbac &c, int comp, const hevcdl_sao_offset &p, const Bd &bd)
{ // codeSAOOffsetParam TEncSbac.cpp:1605-1681
  const int first = comp != 2;
  if (first) {
    const int sym = p.mode == MODE_OFF ? 0 : (p.type == BO ? 1 : 2);
    if (sym == 0) sb_bin(c, c.type_ctx, 0); else { sb_bin(c, c.type_ctx, 1); sb_ep(c, 1); }
  }
  if (p.mode == MODE_NEW) {
    int off[4], k = 0;
    const int ncls = p.type == BO ? 4 : 5;
    for (int i = 0; i < ncls; i++) { if (p.type != BO && i == 2) continue; off[k++] = p.offset[p.type == BO ? (p.aux + i) % 32 : i]; }
    for (int i = 0; i < 4; i++) { const int a = abs(off[i]); sb_ep(c, a == 0 ? 1 : (a < bd.max_off ? a + 1 : a)); }
    if (p.type == BO) { for (int i = 0; i < 4; i++) if (off[i]) sb_ep(c, 1); sb_ep(c, 5); }
    else if (first) sb_ep(c, 2);
  }
}
__device__ void code_blk_param(Sbac &c, const hevcdl_sao_blk &b, int left_avail, int above_avail, int only_merge, const Bd &bd)
{ // codeSAOBlkParam TEncSbac.cpp:1683-1720
  int is_left = 0, is_above = 0;
  if (left_avail) { is_left = b.c[0].mode == MODE_MERGE && b.c[0].type == MERGE_LEFT; sb_bin(c, c.merge_ctx, is_left); }
  if (above_avail && !is_left) { is_above = b.c[0].mode == MODE_MERGE && b.c[0].type == MERGE_ABOVE; sb_bin(c, c.merge_ctx, is_above); }
  if (only_merge) return;
  if (!is_left && !is_above) for (int comp = 0; comp < 3; comp++) code_offset_param(c, comp, b.c[comp], bd);
}
__device__ long long est_dist(long long count, long long offset, long long diff, const Bd &bd) { return (count * offset * offset - diff * offset * 2) >> bd.dist_shift; }   // estSaoDist :459
__device__ int est_iter_offset(int type, double lambda, int offset_in, long long count, long long diff, long long &best_dist, double &best_cost, const Bd &bd)
{ // estIterOffset :465-496
  int it = offset_in, out = 0;
  double min_cost = lambda;
  while (it != 0) {
    long long rate = type == BO ? abs(it) + 2 : abs(it) + 1;
    if (abs(it) == bd.max_off) rate--;
    const long long dist = est_dist(count, it, diff, bd);
    const double cost = (double)dist + lambda * (double)rate;
    if (cost < min_cost) { min_cost = cost; out = it; best_dist = dist; best_cost = cost; }
    it = it > 0 ? it - 1 : it + 1;
  }
  return out;
}
__device__ void derive_offsets(int type, double lambda, const Stat GLB &st, int32_t *q, int32_t &aux, const Bd &bd)
{ // deriveOffsets :498-615
  const int ncls = type == BO ? 32 : 5;
  for (int i = 0; i < 32; i++) q[i] = 0;
  for (int cls = 0; cls < ncls; cls++) {
    if (type != BO && cls == 2) continue;
    if (st.count[cls] == 0) continue;
    const double x = (double)((long long)st.diff[cls] << bd.bits_above_8) / (double)st.count[cls];     // :520-523, offset step log2 0
    int v;
    if (bd.bits_above_8) { const int r = 1 << bd.bits_above_8; v = x > 0 ? ((int)x + (r >> 1)) / r : ((int)x - (r >> 1)) / r; }   // xRoundIbdi2 :49-52
    else v = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);                                                     // xRoundIbdi :54-57
    q[cls] = v < -bd.max_off ? -bd.max_off : (v > bd.max_off ? bd.max_off : v);
  }
  if (type != BO) {
    for (int cls = 0; cls < 5; cls++) {
      long long d; double c;
      if ((cls == 0 || cls == 1) && q[cls] < 0) q[cls] = 0;
      if ((cls == 3 || cls == 4) && q[cls] > 0) q[cls] = 0;
      if (q[cls] != 0) q[cls] = est_iter_offset(type, lambda, q[cls], st.count[cls], st.diff[cls], d, c, bd);
    }
    aux = 0;
  } else {
    double cost[32], min_cost = MAX_DOUBLE;
    for (int cls = 0; cls < 32; cls++) {
      long long d; cost[cls] = lambda;
      if (q[cls] != 0) q[cls] = est_iter_offset(type, lambda, q[cls], st.count[cls], st.diff[cls], d, cost[cls], bd);
    }
    for (int band = 0; band < 29; band++) {
      double c = cost[band]; c += cost[band + 1]; c += cost[band + 2]; c += cost[band + 3];
      if (c < min_cost) { min_cost = c; aux = band; }
    }
    for (int i = 0; i < 32; i++) { const int r = (i - aux) & 31; if (r >= 4) q[i] = 0; }
  }
}
__device__ long long get_dist(int type, int aux, const int32_t *off, const Stat GLB &st, const Bd &bd)
{ // getDistortion :421-457
  long long d = 0;
  if (type != BO) for (int i = 0; i < 5; i++) d += est_dist(st.count[i], off[i], st.diff[i], bd);
  else for (int i = aux; i < aux + 4; i++) d += est_dist(st.count[i % 32], off[i % 32], st.diff[i % 32], bd);
  return d;
}

__device__ void load_off(hevcdl_sao_offset &d, const hevcdl_sao_offset GLB &s_) { d.mode = s_.mode; d.type = s_.type; d.aux = s_.aux; for (int i = 0; i < 32; i++) d.offset[i] = s_.offset[i]; }
__device__ void store_off(hevcdl_sao_offset GLB &d, const hevcdl_sao_offset &s_) { d.mode = s_.mode; d.type = s_.type; d.aux = s_.aux; for (int i = 0; i < 32; i++) d.offset[i] = s_.offset[i]; }

} // namespace

// stats[(frame * ctus + ctu) * 3 + comp][type].  The deblocked CTU (+1 sample halo) is staged in LDS with dword loads
// (sample x of row y at tile[y + 1][x + 4]: dword aligned); a thread handles 4 horizontally adjacent samples per step.
__global__ __launch_bounds__(256) void hevcdl_sao_stats_kernel(hevcdl_sao_params p)
{
  __shared__ int acc[NTYPES][2][32];
  __shared__ uint32_t tile[66][18];                                 // 72 bytes per row: 4 left pad + 64 + 4 right
  const int tid = threadIdx.x, a = blockIdx.x, comp = blockIdx.y, frame = blockIdx.z;
  for (int i = tid; i < NTYPES * 64; i += 256) (&acc[0][0][0])[i] = 0;
  const int cx = p.ctus_x, x0 = (a % cx) * 64, y0 = (a / cx) * 64;
  const int wl = x0 + 64 > p.width ? p.width - x0 : 64, hl = y0 + 64 > p.height ? p.height - y0 : 64;
  const int sh = comp ? 1 : 0, stride = p.width >> sh, w = wl >> sh, h = hl >> sh, ph = p.height >> sh;
  int left, right, above, below; sao_neighbours(p, a, 1, left, right, above, below);
  const int skip_r = comp ? 3 : 5, skip_b = comp ? 2 : 4;
  const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
  const size_t plane = (size_t)frame * fsz + (comp == 0 ? 0 : (comp == 1 ? ysz : ysz + (ysz >> 2)));
  const int bx = x0 >> sh, by = y0 >> sh;
  const uint8_t GLB *src = (const uint8_t GLB *)p.deblocked + plane, *org = (const uint8_t GLB *)p.org + plane;
  const int wd = w >> 2;                                            // dwords per row (w is a multiple of 4)
  for (int i = tid; i < (h + 2) * (wd + 2); i += 256) {             // rows -1..h, dword columns -1..wd (clamped inside the picture: unused there)
    const int r = i / (wd + 2), c = i - r * (wd + 2);
    int yy = by + r - 1, xx = bx + (c - 1) * 4;
    yy = yy < 0 ? 0 : (yy >= ph ? ph - 1 : yy); xx = xx < 0 ? 0 : (xx + 4 > stride ? stride - 4 : xx);
    tile[r][c] = *(const uint32_t GLB *)(src + (size_t)yy * stride + xx);
  }
  __syncthreads();
  // Per-thread edge statistics in packed fields (a thread sees at most 16 samples of a 64x64 block): counts of the five classes in 6-bit
  // fields of one word; sums of (difference + 256) of classes 0..3 in 13-bit fields of a 64-bit word (16 * 511 < 8192; a class-4 sample
  // lands above them and can only carry upwards), class 4 in a word of its own.  One shift-and-add per sample and type instead of five
  // compare-and-adds; unpacked once after the loop.
  uint32_t pc[4] = { 0, 0, 0, 0 }, pd4[4] = { 0, 0, 0, 0 };
  unsigned long long pd[4] = { 0, 0, 0, 0 };
  for (int i = tid; i < h * wd; i += 256) {
    const int y = i / wd, xd = i - y * wd;
    const uint32_t o4 = *(const uint32_t GLB *)(org + (size_t)(by + y) * stride + bx + xd * 4);
    uint32_t row[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) row[r][c] = tile[y + r][xd + c];
    auto px = [&](int r, int k) -> int { const int b = 4 + k; return (int)((row[r][b >> 2] >> (8 * (b & 3))) & 255u); };   // sample (y - 1 + r, 4 xd + k), k = -1..4
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int x = xd * 4 + k, c0 = px(1, k), d = (int)((o4 >> (8 * k)) & 255u) - c0;
#pragma unroll
      for (int t = 0; t < NTYPES; t++) {
        const bool need_lr = (t == EO_0 || t == EO_135 || t == EO_45), need_ab = (t == EO_90 || t == EO_135 || t == EO_45);
        const int sx = need_lr ? (left ? 0 : 1) : 0, ex = right ? w - skip_r : (need_lr ? w - 1 : w);
        const int sy = need_ab ? (above ? 0 : 1) : 0, ey = below ? h - skip_b : (need_ab ? h - 1 : h);
        if (x >= sx && x < ex && y >= sy && y < ey) {
          if (t == BO) { atomicAdd(&acc[BO][0][c0 >> 3], d); atomicAdd(&acc[BO][1][c0 >> 3], 1); }
          else {
            const int n0 = t == EO_0 ? px(1, k - 1) : (t == EO_90 ? px(0, k) : (t == EO_135 ? px(0, k - 1) : px(0, k + 1)));
            const int n1 = t == EO_0 ? px(1, k + 1) : (t == EO_90 ? px(2, k) : (t == EO_135 ? px(2, k + 1) : px(2, k - 1)));
            const int cls = 2 + sgn(c0 - n0) + sgn(c0 - n1);
            const int tt = t < 4 ? t : 0;
            pc[tt] += 1u << (6 * cls);
            pd[tt] += (unsigned long long)(uint32_t)(d + 256) << (13 * cls);
            pd4[tt] += cls == 4 ? (uint32_t)(d + 256) : 0u;
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int k = 0; k < 5; k++) {
      int vc = (int)((pc[t] >> (6 * k)) & 63u);
      int vd = (k < 4 ? (int)((pd[t] >> (13 * k)) & 8191ull) : (int)pd4[t]) - 256 * vc;
      vd += __builtin_amdgcn_update_dpp(0, vd, 0xB1, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0xB1, 0xf, 0xf, false);
      vd += __builtin_amdgcn_update_dpp(0, vd, 0x4E, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0x4E, 0xf, 0xf, false);
      vd += __builtin_amdgcn_update_dpp(0, vd, 0x141, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0x141, 0xf, 0xf, false);
      vd += __builtin_amdgcn_update_dpp(0, vd, 0x140, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0x140, 0xf, 0xf, false);
      if ((tid & 15) == 0) { atomicAdd(&acc[t][0][k], vd); atomicAdd(&acc[t][1][k], vc); }
    }
  __syncthreads();
  Stat GLB *dst = (Stat GLB *)p.stats + ((size_t)(frame * p.ctus_per_frame + a) * 3 + comp) * NTYPES;
  for (int i = tid; i < NTYPES * 64; i += 256) { const int t = i >> 6, r = i & 63; if (r < 32) dst[t].diff[r] = acc[t][0][r]; else dst[t].count[r - 32] = acc[t][1][r - 32]; }
}

// The same statistics for pictures of 16-bit samples (10-bit): the CTU (+1 sample halo) staged in LDS as samples, one sample per
// thread and step; band class = sample >> (bitDepth - 5).
__global__ __launch_bounds__(256) void hevcdl_sao_stats16_kernel(hevcdl_sao_params p)
{
  __shared__ int acc[NTYPES][2][32];
  __shared__ uint16_t tile[66][68];                                 // sample (y, x) of the CTU component at tile[y + 1][x + 1]
  const int tid = threadIdx.x, a = blockIdx.x, comp = blockIdx.y, frame = blockIdx.z;
  for (int i = tid; i < NTYPES * 64; i += 256) (&acc[0][0][0])[i] = 0;
  const int cx = p.ctus_x, x0 = (a % cx) * 64, y0 = (a / cx) * 64;
  const int wl = x0 + 64 > p.width ? p.width - x0 : 64, hl = y0 + 64 > p.height ? p.height - y0 : 64;
  const int sh = comp ? 1 : 0, stride = p.width >> sh, w = wl >> sh, h = hl >> sh, ph = p.height >> sh;
  int left, right, above, below; sao_neighbours(p, a, 1, left, right, above, below);
  const int skip_r = comp ? 3 : 5, skip_b = comp ? 2 : 4, band_shift = p.bit_depth - 5;
  const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
  const size_t plane = (size_t)frame * fsz + (comp == 0 ? 0 : (comp == 1 ? ysz : ysz + (ysz >> 2)));
  const int bx = x0 >> sh, by = y0 >> sh;
  const uint16_t GLB *src = (const uint16_t GLB *)p.deblocked + plane, *org = (const uint16_t GLB *)p.org + plane;
  for (int i = tid; i < (h + 2) * (w + 2); i += 256) {              // rows -1..h, columns -1..w (clamped inside the picture: unused there)
    const int r = i / (w + 2), c = i - r * (w + 2);
    int yy = by + r - 1, xx = bx + c - 1;
    yy = yy < 0 ? 0 : (yy >= ph ? ph - 1 : yy); xx = xx < 0 ? 0 : (xx >= stride ? stride - 1 : xx);
    tile[r][c] = src[(size_t)yy * stride + xx];
  }
  __syncthreads();
  int eo_d[4][5], eo_c[4][5];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int k = 0; k < 5; k++) { eo_d[t][k] = 0; eo_c[t][k] = 0; }
  for (int i = tid; i < h * w; i += 256) {
    const int y = i / w, x = i - y * w;
    auto px = [&](int dy, int dx) -> int { return (int)tile[y + 1 + dy][x + 1 + dx]; };
    const int c0 = px(0, 0), d = (int)org[(size_t)(by + y) * stride + bx + x] - c0;
#pragma unroll
    for (int t = 0; t < NTYPES; t++) {
      const bool need_lr = (t == EO_0 || t == EO_135 || t == EO_45), need_ab = (t == EO_90 || t == EO_135 || t == EO_45);
      const int sx = need_lr ? (left ? 0 : 1) : 0, ex = right ? w - skip_r : (need_lr ? w - 1 : w);
      const int sy = need_ab ? (above ? 0 : 1) : 0, ey = below ? h - skip_b : (need_ab ? h - 1 : h);
      if (x >= sx && x < ex && y >= sy && y < ey) {
        if (t == BO) { atomicAdd(&acc[BO][0][c0 >> band_shift], d); atomicAdd(&acc[BO][1][c0 >> band_shift], 1); }
        else {
          const int n0 = t == EO_0 ? px(0, -1) : (t == EO_90 ? px(-1, 0) : (t == EO_135 ? px(-1, -1) : px(-1, 1)));
          const int n1 = t == EO_0 ? px(0, 1) : (t == EO_90 ? px(1, 0) : (t == EO_135 ? px(1, 1) : px(1, -1)));
          const int cls = 2 + sgn(c0 - n0) + sgn(c0 - n1);
#pragma unroll
          for (int q = 0; q < 5; q++) { eo_c[t < 4 ? t : 0][q] += (cls == q); eo_d[t < 4 ? t : 0][q] += (cls == q) ? d : 0; }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int k = 0; k < 5; k++) {
      int vd = eo_d[t][k], vc = eo_c[t][k];
      vd += __builtin_amdgcn_update_dpp(0, vd, 0xB1, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0xB1, 0xf, 0xf, false);
      vd += __builtin_amdgcn_update_dpp(0, vd, 0x4E, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0x4E, 0xf, 0xf, false);
      vd += __builtin_amdgcn_update_dpp(0, vd, 0x141, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0x141, 0xf, 0xf, false);
      vd += __builtin_amdgcn_update_dpp(0, vd, 0x140, 0xf, 0xf, false); vc += __builtin_amdgcn_update_dpp(0, vc, 0x140, 0xf, 0xf, false);
      if ((tid & 15) == 0) { atomicAdd(&acc[t][0][k], vd); atomicAdd(&acc[t][1][k], vc); }
    }
  __syncthreads();
  Stat GLB *dst = (Stat GLB *)p.stats + ((size_t)(frame * p.ctus_per_frame + a) * 3 + comp) * NTYPES;
  for (int i = tid; i < NTYPES * 64; i += 256) { const int t = i >> 6, r = i & 63; if (r < 32) dst[t].diff[r] = acc[t][0][r]; else dst[t].count[r - 32] = acc[t][1][r - 32]; }
}

// Candidate offsets of every (picture, CTU, component, type): deriveOffsets + getDistortion depend on the statistics and lambda only -- not on the
// coder state or the neighbours' decisions that chain the CTUs of a picture -- so they run ahead of the chain, one thread each.  A set of offsets
// has at most four nonzero entries (edge classes 0, 1, 3, 4; the four bands from the band position on): that is how the chain carries it.
struct Cand { int8_t off[4]; int32_t aux; long long dist; };          // 16 bytes, same order as the statistics
struct Cmp { int mode, type, aux, off[4]; };                            // one component's parameters in the chain (registers)
struct CRec { uint32_t hdr[3], off[3]; int32_t merge; uint32_t pad; };  // what the chain leaves per CTU: resolved parameters (mode | type << 8 | aux << 16, four int8 offsets) + merge direction or -1
__global__ __launch_bounds__(256) void hevcdl_sao_offsets_kernel(hevcdl_sao_params p)
{
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)p.n_frames * p.ctus_per_frame * 3 * NTYPES;
  if (idx >= total) return;
  const int type = (int)(idx % NTYPES), comp = (int)((idx / NTYPES) % 3);
  const Bd bd = { (1 << ((p.bit_depth < 10 ? p.bit_depth : 10) - 5)) - 1, 2 * (p.bit_depth - 8), p.bit_depth - 8 };
  const Stat GLB &st = ((const Stat GLB *)p.stats)[idx];
  int32_t q[32], aux;
  derive_offsets(type, comp ? p.lambda_chroma : p.lambda, st, q, aux, bd);
  Cand GLB &c = ((Cand GLB *)p.cand)[idx];
  for (int i = 0; i < 4; i++) c.off[i] = (int8_t)(type == BO ? q[(aux + i) & 31] : q[i < 2 ? i : i + 1]);
  c.aux = aux; c.dist = get_dist(type, aux, q, st, bd);
}

__device__ void code_offset_cmp(Sbac &c, int comp, const Cmp &p, const Bd &bd)
{ // codeSAOOffsetParam TEncSbac.cpp:1605-1681 on the compact form
  const int first = comp != 2;
  if (first) {
    const int sym = p.mode == MODE_OFF ? 0 : (p.type == BO ? 1 : 2);
    if (sym == 0) sb_bin(c, c.type_ctx, 0); else { sb_bin(c, c.type_ctx, 1); sb_ep(c, 1); }
  }
  if (p.mode == MODE_NEW) {
    for (int i = 0; i < 4; i++) { const int a = abs(p.off[i]); sb_ep(c, a == 0 ? 1 : (a < bd.max_off ? a + 1 : a)); }
    if (p.type == BO) { for (int i = 0; i < 4; i++) if (p.off[i]) sb_ep(c, 1); sb_ep(c, 5); }
    else if (first) sb_ep(c, 2);
  }
}
__device__ void code_blk_cmp(Sbac &c, const Cmp (&b)[3], int left_avail, int above_avail, int only_merge, const Bd &bd)
{ // codeSAOBlkParam TEncSbac.cpp:1683-1720
  int is_left = 0, is_above = 0;
  if (left_avail) { is_left = b[0].mode == MODE_MERGE && b[0].type == MERGE_LEFT; sb_bin(c, c.merge_ctx, is_left); }
  if (above_avail && !is_left) { is_above = b[0].mode == MODE_MERGE && b[0].type == MERGE_ABOVE; sb_bin(c, c.merge_ctx, is_above); }
  if (only_merge) return;
  if (!is_left && !is_above) for (int comp = 0; comp < 3; comp++) code_offset_cmp(c, comp, b[comp], bd);
}
__device__ long long get_dist_cmp(int type, int aux, const int (&off)[4], const Stat GLB &st, const Bd &bd)
{ // getDistortion :421-457 (an edge class 2 / a band outside the four has offset 0 and adds nothing)
  long long d = 0;
  for (int i = 0; i < 4; i++) { const int k = type == BO ? (aux + i) & 31 : (i < 2 ? i : i + 1); d += est_dist(st.count[k], off[i], st.diff[k], bd); }
  return d;
}
__device__ void crec_get(const CRec GLB &r, int c, Cmp &o)
{
  const uint32_t h = r.hdr[c], f = r.off[c];
  o.mode = (int)(h & 255u); o.type = (int)((h >> 8) & 255u); o.aux = (int)(h >> 16);
  for (int i = 0; i < 4; i++) o.off[i] = (int)(int8_t)((f >> (8 * i)) & 255u);
}

// one lane per picture: the CTU chain of decideBlkParams.  Everything it carries is compact (Cmp / CRec): the 32-entry parameter blocks of
// the interface are written afterwards by hevcdl_sao_expand_kernel, in parallel.
__global__ __launch_bounds__(64) void hevcdl_sao_decide_kernel(hevcdl_sao_params p)
{
  const int frame = blockIdx.x * 64 + threadIdx.x;
  if (frame >= p.n_frames) return;
  const int cx = p.ctus_x, nctu = p.ctus_per_frame;
  const double lambda[3] = { p.lambda, p.lambda_chroma, p.lambda_chroma };
  const Bd bd = { (1 << ((p.bit_depth < 10 ? p.bit_depth : 10) - 5)) - 1, 2 * (p.bit_depth - 8), p.bit_depth - 8 };
  const Stat GLB *stats = (const Stat GLB *)p.stats + (size_t)frame * nctu * 3 * NTYPES;
  CRec GLB *crec = (CRec GLB *)p.crec + (size_t)frame * nctu;
  Sbac go, cur, next, mid, temp;
  { // initRDOCabacCoder: I-slice contexts at the slice QP (ContextTables.h:445-458)
    const int init[2] = { 153, 200 };
    for (int i = 0; i < 2; i++) {
      const int v = init[i], slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
      int s = ((slope * p.qp) >> 4) + offset; s = s < 1 ? 1 : (s > 126 ? 126 : s);
      const int mps = s >= 64; const uint8_t st = (uint8_t)(((mps ? s - 64 : 63 - s) << 1) + mps);
      if (i == 0) go.merge_ctx = st; else go.type_ctx = st;
    }
    go.frac = 0;
  }
  next = go;
  const Cmp off_cmp = { MODE_OFF, 0, 0, { 0, 0, 0, 0 } };
  for (int a = 0; a < nctu; a++) {
    const Stat GLB *st = stats + (size_t)a * 3 * NTYPES;
    const Cand GLB *cand = (const Cand GLB *)p.cand + ((size_t)frame * nctu + a) * 3 * NTYPES;      // hevcdl_sao_offsets_kernel
    // merge candidates come from the same tile only (TComPic::getSAOMergeAvailability)
    bool left_av = true, above_av = true;
    for (int t = 0; t < p.tile_cols; t++) if (p.col_bd[t] == a % cx) left_av = false;
    for (int t = 0; t < p.tile_rows; t++) if (p.row_bd[t] == a / cx) above_av = false;
    Cmp best[3], mode[3];
    int best_merge = -1;
    double min_cost = MAX_DOUBLE;
    cur = go;
    auto take = [&](Cmp &d, int type, const Cand GLB &cd) { d.mode = MODE_NEW; d.type = type; d.aux = cd.aux; for (int i = 0; i < 4; i++) d.off[i] = cd.off[i]; };
    { // ---- deriveModeNewRDO :617-758 ----
      Cmp test[3];
      long long dist[3], mode_dist[3] = { 0, 0, 0 };
      for (int c = 0; c < 3; c++) mode[c] = off_cmp;
      go = cur; code_blk_cmp(go, mode, left_av, above_av, 1, bd); mid = go;
      sb_reset(go); code_offset_cmp(go, 0, mode[0], bd);
      double mc = lambda[0] * (double)sb_bits(go), cost;
      temp = go;
      for (int type = 0; type < NTYPES; type++) {
        take(test[0], type, cand[0 * NTYPES + type]); dist[0] = cand[0 * NTYPES + type].dist;
        go = mid; sb_reset(go); code_offset_cmp(go, 0, test[0], bd);
        cost = (double)dist[0] + lambda[0] * (double)(int)sb_bits(go);
        if (cost < mc) { mc = cost; mode_dist[0] = dist[0]; mode[0] = test[0]; temp = go; }
      }
      go = temp; mid = go;
      cost = 0; sb_reset(go);
      { uint32_t prev = 0; for (int c = 1; c < 3; c++) { code_offset_cmp(go, c, mode[c], bd); const uint32_t b = sb_bits(go); cost += lambda[c] * (double)(b - prev); prev = b; } }
      mc = cost;
      for (int type = 0; type < NTYPES; type++) {
        uint32_t prev = 0;
        go = mid; sb_reset(go); cost = 0;
        for (int c = 1; c < 3; c++) {
          take(test[c], type, cand[c * NTYPES + type]); dist[c] = cand[c * NTYPES + type].dist;
          code_offset_cmp(go, c, test[c], bd);
          const uint32_t b = sb_bits(go);
          cost += (double)dist[c] + (lambda[c] * (double)(b - prev));
          prev = b;
        }
        if (cost < mc) { mc = cost; for (int c = 1; c < 3; c++) { mode_dist[c] = dist[c]; mode[c] = test[c]; } }
      }
      double norm = 0;
      for (int c = 0; c < 3; c++) norm += (double)mode_dist[c] / lambda[c];
      go = cur; sb_reset(go); code_blk_cmp(go, mode, left_av, above_av, 0, bd);
      norm += (double)sb_bits(go);
      if (norm < min_cost) { min_cost = norm; for (int c = 0; c < 3; c++) best[c] = mode[c]; best_merge = -1; next = go; }
    }
    // ---- deriveModeMergeRDO :760-812 ----
    for (int mt = 0; mt < 2; mt++) {
      if (!(mt == MERGE_LEFT ? left_av : above_av)) continue;
      const CRec GLB &m = crec[mt == MERGE_LEFT ? a - 1 : a - cx];          // the neighbour's resolved parameters
      Cmp res[3];
      double nd = 0;
      for (int c = 0; c < 3; c++) {
        crec_get(m, c, res[c]);
        if (res[c].mode != MODE_OFF) nd += ((double)get_dist_cmp(res[c].type, res[c].aux, res[c].off, st[c * NTYPES + res[c].type], bd)) / lambda[c];
        mode[c] = res[c]; mode[c].mode = MODE_MERGE; mode[c].type = mt;
      }
      go = cur; sb_reset(go); code_blk_cmp(go, mode, left_av, above_av, 0, bd);
      const double cost = nd + (double)(int)sb_bits(go);
      if (cost < min_cost) { min_cost = cost; for (int c = 0; c < 3; c++) best[c] = res[c]; best_merge = mt; next = go; }
    }
    go = next;
    CRec out;                                                              // reconstructBlkSAOParam: a merged CTU carries the neighbour's parameters
    for (int c = 0; c < 3; c++) {
      out.hdr[c] = (uint32_t)best[c].mode | ((uint32_t)best[c].type << 8) | ((uint32_t)best[c].aux << 16);
      out.off[c] = ((uint32_t)best[c].off[0] & 255u) | (((uint32_t)best[c].off[1] & 255u) << 8) | (((uint32_t)best[c].off[2] & 255u) << 16) | (((uint32_t)best[c].off[3] & 255u) << 24);
    }
    out.merge = best_merge; out.pad = 0;
    { CRec GLB &dst = crec[a]; for (int c = 0; c < 3; c++) { dst.hdr[c] = out.hdr[c]; dst.off[c] = out.off[c]; } dst.merge = out.merge; dst.pad = 0; }
  }
}

// The interface's parameter blocks from the chain's compact records, one thread per (picture, CTU): coded parameters (a merged CTU: mode
// MERGE + direction, offsets of the candidate as the reference's SAOBlkParam holds them) and the resolved ones the reconstruction reads.
__global__ __launch_bounds__(256) void hevcdl_sao_expand_kernel(hevcdl_sao_params p)
{
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)p.n_frames * p.ctus_per_frame;
  if (idx >= total) return;
  const CRec GLB &r = ((const CRec GLB *)p.crec)[idx];
  hevcdl_sao_blk GLB &par = ((hevcdl_sao_blk GLB *)p.params)[idx], &rec = ((hevcdl_sao_blk GLB *)p.recon_params)[idx];
  const int merge = r.merge;
  for (int c = 0; c < 3; c++) {
    Cmp v; crec_get(r, c, v);
    rec.c[c].mode = v.mode; rec.c[c].type = v.type; rec.c[c].aux = v.aux;
    par.c[c].mode = merge >= 0 ? MODE_MERGE : v.mode; par.c[c].type = merge >= 0 ? merge : v.type; par.c[c].aux = v.aux;
    for (int i = 0; i < 32; i++) {
      int o = 0;
      if (v.mode != MODE_OFF) {
        if (v.type == BO) { const int k = (i - v.aux) & 31; if (k < 4) o = v.off[k]; }
        else if (i < 5 && i != 2) o = v.off[i < 2 ? i : i - 1];
      }
      rec.c[c].offset[i] = o; par.c[c].offset[i] = o;
    }
  }
}

template <typename PEL>
__global__ __launch_bounds__(256) void hevcdl_sao_apply_kernel(hevcdl_sao_params p)
{
  const int tid = threadIdx.x, a = blockIdx.x, comp = blockIdx.y, frame = blockIdx.z;
  const hevcdl_sao_offset GLB &prm = ((const hevcdl_sao_blk GLB *)p.recon_params)[(size_t)frame * p.ctus_per_frame + a].c[comp];
  const int cx = p.ctus_x, x0 = (a % cx) * 64, y0 = (a / cx) * 64;
  const int wl = x0 + 64 > p.width ? p.width - x0 : 64, hl = y0 + 64 > p.height ? p.height - y0 : 64;
  const int sh = comp ? 1 : 0, stride = p.width >> sh, w = wl >> sh, h = hl >> sh;
  const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
  const size_t off = (size_t)frame * fsz + (comp == 0 ? 0 : (comp == 1 ? ysz : ysz + (ysz >> 2))) + (size_t)(y0 >> sh) * stride + (x0 >> sh);
  const PEL GLB *src = (const PEL GLB *)p.deblocked + off; PEL GLB *res = (PEL GLB *)p.out + off;
  const int pel_max = (1 << p.bit_depth) - 1, band_shift = p.bit_depth - 5;
  const int mode = prm.mode, type = prm.type;
  __shared__ int offs[32];
  if (tid < 32) offs[tid] = (mode != MODE_OFF && (type == BO || tid < 5)) ? prm.offset[tid] : 0;
  __syncthreads();
  const bool need_lr = (type == EO_0 || type == EO_135 || type == EO_45), need_ab = (type == EO_90 || type == EO_135 || type == EO_45);
  int left, right, above, below; sao_neighbours(p, a, 0, left, right, above, below);
  const int sx = (need_lr && !left) ? 1 : 0, ex = (need_lr && !right) ? w - 1 : w;
  const int sy = (need_ab && !above) ? 1 : 0, ey = (need_ab && !below) ? h - 1 : h;
  for (int i = tid; i < w * h; i += 256) {
    const int y = i / w, x = i - y * w;
    const PEL GLB *s = src + (size_t)y * stride + x;
    int v = s[0];
    if (mode != MODE_OFF && x >= sx && x < ex && y >= sy && y < ey) v = clipbd(v + offs[sao_class<PEL>(type, s, stride, band_shift)], pel_max);
    res[(size_t)y * stride + x] = (PEL)v;
  }
}

// The same for 8-bit pictures with four samples per thread and step: dword loads of the row and of the two neighbouring rows / columns (served
// by the vector L1), one dword store; adjacent lanes on adjacent dwords.  A CTU without offsets (mode off, or a type whose offsets are all 0) is a copy.
__global__ __launch_bounds__(256) void hevcdl_sao_apply8_kernel(hevcdl_sao_params p)
{
  const int tid = threadIdx.x, a = blockIdx.x, comp = blockIdx.y, frame = blockIdx.z;
  const hevcdl_sao_offset GLB &prm = ((const hevcdl_sao_blk GLB *)p.recon_params)[(size_t)frame * p.ctus_per_frame + a].c[comp];
  const int cx = p.ctus_x, x0 = (a % cx) * 64, y0 = (a / cx) * 64;
  const int wl = x0 + 64 > p.width ? p.width - x0 : 64, hl = y0 + 64 > p.height ? p.height - y0 : 64;
  const int sh = comp ? 1 : 0, stride = p.width >> sh, w = wl >> sh, h = hl >> sh, wd = w >> 2;
  const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
  const size_t plane = (size_t)frame * fsz + (comp == 0 ? 0 : (comp == 1 ? ysz : ysz + (ysz >> 2)));
  const size_t plane_bytes = comp == 0 ? ysz : (ysz >> 2);
  const size_t off = (size_t)(y0 >> sh) * stride + (x0 >> sh);
  const uint8_t GLB *pl = (const uint8_t GLB *)p.deblocked + plane; uint8_t GLB *res = (uint8_t GLB *)p.out + plane;
  const int mode = prm.mode, type = prm.type;
  __shared__ int offs[32];
  if (tid < 32) offs[tid] = (mode != MODE_OFF && (type == BO || tid < 5)) ? prm.offset[tid] : 0;
  __syncthreads();
  const bool need_lr = (type == EO_0 || type == EO_135 || type == EO_45), need_ab = (type == EO_90 || type == EO_135 || type == EO_45);
  int left, right, above, below; sao_neighbours(p, a, 0, left, right, above, below);
  const int sx = (need_lr && !left) ? 1 : 0, ex = (need_lr && !right) ? w - 1 : w;
  const int sy = (need_ab && !above) ? 1 : 0, ey = (need_ab && !below) ? h - 1 : h;
  // dword at sample offset o of the plane; o is clamped into the plane (a clamped dword only ever feeds samples outside [sx, ex) x [sy, ey))
  auto ld = [&](long long o) -> uint32_t { o = o < 0 ? 0 : (o > (long long)plane_bytes - 4 ? (long long)plane_bytes - 4 : o); return *(const uint32_t GLB *)(pl + o); };
  const int dx = type == EO_135 ? -1 : (type == EO_45 ? 1 : 0);      // column of the neighbour in the row above (the one below: -dx)
  for (int i = tid; i < wd * h; i += 256) {
    const int y = i / wd, xd = i - y * wd;
    const long long o = (long long)off + (long long)y * stride + 4 * xd;
    const uint32_t c4 = ld(o);
    uint32_t out = c4;
    if (mode != MODE_OFF && y >= sy && y < ey) {
      int cls[4];
      if (type == BO) {
#pragma unroll
        for (int k = 0; k < 4; k++) cls[k] = (int)((c4 >> (8 * k + 3)) & 31u);
      } else {
        // the 6 samples left neighbour .. right neighbour of the row holding the first neighbour (n0) and of the row holding the second (n1)
        unsigned long long r0, r1;
        if (type == EO_0) { r0 = r1 = ((unsigned long long)ld(o + 4) << 40) | ((unsigned long long)c4 << 8) | (ld(o - 4) >> 24); }
        else {
          const long long ou = o - stride, od = o + stride;
          r0 = ((unsigned long long)ld(ou + 4) << 40) | ((unsigned long long)ld(ou) << 8) | (ld(ou - 4) >> 24);
          r1 = ((unsigned long long)ld(od + 4) << 40) | ((unsigned long long)ld(od) << 8) | (ld(od - 4) >> 24);
        }
        // byte j of r = sample 4 xd - 1 + j of that row
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int c0 = (int)((c4 >> (8 * k)) & 255u);
          const int n0 = (int)((r0 >> (8 * (k + 1 + (type == EO_0 ? -1 : dx)))) & 255u), n1 = (int)((r1 >> (8 * (k + 1 + (type == EO_0 ? 1 : -dx)))) & 255u);
          cls[k] = 2 + sgn(c0 - n0) + sgn(c0 - n1);
        }
      }
      out = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int x = 4 * xd + k, c0 = (int)((c4 >> (8 * k)) & 255u);
        const int v = (x >= sx && x < ex) ? clipbd(c0 + offs[cls[k]], 255) : c0;
        out |= (uint32_t)v << (8 * k);
      }
    }
    *(uint32_t GLB *)(res + o) = out;
  }
}

extern "C" void hevcdl_launch_sao(const hevcdl_sao_params *pp, void *stream)
{
  const hevcdl_sao_params p = *pp;
  hipStream_t s = (hipStream_t)stream;
  const dim3 per_ctu(p.ctus_per_frame, 3, p.n_frames);
  if (p.bit_depth == 8) hipLaunchKernelGGL(hevcdl_sao_stats_kernel, per_ctu, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(hevcdl_sao_stats16_kernel, per_ctu, dim3(256), 0, s, p);
  const size_t n_cand = (size_t)p.n_frames * p.ctus_per_frame * 3 * NTYPES;
  hipLaunchKernelGGL(hevcdl_sao_offsets_kernel, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, s, p);
  hipLaunchKernelGGL(hevcdl_sao_decide_kernel, dim3((p.n_frames + 63) / 64), dim3(64), 0, s, p);
  const size_t n_ctu = (size_t)p.n_frames * p.ctus_per_frame;
  hipLaunchKernelGGL(hevcdl_sao_expand_kernel, dim3((unsigned)((n_ctu + 255) / 256)), dim3(256), 0, s, p);
  if (p.bit_depth == 8) hipLaunchKernelGGL(hevcdl_sao_apply8_kernel, per_ctu, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(hevcdl_sao_apply_kernel<uint16_t>, per_ctu, dim3(256), 0, s, p);
}
