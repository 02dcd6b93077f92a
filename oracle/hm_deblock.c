/* oracle/hm_deblock.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the reference's deblocking filter for the configuration of the hot path
 * (all-intra, one slice, constant QP, no PCM / lossless, beta/tc offsets 0, 8- or 10-bit 4:2:0):
 *   TComLoopFilter::loopFilterPic            TLibCommon/TComLoopFilter.cpp:130-156  (all vertical edges, then all horizontal)
 *   xDeblockCU                               :170-239  (edges on the 8x8 grid; chroma on its own 8x8 grid)
 *   xSetEdgefilterTU / xSetEdgefilterPU      :274-360  (left/top edge of every TU and CU, not at the picture border)
 *   xGetBoundaryStrengthSingle               :416-440  (intra on either side => Bs 2)
 *   xEdgeFilterLuma / xEdgeFilterChroma      :557-826
 *   xPelFilterLuma / xPelFilterChroma / xUseStrongFiltering / xCalcDP / xCalcDQ   :830-954
 * Input: the reconstruction compressCtu leaves (before in-loop filters) + the CTU records; output: the picture the
 * reference hands to SAO.  Pinned by tests/golden/rd_*.npz:recon_deblocked (reference run with --SAO=0).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "hm_oracle.h"
#include "hm_tables.h"

static const uint8_t tc_table[54] = { /* TComLoopFilter.cpp:59-62 */
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24 };
static const uint8_t beta_table[52] = { /* :64-67 */
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };

static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
typedef uint16_t spx;                 /* the filter works on 16-bit sample planes; the 8-bit entry widens and narrows */
static int g_pel_max = 255;
static int clip8(int v) { return v < 0 ? 0 : (v > g_pel_max ? g_pel_max : v); }      /* ClipBD */
static int iabs(int v) { return v < 0 ? -v : v; }

static int z_of(int x4, int y4)
{ /* z-scan index of the 4x4 partition (x4, y4) inside its CTU (TComRom.cpp:284-352) */
  int z = 0, b;
  for (b = 0; b < 4; b++) z |= (((x4 >> b) & 1) << (2 * b)) | (((y4 >> b) & 1) << (2 * b + 1));
  return z;
}

/* TU edge at the left (dir 0) / top (dir 1) border of the 4x4 partition at luma (x, y)? */
static int edge_flag(const hm_ctu_record *recs, int ctus_x, int x, int y, int dir)
{
  const hm_ctu_record *r = recs + (y >> 6) * ctus_x + (x >> 6);
  const int z = z_of((x & 63) >> 2, (y & 63) >> 2);
  const int tu = 64 >> (r->depth[z] + r->tr_idx[z]);
  const int pos = dir ? y : x;
  return pos > 0 && (pos % tu) == 0;
}

static void pel_filter_luma(spx *p, int off, int tc, int sw, int thr_cut, int second_p, int second_q)
{ /* xPelFilterLuma :830-893 */
  const int m4 = p[0], m3 = p[-off], m5 = p[off], m2 = p[-2 * off], m6 = p[2 * off], m1 = p[-3 * off], m7 = p[3 * off], m0 = p[-4 * off];
  if (sw) {
    p[-off]     = (spx)clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    p[0]        = (spx)clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    p[-2 * off] = (spx)clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    p[off]      = (spx)clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    p[-3 * off] = (spx)clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    p[2 * off]  = (spx)clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
  } else {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (iabs(delta) < thr_cut) {
      const int tc2 = tc >> 1;
      delta = clip3(-tc, tc, delta);
      p[-off] = (spx)clip8(m3 + delta);
      p[0] = (spx)clip8(m4 - delta);
      if (second_p) p[-2 * off] = (spx)clip8(m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
      if (second_q) p[off] = (spx)clip8(m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
    }
  }
}

static int use_strong(const spx *p, int off, int d, int beta, int tc)
{ /* xUseStrongFiltering :933-943 */
  const int m4 = p[0], m3 = p[-off], m7 = p[3 * off], m0 = p[-4 * off];
  return (iabs(m0 - m3) + iabs(m7 - m4) < (beta >> 3)) && (d < (beta >> 2)) && (iabs(m3 - m4) < ((tc * 5 + 1) >> 1));
}
static int calc_dp(const spx *p, int off) { return iabs(p[-3 * off] - 2 * p[-2 * off] + p[-off]); }
static int calc_dq(const spx *p, int off) { return iabs(p[0] - 2 * p[off] + p[2 * off]); }

int hm_oracle_deblock_frame(uint8_t *frame, int width, int height, int qp, const hm_ctu_record *recs)
{
  const size_t n = (size_t)width * height * 3 / 2;
  size_t i; int rc;
  uint16_t *t;
  if (!frame || width <= 0 || height <= 0) return -1;
  t = (uint16_t *)malloc(n * sizeof *t);
  if (!t) return -2;
  for (i = 0; i < n; i++) t[i] = frame[i];
  rc = hm_oracle_deblock_frame16(t, width, height, qp, recs, 8);
  for (i = 0; i < n; i++) frame[i] = (uint8_t)t[i];
  free(t);
  return rc;
}

int hm_oracle_deblock_frame16(uint16_t *frame, int width, int height, int qp, const hm_ctu_record *recs, int bit_depth)
{
  return hm_oracle_deblock_frame16_tb(frame, width, height, qp, recs, bit_depth, 1, 1, NULL, NULL);
}

/* col_bd / row_bd != NULL: LFCrossTileBoundaryFlag 0 -- edges on the tile borders (CTU units) are left alone (xSetLoopfilterParam
   TComLoopFilter.cpp:362-400: the neighbouring CU of another tile does not exist for the filter) */
static int on_tile_border(int pos, int n_tiles, const int *bd)
{
  int t;
  if (!bd || (pos & 63)) return 0;
  for (t = 1; t < n_tiles; t++) if (bd[t] == (pos >> 6)) return 1;
  return 0;
}

/* LoopFilterBetaOffset_div2 / LoopFilterTcOffset_div2 of the cfg (slice_beta_offset_div2 / slice_tc_offset_div2, TComLoopFilter.cpp:584-585, 701) */
static int g_beta_offset_div2 = 0, g_tc_offset_div2 = 0;
void hm_oracle_set_deblock_offsets(int beta_offset_div2, int tc_offset_div2) { g_beta_offset_div2 = beta_offset_div2; g_tc_offset_div2 = tc_offset_div2; }

int hm_oracle_deblock_frame16_tb(uint16_t *frame, int width, int height, int qp, const hm_ctu_record *recs, int bit_depth,
                                 int tile_cols, int tile_rows, const int *col_bd, const int *row_bd)
{
  const int ctus_x = (width + 63) >> 6, cw = width >> 1, ch = height >> 1;
  spx *Y = frame, *C[2] = { frame + (size_t)width * height, frame + (size_t)width * height + (size_t)cw * ch };
  int dir, x, y, i, c;
  if (!frame || !recs || width <= 0 || height <= 0 || (width & 7) || (height & 7) || qp < 0 || qp > 51 || (bit_depth != 8 && bit_depth != 10)) return -1;
  g_pel_max = (1 << bit_depth) - 1;
  {
    const int bd_scale = 1 << (bit_depth - 8);                                    /* iBitdepthScale :596, 770 */
    const int tc = tc_table[clip3(0, 53, qp + 2 + 2 * g_tc_offset_div2)] * bd_scale, beta = beta_table[clip3(0, 51, qp + 2 * g_beta_offset_div2)] * bd_scale;   /* :623-627, Bs 2 */
    const int side_thr = (beta + (beta >> 1)) >> 3, thr_cut = tc * 10;
    const int qpc = g_chroma_scale_420[clip3(0, 57, qp)];                         /* :782-797, cQpOffset 0 */
    const int tc_c = tc_table[clip3(0, 53, qpc + 2 + 2 * g_tc_offset_div2)] * bd_scale;                                   /* :804 */
    for (dir = 0; dir < 2; dir++) {                                               /* 0: vertical edges (filter across x), 1: horizontal */
      /* luma: 4-sample segments of every edge on the 8x8 grid */
      for (y = 0; y < height; y += dir ? 8 : 4)
        for (x = 0; x < width; x += dir ? 4 : 8) {
          spx *src; int off, step, dp0, dq0, dp3, dq3, d0, d3, d;
          if (!edge_flag(recs, ctus_x, x, y, dir) || on_tile_border(dir ? y : x, dir ? tile_rows : tile_cols, dir ? row_bd : col_bd)) continue;
          src = Y + (size_t)y * width + x; off = dir ? width : 1; step = dir ? 1 : width;
          dp0 = calc_dp(src, off); dq0 = calc_dq(src, off); dp3 = calc_dp(src + 3 * step, off); dq3 = calc_dq(src + 3 * step, off);
          d0 = dp0 + dq0; d3 = dp3 + dq3; d = d0 + d3;
          if (d < beta) {
            const int fp = (dp0 + dp3) < side_thr, fq = (dq0 + dq3) < side_thr;
            const int sw = use_strong(src, off, 2 * d0, beta, tc) && use_strong(src + 3 * step, off, 2 * d3, beta, tc);
            for (i = 0; i < 4; i++) pel_filter_luma(src + i * step, off, tc, sw, thr_cut, fp, fq);
          }
        }
      /* chroma: edges on the chroma 8x8 grid (luma 16), 2 chroma lines per luma partition, Bs 2 everywhere (:746) */
      for (y = 0; y < height; y += dir ? 16 : 4)
        for (x = 0; x < width; x += dir ? 4 : 16) {
          if (!edge_flag(recs, ctus_x, x, y, dir) || on_tile_border(dir ? y : x, dir ? tile_rows : tile_cols, dir ? row_bd : col_bd)) continue;
          for (c = 0; c < 2; c++) {
            spx *src = C[c] + (size_t)(y >> 1) * cw + (x >> 1);
            const int off = dir ? cw : 1, step = dir ? 1 : cw;
            for (i = 0; i < 2; i++) { /* xPelFilterChroma :901-925 */
              spx *p = src + i * step;
              const int m4 = p[0], m3 = p[-off], m5 = p[off], m2 = p[-2 * off];
              const int delta = clip3(-tc_c, tc_c, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
              p[-off] = (spx)clip8(m3 + delta); p[0] = (spx)clip8(m4 - delta);
            }
          }
        }
    }
  }
  return 0;
}
