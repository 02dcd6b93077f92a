/* oracle/hm_sao.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the reference's sample adaptive offset encoder for the configuration of the hot path
 * (8- or 10-bit 4:2:0, one slice, all-intra => temporal layer 0 => SAO always enabled at picture level, SAOLcuBoundary 0,
 * TestSAODisableAtPictureLevel 0, offset step log2 0):
 *   statistics          TEncSampleAdaptiveOffset::getStatistics / getBlkStats   TEncSampleAdaptiveOffset.cpp:295-341, 943-1335
 *   offsets             deriveOffsets / estIterOffset / getDistortion           :421-615
 *   CTU mode decision   deriveModeNewRDO / deriveModeMergeRDO / decideBlkParams :617-941
 *   rate                TEncSbac::codeSAOBlkParam / codeSAOOffsetParam ...      TEncSbac.cpp:1543-1720 (counter coder)
 *   reconstruction      TComSampleAdaptiveOffset::offsetBlock / offsetCTU       TComSampleAdaptiveOffset.cpp:316-620
 * Input: original picture + deblocked picture; output: per-CTU SAO parameters + the final reconstruction.
 * Pinned by tests/golden/rd_*.npz:recon_filtered (the reference's output picture) and :bitstream_sao.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "hm_oracle.h"
#include "hm_tables.h"

enum { EO_0 = 0, EO_90, EO_135, EO_45, BO, NTYPES };
enum { MODE_OFF = 0, MODE_NEW, MODE_MERGE };
enum { MERGE_LEFT = 0, MERGE_ABOVE };

typedef struct { int64_t diff[32], count[32]; } stat_t;
typedef struct { uint8_t merge_ctx, type_ctx; uint64_t frac; } sbac_t;

static int sgn(int v) { return (v > 0) - (v < 0); }
typedef uint16_t spx;                 /* 16-bit sample planes inside; the 8-bit entries widen and narrow */
static int g_bd = 8;                  /* sample bit depth of the run (8 or 10); one run at a time per process */
#define MAX_OFF ((1 << ((g_bd < 10 ? g_bd : 10) - 5)) - 1)      /* getMaxOffsetQVal, TComSampleAdaptiveOffset.h:73: 7 / 31 */
#define DIST_SHIFT (2 * (g_bd - 8))                            /* 2 * DISTORTION_PRECISION_ADJUSTMENT(bitDepth - 8), :501 */
static int clip8(int v) { const int mx = (1 << g_bd) - 1; return v < 0 ? 0 : (v > mx ? mx : v); }

/* ---- statistics: every sample of the CTU whose neighbours exist and that lies outside the not-yet-deblocked margin ---- */
static void blk_stats(stat_t st[NTYPES], const spx *src, const spx *org, int stride, int width, int height,
                      int left, int right, int above, int below, int skip_r, int skip_b)
{
  int t, x, y;
  memset(st, 0, sizeof(stat_t) * NTYPES);
  for (t = 0; t < NTYPES; t++) {
    const int need_lr = (t == EO_0 || t == EO_135 || t == EO_45), need_ab = (t == EO_90 || t == EO_135 || t == EO_45);
    const int sx = need_lr ? (left ? 0 : 1) : 0, ex = right ? width - skip_r : (need_lr ? width - 1 : width);
    const int sy = need_ab ? (above ? 0 : 1) : 0, ey = below ? height - skip_b : (need_ab ? height - 1 : height);
    for (y = sy; y < ey; y++)
      for (x = sx; x < ex; x++) {
        const spx *p = src + (size_t)y * stride + x;
        int cls;
        switch (t) {
          case EO_0:   cls = 2 + sgn(p[0] - p[-1]) + sgn(p[0] - p[1]); break;
          case EO_90:  cls = 2 + sgn(p[0] - p[-stride]) + sgn(p[0] - p[stride]); break;
          case EO_135: cls = 2 + sgn(p[0] - p[-stride - 1]) + sgn(p[0] - p[stride + 1]); break;
          case EO_45:  cls = 2 + sgn(p[0] - p[-stride + 1]) + sgn(p[0] - p[stride - 1]); break;
          default:     cls = p[0] >> (g_bd - 5); break;
        }
        st[t].diff[cls] += (int)org[(size_t)y * stride + x] - (int)p[0];
        st[t].count[cls]++;
      }
  }
}

/* ---- rate (counter coder) ---- */
static void sb_bin(sbac_t *c, uint8_t *ctx, int bin)
{
  const uint8_t s = *ctx;
  c->frac += (uint64_t)g_entropy_bits[s ^ bin];
  *ctx = bin == (s & 1) ? g_next_mps[s] : g_next_lps[s];
}
static void sb_ep(sbac_t *c, int n) { c->frac += (uint64_t)32768 * (uint64_t)n; }
static uint32_t sb_bits(const sbac_t *c) { return (uint32_t)(c->frac >> 15); }
static void sb_reset(sbac_t *c) { c->frac &= 32767; }

static void code_offset_param(sbac_t *c, int comp, const hm_sao_offset *p)
{ /* codeSAOOffsetParam TEncSbac.cpp:1605-1681 (slice enabled) */
  const int first = comp != 2;
  int i;
  if (first) {
    const int sym = p->mode == MODE_OFF ? 0 : (p->type == BO ? 1 : 2);
    if (sym == 0) sb_bin(c, &c->type_ctx, 0); else { sb_bin(c, &c->type_ctx, 1); sb_ep(c, 1); }
  }
  if (p->mode == MODE_NEW) {
    int off[4], k = 0;
    const int ncls = p->type == BO ? 4 : 5;
    for (i = 0; i < ncls; i++) { if (p->type != BO && i == 2) continue; off[k++] = p->offset[p->type == BO ? (p->aux + i) % 32 : i]; }
    for (i = 0; i < 4; i++) { const int a = abs(off[i]); sb_ep(c, a == 0 ? 1 : (a < MAX_OFF ? a + 1 : a)); }       /* codeSaoMaxUvlc */
    if (p->type == BO) { for (i = 0; i < 4; i++) if (off[i]) sb_ep(c, 1); sb_ep(c, 5); }
    else if (first) sb_ep(c, 2);
  }
}
static void code_blk_param(sbac_t *c, const hm_sao_blk *b, int left_avail, int above_avail, int only_merge)
{ /* codeSAOBlkParam TEncSbac.cpp:1683-1720 */
  int is_left = 0, is_above = 0, comp;
  if (left_avail) { is_left = b->c[0].mode == MODE_MERGE && b->c[0].type == MERGE_LEFT; sb_bin(c, &c->merge_ctx, is_left); }
  if (above_avail && !is_left) { is_above = b->c[0].mode == MODE_MERGE && b->c[0].type == MERGE_ABOVE; sb_bin(c, &c->merge_ctx, is_above); }
  if (only_merge) return;
  if (!is_left && !is_above) for (comp = 0; comp < 3; comp++) code_offset_param(c, comp, &b->c[comp]);
}

/* ---- offsets ---- */
static int64_t est_dist(int64_t count, int64_t offset, int64_t diff) { return (count * offset * offset - diff * offset * 2) >> DIST_SHIFT; }   /* estSaoDist :459 */
static int est_iter_offset(int type, double lambda, int offset_in, int64_t count, int64_t diff, int64_t *best_dist, double *best_cost)
{ /* estIterOffset :465-496, offsetTh 7, bitIncrease 0 */
  int it = offset_in, out = 0;
  double min_cost = lambda;
  while (it != 0) {
    int64_t rate = type == BO ? abs(it) + 2 : abs(it) + 1, dist;
    double cost;
    if (abs(it) == MAX_OFF) rate--;
    dist = est_dist(count, it, diff);
    cost = (double)dist + lambda * (double)rate;
    if (cost < min_cost) { min_cost = cost; out = it; *best_dist = dist; *best_cost = cost; }
    it = it > 0 ? it - 1 : it + 1;
  }
  return out;
}
static void derive_offsets(int type, double lambda, const stat_t *st, int *q, int *aux)
{ /* deriveOffsets :498-615 */
  int cls;
  const int ncls = type == BO ? 32 : 5;
  memset(q, 0, sizeof(int) * 32);
  for (cls = 0; cls < ncls; cls++) {
    double x;
    if (type != BO && cls == 2) continue;
    if (st->count[cls] == 0) continue;
    x = (double)(st->diff[cls] * (1 << (g_bd - 8))) / (double)st->count[cls];          /* :520-523, offset step log2 0 */
    if (g_bd > 8) { const int r = 1 << (g_bd - 8); q[cls] = x > 0 ? ((int)x + (r >> 1)) / r : ((int)x - (r >> 1)) / r; }   /* xRoundIbdi2 :49-52 */
    else q[cls] = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);                  /* xRoundIbdi :54-57 */
    q[cls] = q[cls] < -MAX_OFF ? -MAX_OFF : (q[cls] > MAX_OFF ? MAX_OFF : q[cls]);
  }
  if (type != BO) {
    for (cls = 0; cls < 5; cls++) {
      int64_t d; double c;
      if ((cls == 0 || cls == 1) && q[cls] < 0) q[cls] = 0;
      if ((cls == 3 || cls == 4) && q[cls] > 0) q[cls] = 0;
      if (q[cls] != 0) q[cls] = est_iter_offset(type, lambda, q[cls], st->count[cls], st->diff[cls], &d, &c);
    }
    *aux = 0;
  } else {
    int64_t dist[32]; double cost[32], min_cost = 1.7e308; int band, keep[32], i;
    memset(dist, 0, sizeof dist);
    for (cls = 0; cls < 32; cls++) {
      cost[cls] = lambda;
      if (q[cls] != 0) q[cls] = est_iter_offset(type, lambda, q[cls], st->count[cls], st->diff[cls], &dist[cls], &cost[cls]);
    }
    for (band = 0; band < 32 - 4 + 1; band++) {
      double c = cost[band]; c += cost[band + 1]; c += cost[band + 2]; c += cost[band + 3];
      if (c < min_cost) { min_cost = c; *aux = band; }
    }
    memset(keep, 0, sizeof keep);
    for (i = 0; i < 4; i++) keep[(*aux + i) % 32] = q[(*aux + i) % 32];
    memcpy(q, keep, sizeof keep);
  }
}
static int64_t get_dist(int type, int aux, const int *off, const stat_t *st)
{ /* getDistortion :421-457 */
  int64_t d = 0; int i;
  if (type != BO) for (i = 0; i < 5; i++) d += est_dist(st->count[i], off[i], st->diff[i]);
  else for (i = aux; i < aux + 4; i++) d += est_dist(st->count[i % 32], off[i % 32], st->diff[i % 32]);
  return d;
}

static void mode_new(const stat_t st[3][NTYPES], const double lambda[3], const hm_sao_blk *ml[2], hm_sao_blk *out, double *norm_cost,
                     const sbac_t *cur, sbac_t *go)
{ /* deriveModeNewRDO :617-758 */
  sbac_t mid, temp;
  hm_sao_offset test[3];
  int64_t dist[3], mode_dist[3] = { 0, 0, 0 };
  double min_cost, cost;
  int type, comp;
  memset(out, 0, sizeof *out);
  *go = *cur;
  code_blk_param(go, out, ml[MERGE_LEFT] != NULL, ml[MERGE_ABOVE] != NULL, 1);
  mid = *go;
  /* luma */
  out->c[0].mode = MODE_OFF;
  sb_reset(go); code_offset_param(go, 0, &out->c[0]);
  min_cost = lambda[0] * (double)sb_bits(go);
  temp = *go;
  for (type = 0; type < NTYPES; type++) {
    int rate;
    memset(&test[0], 0, sizeof test[0]); test[0].mode = MODE_NEW; test[0].type = type;
    derive_offsets(type, lambda[0], &st[0][type], test[0].offset, &test[0].aux);
    dist[0] = get_dist(type, test[0].aux, test[0].offset, &st[0][type]);
    *go = mid; sb_reset(go); code_offset_param(go, 0, &test[0]);
    rate = (int)sb_bits(go);
    cost = (double)dist[0] + lambda[0] * (double)rate;
    if (cost < min_cost) { min_cost = cost; mode_dist[0] = dist[0]; out->c[0] = test[0]; temp = *go; }
  }
  *go = temp; mid = *go;
  /* chroma: Cb and Cr share the type */
  cost = 0; sb_reset(go);
  { uint32_t prev = 0;
    for (comp = 1; comp < 3; comp++) { uint32_t b; out->c[comp].mode = MODE_OFF; mode_dist[comp] = 0; code_offset_param(go, comp, &out->c[comp]); b = sb_bits(go); cost += lambda[comp] * (double)(b - prev); prev = b; } }
  min_cost = cost;
  for (type = 0; type < NTYPES; type++) {
    uint32_t prev = 0;
    *go = mid; sb_reset(go); cost = 0;
    for (comp = 1; comp < 3; comp++) {
      uint32_t b;
      memset(&test[comp], 0, sizeof test[comp]); test[comp].mode = MODE_NEW; test[comp].type = type;
      derive_offsets(type, lambda[comp], &st[comp][type], test[comp].offset, &test[comp].aux);
      dist[comp] = get_dist(type, test[comp].aux, test[comp].offset, &st[comp][type]);
      code_offset_param(go, comp, &test[comp]);
      b = sb_bits(go);
      cost += (double)dist[comp] + (lambda[comp] * (double)(b - prev));
      prev = b;
    }
    if (cost < min_cost) { min_cost = cost; for (comp = 1; comp < 3; comp++) { mode_dist[comp] = dist[comp]; out->c[comp] = test[comp]; } }
  }
  *norm_cost = 0;
  for (comp = 0; comp < 3; comp++) *norm_cost += (double)mode_dist[comp] / lambda[comp];
  *go = *cur; sb_reset(go);
  code_blk_param(go, out, ml[MERGE_LEFT] != NULL, ml[MERGE_ABOVE] != NULL, 0);
  *norm_cost += (double)sb_bits(go);
}

static void mode_merge(const stat_t st[3][NTYPES], const double lambda[3], const hm_sao_blk *ml[2], hm_sao_blk *out, double *norm_cost,
                       const sbac_t *cur, sbac_t *go)
{ /* deriveModeMergeRDO :760-812 */
  sbac_t temp = *cur;
  int mt, comp;
  *norm_cost = 1.7e308;
  for (mt = 0; mt < 2; mt++) {
    hm_sao_blk test; double nd = 0, cost; int rate;
    if (!ml[mt]) continue;
    test = *ml[mt];
    for (comp = 0; comp < 3; comp++) {
      const hm_sao_offset *m = &ml[mt]->c[comp];
      test.c[comp].mode = MODE_MERGE; test.c[comp].type = mt;
      if (m->mode != MODE_OFF) nd += ((double)get_dist(m->type, m->aux, m->offset, &st[comp][m->type])) / lambda[comp];
    }
    *go = *cur; sb_reset(go);
    code_blk_param(go, &test, ml[MERGE_LEFT] != NULL, ml[MERGE_ABOVE] != NULL, 0);
    rate = (int)sb_bits(go);
    cost = nd + (double)rate;
    if (cost < *norm_cost) { *norm_cost = cost; *out = test; temp = *go; }
  }
  *go = temp;
}

/* ---- reconstruction of one CTU component: samples whose neighbours lie outside the picture are left alone ---- */
static void offset_block(int type, const int *offset, const spx *src, spx *res, int stride, int width, int height,
                         int left, int right, int above, int below)
{
  int x, y;
  const int need_lr = (type == EO_0 || type == EO_135 || type == EO_45), need_ab = (type == EO_90 || type == EO_135 || type == EO_45);
  const int sx = (need_lr && !left) ? 1 : 0, ex = (need_lr && !right) ? width - 1 : width;
  const int sy = (need_ab && !above) ? 1 : 0, ey = (need_ab && !below) ? height - 1 : height;
  for (y = sy; y < ey; y++)
    for (x = sx; x < ex; x++) {
      const spx *p = src + (size_t)y * stride + x;
      int cls;
      switch (type) {
        case EO_0:   cls = 2 + sgn(p[0] - p[-1]) + sgn(p[0] - p[1]); break;
        case EO_90:  cls = 2 + sgn(p[0] - p[-stride]) + sgn(p[0] - p[stride]); break;
        case EO_135: cls = 2 + sgn(p[0] - p[-stride - 1]) + sgn(p[0] - p[stride + 1]); break;
        case EO_45:  cls = 2 + sgn(p[0] - p[-stride + 1]) + sgn(p[0] - p[stride - 1]); break;
        default:     cls = p[0] >> (g_bd - 5); break;
      }
      res[(size_t)y * stride + x] = (spx)clip8(p[0] + offset[cls]);
    }
}

int hm_oracle_sao_frame(const uint8_t *org, const uint8_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint8_t *out)
{
  return hm_oracle_sao_frame_tiles(org, deblocked, width, height, qp, params, out, 1, 1);
}

static int g_lf_across_tiles = 1;
void hm_oracle_sao_set_lf_across_tiles(int flag) { g_lf_across_tiles = flag != 0; }

/* first CTU column / row of a tile */
static int tile_start(int pos, const int *bd, int n_tiles)
{
  int t;
  for (t = 0; t < n_tiles; t++) if (bd[t] == pos) return 1;
  return 0;
}

int hm_oracle_sao_frame_tiles(const uint8_t *org, const uint8_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint8_t *out, int tile_cols, int tile_rows)
{
  const size_t n = (size_t)width * height * 3 / 2;
  size_t i; int rc;
  uint16_t *t;
  if (!org || !deblocked || !out || width <= 0 || height <= 0) return -1;
  t = (uint16_t *)malloc(3 * n * sizeof *t);
  if (!t) return -2;
  for (i = 0; i < n; i++) { t[i] = org[i]; t[n + i] = deblocked[i]; }
  rc = hm_oracle_sao_frame16(t, t + n, width, height, qp, params, t + 2 * n, tile_cols, tile_rows, 8);
  if (rc == 0) for (i = 0; i < n; i++) out[i] = (uint8_t)t[2 * n + i];
  free(t);
  return rc;
}

int hm_oracle_sao_frame16(const uint16_t *org, const uint16_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint16_t *out, int tile_cols, int tile_rows, int bit_depth)
{ /* uniform spacing */
  int col_bd[64], row_bd[64], t;
  if (tile_cols < 1 || tile_rows < 1 || tile_cols > 20 || tile_rows > 22) return -1;
  for (t = 0; t <= tile_cols; t++) col_bd[t] = (t * ((width + 63) >> 6)) / tile_cols;
  for (t = 0; t <= tile_rows; t++) row_bd[t] = (t * ((height + 63) >> 6)) / tile_rows;
  return hm_oracle_sao_frame16_tb(org, deblocked, width, height, qp, params, out, tile_cols, tile_rows, col_bd, row_bd, bit_depth);
}

int hm_oracle_sao_frame16_tb(const uint16_t *org, const uint16_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint16_t *out, int tile_cols, int tile_rows, const int *col_bd, const int *row_bd, int bit_depth)
{
  const int cx = (width + 63) >> 6, cy = (height + 63) >> 6, nctu = cx * cy, cw = width >> 1, ch = height >> 1;
  const size_t ysz = (size_t)width * height, csz = (size_t)cw * ch;
  double lambda[3];
  stat_t (*st)[3][NTYPES];
  hm_sao_blk *recon;
  sbac_t go, cur, next;
  int a, comp, i;
  if (!org || !deblocked || !params || !out || width <= 0 || height <= 0 || (width & 7) || (height & 7) || qp < 0 || qp > 51 || (bit_depth != 8 && bit_depth != 10)) return -1;
  g_bd = bit_depth;
  { /* slice lambdas, TEncSlice.cpp:112-140 */
    const int qpc = g_chroma_scale_420[qp];
    lambda[0] = 0.57 * 1.0 * pow(2.0, (qp - 12) / 3.0);
    lambda[1] = lambda[2] = lambda[0] / pow(2.0, (qp - qpc) / 3.0);
  }
  st = malloc(sizeof(*st) * nctu); recon = malloc(sizeof(hm_sao_blk) * nctu);
  if (!st || !recon) { free(st); free(recon); return -2; }
  for (a = 0; a < nctu; a++) { /* getStatistics :295-341: only picture borders count; 5/4 (luma) and 3/2 (chroma) columns/rows next to a right/lower CTU are skipped */
    const int x0 = (a % cx) * 64, y0 = (a / cx) * 64;
    const int w = x0 + 64 > width ? width - x0 : 64, h = y0 + 64 > height ? height - y0 : 64;
    int left = x0 > 0, above = y0 > 0;
    const int right = x0 + 64 < width, below = y0 + 64 < height;    /* :322-330: for the statistics right / below only look at the picture */
    if (!g_lf_across_tiles) { /* deriveLoopFilterBoundaryAvailibility: a CTU of another tile is missing, like the picture border */
      if (tile_start(a % cx, col_bd, tile_cols)) left = 0;
      if (tile_start(a / cx, row_bd, tile_rows)) above = 0;
    }
    for (comp = 0; comp < 3; comp++) {
      const int sh = comp ? 1 : 0, stride = comp ? cw : width;
      const size_t off = (comp == 0 ? 0 : (comp == 1 ? ysz : ysz + csz)) + (size_t)(y0 >> sh) * stride + (x0 >> sh);
      blk_stats(st[a][comp], deblocked + off, org + off, stride, w >> sh, h >> sh, left, right, above, below, comp ? 3 : 5, comp ? 2 : 4);
    }
  }
  memcpy(out, deblocked, (ysz + 2 * csz) * sizeof(spx));
  { /* initRDOCabacCoder: I-slice contexts (ContextTables.h:445-458: merge 153, type 200) at the slice QP */
    const int init[2] = { 153, 200 };
    uint8_t *dst[2] = { &go.merge_ctx, &go.type_ctx };
    for (i = 0; i < 2; i++) {
      const int v = init[i], slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
      int s = ((slope * qp) >> 4) + offset, mps; s = s < 1 ? 1 : (s > 126 ? 126 : s); mps = s >= 64;
      *dst[i] = (uint8_t)(((mps ? s - 64 : 63 - s) << 1) + mps);
    }
    go.frac = 0;
  }
  for (a = 0; a < nctu; a++) { /* decideBlkParams :814-941 */
    const hm_sao_blk *ml[2] = { NULL, NULL };
    hm_sao_blk mode; double min_cost = 1.7e308, cost;
    const int x0 = (a % cx) * 64, y0 = (a / cx) * 64;
    const int w = x0 + 64 > width ? width - x0 : 64, h = y0 + 64 > height ? height - y0 : 64;
    cur = go;
    /* merge candidates come from the same tile only (TComPic::getSAOMergeAvailability; statistics and offsets do cross tiles:
       LFCrossTileBoundaryFlag 1) */
    if (!tile_start(a % cx, col_bd, tile_cols)) ml[MERGE_LEFT] = &recon[a - 1];
    if (!tile_start(a / cx, row_bd, tile_rows)) ml[MERGE_ABOVE] = &recon[a - cx];
    mode_new((const stat_t (*)[NTYPES])st[a], lambda, ml, &mode, &cost, &cur, &go);
    if (cost < min_cost) { min_cost = cost; params[a] = mode; next = go; }
    mode_merge((const stat_t (*)[NTYPES])st[a], lambda, ml, &mode, &cost, &cur, &go);
    if (cost < min_cost) { min_cost = cost; params[a] = mode; next = go; }
    go = next;
    recon[a] = params[a];
    for (comp = 0; comp < 3; comp++) { /* reconstructBlkSAOParam: offset step 1 => new offsets as coded; merge copies the neighbour */
      hm_sao_offset *p = &recon[a].c[comp];
      if (p->mode == MODE_MERGE) *p = ml[p->type]->c[comp];
    }
    for (comp = 0; comp < 3; comp++) { /* offsetCTU :568-620 */
      const hm_sao_offset *p = &recon[a].c[comp];
      const int sh = comp ? 1 : 0, stride = comp ? cw : width;
      const size_t off = (comp == 0 ? 0 : (comp == 1 ? ysz : ysz + csz)) + (size_t)(y0 >> sh) * stride + (x0 >> sh);
      int o[32];
      if (p->mode == MODE_OFF) continue;
      if (p->type == BO) memcpy(o, p->offset, sizeof o); else { memset(o, 0, sizeof o); memcpy(o, p->offset, sizeof(int) * 5); }
      { int left = x0 > 0, above = y0 > 0, right = x0 + 64 < width, below = y0 + 64 < height;
        if (!g_lf_across_tiles) { /* offsetCTU: all neighbours tile-aware */
          if (tile_start(a % cx, col_bd, tile_cols)) left = 0;
          if (tile_start(a / cx, row_bd, tile_rows)) above = 0;
          if (a % cx + 1 < cx && tile_start(a % cx + 1, col_bd, tile_cols)) right = 0;
          if (a / cx + 1 < cy && tile_start(a / cx + 1, row_bd, tile_rows)) below = 0;
        }
        offset_block(p->type, o, deblocked + off, out + off, stride, w >> sh, h >> sh, left, right, above, below); }
    }
  }
  free(st); free(recon);
  return 0;
}
