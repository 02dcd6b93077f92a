/* oracle/hm_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the reference's depth-pruned all-intra CTU decision path.
 * Flat arrays instead of TComDataCU objects, one explicit CABAC-estimator state struct,
 * table-driven transforms.  Each function cites the reference lines it restates
 * (paths relative to /root/reference/HM_dl/source/Lib).  Build: gcc -O2 -ffp-contract=off.
 * Fixed configuration = /root/reference/encoder_intra_main.cfg (CTU 64, 4 depths, TU 4..32,
 * intra TU depth 3, RDOQ, RDOQTS, TransformSkip + Fast, SignHide, StrongIntraSmoothing,
 * FastUDIUseMPM, 8- or 10-bit 4:2:0, one slice, optionally tiles, optionally wavefront rows (WaveFrontSynchro)).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include "hm_oracle.h"
#include "hm_tables.h"

typedef int16_t pel;
#define MAX_DOUBLE 1.7e+308
#define CTU 64
#define PLANAR 0
#define DC 1
#define HOR 10
#define VER 26
#define DM_CHROMA 36
#define SIZE_2Nx2N 0
#define SIZE_NxN 3
#define SIZE_NONE 8    /* NUMBER_OF_PART_SIZES (TypeDef.h PartSize enum) */
#define SCAN_DIAG 0
#define SCAN_HOR 1
#define SCAN_VER 2

/* =====================================================================================
 * static tables built once
 * ===================================================================================== */
static int      g_init_done = 0;
static uint8_t  g_r2z[256], g_z2r[256];              /* TComRom.cpp:284-352 */
static uint16_t g_scan[3][4][1024];                  /* grouped-4x4 scans, [type][log2-2] TComRom.cpp:179-260 */
static uint8_t  g_scan_cg[3][4][64];                 /* ungrouped scan of the CG grid, [type][log2(grid)] */
static int16_t  g_dct[4][32][32];                    /* T_N[k][n], N = 4<<i */
static uint8_t  g_next_state[128][2];                /* ContextModel.cpp:114-127 */
static FILE    *g_trace = NULL;
static FILE    *g_stage = NULL;                      /* stage trace (F-rd-3): the events HM prints under DEBUG_INTRA_SEARCH_COSTS / DEBUG_TRANSFORM_AND_QUANTISE */

void hm_oracle_set_stage_trace(const char *path)
{
  if (g_stage) { fclose(g_stage); g_stage = NULL; }
  if (path) g_stage = fopen(path, "w");
}

static void stage_block32(const int32_t *v, int n) { for (int i = 0; i < n * n; i++) fprintf(g_stage, "%d ", v[i]); fputc('\n', g_stage); }

void hm_oracle_set_trace(const char *path)
{
  if (g_trace) { fclose(g_trace); g_trace = NULL; }
  if (path) g_trace = fopen(path, "w");
}

static void scan_next(int type, int bw, int bh, int *line, int *col)
{ /* ScanGenerator::GetNextIndex, TComRom.cpp:100-160 */
  if (type == SCAN_DIAG) {
    if (*col == bw - 1 || *line == 0) {
      *line += *col + 1; *col = 0;
      if (*line >= bh) { *col += *line - (bh - 1); *line = bh - 1; }
    } else { (*col)++; (*line)--; }
  } else if (type == SCAN_HOR) {
    if (*col == bw - 1) { (*line)++; *col = 0; } else (*col)++;
  } else {
    if (*line == bh - 1) { (*col)++; *line = 0; } else (*line)++;
  }
}

static void init_tables(void)
{
  if (g_init_done) return;
  for (int r = 0; r < 256; r++) {
    int x = r & 15, y = r >> 4, z = 0;
    for (int b = 0; b < 4; b++) z |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1));
    g_r2z[r] = (uint8_t)z; g_z2r[z] = (uint8_t)r;
  }
  for (int t = 0; t < 3; t++) {
    for (int l = 0; l < 4; l++) {             /* block 4<<l */
      int bw = 4 << l, wg = bw >> 2, ng = wg * wg;
      int gl = 0, gc = 0;
      for (int g = 0; g < ng; g++) {
        int l2 = 0, c2 = 0;
        for (int p = 0; p < 16; p++) {
          g_scan[t][l][g * 16 + p] = (uint16_t)((l2 + gl * 4) * bw + c2 + gc * 4);
          scan_next(t, 4, 4, &l2, &c2);
        }
        scan_next(t, wg, wg, &gl, &gc);
      }
      int ul = 0, uc = 0;                      /* ungrouped scan of a wg x wg grid */
      for (int g = 0; g < ng; g++) { g_scan_cg[t][l][g] = (uint8_t)(ul * wg + uc); scan_next(t, wg, wg, &ul, &uc); }
    }
  }
  for (int i = 0; i < 4; i++) {
    int N = 4 << i, step = 32 / N;
    for (int k = 0; k < N; k++)
      for (int n = 0; n < N; n++) {
        int m = ((2 * n + 1) * k * step) & 127;
        if (m > 64) m = 128 - m;
        g_dct[i][k][n] = (int16_t)(m <= 32 ? g_dct_mag[m] : -g_dct_mag[64 - m]);
      }
  }
  for (int s = 0; s < 128; s++)
    for (int b = 0; b < 2; b++) g_next_state[s][b] = ((s & 1) == b) ? g_next_mps[s] : g_next_lps[s];
  g_init_done = 1;
}

/* =====================================================================================
 * CABAC in estimation mode (TEncBinCoderCABACCounter.cpp:60-140, TEncBinCoderCABAC.cpp:148-173)
 * ===================================================================================== */
typedef struct { uint8_t ctx[NUM_CTX + 1]; uint64_t frac; } cabac_t;

static void cabac_init(cabac_t *c, int qp)
{ /* ContextModel.cpp:56-66 ; TEncSbac.cpp:105-156 ; TEncBinCoderCABAC.cpp:69-79 */
  if (qp < 0) qp = 0; if (qp > 51) qp = 51;
  for (int i = 0; i < NUM_CTX; i++) {
    int v = g_ctx_init[i];
    int slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
    int s = ((slope * qp) >> 4) + offset;
    if (s < 1) s = 1; if (s > 126) s = 126;
    int mps = s >= 64;
    c->ctx[i] = (uint8_t)(((mps ? s - 64 : 63 - s) << 1) + mps);
  }
  c->frac = 0;
}
static inline void enc_bin(cabac_t *c, int ctx, int bin)
{
  uint8_t s = c->ctx[ctx];
  c->frac += (uint64_t)g_entropy_bits[s ^ bin];
  c->ctx[ctx] = g_next_state[s][bin];
}
static inline void enc_ep(cabac_t *c, int n) { c->frac += (uint64_t)32768 * (uint64_t)n; }
static inline void enc_trm(cabac_t *c, int bin) { c->frac += (uint64_t)g_entropy_bits[126 ^ bin]; }
static inline void reset_bits(cabac_t *c) { c->frac &= 32767; }
static inline uint32_t get_bits(const cabac_t *c) { return (uint32_t)(c->frac >> 15); }
static inline int ctx_bits(const cabac_t *c, int ctx, int bin) { return g_entropy_bits[c->ctx[ctx] ^ bin]; }

/* =====================================================================================
 * encoder state
 * ===================================================================================== */
typedef struct {
  uint8_t a[11][256];            /* 0 depth 1 part 2 lumaDir 3 chromaDir 4 trIdx 5..7 cbf 8..10 tskip */
  int32_t coef[3][4096];         /* final coefficients, z-order TU layout (TComDataCU m_pcTrCoeff) */
  uint32_t bits, dist; double cost;
} irec_t;
enum { A_DEPTH = 0, A_PART, A_LDIR, A_CDIR, A_TRIDX, A_CBF, A_TSKIP = 8 };

typedef struct { int x, y, log2, depth, zbase, nparts, part; } cu_t;
/* sample bit depth of the run (8, or 10: InternalBitDepth 10 with RExt__HIGH_BIT_DEPTH_SUPPORT 0, i.e. FULL_NBIT 0, TypeDef.h:162-172).
 * Not thread-safe, like the trace hook: one encode at a time per process. */
static int g_bd = 8;
/* tool switches of the cfg (TAppEncCfg.cpp:900-901,917-918,950,978,1007), bits as HEVCDL_TOOL_* of include/hevcdl.h: 0x01 RDOQ, 0x02 RDOQTS, 0x04 TransformSkip, 0x10 SignHideFlag,
 * 0x20 StrongIntraSmoothing, 0x40 FastUDIUseMPMEnabled can be cleared; TransformSkipFast (0x08) stays on (the reference cfg's value) */
static unsigned g_tools = 0x7fu;
void hm_oracle_set_tools(unsigned tools) { g_tools = tools; }
/* WaveFrontSynchro 1 (entropy_coding_sync_enabled_flag, TAppEncCfg.cpp:975): the coder is re-initialised at the first CTU of every CTU row of a tile and takes over
   the contexts behind the SECOND CTU of the row above when that CTU exists in the tile (TEncSlice.cpp:783-830, 925-928) */
static int g_wpp = 0;
void hm_oracle_set_wpp(int on) { g_wpp = on != 0; }
#define DIST_ADJ(x) (x)          /* DISTORTION_PRECISION_ADJUSTMENT, TypeDef.h:170 */

typedef struct { int x, y, log2, trd, zrel, nparts; } tu_t;   /* luma geometry; zrel relative to the CU */

typedef struct {
  int W, H, cw, ctus_x, ctus_y, qp;
  int qp_c;                                  /* chroma QP, TComTrQuant.cpp:71-100 */
  double lambda, sqrt_lambda, cweight, lambda_c;
  double err_scale[2][4];                    /* [chType][log2-2], TComTrQuant.cpp:3096-3126 */
  pel *org[3], *rec[3];
  irec_t *recs;
  const uint8_t *labels;                     /* labels of the current frame */
  /* per CTU */
  int cx, cy, addr;
  int tx0, ty0, tx1, ty1;                         /* luma rectangle of the current tile (TComPicSym.cpp xInitTiles) */
  irec_t *r;
  cabac_t go, curr[5], next[5], temp[5], root[5], test[5], tbest[5];
  /* CU scratch, CTU-relative addressing */
  pel pred[3][CTU * CTU], resi[3][CTU * CTU], best_rec[3][CTU * CTU];
  pel rec_l[4][3][CTU * CTU];                /* m_pcQTTempTComYuv[layer] */
  int32_t coef_l[4][3][4096];                /* m_ppcQTTempCoeff[comp][layer] */
  pel ts_pred[3][16], ts_rec[3][16];         /* m_pSharedPredTransformSkip / m_pcQTTempTransformSkipTComYuv */
  int32_t ts_coef[3][16];                    /* m_pcQTTempTUCoeff */
  uint64_t est_bits;
} enc_t;

static inline double calc_rd_cost(const enc_t *e, uint32_t bits, uint32_t dist)
{ /* TComRdCost.cpp:62-107, COST_STANDARD_LOSSY / DF_DEFAULT */
  if (g_trace) fprintf(g_trace, "%u %u\n", bits, dist);
  return (double)dist + ((double)bits * e->lambda);
}

static inline int comp_stride(int c) { return c ? 32 : 64; }
/* CTU-relative buffer offset of picture position (x,y) of component c (x,y in component samples) */
static inline int boff(const enc_t *e, int c, int x, int y)
{
  int s = c ? 32 : 64;
  return (y - e->cy * s) * s + (x - e->cx * s);
}
static inline pel *rec_at(const enc_t *e, int c, int x, int y) { return e->rec[c] + (size_t)y * (c ? e->cw : e->W) + x; }
static inline pel *org_at(const enc_t *e, int c, int x, int y) { return e->org[c] + (size_t)y * (c ? e->cw : e->W) + x; }
static inline int pic_stride(const enc_t *e, int c) { return c ? e->cw : e->W; }

/* record of the CTU containing luma 4x4 unit (x4,y4) and its z index */
static inline irec_t *rec_of(const enc_t *e, int x4, int y4, int *z)
{
  *z = g_r2z[((y4 & 15) << 4) | (x4 & 15)];
  return e->recs + (y4 >> 4) * e->ctus_x + (x4 >> 4);
}
static inline void set_parts(uint8_t *a, int z0, int n, int v) { memset(a + z0, v, (size_t)n); }

/* =====================================================================================
 * reference samples (TComPattern.cpp:119-543) -- line[0..4N]: bottom-left ... corner (2N) ... top-right
 * ===================================================================================== */
static int unit_avail(const enc_t *e, int x4, int y4, int cur_x4, int cur_y4)
{ /* TComPattern.cpp:572-749 + TComDataCU.cpp:985-1200: inside the picture and already coded
     (earlier CTU in raster order, or earlier z-order inside the current CTU). */
  if (x4 < 0 || y4 < 0 || x4 * 4 >= e->W || y4 * 4 >= e->H) return 0;
  /* another tile is never available (getPULeft/Above/AboveLeft/AboveRight/BelowLeft, bEnforceTileRestriction, TComDataCU.cpp:985-1200);
     inside one tile the coding order of CTUs is still their raster order */
  if (x4 * 4 < e->tx0 || y4 * 4 < e->ty0 || x4 * 4 >= e->tx1 || y4 * 4 >= e->ty1) return 0;
  int a = (y4 >> 4) * e->ctus_x + (x4 >> 4);
  if (a != e->addr) return a < e->addr;
  return g_r2z[((y4 & 15) << 4) | (x4 & 15)] < g_r2z[((cur_y4 & 15) << 4) | (cur_x4 & 15)];
}

static void build_refs(const enc_t *e, int c, int x, int y, int n, pel *line)
{ /* x,y,n in samples of component c */
  const int u = c ? 2 : 4;                       /* samples per availability unit */
  const int sh = c ? 1 : 2;                      /* component sample -> luma 4x4 unit */
  const int nu = n / u;                          /* units per side */
  const int x4 = x >> sh, y4 = y >> sh;
  const int total = 4 * nu + 1;
  uint8_t fl[4 * 16 + 1];
  int navail = 0;
  for (int k = 0; k < 2 * nu; k++) {             /* left + below-left, from the bottom up */
    int ty4 = y4 + (2 * nu - 1 - k);
    fl[k] = (uint8_t)unit_avail(e, x4 - 1, ty4, x4, y4); navail += fl[k];
  }
  fl[2 * nu] = (uint8_t)unit_avail(e, x4 - 1, y4 - 1, x4, y4); navail += fl[2 * nu];
  for (int k = 0; k < 2 * nu; k++) {             /* above + above-right */
    fl[2 * nu + 1 + k] = (uint8_t)unit_avail(e, x4 + k, y4 - 1, x4, y4); navail += fl[2 * nu + 1 + k];
  }
  const int dcv = 1 << (g_bd - 1);
  if (navail == 0) { for (int i = 0; i <= 4 * n; i++) line[i] = dcv; return; }
  const int st = pic_stride(e, c);
  const pel *p = e->rec[c];
  /* gather */
  for (int k = 0; k < 2 * nu; k++) if (fl[k])
    for (int i = 0; i < u; i++) { int yy = y + 2 * n - 1 - (k * u + i); line[k * u + i] = p[(size_t)yy * st + x - 1]; }
  if (fl[2 * nu]) line[2 * n] = p[(size_t)(y - 1) * st + x - 1];
  for (int k = 0; k < 2 * nu; k++) if (fl[2 * nu + 1 + k])
    for (int i = 0; i < u; i++) line[2 * n + 1 + k * u + i] = p[(size_t)(y - 1) * st + x + k * u + i];
  if (navail == total) return;
  /* substitution (TComPattern.cpp:464-526) */
  #define UNIT_START(k) ((k) < 2 * nu ? (k) * u : ((k) == 2 * nu ? 2 * n : 2 * n + 1 + ((k) - 2 * nu - 1) * u))
  #define UNIT_LEN(k)   ((k) == 2 * nu ? 1 : u)
  int k = 0;
  if (!fl[0]) {
    int nx = 1; while (nx < total && !fl[nx]) nx++;
    pel v = line[UNIT_START(nx)];
    for (; k < nx; k++) for (int i = 0; i < UNIT_LEN(k); i++) line[UNIT_START(k) + i] = v;
  }
  for (; k < total; k++) if (!fl[k]) {
    pel v = line[UNIT_START(k) - 1];
    for (int i = 0; i < UNIT_LEN(k); i++) line[UNIT_START(k) + i] = v;
  }
  #undef UNIT_START
  #undef UNIT_LEN
}

static void filter_refs(const pel *src, pel *dst, int n)
{ /* TComPattern.cpp:203-293, luma only; strong smoothing for n >= 32 (SPS flag on) */
  const int n2 = 2 * n, last = 4 * n;
  int strong = 0;
  if (n >= 32 && (g_tools & 0x20u)) {          /* sps_strong_intra_smoothing_enable_flag, TComPattern.cpp:226 */
    const int thr = 1 << (g_bd - 5);
    int bl = src[0], tl = src[n2], tr = src[last];
    strong = abs(bl + tl - 2 * src[n]) < thr && abs(tl + tr - 2 * src[n2 + n]) < thr;
  }
  dst[0] = src[0]; dst[last] = src[last];
  if (strong) {
    const int shift = (n == 32) ? 6 : 7;        /* log2(2n) */
    int bl = src[0], tl = src[n2], tr = src[last];
    for (int i = 1; i < n2; i++) dst[i] = (pel)(((n2 - i) * bl + i * tl + n) >> shift);
    dst[n2] = src[n2];
    for (int i = 1; i < n2; i++) dst[n2 + i] = (pel)(((n2 - i) * tl + i * tr + n) >> shift);
  } else {
    for (int i = 1; i < last; i++) dst[i] = (pel)((src[i - 1] + 2 * src[i] + src[i + 1] + 2) >> 2);
  }
}

static inline int use_filtered_refs(int c, int mode, int n)
{ /* TComPattern.cpp:545-570; chroma never in 4:2:0 (TComChromaFormat.h:151-154) */
  if (c || mode == DC) return 0;
  int d1 = abs(mode - HOR), d2 = abs(mode - VER), diff = d1 < d2 ? d1 : d2;
  int idx = (n == 4) ? 0 : (n == 8) ? 1 : (n == 16) ? 2 : (n == 32) ? 3 : 4;
  return diff > g_intra_filter_thr[idx];
}

/* =====================================================================================
 * intra prediction (TComPrediction.cpp:183-473, 731-817); ref line as in build_refs
 * ===================================================================================== */
static inline int clip8(int v) { const int mx = (1 << g_bd) - 1; return v < 0 ? 0 : (v > mx ? mx : v); }   /* ClipBD */

static void predict_intra(int c, int mode, const pel *line, int n, pel *dst, int ds)
{
  const pel *above = line + 2 * n + 1;           /* above[-1] = corner, above[0..2n-1] */
  #define LEFT(i) line[2 * n - 1 - (i)]          /* LEFT(-1) = corner, LEFT(0..2n-1) going down */
  const int log2n = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : (n == 32) ? 5 : 6;
  if (mode == PLANAR) {
    int lc[65], tr[65], br[64], rc[64];
    for (int k = 0; k <= n; k++) { tr[k] = above[k]; lc[k] = LEFT(k); }
    int bl = lc[n], trr = tr[n];
    for (int k = 0; k < n; k++) { br[k] = bl - tr[k]; tr[k] <<= log2n; rc[k] = trr - lc[k]; lc[k] <<= log2n; }
    for (int y = 0; y < n; y++) {
      int hp = lc[y] + n;
      for (int x = 0; x < n; x++) { hp += rc[y]; tr[x] += br[x]; dst[y * ds + x] = (pel)((hp + tr[x]) >> (log2n + 1)); }
    }
    return;
  }
  if (mode == DC) {
    int sum = 0;
    for (int i = 0; i < n; i++) sum += above[i] + LEFT(i);
    pel dc = (pel)((sum + n) / (n + n));
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[y * ds + x] = dc;
    if (!c && n <= 16) {                         /* xDCPredFiltering */
      dst[0] = (pel)((above[0] + LEFT(0) + 2 * dst[0] + 2) >> 2);
      for (int x = 1; x < n; x++) dst[x] = (pel)((above[x] + 3 * dst[x] + 2) >> 2);
      for (int y = 1; y < n; y++) dst[y * ds] = (pel)((LEFT(y) + 3 * dst[y * ds] + 2) >> 2);
    }
    return;
  }
  const int is_ver = mode >= 18;
  const int ang_mode = is_ver ? mode - VER : -(mode - HOR);
  const int abs_ang = abs(ang_mode), sign = ang_mode < 0 ? -1 : 1;
  const int inv_angle = g_inv_ang_table[abs_ang];
  const int angle = sign * g_ang_table[abs_ang];
  const int edge = !c && n <= 16;
  pel ref_a[2 * 64 + 1 + 64], ref_l[2 * 64 + 1 + 64];
  pel *ref_main, *ref_side;
  if (angle < 0) {
    const int off = n - 1;
    for (int x = 0; x <= n; x++) ref_a[x + off] = above[x - 1];
    for (int y = 0; y <= n; y++) ref_l[y + off] = LEFT(y - 1);
    ref_main = (is_ver ? ref_a : ref_l) + off;
    ref_side = (is_ver ? ref_l : ref_a) + off;
    int sum = 128;
    for (int k = -1; k > ((n * angle) >> 5); k--) { sum += inv_angle; ref_main[k] = ref_side[sum >> 8]; }
  } else {
    for (int x = 0; x <= 2 * n; x++) ref_a[x] = above[x - 1];
    for (int y = 0; y <= 2 * n; y++) ref_l[y] = LEFT(y - 1);
    ref_main = is_ver ? ref_a : ref_l;
    ref_side = is_ver ? ref_l : ref_a;
  }
  pel tmp[64 * 64];
  pel *pd = is_ver ? dst : tmp;
  const int pds = is_ver ? ds : 64;
  if (angle == 0) {
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) pd[y * pds + x] = ref_main[x + 1];
    if (edge) for (int y = 0; y < n; y++) pd[y * pds] = (pel)clip8(pd[y * pds] + ((ref_side[y + 1] - ref_side[0]) >> 1));
  } else {
    int dpos = angle;
    for (int y = 0; y < n; y++, dpos += angle) {
      int di = dpos >> 5, df = dpos & 31;
      if (df) for (int x = 0; x < n; x++)
        pd[y * pds + x] = (pel)(((32 - df) * ref_main[x + di + 1] + df * ref_main[x + di + 2] + 16) >> 5);
      else for (int x = 0; x < n; x++) pd[y * pds + x] = ref_main[x + di + 1];
    }
  }
  if (!is_ver) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[x * ds + y] = tmp[y * 64 + x];
  #undef LEFT
}

/* =====================================================================================
 * distortion (TComRdCost.cpp:1527-1824 SATD, :1176-1520 SSE, :336-358 getDistPart)
 * ===================================================================================== */
static uint32_t had4(const pel *o, int os, const pel *p, int ps)
{
  int d[16], m[16];
  for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y * 4 + x] = o[y * os + x] - p[y * ps + x];
  for (int y = 0; y < 4; y++) {                 /* rows */
    int a = d[y * 4] + d[y * 4 + 3], b = d[y * 4 + 1] + d[y * 4 + 2], cc = d[y * 4 + 1] - d[y * 4 + 2], dd = d[y * 4] - d[y * 4 + 3];
    m[y * 4] = a + b; m[y * 4 + 1] = a - b; m[y * 4 + 2] = cc + dd; m[y * 4 + 3] = dd - cc;
  }
  uint32_t s = 0;
  for (int x = 0; x < 4; x++) {                 /* columns */
    int a = m[x] + m[12 + x], b = m[4 + x] + m[8 + x], cc = m[4 + x] - m[8 + x], dd = m[x] - m[12 + x];
    s += (uint32_t)(abs(a + b) + abs(a - b) + abs(cc + dd) + abs(dd - cc));
  }
  return (s + 1) >> 1;
}
static uint32_t had8(const pel *o, int os, const pel *p, int ps)
{
  int m[64];
  for (int y = 0; y < 8; y++) {
    int d[8]; for (int x = 0; x < 8; x++) d[x] = o[y * os + x] - p[y * ps + x];
    for (int st = 4; st >= 1; st >>= 1) { int t[8];
      for (int i = 0; i < 8; i++) t[i] = (i & st) ? d[i - st] - d[i] : d[i] + d[i + st];
      memcpy(d, t, sizeof d); }
    memcpy(m + y * 8, d, sizeof d);
  }
  uint32_t s = 0;
  for (int x = 0; x < 8; x++) {
    int d[8]; for (int y = 0; y < 8; y++) d[y] = m[y * 8 + x];
    for (int st = 4; st >= 1; st >>= 1) { int t[8];
      for (int i = 0; i < 8; i++) t[i] = (i & st) ? d[i - st] - d[i] : d[i] + d[i + st];
      memcpy(d, t, sizeof d); }
    for (int y = 0; y < 8; y++) s += (uint32_t)abs(d[y]);
  }
  return (s + 2) >> 2;
}
static uint32_t satd(const pel *o, int os, const pel *p, int ps, int n)
{
  uint32_t s = 0;
  if (n >= 8) { for (int y = 0; y < n; y += 8) for (int x = 0; x < n; x += 8) s += had8(o + y * os + x, os, p + y * ps + x, ps); }
  else s = had4(o, os, p, ps);
  return s >> DIST_ADJ(g_bd - 8);                 /* TComRdCost.cpp xGetHADs */
}
static uint32_t sse(const pel *a, int as, const pel *b, int bs, int n)
{
  uint32_t s = 0;
  const int sh = DIST_ADJ((g_bd - 8) << 1);          /* per sample, TComRdCost.cpp xGetSSE* */
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) { int d = a[y * as + x] - b[y * bs + x]; s += (uint32_t)(d * d) >> sh; }
  return s;
}

/* =====================================================================================
 * transforms (TComTrQuant.cpp:388-987, 2011-2117)
 * ===================================================================================== */
static void fwd_transform(const pel *resi, int rs, int32_t *coef, int n, int use_dst)
{
  const int log2n = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : 5;
  const int s1 = log2n + g_bd - 9, s2 = log2n + 6;
  const int a1 = s1 > 0 ? 1 << (s1 - 1) : 0, a2 = 1 << (s2 - 1);
  static int32_t tmp[32 * 32];
  for (int j = 0; j < n; j++)
    for (int k = 0; k < n; k++) {
      int32_t acc = 0;
      for (int i = 0; i < n; i++) acc += (use_dst ? g_dst4[k][i] : g_dct[log2n - 2][k][i]) * resi[j * rs + i];
      tmp[k * n + j] = (acc + a1) >> s1;
    }
  for (int j = 0; j < n; j++)                    /* j = horizontal frequency (row of tmp) */
    for (int k = 0; k < n; k++) {
      int32_t acc = 0;
      for (int i = 0; i < n; i++) acc += (use_dst ? g_dst4[k][i] : g_dct[log2n - 2][k][i]) * tmp[j * n + i];
      coef[k * n + j] = (acc + a2) >> s2;
    }
}
static inline int32_t clip16(int32_t v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static void inv_transform(const int32_t *coef, pel *resi, int rs, int n, int use_dst)
{
  const int log2n = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : 5;
  const int s1 = 7, s2 = 20 - g_bd;
  static int32_t tmp[32 * 32];
  for (int j = 0; j < n; j++)                    /* column j of coef */
    for (int x = 0; x < n; x++) {
      int32_t acc = 0;
      for (int k = 0; k < n; k++) acc += (use_dst ? g_dst4[k][x] : g_dct[log2n - 2][k][x]) * coef[k * n + j];
      tmp[j * n + x] = clip16((acc + (1 << (s1 - 1))) >> s1);
    }
  for (int j = 0; j < n; j++)                    /* j = output row */
    for (int x = 0; x < n; x++) {
      int32_t acc = 0;
      for (int k = 0; k < n; k++) acc += (use_dst ? g_dst4[k][x] : g_dct[log2n - 2][k][x]) * tmp[k * n + j];
      resi[j * rs + x] = (pel)clip16((acc + (1 << (s2 - 1))) >> s2);
    }
}

/* =====================================================================================
 * coefficient-coding geometry shared by RDOQ and the bit counter
 * ===================================================================================== */
typedef struct {
  int log2, n, ch, scan_type, wg, first_sig_ctx;
  const uint16_t *scan; const uint8_t *scan_cg;
} cparam_t;

static int coef_scan_idx(int c, int n, int dir_mode)
{ /* TComDataCU.cpp:3150-3209 (MDCS: luma <= 8, chroma <= 4) */
  if (n > (c ? 4 : 8)) return SCAN_DIAG;
  if (abs(dir_mode - VER) <= 4) return SCAN_HOR;
  if (abs(dir_mode - HOR) <= 4) return SCAN_VER;
  return SCAN_DIAG;
}
static void get_cparam(cparam_t *cp, int c, int n, int dir_mode)
{ /* TComChromaFormat.cpp:96-160 */
  cp->n = n; cp->log2 = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : 5; cp->ch = c ? 1 : 0;
  cp->scan_type = coef_scan_idx(c, n, dir_mode);
  cp->wg = n >> 2;
  cp->scan = g_scan[cp->scan_type][cp->log2 - 2];
  cp->scan_cg = g_scan_cg[cp->scan_type][cp->log2 - 2];
  if (n == 4) cp->first_sig_ctx = 0;
  else if (n == 8) cp->first_sig_ctx = 9 + ((cp->scan_type != SCAN_DIAG) ? (cp->ch ? 0 : 6) : 0);
  else cp->first_sig_ctx = cp->ch ? 12 : 21;
}
static int pattern_sig_ctx(const uint8_t *cgf, int gx, int gy, int wg)
{ /* TComTrQuant.cpp:2672-2705 */
  if (wg <= 1) return 0;
  int r = (gx < wg - 1) ? (cgf[gy * wg + gx + 1] != 0) : 0;
  int l = (gy < wg - 1) ? (cgf[(gy + 1) * wg + gx] != 0) : 0;
  return r + (l << 1);
}
static int sig_cg_ctx(const uint8_t *cgf, int gx, int gy, int wg)
{ /* TComTrQuant.cpp:3023-3049 */
  int r = (gx < wg - 1) ? (cgf[gy * wg + gx + 1] != 0) : 0;
  int l = (gy < wg - 1) ? (cgf[(gy + 1) * wg + gx] != 0) : 0;
  return (r + l) != 0;
}
static int sig_ctx_inc(const cparam_t *cp, int pat, int scan_pos)
{ /* TComTrQuant.cpp:2707-2803 (no single-context mode) */
  int raster = cp->scan[scan_pos], py = raster >> cp->log2, px = raster - (py << cp->log2);
  if (px + py == 0) return 0;
  int offset;
  if (cp->log2 == 2) offset = g_ctx_ind_map_4x4[4 * py + px];
  else {
    int cnt, xs = px & 3, ys = py & 3;
    switch (pat) {
      case 0: cnt = (xs + ys >= 3) ? 0 : ((xs + ys >= 1) ? 1 : 2); break;
      case 1: cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2); break;
      case 2: cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2); break;
      default: cnt = 2; break;
    }
    int not_first = ((px >> 2) + (py >> 2)) > 0;
    offset = (not_first ? (cp->ch ? 0 : 3) : 0) + cnt;
  }
  return cp->first_sig_ctx + offset;
}
static inline int ctx_set_index(int ch, int subset, int found_gt1)
{ /* TComChromaFormat.h:243-251 */
  return (ch ? 4 : 0) + ((!ch && subset > 0) ? 2 : 0) + (found_gt1 ? 1 : 0);
}
static inline void last_ctx_params(int ch, int n, int *off, int *shift)
{ /* TComChromaFormat.h:211-226 */
  int cw = (n == 4) ? 0 : (n == 8) ? 1 : (n == 16) ? 2 : 3;
  *off = ch ? 0 : (cw * 3 + ((cw + 1) >> 2));
  *shift = ch ? cw : ((cw + 3) >> 2);
}

/* =====================================================================================
 * RDOQ (TComTrQuant.cpp:2119-2661 and helpers :2812-2996)
 * The rate tables of estBitsSbacStruct are read straight from the frozen context states of `cab`
 * (they are refreshed from exactly these states before every TU, TEncSearch.cpp:1282-1286).
 * ===================================================================================== */
static int ic_rate(const cabac_t *cab, uint32_t abs_level, int ctx_one, int ctx_abs, int go_rice, uint32_t c1idx, uint32_t c2idx)
{ /* xGetICRate TComTrQuant.cpp:2881-2955 (no extended precision) */
  int rate = 32768;
  uint32_t base = (c1idx < 8) ? (2 + (c2idx < 1)) : 1;
  if (abs_level >= base) {
    uint32_t symbol = abs_level - base, length;
    if (symbol < (3u << go_rice)) { length = symbol >> go_rice; rate += (int)(length + 1 + go_rice) << 15; }
    else {
      length = go_rice; symbol -= (3u << go_rice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (int)(3 + length + 1 - go_rice + length) << 15;
    }
    if (c1idx < 8) { rate += ctx_bits(cab, CTX_ONE + ctx_one, 1); if (c2idx < 1) rate += ctx_bits(cab, CTX_ABS + ctx_abs, 1); }
  } else if (abs_level == 1) rate += ctx_bits(cab, CTX_ONE + ctx_one, 0);
  else if (abs_level == 2) { rate += ctx_bits(cab, CTX_ONE + ctx_one, 1); rate += ctx_bits(cab, CTX_ABS + ctx_abs, 0); }
  else rate = 0;
  return rate;
}

/* The quantiser without RDOQ (cfg RDOQ 0; RDOQTS 0 for transform-skipped blocks): TComTrQuant::xQuant, TComTrQuant.cpp:1169-1249 -- dead-zone rounding with the intra
 * offset 171 / 512 -- and, with sign data hiding on, signBitHidingHDQ (:991-1113): per coefficient group whose first and last level are four positions or more apart and
 * whose parity disagrees with the first level's sign, the level whose change costs least by deltaU alone goes up or down by one. */
static uint32_t plain_quant(const enc_t *e, int c, int n, int dir_mode, const int32_t *src, int32_t *dst)
{
  const int log2n = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : 5;
  const int qp = (c ? e->qp_c : e->qp) + 6 * (g_bd - 8), per = qp / 6, rem = qp % 6;
  const int qbits = 14 + per + (15 - g_bd - log2n), qbits8 = qbits - 8, ncoef = n * n;
  const int64_t add = (int64_t)171 << (qbits - 9);             /* I slice */
  const int qcoef = g_quant_scales[rem];
  static int32_t delta_u[1024];
  uint32_t abs_sum = 0;
  for (int i = 0; i < ncoef; i++) {
    const int64_t tmp = (int64_t)abs(src[i]) * qcoef;
    const int32_t mag = (int32_t)((tmp + add) >> qbits);
    delta_u[i] = (int32_t)((tmp - ((int64_t)mag << qbits)) >> qbits8);
    abs_sum += (uint32_t)mag;
    int32_t q = src[i] < 0 ? -mag : mag;
    dst[i] = q < -32768 ? -32768 : (q > 32767 ? 32767 : q);
  }
  if ((g_tools & 0x10u) && abs_sum >= 2) {
    cparam_t cp; get_cparam(&cp, c, n, dir_mode);
    int last_cg = -1;
    for (int subset = (ncoef - 1) >> 4; subset >= 0; subset--) {
      int sub_pos = subset << 4, first_nz = 16, last_nz = -1, sum = 0, k;
      for (k = 15; k >= 0; --k) if (dst[cp.scan[k + sub_pos]]) { last_nz = k; break; }
      for (k = 0; k < 16; k++) if (dst[cp.scan[k + sub_pos]]) { first_nz = k; break; }
      for (k = first_nz; k <= last_nz; k++) sum += dst[cp.scan[k + sub_pos]];
      if (last_nz >= 0 && last_cg == -1) last_cg = 1;
      if (last_nz - first_nz >= 4) {
        const uint32_t signbit = dst[cp.scan[sub_pos + first_nz]] > 0 ? 0 : 1;
        if (signbit != ((uint32_t)sum & 1u)) {
          int32_t cur_cost = INT32_MAX, min_cost = INT32_MAX; int min_pos = -1, final_change = 0, cur_change = 0;
          for (k = (last_cg == 1 ? last_nz : 15); k >= 0; --k) {
            const int blk = cp.scan[k + sub_pos];
            if (dst[blk] != 0) {
              if (delta_u[blk] > 0) { cur_cost = -delta_u[blk]; cur_change = 1; }
              else if (k == first_nz && abs(dst[blk]) == 1) cur_cost = INT32_MAX;        /* (curChange keeps its value, as in the reference) */
              else { cur_cost = delta_u[blk]; cur_change = -1; }
            } else if (k < first_nz) {
              const uint32_t ts = src[blk] >= 0 ? 0 : 1;
              if (ts != signbit) cur_cost = INT32_MAX; else { cur_cost = -delta_u[blk]; cur_change = 1; }
            } else { cur_cost = -delta_u[blk]; cur_change = 1; }
            if (cur_cost < min_cost) { min_cost = cur_cost; final_change = cur_change; min_pos = blk; }
          }
          if (dst[min_pos] == 32767 || dst[min_pos] == -32768) final_change = -1;
          if (src[min_pos] >= 0) dst[min_pos] += final_change; else dst[min_pos] -= final_change;
        }
      }
      if (last_cg == 1) last_cg = 0;
    }
  }
  return abs_sum;
}

static uint32_t rdoq(const enc_t *e, const cabac_t *cab, int c, int n, int dir_mode, int is_tskip, int cbf_ctx,
                     const int32_t *src, int32_t *dst)
{
  if (!(g_tools & (is_tskip ? 0x02u : 0x01u))) return plain_quant(e, c, n, dir_mode, src, dst);     /* useRDOQ = transform skip ? RDOQTS : RDOQ, TComTrQuant.cpp:1152 */
  const int ch = c ? 1 : 0;
  const int log2n = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : 5;
  const int qp = (c ? e->qp_c : e->qp) + 6 * (g_bd - 8), per = qp / 6, rem = qp % 6;      /* + qpBdOffset, TComTrQuant.cpp:71-100 */
  const int tshift = 15 - g_bd - log2n;
  const int qbits = 14 + per + tshift;
  const double lambda = c ? e->lambda_c : e->lambda;
  const double err_scale = e->err_scale[ch][log2n - 2];
  const int qcoef = g_quant_scales[rem];
  const int ncoef = n * n;
  cparam_t cp; get_cparam(&cp, c, n, dir_mode);
  static double cost_coeff[1024], cost_sig[1024], cost_coeff0[1024];
  static int rate_inc_up[1024], rate_inc_down[1024], sig_rate_delta[1024];
  static int32_t delta_u[1024];
  double cost_cg_sig[64];
  uint8_t cgf[64];
  memset(cost_coeff, 0, sizeof(double) * ncoef); memset(cost_sig, 0, sizeof(double) * ncoef);
  memset(rate_inc_up, 0, sizeof(int) * ncoef); memset(rate_inc_down, 0, sizeof(int) * ncoef);
  memset(sig_rate_delta, 0, sizeof(int) * ncoef); memset(delta_u, 0, sizeof(int32_t) * ncoef);
  memset(cost_cg_sig, 0, sizeof cost_cg_sig); memset(cgf, 0, sizeof cgf);
  const int sig_off = CTX_SIG + (ch ? 28 : 0);
  const int cg_off = CTX_SIG_CG + (ch ? 2 : 0);
  double block_uncoded = 0, base_cost = 0;
  int cg_last = -1, last_pos = -1;
  int ctx_set = 0, c1 = 1, c2 = 0; uint32_t c1idx = 0, c2idx = 0; int go_rice = 0;
  const int ncg = ncoef >> 4;
  for (int cgpos = ncg - 1; cgpos >= 0; cgpos--) {
    int cgblk = cp.scan_cg[cgpos], gy = cgblk / cp.wg, gx = cgblk - gy * cp.wg;
    double st_sig_cost = 0, st_sig_cost0 = 0, st_coded = 0, st_uncoded = 0; int st_nnz_before0 = 0;
    const int pat = pattern_sig_ctx(cgf, gx, gy, cp.wg);
    for (int pin = 15; pin >= 0; pin--) {
      const int sp = cgpos * 16 + pin;
      const int blk = cp.scan[sp];
      const int64_t tmpl = (int64_t)abs(src[blk]) * qcoef;
      const int64_t lim = (int64_t)0x7fffffff - ((int64_t)1 << (qbits - 1));
      const int32_t ld = (int32_t)(tmpl < lim ? tmpl : lim);
      uint32_t max_abs = (uint32_t)(((int64_t)ld + ((int64_t)1 << (qbits - 1))) >> qbits);
      if (max_abs > 32767u) max_abs = 32767u;
      const double derr = (double)ld;
      cost_coeff0[sp] = derr * derr * err_scale;
      block_uncoded += cost_coeff0[sp];
      dst[blk] = (int32_t)max_abs;
      if (max_abs > 0 && last_pos < 0) { last_pos = sp; ctx_set = ctx_set_index(ch, sp >> 4, 0); cg_last = cgpos; }
      if (last_pos >= 0) {
        uint32_t level;
        const int one_ctx = 4 * ctx_set + c1, abs_ctx = ctx_set + c2;
        /* xGetCodedLevel TComTrQuant.cpp:2812-2879 */
        {
          const int is_last = (sp == last_pos);
          int sig_ctx = 0; double cur_sig = 0; double best = MAX_DOUBLE; uint32_t best_lvl = 0; int done = 0;
          if (!is_last) sig_ctx = sig_off + sig_ctx_inc(&cp, pat, sp);
          if (!is_last && max_abs < 3) {
            cost_sig[sp] = lambda * (double)ctx_bits(cab, sig_ctx, 0);
            cost_coeff[sp] = cost_coeff0[sp] + cost_sig[sp];
            best = cost_coeff[sp];
            if (max_abs == 0) done = 1;
          } else cost_coeff[sp] = MAX_DOUBLE;
          if (!done) {
            if (!is_last) cur_sig = lambda * (double)ctx_bits(cab, sig_ctx, 1);
            uint32_t min_abs = max_abs > 1 ? max_abs - 1 : 1;
            for (int al = (int)max_abs; al >= (int)min_abs; al--) {
              double err = (double)(ld - (int32_t)((uint32_t)al << qbits));
              double cur = err * err * err_scale + lambda * (double)ic_rate(cab, (uint32_t)al, one_ctx, abs_ctx, go_rice, c1idx, c2idx);
              cur += cur_sig;
              if (cur < best) { best_lvl = (uint32_t)al; best = cur; cost_sig[sp] = cur_sig; }
            }
            cost_coeff[sp] = best;
          }
          level = best_lvl;
          if (!is_last) sig_rate_delta[blk] = ctx_bits(cab, sig_ctx, 1) - ctx_bits(cab, sig_ctx, 0);
        }
        delta_u[blk] = (int32_t)((ld - (int32_t)(level << qbits)) >> (qbits - 8));
        if (level > 0) {
          int now = ic_rate(cab, level, one_ctx, abs_ctx, go_rice, c1idx, c2idx);
          rate_inc_up[blk] = ic_rate(cab, level + 1, one_ctx, abs_ctx, go_rice, c1idx, c2idx) - now;
          rate_inc_down[blk] = ic_rate(cab, level - 1, one_ctx, abs_ctx, go_rice, c1idx, c2idx) - now;
        } else rate_inc_up[blk] = ctx_bits(cab, CTX_ONE + one_ctx, 0);
        dst[blk] = (int32_t)level;
        base_cost += cost_coeff[sp];
        uint32_t base_level = (c1idx < 8) ? (2 + (c2idx < 1)) : 1;
        if (level >= base_level) { if (level > 3u * (1u << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4; }
        if (level >= 1) c1idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2idx++; }
        else if (c1 < 3 && c1 > 0 && level) c1++;
        if ((sp & 15) == 0 && sp > 0) {
          ctx_set = ctx_set_index(ch, (sp - 1) >> 4, c1 == 0);
          c1 = 1; c2 = 0; c1idx = 0; c2idx = 0; go_rice = 0;
        }
      } else base_cost += cost_coeff0[sp];
      st_sig_cost += cost_sig[sp];
      if (pin == 0) st_sig_cost0 = cost_sig[sp];
      if (dst[blk]) {
        cgf[cgblk] = 1;
        st_coded += cost_coeff[sp] - cost_sig[sp];
        st_uncoded += cost_coeff0[sp];
        if (pin != 0) st_nnz_before0++;
      }
    }
    if (cg_last >= 0) {
      if (cgpos) {
        if (cgf[cgblk] == 0) {
          int cs = sig_cg_ctx(cgf, gx, gy, cp.wg);
          double r0 = lambda * (double)ctx_bits(cab, cg_off + cs, 0);
          base_cost += r0 - st_sig_cost;
          cost_cg_sig[cgpos] = r0;
        } else if (cgpos < cg_last) {
          if (st_nnz_before0 == 0) { base_cost -= st_sig_cost0; st_sig_cost -= st_sig_cost0; }
          double zero_cost = base_cost;
          int cs = sig_cg_ctx(cgf, gx, gy, cp.wg);
          double r1 = lambda * (double)ctx_bits(cab, cg_off + cs, 1), r0 = lambda * (double)ctx_bits(cab, cg_off + cs, 0);
          base_cost += r1; zero_cost += r0; cost_cg_sig[cgpos] = r1;
          zero_cost += st_uncoded; zero_cost -= st_coded; zero_cost -= st_sig_cost;
          if (zero_cost < base_cost) {
            cgf[cgblk] = 0; base_cost = zero_cost; cost_cg_sig[cgpos] = r0;
            for (int pin = 15; pin >= 0; pin--) {
              int sp = cgpos * 16 + pin, blk = cp.scan[sp];
              if (dst[blk]) { dst[blk] = 0; cost_coeff[sp] = cost_coeff0[sp]; cost_sig[sp] = 0; }
            }
          }
        }
      } else cgf[cgblk] = 1;
    }
  }
  if (last_pos < 0) return 0;
  double best_cost;
  {
    int cctx = CTX_QT_CBF + (ch ? 5 : 0) + cbf_ctx;
    best_cost = block_uncoded + lambda * (double)ctx_bits(cab, cctx, 0);
    base_cost += lambda * (double)ctx_bits(cab, cctx, 1);
  }
  /* last-position rate: TEncSbac.cpp:1910-1930 prefix sums + xGetRateLast TComTrQuant.cpp:2972-2990 */
  int last_x_bits[12], last_y_bits[12];
  {
    int off, shift; last_ctx_params(ch, n, &off, &shift);
    const int bx = CTX_LAST_X + (ch ? 15 : 0), by = CTX_LAST_Y + (ch ? 15 : 0);
    int accx = 0, accy = 0, k, ng = g_group_idx[n - 1];
    for (k = 0; k < ng; k++) {
      last_x_bits[k] = accx + ctx_bits(cab, bx + off + (k >> shift), 0); accx += ctx_bits(cab, bx + off + (k >> shift), 1);
      last_y_bits[k] = accy + ctx_bits(cab, by + off + (k >> shift), 0); accy += ctx_bits(cab, by + off + (k >> shift), 1);
    }
    last_x_bits[k] = accx; last_y_bits[k] = accy;
  }
  int best_last_p1 = 0, found_last = 0;
  for (int cgpos = cg_last; cgpos >= 0 && !found_last; cgpos--) {
    int cgblk = cp.scan_cg[cgpos];
    base_cost -= cost_cg_sig[cgpos];
    if (!cgf[cgblk]) continue;
    for (int pin = 15; pin >= 0; pin--) {
      int sp = cgpos * 16 + pin;
      if (sp > last_pos) continue;
      int blk = cp.scan[sp];
      if (dst[blk]) {
        int py = blk >> log2n, px = blk - (py << log2n);
        if (cp.scan_type == SCAN_VER) { int t = px; px = py; py = t; }
        int gx2 = g_group_idx[px], gy2 = g_group_idx[py];
        double lc = (double)(last_x_bits[gx2] + last_y_bits[gy2]);
        if (gx2 > 3) lc += 32768.0 * (double)((gx2 - 2) >> 1);
        if (gy2 > 3) lc += 32768.0 * (double)((gy2 - 2) >> 1);
        double cost_last = lambda * lc;
        double total = base_cost + cost_last - cost_sig[sp];
        if (total < best_cost) { best_last_p1 = sp + 1; best_cost = total; }
        if (dst[blk] > 1) { found_last = 1; break; }
        base_cost -= cost_coeff[sp]; base_cost += cost_coeff0[sp];
      } else base_cost -= cost_sig[sp];
    }
  }
  uint32_t abs_sum = 0;
  for (int sp = 0; sp < best_last_p1; sp++) {
    int blk = cp.scan[sp]; int32_t lv = dst[blk];
    abs_sum += (uint32_t)lv;
    dst[blk] = src[blk] < 0 ? -lv : lv;
  }
  for (int sp = best_last_p1; sp <= last_pos; sp++) dst[cp.scan[sp]] = 0;
  /* sign data hiding, TComTrQuant.cpp:2530-2660 */
  if (abs_sum >= 2 && (g_tools & 0x10u)) {     /* getSignDataHidingEnabledFlag, TComTrQuant.cpp:2530 */
    const double inv = (double)g_inv_quant_scales[rem];
    int64_t rd_factor = (int64_t)(inv * inv * (1 << (2 * per)) / lambda / 16 / (1 << DIST_ADJ(2 * (g_bd - 8))) + 0.5);
    int last_cg = -1;
    for (int subset = (ncoef - 1) >> 4; subset >= 0; subset--) {
      int sub_pos = subset << 4, first_nz = 16, last_nz = -1, sum = 0, k;
      for (k = 15; k >= 0; --k) if (dst[cp.scan[k + sub_pos]]) { last_nz = k; break; }
      for (k = 0; k < 16; k++) if (dst[cp.scan[k + sub_pos]]) { first_nz = k; break; }
      for (k = first_nz; k <= last_nz; k++) sum += dst[cp.scan[k + sub_pos]];
      if (last_nz >= 0 && last_cg == -1) last_cg = 1;
      if (last_nz - first_nz >= 4) {
        uint32_t signbit = dst[cp.scan[sub_pos + first_nz]] > 0 ? 0 : 1;
        if (signbit != ((uint32_t)sum & 1u)) {
          int64_t min_cost = INT64_MAX, cur_cost = INT64_MAX; int min_pos = -1, final_change = 0, cur_change = 0;
          for (k = (last_cg == 1 ? last_nz : 15); k >= 0; --k) {
            int blk = cp.scan[k + sub_pos];
            if (dst[blk] != 0) {
              int64_t up = rd_factor * (-(int64_t)delta_u[blk]) + rate_inc_up[blk];
              int64_t down = rd_factor * ((int64_t)delta_u[blk]) + rate_inc_down[blk] - ((abs(dst[blk]) == 1) ? sig_rate_delta[blk] : 0);
              if (last_cg == 1 && last_nz == k && abs(dst[blk]) == 1) down -= (4 << 15);
              if (up < down) { cur_cost = up; cur_change = 1; }
              else { cur_change = -1; cur_cost = (k == first_nz && abs(dst[blk]) == 1) ? INT64_MAX : down; }
            } else {
              cur_cost = rd_factor * (-(int64_t)abs(delta_u[blk])) + (1 << 15) + rate_inc_up[blk] + sig_rate_delta[blk];
              cur_change = 1;
              if (k < first_nz) { uint32_t ts = src[blk] >= 0 ? 0 : 1; if (ts != signbit) cur_cost = INT64_MAX; }
            }
            if (cur_cost < min_cost) { min_cost = cur_cost; final_change = cur_change; min_pos = blk; }
          }
          if (dst[min_pos] == 32767 || dst[min_pos] == -32768) final_change = -1;
          if (src[min_pos] >= 0) dst[min_pos] += final_change; else dst[min_pos] -= final_change;
        }
      }
      if (last_cg == 1) last_cg = 0;
    }
  }
  return abs_sum;
}

static void dequant(const enc_t *e, int c, int n, const int32_t *src, int32_t *dst)
{ /* TComTrQuant.cpp:1308-1425, flat scaling */
  const int log2n = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : 5;
  const int qp = (c ? e->qp_c : e->qp) + 6 * (g_bd - 8), per = qp / 6, rem = qp % 6;
  const int tshift = 15 - g_bd - log2n;
  const int rshift = 6 - (tshift + per);
  const int scale = g_inv_quant_scales[rem];
  int tbits = 32 + rshift - 7; if (tbits > 16) tbits = 16;
  const int32_t imin = -(1 << (tbits - 1)), imax = (1 << (tbits - 1)) - 1;
  for (int i = 0; i < n * n; i++) {
    int32_t q = src[i] < imin ? imin : (src[i] > imax ? imax : src[i]);
    int32_t v;
    if (rshift > 0) v = (q * scale + (1 << (rshift - 1))) >> rshift;
    else v = (int32_t)((uint32_t)(q * scale) << (-rshift));
    dst[i] = clip16(v);
  }
}

/* =====================================================================================
 * residual syntax bit counting (TEncSbac.cpp:1115-1541)
 * ===================================================================================== */
static void code_last_xy(cabac_t *c, int px, int py, int n, int ch, int scan_type)
{
  if (scan_type == SCAN_VER) { int t = px; px = py; py = t; }
  int gx = g_group_idx[px], gy = g_group_idx[py], off, shift, k;
  last_ctx_params(ch, n, &off, &shift);
  const int bx = CTX_LAST_X + (ch ? 15 : 0) + off, by = CTX_LAST_Y + (ch ? 15 : 0) + off;
  for (k = 0; k < gx; k++) enc_bin(c, bx + (k >> shift), 1);
  if (gx < g_group_idx[n - 1]) enc_bin(c, bx + (k >> shift), 0);
  for (k = 0; k < gy; k++) enc_bin(c, by + (k >> shift), 1);
  if (gy < g_group_idx[n - 1]) enc_bin(c, by + (k >> shift), 0);
  if (gx > 3) enc_ep(c, (gx - 2) >> 1);
  if (gy > 3) enc_ep(c, (gy - 2) >> 1);
}
static void code_coef_remain(cabac_t *c, uint32_t symbol, int rparam)
{ /* xWriteCoefRemainExGolomb TEncSbac.cpp:337-394 (bit count only) */
  if (symbol < (3u << rparam)) { uint32_t len = symbol >> rparam; enc_ep(c, (int)len + 1); enc_ep(c, rparam); }
  else {
    uint32_t len = (uint32_t)rparam, cn = symbol - (3u << rparam);
    while (cn >= (1u << len)) cn -= (1u << (len++));
    enc_ep(c, (int)(3 + len + 1 - rparam)); enc_ep(c, (int)len);
  }
}
static void code_coeff_nxn(cabac_t *c, const int32_t *coef, int comp, int n, int dir_mode, int tskip_flag)
{
  const int ch = comp ? 1 : 0;
  cparam_t cp; get_cparam(&cp, comp, n, dir_mode);
  const int log2n = cp.log2;
  int num_sig = 0;
  for (int i = 0; i < n * n; i++) num_sig += coef[i] != 0;
  if (num_sig == 0) { fprintf(stderr, "oracle: code_coeff_nxn on empty TU\n"); abort(); }
  if (n == 4 && (g_tools & 0x04u)) enc_bin(c, CTX_TSKIP + ch, tskip_flag);          /* codeTransformSkipFlags :997-1032 (only with transform_skip_enabled_flag) */
  uint8_t cgf[64]; memset(cgf, 0, sizeof cgf);
  int scan_last = -1, pos_last;
  do {
    pos_last = cp.scan[++scan_last];
    if (coef[pos_last] != 0) {
      int py = pos_last >> log2n, px = pos_last - (py << log2n);
      cgf[cp.wg * (py >> 2) + (px >> 2)] = 1;
      num_sig--;
    }
  } while (num_sig > 0);
  { int py = pos_last >> log2n, px = pos_last - (py << log2n); code_last_xy(c, px, py, n, ch, cp.scan_type); }
  const int cg_off = CTX_SIG_CG + (ch ? 2 : 0), sig_off = CTX_SIG + (ch ? 28 : 0);
  const int last_set = scan_last >> 4;
  uint32_t c1 = 1; int go_rice = 0; int sp = scan_last;
  for (int subset = last_set; subset >= 0; subset--) {
    int num_nz = 0, sub_pos = subset << 4;
    go_rice = 0;
    int abs_coeff[16]; int last_nz = -1, first_nz = 16; int escape = 0;
    if (sp == scan_last) { abs_coeff[0] = abs(coef[pos_last]); num_nz = 1; last_nz = sp; first_nz = sp; sp--; }
    int cgblk = cp.scan_cg[subset], gy = cgblk / cp.wg, gx = cgblk - gy * cp.wg;
    if (subset == last_set || subset == 0) cgf[cgblk] = 1;
    else enc_bin(c, cg_off + sig_cg_ctx(cgf, gx, gy, cp.wg), cgf[cgblk] != 0);
    if (cgf[cgblk]) {
      int pat = pattern_sig_ctx(cgf, gx, gy, cp.wg);
      for (; sp >= sub_pos; sp--) {
        int blk = cp.scan[sp], sig = coef[blk] != 0;
        if (sp > sub_pos || subset == 0 || num_nz) enc_bin(c, sig_off + sig_ctx_inc(&cp, pat, sp), sig);
        if (sig) { abs_coeff[num_nz++] = abs(coef[blk]); if (last_nz == -1) last_nz = sp; first_nz = sp; }
      }
    } else sp = sub_pos - 1;
    if (num_nz > 0) {
      int sign_hidden = (g_tools & 0x10u) && (last_nz - first_nz >= 4);
      int cset = ctx_set_index(ch, subset, c1 == 0);
      c1 = 1;
      int n_c1 = num_nz < 8 ? num_nz : 8, first_c2 = -1;
      for (int i = 0; i < n_c1; i++) {
        int sym = abs_coeff[i] > 1;
        enc_bin(c, CTX_ONE + 4 * cset + (int)c1, sym);
        if (sym) { c1 = 0; if (first_c2 == -1) first_c2 = i; else escape = 1; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && first_c2 != -1) {
        int sym = abs_coeff[first_c2] > 2;
        enc_bin(c, CTX_ABS + cset, sym);
        if (sym) escape = 1;
      }
      escape = escape || (num_nz > 8);
      enc_ep(c, sign_hidden ? num_nz - 1 : num_nz);
      int first_coeff2 = 1;
      if (escape) for (int i = 0; i < num_nz; i++) {
        int base = (i < 8) ? (2 + first_coeff2) : 1;
        if (abs_coeff[i] >= base) {
          code_coef_remain(c, (uint32_t)(abs_coeff[i] - base), go_rice);
          if (abs_coeff[i] > (3 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
        }
        if (abs_coeff[i] >= 2) first_coeff2 = 0;
      }
    }
  }
}

/* =====================================================================================
 * mode syntax (TEncSbac.cpp:613-726, TComDataCU.cpp:1334-1461)
 * ===================================================================================== */
static void get_mpm(const enc_t *e, int x, int y, int preds[3], int *nmode)
{ /* getIntraDirPredictor TComDataCU.cpp:1362-1445 for the luma PU whose top-left sample is (x,y) */
  int left = DC, above = DC, z;
  if (x > e->tx0) { irec_t *r = rec_of(e, (x >> 2) - 1, y >> 2, &z); left = r->a[A_LDIR][z]; }
  if ((y & 63) != 0) { irec_t *r = rec_of(e, x >> 2, (y >> 2) - 1, &z); above = r->a[A_LDIR][z]; }
  if (left == above) {
    if (nmode) *nmode = 1;
    if (left > 1) { preds[0] = left; preds[1] = ((left + 29) % 32) + 2; preds[2] = ((left - 1) % 32) + 2; }
    else { preds[0] = PLANAR; preds[1] = DC; preds[2] = VER; }
  } else {
    if (nmode) *nmode = 2;
    preds[0] = left; preds[1] = above;
    if (left && above) preds[2] = PLANAR; else preds[2] = (left + above) < 2 ? VER : DC;
  }
}
static void code_luma_dirs(enc_t *e, cabac_t *c, const cu_t *cu, int first_pu, int npu)
{ /* codeIntraDirLumaAng TEncSbac.cpp:643-696 for PUs first_pu .. first_pu+npu-1 of the CU */
  int preds[4][3], idx[4], dir[4];
  const int pu_size = (cu->part == SIZE_NxN) ? (1 << (cu->log2 - 1)) : (1 << cu->log2);
  for (int j = 0; j < npu; j++) {
    int pu = first_pu + j, px = cu->x + (pu & 1) * pu_size, py = cu->y + (pu >> 1) * pu_size;
    dir[j] = e->r->a[A_LDIR][cu->zbase + pu * (cu->nparts >> 2) * (cu->part == SIZE_NxN)];
    get_mpm(e, px, py, preds[j], NULL);
    idx[j] = -1;
    for (int i = 0; i < 3; i++) if (dir[j] == preds[j][i]) idx[j] = i;
    enc_bin(c, CTX_INTRA_PRED, idx[j] != -1);
  }
  for (int j = 0; j < npu; j++) {
    if (idx[j] != -1) enc_ep(c, idx[j] ? 2 : 1);
    else enc_ep(c, 5);
  }
}
static void code_chroma_dir(enc_t *e, cabac_t *c, const cu_t *cu)
{ /* codeIntraDirChroma TEncSbac.cpp:698-726 */
  int d = e->r->a[A_CDIR][cu->zbase];
  if (d == DM_CHROMA) enc_bin(c, CTX_CHROMA_PRED, 0);
  else { enc_bin(c, CTX_CHROMA_PRED, 1); enc_ep(c, 2); }
}
static int split_ctx(const enc_t *e, int x, int y, int depth)
{ /* getCtxSplitFlag TComDataCU.cpp:1447-1461 */
  int ctx = 0, z;
  if (x > e->tx0) { irec_t *r = rec_of(e, (x >> 2) - 1, y >> 2, &z); ctx += r->a[A_DEPTH][z] > depth; }
  if (y > e->ty0) { irec_t *r = rec_of(e, x >> 2, (y >> 2) - 1, &z); ctx += r->a[A_DEPTH][z] > depth; }
  return ctx;
}
static inline int min_tu_log2(const cu_t *cu)
{ /* getQuadtreeTULog2MinSizeInCU TComDataCU.cpp:1478-1503 with TU log2 2..5, intra TU depth 3 */
  int split = cu->part == SIZE_NxN, r;
  if (cu->log2 < 2 + 3 - 1 + split) r = 2;
  else { r = cu->log2 - (3 - 1 + split); if (r > 5) r = 5; }
  return r;
}

/* chroma geometry of a luma TU: returns 0 when this luma TU carries no chroma block in SEARCH order */
static inline int tu_has_chroma_first(const tu_t *tu) { return tu->log2 > 2 || (tu->zrel & 3) == 0; }
static inline int tu_has_chroma_last(const tu_t *tu)  { return tu->log2 > 2 || (tu->zrel & 3) == 3; }
static inline int tu_csize(const tu_t *tu) { return tu->log2 > 2 ? 1 << (tu->log2 - 1) : 4; }
static inline int tu_czrel(const tu_t *tu) { return tu->log2 > 2 ? tu->zrel : (tu->zrel & ~3); }
static inline int tu_cnparts(const tu_t *tu) { return tu->log2 > 2 ? tu->nparts : 4; }

static inline void tu_child(const tu_t *p, int i, tu_t *ch)
{
  int h = 1 << (p->log2 - 1);
  ch->log2 = p->log2 - 1; ch->trd = p->trd + 1; ch->nparts = p->nparts >> 2;
  ch->x = p->x + (i & 1) * h; ch->y = p->y + (i >> 1) * h; ch->zrel = p->zrel + i * ch->nparts;
}
static inline int mode_of(const enc_t *e, const cu_t *cu, int c, int zrel)
{ /* luma / resolved chroma prediction mode at CU-relative partition zrel (TEncSearch.cpp:1178-1181) */
  if (!c) return e->r->a[A_LDIR][cu->zbase + zrel];
  int m = e->r->a[A_CDIR][cu->zbase + zrel];
  return m == DM_CHROMA ? e->r->a[A_LDIR][cu->zbase + (zrel & ~3)] : m;
}

static void code_qt_cbf(enc_t *e, cabac_t *c, const cu_t *cu, const tu_t *tu, int comp, int lowest)
{ /* codeQtCbf TEncSbac.cpp:920-995 + getCtxQtCbf TComDataCU.cpp:1463-1476 */
  int ctx = comp ? tu->trd : (tu->trd == 0 ? 1 : 0);
  int w = comp ? tu_csize(tu) : (1 << tu->log2);
  int can_split = w >= 8;
  int d = tu->trd + ((!lowest && !can_split) ? 1 : 0);
  int z = cu->zbase + (comp ? tu_czrel(tu) : tu->zrel);
  int cbf = (e->r->a[A_CBF + comp][z] >> d) & 1;
  enc_bin(c, CTX_QT_CBF + (comp ? 5 : 0) + ctx, cbf);
}

/* coefficient source selector for bit counting */
static inline const int32_t *coef_src(const enc_t *e, int real, int comp, int log2_luma, int zabs_comp)
{
  int off = comp ? (zabs_comp * 16) >> 2 : zabs_comp * 16;
  return real ? e->r->coef[comp] + off : e->coef_l[5 - log2_luma][comp] + off;
}

/* xEncSubdivCbfQT TEncSearch.cpp:907-972 */
static void enc_subdiv_cbf(enc_t *e, cabac_t *c, const cu_t *cu, const tu_t *tu, int luma, int chroma)
{
  const int subdiv = e->r->a[A_TRIDX][cu->zbase + tu->zrel] > tu->trd;
  if (cu->part == SIZE_NxN && tu->trd == 0) { }
  else if (tu->log2 > 5) { }
  else if (tu->log2 == 2) { }
  else if (tu->log2 == min_tu_log2(cu)) { }
  else if (luma) enc_bin(c, CTX_SUBDIV + 5 - tu->log2, subdiv);
  if (chroma) for (int comp = 1; comp < 3; comp++)
    if (tu->log2 > 2 && (tu->trd == 0 || ((e->r->a[A_CBF + comp][cu->zbase + tu->zrel] >> (tu->trd - 1)) & 1)))
      code_qt_cbf(e, c, cu, tu, comp, !subdiv);
  if (subdiv) { for (int i = 0; i < 4; i++) { tu_t ch; tu_child(tu, i, &ch); enc_subdiv_cbf(e, c, cu, &ch, luma, chroma); } }
  else if (luma) code_qt_cbf(e, c, cu, tu, 0, 1);
}
/* xEncCoeffQT TEncSearch.cpp:978-1012 (+ TEncEntropy::encodeCoeffNxN cbf test :654-690) */
static void enc_coeff_qt(enc_t *e, cabac_t *c, const cu_t *cu, const tu_t *tu, int comp, int real)
{
  if (e->r->a[A_TRIDX][cu->zbase + tu->zrel] > tu->trd) {
    for (int i = 0; i < 4; i++) { tu_t ch; tu_child(tu, i, &ch); enc_coeff_qt(e, c, cu, &ch, comp, real); }
    return;
  }
  if (comp && !tu_has_chroma_first(tu)) return;
  if (!((e->r->a[A_CBF + comp][cu->zbase + tu->zrel] >> tu->trd) & 1)) return;
  int zc = comp ? tu_czrel(tu) : tu->zrel;
  int n = comp ? tu_csize(tu) : (1 << tu->log2);
  code_coeff_nxn(c, coef_src(e, real, comp, tu->log2, cu->zbase + zc), comp, n, mode_of(e, cu, comp, zc),
                 e->r->a[A_TSKIP + comp][cu->zbase + zc]);
}
/* xEncIntraHeader TEncSearch.cpp:1018-1087 */
static void enc_intra_header(enc_t *e, cabac_t *c, const cu_t *cu, const tu_t *tu, int luma, int chroma)
{
  if (luma) {
    if (tu->zrel == 0 && cu->depth == 3) enc_bin(c, CTX_PART_SIZE, cu->part == SIZE_2Nx2N);
    if (cu->part == SIZE_2Nx2N) { if (tu->zrel == 0) code_luma_dirs(e, c, cu, 0, 1); }
    else { int q = cu->nparts >> 2; if (tu->trd > 0 && (tu->zrel % q) == 0) code_luma_dirs(e, c, cu, tu->zrel / q, 1); }
  }
  if (chroma && tu->zrel == 0) code_chroma_dir(e, c, cu);
}
/* xGetIntraBitsQT TEncSearch.cpp:1093-1117 */
static uint32_t intra_bits_qt(enc_t *e, const cu_t *cu, const tu_t *tu, int luma, int chroma)
{
  cabac_t *c = &e->go;
  reset_bits(c);
  enc_intra_header(e, c, cu, tu, luma, chroma);
  enc_subdiv_cbf(e, c, cu, tu, luma, chroma);
  if (luma) enc_coeff_qt(e, c, cu, tu, 0, 0);
  if (chroma) { enc_coeff_qt(e, c, cu, tu, 1, 0); enc_coeff_qt(e, c, cu, tu, 2, 0); }
  return get_bits(c);
}

/* final transform-tree syntax: TEncEntropy::xEncodeTransform TEncEntropy.cpp:200-398 (real coefficients,
 * chroma of 4x4 luma quads coded with the LAST quadrant) */
static void enc_transform(enc_t *e, cabac_t *c, const cu_t *cu, const tu_t *tu)
{
  const int z = cu->zbase + tu->zrel;
  const int subdiv = e->r->a[A_TRIDX][z] > tu->trd;
  if (cu->part == SIZE_NxN && tu->trd == 0) { }
  else if (tu->log2 > 5) { }
  else if (tu->log2 == 2) { }
  else if (tu->log2 == min_tu_log2(cu)) { }
  else enc_bin(c, CTX_SUBDIV + 5 - tu->log2, subdiv);
  const int first = tu->trd == 0;
  for (int comp = 1; comp < 3; comp++)
    if (first || tu->log2 > 2)
      if (first || ((e->r->a[A_CBF + comp][z] >> (tu->trd - 1)) & 1)) code_qt_cbf(e, c, cu, tu, comp, !subdiv);
  if (subdiv) { for (int i = 0; i < 4; i++) { tu_t ch; tu_child(tu, i, &ch); enc_transform(e, c, cu, &ch); } return; }
  code_qt_cbf(e, c, cu, tu, 0, 1);
  for (int comp = 0; comp < 3; comp++) {
    if (comp && !tu_has_chroma_last(tu)) continue;
    if (!((e->r->a[A_CBF + comp][z] >> tu->trd) & 1)) continue;
    int zc = comp ? tu_czrel(tu) : tu->zrel;
    int n = comp ? tu_csize(tu) : (1 << tu->log2);
    code_coeff_nxn(c, coef_src(e, 1, comp, tu->log2, cu->zbase + zc), comp, n, mode_of(e, cu, comp, zc),
                   e->r->a[A_TSKIP + comp][cu->zbase + zc]);
  }
}
/* whole-CU syntax: TEncCu.cpp:1636-1654 (RD) and xEncodeCU :1222-1270 (true encode); I-slice, no PCM/TQB/DQP */
static void enc_cu_syntax(enc_t *e, cabac_t *c, const cu_t *cu)
{
  if (cu->depth == 3) enc_bin(c, CTX_PART_SIZE, cu->part == SIZE_2Nx2N);
  code_luma_dirs(e, c, cu, 0, cu->part == SIZE_NxN ? 4 : 1);
  code_chroma_dir(e, c, cu);
  tu_t root = { cu->x, cu->y, cu->log2, 0, 0, cu->nparts };
  enc_transform(e, c, cu, &root);
}

/* =====================================================================================
 * TU coding: xIntraCodingTUBlock TEncSearch.cpp:1129-1424
 * mode012: 0 = predict, 1 = predict and save prediction, 2 = reuse saved prediction
 * ===================================================================================== */
static void stage_block_pel(const pel *v, int stride, int n)
{
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) fprintf(g_stage, "%d ", (int)v[j * stride + i]);
  fputc('\n', g_stage);
}

static void code_tu_block(enc_t *e, const cu_t *cu, const tu_t *tu, int comp, int mode012, uint32_t *dist)
{
  const int n = comp ? tu_csize(tu) : (1 << tu->log2);
  const int zrel = comp ? tu_czrel(tu) : tu->zrel;
  const int zabs = cu->zbase + zrel;
  const int x = comp ? tu->x >> 1 : tu->x, y = comp ? tu->y >> 1 : tu->y;     /* for 4x4-luma quads tu is the first quad */
  const int s = comp_stride(comp), bo = boff(e, comp, x, y);
  pel *pred = e->pred[comp] + bo, *resi = e->resi[comp] + bo;
  const pel *org = org_at(e, comp, x, y); const int os = pic_stride(e, comp);
  const int mode = mode_of(e, cu, comp, zrel);
  const int tskip = e->r->a[A_TSKIP + comp][zabs];
  if (mode012 != 2) {
    pel line[4 * 64 + 1], fline[4 * 64 + 1];
    build_refs(e, comp, x, y, n, line);
    const pel *use = line;
    if (use_filtered_refs(comp, mode, n)) { filter_refs(line, fline, n); use = fline; }
    predict_intra(comp, mode, use, n, pred, s);
    if (mode012 == 1) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) e->ts_pred[comp][j * n + i] = pred[j * s + i];
  } else for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) pred[j * s + i] = e->ts_pred[comp][j * n + i];
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) resi[j * s + i] = (pel)(org[j * os + i] - pred[j * s + i]);
  if (!comp) set_parts(e->r->a[A_TRIDX], zabs, tu->nparts, tu->trd);
  /* transform + RDOQ (transformNxN TComTrQuant.cpp:1450-1534) */
  int32_t *coef = e->coef_l[5 - tu->log2][comp] + (comp ? (zabs * 16) >> 2 : zabs * 16);
  static int32_t tc[1024];
  if (tskip) { for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) tc[j * n + i] = (int32_t)resi[j * s + i] << (13 - g_bd); }   /* transform shift 15 - bitDepth - 2 */
  else fwd_transform(resi, s, tc, n, !comp && n == 4);
  const int cbf_ctx = comp ? tu->trd : (tu->trd == 0 ? 1 : 0);
  if (g_stage) { fprintf(g_stage, "F %d %d\n", n, comp); stage_block_pel(resi, s, n); stage_block32(tc, n); }      /* TComTrQuant.cpp:1496-1516 */
  uint32_t abs_sum = rdoq(e, &e->go, comp, n, mode, tskip, cbf_ctx, tc, coef);
  if (g_stage) stage_block32(coef, n);                                                                               /* :1525-1528 */
  set_parts(e->r->a[A_CBF + comp], zabs, comp ? tu_cnparts(tu) : tu->nparts, (abs_sum > 0 ? 1 : 0) << tu->trd);
  if (abs_sum > 0) {
    if (g_stage) { fprintf(g_stage, "I %d %d\n", n, comp); stage_block32(coef, n); }                                 /* :1603-1606 */
    dequant(e, comp, n, coef, tc);
    if (g_stage) stage_block32(tc, n);                                                                               /* :1610-1613 */
    if (tskip) { for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) resi[j * s + i] = (pel)((tc[j * n + i] + (1 << (12 - g_bd))) >> (13 - g_bd)); }
    else inv_transform(tc, resi, s, n, !comp && n == 4);
    if (g_stage) stage_block_pel(resi, s, n);                                                                        /* :1658-1662 */
  } else {
    memset(coef, 0, sizeof(int32_t) * n * n);
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) resi[j * s + i] = 0;
  }
  pel *rq = e->rec_l[5 - tu->log2][comp] + bo, *rp = rec_at(e, comp, x, y); const int ps = pic_stride(e, comp);
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
    pel v = (pel)clip8(pred[j * s + i] + resi[j * s + i]);
    pred[j * s + i] = v; rq[j * s + i] = v; rp[j * ps + i] = v;
  }
  uint32_t d = sse(pred, s, org, os, n);
  if (comp) d = (uint32_t)(e->cweight * (double)d);       /* getDistPart TComRdCost.cpp:350-353 */
  *dist += d;
}

static void store_ts_result(enc_t *e, const cu_t *cu, const tu_t *tu, int comp)
{ /* xStoreIntraResultQT TEncSearch.cpp:1784-1816 (4x4 blocks only) */
  const int zabs = cu->zbase + (comp ? tu_czrel(tu) : tu->zrel);
  const int x = comp ? tu->x >> 1 : tu->x, y = comp ? tu->y >> 1 : tu->y, s = comp_stride(comp), bo = boff(e, comp, x, y);
  memcpy(e->ts_coef[comp], e->coef_l[5 - tu->log2][comp] + (comp ? (zabs * 16) >> 2 : zabs * 16), 16 * sizeof(int32_t));
  for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) e->ts_rec[comp][j * 4 + i] = e->rec_l[5 - tu->log2][comp][bo + j * s + i];
}
static void load_ts_result(enc_t *e, const cu_t *cu, const tu_t *tu, int comp)
{ /* xLoadIntraResultQT TEncSearch.cpp:1819-1870 */
  const int zabs = cu->zbase + (comp ? tu_czrel(tu) : tu->zrel);
  const int x = comp ? tu->x >> 1 : tu->x, y = comp ? tu->y >> 1 : tu->y, s = comp_stride(comp), bo = boff(e, comp, x, y);
  memcpy(e->coef_l[5 - tu->log2][comp] + (comp ? (zabs * 16) >> 2 : zabs * 16), e->ts_coef[comp], 16 * sizeof(int32_t));
  pel *rp = rec_at(e, comp, x, y); const int ps = pic_stride(e, comp);
  for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
    pel v = e->ts_rec[comp][j * 4 + i];
    e->rec_l[5 - tu->log2][comp][bo + j * s + i] = v; rp[j * ps + i] = v;
  }
}

/* xRecurIntraCodingLumaQT TEncSearch.cpp:1430-1738 */
static void recur_luma(enc_t *e, const cu_t *cu, const tu_t *tu, int check_first, uint32_t *dist_out, double *cost_out)
{
  const int full_depth = cu->depth + tu->trd;
  const int zabs = cu->zbase + tu->zrel;
  int check_full = tu->log2 <= 5;
  int check_split = tu->log2 > min_tu_log2(cu);
  if (check_first && check_full) check_split = 0;
  double single_cost = MAX_DOUBLE; uint32_t single_dist = 0, single_cbf = 0; int best_ts = 0;
  const int check_ts = (g_tools & 0x04u) && (tu->log2 == 2) && (cu->part == SIZE_NxN || !(g_tools & 0x08u));      /* TransformSkipFast: only in NxN CUs (TEncSearch.cpp:1502-1505) */
  if (check_full) {
    if (check_ts) {
      e->root[full_depth] = e->go;
      for (int m = 0; m < 2; m++) {
        uint32_t d = 0; double cost;
        set_parts(e->r->a[A_TSKIP + 0], zabs, tu->nparts, m);
        code_tu_block(e, cu, tu, 0, m == 0 ? 1 : 2, &d);
        uint32_t cbf = (e->r->a[A_CBF][zabs] >> tu->trd) & 1;
        if (m == 1 && cbf == 0) cost = MAX_DOUBLE;
        else { uint32_t bits = intra_bits_qt(e, cu, tu, 1, 0); cost = calc_rd_cost(e, bits, d); }
        if (cost < single_cost) {
          single_cost = cost; single_dist = d; single_cbf = cbf; best_ts = m;
          if (m == 0) { store_ts_result(e, cu, tu, 0); e->tbest[full_depth] = e->go; }
        }
        if (m == 0) e->go = e->root[full_depth];
      }
      set_parts(e->r->a[A_TSKIP + 0], zabs, tu->nparts, best_ts);
      if (best_ts == 0) {
        load_ts_result(e, cu, tu, 0);
        set_parts(e->r->a[A_CBF], zabs, tu->nparts, (int)(single_cbf << tu->trd));
        e->go = e->tbest[full_depth];
      }
    } else {
      if (check_split) e->root[full_depth] = e->go;
      set_parts(e->r->a[A_TSKIP + 0], zabs, tu->nparts, 0);
      code_tu_block(e, cu, tu, 0, 0, &single_dist);
      if (check_split) single_cbf = (e->r->a[A_CBF][zabs] >> tu->trd) & 1;
      uint32_t bits = intra_bits_qt(e, cu, tu, 1, 0);
      single_cost = calc_rd_cost(e, bits, single_dist);
    }
  }
  if (check_split) {
    if (check_full) { e->test[full_depth] = e->go; e->go = e->root[full_depth]; }
    else e->root[full_depth] = e->go;
    double split_cost = 0; uint32_t split_dist = 0, split_cbf = 0;
    for (int i = 0; i < 4; i++) {
      tu_t ch; tu_child(tu, i, &ch);
      recur_luma(e, cu, &ch, check_first, &split_dist, &split_cost);
      split_cbf |= (e->r->a[A_CBF][cu->zbase + ch.zrel] >> ch.trd) & 1;
    }
    if (split_cbf) for (int k = 0; k < tu->nparts; k++) e->r->a[A_CBF][zabs + k] |= (uint8_t)(1 << tu->trd);
    e->go = e->root[full_depth];
    uint32_t bits = intra_bits_qt(e, cu, tu, 1, 0);
    split_cost = calc_rd_cost(e, bits, split_dist);
    if (split_cost < single_cost) { *dist_out += split_dist; *cost_out += split_cost; return; }
    e->go = e->test[full_depth];
    set_parts(e->r->a[A_TRIDX], zabs, tu->nparts, tu->trd);
    set_parts(e->r->a[A_CBF], zabs, tu->nparts, (int)(single_cbf << tu->trd));
    set_parts(e->r->a[A_TSKIP + 0], zabs, tu->nparts, best_ts);
    const int n = 1 << tu->log2, bo = boff(e, 0, tu->x, tu->y);
    pel *rp = rec_at(e, 0, tu->x, tu->y);
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) rp[j * e->W + i] = e->rec_l[5 - tu->log2][0][bo + j * 64 + i];
  }
  *dist_out += single_dist; *cost_out += single_cost;
}

/* xSetIntraResultLumaQT / xSetIntraResultChromaQT TEncSearch.cpp:1741-1781, 2150-2198 */
static void set_result(enc_t *e, const cu_t *cu, const tu_t *tu, int comp)
{
  if (e->r->a[A_TRIDX][cu->zbase + tu->zrel] > tu->trd) {
    for (int i = 0; i < 4; i++) { tu_t ch; tu_child(tu, i, &ch); set_result(e, cu, &ch, comp); }
    return;
  }
  if (comp && !tu_has_chroma_first(tu)) return;
  const int n = comp ? tu_csize(tu) : (1 << tu->log2);
  const int zabs = cu->zbase + (comp ? tu_czrel(tu) : tu->zrel);
  const int off = comp ? (zabs * 16) >> 2 : zabs * 16;
  memcpy(e->r->coef[comp] + off, e->coef_l[5 - tu->log2][comp] + off, sizeof(int32_t) * n * n);
  const int x = comp ? tu->x >> 1 : tu->x, y = comp ? tu->y >> 1 : tu->y, s = comp_stride(comp), bo = boff(e, comp, x, y);
  for (int j = 0; j < n; j++) memcpy(e->best_rec[comp] + bo + j * s, e->rec_l[5 - tu->log2][comp] + bo + j * s, sizeof(pel) * n);
}

static uint32_t mode_bits_intra(enc_t *e, const cu_t *cu, int pu, int mode)
{ /* xModeBitsIntra TEncSearch.cpp:5530-5557 */
  const int z = cu->zbase + pu * (cu->nparts >> 2) * (cu->part == SIZE_NxN);
  e->go.frac = e->curr[cu->depth].frac;
  e->go.ctx[CTX_INTRA_PRED] = e->curr[cu->depth].ctx[CTX_INTRA_PRED];
  uint8_t keep = e->r->a[A_LDIR][z];
  e->r->a[A_LDIR][z] = (uint8_t)mode;
  reset_bits(&e->go);
  code_luma_dirs(e, &e->go, cu, pu, 1);
  e->r->a[A_LDIR][z] = keep;
  return get_bits(&e->go);
}

/* estIntraPredLumaQT TEncSearch.cpp:2203-2582 */
static void est_intra_luma(enc_t *e, const cu_t *cu, uint32_t *cu_dist)
{
  const int init_trd = cu->part == SIZE_NxN ? 1 : 0;
  const int npu = init_trd ? 4 : 1;
  const int pu_log2 = cu->log2 - init_trd, pn = 1 << pu_log2;
  const int pu_parts = cu->nparts >> (2 * init_trd);
  uint32_t overall = 0;
  uint8_t sv_tr[256], sv_cbf[3][256], sv_ts[3][256];
  for (int pu = 0; pu < npu; pu++) {
    const int poff = pu * pu_parts, zp = cu->zbase + poff;
    tu_t ptu = { cu->x + (pu & 1) * pn * init_trd, cu->y + (pu >> 1) * pn * init_trd, pu_log2, init_trd, poff, pu_parts };
    /* ---- rough mode decision :2266-2346 ---- */
    pel line[4 * 64 + 1], fline[4 * 64 + 1];
    build_refs(e, 0, ptu.x, ptu.y, pn, line);
    if (pn >= 8 && pn <= 32) filter_refs(line, fline, pn);
    static const uint8_t num_rd_cand_no_mpm[5] = { 9, 9, 4, 4, 5 };          /* g_aucIntraModeNumFast_NotUseMPM, TComRom.cpp:554-562 (TEncSearch.cpp:2269) */
    int nfull = (g_tools & 0x40u) ? g_num_rd_cand[pu_log2 - 2] : num_rd_cand_no_mpm[pu_log2 - 2];
    uint32_t rd_list[16]; double cost_list[16];
    for (int i = 0; i < nfull; i++) cost_list[i] = MAX_DOUBLE;
    const pel *org = org_at(e, 0, ptu.x, ptu.y);
    pel *pred = e->pred[0] + boff(e, 0, ptu.x, ptu.y);
    for (int mode = 0; mode < 35; mode++) {
      predict_intra(0, mode, use_filtered_refs(0, mode, pn) ? fline : line, pn, pred, 64);
      uint32_t sad = satd(org, e->W, pred, 64, pn);
      uint32_t mb = mode_bits_intra(e, cu, pu, mode);
      double cost = (double)sad + (double)mb * e->sqrt_lambda;
      if (g_stage) fprintf(g_stage, "R %d %u %u %.17g\n", mode, sad, mb, cost);                                      /* TEncSearch.cpp:2315-2317 */
      /* xUpdateCandList :5562-5585 */
      int shift = 0;
      while (shift < nfull && cost < cost_list[nfull - 1 - shift]) shift++;
      if (shift) {
        for (int i = 1; i < shift; i++) { rd_list[nfull - i] = rd_list[nfull - 1 - i]; cost_list[nfull - i] = cost_list[nfull - 1 - i]; }
        rd_list[nfull - shift] = (uint32_t)mode; cost_list[nfull - shift] = cost;
      }
    }
    if (g_tools & 0x40u) {                       /* FastUDIUseMPMEnabled, TEncSearch.cpp:2324-2346 */
      int preds[3], nm; get_mpm(e, ptu.x, ptu.y, preds, &nm);
      const int nbase = nfull;
      for (int j = 0; j < nm; j++) {
        int inc = 0;
        for (int i = 0; i < nbase; i++) inc |= (preds[j] == (int)rd_list[i]);
        /* reference compares against the first numModesForFullRD entries, which grows as MPMs are appended */
        for (int i = nbase; i < nfull; i++) inc |= (preds[j] == (int)rd_list[i]);
        if (!inc) rd_list[nfull++] = (uint32_t)preds[j];
      }
    }
    /* ---- RD pass 1 :2355-2443 ---- */
    uint32_t best_mode = 0, best_dist = 0; double best_cost = MAX_DOUBLE;
    for (int m = 0; m <= nfull; m++) {
      const int second = (m == nfull);                     /* pass 2 :2445-2512 */
      const uint32_t org_mode = second ? best_mode : rd_list[m];
      set_parts(e->r->a[A_LDIR], zp, pu_parts, (int)org_mode);
      e->go = e->curr[cu->depth];
      uint32_t d = 0; double cost = 0.0;
      recur_luma(e, cu, &ptu, !second, &d, &cost);
      if (g_stage && !second) fprintf(g_stage, "P %u %.17g\n", org_mode, cost);                                      /* TEncSearch.cpp:2395-2397 (printed for the first loop only) */
      if (cost < best_cost) {
        best_mode = org_mode; best_dist = d; best_cost = cost;
        set_result(e, cu, &ptu, 0);
        memcpy(sv_tr, e->r->a[A_TRIDX] + zp, pu_parts);
        for (int c = 0; c < 3; c++) { memcpy(sv_cbf[c], e->r->a[A_CBF + c] + zp, pu_parts); memcpy(sv_ts[c], e->r->a[A_TSKIP + c] + zp, pu_parts); }
      }
    }
    overall += best_dist;
    memcpy(e->r->a[A_TRIDX] + zp, sv_tr, pu_parts);
    for (int c = 0; c < 3; c++) { memcpy(e->r->a[A_CBF + c] + zp, sv_cbf[c], pu_parts); memcpy(e->r->a[A_TSKIP + c] + zp, sv_ts[c], pu_parts); }
    if (pu != npu - 1) {
      pel *rp = rec_at(e, 0, ptu.x, ptu.y); const int bo = boff(e, 0, ptu.x, ptu.y);
      for (int j = 0; j < pn; j++) for (int i = 0; i < pn; i++) rp[j * e->W + i] = e->best_rec[0][bo + j * 64 + i];
    }
    set_parts(e->r->a[A_LDIR], zp, pu_parts, (int)best_mode);
  }
  if (npu > 1) {
    int comb[3] = { 0, 0, 0 };
    for (int p = 0; p < 4; p++) for (int c = 0; c < 3; c++) comb[c] |= (e->r->a[A_CBF + c][cu->zbase + p * pu_parts] >> 1) & 1;
    for (int k = 0; k < cu->nparts; k++) for (int c = 0; c < 3; c++) e->r->a[A_CBF + c][cu->zbase + k] |= (uint8_t)comb[c];
  }
  e->go = e->curr[cu->depth];
  *cu_dist = overall;
}

/* xRecurIntraChromaCodingQT TEncSearch.cpp:1941-2145 */
static void recur_chroma(enc_t *e, const cu_t *cu, const tu_t *tu, uint32_t *dist_out)
{
  const int z = cu->zbase + tu->zrel;
  if (e->r->a[A_TRIDX][z] == tu->trd) {
    if (!tu_has_chroma_first(tu)) return;
    const int full_depth = cu->depth + tu->trd;
    /* a 4x4 chroma block (under a luma TU of 8x8, or of four 4x4); TransformSkipFast: only under 4x4 luma TUs of which one was transform-skipped (TEncSearch.cpp:1965-1990) */
    int check_ts = (g_tools & 0x04u) && ((g_tools & 0x08u) ? tu->log2 == 2 : tu_csize(tu) == 4);
    if (check_ts && (g_tools & 0x08u)) { int nb = 0; for (int k = 0; k < 4; k++) nb += e->r->a[A_TSKIP + 0][z + k]; check_ts = nb > 0; }
    const int zc = cu->zbase + tu_czrel(tu), np = tu_cnparts(tu);
    for (int comp = 1; comp < 3; comp++) {
      e->root[full_depth] = e->go;
      double single_cost = MAX_DOUBLE, cost_tmp = 0; int best_id = 0, best_ts = 0; uint32_t single_dist = 0, single_cbf = 0;
      const int total = check_ts ? 2 : 1; int cur_id = 0;
      for (int ts = 0; ts < total; ts++) {
        set_parts(e->r->a[A_TSKIP + comp], zc, np, ts);
        cur_id++;
        const int one = (total == 1), last = (cur_id == total);
        const int m012 = one ? 0 : (ts == 0 ? 1 : 2);
        uint32_t d = 0;
        code_tu_block(e, cu, tu, comp, m012, &d);
        uint32_t cbf = (e->r->a[A_CBF + comp][zc] >> tu->trd) & 1;
        if (ts == 1 && cbf == 0) cost_tmp = MAX_DOUBLE;
        else if (!one) {
          reset_bits(&e->go); enc_coeff_qt(e, &e->go, cu, tu, comp, 0);     /* xGetIntraBitsQTChroma :1119-1127 */
          cost_tmp = calc_rd_cost(e, get_bits(&e->go), d);
        }
        if (cost_tmp < single_cost) {
          single_cost = cost_tmp; single_dist = d; best_ts = ts; best_id = cur_id; single_cbf = cbf;
          if (!one && !last) { store_ts_result(e, cu, tu, comp); e->tbest[full_depth] = e->go; }
        }
        if (!one && !last) e->go = e->root[full_depth];
      }
      if (best_id < total) {
        load_ts_result(e, cu, tu, comp);
        set_parts(e->r->a[A_CBF + comp], zc, np, (int)(single_cbf << tu->trd));
        e->go = e->tbest[full_depth];
      }
      set_parts(e->r->a[A_TSKIP + comp], zc, np, best_ts);
      *dist_out += single_dist;
    }
  } else {
    uint32_t split_cbf[3] = { 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
      tu_t ch; tu_child(tu, i, &ch);
      recur_chroma(e, cu, &ch, dist_out);
      for (int comp = 1; comp < 3; comp++) split_cbf[comp] |= (e->r->a[A_CBF + comp][cu->zbase + ch.zrel] >> ch.trd) & 1;
    }
    for (int comp = 1; comp < 3; comp++) if (split_cbf[comp])
      for (int k = 0; k < tu->nparts; k++) e->r->a[A_CBF + comp][z + k] |= (uint8_t)(1 << tu->trd);
  }
}

/* estIntraPredChromaQT TEncSearch.cpp:2588-2737 (4:2:0: one chroma PU per CU) */
static void est_intra_chroma(enc_t *e, const cu_t *cu, uint32_t *cu_dist)
{
  tu_t root = { cu->x, cu->y, cu->log2, 0, 0, cu->nparts };
  uint32_t mode_list[5] = { PLANAR, VER, HOR, DC, DM_CHROMA };
  const int luma_mode = e->r->a[A_LDIR][cu->zbase];
  for (int i = 0; i < 4; i++) if ((int)mode_list[i] == luma_mode) { mode_list[i] = 34; break; }   /* getAllowedChromaDir */
  uint32_t best_mode = 0, best_dist = 0; double best_cost = MAX_DOUBLE;
  uint8_t sv_cbf[3][256], sv_ts[3][256];
  for (int m = 0; m < 5; m++) {
    e->go = e->curr[cu->depth];
    uint32_t d = 0;
    set_parts(e->r->a[A_CDIR], cu->zbase, cu->nparts, (int)mode_list[m]);
    recur_chroma(e, cu, &root, &d);
    e->go = e->curr[cu->depth];                              /* TransformSkip enabled :2648-2651 */
    uint32_t bits = intra_bits_qt(e, cu, &root, 0, 1);
    double cost = calc_rd_cost(e, bits, d);
    if (cost < best_cost) {
      best_cost = cost; best_dist = d; best_mode = mode_list[m];
      set_result(e, cu, &root, 1); set_result(e, cu, &root, 2);
      for (int c = 1; c < 3; c++) { memcpy(sv_cbf[c], e->r->a[A_CBF + c] + cu->zbase, cu->nparts); memcpy(sv_ts[c], e->r->a[A_TSKIP + c] + cu->zbase, cu->nparts); }
    }
  }
  for (int c = 1; c < 3; c++) { memcpy(e->r->a[A_CBF + c] + cu->zbase, sv_cbf[c], cu->nparts); memcpy(e->r->a[A_TSKIP + c] + cu->zbase, sv_ts[c], cu->nparts); }
  set_parts(e->r->a[A_CDIR], cu->zbase, cu->nparts, (int)best_mode);
  *cu_dist += best_dist;
  e->go = e->curr[cu->depth];
}

typedef struct { double cost; uint32_t bits, dist; } rd_t;

static void copy_best_rec_to_pic(enc_t *e, const cu_t *cu, int comp)
{
  const int n = (1 << cu->log2) >> (comp ? 1 : 0), x = cu->x >> (comp ? 1 : 0), y = cu->y >> (comp ? 1 : 0);
  const int s = comp_stride(comp), bo = boff(e, comp, x, y), ps = pic_stride(e, comp);
  pel *rp = rec_at(e, comp, x, y);
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) rp[j * ps + i] = e->best_rec[comp][bo + j * s + i];
}

/* xCheckRDCostIntra TEncCu.cpp:1600-1665; result state left in e->temp[depth] */
static rd_t check_rd_cost_intra(enc_t *e, cu_t *cu, int part)
{
  rd_t r;
  cu->part = part;
  /* initEstData TComDataCU.cpp:525-592 + part/pred mode */
  set_parts(e->r->a[A_DEPTH], cu->zbase, cu->nparts, cu->depth);
  set_parts(e->r->a[A_PART], cu->zbase, cu->nparts, part);
  set_parts(e->r->a[A_LDIR], cu->zbase, cu->nparts, DC);
  set_parts(e->r->a[A_CDIR], cu->zbase, cu->nparts, 0);
  set_parts(e->r->a[A_TRIDX], cu->zbase, cu->nparts, 0);
  for (int c = 0; c < 3; c++) { set_parts(e->r->a[A_CBF + c], cu->zbase, cu->nparts, 0); set_parts(e->r->a[A_TSKIP + c], cu->zbase, cu->nparts, 0); }
  uint32_t dist = 0;
  est_intra_luma(e, cu, &dist);
  copy_best_rec_to_pic(e, cu, 0);
  est_intra_chroma(e, cu, &dist);
  reset_bits(&e->go);
  enc_cu_syntax(e, &e->go, cu);
  e->temp[cu->depth] = e->go;
  r.bits = get_bits(&e->go); r.dist = dist; r.cost = calc_rd_cost(e, r.bits, r.dist);
  return r;
}

/* saved 8x8 CU candidate (2Nx2N vs NxN at depth 3) */
typedef struct { uint8_t a[11][4]; int32_t coef[3][64]; pel rec[3][64]; } cand8_t;
static void save_cand8(enc_t *e, const cu_t *cu, cand8_t *s)
{
  for (int f = 0; f < 11; f++) memcpy(s->a[f], e->r->a[f] + cu->zbase, 4);
  memcpy(s->coef[0], e->r->coef[0] + cu->zbase * 16, 64 * sizeof(int32_t));
  for (int c = 1; c < 3; c++) memcpy(s->coef[c], e->r->coef[c] + cu->zbase * 4, 16 * sizeof(int32_t));
  for (int c = 0; c < 3; c++) {
    int n = c ? 4 : 8, sx = c ? 32 : 64, bo = boff(e, c, cu->x >> (c ? 1 : 0), cu->y >> (c ? 1 : 0));
    for (int j = 0; j < n; j++) memcpy(s->rec[c] + j * n, e->best_rec[c] + bo + j * sx, n * sizeof(pel));
  }
}
static void load_cand8(enc_t *e, const cu_t *cu, const cand8_t *s)
{
  for (int f = 0; f < 11; f++) memcpy(e->r->a[f] + cu->zbase, s->a[f], 4);
  memcpy(e->r->coef[0] + cu->zbase * 16, s->coef[0], 64 * sizeof(int32_t));
  for (int c = 1; c < 3; c++) memcpy(e->r->coef[c] + cu->zbase * 4, s->coef[c], 16 * sizeof(int32_t));
  for (int c = 0; c < 3; c++) {
    int n = c ? 4 : 8, sx = c ? 32 : 64, bo = boff(e, c, cu->x >> (c ? 1 : 0), cu->y >> (c ? 1 : 0));
    for (int j = 0; j < n; j++) memcpy(e->best_rec[c] + bo + j * sx, s->rec[c] + j * n, n * sizeof(pel));
  }
}

/* xCompressCU TEncCu.cpp:470-1104 with the reference's label-pruning edits (:496-520, 815-834, 947-965) */
static rd_t compress_cu(enc_t *e, int x, int y, int depth)
{
  const int log2 = 6 - depth, size = 1 << log2;
  cu_t cu = { x, y, log2, depth, g_r2z[(((y & 63) >> 2) << 4) | ((x & 63) >> 2)], 256 >> (2 * depth), SIZE_2Nx2N };
  const int boundary = !(x + size <= e->W && y + size <= e->H);
  const int pred_depth = e->labels[e->addr * 16 + 4 * ((y & 63) / 16) + (x & 63) / 16];
  const int check_cur = pred_depth == depth, check_next = pred_depth > depth;
  rd_t best = { MAX_DOUBLE, 0, 0 };
  int best_is_real = 0;
  if (!boundary) {
    if (check_cur) {
      rd_t t = check_rd_cost_intra(e, &cu, SIZE_2Nx2N);
      if (t.cost < best.cost) { best = t; e->next[depth] = e->temp[depth]; best_is_real = 1; }
      if (depth == 3) {
        cand8_t keep; save_cand8(e, &cu, &keep);
        rd_t t2 = check_rd_cost_intra(e, &cu, SIZE_NxN);
        if (t2.cost < best.cost) { best = t2; e->next[depth] = e->temp[depth]; }
        else { load_cand8(e, &cu, &keep); cu.part = SIZE_2Nx2N; }
      }
    } else {
      best.cost = MAX_DOUBLE / 16; best.dist = 0xffffffffu >> 3; best.bits = 0xffffffffu >> 3;
    }
    /* split flag of the unsplit candidate :858-867 (state is stale for the dummy candidate; harmless) */
    e->go = e->next[depth];
    reset_bits(&e->go);
    if (depth < 3) enc_bin(&e->go, CTX_SPLIT + split_ctx(e, x, y, depth), 0);
    best.bits += get_bits(&e->go);
    best.cost = calc_rd_cost(e, best.bits, best.dist);
    e->next[depth] = e->go;
  }
  if (best_is_real) {            /* final luma+chroma recon of the chosen candidate -> picture (xCopyYuv2Pic :1093) */
    for (int c = 0; c < 3; c++) copy_best_rec_to_pic(e, &cu, c);
  }
  if (depth < 3) {
    rd_t temp = { 0, 0, 0 };
    const int h = size >> 1, qn = cu.nparts >> 2;
    int any = 0;
    for (int i = 0; i < 4; i++) {
      int sx = x + (i & 1) * h, sy = y + (i >> 1) * h;
      if (sx < e->W && sy < e->H) {
        e->curr[depth + 1] = (i == 0) ? e->curr[depth] : e->next[depth + 1];
        rd_t s;
        if (check_next) s = compress_cu(e, sx, sy, depth + 1);
        else { s.cost = MAX_DOUBLE / 16; s.dist = 0xffffffffu >> 3; s.bits = 0xffffffffu >> 3; }
        temp.cost += s.cost; temp.dist += s.dist; temp.bits += s.bits; any = 1;
      } else if (check_next || boundary) {
        /* initSubCU defaults copied to the picture :989 */
        int z0 = cu.zbase + i * qn;
        set_parts(e->r->a[A_DEPTH], z0, qn, depth + 1); set_parts(e->r->a[A_PART], z0, qn, SIZE_NONE);
        set_parts(e->r->a[A_LDIR], z0, qn, DC); set_parts(e->r->a[A_CDIR], z0, qn, 0); set_parts(e->r->a[A_TRIDX], z0, qn, 0);
        for (int c = 0; c < 3; c++) { set_parts(e->r->a[A_CBF + c], z0, qn, 0); set_parts(e->r->a[A_TSKIP + c], z0, qn, 0); }
      }
    }
    (void)any;
    e->go = e->next[depth + 1];
    if (!boundary) {
      reset_bits(&e->go);
      enc_bin(&e->go, CTX_SPLIT + split_ctx(e, x, y, depth), 1);
      temp.bits += get_bits(&e->go);
    }
    temp.cost = calc_rd_cost(e, temp.bits, temp.dist);
    e->temp[depth] = e->go;
    if (temp.cost < best.cost) { best = temp; e->next[depth] = e->temp[depth]; }
  }
  return best;
}

/* true-state encode of the decided CTU: encodeCtu/xEncodeCU TEncCu.cpp:290-304,1167-1271 + finishCU :1112-1128 */
static void encode_cu_tree(enc_t *e, cabac_t *c, int x, int y, int depth)
{
  const int size = 64 >> depth;
  const int z = g_r2z[(((y & 63) >> 2) << 4) | ((x & 63) >> 2)];
  int boundary = 0;
  if (x + size <= e->W && y + size <= e->H) { if (depth < 3) enc_bin(c, CTX_SPLIT + split_ctx(e, x, y, depth), e->r->a[A_DEPTH][z] > depth); }
  else boundary = 1;
  if ((depth < e->r->a[A_DEPTH][z] && depth < 3) || boundary) {
    const int h = size >> 1;
    for (int i = 0; i < 4; i++) { int sx = x + (i & 1) * h, sy = y + (i >> 1) * h; if (sx < e->W && sy < e->H) encode_cu_tree(e, c, sx, sy, depth + 1); }
    return;
  }
  cu_t cu = { x, y, 6 - depth, depth, z, 256 >> (2 * depth), e->r->a[A_PART][z] };
  enc_cu_syntax(e, c, &cu);
}

static void compress_ctu(enc_t *e, cabac_t *truec, int last_ctu)
{
  e->r = e->recs + e->addr;
  /* initCtu TComDataCU.cpp:420-500 */
  memset(e->r, 0, sizeof *e->r);
  memset(e->r->a[A_PART], SIZE_NONE, 256); memset(e->r->a[A_LDIR], DC, 256);
  e->curr[0] = *truec; e->go = *truec;                    /* TEncSlice.cpp:826-832 */
  rd_t best = compress_cu(e, e->cx * 64, e->cy * 64, 0);
  e->r->bits = best.bits; e->r->dist = best.dist; e->r->cost = best.cost;
  /* TEncSlice.cpp:886-893 */
  reset_bits(truec);
  encode_cu_tree(e, truec, e->cx * 64, e->cy * 64, 0);
  if (!last_ctu) enc_trm(truec, 0);
  e->est_bits += get_bits(truec);
}

int hm_oracle_encode_frames(const uint8_t *yuv, int width, int height, int n_frames, int qp,
                            const uint8_t *labels, hm_ctu_record *out_recs, uint8_t *recon,
                            hm_frame_stats *stats)
{
  return hm_oracle_encode_frames_tiles(yuv, width, height, n_frames, qp, labels, out_recs, recon, stats, 1, 1);
}

int hm_oracle_encode_frames_tiles(const uint8_t *yuv, int width, int height, int n_frames, int qp,
                                  const uint8_t *labels, hm_ctu_record *out_recs, uint8_t *recon,
                                  hm_frame_stats *stats, int tile_cols, int tile_rows)
{
  return hm_oracle_encode_frames_ex(yuv, width, height, n_frames, qp, labels, out_recs, recon, stats, tile_cols, tile_rows, 8);
}

int hm_oracle_encode_frames_ex(const void *yuv_, int width, int height, int n_frames, int qp,
                               const uint8_t *labels, hm_ctu_record *out_recs, void *recon_,
                               hm_frame_stats *stats, int tile_cols, int tile_rows, int bit_depth)
{ /* uniform spacing, TComPicSym.cpp:220-260 */
  int col_bd[64], row_bd[64], t;
  const int cx = (width + 63) >> 6, cy = (height + 63) >> 6;
  if (tile_cols < 1 || tile_rows < 1 || tile_cols > 20 || tile_rows > 22) return -1;
  for (t = 0; t <= tile_cols; t++) col_bd[t] = (t * cx) / tile_cols;
  for (t = 0; t <= tile_rows; t++) row_bd[t] = (t * cy) / tile_rows;
  return hm_oracle_encode_frames_tb(yuv_, width, height, n_frames, qp, labels, out_recs, recon_, stats, tile_cols, tile_rows, col_bd, row_bd, bit_depth);
}

int hm_oracle_encode_frames_tb(const void *yuv_, int width, int height, int n_frames, int qp,
                               const uint8_t *labels, hm_ctu_record *out_recs, void *recon_,
                               hm_frame_stats *stats, int tile_cols, int tile_rows, const int *col_bd, const int *row_bd, int bit_depth)
{
  const uint8_t *yuv = (const uint8_t *)yuv_; uint8_t *recon = (uint8_t *)recon_;
  const int wide = bit_depth > 8;                /* samples are uint16 (little endian) */
  if (bit_depth != 8 && bit_depth != 10) return -1;
  g_bd = bit_depth;
  if (tile_cols < 1 || tile_rows < 1 || tile_cols > (width + 63) >> 6 || tile_rows > (height + 63) >> 6) return -1;
  if (width <= 0 || height <= 0 || (width & 7) || (height & 7) || qp < 0 || qp > 51) return -1;
  init_tables();
  enc_t *e = (enc_t *)calloc(1, sizeof *e);
  if (!e) return -2;
  e->W = width; e->H = height; e->cw = width >> 1; e->qp = qp;
  e->ctus_x = (width + 63) >> 6; e->ctus_y = (height + 63) >> 6;
  const int nctu = e->ctus_x * e->ctus_y;
  /* lambda: TEncSlice.cpp:433-527 (all-intra GOP 1) and setUpLambda :112-140 */
  e->lambda = 0.57 * 1.0 * pow(2.0, (qp - 12) / 3.0);
  e->sqrt_lambda = sqrt(e->lambda);
  e->qp_c = g_chroma_scale_420[qp < 0 ? 0 : (qp > 57 ? 57 : qp)];
  e->cweight = pow(2.0, (qp - e->qp_c) / 3.0);
  e->lambda_c = e->lambda / e->cweight;
  for (int ch = 0; ch < 2; ch++) for (int l = 0; l < 4; l++) {
    int tshift = 15 - g_bd - (l + 2), rem = (ch ? e->qp_c : qp) % 6;      /* (qp + 6k) % 6 == qp % 6 */
    double s = (double)(1 << 15);
    s = s * pow(2.0, -2.0 * tshift);
    e->err_scale[ch][l] = s / g_quant_scales[rem] / g_quant_scales[rem] / (1 << DIST_ADJ(2 * (g_bd - 8)));
  }
  const size_t ysz = (size_t)width * height, csz = ysz >> 2, fsz = ysz + 2 * csz;
  for (int c = 0; c < 3; c++) {
    e->org[c] = (pel *)malloc(sizeof(pel) * (c ? csz : ysz));
    e->rec[c] = (pel *)calloc(c ? csz : ysz, sizeof(pel));
  }
  e->recs = (irec_t *)malloc(sizeof(irec_t) * nctu);
  for (int f = 0; f < n_frames; f++) {
    const uint8_t *src = yuv + (size_t)f * fsz * (wide ? 2 : 1);
    const uint16_t *src16 = (const uint16_t *)src;
    for (size_t i = 0; i < ysz; i++) e->org[0][i] = wide ? (pel)src16[i] : (pel)src[i];
    for (size_t i = 0; i < csz; i++) { e->org[1][i] = wide ? (pel)src16[ysz + i] : (pel)src[ysz + i]; e->org[2][i] = wide ? (pel)src16[ysz + csz + i] : (pel)src[ysz + csz + i]; }
    for (int c = 0; c < 3; c++) memset(e->rec[c], 0, sizeof(pel) * (c ? csz : ysz));
    e->labels = labels + (size_t)f * nctu * 16;
    e->est_bits = 0;
    /* CTUs in tile scan (tiles in raster order, CTUs in raster order inside a tile; boundaries from the caller);
       the coder is re-initialised at the first CTU of every tile (TEncSlice.cpp:719-720, 804-807) */
    for (int tr = 0; tr < tile_rows; tr++) for (int tc = 0; tc < tile_cols; tc++) {
      const int cx0 = col_bd[tc], cx1 = col_bd[tc + 1], cy0 = row_bd[tr], cy1 = row_bd[tr + 1];
      e->tx0 = cx0 * 64; e->ty0 = cy0 * 64; e->tx1 = cx1 * 64; e->ty1 = cy1 * 64;
      cabac_t truec; cabac_init(&truec, qp);
      cabac_t sync; cabac_init(&sync, qp);           /* m_entropyCodingSyncContextState */
      for (int cy = cy0; cy < cy1; cy++) for (int cx = cx0; cx < cx1; cx++) {
        e->addr = cy * e->ctus_x + cx; e->cx = cx; e->cy = cy;
        if (g_wpp && cx == cx0 && cy > cy0) {        /* TEncSlice.cpp:808-823: resetEntropy (contexts from the QP, fraction 0), then the contexts -- not the fraction -- of the sync state */
          cabac_init(&truec, qp);
          if (cx + 1 < cx1) memcpy(truec.ctx, sync.ctx, sizeof truec.ctx);     /* the CTU above and to the right lies in this tile (and slice) */
        }
        compress_ctu(e, &truec, e->addr == nctu - 1);
        if (g_wpp && cx == cx0 + 1) memcpy(sync.ctx, truec.ctx, sizeof sync.ctx);      /* :925-928 */
      }
    }
    for (int a = 0; a < nctu; a++) {
      hm_ctu_record *o = out_recs + (size_t)f * nctu + a; const irec_t *r = e->recs + a;
      memcpy(o->depth, r->a[A_DEPTH], 256); memcpy(o->part_size, r->a[A_PART], 256);
      memcpy(o->luma_dir, r->a[A_LDIR], 256); memcpy(o->chroma_dir, r->a[A_CDIR], 256); memcpy(o->tr_idx, r->a[A_TRIDX], 256);
      for (int c = 0; c < 3; c++) { memcpy(o->cbf[c], r->a[A_CBF + c], 256); memcpy(o->tskip[c], r->a[A_TSKIP + c], 256); }
      o->bits = r->bits; o->dist = r->dist; o->cost = r->cost;
      for (int i = 0; i < 4096; i++) o->coeff_y[i] = (int16_t)r->coef[0][i];
      for (int i = 0; i < 1024; i++) { o->coeff_cb[i] = (int16_t)r->coef[1][i]; o->coeff_cr[i] = (int16_t)r->coef[2][i]; }
    }
    if (recon) {
      uint8_t *dst = recon + (size_t)f * fsz * (wide ? 2 : 1);
      uint16_t *dst16 = (uint16_t *)dst;
      if (wide) {
        for (size_t i = 0; i < ysz; i++) dst16[i] = (uint16_t)e->rec[0][i];
        for (size_t i = 0; i < csz; i++) { dst16[ysz + i] = (uint16_t)e->rec[1][i]; dst16[ysz + csz + i] = (uint16_t)e->rec[2][i]; }
      } else {
        for (size_t i = 0; i < ysz; i++) dst[i] = (uint8_t)e->rec[0][i];
        for (size_t i = 0; i < csz; i++) { dst[ysz + i] = (uint8_t)e->rec[1][i]; dst[ysz + csz + i] = (uint8_t)e->rec[2][i]; }
      }
    }
    if (stats) {
      hm_frame_stats *s = stats + f; memset(s, 0, sizeof *s);
      for (int c = 0; c < 3; c++) { size_t n = c ? csz : ysz; uint64_t acc = 0;
        for (size_t i = 0; i < n; i++) { int d = e->org[c][i] - e->rec[c][i]; acc += (uint64_t)(d * d); } s->sse[c] = acc; }
      s->est_bits = e->est_bits; s->ctus = (uint32_t)nctu;
    }
  }
  for (int c = 0; c < 3; c++) { free(e->org[c]); free(e->rec[c]); }
  free(e->recs); free(e);
  return 0;
}
