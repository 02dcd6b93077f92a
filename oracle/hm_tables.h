/* oracle/hm_tables.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Constant tables of the HEVC intra coding path as used by the reference
 * (HM 16.20 as shipped in /root/reference/HM_dl).  Values are the ones of the
 * HEVC standard / HM; each table cites the reference location it restates.
 */
#ifndef HM_TABLES_H
#define HM_TABLES_H
#include <stdint.h>

/* ---- CABAC context layout (own flat layout; reference order: TEncSbac.cpp:62-92) ---- */
enum {
  CTX_SPLIT      = 0,   /* 3  INIT_SPLIT_FLAG            ContextTables.h (I row) */
  CTX_PART_SIZE  = 3,   /* 1  INIT_PART_SIZE[0]                                  */
  CTX_INTRA_PRED = 4,   /* 1  INIT_INTRA_PRED_MODE                               */
  CTX_CHROMA_PRED= 5,   /* 1  INIT_CHROMA_PRED_MODE[0]                           */
  CTX_QT_CBF     = 6,   /* 10 INIT_QT_CBF  (5 luma, 5 chroma)                    */
  CTX_SUBDIV     = 16,  /* 3  INIT_TRANS_SUBDIV_FLAG                             */
  CTX_SIG_CG     = 19,  /* 4  INIT_SIG_CG_FLAG (2 luma, 2 chroma)                */
  CTX_SIG        = 23,  /* 44 INIT_SIG_FLAG (28 luma, 16 chroma)                 */
  CTX_LAST_X     = 67,  /* 30 INIT_LAST (15 luma, 15 chroma)                     */
  CTX_LAST_Y     = 97,  /* 30 INIT_LAST                                          */
  CTX_ONE        = 127, /* 24 INIT_ONE_FLAG (16 luma, 8 chroma)                  */
  CTX_ABS        = 151, /* 6  INIT_ABS_FLAG (4 luma, 2 chroma)                   */
  CTX_TSKIP      = 157, /* 2  INIT_TRANSFORMSKIP_FLAG (luma, chroma)             */
  NUM_CTX        = 159
};

/* I-slice initialisation values (ContextTables.h:181-480, third row of each table) */
static const uint8_t g_ctx_init[NUM_CTX] = {
  /* split      */ 139, 141, 157,
  /* part size  */ 184,
  /* intra pred */ 184,
  /* chroma pred*/ 63,
  /* qt cbf     */ 111, 141, 154, 154, 154,   94, 138, 182, 154, 154,
  /* subdiv     */ 153, 138, 138,
  /* sig cg     */ 91, 171, 134, 141,
  /* sig luma   */ 111,  111, 125, 110, 110,  94, 124, 108, 124,  107, 125, 141, 179, 153, 125,  107, 125, 141, 179, 153, 125,
                   107, 125, 141, 179, 153, 125,  141,
  /* sig chroma */ 140,  139, 182, 182, 152, 136, 152, 136, 153,  136, 139, 111,  136, 139, 111,  111,
  /* last x     */ 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,
                   108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* last y     */ 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,
                   108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* one luma   */ 140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152,
  /* one chroma */ 140, 179, 166, 182, 140, 227, 122, 197,
  /* abs        */ 138, 153, 136, 167, 152, 152,
  /* tskip      */ 139, 139
};

/* ContextModel.cpp:68-101 */
static const uint8_t g_next_mps[128] = {
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,
  18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49,
  50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81,
  82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113,
  114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127
};
static const uint8_t g_next_lps[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11,
  12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37,
  38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59,
  58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73,
  72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127
};
/* ContextModel.cpp:103-112 (FAST_BIT_EST table, TypeDef.h:125) : 15-bit fractional bits */
static const int32_t g_entropy_bits[128] = {
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb
};

/* TComRom.cpp:354-362 */
static const int g_quant_scales[6]     = { 26214, 23302, 20560, 18396, 16384, 14564 };
static const int g_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };
/* TComRom.cpp:536 (4:2:0 row of g_aucChromaScale) */
static const uint8_t g_chroma_scale_420[58] = {
  0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };
/* TComRom.cpp:597-598 */
static const uint8_t g_min_in_group[10] = { 0,1,2,3,4,6,8,12,16,24 };
static const uint8_t g_group_idx[32]    = { 0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9 };
/* TComRom.cpp:589-595 */
static const uint8_t g_ctx_ind_map_4x4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
/* TComRom.cpp:545-553 : RD candidates kept after RMD, by log2(PU size)-2 (4,8,16,32,64) */
static const uint8_t g_num_rd_cand[5] = { 8, 8, 3, 3, 3 };
/* TComPrediction.cpp:50-58 (luma row), by log2(size)-2 */
static const uint8_t g_intra_filter_thr[5] = { 10, 7, 1, 0, 10 };
/* TComPrediction.cpp:265-266 */
static const int g_ang_table[9]     = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int g_inv_ang_table[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
/* DST-VII 4x4, TComRom.cpp:368-374,475-479 */
static const int8_t g_dst4[4][4] = { {29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29} };
/* magnitudes of the 32-point DCT basis: c[j] ~ 64*sqrt(2)*cos(j*pi/64) as tuned in the standard
 * (TComRom.cpp:376-456, coefficient lists :489-517); T_N[k][n] = +-c[(2n+1)k * 32/N folded]. */
static const int8_t g_dct_mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                      61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
#endif
