// hevcdl_dev.h -- internal host<->kernel parameter blocks (not part of the C ABI).
#ifndef HEVCDL_DEV_H
#define HEVCDL_DEV_H
#include <stdint.h>
#include <stddef.h>

#define HEVCDL_DEV_INPUT_RGB601  0
#define HEVCDL_DEV_INPUT_LUMA    1
#define HEVCDL_DEV_INPUT_RGB_CTU 2

// packed CNN weights (units of floats): every conv is split-f16 MFMA B operands ([N-tile][tap or k-step][hi | lo][64 lanes][8 halves], hevcdl_api.hip pack_conv*)
// + bias + gamma + beta as floats, every fc is [k][j] floats + bias
#define HEVCDL_W_C5   2560         // halves of a 5x5 conv: 5 k-steps x (B1 + B2) x 1 KB = 2560 floats, then 3 * 16 floats
#define HEVCDL_W_C1   0            // 2560 + 3*16
#define HEVCDL_W_C64  2608
#define HEVCDL_W_C2   5216         // 9*32*64 + 3*64
#define HEVCDL_W_C3   23840        // 9*64*128 + 3*128
#define HEVCDL_W_FC1  97952        // 2048*256 + 256
#define HEVCDL_W_FC2  622496       // 256*64 + 64
#define HEVCDL_W_FC3  638944       // 64*16 + 16
#define HEVCDL_W_SCALES 639984      // 8 floats: 1 / (weight scale * HEVCDL_ACT_SCALE) of conv1, conv64, conv2, conv3, fc1 (the scales below), 3 unused
#define HEVCDL_W_TOTAL 639992
// Split-f16 operands (cnn_kernel.hip split_f16) carry 22 significant bits only while the lo half is a NORMAL f16, i.e. for |v| >= 2^-3; below that the pair's error
// is an absolute 2^-25.  Most weights (1e-3 .. 1e-1) and the darker input samples (u8 / 255) lie below 2^-3, so every operand is scaled by an exact power of two
// before it is split: the weights of a layer by 2^k, k the largest exponent that keeps max |w| 2^k below 2^15 (hevcdl_api.hip layer_scale), activations -- the
// input LUT, the maps the BatchNorm epilogues write, conv3's output -- by HEVCDL_ACT_SCALE.  An accumulator then holds (weight scale * HEVCDL_ACT_SCALE) times the
// layer's output; the factor is exact and leaves through the folded BatchNorm map (statistics and alpha unscaled in f64, cnn_kernel.hip bn_fold) or fc1's epilogue.
#define HEVCDL_ACT_SCALE 16.0f

// K order of the 5x5 convolutions (cnn_kernel.hip conv5_mfma, hevcdl_api.hip pack_conv5): 5 k-steps of 16 taps; slot = 16 s + 4 g + j is word j of lane group g in
// k-step s.  Two rules shape it.  (1) One half of a ds_read_b32 serves lane groups 2h and 2h + 1: their words must lie 16 banks apart in the input tile (the 16
// positions of an M-tile, an 8 x 2 block at a row pitch == 8 (mod 32), cover 16 banks whose translate by 16 is the complement) -- partners are (c0, ky, kx) and
// (c1, ky, kx) (channel stride == 16 mod 32), (c2, ky, kx) and (c2, ky + 2, kx), and for what is left a padding slot (weight 0) on the same tap of c1.  (2) In k-steps
// 0..3 the four words of a lane group are kx = 0..3 of ONE row (c, ky): one pointer, two ds_read2_b32 into four consecutive registers = the MFMA operand as it is.
// Step 4 holds the kx = 4 column.  Returns the tap (c * 5 + ky) * 5 + kx, or -(tap + 1) for a padding slot (reads `tap`).
#ifdef __HIPCC__
#define HEVCDL_HD __host__ __device__
#else
#define HEVCDL_HD
#endif
static inline HEVCDL_HD int hevcdl_conv5_slot_tap(int slot)
{
  const int s = slot >> 4, g = (slot >> 2) & 3, j = slot & 3, second = g & 1;
  int c, ky, kx, pad = 0;
  // the eight partner pairs of rows: (c0, ky) | (c1, ky) for ky = 0..4, (c2, 0) | (c2, 2), (c2, 1) | (c2, 3), (c2, 4) | padding on (c1, 4)
  int P;
  if (s < 4) { P = 2 * s + (g >> 1); kx = j; }
  else { kx = 4; P = g < 2 ? j : (j == 0 ? 4 : 4 + j); }        // step 4: lane groups 0 | 1 take pairs 0..3 (words = ky), groups 2 | 3 pairs 4..7
  if (P < 5) { c = second; ky = P; }
  else if (P < 7) { c = 2; ky = (P - 5) + (second ? 2 : 0); }
  else { c = second ? 1 : 2; ky = 4; pad = second; }
  const int tap = (c * 5 + ky) * 5 + kx;
  return pad ? -(tap + 1) : tap;
}

struct hevcdl_cnn_params {
  const uint8_t *input;            // planar 4:2:0 frames, or packed RGB CTUs (input_mode 2)
  const float *weights;            // packed, HEVCDL_W_TOTAL floats
  uint8_t *labels;                 // [ctu][16]
  float *logits;                   // [ctu][4][16] or NULL
  int input_mode, width, height, ctus_x, ctus_per_frame, clamp;
  float *a3;                       // [ctu of the launch][4 quadrants][2048]: flattened conv3 output = input rows of the fully connected head
  int ctu_base;                    // global index of the launch's first CTU
  int n_cus;                       // compute units of the device (first-generation workgroups = 2 per CU)
  int bn_eval;                     // 1: BatchNorm with the checkpoint's running statistics (folded into the packed gamma / beta slots by the host)
};

// fully connected head + labels (fc_kernel.hip), 16 CTUs per workgroup
struct hevcdl_fc_params {
  const float *a3;                 // [n_ctus * 4][2048]
  const float *weights;
  uint8_t *labels;                 // [n_ctus][16] of this launch
  float *logits;                   // [n_ctus][4][16] of this launch, or NULL
  int n_ctus, ctu_base, width, height, ctus_x, ctus_per_frame, clamp;
  const float *logits_in;          // not NULL: skip the three layers and label these logits [n_ctus][4][16] (hevcdl_labels_from_logits)
};

// decision constants (bit patterns computed on the host, see include/hevcdl.h)
struct hevcdl_rd_consts {
  double lambda, sqrt_lambda, chroma_weight, lambda_chroma;
  double err_scale[2][4];
  long long sbh_rd_factor[2];
  int qp, qp_chroma;
  int tools, pad_tools;            // hevcdl_config.tools (HEVCDL_TOOL_*): what the cfg's switches leave on
};

struct hevcdl_rd_params {
  const uint8_t *yuv;              // [frame] planar 4:2:0 originals
  const uint8_t *labels;           // [frame][ctu][16]
  unsigned char *records;          // [frame][ctu] hevcdl_ctu_record
  uint8_t *recon;                  // [frame] planar 4:2:0 reconstruction (also the neighbour source)
  unsigned char *stats;            // [frame] hevcdl_frame_stats or NULL
  unsigned char *scratch;          // [workgroup][wave] workspace
  size_t scratch_per_wave;
  unsigned int *dbgbuf;
  unsigned char *sched;            // hand-over of units between workgroups: [0] finished units, [16 + g] units walked by workgroup g, +8192: one mailbox per workgroup
  int migrate;                     // 1: units travel round the ring of workgroups (uneven dealing)
  int remote;                      // fewer units than workgroups: the workgroups without a unit run second luma passes posted by the others (queue at sched + 4096); 2: the chroma modes too (very few units); 3: passes only while a taker is free (more units than takers)
  const unsigned char *cabac_in;   // [frame] 168-byte coder state to start from, or NULL: slice-start state (only with ctu_begin == 0)
  unsigned char *cabac_out;        // [frame] coder state after the last CTU processed, or NULL
  int ctu_begin, ctu_end;          // CTU address range [begin, end) in coding order
  int width, height, ctus_x, ctus_y, n_frames, debug;
  int tile_cols, tile_rows;        // tiles (1 x 1: none); one wave per (frame, tile)
  int col_bd[21], row_bd[23];      // tile boundaries in CTUs (hevcdl_tile_bounds)
  int tile_begin, tile_count;      // tiles of every frame this launch covers (raster order of tiles); with wpp: the units of a frame (ctus_y rows, or 1)
  // WaveFrontSynchro (hevcdl_config.wavefront): 1 = a unit is one CTU ROW of a frame (tile_count = ctus_y), rows of a frame run two CTUs apart on different waves; 2 = a unit is a
  // whole frame whose rows start from the synchronised contexts (one wave walks them in order: the form that needs no co-residency).  wpp_state: [frame][row] 256 bytes --
  // the contexts behind the row's second CTU (TEncSlice.cpp:925-928) at 0, the number of finished CTUs of the row at 192 (zeroed by the host before the launch)
  int wpp;
  unsigned char *wpp_state;
  // wpp == 1: rows are CLAIMED, not dealt -- a row becomes claimable when the row above has finished the CTU above and to the right of its first one (a ring of ready rows in
  // HBM), a wave without a unit takes a ready row or else the first row of a frame nobody has started; no wave ever holds a slot for a row that cannot start.
  // wpp_masters: waves of a workgroup that claim (the rest help); wpp_queue: ints -- [0] next frame, [32] ring head, [64] ring tail, [96] rows claimed, ring of wpp_ring
  // entries (a power of two >= n_frames: a frame has at most one ready row at a time) from [256]; zeroed by the host
  int wpp_masters, wpp_ring;
  unsigned char *wpp_queue;
  int master_groups;               // workgroups that walk units: in the few-units form the workgroups from this index on take the jobs the others post (without wpp: the number of units)
  hevcdl_rd_consts k;
};

// deblocking passes (deblock_kernel.hip); tc / beta / chroma tc looked up on the host (TComLoopFilter.cpp:59-67, Bs 2, offsets 0)
struct hevcdl_dbk_params {
  const uint8_t *in;               // [frame] planar 4:2:0 reconstruction before the in-loop filters
  uint8_t *out;                    // [frame] deblocked picture (may alias in)
  const unsigned char *records;    // [frame][ctu] hevcdl_ctu_record (depth, trIdx give the TU grid)
  int width, height, ctus_x, ctus_per_frame, n_frames;
  int tc, beta, tc_c;              // already scaled by 1 << (bit depth - 8) (TComLoopFilter.cpp:596, 770)
  int pel_max;                     // (1 << bit depth) - 1; 255: planes of uint8, otherwise planes of uint16
  int lf_across_tiles, tile_cols, tile_rows, col_bd[21], row_bd[23];   // LFCrossTileBoundaryFlag 0: no edge on a tile border is filtered
};

// sample adaptive offset (sao_kernel.hip)
struct hevcdl_sao_params {
  const uint8_t *org, *deblocked;  // [frame] planar 4:2:0
  uint8_t *out;                    // [frame] final reconstruction
  unsigned char *stats;            // [frame][ctu][3][5] {int32 diff[32], count[32]}  (3840 bytes per CTU)
  unsigned char *params;           // [frame][ctu] hevcdl_sao_blk: coded parameters
  unsigned char *recon_params;     // [frame][ctu] hevcdl_sao_blk: merge candidates resolved
  unsigned char *cand;             // [frame][ctu][3][5] candidate offsets of every type {int8 offset[4], int32 aux, int64 distortion} (240 bytes per CTU)
  unsigned char *crec;             // [frame][ctu] the decision chain's compact record (32 bytes), expanded into params / recon_params afterwards
  int width, height, ctus_x, ctus_per_frame, n_frames, qp;
  int tile_cols, tile_rows;        // merge candidates stay inside a tile
  int col_bd[21], row_bd[23];      // tile boundaries in CTUs
  int bit_depth;                   // 8: planes of uint8; 10: planes of uint16 (offset range 31, band = sample >> 5, distortion >> 4)
  int lf_across_tiles;             // 0: a neighbouring CTU of another tile counts as missing (like the picture border)
  double lambda, lambda_chroma;
};

// Tile boundaries in CTUs: bd[0] = 0 < bd[1] < ... < bd[n_tiles] = n_ctus.  Uniform spacing as TComPicSym.cpp xInitTiles; explicit sizes
// name every tile but the last (which takes the rest).  min_size: smallest tile the reference accepts (4 CTU columns, 1 CTU row,
// TComPicSym.cpp:380-392) when there is more than one tile in that direction.  Returns 0 when the layout is valid.
static inline int hevcdl_tile_bounds(int n_ctus, int n_tiles, int uniform, const int32_t *sizes, int min_size, int *bd)
{
  if (n_tiles < 1 || n_tiles > n_ctus) return -1;
  bd[0] = 0;
  for (int t = 0; t < n_tiles; t++) {
    if (uniform) bd[t + 1] = ((t + 1) * n_ctus) / n_tiles;
    else bd[t + 1] = (t == n_tiles - 1) ? n_ctus : bd[t] + sizes[t];
    if (bd[t + 1] - bd[t] < 1 || (n_tiles > 1 && bd[t + 1] - bd[t] < min_size)) return -1;
  }
  return bd[n_tiles] == n_ctus ? 0 : -1;
}

// Boundary policy HEVCDL_BOUNDARY_CLAMP for the 16 labels of the CTU at (x0, y0) (shared by the CNN head kernel and the check of caller labels):
//  1. a cell's label is raised to the smallest depth at which the CU containing the cell lies inside the picture;
//  2. what the reference's walk then READS is made valid, nothing else is touched.  The walk (TEncCu.cpp:496-520) looks at ONE label per CU, the
//     one of its top-left 16x16 cell: label == depth codes the CU, label > depth splits it, label < depth would leave it undecided.  So: a CTU
//     inside the picture whose first label is 0 is one 64x64 CU and the other 15 labels are never read (use_model.py:101-119 does emit such sets:
//     they stay as they are); otherwise every 32x32 quadrant inside the picture is visited -- a first label of 0 becomes 1 -- and where a
//     quadrant is split (first label >= 2, or the picture edge forces it) each of its cells inside the picture is read and raised to 2.
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline void hevcdl_clamp_ctu_labels(uint8_t *lab, int x0, int y0, int width, int height)
{
  const int quads[4][4] = { {0, 1, 4, 5}, {2, 3, 6, 7}, {8, 9, 12, 13}, {10, 11, 14, 15} };
  int md[16], inside[16], forced0 = 0;
  for (int c = 0; c < 16; c++) {
    const int px = x0 + (c & 3) * 16, py = y0 + (c >> 2) * 16;
    md[c] = 0; inside[c] = px < width && py < height;
    if (inside[c]) { while (md[c] < 3) { const int s = 64 >> md[c]; if ((px / s) * s + s <= width && (py / s) * s + s <= height) break; md[c]++; } }
    if (lab[c] < md[c]) lab[c] = (uint8_t)md[c];
    if (md[c] >= 1) forced0 = 1;
  }
  if (!forced0 && lab[0] == 0) return;
  for (int q = 0; q < 4; q++) {
    const int q0 = quads[q][0];
    if (!inside[q0]) continue;
    int forced1 = 0;
    for (int k = 0; k < 4; k++) if (inside[quads[q][k]] && md[quads[q][k]] >= 2) forced1 = 1;
    if (!forced1 && lab[q0] < 1) lab[q0] = 1;
    if (forced1 || lab[q0] >= 2) for (int k = 0; k < 4; k++) if (inside[quads[q][k]] && lab[quads[q][k]] < 2) lab[quads[q][k]] = 2;
  }
}

#ifdef __cplusplus
extern "C" {
#endif
void hevcdl_launch_sao(const struct hevcdl_sao_params *p, void *stream);
void hevcdl_launch_deblock(const struct hevcdl_dbk_params *p, void *stream);
size_t hevcdl_cnn_smem_bytes(void);
size_t hevcdl_fc_smem_bytes(void);
size_t hevcdl_rd_smem_bytes(void);
size_t hevcdl_rd_scratch_bytes(void);      // per wave
int hevcdl_rd_waves_per_group(void);
#ifdef __cplusplus
}
#endif
#endif
