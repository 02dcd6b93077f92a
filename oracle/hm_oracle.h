/* oracle/hm_oracle.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the reference's depth-pruned all-intra CTU decision path
 * (TEncSlice::compressSlice -> TEncCu::compressCtu -> xCompressCU -> xCheckRDCostIntra
 *  -> TEncSearch::estIntraPredLumaQT / estIntraPredChromaQT).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * Pinned against the reference itself: oracle/_ref (built by oracle/build_ref.sh) through
 * oracle/gen_fixtures.py -> tests/golden/rd_*.npz.
 */
#ifndef HM_ORACLE_H
#define HM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as hevcdl_ctu_record (include/hevcdl.h): 15120 bytes. */
typedef struct {
  uint8_t  depth[256], part_size[256], luma_dir[256], chroma_dir[256], tr_idx[256];
  uint8_t  cbf[3][256], tskip[3][256];
  uint32_t bits, dist;
  double   cost;
  int16_t  coeff_y[4096], coeff_cb[1024], coeff_cr[1024];
} hm_ctu_record;

typedef struct {
  uint64_t sse[3];      /* reconstruction SSE per plane (before in-loop filters) */
  uint64_t est_bits;    /* sum over CTUs of the CABAC-estimated bits of the final encode */
  uint32_t ctus;
  uint32_t pad;
} hm_frame_stats;

/* Encode n_frames independent all-intra frames (8-bit 4:2:0 planar, Y then U then V per frame).
 * labels: [n_frames][ctus][16] depth labels (already clamped/valid).  out_recs: [n_frames][ctus].
 * recon: same layout as yuv (may be NULL).  stats: [n_frames] (may be NULL).  Returns 0 on success. */
int hm_oracle_encode_frames(const uint8_t *yuv, int width, int height, int n_frames, int qp,
                            const uint8_t *labels, hm_ctu_record *out_recs, uint8_t *recon,
                            hm_frame_stats *stats);
/* The same with tile_cols x tile_rows uniformly spaced tiles (one slice; TileUniformSpacing 1): each tile is coded from a fresh
 * coder state and sees nothing of the other tiles.  Records stay in raster CTU order. */
int hm_oracle_encode_frames_tiles(const uint8_t *yuv, int width, int height, int n_frames, int qp,
                                  const uint8_t *labels, hm_ctu_record *out_recs, uint8_t *recon,
                                  hm_frame_stats *stats, int tile_cols, int tile_rows);
/* bit_depth 8 (uint8 samples) or 10 (uint16 samples, InputBitDepth = InternalBitDepth = 10, Profile main10). */
int hm_oracle_encode_frames_ex(const void *yuv, int width, int height, int n_frames, int qp,
                               const uint8_t *labels, hm_ctu_record *out_recs, void *recon,
                               hm_frame_stats *stats, int tile_cols, int tile_rows, int bit_depth);
/* explicit tile boundaries in CTUs: col_bd[0] = 0 < ... < col_bd[tile_cols] = CTU columns (TileUniformSpacing 0 with width / height arrays) */
int hm_oracle_encode_frames_tb(const void *yuv, int width, int height, int n_frames, int qp,
                               const uint8_t *labels, hm_ctu_record *out_recs, void *recon,
                               hm_frame_stats *stats, int tile_cols, int tile_rows, const int *col_bd, const int *row_bd, int bit_depth);

/* Deblocking (oracle/hm_deblock.c): filters one planar 4:2:0 frame in place, given the frame's CTU records. */
int hm_oracle_deblock_frame(uint8_t *frame, int width, int height, int qp, const hm_ctu_record *recs);
int hm_oracle_deblock_frame16(uint16_t *frame, int width, int height, int qp, const hm_ctu_record *recs, int bit_depth);   /* uint16 samples, bit_depth 8 or 10 */
/* col_bd / row_bd (tile boundaries in CTUs) non-NULL = LFCrossTileBoundaryFlag 0: no filtering across tile borders */
int hm_oracle_deblock_frame16_tb(uint16_t *frame, int width, int height, int qp, const hm_ctu_record *recs, int bit_depth,
                                 int tile_cols, int tile_rows, const int *col_bd, const int *row_bd);

/* Sample adaptive offset (oracle/hm_sao.c).  mode 0 off / 1 new / 2 merge; type: new -> 0..3 edge offset 0/90/135/45 degrees,
 * 4 band offset; merge -> 0 left, 1 above; aux = band position; offset[class] (edge classes 0..4, bands 0..31). */
typedef struct { int32_t mode, type, aux; int32_t offset[32]; } hm_sao_offset;
typedef struct { hm_sao_offset c[3]; } hm_sao_blk;
/* org, deblocked, out: planar 4:2:0 frames; params: [ctus] coded parameters (as written to the bitstream). */
int hm_oracle_sao_frame(const uint8_t *org, const uint8_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint8_t *out);
/* uint16 sample planes, bit_depth 8 or 10 (offset range 7 / 31, band shift bit_depth - 5, distortion at 8-bit scale) */
int hm_oracle_sao_frame16(const uint16_t *org, const uint16_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint16_t *out, int tile_cols, int tile_rows, int bit_depth);
int hm_oracle_sao_frame16_tb(const uint16_t *org, const uint16_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint16_t *out, int tile_cols, int tile_rows, const int *col_bd, const int *row_bd, int bit_depth);
void hm_oracle_sao_set_lf_across_tiles(int flag);   /* LFCrossTileBoundaryFlag for the next hm_oracle_sao_* calls (default 1) */
int hm_oracle_sao_frame_tiles(const uint8_t *org, const uint8_t *deblocked, int width, int height, int qp, hm_sao_blk *params, uint8_t *out, int tile_cols, int tile_rows);

/* Debug: if non-NULL, every RD cost evaluation appends (bits, dist) to this FILE (text). */
void hm_oracle_set_trace(const char *path);

#ifdef __cplusplus
}
#endif
#endif
