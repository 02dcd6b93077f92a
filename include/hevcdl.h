/* include/hevcdl.h -- C ABI of the MI355X-native all-intra CU-partition hot path.
 *
 * Drop-in boundary: the reference's
 *     Void TEncCu::compressCtu(Int m_iFrame, TComDataCU* pCtu)
 *       HM_dl/source/Lib/TLibEncoder/TEncCu.h:120, TEncCu.cpp:234, single call site TEncSlice.cpp:879
 * together with the label producer it busy-waits for (use_model.py / gen_frames.py, file IPC at
 * TEncCu.cpp:244-253).  The reference walks CTUs one at a time on one CPU thread; CTUs of one slice
 * are strictly serial (reconstructed neighbours + adaptive CABAC state), frames of an all-intra
 * sequence are independent, so the GPU entry points are per BATCH OF FRAMES.  Everything crossing this
 * boundary is plain pointers and sizes; no torch / C++ types.
 *
 * Threading: one hevcdl_ctx per device; a ctx is not thread-safe; different ctxs are independent.
 * Ownership: the caller owns every buffer it passes; the library owns only its internal workspace.
 * Errors: every function returns a status code; nothing aborts or waits forever (the reference hangs at
 * TEncCu.cpp:245 when the label file never appears).
 */
#ifndef HEVCDL_H
#define HEVCDL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  HEVCDL_OK = 0,
  HEVCDL_ERR_INVALID_ARG = 1,
  HEVCDL_ERR_UNSUPPORTED = 2,   /* a cfg key that would change the path and is not implemented */
  HEVCDL_ERR_NO_DEVICE = 3,
  HEVCDL_ERR_HIP = 4,
  HEVCDL_ERR_OOM = 5
} hevcdl_status;

/* tool flags (reference cfg: encoder_intra_main.cfg:37-38,53-54 + TAppEncCfg.cpp defaults :978,950,1007) */
#define HEVCDL_TOOL_RDOQ            (1u << 0)
#define HEVCDL_TOOL_RDOQTS          (1u << 1)
#define HEVCDL_TOOL_TSKIP           (1u << 2)
#define HEVCDL_TOOL_TSKIP_FAST      (1u << 3)
#define HEVCDL_TOOL_SIGN_HIDE       (1u << 4)
#define HEVCDL_TOOL_STRONG_INTRA    (1u << 5)
#define HEVCDL_TOOL_FAST_UDI_MPM    (1u << 6)
#define HEVCDL_TOOLS_REFERENCE      0x7fu
/* Every one of them may be turned off (cfg keys RDOQ, RDOQTS, TransformSkip, TransformSkipFast, SignHideFlag, StrongIntraSmoothing, FastUDIUseMPMEnabled; each pinned by a run of
 * the reference encoder with the switch on its command line: tests/golden/rd_k*.npz).  Without RDOQ the quantiser is the dead-zone rounding of TComTrQuant::xQuant with
 * signBitHidingHDQ behind it (RDOQTS: the same choice for transform-skipped blocks); without TransformSkipFast every 4x4 block is tried both ways, not only those of NxN CUs
 * (TEncSearch.cpp:1502-1505, 1965-1990).  Bits outside HEVCDL_TOOLS_REFERENCE are rejected. */
#define HEVCDL_TOOLS_SWITCHABLE     HEVCDL_TOOLS_REFERENCE
#define HEVCDL_TOOLS_SUPPORTED(t)   ((((t) | HEVCDL_TOOLS_SWITCHABLE) == HEVCDL_TOOLS_REFERENCE) && ((t) & ~HEVCDL_TOOLS_REFERENCE) == 0)

#define HEVCDL_CNN_INPUT_RGB601 0   /* BT.601 limited-range YUV -> RGB, nearest chroma (defined by this project) */
#define HEVCDL_CNN_INPUT_LUMA   1   /* R = G = B = Y */
#define HEVCDL_BN_REFERENCE     0   /* training-mode BatchNorm, as use_model.py:61-63 runs it (the reference never calls model.eval()) */
#define HEVCDL_BN_EVAL          1   /* BatchNorm with the checkpoint's running statistics (what model.eval() would give; NOT the reference pipeline's labels) */
/* HEVCDL_BOUNDARY_CLAMP: every label is raised to the smallest depth at which the CU containing its 16x16 cell lies inside the picture;
 * then exactly what the reference's walk READS is made valid and nothing else is touched.  The walk (TEncCu.cpp:496-520) looks at one label per CU, the one of its
 * top-left cell: a CTU inside the picture whose first label is 0 is ONE 64x64 CU whatever the other 15 labels say (use_model.py:101-119 does write such files, and HM
 * codes them that way -- so does this library); otherwise a visited 32x32 quadrant whose first label is 0 gets 1 and the cells of a split quadrant at least 2. */
#define HEVCDL_BOUNDARY_CLAMP   0

#define HEVCDL_WEIGHT_FLOATS 637712 /* state_dict order of rec/hevc_encoder_model.pt, fp32 tensors only */

typedef struct hevcdl_config {
  uint32_t struct_size;          /* sizeof(hevcdl_config) */
  int32_t  width, height;        /* luma samples, multiples of 8 (min CU) */
  int32_t  bit_depth;            /* 8, or 10: InputBitDepth = InternalBitDepth = 10 (Profile main10); samples are then uint16,
                                    little endian, in every yuv / recon / picture buffer (decision path and in-loop filters); the CNN
                                    stage sees sample >> 2 */
  int32_t  chroma_format;        /* 420 */
  int32_t  qp;                   /* slice QP, 0..51 */
  int32_t  ctu_size;             /* 64   (MaxCUWidth/Height)          */
  int32_t  max_partition_depth;  /* 4    (MaxPartitionDepth)          */
  int32_t  tu_log2_min, tu_log2_max, tu_max_depth_intra;   /* 2, 5, 3 */
  uint32_t tools;                /* HEVCDL_TOOL_* : any subset of HEVCDL_TOOLS_REFERENCE (the reference cfg's value) */
  int32_t  bn_mode, boundary_policy, cnn_input;
  int32_t  device;               /* HIP device ordinal */
  int32_t  max_frames;           /* frames per call the workspace is sized for */
  /* decision constants computed ON THE HOST in IEEE double exactly as the reference does
   * (TEncSlice.cpp:112-140,433-527; TComTrQuant.cpp:3096-3126,2532-2535) and passed as bits: */
  double   lambda, sqrt_lambda, chroma_weight, lambda_chroma;
  double   err_scale[2][4];      /* [luma/chroma][log2(TU)-2] */
  int64_t  sbh_rd_factor[2];     /* [luma/chroma] */
  int32_t  qp_chroma;
  /* uniformly spaced tiles (cfg keys TileUniformSpacing 1, NumTileColumnsMinus1 + 1, NumTileRowsMinus1 + 1; TAppEncCfg.cpp:1024-1028);
   * 1 x 1 = no tiles.  Every tile is at least 4 CTUs wide (the reference's own limit, TComPicSym.cpp:388).  Tiles are independent
   * units of the decision path: one wavefront per (frame, tile). */
  int32_t  tile_columns, tile_rows;
  /* Execution options of the decision kernel (0 = default; formerly a reserved field that had to be 0).
   * HEVCDL_EXEC_NO_UNIT_HANDOVER: never let a frame travel between workgroups.  With more frames than compute units and an uneven split (600 frames on
   * 256 CUs) the kernel hands frames round a ring of workgroups so that every CU is crowded for the same share of the time; workgroups then wait for
   * each other, which needs ALL of them resident at once.  The library therefore launches that form cooperatively (the runtime refuses the launch when
   * it cannot co-schedule the grid -- another context holding CUs or LDS, a CU mask -- and the library falls back to the independent form by itself);
   * set this bit to use the independent form always, e.g. when several contexts / processes share the device by design.  The same holds for launches of at most
   * two thirds as many frames (x tiles) as CUs, which run on ALL CUs so that the empty ones take second luma passes (and, below a twenty-second, chroma modes) from the
   * others: cooperative as well, same fall-back, switched off by this bit (the context then also allocates workspace for its own frames only). */
  int32_t  exec_flags;
#define HEVCDL_EXEC_NO_UNIT_HANDOVER 1
  /* The 8-bit decision kernel exists in two builds: 8 wavefronts per workgroup with the look-ahead of the few-units form, and 10 (csrc/rd_kernel_wide.hip) for
   * launches of three or more units per workgroup; the library picks by the shape of the launch.  HEVCDL_EXEC_RD_WIDE forces the second for every launch of the context (the few-units form is then
   * never used), HEVCDL_EXEC_RD_NARROW the first.  Results do not depend on the choice.  Contexts whose `tools` are not the reference cfg's and 10-bit contexts have
   * one build each (eight waves): both bits are without effect there. */
#define HEVCDL_EXEC_RD_WIDE 2
#define HEVCDL_EXEC_RD_NARROW 4
  /* TileUniformSpacing 0: explicit sizes in CTUs of every tile column / row but the last (TileColumnWidthArray, TileRowHeightArray,
   * TAppEncCfg.cpp:1026-1028); ignored when tile_uniform_spacing != 0 (the default) */
  int32_t  tile_uniform_spacing;
  int32_t  tile_column_width[19], tile_row_height[21];
  int32_t  lf_across_tiles;      /* LFCrossTileBoundaryFlag (default 1): 0 = deblocking and SAO stop at tile borders */
  /* LoopFilterBetaOffset_div2 / LoopFilterTcOffset_div2 (-6 .. 6, default 0; with LoopFilterOffsetInPPS 1, the cfg's value): added twice to the QP the deblocking filter's
   * beta / tc tables are read with (TComLoopFilter.cpp:623-624, 804) */
  int32_t  lf_beta_offset_div2, lf_tc_offset_div2;
  /* WaveFrontSynchro (cfg key of that name, TAppEncCfg.cpp:975; default 0).  1: entropy_coding_sync_enabled_flag -- the coder is re-initialised at the first CTU of every
   * CTU row and takes over the contexts behind the second CTU of the row above (TEncSlice.cpp:783-830, 925-928).  The decisions then differ from the default cfg's (every
   * RD cost is priced with other context states), exactly as the reference's do with the key set; the CTU chain of a frame (2040 steps at 2160p) becomes ctus_y chains two
   * CTUs apart (126 steps), which the decision kernel walks on different waves.  Not together with tiles (the reference refuses the pair in the main profiles).  The per-CTU
   * session (hevcdl_compress_ctu) works on such a context too: the contexts behind the second CTU of every row are kept per session frame, a row's first CTU starts from them
   * whatever state the caller hands in (that IS the reference's behaviour: resetEntropy + loadContexts in front of compressCtu, TEncSlice.cpp:808-823). */
  int32_t  wavefront;
} hevcdl_config;

/* One CTU of decisions: what compressCtu leaves in the picture's CTU record (TEncCu.cpp:1091 copyToPic).
 * 256 entries = 4x4 luma partitions in z-scan order (TComRom.cpp:284-352).  Coefficients use the
 * reference's TComDataCU::m_pcTrCoeff layout: the TU whose first partition is p starts at 16*p (luma)
 * or 4*p (chroma) and is stored row-major with its own width as stride. */
typedef struct hevcdl_ctu_record {
  uint8_t  depth[256], part_size[256], luma_dir[256], chroma_dir[256], tr_idx[256];
  uint8_t  cbf[3][256], tskip[3][256];
  uint32_t bits, dist;           /* pCtu->getTotalBits() / getTotalDistortion() (TEncSlice.cpp:959-961) */
  double   cost;                 /* pCtu->getTotalCost() */
  int16_t  coeff_y[4096], coeff_cb[1024], coeff_cr[1024];
} hevcdl_ctu_record;             /* 15120 bytes */

typedef struct hevcdl_frame_stats {
  uint64_t sse[3];               /* reconstruction SSE per plane before in-loop filters */
  uint64_t est_bits;             /* CABAC-estimated bits of the state-advancing encode (TEncSlice.cpp:886-893) */
  uint32_t ctus, pad;
} hevcdl_frame_stats;

typedef struct hevcdl_profile {
  double   cnn_ms, rd_ms;        /* accumulated kernel time measured with HIP events on the launch stream */
  uint32_t cnn_launches, rd_launches;
  double   cnn_conv_ms;          /* of cnn_ms: the convolution kernel alone (the rest is the fully connected head), summed over the same launches */
} hevcdl_profile;

typedef struct hevcdl_ctx hevcdl_ctx;

/* Fill cfg from (width, height, qp) with the reference configuration (encoder_intra_main.cfg) and the
 * host-computed lambda family.  Replaces TEncSlice::setUpLambda / calculateLambda for this path. */
hevcdl_status hevcdl_config_default(hevcdl_config *cfg, int width, int height, int qp);
/* The same for a given sample bit depth (8 or 10): quantiser QP offset 6 per extra bit, distortion kept at 8-bit scale
 * (the reference's FULL_NBIT 0 build, TypeDef.h:162-172), lambda unchanged (TEncSlice.cpp:433-527). */
hevcdl_status hevcdl_config_default_bd(hevcdl_config *cfg, int width, int height, int qp, int bit_depth);

/* weights: flat fp32 blob, HEVCDL_WEIGHT_FLOATS values (replaces torch.load at use_model.py:62). */
hevcdl_status hevcdl_create(const hevcdl_config *cfg, const float *weights, size_t n_floats, hevcdl_ctx **out);
void          hevcdl_destroy(hevcdl_ctx *ctx);
const char   *hevcdl_last_error(const hevcdl_ctx *ctx);

/* ---- host-buffer entry points (copy in, run, copy out) ------------------------------------- */
/* Replaces gen_frames.py + use_model.py: planar 4:2:0 frames (uint8, or uint16 at bit_depth 10) -> labels[n_frames][ctus][16]
 * (clamped to the picture); logits_opt[n_frames][ctus][4][16] may be NULL. */
hevcdl_status hevcdl_predict_depth(hevcdl_ctx *ctx, const uint8_t *yuv, int n_frames, uint8_t *labels, float *logits_opt);
/* Fixture entry: n_ctus RGB CTUs [n][64][64][3] (the tensor the reference feeds ConvNet2) -> raw labels
 * (use_model.py:101-119, no clamping) and logits[n][4][16]. */
hevcdl_status hevcdl_predict_depth_rgb(hevcdl_ctx *ctx, const uint8_t *ctu_rgb, int n_ctus, uint8_t *labels, float *logits_opt);
/* The label stage alone: logits[n_ctus][4][16] (the four forwards of a CTU, use_model.py:100) -> labels[n_ctus][16] by the 4x argmax and the
 * fix-ups of use_model.py:101-119, run by the same device code that follows the fully connected head.  clamp != 0 also applies the boundary
 * policy, taking CTU i as CTU (i mod ctus-per-frame) of the context's picture. */
hevcdl_status hevcdl_labels_from_logits(hevcdl_ctx *ctx, const float *logits, int n_ctus, int clamp, uint8_t *labels);
/* Replaces the compressCtu loop of TEncSlice::compressSlice for n_frames independent frames.
 * labels_opt == NULL -> labels come from the on-device CNN.  recon_opt / stats_opt may be NULL. */
hevcdl_status hevcdl_compress_frames(hevcdl_ctx *ctx, const uint8_t *yuv, int n_frames, const uint8_t *labels_opt,
                                     hevcdl_ctu_record *records, uint8_t *recon_opt, hevcdl_frame_stats *stats_opt);

/* The same two entry points for pictures that are NOT packed: three planes with their own row pitch, as an encoder's picture buffers hold them
 * (HM: TComPicYuv = one Pel (int16) plane per component with a margin of 80 samples round the picture, stride = width + 160,
 * TComPicYuv.cpp:81-104; getAddr(compID) / getStride(compID) give exactly plane[] / row_stride[]).  Strides are in BYTES.  sample_bytes: 1
 * (uint8 samples, bit_depth 8), or 2 (16-bit containers: the samples of a bit_depth 10 context, or 8-bit samples in int16 as HM keeps them --
 * narrowed on the device).  frame_stride[c]: distance from plane c of frame i to plane c of frame i + 1 (ignored for n_frames 1). */
typedef struct hevcdl_planes {
  const void *plane[3];          /* first sample of Y, Cb, Cr of frame 0 */
  size_t      row_stride[3];     /* bytes from one row to the next, >= the plane's width * sample_bytes */
  size_t      frame_stride[3];   /* bytes from frame i to frame i + 1, per plane */
  int32_t     sample_bytes;      /* 1 or 2 */
} hevcdl_planes;
hevcdl_status hevcdl_predict_depth_planes(hevcdl_ctx *ctx, const hevcdl_planes *src, int n_frames, uint8_t *labels, float *logits_opt);
hevcdl_status hevcdl_compress_frames_planes(hevcdl_ctx *ctx, const hevcdl_planes *src, int n_frames, const uint8_t *labels_opt,
                                            hevcdl_ctu_record *records, uint8_t *recon_opt, hevcdl_frame_stats *stats_opt);

/* ---- deblocking filter (first in-loop filter of the reference) -------------------------------------
 * Replaces TComLoopFilter::loopFilterPic(TComPic*) (TLibCommon/TComLoopFilter.cpp:130, called at TEncGOP.cpp:1742) for
 * the pictures this library decides: recon = what hevcdl_compress_frames returned, records = its CTU records (the
 * TU grid), out = the picture the reference hands to SAO (may alias recon in the device variant). */
hevcdl_status hevcdl_deblock_frames(hevcdl_ctx *ctx, const uint8_t *recon, int n_frames, const hevcdl_ctu_record *records, uint8_t *out);
hevcdl_status hevcdl_deblock_frames_dev(hevcdl_ctx *ctx, const void *d_recon, int n_frames, const void *d_records, void *d_out, void *stream);

/* ---- sample adaptive offset (second in-loop filter) ---------------------------------------------------
 * Replaces TEncSampleAdaptiveOffset::SAOProcess (TLibEncoder/TEncSampleAdaptiveOffset.cpp:244, called at TEncGOP.cpp:1797):
 * statistics, per-CTU parameter decision (new / merge-left / merge-up / off, edge or band offsets) and the offset
 * reconstruction.  org = the original frames, deblocked = output of hevcdl_deblock_frames; params receives the coded
 * parameters of every CTU (what TEncSbac::codeSAOBlkParam writes), out the final reconstruction. */
typedef struct hevcdl_sao_offset {
  int32_t mode;        /* 0 off, 1 new, 2 merge (SAOMode, TypeDef.h:539-545) */
  int32_t type;        /* new: 0..3 edge offset 0/90/135/45 degrees, 4 band offset; merge: 0 left, 1 above */
  int32_t aux;         /* band position */
  int32_t offset[32];  /* per class: edge classes 0..4 (2 = plain, always 0) or bands 0..31 */
} hevcdl_sao_offset;
typedef struct hevcdl_sao_blk { hevcdl_sao_offset c[3]; } hevcdl_sao_blk;      /* Y, Cb, Cr */
hevcdl_status hevcdl_sao_frames(hevcdl_ctx *ctx, const uint8_t *org, const uint8_t *deblocked, int n_frames, hevcdl_sao_blk *params, uint8_t *out);
hevcdl_status hevcdl_sao_frames_dev(hevcdl_ctx *ctx, const void *d_org, const void *d_deblocked, int n_frames, void *d_params, void *d_out, void *stream);

/* ---- the whole picture pipeline in one call (host buffers): CNN labels (labels_opt == NULL) -> decisions -> deblocking (deblock != 0) -> SAO
 * (sao_opt != NULL; with deblock == 0 -- LoopFilterDisable 1 -- it works on the unfiltered reconstruction, as the reference does) with the pictures staying in HBM between the stages.  records + picture_out (the output picture: after the
 * enabled filters; uint16 samples at 10 bits) + sao_opt feed hevcdl_write_access_unit / hevcdl_write_picture_hash_sei.  stats_opt: SSE before
 * the in-loop filters and estimated bits, as hevcdl_compress_frames.  Replaces TEncGOP::compressGOP's per-picture stages (TEncGOP.cpp:1560-1800). */
hevcdl_status hevcdl_encode_pictures(hevcdl_ctx *ctx, const void *yuv, int n_frames, const uint8_t *labels_opt, int deblock, hevcdl_ctu_record *records,
                                     void *picture_out, hevcdl_sao_blk *sao_opt, hevcdl_frame_stats *stats_opt);
/* The same pipeline with the results handed over CHUNK BY CHUNK instead of into caller buffers for the whole batch: the device works on all n_frames at once (a
 * frame is a serial chain of CTUs, so a call costs about the same from 1 to ~250 pictures: large batches are what the GPU wants), then `fn` is called on the
 * calling thread for pictures [first, first + count) -- records, output pictures, SAO parameters (NULL when want_sao == 0) and statistics of the chunk, in
 * page-locked memory owned by the library and valid only during the call -- while the next chunk is copied from HBM behind it.  The host work per picture
 * (hevcdl_write_access_unit, hashes, PSNR: TEncGOP.cpp:1895-1935, 2420-2447) thus overlaps the copies, and the host never holds more than two chunks
 * (a 2160p picture is 43 MB of results).  chunk_frames <= 0: a default (64).  A non-zero return of fn stops the hand-over: HEVCDL_ERR_INVALID_ARG. */
typedef int (*hevcdl_chunk_fn)(void *user, int first, int count, const hevcdl_ctu_record *records, const void *pictures, const hevcdl_sao_blk *sao_opt,
                               const hevcdl_frame_stats *stats);
hevcdl_status hevcdl_encode_pictures_chunked(hevcdl_ctx *ctx, const void *yuv, int n_frames, const uint8_t *labels_opt, int deblock, int want_sao,
                                             int chunk_frames, hevcdl_chunk_fn fn, void *user);

/* ---- bitstream writer (host side; no GPU needed) ---------------------------------------------------
 * One access unit per picture exactly as the reference emits it for its all-intra configuration: VPS, SPS, PPS
 * (ReWriteParamSetsFlag 1), then one slice NAL (IDR_W_RADL for POC 0, CRA afterwards), Annex B start codes.
 * Replaces TEncGOP::compressGOP's parameter-set / slice writing (TEncGOP.cpp:1751-1756, 1895-1935), TEncCavlc::codeVPS /
 * codeSPS / codePPS / codeSliceHeader (TEncCavlc.cpp:677, 500, 189, 755), TEncSlice::encodeSlice (TEncSlice.cpp:985) and
 * the arithmetic coder TEncBinCABAC (TEncBinCoderCABAC.cpp:187-446) for the CTU records hevcdl_compress_frames
 * returns.  sao == NULL (cfg.sao_enabled 0): sample_adaptive_offset_enabled_flag 0; otherwise the [ctus] parameters
 * hevcdl_sao_frames returned are coded in front of every CTU.  loop_filter_disable != 0 -> HEVCDL_ERR_UNSUPPORTED. */
typedef struct hevcdl_stream_config {
  uint32_t struct_size;          /* sizeof(hevcdl_stream_config) */
  int32_t  width, height, qp;    /* as in hevcdl_config */
  int32_t  level_idc;            /* general_level_idc = 30 x Level (cfg key Level, TAppEncCfg.cpp:850): 6.2 -> 186, 3.1 -> 93 */
  int32_t  sao_enabled;          /* 1 iff SAO parameters are passed to hevcdl_write_access_unit */
  int32_t  loop_filter_disable;  /* LoopFilterDisable: 1 = pps_deblocking_filter_disabled_flag (the pictures are then NOT to be passed through hevcdl_deblock_frames) */
  int32_t  tile_columns, tile_rows; /* as in hevcdl_config: PPS tile syntax (loop_filter_across_tiles_enabled_flag 1), CTUs in tile scan,
                                       one sub-stream per tile with entry points in the slice header */
  int32_t  bit_depth;            /* 8 (Profile main) or 10 (Profile main10): profile_tier_level, SPS bit depths, SAO offset range */
  int32_t  tile_uniform_spacing; /* as in the hevcdl_config field, default 1; 0: column_width_minus1 / row_height_minus1 are written */
  int32_t  tile_column_width[19], tile_row_height[21];
  int32_t  lf_across_tiles;      /* loop_filter_across_tiles_enabled_flag of the PPS (default 1) */
  uint32_t tools;                /* as in hevcdl_config: transform_skip_enabled_flag / sign_data_hiding_enabled_flag of the PPS, strong_intra_smoothing_enabled_flag of the SPS,
                                    and the residual syntax that goes with the first two */
  int32_t  lf_beta_offset_div2, lf_tc_offset_div2;   /* as in hevcdl_config: pps_beta_offset_div2 / pps_tc_offset_div2 (deblocking_filter_control_present_flag is set when
                                    either is non-zero or the filter is disabled: TEncTop.cpp:1007-1035) */
  int32_t  rewrite_param_sets;   /* ReWriteParamSetsFlag (default 1: VPS / SPS / PPS in front of every picture, all of them IRAPs here); 0: in front of the first picture only
                                    (TEncGOP.cpp:1751) */
  int32_t  wavefront;            /* WaveFrontSynchro (default 0): 1 = entropy_coding_sync_enabled_flag, one sub-stream per CTU row with entry points in the slice header, every row
                                    starting from the contexts behind the second CTU of the row above (TEncSlice.cpp:1047-1145); the records must come from a context with
                                    hevcdl_config.wavefront 1.  Not together with tiles (the reference refuses the pair in the main profiles) */
                                 /* (LFCrossSliceBoundaryFlag has no field: without slices -- SliceMode 0, the only mode of this path -- the reference sets it to 1 whatever the cfg
                                    says, TAppEncTop.cpp:278-281; tests/golden/stream_c192_q32.npz pins that) */
} hevcdl_stream_config;
hevcdl_status hevcdl_stream_config_default(hevcdl_stream_config *cfg, int width, int height, int qp);
size_t        hevcdl_access_unit_bound(int width, int height);
/* records: the [ctus] records of picture `poc` (= frame index; POC lsb is 8 bits).  out_len is set even when the
 * buffer is too small (HEVCDL_ERR_INVALID_ARG), so the call can be repeated. */
hevcdl_status hevcdl_write_access_unit(const hevcdl_stream_config *cfg, int poc, const hevcdl_ctu_record *records, const hevcdl_sao_blk *sao,
                                       uint8_t *out, size_t capacity, size_t *out_len);

/* Decoded picture hash (cfg key SEIDecodedPictureHash 1 = MD5): the suffix SEI NAL the reference appends to the access unit
 * (TEncGOP.cpp:1938-1960), computed from `picture` = the final reconstruction (planar 4:2:0; uint16 samples at 10 bits).  At most
 * 128 bytes.  A decoder uses it to verify its output bit for bit. */
hevcdl_status hevcdl_write_picture_hash_sei(const hevcdl_stream_config *cfg, const void *picture, uint8_t *out, size_t capacity, size_t *out_len);
/* The three 16-byte MD5 digests (Y, Cb, Cr) themselves, as the reference prints them behind a picture's line (TEncGOP.cpp:2529-2540). */
hevcdl_status hevcdl_picture_md5(const hevcdl_stream_config *cfg, const void *picture, uint8_t digest[48]);
/* The same SEI NAL from digests already computed by hevcdl_picture_md5 (an application that also prints them hashes the picture once). */
hevcdl_status hevcdl_write_digest_sei(const uint8_t digest[48], uint8_t *out, size_t capacity, size_t *out_len);
/* The other two hashes of the key (SEIDecodedPictureHash 2 = CRC, 3 = checksum; 1 = MD5 as above): hevcdl_picture_hash leaves the plane digests of `method` side by side in
 * `digest` (16 / 2 / 4 bytes a plane: *plane_bytes; TComPicYuvMD5.cpp:88-180), hevcdl_write_hash_sei writes the SEI (hash_type = method - 1, SEIwrite.cpp). */
hevcdl_status hevcdl_picture_hash(const hevcdl_stream_config *cfg, const void *picture, int method, uint8_t digest[48], int *plane_bytes);
hevcdl_status hevcdl_write_hash_sei(int method, const uint8_t *digest, uint8_t *out, size_t capacity, size_t *out_len);

/* ---- per-CTU session: the semantic drop-in for the reference's call pair ------------------------
 *   TEncCu::compressCtu(Int m_iFrame, TComDataCU* pCtu)   TEncCu.h:120, called at TEncSlice.cpp:879
 *   TEncCu::encodeCtu(TComDataCU* pCtu)                   TEncCu.h:123, called at TEncSlice.cpp:893
 * One call = both (decide the CTU, then advance the true CABAC state over it).  For CPU-side diff testing against
 * the reference's loop; the batched entry points above are what feeds the GPU.
 *   hevcdl_begin_frames   uploads the frames, takes / predicts their labels (labels_out_opt may be NULL) and clears
 *                         reconstruction and records: the replacement of the label-file poll at TEncCu.cpp:244-253.
 *   hevcdl_compress_ctu   CTUs of a frame must arrive in coding order (neighbours' reconstruction and records live
 *                         in the context).  state_in_opt == NULL: continue from the state the previous call left
 *                         (CTU 0: slice-start contexts of the QP, ContextModel.cpp:56-66); otherwise the caller's
 *                         TEncSbac state (TEncSlice.cpp:826-832).  state_out_opt receives the state after encodeCtu.
 *                         Never hangs and never aborts: ordering / range errors come back as HEVCDL_ERR_INVALID_ARG. */
typedef struct hevcdl_cabac_state {
  uint8_t  ctx[160];    /* (state << 1) | mps of the 159 I-slice contexts in TEncSbac.cpp:62-92 order, 1 pad byte */
  uint64_t frac;        /* TEncBinCABACCounter fractional bit accumulator (15 fractional bits) */
} hevcdl_cabac_state;
hevcdl_status hevcdl_begin_frames(hevcdl_ctx *ctx, const uint8_t *yuv, int n_frames, const uint8_t *labels_opt, uint8_t *labels_out_opt);
hevcdl_status hevcdl_compress_ctu(hevcdl_ctx *ctx, int frame, int ctu_addr, const hevcdl_cabac_state *state_in_opt,
                                  hevcdl_ctu_record *record, hevcdl_cabac_state *state_out_opt);
hevcdl_status hevcdl_get_recon(hevcdl_ctx *ctx, int frame, uint8_t *recon);

/* Page-locked host memory for the yuv / records / picture buffers handed to the host-pointer entry points (optional; ordinary memory works,
 * page-locked memory is copied at the DMA rate).  NULL when no GPU runtime is available. */
/* free / total bytes of a device's memory (hipMemGetInfo): what a front end sizes cfg.max_frames by -- a picture of a call holds about 3 frame buffers, its CTU
 * records (15 120 B per CTU), labels / logits and SAO parameters in HBM */
hevcdl_status hevcdl_device_memory(int device, size_t *free_bytes, size_t *total_bytes);
void *hevcdl_host_alloc(size_t bytes);
void hevcdl_host_free(void *p);

/* ---- device-buffer entry points (inputs/outputs already resident in HBM, asynchronous on `stream`) ---- */
/* All pointers are device pointers; stream is a hipStream_t (NULL = default stream). */
hevcdl_status hevcdl_predict_depth_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, void *d_labels, void *d_logits_opt, void *stream);
/* Labels that do NOT come from hevcdl_predict_depth* (e.g. the unclamped label files of the reference's use_model.py, TEncCu.cpp:244-287):
 * boundary policy HEVCDL_BOUNDARY_CLAMP (see there: label sets that are valid for the reference's walk pass unchanged), in place; a depth above 3 is
 * HEVCDL_ERR_INVALID_ARG.  The host-pointer
 * entry points (labels_opt of hevcdl_compress_frames / hevcdl_encode_pictures / hevcdl_begin_frames) do this themselves; the device
 * entry points below take d_labels as they are and require them to satisfy the policy.  Synchronises `stream`.
 * One context = one launch in flight: the context owns the kernels' workspace, so calls on different streams of the same context must
 * not overlap (use one context per concurrent stream). */
hevcdl_status hevcdl_clamp_labels_dev(hevcdl_ctx *ctx, void *d_labels, int n_frames, void *stream);
hevcdl_status hevcdl_compress_frames_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, const void *d_labels,
                                         void *d_records, void *d_recon, void *d_stats, void *stream);
/* The same for tiles [tile_begin, tile_begin + tile_count) of every frame only (raster order of the cfg's tile grid): the unit of
 * work when the tiles of a picture are spread over several GPUs.  Buffers are whole-frame buffers; only the records, reconstruction
 * samples and statistics (partial sums) of the named tiles are written.  Tiles never read each other's results, so the launches of
 * different ranks need no ordering; the in-loop filters need the assembled picture (hevc-deep-learning-pipeline_amd/sharding.py). */
hevcdl_status hevcdl_compress_tiles_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, const void *d_labels,
                                        void *d_records, void *d_recon, void *d_stats, int tile_begin, int tile_count, void *stream);
/* CNN + RD search back to back: the whole hot path for one batch of frames. */
hevcdl_status hevcdl_encode_frames_dev(hevcdl_ctx *ctx, const void *d_yuv, int n_frames, void *d_labels,
                                       void *d_records, void *d_recon, void *d_stats, void *stream);

/* Which build of the decision kernel and which launch form the context's last decision launch took, as text:
 * "hevcdl_rd_frame_kernel[_wide|_tools|_bd10] form=independent|unit-handover|few-units(...) workgroups=N waves=N units=N" ("" before the first launch).  The selection
 * rule lives in one place (launch_rd); bench.py reports this instead of restating the rule. */
const char   *hevcdl_last_rd_launch(const hevcdl_ctx *ctx);
/* The decision kernel's workspace is allocated by the first launch that needs it (megabytes for a frame in the independent form, 3.4 - 4.3 GB for a launch on every
 * CU) and a launch that cannot get it returns HEVCDL_ERR_OOM; the first launch of a larger shape also synchronises the device while the workspace grows.  This call
 * allocates the largest workspace any launch of the context (1 .. max_frames frames) can ask for, so that a lack of memory shows up here -- the CLI calls it right
 * behind hevcdl_create and retries with a smaller batch -- and no later launch allocates or synchronises. */
hevcdl_status hevcdl_reserve_workspace(hevcdl_ctx *ctx);
/* kernel timing with HIP events recorded on the launch stream */
hevcdl_status hevcdl_profile_enable(hevcdl_ctx *ctx, int enable);
hevcdl_status hevcdl_profile_get(hevcdl_ctx *ctx, hevcdl_profile *out);   /* synchronises, returns and resets */

/* sizes */
int      hevcdl_ctus_per_frame(int width, int height);
size_t   hevcdl_frame_bytes(int width, int height);                       /* 8-bit samples */
size_t   hevcdl_frame_bytes_bd(int width, int height, int bit_depth);    /* 2 bytes per sample above 8 bits */

#ifdef __cplusplus
}
#endif
#endif
