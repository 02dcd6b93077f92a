// rd_kernel.hip -- depth-pruned all-intra CTU decision kernel for gfx950 (MI355X).
//
// Replaces, behind include/hevcdl.h, the reference's per-CTU CPU loop
//   TEncSlice::compressSlice   HM_dl/source/Lib/TLibEncoder/TEncSlice.cpp:698-983
//   TEncCu::compressCtu        TEncCu.cpp:234-287  -> xCompressCU :470-1104 -> xCheckRDCostIntra :1600-1665
//   TEncSearch::estIntraPredLumaQT / estIntraPredChromaQT   TEncSearch.cpp:2203-2737 and everything below them
// (reference-sample gather/filter TComPattern.cpp:119-570, prediction TComPrediction.cpp:183-817, SATD/SSE
//  TComRdCost.cpp, DCT/DST/RDOQ/dequant TComTrQuant.cpp, CABAC-as-rate-estimator TEncSbac.cpp / ContextModel.cpp).
//
// Execution model: the CTUs of one slice are a strict serial chain (reconstructed neighbours + the adaptive CABAC
// state advanced by the final encode of every previous CTU), all-intra frames (and tiles) are independent units.
// A workgroup is NW wavefronts on one CU, each with a private LDS block (RdSmem) and private global scratch.  The
// units of a launch are dealt round-robin to the workgroups; inside a workgroup the first waves are MASTERS (one unit
// each, walking its CTUs in coding order), the others HELPERS.  Where the reference's search evaluates alternatives
// that all start from the same saved state -- the luma candidates of the first RD pass (TEncSearch.cpp:2378-2400,
// each reloads CI_CURR_BEST), the five chroma modes (:2640-2660) -- the master opens a REGION in shared LDS: the
// alternatives become tasks that any wave of the workgroup (the master included) claims with an LDS atomic, runs
// on a private copy of the master's state, and answers with (cost, distortion) + its arrays / levels /
// reconstruction in a result slot; the master then picks the winner exactly as the serial loop would (smallest cost,
// first in list order on ties: the loop's strict <).  With 2048 units in flight every wave is a master and the
// kernel is the one-wave-per-unit design; with fewer units the spare waves shorten each unit's critical path.
// Inside a wave
//   * pixel work is lane-parallel: reference gather, the 35-mode rough mode decision (one lane per
//     (mode, 8x8 block) task: predict + Hadamard), prediction, residual, DCT/DST, dequant, reconstruction, SSE;
//   * RDOQ decides the levels of up to four coefficient groups side by side (rdoq_wave), only its ordered fp64 sums are serial; CABAC bin counting runs
//     wave-uniform with the contexts in registers (code_coeff_wave).
// All decision arithmetic is the reference's: int32 transforms, fp64 costs without contraction (-ffp-contract=off),
// lambda family computed on the host and passed as bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hevcdl.h"
#include "hevcdl_dev.h"

// Sample bit depth is a compile-time property of the kernel: this file is compiled as is for 8-bit samples and once more through
// rd_kernel_bd10.hip (HEVCDL_BD 10, uint16 samples; InternalBitDepth 10 with the reference's FULL_NBIT 0 distortion scaling,
// TypeDef.h:162-172).  Everything that depends on it is derived from BD below.
#ifndef HEVCDL_BD
#define HEVCDL_BD 8
#endif
#if HEVCDL_BD == 8
typedef uint8_t pel_t;
#ifdef HEVCDL_RD_WIDE
#define RD_SYM(name) name##_wide      // rd_kernel_wide.hip: more wavefronts per workgroup (launches with a unit for most CUs)
#elif defined(HEVCDL_RD_TOOLS)
#define RD_SYM(name) name##_tools     // rd_kernel_tools.hip: the tool switches of the cfg read at run time (contexts whose hevcdl_config.tools is not the reference's)
#else
#define RD_SYM(name) name
#endif
#else
typedef uint16_t pel_t;
#define RD_SYM(name) name##_bd10
#endif
// The tool switches (hevcdl_config.tools).  The two 8-bit builds that carry the timed configurations are compiled for the reference cfg's tools: every test on a switch
// folds away (reading them at run time cost 1 % of the 600-frame job: three dependent LDS reads per TU coding, the plain quantiser's code inside every copy of
// code_tu_block).  rd_kernel_tools.hip (8-bit) and the 10-bit build read them from the context.
#ifndef HEVCDL_TOOLS_RT
#define HEVCDL_TOOLS_RT (HEVCDL_BD != 8)
#endif

namespace {
constexpr int BD = HEVCDL_BD, PEL_MAX = (1 << BD) - 1, QP_BD_OFFSET = 6 * (BD - 8);
#ifndef HEVCDL_NW
#define HEVCDL_NW 8
#endif
constexpr int NW = HEVCDL_NW;                                 // wavefronts per workgroup (one workgroup per CU)
#ifndef HEVCDL_NPEND
#define HEVCDL_NPEND 2          // measured with three (the code handles up to three; LDS has room in the eight-wave build): one frame 2.81 -> 2.85 s, 600 frames 6.19 -> 6.34 s -- more work thrown away at a restart, a quarter more workspace per wave
#endif
constexpr int NPEND = BD == 8 ? HEVCDL_NPEND : 1;                         // second luma passes a master may leave running behind it (the 10-bit kernel has LDS for one more region only)
#ifndef HEVCDL_AHEAD
#define HEVCDL_AHEAD 1
#endif
constexpr int AHEAD = (BD == 8 && HEVCDL_AHEAD) ? 1 : 0;       // first-pass candidates of the NEXT CU coded during this CU's chroma search (est_intra_chroma); needs one more region per wave
#ifdef HEVCDL_MICRO_SMALL
// -DHEVCDL_MICRO -DHEVCDL_MICRO_SMALL -DHEVCDL_NW=12|16: the occupancy experiment of tools/micro_rd.py.  Only hevcdl_micro_kernel of such a library may be launched: the
// LDS block of a wave is cut down to what the leaf routines of a TU coding touch (so that 12 / 16 waves fit a CU) and the search functions index beyond it.
constexpr int MSM = 1;
constexpr int NREG = 1;
#else
constexpr int MSM = 0;
constexpr int NREG = 1 + NPEND + AHEAD;
#endif                        // regions per wave: [0] first pass / chroma / rough-mode slices, [1..NPEND] the second passes (master: their tickets; chain owner: [1] its split tasks), [REG_AHEAD] the look-ahead
constexpr int REG_AHEAD = 1 + NPEND;
constexpr int NSLOT = 10 + 5 * NPEND;                                     // result slots of a master: 0..9 first pass, 5..9 chroma, then one set of 5 per second pass that can be pending
// per-wave global scratch: one LAYER SET = coefficient layers [4][6144] int16 + reconstruction layers [4][6144]; the wave's own set is
// followed by the best reconstruction of the CU under test and the task overlay (a CTU of trial reconstruction), then the RDOQ
// per-position arrays, then NSLOT result slots (a layer set + attribute arrays + coder states each)
constexpr int LAYER_SET = 4 * 6144 * 2 + 4 * 6144 * (int)sizeof(pel_t);
constexpr int SAVE_BYTES = 8192, N_SAVE = 1;   // the chain owner's state while it runs one of its own split tasks (spec_children)
constexpr int LOG_AHEAD = 68, LOG_AHEAD_N = 17;               // entries 68..84: the master's context as the look-ahead candidates see it (ahead_open)
constexpr int LOG_JOB = LOG_AHEAD + LOG_AHEAD_N, LOG_JOB_N = 1 + LOG_AHEAD_N;   // per pending second pass: its job block when another workgroup runs it (remote_post): header entry + context snapshot
constexpr int LOG_CJOB = LOG_JOB + NPEND * LOG_JOB_N, LOG_CJOB_N = 10 + LOG_AHEAD_N + 1;   // the five chroma modes of the CU under test as jobs for other workgroups: five headers, one context (+ the coder state they start from)
constexpr int LOG_COLD = LOG_CJOB + LOG_CJOB_N, LOG_COLD_N = 16;   // what a wave touches too rarely to keep in LDS: the coder snapshots next[4] / temp[4] (the CU walk, masters only) and test[4] (unsplit-vs-split of a first-pass TU), 21 words each, then the saved arrays of the best candidate (4 x 256 bytes)
constexpr int LEAF_LOG = 192, LOG_BYTES = (LOG_COLD + LOG_COLD_N) * LEAF_LOG;     // + entry 64: the CTU's entry coder state, 65: end state of the first pass's winner (enc_cu_syntax_fast), 66 / 67: levels / samples of the saved 2Nx2N candidate of an 8x8 CU     // per coded CU of the CTU: cost triple + coder state behind it (compress_cu: replay after a restart)
constexpr int SCR_LAYERS = (LAYER_SET + 2 * 6144 * (int)sizeof(pel_t) + N_SAVE * SAVE_BYTES + LOG_BYTES + 2047) & ~2047;
constexpr int SCR_RDOQ = 16384 + 16384;
constexpr int SLOT_BYTES = (LAYER_SET + 2048 + 2047) & ~2047;   // layer set, 1 KB of attribute arrays, coder state in (168 B at +1024) / out (+1280)
constexpr int SCR_WAVE = SCR_LAYERS + SCR_RDOQ + NSLOT * SLOT_BYTES;
constexpr int RQ_ROWS = BD == 8 ? 4 : 2;                       // coefficient groups RDOQ decides side by side (one per row of 16 lanes; the 10-bit kernel has LDS for two)
constexpr int SSE_SH = 2 * (BD - 8), HAD_SH = BD - 8;         // DISTORTION_PRECISION_ADJUSTMENT: per squared sample / per Hadamard sum

#define DEV __device__ __forceinline__
#define DEVN __device__ __noinline__
constexpr double MAX_DOUBLE = 1.7e+308;
enum { PLANAR = 0, DC = 1, HOR = 10, VER = 26, DM_CHROMA = 36 };
enum { SIZE_2Nx2N = 0, SIZE_NxN = 3, SIZE_NONE = 8 };
enum { SCAN_DIAG = 0, SCAN_HOR = 1, SCAN_VER = 2 };
enum { A_DEPTH = 0, A_PART, A_LDIR, A_CDIR, A_TRIDX, A_CBF, A_TSKIP = 8 };
// record field byte offsets (hevcdl_ctu_record)
enum { REC_BITS = 11 * 256, REC_DIST = REC_BITS + 4, REC_COST = REC_BITS + 8, REC_COEF = REC_BITS + 16, REC_SIZE = 15120 };

// ---- CABAC context layout (I-slice values: ContextTables.h:181-480; reference order TEncSbac.cpp:62-92) ----
enum { CTX_SPLIT = 0, CTX_PART_SIZE = 3, CTX_INTRA_PRED = 4, CTX_CHROMA_PRED = 5, CTX_QT_CBF = 6, CTX_SUBDIV = 16, CTX_SIG_CG = 19,
       CTX_SIG = 23, CTX_LAST_X = 67, CTX_LAST_Y = 97, CTX_ONE = 127, CTX_ABS = 151, CTX_TSKIP = 157, NUM_CTX = 159 };

__constant__ uint8_t c_ctx_init[NUM_CTX] = {
  139, 141, 157, 184, 184, 63,
  111, 141, 154, 154, 154, 94, 138, 182, 154, 154,
  153, 138, 138, 91, 171, 134, 141,
  111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 141,
  140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111, 111,
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
  138, 153, 136, 167, 152, 152, 139, 139 };
// ContextModel.cpp:68-101
__constant__ uint8_t c_next_mps[128] = {
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127 };
__constant__ uint8_t c_next_lps[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };
// ContextModel.cpp:103-112 (FAST_BIT_EST)
__constant__ int32_t c_entropy_bits[128] = {
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };
__constant__ int c_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };        // TComRom.cpp:354-357
__constant__ int c_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                        // TComRom.cpp:359-362
__constant__ uint8_t c_group_idx[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };   // TComRom.cpp:598
__constant__ uint8_t c_ctx_ind_map_4x4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };   // TComRom.cpp:589-595
__constant__ uint8_t c_num_rd_cand[5] = { 8, 8, 3, 3, 3 };                                  // TComRom.cpp:545-553
__constant__ uint8_t c_num_rd_cand_no_mpm[5] = { 9, 9, 4, 4, 5 };                           // TComRom.cpp:554-562 (FastUDIUseMPMEnabled 0)
__constant__ uint8_t c_intra_filter_thr[5] = { 10, 7, 1, 0, 10 };                           // TComPrediction.cpp:50-58
__constant__ int c_ang_table[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };                      // TComPrediction.cpp:265
__constant__ int c_inv_ang_table[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };     // TComPrediction.cpp:266
__constant__ int8_t c_dst4[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };   // TComRom.cpp:368-374
__constant__ int8_t c_dct_mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                      61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };   // TComRom.cpp:376-517

// Address spaces are spelled out: a generic pointer would make every LDS access a FLAT instruction (longer latency,
// and it ties LDS traffic to vmcnt, i.e. to every outstanding global store).
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))
struct __attribute__((aligned(8))) Cabac { uint8_t ctx[160]; unsigned long long frac; };   // 168 bytes
typedef LDS Cabac LCabac;
struct Rd { double cost; uint32_t bits, dist; };
struct DistCost { uint32_t dist; double cost; unsigned long long cfrac; };
struct TuRes { uint32_t dist, bits; };      // code_tu_block: distortion, and the TU's bit count where it was asked for   // cfrac: fractional bits (<< 15) of the TU tree's coefficient bins
struct Cu { int x, y, log2, depth, zbase, nparts, part; };
struct Tu { int x, y, log2, trd, zrel, nparts; };
DEV int uni(int v);

struct K {                             // wave-uniform kernel context (lives in LDS)
  int W, H, cw, ctus_x, addr, cx, cy, nctu;
  GLB const pel_t *org[3];
  GLB pel_t *rec[3];
  GLB unsigned char *records;          // frame's records (global)
  GLB const uint8_t *labels;           // frame's labels
  int tx0, ty0, tx1, ty1;              // luma rectangle of the tile being coded (the whole picture without tiles)
  GLB int16_t *coef_l;                 // ACTIVE layer set: [4 layers][6144] levels (Y 4096, Cb 1024, Cr 1024), z-order TU layout
  GLB pel_t *rec_l;                  //                   [4 layers][6144] CTU-relative reconstruction
  GLB pel_t *best_rec;               // scratch: [6144] best reconstruction of the CU under test (the master's)
  GLB double *q_cost;                  // scratch: RDOQ per-position costs [2][1024] (coded cost, sig cost); written/read lane-parallel,
  GLB int32_t *q_rate;                 //          and SBH inputs [4][1024] (rateIncUp, rateIncDown, sigRateDelta, deltaU)   (the executing wave's)
  GLB pel_t *ovl;                    // task overlay [6144], CTU-relative: trial reconstruction of the task being run (the executing wave's)
  GLB unsigned char *slots;            // the master's NSLOT result slots
  // While a task runs (in_task) its trial reconstruction goes to the overlay instead of the picture, and reference samples inside
  // the task rectangle (luma coordinates) come from there: the picture only ever holds what the master has committed.
  int in_task, trx0, try0, trx1, try1;
  // Origin of the active layer set: levels / reconstruction of a TU are stored relative to the block the set currently serves (the PU of
  // the luma search, the CU of a chroma task, the child of a split task), so that every PU reuses the same few KB of a set and the sets
  // of all waves stay cache resident (laid out CTU-wide they did not: profiles/r02_traffic.json).  lz: z-scan offset in luma samples.
  int lz, lx, ly;
  // The second luma passes of up to two EARLIER CUs of the CTU may still be running (their trial samples are in the picture): luma reference samples
  // inside such a CU's rectangle come from best_rec, which holds the reconstruction the search continued with (the first pass's winner).
  // -> srect[p]: (x0 | y0 << 16 | x1 << 32 | y1 << 48), 0 = none; ONE 8-byte word each, so that a helper copying this context never sees half a rectangle
  unsigned long long srect[3];
  GLB const pel_t *ssrc[3];            // ... and where the samples inside srect[p] are read: a plane of row stride 64 whose sample (0, 0) is the picture's (sorg & 0xffff, sorg >> 16)
  int sorg[4];
  int pset, pad_pset;                  // the slot set of the second pass this wave is running (run_task)
  double lambda, sqrt_lambda, cweight, lambda_c;
  double err_scale[2][4];
  long long sbh[2];
  int qp, qp_c;
  int dbg;
  int tools;                           // HEVCDL_TOOL_* bits the cfg leaves on (TransformSkip, SignHideFlag, StrongIntraSmoothing, FastUDIUseMPMEnabled can be off)
  GLB unsigned int *dbgbuf;
};
typedef const LDS K &KR;

struct __attribute__((aligned(16))) RdSmem {
  K k;
  // the executing wave's own scratch (K is copied from the master when a helper runs one of its tasks; these are not)
  GLB int16_t *my_coef; GLB pel_t *my_rec, *my_ovl; GLB double *my_qcost; GLB int32_t *my_qrate; GLB unsigned long long *my_save; GLB unsigned char *my_slots; GLB unsigned long long *my_log;
  int bound_reg, bound_child;          // a T_LUMA_SPLIT task under way: the chain owner's wave (its region [1] holds the chain's answers) and the child it is the split alternative of (recur_luma's early exit); -1: none
  int p2_pending, pad_p2;              // the second luma pass of the CU under test runs as a task (value: the region that holds its ticket); joined in check_rd_cost_intra or left pending
  Cabac go, curr[MSM ? 1 : 4], root[MSM ? 1 : 5], tbest, truec;   // snapshot slots by CU depth (0..3) / CU+TU depth (root: 0..4); test[] lives in the wave's HBM workspace, and so do next[] / temp[] in the ten-wave build (cold_state)
#if !defined(HEVCDL_RD_WIDE) && !defined(HEVCDL_MICRO_SMALL)
  Cabac nt[8];                        // next[4], temp[4]: the CU walk's snapshots (masters only)
  uint8_t sv[4][256];                 // saved best candidate: luma search trIdx / cbf / tskip in [0..2]; chroma search (later, disjoint in time) cbf Cb, Cr, tskip Cb, Cr
#endif
  uint8_t a[11][MSM ? 16 : 256];      // attribute arrays of the current CTU (flushed to the record at CTU end)
  int16_t line[MSM ? 8 : 264], fline[MSM ? 8 : 264];      // luma reference samples: bottom-left ... corner(2n) ... top-right; [1 2 1]-filtered copy
  int16_t cline[2][MSM ? 4 : 132];    // chroma reference samples (n <= 32, never filtered in 4:2:0)
  // the reference lines are kept across consecutive TU codings of the same block (candidate modes of one PU share
  // their neighbours): key = (log2 n, y, x) of the block each line currently holds, -1 = none
  int ref_key[3], fline_key;
  // residual (row stride n+2: conflict-free column access) and transform / dequantised coefficients (raster) share
  // storage: the residual is dead once the forward transform has consumed it and is rebuilt by the inverse transform.
  // Every value fits 16 bits (HEVC transform dynamic range; the reference clips the inverse stages explicitly).
  // For TUs up to 8x8 (95 % of all codings) the part behind the coefficients also holds RDOQ's per-position outputs (rdoq_wave).
  union __attribute__((aligned(16))) { int16_t resi[32 * 34]; int16_t tc[1024]; };
  // quantised levels of the current TU (raster); the transform intermediate (row stride n+1) lives behind the first 16
  // entries, i.e. a 4x4 block of levels survives the inverse transform (transform-skip bookkeeping needs it)
  int16_t lvl[16 + 32 * 33];
  pel_t pred[MSM ? 16 : 1024];     // prediction, then reconstruction, of the current TU
  pel_t ts_pred[3][16], ts_rec[3][16]; int16_t ts_coef[3][16];
  unsigned int rd_list[16];
  unsigned int bc_u32[4];             // lane-0 -> wave broadcasts
  unsigned int red_u32;
  unsigned long long est_bits, sse_acc[3];
  unsigned long long ctu_frac;        // fractional bits of the CU syntax of the CTU's coded CUs so far (compress_cu; advance_state)
  unsigned long long cfrac_last;      // coefficient part of the last intra_bits_qt count (fractional bits): luma
  unsigned long long cfrac_last_c;    // ... chroma (both components)
  // winners of the CU under test, for the short form of its syntax count (enc_cu_syntax_fast): coefficient bits of the luma / chroma winner, its slot
  unsigned long long lw_cfrac, cw_cfrac; int lw_valid, cw_slot;
  // SATD sums of the NEXT CU's rough mode decision, computed by the master while the workgroup's other waves run this CU's chroma search (rmd_prefetch)
  unsigned int satd_pre[NPEND >= 2 ? 36 : 2]; int pre_key, pre_open, chroma_key, pad_ck;      // chroma_key: key of the CU whose five chroma modes were posted together with its second pass (est_intra_luma; 0: none)   pre_open: key of the PU whose SATD slices are open in this wave's ticket region (0: none)
  // Second passes left running behind the master (compress_cu): carry_ok: the CU being coded may leave its pass pending; pend_*: the passes pending, oldest
  // first (index of their CU among the CTU's coded CUs, region of their ticket); restart: a pending pass chose the split -> the CTU is walked again, CUs
  // [0, replay_upto) from the log, CU nocarry_leaf with the result that pass reached (its luma is not searched again)
  int carry_ok, restart, leaf_idx, replay_upto, nocarry_leaf, pend_n, pend_leaf[3], pend_reg[3], left_pending, resume_reg;   // resume_reg: ticket region of the pass that won, whose result CU nocarry_leaf takes over
  uint8_t lab16[16];                  // the CTU's labels (the walk and the look-ahead read one per CU)
  uint8_t c8a[11][4];                 // saved 2Nx2N candidate of an 8x8 CU: attribute entries (levels and samples: entries 66 / 67 of the wave's log in HBM)
  // Look-ahead (est_intra_chroma -> est_intra_luma of the next CU): ahead_open 0 none / 1 region open / 2 frozen (no further claims); key of the PU, number of
  // candidates, tasks claimed before the freeze, fractional bits the candidates started from
  // a chroma component run for ANOTHER workgroup (remote_serve): where the pair of components of a mode meets -- a counter and the two distortions in the poster's job headers
  GLB int *rp_pair; GLB uint32_t *rp_half_mine, *rp_half_other;
  int chroma_jobs, pad_cj;            // chroma modes posted to other workgroups: 5 jobs (a mode each) or 10 (a component each)
  int ahead_open, ahead_key, ahead_n, ahead_claimed; unsigned int ahead_f0; int xctu;     // xctu: address of the CTU whose first CU the look-ahead was opened for while the CTU before it was finished (-1: none)
#ifdef HEVCDL_KERNEL_PROF
  GLB unsigned long long *my_prof; int prof_task, prof_pad; // timers of the profiling build (8-bit kernel, workgroup 0): 64 accumulators in HBM (cycles in the low 40 bits, calls above), added to with returnless atomics -- no LDS, no wait
#endif
#ifdef HEVCDL_MICRO_T
  unsigned long long mt_acc[12];      // -DHEVCDL_MICRO_T (tools/micro_rd.py --phases): cycles per phase of rdoq_wave, accumulated in scalar registers and added here once per call
#endif
  union {
    double chainb[3][RQ_ROWS * 16 + 1];  // rdoq_wave: zero-level cost, coded cost, significance cost of every position of a batch of groups
    double zb[2][64];                 // RDOQ, run of all-zero groups: zero-level costs / significance costs of 4 groups
    struct { double rmd_cost[36]; unsigned int satd[36]; };   // rough mode decision (never live during RDOQ)
  };
  uint8_t cgf[64];                    // significant-CG flags (RDOQ / bit counter)
  // RDOQ rate tables of the call (the contexts are frozen while a TU is quantised): significance [context - first][bin], greater-1 [set][c1][bin], greater-2 [set][bin]
  int32_t rq_sig[28][2], rq_g1[4][4][2], rq_g2[4][2];
  int last_bits[2][12];
};
typedef LDS RdSmem LSmem;

// Dynamic LDS = NW private blocks (one per wave) followed by the workgroup's shared part; every function reaches its wave's block directly
DEV int wave_id() { return __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_workitem_id_x() >> 6)); }
DEV LDS unsigned char *lds_base() { extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_[]; return (LDS unsigned char *)smem_raw_; }
DEV LSmem &lds_of(int w) { return *(LSmem *)(lds_base() + (size_t)w * sizeof(RdSmem)); }
DEV LSmem &lds() { return lds_of(wave_id()); }
DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// Inside a wave: "barrier" = ordering of this wave's own memory operations as seen by its other lanes.
// The hardware keeps a wave's LDS operations, and its vector-memory operations to one address, in program order, so
// a wavefront-scope fence (a compiler-only constraint: no s_waitcnt vmcnt(0), no cache action) is sufficient.
DEV void wsync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifdef HEVCDL_KERNEL_PROF
#define PROF_T0() const unsigned long long prof_t0_ = __builtin_readcyclecounter()
#define PROF_MARK0() unsigned long long prof_m_ = __builtin_readcyclecounter()
#define PROF_ACC_(id, d) do { GLB unsigned long long *pp_ = lds().my_prof; if (pp_) __hip_atomic_fetch_add(pp_ + (id), ((unsigned long long)(d) & 0xffffffffffull) + (1ull << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#define PROF_MARK(id) do { if (lane_id() == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); PROF_ACC_(id, n_ - prof_m_); prof_m_ = n_; } else prof_m_ = 0; } while (0)
#define PROF_ADD(k, id) do { if (lane_id() == 0) { PROF_ACC_(id, __builtin_readcyclecounter() - prof_t0_); } } while (0)
#if defined(HEVCDL_PROF_GLUE)
// -DHEVCDL_KERNEL_PROF -DHEVCDL_PROF_N=64 -DHEVCDL_PROF_GLUE (tools/phase_profile.py --glue): the RDOQ phase accumulators stay silent and take instead
//   19 / 20 / 21 / 22  the time inside code_tu_block + the bit counts by the kind of task that called them (first-pass candidate / chroma mode / split of a child / second pass's own chain)
//   34 split_bits   35 recur_luma: the unsplit TU put back   4 spec_children: own state saved / restored   5 run_task: before the search   8 run_task: behind it   18 recur_chroma
#define PROF_ADD_T(k, id, tid) do { if (lane_id() == 0) { const unsigned long long d_ = __builtin_readcyclecounter() - prof_t0_; PROF_ACC_(id, d_); const int pk_ = lds().prof_task; if (pk_) { PROF_ACC_(tid, d_); PROF_ACC_(18 + pk_, d_); } } } while (0)
#define PROF_TASK(v) do { if (lane_id() == 0) lds().prof_task = (v); } while (0)
#define PROF_GLUE_T0() const unsigned long long glue_t0_ = __builtin_readcyclecounter()
#define PROF_GLUE(id) do { if (lane_id() == 0) { PROF_ACC_(id, __builtin_readcyclecounter() - glue_t0_); } } while (0)
#else
#define PROF_ADD_T(k, id, tid) do { if (lane_id() == 0) { const unsigned long long d_ = __builtin_readcyclecounter() - prof_t0_; PROF_ACC_(id, d_); if (lds().prof_task) { PROF_ACC_(tid, d_); } } } while (0)
#define PROF_TASK(v) do { if (lane_id() == 0) lds().prof_task = (v); } while (0)
#define PROF_GLUE(id) do { } while (0)
#define PROF_GLUE_T0() do { } while (0)
#endif
#else
#define PROF_T0() do { } while (0)
#define PROF_MARK0() do { } while (0)
#define PROF_MARK(id) do { } while (0)
#define PROF_ADD(k, id) do { } while (0)
#define PROF_ADD_T(k, id, tid) do { } while (0)
#define PROF_TASK(v) do { } while (0)
#define PROF_GLUE(id) do { } while (0)
#define PROF_GLUE_T0() do { } while (0)
#endif
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEV int tools_of(KR k) { return HEVCDL_TOOLS_RT ? __builtin_amdgcn_readfirstlane(k.tools) : (int)HEVCDL_TOOLS_REFERENCE; }
// -DHEVCDL_TIMELINE (build with -DHEVCDL_KERNEL_DEBUG for the buffer): workgroup 0 logs (clock, wave, event, argument) while it codes CTUs [HEVCDL_TL_CTU0, +4) -- tools/timeline.py
#ifdef HEVCDL_TIMELINE
#ifndef HEVCDL_TL_CTU0
#define HEVCDL_TL_CTU0 300
#endif
#define TL(ev, arg) do { if (lane_id() == 0 && blockIdx.x == 0) { GLB unsigned int *b_ = lds().k.dbgbuf; const int a_ = lds().k.addr; if (b_ && a_ >= HEVCDL_TL_CTU0 && a_ < HEVCDL_TL_CTU0 + 4) { \
  const unsigned i_ = __hip_atomic_fetch_add(b_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (i_ < 3990) { b_[1 + 2 * i_] = (unsigned)__builtin_readcyclecounter(); b_[2 + 2 * i_] = ((unsigned)wave_id() << 24) | ((unsigned)(ev) << 16) | ((unsigned)(arg) & 0xffffu); } } } } while (0)
#else
#define TL(ev, arg) do { } while (0)
#endif
// -DHEVCDL_DBG_EXEC: trap (s99 = site) when a function that needs the whole wave is entered with lanes masked off; run under rocgdb
#ifdef HEVCDL_DBG_EXEC
#define CHECK_EXEC(id) do { if (__builtin_amdgcn_read_exec() != ~0ull) { asm volatile("s_mov_b32 s99, %0\n s_trap 2" :: "i"(id) : "s99"); } } while (0)
#else
#define CHECK_EXEC(id) do { } while (0)
#endif
// ---- regions: alternatives of the search handed to the waves of the workgroup (see the header comment) ----
enum { T_LUMA_P1 = 1, T_CHROMA = 2, T_LUMA_SPLIT = 3, T_LUMA_P2 = 4, T_RMD = 5, T_LUMA_AHEAD = 6, T_REMOTE = 7 };   // T_REMOTE: a second pass run by another workgroup (no ticket here; answered through HBM)
enum { SLOT_CHROMA = 5, SLOT_SPLIT = 10, SLOT_P2 = 14, SLOT_PSET = 5 };   // result slots: 0..9 the first pass, 5..9 the chroma modes (after it); per second pass (set p = its region - 1, slots + 5 p): 10..13 its split tasks (by child), 14 its verdict + start state
struct __attribute__((aligned(8))) Region {
  // ticket = (number of tasks << 16) | next task: ONE word, so that a claim (atomic add) returns a consistent pair -- a task index below
  // the count can only come from the region that is open, whose parameters were written before the ticket was
  int ticket, done, kind, owner;            // done: completion counter; owner: wave index of the master
  int cu[7], tu[6], pad_;                   // the CU under test (struct Cu), the PU (luma) or the CU's root TU (chroma) (struct Tu)
  int modes[12];                            // the alternatives: intra directions
  uint32_t dist[12]; double cost[12];       // the answers
  unsigned long long cfrac[12];             // coefficient fractional bits of an alternative's final bit count: first pass / chroma by task; second pass [0..3] split tasks, [4..7] the chain's unsplit codings
};
typedef LDS Region LRegion;
struct Tables {                        // read-only after kernel start, one copy per workgroup
  uint8_t r2z[256];                   // raster -> z-scan of the 16x16 partition grid (TComRom.cpp:284-352)
  // grouped-4x4 coefficient scans (TComRom.cpp:179-260) = CG order x order inside a CG, composed on the fly
  uint8_t scan_cg_all[3][88];         // CG order per [type][1 | 4 | 16 | 64 groups]
  uint8_t scan_in_cg[3][16];          // (y << 2) | x of the 16 positions of a CG, per scan type
  // small constant tables copied to LDS once per kernel: the serial RDOQ / bin-counting code reads them with
  // data-dependent indices, and an LDS read (~64 cycles) is several times cheaper than a constant-memory miss
  int32_t t_ebits[128]; uint8_t t_next[2][128];
  int t_ang[9], t_inv_ang[9]; uint8_t t_group_idx[32], t_ctx_map4[16], t_filter_thr[8];
};
struct __attribute__((aligned(16))) WgShared {
  Region reg[NW][NREG];               // see NREG
  int masters_active, quit, remote, bell;       // waves that currently walk a unit; quit: a workgroup without units has seen the last unit finish; remote: hevcdl_rd_params.remote; bell: counts the times tasks were put up in any region of the workgroup (ring_bell)
  GLB unsigned char *sched; unsigned long long pad2_;
  Tables tab;
};
DEV LDS WgShared &wg_shared() { return *(LDS WgShared *)(lds_base() + (size_t)NW * sizeof(RdSmem)); }
DEV LRegion &my_region(int which = 0) { return wg_shared().reg[wave_id()][which]; }
DEV LDS Tables &tb() { return wg_shared().tab; }
#ifndef HEVCDL_RMD_SLICE_ROUNDS
#define HEVCDL_RMD_SLICE_ROUNDS 3      // rough mode decisions of this many rounds of 64 (mode, block) tasks or more are dealt to the workgroup's waves
#endif
#ifndef HEVCDL_SPEC_MARGIN
#define HEVCDL_SPEC_MARGIN 0
#endif
#ifndef HEVCDL_PREFETCH
#define HEVCDL_PREFETCH 1
#endif
#ifndef HEVCDL_CARRY_MAX
#define HEVCDL_CARRY_MAX 2      // second passes are left pending behind a master only in workgroups with this many masters at most (with waves scarcer the work thrown away at a restart
                                // costs more than the waiting saved); measured on the 600-frame job (2-3 masters per workgroup): 1 -> 7.00 s, 2 -> 6.52 s, 3 -> 6.63 s
#endif
#ifndef HEVCDL_FG_FIRST_MAX
#define HEVCDL_FG_FIRST_MAX HEVCDL_CARRY_MAX   // helpers serve the masters' own regions before the pending passes up to this many masters (helper_step)
#endif
#ifndef HEVCDL_OWNER_LENDS
#define HEVCDL_OWNER_LENDS 0                  // 1: the owner of a split chain runs other masters' tasks while it waits for its split tasks (spec_children).  Measured on the 600-frame job: 6.23 -> 6.28 s (it returns late to its own chain, and the join of its pass waits), so off
#endif
#ifndef HEVCDL_CHROMA_ROOM
#define HEVCDL_CHROMA_ROOM 0                 // launches of 17..170 units: > 0 posts a CU's chroma modes to other workgroups when at least this many of them are idle beyond the jobs already queued.  Measured (16 / 48): 20 frames 3.16 -> 3.9 s, 40 frames 3.17 -> 4.2 / 3.9 s, 75 frames 3.21 -> 3.66 / 3.46 s -- the takers are needed for the second passes; 0 = never
#endif
#ifndef HEVCDL_PREFETCH_MAX
#define HEVCDL_PREFETCH_MAX 3                 // the master computes the next CU's rough-mode SATD during the chroma search up to this many masters (est_intra_chroma)
#endif
DEV int lds_load(LDS int *p);
// spare waves for the second-pass tasks: at least as many waves without a unit as with one (chain owners serve their own split tasks, so no
// wave ever waits on an unserved region)
DEV int spare_waves() { return 2 * lds_load(&wg_shared().masters_active) <= NW - HEVCDL_SPEC_MARGIN; }
DEV int lds_load(LDS int *p) { return uni(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
DEVN int lds_add(LDS int *p, int v)
{ // one atomic per wave (lane 0), result to every lane.  NOT inlined: inlined into a loop whose exit depends on the result, the lane-0
  // branch was merged with the loop's back edge and the loop went on with lane 0 alone (observed with ROCm 7.2's clang)
  int r = 0;
  if (lane_id() == 0) r = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return uni(r);
}
// -DHEVCDL_BELL (round 5, measured and NOT kept): a doorbell.  A wave without work walks the NREG * NW tickets again and again (40 dependent LDS round trips a
// scan, ~4 k cycles, then s_sleep 32).  With the bell it sleeps on ONE word that region_open / region_publish count up, and walks the tickets when it has moved.
// On 256-frame launches (one master, seven helpers per workgroup) that takes 7 % off the kernel's vector + scalar instructions (3.00 -> 2.79 M per CTU,
// profiles/r05b_bell_counters.txt) -- and costs time: one frame 2.82 -> 2.85 s, 600 frames 6.25 -> 6.31 s, 2048 frames 15.68 -> 15.75 s (two runs each, one
// box): the poll every 512 cycles and the call behind every ticket take more issue slots from the busy waves than the scans did, and a helper that finds its task
// 2 k cycles sooner shortens nothing (the tasks are 100 k cycles long).  s_wakeup behind the bell ends the other waves' s_sleep at once (tools/wakeup_probe.hip:
// a wave in s_sleep 32 notices a flag after 370 instead of 1 280 cycles) -- with it the kernel died with GPU memory faults in every launch of the independent
// form (three builds with it against three without), so it is not even an option.
#ifdef HEVCDL_BELL
constexpr int HELPER_BELL = 1;
DEVN void ring_bell() { if (lane_id() == 0) __hip_atomic_fetch_add(&wg_shared().bell, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
constexpr int HELPER_BELL = 0;
DEV void ring_bell() { }
#endif
#ifndef HEVCDL_IDLE_SLEEP
#define HEVCDL_IDLE_SLEEP 8          // x 64 cycles between two looks at the bell
#endif
// hand-over points between waves of the workgroup (same CU: LDS and the vector L1 are shared, workgroup scope is enough)
DEV void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
DEV void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
DEV GLB int16_t *slot_coef(GLB unsigned char *slots, int i) { return (GLB int16_t *)(slots + (size_t)i * SLOT_BYTES); }
DEV GLB pel_t *slot_rec(GLB unsigned char *slots, int i) { return (GLB pel_t *)(slots + (size_t)i * SLOT_BYTES + 4 * 6144 * 2); }
DEV GLB uint8_t *slot_attr(GLB unsigned char *slots, int i) { return (GLB uint8_t *)(slots + (size_t)i * SLOT_BYTES + LAYER_SET); }
DEV GLB unsigned long long *slot_state(GLB unsigned char *slots, int i, int out) { return (GLB unsigned long long *)(slots + (size_t)i * SLOT_BYTES + LAYER_SET + 1024 + 256 * out); }
// Every lane of the wave follows the same control path by construction; the values that steer it are copied to
// SGPRs (readfirstlane) so that branches enclosing barriers / calls are scalar branches, not EXEC-masked regions.
DEV bool ub(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
DEV Cu ucu(const Cu &c) { Cu r = { uni(c.x), uni(c.y), uni(c.log2), uni(c.depth), uni(c.zbase), uni(c.nparts), uni(c.part) }; return r; }
DEV Tu utu(const Tu &t) { Tu r = { uni(t.x), uni(t.y), uni(t.log2), uni(t.trd), uni(t.zrel), uni(t.nparts) }; return r; }
// Wave reductions on the DPP crossbar (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane ends up with
// the result of its row of 16), rows combined through v_readlane -> scalar.  An order of magnitude less latency than
// a butterfly of ds_bpermute shuffles; results are wave-uniform.
#define DPP_ROW_REDUCE(v, OP) do { \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false)); v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false)); \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false)); v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false)); } while (0)
DEV int op_add_(int a, int b) { return a + b; }
DEV int op_max_(int a, int b) { return a > b ? a : b; }
DEV int row_sum_i(int v) { DPP_ROW_REDUCE(v, op_add_); return v; }
DEV int wave_sum_i(int v)
{
  DPP_ROW_REDUCE(v, op_add_);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
DEV int wave_max_i(int v)
{
  DPP_ROW_REDUCE(v, op_max_);
  return op_max_(op_max_(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), op_max_(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
DEV int comp_off(int c) { return c == 0 ? 0 : (c == 1 ? 4096 : 5120); }
DEV int cstride(int c) { return c ? 32 : 64; }
DEV int pstride(KR k, int c) { return c ? k.cw : k.W; }
DEV int boff(KR k, int c, int x, int y) { const int s = c ? 32 : 64; return (y - k.cy * s) * s + (x - k.cx * s); }
// where the reconstruction of the block at (x, y) of component c goes (and where the same task reads it back): the picture, or the
// executing wave's overlay while a task runs
DEV int lay_coef_o(int lz, int log2, int c, int zabs) { const int o = zabs * 16 - lz; return (5 - log2) * 6144 + comp_off(c) + (c ? o >> 2 : o); }
DEV int lay_rec_o(int lx, int ly, int log2, int c, int x, int y) { const int sh = c ? 1 : 0; return (5 - log2) * 6144 + comp_off(c) + (y - (ly >> sh)) * cstride(c) + (x - (lx >> sh)); }
DEV int lay_coef(KR k, int log2, int c, int zabs) { return lay_coef_o(k.lz, log2, c, zabs); }                 // offset of a TU's levels in the active layer set
DEV int lay_rec(KR k, int log2, int c, int x, int y) { return lay_rec_o(k.lx, k.ly, log2, c, x, y); }          // ... of its reconstruction (row stride 64 / 32)
DEV GLB pel_t *rec_target(KR k, int c, int x, int y, int &stride)
{
  if (uni(k.in_task)) { stride = cstride(c); return k.ovl + comp_off(c) + boff(k, c, x, y); }
  stride = pstride(k, c); return k.rec[c] + (size_t)y * stride + x;
}
// log2 of a block size in {4,8,16,32,64}.  NOT 31-clz(n): that form let the compiler fold the constant part of a
// dynamic index into a FLAT instruction's immediate offset with a base BELOW the indexed private object; gfx9-family
// hardware picks the aperture from the base alone (offset ignored) -> HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.
DEV int ilog2(int n) { return n >= 32 ? (n >= 64 ? 6 : 5) : (n >= 16 ? 4 : (n >= 8 ? 3 : 2)); }
DEV int clip8(int v) { return v < 0 ? 0 : (v > PEL_MAX ? PEL_MAX : v); }   // ClipBD
DEV int clip16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// ---------------------------------------------------------------------------------------------------
// CABAC estimator (TEncBinCoderCABACCounter.cpp:60-140); the coder state lives in LDS and is touched by lane 0 only
// ---------------------------------------------------------------------------------------------------
DEV void enc_bin(LCabac *c, int ctx, int bin)
{
  const uint8_t st = c->ctx[ctx];
  c->frac += (unsigned long long)tb().t_ebits[st ^ bin];
  c->ctx[ctx] = tb().t_next[(st & 1) == bin][st];
}
DEV void enc_ep(LCabac *c, int n) { c->frac += 32768ull * (unsigned long long)n; }
DEV void reset_bits(LCabac *c) { c->frac &= 32767ull; }
DEV uint32_t get_bits(const LCabac *c) { return (uint32_t)(c->frac >> 15); }
DEV int ctx_bits(const LCabac *c, int ctx, int bin) { return tb().t_ebits[c->ctx[ctx] ^ bin]; }
DEV void cabac_copy(KR k, LCabac *dst, const LCabac *src)
{ // wave-parallel 168-byte snapshot copy (TEncSbac::load/store, TEncSbac.cpp:396-425)
  PROF_T0();
  wsync();
  if (lane_id() < 21) ((LDS unsigned long long *)dst)[lane_id()] = ((LDS const unsigned long long *)src)[lane_id()];
  wsync();
  PROF_ADD(k, 12);
}
#ifdef HEVCDL_STAGE_TRACE
// Stage-trace build (tests only, lib/libhevcdl_hip_trace.so): the events HM prints under DEBUG_INTRA_SEARCH_COSTS / DEBUG_TRANSFORM_AND_QUANTISE
// (TypeDef.h:59-60) go to a log in HBM, a record per event: 6 header words {kind, a, b, c, cost lo, cost hi} and, for TU events, three blocks of
// a * a values.  dbgbuf[0] = words used, the log starts at dbgbuf[2].  Waves log concurrently: the order of the records means nothing.
constexpr unsigned STAGE_CAP = 12u << 20;
DEVN GLB unsigned *stage_alloc(KR k, int words_)
{
  const int words = uni(words_);
  unsigned off = 0;
  if (lane_id() == 0) off = __hip_atomic_fetch_add(k.dbgbuf, (unsigned)words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  off = (unsigned)uni((int)off);
  return off + (unsigned)words <= STAGE_CAP ? k.dbgbuf + 2 + off : nullptr;
}
DEV void stage_line(KR k, int kind, int a, unsigned b, unsigned c, double cost, bool on)
{ // one record per enabled lane
  if (on) {
    const unsigned off = __hip_atomic_fetch_add(k.dbgbuf, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (off + 6u <= STAGE_CAP) {
      GLB unsigned *t = k.dbgbuf + 2 + off; const unsigned long long cb = (unsigned long long)__double_as_longlong(cost);
      t[0] = (unsigned)kind; t[1] = (unsigned)a; t[2] = b; t[3] = c; t[4] = (unsigned)cb; t[5] = (unsigned)(cb >> 32);
    }
  }
}
#endif
DEV double calc_rd_cost(KR k, uint32_t bits, uint32_t dist)
{ // TComRdCost.cpp:62-107
#if defined(HEVCDL_KERNEL_DEBUG) && !defined(HEVCDL_TIMELINE)
  if (k.dbgbuf && lane_id() == 0) { unsigned int n_ = k.dbgbuf[0]; if (n_ < 100000) { k.dbgbuf[1 + 2 * n_] = bits; k.dbgbuf[2 + 2 * n_] = dist; k.dbgbuf[0] = n_ + 1; } }
#endif
  return (double)dist + ((double)bits * k.lambda);
}

DEV void set_parts(KR k, LDS uint8_t *a, int z0, int n, int v)
{
  for (int i = lane_id(); i < n; i += 64) a[z0 + i] = (uint8_t)v;
}

// attribute of the 4x4 partition (x4,y4) of the picture: current CTU from LDS, earlier CTUs from their records
DEV int part_attr(KR k, int field, int x4, int y4)
{
  const int a = (y4 >> 4) * k.ctus_x + (x4 >> 4), z = tb().r2z[((y4 & 15) << 4) | (x4 & 15)];
  if (a == k.addr) return lds().a[field][z];
  return k.records[(size_t)a * REC_SIZE + field * 256 + z];
}

// ---------------------------------------------------------------------------------------------------
// reference samples (TComPattern.cpp:119-543)
// ---------------------------------------------------------------------------------------------------
DEV LDS int16_t *ref_line(int c) { return c ? lds().cline[c - 1] : lds().line; }
DEV void build_refs_i(KR k, int c_, int x_, int y_, int n_, int force_)
{ // _i: the body, inlined into code_tu_block (HEVCDL_REFS_INLINE); build_refs: the call form
  PROF_T0();
  const int c = uni(c_), x = uni(x_), y = uni(y_), n = uni(n_);
  const int key = (ilog2(n) << 24) | (y << 12) | x;
  if (!uni(force_) && uni(lds().ref_key[c]) == key) { PROF_ADD(k, 0); return; }
  wsync();
  if (lane_id() == 0) { lds().ref_key[c] = key; if (!c) lds().fline_key = -1; }
  LDS int16_t *line_out = ref_line(c);
  const int u = c ? 2 : 4, sh = c ? 1 : 2, nu = n >> sh;          // (every divisor below is a power of two: shifts, not the ~30-instruction division by a run-time value)
  const int x4 = x >> sh, y4 = y >> sh, total = 4 * nu + 1;
  // availability of the <= 65 units: lanes 0..63 + unit 64 (only for a 64x64 luma block) on lane 0's second pass.  unit_avail's rule, with the picture / tile
  // limits read ONCE into scalar registers and the tests combined without branches: as a chain of early returns on fields of the LDS-resident context it came out
  // as six dependent LDS round trips with a wait and a branch each, three times over (the left column, the corner and the top row took their own copies in turn)
  const int kW = uni(k.W), kH = uni(k.H), ktx0 = uni(k.tx0), kty0 = uni(k.ty0), ktx1 = uni(k.tx1), kty1 = uni(k.ty1), kcx = uni(k.ctus_x);
  const int lim_x1 = kW < ktx1 ? kW : ktx1, lim_y1 = kH < kty1 ? kH : kty1;          // (x4 * 4 < W and < tx1, y4 likewise; tx0 / ty0 >= 0)
  const int ca = (y4 >> 4) * kcx + (x4 >> 4), cz = tb().r2z[((y4 & 15) << 4) | (x4 & 15)];
  auto unit_flag = [&](int kk) -> int {
    const int ux = kk < 2 * nu ? x4 - 1 : (kk == 2 * nu ? x4 - 1 : x4 + (kk - 2 * nu - 1)), uy = kk < 2 * nu ? y4 + (2 * nu - 1 - kk) : y4 - 1;
    const int inside = (int)(ux * 4 >= ktx0) & (int)(uy * 4 >= kty0) & (int)(ux * 4 < lim_x1) & (int)(uy * 4 < lim_y1);      // inside the picture and the tile (another tile is never available)
    const int a = (uy >> 4) * kcx + (ux >> 4), z = tb().r2z[((uy & 15) << 4) | (ux & 15)];       // (the index stays inside the table for any coordinates)
    // an earlier CTU of the tile (CTUs of one tile are coded in raster order), or earlier z-order in this one (TComDataCU.cpp:985-1200).  "This one" is the CTU of the
    // block the line is gathered for, not k.addr: the look-ahead gathers for the first CU of the NEXT CTU while this one is finished (process_unit)
    return inside & (a != ca ? (int)(a < ca) : (int)(z < cz));
  };
  const int f0 = (lane_id() < total) ? unit_flag(lane_id()) : 0;
  const unsigned long long m0 = __ballot(f0);
  const int f64 = (total > 64) ? unit_flag(64) : 0;          // uniform
  const int st = pstride(k, c);
  GLB const pel_t *p = k.rec[c];
  // samples inside the rectangle of the running task come from its overlay, everything else from the picture
  const int csh = c ? 1 : 0, cs_ = c ? 32 : 64, task = uni(k.in_task);
  const int rx0 = k.trx0 >> csh, ry0 = k.try0 >> csh, rx1 = k.trx1 >> csh, ry1 = k.try1 >> csh;
  GLB const pel_t *po = k.ovl + comp_off(c);
  const int ox = k.cx * cs_, oy = k.cy * cs_;
  const unsigned long long sr0 = k.srect[0], sr1 = k.srect[1], sr2 = k.srect[2];
  const int spx0 = (int)(sr0 & 0xffff), spy0 = (int)((sr0 >> 16) & 0xffff), spx1 = (int)((sr0 >> 32) & 0xffff), spy1 = (int)(sr0 >> 48);
  const int sqx0 = (int)(sr1 & 0xffff), sqy0 = (int)((sr1 >> 16) & 0xffff), sqx1 = (int)((sr1 >> 32) & 0xffff), sqy1 = (int)(sr1 >> 48);
  GLB const pel_t *ps0 = k.ssrc[0], *ps1 = k.ssrc[1], *ps2 = k.ssrc[2];
  const int srx0 = (int)(sr2 & 0xffff), sry0 = (int)((sr2 >> 16) & 0xffff), srx1 = (int)((sr2 >> 32) & 0xffff), sry1 = (int)(sr2 >> 48), so2x = k.sorg[2] & 0xffff, so2y = k.sorg[2] >> 16;
  const int so0x = k.sorg[0] & 0xffff, so0y = k.sorg[0] >> 16, so1x = k.sorg[1] & 0xffff, so1y = k.sorg[1] >> 16;
  auto unit_start = [&](int kk) { return kk < 2 * nu ? kk * u : (kk == 2 * nu ? 2 * n : 2 * n + 1 + (kk - 2 * nu - 1) * u); };
  auto unit_len = [&](int kk) { return kk == 2 * nu ? 1 : u; };
  auto sample_ptr = [&](int i) -> GLB const pel_t * {        // the sample behind line index i
    int sx, sy;
    if (i < 2 * n) { sy = y + 2 * n - 1 - i; sx = x - 1; }
    else if (i == 2 * n) { sy = y - 1; sx = x - 1; }
    else { sy = y - 1; sx = x + (i - 2 * n - 1); }
    if (task && sx >= rx0 && sx < rx1 && sy >= ry0 && sy < ry1) return po + (sy - oy) * cs_ + (sx - ox);
    if (!c && sx >= spx0 && sx < spx1 && sy >= spy0 && sy < spy1) return ps0 + (sy - so0y) * 64 + (sx - so0x);     // an earlier CU whose second pass is pending
    if (!c && sx >= sqx0 && sx < sqx1 && sy >= sqy0 && sy < sqy1) return ps1 + (sy - so1y) * 64 + (sx - so1x);
    if (!c && sx >= srx0 && sx < srx1 && sy >= sry0 && sy < sry1) return ps2 + (sy - so2y) * 64 + (sx - so2x);
    return p + (size_t)sy * st + sx;
  };
  // Every line element is ONE picture sample: its own when its unit is available, otherwise the last sample of the nearest
  // available unit below, else the first sample of the first available unit above (TComPattern.cpp:350-543).  The source index
  // is pure bit arithmetic on the availability mask, so the (up to 5) loads of a lane are issued back to back -- one memory
  // round trip for the whole line instead of one per 64 elements.
  constexpr int MAX_IT = 5;                                 // 4 * 64 + 1 elements
  int val[MAX_IT]; bool have[MAX_IT];
#pragma unroll
  for (int it = 0; it < MAX_IT; it++) {
    const int i = lane_id() + 64 * it;
    have[it] = false; val[it] = 1 << (BD - 1);
    if (i <= 4 * n) {
      const int kk = i < 2 * n ? i >> sh : (i == 2 * n ? 2 * nu : 2 * nu + 1 + ((i - 2 * n - 1) >> sh));
      const int fl = kk < 64 ? (int)((m0 >> kk) & 1) : f64;
      int si = i;
      if (!fl) {
        const unsigned long long below = kk >= 64 ? m0 : (m0 & ((1ull << kk) - 1ull));
        const unsigned long long above = kk >= 63 ? 0ull : (m0 & ~((2ull << kk) - 1ull));
        if (below) { const int j = 63 - __clzll(below); si = unit_start(j) + unit_len(j) - 1; }
        else if (above) { const int j = __ffsll((long long)above) - 1; si = unit_start(j); }
        else if (f64) si = unit_start(64);
        else si = -1;                                       // nothing available: the default value
      }
      if (si >= 0) { have[it] = true; val[it] = (int)*sample_ptr(si); }
    }
  }
#pragma unroll
  for (int it = 0; it < MAX_IT; it++) { const int i = lane_id() + 64 * it; if (i <= 4 * n) line_out[i] = (int16_t)val[it]; }
  wsync();
  PROF_ADD(k, 0);
}

DEVN void build_refs(KR k, int c_, int x_, int y_, int n_, int force_) { build_refs_i(k, c_, x_, y_, n_, force_); }
#ifndef HEVCDL_REFS_INLINE
#define HEVCDL_REFS_INLINE 1
#endif

DEV void filter_refs(KR k, int n_)
{
  PROF_T0();
  const int n = uni(n_); // TComPattern.cpp:203-293 (luma; strong smoothing for n == 32)
  if (uni(lds().fline_key) == uni(lds().ref_key[0])) { PROF_ADD(k, 1); return; }
  wsync();
  if (lane_id() == 0) lds().fline_key = lds().ref_key[0];
  LDS const int16_t *src = lds().line; LDS int16_t *dst = lds().fline;
  const int n2 = 2 * n, last = 4 * n;
  int strong = 0;
  const int bl = src[0], tl = src[n2], tr = src[last];
  if (n >= 32 && (tools_of(k) & (int)HEVCDL_TOOL_STRONG_INTRA)) strong = (abs(bl + tl - 2 * src[n]) < (1 << (BD - 5))) && (abs(tl + tr - 2 * src[n2 + n]) < (1 << (BD - 5)));     // sps_strong_intra_smoothing_enable_flag
  for (int i = lane_id(); i <= last; i += 64) {
    int v;
    if (i == 0 || i == last) v = src[i];
    else if (strong) {
      const int shift = (n == 32) ? 6 : 7;
      if (i < n2) v = ((n2 - i) * bl + i * tl + n) >> shift;
      else if (i == n2) v = src[n2];
      else v = ((n2 - (i - n2)) * tl + (i - n2) * tr + n) >> shift;
    } else v = (src[i - 1] + 2 * src[i] + src[i + 1] + 2) >> 2;
    dst[i] = (int16_t)v;
  }
  wsync();
  PROF_ADD(k, 1);
}

DEV int use_filtered_refs(int c, int mode, int n)
{ // TComPattern.cpp:545-570; chroma never in 4:2:0
  if (c || mode == DC) return 0;
  const int d1 = abs(mode - HOR), d2 = abs(mode - VER), diff = d1 < d2 ? d1 : d2;
  return diff > tb().t_filter_thr[ilog2(n) - 2];
}

// closed-form intra prediction of one sample (TComPrediction.cpp:183-473, 731-817); dcval only for DC
DEV int pred_pixel(LDS const int16_t *line, int c, int mode, int n, int log2n, int px, int py, int dcval)
{
  const int n2 = 2 * n;
  if (mode == PLANAR) {
    const int left = line[n2 - 1 - py], top = line[n2 + 1 + px], bl = line[n2 - 1 - n], tr = line[n2 + 1 + n];
    return ((n - 1 - px) * left + (px + 1) * tr + (n - 1 - py) * top + (py + 1) * bl + n) >> (log2n + 1);
  }
  if (mode == DC) {
    if (!c && n <= 16) {
      if (px == 0 && py == 0) return (line[n2 + 1] + line[n2 - 1] + 2 * dcval + 2) >> 2;
      if (py == 0) return (line[n2 + 1 + px] + 3 * dcval + 2) >> 2;
      if (px == 0) return (line[n2 - 1 - py] + 3 * dcval + 2) >> 2;
    }
    return dcval;
  }
  const int is_ver = mode >= 18;
  const int ang_mode = is_ver ? mode - VER : -(mode - HOR);
  const int abs_ang = abs(ang_mode);
  const int angle = (ang_mode < 0 ? -1 : 1) * tb().t_ang[abs_ang];
  const int inv_angle = tb().t_inv_ang[abs_ang];
  const int x = is_ver ? px : py, y = is_ver ? py : px;
  auto ref = [&](int i) -> int {
    if (i >= 0) return is_ver ? line[n2 + i] : line[n2 - i];
    const int j = (128 + (-i) * inv_angle) >> 8;
    return is_ver ? line[n2 - j] : line[n2 + j];
  };
  if (angle == 0) {
    int v = ref(x + 1);
    if (!c && n <= 16 && x == 0) { const int s1 = is_ver ? line[n2 - (y + 1)] : line[n2 + (y + 1)], s0 = line[n2]; v = clip8(v + ((s1 - s0) >> 1)); }
    return v;
  }
  const int dpos = (y + 1) * angle, di = dpos >> 5, df = dpos & 31;
  if (df) return ((32 - df) * ref(x + di + 1) + df * ref(x + di + 2) + 16) >> 5;
  return ref(x + di + 1);
}

DEV int dc_value(KR k, LDS const int16_t *line, int n)
{ // predIntraGetPredValDC TComPrediction.cpp:183-201
  const int n2 = 2 * n;
  int s = 0;
  for (int i = lane_id(); i < n; i += 64) s += line[n2 + 1 + i] + line[n2 - 1 - i];
  s = wave_sum_i(s);
  return (s + n) >> (ilog2(n) + 1);                           // / (2 n)
}

// prediction of an n x n TU (n <= 32) into s->pred (stride n)
DEV void predict_block(KR k, int c_, int mode_, int n_)
{
  PROF_T0();
  const int c = uni(c_), mode = uni(mode_), n = uni(n_);
  LDS const int16_t *line = use_filtered_refs(c, mode, n) ? lds().fline : ref_line(c);
  const int log2n = ilog2(n);
  const int dcv = (mode == DC) ? dc_value(k, line, n) : 0;
  for (int i = lane_id(); i < n * n; i += 64) lds().pred[i] = (pel_t)pred_pixel(line, c, mode, n, log2n, i & (n - 1), i >> log2n, dcv);
  wsync();
  PROF_ADD(k, 3);
}

// ---------------------------------------------------------------------------------------------------
// transforms (TComTrQuant.cpp:388-987): lane-parallel dot products, matrices in LDS
// ---------------------------------------------------------------------------------------------------
// ---- transform kernels ----------------------------------------------------------------------------------
// One lane per row/column runs the 1-D transform of the reference as its partial-butterfly factorisation
// (TComTrQuant.cpp:388-855) with compile-time matrix entries; integer arithmetic is exact, so the factorisation
// equals the matrix product the oracle uses.  LDS layouts are padded so that every access is conflict-free:
//   resi  int16  row stride n+2      tmp  int32  row stride n+1      tc  int32  raster (stride n)
constexpr int DCT_MAG[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                              61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
constexpr int dct32(int k, int n) { int m = ((2 * n + 1) * k) & 127; if (m > 64) m = 128 - m; return m <= 32 ? DCT_MAG[m] : -DCT_MAG[64 - m]; }
template <int N> constexpr int dctn(int k, int n) { return dct32(k * (32 / N), n); }     // T_N[k][n]
constexpr int DST4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
DEV int RS(int n) { return n + 2; }
DEV int TS(int n) { return n + 1; }

template <int N> DEV void fwd1d(const int (&x)[N], int (&y)[N])
{ // y[k] = sum_n T_N[k][n] x[n]
  if constexpr (N == 2) { y[0] = 64 * (x[0] + x[1]); y[1] = 64 * (x[0] - x[1]); }
  else {
    int e[N / 2], o[N / 2], ye[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; k++) { e[k] = x[k] + x[N - 1 - k]; o[k] = x[k] - x[N - 1 - k]; }
    fwd1d<N / 2>(e, ye);
#pragma unroll
    for (int r = 0; r < N / 2; r++) {
      y[2 * r] = ye[r];
      int acc = 0;
#pragma unroll
      for (int k = 0; k < N / 2; k++) acc += dctn<N>(2 * r + 1, k) * o[k];
      y[2 * r + 1] = acc;
    }
  }
}
template <int N> DEV void inv1d(const int (&c)[N], int (&x)[N])
{ // x[n] = sum_k T_N[k][n] c[k]
  if constexpr (N == 2) { x[0] = 64 * (c[0] + c[1]); x[1] = 64 * (c[0] - c[1]); }
  else {
    int ce[N / 2], e[N / 2];
#pragma unroll
    for (int r = 0; r < N / 2; r++) ce[r] = c[2 * r];
    inv1d<N / 2>(ce, e);
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
      int o = 0;
#pragma unroll
      for (int r = 0; r < N / 2; r++) o += dctn<N>(2 * r + 1, k) * c[2 * r + 1];
      x[k] = e[k] + o; x[N - 1 - k] = e[k] - o;
    }
  }
}
DEV void dst4_fwd(const int (&x)[4], int (&y)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) y[k] = DST4[k][0] * x[0] + DST4[k][1] * x[1] + DST4[k][2] * x[2] + DST4[k][3] * x[3];
}
DEV void dst4_inv(const int (&c)[4], int (&x)[4]) {
#pragma unroll
  for (int n = 0; n < 4; n++) x[n] = DST4[0][n] * c[0] + DST4[1][n] * c[1] + DST4[2][n] * c[2] + DST4[3][n] * c[3];
}

template <int N, bool DST> DEV void fwd_transform_n(KR k)
{ // s->resi (stride RS) -> s->tc (raster); xTrMxN TComTrQuant.cpp:860-915
  constexpr int LOG2 = (N == 4) ? 2 : (N == 8) ? 3 : (N == 16) ? 4 : 5;
  constexpr int s1 = LOG2 + BD - 9, s2 = LOG2 + 6, a1 = 1 << (s1 - 1), a2 = 1 << (s2 - 1);
  LSmem &s = lds();
  LDS int16_t *tmp_ = s.lvl + 16;
  if (lane_id() < N) {
    int x[N], y[N];
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = s.resi[lane_id() * (N + 2) + i];
    if constexpr (DST) dst4_fwd(x, y); else fwd1d<N>(x, y);
#pragma unroll
    for (int kk = 0; kk < N; kk++) tmp_[lane_id() * (N + 1) + kk] = (int16_t)((y[kk] + a1) >> s1);       // tmp[j][kk]
  }
  wsync();
  if (lane_id() < N) {
    int x[N], y[N];
#pragma unroll
    for (int j = 0; j < N; j++) x[j] = tmp_[j * (N + 1) + lane_id()];                          // column kk = lane
    if constexpr (DST) dst4_fwd(x, y); else fwd1d<N>(x, y);
#pragma unroll
    for (int k2 = 0; k2 < N; k2++) s.tc[k2 * N + lane_id()] = (int16_t)((y[k2] + a2) >> s2);
  }
  wsync();
}
template <int N, bool DST> DEV void inv_transform_n(KR k)
{ // s->tc (dequantised, raster) -> s->resi (stride RS); xITrMxN TComTrQuant.cpp:927-987
  LSmem &s = lds();
  // the intermediate of a transform up to 16 x 16 (544 bytes) lies in RDOQ's addend area, dead outside rdoq_wave: the TU's LEVELS in s.lvl then survive the inverse
  // transform, and the bit count that follows every coding (luma_tu_bits) reads them there instead of fetching them back from the layer buffer in HBM
  static_assert(16 * 17 * 2 <= sizeof(s.chainb), "inverse-transform intermediate");
  LDS int16_t *tmp_ = N <= 16 ? (LDS int16_t *)&s.chainb[0][0] : s.lvl + 16;
  if (lane_id() < N) {
    int c[N], x[N];
#pragma unroll
    for (int kk = 0; kk < N; kk++) c[kk] = s.tc[kk * N + lane_id()];                               // column j = lane
    if constexpr (DST) dst4_inv(c, x); else inv1d<N>(c, x);
#pragma unroll
    for (int i = 0; i < N; i++) tmp_[lane_id() * (N + 1) + i] = (int16_t)clip16((x[i] + 64) >> 7);       // tmp[j][x]
  }
  wsync();
  if (lane_id() < N) {
    int c[N], x[N];
#pragma unroll
    for (int u = 0; u < N; u++) c[u] = tmp_[u * (N + 1) + lane_id()];                           // row y = lane
    if constexpr (DST) dst4_inv(c, x); else inv1d<N>(c, x);
#pragma unroll
    for (int i = 0; i < N; i++) s.resi[lane_id() * (N + 2) + i] = (int16_t)clip16((x[i] + (1 << (19 - BD))) >> (20 - BD));
  }
  wsync();
}
DEV void fwd_transform(KR k, int n_, int use_dst_)
{
  const int n = uni(n_), use_dst = uni(use_dst_);
  if (n == 32) fwd_transform_n<32, false>(k);
  else if (n == 16) fwd_transform_n<16, false>(k);
  else if (n == 8) fwd_transform_n<8, false>(k);
  else if (use_dst) fwd_transform_n<4, true>(k);
  else fwd_transform_n<4, false>(k);
}
DEV void inv_transform(KR k, int n_, int use_dst_)
{
  const int n = uni(n_), use_dst = uni(use_dst_);
  if (n == 32) inv_transform_n<32, false>(k);
  else if (n == 16) inv_transform_n<16, false>(k);
  else if (n == 8) inv_transform_n<8, false>(k);
  else if (use_dst) inv_transform_n<4, true>(k);
  else inv_transform_n<4, false>(k);
}

// ---------------------------------------------------------------------------------------------------
// coefficient-coding geometry shared by RDOQ and the bit counter
// ---------------------------------------------------------------------------------------------------
struct CParam { int log2, n, ch, scan_type, wg, first_sig_ctx; };
// scan position -> raster position of an n x n block: CG order (LDS) composed with the order inside a CG (LDS)
struct ScanFn {
  LDS const uint8_t *cg; LDS const uint8_t *in; int log2n, l;
  DEV int operator[](int sp) const
  {
    const int g = cg[sp >> 4], p = in[sp & 15], gy = g >> l, gx = g - (gy << l);
    return (((gy << 2) + (p >> 2)) << log2n) + (gx << 2) + (p & 3);
  }
};

DEV int coef_scan_idx(int c, int n, int dir_mode)
{ // TComDataCU.cpp:3150-3209
  if (n > (c ? 4 : 8)) return SCAN_DIAG;
  if (abs(dir_mode - VER) <= 4) return SCAN_HOR;
  if (abs(dir_mode - HOR) <= 4) return SCAN_VER;
  return SCAN_DIAG;
}
DEV void get_cparam(CParam &cp, int c, int n, int dir_mode)
{ // TComChromaFormat.cpp:96-160
  cp.n = n; cp.log2 = ilog2(n); cp.ch = c ? 1 : 0; cp.scan_type = coef_scan_idx(c, n, dir_mode); cp.wg = n >> 2;
  if (n == 4) cp.first_sig_ctx = 0;
  else if (n == 8) cp.first_sig_ctx = 9 + ((cp.scan_type != SCAN_DIAG) ? (cp.ch ? 0 : 6) : 0);
  else cp.first_sig_ctx = cp.ch ? 12 : 21;
}
DEV void scan_next(int type, int bw, int bh, int &line, int &col)
{ // ScanGenerator::GetNextIndex TComRom.cpp:100-160
  if (type == SCAN_DIAG) {
    if (col == bw - 1 || line == 0) { line += col + 1; col = 0; if (line >= bh) { col += line - (bh - 1); line = bh - 1; } }
    else { col++; line--; }
  } else if (type == SCAN_HOR) { if (col == bw - 1) { line++; col = 0; } else col++; }
  else { if (line == bh - 1) { col++; line = 0; } else line++; }
}
DEV int pattern_sig_ctx(LDS const uint8_t *cgf, int gx, int gy, int wg)
{ // TComTrQuant.cpp:2672-2705
  if (wg <= 1) return 0;
  const int r = (gx < wg - 1) ? (cgf[gy * wg + gx + 1] != 0) : 0, l = (gy < wg - 1) ? (cgf[(gy + 1) * wg + gx] != 0) : 0;
  return r + (l << 1);
}
DEV int sig_cg_ctx(LDS const uint8_t *cgf, int gx, int gy, int wg)
{ // TComTrQuant.cpp:3023-3049
  const int r = (gx < wg - 1) ? (cgf[gy * wg + gx + 1] != 0) : 0, l = (gy < wg - 1) ? (cgf[(gy + 1) * wg + gx] != 0) : 0;
  return (r + l) != 0;
}
DEV int sig_ctx_inc(const CParam &cp, const ScanFn &scan, int pat, int scan_pos)
{ // TComTrQuant.cpp:2707-2803
  const int raster = scan[scan_pos], py = raster >> cp.log2, px = raster - (py << cp.log2);
  if (px + py == 0) return 0;
  int offset;
  if (cp.log2 == 2) offset = tb().t_ctx_map4[4 * py + px];
  else {
    int cnt; const int xs = px & 3, ys = py & 3;
    if (pat == 0) cnt = (xs + ys >= 3) ? 0 : ((xs + ys >= 1) ? 1 : 2);
    else if (pat == 1) cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2);
    else if (pat == 2) cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2);
    else cnt = 2;
    const int not_first = ((px >> 2) + (py >> 2)) > 0;
    offset = (not_first ? (cp.ch ? 0 : 3) : 0) + cnt;
  }
  return cp.first_sig_ctx + offset;
}
// the count of TComTrQuant.cpp:2740-2760 by significance pattern and position inside the group, two bits per (ys, xs): one 32-bit constant per pattern
constexpr unsigned sig_cnt_table(int pat)
{
  unsigned t = 0;
  for (int ys = 0; ys < 4; ys++) for (int xs = 0; xs < 4; xs++) {
    int cnt = 2;
    if (pat == 0) cnt = (xs + ys >= 3) ? 0 : ((xs + ys >= 1) ? 1 : 2);
    else if (pat == 1) cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2);
    else if (pat == 2) cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2);
    else cnt = 2;
    t |= (unsigned)cnt << (2 * (ys * 4 + xs));
  }
  return t;
}
DEV int sig_ctx_inc_xy(const CParam &cp, int pat, int px, int py)
{ // sig_ctx_inc from the block coordinates of the position; per-lane pattern and position: selects, no branches (the branches were EXEC-masked regions, ~35 instructions)
  int offset;
  if (cp.log2 == 2) offset = tb().t_ctx_map4[4 * py + px];       // (wave-uniform)
  else {
    constexpr unsigned T0 = sig_cnt_table(0), T1 = sig_cnt_table(1), T2 = sig_cnt_table(2), T3 = sig_cnt_table(3);
    const unsigned tbl = pat == 0 ? T0 : (pat == 1 ? T1 : (pat == 2 ? T2 : T3));
    const int cnt = (int)((tbl >> (2 * (((py & 3) << 2) | (px & 3)))) & 3u);
    const int not_first = ((px | py) >> 2) != 0;
    offset = ((not_first && !cp.ch) ? 3 : 0) + cnt;
  }
  return (px | py) == 0 ? 0 : cp.first_sig_ctx + offset;
}
DEV ScanFn scan_of(const LSmem &s, int type, int log2n)
{
  ScanFn f; f.cg = tb().scan_cg_all[type] + (log2n == 2 ? 0 : (log2n == 3 ? 1 : (log2n == 4 ? 5 : 21))); f.in = tb().scan_in_cg[type]; f.log2n = log2n; f.l = log2n - 2;
  return f;
}
DEV LDS const uint8_t *scan_cg_of(const LSmem &s, int type, int log2n) { return tb().scan_cg_all[type] + (log2n == 2 ? 0 : (log2n == 3 ? 1 : (log2n == 4 ? 5 : 21))); }
DEV int ctx_set_index(int ch, int subset, int found_gt1) { return (ch ? 4 : 0) + ((!ch && subset > 0) ? 2 : 0) + (found_gt1 ? 1 : 0); }   // TComChromaFormat.h:243-251
DEV void last_ctx_params(int ch, int n, int &off, int &shift)
{ // TComChromaFormat.h:211-226
  const int cw = ilog2(n) - 2;
  off = ch ? 0 : (cw * 3 + ((cw + 1) >> 2));
  shift = ch ? cw : ((cw + 3) >> 2);
}

// ---------------------------------------------------------------------------------------------------
// RDOQ (TComTrQuant.cpp:2119-2661, helpers :2812-2996).  The rate tables (estBitsSbacStruct, TEncSbac.cpp:1726-1970) are the frozen contexts of `cab`.
// ---------------------------------------------------------------------------------------------------
DEV double rl_d(double v, int l)
{ // value of lane l (wave-uniform l) of a per-lane double
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

#if defined(HEVCDL_KERNEL_PROF) && defined(HEVCDL_PROF_MASTER)
// the master's serial sections, on the accumulators of the RDOQ phase timers (build with -DHEVCDL_PROF_N=64: those then stay silent)
#define MT0() unsigned long long mt_ = __builtin_readcyclecounter()
#define MT(id) do { if (lane_id() == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); PROF_ACC_(id, n_ - mt_); mt_ = n_; } else mt_ = 0; } while (0)
#define MTR() do { mt_ = __builtin_readcyclecounter(); } while (0)
#else
#define MTR() do { } while (0)
#define MT0() do { } while (0)
#define MT(id) do { } while (0)
#endif
#if defined(HEVCDL_MICRO_T)
// phase timers of the micro-benchmark build: a mark costs one s_memtime and two scalar additions (no memory traffic inside the routine)
constexpr int mt_slot(int id) { return id == 18 ? 0 : id == 19 ? 1 : id == 4 ? 2 : id == 5 ? 3 : id == 34 ? 4 : id == 8 ? 5 : id == 20 ? 6 : id == 21 ? 7 : id == 22 ? 8 : id == 6 ? 9 : 10; }
#define RDOQ_MARK(id) do { const unsigned long long n_ = __builtin_readcyclecounter(); mt_[mt_slot(id)] += n_ - pt_; pt_ = n_; } while (0)
#elif defined(HEVCDL_KERNEL_PROF)
#ifndef HEVCDL_PROF_N
#define HEVCDL_PROF_N 0                                 // != 0: the RDOQ phase timers count TUs of this size only
#endif
#define RDOQ_MARK(id) do { if (lane == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); if (HEVCDL_PROF_N == 0 || n == HEVCDL_PROF_N) PROF_ACC_(id, n_ - pt_); pt_ = n_; } } while (0)
#else
#define RDOQ_MARK(id) do { } while (0)
#endif
#ifdef HEVCDL_RDOQ_STOP
#define RDOQ_STOP(id) do { if (HEVCDL_RDOQ_STOP == (id)) return 0; } while (0)
#define RDOQ_SKIP(id) if (HEVCDL_RDOQ_STOP == (id)) { cgpos -= R; continue; }
#else
#define RDOQ_STOP(id) do { } while (0)
#define RDOQ_SKIP(id)
#endif
#ifdef HEVCDL_RDOQ_STOP
#define HEVCDL_RDOQ_STOPV HEVCDL_RDOQ_STOP
#else
#define HEVCDL_RDOQ_STOPV 0
#endif
// RDOQ, whole wave, coefficient groups in batches.  s->tc -> s->lvl ; returns uiAbsSum.  Same arithmetic, same order of every fp64 sum as the reference
// (TComTrQuant.cpp:2119-2661); what changes is how the work inside phase B is laid out:
//   * What a group's level decisions depend on: its own positions (the c1 / c2 / Rice state machine runs inside a group and starts afresh in
//     each), the context set (one carried bit: did the previous group in scan order end with c1 == 0), and the significance pattern (are the
//     groups to the right / below significant AFTER their own RDOQ).  They do NOT depend on the running cost sums.  So up to RQ_ROWS consecutive
//     groups are decided side by side, one per row of 16 lanes, when (a) none of them has a not-yet-final neighbour inside the batch -- groups
//     of one anti-diagonal never are neighbours, groups without a rounded level are final (insignificant) from the start -- and (b) the carried
//     bit of every row but the first is known beforehand: a group whose largest rounded level is >= 3 keeps a level > 1 (the candidates are
//     the rounded level and one below), one whose largest is <= 1 cannot have any; only "largest == 2" ends a batch.
//   * Inside a row the state a position sees is a fold over the decided levels above it (counts of levels, of levels > 1, of levels == 1;
//     the Rice parameter only moves when a level > 3 exists).  Every lane evaluates xGetCodedLevel for its own position under the state the
//     current levels imply, starting from the rounded levels; the levels are re-derived until nothing changes.  The system is triangular (a
//     position only depends on higher scan positions), so the fixed point is the sequential result; it is reached after one pass more than
//     the number of decisions that differ from the guess (typically 2 passes for the whole batch instead of one serial visit per level).
//   * The ordered sums and the group-level tests (zero-out of a group, TComTrQuant.cpp:2385-2440) then run group by group in scan order,
//     addends transposed through LDS once per batch.
// NFIX != 0: the TU size as a compile-time constant (the 4x4 / 8x8 copies: group counts, loop bounds and the masks over the groups fold).  HEVCDL_RDOQ_FIX: the
// largest size that gets a copy of its own.  Measured (round 5): per call 4x4 8 175 -> 6 795 cycles, 8x8 20 035 -> 17 970; whole kernel, none / 4 / 8 / 16:
// one frame 2.62 / 2.60 / 2.59 / 2.59 s, 600 frames 5.77 / 5.72 / 5.70 / 5.71 s, 2048 frames 14.55 / 14.67 / 14.56 / 14.64 s (every copy is 26 KB more code)
#ifndef HEVCDL_RDOQ_FIX
#define HEVCDL_RDOQ_FIX 8
#endif
template <int NFIX = 0> DEV uint32_t rdoq_wave(KR k, const LCabac *cab, int c_, int n_, int dir_mode_, int cbf_ctx_)
{
  const int c = uni(c_), n = NFIX ? NFIX : uni(n_), dir_mode = uni(dir_mode_), cbf_ctx = uni(cbf_ctx_);
  LSmem &s = lds();
  const int lane = lane_id();
  const int ch = c ? 1 : 0, log2n = ilog2(n);
  const int qp = uni(c ? k.qp_c : k.qp) + QP_BD_OFFSET, per = qp / 6, rem = qp % 6;      // + qpBdOffset (TComTrQuant.cpp:71-100)
  const int sign_hide = tools_of(k) & (int)HEVCDL_TOOL_SIGN_HIDE;                       // (read here, with the other words of the context: one wait for all of them)
  const int tshift = 15 - BD - log2n, qbits = 14 + per + tshift;
  const double lambda = c ? k.lambda_c : k.lambda;
  const double err_scale = k.err_scale[ch][log2n - 2];
  const int qcoef = uni(c_quant_scales[rem]);
  const int ncoef = n * n;
  CParam cp; get_cparam(cp, c, n, dir_mode);
  const ScanFn scan = scan_of(s, cp.scan_type, log2n); LDS const uint8_t *scan_cg = scan_cg_of(s, cp.scan_type, log2n);
  LDS const int16_t *src = s.tc; LDS int16_t *dst = s.lvl;
  // Per-position outputs read again by the last-position search and by sign hiding: coded cost and significance cost by scan position, the four rate / error terms of
  // sign hiding by raster position -- 32 bytes per position.  Up to 8x8 they fit behind the coefficients in LDS (the residual that shares that storage is dead until the
  // inverse transform: 128 + 64 * 32 = 2176 bytes); larger TUs use the wave's HBM workspace.
  const bool qlds = n <= 8;
  GLB double *gq_cost = k.q_cost; GLB int32_t *gq_rate = k.q_rate;
  LDS double *lq_cost = (LDS double *)((LDS char *)s.tc + 2 * ncoef); LDS int32_t *lq_rate = (LDS int32_t *)(lq_cost + 2 * ncoef);
  enum { Q_COEFF = 0, Q_SIG = 1, Q_UP = 0, Q_DOWN = 1, Q_SIGDELTA = 2, Q_DELTAU = 3 };
  auto q_cost_st = [&](int a, int i, double v) { if (qlds) lq_cost[a * ncoef + i] = v; else gq_cost[a * 1024 + i] = v; };
  auto q_cost_ld = [&](int a, int i) -> double { double v; if (qlds) v = lq_cost[a * ncoef + i]; else v = gq_cost[a * 1024 + i]; return v; };
  auto q_rate_st = [&](int a, int i, int32_t v) { if (qlds) lq_rate[a * ncoef + i] = v; else gq_rate[a * 1024 + i] = v; };
  auto q_rate_ld = [&](int a, int i) -> int32_t { int32_t v; if (qlds) v = lq_rate[a * ncoef + i]; else v = gq_rate[a * 1024 + i]; return v; };
  // the rate of every group's significance flag as it entered the cost (read again by the last-position search) is one of four values -- flag 0 / 1 under context 0 / 1 --
  // or nothing: three masks over the groups (bit = index in scan order) instead of an array of costs
  unsigned long long cgs_set = 0, cgs_ctx = 0, cgs_one = 0;
  // lLevelDouble = min(|coefficient| * quantiser scale, MAX_INT - (1 << (qbits - 1))) (TComTrQuant.cpp:2180-2183): the product of a 16-bit coefficient (<= 32768) and a
  // scale <= 26214 is below 2^30, the limit at least 2^31 - 1 - 2^26 (qbits <= 27 at every bit depth, QP and TU size of this path): the minimum never binds and
  // everything fits 32 bits -- one v_mul_u32_u24 instead of a 64-bit multiply, compare and select per use
  static_assert(14 + (51 + QP_BD_OFFSET) / 6 + (15 - BD - 2) <= 27, "qbits");
  auto level_double = [&](int blk) -> int32_t { return (int32_t)__umul24((unsigned)abs((int)src[blk]), (unsigned)qcoef); };
  auto cost0_of = [&](int blk) -> double { const double d = (double)level_double(blk); return d * d * err_scale; };
#if defined(HEVCDL_MICRO_T)
  unsigned long long mt_[11] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  unsigned long long pt_ = __builtin_readcyclecounter();
#elif defined(HEVCDL_KERNEL_PROF)
  unsigned long long pt_ = __builtin_readcyclecounter();
#endif
  // ---- phase A: rounded levels; per group (bit = its index in scan order): any level, any level > 1, any level > 2 ----
  unsigned long long cg_nz = 0, cg_ge2 = 0, cg_ge3 = 0;
  int last_pos = -1;
  for (int base = 0; base < ncoef; base += 64) {
    const int sp = base + lane;
    int ma = 0;
    if (sp < ncoef) {
      const int blk = scan[sp];
      const int32_t ld = level_double(blk);
      uint32_t m = ((uint32_t)ld + (1u << (qbits - 1))) >> qbits;          // (< 2^30 + 2^26: no carry out of 32 bits)
      if (m > 32767u) m = 32767u;
      dst[blk] = (int16_t)m; ma = (int)m;
    }
    const unsigned long long b1 = __ballot(ma > 0), b2 = __ballot(ma > 1), b3 = __ballot(ma > 2);
    if (b1) last_pos = base + 63 - __clzll((long long)b1);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int g = (base >> 4) + r;
      if ((b1 >> (16 * r)) & 0xffffull) cg_nz |= 1ull << g;
      if ((b2 >> (16 * r)) & 0xffffull) cg_ge2 |= 1ull << g;
      if ((b3 >> (16 * r)) & 0xffffull) cg_ge3 |= 1ull << g;
    }
  }
  wsync();
  if (last_pos < 0) return 0;
  __builtin_assume(last_pos < ncoef);                       // (with NFIX == 4: one group -- the batch logic, the group test and the walk over the groups fold)
  RDOQ_MARK(18);
  RDOQ_STOP(18);
  const int sig_off = CTX_SIG + (ch ? 28 : 0), cg_off = CTX_SIG_CG + (ch ? 2 : 0);
  int cg_b00, cg_b01, cg_b10, cg_b11, cbf_bits0, cbf_bits1;  // rates of the significant-group flag [context][value] and of the cbf flag [value]
  { // rate tables: ONE pair of dependent LDS reads for everything the call prices with the (frozen) contexts -- significance, greater-1 / greater-2, the
    // significant-group flags and the cbf flag (lanes 48..53), and the last-position prefix tables (TEncSbac.cpp:1910-1930; a second role of lanes 0..31) --
    // instead of a pair here, a pair for the group flags, and two more at the head of the last-position search
    const int set0 = ch ? 4 : 0;
    int off, shift; last_ctx_params(ch, n, off, shift);
    const int ng = tb().t_group_idx[n - 1];
    const int kk = lane & 15, isy = (lane >> 4) & 1;
    const int lctx = (isy ? CTX_LAST_Y : CTX_LAST_X) + (ch ? 15 : 0) + off + (kk >> shift);
    const int lst = cab->ctx[lane < 32 ? lctx : CTX_LAST_X];                                       // (lanes 32..63: any valid context, unused)
    int role_ctx = sig_off + (lane < 28 ? lane : 0);
    if (lane >= 28 && lane < 44) role_ctx = CTX_ONE + 4 * set0 + (lane - 28);
    else if (lane >= 44 && lane < 48) role_ctx = CTX_ABS + set0 + (lane - 44);
    else if (lane >= 48 && lane < 52) role_ctx = cg_off + ((lane - 48) >> 1);
    else if (lane >= 52) role_ctx = CTX_QT_CBF + (ch ? 5 : 0) + cbf_ctx;
    const int st = cab->ctx[role_ctx];
    const int e_0 = tb().t_ebits[st], e_1 = tb().t_ebits[st ^ 1], l_1 = tb().t_ebits[lst ^ 1], l_0 = tb().t_ebits[lst];
    if (lane < 28) { s.rq_sig[lane][0] = e_0; s.rq_sig[lane][1] = e_1; }
    else if (lane < 44) { const int e = lane - 28; s.rq_g1[e >> 2][e & 3][0] = e_0; s.rq_g1[e >> 2][e & 3][1] = e_1; }
    else if (lane < 48) { const int e = lane - 44; s.rq_g2[e][0] = e_0; s.rq_g2[e][1] = e_1; }
    // lanes 48 / 50: the group flag under context 0 / 1, lane 52: the cbf flag (ctx_bits(c, x, bin) = ebits[state ^ bin]): wave-uniform, read across the lanes (no LDS round trip)
    cg_b00 = __builtin_amdgcn_readlane(e_0, 48); cg_b01 = __builtin_amdgcn_readlane(e_1, 48); cg_b10 = __builtin_amdgcn_readlane(e_0, 50); cg_b11 = __builtin_amdgcn_readlane(e_1, 50);
    cbf_bits0 = __builtin_amdgcn_readlane(e_0, 52); cbf_bits1 = __builtin_amdgcn_readlane(e_1, 52);
    { // last position: entry kk = bits of kk ones (+ the terminating zero for kk < ng): prefix sum on the DPP crossbar; lanes 0..15: X, lanes 16..31: Y
      const int b1 = (kk < ng) ? l_1 : 0, b0 = (kk < ng) ? l_0 : 0;
      int inc = b1;
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
      if (lane < 32 && kk <= ng) s.last_bits[isy][kk] = inc - b1 + b0;
    }
    wsync();
  }
  double block_uncoded = 0;
  // zero-level costs above the last position, summed in scan order from the top: 64 positions per round, costs recomputed lane-parallel (a zero coefficient costs
  // exactly 0.0: rounds without any are skipped), the ordered sum itself runs out of LDS
  for (int top = ncoef - 1; top > last_pos; top -= 64) {
    const int sp = top - lane;
    const double c0 = (sp > last_pos) ? cost0_of(scan[sp]) : 0.0;
    const unsigned long long nzc = __ballot(c0 != 0.0);
    if (!nzc) continue;
    if (__popcll(nzc) <= 8) { // a handful: read across the lanes in the same order (lane 0 = the highest position), no transposition through LDS
      for (unsigned long long m = nzc; m; m &= m - 1ull) block_uncoded += rl_d(c0, __ffsll((long long)m) - 1);
      continue;
    }
    wsync();
    s.zb[0][lane] = c0;
    wsync();
    const int nb = (((top - last_pos) < 64 ? (top - last_pos) : 64) + 15) >> 4;     // lanes from top - last_pos on hold +0.0, which leaves the sum as it is
    for (int h = 0; h < nb; h++) {
      double v[16];
#pragma unroll
      for (int t = 0; t < 16; t++) v[t] = s.zb[0][h * 16 + t];
#pragma unroll
      for (int t = 0; t < 16; t++) block_uncoded += v[t];
    }
  }
  double acc = lane < 2 ? block_uncoded : 0.0;          // lanes 0 / 1 / 2: block_uncoded, base_cost, the current group's significance cost (phase B)
  RDOQ_MARK(19);
  RDOQ_STOP(19);
  const int cg_last = last_pos >> 4, wg = cp.wg, lwg = log2n - 2;
  // rates of the significant-group flag, by context (0 / 1) and value
  const double cgr00 = lambda * (double)cg_b00, cgr01 = lambda * (double)cg_b01;
  const double cgr10 = lambda * (double)cg_b10, cgr11 = lambda * (double)cg_b11;
  unsigned long long cgf_mask = 0;                     // significant-group flags after RDOQ, bit = raster index of the group
  unsigned long long cgf_scan = 0;                     // the same flags, bit = index of the group in scan order (the last-position search walks them)
  auto cgf_at = [&](int gx, int gy) -> int { return (int)((cgf_mask >> (gy * wg + gx)) & 1ull); };
  int carry = 0;                                       // the previous group in scan order ended with c1 == 0
  bool cc_needed = true;                               // no group that keeps a level > 1 yet: the last-position search will walk the groups still to come
  const unsigned long long dstart = wg == 8 ? 0xA44208101020844Bull : (wg == 4 ? 0xA44Bull : 0xBull);
  // ---- phase B ----
  int cgpos = cg_last;
  while (cgpos >= 0) {
    // --- the batch: groups cgpos, cgpos - 1, ... (row r of 16 lanes <-> group cgpos - r) ---
    // (the groups' scan order goes anti-diagonal by anti-diagonal in all three scan types: bit i of dstart = group i is the first of its diagonal)
    int R = 1, qnz0 = ((cg_nz >> cgpos) & 1ull) ? cgpos : -1;                                // the first row with a rounded level: its diagonal binds the batch
    while (R < RQ_ROWS && cgpos - R >= 0) {
      const int q = cgpos - R;
      if (((cg_ge2 >> (q + 1)) & 1ull) && !((cg_ge3 >> (q + 1)) & 1ull)) break;            // the bit carried into q is not known beforehand
      const int qnz = (int)((cg_nz >> q) & 1ull);
      if (qnz0 >= 0) { if ((dstart >> (q + 1)) & ((1ull << (qnz0 - q)) - 1ull)) break; }     // another diagonal: might be a neighbour of a row still to be decided
      else if (qnz) qnz0 = q;
      R++;
    }
    const int any_nz = (int)(((cg_nz >> (cgpos - R + 1)) & ((1ull << R) - 1ull)) != 0ull);
    // --- per position: everything that does not depend on the state machine ---
    const int row = lane >> 4, j = lane & 15;
    const bool rv = row < R;
    const int q = rv ? cgpos - row : cgpos;
    const int cgblk = scan_cg[q], gy = cgblk >> lwg, gx = cgblk & (wg - 1);
    const int pin = scan.in[j], px = (gx << 2) + (pin & 3), py = (gy << 2) + (pin >> 2);
    const int sp_j = q * 16 + j, blk_j = (py << log2n) + px;
    const int pat = ((gx < wg - 1) ? cgf_at(gx + 1, gy) : 0) + (((gy < wg - 1) ? cgf_at(gx, gy + 1) : 0) << 1);      // TComTrQuant.cpp:2672-2705
    const int start_pin = (q == cg_last) ? (last_pos & 15) : 15;
    const bool valid = rv && j <= start_pin;
    const int32_t ld_j = level_double(blk_j);
    const int ma_j = valid ? (int)dst[blk_j] : 0;
    const double c0_j = (double)ld_j * (double)ld_j * err_scale;
    const bool is_last = sp_j == last_pos;
    const int sc_j = is_last ? 0 : sig_ctx_inc_xy(cp, pat, px, py);
    const int b0_j = is_last ? 0 : s.rq_sig[sc_j][0], b1_j = is_last ? 0 : s.rq_sig[sc_j][1];
    const double cs0_j = lambda * (double)b0_j, cs1_j = lambda * (double)b1_j;
    RDOQ_MARK(4);
    RDOQ_SKIP(4)
    int lvl_j = 0, ru_j = 0, rd_j = 0;
    double cc_j = c0_j + cs0_j, cs_j = cs0_j;
    unsigned long long g1m = 0;
    if (any_nz) {
      // greater-1 / greater-2 rates of the row's context set (m_greaterOneBits[4 * ctxSet + c1][bin], m_levelAbsBits[ctxSet][bin])
      const int ctx_set = ctx_set_index(ch, q, row == 0 ? carry : (int)((cg_ge3 >> (q + 1)) & 1ull));
      const int cset = ctx_set - (ch ? 4 : 0);
      int g1r0[4], g1r1[4];
#pragma unroll
      for (int t = 0; t < 4; t++) { g1r0[t] = s.rq_g1[cset][t][0]; g1r1[t] = s.rq_g1[cset][t][1]; }
      const int g2r0 = s.rq_g2[cset][0], g2r1 = s.rq_g2[cset][1];
      const unsigned above = (0xfffeu << j) & 0xffffu;                                   // the positions of the row this one comes after
      const bool vis = valid && ma_j > 0;
      int c1s = 1, c1idx = 0, c2idx = 0, gr = 0, rate_a = 0, rate_b = 0;
      // xGetICRate TComTrQuant.cpp:2881-2955 under this position's state
      auto ic_rate = [&](int al, int r1_0, int r1_1) -> int { // branch-free: every lane evaluates both forms of the remainder code
        const bool c1ok = c1idx < 8, c2ok = c2idx < 1;
        const int base = c1ok ? (2 + (c2ok ? 1 : 0)) : 1;
        const uint32_t symbol = (uint32_t)(al - base), thr = 3u << gr;
        // escape: the prefix grows while the remainder reaches 1 << length (xWriteCoefRemainExGolomb's loop) <=> length = floor(log2(symbol - thr + (1 << gr)))
        const int elen = 31 - __clz((int)(symbol - thr + (1u << gr)));
        const uint32_t plen = symbol < thr ? (symbol >> gr) + 1u + (uint32_t)gr : (uint32_t)(3 + elen + 1 - gr + elen);     // (meaningless, and unused, below `base`)
        const int hi = (int)(32768u + (plen << 15)) + (c1ok ? r1_1 + (c2ok ? g2r1 : 0) : 0);
        const int lo = al == 1 ? 32768 + r1_0 : (al == 2 ? 32768 + r1_1 + g2r0 : 0);
        return al >= base ? hi : lo;
      };
      int r1_0 = g1r0[1], r1_1 = g1r1[1];
      lvl_j = vis ? ma_j : 0;
      for (;;) {
        // the state each position sees, from the levels as they stand
        const unsigned long long nzm = __ballot(lvl_j > 0), e1m = __ballot(lvl_j == 1);
        g1m = __ballot(lvl_j > 1);
        const unsigned f_nz = (unsigned)(nzm >> (16 * row)) & above, f_g1 = (unsigned)(g1m >> (16 * row)) & above, f_e1 = (unsigned)(e1m >> (16 * row)) & above;
        c1idx = __popc(f_nz); c2idx = __popc(f_g1);
        { const int e = 1 + __popc(f_e1); c1s = f_g1 ? 0 : (e < 3 ? e : 3); }
        gr = 0;
        if (__ballot(lvl_j > 3)) { // the Rice parameter moves only at a level above 3 << parameter that is coded with an escape (:2360-2366): from 0 at
          // the first such level above 3, then at the next one above 6, above 12, above 24 -- four row masks and four leading-bit searches per lane
          const int base = (c1idx < 8) ? (2 + (c2idx < 1)) : 1;
          const int e_j = (lvl_j >= base) ? lvl_j : 0;
          const unsigned long long m3 = __ballot(e_j > 3), m6 = __ballot(e_j > 6), m12 = __ballot(e_j > 12), m24 = __ballot(e_j > 24);
          const unsigned r3 = (unsigned)(m3 >> (16 * row)) & 0xffffu, r6 = (unsigned)(m6 >> (16 * row)) & 0xffffu, r12 = (unsigned)(m12 >> (16 * row)) & 0xffffu, r24 = (unsigned)(m24 >> (16 * row)) & 0xffffu;
          const int p1 = 31 - __clz((int)r3);
          const int p2 = p1 > 0 ? 31 - __clz((int)(r6 & ((1u << p1) - 1u))) : -1;
          const int p3 = p2 > 0 ? 31 - __clz((int)(r12 & ((1u << p2) - 1u))) : -1;
          const int p4 = p3 > 0 ? 31 - __clz((int)(r24 & ((1u << p3) - 1u))) : -1;
          gr = (p1 > j) + (p2 > j) + (p3 > j) + (p4 > j);
        }
        r1_0 = c1s == 0 ? g1r0[0] : (c1s == 1 ? g1r0[1] : (c1s == 2 ? g1r0[2] : g1r0[3]));
        r1_1 = c1s == 0 ? g1r1[0] : (c1s == 1 ? g1r1[1] : (c1s == 2 ? g1r1[2] : g1r1[3]));
        // xGetCodedLevel TComTrQuant.cpp:2812-2879: zero (only below 3), the rounded level, one below it
        double best = MAX_DOUBLE, cost_s = 0.0; int best_lvl = 0;
        if (!is_last && ma_j < 3) { cost_s = cs0_j; best = c0_j + cs0_j; }
        const double cur_sig = is_last ? 0.0 : cs1_j;
        {
          const double err = (double)(ld_j - (int32_t)((uint32_t)ma_j << qbits));
          rate_a = ic_rate(ma_j, r1_0, r1_1);
          double cur = err * err * err_scale + lambda * (double)rate_a;
          cur += cur_sig;
          if (cur < best) { best_lvl = ma_j; best = cur; cost_s = cur_sig; }
        }
        if (ma_j > 1) {
          const int al = ma_j - 1;
          const double err = (double)(ld_j - (int32_t)((uint32_t)al << qbits));
          rate_b = ic_rate(al, r1_0, r1_1);
          double cur = err * err * err_scale + lambda * (double)rate_b;
          cur += cur_sig;
          if (cur < best) { best_lvl = al; best = cur; cost_s = cur_sig; }
        }
        const int nl = vis ? best_lvl : 0;
        if (vis) { cc_j = best; cs_j = cost_s; }
        const unsigned long long chg = __ballot(nl != lvl_j);
        lvl_j = nl;
        if (!chg) break;
      }
      // rate deltas of level +-1 for sign hiding (:2340-2352); a zero level: the greater-1 rate it would start with
      if (lvl_j > 0) {
        const bool top = lvl_j == ma_j;
        const int now = top ? rate_a : rate_b;
        const int other = ic_rate(top ? lvl_j + 1 : lvl_j - 1, r1_0, r1_1);
        ru_j = (top ? other : rate_a) - now;
        rd_j = (lvl_j == 1 ? 0 : (top ? rate_b : other)) - now;
      } else ru_j = r1_0;
      RDOQ_MARK(5);
      if (valid) {
        dst[blk_j] = (int16_t)lvl_j;
        // Who reads the per-position outputs again: the last-position search walks the groups that keep their flag -- a group with a level, or group 0 -- and sign
        // hiding only touches a group whose first and last level are at least four positions apart (levels only disappear after this point, so the distance
        // can only shrink): the other rows store nothing
        const unsigned nzr = (unsigned)(__ballot(lvl_j > 0) >> (16 * row)) & 0xffffu;
        if (cc_needed && (nzr || q == 0)) { q_cost_st(Q_COEFF, sp_j, cc_j); q_cost_st(Q_SIG, sp_j, cs_j); }   // read again by the last-position search only
        if (nzr && (31 - __clz((int)nzr)) - (__ffs((int)nzr) - 1) >= 4) {
          q_rate_st(Q_SIGDELTA, blk_j, b1_j - b0_j);                   // 0 at the last position
          q_rate_st(Q_DELTAU, blk_j, (int32_t)((ld_j - (int32_t)((uint32_t)lvl_j << qbits)) >> (qbits - 8)));
          q_rate_st(Q_UP, blk_j, ru_j);
          q_rate_st(Q_DOWN, blk_j, rd_j);
        }
      }
    }
    RDOQ_SKIP(5)
    // --- the ordered sums, group by group (each in scan order, pin = 15..0, as the reference accumulates them).  Three of them ride in ONE chain of 16 additions on
    // lanes 0..2 of `acc` (addends transposed through LDS):   0 block_uncoded += c0      1 base_cost += cost_c      2 sig_cost += cost_s (from zero, per group)
    // and stay in their lanes from group to group and from batch to batch: the group's test runs on lane 1 (lane 2's sum comes over the DPP crossbar), nothing is
    // broadcast.  A position that does not contribute adds +0.0, which leaves a sum unchanged.  The two sums over the levels only (coded, uncoded) are wave-uniform.
    const bool nz_j = valid && lvl_j != 0;
    const unsigned long long nzfin = __ballot(nz_j);
    const double d_j = cc_j - cs_j;
    if (rv) {
      const int o = row * 16 + j;
      s.chainb[0][o] = valid ? c0_j : 0.0; s.chainb[1][o] = valid ? cc_j : 0.0; s.chainb[2][o] = valid ? cs_j : 0.0;
    }
    wsync();
    LDS const double *cbl = &s.chainb[lane < 2 ? lane : 2][0];
    for (int r = 0; r < R; r++) {
      const int qq = cgpos - r;
      const int cb = __builtin_amdgcn_readlane(cgblk, 16 * r), ggy = cb >> lwg, ggx = cb & (wg - 1);
      const unsigned nzrow = (unsigned)(nzfin >> (16 * r)) & 0xffffu;
      { // (the addends of the next group are NOT fetched ahead: the sixteen register pairs that takes push code_tu_block over its register budget -- twenty more
        //  registers saved and restored per call, 2 MB of scratch traffic per CTU -- for no measurable gain)
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = cbl[r * 16 + t];
#pragma unroll
        for (int t = 15; t >= 0; t--) acc += v[t];
      }
      const int st_nnz_before0 = __popc(nzrow & 0xfffeu), cg_nonzero = nzrow != 0;
      int flag = cg_nonzero;
      double nb = acc;                                   // lane 1: base_cost behind this group
      if (qq) {
        const int csx = (((ggx < wg - 1) ? cgf_at(ggx + 1, ggy) : 0) + ((ggy < wg - 1) ? cgf_at(ggx, ggy + 1) : 0)) != 0;    // TComTrQuant.cpp:3023-3049
        const double r0 = csx ? cgr10 : cgr00, r1 = csx ? cgr11 : cgr01;
        // lane l takes lane l + 1's sum: lane 1 sees the group's significance cost
        const double sig = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(acc), 0x101, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(acc), 0x101, 0xf, 0xf, false));
        if (!cg_nonzero) {
          nb = acc + (r0 - sig);
          cgs_set |= 1ull << qq; if (csx) cgs_ctx |= 1ull << qq;
        } else if (qq < cg_last) {
          // sums 3 and 4 run over the positions with a level only (adding +0.0 for the others would change nothing): a handful, highest position first
          double st_coded = 0.0, st_uncoded = 0.0;
          for (unsigned m = nzrow; m; ) {
            const int t = 31 - __clz((int)m); m &= ~(1u << t);
            st_coded += rl_d(d_j, 16 * r + t); st_uncoded += rl_d(c0_j, 16 * r + t);
          }
          const double st_sig_cost0 = rl_d(cs_j, 16 * r);
          double b = acc, sg = sig;
          if (st_nnz_before0 == 0) { b -= st_sig_cost0; sg -= st_sig_cost0; }
          double zero_cost = b;
          b += r1; zero_cost += r0;
          zero_cost += st_uncoded; zero_cost -= st_coded; zero_cost -= sg;
          cgs_set |= 1ull << qq; if (csx) cgs_ctx |= 1ull << qq;
          const int take = (int)((__ballot(zero_cost < b) >> 1) & 1ull);                 // lane 1's verdict
          if (take) {
            nb = zero_cost; flag = 0;
            if (row == r && lvl_j) dst[blk_j] = 0;                       // (its per-position costs are never read: the last-position search skips a group without the flag)
          } else { nb = b; cgs_one |= 1ull << qq; }
        }
      } else flag = 1;
      acc = lane == 1 ? nb : (lane == 2 ? 0.0 : acc);
      if (flag) { cgf_mask |= 1ull << cb; cgf_scan |= 1ull << qq; if ((g1m >> (16 * r)) & 0xffffull) cc_needed = false; }   // the last-position search ends in this group (a level > 1 stays)
      RDOQ_MARK(34);
    }
    carry = (int)(((g1m >> (16 * (R - 1))) & 0xffffull) != 0ull);
    wsync();
    RDOQ_MARK(8);
    cgpos -= R;
  }
  block_uncoded = rl_d(acc, 0);
  double base_cost = rl_d(acc, 1);
  RDOQ_MARK(20);
  RDOQ_STOP(20);
  // ---- phase C: last position, TComTrQuant.cpp:2440-2528.  Per CG the 16 positions' costs are fetched
  // lane-parallel, the walk itself is wave-uniform (readlane) and usually ends inside the first group ----
  int best_last_p1 = 0;
  {
    double best_cost;
    best_cost = block_uncoded + lambda * (double)cbf_bits0;
    base_cost += lambda * (double)cbf_bits1;
    LDS int *last_x_bits = s.last_bits[0], *last_y_bits = s.last_bits[1];      // (filled with the rate tables at the top)
    int found_last = 0;
    // the per-position costs of the larger TUs come back from the wave's HBM workspace: those of the NEXT group that keeps its flag are asked for while this one is
    // walked (a round trip of ~1.5 k cycles per group otherwise, a quarter of the routine's time)
    int pg = -1; double pcc = 0.0, pcs = 0.0;
    auto prefetch = [&](int below) {
      const unsigned long long m = below > 0 ? (cgf_scan & ((1ull << below) - 1ull)) : 0ull;
      if (!qlds && m) { pg = 63 - __clzll((long long)m); const int sp = pg * 16 + (lane & 15); pcc = gq_cost[Q_COEFF * 1024 + sp]; pcs = gq_cost[Q_SIG * 1024 + sp]; }
      else pg = -1;
    };
    prefetch(cg_last + 1);
    for (int cgp = cg_last; cgp >= 0 && !found_last; cgp--) {
      if ((cgs_set >> cgp) & 1ull) base_cost -= ((cgs_ctx >> cgp) & 1ull) ? (((cgs_one >> cgp) & 1ull) ? cgr11 : cgr10) : (((cgs_one >> cgp) & 1ull) ? cgr01 : cgr00);
      if (!((cgf_scan >> cgp) & 1ull)) continue;
      const int j = lane & 15, sp_j = cgp * 16 + j, blk_j = scan[sp_j];
      const int lv_j = dst[blk_j];
      double cc_j, cs_j;
      if (pg == cgp) { cc_j = pcc; cs_j = pcs; } else { cc_j = q_cost_ld(Q_COEFF, sp_j); cs_j = q_cost_ld(Q_SIG, sp_j); }
      prefetch(cgp);
      const double c0_j = cost0_of(blk_j);
      int py = blk_j >> log2n, px = blk_j - (py << log2n);
      if (cp.scan_type == SCAN_VER) { const int t = px; px = py; py = t; }
      const int gx2 = tb().t_group_idx[px], gy2 = tb().t_group_idx[py];
      double lc = (double)(last_x_bits[gx2] + last_y_bits[gy2]);
      if (gx2 > 3) lc += 32768.0 * (double)((gx2 - 2) >> 1);
      if (gy2 > 3) lc += 32768.0 * (double)((gy2 - 2) >> 1);
      const double cl_j = lambda * lc;
      // the walk over the group (TComTrQuant.cpp:2478-2527) as one ordered chain: position pin subtracts its coded cost and
      // adds back its zero-level cost when it holds a level, subtracts its significance cost otherwise; the value of the
      // chain BEFORE a position is what its candidate "last position" is priced with.  Chain uniform in registers, prices
      // lane-parallel, then only the positions with a level are compared, highest scan position first.
      const int start_pin = (cgp == cg_last) ? (last_pos & 15) : 15;
      const bool in_j = j <= start_pin;
      const double a1_j = in_j ? (lv_j ? -cc_j : -cs_j) : 0.0, a2_j = (in_j && lv_j) ? c0_j : 0.0;
      const unsigned gt1 = (unsigned)(__ballot(lane < 16 && in_j && lv_j > 1) & 0xffffull);
      const int stop_pin = gt1 ? 31 - __clz((int)gt1) : 0;
      double mine = base_cost, acc = base_cost;
      // the chain matters from the first position of the group that is in the walk (start_pin: the ones above add +0.0) down to the position the walk ends at (a level
      // above 1: nothing below it is priced, and the sum behind the group is not used): when that is a short stretch its addends are read across the lanes
      if (start_pin - stop_pin < 6) {
        for (int pin = start_pin; pin >= stop_pin; pin--) { mine = (j == pin) ? acc : mine; acc = (acc + rl_d(a1_j, pin)) + rl_d(a2_j, pin); }
      } else {
        wsync();
        if (lane < 16) { s.zb[0][j] = a1_j; s.zb[1][j] = a2_j; }
        wsync();
        double v1[16], v2[16];
#pragma unroll
        for (int t = 0; t < 16; t++) { v1[t] = s.zb[0][15 - t]; v2[t] = s.zb[1][15 - t]; }
#pragma unroll
        for (int t = 0; t < 16; t++) { mine = (j == 15 - t) ? acc : mine; acc = (acc + v1[t]) + v2[t]; }
      }
      const double total_j = (mine + cl_j) - cs_j;
      unsigned cand = (unsigned)(__ballot(lane < 16 && in_j && lv_j != 0 && j >= stop_pin) & 0xffffull);
      while (cand) {
        const int pin = 31 - __clz((int)cand);
        cand &= ~(1u << pin);
        const double total = rl_d(total_j, pin);
        if (total < best_cost) { best_last_p1 = cgp * 16 + pin + 1; best_cost = total; }
      }
      if (gt1) found_last = 1;
      base_cost = acc;
    }
  }
  RDOQ_MARK(21);
  RDOQ_STOP(21);
  // signs, absolute sum, uncoded tail (lane-parallel; integer sum is exact) -- and, in the same walk, which groups sign data hiding (TComTrQuant.cpp:2530-2660) has to
  // visit at all: a group whose first and last level are four positions or more apart and whose level parity disagrees with the sign of its first level.  A visit only
  // changes levels of its own group, so the tests of all groups can be made beforehand, four groups (one per row of 16 lanes) at a time.
  uint32_t abs_sum = 0;
  unsigned long long sbh_need = 0;                         // bit = group in scan order
  int top_group = -1;                                      // the first group from the top that holds a level (the reference's lastCG == 1)
  for (int base = 0; base <= last_pos; base += 64) {
    const int sp = base + lane;
    int lv = 0;
    if (sp <= last_pos) {
      const int blk = scan[sp];
      if (sp < best_last_p1) { const int a = dst[blk]; abs_sum += (uint32_t)a; lv = src[blk] < 0 ? -a : a; }
      dst[blk] = (int16_t)lv;
    }
    const unsigned long long nzb = __ballot(lv != 0);
    if (nzb) {
      const unsigned long long ngb = __ballot(lv < 0);
      const int row = lane >> 4;
      const unsigned rm = (unsigned)(nzb >> (16 * row)) & 0xffffu;
      const int first = rm ? __ffs((int)rm) - 1 : 0, last = rm ? 31 - __clz((int)rm) : 0;
      const int sum = row_sum_i(lv);
      const unsigned signbit = (unsigned)(ngb >> (16 * row + first)) & 1u;
      const unsigned long long ndb = __ballot((lane & 15) == 0 && last - first >= 4 && signbit != ((unsigned)sum & 1u));
#pragma unroll
      for (int r = 0; r < 4; r++) if ((ndb >> (16 * r)) & 1ull) sbh_need |= 1ull << ((base >> 4) + r);
      top_group = (base >> 4) + ((63 - __clzll((long long)nzb)) >> 4);
    }
  }
  abs_sum = (uint32_t)wave_sum_i((int)abs_sum);
  wsync();
  if (abs_sum >= 2 && sign_hide) { // (with sign_data_hiding_enabled_flag) lanes 0..15 own the positions of the group being visited
    const long long rd_factor = k.sbh[ch];
    const long long I64MAX = 0x7fffffffffffffffll;
    while (sbh_need) {
      const int subset = 63 - __clzll((long long)sbh_need);
      sbh_need &= ~(1ull << subset);
      const int last_cg = subset == top_group ? 1 : 0;
      const int sub_pos = subset << 4, j = lane & 15, blk_j = scan[sub_pos + j];
      const int lv_j = dst[blk_j];
      const unsigned nzmask = (unsigned)(__ballot(lv_j != 0) & 0xffffull);
      const int last_nz = 31 - __clz((int)nzmask), first_nz = __ffs((int)nzmask) - 1;
      const int lv_first = __builtin_amdgcn_readlane(lv_j, first_nz);
      const uint32_t signbit = lv_first > 0 ? 0 : 1;
      {
        {
          long long cur_cost = I64MAX; int cur_change = 0;
          const int nmax = (last_cg == 1) ? last_nz : 15;
          if (j <= nmax) {
            const int32_t du_j = q_rate_ld(Q_DELTAU, blk_j), riu_j = q_rate_ld(Q_UP, blk_j), rid_j = q_rate_ld(Q_DOWN, blk_j), srd_j = q_rate_ld(Q_SIGDELTA, blk_j);
            if (lv_j != 0) {
              const long long up = rd_factor * (-(long long)du_j) + riu_j;
              long long down = rd_factor * ((long long)du_j) + rid_j - ((abs(lv_j) == 1) ? srd_j : 0);
              if (last_cg == 1 && last_nz == j && abs(lv_j) == 1) down -= (4 << 15);
              if (up < down) { cur_cost = up; cur_change = 1; }
              else { cur_change = -1; cur_cost = (j == first_nz && abs(lv_j) == 1) ? I64MAX : down; }
            } else {
              cur_cost = rd_factor * (-(long long)abs(du_j)) + (1 << 15) + riu_j + srd_j;
              cur_change = 1;
              if (j < first_nz) { const uint32_t ts = src[blk_j] >= 0 ? 0 : 1; if (ts != signbit) cur_cost = I64MAX; }
            }
          }
          // the reference scans n = nmax..0 and keeps the first strict minimum: smallest cost, ties -> largest n
          long long bc = cur_cost; int bn = (j <= nmax && cur_cost != I64MAX) ? j : -1;
          for (int m = 8; m >= 1; m >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)bc, m), hi = (unsigned)__shfl_xor((int)(unsigned)(bc >> 32), m);
            const long long oc = (long long)(((unsigned long long)hi << 32) | lo); const int on = __shfl_xor(bn, m);
            if (on >= 0 && (bn < 0 || oc < bc || (oc == bc && on > bn))) { bc = oc; bn = on; }
          }
          const int min_n = __builtin_amdgcn_readlane(bn, 0);
          if (min_n >= 0) {
            int final_change = __builtin_amdgcn_readlane(cur_change, min_n);
            const int lv_min = __builtin_amdgcn_readlane(lv_j, min_n);
            if (lv_min == 32767 || lv_min == -32768) final_change = -1;
            if (lane == min_n) { if (src[blk_j] >= 0) dst[blk_j] = (int16_t)(lv_j + final_change); else dst[blk_j] = (int16_t)(lv_j - final_change); }
          }
        }
      }
    }
  }
  wsync();
  RDOQ_MARK(22);
#ifdef HEVCDL_MICRO_T
  if (lane < 11) { unsigned long long a_ = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) if (lane == i) a_ = mt_[i];
    s.mt_acc[lane] += a_; }
#endif
  return (uint32_t)uni((int)abs_sum);
}

// The quantiser without RDOQ (cfg RDOQ 0, or RDOQTS 0 for a transform-skipped block): TComTrQuant::xQuant TComTrQuant.cpp:1169-1249 -- dead-zone rounding with the intra offset
// 171 / 512 -- and, with sign data hiding on, signBitHidingHDQ (:991-1113).  Whole wave, s->tc -> s->lvl, returns uiAbsSum (of the levels before the hiding pass, as the reference).
// Not the configuration the bench runs: kept out of line and simple.
DEV uint32_t plain_quant_wave(KR k, int c_, int n_, int dir_mode_)
{
  const int c = uni(c_), n = uni(n_), dir_mode = uni(dir_mode_);
  LSmem &s = lds();
  const int lane = lane_id(), log2n = ilog2(n), ncoef = n * n;
  const int qp = uni(c ? k.qp_c : k.qp) + QP_BD_OFFSET, per = qp / 6, rem = qp % 6;
  const int qbits = 14 + per + (15 - BD - log2n), qbits8 = qbits - 8;
  const uint32_t qcoef = (uint32_t)uni(c_quant_scales[rem]), add = 171u << (qbits - 9);       // (|coefficient| * scale < 2^30, qbits <= 27: 32 bits hold every value, see rdoq_wave)
  LDS const int16_t *src = s.tc; LDS int16_t *dst = s.lvl;
  uint32_t abs_sum = 0;
  for (int i = lane; i < ncoef; i += 64) {
    const int v = src[i];
    const uint32_t mag = (__umul24((unsigned)abs(v), qcoef) + add) >> qbits;
    abs_sum += mag;
    dst[i] = (int16_t)clip16(v < 0 ? -(int)mag : (int)mag);
  }
  abs_sum = (uint32_t)wave_sum_i((int)abs_sum);
  wsync();
  if ((tools_of(k) & (int)HEVCDL_TOOL_SIGN_HIDE) && abs_sum >= 2) {
    CParam cp; get_cparam(cp, c, n, dir_mode);
    const ScanFn scan = scan_of(s, cp.scan_type, log2n);
    bool seen = false;                                     // a group with a level has been met (the reference's lastCG: 1 for the first such group from the top)
    for (int subset = (ncoef - 1) >> 4; subset >= 0; subset--) {
      const int j = lane & 15, blk_j = scan[(subset << 4) + j];
      const int lv_j = lane < 16 ? (int)dst[blk_j] : 0, cf_j = src[blk_j];
      const unsigned nzmask = (unsigned)(__ballot(lv_j != 0) & 0xffffull);
      if (!nzmask) continue;
      const int last_cg = seen ? 0 : 1; seen = true;
      const int last_nz = 31 - __clz((int)nzmask), first_nz = __ffs((int)nzmask) - 1;
      if (last_nz - first_nz < 4) continue;
      const int sum = wave_sum_i(lv_j);
      const unsigned signbit = __builtin_amdgcn_readlane(lv_j, first_nz) > 0 ? 0u : 1u;
      if (signbit == ((unsigned)sum & 1u)) continue;
      const int nmax = last_cg ? last_nz : 15;
      // deltaU of the position, from its coefficient and the level it was given (nothing has changed it yet: a visit only touches its own group, once)
      const int du_j = (int)(__umul24((unsigned)abs(cf_j), qcoef) - ((unsigned)abs(lv_j) << qbits)) >> qbits8;
      int cost = 0x7ffffff, change = 1;                    // (0x7ffffff: "not a candidate"; the real costs are a few hundred at most)
      if (lane < 16 && j <= nmax) {
        if (lv_j != 0) {
          if (du_j > 0) cost = -du_j;
          else if (!(j == first_nz && abs(lv_j) == 1)) { cost = du_j; change = -1; }
        } else if (!(j < first_nz && (cf_j >= 0 ? 0u : 1u) != signbit)) cost = -du_j;
      }
      // the reference walks n = nmax .. 0 and keeps the first strict minimum: smallest cost, ties -> largest n
      const int key = cost < 0x7ffffff ? cost * 16 + (15 - j) : 0x7fffffff;
      const int best = -wave_max_i(-key);
      const int min_n = 15 - (best & 15);
      int final_change = __builtin_amdgcn_readlane(change, min_n);
      const int lv_min = __builtin_amdgcn_readlane(lv_j, min_n);
      if (lv_min == 32767 || lv_min == -32768) final_change = -1;
      if (lane == min_n) dst[blk_j] = (int16_t)(cf_j >= 0 ? lv_j + final_change : lv_j - final_change);
      wsync();
    }
  }
  wsync();
  return (uint32_t)uni((int)abs_sum);
}

DEV void dequant(KR k, int c_, int n_)
{
  PROF_T0();
  const int c = uni(c_), n = uni(n_); // s->lvl -> s->tc  (TComTrQuant.cpp:1308-1425, flat scaling)
  const int log2n = ilog2(n), qp = (c ? k.qp_c : k.qp) + QP_BD_OFFSET, per = qp / 6, rem = qp % 6;
  const int tshift = 15 - BD - log2n, rshift = 6 - (tshift + per), scale = c_inv_quant_scales[rem];
  for (int i = lane_id(); i < n * n; i += 64) {
    const int q = lds().lvl[i];                       // already inside the 16-bit clip range
    int v;
    if (rshift > 0) v = (q * scale + (1 << (rshift - 1))) >> rshift;
    else v = (int)((unsigned)(q * scale) << (-rshift));
    lds().tc[i] = (int16_t)clip16(v);
  }
  wsync();
  PROF_ADD(k, 7);
}

// ---------------------------------------------------------------------------------------------------
// residual syntax bit counting on lane 0 (TEncSbac.cpp:1115-1541); coefficients in s->lvl (TU raster)
// ---------------------------------------------------------------------------------------------------
// Whole-wave version: the coder state is pulled into registers for the duration of the call -- lane L holds contexts
// L, 64+L, 128+L and the matching slices of the rate / transition tables -- so that one bin is a handful of
// v_readlane / v_cndmask and scalar operations instead of five dependent LDS round trips.  Everything a bin
// needs from the coefficients (significance, context increments, magnitudes of a coefficient group) is computed
// lane-parallel first; control flow is wave-uniform.
// pre_: bins coded in front of the coefficients, on contexts below 19 (the CU / TU header flags of a bit count: part size, prev_intra_luma_pred, transform subdivision, cbf)
// -- up to four, packed six bits each from bit 0: context (5 bits) | value << 5, 63 = none; bits 24..27: bypass bins behind them; bit 28: the bit count starts here
// (the integer bits are dropped first: reset_bits); bit 29: the TU has coefficients (s.lvl) to count behind the flags.  PRE_COEF | PRE_NONE = coefficients only.
// The fractional bits the coefficient bins add go to s.cfrac_last (luma_cfrac 1), to s.cfrac_last_c (2) or are added to it (3).  Returns the coder's integer bits.
constexpr int PRE_NONE = 0xffffff, PRE_EP_SHIFT = 24, PRE_RESET = 1 << 28, PRE_COEF = 1 << 29;
template <int NFIX> DEV uint32_t code_coeff_wave_i(KR k, LCabac *c, int comp_, int n_, int dir_mode_, int tskip_flag_, int pre_, int luma_cfrac_)
{ // NFIX != 0: the TU size as a compile-time constant (as for rdoq_wave).  _i: the body, inlined where it is called (code_tu_block's own count); _n: the call form
  const int comp = uni(comp_), n = NFIX ? NFIX : uni(n_), dir_mode = uni(dir_mode_), tskip_flag = uni(tskip_flag_), pre = uni(pre_), luma_cfrac = uni(luma_cfrac_);
  LSmem &s = lds();
  const int lane = lane_id(), ch = comp ? 1 : 0;
  CParam cp; get_cparam(cp, comp, n, dir_mode);
  const int log2n = cp.log2, ncoef = n * n;
  LDS const int16_t *coef = s.lvl; const ScanFn scan = scan_of(s, cp.scan_type, log2n); LDS const uint8_t *scan_cg = scan_cg_of(s, cp.scan_type, log2n);
  LDS uint8_t *cgf = s.cgf;
  int scan_last = -1;
  if (pre & PRE_COEF) {
    cgf[lane] = 0;
    wsync();
    // last significant scan position and the significant-CG flags (TEncSbac.cpp:1170-1200)
    int my_last = -1;
    for (int sp = lane; sp < ncoef; sp += 64) if (coef[scan[sp]] != 0) { my_last = sp; cgf[scan_cg[sp >> 4]] = 1; }
    scan_last = wave_max_i(my_last);
    if (scan_last < 0 && (pre & PRE_NONE) == PRE_NONE && !(pre & PRE_RESET)) return uni((int)get_bits(c));      // never called for an empty TU (cbf checked by the caller)
    wsync();
  }
  // The coefficient contexts of one channel type fit TWO registers whose member is known where a bin is coded (no choice of register at run time):
  //   A = contexts [BA, BA + 64), BA = 19 (luma) / 21 (chroma): significant-group flags (19..20 / 21..22), significance (23..49 / 51..66), last x (67..81 / 82..84: a chroma TU is at most 16 wide)
  //   B = contexts [97, 160): last y (97..111 / 112..115), greater-1 (127..142 / 143..150), greater-2 (151..154 / 155..156), transform skip (157 / 158)
  constexpr int BB = 97;
  const int BA = ch ? 21 : 19;
  static_assert(CTX_SIG_CG == 19 && CTX_LAST_X + 15 == 82 && CTX_LAST_Y == 97 && NUM_CTX <= BB + 64 && CTX_LAST_X + 15 + 2 < 21 + 64 && CTX_LAST_X + 14 < 19 + 64, "context windows of code_coeff_wave");
  const int tools = tools_of(k);
  int cxa = c->ctx[BA + lane], cxb = c->ctx[BB + (lane < 63 ? lane : 62)];
  int cxh = c->ctx[lane < 19 ? lane : 18];                      // H = contexts [0, 19): the header flags (its own register: the windows do not overlap)
  const int eb0 = tb().t_ebits[lane], eb1 = tb().t_ebits[64 + lane];
  // next states by x = state ^ bin (the index the rate tables are read with) and bin: lane l holds, byte by byte, (bin 0, x = l), (bin 0, x = 64 + l), (bin 1, x = l), (bin 1, x = 64 + l);
  // the transition is the MPS one when bin == state & 1, i.e. when x is even, and the state is x ^ bin
  const int mps_t = (lane & 1) ^ 1;
  const int nx = (int)((unsigned)tb().t_next[mps_t][lane] | ((unsigned)tb().t_next[mps_t][64 + lane] << 8) | ((unsigned)tb().t_next[mps_t][lane ^ 1] << 16) | ((unsigned)tb().t_next[mps_t][(64 + lane) ^ 1] << 24));
  unsigned long long frac;
  { const unsigned long long f = c->frac; frac = ((unsigned long long)(unsigned)uni((int)(f >> 32)) << 32) | (unsigned)uni((int)f); }
  if (pre & PRE_RESET) frac &= 32767ull;
  auto step = [&](int &cx, int l, int b) { // TEncBinCABACCounter::encodeBin TEncBinCoderCABACCounter.cpp:90-105 on the context in lane l of cx; b = 0 / 1, wave-uniform
    // integer arithmetic only, no conditions: the compiler then keeps the whole bin on the scalar unit (a comparison in here came out as a lane mask, a v_cndmask to
    // turn it back into a number and two branches: ~35 instructions and six branches a bin before, ~20 and none now)
    const int st = __builtin_amdgcn_readlane(cx, l);
    const int x = st ^ b, xl = x & 63;
    const int e0 = __builtin_amdgcn_readlane(eb0, xl), e1 = __builtin_amdgcn_readlane(eb1, xl);
    frac += (unsigned)(e0 + ((x >> 6) & 1) * (e1 - e0));
    const int w = __builtin_amdgcn_readlane(nx, xl);
    const int nxt = (w >> (((x >> 6) + 2 * b) << 3)) & 0xff;
    // (this clang has no writelane builtin.  Both scalar operands are results of scalar instructions here -- the lane-select hazard of v_writelane is about SGPRs written
    //  by VALU instructions --; the s_nop covers it all the same: inline assembly is outside the compiler's hazard pass)
    // (one SGPR per VALU instruction on gfx9: the lane select goes through M0 -- which nothing else in this file uses (checked in the assembly); it is named as
    //  clobbered all the same, and the compiler's warning that it treats M0 as reserved is silenced for this statement)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(cx) : "s"(nxt), "s"(l) : "m0");
#pragma clang diagnostic pop
  };
  auto bin_a = [&](int ctx, int b) { step(cxa, ctx - BA, b); };
  auto bin_b = [&](int ctx, int b) { step(cxb, ctx - BB, b); };
  auto ep = [&](int cnt) { frac += 32768ull * (unsigned long long)cnt; };
#pragma unroll
  for (int q = 0; q < 4; q++) { const int e = (pre >> (6 * q)) & 63; if (e != 63) step(cxh, e & 31, e >> 5); }
  ep((pre >> PRE_EP_SHIFT) & 15);
  const unsigned long long frac_hdr = frac;
  if (scan_last >= 0) {
  if (n == 4 && (tools & (int)HEVCDL_TOOL_TSKIP)) bin_b(CTX_TSKIP + ch, tskip_flag);                // codeTransformSkipFlags :997-1032 (with transform_skip_enabled_flag)
  { // codeLastSignificantXY TEncSbac.cpp:1051-1113
    const int pos_last = uni(scan[scan_last]);
    int py = pos_last >> log2n, px = pos_last - (py << log2n);
    if (cp.scan_type == SCAN_VER) { const int t = px; px = py; py = t; }
    const int gx = uni(tb().t_group_idx[px]), gy = uni(tb().t_group_idx[py]), gmax = uni(tb().t_group_idx[n - 1]); int off, shift, kk;
    last_ctx_params(ch, n, off, shift);
    const int bx = CTX_LAST_X + (ch ? 15 : 0) + off, by = CTX_LAST_Y + (ch ? 15 : 0) + off;
    for (kk = 0; kk < gx; kk++) bin_a(bx + (kk >> shift), 1);
    if (gx < gmax) bin_a(bx + (kk >> shift), 0);
    for (kk = 0; kk < gy; kk++) bin_b(by + (kk >> shift), 1);
    if (gy < gmax) bin_b(by + (kk >> shift), 0);
    if (gx > 3) ep((gx - 2) >> 1);
    if (gy > 3) ep((gy - 2) >> 1);
  }
  const int cg_off = CTX_SIG_CG + (ch ? 2 : 0), sig_off = CTX_SIG + (ch ? 28 : 0);
  const int last_set = scan_last >> 4;
  int c1 = 1;
  for (int subset = last_set; subset >= 0; subset--) {
    const int sub_pos = subset << 4;
    const int cgblk = uni(scan_cg[subset]), gy = cgblk >> (log2n - 2), gx = cgblk - (gy << (log2n - 2));
    int cg_sig;
    if (subset == last_set || subset == 0) { wsync(); if (lane == 0) cgf[cgblk] = 1; wsync(); cg_sig = 1; }
    else { cg_sig = uni((int)cgf[cgblk]) & 1; bin_a(cg_off + uni(sig_cg_ctx(cgf, gx, gy, cp.wg)), cg_sig); }
    if (!cg_sig) continue;
    // lane-parallel: the 16 positions of the group
    const int j = lane & 15, sp_j = sub_pos + j, blk_j = scan[sp_j];
    const int cf_j = (sp_j <= scan_last) ? (int)coef[blk_j] : 0, abs_j = abs(cf_j);
    const unsigned sigmask = (unsigned)(__ballot(lane < 16 && cf_j != 0) & 0xffffull);
    const int pat = uni(pattern_sig_ctx(cgf, gx, gy, cp.wg));
    const int sigctx_j = sig_off + sig_ctx_inc(cp, scan, pat, sp_j);
    int num_nz = 0, jstart = 15;
    if (subset == last_set) { num_nz = 1; jstart = (scan_last & 15) - 1; }
    for (int jj = jstart; jj >= 0; jj--) {
      const int sig = (sigmask >> jj) & 1;
      if (jj > 0 || subset == 0 || num_nz) bin_a(__builtin_amdgcn_readlane(sigctx_j, jj), sig);
      num_nz += sig;
    }
    if (num_nz > 0) {
      const int last_nz = 31 - __clz((int)sigmask), first_nz = __ffs((int)sigmask) - 1;
      const int sign_hidden = (last_nz - first_nz >= 4) && (tools & (int)HEVCDL_TOOL_SIGN_HIDE);
      const int cset = ctx_set_index(ch, subset, c1 == 0);
      c1 = 1;
      int escape = 0, abs_c2 = -1; unsigned m = sigmask;
      for (int i = 0; i < 8 && m; i++) { // greater-1 flags of the first 8 levels, highest scan position first
        const int p = 31 - __clz((int)m); m &= ~(1u << p);
        const int av = __builtin_amdgcn_readlane(abs_j, p), sym = min(av - 1, 1);      // av > 1 as a number (av >= 1 here): s_min, not a comparison (see step)
        bin_b(CTX_ONE + 4 * cset + c1, sym);
        if (sym) { c1 = 0; if (abs_c2 < 0) abs_c2 = av; else escape = 1; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && abs_c2 >= 0) { const int sym = min(abs_c2 - 2, 1) /* abs_c2 > 2; abs_c2 >= 2 here */; bin_b(CTX_ABS + cset, sym); if (sym) escape = 1; }
      escape = escape || (num_nz > 8);
      ep(sign_hidden ? num_nz - 1 : num_nz);
      if (escape) {
        int first_coeff2 = 1, go_rice = 0, i = 0;
        for (m = sigmask; m; i++) {
          const int p = 31 - __clz((int)m); m &= ~(1u << p);
          const int av = __builtin_amdgcn_readlane(abs_j, p);
          const int base = (i < 8) ? (2 + first_coeff2) : 1;
          if (av >= base) { // xWriteCoefRemainExGolomb TEncSbac.cpp:337-394 (bit count only)
            const uint32_t symbol = (uint32_t)(av - base);
            if (symbol < (3u << go_rice)) ep((int)(symbol >> go_rice) + 1 + go_rice);
            else {
              uint32_t len = (uint32_t)go_rice, cn = symbol - (3u << go_rice);
              while (cn >= (1u << len)) cn -= (1u << (len++));
              ep((int)(3 + len + 1 - go_rice) + (int)len);
            }
            if (av > (3 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
          }
          if (av >= 2) first_coeff2 = 0;
        }
      }
    }
  }
  }
  wsync();
  c->ctx[BA + lane] = (uint8_t)cxa; if (lane < 63) c->ctx[BB + lane] = (uint8_t)cxb;       // (the windows overlap nowhere: 18 < BA, BA + 63 <= 84 < BB)
  if (lane < 19) c->ctx[lane] = (uint8_t)cxh;
  if (lane == 0) { c->frac = frac; if (luma_cfrac == 1) s.cfrac_last = frac - frac_hdr; else if (luma_cfrac == 2) s.cfrac_last_c = frac - frac_hdr; else if (luma_cfrac == 3) s.cfrac_last_c += frac - frac_hdr; }
  wsync();
  return (uint32_t)(frac >> 15);
}
template <int NFIX> DEVN uint32_t code_coeff_wave_n(KR k, LCabac *c, int comp_, int n_, int dir_mode_, int tskip_flag_, int pre_, int luma_cfrac_)
{ return code_coeff_wave_i<NFIX>(k, c, comp_, n_, dir_mode_, tskip_flag_, pre_, luma_cfrac_); }
#ifndef HEVCDL_BITS_FIX
#define HEVCDL_BITS_FIX 16        // per call (tools/micro_rd.py), one copy / own copies: 4x4 4 082 -> 3 495 cycles, 8x8 7 335 -> 6 749
#endif
DEV uint32_t code_coeff_wave(KR k, LCabac *c, int comp, int n_, int dir_mode, int tskip_flag, int pre = PRE_COEF | PRE_NONE, int luma_cfrac = 0)
{
  const int n = uni(n_);
#if HEVCDL_BITS_FIX >= 4
  if (n == 4) return code_coeff_wave_n<4>(k, c, comp, n, dir_mode, tskip_flag, pre, luma_cfrac);
#endif
#if HEVCDL_BITS_FIX >= 8
  if (n == 8) return code_coeff_wave_n<8>(k, c, comp, n, dir_mode, tskip_flag, pre, luma_cfrac);
#endif
#if HEVCDL_BITS_FIX >= 16
  if (n == 16) return code_coeff_wave_n<16>(k, c, comp, n, dir_mode, tskip_flag, pre, luma_cfrac);
#endif
  return code_coeff_wave_n<0>(k, c, comp, n, dir_mode, tskip_flag, pre, luma_cfrac);
}

// ---------------------------------------------------------------------------------------------------
// mode syntax (TEncSbac.cpp:613-726, TComDataCU.cpp:1334-1461); uniform control flow, bins on lane 0
// ---------------------------------------------------------------------------------------------------
DEV void get_mpm(KR k, int x, int y, int preds[3], int *nmode)
{ // getIntraDirPredictor TComDataCU.cpp:1362-1445
  int left = DC, above = DC;
  if (x > k.tx0) left = part_attr(k, A_LDIR, (x >> 2) - 1, y >> 2);
  if ((y & 63) != 0) above = part_attr(k, A_LDIR, x >> 2, (y >> 2) - 1);
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
DEV void code_luma_dirs(KR k, LCabac *c, const Cu &cu, int first_pu, int npu)
{ // codeIntraDirLumaAng TEncSbac.cpp:643-696
  int idx[4];
  const int pu_size = (cu.part == SIZE_NxN) ? (1 << (cu.log2 - 1)) : (1 << cu.log2);
  for (int j = 0; j < npu; j++) {
    const int pu = first_pu + j, px = cu.x + (pu & 1) * pu_size, py = cu.y + (pu >> 1) * pu_size;
    const int dir = lds().a[A_LDIR][cu.zbase + pu * (cu.nparts >> 2) * (cu.part == SIZE_NxN)];
    int preds[3]; get_mpm(k, px, py, preds, nullptr);
    idx[j] = -1;
    for (int i = 0; i < 3; i++) if (dir == preds[i]) idx[j] = i;
    if (lane_id() == 0) enc_bin(c, CTX_INTRA_PRED, idx[j] != -1);
  }
  if (lane_id() == 0) for (int j = 0; j < npu; j++) enc_ep(c, idx[j] != -1 ? (idx[j] ? 2 : 1) : 5);
}
DEV void code_chroma_dir(KR k, LCabac *c, const Cu &cu)
{ // codeIntraDirChroma TEncSbac.cpp:698-726
  if (lane_id() != 0) return;
  if (lds().a[A_CDIR][cu.zbase] == DM_CHROMA) enc_bin(c, CTX_CHROMA_PRED, 0);
  else { enc_bin(c, CTX_CHROMA_PRED, 1); enc_ep(c, 2); }
}
DEV int split_ctx(KR k, int x, int y, int depth)
{ // getCtxSplitFlag TComDataCU.cpp:1447-1461
  int ctx = 0;
  if (x > k.tx0) ctx += part_attr(k, A_DEPTH, (x >> 2) - 1, y >> 2) > depth;
  if (y > k.ty0) ctx += part_attr(k, A_DEPTH, x >> 2, (y >> 2) - 1) > depth;
  return ctx;
}
DEV int min_tu_log2(const Cu &cu)
{ // getQuadtreeTULog2MinSizeInCU TComDataCU.cpp:1478-1503 (TU log2 2..5, intra TU depth 3)
  const int split = cu.part == SIZE_NxN; int r;
  if (cu.log2 < 2 + 3 - 1 + split) r = 2; else { r = cu.log2 - (3 - 1 + split); if (r > 5) r = 5; }
  return r;
}
DEV int tu_has_chroma_first(const Tu &tu) { return tu.log2 > 2 || (tu.zrel & 3) == 0; }
DEV int tu_has_chroma_last(const Tu &tu) { return tu.log2 > 2 || (tu.zrel & 3) == 3; }
DEV int tu_csize(const Tu &tu) { return tu.log2 > 2 ? 1 << (tu.log2 - 1) : 4; }
DEV int tu_czrel(const Tu &tu) { return tu.log2 > 2 ? tu.zrel : (tu.zrel & ~3); }
DEV int tu_cnparts(const Tu &tu) { return tu.log2 > 2 ? tu.nparts : 4; }
DEV Tu tu_child(const Tu &p, int i)
{
  Tu ch; const int h = 1 << (p.log2 - 1);
  ch.log2 = p.log2 - 1; ch.trd = p.trd + 1; ch.nparts = p.nparts >> 2;
  ch.x = p.x + (i & 1) * h; ch.y = p.y + (i >> 1) * h; ch.zrel = p.zrel + i * ch.nparts;
  return ch;
}
DEV int mode_of(KR k, const Cu &cu, int c, int zrel)
{ // TEncSearch.cpp:1178-1181
  if (!c) return lds().a[A_LDIR][cu.zbase + zrel];
  const int m = lds().a[A_CDIR][cu.zbase + zrel];
  return m == DM_CHROMA ? lds().a[A_LDIR][cu.zbase + (zrel & ~3)] : m;
}
DEV void code_qt_cbf(KR k, LCabac *c, const Cu &cu, const Tu &tu, int comp, int lowest)
{ // codeQtCbf TEncSbac.cpp:920-995 + getCtxQtCbf TComDataCU.cpp:1463-1476
  const int ctx = comp ? tu.trd : (tu.trd == 0 ? 1 : 0);
  const int w = comp ? tu_csize(tu) : (1 << tu.log2);
  const int d = tu.trd + ((!lowest && !(w >= 8)) ? 1 : 0);
  const int z = cu.zbase + (comp ? tu_czrel(tu) : tu.zrel);
  const int cbf = (lds().a[A_CBF + comp][z] >> d) & 1;
  if (lane_id() == 0) enc_bin(c, CTX_QT_CBF + (comp ? 5 : 0) + ctx, cbf);
}

// coefficients of one TU -> s->lvl (lane-parallel), source = QT layer buffer or the CTU record
DEV void load_tu_coef(KR k, int real, int comp, int log2_luma, int zabs_comp, int n)
{
  const int off = comp ? (zabs_comp * 16) >> 2 : zabs_comp * 16;
  GLB const int16_t *src = real ? (GLB const int16_t *)(k.records + (size_t)k.addr * REC_SIZE + REC_COEF) + comp_off(comp) + off
                            : k.coef_l + lay_coef(k, log2_luma, comp, zabs_comp);
  PROF_T0();
  wsync();
  for (int i = lane_id(); i < n * n; i += 64) lds().lvl[i] = src[i];
  wsync();
  PROF_ADD(k, 30);
}
// bit-count one coded TU block (cbf already known to be set)
DEV void code_tu_coeffs(KR k, LCabac *c, const Cu &cu, const Tu &tu, int comp, int real)
{
  const int zc = comp ? tu_czrel(tu) : tu.zrel;
  const int n = comp ? tu_csize(tu) : (1 << tu.log2);
  const int mode = uni(mode_of(k, cu, comp, zc));
  load_tu_coef(k, real, comp, tu.log2, cu.zbase + zc, n);
  { PROF_T0(); code_coeff_wave(k, c, comp, n, mode, lds().a[A_TSKIP + comp][cu.zbase + zc]); PROF_ADD(k, 11); }
  wsync();
}

template <int LOG2> DEV void enc_subdiv_cbf(KR k, LCabac *c, const Cu &cu, const Tu &tu, int luma, int chroma)
{ // xEncSubdivCbfQT TEncSearch.cpp:907-972
  const int subdiv = uni(lds().a[A_TRIDX][cu.zbase + tu.zrel]) > tu.trd;
  if (cu.part == SIZE_NxN && tu.trd == 0) { }
  else if (LOG2 > 5) { }
  else if (LOG2 == 2) { }
  else if (LOG2 == min_tu_log2(cu)) { }
  else if (luma && lane_id() == 0) enc_bin(c, CTX_SUBDIV + 5 - LOG2, subdiv);
  if (chroma) for (int comp = 1; comp < 3; comp++)
    if (LOG2 > 2 && (tu.trd == 0 || ((lds().a[A_CBF + comp][cu.zbase + tu.zrel] >> (tu.trd - 1)) & 1)))
      code_qt_cbf(k, c, cu, tu, comp, !subdiv);
  if (subdiv) { if constexpr (LOG2 > 2) for (int i = 0; i < 4; i++) enc_subdiv_cbf<LOG2 - 1>(k, c, cu, tu_child(tu, i), luma, chroma); }
  else if (luma) code_qt_cbf(k, c, cu, tu, 0, 1);
}
template <int LOG2> DEV void enc_coeff_qt(KR k, LCabac *c, const Cu &cu, const Tu &tu, int comp, int real)
{ // xEncCoeffQT TEncSearch.cpp:978-1012 (+ cbf test of TEncEntropy::encodeCoeffNxN :654-690)
  if (uni(lds().a[A_TRIDX][cu.zbase + tu.zrel]) > tu.trd) {
    if constexpr (LOG2 > 2) for (int i = 0; i < 4; i++) enc_coeff_qt<LOG2 - 1>(k, c, cu, tu_child(tu, i), comp, real);
    return;
  }
  if (comp && !tu_has_chroma_first(tu)) return;
  if (!((uni(lds().a[A_CBF + comp][cu.zbase + tu.zrel]) >> tu.trd) & 1)) return;
  code_tu_coeffs(k, c, cu, tu, comp, real);
}
DEV void enc_intra_header(KR k, LCabac *c, const Cu &cu, const Tu &tu, int luma, int chroma)
{ // xEncIntraHeader TEncSearch.cpp:1018-1087
  if (luma) {
    if (tu.zrel == 0 && cu.depth == 3 && lane_id() == 0) enc_bin(c, CTX_PART_SIZE, cu.part == SIZE_2Nx2N);
    if (cu.part == SIZE_2Nx2N) { if (tu.zrel == 0) code_luma_dirs(k, c, cu, 0, 1); }
    else { const int q = cu.nparts >> 2; if (tu.trd > 0 && (tu.zrel & (q - 1)) == 0) code_luma_dirs(k, c, cu, tu.zrel >> (2 * (cu.log2 - 2) - 2), 1); }
  }
  if (chroma && tu.zrel == 0) code_chroma_dir(k, c, cu);
}
// xGetIntraBitsQT (TEncSearch.cpp:1093-1117), luma only, of a TU that has just been coded as ONE transform block (tr_idx == tu.trd over its partitions: the unsplit
// alternative of recur_luma, a first-pass candidate, a child of the chain in spec_children) -- what intra_bits_qt<LOG2>(k, cu, tu, 1, 0) counts there, as one pass of the
// register-resident coder: the header flags (xEncIntraHeader :1018-1087, xEncSubdivCbfQT :907-972) are bins in front of the coefficients (code_coeff_wave) instead of
// three dependent LDS round trips each on lane 0, and the levels are read where code_tu_block left them (lvl_in_lds: TUs up to 16x16, see inv_transform_n) instead of
// coming back from the layer buffer.  Same bins, same contexts, same order per context; leaves `go` and s.cfrac_last as intra_bits_qt does.
#ifndef HEVCDL_BITS_INLINE
#define HEVCDL_BITS_INLINE 1     // the count behind a TU coding runs the bit counter inside code_tu_block's frame (the copies with the block size as a constant): no call frame
#endif                           // (37 scalar registers saved and restored through scratch) and no s_waitcnt vmcnt(0) at a function entry right behind the coding's stores
template <int NFIX = 0> DEV uint32_t luma_tu_bits_body(KR k, const Cu cu_, const Tu tu_, int lvl_in_lds_)
{
  CHECK_EXEC(11);
  PROF_T0();
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_); const int lvl_in_lds = uni(lvl_in_lds_);
  LSmem &s = lds();
  const int z = cu.zbase + tu.zrel, n = 1 << tu.log2;
  int pre = PRE_NONE, nb = 0, nep = 0;
  auto add = [&](int ctx, int v) { pre = (pre & ~(63 << (6 * nb))) | ((ctx | (v << 5)) << (6 * nb)); nb++; };
  wsync();
  if (tu.zrel == 0 && cu.depth == 3) add(CTX_PART_SIZE, cu.part == SIZE_2Nx2N ? 1 : 0);
  int pu = -1;
  if (cu.part == SIZE_2Nx2N) { if (tu.zrel == 0) pu = 0; }
  else { const int q = cu.nparts >> 2; if (tu.trd > 0 && (tu.zrel & (q - 1)) == 0) pu = tu.zrel >> (2 * (cu.log2 - 2) - 2); }
  if (pu >= 0) { // code_luma_dirs for this one PU (codeIntraDirLumaAng TEncSbac.cpp:643-696)
    const int pu_size = (cu.part == SIZE_NxN) ? (1 << (cu.log2 - 1)) : (1 << cu.log2);
    const int px = cu.x + (pu & 1) * pu_size, py = cu.y + (pu >> 1) * pu_size;
    const int dir = uni(s.a[A_LDIR][cu.zbase + pu * (cu.nparts >> 2) * (cu.part == SIZE_NxN)]);
    int preds[3]; get_mpm(k, px, py, preds, nullptr);
    int idx = -1;
    for (int i = 0; i < 3; i++) if (dir == uni(preds[i])) idx = i;
    add(CTX_INTRA_PRED, idx != -1 ? 1 : 0);
    nep = idx != -1 ? (idx ? 2 : 1) : 5;
  }
  if (!(cu.part == SIZE_NxN && tu.trd == 0) && tu.log2 <= 5 && tu.log2 != 2 && tu.log2 != min_tu_log2(cu)) add(CTX_SUBDIV + 5 - tu.log2, 0);
  const int cbf = (uni(s.a[A_CBF][z]) >> tu.trd) & 1;
  add(CTX_QT_CBF + (tu.trd == 0 ? 1 : 0), cbf);
  const int mode = uni(s.a[A_LDIR][z]), tskip = uni(s.a[A_TSKIP][z]);
  if (cbf && !lvl_in_lds) load_tu_coef(k, 0, 0, tu.log2, z, n);
  const int pre_all = pre | (nep << PRE_EP_SHIFT) | PRE_RESET | (cbf ? PRE_COEF : 0);
  uint32_t bits;
  if constexpr (NFIX != 0 && HEVCDL_BITS_INLINE && NFIX <= HEVCDL_BITS_FIX) bits = code_coeff_wave_i<NFIX>(k, &s.go, 0, NFIX, mode, tskip, pre_all, 1);
  else bits = code_coeff_wave(k, &s.go, 0, n, mode, tskip, pre_all, 1);
  wsync();
  PROF_ADD_T(k, 10, 49);
  return bits;
}
// xGetIntraBitsQT, chroma only, of a CU that is ONE transform unit (2Nx2N, tr_idx 0, up to 32x32: one block per component) -- what intra_bits_qt<LOG2>(k, cu, root, 0, 1)
// counts there: intra_chroma_pred_mode, cbf_cb, cbf_cr as bins in front of the first component that has coefficients, then Cb's and Cr's coefficients, on the
// register-resident coder (the flags were three dependent LDS round trips each on lane 0).  Leaves `go` and s.cfrac_last_c as intra_bits_qt does.
DEVN uint32_t chroma_cu_bits_1tu(KR k, const Cu cu_, const Tu tu_)
{
  CHECK_EXEC(11);
  PROF_T0();
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_);
  LSmem &s = lds();
  const int z = cu.zbase, n = tu_csize(tu);
  wsync();
  const int cdir = uni(s.a[A_CDIR][z]), dm = cdir == DM_CHROMA ? 1 : 0, mode = dm ? uni(s.a[A_LDIR][z]) : cdir;
  const int cbf1 = uni(s.a[A_CBF + 1][z]) & 1, cbf2 = uni(s.a[A_CBF + 2][z]) & 1;
  const int hdr = ((CTX_CHROMA_PRED | ((1 - dm) << 5)) | ((CTX_QT_CBF + 5) | (cbf1 << 5)) << 6 | ((CTX_QT_CBF + 5) | (cbf2 << 5)) << 12 | 63 << 18) | ((dm ? 0 : 2) << PRE_EP_SHIFT) | PRE_RESET;
  uint32_t bits;
  if (cbf1) { load_tu_coef(k, 0, 1, tu.log2, z, n); bits = code_coeff_wave(k, &s.go, 1, n, mode, uni(s.a[A_TSKIP + 1][z]), hdr | PRE_COEF, 2); }
  if (cbf2) { load_tu_coef(k, 0, 2, tu.log2, z, n); bits = code_coeff_wave(k, &s.go, 2, n, mode, uni(s.a[A_TSKIP + 2][z]), cbf1 ? (PRE_NONE | PRE_COEF) : (hdr | PRE_COEF), cbf1 ? 3 : 2); }
  if (!cbf1 && !cbf2) bits = code_coeff_wave(k, &s.go, 1, n, mode, 0, hdr, 2);
  wsync();
  PROF_ADD_T(k, 10, 49);
  return bits;
}
DEVN uint32_t luma_tu_bits(KR k, const Cu cu_, const Tu tu_, int lvl_in_lds_) { return luma_tu_bits_body(k, cu_, tu_, lvl_in_lds_); }     // (the call form: the transform-skip trial of a 4x4 TU; everywhere else the count rides inside code_tu_block)
template <int LOG2> DEVN uint32_t intra_bits_qt(KR k, const Cu cu_, const Tu tu_, int luma_, int chroma_)
{
  CHECK_EXEC(11);
  PROF_T0();
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_); const int luma = uni(luma_), chroma = uni(chroma_); // xGetIntraBitsQT TEncSearch.cpp:1093-1117
  LCabac *c = &lds().go;
  wsync();
  if (lane_id() == 0) reset_bits(c);
  enc_intra_header(k, c, cu, tu, luma, chroma);
  enc_subdiv_cbf<LOG2>(k, c, cu, tu, luma, chroma);
  wsync();
  const unsigned long long f0_ = c->frac;
  if (luma) enc_coeff_qt<LOG2>(k, c, cu, tu, 0, 0);
  wsync();
  if (luma && lane_id() == 0) lds().cfrac_last = c->frac - f0_;
  if (chroma) {
    const unsigned long long f1_ = c->frac;
    enc_coeff_qt<LOG2>(k, c, cu, tu, 1, 0); enc_coeff_qt<LOG2>(k, c, cu, tu, 2, 0);
    wsync();
    if (lane_id() == 0) lds().cfrac_last_c = c->frac - f1_;
  }
  wsync();
  PROF_ADD_T(k, 10, 49);
  return uni((int)get_bits(c));
}
DEV unsigned long long uni64(unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)v); }
// Bits of a split TU (xGetIntraBitsQT of the parent after its four children, TEncSearch.cpp:1679-1700) WITHOUT coding the children's
// coefficients a second time.  The count is: mode header + subdivision / cbf flags of the whole tree, then every TU's coefficients in
// coding order, all from the state the parent started with (root).  Coefficient bins only touch the coefficient contexts (significance,
// last position, greater-1/2, transform skip: indices >= CTX_SIG_CG) and the flags only the others, and the children were counted one
// after the other from the same starting state -- so the coefficient bins of the recount are, bin for bin, the ones the children's own
// counts coded: their fractional bits (cfrac, summed by the caller) and their final contexts (the coder `go` now holds them) are taken
// over, and only the flags are coded here.  Leaves `go` exactly as the full recount would.
template <int LOG2> DEVN uint32_t split_bits(KR k, const Cu cu_, const Tu tu_, LCabac *root, unsigned long long cfrac)
{
  PROF_T0();
  PROF_GLUE_T0();
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_);
  LSmem &s = lds();
  LCabac *c = &s.go;
  cabac_copy(k, &s.tbest, c);                   // the children's end state: coefficient contexts (tbest: the transform-skip trial's snapshot, dead once the children are coded)
  cabac_copy(k, c, root);
  if (lane_id() == 0) reset_bits(c);
  enc_intra_header(k, c, cu, tu, 1, 0);
  enc_subdiv_cbf<LOG2>(k, c, cu, tu, 1, 0);
  wsync();
  for (int i = CTX_SIG_CG + lane_id(); i < NUM_CTX; i += 64) c->ctx[i] = s.tbest.ctx[i];
  if (lane_id() == 0) c->frac += cfrac;
  wsync();
#if defined(HEVCDL_PROF_GLUE) && defined(HEVCDL_KERNEL_PROF)
  PROF_GLUE(34);
#else
  PROF_ADD_T(k, 10, 49);
#endif
  return uni((int)get_bits(c));
}
template <int LOG2> DEV void enc_transform(KR k, LCabac *c, const Cu &cu, const Tu &tu)
{ // TEncEntropy::xEncodeTransform TEncEntropy.cpp:200-398 (real coefficients; chroma of a 4x4 quad with its LAST block)
  const int z = cu.zbase + tu.zrel;
  const int subdiv = uni(lds().a[A_TRIDX][z]) > tu.trd;
  if (cu.part == SIZE_NxN && tu.trd == 0) { }
  else if (LOG2 > 5) { }
  else if (LOG2 == 2) { }
  else if (LOG2 == min_tu_log2(cu)) { }
  else if (lane_id() == 0) enc_bin(c, CTX_SUBDIV + 5 - LOG2, subdiv);
  const int first = tu.trd == 0;
  for (int comp = 1; comp < 3; comp++)
    if (first || LOG2 > 2)
      if (first || ((lds().a[A_CBF + comp][z] >> (tu.trd - 1)) & 1)) code_qt_cbf(k, c, cu, tu, comp, !subdiv);
  if (subdiv) { if constexpr (LOG2 > 2) for (int i = 0; i < 4; i++) enc_transform<LOG2 - 1>(k, c, cu, tu_child(tu, i)); return; }
  code_qt_cbf(k, c, cu, tu, 0, 1);
  for (int comp = 0; comp < 3; comp++) {
    if (comp && !tu_has_chroma_last(tu)) continue;
    if (!((uni(lds().a[A_CBF + comp][z]) >> tu.trd) & 1)) continue;
    code_tu_coeffs(k, c, cu, tu, comp, 1);
  }
}
DEVN void enc_cu_syntax(KR k, LCabac *c, const Cu cu_)
{
  PROF_T0();
  const Cu cu = ucu(cu_); // TEncCu.cpp:1636-1654 (RD) and xEncodeCU :1222-1270 (state-advancing encode); I-slice, no PCM/TQB/DQP
  wsync();
  if (cu.depth == 3 && lane_id() == 0) enc_bin(c, CTX_PART_SIZE, cu.part == SIZE_2Nx2N);
  code_luma_dirs(k, c, cu, 0, cu.part == SIZE_NxN ? 4 : 1);
  code_chroma_dir(k, c, cu);
  const Tu root = { cu.x, cu.y, cu.log2, 0, 0, cu.nparts };
  switch (cu.log2) {
    case 6: enc_transform<6>(k, c, cu, root); break;
    case 5: enc_transform<5>(k, c, cu, root); break;
    case 4: enc_transform<4>(k, c, cu, root); break;
    default: enc_transform<3>(k, c, cu, root); break;
  }
  wsync();
  PROF_ADD(k, 13);
}

// ---------------------------------------------------------------------------------------------------
// TU coding: xIntraCodingTUBlock TEncSearch.cpp:1129-1424 (mode012: 0 predict, 1 predict+save, 2 reuse saved, 3 predict and write the
// reconstruction to the layer only: a first-pass candidate of a PU coded as one TU -- nothing reads its picture samples)
// ---------------------------------------------------------------------------------------------------
// NFIX != 0: the block size as a compile-time constant (4x4, 8x8 and 16x16 have copies of their own: one round per loop over the samples, strides and shifts as constants,
// their own copy of rdoq_wave); code_tu_block picks the copy.
template <int NFIX> DEVN TuRes code_tu_block_n(KR k, const Cu cu_, const Tu tu_, int comp_, int mode012_, int count_)
{ // count_ (luma, the TU coded as one transform block): the bit count that follows every such coding (luma_tu_bits) is made before the function returns -- one
  // call frame (37 scalar registers saved and restored, two scratch round trips) per TU coding instead of two
  CHECK_EXEC(1);
  PROF_T0();
  PROF_MARK0();
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_); const int comp = uni(comp_), mode012 = uni(mode012_), count = uni(count_);
  LSmem &s = lds();
  const int n = NFIX ? NFIX : (comp ? tu_csize(tu) : (1 << tu.log2)), log2n = NFIX == 4 ? 2 : (NFIX == 8 ? 3 : (NFIX == 16 ? 4 : ilog2(n)));
  const int zrel = comp ? tu_czrel(tu) : tu.zrel, zabs = cu.zbase + zrel;
  const int x = comp ? tu.x >> 1 : tu.x, y = comp ? tu.y >> 1 : tu.y;
  const int cs = cstride(comp), bo = boff(k, comp, x, y), ps = pstride(k, comp);
  const int mode = uni(mode_of(k, cu, comp, zrel));
  const int tskip = uni(s.a[A_TSKIP + comp][zabs]);
  const int use_rdoq = tools_of(k) & (int)(tskip ? HEVCDL_TOOL_RDOQTS : HEVCDL_TOOL_RDOQ);        // useRDOQ = transform skip ? RDOQTS : RDOQ (TComTrQuant.cpp:1152)
  if (mode012 != 2) {
    if (HEVCDL_REFS_INLINE && NFIX) build_refs_i(k, comp, x, y, n, 0); else build_refs(k, comp, x, y, n, 0);
    if (ub(use_filtered_refs(comp, mode, n))) filter_refs(k, n);
    predict_block(k, comp, mode, n);
    if (mode012 == 1 && lane_id() < 16) s.ts_pred[comp][lane_id()] = s.pred[lane_id()];
  } else { wsync(); if (lane_id() < 16) s.pred[lane_id()] = s.ts_pred[comp][lane_id()]; }
  wsync();
  PROF_MARK(24);
  // (asking for the original samples ahead of the prediction -- four registers a lane for TUs up to 16x16, used in the residual and again in the distortion loop --
  //  measured WORSE: 600 frames 6.06 -> 6.12 s; the registers cost more than the round trip)
  GLB const pel_t *org = k.org[comp] + (size_t)y * ps + x;
  for (int i = lane_id(); i < n * n; i += 64) s.resi[(i >> log2n) * RS(n) + (i & (n - 1))] = (int16_t)((int)org[(size_t)(i >> log2n) * ps + (i & (n - 1))] - (int)s.pred[i]);
  if (!comp) set_parts(k, s.a[A_TRIDX], zabs, tu.nparts, tu.trd);
  wsync();
  PROF_MARK(25);
#ifdef HEVCDL_STAGE_TRACE
  GLB unsigned *tr = stage_alloc(k, 6 + 3 * n * n);                 // transformNxN: residual, coefficients (TComTrQuant.cpp:1496-1516)
  if (tr) {
    if (lane_id() < 6) tr[lane_id()] = lane_id() == 0 ? 2u : (lane_id() == 1 ? (unsigned)n : (lane_id() == 2 ? (unsigned)comp : 0u));
    for (int i = lane_id(); i < n * n; i += 64) tr[6 + i] = (unsigned)(int)s.resi[(i >> log2n) * RS(n) + (i & (n - 1))];
  }
  wsync();
#endif
  if (tskip) { for (int i = lane_id(); i < n * n; i += 64) s.tc[i] = (int16_t)((int)s.resi[(i >> log2n) * RS(n) + (i & (n - 1))] << (13 - BD)); wsync(); }   // n == 4: one pass, every lane reads before any writes
  else fwd_transform(k, n, !comp && n == 4);
  PROF_MARK(26);
  const int cbf_ctx = comp ? tu.trd : (tu.trd == 0 ? 1 : 0);
#ifdef HEVCDL_STAGE_TRACE
  if (tr) for (int i = lane_id(); i < n * n; i += 64) tr[6 + n * n + i] = (unsigned)(int)s.tc[i];
#endif
  uint32_t abs_sum;                                             // (wave-uniform as rdoq_wave returns it)
  { PROF_T0();
    if (use_rdoq) abs_sum = rdoq_wave<NFIX>(k, &s.go, comp, n, mode, cbf_ctx);
    else abs_sum = plain_quant_wave(k, comp, n, mode);
    PROF_ADD(k, 6); PROF_ADD(k, 60 + log2n - 2); }
  wsync();
  PROF_MARK(27);
#ifdef HEVCDL_STAGE_TRACE
  if (tr) for (int i = lane_id(); i < n * n; i += 64) tr[6 + 2 * n * n + i] = abs_sum > 0 ? (unsigned)(int)s.lvl[i] : 0u;      // levels behind the quantiser (:1525-1528)
  GLB unsigned *ti = abs_sum > 0 ? stage_alloc(k, 6 + 3 * n * n) : nullptr;                                                    // invTransformNxN (:1603-1662)
  if (ti) {
    if (lane_id() < 6) ti[lane_id()] = lane_id() == 0 ? 3u : (lane_id() == 1 ? (unsigned)n : (lane_id() == 2 ? (unsigned)comp : 0u));
    for (int i = lane_id(); i < n * n; i += 64) ti[6 + i] = (unsigned)(int)s.lvl[i];
  }
#endif
  set_parts(k, s.a[A_CBF + comp], zabs, comp ? tu_cnparts(tu) : tu.nparts, (abs_sum > 0 ? 1 : 0) << tu.trd);
  GLB int16_t *cl = k.coef_l + lay_coef(k, tu.log2, comp, zabs);
  if (abs_sum > 0) {
    for (int i = lane_id(); i < n * n; i += 64) cl[i] = s.lvl[i];
    dequant(k, comp, n);
#ifdef HEVCDL_STAGE_TRACE
    wsync();
    if (ti) for (int i = lane_id(); i < n * n; i += 64) ti[6 + n * n + i] = (unsigned)(int)s.tc[i];
    wsync();
#endif
    if (tskip) { for (int i = lane_id(); i < n * n; i += 64) s.resi[(i >> log2n) * RS(n) + (i & (n - 1))] = (int16_t)((s.tc[i] + (1 << (12 - BD))) >> (13 - BD)); wsync(); }
    else inv_transform(k, n, !comp && n == 4);
#ifdef HEVCDL_STAGE_TRACE
    wsync();
    if (ti) for (int i = lane_id(); i < n * n; i += 64) ti[6 + 2 * n * n + i] = (unsigned)(int)s.resi[(i >> log2n) * RS(n) + (i & (n - 1))];
#endif
  } else {
    for (int i = lane_id(); i < n * n; i += 64) { cl[i] = 0; s.resi[(i >> log2n) * RS(n) + (i & (n - 1))] = 0; }
    wsync();
  }
  PROF_MARK(28);
  GLB pel_t *rq = k.rec_l + lay_rec(k, tu.log2, comp, x, y);
  int rps; GLB pel_t *rp = rec_target(k, comp, x, y, rps);
  uint32_t d = 0;
  for (int i = lane_id(); i < n * n; i += 64) {
    const int r = i >> log2n, cc = i & (n - 1);
    const int v = clip8((int)s.pred[i] + (int)s.resi[r * RS(n) + cc]);
    s.pred[i] = (pel_t)v;
    rq[r * cs + cc] = (pel_t)v;
    if (mode012 != 3) rp[(size_t)r * rps + cc] = (pel_t)v;
    const int df = v - (int)org[(size_t)r * ps + cc];
    d += (uint32_t)(df * df) >> SSE_SH;                  // per sample, TComRdCost.cpp xGetSSE*
  }
  d = (uint32_t)wave_sum_i((int)d);
  if (comp) d = (uint32_t)(k.cweight * (double)d);            // getDistPart TComRdCost.cpp:350-353
  wsync();
  PROF_MARK(29);
  PROF_ADD_T(k, 9, 48);
  PROF_ADD(k, 56 + log2n - 2);
  TuRes res = { d, 0 };
  if (count) res.bits = luma_tu_bits_body<NFIX>(k, cu, tu, log2n <= 4);
  return res;
}
#ifndef HEVCDL_TU_FIX
#define HEVCDL_TU_FIX 16         // 0: one copy of code_tu_block for every size; 4 / 8 / 16: blocks up to that size get their own.  Measured (round 5), 0 / 4 / 8 / 16:
                                 // one frame 2.62 / 2.60 / 2.55 / 2.54 s, 600 frames 5.755 / 5.74 / 5.65 / 5.61 s, 2048 frames 14.64 / 14.48 / 14.43 / 14.47 s
#endif
DEV TuRes code_tu_block(KR k, const Cu &cu, const Tu &tu, int comp, int mode012, int count = 0)
{
  const int n = uni(comp) ? tu_csize(utu(tu)) : (1 << uni(tu.log2));
#if HEVCDL_TU_FIX >= 4
  if (n == 4) return code_tu_block_n<4>(k, cu, tu, comp, mode012, count);
#endif
#if HEVCDL_TU_FIX >= 8
  if (n == 8) return code_tu_block_n<8>(k, cu, tu, comp, mode012, count);
#endif
#if HEVCDL_TU_FIX >= 16
  if (n == 16) return code_tu_block_n<16>(k, cu, tu, comp, mode012, count);
#endif
  return code_tu_block_n<0>(k, cu, tu, comp, mode012, count);
}

DEV void store_ts_result(KR k, const Cu &cu, const Tu &tu, int comp)
{ // xStoreIntraResultQT TEncSearch.cpp:1784-1816 (4x4 blocks); s->lvl / s->pred still hold the block just coded
  wsync();
  if (lane_id() < 16) { lds().ts_coef[comp][lane_id()] = ((uni((int)((lds().a[A_CBF + comp][cu.zbase + (comp ? tu_czrel(tu) : tu.zrel)] >> tu.trd) & 1)) ? lds().lvl[lane_id()] : (int16_t)0)); lds().ts_rec[comp][lane_id()] = lds().pred[lane_id()]; }
  wsync();
}
DEV void load_ts_result(KR k, const Cu &cu, const Tu &tu, int comp)
{ // xLoadIntraResultQT TEncSearch.cpp:1819-1870
  const int zabs = cu.zbase + (comp ? tu_czrel(tu) : tu.zrel);
  const int x = comp ? tu.x >> 1 : tu.x, y = comp ? tu.y >> 1 : tu.y, cs = cstride(comp), bo = boff(k, comp, x, y);
  int rps; GLB pel_t *rp = rec_target(k, comp, x, y, rps);
  wsync();
  if (lane_id() < 16) {
    k.coef_l[lay_coef(k, tu.log2, comp, zabs) + lane_id()] = lds().ts_coef[comp][lane_id()];
    const int r = lane_id() >> 2, cc = lane_id() & 3; const pel_t v = lds().ts_rec[comp][lane_id()];
    k.rec_l[lay_rec(k, tu.log2, comp, x, y) + r * cs + cc] = v;
    rp[(size_t)r * rps + cc] = v;
  }
  wsync();
}

DEV void chroma_mode_list(int luma_mode, uint32_t (&mode_list)[5])
{ // getAllowedChromaDir TComDataCU.cpp:1334-1353
  mode_list[0] = PLANAR; mode_list[1] = VER; mode_list[2] = HOR; mode_list[3] = DC; mode_list[4] = DM_CHROMA;
  for (int i = 0; i < 4; i++) if ((int)mode_list[i] == luma_mode) { mode_list[i] = 34; break; }
}
DEVN void region_open(LRegion &r, int kind_, int n_, const Cu cu_, const Tu tu_);
DEVN void region_run(KR k, LRegion &r);
DEV void region_close(LRegion &r) { }
DEV void region_publish(LRegion &r) { wg_release(); lds_add(&r.ticket, 1 << 16); ring_bell(); }         // one more task (parameters written before)
DEVN int remote_poll(LRegion &r);
DEVN int remote_room(int need = 1);
DEVN void chroma_post(KR k, const Cu cu_, const Tu tu_, int m0, int m1, int m2, int m3, int m4, int prepare_only);
DEV void chroma_mode_list(int luma_mode, uint32_t (&mode_list)[5]);
DEVN void chroma_collect(LRegion &r);
DEVN void remote_post(KR k, const Cu cu_, const Tu tu_, int reg_, int mode_, double memo_cost, uint32_t memo_dist, int with_chroma);
DEV void region_wait(LRegion &r, int n)
{
  PROF_T0();
  while (lds_load(&r.done) < n) { if (uni(r.kind) == T_REMOTE) { if (!remote_poll(r)) __builtin_amdgcn_s_sleep(32); } else __builtin_amdgcn_s_sleep(2); }
  wg_acquire();
  PROF_ADD(0, 33);
}
DEV void state_to_global(GLB unsigned long long *dst, const LCabac *src) { wsync(); if (lane_id() < 21) dst[lane_id()] = ((LDS const unsigned long long *)src)[lane_id()]; wsync(); }
DEV void state_from_global(LCabac *dst, GLB const unsigned long long *src) { wsync(); if (lane_id() < 21) ((LDS unsigned long long *)dst)[lane_id()] = src[lane_id()]; wsync(); }
// coder snapshots as 21 words, whichever side they live on
typedef unsigned long long u64_;
DEV void st_copy(GLB u64_ *dst, GLB const u64_ *src) { wsync(); if (lane_id() < 21) dst[lane_id()] = src[lane_id()]; wsync(); }
DEV void st_copy(GLB u64_ *dst, LDS const u64_ *src) { wsync(); if (lane_id() < 21) dst[lane_id()] = src[lane_id()]; wsync(); }
DEV void st_copy(LDS u64_ *dst, GLB const u64_ *src) { wsync(); if (lane_id() < 21) dst[lane_id()] = src[lane_id()]; wsync(); }
DEV void st_copy(LDS u64_ *dst, LDS const u64_ *src) { wsync(); if (lane_id() < 21) dst[lane_id()] = src[lane_id()]; wsync(); }
DEV LDS u64_ *st_of(LCabac *c) { return (LDS u64_ *)c; }
// The CU walk's snapshots next[depth] / temp[depth] and the saved arrays of the best candidate: in LDS where a wave's block has the room (the eight-wave builds), in the
// executing wave's HBM workspace (LOG_COLD) in the ten-wave build; test[depth] (unsplit-vs-split of a first-pass TU: rare) always in the workspace.
enum { COLD_NEXT = 0, COLD_TEMP = 4, COLD_TEST = 8 };
DEV GLB u64_ *cold_test(int depth) { return lds().my_log + (size_t)LOG_COLD * (LEAF_LOG / 8) + (size_t)(COLD_TEST + depth) * 21; }
#if !defined(HEVCDL_RD_WIDE) && !defined(HEVCDL_MICRO_SMALL)
DEV LDS u64_ *cold_state(int which, int depth) { return (LDS u64_ *)&lds().nt[which + depth]; }
DEV LDS uint8_t *cold_sv(int c) { return lds().sv[c]; }
#else
DEV GLB u64_ *cold_state(int which, int depth) { return lds().my_log + (size_t)LOG_COLD * (LEAF_LOG / 8) + (size_t)(which + depth) * 21; }
DEV GLB uint8_t *cold_sv(int c) { return (GLB uint8_t *)(lds().my_log + (size_t)LOG_COLD * (LEAF_LOG / 8) + 12 * 21) + c * 256; }
#endif
static_assert(12 * 21 * 8 + 4 * 256 <= LOG_COLD_N * LEAF_LOG, "cold area");
struct DistCbf { uint32_t dist, cbf; unsigned long long cfrac; };
template <int LOG2> DEVN DistCbf spec_children(KR k, const Cu cu_, const Tu tu_);
DEVN int region_claim(LRegion &r);
DEV void import_owner(int owner);
template <bool LEAF> DEVN void run_task(LRegion &r, int idx_);

// xRecurIntraCodingLumaQT TEncSearch.cpp:1430-1738
// memo != 0 (second RD pass, TU == PU): the unsplit coding of this TU with this mode was evaluated in the first pass from
// the same coder state -- (memo_dist, memo_cost) are its results, bit for bit what a re-run would give -- so only the
// split alternative is evaluated; if the unsplit TU wins, its arrays / reconstruction come back from the saved best.
template <int LOG2, bool SPEC = false> DEVN DistCost recur_luma(KR k, const Cu cu_, const Tu tu_, int check_first_, int memo_ = 0, uint32_t memo_dist = 0, double memo_cost = 0.0)
{
  CHECK_EXEC(2);
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_); const int check_first = uni(check_first_), memo = uni(memo_);
  LSmem &s = lds();
  const int full_depth = cu.depth + tu.trd, zabs = cu.zbase + tu.zrel;
  const int check_full = LOG2 <= 5;
  int check_split = LOG2 > min_tu_log2(cu);
  if (check_first && check_full) check_split = 0;
  double single_cost = MAX_DOUBLE; uint32_t single_dist = 0, single_cbf = 0; int best_ts = 0;
  unsigned long long single_cfrac = 0;
  const int check_ts = (LOG2 == 2) && (tools_of(k) & (int)HEVCDL_TOOL_TSKIP) && (cu.part == SIZE_NxN || !(tools_of(k) & (int)HEVCDL_TOOL_TSKIP_FAST));     // TransformSkipFast: only in NxN CUs (TEncSearch.cpp:1502-1505)
  if (memo) { single_cost = memo_cost; single_dist = memo_dist; cabac_copy(k, &s.root[full_depth], &s.go); }
  else if (check_full) {
    if (check_ts) {
      cabac_copy(k, &s.root[full_depth], &s.go);
      for (int m = 0; m < 2; m++) {
        uint32_t d = 0; double cost;
        set_parts(k, s.a[A_TSKIP + 0], zabs, tu.nparts, m); wsync();
        d = code_tu_block(k, cu, tu, 0, m == 0 ? 1 : 2).dist;
        const uint32_t cbf = (uint32_t)(uni(s.a[A_CBF][zabs]) >> tu.trd) & 1;
        if (m == 1 && cbf == 0) cost = MAX_DOUBLE;
        else {
          if (m == 0) store_ts_result(k, cu, tu, 0);           // before the bit count reuses s->lvl
          const uint32_t bits = luma_tu_bits(k, cu, tu, LOG2 <= 4); cost = calc_rd_cost(k, bits, d);
        }
        if (ub(cost < single_cost)) {
          single_cost = cost; single_dist = d; single_cbf = cbf; best_ts = m; single_cfrac = uni64(s.cfrac_last);
          if (m == 0) cabac_copy(k, &s.tbest, &s.go);
        }
        if (m == 0) cabac_copy(k, &s.go, &s.root[full_depth]);
      }
      set_parts(k, s.a[A_TSKIP + 0], zabs, tu.nparts, best_ts); wsync();
      if (best_ts == 0) {
        load_ts_result(k, cu, tu, 0);
        set_parts(k, s.a[A_CBF], zabs, tu.nparts, (int)(single_cbf << tu.trd)); wsync();
        cabac_copy(k, &s.go, &s.tbest);
      }
    } else {
      if (check_split) cabac_copy(k, &s.root[full_depth], &s.go);
      set_parts(k, s.a[A_TSKIP + 0], zabs, tu.nparts, 0); wsync();
      const TuRes tr = code_tu_block(k, cu, tu, 0, (check_first == 2 && !check_split) ? 3 : 0, 1);
      single_dist = tr.dist;
      if (check_split) single_cbf = (uint32_t)(uni(s.a[A_CBF][zabs]) >> tu.trd) & 1;
      const uint32_t bits = tr.bits;
      single_cost = calc_rd_cost(k, bits, single_dist);
      single_cfrac = uni64(s.cfrac_last);
    }
  }
  if constexpr (LOG2 > 2) {
    if (check_split) {
      if (memo) { }
      else if (check_full) { st_copy(cold_test(full_depth), st_of(&s.go)); cabac_copy(k, &s.go, &s.root[full_depth]); }
      else cabac_copy(k, &s.root[full_depth], &s.go);
      double split_cost = 0; uint32_t split_dist = 0, split_cbf = 0;
      unsigned long long split_cfrac = 0;
      bool spec = false, aborted = false;
      if constexpr (SPEC && LOG2 >= 4 && LOG2 <= 5) spec = memo && !uni(k.in_task) && spare_waves() && (LOG2 - 1 > min_tu_log2(cu));
      if (spec) { // second pass of a PU, spare waves in the workgroup: the children's two alternatives run concurrently (spec_children)
        if constexpr (SPEC && LOG2 >= 4 && LOG2 <= 5) { const DistCbf dc = spec_children<LOG2>(k, cu, tu); split_dist = dc.dist; split_cbf = dc.cbf; split_cfrac = dc.cfrac; }
      } else for (int i = 0; i < 4; i++) {
        const Tu ch = tu_child(tu, i);
        { const DistCost r = recur_luma<LOG2 - 1>(k, cu, ch, check_first); split_dist += r.dist; split_cost += r.cost; split_cfrac += r.cfrac; }
        split_cbf |= (uint32_t)(uni(s.a[A_CBF][cu.zbase + ch.zrel]) >> ch.trd) & 1;
#ifndef HEVCDL_STAGE_TRACE
        // The split can no longer win: its cost is (distortion of the four children) + lambda * ((fraction + flag bins + the children's coefficient bins) >> 15), both
        // terms only grow with every child, and calc_rd_cost is monotone in both -- so once the children coded so far reach the unsplit cost, `split < unsplit`
        // (strict, TEncSearch.cpp:1703) is decided and the remaining children are not coded (nothing they would leave behind is read when the unsplit TU wins).
        // (The stage-trace build codes them: the reference's trace lists their blocks.)
        if (i < 3) {
          double bound = single_cost;
          if (memo && single_cost == MAX_DOUBLE && uni(s.bound_child) >= 0) { // the split alternative of a child of a second pass (T_LUMA_SPLIT): it is up against the
            // chain's unsplit coding of that child, which runs meanwhile (spec_children) -- once that cost is there, it is the bound
            LRegion &rr = wg_shared().reg[uni(s.bound_reg)][1];
            if (lds_load(&rr.modes[8 + uni(s.bound_child)])) { wg_acquire(); bound = rr.cost[4 + uni(s.bound_child)]; }
          }
          if (ub(bound < MAX_DOUBLE) && ub(!(calc_rd_cost(k, (uint32_t)(split_cfrac >> 15), split_dist) < bound))) {
            if (memo && single_cost == MAX_DOUBLE) { const DistCost r = { 0, MAX_DOUBLE, 0 }; return r; }     // the task's answer: not cheaper than the unsplit child
            aborted = true; break;
          }
        }
#endif
      }
      if (!aborted) {
      if (split_cbf) { for (int i = lane_id(); i < tu.nparts; i += 64) s.a[A_CBF][zabs + i] |= (uint8_t)(1 << tu.trd); }
      const uint32_t bits = split_bits<LOG2>(k, cu, tu, &s.root[full_depth], split_cfrac);
      split_cost = calc_rd_cost(k, bits, split_dist);
      if (ub(split_cost < single_cost)) { const DistCost r = { split_dist, split_cost, split_cfrac }; return r; }
      }
      PROF_GLUE_T0();
      if (memo) { // the saved best candidate of the first pass IS the unsplit coding (sv_* / best_rec, est_intra_luma)
        wsync();
        for (int i = lane_id(); i < tu.nparts; i += 64) { s.a[A_TRIDX][zabs + i] = cold_sv(0)[i]; s.a[A_CBF][zabs + i] = cold_sv(1)[i]; s.a[A_TSKIP][zabs + i] = cold_sv(2)[i]; }
      } else {
        st_copy(st_of(&s.go), cold_test(full_depth));
        set_parts(k, s.a[A_TRIDX], zabs, tu.nparts, tu.trd);
        set_parts(k, s.a[A_CBF], zabs, tu.nparts, (int)(single_cbf << tu.trd));
        set_parts(k, s.a[A_TSKIP + 0], zabs, tu.nparts, best_ts);
      }
      const int n = 1 << LOG2, bo = boff(k, 0, tu.x, tu.y);
      GLB const pel_t *rq = memo ? k.best_rec + bo : k.rec_l + lay_rec(k, LOG2, 0, tu.x, tu.y);
      int rps; GLB pel_t *rp = rec_target(k, 0, tu.x, tu.y, rps);
      wsync();
      for (int i = lane_id(); i < n * n; i += 64) rp[(size_t)(i >> LOG2) * rps + (i & (n - 1))] = rq[(i >> LOG2) * 64 + (i & (n - 1))];
      wsync();
      PROF_GLUE(35);
    }
  }
  const DistCost r = { single_dist, single_cost, single_cfrac };
  return r;
}

// xSetIntraResultLumaQT / xSetIntraResultChromaQT TEncSearch.cpp:1741-1781, 2150-2198
template <int LOG2> DEV void set_result(KR k, const Cu &cu, const Tu &tu, int comp, GLB const int16_t *src_coef, GLB const pel_t *src_rec, int slz, int slx, int sly)
{ // source: a layer set (the wave's own, or the result slot of the winning task) and its origin
  if (uni(lds().a[A_TRIDX][cu.zbase + tu.zrel]) > tu.trd) {
    if constexpr (LOG2 > 2) for (int i = 0; i < 4; i++) set_result<LOG2 - 1>(k, cu, tu_child(tu, i), comp, src_coef, src_rec, slz, slx, sly);
    return;
  }
  if (comp && !tu_has_chroma_first(tu)) return;
  const int n = comp ? tu_csize(tu) : (1 << LOG2), log2n = ilog2(n);
  const int zabs = cu.zbase + (comp ? tu_czrel(tu) : tu.zrel);
  const int off = comp_off(comp) + (comp ? (zabs * 16) >> 2 : zabs * 16);
  GLB int16_t *dstc = (GLB int16_t *)(k.records + (size_t)k.addr * REC_SIZE + REC_COEF) + off;
  GLB const int16_t *srcc = src_coef + lay_coef_o(slz, LOG2, comp, zabs);
  const int x = comp ? tu.x >> 1 : tu.x, y = comp ? tu.y >> 1 : tu.y, cs = cstride(comp), bo = comp_off(comp) + boff(k, comp, x, y);
  GLB const pel_t *rq = src_rec + lay_rec_o(slx, sly, LOG2, comp, x, y); GLB pel_t *br = k.best_rec + bo;
  for (int i = lane_id(); i < n * n; i += 64) { dstc[i] = srcc[i]; const int o = (i >> log2n) * cs + (i & (n - 1)); br[o] = rq[o]; }
}
DEVN void set_result_cu(KR k, const Cu cu_, const Tu tu_, int comp_, GLB const int16_t *src_coef, GLB const pel_t *src_rec, int slz_, int slx_, int sly_)
{
  const int slz = uni(slz_), slx = uni(slx_), sly = uni(sly_);
  CHECK_EXEC(10);
  PROF_T0();
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_); const int comp = uni(comp_);
  wsync();
  switch (tu.log2) {
    case 6: set_result<6>(k, cu, tu, comp, src_coef, src_rec, slz, slx, sly); break;
    case 5: set_result<5>(k, cu, tu, comp, src_coef, src_rec, slz, slx, sly); break;
    case 4: set_result<4>(k, cu, tu, comp, src_coef, src_rec, slz, slx, sly); break;
    case 3: set_result<3>(k, cu, tu, comp, src_coef, src_rec, slz, slx, sly); break;
    default: set_result<2>(k, cu, tu, comp, src_coef, src_rec, slz, slx, sly); break;
  }
  wsync();
  PROF_ADD(k, 15);
}
// SPEC: only the second pass run as a task of its own (T_LUMA_P2) hands its children's split alternatives to other waves (spec_children)
template <bool SPEC = false> DEV DistCost recur_luma_any(KR k, const Cu &cu, const Tu &tu, int check_first, int memo = 0, uint32_t md = 0, double mc = 0.0)
{
  switch (tu.log2) {
    case 6: return recur_luma<6>(k, cu, tu, check_first, memo, md, mc);
    case 5: return recur_luma<5, SPEC>(k, cu, tu, check_first, memo, md, mc);
    case 4: return recur_luma<4, SPEC>(k, cu, tu, check_first, memo, md, mc);
    case 3: return recur_luma<3>(k, cu, tu, check_first, memo, md, mc);
    default: return recur_luma<2>(k, cu, tu, check_first, memo, md, mc);
  }
}

// The four children of a TU in the second RD pass, when the workgroup has spare waves.  The reference decides child by child
// (xRecurIntraCodingLumaQT :1598-1710): unsplit coding, then the four grandchildren, keep the cheaper, and the next child starts from
// the winner's reconstruction and coder state.  The unsplit coding wins most of the time, so this wave codes the CHAIN of unsplit
// children (picture, own layers and coder state advance exactly as if each had won) while the split alternative of every child runs as a
// task on another wave from the state the chain had when it reached that child.  Then the children are judged in order: as long as the
// unsplit coding wins, the chain WAS the reference's path; the first child whose split wins takes the task's result, and the chain
// restarts behind it.  Every comparison is the reference's (strict <), every coding starts from the state the reference would have.
template <int LOG2> DEVN DistCbf spec_children(KR k, const Cu cu_, const Tu tu_)
{
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_);
  LSmem &s = lds();
  LRegion &r = my_region(1);
  const int sbase = SLOT_SPLIT + SLOT_PSET * uni(k.pset);         // the split tasks' result slots of this pass's set
  uint32_t split_dist = 0, split_cbf = 0;
  unsigned long long split_cfrac = 0;
  int j = 0;
  while (j < 4) {
    wsync();
    if (lane_id() < 4) { r.modes[lane_id()] = j + lane_id(); r.modes[8 + lane_id()] = 0; }          // task i: the split alternative of child j + i; [8 + c]: the chain's answer for child c is there
    if (lane_id() == 0) r.dist[11] = 0;                                 // context copies made by the waves that took split tasks (helper_step)
    if (lane_id() == 0) s.ref_key[0] = -1;                          // a restarted chain meets the same block again with new neighbours
    region_open(r, T_LUMA_SPLIT, 0, cu, tu);
    for (int c = j; c < 4; c++) {
      const Tu ch = tu_child(tu, c);
      state_to_global(slot_state(k.slots, sbase + c, 0), &s.go);
      region_publish(r);
      // the unsplit alternative (the single-TU branch of recur_luma)
      set_parts(k, s.a[A_TSKIP + 0], cu.zbase + ch.zrel, ch.nparts, 0); wsync();
      const TuRes tr = code_tu_block(k, cu, ch, 0, 0, 1);
      const uint32_t d = tr.dist;
      const uint32_t cbf = (uint32_t)(uni(s.a[A_CBF][cu.zbase + ch.zrel]) >> ch.trd) & 1;
      const uint32_t bits = tr.bits;
      const double cost = calc_rd_cost(k, bits, d);
      if (lane_id() == 0) { r.cost[4 + c] = cost; r.dist[4 + c] = d; r.modes[4 + c] = (int)cbf; r.cfrac[4 + c] = s.cfrac_last; }
      wsync();
      wg_release();
      if (lane_id() == 0) __hip_atomic_store(&r.modes[8 + c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // the child's split task may give up against this cost
    }
    { // wait for the split tasks; those nobody has claimed yet this wave runs itself: its own state (everything a task overwrites: kernel context,
      // coder snapshots, attribute arrays) goes to its global scratch and comes back afterwards, the cached reference lines are dropped
      PROF_MARK0();
      constexpr int SAVE_WORDS = (int)(offsetof(RdSmem, line) / 8);
      static_assert(offsetof(RdSmem, line) % 8 == 0 && offsetof(RdSmem, line) <= SAVE_BYTES, "state save area");
      int mine = 0;                                                   // split tasks of this chain run by this wave itself
      while (lds_load(&r.done) < 4 - j) {
        const int idx = region_claim(r);
        if (idx < 0) {
#if HEVCDL_OWNER_LENDS
          // Every split task is taken and this wave would only wait (8.9 % of all wave time on the 600-frame job): it lends itself to the workgroup's masters -- one task
          // of a first-pass / chroma / rough-mode region at a time, its own state parked in its workspace meanwhile, exactly as for a split task of its own.  Not before
          // the waves that took its split tasks have copied its context (helper_step counts the copies in dist[11] of the region).
          const int taken = (int)(lds_load(&r.ticket) & 0xffff) - mine;
          if (lds_load((LDS int *)&r.dist[11]) >= taken) {
            LDS WgShared &sh = wg_shared();
            const int me = wave_id();
            int ran = 0;
            for (int q = 1; q < NW && !ran; q++) {
              LRegion &fr = sh.reg[(me + q) % NW][0];
              const int t = lds_load(&fr.ticket);
              if ((t & 0xffff) >= (int)((unsigned)t >> 16)) continue;
              const int fk = uni(fr.kind);
              if (fk != T_LUMA_P1 && fk != T_CHROMA && fk != T_RMD) continue;
              const int fi = region_claim(fr);
              if (fi < 0) continue;
              wsync();
              for (int i = lane_id(); i < SAVE_WORDS; i += 64) s.my_save[i] = ((LDS const unsigned long long *)&s)[i];
              wg_acquire();
              import_owner(uni(fr.owner));
              run_task<true>(fr, fi);
              wg_release();
              lds_add(&fr.done, 1);
              wsync();
              for (int i = lane_id(); i < SAVE_WORDS; i += 64) ((LDS unsigned long long *)&s)[i] = s.my_save[i];
              wsync();
              if (lane_id() < 3) s.ref_key[lane_id()] = -1;
              if (lane_id() == 0) s.fline_key = -1;
              wsync();
              ran = 1;
            }
            if (ran) continue;
          }
#endif
          __builtin_amdgcn_s_sleep(2); continue;
        }
        mine++;
        wsync();
        { PROF_GLUE_T0(); for (int i = lane_id(); i < SAVE_WORDS; i += 64) s.my_save[i] = ((LDS const unsigned long long *)&s)[i]; wsync(); PROF_GLUE(4); }
        run_task<true>(r, idx);
        wsync();
        { PROF_GLUE_T0(); for (int i = lane_id(); i < SAVE_WORDS; i += 64) ((LDS unsigned long long *)&s)[i] = s.my_save[i]; wsync(); PROF_GLUE(4); }
        wsync();
        if (lane_id() < 3) s.ref_key[lane_id()] = -1;
        if (lane_id() == 0) s.fline_key = -1;
        wg_release();
        lds_add(&r.done, 1);
      }
      wg_acquire();
      PROF_MARK(37);
    }
    wsync();
    int brk = -1;
    for (int c = j; c < 4 && brk < 0; c++) if (ub(r.cost[c - j] < r.cost[4 + c])) brk = c;
    const int last = brk < 0 ? 4 : brk;
    for (int c = j; c < last; c++) { split_dist += (uint32_t)uni((int)r.dist[4 + c]); split_cbf |= (uint32_t)uni(r.modes[4 + c]); split_cfrac += uni64(r.cfrac[4 + c]); }
    if (brk >= 0) { // the split of child brk wins: its arrays, levels, reconstruction and end state replace the chain's
      const Tu ch = tu_child(tu, brk);
      const int zc = cu.zbase + ch.zrel, n = 1 << (LOG2 - 1);
      GLB const uint8_t *at = slot_attr(k.slots, sbase + brk);
      // the split task's slot has the child as its origin; this wave's own set the PU
      GLB const int16_t *sc = slot_coef(k.slots, sbase + brk) + lay_coef_o(zc * 16, LOG2 - 2, 0, zc); GLB int16_t *dc = k.coef_l + lay_coef(k, LOG2 - 2, 0, zc);
      GLB const pel_t *sr = slot_rec(k.slots, sbase + brk) + lay_rec_o(ch.x, ch.y, LOG2 - 2, 0, ch.x, ch.y); GLB pel_t *dr = k.rec_l + lay_rec(k, LOG2 - 2, 0, ch.x, ch.y);
      GLB pel_t *rp = k.rec[0] + (size_t)ch.y * k.W + ch.x;
      wsync();
      for (int i = lane_id(); i < ch.nparts; i += 64) { s.a[A_TRIDX][zc + i] = at[i]; s.a[A_CBF][zc + i] = at[256 + i]; s.a[A_TSKIP][zc + i] = at[512 + i]; }
      for (int i = lane_id(); i < n * n; i += 64) {
        dc[i] = sc[i];
        const int o = (i >> (LOG2 - 1)) * 64 + (i & (n - 1)); const pel_t v = sr[o];
        dr[o] = v; rp[(size_t)(i >> (LOG2 - 1)) * k.W + (i & (n - 1))] = v;
      }
      state_from_global(&s.go, slot_state(k.slots, sbase + brk, 1));
      split_dist += (uint32_t)uni((int)r.dist[brk - j]);
      split_cfrac += uni64(r.cfrac[brk - j]);
      split_cbf |= (uint32_t)(uni(s.a[A_CBF][zc]) >> ch.trd) & 1;
    }
    j = brk < 0 ? 4 : brk + 1;
  }
  const DistCbf res = { split_dist, split_cbf, split_cfrac };
  return res;
}

// rough mode decision for one PU: 35 predictions + SATD (TEncSearch.cpp:2266-2346).  One lane per
// (mode, 8x8 block) task (4x4 blocks for a 4x4 PU): predict the block in registers, Hadamard, add into satd[mode].
// Angular modes are evaluated row by row in the mode's own orientation (the horizontal family on the transposed
// block: the sum of absolute Hadamard coefficients is transpose-invariant), so the 8 samples of a row share one
// (iIdx, iFact) pair and 9 consecutive reference samples (TComPrediction.cpp:731-817).
template <int B> DEV unsigned rmd_block(KR k, const LSmem &s, int mode, int pn, int log2n, int x, int y, int bx, int by, int dcv)
{
  constexpr int PPW = 4 / (int)sizeof(pel_t), NW = B / PPW;   // samples per dword, dwords per row of the block
  const int W = k.W, n2 = 2 * pn;
  GLB const uint32_t *org = (GLB const uint32_t *)(k.org[0] + (size_t)(y + by) * W + x + bx);
  uint32_t o[B][NW];
#pragma unroll
  for (int r = 0; r < B; r++)
#pragma unroll
    for (int w = 0; w < NW; w++) o[r][w] = org[((size_t)r * W / PPW) + w];
  auto opix = [&](int r, int c) -> int { return (int)((o[r][c / PPW] >> (8 * (int)sizeof(pel_t) * (c % PPW))) & (unsigned)((1u << (8 * sizeof(pel_t))) - 1u)); };
  LDS const int16_t *line = use_filtered_refs(0, mode, pn) ? s.fline : s.line;
  int m[B * B];
  if (mode >= 2) {
    const int is_ver = mode >= 18, sgn = is_ver ? 1 : -1;
    const int ang_mode = is_ver ? mode - VER : -(mode - HOR);
    const int abs_ang = abs(ang_mode);
    const int angle = (ang_mode < 0 ? -1 : 1) * tb().t_ang[abs_ang], inv_angle = tb().t_inv_ang[abs_ang];
    const int x0 = is_ver ? bx : by, y0 = is_ver ? by : bx;   // block origin in the mode's orientation
    const int edge = (angle == 0) && (pn <= 16) && (x0 == 0);
    const int s0 = line[n2];
#pragma unroll
    for (int r = 0; r < B; r++) {
      const int dpos = (y0 + r + 1) * angle, di = dpos >> 5, df = dpos & 31;
      int R[B + 1];
#pragma unroll
      for (int t = 0; t <= B; t++) {
        const int i = x0 + t + di + 1;
        const int off = (i >= 0) ? i : -((128 + (-i) * inv_angle) >> 8);
        R[t] = line[n2 + sgn * off];
      }
      int d[B];
#pragma unroll
      for (int c = 0; c < B; c++) {
        int v = ((32 - df) * R[c] + df * R[c + 1] + 16) >> 5;
        if (c == 0 && edge) v = clip8(v + (((int)line[n2 - sgn * (y0 + r + 1)] - s0) >> 1));
        d[c] = (is_ver ? opix(r, c) : opix(c, r)) - v;
      }
      if (B == 8) {
        int e[8];
        e[0] = d[0] + d[4]; e[1] = d[1] + d[5]; e[2] = d[2] + d[6]; e[3] = d[3] + d[7]; e[4] = d[0] - d[4]; e[5] = d[1] - d[5]; e[6] = d[2] - d[6]; e[7] = d[3] - d[7];
        d[0] = e[0] + e[2]; d[1] = e[1] + e[3]; d[2] = e[0] - e[2]; d[3] = e[1] - e[3]; d[4] = e[4] + e[6]; d[5] = e[5] + e[7]; d[6] = e[4] - e[6]; d[7] = e[5] - e[7];
        m[r * B + 0] = d[0] + d[1]; m[r * B + 1] = d[0] - d[1]; m[r * B + 2] = d[2] + d[3]; m[r * B + 3] = d[2] - d[3];
        m[r * B + 4] = d[4] + d[5]; m[r * B + 5] = d[4] - d[5]; m[r * B + 6] = d[6] + d[7]; m[r * B + 7] = d[6] - d[7];
      } else {
        const int a0 = d[0] + d[3], a1 = d[1] + d[2], a2 = d[1] - d[2], a3 = d[0] - d[3];
        m[r * B] = a0 + a1; m[r * B + 1] = a0 - a1; m[r * B + 2] = a2 + a3; m[r * B + 3] = a3 - a2;
      }
    }
  } else {
    // planar / DC (TComPrediction.cpp:183-201, 403-473)
    int top[B];
#pragma unroll
    for (int c = 0; c < B; c++) top[c] = line[n2 + 1 + bx + c];
    const int bl = line[n2 - 1 - pn], tr = line[n2 + 1 + pn];
#pragma unroll
    for (int r = 0; r < B; r++) {
      const int py = by + r, left = line[n2 - 1 - py];
      int d[B];
#pragma unroll
      for (int c = 0; c < B; c++) {
        const int px = bx + c;
        int v;
        if (mode == PLANAR) v = ((pn - 1 - px) * left + (px + 1) * tr + (pn - 1 - py) * top[c] + (py + 1) * bl + pn) >> (log2n + 1);
        else {
          v = dcv;
          if (pn <= 16) {
            if (px == 0 && py == 0) v = (top[0] + left + 2 * dcv + 2) >> 2;
            else if (py == 0) v = (top[c] + 3 * dcv + 2) >> 2;
            else if (px == 0) v = (left + 3 * dcv + 2) >> 2;
          }
        }
        d[c] = opix(r, c) - v;
      }
      if (B == 8) {
        int e[8];
        e[0] = d[0] + d[4]; e[1] = d[1] + d[5]; e[2] = d[2] + d[6]; e[3] = d[3] + d[7]; e[4] = d[0] - d[4]; e[5] = d[1] - d[5]; e[6] = d[2] - d[6]; e[7] = d[3] - d[7];
        d[0] = e[0] + e[2]; d[1] = e[1] + e[3]; d[2] = e[0] - e[2]; d[3] = e[1] - e[3]; d[4] = e[4] + e[6]; d[5] = e[5] + e[7]; d[6] = e[4] - e[6]; d[7] = e[5] - e[7];
        m[r * B + 0] = d[0] + d[1]; m[r * B + 1] = d[0] - d[1]; m[r * B + 2] = d[2] + d[3]; m[r * B + 3] = d[2] - d[3];
        m[r * B + 4] = d[4] + d[5]; m[r * B + 5] = d[4] - d[5]; m[r * B + 6] = d[6] + d[7]; m[r * B + 7] = d[6] - d[7];
      } else {
        const int a0 = d[0] + d[3], a1 = d[1] + d[2], a2 = d[1] - d[2], a3 = d[0] - d[3];
        m[r * B] = a0 + a1; m[r * B + 1] = a0 - a1; m[r * B + 2] = a2 + a3; m[r * B + 3] = a3 - a2;
      }
    }
  }
  // second (column) pass of the Hadamard transform (TComRdCost.cpp:1561-1750; |coefficients| are order independent)
  unsigned sum = 0;
  if (B == 8) {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      int e[8], d[8];
      e[0] = m[c] + m[32 + c]; e[1] = m[8 + c] + m[40 + c]; e[2] = m[16 + c] + m[48 + c]; e[3] = m[24 + c] + m[56 + c];
      e[4] = m[c] - m[32 + c]; e[5] = m[8 + c] - m[40 + c]; e[6] = m[16 + c] - m[48 + c]; e[7] = m[24 + c] - m[56 + c];
      d[0] = e[0] + e[2]; d[1] = e[1] + e[3]; d[2] = e[0] - e[2]; d[3] = e[1] - e[3]; d[4] = e[4] + e[6]; d[5] = e[5] + e[7]; d[6] = e[4] - e[6]; d[7] = e[5] - e[7];
      sum += (unsigned)(abs(d[0] + d[1]) + abs(d[0] - d[1]) + abs(d[2] + d[3]) + abs(d[2] - d[3]) + abs(d[4] + d[5]) + abs(d[4] - d[5]) + abs(d[6] + d[7]) + abs(d[6] - d[7]));
    }
    return (sum + 2) >> 2;
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int a0 = m[c] + m[12 + c], a1 = m[4 + c] + m[8 + c], a2 = m[4 + c] - m[8 + c], a3 = m[c] - m[12 + c];
      sum += (unsigned)(abs(a0 + a1) + abs(a0 - a1) + abs(a2 + a3) + abs(a3 - a2));
    }
    return (sum + 1) >> 1;
  }
}
// rounds [r0, r1) of the rough mode decision's (mode, block) tasks, 64 per round, of the PU at (x, y); the SATD sums go to acc.satd (the
// owner's: integer atomics, any order).  The reference lines are the executing wave's (a helper copies the owner's first).
DEVN void rmd_rounds(KR k, LDS unsigned int *satd_dst, int x_, int y_, int pn_, int dcv_, int r0_, int r1_)
{
  const int x = uni(x_), y = uni(y_), pn = uni(pn_), dcv = uni(dcv_), r0 = uni(r0_), r1 = uni(r1_);
  LSmem &s = lds();
  const int log2n = ilog2(pn), b = pn >= 8 ? 8 : 4, lb = pn >= 8 ? 3 : 2, lnbx = log2n - lb, nbx = 1 << lnbx, nblk = nbx * nbx, ntask = 35 * nblk;
#pragma unroll 1
  for (int t0 = r0 * 64; t0 < r1 * 64 && t0 < ntask; t0 += 64) {
    const int t = t0 + lane_id();
    if (t < ntask) {
      const int mode = t >> (2 * lnbx), blk = t & (nblk - 1), bx = (blk & (nbx - 1)) << lb, by = (blk >> lnbx) << lb;
      const unsigned sum = (b == 8) ? rmd_block<8>(k, s, mode, pn, log2n, x, y, bx, by, dcv) : rmd_block<4>(k, s, mode, pn, log2n, x, y, bx, by, dcv);
      __hip_atomic_fetch_add(&satd_dst[mode], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  wsync();
}
DEVN void rmd_satd(KR k, const Cu cu_, const Tu ptu_)
{
  PROF_T0();
  const Cu cu = ucu(cu_); const Tu ptu = utu(ptu_);
  const int x = ptu.x, y = ptu.y, pn = 1 << ptu.log2;
  LSmem &s = lds();
  const int b = pn >= 8 ? 8 : 4, nbx = pn >> (pn >= 8 ? 3 : 2), ntask = 35 * nbx * nbx, nrounds = (ntask + 63) >> 6;
  if (lane_id() < 36) s.satd[lane_id()] = 0;
  const int dcv = dc_value(k, s.line, pn);
  wsync();
  if (nrounds >= HEVCDL_RMD_SLICE_ROUNDS && lds_load(&wg_shared().masters_active) < NW) {
    // 16x16 PUs and larger (3 / 9 / 35 rounds) with waves without a unit in the workgroup: the rounds are dealt to them in up to NW slices
    LRegion &r = my_region();
    const int ntasks = nrounds < NW ? nrounds : NW;
    wsync();
    if (lane_id() == 0) { r.modes[0] = dcv; r.modes[1] = nrounds; r.modes[2] = 0; }
    region_open(r, T_RMD, ntasks, cu, ptu);
    region_run(k, r);
  } else rmd_rounds(k, s.satd, x, y, pn, dcv, 0, nrounds);
  PROF_ADD(k, 2);
}

// The CU that follows `cu` in the CTU's walk, if it starts inside this CTU and the picture: the walk reads one label per CU (compress_cu)
DEV bool next_leaf(KR k, const Cu &cu, int &nx, int &ny, int &nlog2)
{
  const int z = cu.zbase + cu.nparts;
  if (z >= 256) return false;
  int px = 0, py = 0;
  for (int b = 0; b < 4; b++) { px |= ((z >> (2 * b)) & 1) << b; py |= ((z >> (2 * b + 1)) & 1) << b; }
  const int x = k.cx * 64 + 4 * px, y = k.cy * 64 + 4 * py;
  if (x >= k.W || y >= k.H) return false;
  for (int d = 0; d <= 3; d++) {
    const int size = 64 >> d, ox = x & ~(size - 1), oy = y & ~(size - 1);
    const int straddles = ox + size > k.W || oy + size > k.H;
    const int l = uni(lds().lab16[4 * ((oy & 63) / 16) + (ox & 63) / 16]);
    if (!straddles && l == d) { if (ox != x || oy != y) return false; nx = x; ny = y; nlog2 = 6 - d; return true; }
    if (!(straddles || l > d)) return false;
  }
  return false;
}
// The first CU of the CTU at column / row (ncx, ncy), address na -- the same walk over its labels (read from HBM: the CTU is not the one being coded)
DEV bool ctu_first_leaf(KR k, int na, int ncx, int ncy, int &nx, int &ny, int &nlog2)
{
  const int x = ncx * 64, y = ncy * 64;
  if (x >= k.W || y >= k.H) return false;
  const int l = uni(k.labels[na * 16]);                   // the top-left cell's label is the one every depth of the walk reads there
  for (int d = 0; d <= 3; d++) {
    const int size = 64 >> d;
    const int straddles = x + size > k.W || y + size > k.H;
    if (!straddles && l == d) { nx = x; ny = y; nlog2 = 6 - d; return true; }
    if (!(straddles || l > d)) return false;
  }
  return false;
}
// Rough-mode SATD sums of the PU at (x, y) ahead of time: they depend on the reconstruction around the PU only (final once the CU before it has its first
// pass's winner: a pending second pass is read through best_rec), not on the coder state -- est_intra_luma adds the mode bits when it gets there.
DEVN void rmd_prefetch(KR k, int x_, int y_, int log2_, int sliced_ = 0)
{
  const int x = uni(x_), y = uni(y_), log2 = uni(log2_), pn = 1 << log2, sliced = uni(sliced_);
  LSmem &s = lds();
  const int nbx = pn / 8, nrounds = (35 * nbx * nbx + 63) >> 6;
  if (sliced == 2 && nrounds >= HEVCDL_RMD_SLICE_ROUNDS) { // open only: every slice gathers the PU's reference lines itself (rmd_prefetch_end gathers this wave's)
    LRegion &r = my_region();
    const int ntasks = nrounds < NW ? nrounds : NW;
    const Cu ncu = { x, y, log2, 6 - log2, 0, 1 << (2 * (log2 - 2)), SIZE_2Nx2N }; const Tu ptu = { x, y, log2, 0, 0, 1 << (2 * (log2 - 2)) };
    wsync();
    if (lane_id() < 36) s.satd_pre[NPEND >= 2 ? lane_id() : 0] = 0;
    if (lane_id() == 0) { r.modes[0] = 0; r.modes[1] = nrounds; r.modes[2] = 2; s.pre_key = -1; }
    region_open(r, T_RMD, ntasks, ncu, ptu);
    return;
  }
  build_refs(k, 0, x, y, pn, 1);
  filter_refs(k, pn);
  if (lane_id() < 36) s.satd_pre[NPEND >= 2 ? lane_id() : 0] = 0;
  const int dcv = dc_value(k, s.line, pn);
  wsync();
  if (sliced && nrounds >= HEVCDL_RMD_SLICE_ROUNDS) { // the workgroup's other waves are free (the caller's ticket region too): the rounds in slices, as rmd_satd deals them
    LRegion &r = my_region();
    const int ntasks = nrounds < NW ? nrounds : NW;
    const Cu ncu = { x, y, log2, 6 - log2, 0, 1 << (2 * (log2 - 2)), SIZE_2Nx2N }; const Tu ptu = { x, y, log2, 0, 0, 1 << (2 * (log2 - 2)) };
    wsync();
    if (lane_id() == 0) { r.modes[0] = dcv; r.modes[1] = nrounds; r.modes[2] = 1; }      // [2]: the sums go to satd_pre
    region_open(r, T_RMD, ntasks, ncu, ptu);
    region_run(k, r);
  } else rmd_rounds(k, s.satd_pre, x, y, pn, dcv, 0, nrounds);
  if (lane_id() == 0) s.pre_key = (log2 << 24) | (y << 12) | x;
  wsync();
}
DEVN void rmd_prefetch_end(KR k, int x_, int y_, int log2_)
{ // behind rmd_prefetch(..., 2): this wave takes what slices are left and waits for the others (no-op when the rounds were not sliced: the key is set)
  const int x = uni(x_), y = uni(y_), log2 = uni(log2_);
  LSmem &s = lds();
  if (uni(s.pre_key) == ((log2 << 24) | (y << 12) | x)) return;
  build_refs(k, 0, x, y, 1 << log2, 1);                    // this wave's own lines: the candidates coded ahead copy them, est_intra_luma finds them in place
  filter_refs(k, 1 << log2);
  region_run(k, my_region());
  build_refs(k, 0, x, y, 1 << log2, 0);                    // (a slice run by this wave has gathered them again: same lines; nothing to do when the key matches)
  filter_refs(k, 1 << log2);
  if (lane_id() == 0) s.pre_key = (log2 << 24) | (y << 12) | x;
  wsync();
}

// Join the oldest second pass left pending (compress_cu).  Verdict "nothing changes": its CU is final as the walk assumed (the pass itself has put the
// first pass's reconstruction back into the picture).  The split won: everything coded since stands on the wrong reconstruction and coder state -> the
// other pending pass is waited for (its slots and the picture must be quiet) and the CTU is walked again from the log (process_unit).
DEV void copy_best_rec_to_pic(KR k, const Cu &cu, int comp);
DEV int free_pend_reg()
{ // a ticket region 1..NPEND (= slot set + 1 = srect entry + 1) that no pending pass holds
  LSmem &s = lds();
  unsigned used = 0;
  const int n = uni(s.pend_n);
  for (int i = 0; i < NPEND; i++) if (i < n) used |= 1u << uni(s.pend_reg[i]);
  for (int r = 1; r <= NPEND; r++) if (!((used >> r) & 1u)) return r;
  return 1;
}
DEVN void pend_join_oldest(KR k, int site = 0)
{
  PROF_T0();
  LSmem &s = lds(); LDS K &kk = s.k;
  const int reg = uni(s.pend_reg[0]), leaf = uni(s.pend_leaf[0]), n = uni(s.pend_n);
  LRegion &rp = my_region(reg);
  region_wait(rp, 1);
  wsync();
  const int won = ub(rp.cost[0] < rp.cost[4]);
  if (!won && uni(rp.kind) == T_REMOTE) { // another workgroup ran the pass: the first pass's samples are written back here, by the waves that will read them (remote_poll)
    const Cu pc = { uni(rp.cu[0]), uni(rp.cu[1]), uni(rp.cu[2]), uni(rp.cu[3]), uni(rp.cu[4]), uni(rp.cu[5]), uni(rp.cu[6]) };
    copy_best_rec_to_pic(k, pc, 0);
  }
  if (won) for (int o = 1; o < n; o++) region_wait(my_region(uni(s.pend_reg[o])), 1);
  wsync();
  if (lane_id() == 0) {
    kk.srect[reg - 1] = 0;
    if (won) { kk.srect[0] = 0; kk.srect[1] = 0; kk.srect[2] = 0; s.restart = 1; s.replay_upto = leaf; s.nocarry_leaf = leaf; s.resume_reg = reg; s.pend_n = 0; }
    else { s.pend_leaf[0] = s.pend_leaf[1]; s.pend_reg[0] = s.pend_reg[1]; s.pend_leaf[1] = s.pend_leaf[2]; s.pend_reg[1] = s.pend_reg[2]; s.pend_n = n - 1; }
  }
  wsync();
#ifdef HEVCDL_KERNEL_PROF
  if (uni(site) == 0) PROF_ADD(k, 44); else if (uni(site) == 1) PROF_ADD(k, 54); else PROF_ADD(k, 55);
#endif
}

// Candidates of the first RD pass: rough-mode costs = SATD + mode bits (xModeBitsIntra :5530-5557: 3 possible values, from fractional bits f0 and the state st of
// the prev_intra_luma_pred context), the c_num_rd_cand best in xUpdateCandList's order, then the most probable modes not among them (:2320-2350).  -> s.rd_list, count
template <bool TRACE> DEV int rmd_candidates(KR k, int x, int y, int pu_log2, unsigned long long f0, int st, LDS const unsigned int *satd)
{
  LSmem &s = lds();
  int preds[3], nm; get_mpm(k, x, y, preds, &nm);
  const int use_mpm = tools_of(k) & (int)HEVCDL_TOOL_FAST_UDI_MPM;       // FastUDIUseMPMEnabled: the list is the c_num_rd_cand best + the most probable modes; without it one more and no additions (TEncSearch.cpp:2269, 2322)
  int nfull = use_mpm ? c_num_rd_cand[pu_log2 - 2] : c_num_rd_cand_no_mpm[pu_log2 - 2];
  if (!use_mpm) nm = 0;
  if (lane_id() < 35) {
    const int mode = lane_id();
    int idx = -1; for (int i = 0; i < 3; i++) if (mode == preds[i]) idx = i;
    const unsigned long long fr = f0 + (unsigned long long)tb().t_ebits[st ^ (idx != -1)] + 32768ull * (unsigned long long)(idx != -1 ? (idx ? 2 : 1) : 5);
    s.rmd_cost[mode] = (double)(satd[mode] >> HAD_SH) + (double)(uint32_t)(fr >> 15) * k.sqrt_lambda;
  }
#ifdef HEVCDL_STAGE_TRACE
  if (TRACE) {  // "1st pass mode" lines, TEncSearch.cpp:2315-2317
    const bool on = lane_id() < 35; const int mode = on ? lane_id() : 0; int idx = -1; for (int i = 0; i < 3; i++) if (mode == preds[i]) idx = i;
    const unsigned long long fr = f0 + (unsigned long long)tb().t_ebits[st ^ (idx != -1)] + 32768ull * (unsigned long long)(idx != -1 ? (idx ? 2 : 1) : 5);
    stage_line(k, 0, mode, satd[mode] >> HAD_SH, (unsigned)(fr >> 15), (double)(satd[mode] >> HAD_SH) + (double)(uint32_t)(fr >> 15) * k.sqrt_lambda, on);
  }
#endif
  wsync();
  // xUpdateCandList :5562-5585 == stable sort by (cost, mode); keep the nfull best
  if (lane_id() < 35) {
    const double mc = s.rmd_cost[lane_id()]; int rank = 0;
    for (int j = 0; j < 35; j++) { const double oc = s.rmd_cost[j]; rank += (oc < mc) || (oc == mc && j < lane_id()); }
    if (rank < nfull) s.rd_list[rank] = (unsigned)lane_id();
  }
  wsync();
  if (lane_id() == 0) {
    int nf = nfull;
    for (int j = 0; j < nm; j++) { int inc = 0; for (int i = 0; i < nf; i++) inc |= (preds[j] == (int)s.rd_list[i]); if (!inc) s.rd_list[nf++] = (unsigned)preds[j]; }
    s.bc_u32[1] = (unsigned)nf;
  }
  wsync();
  return uni((int)s.bc_u32[1]);
}

// ---- look-ahead: the next CU's first RD pass during this CU's chroma search -------------------------------------------------------------
// What a first-pass candidate of a one-TU PU needs: the reconstruction around the PU (final once the CU before it has its first pass's winner; a pending second
// pass is read through best_rec), the neighbours' modes, and of the coder state the LUMA-SIDE contexts (part size, prev_intra_luma_pred, transform subdivision, luma
// cbf, luma coefficient contexts) -- the chroma search and the chroma syntax of the CU before touch none of them (chroma has its own contexts), so they are what the
// first pass's winner of that CU left behind (entry 65 of the log).  The fractional bits the count starts from (15 bits) do depend on the chroma syntax: a candidate
// coded ahead of time starts from a guess, and its bit count is corrected when the real value is known (est_intra_luma): bits = (f0 + S) >> 15 with S the candidate's
// own sum, which is in the end state it stored.  Candidate LIST: rough-mode costs with the guessed f0; if the real f0 gives another list, or any luma-side context
// differs after all, the work is dropped.  Claims end when the chroma search does (the master's context is quiet only while it sits in region_run).
#ifndef HEVCDL_AHEAD_MAX
#define HEVCDL_AHEAD_MAX 1                      // measured: with two or three masters per workgroup the candidates coded ahead only take waves from work that is needed now (600 frames: 6.45 -> 6.70 s at 3)
#endif
// xaddr >= 0: the PU is the first CU of the NEXT CTU (address xaddr, column / row xcx / xcy), opened while this CTU is finished (process_unit): the candidates then take
// over a context that names that CTU and attribute arrays as initCtu leaves them (this wave's own stay untouched: the CTU may still be walked again).
DEVN void ahead_open(KR k, const Cu cu_, int nx_, int ny_, int nl_, int xaddr_ = -1, int xcx_ = 0, int xcy_ = 0)
{
  const Cu cu = ucu(cu_); const int nx = uni(nx_), ny = uni(ny_), nl = uni(nl_), xaddr = uni(xaddr_), xcx = uni(xcx_), xcy = uni(xcy_);
  LSmem &s = lds();
  LRegion &r = my_region(REG_AHEAD);
  const int depth = 6 - nl, nparts = 1 << (2 * (nl - 2)), zb = xaddr >= 0 ? 0 : cu.zbase + cu.nparts;
  const Cu ncu = { nx, ny, nl, depth, zb, nparts, SIZE_2Nx2N };
  const Tu ptu = { nx, ny, nl, 0, 0, nparts };
  wsync();
  if (xaddr < 0) for (int i = lane_id(); i < nparts; i += 64) { // initEstData of the next CU (check_rd_cost_intra does it again): the candidates' tasks copy these arrays
    const int z = zb + i;
    s.a[A_DEPTH][z] = (uint8_t)depth; s.a[A_PART][z] = (uint8_t)SIZE_2Nx2N; s.a[A_LDIR][z] = DC; s.a[A_CDIR][z] = 0; s.a[A_TRIDX][z] = 0;
    for (int c = 0; c < 3; c++) { s.a[A_CBF + c][z] = 0; s.a[A_TSKIP + c][z] = 0; }
  }
  wsync();
  { // the context the candidates' waves take over (import_ahead): this wave's own keeps changing while the CU is finished
    static_assert(sizeof(K) + 11 * 256 <= LOG_AHEAD_N * LEAF_LOG, "look-ahead snapshot");
    GLB unsigned long long *d = s.my_log + LOG_AHEAD * (LEAF_LOG / 8);
    LDS const unsigned long long *qk = (LDS const unsigned long long *)&s.k, *qa = (LDS const unsigned long long *)&s.a[0][0];
    static_assert(offsetof(K, addr) == 16 && offsetof(K, cx) == 20 && offsetof(K, cy) == 24 && offsetof(K, nctu) == 28, "K: words 2 and 3 hold addr | cx, cy | nctu");
    for (int i = lane_id(); i < (int)(sizeof(K) / 8); i += 64) {
      unsigned long long w = qk[i];
      if (xaddr >= 0 && i == 2) w = (unsigned long long)(unsigned)xaddr | ((unsigned long long)(unsigned)xcx << 32);
      if (xaddr >= 0 && i == 3) w = (w & 0xffffffff00000000ull) | (unsigned long long)(unsigned)xcy;
      d[i] = w;
    }
    if (xaddr < 0) { for (int i = lane_id(); i < 11 * 256 / 8; i += 64) d[sizeof(K) / 8 + i] = qa[i]; }
    else for (int i = lane_id(); i < 11 * 256 / 8; i += 64) { // initCtu (process_unit) + initEstData of the first CU: eight partitions per word
      const int f = i >> 5, z = (i & 31) * 8;
      const unsigned b = z < nparts ? (f == A_DEPTH ? (unsigned)depth : (f == A_LDIR ? (unsigned)DC : 0u)) : (f == A_PART ? (unsigned)SIZE_NONE : (f == A_LDIR ? (unsigned)DC : 0u));
      d[sizeof(K) / 8 + i] = 0x0101010101010101ull * (unsigned long long)b;
    }
  }
  GLB const unsigned long long *sp = s.my_log + 65 * (LEAF_LOG / 8);
  const unsigned long long f0 = uni64(sp[offsetof(Cabac, frac) / 8]) & 32767ull;
  const int st = uni((int)((GLB const uint8_t *)sp)[CTX_INTRA_PRED]);
  const int nfull = rmd_candidates<false>(k, nx, ny, nl, f0, st, s.satd_pre);
  const int n = nfull < 5 ? nfull : 5;                     // result slots 0..4 (5..9 hold this CU's chroma modes)
  wsync();
  if (lane_id() < n) r.modes[lane_id()] = (int)s.rd_list[lane_id()];
  if (lane_id() == 0) { s.ahead_open = 1; s.ahead_key = (nl << 24) | (ny << 12) | nx; s.ahead_n = n; s.ahead_claimed = 0; s.ahead_f0 = (unsigned)f0; r.pad_ = 0x7fffffff; }
  region_open(r, T_LUMA_AHEAD, n, ncu, ptu);
#if defined(HEVCDL_KERNEL_PROF) && HEVCDL_PROF_N == 64 && !defined(HEVCDL_PROF_MASTER) && !defined(HEVCDL_PROF_GLUE)   // look-ahead statistics on the (then silent) RDOQ accumulators
  if (lane_id() == 0) PROF_ACC_(4, (unsigned long long)n << 10);
#endif
}
DEV void ahead_freeze()
{ // no further claims: count := next
  LSmem &s = lds();
  if (!AHEAD || uni(s.ahead_open) != 1) return;
  LRegion &r = my_region(REG_AHEAD);
  int c = 0;
  if (lane_id() == 0) {
    int t = __hip_atomic_load(&r.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) { c = t & 0xffff; if (__hip_atomic_compare_exchange_strong(&r.ticket, &t, (c << 16) | c, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break; }
    s.ahead_claimed = c; s.ahead_open = 2;
  }
  wsync();
}
DEVN void ahead_drain()
{ // drop the look-ahead: the tasks under way still write their slots and their region
  LSmem &s = lds();
  if (!AHEAD || !uni(s.ahead_open)) return;
  ahead_freeze();
  LRegion &r = my_region(REG_AHEAD);
  const int c = uni(s.ahead_claimed);
  while (lds_load(&r.done) < c) __builtin_amdgcn_s_sleep(2);
  wg_acquire();
  wsync();
  if (lane_id() == 0) { s.ahead_open = 0; s.ahead_key = -1; }
  wsync();
}

// estIntraPredLumaQT TEncSearch.cpp:2203-2582
DEVN uint32_t est_intra_luma(KR k, const Cu cu_)
{
  CHECK_EXEC(5);
  PROF_T0();
  const Cu cu = ucu(cu_);
  LSmem &s = lds();
  const int init_trd = cu.part == SIZE_NxN ? 1 : 0, npu = init_trd ? 4 : 1;
  const int pu_log2 = cu.log2 - init_trd, pn = 1 << pu_log2, pu_parts = cu.nparts >> (2 * init_trd);
  uint32_t overall = 0;
  MT0();
  TL(1, cu.zbase);
  for (int pu = 0; pu < npu; pu++) {
    const int poff = pu * pu_parts, zp = cu.zbase + poff;
    const Tu ptu = { cu.x + (pu & 1) * pn * init_trd, cu.y + (pu >> 1) * pn * init_trd, pu_log2, init_trd, poff, pu_parts };
    { LDS K &kk = s.k; wsync(); kk.lz = zp * 16; kk.lx = ptu.x; kk.ly = ptu.y; wsync(); }       // this wave's own layer set serves the PU (second pass)
    // ---- rough mode decision ----
    const int rkey = (pu_log2 << 24) | (ptu.y << 12) | ptu.x;
    if (NPEND >= 2 && npu == 1 && uni(s.pre_key) == rkey) { // the SATD sums were computed ahead (rmd_prefetch); the gathered lines are still in place
      wsync();
      if (lane_id() < 36) s.satd[lane_id()] = s.satd_pre[NPEND >= 2 ? lane_id() : 0];
      if (lane_id() == 0) { s.ref_key[0] = rkey; s.fline_key = rkey; s.pre_key = -1; }
      wsync();
    } else {
      build_refs(k, 0, ptu.x, ptu.y, pn, 1);
      if (pn >= 8 && pn <= 32) filter_refs(k, pn);
      rmd_satd(k, cu, ptu);
    }
    int nfull;
    { const LCabac *cur = &s.curr[cu.depth]; nfull = rmd_candidates<true>(k, ptu.x, ptu.y, pu_log2, cur->frac & 32767ull, cur->ctx[CTX_INTRA_PRED], s.satd); }   // from the [depth][CI_CURR_BEST] snapshot
    // ---- RD pass 1 (:2355-2443): the candidates are independent (each starts from the [depth][CI_CURR_BEST] snapshot) -> a region ----
    uint32_t best_mode = 0, best_dist = 0; double best_cost = MAX_DOUBLE;
    int early_reg = 0;                                               // the next PU's SATD rounds were opened right behind the first pass's verdict: the ticket region of this CU's second pass
    {
      // candidates coded ahead of time (ahead_open)?  Accepted when they are the PU's candidates in the same order and every context they read is what it was
      // assumed to be; the tasks nobody had claimed when the chroma search of the CU before ended are run now
      bool use_ahead = false;
      if (AHEAD && uni(s.ahead_open)) {
        ahead_freeze();                                                  // claims have gone on while the CU before was finished
        LRegion &ra = my_region(REG_AHEAD);
        const int n_s = uni(s.ahead_n), c = uni(s.ahead_claimed);
        bool ok = npu == 1 && uni(s.ahead_key) == rkey && c > 0 && n_s == (nfull < 5 ? nfull : 5);
        if (ok) ok = __ballot(lane_id() < n_s && ra.modes[lane_id() < n_s ? lane_id() : 0] != (int)s.rd_list[lane_id() < n_s ? lane_id() : 0]) == 0ull;
        if (ok) {
          GLB const uint8_t *sp = (GLB const uint8_t *)(s.my_log + 65 * (LEAF_LOG / 8)); const LCabac *cur = &s.curr[cu.depth];
          bool diff = false;
#pragma unroll
          for (int t = 0; t < 3; t++) {
            const int i = lane_id() + 64 * t;
            const bool luma_side = i == CTX_PART_SIZE || i == CTX_INTRA_PRED || (i >= CTX_QT_CBF && i < CTX_QT_CBF + 5) || (i >= CTX_SUBDIV && i < CTX_SUBDIV + 3) || i == 19 || i == 20 ||
                                   (i >= CTX_SIG && i < CTX_SIG + 28) || (i >= CTX_LAST_X && i < CTX_LAST_X + 15) || (i >= CTX_LAST_Y && i < CTX_LAST_Y + 15) ||
                                   (i >= CTX_ONE && i < CTX_ONE + 16) || (i >= CTX_ABS && i < CTX_ABS + 4) || i == CTX_TSKIP;
            if (luma_side && sp[i] != cur->ctx[i]) diff = true;
          }
          ok = __ballot(diff) == 0ull;
        }
#if defined(HEVCDL_KERNEL_PROF) && HEVCDL_PROF_N == 64 && !defined(HEVCDL_PROF_MASTER) && !defined(HEVCDL_PROF_GLUE)   // look-ahead statistics on the (then silent) RDOQ accumulators
        if (lane_id() == 0) { if (ok) PROF_ACC_(5, (unsigned long long)c << 10); else PROF_ACC_(8, (unsigned long long)((npu == 1 && uni(s.ahead_key) == rkey ? 0 : 1) + (c > 0 ? 0 : 2) + (n_s == (nfull < 5 ? nfull : 5) ? 0 : 4)) << 10); }
#endif
        if (ok) use_ahead = true; else ahead_drain();
      }
      LRegion &r = use_ahead ? my_region(REG_AHEAD) : my_region();
      wsync();
      MT(4);
      TL(2, use_ahead ? uni(s.ahead_claimed) : 0);
      PROF_MARK0();
      if (use_ahead) {
        const int n_s = uni(s.ahead_n), c = uni(s.ahead_claimed);
        if (lane_id() >= n_s && lane_id() < nfull) r.modes[lane_id()] = (int)s.rd_list[lane_id()];
        if (lane_id() == 0) r.pad_ = c;                                // tasks from c on are ordinary candidates: this wave's live context, the real snapshot
        wsync();
        wg_release();
        if (lane_id() == 0) __hip_atomic_store(&r.ticket, (nfull << 16) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (nfull > c) ring_bell();
        region_run(k, r);
        // the bit counts of the candidates coded ahead, from the fractional bits the CU really starts with
        const unsigned long long f0t = s.curr[cu.depth].frac & 32767ull, f0s = (unsigned long long)s.ahead_f0;
        if (lane_id() < c) {
          const unsigned long long sum = slot_state(k.slots, lane_id(), 1)[offsetof(Cabac, frac) / 8] - f0s;
          r.cost[lane_id()] = calc_rd_cost(k, (uint32_t)((f0t + sum) >> 15), r.dist[lane_id()]);
        }
        if (lane_id() == 0) { s.ahead_open = 0; s.ahead_key = -1; }
        wsync();
      } else {
        if (lane_id() < nfull) r.modes[lane_id()] = (int)s.rd_list[lane_id()];
        region_open(r, T_LUMA_P1, nfull, cu, ptu);
        region_run(k, r);
      }
      PROF_MARK(36);
      MT(34);
      TL(3, nfull);
#ifdef HEVCDL_STAGE_TRACE
      { const bool on = lane_id() < nfull; stage_line(k, 1, on ? r.modes[lane_id()] : 0, 0u, 0u, on ? r.cost[lane_id()] : 0.0, on); }   // "2nd pass" lines, :2395-2397
#endif
      // the serial loop keeps a candidate when its cost is strictly smaller: the winner is the smallest cost, first in list order
      int win = -1;
      for (int m = 0; m < nfull; m++) { const double c = r.cost[m]; if (ub(c < best_cost)) { best_cost = c; win = m; } }
      TL(30, win);
      if (win >= 0) { // xSetIntraResultLumaQT + the saved arrays, from the winner's result slot
        best_mode = (uint32_t)uni(r.modes[win]); best_dist = (uint32_t)uni((int)r.dist[win]);
        // Launches of few units: the chain that bounds a frame is luma only -- this CU's winner -> rough modes of the next PU -> its candidates.  The winner's samples
        // sit in its result slot: the SATD rounds of the next PU are handed to the idle waves NOW, reading them there (srect / ssrc), while this wave copies the
        // winner's levels, samples and arrays (the slot is not written again before the rounds are collected: est_intra_chroma, ahead_open)
        if (win < SLOT_CHROMA && npu == 1 && pu_log2 <= 5 && pu_log2 > min_tu_log2(cu) && lds_load(&wg_shared().remote) && HEVCDL_PREFETCH && NPEND >= 2 && cu.depth < 3      // (slots from SLOT_CHROMA on take this CU's chroma modes)
            && lds_load(&wg_shared().masters_active) <= HEVCDL_PREFETCH_MAX) {
          int nx, ny, nl;
          if (next_leaf(k, cu, nx, ny, nl) && nl >= 4 && nl <= 5) {
            if (uni(s.pend_n) == NPEND) { pend_join_oldest(k, 0); if (uni(s.restart)) return 0; }   // the ticket region (= slot set, = srect entry) the second pass below will take
            early_reg = free_pend_reg();
            wsync();
            if (lane_id() == 0) {
              LDS K &kk = s.k;
              kk.ssrc[early_reg - 1] = slot_rec(k.slots, win) + (5 - pu_log2) * 6144; kk.sorg[early_reg - 1] = ptu.x | (ptu.y << 16);
              kk.srect[early_reg - 1] = (unsigned long long)cu.x | ((unsigned long long)cu.y << 16) | ((unsigned long long)(cu.x + (1 << cu.log2)) << 32) | ((unsigned long long)(cu.y + (1 << cu.log2)) << 48);
            }
            wsync();
            rmd_prefetch(k, nx, ny, nl, 2);
            if (lane_id() == 0) s.pre_open = (nl << 24) | (ny << 12) | nx;
            wsync();
            TL(31, 0);
          }
        }
        GLB const uint8_t *at = slot_attr(k.slots, win);
        wsync();
        for (int i = lane_id(); i < pu_parts; i += 64) {
          const uint8_t t0 = at[i], t1 = at[256 + i], t2 = at[512 + i];
          cold_sv(0)[i] = t0; cold_sv(1)[i] = t1; cold_sv(2)[i] = t2;
          s.a[A_TRIDX][zp + i] = t0; s.a[A_CBF][zp + i] = t1; s.a[A_TSKIP][zp + i] = t2;
        }
        wsync();
        TL(32, 0);
        set_result_cu(k, cu, ptu, 0, slot_coef(k.slots, win), slot_rec(k.slots, win), zp * 16, ptu.x, ptu.y);        // a first-pass slot's origin is the PU
        TL(33, 0);
        if (npu == 1 && pu_log2 <= 5) { // the winner's coefficient bits and coder state, kept for the CU's syntax count (its slot may serve a chroma mode next)
          GLB const unsigned long long *src = slot_state(k.slots, win, 1); GLB unsigned long long *dst = s.my_log + 65 * (LEAF_LOG / 8);
          if (lane_id() < 21) dst[lane_id()] = src[lane_id()];
          if (lane_id() == 0) { s.lw_cfrac = r.cfrac[win]; s.lw_valid = 1; }
        }
      }
      region_close(r);
      if (early_reg) { // the copy is in best_rec: from here on the samples are read there (the slot will serve the next CU's candidates)
        wg_release();
        if (lane_id() == 0) s.k.ssrc[early_reg - 1] = k.best_rec + boff(k, 0, ptu.x, ptu.y);        // ONE word changes (a helper may be copying this context): same origin, the PU's corner
        wsync();
      }
    }
    // ---- RD pass 2 (:2445-2512) = the best first-pass mode again, now with TU splitting allowed.  Its unsplit coding is a bit-exact
    // repeat of the first pass (same mode, same references, same coder state): reuse it (memo) ----
    do {
      set_parts(k, s.a[A_LDIR], zp, pu_parts, (int)best_mode);
      cabac_copy(k, &s.go, &s.curr[cu.depth]);
      const int memo = pu_log2 <= 5;
      if (memo && !(pu_log2 > min_tu_log2(cu))) break;              // no split possible: the pass cannot change anything
      if (memo && npu == 1 && (spare_waves() || lds_load(&wg_shared().remote))) {
        // Spare waves: the pass is handed to one of them and joined in check_rd_cost_intra, after the chroma search and the CU's syntax have
        // run on the assumption that it changes nothing (the unsplit TU wins ~95 % of the time) -- or later still (compress_cu).  If the split
        // wins, what was built on the assumption is redone.
        if (uni(s.pend_n) == NPEND) { pend_join_oldest(k, 0); if (uni(s.restart)) return 0; }   // no ticket region free: the oldest pending pass first
        const int reg = early_reg ? early_reg : free_pend_reg();                 // a free ticket region = slot set + 1 (the one the early SATD rounds named, if any)
        LRegion &r2 = my_region(reg);
        wsync();
        if (lane_id() == 0) { r2.modes[0] = (int)best_mode; r2.modes[1] = reg - 1; r2.cost[4] = best_cost; r2.dist[4] = best_dist; s.p2_pending = reg; }
        if (lds_load(&wg_shared().remote) && HEVCDL_PREFETCH && NPEND >= 2 && cu.depth < 3 && lds_load(&wg_shared().masters_active) <= HEVCDL_PREFETCH_MAX) {
          // The chain that bounds a frame in this form is luma only: this CU's winner -> rough modes of the next PU -> its candidates.  The winner's samples
          // are in best_rec now: mark the CU as the pending pass's (readers take best_rec; check_rd_cost_intra writes the same word again) and hand the SATD
          // rounds of the next PU to the idle waves BEFORE the pass and the chroma modes are posted (est_intra_chroma collects them)
          int nx, ny, nl;
          if (!early_reg && next_leaf(k, cu, nx, ny, nl) && nl >= 4 && nl <= 5) {
            wsync();
            if (lane_id() == 0) { LDS K &kk = s.k; kk.ssrc[reg - 1] = k.best_rec + boff(k, 0, cu.x, cu.y); kk.sorg[reg - 1] = cu.x | (cu.y << 16);
              kk.srect[reg - 1] = (unsigned long long)cu.x | ((unsigned long long)cu.y << 16) | ((unsigned long long)(cu.x + (1 << cu.log2)) << 32) | ((unsigned long long)(cu.y + (1 << cu.log2)) << 48); }
            wsync();
            rmd_prefetch(k, nx, ny, nl, 2);
            if (lane_id() == 0) s.pre_open = (nl << 24) | (ny << 12) | nx;
            wsync();
          }
        }
        TL(34, 0);
        state_to_global(slot_state(k.slots, SLOT_P2 + SLOT_PSET * (reg - 1), 0), &s.curr[cu.depth]);
        const int rm = lds_load(&wg_shared().remote);
        // (very few units: the CU's five chroma modes are posted in the same breath -- everything a chroma mode reads is settled once the luma winner is imported, and
        //  posting here instead of in est_intra_chroma brings their answers, which the walk waits for, ~15 k cycles forward and saves a release of its own)
        if (rm && (rm != 3 || remote_room())) remote_post(k, cu, ptu, reg, (int)best_mode, best_cost, best_dist, cu.part == SIZE_2Nx2N && (rm == 2 || (rm == 1 && HEVCDL_CHROMA_ROOM > 0 && remote_room(HEVCDL_CHROMA_ROOM))));    // a workgroup without a unit runs it (few units in the launch)
        else region_open(r2, T_LUMA_P2, 1, cu, ptu);
        TL(35, 0);
        break;
      }
      PROF_MARK0();
      const DistCost dc = recur_luma_any(k, cu, ptu, 0, memo, best_dist, best_cost);
      PROF_MARK(37);
      if (ub(dc.cost < best_cost)) {
        best_dist = dc.dist; best_cost = dc.cost;
        set_result_cu(k, cu, ptu, 0, k.coef_l, k.rec_l, k.lz, k.lx, k.ly);
        for (int i = lane_id(); i < pu_parts; i += 64) {
          cold_sv(0)[i] = s.a[A_TRIDX][zp + i]; cold_sv(1)[i] = s.a[A_CBF][zp + i]; cold_sv(2)[i] = s.a[A_TSKIP][zp + i];   // the luma search leaves chroma entries alone
        }
        wsync();
      }
    } while (0);
    overall += best_dist;
    wsync();
    for (int i = lane_id(); i < pu_parts; i += 64) {
      s.a[A_TRIDX][zp + i] = cold_sv(0)[i]; s.a[A_CBF][zp + i] = cold_sv(1)[i]; s.a[A_TSKIP][zp + i] = cold_sv(2)[i];
    }
    if (pu != npu - 1) {
      GLB pel_t *rp = k.rec[0] + (size_t)ptu.y * k.W + ptu.x; GLB const pel_t *br = k.best_rec + boff(k, 0, ptu.x, ptu.y);
      for (int i = lane_id(); i < pn * pn; i += 64) rp[(size_t)(i >> pu_log2) * k.W + (i & (pn - 1))] = br[(i >> pu_log2) * 64 + (i & (pn - 1))];
    }
    set_parts(k, s.a[A_LDIR], zp, pu_parts, (int)best_mode);
    wsync();
  }
  if (npu > 1) {
    int comb[3] = { 0, 0, 0 };
    for (int p = 0; p < 4; p++) for (int c = 0; c < 3; c++) comb[c] |= (s.a[A_CBF + c][cu.zbase + p * pu_parts] >> 1) & 1;
    wsync();
    for (int i = lane_id(); i < cu.nparts; i += 64) for (int c = 0; c < 3; c++) s.a[A_CBF + c][cu.zbase + i] |= (uint8_t)comb[c];
    wsync();
  }
  cabac_copy(k, &s.go, &s.curr[cu.depth]);
  MT(5);
  TL(4, 0);
  PROF_ADD(k, 16);
  return overall;
}

// xRecurIntraChromaCodingQT TEncSearch.cpp:1941-2145
// TS3: the copy for 8x8 luma TUs that tries transform skip on their 4x4 chroma blocks (TransformSkipFast 0 only: the usual copy stays without that code)
template <int LOG2, bool TS3 = false> DEVN uint32_t recur_chroma(KR k, const Cu cu_, const Tu tu_)
{
  CHECK_EXEC(7);
  if constexpr (LOG2 == 3 && !TS3) { if ((tools_of(k) & (int)(HEVCDL_TOOL_TSKIP | HEVCDL_TOOL_TSKIP_FAST)) == (int)HEVCDL_TOOL_TSKIP) return recur_chroma<3, true>(k, cu_, tu_); }
  uint32_t dist_sum = 0;
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_);
  LSmem &s = lds();
  const int z = cu.zbase + tu.zrel;
  if (uni(s.a[A_TRIDX][z]) == tu.trd) {
    if (!tu_has_chroma_first(tu)) return 0;
    const int full_depth = cu.depth + tu.trd;
    // a 4x4 chroma block (under a luma TU of 8x8, or of four 4x4); TransformSkipFast: only under 4x4 luma TUs of which one was transform-skipped (TEncSearch.cpp:1965-1990)
    const int ts_fast = tools_of(k) & (int)HEVCDL_TOOL_TSKIP_FAST;
    int check_ts = (tools_of(k) & (int)HEVCDL_TOOL_TSKIP) && (LOG2 == 2 || (LOG2 == 3 && TS3 && !ts_fast));
    if (check_ts && ts_fast) { int nb = 0; for (int i = 0; i < 4; i++) nb += s.a[A_TSKIP + 0][z + i]; check_ts = uni(nb) > 0; }
    const int zc = cu.zbase + tu_czrel(tu), np = tu_cnparts(tu);
    for (int comp = 1; comp < 3; comp++) {
      cabac_copy(k, &s.root[full_depth], &s.go);
      double single_cost = MAX_DOUBLE, cost_tmp = 0; int best_id = 0, best_ts = 0; uint32_t single_dist = 0, single_cbf = 0;
      const int total = check_ts ? 2 : 1; int cur_id = 0;
      for (int ts = 0; ts < total; ts++) {
        set_parts(k, s.a[A_TSKIP + comp], zc, np, ts); wsync();
        cur_id++;
        const int one = (total == 1), last = (cur_id == total);
        const int m012 = one ? 0 : (ts == 0 ? 1 : 2);
        const uint32_t d = code_tu_block(k, cu, tu, comp, m012).dist;
        const uint32_t cbf = (uint32_t)(uni(s.a[A_CBF + comp][zc]) >> tu.trd) & 1;
        if (!one && !last) store_ts_result(k, cu, tu, comp);   // before the bit count reuses s->lvl
        if (ts == 1 && cbf == 0) cost_tmp = MAX_DOUBLE;
        else if (!one) { // xGetIntraBitsQTChroma :1119-1127
          wsync(); if (lane_id() == 0) reset_bits(&s.go);
          enc_coeff_qt<LOG2>(k, &s.go, cu, tu, comp, 0);
          wsync(); cost_tmp = calc_rd_cost(k, (uint32_t)uni((int)get_bits(&s.go)), d);
        }
        if (ub(cost_tmp < single_cost)) {
          single_cost = cost_tmp; single_dist = d; best_ts = ts; best_id = cur_id; single_cbf = cbf;
          if (!one && !last) cabac_copy(k, &s.tbest, &s.go);
        }
        if (!one && !last) cabac_copy(k, &s.go, &s.root[full_depth]);
      }
      if (ub(best_id < total)) {
        load_ts_result(k, cu, tu, comp);
        set_parts(k, s.a[A_CBF + comp], zc, np, (int)(single_cbf << tu.trd)); wsync();
        cabac_copy(k, &s.go, &s.tbest);
      }
      set_parts(k, s.a[A_TSKIP + comp], zc, np, best_ts); wsync();
      dist_sum += single_dist;
    }
  } else {
    if constexpr (LOG2 > 2) {
      uint32_t split_cbf[3] = { 0, 0, 0 };
      for (int i = 0; i < 4; i++) {
        const Tu ch = tu_child(tu, i);
        dist_sum += recur_chroma<LOG2 - 1>(k, cu, ch);
        for (int comp = 1; comp < 3; comp++) split_cbf[comp] |= (uint32_t)(uni(s.a[A_CBF + comp][cu.zbase + ch.zrel]) >> ch.trd) & 1;
      }
      wsync();
      for (int comp = 1; comp < 3; comp++) if (split_cbf[comp])
        for (int i = lane_id(); i < tu.nparts; i += 64) s.a[A_CBF + comp][z + i] |= (uint8_t)(1 << tu.trd);
      wsync();
    }
  }
  return dist_sum;
}

// ---------------------------------------------------------------------------------------------------
// regions: open / claim / run / answer
// ---------------------------------------------------------------------------------------------------
DEVN void region_open(LRegion &r, int kind_, int n_, const Cu cu_, const Tu tu_)
{ CHECK_EXEC(12); // r.modes[] already written by the caller
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_); const int kind = uni(kind_), n = uni(n_);
  wsync();
  if (lane_id() == 0) {
    r.kind = kind; r.owner = wave_id(); r.done = 0;
    r.cu[0] = cu.x; r.cu[1] = cu.y; r.cu[2] = cu.log2; r.cu[3] = cu.depth; r.cu[4] = cu.zbase; r.cu[5] = cu.nparts; r.cu[6] = cu.part;
    r.tu[0] = tu.x; r.tu[1] = tu.y; r.tu[2] = tu.log2; r.tu[3] = tu.trd; r.tu[4] = tu.zrel; r.tu[5] = tu.nparts;
  }
  wsync();
  wg_release();                                              // the master's arrays, snapshots and picture writes before the ticket
  if (lane_id() == 0) __hip_atomic_store(&r.ticket, n << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (n > 0) ring_bell();
}

// a helper takes over the master's view of the CTU: kernel context + attribute arrays (the coder snapshot a task starts from is
// read straight from the master's block)
DEV void import_owner(int owner)
{
  LSmem &s = lds(); LSmem &ow = lds_of(owner);
  static_assert(sizeof(K) % 8 == 0 && offsetof(RdSmem, a) % 8 == 0 && offsetof(RdSmem, k) == 0, "8-byte copies");
  static_assert(offsetof(RdSmem, line) % 8 == 0 && offsetof(RdSmem, fline) % 8 == 0 && offsetof(RdSmem, cline) % 8 == 0, "8-byte copies of the reference lines");
  wsync();
  { LDS unsigned long long *d = (LDS unsigned long long *)&s.k; LDS const unsigned long long *q = (LDS const unsigned long long *)&ow.k;
    for (int i = lane_id(); i < (int)(sizeof(K) / 8); i += 64) d[i] = q[i]; }
  { LDS unsigned long long *d = (LDS unsigned long long *)&s.a[0][0]; LDS const unsigned long long *q = (LDS const unsigned long long *)&ow.a[0][0];
    for (int i = lane_id(); i < 11 * 256 / 8; i += 64) d[i] = q[i]; }
  wsync();
  s.k.q_cost = s.my_qcost; s.k.q_rate = s.my_qrate; s.k.ovl = s.my_ovl;      // every lane stores the same values
  if (lane_id() < 3) s.ref_key[lane_id()] = -1;
  if (lane_id() == 0) s.fline_key = -1;
  wsync();
}

// the same for a candidate coded ahead of time: the master's context as it was when it opened the look-ahead (ahead_open), from its log in HBM
DEV void import_ahead(int owner)
{
  LSmem &s = lds(); LSmem &ow = lds_of(owner);
  GLB const unsigned long long *q = ow.my_log + LOG_AHEAD * (LEAF_LOG / 8);
  wsync();
  { LDS unsigned long long *d = (LDS unsigned long long *)&s.k; for (int i = lane_id(); i < (int)(sizeof(K) / 8); i += 64) d[i] = q[i]; }
  { LDS unsigned long long *d = (LDS unsigned long long *)&s.a[0][0]; for (int i = lane_id(); i < 11 * 256 / 8; i += 64) d[i] = q[sizeof(K) / 8 + i]; }
  wsync();
  s.k.q_cost = s.my_qcost; s.k.q_rate = s.my_qrate; s.k.ovl = s.my_ovl;      // every lane stores the same values
  if (lane_id() < 3) s.ref_key[lane_id()] = -1;
  if (lane_id() == 0) s.fline_key = -1;
  wsync();
}

// one alternative, on the executing wave's private state; levels / reconstruction go to the result slot, trial samples to the overlay
template <bool LEAF> DEV void run_task_body(LRegion &r, int idx_)
{ // LEAF: the instance a chain owner uses for its own split tasks (spec_children): every kind but the second-pass task, no nested regions
  CHECK_EXEC(3);
  const int idx = uni(idx_);
  PROF_T0();
  LSmem &s = lds(); LDS K &kk = s.k; KR k = s.k;
  LSmem &ow = lds_of(uni(r.owner));
  const Cu cu = { uni(r.cu[0]), uni(r.cu[1]), uni(r.cu[2]), uni(r.cu[3]), uni(r.cu[4]), uni(r.cu[5]), uni(r.cu[6]) };
  const Tu tu = { uni(r.tu[0]), uni(r.tu[1]), uni(r.tu[2]), uni(r.tu[3]), uni(r.tu[4]), uni(r.tu[5]) };
  const int kind = uni(r.kind);
  TL(20 + kind, idx);
  // chroma modes by component (est_intra_chroma, workgroups with waves to spare): task idx = component (idx & 1) of mode idx >> 1; the two share the mode's slot
  const bool csplit = kind == T_CHROMA && uni(r.pad_) == 1;
  const int mode = uni(r.modes[csplit ? idx >> 1 : idx]);
  wsync();
  if (kind == T_RMD) { // a slice of the rough mode decision's rounds (rmd_satd): SATD sums into the owner's array, nothing else
    const int nrounds = uni(r.modes[1]), ntasks = nrounds < NW ? nrounds : NW, per = (nrounds + ntasks - 1) / ntasks;
    int dcv = uni(r.modes[0]);
    if (uni(r.modes[2]) == 2) { // the slice gathers the lines itself (the owner is busy posting)
      build_refs(k, 0, tu.x, tu.y, 1 << tu.log2, 1);
      filter_refs(k, 1 << tu.log2);
      dcv = dc_value(k, s.line, 1 << tu.log2);
      wsync();
    } else if (&s != &ow) { // the owner's gathered / smoothed reference lines
      wsync();
      for (int i = lane_id(); i < 66; i += 64) { ((LDS unsigned long long *)s.line)[i] = ((LDS const unsigned long long *)ow.line)[i]; ((LDS unsigned long long *)s.fline)[i] = ((LDS const unsigned long long *)ow.fline)[i]; }
      wsync();
    }
    TL(19, idx);
    rmd_rounds(k, uni(r.modes[2]) ? ow.satd_pre : ow.satd, tu.x, tu.y, 1 << tu.log2, dcv, idx * per, (idx + 1) * per < nrounds ? (idx + 1) * per : nrounds);
    TL(40 + T_RMD, idx);
    return;
  }
#ifdef HEVCDL_PROF_GLUE
  PROF_TASK(kind == T_LUMA_P1 || kind == T_LUMA_AHEAD ? 1 : (kind == T_CHROMA ? 2 : (kind == T_LUMA_SPLIT ? 3 : 4)));
#else
  PROF_TASK(kind != T_LUMA_P2);
#endif
  PROF_MARK0();
  PROF_GLUE_T0();
  const int pset = kind == T_LUMA_P2 ? uni(r.modes[1]) : uni(kk.pset);    // slot set of the second pass: given with its ticket; a split task finds it in the chain owner's context
  const int slot = kind == T_LUMA_SPLIT ? SLOT_SPLIT + SLOT_PSET * pset + mode : (kind == T_CHROMA ? SLOT_CHROMA + (csplit ? idx >> 1 : idx) : (kind == T_LUMA_P2 ? SLOT_P2 + SLOT_PSET * pset : idx));   // T_LUMA_SPLIT: child `mode` of r.tu
  const Tu ttu = kind == T_LUMA_SPLIT ? tu_child(tu, mode) : tu;
  const int olz = uni(kk.lz), olx = uni(kk.lx), oly = uni(kk.ly);          // the owner's own origin (it may run this task itself)
  const bool chroma_kind = kind == T_CHROMA;
  kk.lz = (cu.zbase + (chroma_kind ? 0 : ttu.zrel)) * 16; kk.lx = chroma_kind ? cu.x : ttu.x; kk.ly = chroma_kind ? cu.y : ttu.y;
  if (kind == T_LUMA_P2) { // the whole second pass: trial samples go to the picture (the master keeps off the CU's luma until the join), levels to this wave's layers
    kk.coef_l = s.my_coef; kk.rec_l = s.my_rec; kk.in_task = 0;
  } else {
    kk.coef_l = slot_coef(kk.slots, slot); kk.rec_l = slot_rec(kk.slots, slot); kk.in_task = 1;
    kk.trx0 = ttu.x; kk.try0 = ttu.y; kk.trx1 = ttu.x + (1 << ttu.log2); kk.try1 = ttu.y + (1 << ttu.log2);
  }
  wsync();
  LCabac *start = &ow.curr[cu.depth];
  GLB uint8_t *at = slot_attr(kk.slots, slot);
  uint32_t dist; double cost;
  PROF_GLUE(5);
  if (!LEAF && kind == T_LUMA_P2) { // second RD pass of the PU (TEncSearch.cpp:2445-2512) with the first pass's result as the unsplit alternative
    const int zp = cu.zbase + tu.zrel;
    const double memo_cost = r.cost[4]; const uint32_t memo_dist = (uint32_t)uni((int)r.dist[4]);
    wsync();
    kk.srect[pset] = 0; kk.pset = pset;                        // this pass IS the pending one of its set: its own CU's samples are the picture's (the master may be CUs ahead by now)
    wsync();
    state_from_global(&s.go, slot_state(kk.slots, slot, 0));    // the master's [depth][CI_CURR_BEST] snapshot as it was when the pass was handed over
    DistCost dc = { 0, 0.0, 0 };
#ifdef HEVCDL_XP2
    dc.cost = MAX_DOUBLE;                                       // timing experiment: the second pass costs nothing (as if another workgroup ran it)
#else
    if constexpr (!LEAF) dc = recur_luma_any<true>(k, cu, tu, 0, 1, memo_dist, memo_cost);
#endif
    dist = memo_dist; cost = memo_cost;
    if (ub(dc.cost < memo_cost)) { // the split wins: levels -> record, reconstruction -> the master's best, arrays -> the verdict slot
      dist = dc.dist; cost = dc.cost;
      set_result_cu(k, cu, tu, 0, k.coef_l, k.rec_l, k.lz, k.lx, k.ly);
      wsync();
      for (int i = lane_id(); i < tu.nparts; i += 64) { at[i] = s.a[A_TRIDX][zp + i]; at[256 + i] = s.a[A_CBF][zp + i]; at[512 + i] = s.a[A_TSKIP][zp + i]; }
    }
  } else if (kind == T_LUMA_SPLIT) { // second RD pass: the four grandchildren of one child, from the state the chain of unsplit children had there
    const int zc = cu.zbase + ttu.zrel;
    state_from_global(&s.go, slot_state(kk.slots, slot, 0));
    if (lane_id() == 0) { s.bound_reg = uni(r.owner); s.bound_child = mode; }
    const DistCost dc = recur_luma_any(k, cu, ttu, 0, 1, 0, MAX_DOUBLE);     // memo form with an unlimited unsplit cost: the split is evaluated and taken (or given up: recur_luma's early exit)
    wsync();
    if (lane_id() == 0) s.bound_child = -1;
    dist = dc.dist; cost = dc.cost;
    if (lane_id() == 0) r.cfrac[idx] = dc.cfrac;
    state_to_global(slot_state(kk.slots, slot, 1), &s.go);
    for (int i = lane_id(); i < ttu.nparts; i += 64) { at[i] = s.a[A_TRIDX][zc + i]; at[256 + i] = s.a[A_CBF][zc + i]; at[512 + i] = s.a[A_TSKIP][zc + i]; }
  } else if (kind == T_LUMA_P1 || kind == T_LUMA_AHEAD) { // one candidate of the first RD pass (TEncSearch.cpp:2378-2443)
    const int zp = cu.zbase + tu.zrel;
    set_parts(k, s.a[A_LDIR], zp, tu.nparts, mode);
    // look-ahead (ahead_open): the CU before this one is still in its chroma search; what the candidate reads of the coder state -- the luma-side contexts -- is
    // what the first pass's winner of that CU left (entry 65 of the owner's log); the fractional bits it starts from are a guess the master corrects
    if (kind == T_LUMA_AHEAD && idx < uni(r.pad_)) state_from_global(&s.go, ow.my_log + 65 * (LEAF_LOG / 8)); else cabac_copy(k, &s.go, start);
    // a candidate of a PU that is one TU writes its reconstruction to the layer only (code_tu_block mode 3)
    const int one_tu = tu.log2 >= 3 && tu.log2 <= 5;
    if (&s != &ow && tu.log2 <= 5) {
      // the master gathered (and smoothed) the PU's reference samples for the rough mode decision and never rebuilds them while this region
      // is open (a PU of one TU: every candidate meets the same block): take its lines instead of gathering them from the picture again
      wsync();
      for (int i = lane_id(); i < 66; i += 64) { ((LDS unsigned long long *)s.line)[i] = ((LDS const unsigned long long *)ow.line)[i]; ((LDS unsigned long long *)s.fline)[i] = ((LDS const unsigned long long *)ow.fline)[i]; }
      if (lane_id() == 0) { s.ref_key[0] = ow.ref_key[0]; s.fline_key = ow.fline_key; }
      wsync();
    }
    PROF_MARK(50);
    DistCost dc;
    if (one_tu) { // the single-TU branch of recur_luma (first pass: no split to check, no transform-skip trial above 4x4) without its call frame
      set_parts(k, s.a[A_TSKIP + 0], zp, tu.nparts, 0); wsync();
      const TuRes tr = code_tu_block(k, cu, tu, 0, 3, 1);
      dc.dist = tr.dist;
      const uint32_t bits = tr.bits;
      dc.cost = calc_rd_cost(k, bits, dc.dist); dc.cfrac = uni64(s.cfrac_last);
    } else dc = recur_luma_any(k, cu, tu, 1);
    PROF_MARK(51);
    dist = dc.dist; cost = dc.cost;
    wsync();
    if (lane_id() == 0) r.cfrac[idx] = dc.cfrac;
    state_to_global(slot_state(kk.slots, slot, 1), &s.go);          // coder state behind the candidate's bit count: its coefficient contexts are the CU's if it wins (enc_cu_syntax_fast)
    for (int i = lane_id(); i < tu.nparts; i += 64) { at[i] = s.a[A_TRIDX][zp + i]; at[256 + i] = s.a[A_CBF][zp + i]; at[512 + i] = s.a[A_TSKIP][zp + i]; }
  } else { // one chroma mode (TEncSearch.cpp:2640-2700)
    if (&s != &ow && cu.log2 <= 5 && uni(s.a[A_TRIDX][cu.zbase]) == 0) { // one chroma TU per component: the master gathered both lines before it opened the region
      wsync();
      for (int i = lane_id(); i < 66; i += 64) ((LDS unsigned long long *)s.cline)[i] = ((LDS const unsigned long long *)ow.cline)[i];
      if (lane_id() < 2) s.ref_key[1 + lane_id()] = ow.ref_key[1 + lane_id()];
      wsync();
    }
    cabac_copy(k, &s.go, start);
    set_parts(k, s.a[A_CDIR], cu.zbase, cu.nparts, mode); wsync();
    bool counted = true;
    if (csplit) { // one component of the mode (a CU of one TU per component, no transform-skip trial: the single-TU branch of recur_chroma); the component that
      // finishes second counts the mode's bits -- it finds the other one's levels in the slot they share and its flags in the slot's arrays
      const int m = idx >> 1, comp = 1 + (idx & 1), other = 3 - comp;
      GLB uint32_t *half = (GLB uint32_t *)(slot_state(kk.slots, slot, 1) + 32);        // behind the coder state: the two components' distortions
      set_parts(k, s.a[A_TSKIP + comp], cu.zbase, cu.nparts, 0); wsync();
      const uint32_t d = code_tu_block(k, cu, tu, comp, 0).dist;
      wsync();
      for (int i = lane_id(); i < cu.nparts; i += 64) { at[(comp - 1) * 256 + i] = s.a[A_CBF + comp][cu.zbase + i]; at[(comp + 1) * 256 + i] = 0; }
      GLB int *pair = (GLB int *)uni64((unsigned long long)s.rp_pair);          // != 0: the other component runs in another workgroup, on another XCD (remote_serve)
      int prev = 0;
      if (pair) { // the two meet in the poster's job headers (a line each): agent-scope release before the counter, acquire behind it
        if (lane_id() == 0) *s.rp_half_mine = d;
        wsync();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane_id() == 0) prev = __hip_atomic_fetch_add(pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (lane_id() == 0) half[comp - 1] = d;
        wsync();
        wg_release();
        if (lane_id() == 0) prev = __hip_atomic_fetch_add(&r.modes[5 + m], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      counted = uni(prev) == 1;
      if (counted) {
        if (pair) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); else wg_acquire();
        wsync();
        for (int i = lane_id(); i < cu.nparts; i += 64) { s.a[A_CBF + other][cu.zbase + i] = at[(other - 1) * 256 + i]; s.a[A_TSKIP + other][cu.zbase + i] = 0; }
        dist = pair ? d + (uint32_t)uni((int)*s.rp_half_other) : (uint32_t)(uni((int)half[0]) + uni((int)half[1]));
        wsync();
      }
    }
    uint32_t bits = 0;
    const bool one_tu = cu.part == SIZE_2Nx2N && cu.log2 <= 5 && uni(s.a[A_TRIDX][cu.zbase]) == 0;      // the CU is one transform unit: the short count (chroma_cu_bits_1tu)
    if (csplit) {
      if (counted) {
        cabac_copy(k, &s.go, start);
        bits = chroma_cu_bits_1tu(k, cu, tu);            // (components run apart only for a CU of one TU)
      }
    } else {
      switch (cu.log2) {
        case 6: dist = recur_chroma<6>(k, cu, tu); break;
        case 5: dist = recur_chroma<5>(k, cu, tu); break;
        case 4: dist = recur_chroma<4>(k, cu, tu); break;
        default: dist = recur_chroma<3>(k, cu, tu); break;
      }
      cabac_copy(k, &s.go, start);
      if (one_tu) bits = chroma_cu_bits_1tu(k, cu, tu);
      else switch (cu.log2) {
        case 6: bits = intra_bits_qt<6>(k, cu, tu, 0, 1); break;
        case 5: bits = intra_bits_qt<5>(k, cu, tu, 0, 1); break;
        case 4: bits = intra_bits_qt<4>(k, cu, tu, 0, 1); break;
        default: bits = intra_bits_qt<3>(k, cu, tu, 0, 1); break;
      }
    }
    if (csplit) {
      if (counted) { // the mode's answer, where est_intra_chroma looks for it: cost / fractional bits by mode, distortion behind the fractional bits
        const int m = idx >> 1;
        cost = calc_rd_cost(k, bits, dist);
        wsync();
        if (lane_id() == 0) { r.cost[m] = cost; r.cfrac[m] = s.cfrac_last_c; r.cfrac[5 + m] = (unsigned long long)dist; }
        state_to_global(slot_state(kk.slots, slot, 1), &s.go);
      }
    } else {
    cost = calc_rd_cost(k, bits, dist);
    wsync();
    if (lane_id() == 0) r.cfrac[idx] = s.cfrac_last_c;
    state_to_global(slot_state(kk.slots, slot, 1), &s.go);
    for (int i = lane_id(); i < cu.nparts; i += 64) for (int c = 1; c < 3; c++) { at[(c - 1) * 256 + i] = s.a[A_CBF + c][cu.zbase + i]; at[(c + 1) * 256 + i] = s.a[A_TSKIP + c][cu.zbase + i]; }
    }
  }
  if (!csplit && lane_id() == 0) { r.cost[idx] = cost; r.dist[idx] = dist; }
  wsync();
  if (kind == T_LUMA_P1 || kind == T_LUMA_AHEAD) PROF_MARK(52);
  kk.coef_l = s.my_coef; kk.rec_l = s.my_rec; kk.in_task = 0;
  kk.lz = olz; kk.lx = olx; kk.ly = oly;
  wsync();
  TL(40 + kind, idx);
  PROF_TASK(0);
#ifdef HEVCDL_KERNEL_PROF
  { if (kind == T_LUMA_P1 || kind == T_LUMA_AHEAD) PROF_ADD(k, 40); else if (kind == T_CHROMA) PROF_ADD(k, 41); else if (kind == T_LUMA_SPLIT) PROF_ADD(k, 42); else PROF_ADD(k, 43); }
#endif
}
// The call form: a master deep inside the search (region_run, spec_children).  Helpers run the body inside the kernel's own frame (helper_step):
// a kernel has no caller whose registers it must preserve, so the ~27 register saves + restores per task (256 B of scratch each) are not made there.
template <bool LEAF> DEVN void run_task(LRegion &r, int idx_) { run_task_body<LEAF>(r, idx_); }

// claim a task of region r: its index, or -1.  Compare-and-swap, not a blind add: the count grows while a region is open (region_publish),
// and an index taken by a failed claim would be skipped when it becomes valid later.
DEVN int region_claim(LRegion &r)
{
  int res = -1;
  if (lane_id() == 0) {
    int t = __hip_atomic_load(&r.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((t & 0xffff) < (int)((unsigned)t >> 16)) {
      if (__hip_atomic_compare_exchange_strong(&r.ticket, &t, t + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { res = t & 0xffff; break; }
    }
  }
  return uni(res);
}
DEVN void region_run(KR k, LRegion &r)
{ CHECK_EXEC(4); // the master works on its own region, then waits for the helpers' last tasks
  const int n = (int)((unsigned)lds_load(&r.ticket) >> 16);
  for (;;) {
    const int idx = region_claim(r);
    if (idx < 0) break;
#ifdef HEVCDL_RR_INLINE
    run_task_body<false>(r, idx);          // inside this function's frame: its registers are saved once per region, not once per task
#else
    run_task<false>(r, idx);
#endif
    wg_release();
    lds_add(&r.done, 1);
  }
  { PROF_T0(); while (lds_load(&r.done) < n) __builtin_amdgcn_s_sleep(2); PROF_ADD(k, 32); }
  wg_acquire();
}
// waves without a unit (or done with theirs) serve the regions of the workgroup's masters until the last master has finished
DEV int helper_step()
{ // one scan of the workgroup's regions: 1 when a task was run
  LDS WgShared &sh = wg_shared();
  const int me = wave_id();
  {
    int did = 0;
    // the second-pass regions first when they sit on the masters' critical paths; behind the masters' own regions when the passes are left pending
    // (compress_cu): then it is the first pass and the chroma search the master waits for
    const int p2_first = lds_load(&sh.masters_active) > HEVCDL_FG_FIRST_MAX;
    for (int j = 0; j < NREG * NW && !did; j++) {
      const int b = j / NW;             // block of the scan: one region index of every wave; the look-ahead comes right behind the masters' own regions, or last
      #ifdef HEVCDL_AHEAD_LAST
      const int ri = p2_first ? (b < NPEND ? 1 + b : (b == NPEND ? 0 : REG_AHEAD)) : b;
#else
      const int ri = p2_first ? (b < NPEND ? 1 + b : (b == NPEND ? 0 : REG_AHEAD)) : (b == 0 ? 0 : (AHEAD ? (b == 1 ? REG_AHEAD : b - 1) : b));
#endif
      LRegion &r = sh.reg[(me + 1 + (j % NW)) % NW][ri];
      const int t = lds_load(&r.ticket);
      if ((t & 0xffff) >= (int)((unsigned)t >> 16)) continue;
      const int idx = region_claim(r);
      if (idx < 0) continue;
      wg_acquire();
      { PROF_T0(); if (AHEAD && uni(r.kind) == T_LUMA_AHEAD && idx < uni(r.pad_)) import_ahead(uni(r.owner)); else import_owner(uni(r.owner)); PROF_ADD(0, 53); }
      if (uni(r.kind) == T_LUMA_SPLIT) lds_add((LDS int *)&r.dist[11], 1);          // the chain owner's context has been copied: it may lend its wave to other masters now (spec_children)
      run_task_body<false>(r, idx);
      wg_release();
      lds_add(&r.done, 1);
      did = 1;
    }
    return did;
  }
}

// estIntraPredChromaQT TEncSearch.cpp:2588-2737 (4:2:0: one chroma PU per CU)
DEVN uint32_t est_intra_chroma(KR k, const Cu cu_)
{
  CHECK_EXEC(6);
  PROF_T0();
  const Cu cu = ucu(cu_);
  LSmem &s = lds();
  const Tu root = { cu.x, cu.y, cu.log2, 0, 0, cu.nparts };
  uint32_t mode_list[5];
  MT0();
  TL(5, cu.zbase);
  wsync();
  if (lane_id() < 2) s.ref_key[1 + lane_id()] = -1;
  wsync();
  const int luma_mode = uni(s.a[A_LDIR][cu.zbase]);
  chroma_mode_list(luma_mode, mode_list);
  uint32_t best_mode = 0, best_dist = 0; double best_cost = MAX_DOUBLE;
  { // the five modes are independent (each starts from the [depth][CI_CURR_BEST] snapshot, TEncSearch.cpp:2640-2660) -> a region
    LRegion &r = my_region();
    wsync();
    const bool rich = lds_load(&wg_shared().remote) != 0;    // waves to spare: the second passes run on other CUs
    auto look_ahead = [&](int sliced) {
      if (HEVCDL_PREFETCH && NPEND >= 2 && cu.depth < 3 && lds_load(&wg_shared().masters_active) <= HEVCDL_PREFETCH_MAX) { // the other waves have the chroma modes: the master looks ahead
        // (not from an 8x8 CU: its 2Nx2N / NxN choice is still open, so is the reconstruction the next CU will see)
        int nx, ny, nl;
        if (next_leaf(k, cu, nx, ny, nl) && nl >= 4 && nl <= 5) {
          rmd_prefetch(k, nx, ny, nl, sliced);
          TL(7, 0);
#if defined(HEVCDL_KERNEL_PROF) && HEVCDL_PROF_N == 64 && !defined(HEVCDL_PROF_MASTER) && !defined(HEVCDL_PROF_GLUE)   // look-ahead statistics on the (then silent) RDOQ accumulators
          if (lane_id() == 0) PROF_ACC_(18, (unsigned long long)((uni(s.lw_valid) ? 0 : 1) + (uni(s.a[A_TRIDX][cu.zbase]) == 0 ? 0 : 2) + (uni(s.ahead_open) ? 4 : 0) + (spare_waves() ? 0 : 8)) << 10);
#endif
          if (AHEAD && uni(s.lw_valid) && cu.part == SIZE_2Nx2N && uni(s.a[A_TRIDX][cu.zbase]) == 0 && !uni(s.ahead_open) && spare_waves()
              && lds_load(&wg_shared().masters_active) <= HEVCDL_AHEAD_MAX) ahead_open(k, cu, nx, ny, nl);
        }
      }
    };
    // The chain that bounds a frame once waves are plentiful is luma only: reconstruction of this CU -> rough modes of the next -> its candidates.  There the
    // look-ahead runs FIRST, its SATD rounds dealt to the idle waves (this wave's ticket region is still free), and the chroma search of this CU follows.
    // Launches of very few units: the five chroma modes go to other workgroups as well (posted first: they take longest to come back), this workgroup's waves
    // have the candidates of the next CU
    // (with more units the chroma modes go to other workgroups only while enough of them have nothing to do: est_intra_luma decided and posted them already)
    const bool cremote = lds_load(&wg_shared().remote) == 2 || uni(s.chroma_key) == ((cu.log2 << 24) | (cu.y << 12) | cu.x);
    int anx = 0, any = 0, anl = 0;
    const bool la = rich && HEVCDL_PREFETCH && NPEND >= 2 && cu.depth < 3 && lds_load(&wg_shared().masters_active) <= HEVCDL_PREFETCH_MAX && next_leaf(k, cu, anx, any, anl) && anl >= 4 && anl <= 5;
    if (uni(s.pre_open) && (!la || uni(s.pre_open) != ((anl << 24) | (any << 12) | anx))) { region_run(k, r); if (lane_id() == 0) s.pre_open = 0; wsync(); }   // (slices opened for another PU: cannot happen by construction)
    if (la && !uni(s.pre_open)) rmd_prefetch(k, anx, any, anl, 2);               // reference lines of the next PU, its SATD rounds handed to the idle waves (est_intra_luma may have done it already) ...
    const bool posted = cremote && uni(s.chroma_key) == ((cu.log2 << 24) | (cu.y << 12) | cu.x);     // together with the second pass (remote_post)
    wsync();
    if (lane_id() == 0) s.chroma_key = 0;
    if (cremote && !posted) chroma_post(k, cu, root, (int)mode_list[0], (int)mode_list[1], (int)mode_list[2], (int)mode_list[3], (int)mode_list[4], 0);   // ... while the chroma modes are posted
    TL(6, 0);
    if (la) {
      rmd_prefetch_end(k, anx, any, anl);
      if (lane_id() == 0) s.pre_open = 0;
      TL(7, 0);
      if (AHEAD && uni(s.lw_valid) && cu.part == SIZE_2Nx2N && uni(s.a[A_TRIDX][cu.zbase]) == 0 && !uni(s.ahead_open) && spare_waves()
          && lds_load(&wg_shared().masters_active) <= HEVCDL_AHEAD_MAX) ahead_open(k, cu, anx, any, anl);
    }
    TL(8, uni(s.ahead_open));
    // with waves to spare the two components of a mode are tasks of their own
    const bool csplit = rich && !cremote && cu.log2 >= 4 && cu.log2 <= 5 && uni(s.a[A_TRIDX][cu.zbase]) == 0;
    PROF_MARK0();
    if (cremote) { MT(8); chroma_collect(r); }
    else {
    if (lane_id() == 0) { for (int m = 0; m < 5; m++) { r.modes[m] = (int)mode_list[m]; r.modes[5 + m] = 0; } r.pad_ = csplit ? 1 : 0; }
    if (cu.log2 <= 5 && uni(s.a[A_TRIDX][cu.zbase]) == 0) { // the five modes of an unsplit CU share their reference samples: gather them once, the tasks copy them
      const int nc = (1 << cu.log2) >> 1;
      build_refs(k, 1, cu.x >> 1, cu.y >> 1, nc, 1); build_refs(k, 2, cu.x >> 1, cu.y >> 1, nc, 1);
    }
    MT(8);
    region_open(r, T_CHROMA, csplit ? 10 : 5, cu, root);
    if (!rich) look_ahead(0);
    }
    MT(18);
    if (!cremote) region_run(k, r);
    MT(35);
    TL(9, 0);
    PROF_MARK(39);
    int win = -1;
    for (int m = 0; m < 5; m++) { const double c = r.cost[m]; if (ub(c < best_cost)) { best_cost = c; win = m; } }
    if (win >= 0) {
      best_mode = (uint32_t)uni(r.modes[win]); best_dist = csplit ? (uint32_t)uni((int)(unsigned)r.cfrac[5 + win]) : (uint32_t)uni((int)r.dist[win]);
      if (lane_id() == 0) { s.cw_cfrac = r.cfrac[win]; s.cw_slot = SLOT_CHROMA + win; }
      GLB const uint8_t *at = slot_attr(k.slots, SLOT_CHROMA + win);
      wsync();
      for (int i = lane_id(); i < cu.nparts; i += 64) for (int c = 0; c < 4; c++) cold_sv(c)[i] = at[c * 256 + i];
      wsync();
      for (int c = 1; c < 3; c++) set_result_cu(k, cu, root, c, slot_coef(k.slots, SLOT_CHROMA + win), slot_rec(k.slots, SLOT_CHROMA + win), cu.zbase * 16, cu.x, cu.y);   // a chroma slot's origin is the CU
    }
    region_close(r);
  }
  wsync();
  for (int i = lane_id(); i < cu.nparts; i += 64) for (int c = 1; c < 3; c++) { s.a[A_CBF + c][cu.zbase + i] = cold_sv(c - 1)[i]; s.a[A_TSKIP + c][cu.zbase + i] = cold_sv(c + 1)[i]; }
  set_parts(k, s.a[A_CDIR], cu.zbase, cu.nparts, (int)best_mode);
  cabac_copy(k, &s.go, &s.curr[cu.depth]);
  MT(19);
  TL(10, 0);
  PROF_ADD(k, 17);
  return best_dist;
}

DEV void copy_best_rec_to_pic(KR k, const Cu &cu, int comp)
{
  const int n = (1 << cu.log2) >> (comp ? 1 : 0), log2n = ilog2(n), x = cu.x >> (comp ? 1 : 0), y = cu.y >> (comp ? 1 : 0);
  const int cs = cstride(comp), ps = pstride(k, comp);
  GLB const pel_t *br = k.best_rec + comp_off(comp) + boff(k, comp, x, y);
  GLB pel_t *rp = k.rec[comp] + (size_t)y * ps + x;
  wsync();
  for (int i = lane_id(); i < n * n; i += 64) rp[(size_t)(i >> log2n) * ps + (i & (n - 1))] = br[(i >> log2n) * cs + (i & (n - 1))];
  wsync();
}

// The CU's syntax count (enc_cu_syntax) WITHOUT coding the coefficients again, for a 2Nx2N CU of one TU per component.  The count is: the mode / flag bins, then
// the luma, Cb and Cr coefficients, all from the [depth][CI_CURR_BEST] snapshot.  Coefficient bins touch the coefficient contexts only (luma and chroma have
// their own), the flags only the others; the first pass's winner counted exactly these luma coefficient bins from that snapshot, the chroma search's winner
// exactly these chroma bins (Cb then Cr) -- so their fractional bits and final contexts are taken over (as split_bits does for a split TU) and only the flags
// are coded here.  Leaves `c` exactly as the full count would.
DEVN void enc_cu_syntax_fast(KR k, LCabac *c, const Cu cu_, GLB const unsigned long long *luma_end, GLB const unsigned long long *chroma_end, unsigned long long cfrac)
{
  PROF_T0();
  const Cu cu = ucu(cu_);
  const int lane = lane_id();
  GLB const uint8_t *lb = (GLB const uint8_t *)luma_end, *cb = (GLB const uint8_t *)chroma_end;
  uint8_t lv[3], cv[3];
#pragma unroll
  for (int t = 0; t < 3; t++) { const int i = lane + 64 * t; lv[t] = i < NUM_CTX ? lb[i] : (uint8_t)0; cv[t] = i < NUM_CTX ? cb[i] : (uint8_t)0; }      // in flight while the flags are coded
  wsync();
  if (cu.depth == 3 && lane == 0) enc_bin(c, CTX_PART_SIZE, 1);
  code_luma_dirs(k, c, cu, 0, 1);
  code_chroma_dir(k, c, cu);
  const Tu root = { cu.x, cu.y, cu.log2, 0, 0, cu.nparts };
  if (cu.log2 <= 5 && cu.log2 > 2 && cu.log2 != min_tu_log2(cu) && lane == 0) enc_bin(c, CTX_SUBDIV + 5 - cu.log2, 0);        // enc_transform: the root is not split
  code_qt_cbf(k, c, cu, root, 1, 1); code_qt_cbf(k, c, cu, root, 2, 1); code_qt_cbf(k, c, cu, root, 0, 1);
  wsync();
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int i = lane + 64 * t;
    const bool is_l = i == 19 || i == 20 || (i >= CTX_SIG && i < CTX_SIG + 28) || (i >= CTX_LAST_X && i < CTX_LAST_X + 15) || (i >= CTX_LAST_Y && i < CTX_LAST_Y + 15) ||
                      (i >= CTX_ONE && i < CTX_ONE + 16) || (i >= CTX_ABS && i < CTX_ABS + 4) || i == CTX_TSKIP;
    const bool is_c = i == 21 || i == 22 || (i >= CTX_SIG + 28 && i < CTX_LAST_X) || (i >= CTX_LAST_X + 15 && i < CTX_LAST_Y) || (i >= CTX_LAST_Y + 15 && i < CTX_ONE) ||
                      (i >= CTX_ONE + 16 && i < CTX_ABS) || (i >= CTX_ABS + 4 && i < CTX_TSKIP) || i == CTX_TSKIP + 1;
    if (is_l) c->ctx[i] = lv[t]; else if (is_c) c->ctx[i] = cv[t];
  }
  if (lane == 0) c->frac += cfrac;
  wsync();
  PROF_ADD(k, 13);
}

// xCheckRDCostIntra TEncCu.cpp:1600-1665; the end state of the CU syntax is left in s->temp[depth]
DEVN Rd check_rd_cost_intra(KR k, const Cu cu_, int part_, int known_reg_ = 0)
{ // known_reg: the CU's luma search has been done -- by the second pass that was left pending in that ticket region and chose the split (compress_cu)
  CHECK_EXEC(8);
  LSmem &s = lds();
  const int part = uni(part_);
  Cu cu = ucu(cu_); cu.part = part;
  MT0();
  wsync();
  if (lane_id() < 3) s.ref_key[lane_id()] = -1;
  for (int i = lane_id(); i < cu.nparts; i += 64) { // initEstData TComDataCU.cpp:525-592 + part size / pred mode
    const int z = cu.zbase + i;
    s.a[A_DEPTH][z] = (uint8_t)cu.depth; s.a[A_PART][z] = (uint8_t)part; s.a[A_LDIR][z] = DC; s.a[A_CDIR][z] = 0; s.a[A_TRIDX][z] = 0;
    for (int c = 0; c < 3; c++) { s.a[A_CBF + c][z] = 0; s.a[A_TSKIP + c][z] = 0; }
  }
  wsync();
  if (lane_id() == 0) { s.p2_pending = 0; s.left_pending = 0; s.lw_valid = 0; }
  wsync();
  const int known_reg = uni(known_reg_);
  uint32_t dist_l;
  if (known_reg) { // distortion, mode and arrays of the pass's verdict; its levels are in the record, its reconstruction in best_rec and in the picture
    LRegion &rk = my_region(known_reg);
    dist_l = (uint32_t)uni((int)rk.dist[0]);
    GLB const uint8_t *at = slot_attr(k.slots, SLOT_P2 + SLOT_PSET * (known_reg - 1));
    set_parts(k, s.a[A_LDIR], cu.zbase, cu.nparts, uni(rk.modes[0]));
    for (int i = lane_id(); i < cu.nparts; i += 64) { s.a[A_TRIDX][cu.zbase + i] = at[i]; s.a[A_CBF][cu.zbase + i] = at[256 + i]; s.a[A_TSKIP][cu.zbase + i] = at[512 + i]; }
    cabac_copy(k, &s.go, &s.curr[cu.depth]);
    wsync();
  } else { MT(20); dist_l = est_intra_luma(k, cu); MTR(); }
  int pending = uni(s.p2_pending);                   // the second luma pass is running on another wave (est_intra_luma): the region of its ticket
  if (pending) { // until the pass is joined this CU's luma in the picture belongs to it: whoever needs the samples meanwhile (rmd_prefetch, the next CUs) reads best_rec
    wsync();
    if (lane_id() == 0) { LDS K &kk = s.k; kk.ssrc[pending - 1] = k.best_rec + boff(k, 0, cu.x, cu.y); kk.sorg[pending - 1] = cu.x | (cu.y << 16);
      kk.srect[pending - 1] = (unsigned long long)cu.x | ((unsigned long long)cu.y << 16) | ((unsigned long long)(cu.x + (1 << cu.log2)) << 32) | ((unsigned long long)(cu.y + (1 << cu.log2)) << 48); }
    wsync();
  }
  Rd r = { 0.0, 0, 0 };
  if (uni(s.restart)) return r;                      // a pending pass of an earlier CU chose the split: the CTU is walked again (process_unit)
  for (;;) {
    if (!pending) copy_best_rec_to_pic(k, cu, 0);
    MT(21);
    const uint32_t dist = dist_l + est_intra_chroma(k, cu);
    MTR();
    wsync();
    if (lane_id() == 0) reset_bits(&s.go);
    if (part == SIZE_2Nx2N && cu.log2 <= 5 && uni(s.lw_valid) && uni(s.a[A_TRIDX][cu.zbase]) == 0)          // one TU per component, winners known: the short form
      enc_cu_syntax_fast(k, &s.go, cu, s.my_log + 65 * (LEAF_LOG / 8), slot_state(k.slots, uni(s.cw_slot), 1), uni64(s.lw_cfrac) + uni64(s.cw_cfrac));
    else enc_cu_syntax(k, &s.go, cu);
    st_copy(cold_state(COLD_TEMP, cu.depth), st_of(&s.go));
    r.bits = (uint32_t)uni((int)get_bits(&s.go)); r.dist = dist; r.cost = calc_rd_cost(k, r.bits, r.dist);
    MT(22);
    TL(11, 0);
    if (!pending) break;
    if (uni(s.carry_ok)) { // the walk goes on while this pass is still running (compress_cu); until it is joined this CU's luma in the picture belongs to
      // the pass, the search reads best_rec instead
      LDS K &kk = s.k;
      wsync();
      if (lane_id() == 0) {
        const int n = s.pend_n; s.pend_leaf[n] = s.leaf_idx - 1; s.pend_reg[n] = pending; s.pend_n = n + 1; s.p2_pending = 0; s.left_pending = 1;
      }
      wsync();
      break;
    }
    // join: the second pass's verdict
    LRegion &r2 = my_region(pending);
    region_wait(r2, 1);
    const int pset = pending - 1;
    pending = 0;
    wsync();
    if (lane_id() == 0) { s.p2_pending = 0; s.k.srect[pset] = 0; }
    if (!ub(r2.cost[0] < r2.cost[4])) { copy_best_rec_to_pic(k, cu, 0); break; }      // nothing changed: chroma and syntax stand
    // the split won: take its distortion and arrays (its levels and reconstruction are in the record / best reconstruction already) and
    // repeat the chroma search and the syntax, which depend on the TU tree
    dist_l = (uint32_t)uni((int)r2.dist[0]);
    ahead_drain();
    if (lane_id() == 0) s.pre_key = -1;                           // sums (and candidates) computed ahead for the next CU stood on the old reconstruction
    GLB const uint8_t *at = slot_attr(k.slots, SLOT_P2 + SLOT_PSET * pset);
    for (int i = lane_id(); i < cu.nparts; i += 64) { s.a[A_TRIDX][cu.zbase + i] = at[i]; s.a[A_CBF][cu.zbase + i] = at[256 + i]; s.a[A_TSKIP][cu.zbase + i] = at[512 + i]; }
    cabac_copy(k, &s.go, &s.curr[cu.depth]);
    wsync();
  }
  return r;
}

DEV void save_cand8(KR k, const Cu &cu)
{
  LSmem &s = lds();
  GLB int16_t *c8coef = (GLB int16_t *)(s.my_log + 66 * (LEAF_LOG / 8)); GLB pel_t *c8rec = (GLB pel_t *)(s.my_log + 67 * (LEAF_LOG / 8));
  GLB const int16_t *rc = (GLB const int16_t *)(k.records + (size_t)k.addr * REC_SIZE + REC_COEF);
  wsync();
  if (lane_id() < 44) s.c8a[lane_id() >> 2][lane_id() & 3] = s.a[lane_id() >> 2][cu.zbase + (lane_id() & 3)];
  for (int i = lane_id(); i < 96; i += 64) {
    const int c = i < 64 ? 0 : (i < 80 ? 1 : 2), j = i < 64 ? i : (i - 64) & 15;
    c8coef[i] = rc[comp_off(c) + (c ? cu.zbase * 4 : cu.zbase * 16) + j];
    const int n = c ? 4 : 8, sx = c ? 32 : 64, bo = comp_off(c) + boff(k, c, cu.x >> (c ? 1 : 0), cu.y >> (c ? 1 : 0));
    c8rec[i] = k.best_rec[bo + (j / n) * sx + (j % n)];
  }
  wsync();
}
DEV void load_cand8(KR k, const Cu &cu)
{
  LSmem &s = lds();
  GLB const int16_t *c8coef = (GLB const int16_t *)(s.my_log + 66 * (LEAF_LOG / 8)); GLB const pel_t *c8rec = (GLB const pel_t *)(s.my_log + 67 * (LEAF_LOG / 8));
  GLB int16_t *rc = (GLB int16_t *)(k.records + (size_t)k.addr * REC_SIZE + REC_COEF);
  wsync();
  if (lane_id() < 44) s.a[lane_id() >> 2][cu.zbase + (lane_id() & 3)] = s.c8a[lane_id() >> 2][lane_id() & 3];
  for (int i = lane_id(); i < 96; i += 64) {
    const int c = i < 64 ? 0 : (i < 80 ? 1 : 2), j = i < 64 ? i : (i - 64) & 15;
    rc[comp_off(c) + (c ? cu.zbase * 4 : cu.zbase * 16) + j] = c8coef[i];
    const int n = c ? 4 : 8, sx = c ? 32 : 64, bo = comp_off(c) + boff(k, c, cu.x >> (c ? 1 : 0), cu.y >> (c ? 1 : 0));
    k.best_rec[bo + (j / n) * sx + (j % n)] = c8rec[i];
  }
  wsync();
}

// xCompressCU TEncCu.cpp:470-1104 with the reference's label-pruning edits (:496-520, 815-834, 947-965)
//
// Second passes left behind.  With the labels the walk over a CTU is a fixed sequence of CUs ("leaves": each is tested at exactly one depth), every one
// starting from the coder state and the reconstruction its predecessor left.  The second luma pass of a CU (est_intra_luma) changes its result in ~5 % of
// the cases and takes longer than everything else the master does for the CU, so the master does not wait for it: the pass stays pending (up to NPEND of
// them, check_rd_cost_intra) while the walk goes on as if it changed nothing, and is joined when its ticket region is needed again, when it is seen to be
// finished at the start of a later CU, or at the end of the CTU.  Every coded CU is logged (cost triple + the coder state behind it).  If a pass then does
// choose the split, everything after its CU was built on the wrong reconstruction and coder state: the CTU is walked again (process_unit) -- the CUs before
// that one are final (arrays, levels, picture) and only their logged results are replayed, that CU is coded again with its pass joined inside it.
template <int DEPTH> DEVN Rd compress_cu(KR k, int x_, int y_)
{
  CHECK_EXEC(9);
  const int x = uni(x_), y = uni(y_);
  LSmem &s = lds();
  const int log2 = 6 - DEPTH, size = 1 << log2;
  Cu cu = { x, y, log2, DEPTH, (int)tb().r2z[(((y & 63) >> 2) << 4) | ((x & 63) >> 2)], 256 >> (2 * DEPTH), SIZE_2Nx2N };
  const int boundary = uni(!(x + size <= k.W && y + size <= k.H));
  const int pred_depth = uni(s.lab16[4 * ((y & 63) / 16) + (x & 63) / 16]);
  const int check_cur = pred_depth == DEPTH, check_next = pred_depth > DEPTH;
  Rd best = { MAX_DOUBLE, 0, 0 };
  int best_is_real = 0;
  if (!boundary) {
    if (check_cur) {
      const int li = uni(s.leaf_idx);
      wsync();
      if (lane_id() == 0) s.leaf_idx = li + 1;
      wsync();
      GLB unsigned long long *lg = s.my_log + (size_t)(li & 63) * (LEAF_LOG / 8);
      if (li < uni(s.replay_upto)) { // walked before and final: its result and the coder state behind it from the log
        const unsigned long long w0 = lg[0], w1 = lg[1], w23 = lg[23];
        best.cost = __longlong_as_double((long long)uni64(w0)); best.bits = (uint32_t)uni((int)(unsigned)w1); best.dist = (uint32_t)uni((int)(unsigned)(w1 >> 32));
        st_copy(cold_state(COLD_NEXT, DEPTH), lg + 2);
        if (lane_id() == 0) s.ctu_frac += w23;
      } else {
        // a pending pass that has finished meanwhile is joined right away: a restart costs the less the earlier it is seen
        while (uni(s.pend_n)) {
          LRegion &rp0 = my_region(uni(s.pend_reg[0]));
          if (uni(rp0.kind) == T_REMOTE && lds_load(&rp0.done) < 1) remote_poll(rp0);
          if (lds_load(&rp0.done) < 1) break;
          pend_join_oldest(k, 2); if (uni(s.restart)) return best;
        }
        const int carry_ok = NPEND > 0 && DEPTH >= 1 && DEPTH <= 2 && li != uni(s.nocarry_leaf) && lds_load(&wg_shared().masters_active) <= HEVCDL_CARRY_MAX;
        wsync();
        if (lane_id() == 0) s.carry_ok = carry_ok;
        wsync();
        Rd t = check_rd_cost_intra(k, cu, SIZE_2Nx2N, li == uni(s.nocarry_leaf) ? uni(s.resume_reg) : 0);
        if (uni(s.restart)) return t;
        if (ub(t.cost < best.cost)) { best = t; st_copy(cold_state(COLD_NEXT, DEPTH), cold_state(COLD_TEMP, DEPTH)); best_is_real = 1; }
        if (DEPTH == 3) {
          save_cand8(k, cu);
          Rd t2 = check_rd_cost_intra(k, cu, SIZE_NxN);
          if (ub(t2.cost < best.cost)) { best = t2; st_copy(cold_state(COLD_NEXT, DEPTH), cold_state(COLD_TEMP, DEPTH)); }
          else { load_cand8(k, cu); cu.part = SIZE_2Nx2N; }
        }
        wsync();
        if (lane_id() == 0) {
          // what this CU's syntax cost in fractional bits: its count started from the fraction of the CU's entry state (reset_bits keeps the low 15 bits)
          const unsigned long long sf = cold_state(COLD_NEXT, DEPTH)[offsetof(Cabac, frac) / 8] - (s.curr[DEPTH].frac & 32767ull);
          lg[0] = (unsigned long long)__double_as_longlong(best.cost); lg[1] = (unsigned long long)best.bits | ((unsigned long long)best.dist << 32); lg[23] = sf;
          s.ctu_frac += sf;
        }
        st_copy(lg + 2, cold_state(COLD_NEXT, DEPTH));
        TL(12, 0);
      }
    } else { best.cost = MAX_DOUBLE / 16; best.dist = 0xffffffffu >> 3; best.bits = 0xffffffffu >> 3; }
    // split flag of the unsplit candidate (:858-867); for the dummy candidate the loaded state is stale and irrelevant
    st_copy(st_of(&s.go), cold_state(COLD_NEXT, DEPTH));
    const int sctx = (DEPTH < 3) ? split_ctx(k, x, y, DEPTH) : 0;
    if (lane_id() == 0) { reset_bits(&s.go); if (DEPTH < 3) enc_bin(&s.go, CTX_SPLIT + sctx, 0); }
    wsync();
    best.bits += (uint32_t)uni((int)get_bits(&s.go));
    best.cost = calc_rd_cost(k, best.bits, best.dist);
    st_copy(cold_state(COLD_NEXT, DEPTH), st_of(&s.go));
  }
  if (best_is_real) for (int c = uni(s.left_pending) ? 1 : 0; c < 3; c++) copy_best_rec_to_pic(k, cu, c);     // xCopyYuv2Pic :1093 (luma stays with a pass that is still running)
  if constexpr (DEPTH < 3) {
    Rd temp = { 0, 0, 0 };
    const int h = size >> 1, qn = cu.nparts >> 2;
    for (int i = 0; i < 4; i++) {
      const int sx = x + (i & 1) * h, sy = y + (i >> 1) * h;
      if (ub(sx < k.W && sy < k.H)) {
        if (i == 0) cabac_copy(k, &s.curr[DEPTH + 1], &s.curr[DEPTH]); else st_copy(st_of(&s.curr[DEPTH + 1]), cold_state(COLD_NEXT, DEPTH + 1));
        Rd sub;
        if (check_next) sub = compress_cu<DEPTH + 1>(k, sx, sy);
        else { sub.cost = MAX_DOUBLE / 16; sub.dist = 0xffffffffu >> 3; sub.bits = 0xffffffffu >> 3; }
        if (uni(s.restart)) return best;
        temp.cost += sub.cost; temp.dist += sub.dist; temp.bits += sub.bits;
      } else if (check_next || boundary) { // initSubCU defaults copied to the picture (:989)
        const int z0 = cu.zbase + i * qn;
        wsync();
        for (int j = lane_id(); j < qn; j += 64) {
          s.a[A_DEPTH][z0 + j] = DEPTH + 1; s.a[A_PART][z0 + j] = SIZE_NONE; s.a[A_LDIR][z0 + j] = DC; s.a[A_CDIR][z0 + j] = 0; s.a[A_TRIDX][z0 + j] = 0;
          for (int c = 0; c < 3; c++) { s.a[A_CBF + c][z0 + j] = 0; s.a[A_TSKIP + c][z0 + j] = 0; }
        }
        wsync();
      }
    }
    st_copy(st_of(&s.go), cold_state(COLD_NEXT, DEPTH + 1));
    if (!boundary) {
      const int sctx = split_ctx(k, x, y, DEPTH);
      if (lane_id() == 0) { reset_bits(&s.go); enc_bin(&s.go, CTX_SPLIT + sctx, 1); }
      wsync();
      temp.bits += (uint32_t)uni((int)get_bits(&s.go));
    }
    temp.cost = calc_rd_cost(k, temp.bits, temp.dist);
    if (ub(temp.cost < best.cost)) { best = temp; st_copy(cold_state(COLD_NEXT, DEPTH), st_of(&s.go)); }      // (the split's end state is in `go`)
  }
  return best;
}

// state-advancing encode of the decided CTU: encodeCtu/xEncodeCU TEncCu.cpp:290-304,1167-1271
template <int DEPTH> DEVN void encode_cu_tree(KR k, LCabac *c, int x_, int y_)
{
  const int x = uni(x_), y = uni(y_);
  LSmem &s = lds();
  const int size = 64 >> DEPTH;
  const int z = tb().r2z[(((y & 63) >> 2) << 4) | ((x & 63) >> 2)];
  int boundary = 0;
  const int dz = uni(s.a[A_DEPTH][z]);
  if (ub(x + size <= k.W && y + size <= k.H)) {
    if (DEPTH < 3) { const int sctx = split_ctx(k, x, y, DEPTH); if (lane_id() == 0) enc_bin(c, CTX_SPLIT + sctx, dz > DEPTH); }
  } else boundary = 1;
  if constexpr (DEPTH < 3) {
    if (DEPTH < dz || boundary) {
      const int h = size >> 1;
      for (int i = 0; i < 4; i++) { const int sx = x + (i & 1) * h, sy = y + (i >> 1) * h; if (ub(sx < k.W && sy < k.H)) encode_cu_tree<DEPTH + 1>(k, c, sx, sy); }
      return;
    }
  }
  const Cu cu = { x, y, 6 - DEPTH, DEPTH, z, 256 >> (2 * DEPTH), uni(s.a[A_PART][z]) };
  enc_cu_syntax(k, c, cu);
}

// The coder state behind a CTU WITHOUT coding it again (the reference does: encodeCtu with the RD coder, TEncSlice.cpp:886-893, xEncodeCU TEncCu.cpp:1167-1271).
// The state-advancing encode codes, in z-order, every split flag where its CU starts and every CU's syntax.  The search's own chain (compress_cu) coded the same
// CU syntax, CU by CU in the same order, each from the state its predecessor left -- plus the split flags, but those at other places of the sequence (behind
// a CU, behind the four children).  A bin only touches its own context, so every context but the three split-flag ones has seen exactly the encode's bins in
// the encode's order: their end states are the search's (s.next[0]), and the fractional bits of the CU syntax are the sums the search counted (ctu_frac).  What
// is left are the <= 21 split flags: replayed here in the encode's order on the split-flag contexts of the state the CTU started from.
DEVN void advance_state(KR k, LCabac *truec, int x0_, int y0_)
{
  const int x0 = uni(x0_), y0 = uni(y0_), lane = lane_id();
  LSmem &s = lds();
  // node of lane L in the encode's order: 0 the CTU, 1 + 5 i the 32x32 block i, 2 + 5 i + j its 16x16 block j
  int code = 0;
  if (lane < 21) {
    const int r5 = (lane - 1) % 5, i1 = (lane - 1) / 5, d = lane == 0 ? 0 : (r5 == 0 ? 1 : 2);
    auto dz_at = [&](int xx, int yy) { return (int)s.a[A_DEPTH][tb().r2z[(((yy & 63) >> 2) << 4) | ((xx & 63) >> 2)]]; };
    auto inside = [&](int xx, int yy, int size) { return xx + size <= k.W && yy + size <= k.H; };
    int x = x0, y = y0; bool ex = true;
    if (d >= 1) {
      ex = dz_at(x0, y0) > 0 || !inside(x0, y0, 64);                  // the walk goes below a CU that is split or reaches over the picture's edge
      x = x0 + (i1 & 1) * 32; y = y0 + (i1 >> 1) * 32;
      ex = ex && x < k.W && y < k.H;
      if (d == 2) {
        ex = ex && (dz_at(x, y) > 1 || !inside(x, y, 32));
        x += ((r5 - 1) & 1) * 16; y += ((r5 - 1) >> 1) * 16;
        ex = ex && x < k.W && y < k.H;
      }
    }
    if (ex && inside(x, y, 64 >> d)) code = 8 | split_ctx(k, x, y, d) | ((dz_at(x, y) > d) ? 4 : 0);
  }
  wsync();
  s.cgf[lane] = (uint8_t)code;
  wsync();
  if (lane == 0) {
    reset_bits(truec);
    for (int i = 0; i < 21; i++) { const int v = s.cgf[i]; if (v & 8) enc_bin(truec, CTX_SPLIT + (v & 3), (v >> 2) & 1); }
    truec->frac += s.ctu_frac;
  }
  { const auto nx = cold_state(COLD_NEXT, 0); wsync(); for (int i = 3 + lane; i < NUM_CTX; i += 64) truec->ctx[i] = (uint8_t)(nx[i >> 3] >> (8 * (i & 7))); }
  wsync();
}

// one unit = one (frame, tile): tiles are coded from a fresh coder state and see nothing of each other (TEncSlice.cpp:804-807)
// (a launch may cover only tiles [tile_begin, tile_begin + tile_count) of every frame: tile sharding across GPUs)
// Hand-over of a unit between workgroups (p.migrate): when the units do not divide evenly over the workgroups (600 frames on 256 CUs: 88
// workgroups walk three frames, 168 two), every frame is a serial chain and the launch lasts as long as the most crowded workgroup.  A
// master therefore offers its unit to the next workgroup of the ring at a CTU boundary whenever that one currently walks fewer units: the
// whole state of a unit between two CTUs is its position and the coder state (168 B) -- records and reconstruction are in HBM.  The surplus
// units keep travelling round the ring, every workgroup is crowded for the same share of the time, and so is every frame.
#ifndef HEVCDL_HOP
#define HEVCDL_HOP 4
#endif
struct Mbox { int state, unit, next_i, pad_; unsigned long long cabac[21]; };     // state: 0 empty, 2 being filled, 1 full, 3 being taken
DEV GLB int *sched_finished(const hevcdl_rd_params &p) { return (GLB int *)p.sched; }
DEV GLB int *sched_count(const hevcdl_rd_params &p, int g) { return (GLB int *)p.sched + 16 + g; }
DEV GLB Mbox *sched_mbox(const hevcdl_rd_params &p, int g) { return (GLB Mbox *)((GLB unsigned char *)p.sched + 8192) + g; }
DEV int glb_load_lane0(GLB int *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV int glb_load(GLB int *q) { int v = 0; if (lane_id() == 0) v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return uni(v); }
DEVN int glb_add(GLB int *q, int d) { int v = 0; if (lane_id() == 0) v = __hip_atomic_fetch_add(q, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return uni(v); }
DEVN int glb_cas(GLB int *q, int expect, int desired)
{ // 1 when the swap happened
  int ok = 0;
  if (lane_id() == 0) { int e = expect; ok = __hip_atomic_compare_exchange_strong(q, &e, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0; }
  return uni(ok);
}


// ---- second passes on a CU that has nothing to do ------------------------------------------------------------------------------------------------
// A launch of few units (one frame, C2's ten, a GPU's share of a sharded job) leaves most CUs empty, while a frame is bound by the work of its own CU's eight
// waves.  The deferred second pass of a PU is self-contained: start state, result slots, levels and reconstruction already live in HBM.  With
// hevcdl_rd_params.remote the kernel is launched on ALL CUs (cooperatively: the workgroups wait for each other); a master POSTS the pass -- header + the context
// a task takes over (kernel context, attribute arrays), an agent-scope release, a pointer in the ring at sched + 4096 -- and wave 0 of a workgroup without units
// takes it, runs it as the chain owner of ITS workgroup (the split alternatives go to that workgroup's other waves: spec_children) and answers through the block.
// Memory: the workgroups sit on different XCDs, whose L2s are not coherent with each other.  Poster: release before the pointer.  Taker: acquire before the
// context is read, release before the answer.  Joiner: acquire behind the answer; an acquire drops clean lines only, and a 128-byte line of the picture can hold
// samples of the pass's CU next to samples this XCD wrote meanwhile (dirty here, stale in the pass's part) -- so after a pass that changed nothing the joiner
// writes the first pass's samples itself (what the pass restored: same values), and after one that chose the split it writes its own dirty lines back first
// (release) and drops them (acquire): the next read comes from memory, where both parts are.
DEV GLB unsigned long long *job_block(int pset) { return lds().my_log + (size_t)(LOG_JOB + LOG_JOB_N * pset) * (LEAF_LOG / 8); }
DEV GLB int *rq_tail(GLB unsigned char *sched) { return (GLB int *)(sched + 2048); }      // posters add here, takers there: lines of their own
DEV GLB int *rq_head(GLB unsigned char *sched) { return (GLB int *)(sched + 2304); }
DEV GLB unsigned long long *rq_ring(GLB unsigned char *sched) { return (GLB unsigned long long *)(sched + 4096); }
DEV GLB int *rq_idle(GLB unsigned char *sched) { return (GLB int *)(sched + 2560); }     // takers that are not running a job
// hevcdl_rd_params.remote == 3 (more units than takers): a pass is posted only while some taker has nothing to do and nothing waits in the ring; otherwise it
// stays in its own workgroup, as in a launch without takers
DEVN int remote_room(int need_)
{
  const int need = uni(need_);
  int ok = 0;
  if (lane_id() == 0) {
    GLB unsigned char *sched = wg_shared().sched;
    const int idle = __hip_atomic_load(rq_idle(sched), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int tail = __hip_atomic_load(rq_tail(sched), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), head = __hip_atomic_load(rq_head(sched), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = idle - (tail - head) >= need;
  }
  return uni(ok);
}
enum { RQ_SIZE = 2048, JOB_DONE = 0, JOB_DIST = 1, JOB_COST = 2, JOB_CU = 4, JOB_TU = 11, JOB_MODE = 17, JOB_PSET = 18, JOB_MDIST = 19, JOB_MCOST = 20, JOB_KIND = 22, JOB_IDX = 23,
       JOB_CFRAC = 24, JOB_CTXP = 26, JOB_SPLIT = 28, JOB_PAIR = 29, JOB_HALF = 30, JOB_COUNTED = 31,       // int offsets in the header (8-byte values at even offsets)
       JOB_CTX = 24 };                        // 8-byte-word offset of a second pass's own context behind its header
DEV GLB unsigned long long *cjob_block(int m) { return lds().my_log + (size_t)(LOG_CJOB + m) * (LEAF_LOG / 8); }
// Two takers on different XCDs write into neighbouring job headers (a chroma mode's two components: distortion, counter, answer, done flag -- the first 128 bytes of a
// 192-byte log entry).  Their L2s only merge dirty BYTES of a line, and an acquire keeps a line its own XCD has written to, so no two headers may share a 128-byte line:
// with entries 192 bytes apart and the first one 0 or 64 bytes into a line, header e covers bytes [o + 192 e, o + 192 e + 128) and never ends in the line the next begins in.
static_assert(LEAF_LOG == 192 && (LOG_CJOB * LEAF_LOG) % 64 == 0 && (BD != 8 || (4 * 6144 * 2 + 6 * 6144 * (int)sizeof(pel_t) + N_SAVE * SAVE_BYTES) % 128 == 0) && SCR_WAVE % 128 == 0,
              "job headers: the log starts on a line, every header 0 or 64 bytes into one");
static_assert(JOB_COUNTED * 4 + 4 <= 128, "job header fits the first 128 bytes of its log entry");
DEVN void remote_post(KR k, const Cu cu_, const Tu tu_, int reg_, int mode_, double memo_cost, uint32_t memo_dist, int with_chroma_)
{
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_); const int reg = uni(reg_), mode = uni(mode_), pset = reg - 1, with_chroma = uni(with_chroma_);
  LSmem &s = lds(); LRegion &r = my_region(reg);
  GLB unsigned long long *job = job_block(pset); GLB int *hi = (GLB int *)job;
  static_assert(JOB_CTX * 8 <= LEAF_LOG && sizeof(K) + 11 * 256 <= LOG_AHEAD_N * LEAF_LOG, "job block");
  wsync();
  { GLB unsigned long long *d = job + JOB_CTX;
    LDS const unsigned long long *qk = (LDS const unsigned long long *)&s.k, *qa = (LDS const unsigned long long *)&s.a[0][0];
    for (int i = lane_id(); i < (int)(sizeof(K) / 8); i += 64) d[i] = qk[i];
    for (int i = lane_id(); i < 11 * 256 / 8; i += 64) d[sizeof(K) / 8 + i] = qa[i]; }
  if (lane_id() == 0) {
    hi[JOB_DONE] = 0; hi[JOB_MODE] = mode; hi[JOB_PSET] = pset; hi[JOB_MDIST] = (int)memo_dist; *(GLB double *)(hi + JOB_MCOST) = memo_cost;
    hi[JOB_KIND] = T_LUMA_P2; hi[JOB_IDX] = 0; *(GLB unsigned long long *)(hi + JOB_CTXP) = (unsigned long long)(job + JOB_CTX);
    hi[JOB_CU] = cu.x; hi[JOB_CU + 1] = cu.y; hi[JOB_CU + 2] = cu.log2; hi[JOB_CU + 3] = cu.depth; hi[JOB_CU + 4] = cu.zbase; hi[JOB_CU + 5] = cu.nparts; hi[JOB_CU + 6] = cu.part;
    hi[JOB_TU] = tu.x; hi[JOB_TU + 1] = tu.y; hi[JOB_TU + 2] = tu.log2; hi[JOB_TU + 3] = tu.trd; hi[JOB_TU + 4] = tu.zrel; hi[JOB_TU + 5] = tu.nparts;
    r.kind = T_REMOTE; r.owner = wave_id(); r.done = 0;               // no ticket: nobody in this workgroup claims it
    r.cu[0] = cu.x; r.cu[1] = cu.y; r.cu[2] = cu.log2; r.cu[3] = cu.depth; r.cu[4] = cu.zbase; r.cu[5] = cu.nparts; r.cu[6] = cu.part;
  }
  wsync();
  if (with_chroma) { // the CU's chroma modes ride on the same release (the CU is one PU: its luma mode and TU arrays are final unless this very pass chooses the split -- then the chroma search is repeated anyway)
    uint32_t ml[5]; chroma_mode_list(mode, ml);
    const Tu root = { cu.x, cu.y, cu.log2, 0, 0, cu.nparts };
    chroma_post(k, cu, root, (int)ml[0], (int)ml[1], (int)ml[2], (int)ml[3], (int)ml[4], 1);
    if (lane_id() == 0) s.chroma_key = (cu.log2 << 24) | (cu.y << 12) | cu.x;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                  // the block, the start state in the slot, the picture around the CU: before the pointer
  { // one reservation for all the jobs (a taker that finds a reserved entry still empty looks again later), the pointers stored side by side
    GLB unsigned char *sched = wg_shared().sched;
    const int n = with_chroma ? 1 + uni(s.chroma_jobs) : 1;
    int i0 = 0;
    if (lane_id() == 0) i0 = __hip_atomic_fetch_add(rq_tail(sched), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    i0 = uni(i0);
    if (lane_id() < n) __hip_atomic_store(rq_ring(sched) + ((i0 + lane_id()) & (RQ_SIZE - 1)), (unsigned long long)(lane_id() == 0 ? job : cjob_block(lane_id() - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  wsync();
}
// The five chroma modes of a CU as jobs (launches of very few units: there are enough idle workgroups for every mode of every master).  A mode's trial levels and
// reconstruction go to its result slot, never to the picture, so the only lines two XCDs write are slot lines -- the release below also writes back what this XCD
// still holds dirty of them (first-pass candidates used slots 5..9 before), the acquire in chroma_collect drops the then clean copies.
DEVN void chroma_post(KR k, const Cu cu_, const Tu tu_, int m0, int m1, int m2, int m3, int m4, int prepare_only)
{
  const Cu cu = ucu(cu_); const Tu tu = utu(tu_);
  LSmem &s = lds();
  GLB unsigned long long *ctx = cjob_block(10);
  static_assert(sizeof(K) + 11 * 256 + sizeof(Cabac) <= (LOG_AHEAD_N + 1) * LEAF_LOG, "chroma job context");
  // a CU of one TU per component (no transform-skip trial): Cb and Cr of a mode are jobs of their own, the component that finishes second counts the mode's bits (run_task_body)
  const int split = (cu.log2 >= 4 && cu.log2 <= 5 && uni(s.a[A_TRIDX][cu.zbase]) == 0) ? 1 : 0, n = split ? 10 : 5;
  wsync();
  { LDS const unsigned long long *qk = (LDS const unsigned long long *)&s.k, *qa = (LDS const unsigned long long *)&s.a[0][0], *qs = (LDS const unsigned long long *)&s.curr[cu.depth];
    for (int i = lane_id(); i < (int)(sizeof(K) / 8); i += 64) ctx[i] = qk[i];
    for (int i = lane_id(); i < 11 * 256 / 8; i += 64) ctx[sizeof(K) / 8 + i] = qa[i];
    if (lane_id() < 21) ctx[sizeof(K) / 8 + 11 * 256 / 8 + lane_id()] = qs[lane_id()]; }
  if (lane_id() < n) {
    const int j = lane_id(), m = split ? j >> 1 : j, mode = m == 0 ? m0 : (m == 1 ? m1 : (m == 2 ? m2 : (m == 3 ? m3 : m4)));
    GLB int *hi = (GLB int *)cjob_block(j);
    hi[JOB_DONE] = 0; hi[JOB_MODE] = mode; hi[JOB_PSET] = 0; hi[JOB_KIND] = T_CHROMA; hi[JOB_IDX] = j; *(GLB unsigned long long *)(hi + JOB_CTXP) = (unsigned long long)ctx;
    hi[JOB_SPLIT] = split; hi[JOB_PAIR] = 0; hi[JOB_HALF] = 0; hi[JOB_COUNTED] = 0;
    hi[JOB_CU] = cu.x; hi[JOB_CU + 1] = cu.y; hi[JOB_CU + 2] = cu.log2; hi[JOB_CU + 3] = cu.depth; hi[JOB_CU + 4] = cu.zbase; hi[JOB_CU + 5] = cu.nparts; hi[JOB_CU + 6] = cu.part;
    hi[JOB_TU] = tu.x; hi[JOB_TU + 1] = tu.y; hi[JOB_TU + 2] = tu.log2; hi[JOB_TU + 3] = tu.trd; hi[JOB_TU + 4] = tu.zrel; hi[JOB_TU + 5] = tu.nparts;
  }
  if (lane_id() == 0) s.chroma_jobs = n;
  wsync();
  if (uni(prepare_only)) return;                                      // the caller releases and pushes (remote_post)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  {
    GLB unsigned char *sched = wg_shared().sched;
    int i0 = 0;
    if (lane_id() == 0) i0 = __hip_atomic_fetch_add(rq_tail(sched), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    i0 = uni(i0);
    if (lane_id() < n) __hip_atomic_store(rq_ring(sched) + ((i0 + lane_id()) & (RQ_SIZE - 1)), (unsigned long long)cjob_block(lane_id()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  wsync();
}
DEVN void chroma_collect(LRegion &r)
{ // wait for the answers; they go where the tasks of a local chroma region leave theirs
  const int n = uni(lds().chroma_jobs);
  for (;;) {
    int d = 1;
    if (lane_id() < n) d = __hip_atomic_load((GLB int *)cjob_block(lane_id()) + JOB_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__ballot(d == 0) == 0ull) break;
    __builtin_amdgcn_s_sleep(48);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  wsync();
  if (lane_id() < 5) {
    GLB const int *hi = (GLB const int *)cjob_block(n == 10 ? 2 * lane_id() : lane_id());
    if (n == 10 && !hi[JOB_COUNTED]) hi = (GLB const int *)cjob_block(2 * lane_id() + 1);      // the component that finished second carries the mode's answer
    r.modes[lane_id()] = hi[JOB_MODE]; r.dist[lane_id()] = (uint32_t)hi[JOB_DIST]; r.cost[lane_id()] = *(GLB const double *)(hi + JOB_COST); r.cfrac[lane_id()] = *(GLB const unsigned long long *)(hi + JOB_CFRAC);
  }
  wsync();
}
DEVN int remote_poll(LRegion &r)
{ // the master asks for the answer of the pass it posted from region r: 1 when it is there (r.cost[0], r.dist[0], r.done as a local pass would leave them)
  const int pset = uni(r.modes[1]);
  GLB int *hi = (GLB int *)job_block(pset);
  int d = 0;
  if (lane_id() == 0) d = __hip_atomic_load(hi + JOB_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!uni(d)) return 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const double cost = *(GLB const double *)(hi + JOB_COST); const int dist = hi[JOB_DIST];
  wsync();
  if (lane_id() == 0) { r.cost[0] = cost; r.dist[0] = (uint32_t)dist; }
  wsync();
  if (ub(r.cost[0] < r.cost[4])) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }   // see above
  if (lane_id() == 0) r.done = 1;
  wsync();
  return 1;
}
// wave 0 of a workgroup without units: one posted pass, if there is one
DEVN int remote_serve(GLB unsigned char *sched_)
{
  GLB unsigned char *sched = sched_;
  LSmem &s = lds();
  unsigned long long jv = 0;
  if (lane_id() == 0) {
    int h = __hip_atomic_load(rq_head(sched), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long v = __hip_atomic_load(rq_ring(sched) + (h & (RQ_SIZE - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v && __hip_atomic_compare_exchange_strong(rq_head(sched), &h, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      jv = v;                                                          // head == h when v was read non-zero: the entry of this lap
      __hip_atomic_store(rq_ring(sched) + (h & (RQ_SIZE - 1)), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  jv = uni64(jv);
  if (!jv) return 0;
  if (lane_id() == 0) __hip_atomic_fetch_add(rq_idle(sched), -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  GLB unsigned long long *job = (GLB unsigned long long *)jv; GLB int *hi = (GLB int *)job;
  const int kind = uni(hi[JOB_KIND]), idx = uni(hi[JOB_IDX]), depth = uni(hi[JOB_CU + 3]);
  { // the poster's context (import_owner's part, from the block)
    GLB const unsigned long long *q = (GLB const unsigned long long *)uni64(*(GLB const unsigned long long *)(hi + JOB_CTXP));
    wsync();
    { LDS unsigned long long *d = (LDS unsigned long long *)&s.k; for (int i = lane_id(); i < (int)(sizeof(K) / 8); i += 64) d[i] = q[i]; }
    { LDS unsigned long long *d = (LDS unsigned long long *)&s.a[0][0]; for (int i = lane_id(); i < 11 * 256 / 8; i += 64) d[i] = q[sizeof(K) / 8 + i]; }
    if (kind == T_CHROMA && lane_id() < 21) ((LDS unsigned long long *)&s.curr[depth])[lane_id()] = q[sizeof(K) / 8 + 11 * 256 / 8 + lane_id()];    // the snapshot the modes start from
    wsync();
    s.k.q_cost = s.my_qcost; s.k.q_rate = s.my_qrate; s.k.ovl = s.my_ovl;
    if (lane_id() < 3) s.ref_key[lane_id()] = -1;
    if (lane_id() == 0) s.fline_key = -1;
    wsync();
  }
  LRegion &r = my_region(0);                                           // stands in for the poster's ticket region
  if (lane_id() == 0) {
    r.kind = kind; r.owner = wave_id(); r.done = 0; r.ticket = 0; r.pad_ = 0;
    for (int i = 0; i < 7; i++) r.cu[i] = hi[JOB_CU + i];
    for (int i = 0; i < 6; i++) r.tu[i] = hi[JOB_TU + i];
    if (kind == T_LUMA_P2) { r.modes[0] = hi[JOB_MODE]; r.modes[1] = hi[JOB_PSET]; r.dist[4] = (uint32_t)hi[JOB_MDIST]; r.cost[4] = *(GLB const double *)(hi + JOB_MCOST); }
    else r.modes[idx] = hi[JOB_MODE];
  }
  const int split = kind == T_CHROMA && uni(hi[JOB_SPLIT]);
  if (split && lane_id() == 0) { // one component of a chroma mode: job idx = 2 * mode index + component; its twin's header is the neighbouring log entry
    r.pad_ = 1; r.modes[idx >> 1] = hi[JOB_MODE]; r.cost[idx >> 1] = -1.0;
    GLB int *h0 = (idx & 1) ? hi - LEAF_LOG / 4 : hi, *ho = (idx & 1) ? hi - LEAF_LOG / 4 : hi + LEAF_LOG / 4;
    s.rp_pair = h0 + JOB_PAIR; s.rp_half_mine = (GLB uint32_t *)(hi + JOB_HALF); s.rp_half_other = (GLB uint32_t *)(ho + JOB_HALF);
  }
  wsync();
  run_task<false>(r, idx);
  wsync();
  if (split) {
    if (lane_id() == 0) {
      const int m = idx >> 1, counted = r.cost[m] >= 0.0;
      hi[JOB_COUNTED] = counted;
      if (counted) { hi[JOB_DIST] = (int)(unsigned)r.cfrac[5 + m]; *(GLB double *)(hi + JOB_COST) = r.cost[m]; *(GLB unsigned long long *)(hi + JOB_CFRAC) = r.cfrac[m]; }
      s.rp_pair = nullptr;
    }
  } else if (lane_id() == 0) { hi[JOB_DIST] = (int)r.dist[idx]; *(GLB double *)(hi + JOB_COST) = r.cost[idx]; *(GLB unsigned long long *)(hi + JOB_CFRAC) = r.cfrac[idx]; }
  wsync();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                  // answer, arrays, levels, samples: before the flag
  if (lane_id() == 0) { __hip_atomic_store(hi + JOB_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(rq_idle(sched), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  wsync();
  return 1;
}

// slice start: context init from QP (ContextModel.cpp:56-66, TEncSlice.cpp:719-720), fraction 0 (TEncBinCoderCABAC.cpp:69-79)
DEV void slice_start_contexts(LCabac *c, int qp)
{
  const int lane = lane_id();
  for (int i = lane; i < NUM_CTX; i += 64) {
    const int v = c_ctx_init[i], slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
    int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
    const int mps = st >= 64;
    c->ctx[i] = (uint8_t)(((mps ? st - 64 : 63 - st) << 1) + mps);
  }
  if (lane == 0) { c->ctx[159] = 0; c->frac = 0; }
}

// WaveFrontSynchro, rows claimed (hevcdl_dev.h wpp_masters): -> a unit (frame * ctus_y + row) this wave now owns, -1: nothing can start right now, -2: every row has an owner.
// Ready rows first (their frame is under way), then the first row of a frame nobody has started.
DEVN int wpp_claim(const hevcdl_rd_params &p, int n_units)
{
  int r = -1;
  if (lane_id() == 0) {
    GLB int *q = (GLB int *)p.wpp_queue;
    int h = __hip_atomic_load(q + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int v = __hip_atomic_load(q + 256 + (h & (p.wpp_ring - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v && __hip_atomic_compare_exchange_strong(q + 32, &h, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      __hip_atomic_store(q + 256 + (h & (p.wpp_ring - 1)), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (head == h when v was read non-zero: the entry of this lap)
      r = v - 1;
    } else if (__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.n_frames) {
      const int f = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (f < p.n_frames) r = f * p.ctus_y;
    }
    if (r >= 0) __hip_atomic_fetch_add(q + 96, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (__hip_atomic_load(q + 96, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_units) r = -2;
  }
  return uni(r);
}

// -> 0: the unit is finished, 1: handed over to the next workgroup.  i_resume >= 0: continue a unit taken from this workgroup's mailbox.
DEV int process_unit(const hevcdl_rd_params &p, int unit, int i_resume)
{
  LSmem &s = lds();
  const int wpp = p.wpp;                // WaveFrontSynchro: 1 = this unit is one CTU row of its frame, 2 = a whole frame with synchronised rows (hevcdl_dev.h)
  const int ntiles = p.tile_cols * p.tile_rows, frame = unit / p.tile_count, tile = wpp ? 0 : p.tile_begin + (unit - frame * p.tile_count);
  const int wrows = wpp ? p.ctus_y / p.tile_count : 0, wrow0 = wpp ? (unit - frame * p.tile_count) * wrows : 0;      // the unit's CTU rows [wrow0, wrow0 + wrows)
  GLB unsigned char *wstate = wpp ? (GLB unsigned char *)p.wpp_state + (size_t)frame * p.ctus_y * 256 : nullptr;
  int wpp_seen = 0;                     // CTUs of the row above known to be finished (and made visible by an acquire)
  LDS K &k = s.k;                       // every lane stores the same values
  const int lane = lane_id();
  k.W = p.width; k.H = p.height; k.cw = p.width >> 1; k.ctus_x = p.ctus_x; k.nctu = p.ctus_x * p.ctus_y;
  const size_t ysz = (size_t)p.width * p.height, csz = ysz >> 2, fsz = ysz + 2 * csz;
  const int nctu = p.ctus_x * p.ctus_y;
  GLB const pel_t *org0 = (GLB const pel_t *)p.yuv + (size_t)frame * fsz;
  GLB pel_t *rec0 = (GLB pel_t *)p.recon + (size_t)frame * fsz;
  k.org[0] = org0; k.org[1] = org0 + ysz; k.org[2] = org0 + ysz + csz;
  k.rec[0] = rec0; k.rec[1] = rec0 + ysz; k.rec[2] = rec0 + ysz + csz;
  GLB unsigned char *records = (GLB unsigned char *)p.records + (size_t)frame * nctu * REC_SIZE;
  k.records = records;
  k.labels = (GLB const uint8_t *)p.labels + (size_t)frame * nctu * 16;
  k.coef_l = s.my_coef; k.rec_l = s.my_rec; k.best_rec = s.my_rec + 4 * 6144; k.ovl = s.my_ovl;
  k.q_cost = s.my_qcost; k.q_rate = s.my_qrate; k.slots = s.my_slots;        // (a wave that served other masters' tasks holds their context)
  k.in_task = 0; k.trx0 = k.try0 = k.trx1 = k.try1 = 0; k.lz = k.lx = k.ly = 0; k.srect[0] = k.srect[1] = k.srect[2] = 0; k.ssrc[0] = k.ssrc[1] = k.ssrc[2] = k.best_rec; k.sorg[0] = k.sorg[1] = k.sorg[2] = k.sorg[3] = 0; k.pset = 0; k.pad_pset = 0;
  k.lambda = p.k.lambda; k.sqrt_lambda = p.k.sqrt_lambda; k.cweight = p.k.chroma_weight; k.lambda_c = p.k.lambda_chroma;
  for (int a = 0; a < 2; a++) { for (int b = 0; b < 4; b++) k.err_scale[a][b] = p.k.err_scale[a][b]; k.sbh[a] = p.k.sbh_rd_factor[a]; }
  k.qp = p.k.qp; k.qp_c = p.k.qp_chroma; k.dbg = p.debug; k.dbgbuf = (GLB unsigned int *)p.dbgbuf; k.tools = p.k.tools;
  if (lane == 0) { s.est_bits = 0; s.sse_acc[0] = s.sse_acc[1] = s.sse_acc[2] = 0; s.carry_ok = 0; s.restart = 0; s.pend_n = 0; s.p2_pending = 0; s.left_pending = 0; }
  wsync();
  // slice start: context init from QP (ContextModel.cpp:56-66, TEncSlice.cpp:719-720); the true coder of TEncSlice.cpp:719.
  // A per-CTU call (hevcdl_compress_ctu) resumes from the state the previous call left instead.
  LCabac *truec = &s.truec;
  if (i_resume >= 0) { // the coder state travels with the unit; taking it frees the mailbox
    GLB Mbox *mb = sched_mbox(p, (int)blockIdx.x);
    if (lane < 21) ((LDS unsigned long long *)truec)[lane] = mb->cabac[lane];
    wsync();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(&mb->state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (p.cabac_in) {
    GLB const unsigned long long *src = (GLB const unsigned long long *)p.cabac_in + (size_t)frame * 21;
    if (lane < 21) ((LDS unsigned long long *)truec)[lane] = src[lane];
  } else slice_start_contexts(truec, p.k.qp);
  wsync();

  // tile rectangle from the host's boundary tables (TComPicSym.cpp xInitTiles); CTUs of the tile in raster order.  Without tiles the
  // caller may give a CTU range.
  const int tcx = tile % p.tile_cols, tcy = tile / p.tile_cols;
  const int cx0 = p.col_bd[tcx], cx1 = p.col_bd[tcx + 1], cy0 = p.row_bd[tcy], cy1 = p.row_bd[tcy + 1];
  const int tw = cx1 - cx0;
  k.tx0 = cx0 * 64; k.ty0 = cy0 * 64; k.tx1 = cx1 * 64; k.ty1 = cy1 * 64;
  // (a wavefront unit's rows, cut to the caller's CTU range: the per-CTU session walks one CTU a launch in the one-wave form)
  const int i_begin = wpp ? (wrow0 * tw > p.ctu_begin ? wrow0 * tw : p.ctu_begin) : (i_resume >= 0 ? i_resume : (ntiles == 1 ? p.ctu_begin : 0));
  const int i_end = wpp ? ((wrow0 + wrows) * tw < p.ctu_end ? (wrow0 + wrows) * tw : p.ctu_end) : (ntiles == 1 ? p.ctu_end : tw * (cy1 - cy0));
  for (int i = i_begin; i < i_end; i++) {
    if (p.migrate && i > i_begin && ((i - i_begin) & (HEVCDL_HOP - 1)) == 0) { // every HEVCDL_HOP CTUs: does the next workgroup of the ring walk fewer units than this one?
      const int g = (int)blockIdx.x, ng = (g + 1) % (int)gridDim.x;
      if (lds_load(&wg_shared().masters_active) > glb_load(sched_count(p, ng)) && glb_cas(&sched_mbox(p, ng)->state, 0, 2)) {
        GLB Mbox *mb = sched_mbox(p, ng);
        wsync();
        if (lane < 21) mb->cabac[lane] = ((LDS const unsigned long long *)truec)[lane];
        if (lane == 0) { mb->unit = unit; mb->next_i = i; }
        if (p.stats && lane == 0) __hip_atomic_fetch_add(&((GLB hevcdl_frame_stats *)p.stats + frame)->est_bits, (unsigned long long)s.est_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wsync();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // records, reconstruction and the mailbox before the flag: the taker runs on another XCD
        glb_add(sched_count(p, ng), 1); glb_add(sched_count(p, g), -1);
        if (lane == 0) __hip_atomic_store(&mb->state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 1;
      }
    }
    const int cx = cx0 + i % tw, cy = cy0 + i / tw, a = cy * p.ctus_x + cx;
    wsync();
    if (wpp && cy > 0) { // WaveFrontSynchro (TEncSlice.cpp:783-830): this CTU reads the row above up to the CTU above and to the right, a row starts from the contexts behind that CTU
      if (cx == 0) wpp_seen = 0;
      const int need = cx + 2 < tw ? cx + 2 : tw;
      if (wpp == 1 && wpp_seen < need) { // the row above is walked by another wave, most likely on another XCD: its finished-CTU count (agent scope), then an acquire
        GLB int *prog = (GLB int *)(wstate + (size_t)(cy - 1) * 256 + 192);
        PROF_T0();
        // (a look every ~7 us: a CTU takes a wave milliseconds, and thousands of waiting waves must not crowd the fabric with their polls)
        while ((wpp_seen = glb_load(prog)) < need) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        PROF_ADD(k, 23);
      }
      if (cx == 0) { // resetEntropy, then the contexts -- not the fraction -- behind the second CTU of the row above if the picture has one (:808-823)
        slice_start_contexts(truec, p.k.qp);
        wsync();
        if (tw >= 2) { GLB const unsigned int *src = (GLB const unsigned int *)(wstate + (size_t)(cy - 1) * 256); if (lane < 40) ((LDS unsigned int *)truec->ctx)[lane] = src[lane]; }
        wsync();
        if (lane == 0) truec->ctx[159] = 0;
        wsync();
      }
    }
    PROF_MARK0();
    k.addr = a; k.cx = cx; k.cy = cy;
    if (lane < 16) s.lab16[lane] = k.labels[a * 16 + lane];
    // initCtu TComDataCU.cpp:420-500
    for (int i = lane; i < 256; i += 64) {
      for (int f = 0; f < 11; f++) s.a[f][i] = 0;
      s.a[A_PART][i] = SIZE_NONE; s.a[A_LDIR][i] = DC;
    }
    { // coefficient arrays of the record start at zero (initCtu memset)
      GLB uint32_t *rc = (GLB uint32_t *)(records + (size_t)a * REC_SIZE + REC_COEF);
      for (int i = lane; i < 6144 / 2; i += 64) rc[i] = 0;
    }
    cabac_copy(k, &s.curr[0], truec);                         // TEncSlice.cpp:826-832
    cabac_copy(k, &s.go, truec);
    if (lane == 0) {
      s.leaf_idx = 0; s.replay_upto = 0; s.nocarry_leaf = -1; s.resume_reg = 0; s.pend_n = 0; s.restart = 0; s.ctu_frac = 0; s.chroma_key = 0;
      if (!(i > i_begin && s.xctu == a)) { s.pre_key = -1; s.pre_open = 0; s.ahead_open = 0; s.ahead_key = -1; }      // (a look-ahead opened for this very CTU while the one before was finished stays)
      s.xctu = -1;
    }
    wsync();
    PROF_MARK(47);
    TL(13, a);
    Rd best;
    for (;;) {
      best = compress_cu<0>(k, cx * 64, cy * 64);
      TL(14, uni(s.pend_n));
      // Second passes are still out and this wave is about to wait for them, its workgroup's other waves idle: the look-ahead for the FIRST CU of the next CTU -- its
      // rough-mode sums, then its first-pass candidates.  Nothing of this CTU is touched (the candidates get a context of their own naming the next CTU, ahead_open);
      // what they assume is checked when the CU is reached (est_intra_luma), and a restart of this CTU drops them like any other look-ahead.
      // (WaveFrontSynchro: the next CTU's first CU reads the row above one CTU further to the right -- only when that is known to be finished already, never into a row start)
      if (AHEAD && HEVCDL_PREFETCH && NPEND >= 2 && !p.migrate && !uni(s.restart) && uni(s.pend_n) && i + 1 < i_end && uni(s.lw_valid) && !uni(s.ahead_open) && !uni(s.pre_open) &&
          (!wpp || (cx + 1 < tw && (cy == 0 || wpp == 2 || wpp_seen >= (cx + 3 < tw ? cx + 3 : tw)))) &&
          lds_load(&wg_shared().masters_active) <= HEVCDL_AHEAD_MAX && spare_waves()) {
        const int ni = i + 1, ncx = cx0 + ni % tw, ncy = cy0 + ni / tw, na = ncy * p.ctus_x + ncx;
        int nx = 0, ny = 0, nl = 0;
        if (ctu_first_leaf(k, na, ncx, ncy, nx, ny, nl) && nl >= 4 && nl <= 5) {
          { // the candidates read their left neighbours' modes in this CTU's record: its attribute arrays as they stand (written again behind the joins)
            GLB unsigned char *rec = records + (size_t)a * REC_SIZE;
            wsync();
            for (int q = lane; q < 11 * 256 / 4; q += 64) ((GLB uint32_t *)rec)[q] = ((LDS const uint32_t *)&s.a[0][0])[q];
            wsync();
          }
          rmd_prefetch(k, nx, ny, nl, 1);
          const Cu none = { 0, 0, 0, 0, 0, 0, 0 };
          ahead_open(k, none, nx, ny, nl, na, ncx, ncy);
          if (lane == 0) s.xctu = na;
          wsync();
          TL(17, na);
        }
      }
      while (!uni(s.restart) && uni(s.pend_n)) pend_join_oldest(k, 1);      // passes still pending at the end of the CTU
      if (!uni(s.restart)) break;
      // a pending second pass chose the split: the walk again, replaying the CUs before its own (compress_cu)
      wsync();
#ifdef HEVCDL_KERNEL_PROF
      if (lane == 0) PROF_ACC_(38, (unsigned long long)(s.leaf_idx - s.replay_upto) << 10);   // restarts, CUs thrown away
#endif
      if (lane < 3) s.ref_key[lane] = -1;
      ahead_drain();
      if (uni(s.pre_open)) { region_run(k, my_region()); if (lane == 0) s.pre_open = 0; wsync(); }      // SATD slices still out
      if (uni(s.chroma_key)) { chroma_collect(my_region()); if (lane == 0) s.chroma_key = 0; wsync(); }  // (chroma modes posted for a CU whose search was abandoned: cannot happen by construction)
      if (lane == 0) { s.fline_key = -1; s.restart = 0; s.leaf_idx = 0; s.carry_ok = 0; s.p2_pending = 0; s.left_pending = 0; s.pre_key = -1; s.ctu_frac = 0; s.xctu = -1; }
      wsync();
      cabac_copy(k, &s.curr[0], truec);
      cabac_copy(k, &s.go, truec);
    }
    PROF_MARK(45);
    TL(15, 0);
    // the coder state behind the CTU (the reference's state-advancing encode, TEncSlice.cpp:886-893) + end_of_slice_segment_flag = 0 (finishCU TEncCu.cpp:1112-1128)
    wsync();
    advance_state(k, truec, cx * 64, cy * 64);
    if (lane == 0) { if (a != nctu - 1) truec->frac += (unsigned long long)tb().t_ebits[126]; s.est_bits += truec->frac >> 15; }
    TL(16, 0);
    PROF_MARK(46);
    // flush the CTU record
    GLB unsigned char *rec = records + (size_t)a * REC_SIZE;
    for (int i = lane; i < 11 * 256 / 4; i += 64) ((GLB uint32_t *)rec)[i] = ((LDS const uint32_t *)&s.a[0][0])[i];
    if (lane == 0) {
      *(GLB uint32_t *)(rec + REC_BITS) = best.bits; *(GLB uint32_t *)(rec + REC_DIST) = best.dist;
      *(GLB double *)(rec + REC_COST) = best.cost;
    }
    wsync();
    if (wpp) { // the contexts behind the row's second CTU for the row below (TEncSlice.cpp:925-928), then the row's progress: record, samples and contexts before the count
      if (cx == 1 && lane < 40) ((GLB unsigned int *)(wstate + (size_t)cy * 256))[lane] = ((LDS const unsigned int *)truec->ctx)[lane];
      if (wpp == 1) {
        wsync();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) {
          __hip_atomic_store((GLB int *)(wstate + (size_t)cy * 256 + 192), cx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p.wpp_masters && cx + 1 == (tw >= 2 ? 2 : 1) && cy + 1 < p.ctus_y) { // the row below can start now: into the ring of ready rows (the count above is already out)
            GLB int *q = (GLB int *)p.wpp_queue;
            const int t = __hip_atomic_fetch_add(q + 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + 256 + (t & (p.wpp_ring - 1)), frame * p.ctus_y + cy + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // unit + 1: zero marks an empty entry
          }
        }
      }
      wsync();
    }
    PROF_MARK(47);
  }
  if (p.cabac_out) {
    wsync();
    GLB unsigned long long *dstc = (GLB unsigned long long *)p.cabac_out + (size_t)frame * 21;
    if (lane < 21) dstc[lane] = ((LDS const unsigned long long *)truec)[lane];
  }
  if (p.stats) { // per-frame summary: SSE per plane (lane-parallel) + estimated bits; every tile adds its rectangle (the host zeroed the entry)
    GLB hevcdl_frame_stats *st = (GLB hevcdl_frame_stats *)p.stats + frame;
    for (int c = 0; c < 3; c++) {
      const int uy0 = wpp ? wrow0 * 64 : k.ty0, uy1 = wpp ? (wrow0 + wrows) * 64 : k.ty1;        // the unit's sample rows: its tile, or its CTU rows
      const int sh = c ? 1 : 0, ps = p.width >> sh, rx0 = k.tx0 >> sh, ry0 = uy0 >> sh;
      const int rw = ((k.tx1 < p.width ? k.tx1 : p.width) >> sh) - rx0, rh = ((uy1 < p.height ? uy1 : p.height) >> sh) - ry0;
      unsigned long long acc = 0;
      GLB const pel_t *po = org0 + (c == 0 ? 0 : (c == 1 ? ysz : ysz + csz)); GLB const pel_t *pr = rec0 + (c == 0 ? 0 : (c == 1 ? ysz : ysz + csz));
      for (int yy = 0; yy < rh; yy++) {
        const size_t o = (size_t)(ry0 + yy) * ps + rx0;
        for (int xx = lane; xx < rw; xx += 64) { const int d = (int)po[o + xx] - (int)pr[o + xx]; acc += (unsigned long long)(d * d); }
      }
      for (int m = 32; m >= 1; m >>= 1) { unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32); lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m); acc += ((unsigned long long)hi << 32) | lo; }
      if (lane == 0) __hip_atomic_fetch_add(&st->sse[c], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) { __hip_atomic_fetch_add(&st->est_bits, (unsigned long long)s.est_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (wpp ? wrow0 == 0 : tile == p.tile_begin) { st->ctus = (uint32_t)nctu; st->pad = 0; } }
  }
  return 0;
}

// the workgroup's read-only tables (z-scan map, CABAC tables, scans): wave 0, once per kernel
DEV void init_tables(LDS Tables &t)
{
  const int lane = lane_id();
  for (int r = lane; r < 256; r += 64) {
    const int x = r & 15, y = r >> 4; int z = 0;
    for (int b = 0; b < 4; b++) z |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1));
    t.r2z[r] = (uint8_t)z;
  }
  for (int i = lane; i < 128; i += 64) { t.t_ebits[i] = c_entropy_bits[i]; t.t_next[1][i] = c_next_mps[i]; t.t_next[0][i] = c_next_lps[i]; }
  if (lane < 9) { t.t_ang[lane] = c_ang_table[lane]; t.t_inv_ang[lane] = c_inv_ang_table[lane]; }
  if (lane < 16) t.t_ctx_map4[lane] = c_ctx_ind_map_4x4[lane];
  if (lane < 32) t.t_group_idx[lane] = c_group_idx[lane];
  if (lane < 5) t.t_filter_thr[lane] = c_intra_filter_thr[lane];
  if (lane < 12) { // CG order of every (scan type, block size)
    const int type = lane >> 2, l = lane & 3, wg = 1 << l, ng = wg * wg;
    LDS uint8_t *cg = t.scan_cg_all[type] + (l == 0 ? 0 : (l == 1 ? 1 : (l == 2 ? 5 : 21)));
    int ln = 0, c = 0;
    for (int g = 0; g < ng; g++) { cg[g] = (uint8_t)(ln * wg + c); scan_next(type, wg, wg, ln, c); }
  }
  if (lane < 3) { // order inside a CG, per scan type
    int l2 = 0, c2 = 0;
    for (int q = 0; q < 16; q++) { t.scan_in_cg[lane][q] = (uint8_t)((l2 << 2) | c2); scan_next(lane, 4, 4, l2, c2); }
  }
}

} // namespace

extern "C" __global__ __launch_bounds__(NW * 64)
void RD_SYM(hevcdl_rd_frame_kernel)(hevcdl_rd_params p)
{
  LSmem &s = lds();
  const int lane = lane_id(), wave = wave_id();
  // this wave's global scratch
  {
    GLB unsigned char *scr = (GLB unsigned char *)p.scratch + ((size_t)blockIdx.x * NW + wave) * p.scratch_per_wave;
    s.my_coef = (GLB int16_t *)scr; s.my_rec = (GLB pel_t *)(scr + 4 * 6144 * 2); s.my_ovl = s.my_rec + 5 * 6144; s.my_save = (GLB unsigned long long *)(s.my_ovl + 6144);
    s.my_log = s.my_save + N_SAVE * (SAVE_BYTES / 8);
    s.my_qcost = (GLB double *)(scr + SCR_LAYERS); s.my_qrate = (GLB int32_t *)(scr + SCR_LAYERS + 16384);
    s.my_slots = scr + SCR_LAYERS + SCR_RDOQ;
  }
  // units are dealt round-robin: unit u belongs to workgroup u mod G, wave (u div G) mod NW
  const int n_units = p.n_frames * p.tile_count, G = (int)gridDim.x;
  const int dyn = p.wpp == 1 && p.wpp_masters > 0;          // WaveFrontSynchro: rows are claimed as they become startable (wpp_claim), not dealt
  int first = dyn ? n_units : (int)blockIdx.x + G * wave;
  if (p.migrate) { // the surplus units (beyond `base` per workgroup) start evenly spaced round the ring, not bunched in the first workgroups
    const int base = n_units / G, extra = n_units - base * G, g = (int)blockIdx.x;
    if (wave == base) { const int e = (g * extra + G - 1) / G; first = (e < extra && (e * G) / extra == g) ? base * G + e : n_units; }
    else if (wave > base) first = n_units;
  }
  LDS WgShared &sh = wg_shared();
  if (lane == 0) for (int q = 0; q < NREG; q++) { sh.reg[wave][q].ticket = 0; sh.reg[wave][q].done = 0; sh.reg[wave][q].owner = wave; }
  if (lane == 0) { s.bound_reg = 0; s.bound_child = -1; s.rp_pair = nullptr; s.chroma_jobs = 5; }
  if (wave == 0) { // the workgroup's shared part: read-only tables (z-scan map, CABAC tables, scans), master count
    init_tables(sh.tab);
    if (lane == 0) { int m = 0; for (int w = 0; w < NW; w++) m += ((int)blockIdx.x + G * w) < n_units; if (p.migrate) m = glb_load_lane0(sched_count(p, (int)blockIdx.x));
                     if (dyn) m = (int)blockIdx.x < p.master_groups ? (p.wpp_masters < NW ? p.wpp_masters : NW) : 0;
                     sh.masters_active = m;
                     sh.quit = 0; sh.bell = 0; sh.remote = p.remote; sh.sched = (GLB unsigned char *)p.sched; }
  }
#ifdef HEVCDL_KERNEL_PROF
  s.my_prof = (blockIdx.x == 0 && p.dbgbuf) ? (GLB unsigned long long *)(p.dbgbuf + 2) : nullptr; if (lane == 0) s.prof_task = 0;
  const unsigned long long prof_start_ = __builtin_readcyclecounter();
#endif
  __syncthreads();
  if (p.remote && (int)blockIdx.x >= p.master_groups) { // a workgroup without units: wave 0 takes second passes other workgroups post, the other waves serve its regions
    if (wave == 0 && lane == 0) __hip_atomic_fetch_add(rq_idle((GLB unsigned char *)p.sched), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int seen = -1;
    for (;;) {
      if (wave == 0) {
        if (glb_load(sched_finished(p)) >= n_units) { wsync(); if (lane == 0) __hip_atomic_store(&sh.quit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
        // a taker that found nothing waits ~7 us: with hundreds of idle workgroups somebody still looks every few tens of nanoseconds, and the queue's two words
        // are not hammered while the posters need them
        if (!remote_serve((GLB unsigned char *)p.sched)) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
      } else {
        if (lds_load(&sh.quit)) break;
        const int b = lds_load(&sh.bell);                       // (see ring_bell: the tickets are walked only when tasks were put up since the last walk that found none)
        if (HELPER_BELL && b == seen) { __builtin_amdgcn_s_sleep(HEVCDL_IDLE_SLEEP); continue; }
        if (!helper_step()) { seen = b; __builtin_amdgcn_s_sleep(HELPER_BELL ? HEVCDL_IDLE_SLEEP : 32); }
      }
    }
    return;
  }
  // a wave walks a unit (master) or serves the workgroup's regions (helper); with p.migrate units arrive and leave through the mailboxes
  int unit = first < n_units ? first : -1, i_resume = -1;
  int seen = -1, since_check = 0;          // seen: the doorbell's count when this wave last walked the tickets and found no task (ring_bell)
  bool can_claim = dyn && (int)blockIdx.x < p.master_groups && wave < p.wpp_masters;
  for (;;) {
    if (can_claim && unit < 0) {
      const int c = wpp_claim(p, n_units);
      if (c >= 0) { unit = c; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
      else if (c == -2) { can_claim = false; wg_release(); lds_add(&sh.masters_active, -1); }
    }
    if (unit >= 0) {
      PROF_T0();
#ifdef HEVCDL_MASTER_PRIO
      __builtin_amdgcn_s_setprio(HEVCDL_MASTER_PRIO);          // experiment: the wave that walks a unit wins the issue arbitration of its SIMD against the wave it shares it with
#endif
      const int moved = process_unit(p, unit, i_resume);
#ifdef HEVCDL_MASTER_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      PROF_ADD(0, 31);
      int next = -1;
      if (dyn) { if (p.remote) glb_add(sched_finished(p), 1); unit = -1; i_resume = -1; continue; }      // (the wave stays a claimer: masters_active counts it until every row has an owner)
      if (!moved) {
        if (p.migrate) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); glb_add(sched_count(p, (int)blockIdx.x), -1); glb_add(sched_finished(p), 1); }
        else if (p.remote) glb_add(sched_finished(p), 1);               // (every pass this unit posted has been joined)
        else if (unit + G * NW < n_units) next = unit + G * NW;        // more units than wave slots: the next one of this wave's list
      }
      unit = next; i_resume = -1;
      if (unit < 0) { wg_release(); lds_add(&sh.masters_active, -1); }
      continue;
    }
    if (p.migrate) { // (two round trips to HBM: behind every task, and every fourth look of a wave that has none)
      if (!HELPER_BELL || since_check <= 0) {
        since_check = 4;
        if (glb_load(sched_finished(p)) >= n_units) break;
        GLB Mbox *mb = sched_mbox(p, (int)blockIdx.x);
        if (glb_load(&mb->state) == 1 && glb_cas(&mb->state, 1, 3)) { // a unit handed over by the previous workgroup of the ring
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          unit = uni(mb->unit); i_resume = uni(mb->next_i);
          lds_add(&sh.masters_active, 1);
          continue;
        }
      }
      since_check--;
    } else if (lds_load(&sh.masters_active) <= 0) break;
    if (can_claim) { // nothing can start right now: serve the workgroup's regions if there are tasks, look again in a few microseconds (the ring's words are shared by every idle wave of the chip)
      PROF_T0();
      if (!helper_step()) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); PROF_ADD(0, 23); }
      continue;
    }
    { PROF_T0();
      const int b = lds_load(&sh.bell);
      if (HELPER_BELL && b == seen) { __builtin_amdgcn_s_sleep(HEVCDL_IDLE_SLEEP); PROF_ADD(0, 23); }
      else if (helper_step()) since_check = 0;
      else { seen = b; __builtin_amdgcn_s_sleep(HELPER_BELL ? HEVCDL_IDLE_SLEEP : 32); PROF_ADD(0, 23); } }
  }
#ifdef HEVCDL_KERNEL_PROF
  // in-kernel timers of workgroup 0, summed over its waves (masters and helpers): the host decodes the accumulators (tools/phase_profile.py)
  wsync();
  if (blockIdx.x == 0 && p.dbgbuf && lane == 0) { PROF_ACC_(14, __builtin_readcyclecounter() - prof_start_); p.dbgbuf[0] = 64; }
#endif
}

extern "C" size_t RD_SYM(hevcdl_rd_smem_bytes)(void) { return (size_t)NW * sizeof(RdSmem) + sizeof(WgShared); }
extern "C" size_t RD_SYM(hevcdl_rd_scratch_bytes)(void) { return SCR_WAVE; }        // per wave
extern "C" int RD_SYM(hevcdl_rd_waves_per_group)(void) { return NW; }

#if defined(HEVCDL_MICRO) && HEVCDL_BD == 8 && !defined(HEVCDL_RD_WIDE) && !defined(HEVCDL_RD_TOOLS)
// -DHEVCDL_MICRO (tools/micro_rd.py; never in the product library): the leaf routines of a TU coding timed on their own, on synthetic residual blocks.
// Every wave of every workgroup runs `reps` codings (what: 0 RDOQ, 1 the bit counter, 2 forward transform + RDOQ + bit counter + dequant + inverse);
// out[wave] = { cycles inside the timed routine, checksum }.
extern "C" __global__ __launch_bounds__(NW * 64)
void hevcdl_micro_kernel(hevcdl_rd_params p, const int16_t *resi_, int n_blocks, int n, int comp, int mode, int reps, int what, unsigned long long *out_)
{
  LSmem &s = lds();
  const int lane = lane_id(), wave = wave_id();
  GLB const int16_t *resi = (GLB const int16_t *)resi_; GLB unsigned long long *out = (GLB unsigned long long *)out_;
  {
    GLB unsigned char *scr = (GLB unsigned char *)p.scratch + ((size_t)blockIdx.x * NW + wave) * p.scratch_per_wave;
    s.my_coef = (GLB int16_t *)scr; s.my_rec = (GLB pel_t *)(scr + 4 * 6144 * 2); s.my_ovl = s.my_rec + 5 * 6144; s.my_save = (GLB unsigned long long *)(s.my_ovl + 6144);
    s.my_log = s.my_save + N_SAVE * (SAVE_BYTES / 8);
    s.my_qcost = (GLB double *)(scr + SCR_LAYERS); s.my_qrate = (GLB int32_t *)(scr + SCR_LAYERS + 16384);
    s.my_slots = scr + SCR_LAYERS + SCR_RDOQ;
  }
  if (wave == 0) init_tables(wg_shared().tab);
  __syncthreads();
  const int active = what >> 8; what &= 255;                        // waves per workgroup that run (0: all)
  if (active && wave >= active) { if (lane == 0) { out[2 * (blockIdx.x * NW + wave)] = 0; out[2 * (blockIdx.x * NW + wave) + 1] = 0; } return; }
  LDS K &k = s.k;
  if (lane == 0) { s.bound_reg = 0; s.bound_child = -1; }
  k.q_cost = s.my_qcost; k.q_rate = s.my_qrate;
  k.lambda = p.k.lambda; k.sqrt_lambda = p.k.sqrt_lambda; k.cweight = p.k.chroma_weight; k.lambda_c = p.k.lambda_chroma;
  for (int a = 0; a < 2; a++) { for (int b = 0; b < 4; b++) k.err_scale[a][b] = p.k.err_scale[a][b]; k.sbh[a] = p.k.sbh_rd_factor[a]; }
  k.qp = p.k.qp; k.qp_c = p.k.qp_chroma; k.dbg = 0; k.dbgbuf = nullptr; k.tools = p.k.tools;
  LCabac *c0 = &s.truec;
  for (int i = lane; i < NUM_CTX; i += 64) {
    const int v = c_ctx_init[i], slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
    int st = ((slope * p.k.qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
    const int mps = st >= 64;
    c0->ctx[i] = (uint8_t)(((mps ? st - 64 : 63 - st) << 1) + mps);
  }
  if (lane == 0) { c0->ctx[159] = 0; c0->frac = 0; }
  wsync();
  const int log2n = ilog2(n);
  unsigned long long total = 0, sum = 0;
#ifdef HEVCDL_MICRO_T
  if (lane < 12) s.mt_acc[lane] = 0;
  wsync();
#endif
  for (int rep = 0; rep < reps; rep++) {
    GLB const int16_t *src = resi + (size_t)((int)(blockIdx.x * NW + wave + rep) % n_blocks) * 1024;
    wsync();
    for (int i = lane; i < n * n; i += 64) s.resi[(i >> log2n) * RS(n) + (i & (n - 1))] = src[i];
    cabac_copy(k, &s.go, c0);
    wsync();
    unsigned long long t0 = __builtin_readcyclecounter();
    fwd_transform(k, n, !comp && n == 4);
    if (what != 2) t0 = __builtin_readcyclecounter();
#if HEVCDL_RDOQ_FIX
    const uint32_t as = n == 4 ? rdoq_wave<4>(k, &s.go, comp, n, mode, 1) : (n == 8 && HEVCDL_RDOQ_FIX >= 8 ? rdoq_wave<HEVCDL_RDOQ_FIX >= 8 ? 8 : 0>(k, &s.go, comp, n, mode, 1) : rdoq_wave<0>(k, &s.go, comp, n, mode, 1));
#else
    const uint32_t as = rdoq_wave(k, &s.go, comp, n, mode, 1);
#endif
    wsync();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (what != 0) {
      if (what == 1) t0 = __builtin_readcyclecounter();
      if (as) code_coeff_wave(k, &s.go, comp, n, mode, 0);
      if (what == 2 && as) { dequant(k, comp, n); inv_transform(k, n, !comp && n == 4); }
      wsync();
      t1 = __builtin_readcyclecounter();
    }
    total += t1 - t0; sum += as + (s.go.frac >> 15);
  }
  if (lane == 0) { out[2 * (blockIdx.x * NW + wave)] = total; out[2 * (blockIdx.x * NW + wave) + 1] = sum; }
#ifdef HEVCDL_MICRO_T
  wsync();
  if (lane < 11 && blockIdx.x == 0 && wave == 0) out[2 * gridDim.x * NW + lane] = s.mt_acc[lane];      // behind the per-wave pairs: workgroup 0, wave 0
#endif
}

// host side: consts = { lambda, sqrt_lambda, chroma_weight, lambda_chroma, err_scale[2][4] }, sbh[2]; resi: n_blocks x 1024 int16; out: groups * NW * 2 values
extern "C" int hevcdl_micro_run(const double *consts, const long long *sbh, int qp, int qp_c, const int16_t *resi, int n_blocks, int n, int comp, int mode, int reps, int what, int groups,
                                unsigned long long *out)
{
  hevcdl_rd_params p = {};
  p.k.lambda = consts[0]; p.k.sqrt_lambda = consts[1]; p.k.chroma_weight = consts[2]; p.k.lambda_chroma = consts[3];
  for (int a = 0; a < 2; a++) for (int b = 0; b < 4; b++) p.k.err_scale[a][b] = consts[4 + 4 * a + b];
  p.k.sbh_rd_factor[0] = sbh[0]; p.k.sbh_rd_factor[1] = sbh[1]; p.k.qp = qp; p.k.qp_chroma = qp_c; p.k.tools = (int)HEVCDL_TOOLS_REFERENCE;
  p.scratch_per_wave = SCR_WAVE;
  int16_t *d_resi = nullptr; unsigned long long *d_out = nullptr; unsigned char *d_scr = nullptr;
  const size_t smem = (size_t)NW * sizeof(RdSmem) + sizeof(WgShared);
  if (hipMalloc(&d_resi, (size_t)n_blocks * 2048) != hipSuccess || hipMalloc(&d_out, (size_t)groups * NW * 16 + 128) != hipSuccess || hipMalloc(&d_scr, (size_t)groups * NW * SCR_WAVE) != hipSuccess) return -1;
  p.scratch = d_scr;
  hipMemcpy(d_resi, resi, (size_t)n_blocks * 2048, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)hevcdl_micro_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(hevcdl_micro_kernel, dim3(groups), dim3(NW * 64), smem, 0, p, (const int16_t *)d_resi, n_blocks, n, comp, mode, reps, what, d_out);
  const hipError_t e = hipDeviceSynchronize();
  hipMemcpy(out, d_out, (size_t)groups * NW * 16 + 128, hipMemcpyDeviceToHost);      // + 16 values: the phase timers of -DHEVCDL_MICRO_T
  hipFree(d_resi); hipFree(d_out); hipFree(d_scr);
  return e == hipSuccess ? 0 : -2;
}
#endif
