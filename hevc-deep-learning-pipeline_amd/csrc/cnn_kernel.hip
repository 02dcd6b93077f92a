// cnn_kernel.hip -- on-device CNN depth predictor for gfx950 (MI355X), one 256-thread workgroup per CTU.
//
// Replaces the reference's PyTorch sidecar (/root/reference/use_model.py):
//   ConvNet2.forward                      use_model.py:16-58   (BatchNorm in TRAINING mode, :61-63)
//   CTU / quadrant tiling, zero padding   use_model.py:80-95
//   4x argmax + label fix-ups             use_model.py:101-119
//   label file IPC                        use_model.py:121-125 <-> TEncCu.cpp:244-253   (labels stay in HBM)
// plus the YUV->RGB input transform and the boundary clamp defined by this project.
//
// The four convolutions are im2col GEMMs on the matrix cores.  M = output positions (16 per tile, ordered so that the 4 accumulator
// registers of a lane are the members of one max-pool window), N = 16 output channels per tile, K = taps x input channels.
//   All four run on v_mfma_f32_16x16x32_f16 with SPLIT operands: a value v is the pair hi = f16(v), lo = f16(v - hi) (22 significant bits for |v| >= 2^-3, an
//   absolute floor of 2^-25 below: see split_f16), the weights are split the same way on the host, and a product is hi*hi + hi*lo + lo*hi accumulated in f32 (lo*lo,
//   < 2^-21 relative, is dropped) -- f32-like accuracy (the logits stay within 1e-3 of the f32 reference, tests/test_cnn_gpu.py) at 16 / 3 times the f32 MFMA rate.
//   No instruction stands between an LDS read and the MFMA that consumes it (round 4):
//     conv2 / conv3 (76 % of the FLOPs): the maps hold PAIRS of channels per 32-bit word, hi halves and lo halves in separate arrays, position-major with the four
//       words of an operand next to each other: one ds_read_b128 is the hi operand, one the lo operand; three MFMAs per 32 channels (split_pair, pack_conv3);
//     the 5x5 layers: the input tile holds one word (hi | lo) per sample, four RAW words are the A operand, and the two products come from two B operands
//       (conv5_mfma, pack_conv5): 75 taps as 5 k-steps of 16, two MFMAs a step.
//   Every operand read is bank-conflict-free by construction (the strides asserted below, hevcdl_conv5_slot_tap, the 2 x 8 M-tiles of conv2 / conv3).
// conv+BN(train)+ReLU+pool are fused: BN statistics are per sample = per workgroup, so no global reduction exists;
// x -> relu(x*alpha+beta) is monotone, so the pool runs before the affine map (max or min by the sign of gamma).
// The conv64 branch is evaluated once and shared by the 4 quadrants (identical input, identical statistics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hevcdl_dev.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#ifndef HEVCDL_CNN_SKEW
#define HEVCDL_CNN_SKEW 40            // start offset of the second workgroup of a CU, in units of 8128 cycles (see the kernel)
#endif
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))

// conv2's / conv3's input maps are POSITION-major: the four words of an MFMA operand -- channel pairs 4 j + g, j = 0..3, of lane group g; hi halves and lo halves
// in separate arrays -- lie next to each other, one ds_read_b128 per operand.  act12 = [g][hi | lo][18 x 18 positions][4 words], a2 = [k-step][g][hi | lo][10 x 10][4].
// A ds_read_b128 is served 16 lanes at a time, 16 positions of one or two lane groups: with row pitches of 18 / 10 positions the 16 positions of an M-tile (2 columns x
// 8 rows) are distinct mod 16 = distinct quarters of the 64 banks, and an array size == 0 (mod 32 words) keeps the two lane groups of a service group on the same footing.
constexpr int A_ROW = 18, AP = 18 * 18 * 4 + 16;        // words per (g, hi | lo) array of act12
constexpr int A2_ROW = 10, A2P = 10 * 10 * 4 + 16;      // ... of a2
constexpr int T64_ROW = 72, T64_CH = 36 * 72 + 16;   // input tile of HALF the CTU (32 rows + halo 2; conv64 runs in two halves): row pitch == 8, channel stride == 16 (mod 32 banks)
constexpr int T32_ROW = 40, T32_CH = 36 * 40 + 16;   // input tile of one quadrant, halo 2: row pitch == 8, channel stride == 16 (mod 32 banks); hevcdl_conv5_slot_tap

static_assert(AP % 32 == 0 && A2P % 32 == 0 && T64_CH % 32 == 16 && T32_CH % 32 == 16 && T64_ROW % 32 == 8 && T32_ROW % 32 == 8 && A_ROW == 18 && A2_ROW == 10,
              "the operand reads are bank-conflict-free only at these strides (conv2 / conv3: 2 x 8 M-tiles; 5x5 layers: 8 x 2 M-tiles, hevcdl_conv5_slot_tap)");
struct CnnSmem {
  float lut[256];                          // u8 -> u8/255 (ToTensor, use_model.py:94-95), stored split (hi | lo halves): the operand form of the convolutions
  int koff64[80], koff32[80];              // im2col offset, inside the input tiles, of the tap in K slot k (hevcdl_conv5_slot_tap; a padding slot reads a tap that keeps the gather conflict-free, zero weight)
  float act12[8 * AP];                     // conv1 output (channels 0..15 = pairs 0..7) ++ conv64 output (16..31 = pairs 8..15): cat of use_model.py:50; pair m: lane group m & 3, word m >> 2
  union {
    float t64[3 * T64_CH];                 // conv64 input tile of one half of the CTU (only live before the quadrant loop)
    struct {
      union { float t32[3 * T32_CH]; float a2[16 * A2P]; };   // conv1 input tile | conv2 output (pair m of 32: k-step m >> 4, lane group m & 3, word (m & 15) >> 2)
    } q;
  };
  double red[2][4][16];                    // per-wave partial sums / sums of squares (5x5 layers)
  double red2[2][4][64];                   // ... of conv2: every wave holds 4 of the 16 M-tiles of all 64 channels
  float alpha[16], beta[16];               // BN folded to y = x*alpha + beta (conv1 / conv64)
  int bn_eval;                             // HEVCDL_BN_EVAL: the packed gamma / beta slots already hold the folded running statistics
};

__device__ __forceinline__ double shfl_xor_d(double v, int m)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}

// BN in training mode folded to an affine map (nn.BatchNorm2d defaults: biased variance, eps 1e-5)
// The sums come from accumulators that hold 1 / inv_in times the layer's output (operand scales, hevcdl_dev.h HEVCDL_ACT_SCALE: a power of two, so the sums are
// unscaled exactly); the map it returns takes such an accumulator to the activation times HEVCDL_ACT_SCALE, the operand scale of the next layer.
__device__ __forceinline__ void bn_fold(double s, double ss, double n, float gamma, float beta, float inv_in, float &alpha, float &betap)
{
  const double rn = (1.0 / n) * (double)inv_in;    // (n is a power of two at every call: exact)
  double mean = s * rn;
  double var = ss * (rn * (double)inv_in) - mean * mean;
  if (var < 0) var = 0;
  // 1 / sqrt(x): the f32 reciprocal square root (1 ulp) refined by one Newton step in f64 -- 1e-14 relative, against 6e-8 of the float the result is rounded to;
  // the library's f64 sqrt and division are ~70 instructions, this is 8
  const double x = var + 1e-5;
  const double r0 = (double)__builtin_amdgcn_rsqf((float)x);
  const double inv = r0 * (1.5 - (0.5 * x) * (r0 * r0));
  alpha = (float)(inv * (double)gamma * ((double)inv_in * (double)HEVCDL_ACT_SCALE));
  betap = (float)(((double)beta - mean * inv * (double)gamma) * (double)HEVCDL_ACT_SCALE);
}

// v -> one word: low half f16(v) (rounded toward zero), high half f16(v - low half).
// Accuracy of the pair, stated exactly: hi carries 11 significant bits, lo the next 11 WHILE v - hi is a normal f16, i.e. for |v| >= 2^-3 (22 bits, 4.8e-7
// relative); below that lo is an f16 subnormal and the pair's error is bounded absolutely instead, by 2^-25 (lo's last place), and a |v| < 2^-25 remainder is
// lost.  The operands here are activations in [0, ~8] and weights of magnitude 1e-3 .. 1: the absolute floor (3e-8 per operand) is what the 4.2e-5 logit error
// measured against the fp32 graph consists of (tests/test_cnn_gpu.py bounds it at 1e-3; bench.py cnn_label_check counts the labels that differ: 0 of 32 640).
// This relies on the MFMA NOT flushing f16 subnormal inputs, which holds on gfx950 (CDNA3 / 4) and not on gfx90a: hence the guard.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "cnn_kernel.hip: the split-f16 operand form is validated for gfx950 only (f16 subnormal MFMA inputs must not be flushed)"
#endif
__device__ __forceinline__ float split_f16(float v)
{
  const auto h = __builtin_amdgcn_cvt_pkrtz(v, 0.f);
  const float r = v - (float)h[0];
  return __builtin_bit_cast(float, __builtin_amdgcn_cvt_pkrtz(v, r));
}
// two activations -> the word of their hi halves (a in the low half) and the word of their lo halves: the operand form of conv2 / conv3, whose maps hold PAIRS of
// channels per word in separate hi and lo planes, so that four words of a plane are an MFMA operand as they are (no repacking between the LDS read and the MFMA)
__device__ __forceinline__ void split_pair(float a, float b, float &hi, float &lo)
{
  const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
  hi = __builtin_bit_cast(float, h);
  lo = __builtin_bit_cast(float, __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]));
}
// one product on split operands: acc += a * b with a = ah + al, b = bh + bl (al * bl dropped)
__device__ __forceinline__ v4f mfma3(const h8 &ah, const h8 &al, const h8 &bh, const h8 &bl, v4f c)
{
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
}

// 5x5 conv (3 -> 16, zero pad 2 relative to the region) + BN(train) + ReLU + POOLxPOOL max pool -> 16 maps of 16x16.
// tile: region of split words with a 2-pixel zero halo (row pitch ROW, channel stride CH); koff: im2col offsets of the 80 K slots for that pitch.
// Operand form (round 4): the A operand of a k-step is FOUR RAW WORDS of the tile -- the (hi, lo) halves of four taps, exactly as they lie in LDS, no repacking -- and
// the step is two MFMAs: B1 holds the weight's hi half against both halves of the word (a * bh), B2 its lo half against the word's hi half only (ah * bl): the same three
// partial products as conv2 / conv3 form with three MFMAs and eight v_perm per 32 taps.  75 taps -> 5 k-steps of 16 -> 10 MFMAs per tile (9 before), 20 LDS words per lane
// and tile (24 before), no VALU between the reads and the MFMAs of k-steps 0..3: their four words are kx = 0..3 of one tile row (hevcdl_conv5_slot_tap), read through ONE
// per-lane pointer by two ds_read2_b32 into four consecutive registers; step 4 (the kx = 4 column) has a pointer per word.  Everything else in an address is an
// immediate offset (the 4 tiles of an iteration and the words of a row are at fixed distances).
// An M-tile is an 8 x 2 block of positions for BOTH pool sizes (its 16 words cover 16 banks at a row pitch == 8 (mod 32); hevcdl_conv5_slot_tap pairs the taps of
// the two lane groups of a read 16 banks apart).  POOL == 2 (conv1, 32x32 quadrant): row i of the tile = member i & 3 of the 2x2 window i >> 2, lane group g = lane >> 4 ends up with
// window g in its 4 registers.  POOL == 4 (conv64): row i = column i & 3, row (i >> 2) & 1 of the upper or lower half of the 4x4 window i >> 3; a window is finished from the
// two vertically adjacent tiles of an iteration (in registers) and one exchange between lane groups g and g ^ 1.
// The pooled extreme (max for gamma >= 0, min otherwise) is written straight into the halo'd destination map and rescaled in place once the statistics of the whole map are known.
// phase 0: the whole map in one call (conv1 on a quadrant).  phase 1 / 2: upper / lower half of the CTU for conv64 (tile rows are then
// relative to the half); the statistics of the halves meet in sm.red and the map is normalised after the second.
template <int POOL>
__device__ __noinline__ void conv5_mfma(CnnSmem LDS &sm, const float LDS *tile, const int LDS *koff, const float GLB *w, float inv_in, float LDS *map, int pb, int tid, int phase)
{
  constexpr int ROW = (POOL == 4) ? T64_ROW : T32_ROW;
  constexpr int ITERS = (POOL == 4) ? 8 : 4;        // per wave and call: 8 rows of the region, 4 tiles an iteration
  const int lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const u4 GLB *wq = (const u4 GLB *)w + lane;
  h8 b1[5], b2[5];
#pragma unroll
  for (int ks = 0; ks < 5; ks++) { b1[ks] = __builtin_bit_cast(h8, wq[(2 * ks) * 64]); b2[ks] = __builtin_bit_cast(h8, wq[(2 * ks + 1) * 64]); }
  const float bias = w[HEVCDL_W_C5 + i], gamma = w[HEVCDL_W_C5 + 16 + i];
  const int px = (POOL == 4) ? 4 * (i >> 3) + (i & 3) : 2 * (i >> 2) + (i & 1), py = (POOL == 4) ? (i >> 2) & 1 : (i >> 1) & 1;
  const unsigned LDS *kp[8];                         // this lane's pointers into the wave's first tile: [s] word 0 of k-step s < 4 (words 1..3 follow it), [4 + j] word j of step 4
#pragma unroll
  for (int q = 0; q < 8; q++) kp[q] = (const unsigned LDS *)tile + (8 * wave + py) * ROW + px + koff[q < 4 ? 16 * q + 4 * g : 64 + 4 * g + (q - 4)];
  auto word = [&](int ks, int j, int off) { return ks < 4 ? kp[ks][off + j] : kp[4 + j][off]; };
  // the pooled extremes are staged where the operand words of their channel will be: channel i = half i & 1 of pair pb + (i >> 1) -> the hi (even) / lo (odd) array
  float LDS *const out = map + (((pb + (i >> 1)) & 3) * 2 + (i & 1)) * AP + ((pb + (i >> 1)) >> 2);
  double s = 0, ss = 0;
  // tile u of an iteration: POOL 2: columns 8u of one tile row;  POOL 4: tile row u >> 1, columns 8 (u & 1) of a 16 x 4 block (four windows)
  auto toff = [](int u) { return (POOL == 4) ? (u >> 1) * 2 * ROW + 8 * (u & 1) : 8 * u; };
  // A operands: a ring of six register sets, the reads of (k-step, tile) pair p + 5 are issued before the two MFMAs of pair p (32 cycles a pair against an LDS
  // round trip of 64 and more); the first five pairs of the NEXT iteration are requested before this iteration's statistics
  auto rd = [&](int pr) { u4 r; r[0] = word(pr >> 2, 0, toff(pr & 3)); r[1] = word(pr >> 2, 1, toff(pr & 3)); r[2] = word(pr >> 2, 2, toff(pr & 3)); r[3] = word(pr >> 2, 3, toff(pr & 3)); return r; };
  u4 wr[6];
#pragma unroll
  for (int pr = 0; pr < 5; pr++) wr[pr] = rd(pr);
#pragma unroll 1
  for (int it = 0; it < ITERS; it++) {
    v4f acc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) acc[u] = (v4f){ bias, bias, bias, bias };
#pragma unroll
    for (int p = 0; p < 20; p++) {
      if (p + 5 < 20) wr[(p + 5) % 6] = rd(p + 5);
      const h8 a = __builtin_bit_cast(h8, wr[p % 6]);
      acc[p & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2[p >> 2], acc[p & 3], 0, 0, 0);
      acc[p & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1[p >> 2], acc[p & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                      // (left alone the reads are moved next to their use)
    }
    { // next iteration: POOL 2: two rows down;  POOL 4: 16 columns to the right, after four of them four rows down
      const int step = (POOL == 4) ? (((it & 3) == 3) ? 4 * ROW - 48 : 16) : 2 * ROW;
#pragma unroll
      for (int q = 0; q < 8; q++) kp[q] += step;
#pragma unroll
      for (int pr = 0; pr < 5; pr++) wr[pr] = rd(pr);       // (behind the last iteration: words below the wave's rows, never used)
      __builtin_amdgcn_sched_barrier(0);
    }
    float e[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const v4f a = acc[u];
      s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
      ss = __builtin_fma((double)a.x, (double)a.x, ss); ss = __builtin_fma((double)a.y, (double)a.y, ss); ss = __builtin_fma((double)a.z, (double)a.z, ss); ss = __builtin_fma((double)a.w, (double)a.w, ss);
      e[u] = (gamma >= 0.f) ? fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)) : fminf(fminf(a.x, a.y), fminf(a.z, a.w));
    }
    if (POOL == 4) {
      const int prow = 2 * wave + (it >> 2) + ((phase == 2) ? 8 : 0);      // pooled row of the iteration's windows
#pragma unroll
      for (int tc = 0; tc < 2; tc++) {
        float v = (gamma >= 0.f) ? fmaxf(e[tc], e[2 + tc]) : fminf(e[tc], e[2 + tc]);
        const float o = __shfl_xor(v, 16); v = (gamma >= 0.f) ? fmaxf(v, o) : fminf(v, o);
        if (!(g & 1)) out[((prow + 1) * A_ROW + 4 * (it & 3) + 2 * tc + (g >> 1) + 1) * 4] = v;
      }
    } else {
      const int prow = 4 * wave + it;
#pragma unroll
      for (int u = 0; u < 4; u++) out[((prow + 1) * A_ROW + 4 * u + g + 1) * 4] = e[u];
    }
  }
  // per-channel statistics over the whole map: lane groups, then waves
  s += shfl_xor_d(s, 16); s += shfl_xor_d(s, 32); ss += shfl_xor_d(ss, 16); ss += shfl_xor_d(ss, 32);
  if (lane < 16) { if (phase == 2) { sm.red[0][wave][lane] += s; sm.red[1][wave][lane] += ss; } else { sm.red[0][wave][lane] = s; sm.red[1][wave][lane] = ss; } }
  __syncthreads();
  if (phase == 1) return;                 // the lower half follows (the barrier above also frees the tile for it)
  if (tid < 16) {
    double a = 0, b = 0;
    for (int k = 0; k < 4; k++) { a += sm.red[0][k][tid]; b += sm.red[1][k][tid]; }
    constexpr int SIZE = 16 * POOL;
    float al, be;
    if (sm.bn_eval) { al = w[HEVCDL_W_C5 + 16 + tid]; be = w[HEVCDL_W_C5 + 32 + tid]; }      // eval mode: folded on the host (fold_bn_eval)
    else bn_fold(a, b, (double)(SIZE * SIZE), w[HEVCDL_W_C5 + 16 + tid], w[HEVCDL_W_C5 + 32 + tid], inv_in, al, be);
    sm.alpha[tid] = al; sm.beta[tid] = be;
  }
  __syncthreads();
  { // in-place affine + ReLU of the 16 pooled maps: thread = pixel.  The slots then hold conv2's operand form: the hi halves of channels 2m | 2m + 1 in the hi array's
    // word of pair pb + m, their lo halves in the lo array's
    const int py = tid >> 4, px = tid & 15;
    float LDS *d = map + ((py + 1) * A_ROW + px + 1) * 4 + (pb >> 2);
#pragma unroll
    for (int m = 0; m < 8; m++) {                                   // pair pb + m: lane group m & 3, word (pb + m) >> 2; its two channels staged in the hi / lo slots
      float LDS *q = d + ((m & 3) * 2) * AP + (m >> 2);
      const float y0 = fmaxf(q[0] * sm.alpha[2 * m] + sm.beta[2 * m], 0.f), y1 = fmaxf(q[AP] * sm.alpha[2 * m + 1] + sm.beta[2 * m + 1], 0.f);
      float hi, lo; split_pair(y0, y1, hi, lo); q[0] = hi; q[AP] = lo;
    }
  }
  __syncthreads();
}

} // namespace

// 75.9 KB of LDS and at most 256 registers per lane: two workgroups (8 waves) share a CU, so the MFMA phases of one overlap the
// tile fills, statistics and weight fetches of the other
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void hevcdl_cnn_ctu_kernel(hevcdl_cnn_params p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  CnnSmem LDS &sm = *(CnnSmem LDS *)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gctu = p.ctu_base + blockIdx.x;          // global CTU index over all frames (a launch covers a chunk of CTUs)
  const int frame = gctu / p.ctus_per_frame, addr = gctu - frame * p.ctus_per_frame;
  const int x0 = (addr % p.ctus_x) * 64, y0 = (addr / p.ctus_x) * 64;
  const float GLB *W = (const float GLB *)p.weights;
  if (tid == 0) sm.bn_eval = p.bn_eval;      // (read behind the first barrier of the first convolution)
  // Two workgroups share a CU, and every CTU takes the same time: started together they would stay in lockstep -- both in their MFMA phases,
  // then both in their fill / statistics phases.  The workgroup that landed in the second wave slot of its SIMDs (HW_ID.wave_id, measured
  // with tools/hwid_probe.hip: blocks 0..255 take slot 0, blocks 256..511 slot 1 of the same CUs) starts about half a CTU time late; the
  // offset then carries through the whole launch because slots are refilled as they drain.  (-17 % kernel time.)
  if ((int)blockIdx.x < 2 * p.n_cus && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1))      // HW_REG_HW_ID[3:0] = wave slot
    for (int it = 0; it < HEVCDL_CNN_SKEW; it++) __builtin_amdgcn_s_sleep(127);                         // x 8128 cycles
  float GLB *a3_out = (float GLB *)p.a3 + (size_t)blockIdx.x * (4 * 2048);      // this CTU's 4 rows of the fully connected head's input (fc_kernel.hip)
#ifdef HEVCDL_CNN_PROF
  unsigned long long pt_[10] = {0,0,0,0,0,0,0,0,0,0}; unsigned long long pl_ = __builtin_readcyclecounter();
#define CNN_MARK(i) do { __syncthreads(); const unsigned long long n_ = __builtin_readcyclecounter(); pt_[i] += n_ - pl_; pl_ = n_; } while (0)
#else
#define CNN_MARK(i) do { } while (0)
#endif

  // ---- stage 0: CTU input -> LDS (coalesced rows of the planes), LUT, im2col tables, zeroed halo'd maps --------
  sm.lut[tid] = split_f16((float)tid / 255.0f * HEVCDL_ACT_SCALE);       // (a power of two: exact)
  if (tid < 80) {
    const int v = hevcdl_conv5_slot_tap(tid), k = v >= 0 ? v : -v - 1, c = k / 25, r = k - c * 25, ky = r / 5, kx = r - ky * 5;
    sm.koff64[tid] = c * T64_CH + ky * T64_ROW + kx; sm.koff32[tid] = c * T32_CH + ky * T32_ROW + kx;
  }
  // The CTU's RGB samples are read from HBM where a tile is filled (twice per sample: conv64's half tile and conv1's quadrant tile);
  // YUV input is converted on the fly (BT.601 limited range, nearest chroma), samples past the picture edge are 0.
  const size_t ysz_ = (size_t)p.width * p.height, fsz_ = ysz_ + (ysz_ >> 1);
  const uint8_t GLB *srcRGB = (const uint8_t GLB *)p.input + (size_t)gctu * (64 * 64 * 3);
  const uint8_t GLB *Yp = (const uint8_t GLB *)p.input + (size_t)frame * fsz_, *Up = Yp + ysz_, *Vp = Up + (ysz_ >> 2);
  auto pixel = [&](int x, int y, int &r, int &g, int &b) {
    if (p.input_mode == HEVCDL_DEV_INPUT_RGB_CTU) { const uint8_t GLB *q = srcRGB + (y * 64 + x) * 3; r = q[0]; g = q[1]; b = q[2]; return; }
    const int gx = x0 + x, gy = y0 + y, cw = p.width >> 1;
    r = g = b = 0;
    if (gx < p.width && gy < p.height) {
      const int yv = Yp[(size_t)gy * p.width + gx];
      if (p.input_mode == HEVCDL_DEV_INPUT_LUMA) { r = g = b = yv; }
      else {
        const int d = (int)Up[(size_t)(gy >> 1) * cw + (gx >> 1)] - 128, e = (int)Vp[(size_t)(gy >> 1) * cw + (gx >> 1)] - 128, c = yv - 16;
        r = (298 * c + 409 * e + 128) >> 8; g = (298 * c - 100 * d - 208 * e + 128) >> 8; b = (298 * c + 516 * d + 128) >> 8;
        r = r < 0 ? 0 : (r > 255 ? 255 : r); g = g < 0 ? 0 : (g > 255 ? 255 : g); b = b < 0 ? 0 : (b > 255 ? 255 : b);
      }
    }
  };
  // four horizontally adjacent samples (x a multiple of 4) as split words, one row of loads for the strip: a dword of luma + two bytes of each chroma plane
  // (or three dwords of packed RGB) instead of twelve byte loads; strips that straddle the picture's right edge take the per-sample path
  typedef float f2 __attribute__((ext_vector_type(2)));
  auto strip = [&](int x, int y, float (&v0)[4], float (&v1)[4], float (&v2)[4]) {
    int rr[4], gg[4], bb[4];
    if (p.input_mode == HEVCDL_DEV_INPUT_RGB_CTU) {
      const unsigned GLB *q = (const unsigned GLB *)(srcRGB + (y * 64 + x) * 3);
      const unsigned a = q[0], b = q[1], c = q[2];                     // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
      rr[0] = a & 255; gg[0] = (a >> 8) & 255; bb[0] = (a >> 16) & 255; rr[1] = a >> 24; gg[1] = b & 255; bb[1] = (b >> 8) & 255;
      rr[2] = (b >> 16) & 255; gg[2] = b >> 24; bb[2] = c & 255; rr[3] = (c >> 8) & 255; gg[3] = (c >> 16) & 255; bb[3] = c >> 24;
    } else {
      const int gx = x0 + x, gy = y0 + y;
      if (gy < p.height && gx + 3 < p.width) {
        const uint8_t GLB *yq = Yp + (size_t)gy * p.width + gx;
        typedef unsigned u32u __attribute__((aligned(1))); typedef unsigned short u16u __attribute__((aligned(1)));      // (a row need not start on a dword)
        const unsigned yw = *(const u32u GLB *)yq;
        if (p.input_mode == HEVCDL_DEV_INPUT_LUMA) {
#pragma unroll
          for (int k = 0; k < 4; k++) rr[k] = gg[k] = bb[k] = (yw >> (8 * k)) & 255;
        } else {
          const size_t co = (size_t)(gy >> 1) * (p.width >> 1) + (gx >> 1);
          const unsigned uw = *(const u16u GLB *)(Up + co), vw = *(const u16u GLB *)(Vp + co);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int c = (int)((yw >> (8 * k)) & 255) - 16, d = (int)((uw >> (8 * (k >> 1))) & 255) - 128, e = (int)((vw >> (8 * (k >> 1))) & 255) - 128;
            int r = (298 * c + 409 * e + 128) >> 8, g = (298 * c - 100 * d - 208 * e + 128) >> 8, b = (298 * c + 516 * d + 128) >> 8;
            rr[k] = r < 0 ? 0 : (r > 255 ? 255 : r); gg[k] = g < 0 ? 0 : (g > 255 ? 255 : g); bb[k] = b < 0 ? 0 : (b > 255 ? 255 : b);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) pixel(x + k, y, rr[k], gg[k], bb[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { v0[k] = sm.lut[rr[k]]; v1[k] = sm.lut[gg[k]]; v2[k] = sm.lut[bb[k]]; }
  };
  // zero halos only: the interiors of the maps / tiles are rewritten before they are read
  for (int i = tid; i < 3 * 36 * 8; i += 256) {                      // the conv64 tile's columns outside the CTU (2 left, 6 right of the 72): both halves keep them
    const int c = i / (36 * 8), r = (i >> 3) % 36, k = i & 7;
    sm.t64[c * T64_CH + r * T64_ROW + (k < 2 ? k : 64 + k)] = 0.f;
  }
  for (int i = tid; i < 8 * 68; i += 256) {                        // the 68 border positions of the 18x18 maps, four words each, in the 8 arrays of act12
    const int c = i / 68, r = i - c * 68;
    const int o = r < 18 ? r : (r < 36 ? 17 * 18 + (r - 18) : (r < 52 ? (r - 35) * 18 : (r - 51) * 18 + 17));
    *(u4 LDS *)(sm.act12 + c * AP + o * 4) = (u4){ 0u, 0u, 0u, 0u };
  }
  __syncthreads();
  CNN_MARK(0);
  // ---- conv64 branch, once per CTU (use_model.py:38-43), in two halves of 32 rows: the tile holds input rows 32h - 2 .. 32h + 33 ----
  for (int h = 0; h < 2; h++) {
    for (int i = tid; i < 36 * 16; i += 256) {                       // strips of 4 samples: 16 a row
      const int r = i >> 4, x = (i & 15) * 4, y = 32 * h - 2 + r;
      float v0[4] = { 0.f, 0.f, 0.f, 0.f }, v1[4] = { 0.f, 0.f, 0.f, 0.f }, v2[4] = { 0.f, 0.f, 0.f, 0.f };
      if (y >= 0 && y < 64) strip(x, y, v0, v1, v2);
      f2 LDS *d = (f2 LDS *)(sm.t64 + r * T64_ROW + x + 2);             // (even word offsets: 8-byte stores)
      d[0] = (f2){ v0[0], v0[1] }; d[1] = (f2){ v0[2], v0[3] };
      d[T64_CH / 2] = (f2){ v1[0], v1[1] }; d[T64_CH / 2 + 1] = (f2){ v1[2], v1[3] };
      d[T64_CH] = (f2){ v2[0], v2[1] }; d[T64_CH + 1] = (f2){ v2[2], v2[3] };
    }
    __syncthreads();
    conv5_mfma<4>(sm, sm.t64, sm.koff64, W + HEVCDL_W_C64, W[HEVCDL_W_SCALES + 1], sm.act12, 8, tid, h + 1);
  }

  CNN_MARK(1);
  const int i16 = lane & 15, g4 = lane >> 4;
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    // ---- conv1 on the 32x32 quadrant (use_model.py:20-25); its input tile shares storage with conv2's output ----
    for (int i = tid; i < 3 * (4 * T32_ROW + 32 * 8); i += 256) {     // halo of the quadrant tile (storage is shared with conv2's output: rewritten per quadrant)
      const int c = i / (4 * T32_ROW + 256), r = i - c * (4 * T32_ROW + 256);
      int o;
      if (r < 2 * T32_ROW) o = r; else if (r < 4 * T32_ROW) o = 34 * T32_ROW + (r - 2 * T32_ROW);
      else { const int q2 = r - 4 * T32_ROW, y = q2 >> 3, k = q2 & 7; o = (y + 2) * T32_ROW + (k < 2 ? k : 32 + k); }
      sm.q.t32[c * T32_CH + o] = 0.f;
    }
    { // one strip of 4 samples per thread
      const int y = tid >> 3, x = (tid & 7) * 4;
      float v0[4], v1[4], v2[4];
      strip((q & 1) * 32 + x, (q >> 1) * 32 + y, v0, v1, v2);
      f2 LDS *d = (f2 LDS *)(sm.q.t32 + (y + 2) * T32_ROW + x + 2);
      d[0] = (f2){ v0[0], v0[1] }; d[1] = (f2){ v0[2], v0[3] };
      d[T32_CH / 2] = (f2){ v1[0], v1[1] }; d[T32_CH / 2 + 1] = (f2){ v1[2], v1[3] };
      d[T32_CH] = (f2){ v2[0], v2[1] }; d[T32_CH + 1] = (f2){ v2[2], v2[3] };
    }
    __syncthreads();
    CNN_MARK(2);
    conv5_mfma<2>(sm, sm.q.t32, sm.koff32, W + HEVCDL_W_C1, W[HEVCDL_W_SCALES + 0], sm.act12, 0, tid, 0);
    CNN_MARK(3);
    for (int i = tid; i < 16 * 36; i += 256) {                       // the input tile is dead: zero the 36 border positions of conv2's 10 x 10 output maps, four words each, 16 arrays
      const int c = i / 36, r = i - c * 36;                          // 0..9 top row, 10..19 bottom row, 20..35 the sides of rows 1..8
      const int o = r < 10 ? r : (r < 20 ? 80 + r : (1 + ((r - 20) >> 1)) * 10 + ((r - 20) & 1) * 9);
      *(u4 LDS *)(sm.q.a2 + c * A2P + o * 4) = (u4){ 0u, 0u, 0u, 0u };
    }
    __syncthreads();

    // ---- conv2: 32 -> 64, 3x3, on cat(conv1, conv64) (use_model.py:26-31, 50) ----------------------
    // wave w owns M-tiles 4w .. 4w + 3 and ALL four N-tiles: an A tile is gathered from LDS once and multiplied into
    // 64 output channels (gathering it per N-tile, one N-tile per wave, made the operand reads the bound).  The BN statistics of a channel are
    // then spread over the four waves and meet in LDS.  k = tap * 32 + ic.
    // Banks: an operand is one ds_read_b128 (the map layouts at the top of the file).  An M-tile is 2 columns x 8 rows (four pool windows stacked): with a row
    // pitch of 18 positions (10 for conv3's maps) its 16 positions are distinct mod 16, i.e. the 16 lanes a b128 read serves at a time hit 16 different quarters of
    // the 64 banks.  (The 8 x 2 tiles of round 3, read word by word from channel-major maps, had a 2-way conflict in every read: 43 % of the kernel's LDS cycles.)
    {
      // packed weights: [N-tile][tap][hi | lo][64 lanes] x 16 bytes (8 halves: k = 8 * (lane >> 4) + j <-> half j & 1 of channel pair 4 * (j >> 1) + (lane >> 4), pack_conv3)
      // (the layer's base is made opaque per quadrant and the lane's offset is a 32-bit register: every weight load is then `scalar base + register + immediate`;
      // with 64-bit lane pointers the unrolled sequence's 72 addresses were computed ahead of the quadrant loop and spilled)
      const char GLB *wb2 = (const char GLB *)(W + HEVCDL_W_C2); asm volatile("" : "+s"(wb2));
      const unsigned lb = lane * 16u;
      auto w2 = [&](int idx) { return *(const u4 GLB *)(wb2 + (size_t)(lb + (unsigned)idx * 1024u)); };      // idx: 1 KB rows of 64 lanes x 16 bytes
      const float GLB *b2 = W + HEVCDL_W_C2 + 18432;
      v4f acc[4][4];                                                  // [M-tile][N-tile]
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const float bias = b2[n * 16 + i16];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t][n] = (v4f){ bias, bias, bias, bias };
      }
      // lane -> (pool window i16 >> 2 of the tile, member i16 & 3), input channel group g4; tile T = 4 * wave + t sits at rows 8 * (T >> 3), columns 2 * (T & 7)
      const u4 LDS *abase = (const u4 LDS *)(sm.act12 + (2 * g4) * AP) + (2 * (i16 >> 2) + ((i16 >> 1) & 1)) * A_ROW + (i16 & 1) + (8 * (wave >> 1)) * A_ROW + 8 * (wave & 1);      // (in positions = 16 bytes)
      // The 36 (tap, tile) steps are one unrolled sequence: the 8 LDS reads of step p + 1 are issued before the MFMAs of step p -- across taps too --, and the weights
      // of tap + 1 ([N-tile][hi | lo], two register sets) are requested when tap begins.  The scheduling barriers keep that order (left alone the loads sink to their uses).
      u4 bq[2][8];
#pragma unroll
      for (int n = 0; n < 4; n++) { bq[0][2 * n] = w2(n * 18); bq[0][2 * n + 1] = w2(n * 18 + 1); }
      auto aoff = [](int p) { return ((p >> 2) / 3) * A_ROW + ((p >> 2) % 3) + 2 * (p & 3); };
      // the operand pair of a step: the lane group's hi array at the position, and its lo array (AP words on)
      u4 h0 = abase[aoff(0)], l0 = abase[AP / 4 + aoff(0)], h1, l1;
#pragma unroll
      for (int p = 0; p < 36; p++) {
        const int tap = p >> 2, t = p & 3;
        if (t == 0 && tap < 8) {
#pragma unroll
          for (int n = 0; n < 4; n++) { bq[(tap + 1) & 1][2 * n] = w2(n * 18 + (tap + 1) * 2); bq[(tap + 1) & 1][2 * n + 1] = w2(n * 18 + (tap + 1) * 2 + 1); }
        }
        u4 &hc = (p & 1) ? h1 : h0, &lc = (p & 1) ? l1 : l0, &hn = (p & 1) ? h0 : h1, &ln = (p & 1) ? l0 : l1;
        if (p < 35) { hn = abase[aoff(p + 1)]; ln = abase[AP / 4 + aoff(p + 1)]; }
        __builtin_amdgcn_sched_barrier(0);
        const h8 ah = __builtin_bit_cast(h8, hc), al = __builtin_bit_cast(h8, lc);
#pragma unroll
        for (int n = 0; n < 4; n++) acc[t][n] = mfma3(ah, al, __builtin_bit_cast(h8, bq[tap & 1][2 * n]), __builtin_bit_cast(h8, bq[tap & 1][2 * n + 1]), acc[t][n]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // statistics: this wave's 64 positions of every channel, then the four waves through LDS
#pragma unroll
      for (int n = 0; n < 4; n++) {
        double s = 0, ss = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const v4f a = acc[t][n];
          s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
          ss = __builtin_fma((double)a.x, (double)a.x, ss); ss = __builtin_fma((double)a.y, (double)a.y, ss); ss = __builtin_fma((double)a.z, (double)a.z, ss); ss = __builtin_fma((double)a.w, (double)a.w, ss);
        }
        s += shfl_xor_d(s, 16); s += shfl_xor_d(s, 32); ss += shfl_xor_d(ss, 16); ss += shfl_xor_d(ss, 32);
        if (lane < 16) { sm.red2[0][wave][n * 16 + lane] = s; sm.red2[1][wave][n * 16 + lane] = ss; }
      }
      __syncthreads();
      float al_l, be_l;                                               // the affine map of channel `lane` (every wave folds all 64: it needs them for its own tiles)
      if (p.bn_eval) { al_l = b2[64 + lane]; be_l = b2[128 + lane]; }
      else {
        double s = 0, ss = 0;
#pragma unroll
        for (int wv = 0; wv < 4; wv++) { s += sm.red2[0][wv][lane]; ss += sm.red2[1][wv][lane]; }
        bn_fold(s, ss, 256.0, b2[64 + lane], b2[128 + lane], W[HEVCDL_W_SCALES + 2], al_l, be_l);
      }
      // conv3's operand form: plane m < 32 holds the hi halves of a PAIR of channels, plane 32 + m their lo halves; a lane pairs its N-tiles 0 | 1 (channels i16 | 16 + i16,
      // pair i16) and 2 | 3 (32 + i16 | 48 + i16, pair 16 + i16)
#pragma unroll
      for (int np = 0; np < 2; np++) {
        const int c0 = 32 * np + i16, c1 = c0 + 16;
        const float al0 = __shfl(al_l, c0), be0 = __shfl(be_l, c0), al1 = __shfl(al_l, c1), be1 = __shfl(be_l, c1);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const v4f a = acc[t][2 * np], b = acc[t][2 * np + 1];
          const int T = 4 * wave + t;
          const float v0 = fmaxf(fmaxf(a.x * al0 + be0, a.y * al0 + be0), fmaxf(a.z * al0 + be0, a.w * al0 + be0));
          const float v1 = fmaxf(fmaxf(b.x * al1 + be1, b.y * al1 + be1), fmaxf(b.z * al1 + be1, b.w * al1 + be1));
          float hi, lo;
          split_pair(fmaxf(v0, 0.f), fmaxf(v1, 0.f), hi, lo);
          float LDS *d = sm.q.a2 + ((4 * np + (i16 & 3)) * 2) * A2P + ((4 * (T >> 3) + g4 + 1) * A2_ROW + (T & 7) + 1) * 4 + (i16 >> 2);      // window g4 of tile T; pair 16 np + i16
          d[0] = hi; d[A2P] = lo;
        }
      }
      __syncthreads();
    }
    CNN_MARK(4);
    // ---- conv3: 64 -> 128, 3x3 (use_model.py:32-37); wave w owns output channels [32w, 32w+32) = 2 N-tiles, 4 M-tiles ----
    {
      // packed weights: [N-tile][tap][k-step s][hi | lo][64 lanes] x 16 bytes (k = 8 * (lane >> 4) + j <-> half j & 1 of channel pair 16 * s + 4 * (j >> 1) + (lane >> 4), pack_conv3)
      const char GLB *wb3 = (const char GLB *)(W + HEVCDL_W_C3) + (size_t)(2 * wave) * (9 * 4096); asm volatile("" : "+s"(wb3));
      const unsigned lb = lane * 16u;
      auto w3 = [&](int idx) { return *(const u4 GLB *)(wb3 + (size_t)(lb + (unsigned)idx * 1024u)); };
      const float GLB *b3 = W + HEVCDL_W_C3 + 73728;
      v4f acc[2][4];
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const float bias = b3[wave * 32 + n * 16 + i16];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[n][t] = (v4f){ bias, bias, bias, bias };
      }
      const u4 LDS *abase = (const u4 LDS *)(sm.q.a2 + (2 * g4) * A2P) + (2 * (i16 >> 2) + ((i16 >> 1) & 1)) * A2_ROW + (i16 & 1);      // M-tile t: columns 2t, 2t + 1, all 8 rows (in positions = 16 bytes)
      // 72 (tap, k-step, tile) steps as one unrolled sequence, as in conv2; weights [N-tile n][k-step s][hi | lo] of a tap in two register sets
      u4 bq[2][8];
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int q = 0; q < 4; q++) bq[0][n * 4 + q] = w3(n * 36 + q);
      // the operand pair of a step: k-step sk's arrays of the lane group (8 arrays of A2P words on), hi at the position, lo A2P words on
      auto aoff = [](int p) { return ((p >> 2) & 1) * (8 * A2P / 4) + ((p >> 3) / 3) * A2_ROW + ((p >> 3) % 3) + 2 * (p & 3); };
      u4 h0 = abase[aoff(0)], l0 = abase[A2P / 4 + aoff(0)], h1, l1;
#pragma unroll
      for (int p = 0; p < 72; p++) {
        const int tap = p >> 3, sk = (p >> 2) & 1, t = p & 3;
        if ((p & 7) == 0 && tap < 8) {
#pragma unroll
          for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) bq[(tap + 1) & 1][n * 4 + q] = w3(n * 36 + (tap + 1) * 4 + q);
        }
        u4 &hc = (p & 1) ? h1 : h0, &lc = (p & 1) ? l1 : l0, &hn = (p & 1) ? h0 : h1, &ln = (p & 1) ? l0 : l1;
        if (p < 71) { hn = abase[aoff(p + 1)]; ln = abase[A2P / 4 + aoff(p + 1)]; }
        __builtin_amdgcn_sched_barrier(0);
        const h8 ah = __builtin_bit_cast(h8, hc), al = __builtin_bit_cast(h8, lc);
        acc[0][t] = mfma3(ah, al, __builtin_bit_cast(h8, bq[tap & 1][2 * sk]), __builtin_bit_cast(h8, bq[tap & 1][2 * sk + 1]), acc[0][t]);
        acc[1][t] = mfma3(ah, al, __builtin_bit_cast(h8, bq[tap & 1][4 + 2 * sk]), __builtin_bit_cast(h8, bq[tap & 1][4 + 2 * sk + 1]), acc[1][t]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const int ch = wave * 32 + n * 16 + i16;
        double s = 0, ss = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const v4f a = acc[n][t];
          s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
          ss = __builtin_fma((double)a.x, (double)a.x, ss); ss = __builtin_fma((double)a.y, (double)a.y, ss); ss = __builtin_fma((double)a.z, (double)a.z, ss); ss = __builtin_fma((double)a.w, (double)a.w, ss);
        }
        s += shfl_xor_d(s, 16); s += shfl_xor_d(s, 32); ss += shfl_xor_d(ss, 16); ss += shfl_xor_d(ss, 32);
        float al, be;
        if (p.bn_eval) { al = b3[128 + ch]; be = b3[256 + ch]; }
        else bn_fold(s, ss, 64.0, b3[128 + ch], b3[256 + ch], W[HEVCDL_W_SCALES + 3], al, be);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const v4f a = acc[n][t];
          const float v = fmaxf(fmaxf(a.x * al + be, a.y * al + be), fmaxf(a.z * al + be, a.w * al + be));
          a3_out[(size_t)q * 2048 + ch * 16 + g4 * 4 + t] = split_f16(fmaxf(v, 0.f));      // flatten order (C,H,W), use_model.py:53; row 4 * ctu + q of the head's input, stored split: fc1's operand form (fc_kernel.hip)
        }
      }
      __syncthreads();
    }
    CNN_MARK(5);
  }

  // the fully connected head (fc1..fc3) and the label logic run batched over CTUs in fc_kernel.hip
#ifdef HEVCDL_CNN_PROF
  CNN_MARK(7);
  if (tid == 0 && gctu == 0 && p.logits) for (int i = 0; i < 8; i++) ((float GLB *)p.logits)[i] = (float)pt_[i];
#endif
}

#ifdef HEVCDL_CNN_PROF
#endif
extern "C" size_t hevcdl_cnn_smem_bytes(void) { return sizeof(CnnSmem); }
