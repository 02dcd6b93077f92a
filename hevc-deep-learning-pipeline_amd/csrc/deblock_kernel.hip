// deblock_kernel.hip -- HEVC deblocking filter for gfx950 (MI355X), row f-2 of SURVEY.md section 8 (first half; SAO: sao_kernel.hip).
//
// Replaces, for the configuration of the hot path (all-intra => boundary strength 2 on every TU/CU edge, one slice,
// constant QP, beta/tc offsets 0, no PCM / lossless, 8- or 10-bit 4:2:0; with tiles the filter crosses tile borders):
//   TComLoopFilter::loopFilterPic        HM_dl/source/Lib/TLibCommon/TComLoopFilter.cpp:130-156
//   xDeblockCU / xSetEdgefilterTU / PU   :170-360   (left/top edge of every TU, on the 8x8 grid, not at the picture border)
//   xEdgeFilterLuma / xEdgeFilterChroma  :557-826
//   xPelFilterLuma / xPelFilterChroma / xUseStrongFiltering / xCalcDP / xCalcDQ   :830-954
// Input: the reconstruction hevcdl_compress_frames leaves + the CTU records (depth, trIdx give the TU grid).
//
// This stage is genuinely HBM bound (a few ALU ops per sample).  The reference filters all vertical edges of the picture, then all
// horizontal edges; here both happen in one kernel over LDS tiles placed so that no tile needs a halo (hevcdl_deblock_fused_kernel):
// one read and one write of the picture in all.  A thread owns an exclusive 8x4 (vertical edges) / 4x8 (horizontal edges) block of samples
// centred on one 4-sample edge segment -- the unit of the filter decision -- with dword accesses, adjacent lanes on adjacent dwords.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hevcdl_dev.h"

namespace {

#define GLB __attribute__((address_space(1)))
enum { REC_SIZE = 15120, REC_TRIDX = 4 * 256 };

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clipbd(int v, int mx) { return v < 0 ? 0 : (v > mx ? mx : v); }      // ClipBD

// four neighbouring samples of a row as one access: a dword of 8-bit samples, two dwords of 16-bit samples (10-bit pictures)
template <typename PEL> __device__ __forceinline__ void load4(const PEL GLB *p, int (&v)[4])
{
  if constexpr (sizeof(PEL) == 1) { const uint32_t w = *(const uint32_t GLB *)p; v[0] = w & 255; v[1] = (w >> 8) & 255; v[2] = (w >> 16) & 255; v[3] = w >> 24; }
  else { const unsigned long long w = *(const unsigned long long GLB *)p; v[0] = (int)(w & 0xffff); v[1] = (int)((w >> 16) & 0xffff); v[2] = (int)((w >> 32) & 0xffff); v[3] = (int)(w >> 48); }
}
template <typename PEL> __device__ __forceinline__ void store4(PEL GLB *p, int a, int b, int c, int d)
{
  if constexpr (sizeof(PEL) == 1) *(uint32_t GLB *)p = (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
  else *(unsigned long long GLB *)p = (unsigned long long)(uint32_t)a | ((unsigned long long)(uint32_t)b << 16) | ((unsigned long long)(uint32_t)c << 32) | ((unsigned long long)(uint32_t)d << 48);
}

// TU edge at the left (dir 0) / top (dir 1) border of the 4x4 partition at luma (x, y) of this frame?
__device__ __forceinline__ bool edge_flag(const hevcdl_dbk_params &p, const unsigned char GLB *recs, int ctus_x, int x, int y, int dir)
{
  if (!p.lf_across_tiles) { // xSetLoopfilterParam TComLoopFilter.cpp:362-400: the neighbouring CU of another tile does not exist for the filter
    const int pos = (dir ? y : x) >> 6, n = dir ? p.tile_rows : p.tile_cols;
    if ((((dir ? y : x)) & 63) == 0) for (int t = 1; t < n; t++) if ((dir ? p.row_bd[t] : p.col_bd[t]) == pos) return false;
  }
  const unsigned char GLB *r = recs + (size_t)((y >> 6) * ctus_x + (x >> 6)) * REC_SIZE;
  const int x4 = (x & 63) >> 2, y4 = (y & 63) >> 2;
  int z = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) z |= (((x4 >> b) & 1) << (2 * b)) | (((y4 >> b) & 1) << (2 * b + 1));
  const int tu = 64 >> (r[z] + r[REC_TRIDX + z]);
  const int pos = dir ? y : x;
  return pos > 0 && (pos & (tu - 1)) == 0;
}

// Luma decision + filter of one 4-line segment (xEdgeFilterLuma / xPelFilterLuma, TComLoopFilter.cpp:557-700, 830-920), two lines at a time in
// packed 16-bit arithmetic (v_pk_*_i16): every intermediate fits 16 bits at 8 and at 10 bits (largest: 9 * 1023 + 3 * 1023 + 8).  Lines are
// paired (0, 3) and (1, 2): the decisions are taken from lines 0 and 3, so one packed evaluation gives both; the per-line condition of the
// normal filter becomes a half-word mask.  Positions of a line: p3 p2 p1 p0 | q0 q1 q2 q3.
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 pk(int lo, int hi) { return __builtin_bit_cast(s2, ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16)); }
__device__ __forceinline__ s2 pk1(int v) { return pk(v, v); }
__device__ __forceinline__ s2 pk_abs(s2 v) { const s2 n = -v; return __builtin_elementwise_max(v, n); }
__device__ __forceinline__ s2 pk_clip(s2 lo, s2 hi, s2 v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }
__device__ __forceinline__ s2 pk_sel(s2 mask, s2 a, s2 b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ void filter_luma_pairs(s2 (&a)[8], s2 (&b)[8], int tc, int beta, int mx)
{ // a: lines (0, 3); b: lines (1, 2); positions p3 p2 p1 p0 | q0 q1 q2 q3
  const s2 two = pk1(2);
  const s2 dpv = pk_abs(a[1] - two * a[2] + a[3]), dqv = pk_abs(a[4] - two * a[5] + a[6]);
  const int dp0 = dpv.x, dp3 = dpv.y, dq0 = dqv.x, dq3 = dqv.y;
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
  if (d >= beta) return;
  const int side = (beta + (beta >> 1)) >> 3, thr_cut = tc * 10;
  const bool fp = (dp0 + dp3) < side, fq = (dq0 + dq3) < side;
  const s2 flat = pk_abs(a[0] - a[3]) + pk_abs(a[7] - a[4]), gap = pk_abs(a[3] - a[4]);
  const bool sw = flat.x < (beta >> 3) && flat.y < (beta >> 3) && 2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2) && gap.x < ((tc * 5 + 1) >> 1) && gap.y < ((tc * 5 + 1) >> 1);
  const s2 zero = pk1(0), vmx = pk1(mx), vtc = pk1(tc), vtc2 = pk1(tc >> 1), v2tc = pk1(2 * tc);
  auto run = [&](s2 (&v)[8]) {
    const s2 m0 = v[0], m1 = v[1], m2 = v[2], m3 = v[3], m4 = v[4], m5 = v[5], m6 = v[6], m7 = v[7];
    if (sw) {
      v[3] = pk_clip(m3 - v2tc, m3 + v2tc, (m1 + two * m2 + two * m3 + two * m4 + m5 + pk1(4)) >> 3);
      v[4] = pk_clip(m4 - v2tc, m4 + v2tc, (m2 + two * m3 + two * m4 + two * m5 + m6 + pk1(4)) >> 3);
      v[2] = pk_clip(m2 - v2tc, m2 + v2tc, (m1 + m2 + m3 + m4 + two) >> 2);
      v[5] = pk_clip(m5 - v2tc, m5 + v2tc, (m3 + m4 + m5 + m6 + two) >> 2);
      v[1] = pk_clip(m1 - v2tc, m1 + v2tc, (two * m0 + pk1(3) * m1 + m2 + m3 + m4 + pk1(4)) >> 3);
      v[6] = pk_clip(m6 - v2tc, m6 + v2tc, (m3 + m4 + m5 + pk1(3) * m6 + two * m7 + pk1(4)) >> 3);
    } else {
      const s2 delta0 = (pk1(9) * (m4 - m3) - pk1(3) * (m5 - m2) + pk1(8)) >> 4;
      const s2 on = pk_abs(delta0) < pk1(thr_cut);          // half-word mask: this line is filtered
      const s2 delta = pk_clip(-vtc, vtc, delta0);
      v[3] = pk_sel(on, pk_clip(zero, vmx, m3 + delta), m3);
      v[4] = pk_sel(on, pk_clip(zero, vmx, m4 - delta), m4);
      if (fp) v[2] = pk_sel(on, pk_clip(zero, vmx, m2 + pk_clip(-vtc2, vtc2, ((((m1 + m3 + pk1(1)) >> 1) - m2 + delta) >> 1))), m2);
      if (fq) v[5] = pk_sel(on, pk_clip(zero, vmx, m5 + pk_clip(-vtc2, vtc2, ((((m6 + m4 + pk1(1)) >> 1) - m5 - delta) >> 1))), m5);
    }
  };
  run(a); run(b);
}
__device__ __forceinline__ void filter_luma_pk(int (&m)[4][8], int tc, int beta, int mx)
{
  s2 a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { a[k] = pk(m[0][k], m[3][k]); b[k] = pk(m[1][k], m[2][k]); }
  filter_luma_pairs(a, b, tc, beta, mx);
#pragma unroll
  for (int k = 1; k < 7; k++) { m[0][k] = a[k].x; m[3][k] = a[k].y; m[1][k] = b[k].x; m[2][k] = b[k].y; }
}
// 8-bit samples: line pairs straight out of / into the dwords of four samples with v_perm_b32 (selector bytes 0-3: second operand, 4-7: first, 0x0c: zero)
__device__ __forceinline__ s2 perm_s2(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_bit_cast(s2, __builtin_amdgcn_perm(hi, lo, sel)); }
__device__ __forceinline__ uint32_t u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
// rows[i] = the four samples k0 .. k0 + 3 of line i  ->  v[k0 + k] = (line la, line lb) at position k0 + k
__device__ __forceinline__ void rows_to_pairs(uint32_t ra, uint32_t rb, s2 (&v)[8], int k0)
{
#pragma unroll
  for (int k = 0; k < 4; k++) v[k0 + k] = perm_s2(rb, ra, 0x0c000c00u | ((uint32_t)(4 + k) << 16) | (uint32_t)k);
}
__device__ __forceinline__ void pairs_to_rows(const s2 (&v)[8], int k0, uint32_t &ra, uint32_t &rb)
{
  const uint32_t u = __builtin_amdgcn_perm(u32(v[k0 + 1]), u32(v[k0]), 0x06020400u), w = __builtin_amdgcn_perm(u32(v[k0 + 3]), u32(v[k0 + 2]), 0x06020400u);
  ra = __builtin_amdgcn_perm(w, u, 0x05040100u); rb = __builtin_amdgcn_perm(w, u, 0x07060302u);
}
// chroma on two lines at a time: p1 p0 | q0 q1 -> p0, q0
__device__ __forceinline__ void filter_chroma_pairs(s2 p1, s2 &p0, s2 &q0, s2 q1, int tc, int mx)
{
  const s2 delta = pk_clip(pk1(-tc), pk1(tc), (((q0 - p0) << 2) + p1 - q1 + pk1(4)) >> 3);
  p0 = pk_clip(pk1(0), pk1(mx), p0 + delta); q0 = pk_clip(pk1(0), pk1(mx), q0 - delta);
}
// chroma (Bs 2): p1 p0 | q0 q1 -> p0, q0
__device__ __forceinline__ void filter_chroma(int m2, int &m3, int &m4, int m5, int tc, int mx)
{
  const int delta = clip3i(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
  m3 = clipbd(m3 + delta, mx); m4 = clipbd(m4 - delta, mx);
}


} // namespace

// Both passes in one kernel, every sample read once and written once.  A workgroup owns the TW x TH (256 x 32) samples whose top-left corner
// sits 4 samples left of and above a point of the TW x TH grid: [X0 - 4, X0 + TW - 4) x [Y0 - 4, Y0 + TH - 4).  An edge of the 8x8 grid reads 4 samples
// and changes at most 3 on either side, so every vertical edge X0 + 8i and every horizontal edge Y0 + 8j lies with everything it reads and
// writes inside that rectangle -- and the horizontal edges read nothing but vertically filtered samples of the same rectangle: no halo, no
// second trip through HBM.  Stage 1: thread = (edge column, 4-row segment), 8 x 4 samples from HBM (two dword loads per row, adjacent
// lanes adjacent), vertical edge filtered in registers, block to LDS.  Stage 2: thread = (4-sample column group, edge row), 4 x 8 samples from
// LDS, horizontal edge filtered, block to HBM (a dword store per row, adjacent lanes adjacent).  CHROMA == 1: the same on both chroma planes
// in chroma coordinates (edge grid 8 chroma = 16 luma samples; a 4-sample chroma segment spans two luma partitions = two edge flags; the
// filter reads 2 and changes 1 sample per side).  in == out is allowed (a workgroup reads and writes its own rectangle only).
#ifndef DBK_TW
#define DBK_TW 256
#define DBK_TH 32
#endif
#ifndef DBK_WAVES
#define DBK_WAVES 8
#endif
template <int CHROMA, typename PEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DBK_WAVES, DBK_WAVES))) void hevcdl_deblock_fused_kernel(hevcdl_dbk_params p)
{
  constexpr int TW = DBK_TW, TH = DBK_TH, DW = TW * (int)sizeof(PEL) / 4, QD = (int)sizeof(PEL);      // dwords per tile row; dwords per 4 samples
  constexpr int EXN = TW / 8, XGN = TW / 4;                 // edge columns / 4-sample column groups per tile
  static_assert(EXN * (TH / 4) == 256 && XGN * (TH / 8) == 256, "one thread per block in both stages");
  __shared__ uint32_t tile[TH][DW + 1];
  const int frame = blockIdx.z, tid = threadIdx.x;
  const int W = p.width >> CHROMA, H = p.height >> CHROMA;
  // Neighbouring tiles of a row share the cache lines at their common border (the rectangles start 4 samples left of the grid).  Workgroups
  // are dealt to the 8 XCDs round-robin in launch order, each XCD with its own L2: the launch index is therefore remapped so that every XCD
  // gets a contiguous run of tiles -- the two halves of a shared line meet in one L2 and leave it as a whole line.
  const int tiles_x = (W + 4 + TW - 1) / TW, n_tiles = tiles_x * ((H + 4 + TH - 1) / TH), per_xcd = (n_tiles + 7) / 8;
  const int lin = (int)blockIdx.x, vt = (lin & 7) * per_xcd + (lin >> 3);
  if (vt >= n_tiles) return;
  const int X0 = (vt % tiles_x) * TW, Y0 = (vt / tiles_x) * TH;
  const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
  const unsigned char GLB *recs = (const unsigned char GLB *)p.records + (size_t)frame * p.ctus_per_frame * REC_SIZE;
  // stage 1 role: edge column x, rows y .. y + 3
  const int ex = tid % EXN, sy = tid / EXN;
  const int x = X0 + 8 * ex, y = Y0 - 4 + 4 * sy;
  const bool rows_ok = y >= 0 && y < H, has_l = x > 0 && x <= W, has_r = x < W;
  // stage 2 role: columns xh .. xh + 3, edge row yh, rows yh - 4 .. yh + 3
  const int xg = tid % XGN, eb = tid / XGN;
  const int xh = X0 - 4 + 4 * xg, yh = Y0 + 8 * eb;
  const bool cols_ok = xh >= 0 && xh < W, has_u = yh > 0 && yh <= H, has_d = yh < H;
  bool v0 = false, v1 = false, h0 = false, h1 = false;
#pragma unroll 1
  for (int c = 0; c < (CHROMA ? 2 : 1); c++) {
    const size_t plane = (size_t)frame * fsz + (CHROMA ? ysz + (size_t)c * (ysz >> 2) : 0);
    const PEL GLB *src = (const PEL GLB *)p.in + plane; PEL GLB *dst = (PEL GLB *)p.out + plane;
    if (c) __syncthreads();                                 // stage 2 of the first chroma plane has read the tile
    constexpr bool RAW = sizeof(PEL) == 1;                    // 8-bit samples: blocks stay dwords; only blocks with an edge are opened up, into packed line pairs
    if constexpr (RAW) {
      if (rows_ok && (has_l || has_r)) {
        uint32_t l[4] = { 0, 0, 0, 0 }, r[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const size_t o = (size_t)(y + i) * W + x;
          if (has_l) l[i] = *(const uint32_t GLB *)(src + o - 4);
          if (has_r) r[i] = *(const uint32_t GLB *)(src + o);
        }
        if constexpr (!CHROMA) {
          if (has_l && has_r) v0 = edge_flag(p, recs, p.ctus_x, x, y, 0);
          if (v0) {
            s2 a[8], b[8];
            rows_to_pairs(l[0], l[3], a, 0); rows_to_pairs(r[0], r[3], a, 4); rows_to_pairs(l[1], l[2], b, 0); rows_to_pairs(r[1], r[2], b, 4);
            filter_luma_pairs(a, b, p.tc, p.beta, p.pel_max);
            pairs_to_rows(a, 0, l[0], l[3]); pairs_to_rows(a, 4, r[0], r[3]); pairs_to_rows(b, 0, l[1], l[2]); pairs_to_rows(b, 4, r[1], r[2]);
          }
        } else {
          if (c == 0 && has_l && has_r) { v0 = edge_flag(p, recs, p.ctus_x, 2 * x, 2 * y, 0); v1 = edge_flag(p, recs, p.ctus_x, 2 * x, 2 * y + 4, 0); }
#pragma unroll
          for (int h = 0; h < 2; h++) if (h ? v1 : v0) { // rows (0, 1) under the first flag, (2, 3) under the second: p1 p0 = bytes 2, 3 of the left dwords, q0 q1 = bytes 0, 1 of the right ones
            uint32_t &la = l[2 * h], &lb = l[2 * h + 1], &ra = r[2 * h], &rb = r[2 * h + 1];
            const s2 p1 = perm_s2(lb, la, 0x0c060c02u), q1 = perm_s2(rb, ra, 0x0c050c01u);
            s2 p0 = perm_s2(lb, la, 0x0c070c03u), q0 = perm_s2(rb, ra, 0x0c040c00u);
            filter_chroma_pairs(p1, p0, q0, q1, p.tc_c, p.pel_max);
            la = __builtin_amdgcn_perm(u32(p0), la, 0x04020100u); lb = __builtin_amdgcn_perm(u32(p0), lb, 0x06020100u);
            ra = __builtin_amdgcn_perm(u32(q0), ra, 0x03020104u); rb = __builtin_amdgcn_perm(u32(q0), rb, 0x03020106u);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { uint32_t *row = &tile[4 * sy + i][2 * ex]; row[0] = l[i]; row[1] = r[i]; }
      }
    } else
    if (rows_ok && (has_l || has_r)) {
      int m[4][8];                                            // row i: p3 p2 p1 p0 | q0 q1 q2 q3
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const size_t o = (size_t)(y + i) * W + x;
        int l[4] = { 0, 0, 0, 0 }, r[4] = { 0, 0, 0, 0 };
        if (has_l) load4<PEL>(src + o - 4, l);
        if (has_r) load4<PEL>(src + o, r);
#pragma unroll
        for (int k = 0; k < 4; k++) { m[i][k] = l[k]; m[i][4 + k] = r[k]; }
      }
      if (c == 0) { // the TU grid is looked up behind the sample loads: both are in flight together
        if (has_l && has_r) {
          if (CHROMA) { v0 = edge_flag(p, recs, p.ctus_x, 2 * x, 2 * y, 0); v1 = edge_flag(p, recs, p.ctus_x, 2 * x, 2 * y + 4, 0); }
          else v0 = edge_flag(p, recs, p.ctus_x, x, y, 0);
        }
      }
      if (!CHROMA) { if (v0) filter_luma_pk(m, p.tc, p.beta, p.pel_max); }
      else {
#pragma unroll
        for (int i = 0; i < 4; i++) if (i < 2 ? v0 : v1) filter_chroma(m[i][2], m[i][3], m[i][4], m[i][5], p.tc_c, p.pel_max);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        uint32_t *row = &tile[4 * sy + i][2 * QD * ex];
        if constexpr (sizeof(PEL) == 1) {
          row[0] = (uint32_t)m[i][0] | ((uint32_t)m[i][1] << 8) | ((uint32_t)m[i][2] << 16) | ((uint32_t)m[i][3] << 24);
          row[1] = (uint32_t)m[i][4] | ((uint32_t)m[i][5] << 8) | ((uint32_t)m[i][6] << 16) | ((uint32_t)m[i][7] << 24);
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) row[k] = (uint32_t)m[i][2 * k] | ((uint32_t)m[i][2 * k + 1] << 16);
        }
      }
    }
    if (c == 0 && cols_ok && has_u && has_d) { // (issued before the barrier: back by the time stage 2 needs it)
      if (CHROMA) { h0 = edge_flag(p, recs, p.ctus_x, 2 * xh, 2 * yh, 1); h1 = edge_flag(p, recs, p.ctus_x, 2 * xh + 4, 2 * yh, 1); }
      else h0 = edge_flag(p, recs, p.ctus_x, xh, yh, 1);
    }
    __syncthreads();
    if constexpr (RAW) {
      if (cols_ok && (has_u || has_d)) {
        uint32_t w[8];                                        // row k of the band: the four lines (columns) at position k
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = tile[8 * eb + k][xg];
        if constexpr (!CHROMA) {
          if (h0) {
            s2 a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { a[k] = perm_s2(w[k], w[k], 0x0c030c00u); b[k] = perm_s2(w[k], w[k], 0x0c020c01u); }
            filter_luma_pairs(a, b, p.tc, p.beta, p.pel_max);
#pragma unroll
            for (int k = 1; k < 7; k++) w[k] = __builtin_amdgcn_perm(u32(b[k]), u32(a[k]), 0x02060400u);
          }
        } else if (h0 || h1) { // columns (0, 1) under the first flag, (2, 3) under the second; p1 p0 | q0 q1 = rows 2 .. 5 of the band
          s2 a[4], b[4];
#pragma unroll
          for (int k = 0; k < 4; k++) { a[k] = perm_s2(w[2 + k], w[2 + k], 0x0c010c00u); b[k] = perm_s2(w[2 + k], w[2 + k], 0x0c030c02u); }
          if (h0) filter_chroma_pairs(a[0], a[1], a[2], a[3], p.tc_c, p.pel_max);
          if (h1) filter_chroma_pairs(b[0], b[1], b[2], b[3], p.tc_c, p.pel_max);
          w[3] = __builtin_amdgcn_perm(u32(b[1]), u32(a[1]), 0x06040200u); w[4] = __builtin_amdgcn_perm(u32(b[2]), u32(a[2]), 0x06040200u);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < 4 ? has_u : has_d) *(uint32_t GLB *)(dst + (size_t)(yh - 4 + k) * W + xh) = w[k];
      }
    } else
    if (cols_ok && (has_u || has_d)) {
      int m[4][8];                                            // column i: p3 p2 p1 p0 | q0 q1 q2 q3 downwards
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t *row = &tile[8 * eb + k][QD * xg];
        if constexpr (sizeof(PEL) == 1) { const uint32_t w = row[0]; m[0][k] = w & 255; m[1][k] = (w >> 8) & 255; m[2][k] = (w >> 16) & 255; m[3][k] = w >> 24; }
        else { const uint32_t w0 = row[0], w1 = row[1]; m[0][k] = w0 & 0xffff; m[1][k] = w0 >> 16; m[2][k] = w1 & 0xffff; m[3][k] = w1 >> 16; }
      }
      if (!CHROMA) { if (h0) filter_luma_pk(m, p.tc, p.beta, p.pel_max); }
      else {
#pragma unroll
        for (int i = 0; i < 4; i++) if (i < 2 ? h0 : h1) filter_chroma(m[i][2], m[i][3], m[i][4], m[i][5], p.tc_c, p.pel_max);
      }
#pragma unroll
      for (int k = 0; k < 8; k++) if (k < 4 ? has_u : has_d) store4<PEL>(dst + (size_t)(yh - 4 + k) * W + xh, m[0][k], m[1][k], m[2][k], m[3][k]);
    }
  }
}

extern "C" void hevcdl_launch_deblock(const hevcdl_dbk_params *pp, void *stream)
{
  const hevcdl_dbk_params p = *pp;
  hipStream_t s = (hipStream_t)stream;
  const int W = p.width, H = p.height, cw = W >> 1, chh = H >> 1;
  auto grid = [&](int w, int h) { const int n = ((w + 4 + DBK_TW - 1) / DBK_TW) * ((h + 4 + DBK_TH - 1) / DBK_TH); return dim3((unsigned)(((n + 7) / 8) * 8), 1, p.n_frames); };
  const dim3 g0 = grid(W, H), g1 = grid(cw, chh);
  if (p.pel_max == 255) {
    hipLaunchKernelGGL((hevcdl_deblock_fused_kernel<0, uint8_t>), g0, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_fused_kernel<1, uint8_t>), g1, dim3(256), 0, s, p);
  } else { // 16-bit sample planes (10-bit pictures)
    hipLaunchKernelGGL((hevcdl_deblock_fused_kernel<0, uint16_t>), g0, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_fused_kernel<1, uint16_t>), g1, dim3(256), 0, s, p);
  }
}
