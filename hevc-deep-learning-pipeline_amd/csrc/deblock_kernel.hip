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
// This stage is genuinely HBM bound (one read + one write of the picture per pass, a few ALU ops per sample).
// Two passes as in the reference (all vertical edges of the picture, then all horizontal edges).  Each thread owns an
// exclusive 8x4 (vertical edges) / 4x8 (horizontal edges) block of samples centred on one 4-sample edge segment --
// the unit of the filter decision -- so a pass needs no synchronisation, every sample is read once and written once
// per pass with dword accesses, and adjacent lanes touch adjacent dwords (fully coalesced rows).
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

// luma decision + filter of one 4-line segment; m[line][0..7] = p3 p2 p1 p0 | q0 q1 q2 q3
__device__ __forceinline__ void filter_luma(int (&m)[4][8], int tc, int beta, int mx)
{
  auto dp = [&](int i) { return abs(m[i][1] - 2 * m[i][2] + m[i][3]); };
  auto dq = [&](int i) { return abs(m[i][4] - 2 * m[i][5] + m[i][6]); };
  const int dp0 = dp(0), dq0 = dq(0), dp3 = dp(3), dq3 = dq(3);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
  if (d >= beta) return;
  const int side = (beta + (beta >> 1)) >> 3, thr_cut = tc * 10;
  const bool fp = (dp0 + dp3) < side, fq = (dq0 + dq3) < side;
  auto strong = [&](int i, int dd) {
    return (abs(m[i][0] - m[i][3]) + abs(m[i][7] - m[i][4]) < (beta >> 3)) && (dd < (beta >> 2)) && (abs(m[i][3] - m[i][4]) < ((tc * 5 + 1) >> 1));
  };
  const bool sw = strong(0, 2 * d0) && strong(3, 2 * d3);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m0 = m[i][0], m1 = m[i][1], m2 = m[i][2], m3 = m[i][3], m4 = m[i][4], m5 = m[i][5], m6 = m[i][6], m7 = m[i][7];
    if (sw) {
      m[i][3] = clip3i(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
      m[i][4] = clip3i(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
      m[i][2] = clip3i(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
      m[i][5] = clip3i(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
      m[i][1] = clip3i(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
      m[i][6] = clip3i(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    } else {
      int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
      if (abs(delta) < thr_cut) {
        const int tc2 = tc >> 1;
        delta = clip3i(-tc, tc, delta);
        m[i][3] = clipbd(m3 + delta, mx); m[i][4] = clipbd(m4 - delta, mx);
        if (fp) m[i][2] = clipbd(m2 + clip3i(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)), mx);
        if (fq) m[i][5] = clipbd(m5 + clip3i(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)), mx);
      }
    }
  }
}
// chroma (Bs 2): p1 p0 | q0 q1 -> p0, q0
__device__ __forceinline__ void filter_chroma(int m2, int &m3, int &m4, int m5, int tc, int mx)
{
  const int delta = clip3i(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
  m3 = clipbd(m3 + delta, mx); m4 = clipbd(m4 - delta, mx);
}


} // namespace

// Pass 1: vertical edges.  CHROMA == 0: thread = (edge column bx, 4-row segment) of the luma plane, owns samples
// [8bx-4, 8bx+4) x 4 rows.  CHROMA == 1: the same on both chroma planes with chroma coordinates (edge grid 8 chroma =
// 16 luma samples; a 4-row chroma block spans two luma partitions = two edge flags).  Reads `in`, writes `out`
// (every sample of the plane exactly once; in == out is allowed).
template <int CHROMA, typename PEL>
__global__ __launch_bounds__(256) void hevcdl_deblock_ver_kernel(hevcdl_dbk_params p)
{
  const int frame = blockIdx.z;
  const int W = p.width >> CHROMA, H = p.height >> CHROMA;
  const int bx = blockIdx.x * 64 + (threadIdx.x & 63), seg = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (bx * 8 > W || seg * 4 >= H) return;
  const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
  const unsigned char GLB *recs = (const unsigned char GLB *)p.records + (size_t)frame * p.ctus_per_frame * REC_SIZE;
  const int x = bx * 8, y = seg * 4;
  const bool has_l = x > 0, has_r = x < W;
  bool e0 = false, e1 = false;
  if (has_l && has_r) {
    if (CHROMA) { e0 = edge_flag(p, recs, p.ctus_x, 2 * x, 2 * y, 0); e1 = edge_flag(p, recs, p.ctus_x, 2 * x, 2 * y + 4, 0); }
    else e0 = edge_flag(p, recs, p.ctus_x, x, y, 0);
  }
#pragma unroll 1
  for (int c = 0; c < (CHROMA ? 2 : 1); c++) {
    const size_t plane = (size_t)frame * fsz + (CHROMA ? ysz + (size_t)c * (ysz >> 2) : 0);
    const PEL GLB *src = (const PEL GLB *)p.in + plane; PEL GLB *dst = (PEL GLB *)p.out + plane;
    int m[4][8];                                              // row i: p3 p2 p1 p0 | q0 q1 q2 q3
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const size_t o = (size_t)(y + i) * W + x;
      int l[4] = { 0, 0, 0, 0 }, r[4] = { 0, 0, 0, 0 };
      if (has_l) load4<PEL>(src + o - 4, l);
      if (has_r) load4<PEL>(src + o, r);
#pragma unroll
      for (int k = 0; k < 4; k++) { m[i][k] = l[k]; m[i][4 + k] = r[k]; }
    }
    if (!CHROMA) {
      if (e0) filter_luma(m, p.tc, p.beta, p.pel_max);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) if (i < 2 ? e0 : e1) filter_chroma(m[i][2], m[i][3], m[i][4], m[i][5], p.tc_c, p.pel_max);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const size_t o = (size_t)(y + i) * W + x;
      if (has_l) store4<PEL>(dst + o - 4, m[i][0], m[i][1], m[i][2], m[i][3]);
      if (has_r) store4<PEL>(dst + o, m[i][4], m[i][5], m[i][6], m[i][7]);
    }
  }
}

// Pass 2: horizontal edges, in place on `out`.  Thread = (4-sample column group xg, edge row by): samples
// [4xg, 4xg+4) x rows [8by-4, 8by+4); the 4 columns are the 4 lines of the segment.  Only blocks with an edge are
// touched (a block without one keeps the values of pass 1).
template <int CHROMA, typename PEL>
__global__ __launch_bounds__(256) void hevcdl_deblock_hor_kernel(hevcdl_dbk_params p)
{
  const int frame = blockIdx.z;
  const int W = p.width >> CHROMA, H = p.height >> CHROMA;
  const int xg = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int x = xg * 4, y = by * 8;
  if (x >= W || y <= 0 || y >= H) return;
  const size_t ysz = (size_t)p.width * p.height, fsz = ysz + (ysz >> 1);
  const unsigned char GLB *recs = (const unsigned char GLB *)p.records + (size_t)frame * p.ctus_per_frame * REC_SIZE;
  bool e0, e1 = false;
  if (CHROMA) { e0 = edge_flag(p, recs, p.ctus_x, 2 * x, 2 * y, 1); e1 = edge_flag(p, recs, p.ctus_x, 2 * x + 4, 2 * y, 1); }
  else e0 = edge_flag(p, recs, p.ctus_x, x, y, 1);
  if (!e0 && !e1) return;
#pragma unroll 1
  for (int c = 0; c < (CHROMA ? 2 : 1); c++) {
    PEL GLB *pl = (PEL GLB *)p.out + (size_t)frame * fsz + (CHROMA ? ysz + (size_t)c * (ysz >> 2) : 0);
    if (!CHROMA) {
      int m[4][8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        int row[4]; load4<PEL>(pl + (size_t)(y - 4 + k) * W + x, row);
#pragma unroll
        for (int i = 0; i < 4; i++) m[i][k] = row[i];
      }
      filter_luma(m, p.tc, p.beta, p.pel_max);
#pragma unroll
      for (int k = 1; k < 7; k++) store4<PEL>(pl + (size_t)(y - 4 + k) * W + x, m[0][k], m[1][k], m[2][k], m[3][k]);
    } else {
      int v[4][4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int row[4]; load4<PEL>(pl + (size_t)(y - 2 + k) * W + x, row);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i][k] = row[i];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) if (i < 2 ? e0 : e1) filter_chroma(v[i][0], v[i][1], v[i][2], v[i][3], p.tc_c, p.pel_max);
      store4<PEL>(pl + (size_t)(y - 1) * W + x, v[0][1], v[1][1], v[2][1], v[3][1]);
      store4<PEL>(pl + (size_t)y * W + x, v[0][2], v[1][2], v[2][2], v[3][2]);
    }
  }
}

extern "C" void hevcdl_launch_deblock(const hevcdl_dbk_params *pp, void *stream)
{
  const hevcdl_dbk_params p = *pp;
  hipStream_t s = (hipStream_t)stream;
  const int W = p.width, H = p.height, cw = W >> 1, chh = H >> 1;
  const dim3 gv0((W / 8 + 1 + 63) / 64, (H / 4 + 3) / 4, p.n_frames), gv1((cw / 8 + 1 + 63) / 64, (chh / 4 + 3) / 4, p.n_frames);
  const dim3 gh0((W / 4 + 63) / 64, (H / 8 + 1 + 3) / 4, p.n_frames), gh1((cw / 4 + 63) / 64, (chh / 8 + 1 + 3) / 4, p.n_frames);
  if (p.pel_max == 255) {
    hipLaunchKernelGGL((hevcdl_deblock_ver_kernel<0, uint8_t>), gv0, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_ver_kernel<1, uint8_t>), gv1, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_hor_kernel<0, uint8_t>), gh0, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_hor_kernel<1, uint8_t>), gh1, dim3(256), 0, s, p);
  } else { // 16-bit sample planes (10-bit pictures)
    hipLaunchKernelGGL((hevcdl_deblock_ver_kernel<0, uint16_t>), gv0, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_ver_kernel<1, uint16_t>), gv1, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_hor_kernel<0, uint16_t>), gh0, dim3(256), 0, s, p);
    hipLaunchKernelGGL((hevcdl_deblock_hor_kernel<1, uint16_t>), gh1, dim3(256), 0, s, p);
  }
}
