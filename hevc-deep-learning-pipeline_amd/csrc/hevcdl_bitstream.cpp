// hevcdl_bitstream.cpp -- HEVC bitstream writer for the pictures this library decides.
// Host C++ (no GPU): the byte-producing half of the entropy coder is strictly serial per slice and tiny next to the
// decision kernel (a 2160p frame is ~80 KB of output).  SAO parameters (hevcdl_sao_frames) are optional: without them
// the stream signals SAO off.
//
// Mirrors, for the reference's all-intra configuration (one slice per picture, IntraPeriod 1, parameter sets with every
// picture, deblocking on with zero offsets, sign data hiding, transform skip):
//   access unit / NAL order, start codes     TEncGOP.cpp:1751-1756, AnnexBwrite.h:55-80, NALwrite.cpp:47-120
//   VPS / SPS / PPS / slice segment header   TEncCavlc.cpp:677-753, 500-675, 189-341, 755-1109 (+ codePTL :1111-1229)
//   slice data: CTU loop, end_of_slice flag  TEncSlice.cpp:985-1170, TEncCu.cpp:290-304, 1112-1128, 1167-1271
//   CU / TU / residual syntax                TEncSbac.cpp:613-1541, TEncEntropy.cpp:200-398
//   arithmetic coder                         TEncBinCoderCABAC.cpp:70-446, TComCABACTables.cpp:43-121, ContextModel.cpp:56-101
// Pinned byte for byte by tests/golden/rd_*.npz:bitstream_nosao (the reference run with --SAO=0) and :bitstream (default run).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "hevcdl.h"
#include "hevcdl_dev.h"

namespace {

enum { PLANAR = 0, DC = 1, HOR = 10, VER = 26, DM_CHROMA = 36, SIZE_2Nx2N = 0, SIZE_NxN = 3 };
enum { SCAN_DIAG = 0, SCAN_HOR = 1, SCAN_VER = 2 };
enum { CTX_SPLIT = 0, CTX_PART_SIZE = 3, CTX_INTRA_PRED = 4, CTX_CHROMA_PRED = 5, CTX_QT_CBF = 6, CTX_SUBDIV = 16, CTX_SIG_CG = 19,
       CTX_SIG = 23, CTX_LAST_X = 67, CTX_LAST_Y = 97, CTX_ONE = 127, CTX_ABS = 151, CTX_TSKIP = 157, CTX_SAO_MERGE = 159, CTX_SAO_TYPE = 160, NUM_CTX = 161 };

// I-slice context initialisation values in the order of the enum above (ContextTables.h:181-480)
const uint8_t CTX_INIT[NUM_CTX] = {
  139, 141, 157, 184, 184, 63,
  111, 141, 154, 154, 154, 94, 138, 182, 154, 154,
  153, 138, 138, 91, 171, 134, 141,
  111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 141,
  140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111, 111,
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
  138, 153, 136, 167, 152, 152, 139, 139,
  153, 200 };                  // sao_merge_flag, sao_type_idx (ContextTables.h:445-458)
const uint8_t NEXT_MPS[128] = {
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127 };
const uint8_t NEXT_LPS[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };
const uint8_t LPS_TABLE[64][4] = {
  {128,176,208,240},{128,167,197,227},{128,158,187,216},{123,150,178,205},{116,142,169,195},{111,135,160,185},{105,128,152,175},{100,122,144,166},
  {95,116,137,158},{90,110,130,150},{85,104,123,142},{81,99,117,135},{77,94,111,128},{73,89,105,122},{69,85,100,116},{66,80,95,110},
  {62,76,90,104},{59,72,86,99},{56,69,81,94},{53,65,77,89},{51,62,73,85},{48,59,69,80},{46,56,66,76},{43,53,63,72},
  {41,50,59,69},{39,48,56,65},{37,45,54,62},{35,43,51,59},{33,41,48,56},{32,39,46,53},{30,37,43,50},{29,35,41,48},
  {27,33,39,45},{26,31,37,43},{24,30,35,41},{23,28,33,39},{22,27,32,37},{21,26,30,35},{20,24,29,33},{19,23,27,31},
  {18,22,26,30},{17,21,25,28},{16,20,23,27},{15,19,22,25},{14,18,21,24},{14,17,20,23},{13,16,19,22},{12,15,18,21},
  {12,14,17,20},{11,14,16,19},{11,13,15,18},{10,12,15,17},{10,12,14,16},{9,11,13,15},{9,11,12,14},{8,10,12,14},
  {8,9,11,13},{7,9,11,12},{7,9,10,12},{7,8,10,11},{6,8,9,11},{6,7,9,10},{6,7,8,9},{2,2,2,2} };
const uint8_t RENORM[32] = { 6, 5, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1 };
const uint8_t GROUP_IDX[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };   // TComRom.cpp:598
const uint8_t MIN_IN_GROUP[10] = { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24 };                                                          // TComRom.cpp:597
const uint8_t CTX_IND_MAP_4x4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };                                         // TComRom.cpp:589-595

// ---- bit writer (TComOutputBitstream) --------------------------------------------------------------------------------
struct BitOut {
  std::vector<uint8_t> b; uint32_t held = 0; int nheld = 0;
  void write(uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) { held = (held << 1) | ((v >> i) & 1u); if (++nheld == 8) { b.push_back((uint8_t)held); held = 0; nheld = 0; } } }
  void flag(int v) { write(v ? 1u : 0u, 1); }
  void ue(uint32_t v) { uint32_t t = v + 1; int len = 0; while ((t >> len) > 1) len++; write(0, len); write(t, len + 1); }
  void se(int v) { ue(v <= 0 ? (uint32_t)(-2 * v) : (uint32_t)(2 * v - 1)); }
  void align_zero() { if (nheld) write(0, 8 - nheld); }
  void trailing() { write(1, 1); align_zero(); }          // rbsp_trailing_bits / byte_alignment()
};

// NAL unit: 2-byte header, emulation prevention (NALwrite.cpp:47-120), Annex B start code (AnnexBwrite.h:55-80)
void put_nal(std::vector<uint8_t> &out, int type, const std::vector<uint8_t> &rbsp, bool long_start_code)
{
  if (long_start_code) out.push_back(0);
  out.push_back(0); out.push_back(0); out.push_back(1);
  out.push_back((uint8_t)(type << 1)); out.push_back(1);          // nuh_layer_id 0, nuh_temporal_id_plus1 1
  int zeros = 0;
  for (uint8_t v : rbsp) {
    if (zeros >= 2 && v <= 3) { out.push_back(3); zeros = 0; }
    out.push_back(v);
    zeros = v == 0 ? zeros + 1 : 0;
  }
  if (!rbsp.empty() && rbsp.back() == 0) out.push_back(3);          // NALwrite.cpp:112-118 (cabac_zero_words guard)
}

// TComOutputBitstream::countStartCodeEmulations TComBitStream.cpp:198-228: emulation prevention bytes this byte string will receive
uint32_t count_emulations(const std::vector<uint8_t> &b)
{
  uint32_t cnt = 0; int zeros = 0;
  for (uint8_t v : b) { if (zeros >= 2 && v <= 3) { cnt++; zeros = 0; } zeros = v == 0 ? zeros + 1 : 0; }
  return cnt;
}

// MD5 (RFC 1321) of a byte string: the decoded picture hash of the reference (libmd5 + TComPicYuvMD5.cpp:88-130) is the plain
// MD5 of each colour plane, rows packed, samples as 1 byte (8-bit) or 2 bytes little endian.
struct Md5 {
  uint32_t h[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u }; uint64_t len = 0; uint8_t buf[64]; int fill = 0;
  static uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
  static const uint32_t *sines()
  { // K[i] = floor(2^32 * |sin(i + 1)|), RFC 1321 section 3.4, computed once
    struct Table { uint32_t k[64]; Table() { for (int i = 0; i < 64; i++) k[i] = (uint32_t)(int64_t)floor(fabs(sin((double)(i + 1))) * 4294967296.0); } };
    static const Table t;                                               // initialised once, thread-safe
    return t.k;
  }
  const uint32_t *K = sines();
  void block(const uint8_t *p)
  {
    uint32_t m[16], a = h[0], b = h[1], c = h[2], d = h[3];
    memcpy(m, p, 64);                                                   // little-endian hosts only, like the rest of the byte packing here
#define MD5_STEP(F, a, b, c, d, g, i, s) a = b + rol(a + F(b, c, d) + K[i] + m[g], s)
#define MD5_F(x, y, z) (z ^ (x & (y ^ z)))
#define MD5_G(x, y, z) (y ^ (z & (x ^ y)))
#define MD5_H(x, y, z) (x ^ y ^ z)
#define MD5_I(x, y, z) (y ^ (x | ~z))
#define MD5_ROUND(F, g0, gs, i0, s0, s1, s2, s3) \
    for (int q = 0; q < 4; q++) { const int i = i0 + 4 * q; \
      MD5_STEP(F, a, b, c, d, (g0 + gs * (i - i0)) & 15, i, s0); MD5_STEP(F, d, a, b, c, (g0 + gs * (i + 1 - i0)) & 15, i + 1, s1); \
      MD5_STEP(F, c, d, a, b, (g0 + gs * (i + 2 - i0)) & 15, i + 2, s2); MD5_STEP(F, b, c, d, a, (g0 + gs * (i + 3 - i0)) & 15, i + 3, s3); }
    MD5_ROUND(MD5_F, 0, 1, 0, 7, 12, 17, 22)
    MD5_ROUND(MD5_G, 1, 5, 16, 5, 9, 14, 20)
    MD5_ROUND(MD5_H, 5, 3, 32, 4, 11, 16, 23)
    MD5_ROUND(MD5_I, 0, 7, 48, 6, 10, 15, 21)
#undef MD5_ROUND
#undef MD5_STEP
#undef MD5_F
#undef MD5_G
#undef MD5_H
#undef MD5_I
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
  }
  void update(const uint8_t *p, size_t n)
  {
    len += n;
    while (n) {
      if (!fill && n >= 64) { block(p); p += 64; n -= 64; continue; }   // whole blocks straight from the caller's memory
      const size_t take = std::min<size_t>(n, 64 - fill); memcpy(buf + fill, p, take); fill += (int)take; p += take; n -= take; if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void final(uint8_t out[16])
  {
    const uint64_t bits = len * 8; const uint8_t pad = 0x80, zero = 0;
    update(&pad, 1); while (fill != 56) update(&zero, 1);
    uint8_t l[8]; for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i));
    update(l, 8);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (8 * j));
  }
};

void profile_tier_level(BitOut &w, int level_idc, int bit_depth)
{ // codePTL / codeProfileTier: Profile main => idc 1, compatibility flags 1 and 2; main10 at 10 bits => idc 2, flag 2 only (TAppEncTop.cpp:120-135)
  w.write(0, 2); w.flag(0); w.write(bit_depth > 8 ? 2 : 1, 5);
  w.write(bit_depth > 8 ? 0x20000000u : 0x60000000u, 32);
  w.flag(0); w.flag(0); w.flag(0); w.flag(0);                       // progressive / interlaced / non-packed / frame-only
  w.write(0, 16); w.write(0, 16); w.write(0, 11); w.flag(0);        // reserved_zero_43bits, inbld_flag
  w.write((uint32_t)level_idc, 8);
}

// ---- arithmetic coder ------------------------------------------------------------------------------------------------
struct Cabac {
  BitOut &w; uint8_t ctx[NUM_CTX];
  uint32_t low = 0, range = 510; int bits_left = 23, buffered = 0; uint32_t buffered_byte = 0xff;
  Cabac(BitOut &o, int qp) : w(o)
  { // ContextModel::init ContextModel.cpp:56-66
    for (int i = 0; i < NUM_CTX; i++) {
      const int v = CTX_INIT[i], slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
      int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
      const int mps = st >= 64;
      ctx[i] = (uint8_t)(((mps ? st - 64 : 63 - st) << 1) + mps);
    }
  }
  void write_out()
  {
    const uint32_t lead = low >> (24 - bits_left);
    bits_left += 8; low &= 0xffffffffu >> bits_left;
    if (lead == 0xff) buffered++;
    else if (buffered > 0) {
      const uint32_t carry = lead >> 8; uint32_t byte = buffered_byte + carry;
      buffered_byte = lead & 0xff; w.write(byte, 8);
      byte = (0xff + carry) & 0xff;
      while (buffered > 1) { w.write(byte, 8); buffered--; }
    } else { buffered = 1; buffered_byte = lead; }
  }
  void test_write() { if (bits_left < 12) write_out(); }
  void bin(int c, int v)
  {
    const int st = ctx[c] >> 1, mps = ctx[c] & 1;
    const uint32_t lps = LPS_TABLE[st][(range >> 6) & 3];
    range -= lps;
    if (v != mps) {
      const int nb = RENORM[lps >> 3];
      low = (low + range) << nb; range = lps << nb; ctx[c] = NEXT_LPS[ctx[c]]; bits_left -= nb; test_write();
    } else {
      ctx[c] = NEXT_MPS[ctx[c]];
      if (range < 256) { low <<= 1; range <<= 1; bits_left--; test_write(); }
    }
  }
  void ep(int v) { low <<= 1; if (v) low += range; bits_left--; test_write(); }
  void eps(uint32_t v, int n)
  {
    while (n > 8) { n -= 8; const uint32_t pat = v >> n; low <<= 8; low += range * pat; v -= pat << n; bits_left -= 8; test_write(); }
    low <<= n; low += range * v; bits_left -= n; test_write();
  }
  void terminate(int v)
  {
    range -= 2;
    if (v) { low += range; low <<= 7; range = 2 << 7; bits_left -= 7; }
    else if (range >= 256) return;
    else { low <<= 1; range <<= 1; bits_left--; }
    test_write();
  }
  void finish()
  {
    if (low >> (32 - bits_left)) {
      w.write(buffered_byte + 1, 8);
      while (buffered > 1) { w.write(0x00, 8); buffered--; }
      low -= 1u << (32 - bits_left);
    } else {
      if (buffered > 0) w.write(buffered_byte, 8);
      while (buffered > 1) { w.write(0xff, 8); buffered--; }
    }
    w.write(low >> 8, 24 - bits_left);
  }
};

// ---- picture-level view of the CTU records -----------------------------------------------------------------------------
struct Pic {
  const hevcdl_ctu_record *recs; int W, H, ctus_x;
  int tx0 = 0, ty0 = 0;                 // top-left luma sample of the tile being written: nothing left of / above it is a neighbour
  uint8_t r2z[256];
  Pic(const hevcdl_ctu_record *r, int w, int h) : recs(r), W(w), H(h), ctus_x((w + 63) >> 6)
  {
    for (int i = 0; i < 256; i++) { const int x = i & 15, y = i >> 4; int z = 0; for (int b = 0; b < 4; b++) z |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1)); r2z[i] = (uint8_t)z; }
  }
  const hevcdl_ctu_record &rec_at(int x4, int y4, int &z) const { z = r2z[((y4 & 15) << 4) | (x4 & 15)]; return recs[(y4 >> 4) * ctus_x + (x4 >> 4)]; }
};
struct Cu { int x, y, log2, depth, zbase, nparts, part; const hevcdl_ctu_record *r; };
struct Tu { int x, y, log2, trd, zrel, nparts; };

int cbf_of(const Cu &cu, int comp, int z) { return cu.r->cbf[comp][z]; }
Tu tu_child(const Tu &p, int i)
{
  Tu c; const int h = 1 << (p.log2 - 1);
  c.log2 = p.log2 - 1; c.trd = p.trd + 1; c.nparts = p.nparts >> 2; c.x = p.x + (i & 1) * h; c.y = p.y + (i >> 1) * h; c.zrel = p.zrel + i * c.nparts;
  return c;
}
int min_tu_log2(const Cu &cu)
{ // getQuadtreeTULog2MinSizeInCU TComDataCU.cpp:1478-1503 (TU log2 2..5, intra TU depth 3)
  const int split = cu.part == SIZE_NxN; int r;
  if (cu.log2 < 2 + 3 - 1 + split) r = 2; else { r = cu.log2 - (3 - 1 + split); if (r > 5) r = 5; }
  return r;
}

struct CParam { int log2, n, ch, scan_type, wg, first_sig_ctx; const uint16_t *scan; const uint8_t *scan_cg; };      // scan / scan_cg: the tables of (scan type, block size), built once (scan_tables)
void scan_next(int type, int bw, int bh, int &line, int &col)
{ // ScanGenerator::GetNextIndex TComRom.cpp:100-160
  if (type == SCAN_DIAG) {
    if (col == bw - 1 || line == 0) { line += col + 1; col = 0; if (line >= bh) { col += line - (bh - 1); line = bh - 1; } }
    else { col++; line--; }
  } else if (type == SCAN_HOR) { if (col == bw - 1) { line++; col = 0; } else col++; }
  else { if (line == bh - 1) { col++; line = 0; } else line++; }
}
// the grouped 4x4 scans of every (scan type, block size) (TComRom.cpp:179-260): built once, when the library is loaded -- a transform block's coding used to build its own
// (1024 positions for a 32x32 block): 86 % of the writer's time on a 1080p picture
struct ScanTables {
  uint16_t scan[3][4][1024]; uint8_t scan_cg[3][4][64];
  ScanTables()
  {
    for (int type = 0; type < 3; type++) for (int l2 = 0; l2 < 4; l2++) {
      const int n = 4 << l2, wg = n >> 2;
      int gl = 0, gc = 0;
      for (int g = 0; g < wg * wg; g++) {
        scan_cg[type][l2][g] = (uint8_t)(gl * wg + gc);
        int l = 0, col = 0;
        for (int q = 0; q < 16; q++) { scan[type][l2][g * 16 + q] = (uint16_t)((l + gl * 4) * n + col + gc * 4); scan_next(type, 4, 4, l, col); }
        scan_next(type, wg, wg, gl, gc);
      }
    }
  }
};
const ScanTables g_scan_tables;
static_assert(SCAN_DIAG >= 0 && SCAN_DIAG < 3 && SCAN_HOR >= 0 && SCAN_HOR < 3 && SCAN_VER >= 0 && SCAN_VER < 3, "scan types index the tables");
void get_cparam(CParam &cp, int c, int n, int dir_mode)
{ // TComDataCU.cpp:3150-3209 (scan choice), TComChromaFormat.cpp:96-160
  cp.n = n; cp.log2 = n == 4 ? 2 : (n == 8 ? 3 : (n == 16 ? 4 : 5)); cp.ch = c ? 1 : 0; cp.wg = n >> 2;
  cp.scan_type = SCAN_DIAG;
  if (n <= (c ? 4 : 8)) { if (abs(dir_mode - VER) <= 4) cp.scan_type = SCAN_HOR; else if (abs(dir_mode - HOR) <= 4) cp.scan_type = SCAN_VER; }
  if (n == 4) cp.first_sig_ctx = 0;
  else if (n == 8) cp.first_sig_ctx = 9 + ((cp.scan_type != SCAN_DIAG) ? (cp.ch ? 0 : 6) : 0);
  else cp.first_sig_ctx = cp.ch ? 12 : 21;
  cp.scan = g_scan_tables.scan[cp.scan_type][cp.log2 - 2]; cp.scan_cg = g_scan_tables.scan_cg[cp.scan_type][cp.log2 - 2];
}
int sig_ctx_inc(const CParam &cp, int pat, int scan_pos)
{ // TComTrQuant.cpp:2707-2803
  const int raster = cp.scan[scan_pos], py = raster >> cp.log2, px = raster - (py << cp.log2);
  if (px + py == 0) return 0;
  int offset;
  if (cp.log2 == 2) offset = CTX_IND_MAP_4x4[4 * py + px];
  else {
    int cnt; const int xs = px & 3, ys = py & 3;
    if (pat == 0) cnt = (xs + ys >= 3) ? 0 : ((xs + ys >= 1) ? 1 : 2);
    else if (pat == 1) cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2);
    else if (pat == 2) cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2);
    else cnt = 2;
    offset = ((((px >> 2) + (py >> 2)) > 0) ? (cp.ch ? 0 : 3) : 0) + cnt;
  }
  return cp.first_sig_ctx + offset;
}

// the tool switches of the access unit being written (hevcdl_stream_config.tools): transform_skip_enabled_flag and sign_data_hiding_enabled_flag change residual_coding()
thread_local uint32_t t_tools = HEVCDL_TOOLS_REFERENCE;
// residual_coding(): TEncSbac::codeCoeffNxN TEncSbac.cpp:1115-1541
void code_coeff(Cabac &c, const int16_t *coef, int comp, int n, int dir_mode, int tskip_flag)
{
  const int ch = comp ? 1 : 0;
  CParam cp; get_cparam(cp, comp, n, dir_mode);
  const int log2n = cp.log2;
  int num_sig = 0;
  for (int i = 0; i < n * n; i++) num_sig += coef[i] != 0;
  if (num_sig == 0) return;
  if (n == 4 && (t_tools & HEVCDL_TOOL_TSKIP)) c.bin(CTX_TSKIP + ch, tskip_flag);      // transform_skip_flag only with transform_skip_enabled_flag (TEncSbac.cpp:1007)
  uint8_t cgf[64]; memset(cgf, 0, sizeof cgf);
  int scan_last = -1, pos_last;
  do {
    pos_last = cp.scan[++scan_last];
    if (coef[pos_last] != 0) { const int py = pos_last >> log2n, px = pos_last - (py << log2n); cgf[cp.wg * (py >> 2) + (px >> 2)] = 1; num_sig--; }
  } while (num_sig > 0);
  { // codeLastSignificantXY :1115-1181
    int py = pos_last >> log2n, px = pos_last - (py << log2n);
    if (cp.scan_type == SCAN_VER) { const int t = px; px = py; py = t; }
    const int gx = GROUP_IDX[px], gy = GROUP_IDX[py], gmax = GROUP_IDX[n - 1], cw = log2n - 2;
    const int off = ch ? 0 : (cw * 3 + ((cw + 1) >> 2)), shift = ch ? cw : ((cw + 3) >> 2);     // TComChromaFormat.h:211-226
    const int bx = CTX_LAST_X + (ch ? 15 : 0) + off, by = CTX_LAST_Y + (ch ? 15 : 0) + off;
    int k;
    for (k = 0; k < gx; k++) c.bin(bx + (k >> shift), 1);
    if (gx < gmax) c.bin(bx + (k >> shift), 0);
    for (k = 0; k < gy; k++) c.bin(by + (k >> shift), 1);
    if (gy < gmax) c.bin(by + (k >> shift), 0);
    if (gx > 3) { const int cnt = (gx - 2) >> 1, v = px - MIN_IN_GROUP[gx]; for (int i = cnt - 1; i >= 0; i--) c.ep((v >> i) & 1); }
    if (gy > 3) { const int cnt = (gy - 2) >> 1, v = py - MIN_IN_GROUP[gy]; for (int i = cnt - 1; i >= 0; i--) c.ep((v >> i) & 1); }
  }
  const int cg_off = CTX_SIG_CG + (ch ? 2 : 0), sig_off = CTX_SIG + (ch ? 28 : 0);
  const int last_set = scan_last >> 4;
  int c1 = 1, sp = scan_last;
  for (int subset = last_set; subset >= 0; subset--) {
    int num_nz = 0, go_rice = 0; const int sub_pos = subset << 4;
    int abs_coeff[16], last_nz = -1, first_nz = 16; uint32_t signs = 0;
    if (sp == scan_last) { abs_coeff[0] = abs(coef[pos_last]); num_nz = 1; last_nz = sp; first_nz = sp; signs = coef[pos_last] < 0; sp--; }
    const int cgblk = cp.scan_cg[subset], gy = cgblk / cp.wg, gx = cgblk - gy * cp.wg;
    const int right = (gx < cp.wg - 1) ? (cgf[gy * cp.wg + gx + 1] != 0) : 0, lower = (gy < cp.wg - 1) ? (cgf[(gy + 1) * cp.wg + gx] != 0) : 0;
    if (subset == last_set || subset == 0) cgf[cgblk] = 1;
    else c.bin(cg_off + ((right + lower) != 0), cgf[cgblk] != 0);
    if (cgf[cgblk]) {
      const int pat = cp.wg <= 1 ? 0 : right + (lower << 1);
      for (; sp >= sub_pos; sp--) {
        const int blk = cp.scan[sp], sig = coef[blk] != 0;
        if (sp > sub_pos || subset == 0 || num_nz) c.bin(sig_off + sig_ctx_inc(cp, pat, sp), sig);
        if (sig) { abs_coeff[num_nz++] = abs(coef[blk]); signs = 2 * signs + (coef[blk] < 0); if (last_nz == -1) last_nz = sp; first_nz = sp; }
      }
    } else sp = sub_pos - 1;
    if (num_nz > 0) {
      const int sign_hidden = (t_tools & HEVCDL_TOOL_SIGN_HIDE) && (last_nz - first_nz >= 4);
      const int cset = (ch ? 4 : 0) + ((!ch && subset > 0) ? 2 : 0) + (c1 == 0 ? 1 : 0);      // TComChromaFormat.h:243-251
      c1 = 1;
      const int n_c1 = num_nz < 8 ? num_nz : 8; int first_c2 = -1, escape = 0;
      for (int i = 0; i < n_c1; i++) {
        const int sym = abs_coeff[i] > 1;
        c.bin(CTX_ONE + 4 * cset + c1, sym);
        if (sym) { c1 = 0; if (first_c2 == -1) first_c2 = i; else escape = 1; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && first_c2 != -1) { const int sym = abs_coeff[first_c2] > 2; c.bin(CTX_ABS + cset, sym); if (sym) escape = 1; }
      escape = escape || (num_nz > 8);
      if (sign_hidden) c.eps(signs >> 1, num_nz - 1); else c.eps(signs, num_nz);
      int first_coeff2 = 1;
      if (escape) for (int i = 0; i < num_nz; i++) {
        const int base = (i < 8) ? (2 + first_coeff2) : 1;
        if (abs_coeff[i] >= base) { // xWriteCoefRemainExGolomb :337-394
          int code = abs_coeff[i] - base;
          if (code < (3 << go_rice)) { const int len = code >> go_rice; c.eps((1u << (len + 1)) - 2, len + 1); c.eps((uint32_t)(code % (1 << go_rice)), go_rice); }
          else {
            int len = go_rice; code -= 3 << go_rice;
            while (code >= (1 << len)) code -= 1 << (len++);
            c.eps((1u << (3 + len + 1 - go_rice)) - 2, 3 + len + 1 - go_rice); c.eps((uint32_t)code, len);
          }
          if (abs_coeff[i] > (3 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
        }
        if (abs_coeff[i] >= 2) first_coeff2 = 0;
      }
    }
  }
}

int luma_mode_at(const Pic &p, int x4, int y4) { int z; const hevcdl_ctu_record &r = p.rec_at(x4, y4, z); return r.luma_dir[z]; }
int depth_at(const Pic &p, int x4, int y4) { int z; const hevcdl_ctu_record &r = p.rec_at(x4, y4, z); return r.depth[z]; }

void code_luma_dirs(Cabac &c, const Pic &p, const Cu &cu, int npu)
{ // codeIntraDirLumaAng TEncSbac.cpp:643-696, getIntraDirPredictor TComDataCU.cpp:1362-1445
  int preds[4][3], idx[4], dir[4];
  const int pu_size = (cu.part == SIZE_NxN) ? (1 << (cu.log2 - 1)) : (1 << cu.log2);
  for (int j = 0; j < npu; j++) {
    const int px = cu.x + (j & 1) * pu_size, py = cu.y + (j >> 1) * pu_size;
    dir[j] = cu.r->luma_dir[cu.zbase + j * (cu.nparts >> 2) * (cu.part == SIZE_NxN)];
    int left = DC, above = DC;
    if (px > p.tx0) left = luma_mode_at(p, (px >> 2) - 1, py >> 2);
    if ((py & 63) != 0) above = luma_mode_at(p, px >> 2, (py >> 2) - 1);
    if (left == above) {
      if (left > 1) { preds[j][0] = left; preds[j][1] = ((left + 29) % 32) + 2; preds[j][2] = ((left - 1) % 32) + 2; }
      else { preds[j][0] = PLANAR; preds[j][1] = DC; preds[j][2] = VER; }
    } else {
      preds[j][0] = left; preds[j][1] = above;
      preds[j][2] = (left && above) ? PLANAR : ((left + above) < 2 ? VER : DC);
    }
    idx[j] = -1;
    for (int i = 0; i < 3; i++) if (dir[j] == preds[j][i]) idx[j] = i;
    c.bin(CTX_INTRA_PRED, idx[j] != -1);
  }
  for (int j = 0; j < npu; j++) {
    if (idx[j] != -1) { c.ep(idx[j] ? 1 : 0); if (idx[j]) c.ep(idx[j] - 1); }
    else {
      int *q = preds[j], t;
      if (q[0] > q[1]) { t = q[0]; q[0] = q[1]; q[1] = t; }
      if (q[0] > q[2]) { t = q[0]; q[0] = q[2]; q[2] = t; }
      if (q[1] > q[2]) { t = q[1]; q[1] = q[2]; q[2] = t; }
      int d = dir[j];
      for (int i = 2; i >= 0; i--) d = d > q[i] ? d - 1 : d;
      c.eps((uint32_t)d, 5);
    }
  }
}
void code_chroma_dir(Cabac &c, const Cu &cu)
{ // codeIntraDirChroma TEncSbac.cpp:698-726, getAllowedChromaDir TComDataCU.cpp:1334-1353
  const int d = cu.r->chroma_dir[cu.zbase];
  if (d == DM_CHROMA) { c.bin(CTX_CHROMA_PRED, 0); return; }
  c.bin(CTX_CHROMA_PRED, 1);
  const int luma = cu.r->luma_dir[cu.zbase]; int list[4] = { PLANAR, VER, HOR, DC }, k = 0;
  for (int i = 0; i < 4; i++) if (list[i] == luma) { list[i] = 34; break; }
  for (int i = 0; i < 4; i++) if (d == list[i]) { k = i; break; }
  c.eps((uint32_t)k, 2);
}
void code_qt_cbf(Cabac &c, const Cu &cu, const Tu &tu, int comp, int lowest)
{ // codeQtCbf TEncSbac.cpp:920-995
  const int ctx = comp ? tu.trd : (tu.trd == 0 ? 1 : 0);
  const int w = comp ? (tu.log2 > 2 ? 1 << (tu.log2 - 1) : 4) : (1 << tu.log2);
  const int d = tu.trd + ((!lowest && !(w >= 8)) ? 1 : 0);
  const int z = cu.zbase + (comp ? (tu.log2 > 2 ? tu.zrel : (tu.zrel & ~3)) : tu.zrel);
  c.bin(CTX_QT_CBF + (comp ? 5 : 0) + ctx, (cbf_of(cu, comp, z) >> d) & 1);
}
void code_transform(Cabac &c, const Cu &cu, const Tu &tu)
{ // TEncEntropy::xEncodeTransform TEncEntropy.cpp:200-398 (chroma of a 4x4 quad with its last block)
  const int z = cu.zbase + tu.zrel;
  const int subdiv = cu.r->tr_idx[z] > tu.trd;
  if (cu.part == SIZE_NxN && tu.trd == 0) { }
  else if (tu.log2 > 5) { }
  else if (tu.log2 == 2) { }
  else if (tu.log2 == min_tu_log2(cu)) { }
  else c.bin(CTX_SUBDIV + 5 - tu.log2, subdiv);
  const int first = tu.trd == 0;
  for (int comp = 1; comp < 3; comp++)
    if (first || tu.log2 > 2)
      if (first || ((cbf_of(cu, comp, z) >> (tu.trd - 1)) & 1)) code_qt_cbf(c, cu, tu, comp, !subdiv);
  if (subdiv) { for (int i = 0; i < 4; i++) code_transform(c, cu, tu_child(tu, i)); return; }
  code_qt_cbf(c, cu, tu, 0, 1);
  for (int comp = 0; comp < 3; comp++) {
    if (comp && !(tu.log2 > 2 || (tu.zrel & 3) == 3)) continue;
    if (!((cbf_of(cu, comp, z) >> tu.trd) & 1)) continue;
    const int zc = comp ? (tu.log2 > 2 ? tu.zrel : (tu.zrel & ~3)) : tu.zrel, zabs = cu.zbase + zc;
    const int n = comp ? (tu.log2 > 2 ? 1 << (tu.log2 - 1) : 4) : (1 << tu.log2);
    int mode;
    if (!comp) mode = cu.r->luma_dir[zabs];
    else { const int m = cu.r->chroma_dir[zabs]; mode = m == DM_CHROMA ? cu.r->luma_dir[cu.zbase + (zc & ~3)] : m; }
    const int16_t *coef = comp == 0 ? cu.r->coeff_y + zabs * 16 : (comp == 1 ? cu.r->coeff_cb : cu.r->coeff_cr) + zabs * 4;
    code_coeff(c, coef, comp, n, mode, cu.r->tskip[comp][zabs]);
  }
}
// sao(): TEncSbac::codeSAOBlkParam / codeSAOOffsetParam TEncSbac.cpp:1543-1720, called before every CTU (TEncSlice.cpp:1075-1111)
void code_sao_offset(Cabac &c, int comp, const hevcdl_sao_offset &p, int max_off)
{
  const int first = comp != 2;
  if (first) {
    if (p.mode == 0) c.bin(CTX_SAO_TYPE, 0);
    else { c.bin(CTX_SAO_TYPE, 1); c.ep(p.type == 4 ? 0 : 1); }
  }
  if (p.mode != 1) return;
  int off[4], k = 0;
  const int ncls = p.type == 4 ? 4 : 5;
  for (int i = 0; i < ncls; i++) { if (p.type != 4 && i == 2) continue; off[k++] = p.offset[p.type == 4 ? (p.aux + i) % 32 : i]; }
  for (int i = 0; i < 4; i++) { // codeSaoMaxUvlc, maximum (1 << (min(bitDepth, 10) - 5)) - 1: 7 at 8 bits, 31 at 10
    const int a = abs(off[i]);
    if (a == 0) c.ep(0);
    else { c.ep(1); for (int j = 0; j < a - 1; j++) c.ep(1); if (a < max_off) c.ep(0); }
  }
  if (p.type == 4) { for (int i = 0; i < 4; i++) if (off[i]) c.ep(off[i] < 0); c.eps((uint32_t)p.aux, 5); }
  else if (first) c.eps((uint32_t)p.type, 2);
}
void code_sao_blk(Cabac &c, const hevcdl_sao_blk &b, bool left_avail, bool above_avail, int max_off)
{
  bool is_left = false, is_above = false;
  if (left_avail) { is_left = b.c[0].mode == 2 && b.c[0].type == 0; c.bin(CTX_SAO_MERGE, is_left); }
  if (above_avail && !is_left) { is_above = b.c[0].mode == 2 && b.c[0].type == 1; c.bin(CTX_SAO_MERGE, is_above); }
  if (!is_left && !is_above) for (int comp = 0; comp < 3; comp++) code_sao_offset(c, comp, b.c[comp], max_off);
}

void code_cu_tree(Cabac &c, const Pic &p, int x, int y, int depth)
{ // xEncodeCU TEncCu.cpp:1167-1271 (I slice: no skip / pred-mode flags)
  const int size = 64 >> depth;
  int z; const hevcdl_ctu_record &r = p.rec_at(x >> 2, y >> 2, z);
  int boundary = 0;
  if (x + size <= p.W && y + size <= p.H) {
    if (depth < 3) {
      int sctx = 0;
      if (x > p.tx0) sctx += depth_at(p, (x >> 2) - 1, y >> 2) > depth;
      if (y > p.ty0) sctx += depth_at(p, x >> 2, (y >> 2) - 1) > depth;
      c.bin(CTX_SPLIT + sctx, r.depth[z] > depth);
    }
  } else boundary = 1;
  if ((depth < r.depth[z] && depth < 3) || boundary) {
    const int h = size >> 1;
    for (int i = 0; i < 4; i++) { const int sx = x + (i & 1) * h, sy = y + (i >> 1) * h; if (sx < p.W && sy < p.H) code_cu_tree(c, p, sx, sy, depth + 1); }
    return;
  }
  const Cu cu = { x, y, 6 - depth, depth, z, 256 >> (2 * depth), r.part_size[z], &r };
  if (depth == 3) c.bin(CTX_PART_SIZE, cu.part == SIZE_2Nx2N);
  code_luma_dirs(c, p, cu, cu.part == SIZE_NxN ? 4 : 1);
  code_chroma_dir(c, cu);
  const Tu root = { x, y, cu.log2, 0, 0, cu.nparts };
  code_transform(c, cu, root);
}

} // namespace

extern "C" hevcdl_status hevcdl_stream_config_default(hevcdl_stream_config *cfg, int width, int height, int qp)
{
  if (!cfg || width <= 0 || height <= 0 || (width & 7) || (height & 7) || qp < 0 || qp > 51) return HEVCDL_ERR_INVALID_ARG;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = sizeof *cfg; cfg->width = width; cfg->height = height; cfg->qp = qp;
  cfg->level_idc = 186;                    // Level 6.2 (general_level_idc = 30 * level)
  cfg->tile_columns = 1; cfg->tile_rows = 1; cfg->bit_depth = 8; cfg->tile_uniform_spacing = 1; cfg->lf_across_tiles = 1; cfg->wavefront = 0;
  cfg->tools = HEVCDL_TOOLS_REFERENCE; cfg->rewrite_param_sets = 1;
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_picture_md5(const hevcdl_stream_config *cfg, const void *picture, uint8_t digest[48])
{ // calcMD5 TComPicYuvMD5.cpp:88-130: one digest per colour plane
  if (!cfg || cfg->struct_size != sizeof *cfg || !picture || !digest) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7)) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->bit_depth != 8 && cfg->bit_depth != 10) return HEVCDL_ERR_UNSUPPORTED;
  const size_t bps = cfg->bit_depth > 8 ? 2 : 1, ysz = (size_t)cfg->width * cfg->height * bps, csz = ysz / 4;
  const uint8_t *p = (const uint8_t *)picture;
  for (int c = 0; c < 3; c++) { Md5 m; m.update(p + (c == 0 ? 0 : (c == 1 ? ysz : ysz + csz)), c ? csz : ysz); m.final(digest + 16 * c); }
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_write_digest_sei(const uint8_t digest[48], uint8_t *out, size_t capacity, size_t *out_len)
{ // SEIDecodedPictureHash 1 (MD5): suffix SEI NAL after the slice (TEncGOP.cpp:1938-1960, SEIwrite.cpp xWriteSEIDecodedPictureHash)
  if (!digest || !out || !out_len) return HEVCDL_ERR_INVALID_ARG;
  BitOut w;
  w.write(132, 8); w.write(49, 8); w.write(0, 8);          // payload type decoded_picture_hash, size 1 + 3 * 16, hash_type 0 = MD5
  for (int i = 0; i < 48; i++) w.write(digest[i], 8);
  w.trailing();
  std::vector<uint8_t> nal;
  put_nal(nal, 40, w.b, false);                           // SUFFIX_SEI_NUT
  *out_len = nal.size();
  if (nal.size() > capacity) return HEVCDL_ERR_INVALID_ARG;
  memcpy(out, nal.data(), nal.size());
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_picture_hash(const hevcdl_stream_config *cfg, const void *picture, int method, uint8_t digest[48], int *plane_bytes)
{ // calcMD5 / calcCRC / calcChecksum TComPicYuvMD5.cpp:88-180: one digest per colour plane
  if (method == 1) { if (plane_bytes) *plane_bytes = 16; return hevcdl_picture_md5(cfg, picture, digest); }
  if (!cfg || cfg->struct_size != sizeof *cfg || !picture || !digest) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7)) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->bit_depth != 8 && cfg->bit_depth != 10) return HEVCDL_ERR_UNSUPPORTED;
  if (method != 2 && method != 3) return HEVCDL_ERR_UNSUPPORTED;
  const int wide = cfg->bit_depth > 8;
  const size_t ysz = (size_t)cfg->width * cfg->height;
  for (int c = 0; c < 3; c++) {
    const int pw = c ? cfg->width / 2 : cfg->width, ph = c ? cfg->height / 2 : cfg->height;
    const size_t off = c == 0 ? 0 : (c == 1 ? ysz : ysz + ysz / 4);
    auto sample = [&](int x, int y) -> unsigned { const size_t i = off + (size_t)y * pw + x; return wide ? ((const uint16_t *)picture)[i] : ((const uint8_t *)picture)[i]; };
    if (method == 2) { // CRC-16-CCITT over the bytes of the samples, most significant bit first, low byte first (:89-127)
      unsigned crc = 0xffff;
      auto feed = [&](unsigned byte) { for (int b = 0; b < 8; b++) { const unsigned msb = (crc >> 15) & 1, bit = (byte >> (7 - b)) & 1; crc = (((crc << 1) + bit) & 0xffff) ^ (msb * 0x1021); } };
      for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) { const unsigned v = sample(x, y); feed(v & 0xff); if (wide) feed(v >> 8); }
      for (int b = 0; b < 16; b++) { const unsigned msb = (crc >> 15) & 1; crc = ((crc << 1) & 0xffff) ^ (msb * 0x1021); }
      digest[2 * c] = (uint8_t)(crc >> 8); digest[2 * c + 1] = (uint8_t)crc;
    } else {           // checksum: sum of the bytes, each xor-ed with a mask of its position (:141-165)
      uint32_t sum = 0;
      for (int y = 0; y < ph; y++) for (int x = 0; x < pw; x++) {
        const unsigned v = sample(x, y), mask = ((unsigned)x & 0xff) ^ ((unsigned)y & 0xff) ^ ((unsigned)x >> 8) ^ ((unsigned)y >> 8);
        sum += ((v & 0xff) ^ (mask & 0xff));
        if (wide) sum += ((v >> 8) ^ (mask & 0xff));
      }
      for (int k = 0; k < 4; k++) digest[4 * c + k] = (uint8_t)(sum >> (24 - 8 * k));
    }
  }
  if (plane_bytes) *plane_bytes = method == 2 ? 2 : 4;
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_write_hash_sei(int method, const uint8_t *digest, uint8_t *out, size_t capacity, size_t *out_len)
{ // SEIwrite.cpp xWriteSEIDecodedPictureHash: hash_type u(8) = method - 1, then per plane 16 bytes (MD5) / u(16) (CRC) / u(32) (checksum)
  if (method == 1) return hevcdl_write_digest_sei(digest, out, capacity, out_len);
  if (!digest || !out || !out_len || (method != 2 && method != 3)) return HEVCDL_ERR_INVALID_ARG;
  const int pb = method == 2 ? 2 : 4;
  BitOut w;
  w.write(132, 8); w.write(1 + 3 * pb, 8); w.write((uint32_t)(method - 1), 8);
  for (int i = 0; i < 3 * pb; i++) w.write(digest[i], 8);
  w.trailing();
  std::vector<uint8_t> nal;
  put_nal(nal, 40, w.b, false);
  *out_len = nal.size();
  if (nal.size() > capacity) return HEVCDL_ERR_INVALID_ARG;
  memcpy(out, nal.data(), nal.size());
  return HEVCDL_OK;
}

extern "C" hevcdl_status hevcdl_write_picture_hash_sei(const hevcdl_stream_config *cfg, const void *picture, uint8_t *out, size_t capacity, size_t *out_len)
{ // the digests of `picture`, then the SEI above
  if (!cfg || cfg->struct_size != sizeof *cfg || !picture || !out || !out_len) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7)) return HEVCDL_ERR_INVALID_ARG;
  uint8_t d[48];
  const hevcdl_status st = hevcdl_picture_md5(cfg, picture, d);
  if (st != HEVCDL_OK) return st;
  return hevcdl_write_digest_sei(d, out, capacity, out_len);
}

extern "C" size_t hevcdl_access_unit_bound(int width, int height)
{ // worst case is far below the raw picture size x 2 (the arithmetic coder cannot expand 16-bit levels by more than that)
  return (size_t)width * (size_t)height * 3 + 4096;
}

extern "C" hevcdl_status hevcdl_write_access_unit(const hevcdl_stream_config *cfg, int poc, const hevcdl_ctu_record *records, const hevcdl_sao_blk *sao,
                                                 uint8_t *out, size_t capacity, size_t *out_len)
{
  if (!cfg || cfg->struct_size != sizeof *cfg || !records || !out || !out_len || poc < 0) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7) || cfg->qp < 0 || cfg->qp > 51) return HEVCDL_ERR_INVALID_ARG;
  if (cfg->lf_beta_offset_div2 < -6 || cfg->lf_beta_offset_div2 > 6 || cfg->lf_tc_offset_div2 < -6 || cfg->lf_tc_offset_div2 > 6) return HEVCDL_ERR_INVALID_ARG;
  // deblocking control as TEncTop.cpp:1007-1035 sets it with LoopFilterOffsetInPPS 1 (the cfg's value): no override in the slice header; the PPS carries the disabled flag
  // and the offsets, and the control fields are present when any of them differs from the inferred values
  const int dbk_off = cfg->loop_filter_disable != 0;
  const int dbk_beta = dbk_off ? 0 : cfg->lf_beta_offset_div2, dbk_tc = dbk_off ? 0 : cfg->lf_tc_offset_div2;
  const int dbk_control = dbk_off || dbk_beta != 0 || dbk_tc != 0;
  if ((cfg->sao_enabled != 0) != (sao != nullptr)) return HEVCDL_ERR_INVALID_ARG;         // SAO parameters go with sample_adaptive_offset_enabled_flag
  const int bd = cfg->bit_depth;
  if (bd != 8 && bd != 10) return HEVCDL_ERR_UNSUPPORTED;
  if (!HEVCDL_TOOLS_SUPPORTED(cfg->tools)) return HEVCDL_ERR_UNSUPPORTED;
  t_tools = cfg->tools;
  const int tcols = cfg->tile_columns, trows = cfg->tile_rows, tiled = tcols * trows > 1;
  if (tcols < 1 || trows < 1 || tcols > 20 || trows > 22) return HEVCDL_ERR_INVALID_ARG;
  const int wpp = cfg->wavefront != 0;
  if (wpp && tiled) return HEVCDL_ERR_UNSUPPORTED;       // the reference refuses the pair outside the high-throughput profile (TAppEncCfg.cpp xCheckParameter)
  int col_bd[21], row_bd[23];
  const int uniform = cfg->tile_uniform_spacing != 0;
  if (hevcdl_tile_bounds((cfg->width + 63) >> 6, tcols, uniform, cfg->tile_column_width, tiled ? 4 : 1, col_bd) ||      // TComPicSym.cpp:380-392
      hevcdl_tile_bounds((cfg->height + 63) >> 6, trows, uniform, cfg->tile_row_height, 1, row_bd)) return HEVCDL_ERR_INVALID_ARG;
  std::vector<uint8_t> au;
  const bool write_ps = poc == 0 || cfg->rewrite_param_sets != 0;                       // TEncGOP.cpp:1751: the first picture, or every IRAP with ReWriteParamSetsFlag
  if (write_ps) { // VPS  TEncCavlc.cpp:677-753
    BitOut w;
    w.write(0, 4); w.flag(1); w.flag(1); w.write(0, 6); w.write(0, 3); w.flag(1); w.write(0xffff, 16);
    profile_tier_level(w, cfg->level_idc, bd);
    w.flag(1); w.ue(0); w.ue(0); w.ue(0);            // sub_layer_ordering_info_present, max_dec_pic_buffering_minus1, num_reorder, max_latency_plus1
    w.write(0, 6); w.ue(0); w.flag(0); w.flag(0);    // vps_max_layer_id, num_layer_sets_minus1, timing_info_present, extension
    w.trailing();
    put_nal(au, 32, w.b, true);
  }
  if (write_ps) { // SPS  TEncCavlc.cpp:500-675
    BitOut w;
    w.write(0, 4); w.write(0, 3); w.flag(1);
    profile_tier_level(w, cfg->level_idc, bd);
    w.ue(0); w.ue(1); w.ue((uint32_t)cfg->width); w.ue((uint32_t)cfg->height);
    w.flag(1); w.ue(0); w.ue(0); w.ue(0); w.ue(0);   // conformance window present with zero offsets (as the reference writes it)
    w.ue((uint32_t)bd - 8); w.ue((uint32_t)bd - 8); w.ue(4);     // bit depths - 8, log2_max_pic_order_cnt_lsb_minus4 (8 bits)
    w.flag(1); w.ue(0); w.ue(0); w.ue(0);
    w.ue(0); w.ue(3); w.ue(0); w.ue(3); w.ue(2); w.ue(2);    // CB 8..64, TB 4..32, TU depth inter/intra 3
    w.flag(0); w.flag(1); w.flag(cfg->sao_enabled); w.flag(0); // scaling list, AMP, SAO, PCM
    w.ue(2);                                         // two (empty) short-term RPS of the all-intra GOP table
    w.ue(0); w.ue(0);
    w.flag(0); w.ue(0); w.ue(0);                     // RPS 1: inter_ref_pic_set_prediction_flag 0, no pictures
    w.flag(0); w.flag(1); w.flag((cfg->tools & HEVCDL_TOOL_STRONG_INTRA) != 0); w.flag(0); w.flag(0);     // long-term, temporal MVP, strong intra smoothing, VUI, extension
    w.trailing();
    put_nal(au, 33, w.b, true);
  }
  if (write_ps) { // PPS  TEncCavlc.cpp:189-341
    BitOut w;
    w.ue(0); w.ue(0); w.flag(0); w.flag(0); w.write(0, 3); w.flag((cfg->tools & HEVCDL_TOOL_SIGN_HIDE) != 0); w.flag(1); w.ue(3); w.ue(3);   // ... sign_data_hiding_enabled_flag, cabac_init_present_flag, ...
    w.se(0); w.flag(0); w.flag((cfg->tools & HEVCDL_TOOL_TSKIP) != 0); w.flag(0);        // init_qp_minus26 0, constrained intra, transform skip, cu_qp_delta
    w.se(0); w.se(0); w.flag(0); w.flag(0); w.flag(0); w.flag(0);      // chroma qp offsets, slice chroma offsets present, weighted (bi)pred, transquant bypass
    w.flag(tiled); w.flag(wpp);                      // tiles_enabled_flag, entropy_coding_sync_enabled_flag (TEncCavlc.cpp:226-227)
    if (tiled) { // :228-246
      w.ue((uint32_t)tcols - 1); w.ue((uint32_t)trows - 1); w.flag(uniform);
      if (!uniform) {
        for (int i = 0; i < tcols - 1; i++) w.ue((uint32_t)(col_bd[i + 1] - col_bd[i]) - 1);    // column_width_minus1
        for (int i = 0; i < trows - 1; i++) w.ue((uint32_t)(row_bd[i + 1] - row_bd[i]) - 1);    // row_height_minus1
      }
      w.flag(cfg->lf_across_tiles != 0);             // loop_filter_across_tiles_enabled_flag
    }
    w.flag(1); w.flag(dbk_control);                  // loop filter across slices, deblocking_filter_control_present  (TEncCavlc.cpp:247-259)
    if (dbk_control) { w.flag(0); w.flag(dbk_off); if (!dbk_off) { w.se(dbk_beta); w.se(dbk_tc); } }      // override_enabled 0, pps_deblocking_filter_disabled_flag, pps_beta / tc_offset_div2
    w.flag(0); w.flag(0); w.ue(0); w.flag(0); w.flag(0);
    w.trailing();
    put_nal(au, 34, w.b, true);
  }
  { // slice segment: header TEncCavlc.cpp:755-1109, data TEncSlice.cpp:985-1170
    BitOut w;
    const int idr = poc == 0;
    w.flag(1); w.flag(0); w.ue(0); w.ue(2);          // first_slice_segment_in_pic, no_output_of_prior_pics, pps id, slice_type I
    if (!idr) { w.write((uint32_t)poc & 255u, 8); w.flag(0); w.flag(0); w.ue(0); w.ue(0); w.flag(1); }   // POC lsb, RPS coded in the header (empty), slice_temporal_mvp_enabled
    if (cfg->sao_enabled) { w.flag(1); w.flag(1); }
    w.se(cfg->qp - 26);
    if (cfg->sao_enabled || !dbk_off) w.flag(1);     // slice_loop_filter_across_slices_enabled_flag: only when an in-loop filter is on (TEncCavlc.cpp:1097-1104)
    // slice data: one sub-stream per tile, tiles in raster order, CTUs in raster order inside a tile (TEncSlice.cpp:1030-1145).  Every
    // sub-stream starts from the slice-start contexts and ends with a terminating 1 bin (end_of_slice_segment_flag of the last CTU /
    // end_of_subset_one_bit of the others), the coder flush and byte_alignment().
    Pic pic(records, cfg->width, cfg->height);
    const int ctus_y = (cfg->height + 63) >> 6, ctus = pic.ctus_x * ctus_y;
    std::vector<BitOut> sub(wpp ? (size_t)ctus_y : (size_t)tcols * trows);
    if (wpp) { // WaveFrontSynchro: one sub-stream per CTU row (TEncSlice.cpp:1047-1145).  A row starts from the slice-start contexts, or from those behind the SECOND CTU of
      // the row above when the picture is at least two CTUs wide (the CTU above and to the right exists), and ends like a tile: terminating 1 bin, flush, byte_alignment()
      uint8_t sync[NUM_CTX];
      pic.tx0 = 0; pic.ty0 = 0;
      for (int cy = 0; cy < ctus_y; cy++) {
        BitOut &sw = sub[(size_t)cy];
        Cabac c(sw, cfg->qp);
        if (cy > 0 && pic.ctus_x > 1) memcpy(c.ctx, sync, sizeof sync);
        for (int cx = 0; cx < pic.ctus_x; cx++) {
          const int a = cy * pic.ctus_x + cx;
          if (sao) code_sao_blk(c, sao[a], cx > 0, cy > 0, (1 << ((bd < 10 ? bd : 10) - 5)) - 1);
          code_cu_tree(c, pic, cx * 64, cy * 64, 0);
          if (a != ctus - 1) c.terminate(0);
          if (cx == 1) memcpy(sync, c.ctx, sizeof sync);       // :1127-1130
        }
        c.terminate(1);
        c.finish();
        sw.trailing();
      }
    }
    else for (int tr = 0; tr < trows; tr++) for (int tc = 0; tc < tcols; tc++) {
      const int cx0 = col_bd[tc], cx1 = col_bd[tc + 1], cy0 = row_bd[tr], cy1 = row_bd[tr + 1];
      BitOut &sw = sub[(size_t)tr * tcols + tc];
      Cabac c(sw, cfg->qp);
      pic.tx0 = cx0 * 64; pic.ty0 = cy0 * 64;
      for (int cy = cy0; cy < cy1; cy++) for (int cx = cx0; cx < cx1; cx++) {
        const int a = cy * pic.ctus_x + cx;
        if (sao) code_sao_blk(c, sao[a], cx > cx0, cy > cy0, (1 << ((bd < 10 ? bd : 10) - 5)) - 1);      // merge candidates stay inside the tile (TComPic::getSAOMergeAvailability)
        code_cu_tree(c, pic, cx * 64, cy * 64, 0);
        if (a != ctus - 1) c.terminate(0);           // end_of_slice_segment_flag 0 (finishCU TEncCu.cpp:1112-1128)
      }
      c.terminate(1);                                // TEncSlice.cpp:1136
      c.finish();
      sw.trailing();
    }
    if (tiled || wpp) { // entry points: TEncCavlc::codeTilesWPPEntryPoint :1207-1240; a size counts the emulation prevention bytes of its sub-stream
      std::vector<uint32_t> size(sub.size() - 1);
      uint32_t max_size = 0;
      for (size_t i = 0; i + 1 < sub.size(); i++) { size[i] = (uint32_t)sub[i].b.size() + count_emulations(sub[i].b); if (size[i] > max_size) max_size = size[i]; }
      int len_m1 = 0; while (max_size >= (1u << (len_m1 + 1))) len_m1++;
      w.ue((uint32_t)size.size());
      if (!size.empty()) { w.ue((uint32_t)len_m1); for (uint32_t v : size) w.write(v - 1, len_m1 + 1); }
    }
    w.trailing();                                    // byte_alignment()
    for (const BitOut &sw : sub) w.b.insert(w.b.end(), sw.b.begin(), sw.b.end());
    put_nal(au, idr ? 19 : 21, w.b, !write_ps);      // IDR_W_RADL, then CRA (DecodingRefreshType 1); the first NAL unit of an access unit carries the zero_byte
  }
  *out_len = au.size();
  if (au.size() > capacity) return HEVCDL_ERR_INVALID_ARG;
  memcpy(out, au.data(), au.size());
  return HEVCDL_OK;
}
