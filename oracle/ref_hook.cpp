// oracle/ref_hook.cpp -- TEST INFRASTRUCTURE (not product code).
//
// Link-time observer for the *unmodified* reference encoder objects built by
// oracle/build_ref.sh from /root/reference/HM_dl/source.  GNU ld's
// --wrap=<symbol> redirects the single cross-object call of
//   Void TEncCu::compressCtu(Int m_iFrame, TComDataCU* pCtu)
//   (declared TLibEncoder/TEncCu.h:120, defined TEncCu.cpp:234, called TEncSlice.cpp:879)
// to __wrap_<symbol> below, which calls the real function and then serialises
// what the reference left in the picture's CTU record (TEncCu.cpp:1091 copyToPic)
// and in the reconstruction picture (TEncCu.cpp:1093), i.e. exactly the outputs
// of the drop-in boundary.  No reference source is patched
// or copied: this file only *reads* public accessors of TComDataCU / TComPicYuv.
//
// Output: appended to the file named by $HEVCDL_DUMP, one block per CTU:
//   int32 frame, int32 ctuRsAddr,
//   hevcdl_ctu_record (include/hevcdl.h layout, 15120 bytes),
//   uint8 reconY[64*64], reconCb[32*32], reconCr[32*32]  (zeros outside the picture); uint16 samples instead when
//   $HEVCDL_DUMP16 is set (runs with InternalBitDepth > 8)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComPicYuv.h"
#include "TLibEncoder/TEncCu.h"

extern "C" void __real__ZN6TEncCu11compressCtuEiP10TComDataCU(TEncCu*, int, TComDataCU*);

static FILE* g_dump = NULL;

extern "C" void __wrap__ZN6TEncCu11compressCtuEiP10TComDataCU(TEncCu* self, int frame, TComDataCU* ctu)
{
  __real__ZN6TEncCu11compressCtuEiP10TComDataCU(self, frame, ctu);
  if (!g_dump)
  {
    const char* fn = getenv("HEVCDL_DUMP");
    if (!fn) return;
    g_dump = fopen(fn, "ab");
    if (!g_dump) return;
  }
  int32_t hdr[2] = { frame, (int32_t)ctu->getCtuRsAddr() };
  fwrite(hdr, 4, 2, g_dump);
  const int np = 256;
  uint8_t a[11 * 256];
  for (int i = 0; i < np; i++)
  {
    a[0 * 256 + i] = ctu->getDepth(i);
    a[1 * 256 + i] = (uint8_t)ctu->getPartitionSize(i);
    a[2 * 256 + i] = ctu->getIntraDir(CHANNEL_TYPE_LUMA, i);
    a[3 * 256 + i] = ctu->getIntraDir(CHANNEL_TYPE_CHROMA, i);
    a[4 * 256 + i] = ctu->getTransformIdx(i);
    a[5 * 256 + i] = ctu->getCbf(i, COMPONENT_Y);
    a[6 * 256 + i] = ctu->getCbf(i, COMPONENT_Cb);
    a[7 * 256 + i] = ctu->getCbf(i, COMPONENT_Cr);
    a[8 * 256 + i] = ctu->getTransformSkip(i, COMPONENT_Y);
    a[9 * 256 + i] = ctu->getTransformSkip(i, COMPONENT_Cb);
    a[10 * 256 + i] = ctu->getTransformSkip(i, COMPONENT_Cr);
  }
  fwrite(a, 1, sizeof(a), g_dump);
  uint32_t bits = ctu->getTotalBits(), dist = (uint32_t)ctu->getTotalDistortion();
  double cost = ctu->getTotalCost();
  fwrite(&bits, 4, 1, g_dump);
  fwrite(&dist, 4, 1, g_dump);
  fwrite(&cost, 8, 1, g_dump);
  static int16_t c16[4096];
  const int ncoef[3] = { 4096, 1024, 1024 };
  for (int c = 0; c < 3; c++)
  {
    const TCoeff* p = ctu->getCoeff(ComponentID(c));
    for (int i = 0; i < ncoef[c]; i++) c16[i] = (int16_t)p[i];
    fwrite(c16, 2, ncoef[c], g_dump);
  }
  TComPicYuv* rec = ctu->getPic()->getPicYuvRec();
  static uint8_t px[4096];
  static uint16_t px16[4096];
  const bool wide = getenv("HEVCDL_DUMP16") != NULL;
  for (int c = 0; c < 3; c++)
  {
    const ComponentID id = ComponentID(c);
    const int sh = c ? 1 : 0, n = 64 >> sh;
    const int W = rec->getWidth(id), H = rec->getHeight(id), stride = rec->getStride(id);
    const int x0 = (int)ctu->getCUPelX() >> sh, y0 = (int)ctu->getCUPelY() >> sh;
    const Pel* base = rec->getAddr(id);
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++)
      {
        const Pel v = (x0 + x < W && y0 + y < H) ? base[(y0 + y) * stride + x0 + x] : 0;
        px[y * n + x] = (uint8_t)v; px16[y * n + x] = (uint16_t)v;
      }
    if (wide) fwrite(px16, 2, n * n, g_dump); else fwrite(px, 1, n * n, g_dump);
  }
  fflush(g_dump);
}
