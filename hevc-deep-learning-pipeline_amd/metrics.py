"""Per-frame metrics, the encoder's summary table and the Bjontegaard deltas.

Mirrors, for the outputs of the GPU path (`hevcdl_frame_stats`: SSE per plane, bits):
  * PSNR per plane            HM_dl/source/Lib/TLibEncoder/TEncGOP.cpp:2380-2394 (10*log10(maxval^2 * N / SSD), 999.99 for SSD 0)
  * the per-picture log line  TEncGOP.cpp:2500-2541
  * the summary table         TEncAnalyze.h:163-196 (combined YUV PSNR), :198-370 (layout of the 4:2:0 table)
  * BD-PSNR / BD-rate         calc_BDBR/Bjontegaard-python3 (cubic fit over ln(rate), integrals over the common interval)
Feed it the size of the written access unit (the reference's bit count of a picture) and the SSE of the final picture
(after the in-loop filters) to reproduce the reference's log; `hevcdl_frame_stats` itself carries the CABAC estimator's
bits and the SSE before the in-loop filters.  maxval = 255 << (bit depth - 8) (TEncGOP.cpp:2380).
"""
import math

import numpy as np

MAXVAL_8BIT = 255


def psnr_from_sse(sse, n_samples, maxval=MAXVAL_8BIT):
    """TEncGOP.cpp:2391-2393."""
    return 999.99 if sse == 0 else 10.0 * math.log10(float(maxval) * maxval * n_samples / float(sse))


def frame_psnr(sse_yuv, width, height, maxval=MAXVAL_8BIT):
    """(Y, U, V) PSNR in dB of one 4:2:0 picture from its three SSE sums."""
    ny, nc = width * height, (width // 2) * (height // 2)
    return tuple(psnr_from_sse(int(s), n, maxval) for s, n in zip(sse_yuv, (ny, nc, nc)))


def frame_line(poc, qp, bits, psnr, enc_time_s=0.0):
    """The picture line of the encoder log (TEncGOP.cpp:2500-2541), intra slices only."""
    return ("POC %4d TId: %1d ( %c-SLICE, QP %d ) %10d bits [Y %6.4f dB    U %6.4f dB    V %6.4f dB] [ET %5.0f ]"
            % (poc, 0, "I", qp, bits, psnr[0], psnr[1], psnr[2], enc_time_s))


class Summary:
    """Running totals of TEncAnalyze (addResult) and its 4:2:0 printOut."""

    def __init__(self, width, height, frame_rate=30.0, bit_depth=8):
        self.w, self.h, self.fps = width, height, float(frame_rate)
        self.maxval = 255 << (bit_depth - 8)
        self.n = 0
        self.bits = 0.0
        self.psnr = [0.0, 0.0, 0.0]
        self.mse = [0.0, 0.0, 0.0]

    def add(self, bits, sse_yuv):
        ny, nc = self.w * self.h, (self.w // 2) * (self.h // 2)
        p = frame_psnr(sse_yuv, self.w, self.h, self.maxval)
        for c, n in enumerate((ny, nc, nc)):
            self.psnr[c] += p[c]
            self.mse[c] += float(int(sse_yuv[c])) / n
        self.bits += float(bits)
        self.n += 1
        return p

    def bitrate_kbps(self):
        return self.bits * (self.fps / 1000.0 / self.n)

    def yuv_psnr(self):
        """calculateCombinedValues, TEncAnalyze.h:163-196 (4:2:0: weights 4,1,1 over 6)."""
        mse = (4 * self.mse[0] + self.mse[1] + self.mse[2]) / self.n / 6.0
        return 999.99 if mse == 0 else 10.0 * math.log10(float(self.maxval) * self.maxval / mse)

    def averages(self):
        return [p / self.n for p in self.psnr]

    def text(self, delim="a"):
        a = self.averages()
        head = "\tTotal Frames |   Bitrate     Y-PSNR    U-PSNR    V-PSNR    YUV-PSNR  \n"
        return head + "\t %8d    %c %12.4f  %8.4f  %8.4f  %8.4f  %8.4f  " % (self.n, delim, self.bitrate_kbps(), a[0], a[1], a[2], self.yuv_psnr())


def _fit_integral(x, y, lo, hi):
    p = np.polyint(np.poly1d(np.polyfit(x, y, 3)))
    return float(np.polyval(p, hi) - np.polyval(p, lo))


def _interval(a, b, mode):
    """"reference": what the reference's script integrates over -- from the largest to the smallest of ALL points of
    both curves (BjontegaardMetric_Python3.py, `min_int = amax(stack)`, `max_int = amin(stack)`), i.e. the union of the
    two ranges, extrapolating each cubic.  "common": the overlap of the two ranges, as in VCEG-M33."""
    if mode == "reference":
        return max(a.max(), b.max()), min(a.min(), b.min())
    if mode == "common":
        return max(a.min(), b.min()), min(a.max(), b.max())
    raise ValueError("interval must be 'reference' or 'common'")


def bd_psnr(rate_anchor, psnr_anchor, rate_test, psnr_test, interval="reference"):
    """Average PSNR difference (dB), test - anchor, of the cubic fits PSNR(ln rate)."""
    r1, r2 = np.log(np.asarray(rate_anchor, float)), np.log(np.asarray(rate_test, float))
    lo, hi = _interval(r1, r2, interval)
    return (_fit_integral(r2, np.asarray(psnr_test, float), lo, hi) - _fit_integral(r1, np.asarray(psnr_anchor, float), lo, hi)) / (hi - lo)


def bd_rate(rate_anchor, psnr_anchor, rate_test, psnr_test, interval="reference"):
    """Average bit-rate difference (percent), test vs anchor, of the cubic fits ln rate(PSNR)."""
    r1, r2 = np.log(np.asarray(rate_anchor, float)), np.log(np.asarray(rate_test, float))
    p1, p2 = np.asarray(psnr_anchor, float), np.asarray(psnr_test, float)
    lo, hi = _interval(p1, p2, interval)
    avg = (_fit_integral(p2, r2, lo, hi) - _fit_integral(p1, r1, lo, hi)) / (hi - lo)
    return (math.exp(avg) - 1.0) * 100.0
