"""Row f-3: PSNR / summary table / Bjontegaard deltas of the host mirror (CPU only)."""
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLD


@pytest.fixture(scope="module")
def metrics():
    import hevcdl_amd  # noqa: F401  (registers the package)
    import hevcdl_amd.metrics as m
    return m


def test_bd_known_answer_of_the_reference_script(metrics):
    """calc_BDBR/Bjontegaard-python3: RatePsnrSample.txt -> -1.1922290103850435 dB / +31.424376673861843 %."""
    d = np.loadtxt(os.path.join(GOLD, "bd_rate_psnr_sample.txt"))
    ka = json.load(open(os.path.join(GOLD, "bd_known_answer.json")))
    assert metrics.bd_psnr(d[:, 0], d[:, 1], d[:, 2], d[:, 3]) == pytest.approx(ka["bd_psnr_db"], abs=1e-12)
    assert metrics.bd_rate(d[:, 0], d[:, 1], d[:, 2], d[:, 3]) == pytest.approx(ka["bd_rate_percent"], abs=1e-10)


def test_bd_properties(metrics):
    r = np.array([686.76, 309.58, 157.11, 85.95]); p = np.array([40.28, 37.18, 34.24, 31.42])
    for mode in ("reference", "common"):
        assert abs(metrics.bd_psnr(r, p, r, p, interval=mode)) < 1e-9          # identical curves
        assert abs(metrics.bd_rate(r, p, r, p, interval=mode)) < 1e-7
        assert metrics.bd_rate(r, p, r * 1.1, p, interval=mode) == pytest.approx(10.0, abs=1e-6)   # +10 % rate at equal quality
        assert metrics.bd_psnr(r, p, r, p + 0.5, interval=mode) == pytest.approx(0.5, abs=1e-9)
    with pytest.raises(ValueError):
        metrics.bd_rate(r, p, r, p, interval="x")


def test_psnr_matches_the_encoder_formula(metrics):
    rng = np.random.default_rng(5)
    w, h = 64, 48
    org = rng.integers(0, 256, (h, w)).astype(np.int64); rec = np.clip(org + rng.integers(-3, 4, (h, w)), 0, 255)
    sse = int(((org - rec) ** 2).sum())
    assert metrics.psnr_from_sse(sse, w * h) == pytest.approx(10 * math.log10(255.0 * 255.0 * w * h / sse), abs=1e-12)
    assert metrics.psnr_from_sse(0, w * h) == 999.99                              # TEncGOP.cpp:2393
    y, u, v = metrics.frame_psnr((sse, 10, 0), w, h)
    assert u == pytest.approx(10 * math.log10(255.0 * 255.0 * (w // 2) * (h // 2) / 10)) and v == 999.99


def test_summary_table_and_picture_line(metrics):
    s = metrics.Summary(416, 240, frame_rate=30)
    s.add(80000, (300000, 20000, 25000)); s.add(90000, (310000, 21000, 26000))
    assert s.bitrate_kbps() == pytest.approx(170000 * 30 / 1000 / 2)
    mse = (4 * (300000 + 310000) / (416 * 240) + (20000 + 21000) / (208 * 120) + (25000 + 26000) / (208 * 120)) / 2 / 6
    assert s.yuv_psnr() == pytest.approx(10 * math.log10(255 * 255 / mse), abs=1e-12)
    txt = s.text()
    assert txt.splitlines()[0] == "\tTotal Frames |   Bitrate     Y-PSNR    U-PSNR    V-PSNR    YUV-PSNR  "
    assert txt.splitlines()[1].startswith("\t        2    a    2550.0000  ")
    line = metrics.frame_line(3, 32, 80000, (33.5678, 40.1, 41.25), 1.4)
    assert line == "POC    3 TId: 0 ( I-SLICE, QP 32 )      80000 bits [Y 33.5678 dB    U 40.1000 dB    V 41.2500 dB] [ET     1 ]"
