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


def test_bd_against_the_unpruned_anchor_fixture(metrics):
    """F-rd-4: four rate / PSNR points of unpruned HM (the anchor) and of the label-pruned reference, taken from the two encoders' own log
    lines (oracle/gen_fixtures.py gen_bd_anchor); the BD figures are those of the reference's formulas."""
    f = np.load(os.path.join(GOLD, "bd_anchor_small.npz"))
    assert metrics.bd_rate(f["anchor_kbps"], f["anchor_psnr_y"], f["label_kbps"], f["label_psnr_y"]) == pytest.approx(float(f["bd_rate_percent"]), abs=1e-9)
    assert metrics.bd_psnr(f["anchor_kbps"], f["anchor_psnr_y"], f["label_kbps"], f["label_psnr_y"]) == pytest.approx(float(f["bd_psnr_db"]), abs=1e-9)
    # pruning never searches more than the anchor does: at equal QP the anchor's RD cost is at least as good, so the pruned curve is not better overall
    assert float(f["bd_rate_percent"]) > 0


@pytest.mark.gpu
def test_device_path_reproduces_the_label_path_points_of_the_anchor_fixture(metrics):
    """The device path (decisions, deblocking, SAO, bitstream) on the fixture's input and labels gives, QP by QP, exactly the bits and the
    PSNR the reference printed for the same labels -- hence the same BD-rate against the anchor."""
    import hevcdl_amd
    import ref_tools
    f = np.load(os.path.join(GOLD, "bd_anchor_small.npz"))
    w, h, nf = int(f["width"]), int(f["height"]), int(f["frames"])
    yuv = ref_tools.synth_yuv(w, h, nf, int(f["seed"]))
    kbps, psnr = [], []
    for qi, qp in enumerate(f["qps"]):
        enc = hevcdl_amd.Encoder(w, h, int(qp), max_frames=nf)
        recs, final, sao, _ = enc.encode_pictures(yuv, f["labels"])
        enc.close()
        summ = metrics.Summary(w, h, 30.0)
        ysz = w * h
        for i in range(nf):
            au = hevcdl_amd.write_access_unit(w, h, int(qp), 0, recs[i], sao=sao[i])      # POC 0: the fixture's runs code one picture per process
            assert len(au) * 8 == int(f["label_bits"][qi][i]), (qp, i)
            d = (yuv[i].astype(np.int64) - final[i].astype(np.int64)) ** 2
            summ.add(len(au) * 8, (int(d[:ysz].sum()), int(d[ysz:ysz + ysz // 4].sum()), int(d[ysz + ysz // 4:].sum())))
        av = summ.averages()
        assert av[0] == pytest.approx(float(f["label_psnr_yuv"][qi][0]), abs=1e-4)
        kbps.append(summ.bitrate_kbps()); psnr.append(av[0])
    # the reference prints PSNR with four decimals; the cubic fit over these nearly flat points turns that rounding into a few hundredths of a percent
    assert metrics.bd_rate(f["anchor_kbps"], f["anchor_psnr_y"], kbps, psnr) == pytest.approx(float(f["bd_rate_percent"]), abs=0.1)
