"""Container only (needs /root/reference and oracle/_ref): the oracle against fresh runs of the reference encoder over the SAME seeded
configuration sweep that tests/test_rd_gpu.py::test_random_configurations_match_oracle runs on the GPU -- picture size, QP 0..51, bit
depth, content, label policy, tile layout (uniform / explicit), LFCrossTileBoundaryFlag.  Together the two tests tie the HIP path to the
reference on configurations no committed fixture covers."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.exists("/root/reference/encoder_intra_main.cfg") and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_ref"))


def sweep_case(seed):
    """-> (w, h, qp, bit depth, tiles, lf_across_tiles, yuv, labels); keep in step with the GPU test."""
    import ref_tools
    rng = np.random.default_rng(9000 + seed)
    bd = int(rng.choice([8, 10]))
    w, h = int(rng.integers(1, 100)) * 8, int(rng.integers(1, 28)) * 8
    qp = int(rng.integers(0, 52))
    cx, cy = (w + 63) // 64, (h + 63) // 64
    tiles, lf = (1, 1), True
    if cx >= 8 and rng.random() < 0.8:
        cols = int(rng.integers(1, cx // 4 + 1)); rows = int(rng.integers(1, cy + 1))
        if rng.random() < 0.5:
            tiles = (cols, rows)
        else:
            cw = [4] * cols
            for _ in range(cx - 4 * cols):
                cw[int(rng.integers(0, cols))] += 1
            rh = [1] * rows
            for _ in range(cy - rows):
                rh[int(rng.integers(0, rows))] += 1
            tiles = (cw, rh)
        lf = bool(rng.integers(0, 2))
    mx = (1 << bd) - 1
    kind = int(rng.integers(0, 3))
    base = ref_tools.synth_yuv(w, h, 1, 100 + seed).astype(np.int64) * (4 if bd == 10 else 1)
    if kind == 1:
        base = rng.integers(0, mx + 1, base.shape)
    elif kind == 2:
        base = np.where(rng.integers(0, 4, base.shape) > 0, base, rng.integers(0, mx + 1, base.shape))
    yuv = np.clip(base, 0, mx).astype(np.uint8 if bd == 8 else np.uint16)
    labels = ref_tools.make_labels(w, h, 1, ["rand", 0, 1, 2, 3][int(rng.integers(0, 5))], seed + 1)
    return w, h, qp, bd, tiles, lf, yuv, labels


@pytest.mark.skipif(not HAVE_REF, reason="reference sources / build not present")
@pytest.mark.parametrize("seed", list(range(24)))
def test_oracle_equals_a_fresh_reference_run(oracle_built, seed):
    import hevcdl_amd
    import ref_tools as rt
    w, h, qp, bd, tiles, lf, yuv, labels = sweep_case(seed)
    targs = (rt.tile_args(tiles) if tiles != (1, 1) else []) + ([] if lf else ["--LFCrossTileBoundaryFlag=0"])
    dump, out, bits, recon = rt.run_reference(yuv, w, h, qp, labels, extra_args=targs, bit_depth=bd)
    recs, rec, _ = rt.run_oracle(yuv, w, h, qp, labels, tiles=tiles, bit_depth=bd)
    assert rt.compare(dump, recs, rec.reshape(1, -1), w, h, verbose=False) == 0
    dbk = rt.run_deblock(rec.reshape(1, -1), w, h, qp, np.frombuffer(recs.tobytes(), dtype=rt.REC_DTYPE).reshape(1, -1), bit_depth=bd, tiles=tiles, lf_across_tiles=lf)
    sao, fin = rt.run_sao(yuv, dbk, w, h, qp, tiles=tiles, bit_depth=bd, lf_across_tiles=lf)
    assert np.array_equal(fin.reshape(-1), np.frombuffer(recon, np.uint8 if bd == 8 else np.dtype("<u2")))
    # and the product's writer on the oracle's decisions gives the reference's stream, picture-hash SEI included
    au = hevcdl_amd.write_access_unit(w, h, qp, 0, np.frombuffer(recs.tobytes(), hevcdl_amd.REC_DTYPE).reshape(1, -1)[0], sao=sao[0].view(hevcdl_amd.SAO_DTYPE),
                                      tiles=tiles, bit_depth=bd, lf_across_tiles=lf)
    assert au + hevcdl_amd.picture_hash_sei(w, h, fin[0], bd) == bits
