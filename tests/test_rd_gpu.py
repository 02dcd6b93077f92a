"""GPU parity of the HIP CTU-decision kernel (through the C ABI) against the reference-generated golden records
(bit-exact: CU depth, partition, intra modes, TU tree, cbf, transform-skip flags, coefficients, bits/dist/cost,
pre-loop-filter reconstruction) and against the plain-C oracle on further seeded inputs."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu
FIELDS = ["depth", "part_size", "luma_dir", "chroma_dir", "tr_idx", "cbf", "tskip", "bits", "dist", "cost", "coeff_y", "coeff_cb", "coeff_cr"]


def ctu_blocks(recon_frame, w, h, addr):
    import ref_tools
    return ref_tools.ctu_recon_from_frame(recon_frame, w, h, addr)


def assert_records_equal(recs, ref, what):
    for k in FIELDS:
        if not np.array_equal(recs[k], ref[k]):
            bad = np.argwhere(np.asarray(recs[k] != ref[k]).reshape(recs.shape[0], recs.shape[1], -1).any(axis=2))
            raise AssertionError("%s: field %s differs at (frame, ctu) %s" % (what, k, bad[:5].tolist()))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "rd_*.npz"))), ids=lambda p: os.path.basename(p)[3:-4])
def test_golden_records_bit_exact(path):
    import hevcdl_amd
    f = np.load(path)
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    yuv, labels, ref = f["yuv"], f["labels"], f["records"]
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=yuv.shape[0])
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    assert_records_equal(recs, ref, os.path.basename(path))
    for fr in range(yuv.shape[0]):
        for a in range(labels.shape[1]):
            y, u, v = ctu_blocks(recon[fr], w, h, a)
            assert np.array_equal(y, f["rec_y"][fr, a]) and np.array_equal(u, f["rec_cb"][fr, a]) and np.array_equal(v, f["rec_cr"][fr, a]), (fr, a)


@pytest.mark.parametrize("w,h,qp,nf,seed", [(256, 128, 30, 3, 41), (136, 72, 24, 2, 42), (320, 192, 40, 1, 43)])
def test_matches_oracle_on_seeded_inputs(oracle_built, w, h, qp, nf, seed):
    import hevcdl_amd
    import ref_tools
    yuv = ref_tools.synth_yuv(w, h, nf, seed)
    labels = ref_tools.make_labels(w, h, nf, "rand", seed + 1)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels)
    assert_records_equal(recs, o_recs, "oracle %dx%d" % (w, h))
    assert np.array_equal(recon, o_recon)
    assert np.array_equal(stats["sse"], o_stats["sse"]) and np.array_equal(stats["est_bits"], o_stats["est_bits"])
