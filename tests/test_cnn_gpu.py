"""GPU parity of the on-device CNN (through the C ABI) against the reference-generated fixtures and the numpy oracle."""
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3        # fp32 tolerance stated by the north star: summation order in conv / BN statistics only


@pytest.fixture(scope="module")
def enc():
    import hevcdl_amd
    e = hevcdl_amd.Encoder(128, 128, 32, max_frames=2)
    yield e
    e.close()


def test_logits_and_labels_match_reference_fixture(enc):
    f = np.load(os.path.join(GOLD, "cnn_f1.npz"))
    labels, logits = enc.predict_depth_rgb(f["ctu_rgb"])
    err = np.abs(logits - f["logits"]).max()
    assert err < LOGIT_TOL, err
    # labels must be exact wherever the top-2 logit gap of every 4-way argmax exceeds the tolerance band
    lg = f["logits"].reshape(-1, 4, 4, 4)
    srt = np.sort(lg, axis=-1)
    safe = ((srt[..., -1] - srt[..., -2]) > 1e-2).all(axis=(1, 2))
    assert safe.sum() > 32
    assert np.array_equal(labels[safe], f["labels"][safe])


def test_eval_mode_batchnorm_matches_the_reference_model_in_eval_mode():
    """HEVCDL_BN_EVAL (BN mode {reference-train, eval}; F-cnn-1's eval half, tests/golden/cnn_f1_eval.npz = the reference
    model after model.eval() on the CTUs of cnn_f1.npz): BatchNorm with the checkpoint's running statistics.  Same tolerance and label rule
    as the reference mode; the two modes disagree on most labels, so a mix-up cannot pass."""
    import hevcdl_amd
    f, g = np.load(os.path.join(GOLD, "cnn_f1.npz")), np.load(os.path.join(GOLD, "cnn_f1_eval.npz"))
    e = hevcdl_amd.Encoder(128, 128, 32, max_frames=2, bn_mode=1)
    labels, logits = e.predict_depth_rgb(f["ctu_rgb"])
    e.close()
    err = np.abs(logits - g["logits"]).max()
    assert err < LOGIT_TOL, err
    srt = np.sort(g["logits"].reshape(-1, 4, 4, 4), axis=-1)
    safe = ((srt[..., -1] - srt[..., -2]) > 1e-2).all(axis=(1, 2))
    assert safe.sum() > 32
    assert np.array_equal(labels[safe], g["labels"][safe])
    assert (g["labels"] != f["labels"]).mean() > 0.3


def test_yuv_path_matches_oracle(enc):
    import cnn_oracle
    import hevcdl_amd
    import ref_tools
    yuv = ref_tools.synth_yuv(128, 128, 2, seed=9)
    labels, logits = enc.predict_depth(yuv, want_logits=True)
    w = cnn_oracle.load_weights(hevcdl_amd.WEIGHTS_PATH)
    o_labels, o_logits = cnn_oracle.predict_labels(w, yuv, 128, 128)
    assert np.abs(logits - o_logits).max() < LOGIT_TOL
    assert np.array_equal(labels, o_labels)


def test_boundary_picture_zero_fill_and_clamp():
    import cnn_oracle
    import hevcdl_amd
    import ref_tools
    w_, h_ = 200, 136
    e = hevcdl_amd.Encoder(w_, h_, 32, max_frames=1)
    yuv = ref_tools.synth_yuv(w_, h_, 1, seed=3)
    labels, logits = e.predict_depth(yuv, want_logits=True)
    wts = cnn_oracle.load_weights(hevcdl_amd.WEIGHTS_PATH)
    o_labels, o_logits = cnn_oracle.predict_labels(wts, yuv, w_, h_)
    assert np.abs(logits - o_logits).max() < LOGIT_TOL
    assert np.array_equal(labels, o_labels)
    md = cnn_oracle.min_depth_table(w_, h_)
    assert (labels[0] >= md).all()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,cnn_input", [("rgb601", 0), ("luma", 1)])
def test_frame_input_modes_equal_the_rgb_ctu_path_on_the_same_samples(mode, cnn_input):
    """The three input forms of the kernel's tile fill (planar 4:2:0 converted on the fly, luma only, packed RGB CTUs) put the same samples into the tiles: the
    logits of a frame must equal, bit for bit, those of its CTUs converted by the oracle's tiling (oracle/cnn_oracle.py yuv_to_rgb_ctus: use_model.py:80-95 + this
    project's YUV -> RGB transform) and handed over as RGB -- on a picture whose right and bottom CTUs are partly outside (zero fill)."""
    import cnn_oracle
    import hevcdl_amd
    import ref_tools
    w_, h_ = 168, 104
    yuv = ref_tools.synth_yuv(w_, h_, 2, seed=21)
    e = hevcdl_amd.Encoder(w_, h_, 32, max_frames=2, cnn_input=cnn_input)
    labels, logits = e.predict_depth(yuv, want_logits=True)
    for f in range(2):
        ctus = cnn_oracle.yuv_to_rgb_ctus(yuv[f], w_, h_, mode=mode)
        r_labels, r_logits = e.predict_depth_rgb(ctus)
        assert np.array_equal(logits[f].reshape(r_logits.shape), r_logits)
    e.close()
