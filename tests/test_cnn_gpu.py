"""GPU parity of the on-device CNN (through the C ABI) against the reference-generated fixtures and the numpy oracle."""
import os
import re

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


def test_whole_pictures_give_the_label_files_of_the_reference_loop():
    """F-cnn-3 on the device (tests/golden/cnn_f3.npz: whole 416x240 / 200x136 pictures through use_model.py:72-125 itself).  (a) every picture cut by the
    oracle's tiling and handed to the kernel as RGB CTUs: logits within the tolerance, label files exact outside the gap band; (b) the grey pictures also
    through the product's own FRAME path (HEVCDL_CNN_INPUT_LUMA puts R = G = B = Y into the tiles, i.e. sample for sample what the reference loop cropped):
    the kernel's tile fill -- CTU raster order, quadrant origins, zero fill past the right and bottom edge -- against the reference loop itself, labels ==
    the reference's label files after this project's boundary clamp."""
    import cnn_oracle
    import hevcdl_amd
    f = np.load(os.path.join(GOLD, "cnn_f3.npz"))
    e = hevcdl_amd.Encoder(128, 128, 32, max_frames=1)
    for n in range(int(f["n_pictures"])):
        ref_lab, ref_lg = f["labels%d" % n], f["logits%d" % n]
        labels, logits = e.predict_depth_rgb(cnn_oracle.rgb_picture_to_ctus(f["rgb%d" % n]))
        assert np.abs(logits - ref_lg).max() < LOGIT_TOL
        srt = np.sort(ref_lg.reshape(-1, 4, 4, 4), axis=-1)
        safe = ((srt[..., -1] - srt[..., -2]) > 1e-2).all(axis=(1, 2))
        assert safe.sum() >= len(ref_lab) // 2
        assert np.array_equal(labels[safe], ref_lab[safe])
    e.close()
    for n in (1, 3):                                   # the grey pictures
        rgb = f["rgb%d" % n]
        h_, w_ = rgb.shape[:2]
        assert np.array_equal(rgb[..., 0], rgb[..., 1]) and np.array_equal(rgb[..., 0], rgb[..., 2])
        yuv = np.concatenate([rgb[..., 0].reshape(-1), np.full(w_ * h_ // 2, 128, np.uint8)])[None]
        e = hevcdl_amd.Encoder(w_, h_, 32, max_frames=1, cnn_input=1)
        labels, logits = e.predict_depth(yuv, want_logits=True)
        e.close()
        ref_lab, ref_lg = f["labels%d" % n], f["logits%d" % n]
        assert np.abs(logits[0] - ref_lg).max() < LOGIT_TOL
        srt = np.sort(ref_lg.reshape(-1, 4, 4, 4), axis=-1)
        safe = ((srt[..., -1] - srt[..., -2]) > 1e-2).all(axis=(1, 2))
        assert np.array_equal(labels[0][safe], cnn_oracle.clamp_labels(ref_lab[None], w_, h_)[0][safe])


def _one_hot(d):
    fake = np.zeros((len(d), 4, 16), np.float32)
    for k in range(4):
        fake[:, :, 4 * k:4 * k + 4] = np.eye(4, dtype=np.float32)[d[:, :, k]]
    return fake


def test_label_stage_on_the_device_gives_the_reference_lines_on_30000_tuples(enc):
    """F-cnn-2 on the device (row a-3): the digit tuples of tests/golden/cnn_f2.npz (10 000 uniformly random: all 256 raw tuples of a quadrant occur) and
    cnn_f2b.npz (20 000 with every quadrant '0000' half of the time: half of the label sets start with 0, the `pred == "0000" and label[..] != "0"` chain of
    use_model.py:111-119 in all 16 zero / non-zero patterns) -- labels by the reference's own lines -- as one-hot logits through hevcdl_labels_from_logits,
    i.e. through the label stage of hevcdl_fc_kernel itself (same code, the fully connected layers skipped).  Exact; and argmax ties go to the first
    maximum as torch.argmax does."""
    import cnn_oracle
    f, g = np.load(os.path.join(GOLD, "cnn_f2.npz")), np.load(os.path.join(GOLD, "cnn_f2b.npz"))
    assert len(np.unique(f["digits"].reshape(-1, 4) @ np.array([64, 16, 4, 1]))) == 256
    assert 0.4 < (g["labels"][:, 0] == 0).mean() < 0.6
    zero_pat = (g["digits"].reshape(-1, 4, 4).max(axis=2) == 0) @ np.array([8, 4, 2, 1])
    assert len(np.unique(zero_pat)) == 16
    for fx in (f, g):
        fake = _one_hot(fx["digits"])
        assert np.array_equal(enc.labels_from_logits(fake), fx["labels"])
        # other values, same order: only the order matters
        assert np.array_equal(enc.labels_from_logits(fake * 7.5 - 30.0), fx["labels"])
    ties = np.zeros((4, 4, 16), np.float32)
    ties[1][:, [1, 2]] = 3.0
    ties[2][:, [3, 0]] = 1.0
    ties[3][:, 2:4] = 2.0
    assert np.array_equal(enc.labels_from_logits(ties), cnn_oracle.labels_from_logits(ties))


def test_sidecar_writes_the_label_files_of_the_reference_loop(tmp_path):
    """The file-level drop-in for use_model.py (hevcdl_amd.sidecar): the whole pictures of tests/golden/cnn_f3.npz -- which the reference's own loop (use_model.py:72-125) turned
    into label files -- are put where the reference expects its frames (rec/frames/<n>.jpg; stored losslessly, PIL reads by content, so the decoder hands over exactly the
    fixture's RGB) and go through the sidecar: same directory layout, same file text (16 digits, a blank behind each), the reference's labels outside the logits' tie band."""
    import io
    from PIL import Image
    import hevcdl_amd.sidecar as sidecar
    f = np.load(os.path.join(GOLD, "cnn_f3.npz"))
    n_pic = int(f["n_pictures"])
    frames = tmp_path / "rec" / "frames"
    os.makedirs(frames)
    for n in range(n_pic):
        buf = io.BytesIO(); Image.fromarray(f["rgb%d" % n]).save(buf, format="PNG")
        (frames / ("%d.jpg" % (n + 1))).write_bytes(buf.getvalue())
    (tmp_path / "bitstream.cfg").write_text("InputFile : in.yuv\nFramesToBeEncoded            : %d\n" % (n_pic - 1))
    assert sidecar.frames_to_be_encoded(str(tmp_path / "bitstream.cfg")) == n_pic - 1
    done = sidecar.label_frames(str(frames), str(tmp_path / "pred"), n_frames=n_pic - 1, log=lambda *a: None)
    assert done == n_pic - 1 and sorted(os.listdir(tmp_path / "pred")) == [str(i) for i in range(n_pic - 1)]          # FramesToBeEncoded stops the loop (use_model.py:74-75)
    for n in range(n_pic - 1):
        ref_lab, ref_lg = f["labels%d" % n], f["logits%d" % n]
        files = sorted(os.listdir(tmp_path / "pred" / str(n)), key=lambda s: int(s[3:-4]))
        assert files == ["ctu%d.txt" % i for i in range(len(ref_lab))]
        got = []
        for name in files:
            text = (tmp_path / "pred" / str(n) / name).read_text()
            assert re.fullmatch(r"(\d ){16}", text), text
            got.append([int(v) for v in text.split()])
        got = np.array(got, np.uint8)
        srt = np.sort(ref_lg.reshape(-1, 4, 4, 4), axis=-1)
        safe = ((srt[..., -1] - srt[..., -2]) > 1e-2).all(axis=(1, 2))
        assert safe.sum() >= len(ref_lab) // 2 and np.array_equal(got[safe], ref_lab[safe])
    with pytest.raises(FileExistsError):                # labels of an old run are never mixed in (os.mkdir in the reference)
        sidecar.label_frames(str(frames), str(tmp_path / "pred"), n_frames=1, log=lambda *a: None)
