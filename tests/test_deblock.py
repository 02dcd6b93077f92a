"""Row f-2 (deblocking): oracle vs the reference's deblocked reconstruction (CPU), HIP kernels vs both (GPU)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD, fixture_tiles, fixture_lf, fixture_lf_offsets

CASES = sorted(glob.glob(os.path.join(GOLD, "rd_*.npz")))


def bit_depth_of(f):
    return int(f["bit_depth"]) if "bit_depth" in f.files else 8         # rd_x*: reference runs at InternalBitDepth 10 (uint16 samples)


def deblocked_of(f, shape):
    return np.frombuffer(f["recon_deblocked"].tobytes(), np.uint8 if bit_depth_of(f) == 8 else "<u2").reshape(shape)


def prefilter_frames(f):
    """Planar pre-filter frames assembled from the per-CTU reconstruction blocks of a fixture."""
    w, h = int(f["width"]), int(f["height"])
    nf, nctu = f["records"].shape[0], f["records"].shape[1]
    cx = (w + 63) // 64
    dt = f["rec_y"].dtype
    frames = np.zeros((nf, w * h * 3 // 2), dt)
    for fr in range(nf):
        Y = np.zeros((h + 64, w + 64), dt); U = np.zeros((h // 2 + 32, w // 2 + 32), dt); V = U.copy()
        for a in range(nctu):
            x0, y0 = (a % cx) * 64, (a // cx) * 64
            Y[y0:y0 + 64, x0:x0 + 64] = f["rec_y"][fr, a].reshape(64, 64)
            U[y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 32] = f["rec_cb"][fr, a].reshape(32, 32)
            V[y0 // 2:y0 // 2 + 32, x0 // 2:x0 // 2 + 32] = f["rec_cr"][fr, a].reshape(32, 32)
        frames[fr] = np.concatenate([Y[:h, :w].ravel(), U[:h // 2, :w // 2].ravel(), V[:h // 2, :w // 2].ravel()])
    return frames


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[3:-4])
def test_oracle_deblock_matches_reference(oracle_built, path):
    import ref_tools
    f = np.load(path)
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    recs = np.frombuffer(f["records"].tobytes(), dtype=ref_tools.REC_DTYPE).reshape(f["records"].shape[0], -1)
    pre = prefilter_frames(f)
    out = ref_tools.run_deblock(pre, w, h, qp, recs, bit_depth=bit_depth_of(f), tiles=fixture_tiles(f), lf_across_tiles=fixture_lf(f), lf_offsets=fixture_lf_offsets(f))
    ref = deblocked_of(f, out.shape)
    if qp >= 22 and w * h >= 128 * 128:
        assert (pre != ref).sum() > 1000                   # the filter does something on every ordinary fixture (tc = beta = 0 at QP 0)
    assert np.array_equal(out, ref)


def test_oracle_deblock_rejects_bad_arguments(oracle_built):
    import ctypes
    import ref_tools
    lib = ref_tools.oracle_lib()
    lib.hm_oracle_deblock_frame.restype = ctypes.c_int
    lib.hm_oracle_deblock_frame.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    buf = np.zeros(64 * 64 * 3 // 2, np.uint8); rec = np.zeros(1, ref_tools.REC_DTYPE)
    assert lib.hm_oracle_deblock_frame(None, 64, 64, 32, rec.ctypes.data) != 0
    assert lib.hm_oracle_deblock_frame(buf.ctypes.data, 60, 64, 32, rec.ctypes.data) != 0
    assert lib.hm_oracle_deblock_frame(buf.ctypes.data, 64, 64, 52, rec.ctypes.data) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[3:-4])
def test_gpu_deblock_matches_reference(path):
    import hevcdl_amd
    f = np.load(path)
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    nf = f["records"].shape[0]
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nf, -1)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, bit_depth=bit_depth_of(f), tiles=fixture_tiles(f), lf_across_tiles=fixture_lf(f), lf_offsets=fixture_lf_offsets(f))
    out = e.deblock_frames(prefilter_frames(f), recs)
    e.close()
    assert np.array_equal(out, deblocked_of(f, out.shape))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,qp", [(1920, 1080, 32), (3840, 2160, 22), (200, 136, 37)])
def test_gpu_deblock_matches_oracle_at_full_size(oracle_built, w, h, qp):
    """End to end on the GPU path's own decisions: CNN labels -> records + reconstruction -> deblocked picture, checked
    against the oracle filter run on the same records; the filter must leave samples farther than 3 from the 8x8 grid alone."""
    import hevcdl_amd
    import ref_tools
    yuv = ref_tools.synth_yuv(w, h, 1, seed=77)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
    recs, recon, _ = e.compress_frames(yuv)
    out = e.deblock_frames(recon, recs)
    e.close()
    ref = ref_tools.run_deblock(recon, w, h, qp, np.frombuffer(recs.tobytes(), dtype=ref_tools.REC_DTYPE).reshape(1, -1))
    assert np.array_equal(out, ref)
    Y0, Y1 = recon[0, :w * h].reshape(h, w), out[0, :w * h].reshape(h, w)
    assert np.array_equal(Y0[3::8, 3::8], Y1[3::8, 3::8]) and np.array_equal(Y0[4::8, 4::8], Y1[4::8, 4::8]) and (Y0 != Y1).any()
