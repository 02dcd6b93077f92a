"""Row f-2 (SAO): oracle vs the reference's final reconstruction and bitstream (CPU), HIP kernels vs both (GPU)."""
import glob
import os
import sys

import numpy as np
import pytest

from conftest import GOLD, fixture_tiles, fixture_lf, fixture_lf_offsets, fixture_wavefront

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(GOLD, "rd_*.npz")))


def strip_sei(stream):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevc_parse as hp
    return b"".join((b"\x00" if sc == 4 else b"") + b"\x00\x00\x01" + n for sc, n in hp.split_annexb(stream) if ((n[0] >> 1) & 63) != 40)


def load(path):
    f = np.load(path)
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    nf = f["records"].shape[0]
    fb = w * h * 3 // 2
    dt = np.uint8 if bit_depth_of(f) == 8 else np.dtype("<u2")                 # rd_x*: reference runs at InternalBitDepth 10 (uint16 samples)
    planes = [np.frombuffer(f[k].tobytes(), dt).reshape(nf, fb) for k in ("recon_deblocked", "recon_filtered")]
    return f, w, h, qp, nf, f["yuv"].reshape(nf, fb), planes[0], planes[1]


def bit_depth_of(f):
    return int(f["bit_depth"]) if "bit_depth" in f.files else 8


def tiles_of(f):
    return fixture_tiles(f)


def tools_of(f):
    return int(f["tools"]) if "tools" in f.files else 0x7f          # rd_k*: reference runs with a tool switch of the cfg turned off


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[3:-4])
def test_oracle_sao_matches_reference_picture_and_stream(oracle_built, path):
    """Final reconstruction == the reference's output picture, and the decided parameters, written by the product's
    bitstream writer together with the fixture's records, give the reference's default-configuration stream byte for byte
    (its decoded-picture-hash SEI aside)."""
    import hevcdl_amd
    import ref_tools
    f, w, h, qp, nf, org, dbk, final = load(path)
    params, out = ref_tools.run_sao(org, dbk, w, h, qp, tiles=tiles_of(f), bit_depth=bit_depth_of(f), lf_across_tiles=fixture_lf(f))
    assert np.array_equal(out, final)
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nf, -1)
    aus = [hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], sao=params[poc].view(hevcdl_amd.SAO_DTYPE), tiles=tiles_of(f), bit_depth=bit_depth_of(f),
                                        lf_across_tiles=fixture_lf(f), tools=tools_of(f), lf_offsets=fixture_lf_offsets(f), wavefront=fixture_wavefront(f)) for poc in range(nf)]
    assert b"".join(aus) == strip_sei(f["bitstream"].tobytes())
    # with the decoded-picture-hash SEI (MD5 of the final picture) behind every access unit: the reference's stream, every byte
    assert b"".join(au + hevcdl_amd.picture_hash_sei(w, h, out[poc], bit_depth_of(f)) for poc, au in enumerate(aus)) == f["bitstream"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[3:-4])
def test_gpu_sao_matches_reference(path):
    import hevcdl_amd
    f, w, h, qp, nf, org, dbk, final = load(path)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, tiles=tiles_of(f), bit_depth=bit_depth_of(f), lf_across_tiles=fixture_lf(f))
    params, out = e.sao_frames(org, dbk)
    e.close()
    assert np.array_equal(out, final)
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nf, -1)
    ours = b"".join(hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], sao=params[poc], tiles=tiles_of(f), bit_depth=bit_depth_of(f), lf_across_tiles=fixture_lf(f), tools=tools_of(f), lf_offsets=fixture_lf_offsets(f), wavefront=fixture_wavefront(f)) for poc in range(nf))
    assert ours == strip_sei(f["bitstream"].tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,qp", [(1920, 1080, 32), (3840, 2160, 27), (200, 136, 37)])
def test_gpu_pipeline_matches_oracle_at_full_size(oracle_built, w, h, qp):
    """CNN labels -> decisions -> deblocking -> SAO on the GPU; the two filters are checked against the oracle run on the
    same records / pictures (parameters and samples bit-exact)."""
    import hevcdl_amd
    import ref_tools
    yuv = ref_tools.synth_yuv(w, h, 1, seed=55)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
    recs, recon, _ = e.compress_frames(yuv)
    dbk = e.deblock_frames(recon, recs)
    params, out = e.sao_frames(yuv, dbk)
    e.close()
    o_params, o_out = ref_tools.run_sao(yuv, dbk, w, h, qp)
    assert params.tobytes() == o_params.tobytes()
    assert np.array_equal(out, o_out) and (w < 1000 or (out != dbk).any())


def test_oracle_sao_on_a_picture_that_was_not_deblocked(oracle_built):
    """LoopFilterDisable 1 with SAO on: the reference runs SAO on the unfiltered reconstruction; the oracle on the same input gives its final picture, and the writer
    (pps_deblocking_filter_disabled_flag, SAO syntax) its stream -- the picture-hash SEI aside, and with it."""
    import hevcdl_amd
    import ref_tools
    f = np.load(os.path.join(GOLD, "lfoff_c192_q32.npz"))
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"]); nf = f["records"].shape[0]; fb = w * h * 3 // 2
    org, pre, final = f["yuv"].reshape(nf, fb), np.frombuffer(f["recon"].tobytes(), np.uint8).reshape(nf, fb), np.frombuffer(f["recon_sao"].tobytes(), np.uint8).reshape(nf, fb)
    params, out = ref_tools.run_sao(org, pre, w, h, qp)
    assert np.array_equal(out, final)
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nf, -1)
    aus = [hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], sao=params[poc].view(hevcdl_amd.SAO_DTYPE), lf_disable=True) for poc in range(nf)]
    assert b"".join(aus) == strip_sei(f["bitstream_sao"].tobytes())
    assert b"".join(au + hevcdl_amd.picture_hash_sei(w, h, out[poc]) for poc, au in enumerate(aus)) == f["bitstream_sao"].tobytes()
