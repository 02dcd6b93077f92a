"""Row f-1: the bitstream writer (host code, no GPU): byte-exact against the reference's streams, decodable by the
reference decoder to the deblocked reconstruction."""
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, fixture_tiles, fixture_lf, fixture_lf_offsets, fixture_wavefront

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(GOLD, "rd_*.npz")))
REF_DEC = os.path.join(ROOT, "oracle", "_ref", "TAppDecoder_ref")


def tiles_of(f):
    return fixture_tiles(f)


def stream_of(f):
    import hevcdl_amd
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(f["records"].shape[0], -1)
    bd = int(f["bit_depth"]) if "bit_depth" in f.files else 8        # rd_x*: reference runs at InternalBitDepth 10, Profile main10
    tools = int(f["tools"]) if "tools" in f.files else hevcdl_amd.TOOLS_REFERENCE      # rd_k*: reference runs with a tool switch of the cfg turned off
    return b"".join(hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], tiles=tiles_of(f), bit_depth=bd, lf_across_tiles=fixture_lf(f), tools=tools, lf_offsets=fixture_lf_offsets(f), wavefront=fixture_wavefront(f)) for poc in range(recs.shape[0]))


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[3:-4])
def test_stream_is_byte_exact_with_the_reference(path):
    """Same decisions (the fixture's records) -> the very bytes the reference encoder wrote with --SAO=0:
    parameter sets, slice headers, CABAC payload, emulation prevention, start codes."""
    f = np.load(path)
    assert stream_of(f) == f["bitstream_nosao"].tobytes()


@pytest.mark.skipif(not os.path.exists(REF_DEC), reason="reference decoder build (oracle/_ref) only exists in the survey container")
@pytest.mark.parametrize("path", [p for p in CASES if any(k in p for k in ("w416_q32_r", "w64_q32_r", "w128_q22_r", "w200_q30_b10", "c128_q22_r", "c192_q32_r2", "b200_q27_r2", "b416_q32_r", "t520_q37_2x2", "t576_q27_2x3", "x200_q27_r", "x576_q30_2x3", "k128_q22_sbh0", "k128_q27_ts0", "k200_q27_all0"))],
                         ids=lambda p: os.path.basename(p)[3:-4])
def test_reference_decoder_reconstructs_the_deblocked_picture(path, tmp_path):
    f = np.load(path)
    (tmp_path / "s.bin").write_bytes(stream_of(f))
    r = subprocess.run([REF_DEC, "-b", "s.bin", "-o", "dec.yuv"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ERROR" not in r.stdout, r.stdout[-500:]
    assert np.array_equal(np.fromfile(tmp_path / "dec.yuv", np.uint8), f["recon_deblocked"])


@pytest.mark.skipif(not os.path.exists(REF_DEC), reason="reference decoder build (oracle/_ref) only exists in the survey container")
@pytest.mark.parametrize("name", ["c192_q32_r2", "t576_q27_2x3", "x576_q30_2x3"])
def test_reference_decoder_verifies_the_picture_hash(name, tmp_path):
    """Full default-configuration stream (SAO on) + our MD5 SEI of the fixture's final picture: the reference decoder reconstructs the
    picture and finds its own MD5 equal to the one in the stream for every picture."""
    import hevcdl_amd
    sys_path = os.path.join(ROOT, "oracle")
    import sys
    sys.path.insert(0, sys_path)
    import ref_tools
    f = np.load(os.path.join(GOLD, "rd_%s.npz" % name))
    w, h, qp, nf = int(f["width"]), int(f["height"]), int(f["qp"]), f["records"].shape[0]
    bd = int(f["bit_depth"]) if "bit_depth" in f.files else 8
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    fb = w * h * 3 // 2
    final = np.frombuffer(f["recon_filtered"].tobytes(), dt).reshape(nf, fb)
    dbk = np.frombuffer(f["recon_deblocked"].tobytes(), dt).reshape(nf, fb)
    __import__("__graft_entry__").build_oracle()
    params, out = ref_tools.run_sao(f["yuv"].reshape(nf, fb), dbk, w, h, qp, tiles=tiles_of(f), bit_depth=bd)
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nf, -1)
    stream = b"".join(hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], sao=params[poc].view(hevcdl_amd.SAO_DTYPE), tiles=tiles_of(f), bit_depth=bd)
                      + hevcdl_amd.picture_hash_sei(w, h, final[poc], bd) for poc in range(nf))
    (tmp_path / "s.bin").write_bytes(stream)
    r = subprocess.run([REF_DEC, "-b", "s.bin", "-o", "dec.yuv"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ERROR" not in r.stdout and r.stdout.count("(OK)") == nf, r.stdout[-800:]
    assert np.array_equal(np.fromfile(tmp_path / "dec.yuv", dt), final.reshape(-1))
    # a wrong hash is caught
    bad = bytearray(stream); bad[-10] ^= 0xff
    (tmp_path / "bad.bin").write_bytes(bytes(bad))
    r = subprocess.run([REF_DEC, "-b", "bad.bin", "-o", "dec2.yuv"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert "ERROR" in r.stdout or r.returncode != 0


def test_parameter_sets_and_headers_parse_back():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevc_parse as hp
    import hevcdl_amd
    recs = np.zeros(15 * 8, hevcdl_amd.REC_DTYPE)          # 960x512, all-zero records: one 64x64 CU per CTU, no residual
    recs["luma_dir"][:] = 1; recs["chroma_dir"][:] = 36; recs["tr_idx"][:] = 1
    for poc in (0, 3, 300):
        au = hevcdl_amd.write_access_unit(960, 512, 37, poc, recs, level_idc=93)
        nals = hp.split_annexb(au)
        assert [(n[0] >> 1) & 63 for _, n in nals] == [32, 33, 34, 19 if poc == 0 else 21] and [sc for sc, _ in nals] == [4, 4, 4, 3]
        sps, pps = hp.parse_sps(nals[1][1]), hp.parse_pps(nals[2][1])
        assert (sps["width"], sps["height"], sps["level_idc"], sps["sao"], sps["log2_diff_cb"], sps["log2_diff_tb"]) == (960, 512, 93, 0, 3, 3)
        assert (pps["sign_hiding"], pps["tskip"], pps["init_qp_m26"], pps["cu_qp_delta"]) == (1, 1, 0, 0)
        hdr, _ = hp.parse_slice_header(nals[3][1], sps, pps)
        assert hdr["qp_delta"] == 11 and hdr["slice_type"] == 2 and (poc == 0 or hdr["poc_lsb"] == poc % 256)
        # Annex B: no start-code emulation inside a NAL
        for _, n in nals:
            assert b"\x00\x00\x00" not in n and b"\x00\x00\x01" not in n and b"\x00\x00\x02" not in n


def test_writer_rejects_what_it_does_not_implement():
    import ctypes
    import hevcdl_amd
    lib = hevcdl_amd.load_library()
    cfg = hevcdl_amd.StreamConfig()
    assert lib.hevcdl_stream_config_default(ctypes.byref(cfg), 64, 64, 32) == 0
    assert lib.hevcdl_stream_config_default(ctypes.byref(cfg), 60, 64, 32) != 0
    rec = np.zeros(1, hevcdl_amd.REC_DTYPE); buf = np.zeros(4096, np.uint8); n = ctypes.c_size_t(0)
    cfg.sao_enabled = 1
    assert lib.hevcdl_write_access_unit(ctypes.byref(cfg), 0, rec.ctypes.data, None, buf.ctypes.data, 4096, ctypes.byref(n)) == 1    # SAO flag without parameters
    cfg.sao_enabled = 0; cfg.tools = 0x80
    assert lib.hevcdl_write_access_unit(ctypes.byref(cfg), 0, rec.ctypes.data, None, buf.ctypes.data, 4096, ctypes.byref(n)) == 2    # UNSUPPORTED: no tool of the reference's cfg
    cfg.tools = 0x7f; cfg.lf_tc_offset_div2 = 7                                                                                    # -6 .. 6
    assert lib.hevcdl_write_access_unit(ctypes.byref(cfg), 0, rec.ctypes.data, None, buf.ctypes.data, 4096, ctypes.byref(n)) == 1
    cfg.lf_tc_offset_div2 = 0
    assert lib.hevcdl_write_access_unit(ctypes.byref(cfg), 0, rec.ctypes.data, None, buf.ctypes.data, 8, ctypes.byref(n)) == 1 and n.value > 8
    assert lib.hevcdl_write_access_unit(ctypes.byref(cfg), 0, rec.ctypes.data, None, buf.ctypes.data, 4096, ctypes.byref(n)) == 0
    cfg.tile_columns = 2                                  # a 1-CTU picture cannot hold two tile columns (each at least 4 CTUs wide)
    assert lib.hevcdl_write_access_unit(ctypes.byref(cfg), 0, rec.ctypes.data, None, buf.ctypes.data, 4096, ctypes.byref(n)) == 1
    cfg.tile_columns = 1; cfg.tile_rows = 0
    assert lib.hevcdl_write_access_unit(ctypes.byref(cfg), 0, rec.ctypes.data, None, buf.ctypes.data, 4096, ctypes.byref(n)) == 1


def test_tile_syntax_parses_back():
    """PPS tile fields and the slice header's entry points: the offsets add up to the slice data, every sub-stream ends byte aligned."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevc_parse as hp
    f = np.load(os.path.join(GOLD, "rd_t576_q27_2x3.npz"))
    nals = hp.split_annexb(stream_of(f))
    sps, pps = hp.parse_sps(nals[1][1]), hp.parse_pps(nals[2][1])
    assert pps["tiles_enabled"] == 1 and (pps["tile_columns"], pps["tile_rows"], pps["uniform_spacing"], pps["lf_across_tiles"]) == (2, 3, 1, 1)
    hdr, _ = hp.parse_slice_header(nals[3][1], sps, pps)
    assert len(hdr["entry_points"]) == 5 and sum(hdr["entry_points"]) < len(nals[3][1]) - hdr["data_byte_pos"]


def test_stream_with_the_deblocking_filter_disabled_is_byte_exact_with_the_reference():
    """LoopFilterDisable 1 (with SAO 0): pps_deblocking_filter_disabled_flag behind deblocking_filter_control_present_flag, and no
    slice_loop_filter_across_slices_enabled_flag in the slice header (no in-loop filter is on: TEncCavlc.cpp:1097-1104) -- the reference's stream of such a run, byte for byte."""
    import hevcdl_amd
    f = np.load(os.path.join(GOLD, "lfoff_c192_q32.npz"))
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(f["records"].shape[0], -1)
    ours = b"".join(hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], lf_disable=True) for poc in range(recs.shape[0]))
    assert ours == f["bitstream_nosao"].tobytes()


def test_stream_switches_of_the_cfg():
    """ReWriteParamSetsFlag 0: VPS / SPS / PPS in front of the first picture only (the slice NAL unit then opens the access unit: zero_byte).  LFCrossSliceBoundaryFlag 0: no
    effect -- without slices the reference sets the flag to 1 whatever the cfg says (TAppEncTop.cpp:278-281).  Three reference runs of three frames: byte for byte."""
    import hevcdl_amd
    f = np.load(os.path.join(GOLD, "stream_c192_q32.npz"))
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    recs = np.frombuffer(f["records"].tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(f["records"].shape[0], -1)
    stream = lambda **kw: b"".join(hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], **kw) for poc in range(recs.shape[0]))
    assert stream(rewrite_param_sets=False) == f["bitstream_ps0"].tobytes() == f["bitstream_both"].tobytes()
    assert stream() == f["bitstream_ls0"].tobytes()
    assert len(f["bitstream_ps0"]) < len(f["bitstream_ls0"])


@pytest.mark.parametrize("key,method,bd", [("crc", 2, 8), ("sum", 3, 8), ("crc10", 2, 10), ("sum10", 3, 10)])
def test_crc_and_checksum_picture_hashes_match_the_reference(key, method, bd):
    """SEIDecodedPictureHash 2 / 3: access units + the hash SEI of the reference's final pictures == the reference's streams (8- and 10-bit samples)."""
    import hevcdl_amd
    f = np.load(os.path.join(GOLD, "stream_c192_q32.npz"))
    w, h, qp, nf = int(f["width"]), int(f["height"]), int(f["qp"]), f["records"].shape[0]
    recs = np.frombuffer((f["records"] if bd == 8 else f["records10"]).tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nf, -1)
    pics = np.frombuffer(f["recon_" + key].tobytes(), np.uint8 if bd == 8 else "<u2").reshape(nf, w * h * 3 // 2)
    ours = b"".join(hevcdl_amd.write_access_unit(w, h, qp, poc, recs[poc], bit_depth=bd) + hevcdl_amd.picture_hash_sei(w, h, pics[poc], bd, method) for poc in range(nf))
    assert ours == f["bitstream_" + key].tobytes()
