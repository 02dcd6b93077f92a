#!/usr/bin/env python3
"""oracle/gen_fixtures.py -- TEST INFRASTRUCTURE.  Runs ONLY in the container that has /root/reference.

Generates the committed golden vectors under tests/golden/ by running the reference itself:
  * RD fixtures: the reference encoder built in place by oracle/build_ref.sh (oracle/_ref/TAppEncoder_ref,
    observed at the TEncCu::compressCtu boundary by oracle/ref_hook.cpp) on small synthetic YUVs with explicit
    label files -> per-CTU decision records + pre-loop-filter reconstruction (+ bitstream bytes and the
    encoder's per-frame summary line for the later "next" rows).
  * CNN fixtures: the ConvNet2 class text of /root/reference/use_model.py:16-58 exec()'d here with the
    reference checkpoint /root/reference/rec/hevc_encoder_model.pt, BatchNorm left in training mode exactly as
    the reference runs it; label post-processing fixtures by exec()ing use_model.py:101-119.
  * the weight blob (flat little-endian fp32 + JSON manifest) consumed by the product.
No reference source text is written to the repo: fixtures are inputs and expected outputs only.
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_tools as rt  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
WDIR = os.path.join(ROOT, "hevc-deep-learning-pipeline_amd", "weights")
REF = "/root/reference"


EXTREME = {  # name: (W, H, qp, content, bit depth) -- smallest pictures, one-CU-wide strips, QP 0 / 51, constant pictures
    "e8x8_q32": (8, 8, 32, "pattern", 8), "e64x8_q20": (64, 8, 20, "noise", 8), "e8x72_q40": (8, 72, 40, "noise", 8),
    "e136_q0": (136, 72, 0, "noise", 8), "e136_q51": (136, 72, 51, "pattern", 8), "e72_q0_b10": (72, 136, 0, "noise", 10),
    "e200_q51_b10": (200, 72, 51, "noise", 10), "e64_white_b10": (64, 64, 30, "white", 10)}


def extreme_yuv(w, h, qp, kind, bd):
    rng = np.random.default_rng(w * 1000 + h + qp)
    mx = (1 << bd) - 1
    if kind == "noise":
        yuv = rng.integers(0, mx + 1, (1, w * h * 3 // 2))
    elif kind == "white":
        yuv = np.full((1, w * h * 3 // 2), mx, np.int64)
    else:
        yuv = rt.synth_yuv(w, h, 1, w + h).astype(np.int64) * (4 if bd == 10 else 1)
    return yuv.astype(np.uint8 if bd == 8 else np.uint16)


def gen_rd(tiled=False, ten_bit=False, extreme=False):
    cases = []
    #        name            W    H   frames qp  labels seed
    spec = [("c128_q32_d0", 128, 128, 1, 32, 0, 11), ("c128_q32_d1", 128, 128, 1, 32, 1, 12),
            ("c128_q32_d2", 128, 128, 1, 32, 2, 13), ("c128_q32_d3", 128, 128, 1, 32, 3, 14),
            ("c128_q22_r", 128, 128, 1, 22, "rand", 15), ("c128_q27_r", 128, 128, 1, 27, "rand", 16),
            ("c128_q37_r", 128, 128, 1, 37, "rand", 17), ("c192_q32_r2", 192, 128, 2, 32, "rand", 18),
            ("c384_q32_r", 384, 192, 1, 32, "rand", 19),
            ("b416_q32_r", 416, 240, 1, 32, "rand", 20), ("b200_q27_r2", 200, 136, 2, 27, "rand", 21),
            ("b200_q37_d3", 200, 136, 1, 37, 3, 22)]
    spec = [e + ((1, 1),) for e in spec]
    if tiled:
        # tiles (columns, rows), uniformly spaced; the reference needs tiles at least 4 CTUs wide.  520x136 has a ragged right/bottom
        # edge and uneven columns (4 + 5 CTUs)
        spec = [("t512_q32_2x2", 512, 128, 1, 32, "rand", 31, (2, 2)), ("t576_q27_2x3", 576, 192, 1, 27, "rand", 32, (2, 3)),
                ("t520_q37_2x2", 520, 136, 2, 37, "rand", 33, (2, 2)), ("t832_q32_3x1", 832, 128, 1, 32, "rand", 34, (3, 1)),
                # TileUniformSpacing 0 with explicit sizes (all columns / rows listed; the reference is given all but the last)
                ("n832_q32_544x12", 832, 192, 1, 32, "rand", 91, ([5, 4, 4], [1, 2])), ("n712_q27_b10", 712, 136, 1, 27, "rand", 92, ([7, 5], [2, 1])),
                # LFCrossTileBoundaryFlag 0: deblocking and SAO stop at the tile borders
                ("l576_q32_lf0", 576, 192, 1, 32, "rand", 93, (2, 3)), ("l520_q27_lf0_b10", 520, 200, 1, 27, "rand", 94, (2, 2))]
    bd = 8
    if ten_bit:
        # InputBitDepth = InternalBitDepth = 10, Profile main10; samples = the 8-bit pattern * 4 + 2 bits of noise;
        # the last case is C5 in miniature: 10-bit with tiles
        bd = 10
        spec = [("x128_q32_r", 128, 128, 1, 32, "rand", 71, (1, 1)), ("x200_q27_r", 200, 136, 1, 27, "rand", 72, (1, 1)),
                ("x128_q22_d3", 128, 128, 1, 22, 3, 73, (1, 1)), ("x192_q37_r2", 192, 128, 2, 37, "rand", 74, (1, 1)),
                ("x576_q30_2x3", 576, 192, 1, 30, "rand", 75, (2, 3))]
    if extreme:
        spec = [(name, v[0], v[1], 1, v[2], "rand", v[2] + 7 - 100, (1, 1)) for name, v in EXTREME.items()]
    for name, w, h, nf, qp, kind, seed, tiles in spec:
        targs = rt.tile_args(tiles) if tiles != (1, 1) else []
        if name in ("n712_q27_b10", "l520_q27_lf0_b10"):
            bd, ten_bit = 10, True
        else:
            bd, ten_bit = (8, False) if tiled else (bd, ten_bit)
        if name.startswith("l"):
            targs = targs + ["--LFCrossTileBoundaryFlag=0"]
        if extreme:
            bd = EXTREME[name][4]
        yuv = rt.synth_yuv(w, h, nf, seed) if not extreme else None
        if name.startswith("c128_q22"):            # one noisy case: transform skip, escapes, sign hiding
            rng = np.random.default_rng(seed)
            yuv = rng.integers(0, 256, yuv.shape).astype(np.uint8)
        if extreme:
            yuv = extreme_yuv(*EXTREME[name])
        if ten_bit:
            rng = np.random.default_rng(seed)
            yuv = yuv.astype(np.uint16) * 4 + rng.integers(0, 4, yuv.shape).astype(np.uint16)
            if name.startswith("x128_q22"):
                yuv = rng.integers(0, 1024, yuv.shape).astype(np.uint16)
        lab = rt.make_labels(w, h, nf, kind, seed + 100)
        dump, out, bitstream, recon = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs, bit_depth=bd)
        # deblocked-only reconstruction: the same run with SAO switched off (decisions are untouched by the in-loop filters)
        dump2, _, bitstream_nosao, recon_dbk = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs + ["--SAO=0", "--SEIDecodedPictureHash=0"], bit_depth=bd)
        assert dump2.tobytes() == dump.tobytes()
        order = np.lexsort((dump["addr"], dump["frame"]))
        dump = dump[order]
        nctu = lab.shape[1]
        assert len(dump) == nf * nctu
        summary = [ln for ln in out.splitlines() if ln.startswith("POC")]
        np.savez_compressed(os.path.join(GOLD, "rd_%s.npz" % name), width=w, height=h, qp=qp, yuv=yuv, labels=lab, bit_depth=bd, lf_across_tiles=0 if name.startswith("l") else 1,
                            **({"tiles": np.array(tiles)} if isinstance(tiles[0], int) else {"tiles": np.array([len(tiles[0]), len(tiles[1])]), "tile_col_sizes": np.array(tiles[0]), "tile_row_sizes": np.array(tiles[1])}),
                            records=dump["rec"].reshape(nf, nctu), rec_y=dump["rec_y"].reshape(nf, nctu, 4096),
                            rec_cb=dump["rec_cb"].reshape(nf, nctu, 1024), rec_cr=dump["rec_cr"].reshape(nf, nctu, 1024),
                            bitstream=np.frombuffer(bitstream, np.uint8), recon_filtered=np.frombuffer(recon, np.uint8), recon_deblocked=np.frombuffer(recon_dbk, np.uint8), bitstream_nosao=np.frombuffer(bitstream_nosao, np.uint8),
                            summary=np.array(summary))
        cases.append(name)
        print("rd fixture", name, "ctus", nf * nctu, summary[0][:60] if summary else "")
    return cases


def gen_rd_tools():
    """Tool switches of the cfg other than the reference's values (TAppEncCfg.cpp:900-901,917-918,950,978,1007): the reference encoder run with the switch on its command
    line; the fixture carries the tool mask (HEVCDL_TOOL_* of include/hevcdl.h)."""
    T = rt
    #        name             W    H   frames qp labels seed  tools cleared
    spec = [("k128_q22_sbh0", 128, 128, 1, 22, "rand", 15, T.TOOL_SIGN_HIDE),                 # the noisy picture of c128_q22_r: escapes, many levels a group
            ("k128_q27_ts0", 128, 128, 1, 27, "rand", 16, T.TOOL_TSKIP),
            ("k192_q32_sis0", 192, 128, 1, 32, 1, 51, T.TOOL_STRONG_INTRA),                    # 32x32 CUs on smooth content: the strong filter would apply
            ("k200_q32_mpm0", 200, 136, 1, 32, "rand", 52, T.TOOL_FAST_UDI_MPM),
            ("k200_q27_all0", 200, 136, 2, 27, "rand", 53, T.TOOL_SIGN_HIDE | T.TOOL_TSKIP | T.TOOL_STRONG_INTRA | T.TOOL_FAST_UDI_MPM),
            # the quantiser without RDOQ (dead-zone rounding + signBitHidingHDQ): everywhere / in transform-skipped blocks only / without sign hiding as well
            ("k128_q22_rdoq0", 128, 128, 1, 22, "rand", 15, T.TOOL_RDOQ | T.TOOL_RDOQTS), ("k128_q27_rdoqts0", 128, 128, 1, 27, "rand", 16, T.TOOL_RDOQTS),
            ("k128_q27_tsf0", 128, 128, 1, 27, "rand", 16, T.TOOL_TSKIP_FAST), ("k200_q32_tsf0", 200, 136, 1, 32, "rand", 56, T.TOOL_TSKIP_FAST),
            ("k200_q32_rdoq0", 200, 136, 1, 32, "rand", 54, T.TOOL_RDOQ),
            # 10-bit samples (InternalBitDepth 10): the 10-bit build of the kernel reads the switches at run time
            ("k200_q30_b10_mix0", 200, 136, 1, 30, "rand", 57, T.TOOL_RDOQ | T.TOOL_SIGN_HIDE | T.TOOL_TSKIP_FAST | T.TOOL_FAST_UDI_MPM), ("k200_q27_rdoq0_sbh0", 200, 136, 1, 27, "rand", 55, T.TOOL_RDOQ | T.TOOL_RDOQTS | T.TOOL_SIGN_HIDE)]
    cases = []
    for name, w, h, nf, qp, kind, seed, off in spec:
        tools = T.TOOLS_REFERENCE & ~off
        yuv = rt.synth_yuv(w, h, nf, seed)
        if name.startswith("k128_q22") or name.startswith("k128_q27"):
            rng = np.random.default_rng(seed)
            yuv = rng.integers(0, 256, yuv.shape).astype(np.uint8) if name.startswith("k128_q22") else np.clip(yuv.astype(np.int32) + rng.integers(-24, 25, yuv.shape), 0, 255).astype(np.uint8)
        bd = 10 if "_b10" in name else 8
        if bd == 10:
            rng = np.random.default_rng(seed)
            yuv = yuv.astype(np.uint16) * 4 + rng.integers(0, 4, yuv.shape).astype(np.uint16)
        lab = rt.make_labels(w, h, nf, kind, seed + 100)
        targs = rt.tool_args(tools)
        dump, out, bitstream, recon = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs, bit_depth=bd)
        dump2, _, bitstream_nosao, recon_dbk = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs + ["--SAO=0", "--SEIDecodedPictureHash=0"], bit_depth=bd)
        assert dump2.tobytes() == dump.tobytes()
        dump = dump[np.lexsort((dump["addr"], dump["frame"]))]
        nctu = lab.shape[1]
        assert len(dump) == nf * nctu
        summary = [ln for ln in out.splitlines() if ln.startswith("POC")]
        np.savez_compressed(os.path.join(GOLD, "rd_%s.npz" % name), width=w, height=h, qp=qp, yuv=yuv, labels=lab, bit_depth=bd, lf_across_tiles=1, tiles=np.array((1, 1)), tools=tools,
                            records=dump["rec"].reshape(nf, nctu), rec_y=dump["rec_y"].reshape(nf, nctu, 4096),
                            rec_cb=dump["rec_cb"].reshape(nf, nctu, 1024), rec_cr=dump["rec_cr"].reshape(nf, nctu, 1024),
                            bitstream=np.frombuffer(bitstream, np.uint8), recon_filtered=np.frombuffer(recon, np.uint8), recon_deblocked=np.frombuffer(recon_dbk, np.uint8), bitstream_nosao=np.frombuffer(bitstream_nosao, np.uint8),
                            summary=np.array(summary))
        cases.append(name)
        print("rd tools fixture", name, "tools 0x%02x" % tools, "ctus", nf * nctu, summary[0][:60] if summary else "")
    return cases


def gen_rd_lf():
    """Deblocking control of the cfg other than the reference's values: LoopFilterBetaOffset_div2 / LoopFilterTcOffset_div2 (rd_o*: ordinary fixtures that carry
    lf_offsets) and LoopFilterDisable 1 (lfoff_*: the stream and the unfiltered picture of a run with SAO 0 only -- this project runs SAO on the deblocked picture)."""
    for name, w, h, nf, qp, seed, off in (("o192_q32_b2_tm1", 192, 128, 1, 32, 61, (2, -1)), ("o200_q27_bm3_t3", 200, 136, 2, 27, 62, (-3, 3)), ("o128_q37_b6_tm6", 128, 128, 1, 37, 63, (6, -6))):
        yuv = rt.synth_yuv(w, h, nf, seed)
        lab = rt.make_labels(w, h, nf, "rand", seed + 100)
        targs = ["--LoopFilterBetaOffset_div2=%d" % off[0], "--LoopFilterTcOffset_div2=%d" % off[1]]
        dump, out, bitstream, recon = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs)
        dump2, _, bitstream_nosao, recon_dbk = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs + ["--SAO=0", "--SEIDecodedPictureHash=0"])
        assert dump2.tobytes() == dump.tobytes()
        dump = dump[np.lexsort((dump["addr"], dump["frame"]))]
        nctu = lab.shape[1]
        summary = [ln for ln in out.splitlines() if ln.startswith("POC")]
        np.savez_compressed(os.path.join(GOLD, "rd_%s.npz" % name), width=w, height=h, qp=qp, yuv=yuv, labels=lab, bit_depth=8, lf_across_tiles=1, tiles=np.array((1, 1)), lf_offsets=np.array(off),
                            records=dump["rec"].reshape(nf, nctu), rec_y=dump["rec_y"].reshape(nf, nctu, 4096),
                            rec_cb=dump["rec_cb"].reshape(nf, nctu, 1024), rec_cr=dump["rec_cr"].reshape(nf, nctu, 1024),
                            bitstream=np.frombuffer(bitstream, np.uint8), recon_filtered=np.frombuffer(recon, np.uint8), recon_deblocked=np.frombuffer(recon_dbk, np.uint8), bitstream_nosao=np.frombuffer(bitstream_nosao, np.uint8),
                            summary=np.array(summary))
        print("rd lf-offset fixture", name, off, summary[0][:60] if summary else "")
    w, h, nf, qp, seed = 192, 128, 2, 32, 64
    yuv = rt.synth_yuv(w, h, nf, seed)
    lab = rt.make_labels(w, h, nf, "rand", seed + 100)
    dump, out, bitstream, recon = rt.run_reference(yuv, w, h, qp, lab, extra_args=["--LoopFilterDisable=1", "--SAO=0", "--SEIDecodedPictureHash=0"])
    dump = dump[np.lexsort((dump["addr"], dump["frame"]))]
    # ... and with SAO on (the reference's default): SAO then works on the unfiltered reconstruction
    dump_s, out_s, bitstream_s, recon_s = rt.run_reference(yuv, w, h, qp, lab, extra_args=["--LoopFilterDisable=1"])
    np.savez_compressed(os.path.join(GOLD, "lfoff_c192_q32.npz"), width=w, height=h, qp=qp, yuv=yuv, labels=lab, records=dump["rec"].reshape(nf, lab.shape[1]),
                        bitstream_nosao=np.frombuffer(bitstream, np.uint8), recon=np.frombuffer(recon, np.uint8), summary=np.array([ln for ln in out.splitlines() if ln.startswith("POC")]),
                        bitstream_sao=np.frombuffer(bitstream_s, np.uint8), recon_sao=np.frombuffer(recon_s, np.uint8), summary_sao=np.array([ln for ln in out_s.splitlines() if ln.startswith("POC")]))
    print("LoopFilterDisable fixture lfoff_c192_q32")
    # stream-only switches: parameter sets in front of the first picture only, pps_loop_filter_across_slices_enabled_flag 0 (one slice per picture: same pictures)
    w, h, nf, qp, seed = 192, 128, 3, 32, 65
    yuv = rt.synth_yuv(w, h, nf, seed)
    lab = rt.make_labels(w, h, nf, "rand", seed + 100)
    out = {}
    for key, args in (("ps0", ["--ReWriteParamSetsFlag=0"]), ("ls0", ["--LFCrossSliceBoundaryFlag=0"]), ("both", ["--ReWriteParamSetsFlag=0", "--LFCrossSliceBoundaryFlag=0"])):
        dump, _, bitstream, recon = rt.run_reference(yuv, w, h, qp, lab, extra_args=args + ["--SAO=0", "--SEIDecodedPictureHash=0"])
        out["bitstream_" + key] = np.frombuffer(bitstream, np.uint8); out["recon_" + key] = np.frombuffer(recon, np.uint8)
    dump = dump[np.lexsort((dump["addr"], dump["frame"]))]
    # SEIDecodedPictureHash 2 (CRC) / 3 (checksum), 8- and 10-bit samples
    for key, args, bd in (("crc", ["--SEIDecodedPictureHash=2"], 8), ("sum", ["--SEIDecodedPictureHash=3"], 8), ("crc10", ["--SEIDecodedPictureHash=2"], 10), ("sum10", ["--SEIDecodedPictureHash=3"], 10)):
        y = yuv if bd == 8 else yuv.astype(np.uint16) * 4 + 1
        d2, o2, bitstream, recon = rt.run_reference(y, w, h, qp, lab, extra_args=args + ["--SAO=0"], bit_depth=bd)
        d2 = d2[np.lexsort((d2["addr"], d2["frame"]))]
        out["bitstream_" + key] = np.frombuffer(bitstream, np.uint8); out["recon_" + key] = np.frombuffer(recon, np.uint8)
        out["summary_" + key] = np.array([ln for ln in o2.splitlines() if ln.startswith("POC")])
        if bd == 10: out["records10"] = d2["rec"].reshape(nf, lab.shape[1]); out["yuv10"] = y
    np.savez_compressed(os.path.join(GOLD, "stream_c192_q32.npz"), width=w, height=h, qp=qp, yuv=yuv, labels=lab, records=dump["rec"].reshape(nf, lab.shape[1]), **out)
    print("stream-switch fixture stream_c192_q32")


def gen_rd_wpp():
    """WaveFrontSynchro 1 (entropy_coding_sync_enabled_flag, TAppEncCfg.cpp:975): the reference encoder run with the key on its command line -- CTU rows start from the contexts
    behind the second CTU of the row above (TEncSlice.cpp:783-830, 925-928), a sub-stream per row.  The fixtures carry wavefront = 1; everything else as the ordinary ones."""
    #        name             W    H   frames qp labels seed bits
    spec = [("w256_q32_r", 256, 192, 1, 32, "rand", 81, 8),           # 4 x 3 CTUs
            ("w200_q27_r2", 200, 136, 2, 27, "rand", 82, 8),          # ragged right / bottom edge, two pictures
            ("w64_q32_r", 64, 192, 1, 32, "rand", 83, 8),             # one CTU wide: no CTU above and to the right, every row starts from the slice-start contexts
            ("w128_q22_r", 128, 256, 1, 22, "rand", 84, 8),           # two CTUs wide: the sync state is the state behind the LAST CTU of the row above; noise (escapes, sign hiding)
            ("w416_q32_r", 416, 240, 1, 32, "rand", 85, 8),           # C1's picture size
            ("w384_q37_d1", 384, 256, 1, 37, 1, 86, 8),               # 32x32 CUs
            ("w200_q30_b10", 200, 200, 1, 30, "rand", 87, 10),        # 10-bit samples
            ("w192_q27_k", 192, 192, 1, 27, "rand", 88, 8)]           # with tool switches off as well (sign hiding, transform skip): the build of the kernel that reads them at run time
    for name, w, h, nf, qp, kind, seed, bd in spec:
        tools = rt.TOOLS_REFERENCE & ~(rt.TOOL_SIGN_HIDE | rt.TOOL_TSKIP) if name.endswith("_k") else rt.TOOLS_REFERENCE
        yuv = rt.synth_yuv(w, h, nf, seed)
        if name.startswith("w128_q22"):
            yuv = np.random.default_rng(seed).integers(0, 256, yuv.shape).astype(np.uint8)
        if bd == 10:
            yuv = yuv.astype(np.uint16) * 4 + np.random.default_rng(seed).integers(0, 4, yuv.shape).astype(np.uint16)
        lab = rt.make_labels(w, h, nf, kind, seed + 100)
        targs = ["--WaveFrontSynchro=1"] + (rt.tool_args(tools) if tools != rt.TOOLS_REFERENCE else [])
        if name.endswith("_k"):
            yuv = np.clip(yuv.astype(np.int32) + np.random.default_rng(seed).integers(-24, 25, yuv.shape), 0, 255).astype(np.uint8)
        dump, out, bitstream, recon = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs, bit_depth=bd)
        dump2, _, bitstream_nosao, recon_dbk = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs + ["--SAO=0", "--SEIDecodedPictureHash=0"], bit_depth=bd)
        assert dump2.tobytes() == dump.tobytes()
        dump0, _, _, _ = rt.run_reference(yuv, w, h, qp, lab, extra_args=targs[1:] + ["--SAO=0", "--SEIDecodedPictureHash=0"], bit_depth=bd)      # the same picture without the key: other decisions (the fixture is not vacuous)
        dump = dump[np.lexsort((dump["addr"], dump["frame"]))]
        dump0 = dump0[np.lexsort((dump0["addr"], dump0["frame"]))]
        nctu = lab.shape[1]
        assert len(dump) == nf * nctu
        differs = int(sum(not np.array_equal(a, b) for a, b in zip(dump["rec"], dump0["rec"])))
        summary = [ln for ln in out.splitlines() if ln.startswith("POC")]
        np.savez_compressed(os.path.join(GOLD, "rd_%s.npz" % name), width=w, height=h, qp=qp, yuv=yuv, labels=lab, bit_depth=bd, lf_across_tiles=1, tiles=np.array((1, 1)), wavefront=1, **({"tools": tools} if tools != rt.TOOLS_REFERENCE else {}),
                            records=dump["rec"].reshape(nf, nctu), rec_y=dump["rec_y"].reshape(nf, nctu, 4096),
                            rec_cb=dump["rec_cb"].reshape(nf, nctu, 1024), rec_cr=dump["rec_cr"].reshape(nf, nctu, 1024),
                            bitstream=np.frombuffer(bitstream, np.uint8), recon_filtered=np.frombuffer(recon, np.uint8), recon_deblocked=np.frombuffer(recon_dbk, np.uint8), bitstream_nosao=np.frombuffer(bitstream_nosao, np.uint8),
                            summary=np.array(summary), ctus_differing_from_the_default_cfg=differs)
        print("rd wavefront fixture", name, "ctus", nf * nctu, "records that differ from the run without the key:", differs, summary[0][:60] if summary else "")


def load_ref_model():
    import torch
    import torch.nn as nn
    src = open(os.path.join(REF, "use_model.py"), encoding="utf-8").read()
    cls = src[src.index("class ConvNet2"):src.index("DEVICE = ")]
    ns = {"torch": torch, "nn": nn}
    exec(cls, ns)
    model = ns["ConvNet2"]()
    sd = torch.load(os.path.join(REF, "rec", "hevc_encoder_model.pt"), map_location="cpu")
    model.load_state_dict(sd)
    assert model.training            # reference never calls .eval()
    return model, sd, src


def gen_cnn_eval():
    """F-cnn-1, second half ("... in reference (train-BN) mode and in eval mode"): the CTUs of cnn_f1.npz through a freshly
    loaded reference model after model.eval() -- BatchNorm then uses the checkpoint's running statistics (a model that has run in training
    mode has already moved them, hence the fresh load).  This is the selectable HEVCDL_BN_EVAL mode, not what the reference pipeline runs."""
    import torch
    model, _, _ = load_ref_model()
    model.eval()
    ctus = np.load(os.path.join(GOLD, "cnn_f1.npz"))["ctu_rgb"]
    logits = np.zeros((len(ctus), 4, 16), np.float32)
    with torch.no_grad():
        for i in range(len(ctus)):
            x = torch.from_numpy(ctus[i].astype(np.float32) / 255.0).permute(2, 0, 1)
            for q in range(4):
                ox, oy = (q % 2) * 32, (q // 2) * 32
                logits[i, q] = model(x[:, oy:oy + 32, ox:ox + 32].unsqueeze(0).contiguous(), x.unsqueeze(0))[0].numpy()
    import cnn_oracle
    labels = cnn_oracle.labels_from_logits(logits)          # post-processing pinned by cnn_f1 / cnn_f2
    np.savez_compressed(os.path.join(GOLD, "cnn_f1_eval.npz"), logits=logits, labels=labels)
    print("cnn eval fixture:", len(ctus), "CTUs; label histogram", np.bincount(labels.ravel(), minlength=4))


def gen_weights(sd):
    os.makedirs(WDIR, exist_ok=True)
    tensors, chunks, off = [], [], 0
    for k, v in sd.items():
        a = v.detach().cpu().numpy()
        if a.dtype != np.float32:      # num_batches_tracked (int64): unused in training-mode BN
            continue
        tensors.append({"name": k, "shape": list(a.shape), "offset": off})
        chunks.append(a.astype("<f4").ravel())
        off += a.size
    np.concatenate(chunks).tofile(os.path.join(WDIR, "hevc_encoder_model.f32"))
    json.dump({"source": "wolverinn/HEVC-deep-learning-pipeline rec/hevc_encoder_model.pt", "dtype": "f32le",
               "floats": off, "tensors": tensors}, open(os.path.join(WDIR, "hevc_encoder_model.json"), "w"), indent=1)
    print("weights:", off, "floats,", len(tensors), "tensors")


def ref_label_fn(src):
    """Label post-processing through the reference's own lines (use_model.py:101-119), driven with given outputs: -> f(logits [4,16]) -> 16 labels."""
    import torch
    lines = src.splitlines()
    start = next(i for i, l in enumerate(lines) if l.strip().startswith("pred = str(int(torch.argmax"))
    end = next(i for i, l in enumerate(lines) if "label[10],label[11],label[14],label[15]" in l)
    body = "\n".join(l[16:] if len(l) > 16 else l.strip() for l in lines[start:end + 1])
    code = compile(body, "use_model_101_119", "exec")

    def ref_labels(lg4):                           # lg4 [4,16] logits
        label = [str(i) for i in range(16)]
        for layer2 in range(4):
            ns = {"torch": torch, "output": torch.from_numpy(lg4[layer2:layer2 + 1].copy()), "layer2": layer2, "label": label}
            exec(code, ns)
        return [int(v) for v in label]
    return ref_labels


def gen_cnn_label_chain(src):
    """F-cnn-2, second set: uniformly random digits make a quadrant '0000' once in 256, so the `pred == "0000" and label[..] != "0"` chain of
    use_model.py:111-119 is hardly walked by cnn_f2.npz (43 of 10 000 sets start with label 0).  Here every quadrant is '0000' with probability 1/2, otherwise
    random digits: 20 000 tuples, half of the sets start with 0, all 16 zero / non-zero patterns of the four quadrants occur ~1 250 times each."""
    rng = np.random.default_rng(4321)
    ref_labels = ref_label_fn(src)
    m = 20000
    digits = rng.integers(0, 4, (m, 4, 4))
    digits[rng.integers(0, 2, (m, 4)) == 1] = 0
    fake = np.zeros((m, 4, 16), np.float32)
    for k in range(4):
        fake[:, :, 4 * k:4 * k + 4] = np.eye(4, dtype=np.float32)[digits[:, :, k]]
    lab = np.array([ref_labels(fake[i]) for i in range(m)], np.uint8)
    np.savez_compressed(os.path.join(GOLD, "cnn_f2b.npz"), digits=digits.astype(np.uint8), labels=lab)
    print("cnn label-chain fixture:", m, "tuples; first label 0 in", int((lab[:, 0] == 0).sum()), "; label histogram", np.bincount(lab.ravel(), minlength=4))


def gen_cnn(model, src):
    import torch
    rng = np.random.default_rng(1234)
    n = 256                                        # >= 256 CTUs
    ctus = np.zeros((n, 64, 64, 3), np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in range(n):
        k = i % 8
        if k == 0:
            ctus[i] = rng.integers(0, 256, (64, 64, 3))
        elif k == 1:
            base = 128 + 60 * np.sin(xx / (3.0 + i)) * np.cos(yy / (5.0 + i / 3))
            ctus[i] = np.clip(base[..., None] + rng.normal(0, 4, (64, 64, 3)), 0, 255)
        elif k == 2:
            ctus[i] = int(rng.integers(0, 256))
        elif k == 3:
            b = (rng.integers(0, 2, (8, 8)) * 200 + 20).repeat(8, 0).repeat(8, 1)
            ctus[i] = np.clip(b[..., None] + rng.integers(-10, 10, (64, 64, 3)), 0, 255)
        elif k == 4:
            b = (((xx * (1 + i % 5) + yy * 3) // 23) % 2) * 150 + 40
            ctus[i] = np.clip(b[..., None] + rng.normal(0, 2, (64, 64, 3)), 0, 255)
        elif k == 5:                               # picture-edge CTU: zero fill right/bottom (PIL crop)
            ctus[i] = np.clip(128 + 50 * np.sin((xx + yy) / 9.0)[..., None] + rng.normal(0, 6, (64, 64, 3)), 0, 255)
            ctus[i, :, 32 + (i % 3) * 8:] = 0
            ctus[i, 48:, :] = 0
        elif k == 6:
            b = (rng.integers(0, 2, (16, 16)) * 255).repeat(4, 0).repeat(4, 1)
            ctus[i] = b[..., None]
        else:
            g = np.clip(xx * 4 * (i % 3 == 0) + yy * 4 * (i % 3 != 0), 0, 255)
            ctus[i] = np.stack([g, 255 - g, (g // 2 + 60)], -1)
    logits = np.zeros((n, 4, 16), np.float32)
    with torch.no_grad():
        for i in range(n):
            x = torch.from_numpy(ctus[i].astype(np.float32) / 255.0).permute(2, 0, 1)   # ToTensor: HWC u8 -> CHW /255
            for q in range(4):
                ox, oy = (q % 2) * 32, (q // 2) * 32
                logits[i, q] = model(x[:, oy:oy + 32, ox:ox + 32].unsqueeze(0).contiguous(), x.unsqueeze(0))[0].numpy()
    ref_labels = ref_label_fn(src)
    labels = np.array([ref_labels(logits[i]) for i in range(n)], np.uint8)
    np.savez_compressed(os.path.join(GOLD, "cnn_f1.npz"), ctu_rgb=ctus, logits=logits, labels=labels)
    m = 10000                                      # F-cnn-2: >= 10 000 tuples
    digits = rng.integers(0, 4, (m, 4, 4))
    fake = np.zeros((m, 4, 16), np.float32)
    for k in range(4):
        fake[:, :, 4 * k:4 * k + 4] = np.eye(4, dtype=np.float32)[digits[:, :, k]]
    lab2 = np.array([ref_labels(fake[i]) for i in range(m)], np.uint8)
    np.savez_compressed(os.path.join(GOLD, "cnn_f2.npz"), digits=digits.astype(np.uint8), labels=lab2)
    print("cnn fixtures:", n, "CTUs,", m, "label tuples; label histogram", np.bincount(labels.ravel(), minlength=4))


def gen_cnn_pictures(model, src):
    """F-cnn-3: whole pictures through the reference's OWN frame loop -- use_model.py from `total_frames = ...` to the end of
    the file (lines 72-125: CTU count :80, raster order :86-87, quadrant origin :89-90, img.crop beyond the picture :91-92, ToTensor :93-94, the four
    forwards, argmax + fix-ups :101-119, one label file per CTU :121-125), exec()'d as it stands.  What the namespace supplies instead of the files the
    loop expects: `Image.open` hands back an in-memory PIL picture (the JPEG the reference reads is a lossy copy of exactly such a picture),
    `transforms.ToTensor` is torchvision's published rule for 8-bit RGB PIL pictures (HWC uint8 -> CHW float32 / 255; torchvision is not installed here),
    `model` records every output it returns; `os` is the real module in a scratch directory, so the label files are the loop's own.  Stored: the pictures,
    the labels read back from pred/<frame>/ctu<i>.txt and the logits of every (CTU, quadrant) forward.  Sizes: 416x240 (C1) and 200x136, both ragged on
    both edges; one colour and one grey picture each (a grey one can also enter the product's frame path through HEVCDL_CNN_INPUT_LUMA sample for sample)."""
    import tempfile
    import torch
    from PIL import Image as PILImage
    pics = []
    for w, h, seed in ((416, 240, 501), (200, 136, 502)):
        yuv = rt.synth_yuv(w, h, 1, seed)[0]
        import cnn_oracle
        rgb = cnn_oracle.yuv_to_rgb_picture(yuv, w, h, "rgb601")
        rng = np.random.default_rng(seed)
        rgb = np.clip(rgb.astype(np.int32) + rng.integers(-6, 7, rgb.shape), 0, 255).astype(np.uint8)      # decorrelate the three channels a little
        grey = np.repeat(yuv[:w * h].reshape(h, w)[..., None], 3, axis=2).astype(np.uint8)
        pics += [rgb, grey]
    lines = src.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("total_frames = "))
    code = compile("\n".join(lines[start:]), "use_model_72_125", "exec")
    recorded = []

    class Recorder:
        def __call__(self, a, b):
            out = model(a, b)
            recorded.append(out[0].detach().numpy().copy())
            return out

    class ImageShim:
        @staticmethod
        def open(path):
            n = int(re.search(r"(\d+)\.jpg$", path).group(1))
            return PILImage.fromarray(pics[n - 1], "RGB")

    class ToTensorShim:
        def __call__(self, pic):
            a = np.array(pic, np.uint8, copy=True)
            assert a.ndim == 3 and a.shape[2] == 3
            return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    class TransformsShim:
        ToTensor = ToTensorShim

    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="cnnf3_")
    try:
        os.chdir(tmp)
        os.makedirs("rec/frames")
        os.mkdir("pred")
        for n in range(len(pics)):
            open("rec/frames/%d.jpg" % (n + 1), "wb").close()          # the loop counts and removes these; it never reads them (Image.open is the shim)
        import math
        ns = {"torch": torch, "os": os, "math": math, "Image": ImageShim, "transforms": TransformsShim, "model": Recorder(),
              "DEVICE": torch.device("cpu"), "frame_tobe_encoded": str(len(pics))}
        exec(code, ns)
        out = {}
        pos = 0
        for n, pic in enumerate(pics):
            h, w = pic.shape[:2]
            nctu = ((w + 63) // 64) * ((h + 63) // 64)
            lab = np.array([[int(v) for v in open("pred/%d/ctu%d.txt" % (n, i)).read().split()] for i in range(nctu)], np.uint8)
            assert lab.shape == (nctu, 16) and not os.path.exists("rec/frames/%d.jpg" % (n + 1))
            out["rgb%d" % n], out["labels%d" % n] = pic, lab
            out["logits%d" % n] = np.array(recorded[pos:pos + 4 * nctu], np.float32).reshape(nctu, 4, 16)
            pos += 4 * nctu
        assert pos == len(recorded)
    finally:
        os.chdir(cwd)
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(GOLD, "cnn_f3.npz"), n_pictures=len(pics), **out)
    print("cnn picture fixture:", [(out["rgb%d" % n].shape, np.bincount(out["labels%d" % n].ravel(), minlength=4).tolist()) for n in range(len(pics))])


def gen_full():
    """One whole 1920x1080 frame through the reference (510 CTUs, labels = what the label CNN gives on this frame, computed by the numpy
    restatement of the CNN): the full-size pin of the decision path ("cross-check on large frames").  The input is
    regenerated from its seed (ref_tools.synth_yuv); stored are the labels, the reference's CTU records and checksums of its reconstruction."""
    import zlib
    import cnn_oracle
    w, h, qp, seed = 1920, 1080, 32, 77
    yuv = rt.synth_yuv(w, h, 1, seed)
    wts = cnn_oracle.load_weights(os.path.join(WDIR, "hevc_encoder_model.f32"))
    lab, _ = cnn_oracle.predict_labels(wts, yuv, w, h)
    dump, out, bitstream, recon = rt.run_reference(yuv, w, h, qp, lab)
    dump = dump[np.lexsort((dump["addr"], dump["frame"]))]
    nctu = lab.shape[1]
    assert len(dump) == nctu
    crc = np.array([[zlib.crc32(e[k].tobytes()) for k in ("rec_y", "rec_cb", "rec_cr")] for e in dump], np.uint32)
    np.savez_compressed(os.path.join(GOLD, "full_f1080_q32.npz"), width=w, height=h, qp=qp, seed=seed, labels=lab, records=dump["rec"].reshape(1, nctu), recon_crc32=crc,
                        summary=np.array([ln for ln in out.splitlines() if ln.startswith("POC")]))
    print("full-frame fixture: %d CTUs, label histogram %s" % (nctu, np.bincount(lab.ravel(), minlength=4)))


def gen_stage_traces():
    """F-rd-3: HM's own stage traces (TAppEncoder_trace of oracle/build_ref.sh = the reference with DEBUG_INTRA_SEARCH_COSTS and
    DEBUG_TRANSFORM_AND_QUANTISE on): every line of the mode search and every TU block on its way through transform / quantiser / dequantiser /
    inverse transform, in search order.  Inputs are regenerated from their seeds; stored are the labels and the parsed events
    (ref_tools.parse_reference_stage_trace)."""
    cases = [("stage_a64_q32", 64, 64, 32, 11, [[1, 1, 2, 2, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3, 2, 2]]),
             ("stage_b128_q27", 128, 64, 27, 12, [[2, 2, 3, 3, 2, 2, 3, 3, 0, 0, 0, 0, 0, 0, 0, 0], [3, 3, 1, 1, 3, 3, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]])]
    saved = rt.REF_ENC
    rt.REF_ENC = rt.STAGE_REF
    try:
        for name, w, h, qp, seed, labels in cases:
            yuv = rt.synth_yuv(w, h, 1, seed)
            lab = np.array([[rt.fixup_labels(l) for l in labels]], np.uint8)
            dump, out, _, _ = rt.run_reference(yuv, w, h, qp, lab)
            ev = rt.parse_reference_stage_trace(out)
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), width=w, height=h, qp=qp, seed=seed, labels=lab, **ev)
            print(name, "events", len(ev["kind"]), "by kind", np.bincount(ev["kind"], minlength=4), "block values", ev["blk"].size)
    finally:
        rt.REF_ENC = saved


def gen_bd_anchor():
    """F-rd-4: rate / PSNR points of the unpruned anchor (oracle/_ref/TAppEncoder_anchor) and of the reference as shipped (label files) on a
    small input at QP 22 / 27 / 32 / 37, with the BD figures of the reference's formulas: pins metrics.py and, on the GPU, the device path's
    rate / PSNR against the reference's own log lines."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ROOT)
    import bd_anchor
    import cnn_oracle
    import hevcdl_amd.metrics as metrics
    w, h, nf, seed = 384, 192, 2, 3000
    yuv = rt.synth_yuv(w, h, nf, seed)
    wts = cnn_oracle.load_weights(os.path.join(WDIR, "hevc_encoder_model.f32"))
    lab, _ = cnn_oracle.predict_labels(wts, yuv, w, h)
    base = tempfile.mkdtemp(prefix="bdfix_")
    pts = {}
    for name, binary in (("label_path", bd_anchor.REF), ("anchor", bd_anchor.ANCHOR)):
        pts[name] = [bd_anchor.curve_from_runs([bd_anchor.run_encoder(binary, yuv[i], lab[i], w, h, qp, base, "%s%d_%d" % (name, qp, i)) for i in range(nf)]) for qp in bd_anchor.QPS]
    an, lp = pts["anchor"], pts["label_path"]
    rate = lambda c: [p["kbps"] for p in c]
    psnr = lambda c: [p["psnr_y"] for p in c]
    np.savez_compressed(os.path.join(GOLD, "bd_anchor_small.npz"), width=w, height=h, frames=nf, seed=seed, qps=np.array(bd_anchor.QPS), labels=lab,
                        anchor_kbps=np.array(rate(an)), anchor_psnr_y=np.array(psnr(an)), label_kbps=np.array(rate(lp)), label_psnr_y=np.array(psnr(lp)),
                        label_bits=np.array([p["bits_per_frame"] for p in lp]), label_psnr_yuv=np.array([[p["psnr_y"], p["psnr_u"], p["psnr_v"]] for p in lp]),
                        bd_rate_percent=metrics.bd_rate(rate(an), psnr(an), rate(lp), psnr(lp)), bd_psnr_db=metrics.bd_psnr(rate(an), psnr(an), rate(lp), psnr(lp)))
    print("bd anchor fixture: BD-rate %.3f %%, BD-PSNR %.4f dB" % (metrics.bd_rate(rate(an), psnr(an), rate(lp), psnr(lp)), metrics.bd_psnr(rate(an), psnr(an), rate(lp), psnr(lp))))


def gen_bd():
    """Known-answer for a BD-rate script: values computed from the reference's
    calc_BDBR sample with its own bundled formula."""
    json.dump({"bd_psnr_db": -1.1922290103850435, "bd_rate_percent": 31.424376673861843,
               "source": "calc_BDBR/Bjontegaard-python3.zip:RatePsnrSample.txt"}, open(os.path.join(GOLD, "bd_known_answer.json"), "w"))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["rd", "rdtiles", "rd10", "rdx", "rdtools", "rdlf", "rdwpp", "cnn", "weights", "bd", "full", "bdanchor", "stage", "cnneval", "cnnpic", "cnnchain"]
    if "stage" in what:
        gen_stage_traces()
    if "rd" in what:
        gen_rd()
    if "rdtiles" in what:
        gen_rd(tiled=True)
    if "rd10" in what:
        gen_rd(ten_bit=True)
    if "rdx" in what:
        gen_rd(extreme=True)
    if "rdtools" in what:
        gen_rd_tools()
    if "rdlf" in what:
        gen_rd_lf()
    if "rdwpp" in what:
        gen_rd_wpp()
    if "cnn" in what or "weights" in what:
        model, sd, src = load_ref_model()
        if "weights" in what:
            gen_weights(sd)
        if "cnn" in what:
            gen_cnn(model, src)
    if "cnneval" in what:
        gen_cnn_eval()
    if "cnnpic" in what:
        model, sd, src = load_ref_model()
        gen_cnn_pictures(model, src)
    if "cnnchain" in what:
        gen_cnn_label_chain(load_ref_model()[2])
    if "bd" in what:
        gen_bd()
    if "full" in what:
        gen_full()
    if "bdanchor" in what:
        gen_bd_anchor()
