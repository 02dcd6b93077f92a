"""oracle/ref_tools.py -- TEST INFRASTRUCTURE.

Helpers that (a) drive the reference encoder built by oracle/build_ref.sh (only available in the
container that has /root/reference) and (b) call the plain-C oracle (oracle/hm_oracle.c) through ctypes.
Nothing here is imported by the product package.
"""
import ctypes
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ENC = os.path.join(HERE, "_ref", "TAppEncoder_ref")
REF_DEC = os.path.join(HERE, "_ref", "TAppDecoder_ref")
REF_CFG = "/root/reference/encoder_intra_main.cfg"
ORACLE_SO = os.path.join(HERE, "_build", "libhm_oracle.so")

# hevcdl_ctu_record (include/hevcdl.h) as a numpy dtype: 15120 bytes
REC_DTYPE = np.dtype([
    ("depth", "u1", 256), ("part_size", "u1", 256), ("luma_dir", "u1", 256), ("chroma_dir", "u1", 256),
    ("tr_idx", "u1", 256), ("cbf", "u1", (3, 256)), ("tskip", "u1", (3, 256)),
    ("bits", "<u4"), ("dist", "<u4"), ("cost", "<f8"),
    ("coeff_y", "<i2", 4096), ("coeff_cb", "<i2", 1024), ("coeff_cr", "<i2", 1024)])
assert REC_DTYPE.itemsize == 15120
DUMP_DTYPE = np.dtype([("frame", "<i4"), ("addr", "<i4"), ("rec", REC_DTYPE),
                       ("rec_y", "u1", 4096), ("rec_cb", "u1", 1024), ("rec_cr", "u1", 1024)])
DUMP16_DTYPE = np.dtype([("frame", "<i4"), ("addr", "<i4"), ("rec", REC_DTYPE),
                         ("rec_y", "<u2", 4096), ("rec_cb", "<u2", 1024), ("rec_cr", "<u2", 1024)])      # runs with InternalBitDepth > 8
STATS_DTYPE = np.dtype([("sse", "<u8", 3), ("est_bits", "<u8"), ("ctus", "<u4"), ("pad", "<u4")])


def synth_yuv(width, height, n_frames, seed):
    """Synthetic 8-bit 4:2:0 frames (generator of SURVEY.md section 8d, reduced block size for small pictures)."""
    rng = np.random.default_rng(seed)
    out = []
    y, x = np.mgrid[0:height, 0:width]
    for f in range(n_frames):
        Y = 128 + 50 * np.sin(x / 57.0) * np.cos(y / 43.0) + 30 * np.sin((x + y + 3 * f) / 19.0) + rng.normal(0, 5, (height, width))
        bw, bh = max(16, width // 3), max(16, height // 3)
        bx, by = (width // 5 + 4 * f) % max(1, width - bw), height // 4
        blk = Y[by:by + bh, bx:bx + bw]
        Y[by:by + bh, bx:bx + bw] = np.floor(blk / 24) * 24
        Y = np.clip(Y, 0, 255).astype(np.uint8)
        xc, yc = x[::2, ::2], y[::2, ::2]
        U = np.clip(128 + 25 * np.sin(xc / 61.0) + rng.normal(0, 2, xc.shape), 0, 255).astype(np.uint8)
        V = np.clip(128 + 25 * np.cos(yc / 47.0) + rng.normal(0, 2, yc.shape), 0, 255).astype(np.uint8)
        out.append(np.concatenate([Y.ravel(), U.ravel(), V.ravel()]))
    return np.stack(out)


def min_depth_table(width, height):
    """Per CTU, per 16x16 cell: smallest depth at which the CU containing the cell lies inside the picture
    (SURVEY.md section 5 fact 2).  Cells outside the picture get 0."""
    cx, cy = (width + 63) // 64, (height + 63) // 64
    t = np.zeros((cy * cx, 16), np.uint8)
    for a in range(cx * cy):
        x0, y0 = (a % cx) * 64, (a // cx) * 64
        for c in range(16):
            px, py = x0 + (c % 4) * 16, y0 + (c // 4) * 16
            if px >= width or py >= height:
                continue
            for d in range(4):
                s = 64 >> d
                bx, by = px // s * s, py // s * s
                if bx + s <= width and by + s <= height:
                    t[a, c] = d
                    break
            else:
                t[a, c] = 3
    return t


def fixup_labels(lab):
    """Quadrant consistency rules of /root/reference/use_model.py:102-119 applied to a [16] label vector
    (after any clamping), so that the labels describe a valid quadtree."""
    lab = [int(v) for v in lab]
    quads = [(0, 1, 4, 5), (2, 3, 6, 7), (8, 9, 12, 13), (10, 11, 14, 15)]
    for qi, q in enumerate(quads):
        p = [lab[i] for i in q]
        if 0 in p and p != [0, 0, 0, 0]:
            p = [1 if v == 0 else v for v in p]
        if 1 in p and p != [1, 1, 1, 1]:
            p = [2 if v == 1 else v for v in p]
        if qi == 1 and p == [0, 0, 0, 0] and lab[0] != 0:
            p = [1, 1, 1, 1]
        if qi == 2 and p == [0, 0, 0, 0] and lab[2] != 0:
            p = [1, 1, 1, 1]
        if qi == 3 and p == [0, 0, 0, 0] and lab[8] != 0:
            p = [1, 1, 1, 1]
        for i, v in zip(q, p):
            lab[i] = v
    return lab


def clamp_labels(lab, md, inside=None):
    """Boundary policy for the 16 labels of one CTU (cnn_oracle.clamp_ctu_labels: raise to the in-picture depth, then make valid exactly what
    the reference's walk reads)."""
    import cnn_oracle
    if inside is None:
        inside = np.ones(16, bool)
    return cnn_oracle.clamp_ctu_labels(lab, md, inside)


def make_labels(width, height, n_frames, kind, seed=0):
    """kind: 0..3 = constant depth; 'rand' = random valid quadtrees.  Always clamped to the picture."""
    cx, cy = (width + 63) // 64, (height + 63) // 64
    import cnn_oracle
    md = min_depth_table(width, height)
    ins = cnn_oracle.inside_table(width, height)
    rng = np.random.default_rng(seed)
    labs = np.zeros((n_frames, cx * cy, 16), np.uint8)
    for f in range(n_frames):
        for a in range(cx * cy):
            lab = [0] * 16
            if kind == "rand":
                if rng.integers(0, 4) != 0:
                    for q in [(0, 1, 4, 5), (2, 3, 6, 7), (8, 9, 12, 13), (10, 11, 14, 15)]:
                        v = [1, 1, 1, 1] if rng.integers(0, 3) == 0 else [int(t) for t in rng.integers(2, 4, 4)]
                        for i, t in zip(q, v):
                            lab[i] = t
            else:
                lab = [int(kind)] * 16
            labs[f, a] = clamp_labels(lab, md[a], ins[a])
    return labs


def run_reference(yuv, width, height, qp, labels, extra_args=(), keep_dir=None, trace=False, bit_depth=8):
    """Run the reference encoder; returns (dump records array, stdout text, bitstream bytes, recon bytes).
    bit_depth 10: yuv holds 10-bit samples (uint16), coded with InputBitDepth = InternalBitDepth = 10, Profile main10."""
    n_frames = yuv.shape[0]
    d = keep_dir or tempfile.mkdtemp(prefix="hmref_")
    os.makedirs(os.path.join(d, "rec"), exist_ok=True)
    if os.path.isdir(os.path.join(d, "pred")):
        shutil.rmtree(os.path.join(d, "pred"))
    yuv.astype(np.uint8 if bit_depth == 8 else "<u2").tofile(os.path.join(d, "in.yuv"))
    for f in range(n_frames):
        os.makedirs(os.path.join(d, "pred", str(f)))
        for a in range(labels.shape[1]):
            with open(os.path.join(d, "pred", str(f), "ctu%d.txt" % a), "w") as fh:
                fh.write(" ".join(str(int(v)) for v in labels[f, a]) + " ")
    cfg = open(REF_CFG).read().replace(".\\rec\\", "rec/")
    open(os.path.join(d, "enc.cfg"), "w").write(cfg)
    open(os.path.join(d, "bs.cfg"), "w").write(
        "InputFile : in.yuv\nInputBitDepth : 8\nInputChromaFormat : 420\nFrameRate : 30\nFrameSkip : 0\n"
        "SourceWidth : %d\nSourceHeight : %d\nFramesToBeEncoded : %d\nLevel : 6.2\n" % (width, height, n_frames))
    env = dict(os.environ, HEVCDL_DUMP=os.path.join(d, "dump.bin"))
    if trace:
        env["HEVCDL_TRACE"] = os.path.join(d, "trace.txt")
    if os.path.exists(env["HEVCDL_DUMP"]):
        os.remove(env["HEVCDL_DUMP"])
    cmd = [REF_ENC, "-c", "enc.cfg", "-c", "bs.cfg", "-q", str(qp), "--SEIDecodedPictureHash=1"] + list(extra_args)
    if bit_depth != 8:
        cmd += ["--InputBitDepth=%d" % bit_depth, "--InternalBitDepth=%d" % bit_depth, "--Profile=main10"]
        env["HEVCDL_DUMP16"] = "1"
    p = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("reference encoder failed: " + p.stdout[-2000:] + p.stderr[-2000:])
    dump = np.fromfile(env["HEVCDL_DUMP"], dtype=DUMP_DTYPE if bit_depth == 8 else DUMP16_DTYPE)
    bitstream = open(os.path.join(d, "rec", "str.bin"), "rb").read()
    recon = open(os.path.join(d, "rec", "rec.yuv"), "rb").read()
    if keep_dir is None:
        shutil.rmtree(d)
    return dump, p.stdout, bitstream, recon


_lib = None


def oracle_lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(ORACLE_SO)
        _lib.hm_oracle_encode_frames.restype = ctypes.c_int
        _lib.hm_oracle_encode_frames.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib.hm_oracle_set_trace.argtypes = [ctypes.c_char_p]
    return _lib


def tile_args(tiles):
    """Reference command-line switches for tiles = (columns, rows) uniformly spaced, or ([widths], [heights]) in CTUs (all of them)."""
    if isinstance(tiles[0], (int, np.integer)):
        return ["--TileUniformSpacing=1", "--NumTileColumnsMinus1=%d" % (tiles[0] - 1), "--NumTileRowsMinus1=%d" % (tiles[1] - 1)]
    args = ["--TileUniformSpacing=0", "--NumTileColumnsMinus1=%d" % (len(tiles[0]) - 1), "--NumTileRowsMinus1=%d" % (len(tiles[1]) - 1)]
    if len(tiles[0]) > 1:
        args.append("--TileColumnWidthArray=" + " ".join(str(int(v)) for v in tiles[0][:-1]))
    if len(tiles[1]) > 1:
        args.append("--TileRowHeightArray=" + " ".join(str(int(v)) for v in tiles[1][:-1]))
    return args


def tile_bounds(tiles, width, height):
    """(columns, rows, int32 column boundaries, int32 row boundaries) in CTUs for either form of `tiles`."""
    cx, cy = (width + 63) // 64, (height + 63) // 64
    if isinstance(tiles[0], (int, np.integer)):
        c, r = int(tiles[0]), int(tiles[1])
        cb, rb = [(i * cx) // c for i in range(c + 1)], [(i * cy) // r for i in range(r + 1)]
    else:
        c, r = len(tiles[0]), len(tiles[1])
        cb, rb = [int(sum(tiles[0][:i])) for i in range(c + 1)], [int(sum(tiles[1][:i])) for i in range(r + 1)]
    return c, r, np.array(cb, np.int32), np.array(rb, np.int32)


TOOLS_REFERENCE = 0x7f
TOOL_TSKIP_FAST = 0x08
TOOL_RDOQ, TOOL_RDOQTS, TOOL_TSKIP, TOOL_SIGN_HIDE, TOOL_STRONG_INTRA, TOOL_FAST_UDI_MPM = 0x01, 0x02, 0x04, 0x10, 0x20, 0x40     # HEVCDL_TOOL_* of include/hevcdl.h


def tool_args(tools):
    """The reference's command-line switches for a tool mask (TAppEncCfg.cpp:900-901,917-918,950,978,1007)."""
    a = []
    if not tools & TOOL_RDOQ: a.append("--RDOQ=0")
    if not tools & TOOL_RDOQTS: a.append("--RDOQTS=0")
    if not tools & TOOL_TSKIP: a.append("--TransformSkip=0")
    if not tools & TOOL_TSKIP_FAST: a.append("--TransformSkipFast=0")
    if not tools & TOOL_SIGN_HIDE: a.append("--SignHideFlag=0")
    if not tools & TOOL_STRONG_INTRA: a.append("--StrongIntraSmoothing=0")
    if not tools & TOOL_FAST_UDI_MPM: a.append("--FastUDIUseMPMEnabled=0")
    return a


def run_oracle(yuv, width, height, qp, labels, trace_path=None, tiles=(1, 1), bit_depth=8, tools=TOOLS_REFERENCE, wpp=False):
    """Returns (records [frames][ctus] REC_DTYPE, recon [frames][w*h*3/2] (uint8, or uint16 for bit_depth 10), stats [frames])."""
    lib = oracle_lib()
    yuv = np.ascontiguousarray(yuv, np.uint8 if bit_depth == 8 else np.uint16)
    labels = np.ascontiguousarray(labels, np.uint8)
    n_frames, nctu = labels.shape[0], labels.shape[1]
    recs = np.zeros((n_frames, nctu), REC_DTYPE)
    recon = np.zeros_like(yuv)
    stats = np.zeros(n_frames, STATS_DTYPE)
    lib.hm_oracle_set_trace(trace_path.encode() if trace_path else None)
    lib.hm_oracle_set_tools(ctypes.c_uint(tools))
    lib.hm_oracle_set_wpp(1 if wpp else 0)              # WaveFrontSynchro 1: a sub-stream per CTU row, contexts synchronised with the row above
    lib.hm_oracle_encode_frames_tb.restype = ctypes.c_int
    lib.hm_oracle_encode_frames_tb.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    tc, tr, cb, rb = tile_bounds(tiles, width, height)
    rc = lib.hm_oracle_encode_frames_tb(yuv.ctypes.data, width, height, n_frames, qp, labels.ctypes.data,
                                        recs.ctypes.data, recon.ctypes.data, stats.ctypes.data, tc, tr, cb.ctypes.data, rb.ctypes.data, bit_depth)
    lib.hm_oracle_set_trace(None)
    lib.hm_oracle_set_tools(ctypes.c_uint(TOOLS_REFERENCE))
    lib.hm_oracle_set_wpp(0)
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    return recs, recon, stats


def run_deblock(recon, width, height, qp, recs, bit_depth=8, tiles=(1, 1), lf_across_tiles=True, lf_offsets=(0, 0)):
    """Oracle deblocking of pre-filter reconstructions [frames][w*h*3/2] given the records [frames][ctus] -> filtered copy.
    lf_across_tiles False = LFCrossTileBoundaryFlag 0: edges on the borders of `tiles` are left alone.  lf_offsets: (LoopFilterBetaOffset_div2, LoopFilterTcOffset_div2)."""
    lib = oracle_lib()
    lib.hm_oracle_set_deblock_offsets(int(lf_offsets[0]), int(lf_offsets[1]))
    try:
        return _run_deblock(lib, recon, width, height, qp, recs, bit_depth, tiles, lf_across_tiles)
    finally:
        lib.hm_oracle_set_deblock_offsets(0, 0)


def _run_deblock(lib, recon, width, height, qp, recs, bit_depth, tiles, lf_across_tiles):
    if not lf_across_tiles:
        lib.hm_oracle_deblock_frame16_tb.restype = ctypes.c_int
        lib.hm_oracle_deblock_frame16_tb.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        tc, tr, cb, rb = tile_bounds(tiles, width, height)
        dt = np.uint8 if bit_depth == 8 else np.uint16
        out = np.ascontiguousarray(np.asarray(recon, dt), np.uint16).copy().reshape(recs.shape[0], -1)
        recs = np.ascontiguousarray(recs)
        for f in range(out.shape[0]):
            if lib.hm_oracle_deblock_frame16_tb(out[f].ctypes.data, width, height, qp, recs[f].ctypes.data, bit_depth, tc, tr, cb.ctypes.data, rb.ctypes.data) != 0:
                raise RuntimeError("oracle deblock failed")
        return out.astype(dt)
    if bit_depth != 8:
        lib.hm_oracle_deblock_frame16.restype = ctypes.c_int
        lib.hm_oracle_deblock_frame16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        out = np.ascontiguousarray(recon, np.uint16).copy().reshape(recs.shape[0], -1)
        recs = np.ascontiguousarray(recs)
        for f in range(out.shape[0]):
            if lib.hm_oracle_deblock_frame16(out[f].ctypes.data, width, height, qp, recs[f].ctypes.data, bit_depth) != 0:
                raise RuntimeError("oracle deblock failed")
        return out
    lib.hm_oracle_deblock_frame.restype = ctypes.c_int
    lib.hm_oracle_deblock_frame.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = np.ascontiguousarray(recon, np.uint8).copy().reshape(recs.shape[0], -1)
    recs = np.ascontiguousarray(recs)
    for f in range(out.shape[0]):
        rc = lib.hm_oracle_deblock_frame(out[f].ctypes.data, width, height, qp, recs[f].ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle deblock failed rc=%d" % rc)
    return out


SAO_DTYPE = np.dtype([("mode", "<i4"), ("type", "<i4"), ("aux", "<i4"), ("offset", "<i4", 32)])      # hm_sao_offset


def run_sao(org, deblocked, width, height, qp, tiles=(1, 1), bit_depth=8, lf_across_tiles=True):
    """Oracle SAO of frames [n][w*h*3/2] -> (params [n][ctus][3] SAO_DTYPE, final reconstruction)."""
    lib = oracle_lib()
    lib.hm_oracle_sao_set_lf_across_tiles(1 if lf_across_tiles else 0)
    lib.hm_oracle_sao_frame16_tb.restype = ctypes.c_int
    lib.hm_oracle_sao_frame16_tb.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    dt = np.uint8 if bit_depth == 8 else np.uint16
    org16 = np.ascontiguousarray(np.asarray(org, dt).reshape(-1, width * height * 3 // 2), np.uint16)
    dbk16 = np.ascontiguousarray(np.asarray(deblocked, dt).reshape(org16.shape), np.uint16)
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    params = np.zeros((org16.shape[0], nctu, 3), SAO_DTYPE)
    out = np.zeros_like(org16)
    tc, tr, cb, rb = tile_bounds(tiles, width, height)
    for f in range(org16.shape[0]):
        rc = lib.hm_oracle_sao_frame16_tb(org16[f].ctypes.data, dbk16[f].ctypes.data, width, height, qp, params[f].ctypes.data, out[f].ctypes.data,
                                          tc, tr, cb.ctypes.data, rb.ctypes.data, bit_depth)
        if rc != 0:
            raise RuntimeError("oracle sao failed rc=%d" % rc)
    lib.hm_oracle_sao_set_lf_across_tiles(1)
    return params, out.astype(dt)


def ctu_recon_from_frame(recon_frame, width, height, addr):
    """Cut the 64x64 / 32x32 / 32x32 CTU blocks (zeros outside the picture) out of one planar frame."""
    cx = (width + 63) // 64
    x0, y0 = (addr % cx) * 64, (addr // cx) * 64
    Y = recon_frame[:width * height].reshape(height, width)
    U = recon_frame[width * height:width * height * 5 // 4].reshape(height // 2, width // 2)
    V = recon_frame[width * height * 5 // 4:].reshape(height // 2, width // 2)
    out = []
    for P, n, xx, yy in ((Y, 64, x0, y0), (U, 32, x0 // 2, y0 // 2), (V, 32, x0 // 2, y0 // 2)):
        b = np.zeros((n, n), recon_frame.dtype)
        sub = P[yy:yy + n, xx:xx + n]
        b[:sub.shape[0], :sub.shape[1]] = sub
        out.append(b.ravel())
    return out


FIELDS = ["depth", "part_size", "luma_dir", "chroma_dir", "tr_idx", "cbf", "tskip", "bits", "dist", "cost",
          "coeff_y", "coeff_cb", "coeff_cr"]


def compare(dump, recs, recon, width, height, verbose=True):
    """Compare reference dump with oracle output; returns number of mismatching (ctu, field) pairs."""
    bad = 0
    for e in dump:
        f, a = int(e["frame"]), int(e["addr"])
        r = recs[f, a]
        for k in FIELDS:
            if not np.array_equal(e["rec"][k], r[k]):
                bad += 1
                if verbose and bad <= 12:
                    ev, rv = np.asarray(e["rec"][k]), np.asarray(r[k])
                    if ev.ndim:
                        idx = np.argwhere(ev != rv)[:4].tolist()
                        print("MISMATCH frame %d ctu %d field %s at %s ref=%s ours=%s" % (f, a, k, idx, ev[ev != rv][:6], rv[ev != rv][:6]))
                    else:
                        print("MISMATCH frame %d ctu %d field %s ref=%s ours=%s" % (f, a, k, ev, rv))
        ry, ru, rv_ = ctu_recon_from_frame(recon[f], width, height, a)
        for k, ours in (("rec_y", ry), ("rec_cb", ru), ("rec_cr", rv_)):
            if not np.array_equal(e[k], ours):
                bad += 1
                if verbose and bad <= 12:
                    idx = np.argwhere(e[k] != ours)[:4].ravel().tolist()
                    print("MISMATCH frame %d ctu %d %s at %s" % (f, a, k, idx))
    return bad


# ---- stage traces ---------------------------------------------------------------------------------------------
# One event per line HM prints under DEBUG_INTRA_SEARCH_COSTS / DEBUG_TRANSFORM_AND_QUANTISE (oracle/build_ref.sh: TAppEncoder_trace), in the
# order of the search.  kind 0: "1st pass" line (a = mode, b = SATD, c = mode bits, cost); 1: "2nd pass" line = one candidate of the first RD loop
# (a = luma mode, cost); 2: a TU through transformNxN (a = size, b = component; blocks: residual, coefficients, levels); 3: a TU through
# invTransformNxN (blocks: levels, dequantised coefficients, residual).  Costs are the 6 significant digits std::cout prints.
STAGE_REF = os.path.join(os.path.dirname(REF_ENC), "TAppEncoder_trace")
_RE_P1 = re.compile(r"^1st pass mode (\d+) SAD = (\d+), mode bits = (\d+), cost = (\S+)$")
_RE_P2 = re.compile(r"^2nd pass \[luma,chroma\] mode \[(\d+),(\d+)\] cost = (\S+)$")
_RE_TU = re.compile(r"^(\d+): (\d+)x(\d+) channel (\d+) TU (at input to transform|between transform and quantiser|at output of quantiser|"
                    r"at input to dequantiser|between dequantiser and inverse-transform|at output of inverse-transform)$")
_TU_STEP = {"at input to transform": (2, 0), "between transform and quantiser": (2, 1), "at output of quantiser": (2, 2),
            "at input to dequantiser": (3, 0), "between dequantiser and inverse-transform": (3, 1), "at output of inverse-transform": (3, 2)}


def _pack_events(ev, blocks):
    off = np.zeros(len(ev) + 1, np.int64)
    for i, b in enumerate(blocks):
        off[i + 1] = off[i] + (0 if b is None else b.size)
    return {"kind": np.array([e[0] for e in ev], np.uint8), "a": np.array([e[1] for e in ev], np.int32), "b": np.array([e[2] for e in ev], np.int64),
            "c": np.array([e[3] for e in ev], np.int32), "cost": np.array([e[4] for e in ev], np.float64), "blk_off": off,
            "blk": np.concatenate([b for b in blocks if b is not None] or [np.zeros(0, np.int32)]).astype(np.int32)}


def parse_reference_stage_trace(text):
    lines = text.splitlines()
    ev, blocks, i = [], [], 0
    while i < len(lines):
        ln = lines[i]
        m = _RE_P1.match(ln)
        if m:
            ev.append((0, int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)))); blocks.append(None); i += 1; continue
        m = _RE_P2.match(ln)
        if m:
            ev.append((1, int(m.group(1)), int(m.group(2)), 0, float(m.group(3)))); blocks.append(None); i += 1; continue
        m = _RE_TU.match(ln)
        if m:
            n, comp = int(m.group(2)), int(m.group(4))
            kind, step = _TU_STEP[m.group(5)]
            vals = np.array([int(v) for r in lines[i + 1:i + 1 + n] for v in r.split()], np.int32)
            assert vals.size == n * n, (ln, vals.size)
            if step == 0:
                ev.append((kind, n, comp, 0, 0.0)); blocks.append(vals)
            else:
                assert ev[-1][:3] == (kind, n, comp), (ln, ev[-1])
                blocks[-1] = np.concatenate([blocks[-1], vals])
            i += 1 + n; continue
        i += 1
    return _pack_events(ev, blocks)


def parse_oracle_stage_trace(path):
    ev, blocks = [], []
    with open(path) as fh:
        lines = fh.read().splitlines()
    i = 0
    while i < len(lines):
        t = lines[i].split()
        if t[0] == "R":
            ev.append((0, int(t[1]), int(t[2]), int(t[3]), float("%g" % float(t[4])))); blocks.append(None); i += 1
        elif t[0] == "P":
            ev.append((1, int(t[1]), 0, 0, float("%g" % float(t[2])))); blocks.append(None); i += 1
        else:
            ev.append((2 if t[0] == "F" else 3, int(t[1]), int(t[2]), 0, 0.0))
            blocks.append(np.array(" ".join(lines[i + 1:i + 4]).split(), np.int64).astype(np.int32)); i += 4
    return _pack_events(ev, blocks)


def run_oracle_stage_trace(yuv, width, height, qp, labels):
    """The oracle's own events for the same input (records / reconstruction are returned as by run_oracle)."""
    lib = oracle_lib()
    lib.hm_oracle_set_stage_trace.argtypes = [ctypes.c_char_p]
    fd, path = tempfile.mkstemp(prefix="hm_stage_", suffix=".txt")
    os.close(fd)
    try:
        lib.hm_oracle_set_stage_trace(path.encode())
        out = run_oracle(yuv, width, height, qp, labels)
        lib.hm_oracle_set_stage_trace(None)
        return parse_oracle_stage_trace(path), out
    finally:
        lib.hm_oracle_set_stage_trace(None)
        os.remove(path)
