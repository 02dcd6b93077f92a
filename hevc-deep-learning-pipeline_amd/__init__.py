"""hevc-deep-learning-pipeline_amd -- host-side mirror (Python, ctypes) of the C ABI in include/hevcdl.h.

The compute path is the hand-written HIP library csrc/*.hip -> lib/libhevcdl_hip.so (gfx950).  There is NO CPU
fallback: if the library is missing or no GPU is visible, every entry point raises.  The CPU oracle under
oracle/ is test infrastructure and is never imported from here.

Reference interfaces this replaces:
  * TEncCu::compressCtu loop of TEncSlice::compressSlice (HM_dl/source/Lib/TLibEncoder/TEncSlice.cpp:792-983,
    TEncCu.cpp:234-287)                                 -> Encoder.compress_frames / encode_frames_dev
  * python gen_frames.py + python use_model.py sidecar  -> Encoder.predict_depth
"""
import ctypes
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("HEVCDL_LIB") or os.path.join(PKG_DIR, "lib", "libhevcdl_hip.so")
TRACE_LIB_PATH = os.path.join(PKG_DIR, "lib", "libhevcdl_hip_trace.so")      # -DHEVCDL_STAGE_TRACE build, loaded by tests/test_rd_gpu.py only
WEIGHTS_PATH = os.path.join(PKG_DIR, "weights", "hevc_encoder_model.f32")
WEIGHT_FLOATS = 637712
SOURCES = ["cnn_kernel.hip", "fc_kernel.hip", "rd_kernel.hip", "rd_kernel_bd10.hip", "rd_kernel_wide.hip", "rd_kernel_tools.hip", "deblock_kernel.hip", "sao_kernel.hip", "hevcdl_api.hip", "hevcdl_bitstream.cpp"]

STATUS = {0: "OK", 1: "INVALID_ARG", 2: "UNSUPPORTED", 3: "NO_DEVICE", 4: "HIP", 5: "OOM"}

REC_DTYPE = np.dtype([
    ("depth", "u1", 256), ("part_size", "u1", 256), ("luma_dir", "u1", 256), ("chroma_dir", "u1", 256),
    ("tr_idx", "u1", 256), ("cbf", "u1", (3, 256)), ("tskip", "u1", (3, 256)),
    ("bits", "<u4"), ("dist", "<u4"), ("cost", "<f8"),
    ("coeff_y", "<i2", 4096), ("coeff_cb", "<i2", 1024), ("coeff_cr", "<i2", 1024)])
STATS_DTYPE = np.dtype([("sse", "<u8", 3), ("est_bits", "<u8"), ("ctus", "<u4"), ("pad", "<u4")])
SAO_DTYPE = np.dtype([("mode", "<i4"), ("type", "<i4"), ("aux", "<i4"), ("offset", "<i4", 32)])       # hevcdl_sao_offset; a CTU has 3 (Y, Cb, Cr)
CABAC_DTYPE = np.dtype([("ctx", "u1", 160), ("frac", "<u8")])      # hevcdl_cabac_state
assert REC_DTYPE.itemsize == 15120 and STATS_DTYPE.itemsize == 40 and CABAC_DTYPE.itemsize == 168


class HevcdlError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("hevcdl status %s%s" % (STATUS.get(status, status), (": " + msg) if msg else ""))
        self.status = status


class Planes(ctypes.Structure):
    """hevcdl_planes of include/hevcdl.h."""
    _fields_ = [("plane", ctypes.c_void_p * 3), ("row_stride", ctypes.c_size_t * 3), ("frame_stride", ctypes.c_size_t * 3), ("sample_bytes", ctypes.c_int32)]


class Config(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("bit_depth", ctypes.c_int32), ("chroma_format", ctypes.c_int32), ("qp", ctypes.c_int32),
                ("ctu_size", ctypes.c_int32), ("max_partition_depth", ctypes.c_int32),
                ("tu_log2_min", ctypes.c_int32), ("tu_log2_max", ctypes.c_int32), ("tu_max_depth_intra", ctypes.c_int32),
                ("tools", ctypes.c_uint32), ("bn_mode", ctypes.c_int32), ("boundary_policy", ctypes.c_int32),
                ("cnn_input", ctypes.c_int32), ("device", ctypes.c_int32), ("max_frames", ctypes.c_int32),
                ("lambda_", ctypes.c_double), ("sqrt_lambda", ctypes.c_double), ("chroma_weight", ctypes.c_double),
                ("lambda_chroma", ctypes.c_double), ("err_scale", (ctypes.c_double * 4) * 2),
                ("sbh_rd_factor", ctypes.c_int64 * 2), ("qp_chroma", ctypes.c_int32),
                ("tile_columns", ctypes.c_int32), ("tile_rows", ctypes.c_int32), ("exec_flags", ctypes.c_int32),
                ("tile_uniform_spacing", ctypes.c_int32), ("tile_column_width", ctypes.c_int32 * 19), ("tile_row_height", ctypes.c_int32 * 21),
                ("lf_across_tiles", ctypes.c_int32), ("lf_beta_offset_div2", ctypes.c_int32), ("lf_tc_offset_div2", ctypes.c_int32), ("wavefront", ctypes.c_int32)]


def tile_layout(tiles, width, height):
    """tiles = (columns, rows): uniformly spaced; or ([w0, w1, ...], [h0, h1, ...]): explicit sizes in CTUs of EVERY tile column / row
    (TileUniformSpacing 0; the last one must take exactly the rest).  -> (columns, rows, uniform, column boundaries, row boundaries)."""
    cx, cy = (width + 63) // 64, (height + 63) // 64
    if isinstance(tiles[0], (int, np.integer)):
        c, r = int(tiles[0]), int(tiles[1])
        return c, r, 1, [(i * cx) // c for i in range(c + 1)], [(i * cy) // r for i in range(r + 1)]
    cw, rh = [int(v) for v in tiles[0]], [int(v) for v in tiles[1]]
    if sum(cw) != cx or sum(rh) != cy:
        raise ValueError("explicit tile sizes must add up to the picture: %d CTU columns, %d CTU rows" % (cx, cy))
    return len(cw), len(rh), 0, [sum(cw[:i]) for i in range(len(cw) + 1)], [sum(rh[:i]) for i in range(len(rh) + 1)]


def _set_tiles(cfg, tiles, width, height):
    c, r, uniform, cb, rb = tile_layout(tiles, width, height)
    cfg.tile_columns, cfg.tile_rows, cfg.tile_uniform_spacing = c, r, uniform
    for i in range(min(c - 1, 19)):
        cfg.tile_column_width[i] = cb[i + 1] - cb[i]
    for i in range(min(r - 1, 21)):
        cfg.tile_row_height[i] = rb[i + 1] - rb[i]


class StreamConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("qp", ctypes.c_int32),
                ("level_idc", ctypes.c_int32), ("sao_enabled", ctypes.c_int32), ("loop_filter_disable", ctypes.c_int32),
                ("tile_columns", ctypes.c_int32), ("tile_rows", ctypes.c_int32), ("bit_depth", ctypes.c_int32),
                ("tile_uniform_spacing", ctypes.c_int32), ("tile_column_width", ctypes.c_int32 * 19), ("tile_row_height", ctypes.c_int32 * 21),
                ("lf_across_tiles", ctypes.c_int32), ("tools", ctypes.c_uint32), ("lf_beta_offset_div2", ctypes.c_int32), ("lf_tc_offset_div2", ctypes.c_int32),
                ("rewrite_param_sets", ctypes.c_int32), ("wavefront", ctypes.c_int32)]


class Profile(ctypes.Structure):
    _fields_ = [("cnn_ms", ctypes.c_double), ("rd_ms", ctypes.c_double), ("cnn_launches", ctypes.c_uint32), ("rd_launches", ctypes.c_uint32), ("cnn_conv_ms", ctypes.c_double)]


def _includes(path, seen):
    """The project-local files `path` includes (transitively): the dependency set of one object."""
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for ln in open(path, encoding="utf-8", errors="replace"):
        ln = ln.strip()
        if ln.startswith("#include \""):
            name = ln.split('"')[1]
            for d in (os.path.dirname(path), os.path.join(ROOT, "include"), os.path.join(PKG_DIR, "csrc")):
                if os.path.exists(os.path.join(d, name)):
                    _includes(os.path.join(d, name), seen)
                    break
    return seen


def build_ext(force=False, verbose=False, defines=(), out=None, extra_flags=(), jobs=None):
    """Compile every HIP source for gfx950 into lib/libhevcdl_hip.so (hipcc cross-compiles without a GPU).  One object per source under build/<variant>/,
    recompiled only when the source or a header it includes is newer, the stale ones in parallel (the decision kernel is three translation units of ~80 s each)."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(PKG_DIR, "csrc", s) for s in SOURCES]
    out = out or os.path.join(PKG_DIR, "lib", "libhevcdl_hip.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG_DIR, "csrc")] + ["-D" + d for d in defines] + list(extra_flags)
    import hashlib
    variant = os.path.splitext(os.path.basename(out))[0] + "-" + hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    odir = os.path.join(PKG_DIR, "build", variant)
    os.makedirs(odir, exist_ok=True)
    # a library newer than every source and every header a source includes is current, whether or not its objects are here (build/ does not travel to the
    # GPU box, the built lib/*.so does: without this a fresh box would recompile every translation unit, or fail where hipcc is absent)
    deps = set()
    for src in srcs:
        _includes(src, deps)
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        tag = out + ".flags"
        if not os.path.exists(tag) or open(tag).read() == variant:
            return out
    objs, stale = [], []
    for src in srcs:
        obj = os.path.join(odir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _includes(src, set())):
            stale.append((src, obj))
    if not stale and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(o) for o in objs):
        return out

    def compile_one(job):
        cmd = [hipcc] + flags + ["-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=jobs or min(len(stale), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(compile_one, stale))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
    with open(out + ".flags", "w") as f:       # which flag set built `out` (the fast path above must not take a library of other -D's for this one)
        f.write(variant)
    return out


APP_PATH = os.path.join(PKG_DIR, "bin", "TAppEncoderHevcdl")


def build_app(force=False):
    """Compile the command-line front end (csrc/hevcdl_app.cpp, host C++ over the C ABI) into bin/TAppEncoderHevcdl."""
    src = os.path.join(PKG_DIR, "csrc", "hevcdl_app.cpp")
    lib = build_ext()
    if not force and os.path.exists(APP_PATH) and os.path.getmtime(APP_PATH) >= max(os.path.getmtime(src), os.path.getmtime(lib)):
        return APP_PATH
    os.makedirs(os.path.dirname(APP_PATH), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "-O2", "-std=c++17", "-x", "c++", src, "-x", "none", "-I" + os.path.join(ROOT, "include"), "-L" + os.path.dirname(lib), "-lhevcdl_hip",
                    "-Wl,-rpath,$ORIGIN/../lib", "-pthread", "-ldl", "-o", APP_PATH], check=True)
    return APP_PATH


_lib = None


def load_library():
    """Load the HIP library; fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(this package has no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.hevcdl_config_default.argtypes = [ctypes.POINTER(Config), ci, ci, ci]
    lib.hevcdl_create.argtypes = [ctypes.POINTER(Config), vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.hevcdl_destroy.argtypes = [vp]
    lib.hevcdl_destroy.restype = None
    lib.hevcdl_last_error.argtypes = [vp]
    lib.hevcdl_last_error.restype = ctypes.c_char_p
    lib.hevcdl_predict_depth.argtypes = [vp, vp, ci, vp, vp]
    lib.hevcdl_predict_depth_rgb.argtypes = [vp, vp, ci, vp, vp]
    lib.hevcdl_labels_from_logits.argtypes = [vp, vp, ci, ci, vp]
    lib.hevcdl_compress_frames.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    lib.hevcdl_predict_depth_planes.argtypes = [vp, vp, ci, vp, vp]
    lib.hevcdl_compress_frames_planes.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    lib.hevcdl_stream_config_default.argtypes = [ctypes.POINTER(StreamConfig), ci, ci, ci]
    lib.hevcdl_access_unit_bound.argtypes = [ci, ci]
    lib.hevcdl_access_unit_bound.restype = ctypes.c_size_t
    lib.hevcdl_write_access_unit.argtypes = [ctypes.POINTER(StreamConfig), ci, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    lib.hevcdl_sao_frames.argtypes = [vp, vp, vp, ci, vp, vp]
    lib.hevcdl_sao_frames_dev.argtypes = [vp, vp, vp, ci, vp, vp, vp]
    lib.hevcdl_deblock_frames.argtypes = [vp, vp, ci, vp, vp]
    lib.hevcdl_deblock_frames_dev.argtypes = [vp, vp, ci, vp, vp, vp]
    lib.hevcdl_begin_frames.argtypes = [vp, vp, ci, vp, vp]
    lib.hevcdl_compress_ctu.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.hevcdl_get_recon.argtypes = [vp, ci, vp]
    lib.hevcdl_predict_depth_dev.argtypes = [vp, vp, ci, vp, vp, vp]
    lib.hevcdl_clamp_labels_dev.argtypes = [vp, vp, ci, vp]
    lib.hevcdl_compress_frames_dev.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp]
    lib.hevcdl_encode_frames_dev.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp]
    lib.hevcdl_compress_tiles_dev.argtypes = [vp, vp, ci, vp, vp, vp, vp, ci, ci, vp]
    lib.hevcdl_encode_pictures.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, vp]
    lib.hevcdl_reserve_workspace.argtypes = [vp]
    lib.hevcdl_last_rd_launch.argtypes = [vp]
    lib.hevcdl_last_rd_launch.restype = ctypes.c_char_p
    lib.hevcdl_profile_enable.argtypes = [vp, ci]
    lib.hevcdl_profile_get.argtypes = [vp, ctypes.POINTER(Profile)]
    lib.hevcdl_ctus_per_frame.argtypes = [ci, ci]
    lib.hevcdl_frame_bytes.argtypes = [ci, ci]
    lib.hevcdl_frame_bytes.restype = ctypes.c_size_t
    lib.hevcdl_frame_bytes_bd.argtypes = [ci, ci, ci]
    lib.hevcdl_frame_bytes_bd.restype = ctypes.c_size_t
    lib.hevcdl_config_default_bd.argtypes = [ctypes.POINTER(Config), ci, ci, ci, ci]
    _lib = lib
    return lib


EXPORTS = ["hevcdl_config_default", "hevcdl_create", "hevcdl_destroy", "hevcdl_last_error", "hevcdl_predict_depth",
           "hevcdl_predict_depth_rgb", "hevcdl_labels_from_logits", "hevcdl_compress_frames", "hevcdl_predict_depth_planes", "hevcdl_compress_frames_planes", "hevcdl_predict_depth_dev", "hevcdl_compress_frames_dev",
           "hevcdl_encode_frames_dev", "hevcdl_compress_tiles_dev", "hevcdl_clamp_labels_dev", "hevcdl_device_memory", "hevcdl_host_alloc", "hevcdl_host_free", "hevcdl_encode_pictures", "hevcdl_encode_pictures_chunked", "hevcdl_profile_enable", "hevcdl_profile_get", "hevcdl_last_rd_launch", "hevcdl_reserve_workspace", "hevcdl_ctus_per_frame", "hevcdl_frame_bytes", "hevcdl_frame_bytes_bd", "hevcdl_config_default_bd",
           "hevcdl_begin_frames", "hevcdl_compress_ctu", "hevcdl_get_recon", "hevcdl_deblock_frames", "hevcdl_deblock_frames_dev",
           "hevcdl_sao_frames", "hevcdl_sao_frames_dev", "hevcdl_stream_config_default", "hevcdl_access_unit_bound", "hevcdl_write_access_unit", "hevcdl_write_picture_hash_sei", "hevcdl_picture_md5", "hevcdl_write_digest_sei", "hevcdl_picture_hash", "hevcdl_write_hash_sei"]


def picture_hash_sei(width, height, picture, bit_depth=8, method=1):
    """Suffix SEI NAL with the hash of the three planes of `picture` (the final reconstruction) -> bytes.  method: SEIDecodedPictureHash 1 MD5, 2 CRC, 3 checksum."""
    if method != 1:
        lib = load_library()
        cfg = StreamConfig()
        if lib.hevcdl_stream_config_default(ctypes.byref(cfg), width, height, 32) != 0:
            raise HevcdlError(1, "stream config")
        cfg.bit_depth = bit_depth
        pic = np.ascontiguousarray(picture, np.uint8 if bit_depth == 8 else np.dtype("<u2")).reshape(-1)
        dg = np.zeros(48, np.uint8); pb = ctypes.c_int(0); buf = np.zeros(128, np.uint8); n = ctypes.c_size_t(0)
        lib.hevcdl_picture_hash.argtypes = [ctypes.POINTER(StreamConfig), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        lib.hevcdl_write_hash_sei.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        st = lib.hevcdl_picture_hash(ctypes.byref(cfg), pic.ctypes.data, method, dg.ctypes.data, ctypes.byref(pb))
        if st == 0:
            st = lib.hevcdl_write_hash_sei(method, dg.ctypes.data, buf.ctypes.data, 128, ctypes.byref(n))
        if st != 0:
            raise HevcdlError(st, "picture hash SEI")
        return buf[:n.value].tobytes()
    lib = load_library()
    cfg = StreamConfig()
    if lib.hevcdl_stream_config_default(ctypes.byref(cfg), width, height, 32) != 0:
        raise HevcdlError(1, "stream config")
    cfg.bit_depth = bit_depth
    pic = np.ascontiguousarray(picture, np.uint8 if bit_depth == 8 else np.dtype("<u2")).reshape(-1)
    if pic.size != width * height * 3 // 2:
        raise ValueError("picture must hold width * height * 3 / 2 samples")
    buf = np.zeros(128, np.uint8)
    n = ctypes.c_size_t(0)
    lib.hevcdl_write_picture_hash_sei.argtypes = [ctypes.POINTER(StreamConfig), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    st = lib.hevcdl_write_picture_hash_sei(ctypes.byref(cfg), pic.ctypes.data, buf.ctypes.data, 128, ctypes.byref(n))
    if st != 0:
        raise HevcdlError(st, "write_picture_hash_sei")
    return buf[:n.value].tobytes()


def load_weights(path=WEIGHTS_PATH):
    w = np.fromfile(path, dtype="<f4")
    if w.size != WEIGHT_FLOATS:
        raise ValueError("weight blob must hold %d floats, got %d" % (WEIGHT_FLOATS, w.size))
    return w


EXEC_NO_UNIT_HANDOVER, EXEC_RD_WIDE, EXEC_RD_NARROW = 1, 2, 4      # HEVCDL_EXEC_* (hevcdl_config.exec_flags)
TOOLS_REFERENCE = 0x7f
TOOL_RDOQ, TOOL_RDOQTS, TOOL_TSKIP, TOOL_TSKIP_FAST, TOOL_SIGN_HIDE, TOOL_STRONG_INTRA, TOOL_FAST_UDI_MPM = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40      # HEVCDL_TOOL_* (include/hevcdl.h): each may be turned off


def default_config(width, height, qp, max_frames=1, device=0, cnn_input=0, tiles=(1, 1), bit_depth=8, lf_across_tiles=True, bn_mode=0, tools=TOOLS_REFERENCE, lf_offsets=(0, 0), wavefront=False):
    lib = load_library()
    cfg = Config()
    st = lib.hevcdl_config_default_bd(ctypes.byref(cfg), width, height, qp, bit_depth)
    if st:
        raise HevcdlError(st, "hevcdl_config_default(%d,%d,%d)" % (width, height, qp))
    cfg.max_frames, cfg.device, cfg.cnn_input = max_frames, device, cnn_input
    cfg.bn_mode = bn_mode                      # 0: training-mode BatchNorm as the reference runs it; 1 (HEVCDL_BN_EVAL): the checkpoint's running statistics
    _set_tiles(cfg, tiles, width, height)        # (columns, rows) uniformly spaced, or explicit sizes: see tile_layout
    cfg.lf_across_tiles = 1 if lf_across_tiles else 0                   # LFCrossTileBoundaryFlag
    cfg.tools = tools                          # cfg keys TransformSkip / SignHideFlag / StrongIntraSmoothing / FastUDIUseMPMEnabled
    cfg.lf_beta_offset_div2, cfg.lf_tc_offset_div2 = lf_offsets      # LoopFilterBetaOffset_div2, LoopFilterTcOffset_div2
    cfg.wavefront = 1 if wavefront else 0      # WaveFrontSynchro: CTU rows start from the contexts behind the second CTU of the row above (and run as units of their own)
    return cfg


def write_access_unit(width, height, qp, poc, records, level_idc=186, sao=None, tiles=(1, 1), bit_depth=8, lf_across_tiles=True, tools=TOOLS_REFERENCE, lf_offsets=(0, 0), lf_disable=False,
                      rewrite_param_sets=True, wavefront=False):
    """Host-side bitstream writer (no GPU): VPS+SPS+PPS+slice NAL of one picture from its CTU records -> bytes."""
    lib = load_library()
    cfg = StreamConfig()
    st = lib.hevcdl_stream_config_default(ctypes.byref(cfg), width, height, qp)
    if st != 0:
        raise HevcdlError(st, "stream config")
    cfg.level_idc = level_idc
    _set_tiles(cfg, tiles, width, height)
    cfg.lf_across_tiles = 1 if lf_across_tiles else 0
    cfg.bit_depth = bit_depth
    cfg.tools = tools
    cfg.lf_beta_offset_div2, cfg.lf_tc_offset_div2 = lf_offsets
    cfg.loop_filter_disable = 1 if lf_disable else 0
    cfg.rewrite_param_sets = 1 if rewrite_param_sets else 0
    cfg.wavefront = 1 if wavefront else 0      # a sub-stream per CTU row, entry points in the slice header
    sao_ptr = None
    if sao is not None:
        sao = np.ascontiguousarray(sao, SAO_DTYPE)
        cfg.sao_enabled = 1
        sao_ptr = sao.ctypes.data
    records = np.ascontiguousarray(records)
    cap = lib.hevcdl_access_unit_bound(width, height)
    buf = np.zeros(cap, np.uint8)
    n = ctypes.c_size_t(0)
    st = lib.hevcdl_write_access_unit(ctypes.byref(cfg), int(poc), records.ctypes.data, sao_ptr, buf.ctypes.data, cap, ctypes.byref(n))
    if st != 0:
        raise HevcdlError(st, "write_access_unit")
    return buf[:n.value].tobytes()


class Encoder:
    """One context per device.  Frames are planar 8-bit 4:2:0, numpy [n_frames, w*h*3/2] uint8."""

    def __init__(self, width, height, qp, max_frames=1, device=0, cnn_input=0, weights=None, cfg=None, tiles=(1, 1), bit_depth=8, lf_across_tiles=True, bn_mode=0, tools=TOOLS_REFERENCE, lf_offsets=(0, 0), wavefront=False):
        """bit_depth 10: every yuv / recon array of the decision path holds uint16 samples (frame_bytes counts bytes)."""
        self.lib = load_library()
        self.cfg = cfg or default_config(width, height, qp, max_frames, device, cnn_input, tiles, bit_depth, lf_across_tiles, bn_mode, tools, lf_offsets, wavefront)
        self.bit_depth = self.cfg.bit_depth
        self.sample_dtype = np.uint8 if self.bit_depth == 8 else np.dtype("<u2")
        self.tiles = (self.cfg.tile_columns, self.cfg.tile_rows)
        self.width, self.height, self.qp = self.cfg.width, self.cfg.height, self.cfg.qp
        self.ctus = self.lib.hevcdl_ctus_per_frame(self.width, self.height)
        self.frame_bytes = self.lib.hevcdl_frame_bytes_bd(self.width, self.height, self.bit_depth)
        self.frame_samples = self.width * self.height * 3 // 2
        w = np.ascontiguousarray(load_weights() if weights is None else weights, dtype="<f4")
        self._h = ctypes.c_void_p()
        st = self.lib.hevcdl_create(ctypes.byref(self.cfg), w.ctypes.data, w.size, ctypes.byref(self._h))
        if st:
            self._h = None
            raise HevcdlError(st, "hevcdl_create")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.hevcdl_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, st):
        if st:
            raise HevcdlError(st, (self.lib.hevcdl_last_error(self._h) or b"").decode())

    def _frames(self, yuv):
        yuv = np.ascontiguousarray(yuv, self.sample_dtype).reshape(-1, self.frame_samples)
        return yuv, yuv.shape[0]

    def predict_depth(self, yuv, want_logits=False):
        yuv, n = self._frames(yuv)
        labels = np.zeros((n, self.ctus, 16), np.uint8)
        logits = np.zeros((n, self.ctus, 4, 16), np.float32) if want_logits else None
        self._check(self.lib.hevcdl_predict_depth(self._h, yuv.ctypes.data, n, labels.ctypes.data, logits.ctypes.data if want_logits else None))
        return (labels, logits) if want_logits else labels

    def labels_from_logits(self, logits, clamp=False):
        """logits [n,4,16] float32 -> labels [n,16]: the device's label stage alone (use_model.py:101-119; clamp: + the boundary policy)."""
        logits = np.ascontiguousarray(logits, np.float32)
        n = logits.shape[0]
        assert logits.shape == (n, 4, 16)
        labels = np.zeros((n, 16), np.uint8)
        self._check(self.lib.hevcdl_labels_from_logits(self._h, logits.ctypes.data, n, int(bool(clamp)), labels.ctypes.data))
        return labels

    def predict_depth_rgb(self, ctu_rgb):
        ctu_rgb = np.ascontiguousarray(ctu_rgb, np.uint8).reshape(-1, 64, 64, 3)
        n = ctu_rgb.shape[0]
        labels = np.zeros((n, 16), np.uint8)
        logits = np.zeros((n, 4, 16), np.float32)
        self._check(self.lib.hevcdl_predict_depth_rgb(self._h, ctu_rgb.ctypes.data, n, labels.ctypes.data, logits.ctypes.data))
        return labels, logits

    def _planes(self, y, u, v):
        """Three arrays [frames][rows][cols] (views with any row / frame pitch, uint8 or 16-bit samples) -> (hevcdl_planes, n_frames, keep-alive)."""
        arrs = [np.asarray(a) for a in (y, u, v)]
        arrs = [a[None] if a.ndim == 2 else a for a in arrs]
        sb = arrs[0].dtype.itemsize
        pl = Planes()
        for c, a in enumerate(arrs):
            if a.ndim != 3 or a.dtype.itemsize != sb or a.strides[2] != sb or a.shape[1:] != ((self.height, self.width) if c == 0 else (self.height // 2, self.width // 2)):
                raise ValueError("plane %d: expected [frames][%d][%d] with unit column stride" % (c, self.height if c == 0 else self.height // 2, self.width if c == 0 else self.width // 2))
            pl.plane[c], pl.row_stride[c], pl.frame_stride[c] = a.ctypes.data, a.strides[1], a.strides[0]
        pl.sample_bytes = sb
        return pl, arrs[0].shape[0], arrs

    def predict_depth_planes(self, y, u, v):
        pl, n, keep = self._planes(y, u, v)
        labels = np.zeros((n, self.ctus, 16), np.uint8)
        self._check(self.lib.hevcdl_predict_depth_planes(self._h, ctypes.byref(pl), n, labels.ctypes.data, None))
        return labels

    def compress_frames_planes(self, y, u, v, labels=None):
        """compress_frames for pictures held as three planes with their own pitch (e.g. views into an encoder's padded picture buffers)."""
        pl, n, keep = self._planes(y, u, v)
        recs = np.zeros((n, self.ctus), REC_DTYPE)
        recon = np.zeros((n, self.frame_bytes), np.uint8)
        stats = np.zeros(n, STATS_DTYPE)
        lab_ptr = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.uint8).reshape(n, self.ctus, 16)
            lab_ptr = labels.ctypes.data
        self._check(self.lib.hevcdl_compress_frames_planes(self._h, ctypes.byref(pl), n, lab_ptr, recs.ctypes.data, recon.ctypes.data, stats.ctypes.data))
        return recs, (recon.view(np.uint16) if self.bit_depth > 8 else recon), stats

    def compress_frames(self, yuv, labels=None):
        """-> (records [n, ctus] REC_DTYPE, recon [n, frame_bytes] uint8, stats [n] STATS_DTYPE)."""
        yuv, n = self._frames(yuv)
        recs = np.zeros((n, self.ctus), REC_DTYPE)
        recon = np.zeros_like(yuv)
        stats = np.zeros(n, STATS_DTYPE)
        lab_ptr = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.uint8).reshape(n, self.ctus, 16)
            lab_ptr = labels.ctypes.data
        self._check(self.lib.hevcdl_compress_frames(self._h, yuv.ctypes.data, n, lab_ptr, recs.ctypes.data, recon.ctypes.data, stats.ctypes.data))
        return recs, recon, stats

    def encode_pictures(self, yuv, labels=None, deblock=True, sao=True):
        """Whole picture pipeline in one call (the pictures stay in HBM between the stages) -> (records [n, ctus], output pictures [n, samples],
        SAO parameters [n, ctus, 3] or None, stats [n])."""
        yuv, n = self._frames(yuv)
        recs = np.zeros((n, self.ctus), REC_DTYPE)
        out = np.zeros_like(yuv)
        stats = np.zeros(n, STATS_DTYPE)
        params = np.zeros((n, self.ctus, 3), SAO_DTYPE) if sao else None
        lab_ptr = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.uint8).reshape(n, self.ctus, 16)
            lab_ptr = labels.ctypes.data
        self._check(self.lib.hevcdl_encode_pictures(self._h, yuv.ctypes.data, n, lab_ptr, int(bool(deblock)), recs.ctypes.data, out.ctypes.data,
                                                    params.ctypes.data if sao else None, stats.ctypes.data))
        return recs, out, params, stats

    def encode_pictures_chunked(self, yuv, labels=None, deblock=True, sao=True, chunk_frames=0):
        """The same pipeline with the results handed over chunk by chunk (hevcdl_encode_pictures_chunked): a generator-like list of
        (first, records [count, ctus], pictures [count, samples], SAO parameters or None, stats [count]) copies, one per chunk."""
        yuv, n = self._frames(yuv)
        lab_ptr = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.uint8).reshape(n, self.ctus, 16)
            lab_ptr = labels.ctypes.data
        chunks = []
        fn_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)

        def view(ptr, dtype, shape):
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer(ctypes.string_at(ptr, nbytes), dtype).reshape(shape).copy()

        def on_chunk(_user, first, count, recs, pics, sao_p, stats):
            chunks.append((first, view(recs, REC_DTYPE, (count, self.ctus)), view(pics, yuv.dtype, (count, yuv.shape[1])),
                           view(sao_p, SAO_DTYPE, (count, self.ctus, 3)) if sao_p else None, view(stats, STATS_DTYPE, (count,))))
            return 0
        cb = fn_t(on_chunk)
        self.lib.hevcdl_encode_pictures_chunked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, fn_t, ctypes.c_void_p]
        self._check(self.lib.hevcdl_encode_pictures_chunked(self._h, yuv.ctypes.data, n, lab_ptr, int(bool(deblock)), int(bool(sao)), int(chunk_frames), cb, None))
        return chunks

    # ---- deblocking filter (TComLoopFilter::loopFilterPic) ----
    def deblock_frames(self, recon, records):
        """recon [n, frame_bytes] (before the in-loop filters) + records [n, ctus] -> deblocked pictures."""
        recon, n = self._frames(recon)
        records = np.ascontiguousarray(records).reshape(n, self.ctus)
        out = np.zeros_like(recon)
        self._check(self.lib.hevcdl_deblock_frames(self._h, recon.ctypes.data, n, records.ctypes.data, out.ctypes.data))
        return out

    def sao_frames(self, org, deblocked):
        """original + deblocked frames -> (SAO parameters [n, ctus, 3] SAO_DTYPE, final reconstruction)."""
        org, n = self._frames(org)
        dbk, _ = self._frames(deblocked)
        params = np.zeros((n, self.ctus, 3), SAO_DTYPE)
        out = np.zeros_like(org)
        self._check(self.lib.hevcdl_sao_frames(self._h, org.ctypes.data, dbk.ctypes.data, n, params.ctypes.data, out.ctypes.data))
        return params, out

    def sao_frames_dev(self, d_org, d_deblocked, n, d_params, d_out, stream=None):
        self._check(self.lib.hevcdl_sao_frames_dev(self._h, d_org, d_deblocked, n, d_params, d_out, stream))

    def deblock_frames_dev(self, d_recon, n, d_records, d_out, stream=None):
        self._check(self.lib.hevcdl_deblock_frames_dev(self._h, d_recon, n, d_records, d_out, stream))

    # ---- per-CTU session: compressCtu + encodeCtu of the reference, one CTU per call (TEncSlice.cpp:879,893) ----
    def begin_frames(self, yuv, labels=None):
        """Upload frames, take / predict labels -> labels [n, ctus, 16] actually used."""
        yuv, n = self._frames(yuv)
        out = np.zeros((n, self.ctus, 16), np.uint8)
        lab_ptr = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, np.uint8).reshape(n, self.ctus, 16)
            lab_ptr = labels.ctypes.data
        self._check(self.lib.hevcdl_begin_frames(self._h, yuv.ctypes.data, n, lab_ptr, out.ctypes.data))
        return out

    def compress_ctu(self, frame, ctu_addr, state_in=None):
        """-> (record REC_DTYPE scalar array, cabac state after the CTU CABAC_DTYPE scalar array)."""
        rec = np.zeros(1, REC_DTYPE)
        st_out = np.zeros(1, CABAC_DTYPE)
        st_in = None
        if state_in is not None:
            st_in = np.ascontiguousarray(state_in, CABAC_DTYPE).reshape(1)
        self._check(self.lib.hevcdl_compress_ctu(self._h, frame, ctu_addr, None if st_in is None else st_in.ctypes.data, rec.ctypes.data, st_out.ctypes.data))
        return rec, st_out

    def get_recon(self, frame):
        out = np.zeros(self.frame_samples, self.sample_dtype)
        self._check(self.lib.hevcdl_get_recon(self._h, frame, out.ctypes.data))
        return out

    # ---- device-resident entry points: arguments are raw device pointers (ints), e.g. torch.Tensor.data_ptr() ----
    def predict_depth_dev(self, d_yuv, n, d_labels, d_logits=None, stream=None):
        self._check(self.lib.hevcdl_predict_depth_dev(self._h, d_yuv, n, d_labels, d_logits, stream))

    def clamp_labels_dev(self, d_labels, n, stream=None):
        """Caller-made labels on the device: boundary clamp + quadtree consistency in place (hevcdl_clamp_labels_dev)."""
        self._check(self.lib.hevcdl_clamp_labels_dev(self._h, d_labels, n, stream))

    def compress_frames_dev(self, d_yuv, n, d_labels, d_records, d_recon, d_stats=None, stream=None):
        self._check(self.lib.hevcdl_compress_frames_dev(self._h, d_yuv, n, d_labels, d_records, d_recon, d_stats, stream))

    def compress_tiles_dev(self, d_yuv, n, d_labels, d_records, d_recon, d_stats, tile_begin, tile_count, stream=None):
        """Tiles [tile_begin, tile_begin + tile_count) of every frame only (whole-frame device buffers): see sharding.py."""
        self._check(self.lib.hevcdl_compress_tiles_dev(self._h, d_yuv, n, d_labels, d_records, d_recon, d_stats, tile_begin, tile_count, stream))

    def encode_frames_dev(self, d_yuv, n, d_labels, d_records, d_recon, d_stats=None, stream=None):
        self._check(self.lib.hevcdl_encode_frames_dev(self._h, d_yuv, n, d_labels, d_records, d_recon, d_stats, stream))

    def profile_enable(self, on=True):
        self._check(self.lib.hevcdl_profile_enable(self._h, int(on)))

    def reserve_workspace(self):
        """Allocate the decision kernel's largest workspace now (hevcdl_reserve_workspace): raises HevcdlError(OOM) here instead of at the first launch."""
        self._check(self.lib.hevcdl_reserve_workspace(self._h))

    def last_rd_launch(self):
        """Build of the decision kernel and launch form of the last decision launch (hevcdl_last_rd_launch)."""
        return (self.lib.hevcdl_last_rd_launch(self._h) or b"").decode()

    def profile_get(self):
        p = Profile()
        self._check(self.lib.hevcdl_profile_get(self._h, ctypes.byref(p)))
        return {"cnn_ms": p.cnn_ms, "rd_ms": p.rd_ms, "cnn_launches": p.cnn_launches, "rd_launches": p.rd_launches, "cnn_conv_ms": p.cnn_conv_ms}
