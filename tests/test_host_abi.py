"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol of include/hevcdl.h, computes the host-side
decision constants exactly like the reference, and fails loudly (no fallback) where it must.  No compute calls here."""
import ctypes
import math
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import hevcdl_amd
    hevcdl_amd.build_ext()
    return hevcdl_amd.load_library()


def test_every_declared_symbol_is_exported(lib):
    import hevcdl_amd
    hdr = open(os.path.join(ROOT, "include", "hevcdl.h")).read()
    declared = set(re.findall(r"\b(hevcdl_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hevcdl_ctx"}
    assert declared, "no declarations parsed"
    raw = ctypes.CDLL(hevcdl_amd.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "symbol %s declared in include/hevcdl.h is not exported" % name
    assert declared == set(hevcdl_amd.EXPORTS)


def test_struct_layouts_match_the_header(lib):
    import hevcdl_amd
    assert hevcdl_amd.REC_DTYPE.itemsize == 15120
    assert ctypes.sizeof(hevcdl_amd.Config) == hevcdl_amd.default_config(64, 64, 32).struct_size


@pytest.mark.parametrize("qp,lam,qpc,cw", [(22, 5.745, 22, 1.0), (27, 18.24, 27, 1.0), (32, 57.91, 31, 1.2599), (37, 183.9, 34, 2.0)])
def test_lambda_family_matches_reference_values(lib, qp, lam, qpc, cw):
    """lambda = 0.57*2^((QP-12)/3); chroma QP map; chroma weight 2^((QP-QPc)/3)."""
    import hevcdl_amd
    c = hevcdl_amd.default_config(1920, 1080, qp)
    assert c.lambda_ == 0.57 * 2.0 ** ((qp - 12) / 3.0)
    assert abs(c.lambda_ - lam) / lam < 1e-3
    assert c.sqrt_lambda == math.sqrt(c.lambda_)
    assert c.qp_chroma == qpc and abs(c.chroma_weight - cw) < 1e-4
    assert c.lambda_chroma == c.lambda_ / c.chroma_weight
    qs = [26214, 23302, 20560, 18396, 16384, 14564]
    for ch, q in ((0, qp), (1, qpc)):
        for l in range(4):
            ts = 15 - 8 - (l + 2)
            assert c.err_scale[ch][l] == (32768.0 * 2.0 ** (-2.0 * ts)) / qs[q % 6] / qs[q % 6] / 1
    assert c.tools == 0x7f and c.ctu_size == 64 and c.tu_log2_max == 5


def test_invalid_and_unsupported_configurations_are_rejected(lib):
    import hevcdl_amd
    cfg = hevcdl_amd.Config()
    assert lib.hevcdl_config_default(ctypes.byref(cfg), 100, 64, 32) == 1       # not a multiple of 8
    assert lib.hevcdl_config_default(ctypes.byref(cfg), 64, 64, 52) == 1        # QP out of range
    assert lib.hevcdl_config_default(None, 64, 64, 32) == 1
    cfg = hevcdl_amd.default_config(64, 64, 32)
    h = ctypes.c_void_p()
    w = hevcdl_amd.load_weights()
    assert lib.hevcdl_create(ctypes.byref(cfg), w.ctypes.data, w.size - 1, ctypes.byref(h)) == 1    # wrong blob size
    for tools in (0xff, 0x17f, 0x80000000):                                                         # bits that are no tool of the reference's cfg: rejected, not ignored
        cfg.tools = tools                                                                           # (each of the seven switches may be off: tests/golden/rd_k*)
        assert lib.hevcdl_create(ctypes.byref(cfg), w.ctypes.data, w.size, ctypes.byref(h)) == 2
    cfg = hevcdl_amd.default_config(64, 64, 32)
    cfg.bit_depth = 12                                                                              # 8 and 10 exist
    assert lib.hevcdl_create(ctypes.byref(cfg), w.ctypes.data, w.size, ctypes.byref(h)) == 2
    assert lib.hevcdl_config_default_bd(ctypes.byref(cfg), 64, 64, 32, 12) == 2
    c8, c10 = hevcdl_amd.default_config(64, 64, 32), hevcdl_amd.default_config(64, 64, 32, bit_depth=10)
    assert c10.bit_depth == 10 and c10.lambda_ == c8.lambda_ if hasattr(c8, "lambda_") else True
    # distortion stays at 8-bit scale (FULL_NBIT 0): error scale of a TU drops by 2^(2*2) for the distortion shift and rises by 2^(2*2) for
    # the transform shift -> unchanged; the sign-hiding factor gains 2^(2*2) from the QP offset of 12 and loses 2^4 from the distortion shift
    assert [list(r) for r in c10.err_scale] == [list(r) for r in c8.err_scale] and list(c10.sbh_rd_factor) == list(c8.sbh_rd_factor)
    assert lib.hevcdl_frame_bytes_bd(1920, 1080, 10) == 1920 * 1080 * 3
    assert lib.hevcdl_ctus_per_frame(3840, 2160) == 2040 and lib.hevcdl_ctus_per_frame(416, 240) == 28
    assert lib.hevcdl_frame_bytes(1920, 1080) == 1920 * 1080 * 3 // 2


def test_no_gpu_means_loud_failure_not_fallback(lib):
    """In a container without a GPU the product path must fail (NO_DEVICE), never route to a CPU implementation."""
    import torch
    import hevcdl_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hevcdl_amd.HevcdlError) as e:
        hevcdl_amd.Encoder(64, 64, 32)
    assert e.value.status == 3


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hevc-deep-learning-pipeline_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle/" not in txt.replace("oracle/ is test", "").replace("under\noracle/", "") or fn == "__init__.py", fn
                assert "hm_oracle" not in txt and "cnn_oracle" not in txt and "ref_tools" not in txt, fn


def test_weight_blob_matches_manifest():
    import json
    import hevcdl_amd
    man = json.load(open(os.path.splitext(hevcdl_amd.WEIGHTS_PATH)[0] + ".json"))
    assert man["floats"] == hevcdl_amd.WEIGHT_FLOATS == hevcdl_amd.load_weights().size
    names = [t["name"] for t in man["tensors"]]
    assert names[0] == "conv1.0.weight" and names[-1] == "conv64.1.running_var" and len(names) == 30


def test_k_slot_map_of_the_5x5_layers_covers_every_tap_once_without_bank_conflicts(tmp_path):
    """hevcdl_conv5_slot_tap (csrc/hevcdl_dev.h) is shared by the host's weight packing and the kernel's operand reads: all 75 taps exactly once in the 80 slots,
    the four words of a lane group in k-steps 0..3 are consecutive taps of one tile row, and the two lane groups of one half of a ds_read_b32 (slots a, a + 4) read
    words 16 LDS banks apart at both tile pitches -- the properties cnn_kernel.hip's conv5_mfma is built on."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = tmp_path / "slotmap.cpp"
    src.write_text('#include <cstdio>\n#include "hevcdl.h"\n#include "hevcdl_dev.h"\nint main() { for (int s = 0; s < 80; s++) printf("%d\\n", hevcdl_conv5_slot_tap(s)); return 0; }\n')
    exe = tmp_path / "slotmap"
    subprocess.run(["g++", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "hevc-deep-learning-pipeline_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert len(vals) == 80
    real = sorted(v for v in vals if v >= 0)
    assert real == list(range(75)), "every tap exactly once"
    tap = [v if v >= 0 else -v - 1 for v in vals]
    for slot in range(64):
        if slot & 3:
            assert tap[slot] == tap[slot & ~3] + (slot & 3), "k-steps 0..3: four consecutive words of a row"
    for row, ch in ((72, 36 * 72 + 16), (40, 36 * 40 + 16)):          # T64 / T32 pitches of cnn_kernel.hip
        def off(t):
            return (t // 25) * ch + ((t % 25) // 5) * row + t % 5
        for slot in range(80):
            if not (slot >> 2) & 1:
                assert (off(tap[slot + 4]) - off(tap[slot])) % 32 == 16, (slot, tap[slot], tap[slot + 4])


def test_issue_bound_file_is_keyed_to_the_shipped_kernel_and_adds_up():
    """bench.py reports `roofline.issue` (the instruction-issue bound of the decision kernel) and `roofline.traffic` from profiles/r*_issue.json / r*_traffic.json while their
    hash equals that of the shipped csrc/rd_kernel.hip: the newest committed files must carry that hash (a kernel edit without a new counter pass makes the line say "not
    reported"), and the ceiling must follow from the counters the way tools/issue_json.py states (256 CUs x 4 SIMDs x 2.4 GHz / (VALU wave-instructions per CTU x 4 cycles))."""
    import glob
    import hashlib
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    sha = hashlib.sha256(open(bench.RD_KERNEL_SRC, "rb").read()).hexdigest()[:16]
    issue = json.load(open(sorted(glob.glob(os.path.join(root, "profiles", "r*_issue.json")))[-1]))
    traffic = json.load(open(sorted(glob.glob(os.path.join(root, "profiles", "r*_traffic.json")))[-1]))
    assert issue["rd_kernel_sha16"] == sha and traffic["rd_kernel_sha16"] == sha
    for frames in ("600", "2560"):
        sh = issue["shapes"][frames]
        assert abs(sh["ceiling_ctus_per_s"] - 256 * 4 * 2.4e9 / (sh["valu_per_ctu"] * 4)) < 1.0
        assert 0.2 < sh["frac_under_counters"] < 1.0 and 16 < sh["lanes_enabled"] <= 64
    got = bench.measured_issue(600, 234000.0)
    assert got["frac"] is not None and abs(got["frac"] - 234000.0 / issue["shapes"]["600"]["ceiling_ctus_per_s"]) < 1e-9 and got["valu_per_ctu"] > 5e5
    assert bench.measured_issue(123, 1.0)["frac"] is None                   # no counter pass on that launch shape: said so, not guessed
    t, src = bench.measured_traffic(1224000)
    assert t is not None and t > 27408 * 1224000
