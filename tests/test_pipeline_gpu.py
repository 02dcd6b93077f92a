"""GPU: the whole-sequence pipeline (pipeline.py) -- stand-alone == two ranks sharding the frames (gloo for the summary gather, both on
GPU 0) == the C++ command line, byte for byte (bitstream with picture-hash SEI, reconstruction), and the reference decoder accepts it."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DEC = os.path.join(ROOT, "oracle", "_ref", "TAppDecoder_ref")


@pytest.mark.parametrize("w,h,nf,tiles,bd", [(192, 128, 5, "1x1", 8), (512, 128, 3, "2x2", 10)])
def test_sharded_encode_equals_single_process_and_cli(tmp_path, w, h, nf, tiles, bd):
    import hevcdl_amd
    import ref_tools
    yuv = ref_tools.synth_yuv(w, h, nf, seed=123)
    if bd == 10:
        yuv = yuv.astype(np.uint16) * 4 + np.random.default_rng(3).integers(0, 4, yuv.shape).astype(np.uint16)
    yuv.astype(np.uint8 if bd == 8 else "<u2").tofile(tmp_path / "in.yuv")
    script = os.path.join(ROOT, "tools", "encode_sharded.py")
    common = ["-i", "in.yuv", "-wdt", str(w), "-hgt", str(h), "-q", "30", "-f", str(nf), "--batch", "2", "--tiles", tiles, "--bit-depth", str(bd), "--hash"]
    r1 = subprocess.run([sys.executable, script] + common + ["-b", "one.bin", "-o", "one.yuv"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                         script] + common + ["-b", "two.bin", "-o", "two.yuv", "--backend", "gloo"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    assert (tmp_path / "one.bin").read_bytes() == (tmp_path / "two.bin").read_bytes()
    assert (tmp_path / "one.yuv").read_bytes() == (tmp_path / "two.yuv").read_bytes()
    lines = lambda t: [l for l in t.splitlines() if l.startswith("POC") or l.startswith("\t ")]
    assert lines(r1.stdout) == lines(r2.stdout) and len(lines(r1.stdout)) == nf + 1
    # the C++ front end on the same input
    app = hevcdl_amd.build_app()
    tc, tr = (int(v) for v in tiles.split("x"))
    extra = ["--TileUniformSpacing=1", "--NumTileColumnsMinus1=%d" % (tc - 1), "--NumTileRowsMinus1=%d" % (tr - 1), "--SEIDecodedPictureHash=1", "--Level=6.2"]
    if bd == 10:
        extra += ["--InputBitDepth=10", "--InternalBitDepth=10", "--Profile=main10"]
    r3 = subprocess.run([app, "-i", "in.yuv", "-wdt", str(w), "-hgt", str(h), "-q", "30", "-f", str(nf), "-b", "cli.bin", "-o", "cli.yuv"] + extra, cwd=tmp_path,
                        capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, r3.stdout[-2000:] + r3.stderr[-2000:]
    assert (tmp_path / "cli.bin").read_bytes() == (tmp_path / "one.bin").read_bytes() and (tmp_path / "cli.yuv").read_bytes() == (tmp_path / "one.yuv").read_bytes()
    if os.path.exists(REF_DEC):          # only in the container that built the reference
        r4 = subprocess.run([REF_DEC, "-b", "one.bin", "-o", "dec.yuv"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r4.returncode == 0 and "ERROR" not in r4.stdout and r4.stdout.count("(OK)") == nf
        assert (tmp_path / "dec.yuv").read_bytes() == (tmp_path / "one.yuv").read_bytes()


@pytest.mark.parametrize("w,h,nf,tiles,bd", [(512, 192, 3, "2x3", 8), (520, 136, 2, "2x2", 10)])
def test_tile_sharded_encode_equals_single_process(tmp_path, w, h, nf, tiles, bd):
    """--shard tiles: the two ranks decide different tiles of every picture, exchange them, the owner filters and writes: same files."""
    import ref_tools
    yuv = ref_tools.synth_yuv(w, h, nf, seed=321)
    if bd == 10:
        yuv = yuv.astype(np.uint16) * 4 + np.random.default_rng(4).integers(0, 4, yuv.shape).astype(np.uint16)
    yuv.astype(np.uint8 if bd == 8 else "<u2").tofile(tmp_path / "in.yuv")
    script = os.path.join(ROOT, "tools", "encode_sharded.py")
    common = ["-i", "in.yuv", "-wdt", str(w), "-hgt", str(h), "-q", "33", "-f", str(nf), "--batch", "2", "--tiles", tiles, "--bit-depth", str(bd), "--hash"]
    r1 = subprocess.run([sys.executable, script] + common + ["-b", "one.bin", "-o", "one.yuv"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                         script] + common + ["-b", "two.bin", "-o", "two.yuv", "--backend", "gloo", "--shard", "tiles"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    assert (tmp_path / "one.bin").read_bytes() == (tmp_path / "two.bin").read_bytes()
    assert (tmp_path / "one.yuv").read_bytes() == (tmp_path / "two.yuv").read_bytes()
    lines = lambda t: [l.rsplit(" [ET", 1)[0] for l in t.splitlines() if l.startswith("POC") or l.startswith("\t ")]
    assert lines(r1.stdout) == lines(r2.stdout) and len(lines(r1.stdout)) == nf + 1
