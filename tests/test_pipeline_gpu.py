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


@pytest.mark.parametrize("w,h,nf,tiles,bd", [(192, 128, 5, "1x1", 8), (512, 128, 3, "2x2", 10), (256, 192, 4, "wavefront", 8)])
def test_sharded_encode_equals_single_process_and_cli(tmp_path, w, h, nf, tiles, bd):
    import hevcdl_amd
    import ref_tools
    wavefront = tiles == "wavefront"          # WaveFrontSynchro 1 instead of tiles: a sub-stream per CTU row
    tiles = "1x1" if wavefront else tiles
    yuv = ref_tools.synth_yuv(w, h, nf, seed=123)
    if bd == 10:
        yuv = yuv.astype(np.uint16) * 4 + np.random.default_rng(3).integers(0, 4, yuv.shape).astype(np.uint16)
    yuv.astype(np.uint8 if bd == 8 else "<u2").tofile(tmp_path / "in.yuv")
    script = os.path.join(ROOT, "tools", "encode_sharded.py")
    common = ["-i", "in.yuv", "-wdt", str(w), "-hgt", str(h), "-q", "30", "-f", str(nf), "--batch", "2", "--tiles", tiles, "--bit-depth", str(bd), "--hash"] + (["--wavefront"] if wavefront else [])
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
    if wavefront:
        extra += ["--WaveFrontSynchro=1"]
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


@pytest.mark.parametrize("w,h,bd,tiles", [(200, 136, 8, (1, 1)), (8, 8, 8, (1, 1)), (520, 200, 10, (2, 2)), (1928, 1080, 8, (1, 1))])
def test_device_entry_points_stay_inside_their_buffers(w, h, bd, tiles):
    """Bounds check of every device-pointer entry point (the image has no address sanitizer for device code): each output buffer sits between
    two 64 KiB guard areas filled with a pattern -- inputs likewise, so an out-of-range write of a neighbouring kernel would show -- and the
    guards must be untouched after CNN, decisions, deblocking and SAO on ragged picture sizes."""
    import torch
    import hevcdl_amd
    import ref_tools
    nf, G = 2, 65536
    dev = torch.device("cuda", 0)
    e = hevcdl_amd.Encoder(w, h, 30, max_frames=nf, tiles=tiles, bit_depth=bd)
    yuv = ref_tools.synth_yuv(w, h, nf, seed=91)
    if bd == 10:
        yuv = (yuv.astype(np.uint16) << 2) | 2
    sizes = {"yuv": e.frame_bytes * nf, "labels": e.ctus * 16 * nf, "records": e.ctus * 15120 * nf, "recon": e.frame_bytes * nf, "stats": 40 * nf,
             "dbk": e.frame_bytes * nf, "sao": e.ctus * np.dtype(hevcdl_amd.SAO_DTYPE).itemsize * 3 * nf, "final": e.frame_bytes * nf}
    bufs = {}
    for k, n in sizes.items():
        n = (n + 255) // 256 * 256
        t = torch.full((n + 2 * G,), 0xA5, dtype=torch.uint8, device=dev)
        bufs[k] = (t, n)
    ptr = lambda k: bufs[k][0].data_ptr() + G
    bufs["yuv"][0][G:G + sizes["yuv"]] = torch.from_numpy(np.ascontiguousarray(yuv).view(np.uint8).reshape(-1)).to(dev)
    e.predict_depth_dev(ptr("yuv"), nf, ptr("labels"))
    e.compress_frames_dev(ptr("yuv"), nf, ptr("labels"), ptr("records"), ptr("recon"), ptr("stats"))
    e.deblock_frames_dev(ptr("recon"), nf, ptr("records"), ptr("dbk"))
    e.sao_frames_dev(ptr("yuv"), ptr("dbk"), nf, ptr("sao"), ptr("final"))
    torch.cuda.synchronize()
    for k, (t, n) in bufs.items():
        assert bool((t[:G] == 0xA5).all()) and bool((t[G + n:] == 0xA5).all()), "guard area of %s overwritten" % k
        if k != "yuv":
            assert bool((t[G:G + sizes[k]] != 0xA5).any()), "%s not written" % k
    # and the results are the ones the host entry points give
    recs, recon, _ = e.compress_frames(yuv)
    assert bufs["records"][0][G:G + sizes["records"]].cpu().numpy().tobytes() == recs.tobytes()
    assert bufs["recon"][0][G:G + sizes["recon"]].cpu().numpy().tobytes() == np.ascontiguousarray(recon).tobytes()
    e.close()
