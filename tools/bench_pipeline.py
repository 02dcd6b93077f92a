#!/usr/bin/env python3
"""End-to-end pictures per second of the whole encoder (file -> CNN -> decisions -> deblocking -> SAO -> bitstream with picture hash -> files)
on one GPU: python tools/bench_pipeline.py [frames] [WxH]   (not the headline metric: that is bench.py's CTU decisions per second)"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.cuda.init()
    import bench
    import hevcdl_amd.pipeline as pipeline
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    w, h = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "3840x2160").split("x"))
    d = tempfile.mkdtemp(prefix="hevcdl_e2e_")
    yuv = bench.synth_frames_torch(torch, torch.device("cuda", 0), w, h, list(range(nf)), seed=4000).cpu().numpy()
    yuv.tofile(os.path.join(d, "in.yuv"))
    del yuv
    for tiles in ((1, 1), (4, 2)):
        t0 = time.time()
        summ, rows = pipeline.encode_sequence(os.path.join(d, "in.yuv"), w, h, 32, nf, os.path.join(d, "out.bin"), os.path.join(d, "rec.yuv"),
                                              batch=nf, tiles=tiles, hash_sei=True, log=lambda *a: None)
        dt = time.time() - t0
        print(json.dumps({"stage": "pipeline", "size": "%dx%d" % (w, h), "frames": nf, "tiles": list(tiles), "seconds": dt, "pictures_per_s": nf / dt,
                          "ctus_per_s": nf * (((w + 63) // 64) * ((h + 63) // 64)) / dt, "kbps": summ.bitrate_kbps(), "psnr_y": summ.averages()[0]}), flush=True)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)


if __name__ == "__main__":
    main()
