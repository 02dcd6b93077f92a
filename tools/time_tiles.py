"""RD-kernel throughput with and without tiles: python tools/time_tiles.py W H frames cols rows [bit_depth]"""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
W, H, nf, tc, tr = (int(v) for v in sys.argv[1:6])
bd = int(sys.argv[6]) if len(sys.argv) > 6 else 8
yuv = np.repeat(ref_tools.synth_yuv(W, H, 1, seed=1), nf, axis=0)
if bd == 10:
    yuv = yuv.astype(np.uint16) * 4 + np.random.default_rng(1).integers(0, 4, yuv.shape).astype(np.uint16)
for tiles in ((tc, tr),) if os.environ.get('TILED_ONLY') else ((1, 1), (tc, tr)):
    enc = hevcdl_amd.Encoder(W, H, 32, max_frames=nf, tiles=tiles, bit_depth=bd)
    enc.profile_enable(True)
    lab = enc.predict_depth(yuv)
    enc.profile_get()
    recs, recon, stats = enc.compress_frames(yuv, lab)
    pr = enc.profile_get()
    ct = lab.shape[0] * lab.shape[1]
    print("bit depth", bd, "tiles", tiles, "frames", nf, "ctus", ct, "rd_ms %.1f" % pr['rd_ms'], "RD CTU/s %.0f" % (ct / (pr['rd_ms'] / 1e3)), "bits/frame %.0f" % stats["est_bits"].mean(), flush=True)
    enc.close()
