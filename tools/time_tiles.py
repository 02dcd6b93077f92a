"""RD-kernel throughput with and without tiles: python tools/time_tiles.py W H frames cols rows"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
W, H, nf, tc, tr = (int(v) for v in sys.argv[1:6])
yuv = np.repeat(ref_tools.synth_yuv(W, H, 1, seed=1), nf, axis=0)
for tiles in ((1, 1), (tc, tr)):
    enc = hevcdl_amd.Encoder(W, H, 32, max_frames=nf, tiles=tiles)
    enc.profile_enable(True)
    lab = enc.predict_depth(yuv)
    enc.profile_get()
    recs, recon, stats = enc.compress_frames(yuv, lab)
    pr = enc.profile_get()
    ct = lab.shape[0] * lab.shape[1]
    print("tiles", tiles, "frames", nf, "ctus", ct, "rd_ms %.1f" % pr['rd_ms'], "RD CTU/s %.0f" % (ct / (pr['rd_ms'] / 1e3)), "bits/frame %.0f" % stats["est_bits"].mean(), flush=True)
    enc.close()
