"""Does the label CNN of the next step fit into the tail of the decision kernel?  tools/overlap_probe.py [frames]: the two stages of one context on two streams,
serial (one stream) against concurrent (the CNN queued on a second stream right behind the decision kernel's launch)."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import hevcdl_amd, ref_tools
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
W, H = 3840, 2160
base = ref_tools.synth_yuv(W, H, 4, seed=4000)
enc = hevcdl_amd.Encoder(W, H, 32, max_frames=n)
hb = torch.from_numpy(np.ascontiguousarray(base.reshape(4, -1))).cuda()
yuv = torch.empty((n, hb.shape[1]), dtype=torch.uint8, device='cuda')
for f in range(n):
    yuv[f] = hb[f % 4]
ctus = 2040
lab = [torch.empty((n, ctus, 16), dtype=torch.uint8, device='cuda') for _ in range(2)]
recs = torch.empty((n, ctus, 15120), dtype=torch.uint8, device='cuda')
recon = torch.empty_like(yuv)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
enc.predict_depth_dev(yuv.data_ptr(), n, lab[0].data_ptr(), None, s1.cuda_stream)
enc.compress_frames_dev(yuv.data_ptr(), n, lab[0].data_ptr(), recs.data_ptr(), recon.data_ptr(), None, s1.cuda_stream)
torch.cuda.synchronize()
for mode in ("serial", "concurrent", "serial", "concurrent"):
    t0 = time.time()
    enc.compress_frames_dev(yuv.data_ptr(), n, lab[0].data_ptr(), recs.data_ptr(), recon.data_ptr(), None, s1.cuda_stream)
    enc.predict_depth_dev(yuv.data_ptr(), n, lab[1].data_ptr(), None, (s1 if mode == "serial" else s2).cuda_stream)
    torch.cuda.synchronize()
    print("%-10s decisions + CNN of %d frames: %.3f s" % (mode, n, time.time() - t0), flush=True)
assert torch.equal(lab[0], lab[1])
enc.close()
