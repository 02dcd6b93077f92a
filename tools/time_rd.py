"""Decision-kernel timing at 2160p (or WxH): tools/time_rd.py frames [frames ...] [--size WxH] [--flags=N] [--wavefront]; labels from the on-device CNN."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import hevcdl_amd, ref_tools
args = [a for a in sys.argv[1:] if not a.startswith('--')]
W, H = 3840, 2160
for a in sys.argv[1:]:
    if a.startswith('--size='):
        W, H = [int(v) for v in a[7:].split('x')]
flags = 0
for a in sys.argv[1:]:
    if a.startswith('--flags='):
        flags = int(a[8:])          # hevcdl_config.exec_flags: 1 independent form only, 2 / 4 the ten- / eight-wave build of the kernel
wavefront = '--wavefront' in sys.argv[1:]      # WaveFrontSynchro 1: CTU rows as units of the launch
tools = 0x7f
for a in sys.argv[1:]:
    if a.startswith('--tools='):
        tools = int(a[8:], 0)        # hevcdl_config.tools: a mask other than 0x7f runs the build of the kernel that reads the switches at run time (csrc/rd_kernel_tools.hip)
counts = [int(a) for a in args] or [1, 75, 600]
nmax = max(counts)
base = ref_tools.synth_yuv(W, H, 4, seed=4000)
cfg = hevcdl_amd.default_config(W, H, 32, max_frames=nmax, wavefront=wavefront, tools=tools)
cfg.exec_flags = flags
enc = hevcdl_amd.Encoder(W, H, 32, cfg=cfg)
fb = hevcdl_amd.frame_bytes(W, H) if hasattr(hevcdl_amd, 'frame_bytes') else W * H * 3 // 2
yuv = torch.empty((nmax, fb), dtype=torch.uint8, device='cuda')
hb = torch.from_numpy(np.ascontiguousarray(base.reshape(4, -1))).cuda()
for f in range(nmax):
    yuv[f] = hb[f % 4]
ctus = ((W + 63) // 64) * ((H + 63) // 64)
labels = torch.empty((nmax, ctus, 16), dtype=torch.uint8, device='cuda')
recs = torch.empty((nmax, ctus, 15120), dtype=torch.uint8, device='cuda')
recon = torch.empty_like(yuv)
stats = torch.zeros((nmax, 40), dtype=torch.uint8, device='cuda')
enc.predict_depth_dev(yuv.data_ptr(), nmax, labels.data_ptr())
torch.cuda.synchronize()
enc.compress_frames_dev(yuv.data_ptr(), min(counts), labels.data_ptr(), recs.data_ptr(), recon.data_ptr(), stats.data_ptr())      # (workspace allocation, code load)
torch.cuda.synchronize()
for n in counts:
    t0 = time.time()
    enc.compress_frames_dev(yuv.data_ptr(), n, labels.data_ptr(), recs.data_ptr(), recon.data_ptr(), stats.data_ptr())
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("flags %d  frames %5d  rd %.3f s  %.1f CTU/s  %s" % (flags, n, dt, n * ctus / dt, enc.last_rd_launch()), flush=True)
enc.close()
