"""Label-CNN timing at 2160p: tools/time_cnn.py [frames] [repeats] [file to save 8 frames of logits / labels]; prints the convolution kernel's and the whole stage's HIP-event times and a checksum of
the labels and logits (A/B two builds through HEVCDL_LIB: the checksums must agree)."""
import sys, zlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import hevcdl_amd, ref_tools
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H = 3840, 2160
base = ref_tools.synth_yuv(W, H, 4, seed=4000)
enc = hevcdl_amd.Encoder(W, H, 32, max_frames=n)
hb = torch.from_numpy(np.ascontiguousarray(base.reshape(4, -1))).cuda()
yuv = torch.empty((n, hb.shape[1]), dtype=torch.uint8, device='cuda')
for f in range(n):
    yuv[f] = hb[f % 4]
ctus = ((W + 63) // 64) * ((H + 63) // 64)
labels = torch.empty((n, ctus, 16), dtype=torch.uint8, device='cuda')
logits = torch.empty((n, ctus, 4, 16), dtype=torch.float32, device='cuda')
enc.predict_depth_dev(yuv.data_ptr(), n, labels.data_ptr(), logits.data_ptr())
torch.cuda.synchronize()
if len(sys.argv) > 3:
    np.save(sys.argv[3], logits[:8].cpu().numpy()); np.save(sys.argv[3] + ".labels", labels[:8].cpu().numpy())
print("labels crc %08x  logits crc %08x" % (zlib.crc32(labels[:4].cpu().numpy().tobytes()), zlib.crc32(logits[:4].cpu().numpy().tobytes())))
enc.profile_enable(True)
for r in range(reps):
    enc.predict_depth_dev(yuv.data_ptr(), n, labels.data_ptr())
    torch.cuda.synchronize()
    p = enc.profile_get()
    print("frames %d  conv %.2f ms  stage %.2f ms  = %.0f k CTU/s (conv kernel alone %.0f k)" % (n, p["cnn_conv_ms"], p["cnn_ms"], n * ctus / p["cnn_ms"], n * ctus / p["cnn_conv_ms"]), flush=True)
    enc.profile_enable(True)
enc.close()
