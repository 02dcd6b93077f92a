"""Residual blocks of real TU codings for tools/micro_rd.py: the stage-trace build of the library (lib/libhevcdl_hip_trace.so) runs the decision kernel on a
window of the bench content (ref_tools.synth_yuv at 3840x2160, QP 32, labels from the on-device CNN) and a random sample of the residual blocks it logs
(luma, every TU size, trial codings included) is written to tools/data/blocks_q32.npz.  Run on a GPU box: python tools/capture_blocks.py"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import ctypes, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import hevcdl_amd, ref_tools
W, H, w, h, x0, y0 = 3840, 2160, 512, 256, 1280, 640
full = ref_tools.synth_yuv(W, H, 1, 4000)[0]
Y = full[:W * H].reshape(H, W)[y0:y0 + h, x0:x0 + w]
U = full[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)[y0 // 2:(y0 + h) // 2, x0 // 2:(x0 + w) // 2]
V = full[W * H * 5 // 4:].reshape(H // 2, W // 2)[y0 // 2:(y0 + h) // 2, x0 // 2:(x0 + w) // 2]
yuv = np.concatenate([Y.ravel(), U.ravel(), V.ravel()])[None]
enc = hevcdl_amd.Encoder(w, h, 32, max_frames=1)
labels = enc.predict_depth(yuv)
enc.compress_frames(yuv, labels)
lib = hevcdl_amd.load_library()
lib.hevcdl_stage_trace_fetch.restype = ctypes.c_size_t
lib.hevcdl_stage_trace_fetch.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(12 << 20, np.uint32)
used = int(lib.hevcdl_stage_trace_fetch(buf.ctypes.data, buf.size))
enc.close()
np.save(sys.argv[1], buf[:min(used, buf.size)])
"""
sys.path.insert(0, ROOT)
import hevcdl_amd
tmp = "/tmp/stage_words.npy"
r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "oracle")), tmp], env=dict(os.environ, HEVCDL_LIB=hevcdl_amd.TRACE_LIB_PATH), capture_output=True, text=True)
assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
words = np.load(tmp)
by_n = {4: [], 8: [], 16: [], 32: []}
i = 0
while i < len(words):
    kind, a, b = int(words[i]), int(words[i + 1]), int(words[i + 2])
    if kind < 2:
        i += 6
        continue
    if kind == 2 and b == 0:
        by_n[a].append(words[i + 6:i + 6 + a * a].copy().view(np.int32).astype(np.int16))
    i += 6 + 3 * a * a
rng = np.random.default_rng(1)
out = {}
for n, lst in by_n.items():
    pick = rng.choice(len(lst), size=min(64, len(lst)), replace=False)
    blk = np.zeros((len(pick), 1024), np.int16)
    for j, p in enumerate(pick):
        blk[j, :n * n] = lst[p]
    out["n%d" % n] = blk
    print("n %2d: %6d luma TU codings logged, %d kept, mean |residual| %.2f" % (n, len(lst), len(pick), float(np.abs(blk[:, :n * n]).mean())))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "blocks_q32.npz"), **out)
