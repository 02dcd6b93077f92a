import sys, os, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import bench
nf, w, h = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), 3840, 2160
d = tempfile.mkdtemp(prefix="hevcdl_cli_")
bench.synth_frames_torch(torch, torch.device("cuda", 0), w, h, list(range(nf)), seed=4000).cpu().numpy().tofile(os.path.join(d, "in.yuv"))
app = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hevc-deep-learning-pipeline_amd", "bin", "TAppEncoderHevcdl")
for extra in ([], ["--TileUniformSpacing=1", "--NumTileColumnsMinus1=3", "--NumTileRowsMinus1=1"], ["--WaveFrontSynchro=1"]):
    t = time.time()
    r = subprocess.run([app, "-i", "in.yuv", "-wdt", str(w), "-hgt", str(h), "-q", "32", "-b", "o.bin", "-o", "o.yuv", "--SEIDecodedPictureHash=1", "--Level=6.2"] + extra, cwd=d, capture_output=True, text=True)
    dt = time.time() - t
    print("cli", ("wavefront" if "WaveFront" in extra[0] else "tiles 4x2") if extra else "default cfg", "rc", r.returncode, "%.1f s" % dt, "%.1f pictures/s" % (nf / dt), r.stdout.splitlines()[-1] if r.stdout else "", r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "")
