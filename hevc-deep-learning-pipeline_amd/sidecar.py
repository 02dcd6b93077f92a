"""Drop-in for the reference's Python sidecar at the FILE level: /root/reference/use_model.py:60-125.

The reference pipeline is: gen_frames.py (ffmpeg: in.yuv -> rec/frames/<n>.jpg, 1-based) -> use_model.py (PIL decode, 64x64 CTUs in raster order with zero fill past
the picture edge, four ConvNet2 forwards per CTU, argmax + fix-ups, one text file pred/<n-1>/ctu<i>.txt per CTU: 16 digits separated by blanks, written under a
temporary name and renamed) -> the encoder polls for those files (TEncCu.cpp:244-253).  This module is the middle step on the MI355X: the SAME JPEG files through the
SAME decoder (PIL), the CNN and the label stage on the device (hevcdl_predict_depth_rgb: the tensor the reference feeds its model, labels exactly as its lines 101-119
write them -- no boundary clamp: the file is what use_model.py would have written; the encoder side applies its policy when it reads labels), the same files out.

    python -m hevcdl_amd.sidecar [--frames-dir ./rec/frames] [--pred-dir ./pred] [--frames N | --cfg bitstream.cfg] [--device 0]

(With the planar YUV at hand the JPEG detour is not needed at all: hevcdl_predict_depth / bin/TAppEncoderHevcdl take the frames themselves; DESIGN.md section 2 says how
far that defined input is from a JPEG round trip.)  No CPU path: without the built library or without a GPU this fails."""
import argparse
import math
import os
import re

import numpy as np


def rgb_picture_to_ctus(rgb):
    """use_model.py:80-95: [height, width, 3] uint8 -> [ctus][64][64][3], CTUs in raster order (ceil(width / 64) a row), zero fill where img.crop reaches past the picture."""
    rgb = np.asarray(rgb, np.uint8)
    height, width = rgb.shape[:2]
    cx, cy = (width + 63) // 64, (height + 63) // 64
    pad = np.zeros((cy * 64, cx * 64, 3), np.uint8)
    pad[:height, :width] = rgb[:, :, :3]
    return pad.reshape(cy, 64, cx, 64, 3).transpose(0, 2, 1, 3, 4).reshape(cy * cx, 64, 64, 3)


def frames_to_be_encoded(cfg_path):
    """FramesToBeEncoded of a reference-style cfg (use_model.py:65-71 reads its eighth line; this reads the key)."""
    for line in open(cfg_path, encoding="utf-8", errors="replace"):
        m = re.match(r"\s*FramesToBeEncoded\s*:\s*(\d+)", line)
        if m:
            return int(m.group(1))
    raise ValueError("%s holds no FramesToBeEncoded" % cfg_path)


def write_label_files(pred_dir, frame_index, labels):
    """pred/<frame_index>/ctu<i>.txt for every CTU: 16 digits, a blank behind each; under a temporary name first, then renamed (use_model.py:121-125: the encoder polls)."""
    d = os.path.join(pred_dir, str(frame_index))
    os.makedirs(d, exist_ok=False)                      # (as os.mkdir in the reference: an existing directory is an error, labels of an old run must not be mixed in)
    for i, lab in enumerate(labels):
        tmp = os.path.join(d, "ctu.txt")
        with open(tmp, "w", encoding="utf-8") as f:
            f.write("".join("%d " % int(v) for v in lab))
        os.rename(tmp, os.path.join(d, "ctu%d.txt" % i))


def label_frames(frames_dir, pred_dir, n_frames=None, device=0, log=print):
    """The reference's loop over rec/frames/1.jpg ... (use_model.py:72-125) -> label files; returns the number of frames labelled."""
    from PIL import Image
    from . import Encoder
    total = len(os.listdir(frames_dir))
    enc = None
    done = 0
    try:
        for number in range(1, total + 1):
            if n_frames is not None and number > n_frames:
                break
            img = Image.open(os.path.join(frames_dir, "%d.jpg" % number)).convert("RGB")
            width, height = img.size
            ctus = rgb_picture_to_ctus(np.asarray(img))
            assert ctus.shape[0] == math.ceil(width / 64) * math.ceil(height / 64)
            if enc is None:                             # (any geometry will do: the RGB entry point works on CTUs; QP does not enter the CNN)
                enc = Encoder(64, 64, 32, max_frames=1, device=device)
            labels, _ = enc.predict_depth_rgb(ctus)
            write_label_files(pred_dir, number - 1, labels)
            done += 1
            log("frame %d: %dx%d, %d CTUs labelled" % (number, width, height, ctus.shape[0]))
    finally:
        if enc is not None:
            enc.close()
    return done


def main():
    ap = argparse.ArgumentParser(description="MI355X drop-in for the reference's use_model.py: rec/frames/<n>.jpg -> pred/<n-1>/ctu<i>.txt")
    ap.add_argument("--frames-dir", default="./rec/frames")
    ap.add_argument("--pred-dir", default="./pred")
    ap.add_argument("--frames", type=int, default=None, help="FramesToBeEncoded (default: from --cfg, else every JPEG)")
    ap.add_argument("--cfg", default=None, help="reference-style cfg holding FramesToBeEncoded (bitstream.cfg)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    n = a.frames if a.frames is not None else (frames_to_be_encoded(a.cfg) if a.cfg else None)
    os.makedirs(a.pred_dir, exist_ok=True)
    label_frames(a.frames_dir, a.pred_dir, n, a.device)


if __name__ == "__main__":
    main()
