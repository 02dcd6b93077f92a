#!/usr/bin/env python3
"""a-5 evidence (container-only, no GPU): how often do the labels from this project's DEFINED CNN input (BT.601 integer YUV->RGB, include/hevcdl.h) equal the
labels the reference's own input path would give?  The reference goes planar YUV -> ffmpeg -> JPEG files (gen_frames.py:21, default mjpeg settings of an
unpinned ffmpeg) -> PIL -> RGB crops (use_model.py:77-95).  ffmpeg is not in this image; Pillow is, so the lossy step is emulated: the BT.601 RGB picture is
written as a 4:2:0 baseline JPEG at a sweep of qualities (ffmpeg's default -q:v for mjpeg corresponds roughly to Pillow quality 75-90), read back, cut into
CTUs and sent through the same numpy CNN (oracle/cnn_oracle.py, pinned to the reference model).  Labels compared before the boundary clamp.
    python tools/jpeg_label_agreement.py [frames of 1920x1080, default 2 = 1020 CTUs] [out.json]"""
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    from PIL import Image
    import cnn_oracle
    import ref_tools
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r03_a5_jpeg_label_agreement.json")
    w, h = 1920, 1080
    wts = cnn_oracle.load_weights(os.path.join(ROOT, "hevc-deep-learning-pipeline_amd", "weights", "hevc_encoder_model.f32"))
    yuv = ref_tools.synth_yuv(w, h, nf, seed=1000)
    cx, cy = (w + 63) // 64, (h + 63) // 64

    def labels_of(ctus):
        return np.stack([cnn_oracle.labels_from_logits(cnn_oracle.ctu_logits(wts, ctus[i:i + 1]))[0] for i in range(ctus.shape[0])])

    def picture_of(ctus):                       # [ctus,64,64,3] -> the picture (the zero fill past the edge dropped)
        pic = ctus.reshape(cy, cx, 64, 64, 3).transpose(0, 2, 1, 3, 4).reshape(cy * 64, cx * 64, 3)
        return pic[:h, :w]

    def ctus_of(pic):                           # PIL crop semantics: zero fill past the picture edge (use_model.py:92-93)
        full = np.zeros((cy * 64, cx * 64, 3), np.uint8)
        full[:h, :w] = pic
        return full.reshape(cy, 64, cx, 64, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 64, 64, 3)

    res = {"size": "%dx%d" % (w, h), "frames": nf, "ctus": nf * cx * cy, "qualities": []}
    t0 = time.time()
    exact_ctus = [cnn_oracle.yuv_to_rgb_ctus(yuv[f], w, h) for f in range(nf)]
    exact = np.concatenate([labels_of(c) for c in exact_ctus])
    print("exact-input labels: %.0f s, depth histogram %s" % (time.time() - t0, np.bincount(exact.ravel(), minlength=4).tolist()), flush=True)
    for q in (60, 75, 85, 90, 95, 100):
        labs, psnr = [], []
        for f in range(nf):
            pic = picture_of(exact_ctus[f])
            buf = io.BytesIO()
            Image.fromarray(pic, "RGB").save(buf, "JPEG", quality=q, subsampling=2)          # 4:2:0, what mjpeg from yuv420p gives
            back = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
            psnr.append(10 * np.log10(255.0 ** 2 / max(1e-9, ((back.astype(np.float64) - pic) ** 2).mean())))
            labs.append(labels_of(ctus_of(back)))
        labs = np.concatenate(labs)
        row = {"jpeg_quality": q, "rgb_psnr_db": float(np.mean(psnr)), "cells_equal": float((labs == exact).mean()), "ctus_all_16_equal": float((labs == exact).all(axis=1).mean()),
               "first_label_equal": float((labs[:, 0] == exact[:, 0]).mean()), "mean_abs_depth_difference": float(np.abs(labs.astype(int) - exact.astype(int)).mean())}
        res["qualities"].append(row)
        print(row, flush=True)
    res["note"] = ("labels of the defined BT.601 input vs labels after a JPEG round trip of the same RGB picture (Pillow, 4:2:0), numpy CNN pinned to the reference model, "
                   "training-mode BatchNorm as the reference runs it; per 16x16 cell and per CTU.  The reference's ffmpeg is unpinned and absent: an emulation, not a pin.")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
