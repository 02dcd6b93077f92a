"""oracle/cnn_oracle.py -- TEST INFRASTRUCTURE (CPU oracle), not product code.

numpy restatement of the reference's CNN depth predictor:
  * ConvNet2 forward                         /root/reference/use_model.py:16-58
    (BatchNorm in TRAINING mode: per-sample, per-channel spatial mean / biased variance, eps 1e-5,
     because use_model.py:61-63 never calls .eval())
  * CTU / quadrant tiling, zero padding      /root/reference/use_model.py:80-95
  * 4x argmax + label fix-ups                /root/reference/use_model.py:101-119
plus the two steps this project defines itself (parity unpinned in the reference):
  * YUV 4:2:0 -> RGB input transform (the reference goes through ffmpeg -> JPEG -> PIL)
  * clamping labels to the picture boundary
Pinned by tests/golden/cnn_*.npz, generated from the reference model by oracle/gen_fixtures.py.
"""
import json
import os

import numpy as np

EPS = np.float32(1e-5)
QUADS = ((0, 1, 4, 5), (2, 3, 6, 7), (8, 9, 12, 13), (10, 11, 14, 15))


def load_weights(blob_path, manifest_path=None):
    """Flat little-endian fp32 blob + JSON manifest (name, shape, offset in floats)."""
    manifest_path = manifest_path or os.path.splitext(blob_path)[0] + ".json"
    man = json.load(open(manifest_path))
    blob = np.fromfile(blob_path, dtype="<f4")
    return {t["name"]: blob[t["offset"]:t["offset"] + int(np.prod(t["shape"]))].reshape(t["shape"]) for t in man["tensors"]}


def _conv2d(x, w, b, pad):
    """x [N,C,H,W] fp32, w [O,C,k,k]; direct im2col GEMM in fp32."""
    n, c, h, wd = x.shape
    o, _, k, _ = w.shape
    xp = np.zeros((n, c, h + 2 * pad, wd + 2 * pad), np.float32)
    xp[:, :, pad:pad + h, pad:pad + wd] = x
    cols = np.empty((n, c * k * k, h * wd), np.float32)
    i = 0
    for ci in range(c):
        for ky in range(k):
            for kx in range(k):
                cols[:, i, :] = xp[:, ci, ky:ky + h, kx:kx + wd].reshape(n, -1)
                i += 1
    out = np.matmul(w.reshape(o, -1).astype(np.float32), cols) + b.reshape(1, o, 1)
    return out.reshape(n, o, h, wd).astype(np.float32)


def _bn_train(x, gamma, beta):
    mean = x.mean(axis=(2, 3), keepdims=True, dtype=np.float64)
    var = ((x.astype(np.float64) - mean) ** 2).mean(axis=(2, 3), keepdims=True)
    inv = (1.0 / np.sqrt(var + 1e-5)).astype(np.float32)
    return ((x - mean.astype(np.float32)) * inv * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)).astype(np.float32)


def _bn_eval(x, gamma, beta, mean, var):
    """model.eval(): the checkpoint's running statistics (HEVCDL_BN_EVAL; not what the reference pipeline runs)."""
    inv = (1.0 / np.sqrt(var.astype(np.float64) + 1e-5)).astype(np.float32)
    return ((x - mean.reshape(1, -1, 1, 1)) * inv.reshape(1, -1, 1, 1) * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)).astype(np.float32)


def _pool(x, k):
    n, c, h, w = x.shape
    return x.reshape(n, c, h // k, k, w // k, k).max(axis=(3, 5))


def _block(x, w, name, pad, pool, bn_eval=False):
    y = _conv2d(x, w[name + ".0.weight"], w[name + ".0.bias"], pad)
    if bn_eval:
        y = _bn_eval(y, w[name + ".1.weight"], w[name + ".1.bias"], w[name + ".1.running_mean"], w[name + ".1.running_var"])
    else:
        y = _bn_train(y, w[name + ".1.weight"], w[name + ".1.bias"])
    return _pool(np.maximum(y, 0), pool)


def forward(w, x32, x64, bn_eval=False):
    """use_model.py:48-58.  x32 [N,3,32,32], x64 [N,3,64,64] fp32 in [0,1] -> logits [N,16]."""
    a = _block(x32, w, "conv1", 2, 2, bn_eval)
    b = _block(x64, w, "conv64", 2, 4, bn_eval)
    out = np.concatenate([a, b], axis=1)
    out = _block(out, w, "conv2", 1, 2, bn_eval)
    out = _block(out, w, "conv3", 1, 2, bn_eval)
    out = out.reshape(out.shape[0], -1)
    out = np.maximum(out @ w["fc1.0.weight"].T + w["fc1.0.bias"], 0).astype(np.float32)
    out = np.maximum(out @ w["fc2.0.weight"].T + w["fc2.0.bias"], 0).astype(np.float32)
    return (out @ w["fc3.weight"].T + w["fc3.bias"]).astype(np.float32)


def ctu_logits(w, ctu_rgb, bn_eval=False):
    """ctu_rgb [N,64,64,3] uint8 -> logits [N,4,16] (quadrant layer2 = 0..3, use_model.py:89-100)."""
    x = (ctu_rgb.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2)     # ToTensor
    outs = []
    for q in range(4):
        ox, oy = (q % 2) * 32, (q // 2) * 32
        outs.append(forward(w, np.ascontiguousarray(x[:, :, oy:oy + 32, ox:ox + 32]), x, bn_eval))
    return np.stack(outs, axis=1)


def labels_from_logits(logits):
    """use_model.py:101-119 on logits [N,4,16] -> labels [N,16] uint8."""
    n = logits.shape[0]
    labels = np.zeros((n, 16), np.uint8)
    for i in range(n):
        lab = [0] * 16
        for q in range(4):
            p = [int(np.argmax(logits[i, q, 4 * k:4 * k + 4])) for k in range(4)]
            if 0 in p and p != [0, 0, 0, 0]:
                p = [1 if v == 0 else v for v in p]
            if 1 in p and p != [1, 1, 1, 1]:
                p = [2 if v == 1 else v for v in p]
            if q == 1 and p == [0, 0, 0, 0] and lab[0] != 0:
                p = [1, 1, 1, 1]
            if q == 2 and p == [0, 0, 0, 0] and lab[2] != 0:
                p = [1, 1, 1, 1]
            if q == 3 and p == [0, 0, 0, 0] and lab[8] != 0:
                p = [1, 1, 1, 1]
            for idx, v in zip(QUADS[q], p):
                lab[idx] = v
        labels[i] = lab
    return labels


def min_depth_table(width, height):
    """Smallest depth at which the CU holding each 16x16 cell lies inside the picture; 0 for cells outside."""
    cx, cy = (width + 63) // 64, (height + 63) // 64
    t = np.zeros((cy * cx, 16), np.uint8)
    for a in range(cx * cy):
        x0, y0 = (a % cx) * 64, (a // cx) * 64
        for c in range(16):
            px, py = x0 + (c % 4) * 16, y0 + (c // 4) * 16
            if px >= width or py >= height:
                continue
            d = 0
            while d < 3:
                s = 64 >> d
                if px // s * s + s <= width and py // s * s + s <= height:
                    break
                d += 1
            t[a, c] = d
    return t


def clamp_ctu_labels(lab, md, inside):
    """Boundary policy for the 16 labels of one CTU (restates hevcdl_clamp_ctu_labels, csrc/hevcdl_dev.h): raise every cell to the smallest
    depth whose CU lies inside the picture, then make valid what the reference's walk reads -- ONE label per CU, the one of its top-left cell
    (TEncCu.cpp:496-520) -- and nothing else: a first label of 0 in a CTU inside the picture means one 64x64 CU (the other labels are not
    read and stay); otherwise a visited 32x32 quadrant whose first label is 0 gets 1, and the cells of a split quadrant get at least 2."""
    lab = np.maximum(np.asarray(lab, np.uint8), md)
    forced0 = bool((md[inside] >= 1).any())
    if not forced0 and lab[0] == 0:
        return lab
    for q in QUADS:
        q = list(q)
        if not inside[q[0]]:
            continue
        forced1 = any(inside[c] and md[c] >= 2 for c in q)
        if not forced1 and lab[q[0]] < 1:
            lab[q[0]] = 1
        if forced1 or lab[q[0]] >= 2:
            for c in q:
                if inside[c] and lab[c] < 2:
                    lab[c] = 2
    return lab


def inside_table(width, height):
    """Per CTU, per 16x16 cell: does the cell lie inside the picture."""
    cx, cy = (width + 63) // 64, (height + 63) // 64
    t = np.zeros((cy * cx, 16), bool)
    for a in range(cx * cy):
        for c in range(16):
            t[a, c] = (a % cx) * 64 + (c % 4) * 16 < width and (a // cx) * 64 + (c // 4) * 16 < height
    return t


def clamp_labels(labels, width, height):
    """Boundary policy of this project for [frames, ctus, 16] labels: see clamp_ctu_labels."""
    md, ins = min_depth_table(width, height), inside_table(width, height)
    out = np.array(labels, np.uint8).reshape(-1, md.shape[0], 16)
    for f in range(out.shape[0]):
        for a in range(md.shape[0]):
            out[f, a] = clamp_ctu_labels(out[f, a], md[a], ins[a])
    return out.reshape(np.shape(labels))


def yuv_to_rgb_picture(yuv_frame, width, height, mode="rgb601"):
    """One planar 8-bit 4:2:0 frame -> [height, width, 3] uint8 RGB: this project's input transform (the reference's own is ffmpeg -> JPEG -> PIL, unpinned).
    mode 'rgb601': BT.601 limited-range integer conversion, nearest-neighbour chroma:
        C=Y-16, D=U-128, E=V-128; R=clip((298C+409E+128)>>8), G=clip((298C-100D-208E+128)>>8), B=clip((298C+516D+128)>>8)
    mode 'luma': R=G=B=Y."""
    Y = yuv_frame[:width * height].reshape(height, width).astype(np.int32)
    if mode == "luma":
        rgb = np.stack([Y, Y, Y], axis=-1)
    else:
        U = yuv_frame[width * height:width * height * 5 // 4].reshape(height // 2, width // 2).astype(np.int32)
        V = yuv_frame[width * height * 5 // 4:].reshape(height // 2, width // 2).astype(np.int32)
        D = (U - 128).repeat(2, 0).repeat(2, 1)
        E = (V - 128).repeat(2, 0).repeat(2, 1)
        C = Y - 16
        R = (298 * C + 409 * E + 128) >> 8
        G = (298 * C - 100 * D - 208 * E + 128) >> 8
        B = (298 * C + 516 * D + 128) >> 8
        rgb = np.stack([R, G, B], axis=-1)
    return np.clip(rgb, 0, 255).astype(np.uint8)


def rgb_picture_to_ctus(rgb):
    """The tiling of use_model.py:80-95: [height, width, 3] uint8 -> [ctus,64,64,3] CTUs in raster order (ceil(w/64) per row, :80,:86-87), what lies past the
    picture edge zero-filled (PIL's img.crop beyond the picture, :91-92).  Pinned by tests/golden/cnn_f3.npz (the reference's loop itself on whole pictures)."""
    height, width = rgb.shape[:2]
    cx, cy = (width + 63) // 64, (height + 63) // 64
    pad = np.zeros((cy * 64, cx * 64, 3), np.uint8)
    pad[:height, :width] = rgb
    return pad.reshape(cy, 64, cx, 64, 3).transpose(0, 2, 1, 3, 4).reshape(cy * cx, 64, 64, 3)


def yuv_to_rgb_ctus(yuv_frame, width, height, mode="rgb601"):
    """One planar 8-bit 4:2:0 frame -> RGB CTUs: the input transform, then the reference's tiling."""
    return rgb_picture_to_ctus(yuv_to_rgb_picture(yuv_frame, width, height, mode))


def predict_labels(w, yuv_frames, width, height, mode="rgb601", clamp=True):
    """Whole 'frame -> labels' stage for [frames, w*h*3/2] uint8.  Returns (labels [F,ctus,16], logits [F,ctus,4,16])."""
    labs, logs = [], []
    for fr in yuv_frames:
        ctus = yuv_to_rgb_ctus(fr, width, height, mode)
        lg = ctu_logits(w, ctus)
        lb = labels_from_logits(lg)
        labs.append(lb)
        logs.append(lg)
    labs = np.stack(labs)
    if clamp:
        labs = clamp_labels(labs, width, height)
    return labs, np.stack(logs)
