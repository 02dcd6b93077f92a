"""oracle/cnn_torch.py -- TEST INFRASTRUCTURE (checker), not product code.

Plain PyTorch fp32 restatement of oracle/cnn_oracle.py (which restates /root/reference/use_model.py:16-58, 80-119) for sizes the numpy oracle cannot finish in a
bench run (0.3 s per CTU): the same graph -- conv / BatchNorm in TRAINING mode (per-sample statistics, biased variance, eps 1e-5) / ReLU / max-pool blocks, cat, three
linear layers -- in torch.nn.functional, fp32 throughout, on whatever device the tensors live.  Pinned against cnn_oracle.forward by tests/test_oracle_golden.py
(CPU, a handful of CTUs); bench.py's `cnn_label_check` leg runs it on the GPU as the fp32 reference the split-f16 MFMA kernel's labels are counted against.
Only tests/ and bench.py's checker legs may import this module.
"""
import numpy as np

import cnn_oracle


def _block(F, x, w, name, pad, pool):
    y = F.conv2d(x, w[name + ".0.weight"], w[name + ".0.bias"], padding=pad)
    mean = y.double().mean(dim=(2, 3), keepdim=True)
    var = ((y.double() - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    inv = (1.0 / (var + 1e-5).sqrt()).float()
    y = (y - mean.float()) * inv * w[name + ".1.weight"].view(1, -1, 1, 1) + w[name + ".1.bias"].view(1, -1, 1, 1)
    return F.max_pool2d(F.relu(y), pool)


def forward(torch, w, x32, x64):
    """cnn_oracle.forward on torch tensors: x32 [N,3,32,32], x64 [N,3,64,64] fp32 in [0,1] -> logits [N,16]."""
    F = torch.nn.functional
    a = _block(F, x32, w, "conv1", 2, 2)
    b = _block(F, x64, w, "conv64", 2, 4)
    out = _block(F, torch.cat([a, b], dim=1), w, "conv2", 1, 2)
    out = _block(F, out, w, "conv3", 1, 2)
    out = out.reshape(out.shape[0], -1)
    out = F.relu(F.linear(out, w["fc1.0.weight"], w["fc1.0.bias"]))
    out = F.relu(F.linear(out, w["fc2.0.weight"], w["fc2.0.bias"]))
    return F.linear(out, w["fc3.weight"], w["fc3.bias"])


def ctu_logits(torch, w, ctu_rgb, batch=2048):
    """ctu_rgb [N,64,64,3] uint8 (numpy or tensor) -> logits [N,4,16] fp32 tensor on w's device (quadrant order of use_model.py:89-100)."""
    dev = w["fc3.weight"].device
    x_all = torch.as_tensor(ctu_rgb)
    outs = []
    with torch.no_grad():
        for i in range(0, x_all.shape[0], batch):
            x = (x_all[i:i + batch].to(dev).float() / 255.0).permute(0, 3, 1, 2).contiguous()
            qs = []
            for q in range(4):
                ox, oy = (q % 2) * 32, (q // 2) * 32
                qs.append(forward(torch, w, x[:, :, oy:oy + 32, ox:ox + 32].contiguous(), x))
            outs.append(torch.stack(qs, dim=1))
    return torch.cat(outs)


def weights_to(torch, w_np, dev):
    return {k: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32, device=dev) for k, v in w_np.items()}


def label_check(torch, w_np, yuv_frames, width, height, dev, gpu_labels, gap=1e-2):
    """Labels of the fp32 graph for whole frames (numpy [F, w*h*3/2] uint8) against `gpu_labels` [F, ctus, 16]:
    -> {ctus, in_gap_band, labels_differing_from_fp32_oracle, ...}.  A CTU is `in the gap band` when, in any of its 16 argmax decisions, the two largest logits of the
    fp32 graph are closer than `gap` (there a rounding difference of the kernel's split-f16 operands may legitimately pick the other class)."""
    w = weights_to(torch, w_np, dev)
    ctus = band = differ = differ_outside = cells = 0
    for f in range(yuv_frames.shape[0]):
        rgb = cnn_oracle.yuv_to_rgb_ctus(yuv_frames[f], width, height)
        lg = ctu_logits(torch, w, rgb).cpu().numpy()
        lab = cnn_oracle.clamp_labels(cnn_oracle.labels_from_logits(lg)[None], width, height)[0]
        top2 = np.sort(lg.reshape(lg.shape[0], 4, 4, 4), axis=3)
        near = ((top2[..., 3] - top2[..., 2]) < gap).reshape(lg.shape[0], -1).any(axis=1)
        d = (lab != gpu_labels[f]).any(axis=1)
        ctus += lab.shape[0]; band += int(near.sum()); differ += int(d.sum()); differ_outside += int((d & ~near).sum()); cells += int((lab != gpu_labels[f]).sum())
    return {"ctus": ctus, "in_gap_band": band, "labels_differing_from_fp32_oracle": differ, "differing_outside_the_band": differ_outside, "differing_cells": cells,
            "gap": gap, "frames": int(yuv_frames.shape[0]),
            "reference": "oracle/cnn_torch.py: the graph of oracle/cnn_oracle.py (use_model.py:16-58, BatchNorm in training mode) in fp32 torch.nn.functional on the same GPU"}
