#!/usr/bin/env python3
"""C3 of BASELINE.json: BD-rate / BD-PSNR against the HM anchor.

Three rate / PSNR curves over QP {22, 27, 32, 37} on the same synthetic pictures:
  anchor      unpruned HM 16.20 = the reference encoder with both depth checks forced on (oracle/_ref/TAppEncoder_anchor, see
              oracle/build_ref.sh) -- every CU depth is searched, the label files are ignored;
  label_path  the reference encoder as shipped (oracle/_ref/TAppEncoder_ref), pruned by label files -- here the labels of the device CNN,
              because the reference's own label producer (ffmpeg -> JPEG -> PIL -> PyTorch) cannot run;
  device      this framework end to end on the GPU: CNN labels, decisions, deblocking, SAO, bitstream.
label_path and device must coincide point for point (same labels => byte-identical streams); BD figures come from metrics.py (the
formulas of the reference's calc_BDBR script).
    python tools/bd_anchor.py [--frames N] [--size WxH] [--out file.json] [--no-device] [--procs P]
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_ref")
ANCHOR = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_anchor")
POC_LINE = re.compile(r"^POC\s+(\d+).*?(\d+) bits \[Y ([\d.]+) dB\s+U ([\d.]+) dB\s+V ([\d.]+) dB")
QPS = (22, 27, 32, 37)


def run_encoder(binary, frame, labels, width, height, qp, base, idx):
    """One picture through a reference-encoder binary -> (bits, psnr_y, psnr_u, psnr_v, seconds) from its own log line."""
    import bench
    import ref_args
    d = bench._prepare_ref_run((idx, frame, width, height, qp, labels[None], base))
    cmd = [binary, "-i", "in.yuv", "-b", "rec/str.bin", "-o", "rec/rec.yuv"] + ref_args.reference_args(width, height, 1, qp)
    t = time.time()
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    dt = time.time() - t
    if r.returncode != 0:
        raise RuntimeError("encoder failed: " + r.stdout[-400:] + r.stderr[-400:])
    m = next((POC_LINE.match(ln) for ln in r.stdout.splitlines() if POC_LINE.match(ln)), None)
    if not m:
        raise RuntimeError("no POC line: " + r.stdout[-400:])
    size = os.path.getsize(os.path.join(d, "rec", "str.bin"))
    shutil.rmtree(d, ignore_errors=True)
    return int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), dt, size


def curve_from_runs(runs, fps=30.0):
    bits = np.array([r[0] for r in runs], np.float64)
    return {"kbps": float(bits.mean() * fps / 1000.0), "psnr_y": float(np.mean([r[1] for r in runs])), "psnr_u": float(np.mean([r[2] for r in runs])),
            "psnr_v": float(np.mean([r[3] for r in runs])), "bits_per_frame": [int(b) for b in bits], "cpu_seconds_per_frame": float(np.mean([r[4] for r in runs])),
            "stream_bytes": [int(r[5]) for r in runs]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "c3_bd.json"))
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--no-device", action="store_true", help="CPU curves only (labels: numpy CNN oracle)")
    ap.add_argument("--seed", type=int, default=3000)
    ap.add_argument("--reuse", default="", help="a previous result file of the same --frames / --size / --seed: its anchor and label_path curves (runs of the reference encoder on the host: they do not "
                                               "depend on this project's kernels) are taken over, only the device path is run again and compared with them")
    a = ap.parse_args()
    w, h = [int(v) for v in a.size.split("x")]
    nf = a.frames
    import bench
    import hevcdl_amd
    import hevcdl_amd.metrics as metrics
    procs = a.procs or bench.effective_cores()
    if a.no_device:
        import cnn_oracle
        import ref_tools
        yuv = ref_tools.synth_yuv(w, h, nf, a.seed)
        wts = cnn_oracle.load_weights(hevcdl_amd.WEIGHTS_PATH)
        labels_by_qp = {qp: cnn_oracle.predict_labels(wts, yuv, w, h)[0] for qp in QPS[:1]}
        labels_by_qp = {qp: labels_by_qp[QPS[0]] for qp in QPS}
    else:
        import torch
        torch.cuda.init()
        yuv = bench.synth_frames_torch(torch, torch.device("cuda", 0), w, h, list(range(nf)), seed=a.seed).cpu().numpy()
        labels_by_qp = {}
    res = {"width": w, "height": h, "frames": nf, "qps": list(QPS), "content": "synthetic (bench.synth_frames_torch / ref_tools.synth_yuv, seed %d)" % a.seed,
           "cpu_procs": procs, "curves": {"anchor": [], "label_path": [], "device": []}}
    import hashlib
    res["rd_kernel_sha16"] = hashlib.sha256(open(os.path.join(ROOT, "hevc-deep-learning-pipeline_amd", "csrc", "rd_kernel.hip"), "rb").read()).hexdigest()[:16]
    reuse = None
    if a.reuse:
        reuse = json.load(open(a.reuse))
        if (reuse["width"], reuse["height"], reuse["frames"], reuse["content"]) != (w, h, nf, res["content"]):
            raise SystemExit("--reuse: %s was made with another size / frame count / seed" % a.reuse)
        res["cpu_curves_from"] = "%s (reference-encoder runs of kernel sha %s's session)" % (os.path.basename(a.reuse), reuse.get("rd_kernel_sha16", "?"))
    base = tempfile.mkdtemp(prefix="hevcdl_bd_")
    try:
        for qp in QPS:
            if not a.no_device:     # the device path: labels, decisions, in-loop filters, access units
                t0 = time.time()
                enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
                labels = enc.predict_depth(yuv)
                recs, final, sao, _ = enc.encode_pictures(yuv, labels)
                enc.close()
                summ = metrics.Summary(w, h, 30.0)
                ysz = w * h
                bits = []
                for i in range(nf):
                    au = hevcdl_amd.write_access_unit(w, h, qp, 0, recs[i], sao=sao[i])      # POC 0: the reference runs code one picture per process
                    d = (yuv[i].astype(np.int64) - final[i].astype(np.int64)) ** 2
                    summ.add(len(au) * 8, (int(d[:ysz].sum()), int(d[ysz:ysz + ysz // 4].sum()), int(d[ysz + ysz // 4:].sum())))
                    bits.append(len(au) * 8)
                av = summ.averages()
                res["curves"]["device"].append({"qp": qp, "kbps": summ.bitrate_kbps(), "psnr_y": av[0], "psnr_u": av[1], "psnr_v": av[2], "bits_per_frame": bits,
                                                "gpu_seconds": time.time() - t0, "depth_hist": np.bincount(labels.ravel(), minlength=4).tolist()})
                labels_by_qp[qp] = labels
            labels = labels_by_qp[qp]
            for name, binary in (("label_path", REF), ("anchor", ANCHOR)):
                if reuse is not None:
                    c = next(c for c in reuse["curves"][name] if c["qp"] == qp)
                    if name == "label_path" and not a.no_device and reuse.get("depth_hist", {}).get(str(qp)) not in (None, res["curves"]["device"][-1]["depth_hist"]):
                        raise SystemExit("--reuse: the labels of QP %d differ from those the reused label-path runs were made with" % qp)
                    res["curves"][name].append(c)
                    continue
                t0 = time.time()
                with ThreadPoolExecutor(max_workers=procs) as pool:
                    runs = list(pool.map(lambda i: run_encoder(binary, yuv[i], labels[i], w, h, qp, base, "%s%d_%d" % (name, qp, i)), range(nf)))
                c = curve_from_runs(runs)
                c["qp"] = qp
                c["wall_seconds"] = time.time() - t0
                res["curves"][name].append(c)
                print(name, qp, {k: c[k] for k in ("kbps", "psnr_y", "cpu_seconds_per_frame", "wall_seconds")}, flush=True)
            if not a.no_device:
                dv, lp = res["curves"]["device"][-1], res["curves"]["label_path"][-1]
                print("device", qp, {k: dv[k] for k in ("kbps", "psnr_y", "gpu_seconds")}, "== label path:", dv["bits_per_frame"] == lp["bits_per_frame"], flush=True)
    finally:
        shutil.rmtree(base, ignore_errors=True)

    def bd(test):
        an = res["curves"]["anchor"]
        return {"bd_rate_percent": metrics.bd_rate([p["kbps"] for p in an], [p["psnr_y"] for p in an], [p["kbps"] for p in test], [p["psnr_y"] for p in test]),
                "bd_psnr_db": metrics.bd_psnr([p["kbps"] for p in an], [p["psnr_y"] for p in an], [p["kbps"] for p in test], [p["psnr_y"] for p in test])}
    if not a.no_device:
        res["depth_hist"] = {str(c["qp"]): c["depth_hist"] for c in res["curves"]["device"]}
    res["label_path_vs_anchor"] = bd(res["curves"]["label_path"])
    if not a.no_device:
        res["device_vs_anchor"] = bd(res["curves"]["device"])
        res["device_equals_label_path"] = all(d["bits_per_frame"] == l["bits_per_frame"] and abs(d["psnr_y"] - l["psnr_y"]) < 1e-3
                                              for d, l in zip(res["curves"]["device"], res["curves"]["label_path"]))
    an, lp = res["curves"]["anchor"], res["curves"]["label_path"]
    res["encoder_time_ratio_anchor_over_label_path"] = float(np.mean([x["cpu_seconds_per_frame"] for x in an]) / np.mean([x["cpu_seconds_per_frame"] for x in lp]))
    res["note"] = ("Y-PSNR and bits are those of the encoders' own picture lines (device: access unit size, SSE of the final picture).  The anchor is unpruned HM "
                   "(both depth checks forced on in front of TEncCu.cpp:522); it is an anchor, not a parity pin.")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "curves"}, indent=1))


if __name__ == "__main__":
    main()
