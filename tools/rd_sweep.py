#!/usr/bin/env python3
"""C3 of the survey: 3840x2160, N frames x QP {22, 27, 32, 37} -> rate / PSNR points of this encoder (CNN labels), and -- because the
reference repository holds no unpruned-HM numbers for synthetic content -- the same sweep with a fixed label policy (every CU 16x16) as a
stand-in second curve, so that the BD-rate / BD-PSNR tooling (metrics.py, the reference's calc_BDBR formulas) is exercised end to end.
    python tools/rd_sweep.py [frames] [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def main():
    import torch
    torch.cuda.init()
    import bench
    import hevcdl_amd
    import hevcdl_amd.metrics as metrics
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "c3_sweep.json")
    w, h = 3840, 2160
    dev = torch.device("cuda", 0)
    yuv = bench.synth_frames_torch(torch, dev, w, h, nf, seed=3000).cpu().numpy()
    res = {"width": w, "height": h, "frames": nf, "curves": {}}
    for policy in ("cnn", "depth2"):
        pts = []
        for qp in (22, 27, 32, 37):
            t0 = time.time()
            enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
            labels = enc.predict_depth(yuv) if policy == "cnn" else np.full((nf, enc.ctus, 16), 2, np.uint8)
            if policy == "depth2":                       # the boundary clamp the CNN path applies (last CTU row of 2160p is half outside)
                cl = enc.predict_depth(yuv[:1])[0]
                labels = np.maximum(labels, np.where(cl > 2, cl, 0)[None])
            recs, recon, _ = enc.compress_frames(yuv, labels)
            dbk = enc.deblock_frames(recon, recs)
            sao, final = enc.sao_frames(yuv, dbk)
            enc.close()
            summ = metrics.Summary(w, h, 30.0)
            ysz = w * h
            for i in range(nf):
                au = hevcdl_amd.write_access_unit(w, h, qp, i, recs[i], sao=sao[i])
                d = (yuv[i].astype(np.int64) - final[i].astype(np.int64)) ** 2
                summ.add(len(au) * 8, (int(d[:ysz].sum()), int(d[ysz:ysz + ysz // 4].sum()), int(d[ysz + ysz // 4:].sum())))
            a = summ.averages()
            pts.append({"qp": qp, "kbps": summ.bitrate_kbps(), "psnr_y": a[0], "psnr_u": a[1], "psnr_v": a[2], "psnr_yuv": summ.yuv_psnr(),
                        "depth_hist": np.bincount(labels.ravel(), minlength=4).tolist(), "seconds": time.time() - t0})
            print(policy, pts[-1], flush=True)
        res["curves"][policy] = pts
    a, b = res["curves"]["depth2"], res["curves"]["cnn"]
    res["bd_rate_cnn_vs_depth2_percent"] = metrics.bd_rate([p["kbps"] for p in a], [p["psnr_y"] for p in a], [p["kbps"] for p in b], [p["psnr_y"] for p in b])
    res["bd_psnr_cnn_vs_depth2_db"] = metrics.bd_psnr([p["kbps"] for p in a], [p["psnr_y"] for p in a], [p["kbps"] for p in b], [p["psnr_y"] for p in b])
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print("BD-rate (CNN labels vs all-16x16 labels): %.2f %%, BD-PSNR %.3f dB" % (res["bd_rate_cnn_vs_depth2_percent"], res["bd_psnr_cnn_vs_depth2_db"]))


if __name__ == "__main__":
    main()
