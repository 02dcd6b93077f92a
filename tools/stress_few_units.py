"""Stress of the few-units form (second passes / chroma modes posted to idle workgroups): many launches of 1..40 frames of random content and labels, each
compared byte for byte with the independent form (exec_flags 1).  python tools/stress_few_units.py [seconds]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import hevcdl_amd, ref_tools
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2026)
t0 = time.time(); runs = 0; ctus = 0
sizes = [(512, 320), (416, 240), (832, 480), (1280, 720)]
encs = {}
while time.time() - t0 < budget:
    w, h = sizes[int(rng.integers(0, len(sizes)))]
    qp = int(rng.integers(22, 40))
    nf = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 17, 24, 40]))
    key = (w, h, qp)
    if key not in encs:
        pair = []
        for flags in (0, 1):
            cfg = hevcdl_amd.default_config(w, h, qp, max_frames=40)
            cfg.exec_flags = flags
            pair.append(hevcdl_amd.Encoder(w, h, qp, cfg=cfg))
        encs[key] = pair
    base = ref_tools.synth_yuv(w, h, 2, int(rng.integers(0, 1 << 30)))
    yuv = np.stack([np.clip(base[i % 2].astype(np.int16) + rng.integers(-4, 5, base.shape[1]) * (1 + i % 3), 0, 255).astype(np.uint8) for i in range(nf)])
    mode = int(rng.integers(0, 3))
    labels = encs[key][0].predict_depth(yuv) if mode == 0 else ref_tools.make_labels(w, h, nf, "rand", int(rng.integers(0, 1000)))
    a = encs[key][0].compress_frames(yuv, labels)
    b = encs[key][1].compress_frames(yuv, labels)
    for f in ref_tools.FIELDS:
        assert np.array_equal(a[0][f], b[0][f]), ("records", f, w, h, qp, nf, runs)
    assert np.array_equal(a[1], b[1]), ("recon", w, h, qp, nf, runs)
    runs += 1; ctus += nf * a[0].shape[1]
    if len(encs) > 12:
        for p in encs.pop(next(iter(encs))): p.close()
print("stress ok: %d launch pairs, %d CTUs, %.0f s" % (runs, ctus, time.time() - t0))
