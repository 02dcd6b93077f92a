"""Stress of WaveFrontSynchro on the device: many launches of random sizes / content / labels / QP / frame counts, CTU rows on waves of their own (rows wait for each other
through finished-CTU counts in HBM; few units: idle workgroups take posted second passes; many units: rows queue on wave slots) against the form in which one wave walks a
frame's rows in order (exec_flags 1), byte for byte.  python tools/stress_wavefront.py [seconds]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import hevcdl_amd, ref_tools
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2027)
t0 = time.time(); runs = 0; ctus = 0
sizes = [(512, 320), (416, 240), (832, 480), (1280, 720), (128, 448), (64, 256), (200, 136), (1920, 1080), (3840, 2160)]
encs = {}
while time.time() - t0 < budget:
    w, h = sizes[int(rng.integers(0, len(sizes)))]
    qp = int(rng.integers(22, 40))
    nf = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 130, 400]))
    if nf > 40 and w * h > 512 * 320:
        nf = 40
    if w * h >= 1920 * 1080:
        nf = min(nf, 3)               # (the one-wave form walks such a frame for seconds)
    key = (w, h, qp)
    if key not in encs:
        pair = []
        for flags in (0, 1):
            cfg = hevcdl_amd.default_config(w, h, qp, max_frames=400 if w * h <= 512 * 320 else 40, wavefront=True)
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
    assert np.array_equal(a[2]["est_bits"], b[2]["est_bits"]) and np.array_equal(a[2]["sse"], b[2]["sse"]), ("stats", w, h, qp, nf, runs)
    runs += 1; ctus += nf * a[0].shape[1]
    if len(encs) > 8:
        for p in encs.pop(next(iter(encs))): p.close()
print("wavefront stress ok: %d launch pairs, %d CTUs, %.0f s" % (runs, ctus, time.time() - t0))
