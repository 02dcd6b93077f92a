"""Stress of the unit hand-over (more units than workgroups, unevenly dealt: frames travel round the ring of workgroups -- the form that carries the 600-frame job) on both
builds of the 8-bit kernel: many launches of 257..900 small frames of random content and labels, each compared byte for byte with the independent form (exec_flags 1).
python tools/stress_handover.py [seconds]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import hevcdl_amd, ref_tools
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2027)
t0 = time.time(); runs = 0; ctus = 0; forms = {}
sizes = [(128, 128), (192, 128), (256, 192), (320, 192)]
NMAX = 900
while time.time() - t0 < budget:
    w, h = sizes[int(rng.integers(0, len(sizes)))]
    qp = int(rng.integers(22, 40))
    nf = int(rng.choice([257, 300, 383, 512, 513, 640, 767, 768, 900]))
    flags = int(rng.choice([0, 2, 4]))                 # the library's choice / the ten-wave build / the eight-wave build
    encs = []
    for fl in (flags, 1):
        cfg = hevcdl_amd.default_config(w, h, qp, max_frames=NMAX)
        cfg.exec_flags = fl
        encs.append(hevcdl_amd.Encoder(w, h, qp, cfg=cfg))
    base = ref_tools.synth_yuv(w, h, 4, int(rng.integers(0, 1 << 30)))
    noise = rng.integers(-6, 7, (8, base.shape[1]))
    yuv = np.stack([np.clip(base[i % 4].astype(np.int16) + noise[i % 8] * (1 + i % 3), 0, 255).astype(np.uint8) for i in range(nf)])
    labels = ref_tools.make_labels(w, h, nf, "rand", int(rng.integers(0, 1000)))
    a = encs[0].compress_frames(yuv, labels)
    form = encs[0].last_rd_launch()
    b = encs[1].compress_frames(yuv, labels)
    for f in ref_tools.FIELDS:
        assert np.array_equal(a[0][f], b[0][f]), ("records", f, w, h, qp, nf, flags, runs)
    assert np.array_equal(a[1], b[1]), ("recon", w, h, qp, nf, flags, runs)
    key = " ".join(form.split(" ")[:2]); forms[key] = forms.get(key, 0) + 1
    runs += 1; ctus += nf * a[0].shape[1]
    for e in encs: e.close()
print("stress ok: %d launch pairs, %d CTUs, %.0f s; forms: %s" % (runs, ctus, time.time() - t0, forms))
