"""Largest difference of the on-device CNN's logits from the reference model's (tests/golden/cnn_f1.npz, cnn_f3.npz): python tools/cnn_err.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hevcdl_amd, cnn_oracle
G = os.path.join(ROOT, "tests", "golden")
e = hevcdl_amd.Encoder(128, 128, 32, max_frames=1)
f = np.load(os.path.join(G, "cnn_f1.npz"))
lab, lg = e.predict_depth_rgb(f["ctu_rgb"])
print("cnn_f1: max |logit - reference model| %.3e   labels differing %d of %d CTUs" % (np.abs(lg - f["logits"]).max(), int((lab != f["labels"]).any(axis=1).sum()), len(lab)))
g = np.load(os.path.join(G, "cnn_f3.npz"))
for n in range(int(g["n_pictures"])):
    lab, lg = e.predict_depth_rgb(cnn_oracle.rgb_picture_to_ctus(g["rgb%d" % n]))
    print("cnn_f3 picture %d: max |logit - reference loop| %.3e   label files differing %d of %d" % (n, np.abs(lg - g["logits%d" % n]).max(), int((lab != g["labels%d" % n]).any(axis=1).sum()), len(lab)))
e.close()
