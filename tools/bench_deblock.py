#!/usr/bin/env python3
"""Roofline measurement of the deblocking stage (HBM bound): real TU grids from the RD kernel, replicated to a batch
that does not fit the 256 MiB Infinity Cache, timed with events on the launch stream."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import hevcdl_amd
import ref_tools

W, H, QP = 3840, 2160, 32
SRC, F, REPS = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 96, 5
dev = torch.device("cuda", 0)
yuv = ref_tools.synth_yuv(W, H, SRC, seed=1000)
enc = hevcdl_amd.Encoder(W, H, QP, max_frames=F)
recs, recon, _ = enc.compress_frames(yuv)
idx = np.arange(F) % SRC
d_recon = torch.from_numpy(recon[idx].copy()).to(dev)
d_recs = torch.from_numpy(np.frombuffer(recs[idx].tobytes(), np.uint8).copy()).to(dev)
d_out = torch.empty_like(d_recon)
s = torch.cuda.current_stream(dev)
enc.deblock_frames_dev(d_recon.data_ptr(), F, d_recs.data_ptr(), d_out.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
ref = ref_tools.run_deblock(recon[:1], W, H, QP, np.frombuffer(recs[:1].tobytes(), dtype=ref_tools.REC_DTYPE).reshape(1, -1))
assert np.array_equal(d_out[0].cpu().numpy(), ref[0]), "deblocked frame differs from the oracle"
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(s)
for _ in range(REPS):
    enc.deblock_frames_dev(d_recon.data_ptr(), F, d_recs.data_ptr(), d_out.data_ptr(), s.cuda_stream)
b.record(s); torch.cuda.synchronize()
ms = a.elapsed_time(b) / REPS
fb = W * H * 3 // 2
algo = F * (2 * fb + 2 * (W // 4) * (H // 4))            # read picture + write picture + depth/trIdx bytes of the records
out_line = (json.dumps({"stage": "deblock", "frames": F, "ms_per_batch": ms, "frames_per_s": F / ms * 1e3, "ctus_per_s": F * enc.ctus / ms * 1e3,
                  "algorithmic_GB": algo / 1e9, "achieved_GBps": algo / ms / 1e6, "peak_GBps": 8000.0, "frac": algo / ms / 1e6 / 8000.0}))
print(out_line)
# ---- SAO on the deblocked pictures ----
d_org = torch.from_numpy(yuv[idx].copy()).to(dev)
d_params = torch.empty((F, enc.ctus, 3 * hevcdl_amd.SAO_DTYPE.itemsize), dtype=torch.uint8, device=dev)
d_final = torch.empty_like(d_recon)
enc.sao_frames_dev(d_org.data_ptr(), d_out.data_ptr(), F, d_params.data_ptr(), d_final.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
o_par, o_fin = ref_tools.run_sao(yuv[:1], d_out[:1].cpu().numpy(), W, H, QP)
assert np.array_equal(d_final[0].cpu().numpy(), o_fin[0]) and d_params[0].cpu().numpy().tobytes() == o_par[0].tobytes(), "SAO differs from the oracle"
a.record(s)
for _ in range(REPS):
    enc.sao_frames_dev(d_org.data_ptr(), d_out.data_ptr(), F, d_params.data_ptr(), d_final.data_ptr(), s.cuda_stream)
b.record(s); torch.cuda.synchronize()
ms = a.elapsed_time(b) / REPS
algo = F * 3 * fb                          # read original + read deblocked + write final (statistics re-read the two inputs: counted once)
print(json.dumps({"stage": "sao", "frames": F, "ms_per_batch": ms, "frames_per_s": F / ms * 1e3, "algorithmic_GB": algo / 1e9, "achieved_GBps": algo / ms / 1e6, "peak_GBps": 8000.0, "frac": algo / ms / 1e6 / 8000.0}))
enc.close()
