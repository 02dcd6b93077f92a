#!/usr/bin/env python3
"""gpurun_out/prof/<tag>_summary.txt (tools/profile_round.sh: rocprofv3 kernel trace + the FETCH_SIZE / WRITE_SIZE counter passes of a 300-frame launch)
-> profiles/<tag>_traffic.json, the file bench.py reports `roofline.traffic` from while the hash of rd_kernel.hip matches.
    python tools/traffic_json.py <tag> [frames of the counter launch, default 128]"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 600
txt = open(os.path.join(ROOT, "gpurun_out", "prof", tag + "_summary.txt")).read()
vals = {}
for m in re.finditer(r"hevcdl_rd_frame_kernel[^|]*\| (FETCH_SIZE|WRITE_SIZE) = ([\d.eE+]+)", txt):
    vals[m.group(1)] = float(m.group(2))
ctus = frames * 2040
sha = hashlib.sha256(open(os.path.join(ROOT, "hevc-deep-learning-pipeline_amd", "csrc", "rd_kernel.hip"), "rb").read()).hexdigest()[:16]
out = {"kernel": re.search(r"(hevcdl_rd_frame_kernel\w*) \| FETCH_SIZE", txt).group(1) if re.search(r"(hevcdl_rd_frame_kernel\w*) \| FETCH_SIZE", txt) else "hevcdl_rd_frame_kernel", "rd_kernel_sha16": sha,
       "source": "profiles/%s_rocprofv3_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, one launch of %d frames of 3840x2160)" % (tag, frames),
       "ctus": ctus, "fetch_kb": vals.get("FETCH_SIZE"), "write_kb": vals.get("WRITE_SIZE"),
       "fetch_bytes_per_ctu": vals["FETCH_SIZE"] * 1024 / ctus if "FETCH_SIZE" in vals else None,
       "write_bytes_per_ctu": vals["WRITE_SIZE"] * 1024 / ctus if "WRITE_SIZE" in vals else None,
       "note": "raw counter values (KB) of the L2 <-> fabric interface (requests served by the 256 MB Infinity Cache are counted as well); the guide's x2 FETCH_SIZE correction "
               "is calibrated for wide coalesced streaming reads only and is NOT applied to this narrow access pattern (uncalibrated), WRITE_SIZE is uncalibrated; one counter per "
               "pass (--kernel-include-regex hevcdl_rd_frame_kernel) on a %d-frame launch%s" % (frames, ": the launch shape of the timed step itself" if frames == 600 else "")}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
