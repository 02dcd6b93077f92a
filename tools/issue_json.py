#!/usr/bin/env python3
"""SQ counter passes of the decision kernel (tools/r05_pmc.sh kernel <frames> -> gpurun_out/prof/r05pmc_k<frames>.txt, committed as
profiles/<tag>_rd_counters_f<frames>.txt) -> profiles/<tag>_issue.json: the instruction-issue bound bench.py reports as `roofline.issue`
while the hash of rd_kernel.hip matches.

The decision kernel is a chain of dependent integer / fp64 operations; its HBM fraction is ~1e-3 by construction and
steers nothing.  What a CTU costs the chip is vector-ALU issue: a wave64 VALU instruction occupies its SIMD for 4 cycles, so
    ceiling [CTU/s] = CUs x 4 SIMDs x clock / (VALU wave-instructions per CTU x 4)
at the instruction count the kernel has TODAY on that launch shape (fewer instructions per CTU raise the ceiling; idle waves lower `frac`).
    python tools/issue_json.py <tag> <frames> [<frames> ...]        (reads profiles/<tag>_rd_counters_f<frames>.txt)
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS, SIMDS, CLOCK_HZ, VALU_CYCLES = 256, 4, 2.4e9, 4        # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 VALU instruction issues over 4 cycles


def shape(path, frames, ctus_per_frame=2040):
    txt = open(path).read()
    vals, launches, dur = {}, None, []
    for m in re.finditer(r"hevcdl_rd_frame_kernel\w* \| (\w+) = ([\d.eE+]+) \(sum over (\d+) rows\)", txt):
        vals[m.group(1)] = float(m.group(2))
        launches = int(m.group(3))
    for m in re.finditer(r"^hevcdl_rd_frame_kernel\w* \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", txt, re.M):
        dur.append(float(m.group(3)) * 1e-6)           # average duration of a launch in that pass, ms -> s  (rocpd: microseconds in this view)
    ctus = frames * ctus_per_frame * launches
    valu = vals["SQ_INSTS_VALU"] / ctus
    out = {"frames": frames, "launches_per_pass": launches, "valu_per_ctu": valu, "salu_per_ctu": vals["SQ_INSTS_SALU"] / ctus, "smem_per_ctu": vals["SQ_INSTS_SMEM"] / ctus,
           "lds_per_ctu": vals["SQ_INSTS_LDS"] / ctus, "vmem_rd_per_ctu": vals["SQ_INSTS_VMEM_RD"] / ctus, "vmem_wr_per_ctu": vals["SQ_INSTS_VMEM_WR"] / ctus,
           "lanes_enabled": vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_ACTIVE_INST_VALU"],          # of 64; wave-uniform work runs with every lane on: an upper bound of the useful lanes
           "active_inst_share_of_wave_cycles": vals["SQ_ACTIVE_INST_ANY"] / vals["SQ_WAVE_CYCLES"], "wait_share_of_wave_cycles": vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"],
           "ceiling_ctus_per_s": CUS * SIMDS * CLOCK_HZ / (valu * VALU_CYCLES)}
    if dur:
        out["kernel_s_under_counters"] = sum(dur) / len(dur)
        out["ctus_per_s_under_counters"] = frames * ctus_per_frame / out["kernel_s_under_counters"]
        out["frac_under_counters"] = out["ctus_per_s_under_counters"] / out["ceiling_ctus_per_s"]
    return out


def main():
    tag = sys.argv[1]
    sha = hashlib.sha256(open(os.path.join(ROOT, "hevc-deep-learning-pipeline_amd", "csrc", "rd_kernel.hip"), "rb").read()).hexdigest()[:16]
    out = {"kernel": "hevcdl_rd_frame_kernel", "rd_kernel_sha16": sha, "unit_of_work": "one CTU (SURVEY.md section 8d)",
           "model": "ceiling = %d CUs x %d SIMDs x %.1f GHz / (VALU wave-instructions per CTU x %d issue cycles)" % (CUS, SIMDS, CLOCK_HZ / 1e9, VALU_CYCLES),
           "source": ", ".join("profiles/%s_rd_counters_f%s.txt" % (tag, f) for f in sys.argv[2:]) + " (rocprofv3 --pmc, one pass per counter group, tools/r05_pmc.sh)", "shapes": {}}
    for f in sys.argv[2:]:
        out["shapes"][f] = shape(os.path.join(ROOT, "profiles", "%s_rd_counters_f%s.txt" % (tag, f)), int(f))
    json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_issue.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
