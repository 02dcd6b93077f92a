#!/usr/bin/env python3
"""Regenerate the measured sections of DESIGN.md (4.1, 4.2, 5, 7, 8) from tools/doc_templates/*.md and a bench line:
    python tools/regen_design.py profiles/r04f_bench_c4_f600.json
Every @KEY@ of the templates is a number of that line (formatted below); the sections are spliced between their headers, the rest of DESIGN.md is left alone."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rc, sp, c2, sat = d["roofline_cnn"], d["scale_projection"], d["c2"], d["saturated"]
keys = {
    "VALUE": "%.1f" % (d["value"] / 1e3), "STEP": "%.2f" % (d["ms_per_step"] / 1e3), "RDMS": "%.2f" % (d["roofline"]["kernel_ms"] / 1e3),
    "CNNS": "%.3f" % (d["roofline"]["cnn_kernel_ms"] / 1e3), "ACH": "%.2f" % d["roofline"]["achieved"], "FRAC": "%.2e" % d["roofline"]["frac"],
    "CNNEX": "%.0f" % rc["executed_tflops"], "CNNFRAC": "%.2f" % rc["frac"], "CNNUS": "%.0f" % rc["useful_tflops"], "CNNMS": "%.0f" % rc["kernel_ms"],
    "HEADMS": "%.0f" % rc["head_kernel_ms"], "CNNMF": "%.0f" % rc["executed_mflop_per_ctu"], "CNNRATE": "%.2f" % (d["config"]["frames"] * d["config"]["ctus_per_frame"] / rc["kernel_ms"] / 1e3),
    "FLOOR": "%.2f" % d["latency_floor_s"], "SHARE8": "%.2f" % d["share_8gpu_s"],
    "P1": "%.2f" % sp["seconds"]["1"], "P2": "%.2f" % sp["seconds"]["2"], "P4": "%.2f" % sp["seconds"]["4"], "P8": "%.2f" % sp["seconds"]["8"],
    "V1": "%.0f" % (sp["value"]["1"] / 1e3), "V2": "%.0f" % (sp["value"]["2"] / 1e3), "V4": "%.0f" % (sp["value"]["4"] / 1e3), "V8": "%.0f" % (sp["value"]["8"] / 1e3),
    "S8": "%.2f" % (sp["value"]["8"] / sp["value"]["1"]),
    "SAT": "%.1f" % (sat["value"] / 1e3), "SATCU": "%.0f" % sat["per_cu"], "SATRD": "%.1f" % (sat["kernel_ms"] / 1e3),
    "SATRDV": "%.1f" % (sat["frames_per_gpu"] * d["config"]["ctus_per_frame"] / sat["kernel_ms"]),
    "C2MS": "%.3f" % (c2["ms_per_step"] / 1e3), "C2R": "%.2f" % c2["gpu_over_cpu"], "C2CPU": "%.2f" % (10 * 510 / c2["cpu_reference"]["value"]),
    "CPU": "%.1f" % (d["cpu_baseline"]["value"] / 1e3), "CPU1": "%.0f" % d["cpu_baseline"]["one_process"]["value"],
    "E2E": "%.0f" % (d["e2e"]["value"] / 1e3), "E2EP": "%.1f" % d["e2e"]["pictures_per_s"],
    "LINE": os.path.relpath(os.path.abspath(sys.argv[1]), ROOT),
}


def fill(name):
    t = open(os.path.join(ROOT, "tools", "doc_templates", name)).read()
    for k, v in keys.items():
        t = t.replace("@%s@" % k, v)
    assert "@" not in "".join(w for w in t.split() if w.startswith("@") and w.endswith("@") and w[1:-1].isupper()), "unfilled key in " + name
    return t.rstrip("\n") + "\n\n"


path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()


def splice(s, start, end, new):
    i0 = s.index(start)
    i1 = s.index(end, i0 + len(start)) if end else len(s)
    return s[:i0] + new + s[i1:]


s = splice(s, "### 4.1 ", "### 4.2 ", fill("new41.md"))
s = splice(s, "### 4.2 ", "### 4.3 ", fill("new42.md"))
s = splice(s, "## 5. Measurement", "## 5c. ", fill("new5.md"))
s = splice(s, "## 7. Where the time is now", None, fill("new78.md"))
open(path, "w").write(s.rstrip("\n") + "\n")
print("DESIGN.md regenerated from", sys.argv[1])
