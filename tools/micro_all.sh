#!/bin/bash
# run tools/micro_rd.py on every lib/libhevcdl_hip_micro_*.so named on the command line (names without the prefix); output -> gpurun_out/micro_<name>.txt
for v in "$@"; do python tools/micro_rd.py --lib hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_micro_$v.so --reps 100 > gpurun_out/micro_$v.txt 2>&1; done
