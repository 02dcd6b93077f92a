#!/bin/bash
# A/B of decision-kernel libraries in one GPU session: tools/ab.sh <tag> <lib name under lib/ ...> ; times -> gpurun_out/<tag>_time_<lib>.txt
tag=$1; shift
for rep in 1 2; do
for l in "$@"; do
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l python tools/time_rd.py 1 600 2048 >> gpurun_out/${tag}_time_${l%.so}.txt 2>&1
done
done
for l in "$@"; do echo "== $l"; cat gpurun_out/${tag}_time_${l%.so}.txt; done
