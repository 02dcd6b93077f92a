#!/bin/bash
# one GPU session: A/B times of few-unit launches: tools/job.sh <tag> <lib> <lib> ...
tag=$1; shift
rm -f gpurun_out/${tag}_time_*.txt
for rep in 1 2 3; do
for l in "$@"; do
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l python tools/time_rd.py 1 75 150 >> gpurun_out/${tag}_time_${l%.so}.txt 2>&1
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l python tools/time_rd.py 10 --size=1920x1080 >> gpurun_out/${tag}_time_${l%.so}.txt 2>&1
done
done
for l in "$@"; do echo "== $l"; grep -a 'flags\|rror' gpurun_out/${tag}_time_${l%.so}.txt; done
