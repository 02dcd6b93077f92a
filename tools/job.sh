#!/bin/bash
# one GPU session: the GPU suite + A/B times: tools/job.sh <tag> <lib> <lib> ...
tag=$1; shift
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.txt 2>&1; tail -4 gpurun_out/${tag}_pytest.txt
rm -f gpurun_out/${tag}_time_*.txt
for rep in 1 2 3; do
for l in "$@"; do
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l python tools/time_rd.py 1 600 2048 >> gpurun_out/${tag}_time_${l%.so}.txt 2>&1
done
done
for l in "$@"; do echo "== $l"; grep -a 'flags\|rror' gpurun_out/${tag}_time_${l%.so}.txt; done
