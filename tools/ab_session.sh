#!/bin/bash
# A/B of decision-kernel libraries in one GPU session (boxes differ by 1 - 2 %: only numbers of one session compare).
#   tools/job.sh <tag> <library under hevc-deep-learning-pipeline_amd/lib> ...      (variants: hevcdl_amd.build_ext(defines=(...), out=".../lib/ab_x.so"))
# parity of the decision kernel first, then every library three times over 1 / 600 / 2048 frames of 2160p; times -> gpurun_out/<tag>_time_<lib>.txt
tag=$1; shift
python -m pytest tests/test_rd_gpu.py -x -q > gpurun_out/${tag}_pytest.txt 2>&1; tail -3 gpurun_out/${tag}_pytest.txt
rm -f gpurun_out/${tag}_time_*.txt
for rep in 1 2 3; do
for l in "$@"; do
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l python tools/time_rd.py 1 600 2048 >> gpurun_out/${tag}_time_${l%.so}.txt 2>&1
done
done
for l in "$@"; do echo "== $l"; grep -a 'flags\|rror' gpurun_out/${tag}_time_${l%.so}.txt; done
