#!/bin/bash
# one GPU session: parity of the decision kernel, micro-benchmark of the leaves (new / base), A/B times
tag=$1
python -m pytest tests/test_rd_gpu.py -x -q > gpurun_out/${tag}_pytest.txt 2>&1; tail -3 gpurun_out/${tag}_pytest.txt
python tools/micro_rd.py --lib hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_micro.so --reps 100 > gpurun_out/${tag}_micro_new.txt 2>&1
python tools/micro_rd.py --lib hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_micro_base.so --reps 100 > gpurun_out/${tag}_micro_base.txt 2>&1
grep rdoq gpurun_out/${tag}_micro_base.txt; grep rdoq gpurun_out/${tag}_micro_new.txt
rm -f gpurun_out/${tag}_time_*.txt
bash tools/ab.sh $tag ab_base.so libhevcdl_hip.so
