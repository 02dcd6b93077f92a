#!/bin/bash
# usage (GPU box): traffic_exp.sh variant...   -> WRITE_SIZE / FETCH_SIZE of the decision kernel on a 128-frame launch for lib/libhevcdl_hip_<variant>.so
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/tx
QC="--frames 128 --steps 1 --warmup 0 --no-cpu-baseline --no-c2 --saturated-frames 0"
for v in "$@"; do
  export HEVCDL_LIB=$R/hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_$v.so
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf gpurun_out/tx/${v}_$c
    timeout 240 rocprofv3 --pmc $c -d gpurun_out/tx/${v}_$c -o x -- python bench.py $QC > gpurun_out/tx/${v}_$c.log 2>&1
    echo "$v $c rc=$?"
  done
  python tools/rocpd_summary.py gpurun_out/tx/${v}_WRITE_SIZE gpurun_out/tx/${v}_FETCH_SIZE 2>&1 | grep -E "rd_frame" | cut -c1-150
done
