#!/bin/bash
# usage (GPU box): pmc_exp.sh "<counters>" ...   -> counter values of the decision kernel on a 128-frame launch, one pass per argument
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pmc
QC="--frames 128 --steps 1 --warmup 0 --no-cpu-baseline --no-c2 --saturated-frames 0"
i=0
for c in "$@"; do
  i=$((i+1)); rm -rf gpurun_out/pmc/p$i
  timeout 240 rocprofv3 --pmc $c -d gpurun_out/pmc/p$i -o x -- python bench.py $QC > gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i ($c) rc=$?"
  python tools/rocpd_summary.py gpurun_out/pmc/p$i 2>&1 | grep -E "rd_frame" | cut -c1-150
  rm -rf gpurun_out/pmc/p$i
done
