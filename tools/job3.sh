cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in libhevcdl_hip ab_ringonly ab_nowakeup ab_noahead ab_noprefetch; do
  echo "== $L"
  for cfg in "--flags=1 1" "--flags=1 8" "256"; do
    HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 300 python tools/time_rd.py $cfg > gpurun_out/j3.txt 2>&1
    echo "   [$cfg] rc=$? $(grep -c 'frames' gpurun_out/j3.txt) lines: $(grep 'frames\|fault' gpurun_out/j3.txt | tail -2 | tr '\n' ' ')"
  done
done
