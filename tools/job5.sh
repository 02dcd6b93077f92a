cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in ab_nobell libhevcdl_hip ab_prio ab_inltu ab_nobell libhevcdl_hip ab_prio ab_inltu; do
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 600 python tools/time_rd.py 1 75 256 600 2048 > gpurun_out/j5_time_$L.txt 2>&1
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 300 python tools/time_rd.py 10 --size=1920x1080 >> gpurun_out/j5_time_$L.txt 2>&1
  echo $L; grep "frames\|fault\|Error" gpurun_out/j5_time_$L.txt
done
