cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 tools/wakeup_probe.hip -o /tmp/wakeup_probe 2>/dev/null && /tmp/wakeup_probe > gpurun_out/j2_wakeup.txt 2>&1
cat gpurun_out/j2_wakeup.txt
python tools/cnn_err.py > gpurun_out/j2_cnn_err.txt 2>&1; cat gpurun_out/j2_cnn_err.txt
for L in ab_nobell libhevcdl_hip ab_prio ab_nobell libhevcdl_hip; do
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so python tools/time_rd.py 1 16 75 256 600 > gpurun_out/j2_time_$L.txt 2>&1
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so python tools/time_rd.py 10 --size=1920x1080 >> gpurun_out/j2_time_$L.txt 2>&1
  echo $L; grep frames gpurun_out/j2_time_$L.txt
done
(time python -m pytest tests -m gpu -x -q) > gpurun_out/j2_pytest.txt 2>&1
tail -5 gpurun_out/j2_pytest.txt
