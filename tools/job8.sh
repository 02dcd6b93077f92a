cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests/test_rd_gpu.py -m gpu -x -q) > gpurun_out/j8_pytest.txt 2>&1
tail -4 gpurun_out/j8_pytest.txt
for L in ab_base libhevcdl_hip ab_base libhevcdl_hip; do
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 600 python tools/time_rd.py 1 256 600 2048 > gpurun_out/j8_time_$L.txt 2>&1
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 300 python tools/time_rd.py 10 --size=1920x1080 >> gpurun_out/j8_time_$L.txt 2>&1
  echo $L; grep "frames\|fault\|Error" gpurun_out/j8_time_$L.txt
done
