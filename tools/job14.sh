cd $GRAFT_REPO_ROOT
for i in 1 2; do for L in ab_tu0 ab_tu4 libhevcdl_hip; do
  echo $L; HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so python tools/time_rd.py 1 600 2048 2>&1 | grep frames
done; done
(time python -m pytest tests/test_rd_gpu.py -m gpu -x -q) > gpurun_out/j14_pytest.txt 2>&1
grep -E "passed|failed" gpurun_out/j14_pytest.txt
