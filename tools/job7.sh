cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/micro_rd.py --lib hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_micro_base.so --reps 100 > gpurun_out/j7_micro_base.txt 2>&1
python tools/micro_rd.py --reps 100 > gpurun_out/j7_micro_new.txt 2>&1
paste -d'\n' gpurun_out/j7_micro_base.txt gpurun_out/j7_micro_new.txt | cut -c1-118
for L in ab_base ab_noopre libhevcdl_hip ab_base ab_noopre libhevcdl_hip; do
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 600 python tools/time_rd.py 1 256 600 2048 > gpurun_out/j7_time_$L.txt 2>&1
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 300 python tools/time_rd.py 10 --size=1920x1080 >> gpurun_out/j7_time_$L.txt 2>&1
  echo $L; grep "frames\|fault\|Error" gpurun_out/j7_time_$L.txt
done
(time python -m pytest tests -m gpu -x -q) > gpurun_out/j7_pytest.txt 2>&1
tail -4 gpurun_out/j7_pytest.txt
