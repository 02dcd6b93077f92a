cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for L in ab_nobell libhevcdl_hip ab_sleep24 ab_prio ab_nobell libhevcdl_hip; do
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 600 python tools/time_rd.py 1 16 75 256 600 > gpurun_out/j4_time_$L.txt 2>&1
  HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 300 python tools/time_rd.py 10 --size=1920x1080 >> gpurun_out/j4_time_$L.txt 2>&1
  echo $L; grep "frames\|fault" gpurun_out/j4_time_$L.txt
done
(time python -m pytest tests -m gpu -x -q) > gpurun_out/j4_pytest.txt 2>&1
tail -5 gpurun_out/j4_pytest.txt
for L in ab_nobell libhevcdl_hip; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so timeout 600 rocprofv3 --pmc $set --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/j4_${L}_$i -o c -- python tools/time_rd.py 256 > gpurun_out/prof/j4_${L}_$i.log 2>&1
  done
  echo "== PMC 256 frames $L"; python tools/rocpd_summary.py gpurun_out/prof/j4_${L}_1 gpurun_out/prof/j4_${L}_2 | grep "hevcdl_rd" | cut -c1-120
done
