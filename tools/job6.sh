cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/micro_rd.py --lib hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_micro_base.so --reps 100 > gpurun_out/j6_micro_base.txt 2>&1
python tools/micro_rd.py --reps 100 > gpurun_out/j6_micro_new.txt 2>&1
paste -d'\n' gpurun_out/j6_micro_base.txt gpurun_out/j6_micro_new.txt | cut -c1-150
