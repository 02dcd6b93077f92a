cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/micro_rd.py --lib hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_micro_base.so --reps 100 > gpurun_out/j11_micro_base.txt 2>&1
python tools/micro_rd.py --reps 100 > gpurun_out/j11_micro_new.txt 2>&1
paste -d'\n' gpurun_out/j11_micro_base.txt gpurun_out/j11_micro_new.txt | grep "rdoq  " | cut -c1-118
(time python -m pytest tests/test_rd_gpu.py -m gpu -x -q) > gpurun_out/j11_pytest.txt 2>&1
grep -E "passed|failed" gpurun_out/j11_pytest.txt
for i in 1 2; do python tools/time_rd.py 1 600 2048 2>&1 | grep frames; python tools/time_rd.py 10 --size=1920x1080 2>&1 | grep frames; done
