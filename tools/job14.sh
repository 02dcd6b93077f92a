cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/time_rd.py 1 600 2048 2>&1 | grep frames; python tools/time_rd.py 10 --size=1920x1080 2>&1 | grep frames; done
(time python -m pytest tests/test_rd_gpu.py -m gpu -x -q) > gpurun_out/j14_pytest.txt 2>&1
grep -E "passed|failed" gpurun_out/j14_pytest.txt
