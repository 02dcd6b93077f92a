cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rd_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_rdtests.txt
cat gpurun_out/r03_rdtests.txt
timeout 300 python tools/time_rd.py 1 600 > gpurun_out/r03_time.txt 2>&1
cat gpurun_out/r03_time.txt
