cd $GRAFT_REPO_ROOT
timeout 300 python tools/time_rd.py 1 600 2>&1 | tail -2
python tools/phase_profile.py 3840 2160 1 > gpurun_out/r03_phase_f1.txt 2>&1
cat gpurun_out/r03_phase_f1.txt | grep -E "IDLE|chain owner|MASTER|est_|region|restarts|task:"
timeout 900 python -m pytest tests/test_rd_gpu.py -x -q -m gpu 2>&1 | tail -4
