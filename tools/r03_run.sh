cd $GRAFT_REPO_ROOT
timeout 300 python tools/time_rd.py 1 600 2>&1 | tail -2
timeout 900 python -m pytest tests/test_rd_gpu.py -q -x -m gpu 2>&1 | tail -4
