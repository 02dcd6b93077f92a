cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/time_rd.py 1 600 1 600 2>&1 | tail -4
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
