set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_metrics.py -m gpu -x -q 2>&1 | tail -3
timeout 1500 python tools/bd_anchor.py --frames 16 --out gpurun_out/r02_c3_bd.json > gpurun_out/bd.log 2>&1
tail -30 gpurun_out/bd.log
