cd $GRAFT_REPO_ROOT
timeout 1700 python tools/bd_anchor.py --frames 100 --out gpurun_out/r03_c3_bd.json > gpurun_out/r03_c3_bd.log 2>&1
tail -30 gpurun_out/r03_c3_bd.log
