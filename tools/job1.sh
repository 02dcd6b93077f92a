cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
(time python -m pytest tests/test_cnn_gpu.py tests/test_app_cli.py -m gpu -x -q) > gpurun_out/j1_pytest.txt 2>&1
tail -5 gpurun_out/j1_pytest.txt
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
grep -c . gpurun_out/prof/counters_list.txt
bash tools/r05_pmc.sh micro
bash tools/r05_pmc.sh kernel 600
