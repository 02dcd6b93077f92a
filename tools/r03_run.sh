cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_app_cli.py tests/test_host_abi.py -q -x 2>&1 | tail -4
timeout 900 python tools/bench_cli.py 600 2>&1 | tail -3
