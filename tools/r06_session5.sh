#!/bin/bash
# round 6, fifth GPU session: the bench line with the wavefront leg, the two-rank path, the wavefront tests on the final claim rule
timeout 900 python -m pytest tests/test_rd_gpu.py -x -q -k "wavefront or two_ranks" > gpurun_out/r06e_pytest.txt 2>&1; tail -5 gpurun_out/r06e_pytest.txt
timeout 1200 python bench.py > gpurun_out/r06e_bench.json 2> gpurun_out/r06e_bench.err; tail -c 6000 gpurun_out/r06e_bench.json; tail -5 gpurun_out/r06e_bench.err
