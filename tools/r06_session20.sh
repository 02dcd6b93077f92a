#!/bin/bash
# the faster bitstream writer on the device box: the GPU suite (CLI / pipeline cases code streams with it), then the bench line (its e2e legs)
python -m pytest tests -m gpu -x -q > gpurun_out/r06r_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r06r_pytest_gpu.txt
timeout 1200 python bench.py > gpurun_out/r06r_bench.json 2> gpurun_out/r06r_bench.err; tail -c 300 gpurun_out/r06r_bench.json; tail -2 gpurun_out/r06r_bench.err
