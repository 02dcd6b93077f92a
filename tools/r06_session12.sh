#!/bin/bash
# the driver's round-end sequence on the final tree: build check is done in the container; here the GPU suite, smoke(), and a short bench
python -m pytest tests -m gpu -x -q > gpurun_out/r06k_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r06k_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06k_smoke.txt 2>&1; tail -1 gpurun_out/r06k_smoke.txt
