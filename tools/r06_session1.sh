#!/bin/bash
# round 6, first GPU session: the GPU suite on the round's starting kernel, then C3 of BASELINE.json as written (100 frames x 4 QPs, tools/bd_anchor.py)
python -m pytest tests -m gpu -x -q > gpurun_out/r06a_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r06a_pytest_gpu.txt
timeout 1500 python tools/bd_anchor.py --frames 100 --out gpurun_out/r06_c3_bd.json > gpurun_out/r06_c3_bd.log 2>&1; tail -25 gpurun_out/r06_c3_bd.log
