#!/bin/bash
# round 6, second GPU session: WaveFrontSynchro on the device for the first time -- parity first (every run under its own timeout: rows wait for each other), then timings
timeout 900 python -m pytest tests/test_rd_gpu.py -x -q -k "wavefront or w256 or w200 or w64 or w128 or w416 or w384" > gpurun_out/r06b_pytest_wpp.txt 2>&1; tail -15 gpurun_out/r06b_pytest_wpp.txt
timeout 600 python tools/time_rd.py 1 2 75 600 --wavefront > gpurun_out/r06b_time_wpp.txt 2>&1; cat gpurun_out/r06b_time_wpp.txt | tail -6
timeout 300 python tools/time_rd.py 10 --size=1920x1080 --wavefront > gpurun_out/r06b_time_wpp_1080.txt 2>&1; tail -2 gpurun_out/r06b_time_wpp_1080.txt
timeout 600 python tools/time_rd.py 1 2 75 600 > gpurun_out/r06b_time_plain.txt 2>&1; tail -4 gpurun_out/r06b_time_plain.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r06b_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r06b_pytest_gpu.txt
