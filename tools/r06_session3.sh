#!/bin/bash
# round 6, third GPU session: WaveFrontSynchro with rows claimed as they become startable
timeout 900 python -m pytest tests/test_rd_gpu.py -x -q -k "wavefront or w256 or w200 or w64 or w128 or w416 or w384" > gpurun_out/r06c_pytest_wpp.txt 2>&1; tail -5 gpurun_out/r06c_pytest_wpp.txt
timeout 600 python tools/time_rd.py 1 2 75 150 300 600 --wavefront > gpurun_out/r06c_time_wpp.txt 2>&1; cat gpurun_out/r06c_time_wpp.txt | tail -7
timeout 300 python tools/time_rd.py 10 --size=1920x1080 --wavefront > gpurun_out/r06c_time_wpp_1080.txt 2>&1; tail -2 gpurun_out/r06c_time_wpp_1080.txt
timeout 400 python tools/stress_wavefront.py 240 > gpurun_out/r06c_stress_wpp.txt 2>&1; tail -3 gpurun_out/r06c_stress_wpp.txt
