#!/bin/bash
# the saturated regime with and without WaveFrontSynchro: 1200 / 2560 frames (rows claimed dynamically have no tail of unequal frames)
timeout 600 python tools/time_rd.py 1200 2560 --wavefront > gpurun_out/r06n_saturated_wavefront.txt 2>&1; grep -a flags gpurun_out/r06n_saturated_wavefront.txt | cut -c1-150
timeout 600 python tools/time_rd.py 1200 2560 > gpurun_out/r06n_saturated_default.txt 2>&1; grep -a flags gpurun_out/r06n_saturated_default.txt | cut -c1-150
