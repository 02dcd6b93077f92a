#!/bin/bash
# round 6, last GPU session: the bench line once more (now that the issue / traffic files of this kernel are committed, the line carries them), then the randomized stress
# of the three cooperative forms on the final build
timeout 1200 python bench.py > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err; tail -c 400 gpurun_out/r06h_bench.json; tail -3 gpurun_out/r06h_bench.err
timeout 400 python tools/stress_few_units.py 150 > gpurun_out/r06h_stress_few_units.txt 2>&1; tail -2 gpurun_out/r06h_stress_few_units.txt
timeout 400 python tools/stress_handover.py 150 > gpurun_out/r06h_stress_handover.txt 2>&1; tail -2 gpurun_out/r06h_stress_handover.txt
timeout 400 python tools/stress_wavefront.py 200 > gpurun_out/r06h_stress_wavefront.txt 2>&1; tail -2 gpurun_out/r06h_stress_wavefront.txt
du -sh gpurun_out
