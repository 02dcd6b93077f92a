#!/bin/bash
# round 6: the new fixture / pipeline / CLI cases on the device, then the command line end to end (file in, stream + reconstruction out): default cfg, tiles, WaveFrontSynchro
timeout 900 python -m pytest tests -x -q -m gpu -k "w192_q27_k or sharded_encode or (cli and w)" > gpurun_out/r06i_pytest.txt 2>&1; tail -4 gpurun_out/r06i_pytest.txt
timeout 900 python tools/bench_cli.py 128 > gpurun_out/r06i_bench_cli_128.txt 2>&1; grep -a "^cli" gpurun_out/r06i_bench_cli_128.txt | cut -c1-250
timeout 1200 python tools/bench_cli.py 384 > gpurun_out/r06i_bench_cli_384.txt 2>&1; grep -a "^cli" gpurun_out/r06i_bench_cli_384.txt | cut -c1-250
