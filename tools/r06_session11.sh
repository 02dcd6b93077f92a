#!/bin/bash
# where does the command line's time go with WaveFrontSynchro?  (set-up / read / device / host / write / whole run on stderr)
python tools/bench_cli.py 16 > gpurun_out/r06j_bench_cli_16.txt 2>&1; grep -a "^cli" gpurun_out/r06j_bench_cli_16.txt | cut -c1-60,120-520
python tools/bench_cli.py 384 > gpurun_out/r06j_bench_cli_384.txt 2>&1; grep -a "^cli" gpurun_out/r06j_bench_cli_384.txt | cut -c1-60,120-520
