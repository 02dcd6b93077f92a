#!/bin/bash
# the per-CTU session under WaveFrontSynchro, then -- only if the suite is green -- the round profile on the kernel as it now stands
python -m pytest tests -m gpu -x -q > gpurun_out/r06q_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r06q_pytest_gpu.txt
if grep -q " failed" gpurun_out/r06q_pytest_gpu.txt; then echo "suite not green: no profile"; exit 1; fi
sed -i 's/^python -m pytest tests -m gpu.*$/echo "(suite: r06q_pytest_gpu.txt)"/' tools/final_round.sh
bash tools/final_round.sh r06q
