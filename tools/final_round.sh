#!/bin/bash
# End-of-round measurement in one GPU session: the GPU test suite, the round profile (bench line, kernel trace, HBM counter passes), the in-kernel phase profiles
# and the SQ counter passes of the decision kernel.  usage (through gpurun): bash tools/final_round.sh r05d
TAG=${1:-r05d}
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.txt
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; tail -12 gpurun_out/${TAG}_profile.log
python tools/phase_profile.py 3840 2160 600 > gpurun_out/${TAG}_phase_f600.txt 2>&1
python tools/phase_profile.py 3840 2160 1 > gpurun_out/${TAG}_phase_f1.txt 2>&1
bash tools/r05_pmc.sh kernel 600 > gpurun_out/${TAG}_pmc600.log 2>&1; cp gpurun_out/prof/r05pmc_k600.txt gpurun_out/${TAG}_pmc_k600.txt
bash tools/r05_pmc.sh kernel 256 > gpurun_out/${TAG}_pmc256.log 2>&1; cp gpurun_out/prof/r05pmc_k256.txt gpurun_out/${TAG}_pmc_k256.txt
grep hevcdl_rd gpurun_out/${TAG}_pmc_k600.txt | cut -c1-200
