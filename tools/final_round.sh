#!/bin/bash
# End-of-round measurement in one GPU session: the GPU test suite, the round profile (bench line, kernel trace, HBM counter passes), the in-kernel phase profiles
# and the SQ counter passes of the decision kernel (timed 600-frame launch, 256-frame launch, the saturated 2560-frame launch).  usage (through gpurun): bash tools/final_round.sh r06q
# afterwards, here: cp the summaries into profiles/, python tools/traffic_json.py <tag>; python tools/issue_json.py <tag> 600 256 2560
TAG=${1:-r06q}
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.txt
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; tail -12 gpurun_out/${TAG}_profile.log
python tools/phase_profile.py 3840 2160 600 > gpurun_out/${TAG}_phase_f600.txt 2>&1
python tools/phase_profile.py 3840 2160 1 > gpurun_out/${TAG}_phase_f1.txt 2>&1
PROF_WAVEFRONT=1 python tools/phase_profile.py 3840 2160 600 > gpurun_out/${TAG}_phase_wavefront_f600.txt 2>&1
PROF_WAVEFRONT=1 python tools/phase_profile.py 3840 2160 1 > gpurun_out/${TAG}_phase_wavefront_f1.txt 2>&1
bash tools/r05_pmc.sh kernel 600 > gpurun_out/${TAG}_pmc600.log 2>&1; cp gpurun_out/prof/r05pmc_k600.txt gpurun_out/${TAG}_pmc_k600.txt
bash tools/r05_pmc.sh kernel 256 > gpurun_out/${TAG}_pmc256.log 2>&1; cp gpurun_out/prof/r05pmc_k256.txt gpurun_out/${TAG}_pmc_k256.txt
bash tools/r05_pmc.sh kernel 2560 > gpurun_out/${TAG}_pmc2560.log 2>&1; cp gpurun_out/prof/r05pmc_k2560.txt gpurun_out/${TAG}_pmc_k2560.txt
timeout 300 python tools/time_rd.py 1 75 600 --tools=0x6b > gpurun_out/${TAG}_time_tools_build.txt 2>&1; timeout 300 python tools/time_rd.py 1 75 600 >> gpurun_out/${TAG}_time_tools_build.txt 2>&1
grep hevcdl_rd gpurun_out/${TAG}_pmc_k600.txt | cut -c1-200
du -sh gpurun_out
