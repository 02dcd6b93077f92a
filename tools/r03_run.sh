cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
bash tools/profile_round.sh r03c > gpurun_out/prof/r03c_profile_round.log 2>&1
tail -3 gpurun_out/prof/r03c_profile_round.log | cut -c1-300
python tools/phase_profile.py 3840 2160 1 > gpurun_out/prof/r03c_phase_f1.txt 2>&1
python tools/phase_profile.py 3840 2160 600 > gpurun_out/prof/r03c_phase_f600.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
