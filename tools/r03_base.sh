cd $GRAFT_REPO_ROOT
python tools/phase_profile.py 3840 2160 1 > gpurun_out/r03_base_phase_f1.txt 2>&1
tail -12 gpurun_out/r03_base_phase_f1.txt
