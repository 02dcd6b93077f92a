cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/phase_profile.py 3840 2160 600 > gpurun_out/prof5_f600.txt 2>&1
python tools/phase_profile.py 3840 2160 1 > gpurun_out/prof5_f1.txt 2>&1
cut -c1-110 gpurun_out/prof5_f600.txt | head -70
