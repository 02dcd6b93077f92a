cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r05c > gpurun_out/j15_profile.log 2>&1
tail -12 gpurun_out/j15_profile.log | cut -c1-200
bash tools/r05_pmc.sh kernel 600
bash tools/r05_pmc.sh kernel 256
