cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
bash tools/profile_round.sh r03d > gpurun_out/prof/r03d_profile_round.log 2>&1
tail -3 gpurun_out/prof/r03d_profile_round.log | cut -c1-300
