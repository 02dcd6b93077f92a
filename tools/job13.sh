cd $GRAFT_REPO_ROOT
for v in b0 b8; do python tools/micro_rd.py --lib hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_micro_$v.so --reps 100 2>&1 | grep "n  [48]" | cut -c1-100; done
