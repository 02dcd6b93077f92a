cd $GRAFT_REPO_ROOT
for i in 1 2; do for L in libhevcdl_hip ab_tu16 ab_tu16b8; do
  echo $L; HEVCDL_LIB=$GRAFT_REPO_ROOT/hevc-deep-learning-pipeline_amd/lib/$L.so python tools/time_rd.py 1 600 2048 2>&1 | grep frames
done; done
