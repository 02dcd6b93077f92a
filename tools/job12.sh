cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do
python tools/time_rd.py 450 600 768 1024 --flags=4 2>&1 | grep frames
python tools/time_rd.py 450 600 768 1024 --flags=2 2>&1 | grep frames
done
