cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
QC="--frames 128 --steps 1 --warmup 0 --no-cpu-baseline --no-c2 --no-e2e --no-latency-floor --saturated-frames 0"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/x1_write -o x1 -- python bench.py $QC > gpurun_out/prof/x1_write.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof/x1_write gpurun_out/prof/x1_write gpurun_out/prof/x1_write 2>&1 | grep -E "hevcdl_rd|##" | cut -c1-200
