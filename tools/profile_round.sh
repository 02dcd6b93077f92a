set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py > gpurun_out/prof/r01k_bench.json 2> gpurun_out/prof/r01k_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/r01k_trace -o r01k -- python bench.py --no-cpu-baseline > gpurun_out/prof/r01k_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/r01k_fetch -o r01k -- python bench.py --frames 512 --no-cpu-baseline > gpurun_out/prof/r01k_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/r01k_write -o r01k -- python bench.py --frames 512 --no-cpu-baseline > gpurun_out/prof/r01k_write.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof/r01k_trace gpurun_out/prof/r01k_fetch gpurun_out/prof/r01k_write > gpurun_out/prof/r01k_summary.txt 2>&1
tail -1 gpurun_out/prof/r01k_bench.json | cut -c1-400
cat gpurun_out/prof/r01k_summary.txt
