set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py > gpurun_out/prof/r01i_bench.json 2> gpurun_out/prof/r01i_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/r01i_trace -o r01i -- python bench.py --no-cpu-baseline > gpurun_out/prof/r01i_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/r01i_fetch -o r01i -- python bench.py --frames 512 --no-cpu-baseline > gpurun_out/prof/r01i_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/r01i_write -o r01i -- python bench.py --frames 512 --no-cpu-baseline > gpurun_out/prof/r01i_write.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof/r01i_trace gpurun_out/prof/r01i_fetch gpurun_out/prof/r01i_write > gpurun_out/prof/r01i_summary.txt 2>&1
tail -1 gpurun_out/prof/r01i_bench.json | cut -c1-400
cat gpurun_out/prof/r01i_summary.txt
