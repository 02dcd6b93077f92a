# Round profile: bench line, rocprofv3 kernel trace of the same command, and the two HBM counter passes (one counter per pass, the decision kernel only) on the very
# launch shape that is timed: 600 frames (the ten-wave build, frames migrating between workgroups).
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh r04
set -x
TAG=${1:-r04}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py --steps 3 --warmup 1 > gpurun_out/prof/${TAG}_bench.json 2> gpurun_out/prof/${TAG}_bench.err
Q="--steps 1 --warmup 0 --no-cpu-baseline --no-c2 --no-e2e --no-latency-floor --no-label-check --no-projection --no-wavefront --saturated-frames 0"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/${TAG}_trace -o ${TAG} -- python bench.py $Q > gpurun_out/prof/${TAG}_trace.log 2>&1
QC="--frames 600 $Q"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/${TAG}_fetch -o ${TAG} -- python bench.py $QC > gpurun_out/prof/${TAG}_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/${TAG}_write -o ${TAG} -- python bench.py $QC > gpurun_out/prof/${TAG}_write.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof/${TAG}_trace gpurun_out/prof/${TAG}_fetch gpurun_out/prof/${TAG}_write > gpurun_out/prof/${TAG}_summary.txt 2>&1
sha256sum hevc-deep-learning-pipeline_amd/csrc/rd_kernel.hip | cut -c1-16 > gpurun_out/prof/${TAG}_rd_kernel_sha16.txt
tail -1 gpurun_out/prof/${TAG}_bench.json | cut -c1-300
grep -E 'hevcdl|##' gpurun_out/prof/${TAG}_summary.txt | cut -c1-160
# the rocpd databases are tens of MB each and gpurun_out/ only travels back up to 64 MiB: keep the summaries, drop the databases
rm -rf gpurun_out/prof/${TAG}_trace gpurun_out/prof/${TAG}_fetch gpurun_out/prof/${TAG}_write
