# HBM counter passes of the decision kernel on a launch in the regime of the timed one (more frames than CUs: masters share workgroups, frames migrate):
# usage (on the GPU box, through gpurun): bash tools/traffic_pass.sh <tag> [frames, default 300]
TAG=${1:-r04}; FR=${2:-300}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
QC="--frames $FR --steps 1 --warmup 0 --no-cpu-baseline --no-c2 --no-e2e --no-latency-floor --no-label-check --no-projection --saturated-frames 0"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/${TAG}_fetch -o ${TAG} -- python bench.py $QC > gpurun_out/prof/${TAG}_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/${TAG}_write -o ${TAG} -- python bench.py $QC > gpurun_out/prof/${TAG}_write.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof/${TAG}_fetch gpurun_out/prof/${TAG}_write > gpurun_out/prof/${TAG}_counters.txt 2>&1
grep -E 'hevcdl|##' gpurun_out/prof/${TAG}_counters.txt | cut -c1-200
tail -2 gpurun_out/prof/${TAG}_fetch.log | cut -c1-300
