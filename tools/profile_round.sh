# Round profile: bench line, rocprofv3 kernel trace and the two HBM counter passes of the same command (C4: 600 frames of 2160p).
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh r02a
set -x
TAG=${1:-r03}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py --steps 3 --warmup 1 > gpurun_out/prof/${TAG}_bench.json 2> gpurun_out/prof/${TAG}_bench.err
Q="--steps 1 --warmup 0 --no-cpu-baseline --no-c2 --no-e2e --no-latency-floor --saturated-frames 0"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/${TAG}_trace -o ${TAG} -- python bench.py $Q > gpurun_out/prof/${TAG}_trace.log 2>&1
# counter passes on a 192-frame launch (more than two thirds of the CUs: the regime of the bench launch, second passes local): under --pmc the 600-frame launch did not finish within 20 minutes (WRITE_SIZE pass, round 2)
QC="--frames 192 --steps 1 --warmup 0 --no-cpu-baseline --no-c2 --no-e2e --no-latency-floor --saturated-frames 0"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/${TAG}_fetch -o ${TAG} -- python bench.py $QC > gpurun_out/prof/${TAG}_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/${TAG}_write -o ${TAG} -- python bench.py $QC > gpurun_out/prof/${TAG}_write.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof/${TAG}_trace gpurun_out/prof/${TAG}_fetch gpurun_out/prof/${TAG}_write > gpurun_out/prof/${TAG}_summary.txt 2>&1
sha256sum hevc-deep-learning-pipeline_amd/csrc/rd_kernel.hip | cut -c1-16 > gpurun_out/prof/${TAG}_rd_kernel_sha16.txt
tail -1 gpurun_out/prof/${TAG}_bench.json | cut -c1-300
grep -E 'hevcdl|##' gpurun_out/prof/${TAG}_summary.txt | cut -c1-160
