export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_WAVE[A-Z_0-9]*\|TCP_[A-Z_]*STALL[A-Z_]*" | sort -u | tr '\n' ' ' | cut -c1-3000
echo
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/rdpmcb_$i -o c -- python tools/time_rd.py 256 > gpurun_out/prof/rdpmcb_$i.log 2>&1
  tail -3 gpurun_out/prof/rdpmcb_$i.log | cut -c1-200
done
python tools/rocpd_summary.py gpurun_out/prof/rdpmcb_1 gpurun_out/prof/rdpmcb_2 gpurun_out/prof/rdpmcb_3 2>&1 | grep "hevcdl_rd" | cut -c1-200
