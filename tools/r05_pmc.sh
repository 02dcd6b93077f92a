# Round 5 measurement of the decision kernel's instruction stream (review item 2): lane utilisation of the vector ALU and the instruction mix
#   (a) on the launch shapes of the bench: bash tools/r05_pmc.sh kernel 600   (tools/time_rd.py <frames>: two launches; counters summed over both)
#   (b) per leaf routine, with the micro kernel (tools/micro_rd.py: RDOQ / bit counter / transform + RDOQ + bits + inverse at 4..32, one dispatch each): bash tools/r05_pmc.sh micro
# One rocprofv3 --pmc pass per counter set (never together with a trace).  Output: gpurun_out/prof/r05pmc_<what>.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WHAT=${1:-kernel}
FR=${2:-600}
mkdir -p $R/gpurun_out/prof
cd $R
SETS=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_THREAD_CYCLES_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
i=0
dirs=""
for set in "${SETS[@]}"; do
  i=$((i+1))
  if [ "$WHAT" = "micro" ]; then
    timeout 600 rocprofv3 --pmc $set --kernel-include-regex hevcdl_micro_kernel -d gpurun_out/prof/r05m_$i -o c -- python tools/micro_rd.py --reps 50 > gpurun_out/prof/r05m_$i.log 2>&1
    dirs="$dirs gpurun_out/prof/r05m_$i"
  else
    timeout 900 rocprofv3 --pmc $set --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/r05k${FR}_$i -o c -- python tools/time_rd.py $FR > gpurun_out/prof/r05k${FR}_$i.log 2>&1
    dirs="$dirs gpurun_out/prof/r05k${FR}_$i"
  fi
done
if [ "$WHAT" = "micro" ]; then
  python tools/pmc_rows.py $dirs > gpurun_out/prof/r05pmc_micro.txt 2>&1
  tail -40 gpurun_out/prof/r05pmc_micro.txt | cut -c1-220
else
  python tools/rocpd_summary.py $dirs > gpurun_out/prof/r05pmc_k$FR.txt 2>&1
  grep "hevcdl_rd" gpurun_out/prof/r05pmc_k$FR.txt | cut -c1-200
fi
rm -rf $dirs          # (the rocpd databases: summarised above; gpurun_out/ only travels back up to 64 MiB)
