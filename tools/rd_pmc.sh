# SQ counters of the decision kernel on a short launch: bash tools/rd_pmc.sh [frames, default 64]   (on the GPU box, through gpurun)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FR=${1:-64}
mkdir -p $R/gpurun_out/prof
cd $R
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex hevcdl_rd_frame_kernel -d gpurun_out/prof/rdpmc_$i -o c -- python tools/time_rd.py $FR > gpurun_out/prof/rdpmc_$i.log 2>&1
done
python tools/rocpd_summary.py gpurun_out/prof/rdpmc_1 gpurun_out/prof/rdpmc_2 gpurun_out/prof/rdpmc_3 gpurun_out/prof/rdpmc_4 gpurun_out/prof/rdpmc_5 gpurun_out/prof/rdpmc_6 gpurun_out/prof/rdpmc_7 > gpurun_out/prof/rdpmc.txt 2>&1
grep "hevcdl_rd" gpurun_out/prof/rdpmc.txt | cut -c1-200
tail -2 gpurun_out/prof/rdpmc_1.log
