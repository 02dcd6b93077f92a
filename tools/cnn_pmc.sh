export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R; export HEVCDL_LIB=${HEVCDL_LIB_AB:-}; [ -z "$HEVCDL_LIB" ] && unset HEVCDL_LIB
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex hevcdl_cnn_ctu_kernel -d gpurun_out/prof/cnnpmc_$i -o c -- python tools/time_cnn.py 64 1 > gpurun_out/prof/cnnpmc_$i.log 2>&1
done
python tools/rocpd_summary.py gpurun_out/prof/cnnpmc_1 gpurun_out/prof/cnnpmc_2 gpurun_out/prof/cnnpmc_3 gpurun_out/prof/cnnpmc_4 gpurun_out/prof/cnnpmc_5 gpurun_out/prof/cnnpmc_6 > gpurun_out/prof/cnnpmc.txt 2>&1
cat gpurun_out/prof/cnnpmc.txt | cut -c1-220 | tail -40
tail -3 gpurun_out/prof/cnnpmc_1.log
