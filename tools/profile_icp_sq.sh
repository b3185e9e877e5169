cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY"; do
  i=$((i+1))
  TSD_SQ_COUNTERS="$ctrs" tools/profile_sq.sh r6_icp_c$i 2>&1 | grep "k_icp"
done
find gpurun_out -name "*counter_collection.csv" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null
