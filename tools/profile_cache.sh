#!/bin/bash
# L2 (TCC) and vector-L1 (TCP) counters of the push kernels, separate PMC passes of at most four counters: where the HBM traffic above
# the algorithmic bytes comes from.   usage: tools/profile_cache.sh <tag> [bench args]
set -u
tag=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for ctrs in "TCC_HIT TCC_MISS TCC_READ TCC_WRITE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCP_TOTAL_READ TCP_TOTAL_WRITE TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" "TCC_REQ TCC_ATOMIC TCC_NORMAL_WRITEBACK TCC_NORMAL_EVICT"; do
  i=$((i+1))
  TSD_SQ_COUNTERS="$ctrs" tools/profile_sq.sh ${tag}_c$i "$@" 2>&1 | grep "k_push_update\|k_push_classify\|k_push_halo\|k_raycast\|k_calib"
done
find gpurun_out -name "*counter_collection.csv" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null
