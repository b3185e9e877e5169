#!/bin/bash
# A/B of registration-kernel variants on three fixed inputs (tools/icp_repeat.py: 100 registrations each, HIP-event dispatch time, results
# checked for determinism and against the oracle), every variant built into lib/diag_rep and timed in turn, the whole list twice, inside
# ONE gpurun call (the pool's boxes differ by several per cent).  Each argument = the hipcc flags of a variant ("" = the defaults).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    DIAG_DIR=diag_rep tools/diag_build.sh icp_kernels $v > /dev/null 2>&1 || { echo "variant [$v] failed to build"; continue; }
    echo "== [$v]"
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_rep python3 tools/icp_repeat.py 100 2>/dev/null | grep input
  done
done
