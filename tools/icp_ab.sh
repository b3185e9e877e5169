#!/bin/bash
# A/B of the registration kernel in ONE gpurun call (boxes differ by several per cent).  Each argument is "<source file>|<flags>"
# (source relative to csrc/, empty = icp_kernels.hip), built into lib/diag_icp<i>; the variants are then timed in turn, twice.
cd $GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1)); src=${v%%|*}; fl=${v#*|}
  DIAG_SRC=${src:-icp_kernels.hip} DIAG_DIR=diag_icp$i tools/diag_build.sh icp_kernels $fl > /dev/null 2>&1 || { echo "variant [$v] failed to build"; continue; }
done
for rep in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_icp$i python3 tools/icp_ablate.py "[$v]" 2>/dev/null | tail -1
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_icp$i python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   slam: %.0f scans/s, ms_icp %.4f, max/mean %.2f' % (d['value'], d['ms_icp_iterate'], d['ms_icp_iterate_spread']['max_over_mean']))"
  done
done
