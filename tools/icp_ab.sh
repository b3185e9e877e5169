#!/bin/bash
# A/B of k_icp variants in ONE gpurun call (the pool's boxes differ by several per cent): tools/icp_ab.sh "<src>|<flags>" ...
# (src beside icp_kernels.hip in csrc/, empty = icp_kernels.hip itself).  Each variant -> lib/diag_ab<i>; then, twice in turn,
# 100 registrations of three fixed inputs (tools/icp_repeat.py: dispatch time, determinism, oracle check) and the SLAM bench.
cd $GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1)); src=${v%%|*}; fl=${v#*|}
  DIAG_SRC=${src:-icp_kernels.hip} DIAG_DIR=diag_ab$i tools/diag_build.sh icp_kernels $fl > /dev/null 2>&1 || { echo "variant [$v] failed to build"; continue; }
done
for rep in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    echo "[$v]"
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_ab$i python3 tools/icp_repeat.py 100 2>&1 | grep "^input" | sed 's/, max .*//; s/100 runs, //'
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_ab$i python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('   slam: %.0f scans/s | icp %.2f us' % (d['value'], 1e3*s['icp']))"
  done
done
