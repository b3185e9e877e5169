#!/bin/bash
# the registration's spread over 200 scans (every dispatch timed) for variants of k_icp, side by side in ONE call:
#   gpurun -- tools/spread_ab.sh "<src>|<flags>" ...      (as tools/icp_ab.sh)
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
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_ab$i python3 bench.py --no-cpu-baseline --no-stream --comparison-passes 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['ms_icp_iterate_spread']
print('[$v] value %.0f | icp mean %.4f p50 %.4f p90 %.4f p99 %.4f max %.4f' % (d['value'], s['mean'], s['p50'], s['p90'], s['p99'], s['max']))"
  done
done
