#!/bin/bash
# A/B of one translation unit in ONE gpurun call (the pool's boxes differ by several per cent): tools/tu_ab.sh <tu> "<src>|<flags>" ...
# (src beside <tu>.hip in csrc/, empty = <tu>.hip itself).  Each variant -> lib/diag_ab<i>; then the slam bench in turn, three times.
cd $GRAFT_REPO_ROOT
tu=$1; shift
i=0
for v in "$@"; do
  i=$((i+1)); src=${v%%|*}; fl=${v#*|}
  DIAG_SRC=${src:-$tu.hip} DIAG_DIR=diag_ab$i tools/diag_build.sh $tu $fl > /dev/null 2>&1 || { echo "variant [$v] failed to build"; continue; }
done
for rep in 1 2 3; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_ab$i python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('[$v] slam: %.0f scans/s | raycast %.2f icp %.2f classify %.2f update %.2f halo %.2f us' % (d['value'], 1e3*s['raycast'], 1e3*s['icp'], 1e3*s['push_classify'], 1e3*s['push_update'], 1e3*s['push_halo']))"
  done
done
