#!/bin/bash
# A/B of the bound-renewal schedule of k_icp (results are exact whatever the schedule) IN THE BENCH LOOP: each argument = hipcc flags of a
# variant ("" = defaults), built into lib/diag_sched, timed with bench.py (100 scans), the list twice, inside one gpurun call.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    DIAG_DIR=diag_sched tools/diag_build.sh icp_kernels $v > /dev/null 2>&1 || { echo "variant [$v] failed to build"; continue; }
    TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_sched timeout 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); sp=d.get('ms_icp_iterate_spread') or {}
        print('[%s]' % '$v', 'value', round(d['value'],1), 'ms_icp_iterate', round(d['ms_icp_iterate'],5), 'max/mean', round(sp.get('max_over_mean',0),3))
"
  done
done
