#!/bin/bash
# diagnostic (GPU box): k_push_update variants (-DTSD_UPDATE_WPS=<waves per SIMD> -DTSD_UPDATE_CB=<cells per lane and pass of phase C>)
# built into lib/diag_<name>, timed on the push-only benches and the slam bench; no parity here (the product build has the tests)
cd $GRAFT_REPO_ROOT
# a variant is "<flags>" or "<source file beside push_kernels.hip>|<flags>" (an A/B against a previous version of the file, same call)
for v in "$@"; do
  name=$(echo "$v" | tr -d ' =-|._' | tr 'A-Z' 'a-z')
  src=push_kernels.hip; fl=$v
  case "$v" in *"|"*) src=${v%%|*}; fl=${v#*|};; esac
  DIAG_SRC=$src DIAG_DIR=diag_$name tools/diag_build.sh push_kernels $fl > /dev/null 2>&1 || { echo "variant [$v] failed to build"; continue; }
  export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_$name
  for w in "cfg3 comb" "cfg3 pillars" "cfg2 comb" "cfg2 pillars"; do set -- $w
    python3 bench.py --config $1 --scene $2 --mode push --steps 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['stages_ms']
print('variant [$v] $1/$2 push: update us %.1f frac %.3f | classify %.1f halo %.1f | step us %.1f' % (1e3*r['avg_launch_ms'], r['frac'], 1e3*s['push_classify'], 1e3*s['push_halo'], 1e3*d['ms_per_step']))"
  done
  python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-second-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant [$v] cfg2 slam: %.0f scans/s, update us %.1f frac %.3f' % (d['value'], 1e3*r['avg_launch_ms'], r['frac']))"
  unset TSD_LIB_DIR
done
