#!/bin/bash
# diagnostic: rebuild push_kernels with extra -D flags ($TSD_EXTRA) into lib/diag on the GPU box and time the push kernels
# (no parity!) on the slam bench (cfg2) and the push-only benches (cfg3 comb / cfg2 pillars)
$GRAFT_REPO_ROOT/tools/diag_build.sh push_kernels $TSD_EXTRA > /dev/null
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$TSD_EXTRA] cfg2 slam', round(d['value']), 'update us', round(1e3*d['roofline']['avg_launch_ms'],1))"
for w in "cfg3 comb" "cfg3 pillars" "cfg2 pillars"; do set -- $w
python3 bench.py --config $1 --scene $2 --mode push --steps 100 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$TSD_EXTRA] $1 $2 push: update us', round(1e3*d['roofline']['avg_launch_ms'],1), 'frac', round(d['roofline']['frac'],3))"
done
