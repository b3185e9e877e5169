#!/bin/bash
# diagnostic: rebuild push_kernels with extra -D flags ($TSD_EXTRA) on the GPU box and run the bench (no parity!)
$GRAFT_REPO_ROOT/tools/diag_build.sh push_kernels $TSD_EXTRA
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$TSD_EXTRA]', d['value'], 'push kernels', d['ms_push_kernels'], 'update', d['roofline']['avg_launch_ms'])"
