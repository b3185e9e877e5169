cd $GRAFT_REPO_ROOT
run() { python3 bench.py --robots 8 --no-cpu-baseline 2>gpurun_out/mr.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.0f scans/s, batch %.2f, icp %.1f us, update %.1f' % (d['value'], d['config']['scans_per_batch'], 1e3*d['ms_icp_iterate'], 1e3*d['stages_ms']['push_update']))"; grep -m2 "probe" gpurun_out/mr.err; }
TSD_BATCH_VERBOSE=1 run "default"
TSD_BATCH_VERBOSE=1 run "default again"
TSD_BATCH_EVENT_WAIT=1 run "event waits"
DIAG_DIR=diag_w2 tools/diag_build.sh push_kernels -DTSD_UPDATE_WPS=2 > /dev/null 2>&1
TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_w2 run "update WPS=2 (2 workgroups per CU)"
DIAG_DIR=diag_w3 tools/diag_build.sh push_kernels -DTSD_UPDATE_WPS=3 > /dev/null 2>&1
TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_w3 run "update WPS=3"
