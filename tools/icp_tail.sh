#!/bin/bash
# Which scans of the bench trajectory make the registration's tail, and which steps of those registrations are long?
#   gpurun -- tools/icp_tail.sh [n_scans=220] [--fused] [extra hipcc flags]
# (timeline build of k_icp, the unfused C ABI loop of tests/slam_driver.py so that every registration's per-step record is readable)
cd $GRAFT_REPO_ROOT
N=${1:-220}; shift
MODE=""; if [ "$1" = "--fused" ]; then MODE="--fused"; shift; fi
DIAG_DIR=diag_tl tools/diag_build.sh icp_kernels -DTSD_ICP_TIMELINE -DTSD_ICP_TL_FIRST=0 -DTSD_ICP_TL_STEPS=30 "$@" > /dev/null 2>&1 || { echo "timeline build failed"; exit 1; }
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_tl
python3 tools/icp_tail.py $N $MODE
