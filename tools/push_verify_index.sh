#!/bin/bash
# k_push_update's fp32 beam-index estimate against the exact formulation, cell by cell (diagnostic build into lib/diag_verify)
DIAG_DIR=diag_verify $GRAFT_REPO_ROOT/tools/diag_build.sh push_kernels -DTSD_PUSH_VERIFY_INDEX
TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_verify python3 $GRAFT_REPO_ROOT/tools/push_verify_index.py
