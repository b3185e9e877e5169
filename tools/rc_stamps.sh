#!/bin/bash
# diagnostic: rebuild raycast_kernels with per-phase cycle stamps on the GPU box and print the phase shares
$GRAFT_REPO_ROOT/tools/diag_build.sh raycast_kernels -DTSD_RC_STAMPS
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests.test_gpu_parity import build_map
from tests import helpers as H
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
pose, _ = H.sensor_pose(world, 5)
rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
for rep in range(2):
    z = np.zeros(1024); 
    import ctypes as C
    dg.profile(True, "raycast"); dg.profile_reset()
    # zero the debug buffer through a dummy icp trace read is not possible; accept accumulation across reps
    cd, nd, md, cnt = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    ms, n = dg.profile_get("raycast")
    tr = np.zeros((256, 8)); dg.lib.tsd_icp_trace(dg.h, tr.ctypes.data_as(capi._dp), 256)
    d = tr[:128]
    names = ["clip+coarse", "segments", "cand blocks", "blocks looked at", "march", "normal", "serial", "total"]
    print("kernel us %.1f hits %d" % (1e3 * ms / max(n, 1), cnt), "avg cycles over 128 sampled beams:",
          {nm: "%.0f" % d[:, i].mean() for i, nm in enumerate(names)}, "max total %.0f" % d[:, 7].max())
PY
