#!/bin/bash
# diagnostic: per-phase cycle stamps of k_raycast inside the closed SLAM loop of bench.py (cfg2)
$GRAFT_REPO_ROOT/tools/diag_build.sh raycast_kernels -DTSD_RC_STAMPS
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from ohm_tsd_slam_amd import capi, facade, synth
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc, start_xy=[0.5 * gc.width, 0.5 * gc.width - 0.21])
N = 120
poses = synth.trajectory(world, N)
scans = synth.scans_for(world, geo, poses)
node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
grid = node.grid()
names = ["clip+coarse", "segments", "cand blocks", "blocks looked at", "march", "normal", "serial", "total"]
grid.profile(True, "raycast"); grid.profile_reset()
for k in range(N):
    node.laser(scans[k], geo.angle_min, geo.angle_increment)
    if k in (10, 60, 119):
        grid.sync()
        tr = np.zeros((256, 8)); grid.lib.tsd_icp_trace(grid.h, tr.ctypes.data_as(capi._dp), 256)
        d = tr.reshape(128, 8)
        print("scan", k, {nm: "%.0f" % d[:, i].mean() for i, nm in enumerate(names)}, "max total %.0f" % d[:, 7].max(),
              "max cand %d" % d[:, 2].max())
ms, n = grid.profile_get("raycast")
print("avg raycast us %.1f over %d" % (1e3 * ms / max(n, 1), n))
node.close()
PY
