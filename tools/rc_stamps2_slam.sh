#!/bin/bash
# diagnostic: cycles of EVERY beam of k_raycast (up to the end of the march) inside the SLAM loop; serial-chain beams flagged
$GRAFT_REPO_ROOT/tools/diag_build.sh raycast_kernels -DTSD_RC_STAMPS2
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from ohm_tsd_slam_amd import capi, facade, synth
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc, start_xy=[0.5 * gc.width, 0.5 * gc.width - 0.21])
N = 60
poses = synth.trajectory(world, N)
scans = synth.scans_for(world, geo, poses)
node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
grid = node.grid()
grid.profile(True, "raycast")
for k in range(N):
    grid.profile_reset()
    node.laser(scans[k], geo.angle_min, geo.angle_increment)
    grid.sync()
    if k % 6 == 5:
        ms, n = grid.profile_get("raycast")
        tr = np.zeros((256, 8)); grid.lib.tsd_icp_trace(grid.h, tr.ctypes.data_as(capi._dp), 256)
        d = tr.reshape(-1)
        c = np.abs(d[d != 0])
        print("scan %d kernel %.1f us; beams: median %.0f p90 %.0f p99 %.0f max %.0f cycles (%.1f us); serial-chain beams %d" %
              (k, 1e3 * ms / max(n, 1), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max(), c.max() / 2400.0, (d < 0).sum()))
node.close()
PY
