#!/bin/bash
# diagnostic: phase timeline (100 MHz wall clock) of sampled workgroups of k_push_update inside the bench loop
$GRAFT_REPO_ROOT/tools/diag_build.sh push_kernels -DTSD_PUSH_STAMPS $TSD_EXTRA
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from ohm_tsd_slam_amd import capi, facade, synth
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc, start_xy=[0.5 * gc.width, 0.5 * gc.width - 0.21])
N = 40
poses = synth.trajectory(world, N)
scans = synth.scans_for(world, geo, poses)
node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
grid = node.grid()
for k in range(N):
    node.laser(scans[k], geo.angle_min, geo.angle_increment)
    grid.sync()
    if k in (20, 30, 39):
        tr = np.zeros((256, 8)); grid.lib.tsd_icp_trace(grid.h, tr.ctypes.data_as(capi._dp), 256)
        st = tr.reshape(128, 8)
        st = st[st[:, 1] > 0]
        t0 = st[:, 0].min()
        rel = (st[:, :7] - t0) * 0.01     # microseconds since the first sampled workgroup started
        np.set_printoptions(precision=2, suppress=True, linewidth=200)
        print("scan", k, "sampled groups", len(st))
        print(" start   : min %.2f median %.2f max %.2f" % (rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max()))
        names = ["args+list", "staged(barrier)", "indices+sd", "reads back", "addTsd+writes issued", "final barrier"]
        for i, nm in enumerate(names):
            d = rel[:, i + 1] - rel[:, i]
            print(" %-22s: median %.2f  p90 %.2f  max %.2f us" % (nm, np.median(d), np.percentile(d, 90), d.max()))
        print(" end     : median %.2f max %.2f" % (np.median(rel[:, 6]), rel[:, 6].max()))
node.close()
PY
