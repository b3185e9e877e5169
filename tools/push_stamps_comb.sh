#!/bin/bash
# diagnostic: phase timeline (100 MHz wall clock) of sampled workgroups of k_push_update, push-only, cfg3 / comb
$GRAFT_REPO_ROOT/tools/diag_build.sh push_kernels -DTSD_PUSH_STAMPS $TSD_EXTRA
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests import helpers as H
gc, geo, _ = synth.CONFIGS["cfg3"]
world = synth.World("comb", gc)
g = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
for k in range(12):
    pose, (x, y, yaw) = H.sensor_pose(world, k)
    data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
    g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0, want_stats=False)
    g.sync()
    if k in (6, 11):
        tr = np.zeros((256, 8)); g.lib.tsd_icp_trace(g.h, tr.ctypes.data_as(capi._dp), 256)
        st = tr.reshape(-1)[:1024].reshape(128, 8)
        st = st[st[:, 1] > 0]
        t0 = st[:, 0].min()
        rel = (st[:, :7] - t0) * 0.01
        print("push", k, "sampled groups", len(st), "(stamps of the LAST tile each sampled workgroup processed)")
        print(" start   : min %.2f median %.2f max %.2f" % (rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max()))
        names = ["args+list", "staged(barrier)", "pass A + drain", "pass B (reads, exact, addTsd, writes)", "halo init", "final barrier"]
        for i, nm in enumerate(names):
            d = rel[:, i + 1] - rel[:, i]
            print(" %-40s: median %.2f  p90 %.2f  max %.2f us" % (nm, np.median(d), np.percentile(d, 90), d.max()))
        print(" end     : median %.2f max %.2f" % (np.median(rel[:, 6]), rel[:, 6].max()))
PY
