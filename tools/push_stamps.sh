#!/bin/bash
# diagnostic (GPU box): where k_push_update's workgroups spend their cycles, push-only (thread 0 of every 8th workgroup, summed
# over its tiles; -DTSD_PUSH_STAMPS build into lib/diag_stamps).  usage: tools/push_stamps.sh [cfg3 comb] [extra -D flags]
cfg=${1:-cfg3}; scene=${2:-comb}; shift; shift
DIAG_DIR=diag_stamps $GRAFT_REPO_ROOT/tools/diag_build.sh push_kernels -DTSD_PUSH_STAMPS "$@" > /dev/null
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_stamps
cd $GRAFT_REPO_ROOT && python3 - $cfg $scene <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests import helpers as H
gc, geo, _ = synth.CONFIGS[sys.argv[1]]
world = synth.World(sys.argv[2], gc)
g = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
for k in range(12):
    pose, (x, y, yaw) = H.sensor_pose(world, k)
    data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
    g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0, want_stats=False)
    g.sync()
    if k in (6, 11):
        tr2 = np.zeros((512, 8)); g.lib.tsd_icp_trace(g.h, tr2.ctypes.data_as(capi._dp), 512)
        tr, sub = tr2[:256], tr2[256:]
        sel = (tr[:, 1] > 0) & (tr[:, 1] > tr[:, 1].max() - 100000)          # this push's workgroups only (100 MHz clock: within 1 ms of the last one to end)
        st, sub = tr[sel], sub[sel]
        t0 = st[:, 0].min()
        life = (st[:, 1] - st[:, 0]) * 0.01
        print(f"   workgroup start after the first: median {np.median((st[:,0]-t0)*0.01):.2f} p90 {np.percentile((st[:,0]-t0)*0.01, 90):.2f} max {((st[:,0]-t0)*0.01).max():.2f} us; "
              f"end after the first start: median {np.median((st[:,1]-t0)*0.01):.2f} p90 {np.percentile((st[:,1]-t0)*0.01, 90):.2f} max {((st[:,1]-t0)*0.01).max():.2f} us; life p10 {np.percentile(life,10):.2f} p90 {np.percentile(life,90):.2f} max {life.max():.2f}")
        sub = sub[st[:, 2] > 0]; st = st[st[:, 2] > 0]
        span = (st[:, 1].max() - t0) * 0.01
        tiles = st[:, 2]
        print(f"{sys.argv[1]}/{sys.argv[2]} push {k}: {len(st)} sampled workgroups, UPDATE tiles each: median {np.median(tiles):.0f} max {tiles.max():.0f}; "
              f"start spread {((st[:,0]-t0)*0.01).max():.2f} us, kernel span seen {span:.2f} us, workgroup life median {np.median((st[:,1]-st[:,0])*0.01):.2f} us")
        names = ["staging / list / record of the previous tile", "phase A + fix-up", "barrier wait after A", "phase C", "barrier wait after C"]
        tot = st[:, 3:8].sum(axis=1)
        for i, nm in enumerate(names):
            c = st[:, 3 + i]
            print(f"   {nm:46s}: {100*np.median(c/tot):5.1f} % of thread 0's cycles; per tile median {np.median(c/tiles):8.0f} cycles")
        print(f"   cycles per tile (thread 0): median {np.median(tot/tiles):.0f}")
        # the slowest decile of the workgroups against the rest: which phase makes them slow
        life_s = (st[:, 1] - st[:, 0]) * 0.01
        slow = life_s >= np.percentile(life_s, 90)
        for lab, m in (("slowest 10 %", slow), ("the others", ~slow)):
            if m.sum() == 0: continue
            parts = " ".join(f"{nm.split()[0]}:{np.median(st[m, 3 + i] / tiles[m]):.0f}" for i, nm in enumerate(names))
            subs = " ".join(f"{lab2}:{np.median(sub[m, i] / tiles[m]):.0f}" for i, lab2 in enumerate(["A-class", "A-lim", "A-compact", "fix-up", "record"]))
            print(f"   {lab:13s} ({m.sum()} wg, life median {np.median(life_s[m]):.2f} us, tiles {np.median(tiles[m]):.0f}): {parts} | {subs}")
        for i, nm in enumerate(["A: d2 table, setup, classification", "A: limits, candidate test", "A: compaction", "A: fix-up", "record of the tile (thread 0)"]):
            print(f"      {nm:40s}: per tile median {np.median(sub[:, i]/tiles):8.0f} cycles")
PY
