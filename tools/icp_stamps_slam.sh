#!/bin/bash
# diagnostic: per-phase cycle stamps of k_icp inside the closed SLAM loop of bench.py (cfg2)
$GRAFT_REPO_ROOT/tools/diag_build.sh icp_kernels -DTSD_ICP_STAMPS $TSD_EXTRA
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 - <<'PY' 2>&1 | grep -E "${TSD_GREP:-ICPDBG|avg over|SLOW|cycles/step|searched/step}" | tail -${TSD_TAIL:-12}
import numpy as np, sys
sys.path.insert(0, '.')
from ohm_tsd_slam_amd import capi, facade, synth
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc, start_xy=[0.5 * gc.width, 0.5 * gc.width - 0.21])
N = 130
poses = synth.trajectory(world, N)
scans = synth.scans_for(world, geo, poses)
node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
grid = node.grid()
import os
names = ["setup+tier0", "A search", "BC reciprocal", "D sums1", "F sums2+trig", "G transform+ctl"] if "SETUP" not in os.environ.get("TSD_EXTRA", "") else ["s0 loads", "s1 ballot+bar", "s2 compaction", "s3 take regs", "s4 unit+pads", "s5 rmax"]
acc = np.zeros(8); cnt = 0; worst = []
grid.profile(True, "icp"); grid.profile_reset()
for k in range(N):
    node.laser(scans[k], geo.angle_min, geo.angle_increment)
    if k < 10: continue
    grid.sync()
    tr = np.zeros((256, 8)); grid.lib.tsd_icp_trace(grid.h, tr.ctypes.data_as(capi._dp), 256)
    st = tr.reshape(-1)[-8:]
    acc += st; cnt += 1
    worst.append((st[:6].sum() + st[6], k, np.diff(np.concatenate([[0], tr[:30, 2]])).astype(int).tolist(),
                  np.diff(np.concatenate([[0], tr[:30, 1]])).astype(int).tolist(), int(st[7]), [int(x) for x in st[:6] / 30]))
    if k % 23 == 0:
        print("scan", k, "A cycles/step:", np.diff(np.concatenate([[0], tr[:30, 2]])).astype(int).tolist())
        print("   searched/step:", np.diff(np.concatenate([[0], tr[:30, 1]])).astype(int).tolist(), "wave searches", int(st[7]))
ms, n = grid.profile_get("icp")
print("avg over %d scans: kernel ms %.3f, cycles total %.0f" % (cnt, ms / max(n, 1), acc[:6].sum() / cnt),
      {nm: "%.0f" % (c / cnt / 30) for nm, c in zip(names, acc[:6])}, "setup cycles %.0f wave searches %.1f" % (acc[6] / cnt, acc[7] / cnt))
worst.sort(reverse=True)
for w in worst[:4]:
    print('SLOW scan %d total %d cycles; per-step phases %s; wave searches %d' % (w[1], w[0], w[5], w[4]))
    print('   A cycles/step:', w[2])
    print('   searched/step:', w[3])
node.close()
PY
