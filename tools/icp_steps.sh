#!/bin/bash
# Length of EVERY step of a registration (wave 0's stamps of the timeline build, all 30 steps), with the phases of the search steps:
#   gpurun -- tools/icp_steps.sh [extra hipcc flags]
cd $GRAFT_REPO_ROOT
DIAG_DIR=diag_tl tools/diag_build.sh icp_kernels -DTSD_ICP_TIMELINE -DTSD_ICP_TL_FIRST=0 -DTSD_ICP_TL_STEPS=30 "$@" > /dev/null 2>&1 || { echo "timeline build failed"; exit 1; }
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_tl
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests.test_gpu_parity import build_map, icp_inputs
O.build()
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
for k in (5, 12, 17):
    pose, rl, rw, data, mask, M, S = icp_inputs(O, gc, geo, world, k, og)
    p = dg.icp_params(30, 0.4, 0.02)
    for rep in range(3):
        dg.profile(True, "icp"); dg.profile_reset()
        r = dg.localize(pose, rw, rl, data, mask, 0.001, 30.0, p)
        ms, n = dg.profile_get("icp")
    tr = np.zeros((512, 8)); dg.lib.tsd_icp_trace(dg.h, tr.ctypes.data_as(capi._dp), 512)
    tl = tr.reshape(-1)[256 * 8: 256 * 8 + 30 * 16].reshape(30, 16)
    t0 = tl[:, 0]
    print(f"input {k}: kernel {1e3 * ms / n:.1f} us = {1e3 * ms / n * 2400:.0f} cycles at 2.4 GHz; first step starts at its own zero; pairs {r.pairs}")
    print("  step lengths (wave 0, T0 -> next T0):", np.diff(t0).astype(int).tolist())
    print("  barrier 1 -> winners known (the search block when there is one):", (tl[:, 5] - tl[:, 4]).astype(int).tolist())
    print("  top -> barrier 1:", (tl[:, 4] - tl[:, 0]).astype(int).tolist())
    ws = tr.reshape(-1)[(256 + 128) * 8: (256 + 128) * 8 + 30 * 8].reshape(30, 8)
    print("  list pass, wave 0's first round, per step: (entries on the list, lanes active, largest / mean window rounds of a lane, cycles of the window_search call)")
    print("   ", [(int(w[0]), int(w[4]), int(w[1]), round(w[2] / max(w[4], 1), 2), int(w[3])) for w in ws if w[0] > 0])
    srch = [i for i in range(30) if tl[i, 13] > tl[i, 4]]
    print("  search steps", srch)
    print("    barrier 1 -> window pass done:", [int(tl[i, 13] - tl[i, 4]) for i in srch])
    print("    -> whole-wave searches done:  ", [int(tl[i, 14] - tl[i, 13]) for i in srch])
    print("    -> results read, winners:     ", [int(tl[i, 5] - tl[i, 14]) for i in srch])
    print("  sum of the 29 step lengths %d; steady-state median %d" % (np.diff(t0).sum(), np.median(np.diff(t0)[18:])))
PY
