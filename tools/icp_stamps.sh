#!/bin/bash
# diagnostic: rebuild icp_kernels with per-phase cycle stamps on the GPU box and print the phase shares
$GRAFT_REPO_ROOT/tools/diag_build.sh icp_kernels -DTSD_ICP_STAMPS $TSD_EXTRA
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, '.')
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests.test_gpu_parity import build_map, icp_inputs
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
pose, rl, rw, data, mask, M, S = icp_inputs(O, gc, geo, world, 5, og)
p = dg.icp_params(30, 0.4, 0.02)
for rep in range(3):
    dg.profile(True, "icp"); dg.profile_reset()
    rd = dg.icp(M, S, pose, p)
    ms, n = dg.profile_get("icp")
    tr = np.zeros((256, 8)); dg.lib.tsd_icp_trace(dg.h, tr.ctypes.data_as(capi._dp), 256)
    st = tr.reshape(-1)[-8:]
    names = ["setup+tier0", "A list", "BC reciprocal", "D sums1", "F sums2+trig", "G transform+ctl"]
    tot = st[:6].sum()
    print("phase A cycles per step:", np.diff(np.concatenate([[0], tr[:30, 2]])).astype(int).tolist())
    print("whole-wave searches per step:", np.diff(np.concatenate([[0], tr[:30, 3]])).astype(int).tolist())
    print("searched points per step:", np.diff(np.concatenate([[0], tr[:30, 1]])).astype(int).tolist())
    print("searched points total %d, of which whole-wave searches %d" % (st[6], st[7]))
    print("shape", os.environ.get("TSD_ICP_SHAPE", "0"), "kernel ms %.3f" % (ms / n), "cycles total %.0f" % tot,
          {nm: "%.0f" % (c / 30) for nm, c in zip(names, st[:6])}, "nM", len(M), "nS", len(S), "pairs", rd.pairs)
PY
