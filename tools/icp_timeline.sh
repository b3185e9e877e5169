#!/bin/bash
# Timeline of the steady-state step of k_icp: lane 0 of every wave stamps the shader clock at 13 points of four steps (-DTSD_ICP_TIMELINE,
# a diagnostic build in lib/diag_tl; the product library is not touched).  Prints, per interval, the median / min / max over waves and
# steps, the waits at the two barriers, and each wave's own busy chain -- the measured side of profiles/r4_icp_critical_path.txt.
#   gpurun -- tools/icp_timeline.sh [extra hipcc flags for the variant under test]
cd $GRAFT_REPO_ROOT
echo "=== timeline build flags: $*"
DIAG_DIR=diag_tl tools/diag_build.sh icp_kernels -DTSD_ICP_TIMELINE "$@" > /dev/null 2>&1 || { echo "timeline build failed"; exit 1; }
export TSD_LIB_DIR=$GRAFT_REPO_ROOT/ohm_tsd_slam_amd/lib/diag_tl
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, '.')
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests.test_gpu_parity import build_map, icp_inputs
O.build()
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
NAMES = ["0>1 neighbours' coordinates (6 ds_read_b128 + wait)", "1>2 tier 0 arithmetic", "2>3 reciprocal atomics (3 ds_min_rtn_u64) + list append",
         "3>4 BARRIER 1", "4>5 counter + slot minima read, winners", "5>6 pair sums (registers)", "6>7 transpose reduction inside the wave",
         "7>8 BARRIER 2", "8>10 partials -> totals (2 LDS round trips)", "10>11 closed form", "11>12 transform + bounds", "12>0' loop control"]
IDX = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 10), (10, 11), (11, 12)]
for k in (5, 12):
    pose, rl, rw, data, mask, M, S = icp_inputs(O, gc, geo, world, k, og)
    p = dg.icp_params(30, 0.4, 0.02)
    for rep in range(3):
        dg.profile(True, "icp"); dg.profile_reset()
        r = dg.localize(pose, rw, rl, data, mask, 0.001, 30.0, p)
        ms, n = dg.profile_get("icp")
    tr = np.zeros((512, 8)); dg.lib.tsd_icp_trace(dg.h, tr.ctypes.data_as(capi._dp), 512)
    W = int(os.environ.get("TL_W", "6"))
    tl = tr.reshape(-1)[256 * 8: 256 * 8 + 4 * W * 16].reshape(4, W, 16)
    print(f"input {k}: kernel {1e3 * ms / n:.1f} us (timeline build), pairs {r.pairs} it {r.iterations} rms {r.rms!r} T {np.asarray(r.T).reshape(-1)[:6].tolist()!r}")
    iv = np.array([[[tl[s, w, b] - tl[s, w, a] for (a, b) in IDX] for w in range(W)] for s in range(4)])       # [step][wave][interval]
    step_len = np.array([[tl[s + 1, w, 0] - tl[s, w, 0] for w in range(W)] for s in range(3)])
    ctl = np.array([[tl[s + 1, w, 0] - tl[s, w, 12] for w in range(W)] for s in range(3)])
    print("  step length (T0 -> next T0), per wave, steps 19..21:", np.median(step_len, axis=0).astype(int).tolist())
    for j, nm in enumerate(NAMES[:-1]):
        x = iv[:, :, j]
        print(f"  {nm:66s} median {int(np.median(x)):5d}   min {int(x.min()):5d}   max {int(x.max()):5d}   per wave {np.median(x, axis=0).astype(int).tolist()}")
    print(f"  {NAMES[-1]:66s} median {int(np.median(ctl)):5d}   min {int(ctl.min()):5d}   max {int(ctl.max()):5d}")
    busy = iv.sum(axis=2) - iv[:, :, 3] - iv[:, :, 7]
    print("  busy chain of a wave (everything but the two barrier waits), median per wave:", np.median(busy, axis=0).astype(int).tolist())
    print("  waits at barrier 1 / barrier 2, median per wave:", np.median(iv[:, :, 3], axis=0).astype(int).tolist(), np.median(iv[:, :, 7], axis=0).astype(int).tolist())
PY
