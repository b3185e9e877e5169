"""Determinism of the registration kernel: the same input registered N times must give ONE bit pattern (a race between waves
would show as several).  usage: [TSD_LIB_DIR=...] python tools/icp_repeat.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests.test_gpu_parity import build_map, icp_inputs
O.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
for k in (5, 12, 17):
    pose, rl, rw, data, mask, M, S = icp_inputs(O, gc, geo, world, k, og)
    p = dg.icp_params(30, 0.4, 0.02)
    seen = {}
    dg.profile(True, "icp"); dg.profile_reset()
    for rep in range(N):
        r = dg.localize(pose, rw, rl, data, mask, 0.001, 30.0, p)
        key = (np.asarray(r.T).tobytes(), float(r.rms).hex(), int(r.pairs))
        seen[key] = seen.get(key, 0) + 1
    ms, n = dg.profile_get("icp")
    ro = O.icp(M, S, pose, 30, 0.4, 0.02, (0.0, og.max_x, 0.0, og.max_x), nn_mode=1)
    dT = max(np.max(np.abs(np.frombuffer(kk[0]).reshape(3, 3) - ro["T"])) for kk in seen)
    print(f"input {k}: {N} runs, {len(seen)} distinct results {sorted(seen.values(), reverse=True)}, {1e3 * ms / n:.1f} us, pairs {[kk[2] for kk in seen]} oracle {ro['pairs']}, max |T - oracle| {dT:.2e}")
