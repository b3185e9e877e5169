"""What of the registration kernel's time is NOT the 30 steps: tsd_localize (ray cast + k_icp, no scan epilogue) on fixed inputs with
icp_iterations swept -> per-step cost and fixed cost (setup + result) by a linear fit; and the fused scan path's dispatch time for
comparison (the difference to the fit at 30 = the scan epilogue: gates, Sensor::transform, next arguments, result record).
usage: [TSD_LIB_DIR=...] python tools/icp_overheads.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests.test_gpu_parity import build_map, icp_inputs
from tests.slam_driver import HipSlamFused, slam_kwargs
O.build()
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
for k in (5, 12):
    pose, rl, rw, data, mask, M, S = icp_inputs(O, gc, geo, world, k, og)
    xs, ys = [], []
    for it in (11, 12, 14, 18, 22, 26, 30):
        p = dg.icp_params(it, 0.4, 0.02)
        for rep in range(3):
            dg.localize(pose, rw, rl, data, mask, 0.001, 30.0, p)
        dg.profile(True, "icp"); dg.profile_reset()
        for rep in range(20):
            r = dg.localize(pose, rw, rl, data, mask, 0.001, 30.0, p)
        ms, n = dg.profile_get("icp")
        xs.append(r.iterations); ys.append(1e3 * ms / n)
    a, b = np.polyfit(xs, ys, 1)
    print(f"input {k}: dispatch us by iterations {dict(zip(xs, [round(y, 1) for y in ys]))}; fit: {a:.2f} us per step + {b:.1f} us fixed")
poses = synth.trajectory(world, 60)
scans = synth.scans_for(world, geo, poses)
s = HipSlamFused(O, **slam_kwargs(gc, geo))
for k in range(60):
    if k == 10:
        s.grid.profile(True, "icp"); s.grid.profile_reset()
    s.process_scan(scans[k])
ms, n = s.grid.profile_get("icp")
print(f"fused scan path (tsd_scan, epilogue included), 50 scans: {1e3 * ms / n:.1f} us per dispatch")
