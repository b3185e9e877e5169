"""Times the registration kernel on ONE fixed input (cfg2 map after 20 pushes, tsd_localize repeated), for the stock library
or a variant built into lib/diag_<name> (TSD_LIB_DIR).  usage: TSD_LIB_DIR=... python tools/icp_ablate.py [label]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests.test_gpu_parity import build_map, icp_inputs
O.build()
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
out = []
for k in (5, 12):
    pose, rl, rw, data, mask, M, S = icp_inputs(O, gc, geo, world, k, og)
    p = dg.icp_params(30, 0.4, 0.02)
    for rep in range(3):
        dg.localize(pose, rw, rl, data, mask, 0.001, 30.0, p)
    dg.profile(True, "icp"); dg.profile_reset()
    for rep in range(20):
        r = dg.localize(pose, rw, rl, data, mask, 0.001, 30.0, p)
    ms, n = dg.profile_get("icp")
    out.append(f"input {k}: {1e3 * ms / n:6.1f} us (pairs {r.pairs}, it {r.iterations})")
print((sys.argv[1] if len(sys.argv) > 1 else "stock").ljust(28), "  ".join(out))
