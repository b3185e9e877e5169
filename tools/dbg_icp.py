import numpy as np, sys
sys.path.insert(0, '.')
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests import helpers as H
from tests.test_gpu_parity import build_map, icp_inputs
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = build_map(O, gc, geo, world)
for k in (2, 5, 9):
    pose, rl, rw, data, mask, M, S = icp_inputs(O, gc, geo, world, k, og)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    ro = O.icp(M, S, pose, 30, 0.4, 0.02, bounds, nn_mode=0, trace=True)
    rd = dg.icp(M, S, pose, dg.icp_params(30, 0.4, 0.02))
    td = dg.icp_trace(30)
    to = ro["trace"]
    print("k", k, "final rms diff", ro["rms"] - rd.rms, "T diff", np.max(np.abs(ro["T"] - rd.T)))
    for i in range(30):
        flag = "" if to[i,0] == td[i,0] else "  <-- pairs differ"
        print(i, int(to[i,0]), int(td[i,0]), "%.3e" % (to[i,1]-td[i,1]), "%.3e" % (to[i,2]-td[i,2]), flag)
