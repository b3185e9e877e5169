"""k_icp duration against the number of beams (same field of view and scene, cfg2 grid): separates the per-wave fixed cost of a
registration step from the per-point cost.  usage: python tools/icp_vs_beams.py [beams ...]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests.slam_driver import slam_kwargs, HipSlamFused

O.build()
gc, geo0, scene = synth.CONFIGS["cfg2"]
fov = geo0.angle_increment * (geo0.beams - 1)
for beams in [int(a) for a in sys.argv[1:]] or [1081, 897, 769, 641, 513, 385, 257]:
    geo = synth.ScanGeometry(beams, geo0.angle_min, fov / (beams - 1))
    geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
    world = synth.World(scene, gc)
    n = 60
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    hs = HipSlamFused(O, **slam_kwargs(gc, geo_msg))
    for k in range(10):
        hs.process_scan(scans[k])
    hs.grid.sync()
    hs.grid.profile(True, kernels="all/1")
    hs.grid.profile_reset()
    for k in range(10, n):
        out = hs.process_scan(scans[k])
    hs.grid.sync()
    ms, c = hs.grid.profile_get("icp")
    rc, c2 = hs.grid.profile_get("raycast")
    err = math.hypot(out["pose"][0, 2] - poses[-1, 0], out["pose"][1, 2] - poses[-1, 1])
    T = ((beams + 2) // 3 + 63) // 64 * 64
    print(f"beams {beams:5d}  blocks {(beams + 63) // 64:2d}  threads {T:4d}  icp {1e3 * ms / c:7.1f} us  raycast {1e3 * rc / c2:5.1f} us  pairs {out['pairs']}  err {err:.3f}")
    hs.grid.close()
