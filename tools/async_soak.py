#!/usr/bin/env python3
"""Diagnostic: 200 scans of cfg 2 through the fused scan with asynchronous mapping against the same order on the oracle's primitives
(tests/test_gpu_async_mapping.py runs 30 + 14).  usage (GPU box): python3 tools/async_soak.py"""
import sys, numpy as np
sys.path.insert(0, '.')
from ohm_tsd_slam_amd import synth
from oracle import pyoracle as O
from tests import helpers as H
from tests.slam_driver import HipSlamFused, slam_kwargs
from tests.test_gpu_async_mapping import OracleOnePushBehind
O.build()
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
n = 200
poses = synth.trajectory(world, n)
scans = synth.scans_for(world, geo, poses)
kw = slam_kwargs(gc, geo)
hs = HipSlamFused(O, **kw); oa = OracleOnePushBehind(O, **kw)
worst = 0.0; flips = 0
for k in range(n):
    rh = hs.process_scan(scans[k])
    if k == 0: hs.sensor.set_async_mapping(True)
    ro = oa.process_scan(scans[k])
    d, a = H.pose_delta(ro["pose"], rh["pose"]); worst = max(worst, d)
    if (rh["pushed"], rh["reg_error"], rh["pairs"]) != (ro["pushed"], ro["reg_error"], ro["pairs"]): flips += 1
oa.flush(); hs.grid.sync()
print("200 scans cfg2 async: worst pose diff %.3e m, mismatching scans %d" % (worst, flips))
H.assert_grids_equal(oa.g.dump(), hs.grid.download_tiles(), 1e-6)
print("grids equal")
