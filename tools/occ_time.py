#!/usr/bin/env python3
"""Diagnostic: time of the occupancy extraction (row N1) on a cfg2 grid after a short SLAM run."""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()          # (torch first: its HIP runtime has to come up before the library's context)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import facade, synth
gc, geo, scene = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
world = synth.World(scene, gc)
poses = synth.trajectory(world, 60)
scans = synth.scans_for(world, geo, poses)
node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
for s in scans:
    node.laser(s, geo.angle_min, geo.angle_increment)
grid = node.grid()
buf = torch.empty(gc.cells * gc.cells, dtype=torch.int8, device="cuda:0")
grid.profile(True, "occupancy"); grid.profile_reset()
for i in range(20):
    grid.occupancy_into(buf.data_ptr(), False, 2)
ms, n = grid.profile_get("occupancy")
cells = gc.cells * gc.cells
tiles = (gc.cells // 32) ** 2
bytes_alg = tiles * 1089 * 8 + cells      # every tile's tsd read once (as stored: 33x33), one int8 written per cell
print("occupancy: %.1f us per extraction (%d launches), algorithmic %.1f MB -> %.2f TB/s" % (1e3 * ms / n, n, bytes_alg / 1e6, bytes_alg / (ms / n * 1e-3) / 1e12))
node.close()
