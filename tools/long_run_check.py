#!/usr/bin/env python3
"""Diagnostic: the HIP facade and the CPU oracle SLAM side by side over a long cfg2 trajectory (test infrastructure:
uses oracle/).  usage: python tools/long_run_check.py [n_scans] [--r4-angles]

The facade receives sensor_msgs/LaserScan, whose angle_min / angle_increment are float32; the oracle loop must be given the same
float32-rounded values (tests/test_gpu_bench_workloads.py does).  Round 4's version of this tool handed the oracle the float64 angles
instead -- a scanner 1.5e-9 rad per beam different from the facade's -- and that, not a kernel, is what the 1e-2 m "divergence" of
profiles/r4_soak_checks.txt (scans 326-339) was: --r4-angles reproduces it, the default run does not show it (profiles/r5_first_flip.txt)."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import facade, synth
from oracle import pyoracle as O
from tests.slam_driver import slam_kwargs

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 340
r4_angles = "--r4-angles" in sys.argv
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
poses = synth.trajectory(world, n)
scans = synth.scans_for(world, geo, poses)
node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
print("oracle scanner geometry: %s" % ("float64 angles (round 4's mistake)" if r4_angles else "float32-rounded angles, as the LaserScan message carries them"))
osl = O.Slam(**slam_kwargs(gc, geo if r4_angles else geo_msg, threads=min(64, os.cpu_count() or 8), nn_mode=int(os.environ.get("NN_MODE", "0"))))
worst = 0.0; prev_dd = 1e-12
for k in range(n):
    node.laser(scans[k], geo.angle_min, geo.angle_increment)
    ro = osl.process_scan(scans[k])
    rh = node.report()
    ph = rh["pose"]; po = np.array(ro.pose).reshape(3, 3)
    eh = math.hypot(ph[0, 2] - poses[k, 0], ph[1, 2] - poses[k, 1]); eo = math.hypot(po[0, 2] - poses[k, 0], po[1, 2] - poses[k, 1])
    dd = math.hypot(ph[0, 2] - po[0, 2], ph[1, 2] - po[1, 2])
    worst = max(worst, dd)
    if k % 20 == 0 or (dd > 1e-6 and dd > 3 * prev_dd) or (eh > 0.5 and k % 10 == 0):
        print("scan %4d truth x %.2f yaw %.2f | hip err %.3f pairs %d state %d reg_err %d pushed %d | oracle err %.3f pairs %d state %d reg_err %d pushed %d | hip-oracle %.2e"
              % (k, poses[k, 0], poses[k, 2], eh, rh.get("pairs", -1), rh.get("icp_state", -1), rh.get("reg_error", -1), rh.get("pushed", -1),
                 eo, ro.pairs, ro.icp_state, ro.reg_error, ro.pushed, dd))
    prev_dd = max(dd, 1e-12)
    if eh > 2.0 and eo > 2.0:
        print("both lost at scan", k); break
print("max |hip - oracle| position difference %.3e over %d scans" % (worst, k + 1))
node.close()
