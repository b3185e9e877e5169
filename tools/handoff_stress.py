#!/usr/bin/env python3
"""The two in-launch hand-offs of the fused scan under UNEVEN LOAD (test infrastructure).

The push's halo pass rides in the prologue of the ray cast behind it, and -- registration_mode 3 -- the pre-registration's arg-max
is the first workgroup of the registration's launch.  Both hand data between workgroups of ONE launch.  This
tool runs the facade's fused closed loop over n scans and prints a hash of every pose it produced and the grid's digest (halo cells
included); with --load a second context on the same GPU pushes a 16384^2 grid and extracts its occupancy map in a loop on another
thread the whole time (1 280 resident workgroups streaming through HBM beside the scans).  The runs to compare:

    TSD_HALO_KERNEL=1 TSD_PDF_ARGMAX_KERNEL=1 python tools/handoff_stress.py 400 3          # both passes as kernels of their own, idle chip
    python tools/handoff_stress.py 400 3 --load                                             # both hand-offs in-launch, loaded chip

must print the same two hashes (tests/test_gpu_fuzz.py runs a short pair; tools/soak_r6.sh a long one)."""
import hashlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import capi, facade, synth

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(argv[0]) if argv else 400
mode = int(argv[1]) if len(argv) > 1 else 0
LOAD = "--load" in sys.argv
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
poses = synth.trajectory(world, n)
scans = synth.scans_for(world, geo, poses)
params = facade.node_params(gc, geo, registration_mode=mode)
if mode == 3:
    params["tsdpdf_seed"] = 4711                   # reproducible draws
node = facade.SlamNode(params, device=0, synchronous=True)

stop = threading.Event()
pushes = [0]


def load():
    from oracle import pyoracle as O            # (ingest only)
    from tests import helpers as H
    g3, geo3, _ = synth.CONFIGS["cfg3"]
    w3 = synth.World("comb", g3)
    big = capi.TsdGridDevice(g3.map_size_log2, g3.cell_size, g3.max_trunc)
    k = 0
    while not stop.is_set():
        pose, (x, y, yaw) = H.sensor_pose(w3, k % 12)
        data, mask = O.ingest_f32(w3.scan(x, y, yaw, geo3), 30.0, geo3.angle_increment)
        for _ in range(8):
            big.push(pose, data, mask, geo3.angle_increment, geo3.angle_min, 30.0, 0.001, 2.0, want_stats=False)
        big.occupancy(False, 2)
        pushes[0] += 8
        k += 1
    big.close()


th = None
if LOAD:
    th = threading.Thread(target=load, daemon=True)
    th.start()
    time.sleep(1.5)                              # (the big grid is up and pushing)
h = hashlib.sha256()
t0 = time.time()
for k in range(n):
    node.laser(scans[k], geo.angle_min, geo.angle_increment)
    h.update(np.ascontiguousarray(node.report()["pose"], dtype=np.float64).tobytes())
dt = time.time() - t0
stop.set()
if th:
    th.join()
d = node.grid().digest()
print(f"mode {mode} scans {n} load {'yes: %d pushes of a 16384^2 grid beside them' % pushes[0] if LOAD else 'no'}; {n / dt:.0f} scans/s")
print("poses", h.hexdigest()[:32], "grid", "%016x" % d["hash"], "cells", d["cells_valid"])
node.close()
