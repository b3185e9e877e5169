"""Probe of the batched multi-robot path through the raw C ABI (tsd_batch_*): R robots on the cfg2 grid, `slots` batch slots
used in turn by ONE host thread with the pushes enqueued ahead of the results.  Prints scans/s and the sampled kernel times.
usage: python tools/batch_probe.py [robots] [slots] [scans]"""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests.slam_driver import slam_kwargs
from tests.test_gpu_batch import Robot

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
O.build()
gc, geo, scene = synth.CONFIGS["cfg2"]
geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
kw = slam_kwargs(gc, geo_msg)
world = synth.World(scene, gc, start_xy=[0.5 * gc.width + 0.37, 0.5 * gc.width - 0.21])
leg = 50
lanes = synth.free_lanes(world, R, 0.06 * leg, clearance=0.6) if R > 1 else [(float(world.start[0]), float(world.start[1]))]
dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.truncation_radius * gc.cell_size)
robots, scans, sensors, truth = [], [], [], []
for r in range(R):
    p = synth.trajectory(world, 12 + K, leg=leg)
    p[:, 1] += lanes[r][1] - world.start[1]
    p[:, 0] += lanes[r][0] - world.start[0]
    truth.append(p)
    scans.append(synth.scans_for(world, geo, p))
    rb = Robot(O, gc, geo, (lanes[r][0] - 0.5 * gc.width, lanes[r][1] - 0.5 * gc.width, 0.1), kw)
    robots.append(rb)
for rb, sc in zip(robots, scans):
    data, mask, _ = rb.ingest(sc[0])
    dg.free_footprint([rb.sx + kw["footprint_x_offset"], rb.sy], kw["footprint_width"], kw["footprint_height"])
    dg.push(rb.pose, data, mask, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])
    rb.rays = O.rays_rescale(rb.rays, rb.cs, 1.0)
    s = capi.TsdSensorDevice(dg, geo.beams, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])
    s.set_pose(rb.pose, rb.rays, rb.rays_local)
    sensors.append(s)
params = dg.icp_params(kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"])
gates = capi.GateParams(kw["reg_trs_max"], kw["reg_sin_rot_max"], 0.05, 0.03)
ing = [[rb.ingest(sc[k]) for k in range(1, 12 + K)] for rb, sc in zip(robots, scans)]     # host ingest outside the timed loop
groups = [list(range(R))[i::NS] for i in range(NS)]
groups = [g for g in groups if g]
slots = [capi.TsdBatch(dg, len(g)) for g in groups]
last = [None] * R


def begin(si, k):
    g = groups[si]
    slots[si].begin([sensors[i] for i in g], [ing[i][k][0] for i in g], [ing[i][k][1] for i in g], [ing[i][k][2] for i in g], params, gates)


def collect(si):
    for i, sr in zip(groups[si], slots[si].results()):
        last[i] = sr


def run(k0, k1):
    for si in range(len(slots)):
        begin(si, k0)
    for k in range(k0, k1):
        for si in range(len(slots)):
            slots[si].push()            # behind the other slots' begins of this round
            collect(si)
            if k + 1 < k1:
                begin(si, k + 1)


run(0, 10)
dg.sync()
dg.profile(True, kernels="all/4")
dg.profile_reset()
t0 = time.perf_counter()
run(10, 10 + K)
dg.sync()
el = time.perf_counter() - t0
names = ["raycast", "icp", "push_classify", "push_update", "push_halo"]
st = {n: dg.profile_get(n) for n in names}
err = max(math.hypot(last[i].pose[2] - truth[i][10 + K, 0], last[i].pose[5] - truth[i][10 + K, 1]) for i in range(R))
print(f"robots {R} slots {len(slots)} scans/s {R * K / el:.0f}  ms/round {1e3 * el / K:.3f}  tracking error {err:.3f} m  " +
      "  ".join(f"{n} {1e3 * ms / max(c, 1):.1f} us x{c}" for n, (ms, c) in st.items()))
