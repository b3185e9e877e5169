#!/usr/bin/env python3
"""Randomised sweep of the batched multi-robot path (tsd_batch_*: several robots on ONE grid, the reference's multi-robot mode) against
the same order on the oracle's primitives (tests/test_gpu_batch.py's Robot; test infrastructure: uses oracle/): 2-6 robots started at
random offsets, each moving at random (steps below and above the push gate, turns), spoiled readings; every round the robots are split
at random into one or two batch slots that are BEGUN together (both slots' ray casts read the grid before any push of the round), the
pushes follow slot by slot in robot order, the results are collected afterwards -- the two-slot pattern of the facade's dispatcher, with
the device-side waits and gates it uses.  Per robot and round: ray-cast hits, gates, pairs / iterations / state exact, pose 1e-9; final
grid 1e-9.  "mode3": in half of the cases a random subset of the robots has its TSD_PDF pre-registration armed every round
(tsd_scan_preregister with random draws; armed and unarmed scans share batches), winner and candidate count compared too.
usage (GPU box): python3 tools/fuzz_batch.py [cases] [first_seed] [mode3]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests import helpers as H
from tests.slam_driver import slam_kwargs
import tests.test_gpu_batch as T

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with_mode3 = len(sys.argv) > 3 and sys.argv[3] == "mode3"
t_start = time.time()
tot = dict(rounds=0, scans=0, pushes=0, two_slot_rounds=0, armed_scans=0)


def spoil(rng, r32):
    r = r32.copy()
    n = len(r)
    for val in (0.0, np.nan, 45.0, 0.1):
        r[rng.integers(0, n, rng.integers(0, max(2, n // 40)))] = val
    return r


for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    map_log2 = int(rng.choice([9, 10, 10]))
    cs = 0.05
    scene = str(rng.choice(["room", "pillars"]))
    geo = synth.ScanGeometry.full_circle_360() if rng.random() < 0.4 else synth.ScanGeometry.utm30lx()
    gc = synth.GridConfig(map_log2, cs)
    R = int(rng.integers(2, 7))
    n = int(rng.integers(5, 14))
    geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
    kw = slam_kwargs(gc, geo_msg)
    mode3 = with_mode3 and rng.random() < 0.5
    if mode3:
        kw.update(trials=int(rng.choice([20, 40, 100])), size_control_set=int(rng.choice([60, 120, 140])))
        phi_max3 = np.radians(kw["ransac_phi_max"])
    og = O.Grid(gc.map_size_log2, gc.cell_size, gc.truncation_radius * gc.cell_size)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.truncation_radius * gc.cell_size)
    tag = f"seed {seed}: 2^{map_log2} cells, {scene}, {geo.beams} beams, {R} robots, {n} rounds" + (", mode 3" if mode3 else "")
    robots, scans, sensors, slots = [], [], [], []
    try:
        world0 = synth.World(scene, gc)
        for r in range(R):
            off = (float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-0.5, 0.5)))
            w = synth.World(scene, gc, start_xy=[0.5 * gc.width + off[0], 0.5 * gc.width + off[1]])
            x, y, yaw = w.start[0], w.start[1], off[2]
            sc = []
            for k in range(n):
                r32 = world0.scan(x, y, yaw, geo)        # (ONE world for all robots: they map the same room)
                sc.append(spoil(rng, r32) if rng.random() < 0.3 else r32)
                u = rng.random()
                step = rng.uniform(0.0, 0.04) if u < 0.2 else rng.uniform(0.05, 0.1)
                x += step * math.cos(yaw); y += step * math.sin(yaw); yaw += rng.uniform(-0.03, 0.03)
            scans.append(sc)
            robots.append(T.Robot(O, gc, geo, off, kw))
        for rb, sc in zip(robots, scans):
            rb.init_both(og, dg, sc[0])
        for rb in robots:
            s = capi.TsdSensorDevice(dg, geo.beams, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])
            s.set_pose(rb.pose, rb.rays, rb.rays_local)
            sensors.append(s)
        params = dg.icp_params(kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"])
        gates = capi.GateParams(kw["reg_trs_max"], kw["reg_sin_rot_max"], 0.05, 0.03)
        slots = [capi.TsdBatch(dg, R), capi.TsdBatch(dg, R)]
        bounds = (dg.min_x, dg.max_x, dg.min_y, dg.max_y)
        for k in range(1, n):
            # this round's split: a random permutation cut at a random place (second group possibly empty)
            perm = [int(i) for i in rng.permutation(R)]
            cut = int(rng.integers(1, R + 1))
            groups = [g for g in (perm[:cut], perm[cut:]) if g]
            order = [i for g in groups for i in g]
            ing = {i: robots[i].ingest(scans[i][k]) for i in order}
            draws = {i: (tuple(rng.integers(0, 2 ** 31 - 1, m) for m in (geo.beams, kw["size_control_set"], kw["trials"])) if (mode3 and rng.random() < 0.7) else None)
                     for i in order}
            ros = {i: robots[i].localise(og, ing[i][0], ing[i][1], bounds, draws[i]) for i in order}      # all against the grid before the round's pushes
            for i in order:
                if draws[i] is not None:
                    sc_, ms_, _n = O.scene_from_scan(robots[i].rays_local, ing[i][0], ing[i][1])
                    sensors[i].preregister(sc_, ms_, kw["trials"], kw["size_control_set"], kw["zrand"], phi_max3, kw["angle_increment"], *draws[i])
                    tot["armed_scans"] += 1
            for i in order:
                robots[i].apply_push(og)                                                            # ... the pushes in the round's order
            for si, grp in enumerate(groups):
                slots[si].begin([sensors[i] for i in grp], [ing[i][0] for i in grp], [ing[i][1] for i in grp], [ing[i][2] for i in grp], params, gates)
            for si, grp in enumerate(groups):
                slots[si].push()
            for si, grp in enumerate(groups):
                for i, sr in zip(grp, slots[si].results()):
                    ro = ros[i]
                    T._compare(k, i, ro, sr)
                    d, a = H.pose_delta(ro["pose"], np.array(sr.pose[:]).reshape(3, 3))
                    assert d <= 1e-9 and a <= 1e-9, f"round {k} robot {i}: |dpose| {d} m {a} rad"
                    if draws[i] is not None and not ro["no_model"]:
                        pr = sensors[i].preregistration_result()
                        assert (pr["candidates"], pr["idx"], pr["i"]) == ro["pre"], f"round {k} robot {i}: pre-registration {pr} vs {ro['pre']}"
                    tot["scans"] += 1; tot["pushes"] += int(ro["pushed"])
            tot["rounds"] += 1; tot["two_slot_rounds"] += int(len(groups) == 2)
        dg.sync()
        H.assert_grids_equal(og.dump(), dg.download_tiles(), 1e-9)
    except AssertionError as e:
        print("MISMATCH", tag, "--", e)
        sys.exit(1)
    finally:
        for b in slots:
            b.close()
        for s in sensors:
            s.close()
    if case % 10 == 9:
        print(f"{case + 1} cases ok ({tag}); {tot}; {time.time() - t_start:.0f} s", flush=True)
print(f"all {n_cases} cases ok from seed {seed0}: {tot}; {time.time() - t_start:.0f} s")
