#!/usr/bin/env python3
"""Generates tests/golden/oracle_*.npz: small input/output vectors of the hot path computed by the CPU
oracle (oracle/tsd_oracle.c) on the deterministic synthetic worlds of ohm_tsd_slam_amd/synth.py.

What they pin: the oracle against regressions, and -- on the GPU box, where nothing else of the build
container exists -- the HIP path against numbers that were fixed at commit time.  They are NOT outputs
of the reference (only ref_chain_pairs.npz is, see make_ref_chain_fixture.py); DESIGN.md section 6.

    python tests/golden/make_oracle_fixtures.py        (needs only gcc + numpy)
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from ohm_tsd_slam_amd import synth  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.slam_driver import slam_kwargs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def push_raycast_icp():
    gc = synth.GridConfig(7, 0.1)                      # 128 x 128 cells = 16 tiles
    geo = synth.ScanGeometry(181, math.radians(-90.0), math.radians(1.0))
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    poses, scans, stats = [], [], []
    for k in range(3):
        pose, (x, y, yaw) = H.sensor_pose(world, 3 * k)
        r = world.scan(x, y, yaw, geo).copy()
        if k == 1:
            r[5:9] = 0.0; r[40:43] = np.nan; r[100:110] = 45.0        # zero / NaN / over-range readings
        data, mask = O.ingest_f32(r, H.MAX_RANGE, geo.angle_increment)
        st = g.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        poses.append(pose); scans.append(r); stats.append([st[k_] for k_ in sorted(st)])
    init, iw, tsd, w = g.dump()
    pose, (x, y, yaw) = H.sensor_pose(world, 4)
    rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
    co, no, mo, cnt = g.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    r32 = world.scan(x + 0.04, y - 0.02, yaw + 0.01, geo)
    data, mask = O.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
    scene, ms, _ = O.scene_from_scan(rl, data, mask)
    M = co.reshape(-1, 2)[mo.astype(bool)]
    S = scene.reshape(-1, 2)[ms.astype(bool)]
    icp = O.icp(M, S, pose, 30, 0.4, 0.02, (0.0, g.max_x, 0.0, g.max_x), nn_mode=0, trace=True)
    np.savez_compressed(
        os.path.join(HERE, "oracle_push_raycast_icp.npz"),
        map_size_log2=gc.map_size_log2, cell_size=gc.cell_size, max_trunc=gc.max_trunc,
        beams=geo.beams, angle_min=geo.angle_min, angle_increment=geo.angle_increment,
        push_poses=np.array(poses), push_scans=np.array(scans), push_stats=np.array(stats),
        push_stat_names=np.array(sorted(st)), init=init, init_weight=iw, tsd=tsd, weight=w,
        rc_pose=pose, rc_rays_world=rw, rc_rays_local=rl, rc_mask=mo, rc_coords=co, rc_normals=no,
        icp_ranges=data, icp_mask=mask, icp_model=M, icp_scene=S, icp_T=icp["T"], icp_rms=icp["rms"],
        icp_pairs=icp["pairs"], icp_iterations=icp["iterations"], icp_state=icp["state"], icp_trace=icp["trace"])


def trajectory():
    gc = synth.GridConfig(8, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    poses = synth.trajectory(world, 10)
    scans = synth.scans_for(world, geo, poses)
    slam = O.Slam(**slam_kwargs(gc, geo))
    rows = []
    for k in range(len(scans)):
        r = slam.process_scan(scans[k])
        rows.append(list(r.pose[:]) + [r.pairs, r.valid_model, r.pushed, r.reg_error, r.rms])
    init, iw, tsd, w = slam.grid.dump()
    sel = init.astype(bool)
    m = ~np.isnan(tsd[sel])
    summary = np.array([init.sum(), iw.sum(), np.nansum(tsd[sel]), w[sel].sum(), np.count_nonzero(m)])
    np.savez_compressed(os.path.join(HERE, "oracle_trajectory.npz"), map_size_log2=gc.map_size_log2,
                        cell_size=gc.cell_size, scans=scans, rows=np.array(rows), init=init, grid_summary=summary)


DIGEST_KEYS = ("hash", "cells_valid", "tiles_initialized", "sum_tsd", "sum_weight")


def baseline_configs():
    """SURVEY 8(c) "golden vectors to commit": for BASELINE configs 1-3 (and cfg3 / comb) per push the work counters and
    the digest of the whole grid (64-bit hash of the canonical dump, valid cells, sum tsd, sum weight: tsd_grid_digest /
    ora_grid_digest), a ray cast, a registration with its per-iteration (pairs, rms, threshold, state, Tlast), and a
    short closed loop (per-scan pose, pairs, pushed) -- everything the GPU box needs to check the HIP path at FULL size
    without the oracle.  Inputs are the raw float32 scans (the ingest is part of what is pinned)."""
    out = {}
    threads = min(8, os.cpu_count() or 1)
    for tag, cfg, scene, n_push in (("cfg1", "cfg1", "room", 4), ("cfg2", "cfg2", "pillars", 4), ("cfg3", "cfg3", "pillars", 3),
                                    ("cfg3comb", "cfg3", "comb", 3)):
        gc, geo, _ = synth.CONFIGS[cfg]
        world = synth.World(scene, gc)
        g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
        poses, scans, stats, digs = [], [], [], []
        for k in range(n_push):
            pose, (x, y, yaw) = H.sensor_pose(world, 5 * k)
            r = world.scan(x, y, yaw, geo)
            data, mask = O.ingest_f32(r, H.MAX_RANGE, geo.angle_increment)
            st = g.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, threads=threads)
            d = g.digest()
            poses.append(pose); scans.append(r); stats.append([st[k_] for k_ in sorted(st)])
            digs.append([np.uint64(d["hash"]), d["cells_valid"], d["tiles_initialized"]]); out[f"{tag}_sums_{k}"] = np.array([d["sum_tsd"], d["sum_weight"]])
        out[f"{tag}_push_poses"] = np.array(poses); out[f"{tag}_push_scans"] = np.array(scans)
        out[f"{tag}_push_stats"] = np.array(stats); out[f"{tag}_stat_names"] = np.array(sorted(st))
        out[f"{tag}_digest_hash"] = np.array([d[0] for d in digs], dtype=np.uint64)
        out[f"{tag}_digest_counts"] = np.array([[d[1], d[2]] for d in digs], dtype=np.int64)
        # ray cast + registration from a pose between the pushes
        pose, (x, y, yaw) = H.sensor_pose(world, 7)
        rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
        co, no, mo, cnt = g.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE, threads=threads)
        out[f"{tag}_rc_pose"] = pose; out[f"{tag}_rc_rays_world"] = rw; out[f"{tag}_rc_rays_local"] = rl
        out[f"{tag}_rc_mask"] = mo; out[f"{tag}_rc_coords"] = co; out[f"{tag}_rc_normals"] = no
        if scene != "comb":
            r32 = world.scan(x + 0.04, y - 0.02, yaw + 0.01, geo)
            data, mask = O.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
            sc, ms, _ = O.scene_from_scan(rl, data, mask)
            M = co.reshape(-1, 2)[mo.astype(bool)]
            S = sc.reshape(-1, 2)[ms.astype(bool)]
            icp = O.icp(M, S, pose, 30, 0.4, 0.02, (0.0, g.max_x, 0.0, g.max_x), nn_mode=1, trace=True)
            out[f"{tag}_icp_scan"] = r32; out[f"{tag}_icp_T"] = icp["T"]; out[f"{tag}_icp_trace"] = icp["trace"]
            out[f"{tag}_icp_counts"] = np.array([icp["pairs"], icp["iterations"], icp["state"], len(M), len(S)])
            # closed loop through the SLAM loop (a LaserScan carries angle_min / angle_increment as float32)
            geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
            n = 8
            tposes = synth.trajectory(world, n)
            tscans = synth.scans_for(world, geo, tposes)
            slam = O.Slam(**slam_kwargs(gc, geo_msg, nn_mode=1, threads=threads))
            rows = []
            for k in range(n):
                rr = slam.process_scan(tscans[k])
                rows.append(list(rr.pose[:]) + [rr.pairs, rr.iterations, rr.icp_state, rr.valid_model, rr.valid_scene, rr.pushed, rr.reg_error, rr.rms])
            dd = slam.grid.digest()
            out[f"{tag}_traj_scans"] = tscans; out[f"{tag}_traj_rows"] = np.array(rows)
            out[f"{tag}_traj_grid"] = np.array([dd["cells_valid"], dd["tiles_initialized"], dd["sum_tsd"], dd["sum_weight"]])
            slam.close()
        g.close()
    np.savez_compressed(os.path.join(HERE, "oracle_baseline_configs.npz"), **out)


def n1_n4():
    """SURVEY 8(f) rows N1 and N4.  N1 on a grid of its own (256 x 256 cells = 8 x 8 tiles: the extraction skips the outer ring of
    tiles, so the small grid of the push fixture shows nothing): the occupancy map after every push, plain and inflated (`content`
    persists like ThreadGrid::_occGridContent), and the colour image of the final grid.  N4 on the model (with the ray cast's normals)
    and scene of oracle_push_raycast_icp.npz (which must exist): the point-to-line registration."""
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    content = {False: np.full(gc.cells * gc.cells, -1, dtype=np.int8), True: np.full(gc.cells * gc.cells, -1, dtype=np.int8)}
    occ = {False: [], True: []}; marks = {False: [], True: []}
    poses, scans = [], []
    for k in range(4):
        pose, (x, y, yaw) = H.sensor_pose(world, 4 * k)
        r = world.scan(x, y, yaw, geo).copy()
        data, mask = O.ingest_f32(r, H.MAX_RANGE, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        poses.append(pose); scans.append(r)
        for inflate in (False, True):
            o, n = g.occupancy(content[inflate], inflate, 2)
            occ[inflate].append(np.array(o, dtype=np.int8).reshape(gc.cells, gc.cells).copy()); marks[inflate].append(n)
    img = g.color_image()
    g.close()
    f = np.load(os.path.join(HERE, "oracle_push_raycast_icp.npz"))
    g2 = O.Grid(int(f["map_size_log2"]), float(f["cell_size"]), float(f["max_trunc"]))
    M, S = f["icp_model"], f["icp_scene"]
    N = f["rc_normals"].reshape(-1, 2)[f["rc_mask"].astype(bool)]
    ptl = O.icp(M, S, f["rc_pose"], 30, 0.4, 0.02, (0.0, g2.max_x, 0.0, g2.max_x), nn_mode=0, model_normals_xy=N)
    g2.close()
    np.savez_compressed(
        os.path.join(HERE, "oracle_n1_n4.npz"),
        map_size_log2=gc.map_size_log2, cell_size=gc.cell_size, max_trunc=gc.max_trunc,
        beams=geo.beams, angle_min=geo.angle_min, angle_increment=geo.angle_increment,
        push_poses=np.array(poses), push_scans=np.array(scans),
        occ_plain=np.array(occ[False]), occ_inflated=np.array(occ[True]), marks_plain=np.array(marks[False]),
        marks_inflated=np.array(marks[True]), color_image=np.asarray(img),
        ptl_normals=N, ptl_T=ptl["T"], ptl_rms=ptl["rms"], ptl_pairs=ptl["pairs"], ptl_iterations=ptl["iterations"], ptl_state=ptl["state"])
    print("oracle_n1_n4.npz", os.path.getsize(os.path.join(HERE, "oracle_n1_n4.npz")), "bytes; marks", marks[False], marks[True],
          "ptl pairs", ptl["pairs"], "rms", ptl["rms"])


def tsdpdf():
    """TSD_PDFMatching::match with fixed rand() draws on a small map, and Icp::iterate with its result as Tinit
    (registration_mode 3).  Inputs: the pushes that build the map, the pose the robot believes, the float32 scan; the three draw
    streams.  Outputs: the match (winner, counts, T, probability) and the registration."""
    gc = synth.GridConfig(8, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    poses, scans = [], []
    for k in range(4):
        pose, (x, y, yaw) = H.sensor_pose(world, k)
        r = world.scan(x, y, yaw, geo)
        data, mask = O.ingest_f32(r, H.MAX_RANGE, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        poses.append(pose); scans.append(r)
    pose, _ = H.sensor_pose(world, 3)                              # where the robot believes it is
    _, (x, y, yaw) = H.sensor_pose(world, 8)                       # where the scan is taken
    rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
    co, no, mo, cnt = g.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    r32 = world.scan(x, y, yaw + 0.05, geo)
    data, mask = O.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
    sc, ms, _ = O.scene_from_scan(rl, data, mask)
    rng = np.random.default_rng(20261003)
    trials, ctrl, zrand, phi_max = 100, 140, 0.25, math.radians(30.0)
    ds, dc, dt = (rng.integers(0, 2 ** 31 - 1, n).astype(np.int32) for n in (geo.beams, ctrl, trials))
    m = O.tsdpdf_match(g, pose, co, mo, sc, ms, trials, ctrl, zrand, phi_max, geo.angle_increment, ds, dc, dt)
    M = co.reshape(-1, 2)[mo.astype(bool)]
    S = sc.reshape(-1, 2)[ms.astype(bool)]
    icp = O.icp_init(M, S, pose, 30, 0.4, 0.02, (0.0, g.max_x, 0.0, g.max_x), m["T"], nn_mode=1)
    np.savez_compressed(
        os.path.join(HERE, "oracle_tsdpdf.npz"), map_size_log2=gc.map_size_log2, cell_size=gc.cell_size, max_trunc=gc.max_trunc,
        beams=geo.beams, angle_min=geo.angle_min, angle_increment=geo.angle_increment, push_poses=np.array(poses), push_scans=np.array(scans),
        pose=pose, rays_world=rw, rays_local=rl, scan=r32, rc_mask=mo, rc_coords=co, scene=sc, scene_mask=ms,
        trials=trials, size_control_set=ctrl, zrand=zrand, phi_max=phi_max, draws_sub=ds, draws_ctrl=dc, draws_trials=dt,
        match_T=m["T"], match_prob=m["prob"], match_counts=np.array([m["candidates"], m["idx"], m["i"]]),
        icp_T=icp["T"], icp_counts=np.array([icp["pairs"], icp["iterations"], icp["state"], len(M), len(S)]), icp_rms=icp["rms"])
    g.close()


if __name__ == "__main__":
    O.build()
    if len(sys.argv) > 1 and sys.argv[1] == "tsdpdf":                 # (only the N3 fixture: the others take minutes)
        tsdpdf()
        print("oracle_tsdpdf.npz", os.path.getsize(os.path.join(HERE, "oracle_tsdpdf.npz")), "bytes")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n1n4":                   # (only the N1 / N4 fixture, on top of the committed push fixture)
        n1_n4()
        sys.exit(0)
    push_raycast_icp()
    trajectory()
    baseline_configs()
    tsdpdf()
    n1_n4()
    for f in ("oracle_push_raycast_icp.npz", "oracle_trajectory.npz", "oracle_baseline_configs.npz", "oracle_tsdpdf.npz", "oracle_n1_n4.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
