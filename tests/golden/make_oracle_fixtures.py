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


if __name__ == "__main__":
    O.build()
    push_raycast_icp()
    trajectory()
    for f in ("oracle_push_raycast_icp.npz", "oracle_trajectory.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
