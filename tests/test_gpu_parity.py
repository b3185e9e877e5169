"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): tile flags / initWeight / hit masks / pair counts bit-exact; SDF and
weight within 1e-5 (observed: 0 with fp64 storage); ICP pose within 1e-4 m / 1e-4 rad.
"""
import math

import numpy as np
import pytest

from ohm_tsd_slam_amd import capi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_CELL = 1e-5      # SDF / weight tolerance stated by north_star
TOL_POSE_M = 1e-4
TOL_POSE_RAD = 1e-4


def make_pair(oracle, gc):
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    return og, dg


def push_both(oracle, og, dg, world, geo, k, ranges_f32=None, remask=False):
    pose, (x, y, yaw) = H.sensor_pose(world, k)
    r32 = world.scan(x, y, yaw, geo) if ranges_f32 is None else ranges_f32
    data, mask = oracle.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
    if remask:
        data, mask = oracle.ingest_f64(data, H.MAX_RANGE, geo.angle_increment)
    so = og.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
    sd = dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
    return so, sd


@pytest.mark.parametrize("scene,map_log2,cs,geo", [
    ("room", 8, 0.1, synth.ScanGeometry.full_circle_360()),
    ("room", 9, 0.05, synth.ScanGeometry.full_circle_360()),          # BASELINE cfg 1
    ("pillars", 10, 0.05, synth.ScanGeometry.utm30lx()),
])
def test_push_matches_oracle(oracle, scene, map_log2, cs, geo):
    gc = synth.GridConfig(map_log2, cs)
    world = synth.World(scene, gc)
    og, dg = make_pair(oracle, gc)
    for k in range(6):
        so, sd = push_both(oracle, og, dg, world, geo, k * 5)
        assert so == sd, f"push {k}: stats differ\n oracle {so}\n hip    {sd}"
        dt, dw = H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
    assert so["cells_updated"] > 0


def test_push_special_readings(oracle):
    """0.0, NaN, > max_range and +inf readings (Sensor.cpp:252-272, TsdGrid.cpp:266-271)."""
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    pose, (x, y, yaw) = H.sensor_pose(world, 0)
    r = world.scan(x, y, yaw, geo).copy()
    r[10:40] = 0.0
    r[100:130] = np.nan
    r[200:260] = 45.0
    r[300:330] = np.inf
    r[500] = 1.0      # depth discontinuity
    for remask in (False, True):      # initPush path and queuePush re-mask path
        so, sd = push_both(oracle, og, dg, world, geo, 0, ranges_f32=r, remask=remask)
        assert so == sd
        H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)


def test_push_empties_and_reinit(oracle):
    """Tiles that become EMPTY (increaseEmptiness on uninitialised and on initialised tiles) and tiles
    later materialised from _initWeight > 0 (TsdGridComponent.cpp:104-121, TsdGridPartition.cpp:98-108)."""
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    pose, (x, y, yaw) = H.sensor_pose(world, 0)
    near = np.full(geo.beams, 2.0, dtype=np.float32)
    far = np.full(geo.beams, 9.0, dtype=np.float32)
    seen = {"emptied_init": 0, "emptied_uninit": 0, "new_from_empty": 0}
    for r in (far, near, far, far, near, far):
        so, sd = push_both(oracle, og, dg, world, geo, 0, ranges_f32=r)
        assert so == sd
        H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
        seen["emptied_init"] += so["tiles_emptied_init"]
        seen["emptied_uninit"] += so["tiles_emptied_uninit"]
        seen["new_from_empty"] += so["tiles_new_from_empty"]
    assert all(v > 0 for v in seen.values()), seen


def test_free_footprint_then_push(oracle):
    gc = synth.GridConfig(8, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    c = [world.start[0] + 0.28, world.start[1]]
    assert og.free_footprint(c, 1.0, 1.0) and dg.free_footprint(c, 1.0, 1.0)
    H.assert_grids_equal(og.dump(), dg.download_tiles(), 0.0)
    so, sd = push_both(oracle, og, dg, world, geo, 0)
    assert so == sd
    H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
    # out of bounds rectangle is rejected by both (TsdGrid.cpp:617-622)
    assert not og.free_footprint([1e4, 1e4], 1.0, 1.0)
    assert not dg.free_footprint([1e4, 1e4], 1.0, 1.0)


def test_push_launch_window_follows_the_sensor(oracle):
    """The push kernels are launched over the tile window the scan can reach (union with the previous window and
    with freeFootprint marks): jumps across the map, a footprint freed far away from the sensor and a push with a
    longer range must leave the same grid and the same statistics as the reference's sweep over all tiles."""
    gc = synth.GridConfig(10, 0.05)                      # 51.2 m, 32 x 32 tiles
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("pillars", gc)
    og, dg = make_pair(oracle, gc)
    w = gc.width
    stops = [(0.2 * w, 0.25 * w, 0.3, 4.0), (0.8 * w, 0.7 * w, 2.0, 4.0), (0.8 * w, 0.7 * w, 2.1, 4.0),
             (0.15 * w, 0.85 * w, -1.0, 3.0), (0.5 * w, 0.5 * w, 0.0, 30.0), (0.21 * w, 0.26 * w, 0.35, 4.0)]
    total = None
    for n, (x, y, yaw, max_range) in enumerate(stops):
        if n == 2:      # far from every window so far
            c = [0.3 * w, 0.6 * w]
            assert og.free_footprint(c, 1.5, 1.0) and dg.free_footprint(c, 1.5, 1.0)
        pose = synth.pose_matrix(x, y, yaw)
        r32 = np.minimum(world.scan(x, y, yaw, geo), np.float32(max_range + 1.0))
        data, mask = oracle.ingest_f32(r32, max_range, geo.angle_increment)
        so = og.push(pose, data, mask, geo.angle_increment, geo.angle_min, max_range, H.MIN_RANGE, H.LOW_REFL)
        sd = dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, max_range, H.MIN_RANGE, H.LOW_REFL)
        assert so == sd, f"stop {n}: stats differ\n oracle {so}\n hip    {sd}"
        H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
        total = dict(so) if total is None else {k: total[k] + v for k, v in so.items()}
    st, pushes = dg.push_stats_total()
    assert pushes == len(stops)
    total.pop("tiles_total")                            # a property of the grid, not a sum over pushes
    assert {k: st[k] for k in total} == total


def test_push_beam_index_without_atan2_is_exact(oracle):
    """k_push_update names the beam of a cell from a fp32 angle estimate proven by two fp64 cross products and falls
    back to the reference's atan2 + round near a beam boundary: every rotation of the sensor, axis-aligned ones
    included (cells exactly on boundaries), must give the reference's grid."""
    gc = synth.GridConfig(8, 0.05)
    world = synth.World("room", gc)
    for geo in (synth.ScanGeometry.utm30lx(), synth.ScanGeometry.full_circle_360()):
        og, dg = make_pair(oracle, gc)
        x, y = world.start[0], world.start[1]
        for yaw in (0.0, math.pi / 2, math.pi, -math.pi / 2, math.pi / 4, 0.5 * geo.angle_increment, 1.2345, -2.9):
            pose = synth.pose_matrix(x + 0.025, y + 0.025, yaw)      # the sensor sits exactly on a cell centre
            data, mask = oracle.ingest_f32(world.scan(x + 0.025, y + 0.025, yaw, geo), H.MAX_RANGE, geo.angle_increment)
            so = og.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
            sd = dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
            assert so == sd, f"yaw {yaw}: {so} != {sd}"
            H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)


def build_map(oracle, gc, geo, world, pushes=4):
    og, dg = make_pair(oracle, gc)
    for k in range(pushes):
        push_both(oracle, og, dg, world, geo, k * 3)
    # identical grids on both sides from here on (upload the oracle's dump so that later comparisons
    # isolate the kernel under test)
    dg.upload_tiles(*og.dump())
    return og, dg


@pytest.mark.parametrize("scene,map_log2,cs,geo", [
    ("room", 9, 0.05, synth.ScanGeometry.full_circle_360()),
    ("pillars", 10, 0.05, synth.ScanGeometry.utm30lx()),
])
def test_raycast_matches_oracle(oracle, scene, map_log2, cs, geo):
    gc = synth.GridConfig(map_log2, cs)
    world = synth.World(scene, gc)
    og, dg = build_map(oracle, gc, geo, world)
    for k in (1, 4, 10):
        pose, _ = H.sensor_pose(world, k)
        rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
        co, no, mo, cnt_o = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
        cd, nd, md, cnt_d = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
        assert np.array_equal(mo, md), f"hit masks differ at beams {np.nonzero(mo != md)[0][:10]}"
        assert cnt_o == cnt_d and cnt_o > 0.5 * geo.beams
        sel = np.repeat(mo.astype(bool), 2)
        assert np.max(np.abs(co[sel] - cd[sel])) <= 1e-9
        assert np.max(np.abs(no[sel] - nd[sel])) <= 1e-9


def test_raycast_outside_and_empty(oracle):
    """Sensor outside the grid and an empty map: no hits, no crash (RayCastPolar2D.cpp:137-146)."""
    gc = synth.GridConfig(8, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    og, dg = make_pair(oracle, gc)
    for pose in (synth.pose_matrix(-3.0, 2.0, 0.3), synth.pose_matrix(6.4, 6.4, 0.0)):
        rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
        _, _, mo, cnt_o = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
        _, _, md, cnt_d = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
        assert cnt_o == cnt_d == 0 and not md.any()


def icp_inputs(oracle, gc, geo, world, k, og):
    pose, (x, y, yaw) = H.sensor_pose(world, k)
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    co, no, mo, _ = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    # scan taken from a slightly different true pose => non-trivial registration
    r32 = world.scan(x + 0.05, y - 0.03, yaw + 0.015, geo)
    data, mask = oracle.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
    scene, ms, _ = oracle.scene_from_scan(rl, data, mask)
    M = co.reshape(-1, 2)[mo.astype(bool)]
    S = scene.reshape(-1, 2)[ms.astype(bool)]
    return pose, rl, rw, data, mask, M, S


@pytest.mark.parametrize("iters", [30, 25, 10, 11, 5])
def test_icp_matches_oracle(oracle, iters):
    gc = synth.GridConfig(10, 0.05)
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("pillars", gc)
    og, dg = build_map(oracle, gc, geo, world)
    pose, rl, rw, data, mask, M, S = icp_inputs(oracle, gc, geo, world, 2, og)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    ro = oracle.icp(M, S, pose, iters, 0.4, 0.02, bounds, nn_mode=0)
    rd = dg.icp(M, S, pose, dg.icp_params(iters, 0.4, 0.02))
    assert (ro["pairs"], ro["iterations"], ro["state"]) == (rd.pairs, rd.iterations, rd.state)
    d, a = H.pose_delta(ro["T"], rd.T)
    assert d <= TOL_POSE_M and a <= TOL_POSE_RAD, (d, a)
    assert abs(ro["rms"] - rd.rms) <= 1e-9
    # the registration actually moved the scan
    assert np.hypot(rd.T[0, 2], rd.T[1, 2]) > 0.02


@pytest.mark.parametrize("iters", [30, 11])
def test_icp_point_to_line_matches_oracle(oracle, iters):
    """PointToLine2DEstimator (PointToLineEstimator2D.cpp:52-157) on the ray cast's normals, the
    estimator north_star names; the node itself constructs the closed form.  Direct call and fused localize."""
    gc = synth.GridConfig(10, 0.05)
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("pillars", gc)
    og, dg = build_map(oracle, gc, geo, world)
    pose, (x, y, yaw) = H.sensor_pose(world, 2)
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    co, no, mo, _ = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    data, mask = oracle.ingest_f32(world.scan(x + 0.05, y - 0.03, yaw + 0.015, geo), H.MAX_RANGE, geo.angle_increment)
    scene, ms, _ = oracle.scene_from_scan(rl, data, mask)
    M = co.reshape(-1, 2)[mo.astype(bool)]
    N = no.reshape(-1, 2)[mo.astype(bool)]
    S = scene.reshape(-1, 2)[ms.astype(bool)]
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    ro = oracle.icp(M, S, pose, iters, 0.4, 0.02, bounds, nn_mode=0, model_normals_xy=N)
    rc = oracle.icp(M, S, pose, iters, 0.4, 0.02, bounds, nn_mode=0)
    p = dg.icp_params(iters, 0.4, 0.02, estimator=1)
    rd = dg.icp(M, S, pose, p, model_normals_xy=N)
    assert (ro["pairs"], ro["iterations"], ro["state"]) == (rd.pairs, rd.iterations, rd.state)
    d, a = H.pose_delta(ro["T"], rd.T)
    assert d <= TOL_POSE_M and a <= TOL_POSE_RAD, (d, a)
    assert abs(ro["rms"] - rd.rms) <= 1e-9
    # a different estimator, not the closed form under another name: its "rms" is a mean distance, not a mean square
    assert abs(ro["rms"] - rc["rms"]) > 1e-6 and np.hypot(rd.T[0, 2], rd.T[1, 2]) > 0.02
    # the unsorted / shuffled model takes its normals along
    perm = np.random.default_rng(5).permutation(len(M))
    rs = dg.icp(M[perm], S, pose, p, model_normals_xy=N[perm])
    assert (rs.pairs, rs.iterations, rs.state) == (rd.pairs, rd.iterations, rd.state)
    d, a = H.pose_delta(rs.T, rd.T)
    assert d <= 1e-9 and a <= 1e-9
    # fused: ray cast -> compaction -> registration on the device, normals included
    rf = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, p)
    assert (rf.n_model, rf.n_scene) == (len(M), len(S))
    assert (ro["pairs"], ro["iterations"], ro["state"]) == (rf.pairs, rf.iterations, rf.state)
    d, a = H.pose_delta(ro["T"], rf.T)
    assert d <= TOL_POSE_M and a <= TOL_POSE_RAD
    # the estimator needs normals and a known id
    with pytest.raises(capi.TsdError):
        dg.icp(M, S, pose, p)
    with pytest.raises(capi.TsdError):
        dg.icp(M, S, pose, dg.icp_params(iters, 0.4, 0.02, estimator=7))


def test_icp_degenerate(oracle):
    """<= 2 pairs: NOTMATCHABLE on the first step is reported as SUCCESS with T = I (Icp.cpp:489-505)."""
    gc = synth.GridConfig(8, 0.05)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    M = np.array([[1.0, 1.0], [2.0, 2.0], [3.0, 1.5]])
    S = M + 5.0     # farther than dist_filter_max
    pose = synth.pose_matrix(6.0, 6.0, 0.0)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    ro = oracle.icp(M, S, pose, 30, 0.4, 0.02, bounds)
    rd = dg.icp(M, S, pose, dg.icp_params(30, 0.4, 0.02))
    assert (ro["pairs"], ro["iterations"], ro["state"]) == (rd.pairs, rd.iterations, rd.state) == (0, 1, 5)
    assert np.array_equal(rd.T, np.eye(3))
    # empty model / scene (Icp.cpp:467-471)
    rd = dg.icp(np.zeros((0, 2)), S, pose, dg.icp_params(30, 0.4, 0.02))
    assert rd.state == 2


def test_localize_fused_matches_pieces(oracle):
    """tsd_localize == tsd_raycast + host compaction + tsd_icp, and == the oracle chain."""
    gc = synth.GridConfig(10, 0.05)
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("pillars", gc)
    og, dg = build_map(oracle, gc, geo, world)
    pose, rl, rw, data, mask, M, S = icp_inputs(oracle, gc, geo, world, 3, og)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    ro = oracle.icp(M, S, pose, 30, 0.4, 0.02, bounds)
    p = dg.icp_params(30, 0.4, 0.02)
    rf = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, p)
    assert (rf.n_model, rf.n_scene) == (len(M), len(S))
    assert (ro["pairs"], ro["iterations"], ro["state"]) == (rf.pairs, rf.iterations, rf.state)
    d, a = H.pose_delta(ro["T"], rf.T)
    assert d <= TOL_POSE_M and a <= TOL_POSE_RAD


def test_localize_fused_with_t_init_matches_direct_call_and_oracle(oracle):
    """registration_mode 3's registration: tsd_localize with tsd_icp_params.t_init (Tinit applied while the device stages the
    scene, Icp.cpp:481-486) == tsd_icp on the maskMatrix-compacted sets with the same Tinit == the oracle's Icp::iterate(Tinit)."""
    gc = synth.GridConfig(10, 0.05)
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("pillars", gc)
    og, dg = build_map(oracle, gc, geo, world)
    pose, rl, rw, data, mask, M, S = icp_inputs(oracle, gc, geo, world, 3, og)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    Tinit = synth.pose_matrix(0.04, -0.03, 0.02)
    ro = oracle.icp_init(M, S, pose, 30, 0.4, 0.02, bounds, Tinit, nn_mode=1)
    p = dg.icp_params(30, 0.4, 0.02, t_init=Tinit)
    rd = dg.icp(M, S, pose, p)
    rf = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, p)
    assert (rf.n_model, rf.n_scene) == (len(M), len(S))
    assert (rf.pairs, rf.iterations, rf.state) == (rd.pairs, rd.iterations, rd.state) == (ro["pairs"], ro["iterations"], ro["state"])
    d, a = H.pose_delta(rd.T, rf.T)
    assert d <= 1e-12 and a <= 1e-12 and abs(rf.rms - rd.rms) <= 1e-12
    d, a = H.pose_delta(ro["T"], rf.T)
    assert d <= 1e-9 and a <= 1e-9
    # and the identity Tinit is the plain call, bit for bit (x*1 + y*0 + 0 == x)
    r0 = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, dg.icp_params(30, 0.4, 0.02))
    r1 = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, dg.icp_params(30, 0.4, 0.02, t_init=np.eye(3)))
    assert np.array_equal(np.asarray(r0.T), np.asarray(r1.T)) and r0.rms == r1.rms and r0.pairs == r1.pairs


@pytest.mark.parametrize("fused", [False, True])
def test_closed_loop_trajectory(oracle, fused):
    """init -> [raycast -> ICP -> transform -> push] x K on both sides, poses compared every scan.
    fused = tsd_scan (one call per scan, gates and pose bookkeeping on the device)."""
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    poses = synth.trajectory(world, 12)
    scans = synth.scans_for(world, geo, poses)
    from tests.slam_driver import HipSlam, HipSlamFused, slam_kwargs
    kw = slam_kwargs(gc, geo)
    so = oracle.Slam(**kw)
    sh = (HipSlamFused if fused else HipSlam)(oracle, **kw)
    for k in range(len(scans)):
        ro = so.process_scan(scans[k])
        rh = sh.process_scan(scans[k])
        Po = np.array(ro.pose[:]).reshape(3, 3)
        d, a = H.pose_delta(Po, rh["pose"])
        assert d <= TOL_POSE_M and a <= TOL_POSE_RAD, f"scan {k}: {d} {a}"
        assert ro.pushed == rh["pushed"] and ro.reg_error == rh["reg_error"]
        if k > 0:
            assert ro.pairs == rh["pairs"] and ro.valid_model == rh["valid_model"]
            # tracks ground truth
            assert np.hypot(Po[0, 2] - poses[k, 0], Po[1, 2] - poses[k, 1]) < 0.15
    oi, oiw, ot, ow = so.grid.dump()
    gi, giw, gt, gw = sh.grid.download_tiles()
    assert np.array_equal(oi, gi)
    sel = oi.astype(bool)
    m = ~np.isnan(ot[sel])
    assert np.array_equal(np.isnan(ot[sel]), np.isnan(gt[sel]))
    assert np.max(np.abs(ot[sel][m] - gt[sel][m])) <= TOL_CELL
    assert np.max(np.abs(ow[sel] - gw[sel])) <= TOL_CELL


# ------------------------------------------------------------------------------------------------
# occupancy extraction (RayCastAxisAligned2D::calcCoords + ThreadGrid marking)
@pytest.mark.parametrize("inflate", [False, True])
def test_occupancy_matches_oracle(oracle, inflate):
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    content = np.full(gc.cells * gc.cells, -1, dtype=np.int8)
    for k in range(4):
        push_both(oracle, og, dg, world, geo, k * 4)
        oo, no = og.occupancy(content, inflate, 2)      # `content` persists like ThreadGrid::_occGridContent
        od, nd = dg.occupancy(inflate, 2)
        assert no == nd and no > 0
        assert np.array_equal(oo.reshape(gc.cells, gc.cells), od), \
            f"push {k}: {np.count_nonzero(oo.reshape(gc.cells, gc.cells) != od)} cells differ"
    assert set(np.unique(od)) <= {-1, 0, 100} and (od == 100).sum() > 50


# ------------------------------------------------------------------------------------------------
# edge cases of the scan itself
def test_color_image_matches_oracle(oracle):
    """TsdGrid::grid2ColorImage (TsdGrid.cpp:429-488), the RGB image ThreadGrid publishes with the occupancy map:
    byte for byte, at the grid's own size and at an arbitrary one (px / py accumulate by repeated addition)."""
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    near = np.full(geo.beams, 2.0, dtype=np.float32)
    for k, r in enumerate((None, None, near, None)):          # content, empty-with-weight and untouched tiles
        push_both(oracle, og, dg, world, geo, 4 * k, ranges_f32=r)
    for (w, h) in ((gc.cells, gc.cells), (300, 200), (1000, 37)):
        io = og.color_image(w, h)
        ih = dg.color_image(w, h)
        assert io.shape == ih.shape == (h, w, 3)
        assert np.array_equal(io, ih), f"{w}x{h}: {np.argwhere(io != ih)[:5]}"
    full = dg.color_image()
    assert (full[..., 1] == 255).any() and ((full[..., 1] == 0) & (full[..., 0] > 0)).any()      # free (green), behind the surface (red)
    assert (full.sum(axis=2) == 0).any()                                                     # unseen (black)


def test_text_grid_file_matches_oracle(oracle, tmp_path):
    """TsdGrid::storeGrid / TsdGrid(file) (TsdGrid.cpp:548-607, :25-110), the reference's text format:
    the file written from the device grid is byte-identical to the oracle's, loading it gives the same grid on both
    sides (interior cells at the file's 6 digits, halos at their init value until the next push), and the ray cast /
    the next push on the loaded grids agree."""
    gc = synth.GridConfig(8, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    near = np.full(geo.beams, 2.0, dtype=np.float32)
    for k, r in enumerate((None, near, None)):                # content tiles, empty-with-weight tiles, untouched tiles
        push_both(oracle, og, dg, world, geo, 3 * k, ranges_f32=r)
    fo, fh = tmp_path / "oracle.grid", tmp_path / "hip.grid"
    assert og.store_text(fo)
    dg.store_text(fh)
    bo, bh = fo.read_bytes(), fh.read_bytes()
    assert bo == bh and len(bo) > 1000
    head = bo.split(b"\n")[:4]
    assert head == [b"0.05", b"5", b"8", b"0.15"]
    # load into fresh grids
    og2 = oracle.Grid.load_text(fo, gc.cell_size)
    dg2 = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    dg2.load_text(fh)
    d_o, d_h = og2.dump(), dg2.download_tiles()
    H.assert_grids_equal(d_o, d_h, 0.0)
    flags, iw, tsd, w = d_h
    f0 = og.dump()
    un = flags == 0                                           # (a content tile's _initWeight is not in the file)
    assert np.array_equal(flags, f0[0]) and np.allclose(iw[un], f0[1][un], rtol=1e-5) and (iw[~un] == 0).all()
    t = tsd.reshape(-1, 33, 33)[flags.astype(bool)]
    t0 = f0[2].reshape(-1, 33, 33)[flags.astype(bool)]
    assert np.isnan(t[:, 32, :]).all() and np.isnan(t[:, :, 32]).all()          # the halo is not stored
    both = ~np.isnan(t0[:, :32, :32])
    assert np.array_equal(np.isnan(t[:, :32, :32]), ~both)
    assert np.allclose(t[:, :32, :32][both], t0[:, :32, :32][both], rtol=1e-5, atol=1e-12)
    # the loaded grids behave alike: ray cast and the next push
    pose, _ = H.sensor_pose(world, 2)
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    co, no, mo, cnt_o = og2.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    cd, nd, md, cnt_d = dg2.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    assert cnt_o == cnt_d and np.array_equal(mo, md)
    so, sd = push_both(oracle, og2, dg2, world, geo, 9)
    assert so == sd
    H.assert_grids_equal(og2.dump(), dg2.download_tiles(), TOL_CELL)
    # a file of another layout is refused
    other = capi.TsdGridDevice(9, gc.cell_size, gc.max_trunc)
    with pytest.raises(capi.TsdError):
        other.load_text(fh)


def test_fused_scan_with_host_calls_in_between(oracle):
    """tsd_scan enqueues the NEXT scan's ray cast right behind its push.  Whatever touches the grid, the sensor or
    the context's ray-cast outputs in between (another push, a freed footprint, a new sensor pose, an unfused ray
    cast, a map upload) must make the next tsd_scan cast again: the loop with such calls interleaved has to equal
    the same loop run through the unfused calls."""
    from tests.slam_driver import HipSlam, HipSlamFused, slam_kwargs
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    poses = synth.trajectory(world, 10)
    scans = synth.scans_for(world, geo, poses)
    kw = slam_kwargs(gc, geo)
    a, b = HipSlamFused(oracle, **kw), HipSlam(oracle, fused=False, **kw)

    def meddle(s, k):
        g = s.grid
        if k == 2:      # an extra push from somewhere else (e.g. a second sensor's mapping thread)
            pose = synth.pose_matrix(world.start[0] - 1.0, world.start[1] + 0.5, 1.0)
            data, mask = oracle.ingest_f32(world.scan(world.start[0] - 1.0, world.start[1] + 0.5, 1.0, geo), H.MAX_RANGE, geo.angle_increment)
            g.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        elif k == 4:
            g.free_footprint([world.start[0] + 2.0, world.start[1] - 1.0], 0.6, 0.6)
        elif k == 5:    # an unfused ray cast overwrites the context's model buffers
            rl, rw = H.world_rays(oracle, geo, synth.pose_matrix(world.start[0], world.start[1], 0.7), gc.cell_size)
            g.raycast(synth.pose_matrix(world.start[0], world.start[1], 0.7), rw, H.MIN_RANGE, H.MAX_RANGE)
        elif k == 7:    # the map is replaced by itself through the host
            g.upload_tiles(*g.download_tiles())

    for k in range(len(scans)):
        ra, rb = a.process_scan(scans[k]), b.process_scan(scans[k])
        d, ang = H.pose_delta(ra["pose"], rb["pose"])
        assert d <= 1e-9 and ang <= 1e-9, f"scan {k}: {d} {ang}"
        assert ra["pushed"] == rb["pushed"] and ra["pairs"] == rb["pairs"] and ra["valid_model"] == rb["valid_model"]
        meddle(a, k); meddle(b, k)
    H.assert_grids_equal(a.grid.download_tiles(), b.grid.download_tiles(), 0.0)


def test_staged_scan_dropped_every_scan_at_cfg3(oracle):
    """A scan staged ahead that is NOT the one that comes next is dropped by tsd_scan_submit.  The three scan / table buffers
    of the sensor are used in turn, and a drop must not advance that rotation twice per scan: the decoy staged after scan k+1
    would then overwrite the buffers the push of scan k may still be reading (nothing the host has seen orders that push).  A
    decoy is staged and dropped after EVERY scan here, at cfg 3 where a push takes longest; poses, push decisions and the
    whole-grid digest must equal the plain tsd_scan loop."""
    from tests.slam_driver import HipSlamFused, slam_kwargs
    gc, geo, _ = synth.CONFIGS["cfg3"]
    world = synth.World("pillars", gc)
    poses = synth.trajectory(world, 14)
    scans = synth.scans_for(world, geo, poses)
    kw = slam_kwargs(gc, geo)
    a, b = HipSlamFused(oracle, **kw), HipSlamFused(oracle, **kw)

    def ingest(r32):
        r = np.array(r32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        data, mask = oracle.ingest_f32(r, kw["max_range"], kw["angle_increment"])
        _, mask_push = oracle.ingest_f64(data, kw["max_range"], kw["angle_increment"])
        return data, mask, mask_push

    for k in range(len(scans)):
        ra = a.process_scan(scans[k])
        if k == 0:
            rb = b.process_scan(scans[k])
        else:
            data, mask, mask_push = ingest(scans[k])
            decoy = ingest(np.roll(np.asarray(scans[k - 1], dtype=np.float32), 97) * np.float32(0.5))   # nothing like the next scan
            sr = b.sensor.scan_ahead(data, mask, mask_push, b.params, b.gates, nxt=decoy)
            rb = dict(pose=np.array(sr.pose[:]).reshape(3, 3), pushed=int(sr.pushed), pairs=int(sr.icp.pairs), valid_model=int(sr.icp.n_model))
        d, ang = H.pose_delta(ra["pose"], rb["pose"])
        assert d == 0.0 and ang == 0.0, f"scan {k}: {d} {ang}"
        assert ra["pushed"] == rb["pushed"] and ra["pairs"] == rb["pairs"] and ra["valid_model"] == rb["valid_model"], k
    assert a.grid.digest() == b.grid.digest()


def test_hip_pair_chain_equals_compiled_reference_fixture():
    """tests/golden/ref_chain_pairs.npz holds pair lists produced by the COMPILED reference (its own PairAssignment.cpp,
    DistanceFilter.cpp, ReciprocalFilter.cpp; tests/golden/make_ref_chain_fixture.py) for repeated determinePairs() calls on a
    static scene.  The registration kernel's own pair formation (tsd_icp_pairs: the k_icp code path with the scene held still)
    must give the same lists, bit for bit and in the same order, for all five cases (icp_iterations 30 / 25 / 11 / 10 / 4, i.e.
    the threshold schedule incl. its unsigned wrap): row I4 of the HIP side pinned to the reference directly, no oracle in between."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_chain_pairs.npz"))
    g = capi.TsdGridDevice(8, 0.05, 0.15)
    total = 0
    for case in range(int(z["n_cases"])):
        model, scene = z[f"model_{case}"], z[f"scene_{case}"]
        iters, calls = int(z[f"iters_{case}"]), int(z[f"calls_{case}"])
        b = z[f"bounds_{case}"]
        prm = g.icp_params(iters, 0.4, 0.02)
        prm.min_x, prm.max_x, prm.min_y, prm.max_y = float(b[0]), float(b[1]), float(b[2]), float(b[3])
        got = g.icp_pairs(model, scene, np.eye(3), prm, calls)
        for k in range(calls):
            assert np.array_equal(z[f"pm_{case}_{k}"], got[k][0]), (case, k, "model indices")
            assert np.array_equal(z[f"ps_{case}_{k}"], got[k][1]), (case, k, "scene indices")
            total += len(got[k][0])
        # the same inputs through the ordinary registration: its first step's pair count is the first list's length
        r = g.icp(model, scene, np.eye(3), prm)
        assert int(g.icp_trace(1)[0][0]) == len(z[f"pm_{case}_0"])
    assert total > 2000
    g.close()


def test_push_degenerate_scans(oracle):
    """All beams masked / all infinite / a single valid beam / every beam at max range: same tile
    classification and cells on both sides, no crash, nothing updated where nothing is visible."""
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    zeros = np.zeros(geo.beams, dtype=np.float32)                   # maskZeroDepth: everything masked
    infs = np.full(geo.beams, np.inf, dtype=np.float32)             # valid-infinite: carves within lowReflectivityRange
    one = zeros.copy(); one[geo.beams // 2] = 3.0
    far = np.full(geo.beams, 29.999, dtype=np.float32)
    for name, r in (("zeros", zeros), ("infs", infs), ("one", one), ("far", far), ("zeros again", zeros)):
        so, sd = push_both(oracle, og, dg, world, geo, 0, ranges_f32=r)
        assert so == sd, name
        H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
        if name.startswith("zeros"):
            assert sd["cells_updated"] == 0 and sd["tiles_update"] == 0


def test_push_sensor_outside_grid_and_small_beam_counts(oracle):
    gc = synth.GridConfig(7, 0.1)       # 128 x 128 cells, 16 tiles
    og, dg = make_pair(oracle, gc)
    for beams in (1, 2, 3, 64, 65):
        geo = synth.ScanGeometry(beams, -0.4, 0.8 / max(beams - 1, 1))
        r = np.linspace(2.0, 5.0, beams).astype(np.float32)
        data, mask = oracle.ingest_f32(r, H.MAX_RANGE, geo.angle_increment)
        for pose in (synth.pose_matrix(6.4, 6.4, 0.3), synth.pose_matrix(-2.0, 6.0, 0.0), synth.pose_matrix(14.0, 14.0, 3.0)):
            so = og.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
            sd = dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
            assert so == sd, (beams, pose[0, 2])
    H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
    # more beams than the kernels are built for is an error, not a crash
    with pytest.raises(capi.TsdError):
        dg.push(synth.pose_matrix(6.4, 6.4, 0.0), np.ones(capi.MAX_BEAMS + 1), np.ones(capi.MAX_BEAMS + 1, dtype=np.uint8),
                1e-3, 0.0, 30.0, 0.001, 2.0)


def test_max_beams_scan(oracle):
    """TSD_MAX_BEAMS = 4096 beams over 360 degrees: push parity and an ICP with 2048 points (the limits of
    the LDS-resident scan / model)."""
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry(capi.MAX_BEAMS, -math.pi, 2.0 * math.pi / capi.MAX_BEAMS)
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    for k in range(2):
        so, sd = push_both(oracle, og, dg, world, geo, k * 5)
        assert so == sd
    H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
    rng = np.random.default_rng(7)
    ang = np.sort(rng.uniform(-math.pi, math.pi, capi.MAX_ICP_POINTS))
    M = np.stack([6.0 * np.cos(ang) + 0.3 * np.cos(5 * ang), 4.0 * np.sin(ang)], axis=1)
    c, s_ = math.cos(0.02), math.sin(0.02)
    S = (M[::1] @ np.array([[c, -s_], [s_, c]]).T + np.array([0.05, -0.03]))[rng.permutation(capi.MAX_ICP_POINTS)]
    pose = synth.pose_matrix(12.8, 12.8, 0.0)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    ro = oracle.icp(M, S, pose, 30, 0.4, 0.02, bounds, nn_mode=1)
    rd = dg.icp(M, S, pose, dg.icp_params(30, 0.4, 0.02))
    assert (ro["pairs"], ro["iterations"], ro["state"]) == (rd.pairs, rd.iterations, rd.state)
    d, a = H.pose_delta(ro["T"], rd.T)
    assert d <= TOL_POSE_M and a <= TOL_POSE_RAD


def test_icp_exact_ties_and_unsorted_models(oracle):
    """Model points on a lattice and scene points exactly between them: exact d2 ties (lowest model index
    wins, like a first-minimum linear scan); model given in random order (tsd_icp sorts by angle itself)."""
    gc = synth.GridConfig(8, 0.05)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    xs, ys = np.meshgrid(np.arange(-2.0, 2.01, 0.25), np.arange(-1.5, 1.51, 0.25))
    M = np.stack([xs.ravel(), ys.ravel()], axis=1)
    M = M[np.hypot(M[:, 0], M[:, 1]) > 0.3]
    rng = np.random.default_rng(3)
    M = M[rng.permutation(len(M))]
    S = M[: len(M) // 2] + np.array([0.125, 0.0])          # exactly half way between two lattice points
    pose = synth.pose_matrix(6.4, 6.4, 0.0)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    for iters in (1, 3, 30):
        ro = oracle.icp(M, S, pose, iters, 0.4, 0.02, bounds, nn_mode=0)
        rd = dg.icp(M, S, pose, dg.icp_params(iters, 0.4, 0.02))
        assert (ro["pairs"], ro["iterations"], ro["state"]) == (rd.pairs, rd.iterations, rd.state), iters
        d, a = H.pose_delta(ro["T"], rd.T)
        assert d <= TOL_POSE_M and a <= TOL_POSE_RAD


def test_push_sensor_exactly_on_a_cell_centre(oracle):
    """The cell whose centre IS the sensor position (heading 0: PoseInv * centre is exactly (0, 0), atan2(0, 0) = 0 names the beam at
    angle 0, the distance is 0 and sd = the reading): round 4, tools/fuzz_parity.py with exact poses -- the update kernel's square root
    without the zero pass-through made 0 * inf = NaN of that distance and silently skipped the cell."""
    gc = synth.GridConfig(9, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("pillars", gc)
    og, dg = make_pair(oracle, gc)
    ix, iy = 260, 250
    x, y = (ix + 0.5) * gc.cell_size, (iy + 0.5) * gc.cell_size
    for k, yaw in enumerate((0.0, 0.0, math.pi / 2)):
        pose = synth.pose_matrix(x, y, yaw)
        data, mask = oracle.ingest_f32(world.scan(x, y, yaw, geo), H.MAX_RANGE, geo.angle_increment)
        so = og.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        sd = dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        if yaw == 0.0:
            assert so == sd, f"push {k}: {so} {sd}"
            H.assert_grids_equal(og.dump(), dg.download_tiles(), 0.0)
    # the cell under the sensor was updated by both heading-0 pushes (weight = two measurements' worth)
    oi, oiw, ot, ow = og.dump()
    p = (iy // 32) * (gc.cells // 32) + ix // 32
    w_cell = ow.reshape(-1, 33, 33)[p, iy % 32, ix % 32]
    t_cell = ot.reshape(-1, 33, 33)[p, iy % 32, ix % 32]
    assert w_cell > 0.0 and t_cell == 1.0, (w_cell, t_cell)


def test_icp_four_way_exact_ties(oracle):
    """Scene points at the CENTRE of a lattice cell: four model points at exactly the same distance, given in random order, so that
    the lowest original index is as often the third or fourth that the search meets as the first (round 4, tools/fuzz_icp.py seed 11:
    the whole-wave walk compared distances alone and kept whichever of the tied points it met first; pair counts agreed, pairs and T
    did not).  The first pair list must equal the oracle's, pair for pair."""
    gc = synth.GridConfig(8, 0.05)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    for seed, h in ((1, 0.125), (2, 0.25), (3, 0.125)):
        rng = np.random.default_rng(seed)
        xs, ys = np.meshgrid(np.arange(-2.0, 2.01, h), np.arange(-1.5, 1.51, h))
        M = np.stack([xs.ravel(), ys.ravel()], axis=1)
        M = M[np.hypot(M[:, 0], M[:, 1]) > 0.3]
        M = M[rng.permutation(len(M))]
        S = M[rng.choice(len(M), len(M) // 3, replace=False)] + np.array([h / 2, h / 2])
        pose = synth.pose_matrix(6.4, 6.4, float(rng.uniform(-3, 3)))
        pm, ps, _ = oracle.icp_pairs(M, S, pose, 3, 0.2, 0.02, bounds, 0.2 * 0.2, nn_mode=0)
        hm, hs = dg.icp_pairs(M, S, pose, dg.icp_params(3, 0.2, 0.02), 1)[0]
        assert sorted(zip(ps.tolist(), pm.tolist())) == sorted(zip(hs.tolist(), hm.tolist())), f"lattice {h}: pair lists differ"
        assert len(pm) > 20
        for iters in (1, 3, 30):
            ro = oracle.icp(M, S, pose, iters, 0.2, 0.02, bounds, nn_mode=0)
            rd = dg.icp(M, S, pose, dg.icp_params(iters, 0.2, 0.02))
            assert (ro["pairs"], ro["iterations"], ro["state"]) == (rd.pairs, rd.iterations, rd.state), (h, iters)
            d, a = H.pose_delta(ro["T"], rd.T)
            assert d <= 1e-9 and a <= 1e-9, (h, iters, d, a)


def test_icp_helpers_change_nothing(oracle):
    """Step 0's nearest-neighbour searches come from helper workgroups by default (icp_kernels.hip: IcpSeed) -- the same search function on
    the same inputs, run on other compute units while the registering workgroup sets itself up.  With the helpers switched off
    (tsd_debug_set_icp_helpers) the registration searches itself: T, rms, counts and the WHOLE per-iteration trace (pairs, rms,
    threshold, state, Tlast) must be bit-identical, for the closed form and the point-to-line estimator, direct and fused calls, point
    sets with unresolvable windows (lattices: exact ties; sparse far points: whole-wave searches), and agree with the oracle."""
    gc = synth.GridConfig(9, 0.05)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    rng = np.random.default_rng(77)
    cases = []
    for n in (40, 300, 1081, 1500, 2000):                       # one helper .. several, both workgroup shapes
        phi = np.sort(rng.uniform(-2.3, 2.3, n))
        r = 3.0 + 1.5 * np.sin(3 * phi) + rng.normal(0, 0.01, n)
        M = np.stack([r * np.cos(phi), r * np.sin(phi)], axis=1)
        S = M[rng.permutation(n)[: max(8, int(0.9 * n))]] + rng.normal(0, 0.02, (max(8, int(0.9 * n)), 2))
        a = rng.uniform(-0.03, 0.03)
        R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        S = S @ R.T + rng.uniform(-0.05, 0.05, 2)
        S[:: 7] += rng.uniform(0.3, 0.8, (len(S[:: 7]), 2))    # far from the model: whole-wave searches
        if n >= 1081:
            S[n // 3: n // 3 + n // 5] += (1.5, 0.7)           # a whole sector of them (a scan that turns into unseen space): the helpers' sweeps
        cases.append((M, S))
    xs, ys = np.meshgrid(np.arange(-2.0, 2.01, 0.25), np.arange(-1.5, 1.51, 0.25))
    L = np.stack([xs.ravel(), ys.ravel()], axis=1)
    L = L[np.hypot(L[:, 0], L[:, 1]) > 0.3]
    cases.append((L[rng.permutation(len(L))], L[rng.choice(len(L), len(L) // 3, replace=False)] + 0.125))       # exact ties
    pose = synth.pose_matrix(12.8, 12.8, 0.4)
    for ci, (M, S) in enumerate(cases):
        for est in (0, 1):
            nrm = None
            if est == 1:
                nrm = M / np.linalg.norm(M, axis=1, keepdims=True)
            out = []
            if est == 1 and len(M) > 1536:
                continue                                           # (the model normals of that many points do not fit the LDS: TSD_E_CAPACITY)
            for on in (True, False):
                dg.set_icp_helpers(on)
                p = dg.icp_params(30, 0.4, 0.02, estimator=est)
                r = dg.icp(M, S, pose, p, model_normals_xy=nrm)
                out.append((r, dg.icp_trace(r.iterations)))
            dg.set_icp_helpers(True)
            (ra, ta), (rb, tb) = out
            # the hand-off DELIVERS: with the helpers on, (nearly) every scene point starts from their granules -- a broken tag or stride,
            # or helpers that never ran, would fall back to the self-search silently (same results, only slower) -- and none without
            # (the lattice: two thirds of its points are exact ties that a helper's single wave walks one by one -- longer than the
            # registration waits (25 us), which then searches those itself: some delivered is all that can be asked there)
            need = 1 if ci == len(cases) - 1 else 0.9 * len(S)
            assert ra.seeded >= need and rb.seeded == 0, (ci, est, ra.seeded, rb.seeded, len(S))
            assert (ra.pairs, ra.iterations, ra.state) == (rb.pairs, rb.iterations, rb.state), (ci, est)
            assert np.array_equal(ra.T, rb.T) and ra.rms == rb.rms, (ci, est, ra.T - rb.T)
            assert np.array_equal(ta, tb, equal_nan=True), (ci, est)
            if est == 0:
                ro = oracle.icp(M, S, pose, 30, 0.4, 0.02, bounds, nn_mode=0)
                assert (ro["pairs"], ro["iterations"], ro["state"]) == (ra.pairs, ra.iterations, ra.state), ci
                d, a = H.pose_delta(ro["T"], ra.T)
                assert d <= 1e-9 and a <= 1e-9, (ci, d, a)
    # fused: ray cast + registration from a mapped grid
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("room", gc)
    pose0, (x, y, yaw) = H.sensor_pose(world, 0)
    r32 = world.scan(x, y, yaw, geo)
    data, mask = oracle.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
    dg.push(pose0, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)
    pose1, (x1, y1, yaw1) = H.sensor_pose(world, 2)
    d1, m1 = oracle.ingest_f32(world.scan(x1, y1, yaw1, geo), H.MAX_RANGE, geo.angle_increment)
    rl, rw = H.world_rays(oracle, geo, pose0, gc.cell_size)
    out = []
    for on in (True, False):
        dg.set_icp_helpers(on)
        r = dg.localize(pose0, rw, rl, d1, m1, H.MIN_RANGE, H.MAX_RANGE, dg.icp_params(30, 0.4, 0.02))
        out.append((r, dg.icp_trace(r.iterations)))
    dg.set_icp_helpers(True)
    (ra, ta), (rb, tb) = out
    assert ra.seeded >= 0.9 * ra.n_scene and rb.seeded == 0, (ra.seeded, rb.seeded, ra.n_scene)
    assert ra.pairs > 500 and (ra.pairs, ra.iterations, ra.state, ra.n_model, ra.n_scene) == (rb.pairs, rb.iterations, rb.state, rb.n_model, rb.n_scene)
    assert np.array_equal(ra.T, rb.T) and np.array_equal(ta, tb, equal_nan=True)


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: cfg 2 (4096^2) cell-for-cell against the oracle, cfg 3 (16384^2) through
# size-independent properties plus the oracle on the touched tiles
def grid_properties(init, iw, tsd, w):
    sel = init.astype(bool)
    assert np.all(iw >= 0.0) and np.all(iw <= 32.0)
    assert np.all(w[sel] >= 0.0) and np.all(w[sel] <= 32.0), "weights are capped at TSDGRIDMAXWEIGHT"
    t = tsd[sel]
    m = ~np.isnan(t)
    assert np.all(t[m] <= 1.0), "tsd is truncated at +1"
    assert np.all(w[sel][~m] == 0.0), "a NaN cell carries no weight"


def test_push_comb_scene(oracle):
    """BASELINE's bandwidth-stress scene: ranges alternate 5 m / 25 m every 8 beams, so nearly every tile within
    25 m is seen by some beam and hidden from its neighbours -- the range-maximum / range-minimum tables of
    isInRange and the beam windows of the update kernel at their worst."""
    gc = synth.GridConfig(11, 0.025)                     # 51.2 m: the 25 m beams stay inside
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("comb", gc)
    og, dg = make_pair(oracle, gc)
    for k in range(3):
        so, sd = push_both(oracle, og, dg, world, geo, 7 * k)
        assert so == sd, f"push {k}: stats differ\n oracle {so}\n hip    {sd}"
        H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
    assert sd["tiles_update"] > 1000 and sd["cells_updated"] > 400000


def test_cfg2_full_size_push_and_localize(oracle):
    gc, geo, scene = synth.CONFIGS["cfg2"]
    world = synth.World(scene, gc)
    og, dg = make_pair(oracle, gc)
    for k in range(4):
        so, sd = push_both(oracle, og, dg, world, geo, k * 4)
        assert so == sd
    assert sd["tiles_total"] == 16384 and sd["cells_updated"] > 200000
    od, dd = og.dump(), dg.download_tiles()
    H.assert_grids_equal(od, dd, TOL_CELL)
    grid_properties(*dd)
    # a push from the same pose twice only moves weights / averages, never the tile set
    init0 = dd[0].copy()
    push_both(oracle, og, dg, world, geo, 12)
    push_both(oracle, og, dg, world, geo, 12)
    dd2 = dg.download_tiles()
    assert np.all(dd2[0] >= init0)
    H.assert_grids_equal(og.dump(), dd2, TOL_CELL)
    pose, rl, rw, data, mask, M, S = icp_inputs(oracle, gc, geo, world, 6, og)
    p = dg.icp_params(30, 0.4, 0.02)
    rf = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, p)
    ro = oracle.icp(M, S, pose, 30, 0.4, 0.02, (0.0, og.max_x, 0.0, og.max_x), nn_mode=1)
    assert (rf.n_model, rf.n_scene, rf.pairs) == (len(M), len(S), ro["pairs"])
    d, a = H.pose_delta(ro["T"], rf.T)
    assert d <= TOL_POSE_M and a <= TOL_POSE_RAD


def test_cfg3_full_size_properties(oracle):
    """16384 x 16384 cells @ 0.01 m (262144 tiles, 4.6 GB of fp64 cells): stats equal to the oracle's,
    properties of the grid, and cell-for-cell equality on the tiles the pushes touched."""
    gc, geo, scene = synth.CONFIGS["cfg3"]
    world = synth.World(scene, gc)
    og, dg = make_pair(oracle, gc)
    for k in range(2):
        so, sd = push_both(oracle, og, dg, world, geo, k * 5)
        assert so == sd
    assert sd["tiles_total"] == 262144 and sd["cells_updated"] > 1000000
    oi, oiw = og.tile_state()
    di, diw = dg.download_tile_state()
    assert np.array_equal(oi, di) and np.array_equal(oiw, diw)
    # compare cells on a sample of initialised tiles through the ray caster: same hits from both grids
    pose, (x, y, yaw) = H.sensor_pose(world, 7)
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    co, no, mo, cnt_o = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    cd, nd, md, cnt_d = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    assert np.array_equal(mo, md) and cnt_o > 0.5 * geo.beams
    sel = np.repeat(mo.astype(bool), 2)
    assert np.max(np.abs(co[sel] - cd[sel])) <= 1e-9 and np.max(np.abs(no[sel] - nd[sel])) <= 1e-9


def test_map_size_15_smoke(oracle):
    """The largest grid the node accepts (/root/reference/src/SlamNode.cpp:71-75: map_size up to 15 = 32 768 x 32 768 cells, 1 048 576
    tiles; 18.8 GB of fp64 cells on the device): two pushes and a ray cast against the oracle -- push statistics, the state of every
    tile and the hit mask exact, coordinates within 1e-9.  The device side of this test allocates 18.8 GB of cells + 1 GB of occupancy map
    + the per-tile arrays (20.1 GB in all); it is skipped only on a device with less than 24 GB free, and says what it saw."""
    need = 24 * (1 << 30)
    free_b, total_b = capi.device_memory(0)        # (through the product: the HIP runtime libtsd_hip.so itself is linked against)
    if free_b < need:
        pytest.skip(f"map_size 15 needs {need / 2**30:.0f} GiB of device memory: {free_b / 2**30:.1f} GiB free of {total_b / 2**30:.1f} GiB")
    gc = synth.GridConfig(15, 0.01)
    geo = synth.ScanGeometry.utm30lx()
    world = synth.World("pillars", gc)
    og, dg = make_pair(oracle, gc)
    assert dg.tiles == 1 << 20 and og.tiles == dg.tiles
    for k in range(2):
        so, sd = push_both(oracle, og, dg, world, geo, k * 5)
        assert so == sd, f"push {k}: {so} / {sd}"
    assert sd["tiles_total"] == 1 << 20 and sd["cells_updated"] > 1000000
    oi, oiw = og.tile_state()
    di, diw = dg.download_tile_state()
    assert np.array_equal(oi, di) and np.array_equal(oiw, diw) and int(oi.sum()) > 1000
    pose, _ = H.sensor_pose(world, 7)
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    co, no, mo, cnt_o = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    cd, nd, md, cnt_d = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    assert np.array_equal(mo, md) and cnt_o == cnt_d and cnt_o > 0.5 * geo.beams
    sel = np.repeat(mo.astype(bool), 2)
    assert np.max(np.abs(co[sel] - cd[sel])) <= 1e-9 and np.max(np.abs(no[sel] - nd[sel])) <= 1e-9
    dg.close()


# ------------------------------------------------------------------------------------------------
# whole-grid digest (tsd_grid_digest): the hash the cfg 1-3 fixtures pin, against the oracle's digest of its own dump
def test_grid_digest_matches_oracle(oracle):
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    og, dg = make_pair(oracle, gc)
    assert og.digest() == dg.digest()                                  # empty grid
    near = np.full(geo.beams, 2.0, dtype=np.float32)
    for k, r in enumerate((None, None, near, None)):                   # content, empty-with-weight, emptied tiles
        push_both(oracle, og, dg, world, geo, 4 * k, ranges_f32=r)
        do, dd = og.digest(), dg.digest()
        assert do["hash"] == dd["hash"] and do["cells_valid"] == dd["cells_valid"] > 0
        assert do["tiles_initialized"] == dd["tiles_initialized"]
        assert abs(do["sum_tsd"] - dd["sum_tsd"]) <= 1e-9 * max(1.0, abs(do["sum_tsd"]))
        assert abs(do["sum_weight"] - dd["sum_weight"]) <= 1e-9 * max(1.0, abs(do["sum_weight"]))
    # the hash sees a single changed cell
    i, iw, t, w = dg.download_tiles()
    p = int(np.nonzero(i)[0][0])
    t[p, 5] = 0.123 if not t[p, 5] == 0.123 else 0.5
    dg.upload_tiles(i, iw, t, w)
    assert dg.digest()["hash"] != do["hash"]


# ------------------------------------------------------------------------------------------------
# 32-bit fixed-point cell storage (lib/libtsd_hip_q32.so, -DTSD_STORAGE_Q32): 8 bytes per cell instead of 16.
# north_star: "SDF/weight within 1e-5"; SURVEY 7 hard parts: "verify <= 1e-5 after >= 1000 pushes".
def test_q32_storage_1000_pushes(oracle):
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc, storage="q32")
    assert dg.lib.tsd_storage_bits() == 32
    n = 1200
    for k in range(n):
        kk = k % 40                                       # back and forth inside the room: cells collect hundreds of updates
        pose, (x, y, yaw) = H.sensor_pose(world, kk if (k // 40) % 2 == 0 else 40 - kk)
        r32 = world.scan(x, y, yaw, geo)
        data, mask = oracle.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
        so = og.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        if k % 100 == 0 or k == n - 1:
            sd = dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
            assert so == sd, f"push {k}: the work counters do not depend on the storage"
        else:
            dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)
    dt, dw = H.assert_grids_equal(og.dump(), dg.download_tiles(), TOL_CELL)
    _, _, _, ow = og.dump()
    assert ow.max() > 5.0, "cells must have collected many updates for this to mean anything"
    assert dt <= n * 2.0 ** -31 * 4 and dw <= n * 2.0 ** -27       # the bound of csrc/tsd_device.hpp (tsd: + the weight's share)
    print("q32 after %d pushes: max |d tsd| %.3e  max |d weight| %.3e  (max weight %.2f)" % (n, dt, dw, ow.max()))


def test_q32_closed_loop(oracle):
    from tests.slam_driver import HipSlam, slam_kwargs
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, 40)
    scans = synth.scans_for(world, geo, poses)
    kw = slam_kwargs(gc, geo)
    so, sh = oracle.Slam(**kw), HipSlam(oracle, fused=True, storage="q32", **kw)
    for k in range(len(scans)):
        ro, rh = so.process_scan(scans[k]), sh.process_scan(scans[k])
        d, a = H.pose_delta(np.array(ro.pose[:]).reshape(3, 3), rh["pose"])
        assert d <= TOL_POSE_M and a <= TOL_POSE_RAD, f"scan {k}: {d} m {a} rad"
    H.assert_grids_equal(so.grid.dump(), sh.grid.download_tiles(), TOL_CELL)


# ------------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/oracle_*.npz): the HIP path against numbers fixed at commit
# time, without calling the oracle
def test_golden_push_raycast_icp_fixture():
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_push_raycast_icp.npz"))
    dg = capi.TsdGridDevice(int(f["map_size_log2"]), float(f["cell_size"]), float(f["max_trunc"]))
    res, phi = float(f["angle_increment"]), float(f["angle_min"])
    from ohm_tsd_slam_amd import facade
    import ctypes as C
    HL = facade.load_library()
    for k in range(len(f["push_poses"])):
        r = np.ascontiguousarray(f["push_scans"][k], dtype=np.float32)
        data = np.zeros(r.size); mask = np.zeros(r.size, dtype=np.uint8)
        HL.tsd_host_sensor_ingest_f32(r.ctypes.data_as(C.POINTER(C.c_float)), r.size, res, phi, H.MAX_RANGE,
                                      data.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), 0)
        st = dg.push(f["push_poses"][k], data, mask, res, phi, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        assert [st[n] for n in sorted(st)] == list(f["push_stats"][k])
    H.assert_grids_equal((f["init"], f["init_weight"], f["tsd"], f["weight"]), dg.download_tiles(), TOL_CELL)
    cd, nd, md, cnt = dg.raycast(f["rc_pose"], f["rc_rays_world"], H.MIN_RANGE, H.MAX_RANGE)
    assert np.array_equal(md, f["rc_mask"])
    sel = np.repeat(md.astype(bool), 2)
    assert np.max(np.abs(cd[sel] - f["rc_coords"][sel])) <= 1e-9 and np.max(np.abs(nd[sel] - f["rc_normals"][sel])) <= 1e-9
    p = dg.icp_params(30, 0.4, 0.02)
    rd = dg.icp(f["icp_model"], f["icp_scene"], f["rc_pose"], p)
    assert (rd.pairs, rd.iterations, rd.state) == (int(f["icp_pairs"]), int(f["icp_iterations"]), int(f["icp_state"]))
    d, a = H.pose_delta(f["icp_T"], rd.T)
    assert d <= TOL_POSE_M and a <= TOL_POSE_RAD
    tr = dg.icp_trace(rd.iterations)
    assert np.array_equal(tr[:, 0], f["icp_trace"][:, 0]), "pairs per step"
    assert np.max(np.abs(tr[:, 1] - f["icp_trace"][:, 1])) <= 1e-9, "mean squared distance per step"
    # fused localize on the same inputs
    rf = dg.localize(f["rc_pose"], f["rc_rays_world"], f["rc_rays_local"], f["icp_ranges"], f["icp_mask"], H.MIN_RANGE,
                     H.MAX_RANGE, p)
    assert (rf.pairs, rf.n_model, rf.n_scene) == (int(f["icp_pairs"]), len(f["icp_model"]), len(f["icp_scene"]))


def test_golden_n1_n4_fixture():
    """tests/golden/oracle_n1_n4.npz WITHOUT the oracle: occupancy maps (plain / inflated) after every push and the colour image
    (row N1) byte for byte; the point-to-line registration (row N4) on the model / scene of the push fixture."""
    import os
    import ctypes as C
    from ohm_tsd_slam_amd import facade
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    f = np.load(os.path.join(gold, "oracle_n1_n4.npz"))
    res, phi = float(f["angle_increment"]), float(f["angle_min"])
    HL = facade.load_library()
    for inflate, key in ((False, "plain"), (True, "inflated")):
        dg = capi.TsdGridDevice(int(f["map_size_log2"]), float(f["cell_size"]), float(f["max_trunc"]))
        for k in range(len(f["push_poses"])):
            r = np.ascontiguousarray(f["push_scans"][k], dtype=np.float32)
            data = np.zeros(r.size); mask = np.zeros(r.size, dtype=np.uint8)
            HL.tsd_host_sensor_ingest_f32(r.ctypes.data_as(C.POINTER(C.c_float)), r.size, res, phi, H.MAX_RANGE,
                                          data.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), 0)
            dg.push(f["push_poses"][k], data, mask, res, phi, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)
            od, nd = dg.occupancy(inflate, 2)
            assert nd == int(f["marks_" + key][k]) and np.array_equal(od, f["occ_" + key][k]), (key, k)
        if not inflate:
            assert np.array_equal(np.asarray(dg.color_image()), f["color_image"])
    p = np.load(os.path.join(gold, "oracle_push_raycast_icp.npz"))
    dg = capi.TsdGridDevice(int(p["map_size_log2"]), float(p["cell_size"]), float(p["max_trunc"]))
    rd = dg.icp(p["icp_model"], p["icp_scene"], p["rc_pose"], dg.icp_params(30, 0.4, 0.02, estimator=1), model_normals_xy=f["ptl_normals"])
    assert (rd.pairs, rd.iterations, rd.state) == (int(f["ptl_pairs"]), int(f["ptl_iterations"]), int(f["ptl_state"]))
    d, a = H.pose_delta(f["ptl_T"], rd.T)
    assert d <= 1e-9 and a <= 1e-9 and abs(rd.rms - float(f["ptl_rms"])) <= 1e-9


def test_golden_trajectory_fixture():
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_trajectory.npz"))
    from ohm_tsd_slam_amd import facade
    gc = synth.GridConfig(int(f["map_size_log2"]), float(f["cell_size"]))
    geo = synth.ScanGeometry.full_circle_360()
    # the facade receives angle_min / angle_increment as float32 like a LaserScan; the fixture was made with
    # the double geometry, so poses are compared at the tolerance, flags and counts exactly
    node = facade.SlamNode(facade.node_params(gc, geo), synchronous=True)
    for k in range(len(f["scans"])):
        node.laser(f["scans"][k], geo.angle_min, geo.angle_increment)
        rep = node.report()
        row = f["rows"][k]
        d, a = H.pose_delta(row[:9].reshape(3, 3), rep["pose"])
        assert d <= TOL_POSE_M and a <= TOL_POSE_RAD, f"scan {k}"
        assert (int(row[11]), int(row[12])) == (rep["pushed"], rep["reg_error"])
    init, _ = node.grid().download_tile_state()
    assert np.array_equal(init, f["init"])
    node.close()


@pytest.mark.parametrize("tag,cfg,scene", [("cfg1", "cfg1", "room"), ("cfg2", "cfg2", "pillars"), ("cfg3", "cfg3", "pillars"),
                                           ("cfg3comb", "cfg3", "comb")])
def test_golden_baseline_config_fixture(tag, cfg, scene):
    """tests/golden/oracle_baseline_configs.npz, checked WITHOUT the oracle: BASELINE configs 1-3 (+ cfg3 /
    comb) at full size.  Per push the work counters and the digest of the whole grid -- the 64-bit hash of the canonical
    dump is bit-exact or it is not equal; ray cast; registration per iteration (pairs, rms, threshold, state, Tlast);
    closed loop through the C++ facade (per-scan pose, counts, push decisions, final grid sums)."""
    import ctypes as C
    import os
    from ohm_tsd_slam_amd import facade
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_baseline_configs.npz"))
    gc, geo, _ = synth.CONFIGS[cfg]
    HL = facade.load_library()

    def ingest(r32):
        r = np.ascontiguousarray(r32, dtype=np.float32)
        data = np.zeros(r.size); mask = np.zeros(r.size, dtype=np.uint8)
        HL.tsd_host_sensor_ingest_f32(r.ctypes.data_as(C.POINTER(C.c_float)), r.size, geo.angle_increment, geo.angle_min, H.MAX_RANGE,
                                      data.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), 0)
        return data, mask

    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(len(f[f"{tag}_push_poses"])):
        data, mask = ingest(f[f"{tag}_push_scans"][k])
        st = dg.push(f[f"{tag}_push_poses"][k], data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        assert [st[n] for n in sorted(st)] == list(f[f"{tag}_push_stats"][k]), f"push {k}"
        d = dg.digest()
        assert np.uint64(d["hash"]) == f[f"{tag}_digest_hash"][k], f"push {k}: the grid is not bit-identical to the fixture's"
        assert [d["cells_valid"], d["tiles_initialized"]] == list(f[f"{tag}_digest_counts"][k])
        so = f[f"{tag}_sums_{k}"]
        assert abs(d["sum_tsd"] - so[0]) <= 1e-9 * max(1.0, abs(so[0])) and abs(d["sum_weight"] - so[1]) <= 1e-9 * max(1.0, abs(so[1]))
    cd, nd, md, cnt = dg.raycast(f[f"{tag}_rc_pose"], f[f"{tag}_rc_rays_world"], H.MIN_RANGE, H.MAX_RANGE)
    assert np.array_equal(md, f[f"{tag}_rc_mask"])
    sel = np.repeat(md.astype(bool), 2)
    assert np.max(np.abs(cd[sel] - f[f"{tag}_rc_coords"][sel])) <= 1e-9 and np.max(np.abs(nd[sel] - f[f"{tag}_rc_normals"][sel])) <= 1e-9
    if scene == "comb":
        return
    data, mask = ingest(f[f"{tag}_icp_scan"])
    p = dg.icp_params(30, 0.4, 0.02)
    rf = dg.localize(f[f"{tag}_rc_pose"], f[f"{tag}_rc_rays_world"], f[f"{tag}_rc_rays_local"], data, mask, H.MIN_RANGE, H.MAX_RANGE, p)
    want = f[f"{tag}_icp_counts"]
    assert [rf.pairs, rf.iterations, rf.state, rf.n_model, rf.n_scene] == list(want)
    d, a = H.pose_delta(f[f"{tag}_icp_T"], rf.T)
    assert d <= 1e-9 and a <= 1e-9
    tr, to = dg.icp_trace(rf.iterations), f[f"{tag}_icp_trace"]
    assert np.array_equal(tr[:, 0], to[:, 0]) and np.array_equal(tr[:, 3], to[:, 3]), "pairs / state per iteration"
    assert np.max(np.abs(tr[:, 1] - to[:, 1])) <= 1e-9 and np.max(np.abs(tr[:, 2] - to[:, 2])) <= 1e-15, "rms / threshold per iteration"
    assert np.max(np.abs(tr[:, 4:8] - to[:, 4:8])) <= 1e-9, "Tlast per iteration"
    # closed loop through the facade
    node = facade.SlamNode(facade.node_params(gc, geo), synchronous=True)
    rows = f[f"{tag}_traj_rows"]
    for k in range(len(rows)):
        node.laser(f[f"{tag}_traj_scans"][k], geo.angle_min, geo.angle_increment)
        rep = node.report()
        d, a = H.pose_delta(rows[k][:9].reshape(3, 3), rep["pose"])
        assert d <= TOL_POSE_M and a <= TOL_POSE_RAD, f"scan {k}"
        if k > 0:
            assert [int(x) for x in rows[k][9:14]] == [rep["pairs"], rep["iterations"], rep["icp_state"], rep["valid_model"], rep["valid_scene"]], f"scan {k}"
        assert (int(rows[k][14]), int(rows[k][15])) == (rep["pushed"], rep["reg_error"])
    dd = node.grid().digest()
    tg = f[f"{tag}_traj_grid"]
    assert [dd["cells_valid"], dd["tiles_initialized"]] == [int(tg[0]), int(tg[1])]
    assert abs(dd["sum_tsd"] - tg[2]) <= 1e-6 * max(1.0, abs(tg[2])) and abs(dd["sum_weight"] - tg[3]) <= 1e-6 * max(1.0, abs(tg[3]))
    node.close()


def test_push_on_a_device_with_few_compute_units():
    """k_push_update's ticket queue handed tiles out through 32 heads owned by blockIdx % 32 -- with a resident grid of
    fewer than 32 workgroups (a small compute partition, a CU mask) whole residue classes of the UPDATE list would never be
    processed, silently.  TSD_DEBUG_N_CUS=4 makes the context size its launches for four compute units (20 workgroups): the
    cfg 2 golden digests (bit-exact 64-bit hashes of the whole grid) must still come out."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSD_DEBUG_N_CUS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_parity.py", "-k",
                        "golden_baseline_config_fixture and cfg2 or push_comb_scene"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
