"""GPU tests of the C++ facade (ThreadLocalize / ThreadMapping over the C ABI): parity of the whole
loop with the oracle, and the reference's threading contract (first scan initialises synchronously,
newest scan wins, mapping queue drains, clean shutdown)."""
import math
import time

import numpy as np
import pytest

from ohm_tsd_slam_amd import facade, synth
from tests import helpers as H
from tests.slam_driver import slam_kwargs

pytestmark = pytest.mark.gpu


def run_pair(oracle, gc, geo, scene, n, fused=True, **over):
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    # a sensor_msgs/LaserScan carries angle_min / angle_increment as float32 (ThreadLocalize.cpp:487-488
    # widens them again): give the oracle the same rounded geometry the facade receives
    geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
    so = oracle.Slam(**slam_kwargs(gc, geo_msg, **over))
    node = facade.SlamNode(facade.node_params(gc, geo, **{k: v for k, v in over.items() if k in ("icp_iterations",)}),
                           synchronous=True, fused=fused)
    return world, poses, scans, so, node


@pytest.mark.parametrize("cfg,n,fused", [("cfg1", 15, True), ("cfg1", 15, False), ("cfg2", 8, True), ("cfg2", 8, False)])
def test_facade_sync_loop_matches_oracle(oracle, cfg, n, fused):
    """fused: tsd_scan (gates, Sensor::transform and the push decided on the device in stream order);
    unfused: tsd_localize + host gates + tsd_push, the reference's call structure."""
    gc, geo, scene = synth.CONFIGS[cfg]
    world, poses, scans, so, node = run_pair(oracle, gc, geo, scene, n, fused=fused)
    pushes = 0
    for k in range(n):
        ro = so.process_scan(scans[k])
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
        rh = node.report()
        Po = np.array(ro.pose[:]).reshape(3, 3)
        d, a = H.pose_delta(Po, rh["pose"])
        assert d <= 1e-4 and a <= 1e-4, f"scan {k}: {d} m {a} rad"
        assert bool(ro.pushed) == bool(rh["pushed"]) and bool(ro.reg_error) == bool(rh["reg_error"])
        if k > 0:
            assert (ro.pairs, ro.iterations, ro.icp_state) == (rh["pairs"], rh["iterations"], rh["icp_state"])
            assert (ro.valid_model, ro.valid_scene) == (rh["valid_model"], rh["valid_scene"])
            assert abs(ro.rms - rh["rms"]) <= 1e-9
        pushes += rh["pushed"]
    assert pushes >= n // 2
    H.assert_grids_equal(so.grid.dump(), node.grid().download_tiles(), 1e-5)
    # PoseStamped on <node>/estimated_pose: position = pose + grid offset, yaw quaternion
    msg = node.pose_msg()
    assert msg["topic"] == "tsd_slam/estimated_pose" and msg["count"] == n - 1
    W = gc.cells * gc.cell_size
    P = node.report()["pose"]
    assert msg["position"][0] == P[0, 2] - 0.5 * W and msg["position"][1] == P[1, 2] - 0.5 * W
    yaw = 2.0 * math.atan2(msg["orientation_xyzw"][2], msg["orientation_xyzw"][3])
    assert abs(yaw - math.atan2(P[1, 0], P[0, 0])) < 1e-9
    node.close()


def _hom(t, q):
    x, y, z, w = q
    M = np.eye(4)
    M[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                 [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                 [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    M[:3, 3] = t
    return M


@pytest.mark.parametrize("fused", [True, False])
def test_facade_tf_map_to_odom(fused):
    """ThreadLocalize::sendTransform's tf half (/root/reference/src/ThreadLocalize.cpp:617-661) on the real device loop: no tf tree ->
    the broadcast transform is never written (identity), the PoseStamped is the laser pose; with odom -> base_footprint -> laser heard
    -> map -> odom = laser pose * T(laser <- base_footprint) * T(base_footprint <- odom), against a numpy product."""
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    scans = synth.scans_for(world, geo, synth.trajectory(world, 6))
    node = facade.SlamNode(facade.node_params(gc, geo), synchronous=True, fused=fused)
    for k in range(3):
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
    tf = node.tf_msg()
    assert (tf["frame_id"], tf["child_frame_id"]) == ("map", "odom") and tf["count"] == 2
    assert np.array_equal(tf["translation"], [0, 0, 0]) and np.array_equal(tf["rotation_xyzw"], [0, 0, 0, 1])
    q_ob = np.array([0.0, 0.0, math.sin(0.35), math.cos(0.35)])
    q_bl = np.array([0.0, 0.0, math.sin(-0.05), math.cos(-0.05)])
    node.set_transform("odom", "base_footprint", [1.25, -0.5, 0.0], q_ob)
    node.set_transform("base_footprint", "laser", [0.3, 0.05, 0.2], q_bl)
    for k in range(3, 6):
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
        tf, msg = node.tf_msg(), node.pose_msg()
        laser = _hom(msg["position"], msg["orientation_xyzw"])
        expect = laser @ np.linalg.inv(_hom([0.3, 0.05, 0.2], q_bl)) @ np.linalg.inv(_hom([1.25, -0.5, 0.0], q_ob))
        assert np.max(np.abs(_hom(tf["translation"], tf["rotation_xyzw"]) - expect)) <= 1e-12, f"scan {k}"
    assert tf["count"] == 5
    node.close()


@pytest.mark.parametrize("fused", [True, False])
def test_facade_threads_contract(oracle, fused):
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, 40)
    scans = synth.scans_for(world, geo, poses)
    node = facade.SlamNode(facade.node_params(gc, geo), synchronous=False, fused=fused)
    # first scan: init + synchronous initPush on the caller's thread (ThreadLocalize.cpp:257-267)
    node.laser(scans[0], geo.angle_min, geo.angle_increment)
    assert node.processed() == 1 and node.report()["pushed"] == 1 and node.report()["initialised"] == 1
    init, _ = node.grid().download_tile_state()
    assert init.sum() > 10
    # a burst of scans: the localiser consumes the newest and drops the rest (:319-332)
    for k in range(1, 30):
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
    assert node.wait_idle(20000)
    done = node.processed() - 1
    assert 1 <= done <= 29
    # the last processed scan is the newest one (its stamp), whatever was dropped in between
    assert node.report()["stamp_ns"] == 30 * 25_000_000
    # steady feeding (wait between scans): every scan is processed, map keeps growing
    before = node.processed()
    for k in range(30, 40):
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
        assert node.wait_idle(20000)
    assert node.processed() - before == 10
    t0 = time.time()
    node.close()          # terminateThread + alive() polling + join, as SlamNode::~SlamNode
    assert time.time() - t0 < 5.0


@pytest.mark.parametrize("fused", [True, False])
def test_facade_registration_error_publishes_nan(oracle, fused):
    """A scan that cannot be registered within reg_trs_max -> NaN pose, pose unchanged, no push
    (ThreadLocalize.cpp:381-387, 691-713)."""
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, 3)
    scans = synth.scans_for(world, geo, poses)
    node = facade.SlamNode(facade.node_params(gc, geo, reg_trs_max=1e-4), synchronous=True, fused=fused)
    node.laser(scans[0], geo.angle_min, geo.angle_increment)
    p0 = node.report()["pose"].copy()
    init0, _ = node.grid().download_tile_state()
    node.laser(scans[2], geo.angle_min, geo.angle_increment)
    r = node.report()
    assert r["reg_error"] == 1 and r["pushed"] == 0 and np.array_equal(r["pose"], p0)
    assert np.isnan(node.pose_msg()["position"]).all()
    # the gated-off push left the grid alone, and the next good scan is still registered from p0
    init1, _ = node.grid().download_tile_state()
    assert np.array_equal(init0, init1)
    node.close()


def test_facade_multi_robot_shares_one_grid(oracle):
    """robot_nbr = 2: two ThreadLocalize on ONE TsdGrid + ONE ThreadMapping (SlamNode.cpp:101-122);
    only the first robot's init pushes (ThreadMapping::initialized gate, ThreadLocalize.cpp:506-507)."""
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    params = facade.node_params(gc, geo, robot_nbr=2)
    params.update({"robot_0/name": "georg", "robot_1/name": "simon",
                   "tsd_slam/georg/local_offset_x": 0.37, "tsd_slam/georg/local_offset_y": -0.21,
                   "tsd_slam/georg/local_offset_yaw": 0.1,
                   "tsd_slam/simon/local_offset_x": -0.7, "tsd_slam/simon/local_offset_y": 0.4})
    node = facade.SlamNode(params, synchronous=True)
    s0 = world.scan(world.start[0], world.start[1], 0.1, geo)
    s1 = world.scan(world.cx - 0.7, world.cy + 0.4, 0.0, geo)
    node.laser(s0, geo.angle_min, geo.angle_increment, robot=0)
    node.laser(s1, geo.angle_min, geo.angle_increment, robot=1)
    assert node.report(0)["pushed"] == 1 and node.report(1)["pushed"] == 0
    assert node.pose_msg(0)["topic"] == "tsd_slam/georg/estimated_pose"
    assert node.pose_msg(1)["topic"] == "tsd_slam/simon/estimated_pose"
    for _ in range(3):
        node.laser(s0, geo.angle_min, geo.angle_increment, robot=0)
        node.laser(s1, geo.angle_min, geo.angle_increment, robot=1)
    P1 = node.report(1)["pose"]
    assert math.hypot(P1[0, 2] - (world.cx - 0.7), P1[1, 2] - (world.cy + 0.4)) < 0.1
    node.close()


def _two_robot_setup(oracle, cfg, n):
    gc, geo, scene = synth.CONFIGS[cfg]
    offs = [(0.37, -0.21, 0.1), (-0.7, 0.4, 0.0)]
    geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
    worlds, scans = [], []
    for (ox, oy, yaw) in offs:
        w = synth.World(scene, gc, start_xy=[0.5 * gc.width + ox, 0.5 * gc.width + oy])
        worlds.append(w)
        scans.append(synth.scans_for(w, geo, synth.trajectory(w, n, yaw0=yaw)))
    so0 = oracle.Slam(**slam_kwargs(gc, geo_msg, local_offset_x=offs[0][0], local_offset_y=offs[0][1], local_offset_yaw=offs[0][2]))
    so1 = oracle.Slam(shared_with=so0, **slam_kwargs(gc, geo_msg, local_offset_x=offs[1][0], local_offset_y=offs[1][1],
                                                       local_offset_yaw=offs[1][2]))
    params = facade.node_params(gc, geo, robot_nbr=2)
    params.update({"robot_0/name": "georg", "robot_1/name": "simon"})
    for name, (ox, oy, yaw) in zip(("georg", "simon"), offs):
        params.update({f"tsd_slam/{name}/local_offset_x": ox, f"tsd_slam/{name}/local_offset_y": oy,
                       f"tsd_slam/{name}/local_offset_yaw": yaw,
                       # the ICP keys are per robot in multi-robot mode (ThreadLocalize.cpp:86-88: _robotName + ...)
                       f"{name}/dist_filter_max": 0.4, f"{name}/dist_filter_min": 0.02, f"{name}/icp_iterations": 30})
    params = {k: v for k, v in params.items() if not k.startswith("tsd_slam/local_offset")}
    node = facade.SlamNode(params, synchronous=True)
    return gc, geo, scans, (so0, so1), node


@pytest.mark.parametrize("cfg,n", [("cfg1", 12), ("cfg2", 8)])
def test_facade_two_robots_one_grid_match_oracle(oracle, cfg, n):
    """The shared-grid multi-robot loop against the oracle's: two localisers on ONE grid, scans fed in turn
    (robot 0 scan k, robot 1 scan k, ...).  The facade uses the split scan (tsd_scan_begin / _wait / _finish: ray cast +
    registration on the sensor's own stream, push on the grid's); fed in turn its results are those of the serial loop."""
    gc, geo, scans, sos, node = _two_robot_setup(oracle, cfg, n)
    for k in range(n):
        for r in (0, 1):
            ro = sos[r].process_scan(scans[r][k])
            node.laser(scans[r][k], geo.angle_min, geo.angle_increment, robot=r)
            rh = node.report(r)
            d, a = H.pose_delta(np.array(ro.pose[:]).reshape(3, 3), rh["pose"])
            assert d <= 1e-4 and a <= 1e-4, f"scan {k} robot {r}: {d} m {a} rad"
            assert bool(ro.pushed) == bool(rh["pushed"]), f"scan {k} robot {r}"
            if k > 0:
                assert (ro.pairs, ro.iterations, ro.icp_state) == (rh["pairs"], rh["iterations"], rh["icp_state"]), f"scan {k} robot {r}"
                assert (ro.valid_model, ro.valid_scene) == (rh["valid_model"], rh["valid_scene"])
    H.assert_grids_equal(sos[0].grid.dump(), node.grid().download_tiles(), 1e-5)
    node.close()


def test_facade_two_robots_concurrently(oracle):
    """The same two robots fed from two threads at once (what bench.py --robots does): every scan is processed, both
    robots keep tracking, the grid stays consistent (properties; the interleaving is the scheduler's, like the reference's)."""
    import threading
    n = 40
    gc, geo, scans, sos, node = _two_robot_setup(oracle, "cfg2", n)
    for r in (0, 1):
        node.laser(scans[r][0], geo.angle_min, geo.angle_increment, robot=r)
    errs = []

    def feed(r):
        try:
            for k in range(1, n):
                node.laser(scans[r][k], geo.angle_min, geo.angle_increment, robot=r)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=feed, args=(r,)) for r in (0, 1)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    node.grid().sync()
    for r, (ox, oy) in enumerate(((0.37, -0.21), (-0.7, 0.4))):
        assert node.processed(r) == n
        P = node.report(r)["pose"]
        truth_x = 0.5 * gc.width + ox + 0.06 * (n - 1)
        assert math.hypot(P[0, 2] - truth_x, P[1, 2] - (0.5 * gc.width + oy)) < 0.15, (r, P)
    init, iw, tsd, w = node.grid().download_tiles()
    sel = init.astype(bool)
    assert sel.sum() > 500 and np.all(w[sel] >= 0.0) and np.all(w[sel] <= 32.0)
    t = tsd[sel]; m = ~np.isnan(t)
    assert np.all(t[m] <= 1.0) and np.all(t[m] >= -1.0)
    node.close()


def test_facade_next_scan_staged_ahead_is_bit_identical(oracle):
    """The next scan announced to the localiser (ThreadLocalize::announceNext -> tsd_scan_stage during the current registration,
    what bench.py does) must change nothing: same reports and the same grid, bit for bit, as feeding the scans one by one.
    Every fifth announcement is a WRONG scan (another stamp comes next): the staged scan has to be dropped."""
    gc, geo, scene = synth.CONFIGS["cfg2"]
    world = synth.World(scene, gc)
    n = 30
    scans = synth.scans_for(world, geo, synth.trajectory(world, n))
    out = []
    for ahead in (False, True):
        node = facade.SlamNode(facade.node_params(gc, geo), synchronous=True)
        reps = []
        for k in range(n):
            nxt = None
            if ahead and k + 1 < n:
                nxt = scans[k + 1] if k % 5 != 4 else scans[0]
            if ahead and k % 5 == 4 and k + 1 < n:
                # announce with a stamp that will not come: laser() announces for stamp + 25 ms, the next call then jumps by 50 ms
                node.laser(scans[k], geo.angle_min, geo.angle_increment, ahead=nxt)
                node._stamp += 25_000_000
            else:
                node.laser(scans[k], geo.angle_min, geo.angle_increment, ahead=nxt)
            r = node.report()
            reps.append((r["pose"].copy(), r["pairs"], r["iterations"], r["icp_state"], r["pushed"], r["valid_model"], r["valid_scene"]))
        dig = node.grid().digest()
        out.append((reps, dig))
        node.close()
    (ra, da), (rb, db) = out
    for k, (x, y) in enumerate(zip(ra, rb)):
        assert np.array_equal(x[0], y[0]) and x[1:] == y[1:], f"scan {k}: {x} != {y}"
    assert da == db


def test_facade_four_robots_replayed_through_the_dispatcher(oracle):
    """Four robots on one grid fed by the native replay (one publisher thread per robot, what bench.py --robots does): the
    facade's dispatcher groups their scans into batches (tsd_batch_*).  Every scan is processed, the scans travel in batches
    of more than one, every robot keeps tracking, the grid stays consistent."""
    n, R = 40, 4
    gc, geo, scene = synth.CONFIGS["cfg2"]
    world = synth.World(scene, gc, start_xy=[0.5 * gc.width + 0.37, 0.5 * gc.width - 0.21])
    lanes = synth.free_lanes(world, R, 0.06 * 50, clearance=0.6)
    poses, scans = [], []
    for r in range(R):
        p = synth.trajectory(world, n, leg=50)
        p[:, 1] += lanes[r][1] - world.start[1]
        poses.append(p); scans.append(np.stack(synth.scans_for(world, geo, p)).astype(np.float32))
    params = facade.node_params(gc, geo, robot_nbr=R)
    for r in range(R):
        params.update({f"robot_{r}/name": f"robot{r}", f"tsd_slam/robot{r}/local_offset_x": lanes[r][0] - 0.5 * gc.width,
                       f"tsd_slam/robot{r}/local_offset_y": lanes[r][1] - 0.5 * gc.width, f"tsd_slam/robot{r}/local_offset_yaw": 0.1,
                       f"robot{r}/dist_filter_max": 0.4, f"robot{r}/dist_filter_min": 0.02, f"robot{r}/icp_iterations": 30,
                       f"robot{r}/registration_mode": 0})
    params = {k: v for k, v in params.items() if not k.startswith("tsd_slam/local_offset")}
    node = facade.SlamNode(params, synchronous=True)
    for r in range(R):
        node.laser(scans[r][0], geo.angle_min, geo.angle_increment, robot=r)
    b0 = node.batch_stats()
    node.play(scans, 1, n - 1, geo.angle_min, geo.angle_increment)
    node.grid().sync()
    batches, carried = (x - y for x, y in zip(node.batch_stats(), b0))
    assert carried == R * (n - 1)
    assert batches < carried, "the robots' scans never shared a batch"
    for r in range(R):
        assert node.processed(r) == n
        P = node.report(r)["pose"]
        assert math.hypot(P[0, 2] - poses[r][-1, 0], P[1, 2] - poses[r][-1, 1]) < 0.15, (r, P, poses[r][-1])
    init, iw, tsd, w = node.grid().download_tiles()
    sel = init.astype(bool)
    assert sel.sum() > 500 and np.all(w[sel] >= 0.0) and np.all(w[sel] <= 32.0)
    t = tsd[sel]; m = ~np.isnan(t)
    assert np.all(t[m] <= 1.0) and np.all(t[m] >= -1.0)
    node.close()


@pytest.mark.parametrize("fused", [True, False])
def test_facade_point_to_line_estimator_tracks(oracle, fused):
    """`icp_estimator` = 1 (an addition: the reference node has no such key) runs the loop with
    PointToLine2DEstimator on the ray cast's normals; it has to track the synthetic trajectory like the closed form."""
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    n = 25
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    out = {}
    for est in (0, 1):
        node = facade.SlamNode(facade.node_params(gc, geo, **{"icp_estimator": est}), synchronous=True, fused=fused)
        rms = []
        for k in range(n):
            node.laser(scans[k], geo.angle_min, geo.angle_increment)
            rh = node.report()
            if k > 0:
                assert rh["pairs"] > 100 and not rh["reg_error"]
                rms.append(rh["rms"])
        err = math.hypot(rh["pose"][0, 2] - poses[-1, 0], rh["pose"][1, 2] - poses[-1, 1])
        out[est] = (err, float(np.mean(rms)))
        node.close()
    assert out[0][0] < 0.15 and out[1][0] < 0.15, out
    # the two estimators report different error measures (mean squared distance vs mean |n.(s - m)|)
    assert out[0][1] != out[1][1]
