"""GPU tests of SURVEY 8(f) row N3: the TSD_PDF pre-registration (registration_mode 3; TSD_PDFMatching.cpp:31-294) and
Icp::iterate with its result as Tinit (ThreadLocalize.cpp:557-581).  The reference's three rand() streams are inputs
here (fixed draw arrays), so the HIP result is compared with the oracle's restatement for identical draws: the same
winning (model, scene) pair, candidates and point counts exact, T / probability to rounding (the device's cos / sin
and the product of 140-360 factors differ from libm's in the last bits).  PARITY UNPINNED like the other GSL-bound rows.
"""
import ctypes as C
import math

import numpy as np
import pytest

from ohm_tsd_slam_amd import capi, facade, synth
from tests import helpers as H
from tests.slam_driver import slam_kwargs

pytestmark = pytest.mark.gpu


def _map_and_scan(oracle, gc, geo, scene, k_pose, k_scan, dyaw=0.0, pushes=4):
    world = synth.World(scene, gc)
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(pushes):
        pose, (x, y, yaw) = H.sensor_pose(world, k)
        data, mask = oracle.ingest_f32(world.scan(x, y, yaw, geo), H.MAX_RANGE, geo.angle_increment)
        og.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)
    pose, _ = H.sensor_pose(world, k_pose)                        # where the robot believes it is
    _, (x, y, yaw) = H.sensor_pose(world, k_scan)                 # where the scan is really taken
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    co, no, mo, cnt = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    data, mask = oracle.ingest_f32(world.scan(x, y, yaw + dyaw, geo), H.MAX_RANGE, geo.angle_increment)
    sc, ms, ns = oracle.scene_from_scan(rl, data, mask)
    Ttrue = np.linalg.inv(pose) @ synth.pose_matrix(x, y, yaw + dyaw)
    return og, dg, pose, co, mo, sc, ms, Ttrue


@pytest.mark.parametrize("cfg,scene,ctrl,zrand,seed", [
    ("cfg1", "room", 140, 0.25, 1), ("cfg2", "pillars", 140, 0.25, 2), ("cfg2", "pillars", 360, 0.05, 3),   # defaults / shipped YAML
])
def test_tsdpdf_match_matches_oracle(oracle, cfg, scene, ctrl, zrand, seed):
    gc, geo, _ = synth.CONFIGS[cfg]
    og, dg, pose, co, mo, sc, ms, Ttrue = _map_and_scan(oracle, gc, geo, scene, 3, 8, dyaw=0.05)
    rng = np.random.default_rng(seed)
    trials = 100
    ds, dc, dt = (rng.integers(0, 2 ** 31 - 1, n) for n in (geo.beams, ctrl, trials))
    phi_max = math.radians(30.0)
    ro = oracle.tsdpdf_match(og, pose, co, mo, sc, ms, trials, ctrl, zrand, phi_max, geo.angle_increment, ds, dc, dt)
    rh = dg.tsdpdf_match(pose, co, mo, sc, ms, trials, ctrl, zrand, phi_max, geo.angle_increment, ds, dc, dt)
    assert ro["rc"] == 0 and ro["candidates"] > 200
    assert (rh["candidates"], rh["idx"], rh["i"]) == (ro["candidates"], ro["idx"], ro["i"]), (ro, rh)
    assert abs(rh["prob"] - ro["prob"]) <= 1e-9 * ro["prob"]
    assert np.max(np.abs(rh["T"] - ro["T"])) <= 1e-12
    # and it is a sensible pre-registration: close to the true motion between the believed and the real pose
    d, a = H.pose_delta(Ttrue, rh["T"])
    assert d < 0.15 and a < 0.05, (d, a)


# (zrand = 1.0: every factor is exactly 1.0, every candidate ties -- the first in the reference's serial trial / i order has to win, in the
# unfused call (list in serial order) and in the fused one (list in whatever order the waves arrived, the order carried as a key))
@pytest.mark.parametrize("cfg,scene,ctrl,zrand,seed", [("cfg2", "pillars", 140, 0.25, 2), ("cfg1", "room", 360, 0.05, 7), ("cfg2", "pillars", 140, 1.0, 11)])
def test_fused_preregistration_matches_the_unfused_calls(oracle, cfg, scene, ctrl, zrand, seed):
    """tsd_scan_preregister + tsd_scan (everything between the ray cast and the registration on the device: normals, sample lists,
    control set, trial picks, candidates, scoring, arg-max, Tinit handed over on the device) == tsd_tsdpdf_match on the ray cast's
    outputs + tsd_localize with its T as t_init, and == the oracle."""
    gc, geo, _ = synth.CONFIGS[cfg]
    og, dg, pose, co, mo, sc, ms, Ttrue = _map_and_scan(oracle, gc, geo, scene, 3, 8, dyaw=0.05)
    world = synth.World(scene, gc)
    _, (x, y, yaw) = H.sensor_pose(world, 8)
    data, mask = oracle.ingest_f32(world.scan(x, y, yaw + 0.05, geo), H.MAX_RANGE, geo.angle_increment)
    _, mask_push = oracle.ingest_f64(data, H.MAX_RANGE, geo.angle_increment)
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    rng = np.random.default_rng(seed)
    trials = 100
    ds, dc, dt = (rng.integers(0, 2 ** 31 - 1, n) for n in (geo.beams, ctrl, trials))
    phi_max = math.radians(30.0)
    # the unfused reference: pre-registration on the host-visible ray cast, then the registration with Tinit
    ch, nh, mh, cnt = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    rm = dg.tsdpdf_match(pose, ch, mh, sc, ms, trials, ctrl, zrand, phi_max, geo.angle_increment, ds, dc, dt)
    ro = oracle.tsdpdf_match(og, pose, co, mo, sc, ms, trials, ctrl, zrand, phi_max, geo.angle_increment, ds, dc, dt)
    assert (rm["candidates"], rm["idx"], rm["i"]) == (ro["candidates"], ro["idx"], ro["i"])
    p = dg.icp_params(30, 0.4, 0.02, t_init=rm["T"])
    rf = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, p)
    # fused: one scan of a device sensor
    sensor = capi.TsdSensorDevice(dg, geo.beams, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
    sensor.set_pose(pose, rw, rl)
    gates = capi.GateParams(10.0, 1.0, 1e9, 2.0)            # (wide registration gate, no push: this test is about the registration's inputs)
    sensor.preregister(sc, ms, trials, ctrl, zrand, phi_max, geo.angle_increment, ds, dc, dt)
    sr = sensor.scan(data, mask, mask_push, dg.icp_params(30, 0.4, 0.02), gates)
    pr = sensor.preregistration_result()
    assert (pr["candidates"], pr["idx"], pr["i"]) == (rm["candidates"], rm["idx"], rm["i"]), (pr, rm)
    assert (pr["valid_model"], pr["valid_scene"], pr["control_points"]) == (rm["valid_model"], rm["valid_scene"], rm["control"])
    assert np.max(np.abs(pr["T"] - rm["T"])) <= 1e-12 and abs(pr["prob"] - rm["prob"]) <= 1e-9 * rm["prob"]
    assert (sr.icp.pairs, sr.icp.iterations, sr.icp.state, sr.icp.n_model, sr.icp.n_scene) == (rf.pairs, rf.iterations, rf.state, rf.n_model, rf.n_scene)
    d, a = H.pose_delta(np.asarray(rf.T), np.array(sr.icp.T[:]).reshape(3, 3))
    assert d <= 1e-11 and a <= 1e-11
    # a scan without an armed pre-registration is the plain scan again (one-shot)
    sensor.set_pose(pose, rw, rl)
    s0 = sensor.scan(data, mask, mask_push, dg.icp_params(30, 0.4, 0.02), gates)
    r0 = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, dg.icp_params(30, 0.4, 0.02))
    assert (s0.icp.pairs, s0.icp.iterations) == (r0.pairs, r0.iterations)
    sensor.close()


def test_preregistration_armed_ahead_of_the_collect(oracle):
    """tsd_scan_preregister while the previous scan is in flight (what the facade does for a staged scan: scene points, draws and the
    inputs' copy while the device registers) == arming it between the two scans: same winner, same registration, same pose."""
    gc, geo, scene = synth.CONFIGS["cfg2"]
    og, dg, pose, co, mo, sc, ms, Ttrue = _map_and_scan(oracle, gc, geo, scene, 3, 8, dyaw=0.05)
    world = synth.World(scene, gc)
    rl, rw = H.world_rays(oracle, geo, pose, gc.cell_size)
    rng = np.random.default_rng(31)
    phi_max, res = math.radians(30.0), geo.angle_increment
    scans = []
    for k, dy in ((8, 0.05), (9, 0.02)):
        _, (x, y, yaw) = H.sensor_pose(world, k)
        data, mask = oracle.ingest_f32(world.scan(x, y, yaw + dy, geo), H.MAX_RANGE, res)
        _, mask_push = oracle.ingest_f64(data, H.MAX_RANGE, res)
        scn, msk, _ = oracle.scene_from_scan(rl, data, mask)
        draws = tuple(rng.integers(0, 2 ** 31 - 1, n) for n in (geo.beams, 140, 100))
        scans.append((data, mask, mask_push, (scn, msk, 100, 140, 0.25, phi_max, res) + draws))
    gates = capi.GateParams(1.0, 0.5, 0.05, 0.03)            # (the node's gates: the first scan pushes, the second sees that push)
    outs = []
    for ahead in (False, True):
        g2 = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
        g2.upload_tiles(*dg.download_tiles())
        sensor = capi.TsdSensorDevice(g2, geo.beams, res, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
        sensor.set_pose(pose, rw, rl)
        p = g2.icp_params(30, 0.4, 0.02)
        sensor.preregister(*scans[0][3])
        if not ahead:
            r1 = sensor.scan(scans[0][0], scans[0][1], scans[0][2], p, gates)
            pr1 = sensor.preregistration_result()
            sensor.preregister(*scans[1][3])
            r2 = sensor.scan(scans[1][0], scans[1][1], scans[1][2], p, gates)
        else:
            r1 = sensor.scan_ahead(scans[0][0], scans[0][1], scans[0][2], p, gates, nxt=scans[1][:3], nxt_pre=scans[1][3])
            pr1 = sensor.preregistration_result()          # (the first scan's, although the second is armed already)
            r2 = sensor.scan_ahead(None, None, None, p, gates)
        pr2 = sensor.preregistration_result()
        outs.append((pr1, pr2, r1, r2))
        sensor.close()
    (a1, a2, ra1, ra2), (b1, b2, rb1, rb2) = outs
    for x, y in ((a1, b1), (a2, b2)):
        assert (x["candidates"], x["idx"], x["i"]) == (y["candidates"], y["idx"], y["i"]) and np.array_equal(x["T"], y["T"])
    for x, y in ((ra1, rb1), (ra2, rb2)):
        assert (x.icp.pairs, x.icp.iterations, x.pushed) == (y.icp.pairs, y.icp.iterations, y.pushed)
        assert np.array_equal(np.array(x.pose[:]), np.array(y.pose[:]))
    assert ra1.pushed == 1 and a2["candidates"] > 0


def test_hip_matches_the_committed_tsdpdf_fixture():
    """No oracle at run time: the committed vectors of tests/golden/oracle_tsdpdf.npz (make_oracle_fixtures.py) against the HIP path,
    through tsd_tsdpdf_match + tsd_localize(t_init) and through the fused scan (tsd_scan_preregister)."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_tsdpdf.npz"))
    from oracle import pyoracle as O        # (only the scan ingest: part of the inputs' preparation, not of what is checked)
    res, phi = float(f["angle_increment"]), float(f["angle_min"])
    dg = capi.TsdGridDevice(int(f["map_size_log2"]), float(f["cell_size"]), float(f["max_trunc"]))
    for k in range(len(f["push_poses"])):
        data, mask = O.ingest_f32(f["push_scans"][k], H.MAX_RANGE, res)
        dg.push(f["push_poses"][k], data, mask, res, phi, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)
    pose, rw, rl = f["pose"], f["rays_world"], f["rays_local"]
    ch, nh, mh, cnt = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    sel = np.repeat(mh.astype(bool), 2)
    assert np.array_equal(mh, f["rc_mask"]) and np.max(np.abs(ch[sel] - f["rc_coords"][sel])) <= 1e-9
    args = (int(f["trials"]), int(f["size_control_set"]), float(f["zrand"]), float(f["phi_max"]), res, f["draws_sub"], f["draws_ctrl"], f["draws_trials"])
    m = dg.tsdpdf_match(pose, ch, mh, f["scene"], f["scene_mask"], *args)
    assert [m["candidates"], m["idx"], m["i"]] == list(f["match_counts"])
    assert np.max(np.abs(m["T"] - f["match_T"])) <= 1e-12 and abs(m["prob"] - float(f["match_prob"])) <= 1e-9 * float(f["match_prob"])
    data, mask = O.ingest_f32(f["scan"], H.MAX_RANGE, res)
    r = dg.localize(pose, rw, rl, data, mask, H.MIN_RANGE, H.MAX_RANGE, dg.icp_params(30, 0.4, 0.02, t_init=m["T"]))
    assert [r.pairs, r.iterations, r.state, r.n_model, r.n_scene] == list(f["icp_counts"])
    d, a = H.pose_delta(f["icp_T"], np.asarray(r.T))
    assert d <= 1e-9 and a <= 1e-9 and abs(r.rms - float(f["icp_rms"])) <= 1e-9
    # the fused scan
    _, mask_push = O.ingest_f64(data, H.MAX_RANGE, res)
    sensor = capi.TsdSensorDevice(dg, int(f["beams"]), res, phi, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
    sensor.set_pose(pose, rw, rl)
    sensor.preregister(f["scene"], f["scene_mask"], *args)
    sr = sensor.scan(data, mask, mask_push, dg.icp_params(30, 0.4, 0.02), capi.GateParams(10.0, 1.0, 1e9, 2.0))
    pr = sensor.preregistration_result()
    assert [pr["candidates"], pr["idx"], pr["i"]] == list(f["match_counts"]) and np.max(np.abs(pr["T"] - f["match_T"])) <= 1e-12
    assert [sr.icp.pairs, sr.icp.iterations, sr.icp.state, sr.icp.n_model, sr.icp.n_scene] == list(f["icp_counts"])
    d, a = H.pose_delta(f["icp_T"], np.array(sr.icp.T[:]).reshape(3, 3))
    assert d <= 1e-9 and a <= 1e-9
    sensor.close()


def test_tsdpdf_degenerate_inputs(oracle):
    """too few valid points -> identity (TSD_PDFMatching.cpp:53-57, :129-139); all-false masks; tiny control set"""
    gc, geo, _ = synth.CONFIGS["cfg1"]
    og, dg, pose, co, mo, sc, ms, _ = _map_and_scan(oracle, gc, geo, "room", 2, 3)
    rng = np.random.default_rng(5)
    ds, dc, dt = (rng.integers(0, 2 ** 31 - 1, n) for n in (geo.beams, 140, 100))
    z = np.zeros_like(mo)
    for (mm, msk) in ((z, ms), (mo, z)):
        rh = dg.tsdpdf_match(pose, co, mm, sc, msk, 100, 140, 0.25, 0.5, geo.angle_increment, ds, dc, dt)
        ro = oracle.tsdpdf_match(og, pose, co, mm, sc, msk, 100, 140, 0.25, 0.5, geo.angle_increment, ds, dc, dt)
        assert np.array_equal(rh["T"], np.eye(3)) and np.array_equal(ro["T"], np.eye(3)) and rh["idx"] == -1
    rh = dg.tsdpdf_match(pose, co, mo, sc, ms, 5, 3, 0.25, 0.5, geo.angle_increment, ds, dc, dt)
    ro = oracle.tsdpdf_match(og, pose, co, mo, sc, ms, 5, 3, 0.25, 0.5, geo.angle_increment, ds, dc, dt)
    assert (rh["candidates"], rh["idx"], rh["i"]) == (ro["candidates"], ro["idx"], ro["i"])
    assert np.max(np.abs(rh["T"] - ro["T"])) <= 1e-12


@pytest.mark.parametrize("iters", [30, 11])
def test_icp_with_t_init_matches_oracle(oracle, iters):
    gc, geo, scene = synth.CONFIGS["cfg2"]
    og, dg, pose, co, mo, sc, ms, Ttrue = _map_and_scan(oracle, gc, geo, scene, 3, 8, dyaw=0.05)
    M = co.reshape(-1, 2)[mo.astype(bool)]
    S = sc.reshape(-1, 2)[ms.astype(bool)]
    Tinit = synth.pose_matrix(0.28, -0.03, 0.09)                   # a rough pre-registration
    ro = oracle.icp_init(M, S, pose, iters, 0.4, 0.02, (0.0, og.max_x, 0.0, og.max_x), Tinit, nn_mode=1)
    rh = dg.icp(M, S, pose, dg.icp_params(iters, 0.4, 0.02, t_init=Tinit))
    assert (rh.pairs, rh.iterations, rh.state) == (ro["pairs"], ro["iterations"], ro["state"])
    d, a = H.pose_delta(ro["T"], rh.T)
    assert d <= 1e-9 and a <= 1e-9 and abs(rh.rms - ro["rms"]) <= 1e-9
    d, a = H.pose_delta(Ttrue, rh.T)
    assert d < 0.05 and a < 0.02                                   # Tfinal includes Tinit (Icp.cpp:485)
    # without Tinit the same call starts from the identity
    r0 = dg.icp(M, S, pose, dg.icp_params(iters, 0.4, 0.02))
    o0 = oracle.icp(M, S, pose, iters, 0.4, 0.02, (0.0, og.max_x, 0.0, og.max_x), nn_mode=1)
    assert (r0.pairs, r0.iterations) == (o0["pairs"], o0["iterations"])


def _libc_draws(seed, n_sub, n_ctrl, n_trials):
    """what obvious::TSD_PDFMatching::match of the facade draws for `tsdpdf_seed` >= 0: srand(seed + call), then rand()"""
    libc = C.CDLL(None)
    libc.srand(C.c_uint(seed))
    return ([libc.rand() for _ in range(n_sub)], [libc.rand() for _ in range(n_ctrl)], [libc.rand() for _ in range(n_trials)])


@pytest.mark.parametrize("cfg,n,trials", [("cfg1", 12, 100), ("cfg2", 10, 100), ("cfg1", 8, 600)])
def test_facade_registration_mode_3_matches_oracle(oracle, cfg, n, trials):
    """config/single-laser.yaml's mode: ThreadLocalize with the TSD_PDF pre-registration in front of the ICP, the whole
    closed loop against the oracle's SLAM loop in the same mode, both fed the same rand() draws.  The facade runs the
    pre-registration inside the fused scan (tsd_scan_preregister); 600 trials are more than its device-side list building holds,
    so that case exercises the fall-back to the reference's call structure (tsd_raycast -> tsd_tsdpdf_match -> tsd_localize)."""
    gc, geo, scene = synth.CONFIGS[cfg]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
    ctrl, zrand, phimax, seed = 140, 0.25, 30.0, 4711
    so = oracle.Slam(**slam_kwargs(gc, geo_msg, registration_mode=3, trials=trials, size_control_set=ctrl, zrand=zrand,
                                   ransac_phi_max=phimax))
    params = facade.node_params(gc, geo)
    params.update({"registration_mode": 3, "trials": trials, "sizeControlSet": ctrl, "zrand": zrand, "ransac_phi_max": phimax,
                   "tsdpdf_seed": seed})
    node = facade.SlamNode(params, synchronous=True)
    pushes = 0
    for k in range(n):
        if k > 0:
            so.set_draws(*_libc_draws(seed + (k - 1), geo.beams, ctrl, trials))
        ro = so.process_scan(scans[k])
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
        rh = node.report()
        d, a = H.pose_delta(np.array(ro.pose[:]).reshape(3, 3), rh["pose"])
        assert d <= 1e-4 and a <= 1e-4, f"scan {k}: {d} m {a} rad"
        if k > 0:
            assert (ro.pairs, ro.iterations, ro.icp_state) == (rh["pairs"], rh["iterations"], rh["icp_state"]), f"scan {k}"
            assert (ro.valid_model, ro.valid_scene) == (rh["valid_model"], rh["valid_scene"])
            assert bool(ro.pushed) == bool(rh["pushed"]) and bool(ro.reg_error) == bool(rh["reg_error"])
        pushes += rh["pushed"]
    assert pushes >= n // 2
    H.assert_grids_equal(so.grid.dump(), node.grid().download_tiles(), 1e-5)
    e = math.hypot(rh["pose"][0, 2] - poses[-1, 0], rh["pose"][1, 2] - poses[-1, 1])
    assert e < 0.1, f"tracking error {e} m"
    node.close()


def test_facade_two_robots_registration_mode_3_through_the_batched_path(oracle):
    """registration_mode 3 for the reference's multi-robot mode (every ThreadLocalize runs `case TSD` itself,
    /root/reference/src/ThreadLocalize.cpp:557-567; SlamNode.cpp:101-122: one shared grid): the facade's dispatcher batches the robots'
    scans and the pre-registration runs fused inside the batch (tsd_scan_preregister + tsd_batch_*), no longer through the unfused host
    calls.  Fed in turn (a batch of one each time = the serial loop) against the oracle's shared-grid loop in mode 3, each robot with
    its own seeded rand() sequence."""
    gc, geo, scene = synth.CONFIGS["cfg1"]
    n = 10
    offs = [(0.37, -0.21, 0.1), (-0.7, 0.4, 0.0)]
    geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
    ctrl, zrand, phimax, trials, seeds = 120, 0.25, 30.0, 60, (4711, 815)
    scans = []
    for (ox, oy, yaw) in offs:
        w = synth.World(scene, gc, start_xy=[0.5 * gc.width + ox, 0.5 * gc.width + oy])
        scans.append(synth.scans_for(w, geo, synth.trajectory(w, n, yaw0=yaw)))
    mode3 = dict(registration_mode=3, trials=trials, size_control_set=ctrl, zrand=zrand, ransac_phi_max=phimax)
    so0 = oracle.Slam(**slam_kwargs(gc, geo_msg, local_offset_x=offs[0][0], local_offset_y=offs[0][1], local_offset_yaw=offs[0][2], **mode3))
    so1 = oracle.Slam(shared_with=so0, **slam_kwargs(gc, geo_msg, local_offset_x=offs[1][0], local_offset_y=offs[1][1],
                                                       local_offset_yaw=offs[1][2], **mode3))
    sos = (so0, so1)
    params = facade.node_params(gc, geo, robot_nbr=2)
    params.update({"robot_0/name": "georg", "robot_1/name": "simon", "trials": trials, "sizeControlSet": ctrl, "zrand": zrand})
    for name, (ox, oy, yaw), seed in zip(("georg", "simon"), offs, seeds):
        params.update({f"tsd_slam/{name}/local_offset_x": ox, f"tsd_slam/{name}/local_offset_y": oy, f"tsd_slam/{name}/local_offset_yaw": yaw,
                       f"{name}/dist_filter_max": 0.4, f"{name}/dist_filter_min": 0.02, f"{name}/icp_iterations": 30,
                       f"{name}/registration_mode": 3, f"{name}/ransac_phi_max": phimax, f"{name}/tsdpdf_seed": seed})
    params = {k: v for k, v in params.items() if not k.startswith("tsd_slam/local_offset")}
    node = facade.SlamNode(params, synchronous=True)
    b0 = node.batch_stats()
    for k in range(n):
        for r in (0, 1):
            if k > 0:
                sos[r].set_draws(*_libc_draws(seeds[r] + (k - 1), geo.beams, ctrl, trials))
            ro = sos[r].process_scan(scans[r][k])
            node.laser(scans[r][k], geo.angle_min, geo.angle_increment, robot=r)
            rh = node.report(r)
            d, a = H.pose_delta(np.array(ro.pose[:]).reshape(3, 3), rh["pose"])
            assert d <= 1e-4 and a <= 1e-4, f"scan {k} robot {r}: {d} m {a} rad"
            if k > 0:
                assert (ro.pairs, ro.iterations, ro.icp_state) == (rh["pairs"], rh["iterations"], rh["icp_state"]), f"scan {k} robot {r}"
                assert bool(ro.pushed) == bool(rh["pushed"]) and bool(ro.reg_error) == bool(rh["reg_error"])
    b1 = node.batch_stats()
    assert b1[1] - b0[1] == 2 * (n - 1), "the robots' scans did not go through the batched dispatcher"
    H.assert_grids_equal(so0.grid.dump(), node.grid().download_tiles(), 1e-5)
    node.close()
