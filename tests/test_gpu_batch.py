"""Batched multi-robot scans (tsd_batch_*) against the oracle.

Semantics of a batch (include/tsd_hip.h): every ray cast of the batch reads the grid as it was before any push of the batch,
the pushes are applied in the order of the batch.  The oracle side below runs exactly that with the oracle's primitives
(ray cast, scene, Icp::iterate, gates, push: the statements of ThreadLocalize::eventLoop, ThreadLocalize.cpp:305-411) on ONE
oracle grid shared by the robots (SlamNode.cpp:101-122).
"""
import math

import numpy as np
import pytest

from ohm_tsd_slam_amd import capi, synth
from tests import helpers as H
from tests.slam_driver import slam_kwargs

pytestmark = pytest.mark.gpu

OFFSETS = [(0.37, -0.21, 0.1), (-0.7, 0.4, 0.0), (0.9, 0.8, -0.2), (-0.3, -0.9, 0.3)]


class Robot:
    """Host state of one robot (pose bookkeeping of ThreadLocalize) for BOTH sides: the oracle side advances it with the
    oracle's results, the HIP side keeps the same state on the device (tsd_sensor)."""

    def __init__(self, o, gc, geo, off, kw):
        self.o, self.kw = o, kw
        W = gc.width
        self.phi = off[2]
        self.sx, self.sy = W * 0.5 + off[0], W * 0.5 + off[1]
        Tinit = np.array([[math.cos(self.phi), -math.sin(self.phi), self.sx], [math.sin(self.phi), math.cos(self.phi), self.sy], [0, 0, 1.0]])
        self.rays_local = o.rays_local(geo.beams, kw["angle_min"], kw["angle_increment"])
        self.rays = o.rays_transform(Tinit, self.rays_local)
        self.pose = o.mat3_mul(np.eye(3), Tinit)
        self.last_pose = None
        self.cs = gc.cell_size

    def ingest(self, r32):
        kw = self.kw
        r = np.array(r32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        data, mask = self.o.ingest_f32(r, kw["max_range"], kw["angle_increment"])
        _, mask_push = self.o.ingest_f64(data, kw["max_range"], kw["angle_increment"])
        return data, mask, mask_push

    def init_both(self, og, dg, r32):
        kw = self.kw
        data, mask, _ = self.ingest(r32)
        for g in (og, dg):
            g.free_footprint([self.sx + kw["footprint_x_offset"], self.sy], kw["footprint_width"], kw["footprint_height"])
            g.push(self.pose, data, mask, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])
        self.rays = self.o.rays_rescale(self.rays, self.cs, 1.0)

    # ---- oracle side, split like a batch: localise against the grid as it is, apply the push later
    def localise(self, og, data, mask, bounds, draws=None):
        o, kw = self.o, self.kw
        out = dict(pushed=0, reg_error=0, pairs=0, valid_model=0, no_model=0, iterations=0, state=0)
        if self.last_pose is None:
            self.last_pose = self.pose.copy()
        co, no, mo, cnt = og.raycast(self.pose, self.rays, kw["min_range"], kw["max_range"])
        out["valid_model"] = cnt
        self._push = None
        if cnt == 0:
            out.update(pose=self.pose.copy(), no_model=1)
            return out
        scene, ms, _ = o.scene_from_scan(self.rays_local, data, mask)
        M = co.reshape(-1, 2)[mo.astype(bool)]
        S = scene.reshape(-1, 2)[ms.astype(bool)]
        if draws is None:
            r = o.icp(M, S, self.pose, kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"], bounds)
        else:
            # registration_mode 3 (ThreadLocalize.cpp:557-567): TSD_PDFMatching::match on the beam-indexed sets, its result is Tinit
            m = o.tsdpdf_match(og, self.pose, co, mo, scene, ms, kw["trials"], kw["size_control_set"], kw["zrand"],
                               np.radians(kw["ransac_phi_max"]), kw["angle_increment"], *draws)
            out["pre"] = (m["candidates"], m["idx"], m["i"])
            self.scene_last = (scene, ms)
            r = o.icp_init(M, S, self.pose, kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"], bounds, m["T"])
        T = r["T"]
        out.update(pairs=r["pairs"], iterations=r["iterations"], state=r["state"])
        Tf = o.f64(T).reshape(9)
        if o.lib().ora_is_registration_error(o.d(Tf), kw["reg_trs_max"], kw["reg_sin_rot_max"]):
            out.update(pose=self.pose.copy(), reg_error=1)
            return out
        self.rays = o.rays_transform(T, self.rays)
        self.pose = o.mat3_mul(self.pose, T)
        out["pose"] = self.pose.copy()
        lp, cp = o.f64(self.last_pose).reshape(9), o.f64(self.pose).reshape(9)
        if o.lib().ora_is_pose_change_significant(o.d(lp), o.d(cp)):
            self.last_pose = self.pose.copy()
            d2, m2 = o.ingest_f64(data, kw["max_range"], kw["angle_increment"])
            self._push = (self.pose.copy(), d2, m2)
            out["pushed"] = 1
        return out

    def apply_push(self, og):
        kw = self.kw
        if self._push is not None:
            pose, d2, m2 = self._push
            og.push(pose, d2, m2, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])


def _setup(oracle, cfg, n_robots, n_scans):
    gc, geo, scene = synth.CONFIGS[cfg]
    geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
    kw = slam_kwargs(gc, geo_msg)
    og = oracle.Grid(gc.map_size_log2, gc.cell_size, gc.truncation_radius * gc.cell_size)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.truncation_radius * gc.cell_size)
    robots, scans, sensors = [], [], []
    for off in OFFSETS[:n_robots]:
        w = synth.World(scene, gc, start_xy=[0.5 * gc.width + off[0], 0.5 * gc.width + off[1]])
        scans.append(synth.scans_for(w, geo, synth.trajectory(w, n_scans, yaw0=off[2])))
        robots.append(Robot(oracle, gc, geo, off, kw))
    for rb, sc in zip(robots, scans):
        rb.init_both(og, dg, sc[0])
    for rb in robots:
        s = capi.TsdSensorDevice(dg, geo.beams, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])
        s.set_pose(rb.pose, rb.rays, rb.rays_local)
        sensors.append(s)
    params = dg.icp_params(kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"])
    gates = capi.GateParams(kw["reg_trs_max"], kw["reg_sin_rot_max"], 0.05, 0.03)
    return gc, geo, kw, og, dg, robots, scans, sensors, params, gates


def _compare(k, i, ro, sr):
    assert int(sr.icp.n_model) == ro["valid_model"], f"round {k} robot {i}: ray-cast hits"
    assert (int(sr.no_model), int(sr.reg_error), int(sr.pushed)) == (ro["no_model"], ro["reg_error"], ro["pushed"]), f"round {k} robot {i}: gates"
    if not ro["no_model"]:
        assert (int(sr.icp.pairs), int(sr.icp.iterations), int(sr.icp.state)) == (ro["pairs"], ro["iterations"], ro["state"]), f"round {k} robot {i}"
    d, a = H.pose_delta(ro["pose"], np.array(sr.pose[:]).reshape(3, 3))
    assert d <= 1e-4 and a <= 1e-4, f"round {k} robot {i}: pose differs by {d} m, {a} rad"
    return d


@pytest.mark.parametrize("cfg,n_robots,n_scans", [("cfg1", 3, 10), ("cfg2", 4, 6)])
def test_batch_matches_oracle(oracle, cfg, n_robots, n_scans):
    """One batch per round with all robots: counts / gates exact, pose within the bar, grids cell for cell."""
    gc, geo, kw, og, dg, robots, scans, sensors, params, gates = _setup(oracle, cfg, n_robots, n_scans)
    batch = capi.TsdBatch(dg, n_robots)
    bounds = (dg.min_x, dg.max_x, dg.min_y, dg.max_y)
    worst = 0.0
    for k in range(1, n_scans):
        ing = [rb.ingest(sc[k]) for rb, sc in zip(robots, scans)]
        ros = [rb.localise(og, d_, m_, bounds) for rb, (d_, m_, _) in zip(robots, ing)]
        for rb in robots:
            rb.apply_push(og)
        batch.begin(sensors, [x[0] for x in ing], [x[1] for x in ing], [x[2] for x in ing], params, gates)
        res = batch.results()
        for i, (ro, sr) in enumerate(zip(ros, res)):
            worst = max(worst, _compare(k, i, ro, sr))
    H.assert_grids_equal(og.dump(), dg.download_tiles(), 1e-5)
    batch.close()
    for s in sensors:
        s.close()


@pytest.mark.parametrize("cfg,n_robots,n_scans", [("cfg1", 4, 12), ("cfg2", 6, 8)])
def test_batch_push_in_one_pass_equals_serial_pushes(oracle, cfg, n_robots, n_scans):
    """tsd_batch_push applies the robots' pushes in ONE pass per tile (csrc/push_multi.hip: every tile read and written once, the
    robots' updates in the batch's order inside it, one halo pass at the end).  With tsd_debug_set_push_multi(0) it enqueues one push
    per robot like the reference's mapper (ThreadMapping.cpp:43-76).  Two device grids driven identically, one in each mode: every
    scan result, the accumulated push statistics and the WHOLE grid -- tile states, cells, halos: the canonical dump and its 64-bit
    digest -- are bit-identical, and both equal the oracle's serial pushes within the suite's bar."""
    runs = []
    for multi in (True, False):
        gc, geo, kw, og, dg, robots, scans, sensors, params, gates = _setup(oracle, cfg, n_robots, n_scans)
        dg.set_push_multi(multi)
        batch = capi.TsdBatch(dg, n_robots)
        bounds = (dg.min_x, dg.max_x, dg.min_y, dg.max_y)
        dg.push_stats_total(reset=True)
        results = []
        for k in range(1, n_scans):
            ing = [rb.ingest(sc[k]) for rb, sc in zip(robots, scans)]
            ros = [rb.localise(og, d_, m_, bounds) for rb, (d_, m_, _) in zip(robots, ing)]
            for rb in robots:
                rb.apply_push(og)
            batch.begin(sensors, [x[0] for x in ing], [x[1] for x in ing], [x[2] for x in ing], params, gates)
            res = batch.results()
            for i, (ro, sr) in enumerate(zip(ros, res)):
                _compare(k, i, ro, sr)
            results.append([(np.array(sr.pose[:]).tobytes(), int(sr.pushed), int(sr.icp.pairs)) for sr in res])
        H.assert_grids_equal(og.dump(), dg.download_tiles(), 1e-5)
        runs.append((results, dg.push_stats_total(), dg.digest(), dg.download_tiles()))
        batch.close()
        for s in sensors:
            s.close()
        dg.close()
    (ra, sa, da, ta), (rb_, sb, db, tb) = runs
    assert ra == rb_, "scan results differ between the one-pass and the serial pushes"
    assert sa == sb, f"push statistics differ: {sa} / {sb}"
    assert sa[1] >= (n_scans - 1) * n_robots // 2
    assert da == db, f"grid digests differ: {da} / {db}"
    for x, y in zip(ta, tb):
        assert np.array_equal(x, y, equal_nan=True)


def test_batch_with_the_pre_registration_matches_oracle(oracle):
    """registration_mode 3 for several robots on one grid (the reference runs `case TSD` in every robot's thread,
    /root/reference/src/ThreadLocalize.cpp:557-567): every robot's pre-registration is armed (tsd_scan_preregister: scene points + the
    three rand() streams) and runs INSIDE the batch -- normals, lists, scoring, arg-max on the device behind the batch's ray casts,
    against the grid as it is before any push of the batch -- and its result is the Tinit of that robot's registration.  Oracle side:
    the same order on the oracle's primitives.  Winner, candidate count, pairs, iterations, state, gates exact; poses within the bar."""
    n_robots, n_scans = 3, 8
    gc, geo, kw, og, dg, robots, scans, sensors, params, gates = _setup(oracle, "cfg1", n_robots, n_scans)
    kw.update(trials=40, size_control_set=120)
    batch = capi.TsdBatch(dg, n_robots)
    bounds = (dg.min_x, dg.max_x, dg.min_y, dg.max_y)
    rng = np.random.default_rng(4242)
    phi_max, res = np.radians(kw["ransac_phi_max"]), kw["angle_increment"]
    for k in range(1, n_scans):
        ing = [rb.ingest(sc[k]) for rb, sc in zip(robots, scans)]
        draws = [tuple(rng.integers(0, 2 ** 31 - 1, m) for m in (geo.beams, kw["size_control_set"], kw["trials"])) for _ in robots]
        # robot 1 sits every third round out of the pre-registration: armed and unarmed scans may share a batch
        armed = [not (i == 1 and k % 3 == 0) for i in range(n_robots)]
        ros = [rb.localise(og, d_, m_, bounds, dr if a else None) for rb, (d_, m_, _), dr, a in zip(robots, ing, draws, armed)]
        for rb in robots:
            rb.apply_push(og)
        for rb, s, (d_, m_, _), dr, a in zip(robots, sensors, ing, draws, armed):
            if a:
                sc, ms, _n = oracle.scene_from_scan(rb.rays_local, d_, m_)
                s.preregister(sc, ms, kw["trials"], kw["size_control_set"], kw["zrand"], phi_max, res, *dr)
        batch.begin(sensors, [x[0] for x in ing], [x[1] for x in ing], [x[2] for x in ing], params, gates)
        res_h = batch.results()
        for i, (ro, sr) in enumerate(zip(ros, res_h)):
            _compare(k, i, ro, sr)
            if armed[i] and not ro["no_model"]:
                pr = sensors[i].preregistration_result()
                assert (pr["candidates"], pr["idx"], pr["i"]) == ro["pre"], f"round {k} robot {i}: pre-registration {pr} vs {ro['pre']}"
    H.assert_grids_equal(og.dump(), dg.download_tiles(), 1e-5)
    batch.close()
    for s in sensors:
        s.close()


def test_two_slots_with_pushes_enqueued_ahead(oracle):
    """Two batch slots used the way the facade's dispatcher uses them: both begin before either push is enqueued, the pushes
    are enqueued BEFORE the results are collected (gated on the device).  Both slots' ray casts then read the grid before any
    push of the round and the pushes follow in slot order: the same as one batch of all robots."""
    n_robots, n_scans = 4, 8
    gc, geo, kw, og, dg, robots, scans, sensors, params, gates = _setup(oracle, "cfg1", n_robots, n_scans)
    slots = [capi.TsdBatch(dg, 2), capi.TsdBatch(dg, 2)]
    groups = [[0, 1], [2, 3]]
    bounds = (dg.min_x, dg.max_x, dg.min_y, dg.max_y)
    for k in range(1, n_scans):
        ing = [rb.ingest(sc[k]) for rb, sc in zip(robots, scans)]
        ros = [rb.localise(og, d_, m_, bounds) for rb, (d_, m_, _) in zip(robots, ing)]
        for rb in robots:
            rb.apply_push(og)
        for slot, grp in zip(slots, groups):
            slot.begin([sensors[i] for i in grp], [ing[i][0] for i in grp], [ing[i][1] for i in grp], [ing[i][2] for i in grp], params, gates)
        for slot in slots:
            slot.push()
        for slot, grp in zip(slots, groups):
            for i, sr in zip(grp, slot.results()):
                _compare(k, i, ros[i], sr)
    H.assert_grids_equal(og.dump(), dg.download_tiles(), 1e-5)
    for slot in slots:
        slot.close()


def test_batch_argument_errors(oracle):
    gc, geo, kw, og, dg, robots, scans, sensors, params, gates = _setup(oracle, "cfg1", 2, 2)
    batch = capi.TsdBatch(dg, 1)
    ing = [rb.ingest(sc[1]) for rb, sc in zip(robots, scans)]
    with pytest.raises(capi.TsdError):      # more scans than the slot holds
        batch.begin(sensors, [x[0] for x in ing], [x[1] for x in ing], None, params, gates)
    b2 = capi.TsdBatch(dg, 2)
    with pytest.raises(capi.TsdError):      # the same sensor twice
        b2.begin([sensors[0], sensors[0]], [ing[0][0]] * 2, [ing[0][1]] * 2, None, params, gates)
    b2.begin(sensors, [x[0] for x in ing], [x[1] for x in ing], None, params, gates)
    with pytest.raises(capi.TsdError):      # slot not collected yet
        b2.begin(sensors, [x[0] for x in ing], [x[1] for x in ing], None, params, gates)
    with pytest.raises(capi.TsdError):      # sensor still in flight in the other slot
        batch.begin([sensors[0]], [ing[0][0]], [ing[0][1]], None, params, gates)
    assert len(b2.results()) == 2
    b2.close(); batch.close()


import os      # noqa: E402
import subprocess  # noqa: E402
import sys     # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env_extra,must_be_clean", [
    ({"TSD_BATCH_VERBOSE": "1"}, True),                                   # the box as it is: the probe decides, every result right
    ({"GPU_MAX_HW_QUEUES": "1", "TSD_BATCH_VERBOSE": "1"}, False),        # every stream on ONE in-order hardware queue
    ({"GPU_MAX_HW_QUEUES": "1", "TSD_BATCH_FORCE_DEVICE_WAIT": "1", "TSD_BATCH_POLL_BOUND": "30000", "TSD_BATCH_VERBOSE": "1"}, False),
    ({"TSD_BATCH_EVENT_WAIT": "1"}, True),
])
def test_batched_path_never_delivers_stale_registrations(env_extra, must_be_clean):
    """The hand-offs inside a batch are waits on the device only where a start-up probe has shown that the streams involved run
    side by side; and where such a wait still runs out (forced here: one hardware queue, device waits insisted on, a short
    bound), the caller gets a non-zero code -- correct results or an error, never a pose from stale ray-cast output with rc 0
    (tests/batch_serial_check.py exits 9 for that)."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("TSD_BATCH_") and k != "GPU_MAX_HW_QUEUES"}
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-m", "tests.batch_serial_check"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "batch_serial_check ok" in out.stdout
    if must_be_clean:
        assert ", 0 calls returned an error" in out.stdout, out.stdout[-2000:]
    print(out.stdout[-600:], out.stderr[-600:])
