"""Asynchronous mapping (tsd_sensor_set_async_mapping): the fused scan's push runs beside the NEXT registration, so the next scan's ray
cast sees the grid one push behind -- one of the interleavings of the reference's ThreadLocalize / ThreadMapping pair
(ThreadMapping.cpp:51-76: queuePush returns at once), and a deterministic one.  The oracle side is that order on the oracle's
primitives: ray cast, THEN the previous scan's push, registration, gates."""
import numpy as np
import pytest

from ohm_tsd_slam_amd import synth
from tests import helpers as H
from tests.slam_driver import HipSlamFused, slam_kwargs

pytestmark = pytest.mark.gpu


class OracleOnePushBehind:
    def __init__(self, o, **kw):
        self.o, self.kw = o, kw
        self.g = o.Grid(kw["map_size_log2"], kw["cell_size"], kw["truncation_radius"] * kw["cell_size"])
        self.initialized = False
        self.pending = None

    def flush(self):
        if self.pending is not None:
            kw = self.kw
            pose, d2, m2 = self.pending
            self.g.push(pose, d2, m2, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])
            self.pending = None

    def process_scan(self, ranges_f32, draws=None):
        import math
        o, kw, g = self.o, self.kw, self.g
        r = np.array(ranges_f32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        B, res, phi_min = kw["beams"], kw["angle_increment"], kw["angle_min"]
        out = dict(pushed=0, reg_error=0, pairs=0)
        if not self.initialized:
            W = (1 << kw["map_size_log2"]) * kw["cell_size"]
            phi = kw["local_offset_yaw"]
            sx = W * 0.5 + kw["x_offset"] + kw["local_offset_x"]
            sy = W * 0.5 + kw["y_offset"] + kw["local_offset_y"]
            Tinit = np.array([[math.cos(phi), -math.sin(phi), sx], [math.sin(phi), math.cos(phi), sy], [0, 0, 1.0]])
            self.rays_local = o.rays_local(B, phi_min, res)
            self.rays = o.rays_transform(Tinit, self.rays_local)
            self.ray_norm = 1.0
            self.pose = o.mat3_mul(np.eye(3), Tinit)
            data, mask = o.ingest_f32(r, kw["max_range"], res)
            g.free_footprint([sx + kw["footprint_x_offset"], sy], kw["footprint_width"], kw["footprint_height"])
            g.push(self.pose, data, mask, res, phi_min, kw["max_range"], kw["min_range"], kw["low_refl_range"])     # initPush: synchronous
            self.initialized = True
            self.last_pose = None
            out.update(pose=self.pose.copy(), pushed=1)
            return out
        data, mask = o.ingest_f32(r, kw["max_range"], res)
        if self.last_pose is None:
            self.last_pose = self.pose.copy()
        self.rays = o.rays_rescale(self.rays, kw["cell_size"], self.ray_norm)
        self.ray_norm = kw["cell_size"]
        co, no, mo, cnt = g.raycast(self.pose, self.rays, kw["min_range"], kw["max_range"])      # the grid WITHOUT the previous scan's push
        self.flush()                                                                             # ... which lands now
        if cnt == 0:
            out.update(pose=self.pose.copy(), no_model=1)
            return out
        scene, ms, _ = o.scene_from_scan(self.rays_local, data, mask)
        M = co.reshape(-1, 2)[mo.astype(bool)]
        S = scene.reshape(-1, 2)[ms.astype(bool)]
        if draws is None:
            icp = o.icp(M, S, self.pose, kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"], (0.0, g.max_x, 0.0, g.max_x), nn_mode=1)
        else:
            # registration_mode 3: the pre-registration scores on the grid WITH the previous scan's push (it landed above)
            m = o.tsdpdf_match(g, self.pose, co, mo, scene, ms, kw["trials"], kw["size_control_set"], kw["zrand"],
                               np.radians(kw["ransac_phi_max"]), res, *draws)
            out["pre"] = (m["candidates"], m["idx"], m["i"])
            icp = o.icp_init(M, S, self.pose, kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"],
                             (0.0, g.max_x, 0.0, g.max_x), m["T"], nn_mode=1)
        T = icp["T"]
        out.update(pairs=icp["pairs"], T=T)
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
            d2, m2 = o.ingest_f64(data, kw["max_range"], res)
            self.pending = (self.pose.copy(), d2, m2)
            out["pushed"] = 1
        return out


@pytest.mark.parametrize("cfg,n", [("cfg1", 30), ("cfg2", 14)])
def test_async_mapping_is_exactly_one_push_behind(oracle, cfg, n):
    gc, geo, scene = synth.CONFIGS[cfg]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    kw = slam_kwargs(gc, geo)
    hs = HipSlamFused(oracle, **kw)
    hstrict = HipSlamFused(oracle, **kw)              # the default order beside it: the two must not be the same thing
    oa = OracleOnePushBehind(oracle, **kw)
    strict_differs = False
    for k in range(n):
        rh = hs.process_scan(scans[k])
        if k == 0:
            hs.sensor.set_async_mapping(True)
        rs = hstrict.process_scan(scans[k])
        strict_differs |= rs["pairs"] != rh["pairs"] or H.pose_delta(rs["pose"], rh["pose"])[0] > 1e-7
        ro = oa.process_scan(scans[k])
        assert (rh["pushed"], rh["reg_error"], rh["pairs"]) == (ro["pushed"], ro["reg_error"], ro["pairs"]), (k, rh, ro)
        d, a = H.pose_delta(ro["pose"], rh["pose"])
        assert d <= 1e-9 and a <= 1e-9, (k, d, a)
    assert strict_differs, "one push behind gave the strict order's results: the mode did nothing"
    oa.flush()
    hs.grid.sync()
    H.assert_grids_equal(oa.g.dump(), hs.grid.download_tiles(), 1e-9)      # (the poses agree to 1e-13, the cells pushed from them likewise)
    # switching back is allowed between scans and is the strict order again
    hs.sensor.set_async_mapping(False)
    rs = hs.process_scan(scans[-1])
    assert rs["pairs"] > 0


class HipFusedAhead(HipSlamFused):
    """HipSlamFused with the NEXT scan staged ahead of the collect (tsd_scan_submit / _stage / _collect): what the facade does when
    the next LaserScan is known (ThreadLocalize::announceNext)."""

    def _ingest(self, ranges_f32):
        o, kw = self.o, self.kw
        r = np.array(ranges_f32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        data, mask = o.ingest_f32(r, kw["max_range"], kw["angle_increment"])
        _, mask_push = o.ingest_f64(data, kw["max_range"], kw["angle_increment"])
        return data, mask, mask_push

    def process_scan(self, ranges_f32, nxt_f32=None, staged=False):
        if not self.initialized:
            return super().process_scan(ranges_f32)
        nxt = self._ingest(nxt_f32) if nxt_f32 is not None else None
        cur = (None, None, None) if staged else self._ingest(ranges_f32)
        sr = self.sensor.scan_ahead(*cur, self.params, self.gates, nxt=nxt)
        self.pose = np.array(sr.pose[:]).reshape(3, 3)
        return dict(pose=self.pose.copy(), pushed=int(sr.pushed), reg_error=int(sr.reg_error), pairs=int(sr.icp.pairs))


@pytest.mark.parametrize("stall_us", [0, 3000])
def test_async_mapping_staged_ahead_under_a_lagging_push_stream(oracle, stall_us):
    """with the mapper asynchronous, scan k+3 is staged into the scan / table buffers of scan k while nothing the
    host has seen proves that push k -- on the push stream, beside registration k+1 -- has finished reading them.  Every push is held
    back by 3 ms here (tsd_debug_stall_push_stream; a registration takes 0.15 ms), so an unguarded staging WOULD overwrite the
    buffers under the push; the per-buffer push event keeps the order, and results and grid equal the one-push-behind order."""
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    n = 24
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    kw = slam_kwargs(gc, geo)
    hs = HipFusedAhead(oracle, **kw)
    oa = OracleOnePushBehind(oracle, **kw)
    for k in range(n):
        nxt = scans[k + 1] if (k >= 1 and k + 1 < n) else None
        rh = hs.process_scan(scans[k], nxt, staged=(k >= 2))
        if k == 0:
            hs.sensor.set_async_mapping(True)
            hs.grid._check(hs.grid.lib.tsd_debug_stall_push_stream(hs.grid.h, stall_us), "tsd_debug_stall_push_stream")
        ro = oa.process_scan(scans[k])
        assert (rh["pushed"], rh["reg_error"], rh["pairs"]) == (ro["pushed"], ro["reg_error"], ro["pairs"]), (k, rh, ro)
        d, a = H.pose_delta(ro["pose"], rh["pose"])
        assert d <= 1e-9 and a <= 1e-9, (k, d, a)
    oa.flush()
    hs.grid.sync()
    H.assert_grids_equal(oa.g.dump(), hs.grid.download_tiles(), 1e-9)
    hs.grid._check(hs.grid.lib.tsd_debug_stall_push_stream(hs.grid.h, 0), "tsd_debug_stall_push_stream")


def test_async_mapping_with_one_hardware_queue():
    """All streams of the process on ONE in-order hardware queue (GPU_MAX_HW_QUEUES=1): the two hand-offs between the context's stream
    and the push stream must neither deadlock nor change a result -- the cfg 1 case above in a child process (the variable is read
    when the runtime comes up)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_async_mapping.py", "-k", "one_push_behind and cfg1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


class HipFusedMode3(HipSlamFused):
    """HipSlamFused with the TSD_PDF pre-registration armed ahead of every scan (tsd_scan_preregister), given draws."""

    def process_scan(self, ranges_f32, draws=None):
        if not self.initialized or draws is None:
            return super().process_scan(ranges_f32)
        o, kw = self.o, self.kw
        r = np.array(ranges_f32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        res = kw["angle_increment"]
        data, mask = o.ingest_f32(r, kw["max_range"], res)
        _, mask_push = o.ingest_f64(data, kw["max_range"], res)
        sc, ms, _ = o.scene_from_scan(self.rays_local, data, mask)
        self.sensor.preregister(sc, ms, kw["trials"], kw["size_control_set"], kw["zrand"], np.radians(kw["ransac_phi_max"]), res, *draws)
        sr = self.sensor.scan(data, mask, mask_push, self.params, self.gates)
        pr = self.sensor.preregistration_result()
        self.pose = np.array(sr.pose[:]).reshape(3, 3)
        return dict(pose=self.pose.copy(), pushed=int(sr.pushed), reg_error=int(sr.reg_error), pairs=int(sr.icp.pairs),
                    pre=(pr["candidates"], pr["idx"], pr["i"]))


def test_async_mapping_with_the_pre_registration(oracle):
    """registration_mode 3 with the mapper asynchronous: ray cast (one push behind), the previous push lands -- beside the normals and
    the list building, which do not read the grid --, THEN the scoring, the arg-max and the registration."""
    gc, geo, scene = synth.CONFIGS["cfg1"]
    world = synth.World(scene, gc)
    n = 16
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    kw = slam_kwargs(gc, geo, trials=40, size_control_set=120)
    hs = HipFusedMode3(oracle, **kw)
    oa = OracleOnePushBehind(oracle, **kw)
    rng = np.random.default_rng(77)
    for k in range(n):
        draws = None if k == 0 else tuple(rng.integers(0, 2 ** 31 - 1, m) for m in (geo.beams, kw["size_control_set"], kw["trials"]))
        rh = hs.process_scan(scans[k], draws)
        if k == 0:
            hs.sensor.set_async_mapping(True)
        ro = oa.process_scan(scans[k], draws)
        if k > 0:
            assert rh["pre"] == ro["pre"], (k, rh["pre"], ro["pre"])
        assert (rh["pushed"], rh["reg_error"], rh["pairs"]) == (ro["pushed"], ro["reg_error"], ro["pairs"]), (k, rh, ro)
        d, a = H.pose_delta(ro["pose"], rh["pose"])
        assert d <= 1e-9 and a <= 1e-9, (k, d, a)
    oa.flush()
    hs.grid.sync()
    H.assert_grids_equal(oa.g.dump(), hs.grid.download_tiles(), 1e-9)
