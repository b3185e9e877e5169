"""Test-side closed loop around the HIP kernels: ThreadLocalize::init / eventLoop with the device calls
of include/tsd_hip.h in place of RayCastPolar2D / Icp / TsdGrid::push, and the O(beams) host steps
(scan ingest, ray maps, gates) taken from the oracle so that both sides see identical inputs.
The product's own host side is the C++ facade (ohm_tsd_slam_amd/csrc/host), tested in test_facade*.py.
"""
import math

import numpy as np

from ohm_tsd_slam_amd import capi


def slam_kwargs(gc, geo, **over):
    kw = dict(
        map_size_log2=gc.map_size_log2, cell_size=gc.cell_size, truncation_radius=gc.truncation_radius,
        beams=geo.beams, angle_min=geo.angle_min, angle_increment=geo.angle_increment,
        max_range=30.0, min_range=0.001, low_refl_range=2.0,
        x_offset=0.0, y_offset=0.0, local_offset_x=0.37, local_offset_y=-0.21, local_offset_yaw=0.1,
        footprint_width=1.0, footprint_height=1.0, footprint_x_offset=0.28,
        laser_min_range=0.26, icp_iterations=30, dist_filter_max=0.4, dist_filter_min=0.02,
        reg_trs_max=1.0, reg_sin_rot_max=0.5, nn_mode=0, threads=1,
        registration_mode=0, trials=100, size_control_set=140, zrand=0.25, ransac_phi_max=30.0,
    )
    kw.update(over)
    return kw


class HipSlam:
    def __init__(self, oracle, fused=True, storage="f64", **kw):
        self.o = oracle
        self.kw = kw
        self.fused = fused
        self.grid = capi.TsdGridDevice(kw["map_size_log2"], kw["cell_size"], kw["truncation_radius"] * kw["cell_size"],
                                       storage=storage)
        self.initialized = False
        self.params = self.grid.icp_params(kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"])

    def process_scan(self, ranges_f32, T_override=None):
        """T_override: the registration result the STATE is advanced with (the oracle's, in the re-synced long runs:
        out["T"] stays the device's own result, which is what gets compared); None = the device's."""
        o, kw, g = self.o, self.kw, self.grid
        r = np.array(ranges_f32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        B, res, phi_min = kw["beams"], kw["angle_increment"], kw["angle_min"]
        out = dict(pushed=0, reg_error=0, pairs=0, valid_model=0, no_model=0)
        if not self.initialized:
            W = g.cells * g.cell_size
            phi = kw["local_offset_yaw"]
            sx = W * 0.5 + kw["x_offset"] + kw["local_offset_x"]
            sy = W * 0.5 + kw["y_offset"] + kw["local_offset_y"]
            Tinit = np.array([[math.cos(phi), -math.sin(phi), sx], [math.sin(phi), math.cos(phi), sy], [0, 0, 1.0]])
            self.rays_local = o.rays_local(B, phi_min, res)
            self.rays = o.rays_transform(Tinit, self.rays_local)
            self.ray_norm = 1.0
            self.pose = o.mat3_mul(np.eye(3), Tinit)
            self.data, self.mask = o.ingest_f32(r, kw["max_range"], res)
            g.free_footprint([sx + kw["footprint_x_offset"], sy], kw["footprint_width"], kw["footprint_height"])
            g.push(self.pose, self.data, self.mask, res, phi_min, kw["max_range"], kw["min_range"], kw["low_refl_range"])
            self.initialized = True
            self.last_pose = None
            out.update(pose=self.pose.copy(), pushed=1)
            return out
        self.data, self.mask = o.ingest_f32(r, kw["max_range"], res)
        if self.last_pose is None:
            self.last_pose = self.pose.copy()
        self.rays = o.rays_rescale(self.rays, g.cell_size, self.ray_norm)
        self.ray_norm = g.cell_size
        if self.fused:
            res_icp = g.localize(self.pose, self.rays, self.rays_local, self.data, self.mask, kw["min_range"],
                                 kw["max_range"], self.params)
            out["valid_model"] = res_icp.n_model
            if res_icp.n_model == 0:
                out.update(pose=self.pose.copy(), no_model=1)
                return out
        else:
            co, no, mo, cnt = g.raycast(self.pose, self.rays, kw["min_range"], kw["max_range"])
            out["valid_model"] = cnt
            if cnt == 0:
                out.update(pose=self.pose.copy(), no_model=1)
                return out
            scene, ms, _ = o.scene_from_scan(self.rays_local, self.data, self.mask)
            M = co.reshape(-1, 2)[mo.astype(bool)]
            S = scene.reshape(-1, 2)[ms.astype(bool)]
            res_icp = g.icp(M, S, self.pose, self.params)
        T = res_icp.T
        out.update(pairs=res_icp.pairs, T=T, iterations=res_icp.iterations, icp_state=res_icp.state, rms=res_icp.rms)
        if T_override is not None:
            T = np.array(T_override, dtype=np.float64).reshape(3, 3)
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
            d2, m2 = o.ingest_f64(self.data, kw["max_range"], res)
            g.push(self.pose, d2, m2, res, phi_min, kw["max_range"], kw["min_range"], kw["low_refl_range"],
                   want_stats=False)
            out["pushed"] = 1
        return out


class HipSlamFused(HipSlam):
    """The same closed loop through tsd_scan: sensor state, gates, Sensor::transform and the push
    decision live on the device; the host only ingests the scan and reads the result."""

    def process_scan(self, ranges_f32):
        o, kw, g = self.o, self.kw, self.grid
        if not self.initialized:
            out = super().process_scan(ranges_f32)
            self.rays = o.rays_rescale(self.rays, g.cell_size, self.ray_norm)
            self.ray_norm = g.cell_size
            self.sensor = capi.TsdSensorDevice(g, kw["beams"], kw["angle_increment"], kw["angle_min"], kw["max_range"],
                                               kw["min_range"], kw["low_refl_range"])
            self.sensor.set_pose(self.pose, self.rays, self.rays_local)
            self.gates = capi.GateParams(kw["reg_trs_max"], kw["reg_sin_rot_max"], 0.05, 0.03)
            return out
        r = np.array(ranges_f32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        res = kw["angle_increment"]
        data, mask = o.ingest_f32(r, kw["max_range"], res)
        _, mask_push = o.ingest_f64(data, kw["max_range"], res)
        sr = self.sensor.scan(data, mask, mask_push, self.params, self.gates)
        self.pose = np.array(sr.pose[:]).reshape(3, 3)
        return dict(pose=self.pose.copy(), pushed=int(sr.pushed), reg_error=int(sr.reg_error), pairs=int(sr.icp.pairs),
                    valid_model=int(sr.icp.n_model), no_model=int(sr.no_model), T=np.array(sr.icp.T[:]).reshape(3, 3))


class PrimitiveLoop:
    """ThreadLocalize::eventLoop (ThreadLocalize.cpp:310-409) on ONE backend's primitives -- the oracle's (ray cast, Icp::iterate
    with its per-iteration trace, push) or the unfused C ABI calls (tsd_raycast / tsd_icp + tsd_icp_trace / tsd_push) -- keeping the
    inputs of the last registration on the host.  T_override advances the state with somebody else's registration result
    (re-synced runs); out["T"] stays the backend's own."""

    def __init__(self, oracle, kw, grid, is_hip, threads=1):
        self.o, self.kw, self.g, self.hip, self.threads = oracle, kw, grid, is_hip, threads
        O = oracle
        self.cs = grid.cell_size
        W = (1 << kw["map_size_log2"]) * self.cs
        phi = kw["local_offset_yaw"]
        self.sx = W * 0.5 + kw["x_offset"] + kw["local_offset_x"]
        self.sy = W * 0.5 + kw["y_offset"] + kw["local_offset_y"]
        Tinit = np.array([[math.cos(phi), -math.sin(phi), self.sx], [math.sin(phi), math.cos(phi), self.sy], [0, 0, 1.0]])
        self.rays_local = O.rays_local(kw["beams"], kw["angle_min"], kw["angle_increment"])
        self.rays = O.rays_transform(Tinit, self.rays_local)
        self.pose = O.mat3_mul(np.eye(3), Tinit)
        self.last_pose = None
        self.first = True
        if is_hip:
            self.params = grid.icp_params(kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"])
            self.bounds = (grid.min_x, grid.max_x, grid.min_y, grid.max_y)
        else:
            self.bounds = (0.0, grid.max_x, 0.0, grid.max_x)        # TsdGrid's _minX/_maxX/_minY/_maxY (tsd_oracle.c: ora_grid_create)

    def push(self, pose, data, mask):
        kw = self.kw
        a = (pose, data, mask, kw["angle_increment"], kw["angle_min"], kw["max_range"], kw["min_range"], kw["low_refl_range"])
        return self.g.push(*a) if self.hip else self.g.push(*a, threads=self.threads)

    def icp(self, M, S, pose):
        kw, O = self.kw, self.o
        if self.hip:
            r = self.g.icp(M, S, pose, self.params)
            tr = self.g.icp_trace(r.iterations)
            return dict(T=r.T, pairs=r.pairs, iterations=r.iterations, state=r.state, rms=r.rms, trace=tr)
        return O.icp(M, S, pose, kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"], self.bounds,
                     nn_mode=kw.get("nn_mode", 0), trace=True)

    def scan(self, r32, T_override=None):
        kw, O = self.kw, self.o
        r = np.array(r32, dtype=np.float32)
        r[r < kw["laser_min_range"]] = 0.0
        data, mask = O.ingest_f32(r, kw["max_range"], kw["angle_increment"])
        out = dict(pushed=0, reg_error=0, no_model=0, stats=None, trace=None, hit=None)
        if self.first:
            self.first = False
            self.g.free_footprint([self.sx + kw["footprint_x_offset"], self.sy], kw["footprint_width"], kw["footprint_height"])
            out["stats"] = self.push(self.pose, data, mask)
            self.rays = O.rays_rescale(self.rays, self.cs, 1.0)
            out.update(pose=self.pose.copy(), pushed=1)
            return out
        if self.last_pose is None:
            self.last_pose = self.pose.copy()
        a = (self.pose, self.rays, kw["min_range"], kw["max_range"])
        co, no, mo, cnt = self.g.raycast(*a) if self.hip else self.g.raycast(*a, threads=self.threads)
        out["hit"] = mo.copy()
        if cnt == 0:
            out.update(pose=self.pose.copy(), no_model=1)
            return out
        scene, ms, _ = O.scene_from_scan(self.rays_local, data, mask)
        M = co.reshape(-1, 2)[mo.astype(bool)].copy()
        S = scene.reshape(-1, 2)[ms.astype(bool)].copy()
        self.inputs = (M, S, self.pose.copy())
        res = self.icp(M, S, self.pose)
        out.update(trace=res["trace"], pairs=res["pairs"], iterations=res["iterations"], state=res["state"], T=np.array(res["T"]))
        T = res["T"] if T_override is None else np.array(T_override, dtype=np.float64).reshape(3, 3)
        if O.lib().ora_is_registration_error(O.d(O.f64(T).reshape(9)), kw["reg_trs_max"], kw["reg_sin_rot_max"]):
            out.update(pose=self.pose.copy(), reg_error=1)
            return out
        self.rays = O.rays_transform(T, self.rays)
        self.pose = O.mat3_mul(self.pose, T)
        out["pose"] = self.pose.copy()
        lp, cp = O.f64(self.last_pose).reshape(9), O.f64(self.pose).reshape(9)
        if O.lib().ora_is_pose_change_significant(O.d(lp), O.d(cp)):
            self.last_pose = self.pose.copy()
            d2, m2 = O.ingest_f64(data, kw["max_range"], kw["angle_increment"])
            out["stats"] = self.push(self.pose, d2, m2)
            out["pushed"] = 1
        return out
