"""ctypes binding of the CPU oracle (oracle/libtsd_oracle.so) and of the compiled reference pieces
(oracle/_ref/libtsd_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under ohm_tsd_slam_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libtsd_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libtsd_ref.so")
REFERENCE_ROOT = "/root/reference"
TILE_CELLS = 1089

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)
_ip = C.POINTER(C.c_int)


class PushStats(C.Structure):
    _fields_ = [
        ("cells_updated", C.c_int64), ("cells_visited", C.c_int64), ("tiles_total", C.c_int32),
        ("tiles_range_pass", C.c_int32), ("tiles_update", C.c_int32), ("tiles_new", C.c_int32),
        ("tiles_new_from_empty", C.c_int32), ("tiles_emptied_init", C.c_int32),
        ("tiles_emptied_uninit", C.c_int32),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class IcpParams(C.Structure):
    _fields_ = [("iterations", C.c_int), ("dist_filter_max", C.c_double), ("dist_filter_min", C.c_double),
                ("min_x", C.c_double), ("max_x", C.c_double), ("min_y", C.c_double), ("max_y", C.c_double),
                ("nn_mode", C.c_int)]


class IcpResult(C.Structure):
    _fields_ = [("T", C.c_double * 9), ("rms", C.c_double), ("pairs", C.c_int), ("iterations", C.c_int),
                ("state", C.c_int)]


class SlamConfig(C.Structure):
    _fields_ = [
        ("map_size_log2", C.c_int), ("cell_size", C.c_double), ("truncation_radius", C.c_int),
        ("beams", C.c_int), ("angle_min", C.c_double), ("angle_increment", C.c_double),
        ("max_range", C.c_double), ("min_range", C.c_double), ("low_refl_range", C.c_double),
        ("x_offset", C.c_double), ("y_offset", C.c_double), ("local_offset_x", C.c_double),
        ("local_offset_y", C.c_double), ("local_offset_yaw", C.c_double),
        ("footprint_width", C.c_double), ("footprint_height", C.c_double), ("footprint_x_offset", C.c_double),
        ("laser_min_range", C.c_double),
        ("icp_iterations", C.c_int), ("dist_filter_max", C.c_double), ("dist_filter_min", C.c_double),
        ("reg_trs_max", C.c_double), ("reg_sin_rot_max", C.c_double),
        ("nn_mode", C.c_int), ("threads", C.c_int),
        ("registration_mode", C.c_int), ("trials", C.c_int), ("size_control_set", C.c_int), ("zrand", C.c_double),
        ("ransac_phi_max", C.c_double),
    ]


class ScanResult(C.Structure):
    _fields_ = [
        ("pose", C.c_double * 9), ("T", C.c_double * 9), ("rms", C.c_double), ("pairs", C.c_int),
        ("iterations", C.c_int), ("icp_state", C.c_int), ("valid_model", C.c_int), ("valid_scene", C.c_int),
        ("reg_error", C.c_int), ("pushed", C.c_int), ("no_model", C.c_int),
        ("t_raycast", C.c_double), ("t_icp", C.c_double), ("t_push", C.c_double),
    ]


def build(force: bool = False, with_ref: bool | None = None):
    """Compile the oracle (and, when /root/reference is present, oracle/_ref)."""
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "tsd_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libtsd_oracle.so"], stdout=subprocess.DEVNULL)
    if with_ref is None:
        with_ref = os.path.isdir(os.path.join(REFERENCE_ROOT, "src"))
    if with_ref and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "_ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.ora_grid_create.restype = C.c_void_p
        L.ora_grid_create.argtypes = [C.c_int, C.c_double, C.c_double]
        L.ora_grid_destroy.argtypes = [C.c_void_p]
        for n in ("ora_grid_cells", "ora_grid_tiles"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = C.c_int
        for n in ("ora_grid_max_trunc", "ora_grid_max_x"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = C.c_double
        L.ora_free_footprint.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double]
        L.ora_grid_dump.argtypes = [C.c_void_p, _u8p, _dp, _dp, _dp]
        L.ora_grid_load.argtypes = [C.c_void_p, _u8p, _dp, _dp, _dp]
        L.ora_grid_tile_state.argtypes = [C.c_void_p, _u8p, _dp]
        L.ora_grid_digest.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), _dp, _dp]
        L.ora_grid_digest.restype = None
        L.ora_push.argtypes = [C.c_void_p, _dp, _dp, _u8p, C.c_int, C.c_double, C.c_double, C.c_double,
                               C.c_double, C.c_double, C.c_int, C.POINTER(PushStats)]
        L.ora_raycast.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp, _u8p]
        L.ora_raycast.restype = C.c_int
        L.ora_interpolate_bilinear.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp]
        L.ora_icp.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(IcpParams), C.POINTER(IcpResult), _dp]
        L.ora_icp_point_to_line.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(IcpParams), C.POINTER(IcpResult), _dp]
        L.ora_icp_point_to_line.restype = None
        L.ora_icp_pairs.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(IcpParams), _dp, _ip, _ip]
        L.ora_icp_pairs.restype = C.c_int
        L.ora_distance_filter_multiplier.argtypes = [C.c_double, C.c_double, C.c_int]
        L.ora_distance_filter_multiplier.restype = C.c_double
        L.ora_sensor_ingest_f32.argtypes = [_fp, C.c_int, C.c_double, C.c_double, _dp, _u8p]
        L.ora_sensor_ingest_f64.argtypes = [_dp, C.c_int, C.c_double, C.c_double, _dp, _u8p]
        L.ora_laser_min_range_clamp.argtypes = [_fp, C.c_int, C.c_double]
        L.ora_backproject.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.ora_backproject.restype = C.c_int
        L.ora_rays_local.argtypes = [C.c_int, C.c_double, C.c_double, _dp]
        L.ora_rays_transform.argtypes = [_dp, _dp, C.c_int]
        L.ora_rays_rescale.argtypes = [_dp, C.c_int, C.c_double, C.c_double]
        L.ora_scene_from_scan.argtypes = [_dp, _dp, _u8p, C.c_int, _dp, _u8p]
        L.ora_scene_from_scan.restype = C.c_int
        L.ora_mat3_mul.argtypes = [_dp, _dp, _dp]
        L.ora_mat3_inv.argtypes = [_dp, _dp]
        L.ora_calc_angle.argtypes = [_dp]
        L.ora_calc_angle.restype = C.c_double
        L.ora_is_registration_error.argtypes = [_dp, C.c_double, C.c_double]
        L.ora_is_pose_change_significant.argtypes = [_dp, _dp]
        L.ora_occupancy.argtypes = [C.c_void_p, _i8p, _i8p, C.c_int, C.c_int]
        L.ora_occupancy.restype = C.c_int
        L.ora_grid_color_image.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte), C.c_uint, C.c_uint]
        L.ora_grid_color_image.restype = None
        L.ora_grid_store_text.argtypes = [C.c_void_p, C.c_char_p]
        L.ora_grid_store_text.restype = C.c_int
        L.ora_text_lines.argtypes = [C.c_char_p, _ip, C.c_int, _dp]
        L.ora_text_lines.restype = C.c_int
        L.ora_grid_load_text.argtypes = [C.c_char_p]
        L.ora_grid_load_text.restype = C.c_void_p
        L.ora_slam_create.restype = C.c_void_p
        L.ora_slam_create.argtypes = [C.POINTER(SlamConfig)]
        L.ora_slam_destroy.argtypes = [C.c_void_p]
        L.ora_slam_create_shared.restype = C.c_void_p
        L.ora_slam_create_shared.argtypes = [C.POINTER(SlamConfig), C.c_void_p]
        L.ora_slam_grid.restype = C.c_void_p
        L.ora_slam_grid.argtypes = [C.c_void_p]
        L.ora_slam_process_scan.argtypes = [C.c_void_p, _fp, C.POINTER(ScanResult)]
        L.ora_slam_last_push_stats.argtypes = [C.c_void_p, C.POINTER(PushStats)]
        L.ora_slam_set_draws.argtypes = [C.c_void_p, _ip, _ip, _ip]
        L.ora_slam_set_draws.restype = None
        L.ora_slam_last_prereg.argtypes = [C.c_void_p, _dp, _dp, _ip, _ip]
        L.ora_slam_last_prereg.restype = None
        L.ora_tsdpdf_match.argtypes = [C.c_void_p, _dp, _dp, _u8p, _dp, _u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_double, _ip, _ip, _ip, _dp, _dp, _ip, _ip, _ip]
        L.ora_tsdpdf_match.restype = C.c_int
        L.ora_icp_init.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(IcpParams), _dp, C.POINTER(IcpResult), _dp]
        L.ora_icp_init.restype = None
        _lib = L
    return _lib


def ref_available() -> bool:
    return os.path.exists(REF_SO)


def ref():
    """The compiled reference TUs (PairAssignment / DistanceFilter / ReciprocalFilter + mathbase.h)."""
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.ref_chain_create.restype = C.c_void_p
        R.ref_chain_create.argtypes = [C.c_double, C.c_double, C.c_int]
        R.ref_chain_destroy.argtypes = [C.c_void_p]
        R.ref_chain_reset.argtypes = [C.c_void_p]
        R.ref_chain_pairs.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_int, _u8p, _ip, _ip]
        R.ref_chain_pairs.restype = C.c_int
        R.ref_minmax4.argtypes = [_ip, _ip, _ip]
        R.ref_euklid2.argtypes = [_dp, _dp]
        R.ref_euklid2.restype = C.c_double
        R.ref_dist_sqr2d.argtypes = [_dp, _dp]
        R.ref_dist_sqr2d.restype = C.c_double
        R.ref_norm2.argtypes = [_dp]
        R.ref_deg2rad.argtypes = [C.c_double]
        R.ref_deg2rad.restype = C.c_double
        R.ref_text_lines.argtypes = [C.c_char_p, _ip, C.c_int, _dp]
        R.ref_text_lines.restype = None
        _ref = R
    return _ref


def d(a):
    return a.ctypes.data_as(_dp)


def u8(a):
    return a.ctypes.data_as(_u8p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------------------------------------------
class Grid:
    def __init__(self, map_size_log2, cell_size, max_trunc, handle=None):
        self.L = lib()
        self.own = handle is None
        self.h = handle if handle is not None else self.L.ora_grid_create(map_size_log2, cell_size, max_trunc)
        self.cells = self.L.ora_grid_cells(self.h)
        self.tiles = self.L.ora_grid_tiles(self.h)
        self.cell_size = cell_size
        self.max_trunc = self.L.ora_grid_max_trunc(self.h)
        self.max_x = self.L.ora_grid_max_x(self.h)

    def close(self):
        if self.own and self.h:
            self.L.ora_grid_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def store_text(self, path) -> bool:
        """TsdGrid::storeGrid: the reference's text format."""
        return bool(self.L.ora_grid_store_text(self.h, str(path).encode()))

    @classmethod
    def load_text(cls, path, cell_size):
        """TsdGrid(file, FILE_SOURCE): a new grid from a stored file (None if the file is not one)."""
        h = lib().ora_grid_load_text(str(path).encode())
        if not h:
            return None
        g = cls(0, cell_size, 0.0, handle=h)
        g.own = True
        return g

    def free_footprint(self, center, w, h):
        c = f64(center)
        return bool(self.L.ora_free_footprint(self.h, d(c), w, h))

    def push(self, pose, ranges, mask, ang_res, phi_min, max_range, min_range, low_refl, threads=1):
        pose = f64(pose).reshape(9)
        ranges = f64(ranges)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        st = PushStats()
        self.L.ora_push(self.h, d(pose), d(ranges), u8(mask), ranges.size, ang_res, phi_min, max_range,
                        min_range, low_refl, threads, C.byref(st))
        return st.as_dict()

    def raycast(self, pose, rays_world, min_range, max_range, threads=1):
        pose = f64(pose).reshape(9)
        rays = f64(rays_world)
        beams = rays.size // 2
        coords = np.zeros(2 * beams)
        normals = np.zeros(2 * beams)
        mask = np.zeros(beams, dtype=np.uint8)
        n = self.L.ora_raycast(self.h, d(pose), d(rays), beams, min_range, max_range, threads, d(coords),
                               d(normals), u8(mask))
        return coords, normals, mask, n

    def dump(self):
        init = np.zeros(self.tiles, dtype=np.uint8)
        iw = np.zeros(self.tiles)
        tsd = np.zeros((self.tiles, TILE_CELLS))
        w = np.zeros((self.tiles, TILE_CELLS))
        self.L.ora_grid_dump(self.h, u8(init), d(iw), d(tsd), d(w))
        return init, iw, tsd, w

    def digest(self) -> dict:
        h, nv, ni = C.c_uint64(), C.c_int64(), C.c_int32()
        st, sw = C.c_double(), C.c_double()
        self.L.ora_grid_digest(self.h, C.byref(h), C.byref(nv), C.byref(ni), C.byref(st), C.byref(sw))
        return dict(hash=int(h.value), cells_valid=int(nv.value), tiles_initialized=int(ni.value),
                    sum_tsd=float(st.value), sum_weight=float(sw.value))

    def load(self, init, iw, tsd, w):
        init = np.ascontiguousarray(init, dtype=np.uint8)
        iw, tsd, w = f64(iw), f64(tsd), f64(w)
        self.L.ora_grid_load(self.h, u8(init), d(iw), d(tsd), d(w))

    def tile_state(self):
        init = np.zeros(self.tiles, dtype=np.uint8)
        iw = np.zeros(self.tiles)
        self.L.ora_grid_tile_state(self.h, u8(init), d(iw))
        return init, iw

    def bilinear(self, x, y):
        v = C.c_double(0.0)
        st = self.L.ora_interpolate_bilinear(self.h, x, y, C.byref(v))
        return st, v.value

    def color_image(self, width=None, height=None):
        """TsdGrid::grid2ColorImage: (height, width, 3) uint8."""
        n = self.cells
        width = n if width is None else width
        height = n if height is None else height
        img = np.zeros((height, width, 3), dtype=np.uint8)
        self.L.ora_grid_color_image(self.h, img.ctypes.data_as(C.POINTER(C.c_ubyte)), width, height)
        return img

    def occupancy(self, content, inflate=False, factor=2):
        out = np.zeros_like(content)
        n = self.L.ora_occupancy(self.h, content.ctypes.data_as(_i8p), out.ctypes.data_as(_i8p), int(inflate), factor)
        return out, n


def ingest_f32(ranges_f32, max_range, ang_res):
    r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
    data = np.zeros(r.size)
    mask = np.zeros(r.size, dtype=np.uint8)
    lib().ora_sensor_ingest_f32(r.ctypes.data_as(_fp), r.size, max_range, ang_res, d(data), u8(mask))
    return data, mask


def ingest_f64(ranges, max_range, ang_res):
    r = f64(ranges).copy()
    data = np.zeros(r.size)
    mask = np.zeros(r.size, dtype=np.uint8)
    lib().ora_sensor_ingest_f64(d(r), r.size, max_range, ang_res, d(data), u8(mask))
    return data, mask


def rays_local(beams, phi_min, ang_res):
    r = np.zeros(2 * beams)
    lib().ora_rays_local(beams, phi_min, ang_res, d(r))
    return r


def rays_transform(T, rays):
    T = f64(T).reshape(9)
    r = f64(rays).copy()
    lib().ora_rays_transform(d(T), d(r), r.size // 2)
    return r


def rays_rescale(rays, norm_new, norm_old):
    r = f64(rays).copy()
    lib().ora_rays_rescale(d(r), r.size // 2, norm_new, norm_old)
    return r


def scene_from_scan(rays_loc, data, mask):
    beams = data.size
    scene = np.zeros(2 * beams)
    ms = np.zeros(beams, dtype=np.uint8)
    rl, dd = f64(rays_loc), f64(data)
    mk = np.ascontiguousarray(mask, dtype=np.uint8)
    n = lib().ora_scene_from_scan(d(rl), d(dd), u8(mk), beams, d(scene), u8(ms))
    return scene, ms, n


def mat3_inv(A):
    A = f64(A).reshape(9)
    out = np.zeros(9)
    lib().ora_mat3_inv(d(A), d(out))
    return out.reshape(3, 3)


def mat3_mul(A, B):
    A, B = f64(A).reshape(9), f64(B).reshape(9)
    out = np.zeros(9)
    lib().ora_mat3_mul(d(A), d(B), d(out))
    return out.reshape(3, 3)


def icp(model_xy, scene_xy, pose, iterations, dist_max, dist_min, bounds, nn_mode=0, trace=False, model_normals_xy=None):
    """model_normals_xy given: PointToLine2DEstimator instead of the node's ClosedFormEstimator2D."""
    m, s = f64(model_xy).reshape(-1), f64(scene_xy).reshape(-1)
    pose = f64(pose).reshape(9)
    p = IcpParams(iterations, dist_max, dist_min, bounds[0], bounds[1], bounds[2], bounds[3], nn_mode)
    r = IcpResult()
    tr = np.zeros((max(iterations, 1), 8)) if trace else None       # pairs, rms, thr, state, Tlast (c, s, tx, ty)
    if model_normals_xy is None:
        lib().ora_icp(d(m), m.size // 2, d(s), s.size // 2, d(pose), C.byref(p), C.byref(r), d(tr) if trace else None)
    else:
        nrm = f64(model_normals_xy).reshape(-1)
        assert nrm.size == m.size
        lib().ora_icp_point_to_line(d(m), d(nrm), m.size // 2, d(s), s.size // 2, d(pose), C.byref(p), C.byref(r),
                                    d(tr) if trace else None)
    out = {"T": np.array(r.T[:]).reshape(3, 3), "rms": r.rms, "pairs": r.pairs, "iterations": r.iterations,
           "state": r.state}
    if trace:
        out["trace"] = tr[: r.iterations]
    return out


def tsdpdf_match(grid, pose, M, maskM, S, maskS, trials, size_control_set, zrand, phi_max, resolution, draws_sub, draws_ctrl,
                 draws_trials):
    """TSD_PDFMatching::match with the rand() draws as inputs -> dict(T, prob, idx, i, candidates, rc)"""
    M, S = f64(M).reshape(-1), f64(S).reshape(-1)
    n = M.size // 2
    mM, mS = np.ascontiguousarray(maskM, dtype=np.uint8), np.ascontiguousarray(maskS, dtype=np.uint8)
    ds = np.ascontiguousarray(draws_sub, dtype=np.int32); dc = np.ascontiguousarray(draws_ctrl, dtype=np.int32)
    dt = np.ascontiguousarray(draws_trials, dtype=np.int32)
    assert ds.size >= n and dc.size >= size_control_set and dt.size >= trials
    T = np.zeros(9); prob = C.c_double(); idx = C.c_int(); ii = C.c_int(); cand = C.c_int()
    rc = lib().ora_tsdpdf_match(grid.h, d(f64(pose).reshape(9)), d(M), u8(mM), d(S), u8(mS), n, trials, size_control_set, zrand,
                                phi_max, resolution, ds.ctypes.data_as(_ip), dc.ctypes.data_as(_ip), dt.ctypes.data_as(_ip),
                                d(T), C.byref(prob), C.byref(idx), C.byref(ii), C.byref(cand))
    return dict(T=T.reshape(3, 3), prob=prob.value, idx=idx.value, i=ii.value, candidates=cand.value, rc=rc)


def icp_init(model_xy, scene_xy, pose, iterations, dist_max, dist_min, bounds, T_init, nn_mode=0):
    m, s = f64(model_xy).reshape(-1), f64(scene_xy).reshape(-1)
    p = IcpParams(iterations, dist_max, dist_min, bounds[0], bounds[1], bounds[2], bounds[3], nn_mode)
    r = IcpResult()
    lib().ora_icp_init(d(m), m.size // 2, d(s), s.size // 2, d(f64(pose).reshape(9)), C.byref(p), d(f64(T_init).reshape(9)),
                       C.byref(r), None)
    return dict(T=np.array(r.T[:]).reshape(3, 3), rms=r.rms, pairs=r.pairs, iterations=r.iterations, state=r.state)


def icp_pairs(model_xy, scene_xy, pose, iterations, dist_max, dist_min, bounds, thr_sqr, nn_mode=0):
    m, s = f64(model_xy).reshape(-1), f64(scene_xy).reshape(-1)
    pose = f64(pose).reshape(9)
    p = IcpParams(iterations, dist_max, dist_min, bounds[0], bounds[1], bounds[2], bounds[3], nn_mode)
    thr = C.c_double(thr_sqr)
    pm = np.zeros(s.size // 2 + 1, dtype=np.int32)
    ps = np.zeros(s.size // 2 + 1, dtype=np.int32)
    n = lib().ora_icp_pairs(d(m), m.size // 2, d(s), s.size // 2, d(pose), C.byref(p), C.byref(thr),
                            pm.ctypes.data_as(_ip), ps.ctypes.data_as(_ip))
    return pm[:n].copy(), ps[:n].copy(), thr.value


class Slam:
    """ThreadLocalize::init + eventLoop body + synchronous ThreadMapping pushes on the CPU."""

    def __init__(self, shared_with=None, **kw):
        """shared_with: another Slam whose grid this localiser works on (multi-robot mode, SlamNode.cpp:101-122)"""
        self.L = lib()
        self.cfg = SlamConfig(**kw)
        self._first = shared_with          # (keeps the grid's owner alive)
        if shared_with is None:
            self.h = self.L.ora_slam_create(C.byref(self.cfg))
        else:
            self.h = self.L.ora_slam_create_shared(C.byref(self.cfg), shared_with.h)
        self.grid = Grid(kw["map_size_log2"], kw["cell_size"], 0.0, handle=self.L.ora_slam_grid(self.h))

    def close(self):
        if self.h:
            self.L.ora_slam_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_scan(self, ranges_f32):
        r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
        out = ScanResult()
        self.L.ora_slam_process_scan(self.h, r.ctypes.data_as(_fp), C.byref(out))
        return out

    def set_draws(self, sub, ctrl, trials):
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (sub, ctrl, trials)]
        self.L.ora_slam_set_draws(self.h, *[x.ctypes.data_as(_ip) for x in a])

    def last_prereg(self):
        T = np.zeros(9); prob = C.c_double(); idx = C.c_int(); ii = C.c_int()
        self.L.ora_slam_last_prereg(self.h, d(T), C.byref(prob), C.byref(idx), C.byref(ii))
        return dict(T=T.reshape(3, 3), prob=prob.value, idx=idx.value, i=ii.value)

    def last_push_stats(self):
        st = PushStats()
        self.L.ora_slam_last_push_stats(self.h, C.byref(st))
        return st.as_dict()
