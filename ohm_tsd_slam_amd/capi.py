"""ctypes binding of the C ABI in ``include/tsd_hip.h`` (``lib/libtsd_hip.so``).

This is plumbing for the Python test / bench drivers; the product is the HIP library itself and the
C++ facade in ``csrc/host``.  There is no CPU fall-back: importing works without a GPU (so the ABI can
be checked), but every compute call needs a gfx950 device and raises :class:`TsdError` otherwise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSD_LIB_DIR: developer override used by tools/*_stamps*.sh, which build INSTRUMENTED variants of the kernels into
# lib/diag/ so that the product libraries under lib/ are never replaced by a build without parity checks
LIB_DIR = os.environ.get("TSD_LIB_DIR") or os.path.join(_HERE, "lib")
if os.environ.get("TSD_LIB_DIR"):
    import sys as _sys
    print(f"ohm_tsd_slam_amd: using the libraries under TSD_LIB_DIR={LIB_DIR} (diagnostic build)", file=_sys.stderr)
LIB_PATH = os.path.join(LIB_DIR, "libtsd_hip.so")
# the same library built with 32-bit fixed-point cell storage (-DTSD_STORAGE_Q32, csrc/tsd_device.hpp)
LIB_PATH_Q32 = os.path.join(LIB_DIR, "libtsd_hip_q32.so")

TILE_CELLS = 1089
MAX_BEAMS = 4096
MAX_ICP_POINTS = 2048

ICP_STATE = {1: "PROCESSING", 2: "NOTMATCHABLE", 3: "MAXITERATIONS", 5: "SUCCESS"}


class TsdError(RuntimeError):
    pass


class PushStats(C.Structure):
    _fields_ = [
        ("cells_updated", C.c_int64),
        ("cells_visited", C.c_int64),
        ("tiles_total", C.c_int32),
        ("tiles_range_pass", C.c_int32),
        ("tiles_update", C.c_int32),
        ("tiles_new", C.c_int32),
        ("tiles_new_from_empty", C.c_int32),
        ("tiles_emptied_init", C.c_int32),
        ("tiles_emptied_uninit", C.c_int32),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class IcpParams(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("estimator", C.c_int),          # 0 ClosedFormEstimator2D (the node's), 1 PointToLine2DEstimator
        ("dist_filter_max", C.c_double),
        ("dist_filter_min", C.c_double),
        ("min_x", C.c_double),
        ("max_x", C.c_double),
        ("min_y", C.c_double),
        ("max_y", C.c_double),
        ("t_init", C.c_double * 9),      # Tinit of Icp::iterate (pre-registration result), used when use_t_init != 0
        ("use_t_init", C.c_int),
        ("reserved", C.c_int),
    ]


class TsdPdfParams(C.Structure):
    """tsd_tsdpdf_params"""
    _fields_ = [("trials", C.c_int), ("size_control_set", C.c_int), ("eps_thresh", C.c_double), ("zrand", C.c_double),
                ("phi_max", C.c_double), ("ang_res", C.c_double)]


class TsdPdfResult(C.Structure):
    """tsd_tsdpdf_result"""
    _fields_ = [("T", C.c_double * 9), ("probability", C.c_double), ("idx_model", C.c_int32), ("idx_scene", C.c_int32),
                ("candidates", C.c_int32), ("valid_model", C.c_int32), ("valid_scene", C.c_int32), ("control_points", C.c_int32),
                ("reserved", C.c_int32)]


class IcpResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 9),
        ("rms", C.c_double),
        ("pairs", C.c_int32),
        ("iterations", C.c_int32),
        ("state", C.c_int32),
        ("n_model", C.c_int32),
        ("n_scene", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GridDigest(C.Structure):
    """tsd_grid_digest_t"""
    _fields_ = [("hash", C.c_uint64), ("cells_valid", C.c_int64), ("tiles_initialized", C.c_int32),
                ("reserved", C.c_int32), ("sum_tsd", C.c_double), ("sum_weight", C.c_double)]

    def as_dict(self):
        return dict(hash=int(self.hash), cells_valid=int(self.cells_valid), tiles_initialized=int(self.tiles_initialized),
                    sum_tsd=float(self.sum_tsd), sum_weight=float(self.sum_weight))


class GateParams(C.Structure):
    _fields_ = [("reg_trs_max", C.c_double), ("reg_sin_rot_max", C.c_double),
                ("trs_min", C.c_double), ("rot_min", C.c_double)]


class ScanResult(C.Structure):
    _fields_ = [("icp", IcpResult), ("pose", C.c_double * 9), ("reg_error", C.c_int32), ("pushed", C.c_int32),
                ("no_model", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/tsd_hip.h declares: name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)
_ip = C.POINTER(C.c_int)
ABI = {
    "tsd_device_count": (C.c_int, []),
    "tsd_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "tsd_create": (C.c_void_p, [C.c_int, C.c_int, C.c_double, C.c_double]),
    "tsd_destroy": (None, [C.c_void_p]),
    "tsd_reset": (C.c_int, [C.c_void_p]),
    "tsd_set_max_truncation": (C.c_int, [C.c_void_p, C.c_double]),
    "tsd_sync": (C.c_int, [C.c_void_p]),
    "tsd_device": (C.c_int, [C.c_void_p]),
    "tsd_stream": (C.c_void_p, [C.c_void_p]),
    "tsd_last_error": (C.c_char_p, [C.c_void_p]),
    "tsd_cells": (C.c_int, [C.c_void_p]),
    "tsd_tiles": (C.c_int, [C.c_void_p]),
    "tsd_cell_size": (C.c_double, [C.c_void_p]),
    "tsd_max_truncation": (C.c_double, [C.c_void_p]),
    "tsd_min_x": (C.c_double, [C.c_void_p]),
    "tsd_max_x": (C.c_double, [C.c_void_p]),
    "tsd_min_y": (C.c_double, [C.c_void_p]),
    "tsd_max_y": (C.c_double, [C.c_void_p]),
    "tsd_free_footprint": (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_double]),
    "tsd_push": (C.c_int, [C.c_void_p, _dp, _dp, _u8p, C.c_int, C.c_double, C.c_double, C.c_double,
                           C.c_double, C.c_double, C.POINTER(PushStats)]),
    "tsd_raycast": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int, C.c_double, C.c_double, _dp, _dp, _u8p, _ip]),
    "tsd_icp": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(IcpParams),
                          C.POINTER(IcpResult)]),
    "tsd_icp_normals": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(IcpParams),
                                  C.POINTER(IcpResult)]),
    "tsd_localize": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _u8p, C.c_int, C.c_double, C.c_double,
                               C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "tsd_sensor_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "tsd_sensor_destroy": (None, [C.c_void_p]),
    "tsd_sensor_set_pose": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "tsd_scan": (C.c_int, [C.c_void_p, _dp, _u8p, _u8p, C.POINTER(IcpParams), C.POINTER(GateParams),
                           C.POINTER(ScanResult)]),
    "tsd_scan_submit": (C.c_int, [C.c_void_p, _dp, _u8p, _u8p, C.POINTER(IcpParams), C.POINTER(GateParams)]),
    "tsd_scan_stage": (C.c_int, [C.c_void_p, _dp, _u8p, _u8p]),
    "tsd_scan_collect": (C.c_int, [C.c_void_p, C.POINTER(ScanResult)]),
    "tsd_scan_begin": (C.c_int, [C.c_void_p, _dp, _u8p, _u8p, C.POINTER(IcpParams), C.POINTER(GateParams)]),
    "tsd_scan_wait": (C.c_int, [C.c_void_p]),
    "tsd_scan_finish": (C.c_int, [C.c_void_p, C.POINTER(ScanResult)]),
    "tsd_batch_create": (C.c_void_p, [C.c_void_p, C.c_int]),
    "tsd_batch_destroy": (None, [C.c_void_p]),
    "tsd_batch_capacity": (C.c_int, [C.c_void_p]),
    "tsd_batch_inflight": (C.c_int, [C.c_void_p]),
    "tsd_batch_begin": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(_dp), C.POINTER(_u8p), C.POINTER(_u8p),
                                  C.POINTER(IcpParams), C.POINTER(GateParams)]),
    "tsd_batch_push": (C.c_int, [C.c_void_p]),
    "tsd_batch_poll": (C.c_int, [C.c_void_p]),
    "tsd_batch_results": (C.c_int, [C.c_void_p, C.POINTER(ScanResult)]),
    "tsd_icp_trace": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "tsd_tsdpdf_match": (C.c_int, [C.c_void_p, _dp, _dp, _u8p, _dp, _u8p, C.c_int, C.POINTER(TsdPdfParams), _ip, _ip, _ip,
                                   C.POINTER(TsdPdfResult)]),
    "tsd_scan_preregister": (C.c_int, [C.c_void_p, C.POINTER(TsdPdfParams), _dp, _u8p, _ip, _ip, _ip]),
    "tsd_scan_preregistration_result": (C.c_int, [C.c_void_p, C.POINTER(TsdPdfResult)]),
    "tsd_sensor_set_async_mapping": (C.c_int, [C.c_void_p, C.c_int]),
    "tsd_debug_stall_push_stream": (C.c_int, [C.c_void_p, C.c_uint]),
    "tsd_debug_set_icp_helpers": (C.c_int, [C.c_void_p, C.c_int]),
    "tsd_debug_sensor_scan_path": (C.c_int, [C.c_void_p]),
    "tsd_debug_set_push_multi": (C.c_int, [C.c_void_p, C.c_int]),
    "tsd_color_image": (C.c_int, [C.c_void_p, _u8p, C.c_uint, C.c_uint]),
    "tsd_store_grid_text": (C.c_int, [C.c_void_p, C.c_char_p]),
    "tsd_load_grid_text": (C.c_int, [C.c_void_p, C.c_char_p]),
    "tsd_download_tiles": (C.c_int, [C.c_void_p, _u8p, _dp, _dp, _dp]),
    "tsd_upload_tiles": (C.c_int, [C.c_void_p, _u8p, _dp, _dp, _dp]),
    "tsd_download_tile_state": (C.c_int, [C.c_void_p, _u8p, _dp]),
    "tsd_grid_digest": (C.c_int, [C.c_void_p, C.POINTER(GridDigest)]),
    "tsd_storage_bits": (C.c_int, []),
    "tsd_abi_sizeof": (C.c_int, [C.c_char_p]),
    "tsd_occupancy": (C.c_int, [C.c_void_p, _i8p, C.c_int, C.c_int, _ip]),
    "tsd_occupancy_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "tsd_occupancy_dev_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "tsd_icp_pairs": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, _dp, C.POINTER(IcpParams), C.c_int, _ip, _ip, _ip]),
    "tsd_calibrate_rmw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "tsd_measure_stream": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, _dp, _dp]),
    "tsd_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "tsd_profile_select": (C.c_int, [C.c_void_p, C.c_char_p]),
    "tsd_profile_reset": (C.c_int, [C.c_void_p]),
    "tsd_push_stats_total": (C.c_int, [C.c_void_p, C.POINTER(PushStats), C.POINTER(C.c_int64), C.c_int]),
    "tsd_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, _dp, _ip]),
    "tsd_profile_get_spread": (C.c_int, [C.c_void_p, C.c_char_p, _dp, _dp, _dp]),
    "tsd_profile_get_samples": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int]),
}

_lib = None


def load_library(path: str | None = None):
    """Load ``libtsd_hip.so`` and bind every ABI symbol.  Raises if the library is missing: the HIP
    extension is the only implementation."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise TsdError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fall-back")
    lib = C.CDLL(p)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)     # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the ctypes mirrors of the public structs must have the library's layout
    for cname, mirror in (("tsd_push_stats", PushStats), ("tsd_icp_params", IcpParams), ("tsd_icp_result", IcpResult),
                          ("tsd_gate_params", GateParams), ("tsd_scan_result", ScanResult), ("tsd_grid_digest_t", GridDigest),
                          ("tsd_tsdpdf_params", TsdPdfParams), ("tsd_tsdpdf_result", TsdPdfResult)):
        if lib.tsd_abi_sizeof(cname.encode()) != C.sizeof(mirror):
            raise TsdError(f"ABI mismatch: sizeof({cname}) = {lib.tsd_abi_sizeof(cname.encode())} in {p}, {C.sizeof(mirror)} in capi.py")
    if path is None:
        _lib = lib
    return lib


def device_memory(device: int = 0):
    """(free, total) bytes of `device` as the product's own HIP runtime reports them (tsd_device_memory)."""
    lib = load_library()
    f, t = C.c_uint64(0), C.c_uint64(0)
    rc = lib.tsd_device_memory(device, C.byref(f), C.byref(t))
    if rc != 0:
        raise TsdError(f"tsd_device_memory({device}) failed ({rc})")
    return f.value, t.value


def _d(a):
    return a.ctypes.data_as(_dp)


def _u8(a):
    return a.ctypes.data_as(_u8p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


@dataclass
class IcpOut:
    T: np.ndarray
    rms: float
    pairs: int
    iterations: int
    state: int
    n_model: int = 0
    n_scene: int = 0
    seeded: int = 0         # tsd_icp_result.reserved: scene points whose step-0 search came from the helper workgroups


class TsdGridDevice:
    """One TSD grid resident in the HBM of one GPU (``obvious::TsdGrid`` as constructed by
    ``SlamNode::initialize``, SlamNode.cpp:77-78)."""

    def __init__(self, map_size_log2: int, cell_size: float, max_trunc: float, device: int = 0, storage: str = "f64"):
        """storage: "f64" (the reference's cells, lib/libtsd_hip.so) or "q32" (8 bytes per cell, lib/libtsd_hip_q32.so)"""
        self.lib = load_library() if storage == "f64" else load_library(LIB_PATH_Q32)
        assert self.lib.tsd_storage_bits() == (64 if storage == "f64" else 32)
        if self.lib.tsd_device_count() <= 0:
            raise TsdError("no HIP device visible: the TSD hot path only exists as gfx950 kernels")
        self.h = self.lib.tsd_create(device, map_size_log2, cell_size, max_trunc)
        if not self.h:
            raise TsdError("tsd_create failed")
        self.cells = self.lib.tsd_cells(self.h)
        self.tiles = self.lib.tsd_tiles(self.h)
        self.cell_size = self.lib.tsd_cell_size(self.h)
        self.max_trunc = self.lib.tsd_max_truncation(self.h)
        self.min_x, self.max_x = self.lib.tsd_min_x(self.h), self.lib.tsd_max_x(self.h)
        self.min_y, self.max_y = self.lib.tsd_min_y(self.h), self.lib.tsd_max_y(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise TsdError(f"{what} failed ({rc}): {self.lib.tsd_last_error(self.h).decode()}")

    def reset(self):
        self._check(self.lib.tsd_reset(self.h), "tsd_reset")

    def sync(self):
        self._check(self.lib.tsd_sync(self.h), "tsd_sync")

    def free_footprint(self, center, width, height) -> bool:
        c = _f64(center)
        rc = self.lib.tsd_free_footprint(self.h, _d(c), float(width), float(height))
        if rc == -4:
            return False
        self._check(rc, "tsd_free_footprint")
        return True

    def push(self, pose, ranges, mask, ang_res, phi_min, max_range, min_range, low_refl, want_stats=True):
        pose = _f64(pose).reshape(9)
        ranges = _f64(ranges)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        st = PushStats()
        rc = self.lib.tsd_push(self.h, _d(pose), _d(ranges), _u8(mask), ranges.size, ang_res, phi_min,
                               max_range, min_range, low_refl, C.byref(st) if want_stats else None)
        self._check(rc, "tsd_push")
        return st.as_dict() if want_stats else None

    def raycast(self, pose, rays_world, min_range, max_range):
        pose = _f64(pose).reshape(9)
        rays = _f64(rays_world)
        beams = rays.size // 2
        coords = np.zeros(2 * beams)
        normals = np.zeros(2 * beams)
        mask = np.zeros(beams, dtype=np.uint8)
        n = C.c_int(0)
        rc = self.lib.tsd_raycast(self.h, _d(pose), _d(rays), beams, min_range, max_range, _d(coords),
                                  _d(normals), _u8(mask), C.byref(n))
        self._check(rc, "tsd_raycast")
        return coords, normals, mask, n.value

    def icp_params(self, iterations, dist_max, dist_min, estimator: int = 0, t_init=None) -> IcpParams:
        p = IcpParams(iterations, estimator, dist_max, dist_min, self.min_x, self.max_x, self.min_y, self.max_y)
        if t_init is not None:
            p.t_init = (C.c_double * 9)(*_f64(t_init).reshape(9))
            p.use_t_init = 1
        return p

    def tsdpdf_match(self, pose, model_xy, mask_m, scene_xy, mask_s, trials, size_control_set, zrand, phi_max, ang_res,
                     draws_sub, draws_ctrl, draws_trials, eps_thresh=0.15) -> dict:
        """obvious::TSD_PDFMatching::match with the rand() draws as inputs (tsd_tsdpdf_match)"""
        M, S = _f64(model_xy).reshape(-1), _f64(scene_xy).reshape(-1)
        n = M.size // 2
        mM, mS = np.ascontiguousarray(mask_m, dtype=np.uint8), np.ascontiguousarray(mask_s, dtype=np.uint8)
        ds, dc, dt = (np.ascontiguousarray(x, dtype=np.int32) for x in (draws_sub, draws_ctrl, draws_trials))
        assert ds.size >= n and dc.size >= size_control_set and dt.size >= trials
        prm = TsdPdfParams(trials, size_control_set, eps_thresh, zrand, phi_max, ang_res)
        r = TsdPdfResult()
        rc = self.lib.tsd_tsdpdf_match(self.h, _d(_f64(pose).reshape(9)), _d(M), _u8(mM), _d(S), _u8(mS), n, C.byref(prm),
                                       ds.ctypes.data_as(_ip), dc.ctypes.data_as(_ip), dt.ctypes.data_as(_ip), C.byref(r))
        self._check(rc, "tsd_tsdpdf_match")
        return dict(T=np.array(r.T[:]).reshape(3, 3), prob=r.probability, idx=r.idx_model, i=r.idx_scene,
                    candidates=r.candidates, valid_model=r.valid_model, valid_scene=r.valid_scene, control=r.control_points)

    def icp(self, model_xy, scene_xy, pose, params: IcpParams, model_normals_xy=None) -> IcpOut:
        m = _f64(model_xy).reshape(-1)
        s = _f64(scene_xy).reshape(-1)
        pose = _f64(pose).reshape(9)
        r = IcpResult()
        if model_normals_xy is None:
            rc = self.lib.tsd_icp(self.h, _d(m), m.size // 2, _d(s), s.size // 2, _d(pose), C.byref(params), C.byref(r))
        else:
            nrm = _f64(model_normals_xy).reshape(-1)
            assert nrm.size == m.size
            rc = self.lib.tsd_icp_normals(self.h, _d(m), _d(nrm), m.size // 2, _d(s), s.size // 2, _d(pose),
                                          C.byref(params), C.byref(r))
        self._check(rc, "tsd_icp")
        return IcpOut(np.array(r.T[:]).reshape(3, 3), r.rms, r.pairs, r.iterations, r.state, seeded=r.reserved)

    def localize(self, pose, rays_world, rays_local, ranges, mask, min_range, max_range, params: IcpParams) -> IcpOut:
        pose = _f64(pose).reshape(9)
        rw, rl, rg = _f64(rays_world), _f64(rays_local), _f64(ranges)
        mk = np.ascontiguousarray(mask, dtype=np.uint8)
        r = IcpResult()
        rc = self.lib.tsd_localize(self.h, _d(pose), _d(rw), _d(rl), _d(rg), _u8(mk), rg.size, min_range,
                                   max_range, C.byref(params), C.byref(r))
        self._check(rc, "tsd_localize")
        return IcpOut(np.array(r.T[:]).reshape(3, 3), r.rms, r.pairs, r.iterations, r.state, r.n_model, r.n_scene, seeded=r.reserved)

    def set_icp_helpers(self, on: bool):
        """test hook: off = every registration does its first step's searches itself (same results, tsd_debug_set_icp_helpers)"""
        self._check(self.lib.tsd_debug_set_icp_helpers(self.h, 1 if on else 0), "tsd_debug_set_icp_helpers")

    def set_push_multi(self, on: bool):
        """test hook: off = tsd_batch_push enqueues one push per robot instead of one pass per tile (same grid, tsd_debug_set_push_multi)"""
        self._check(self.lib.tsd_debug_set_push_multi(self.h, 1 if on else 0), "tsd_debug_set_push_multi")

    def icp_pairs(self, model_xy, scene_xy, pose, params: IcpParams, calls: int):
        """tsd_icp_pairs: [(model_idx, scene_idx)] of each of the first ``calls`` determinePairs calls on the static scene"""
        m, sc, ps = _f64(model_xy).reshape(-1), _f64(scene_xy).reshape(-1), _f64(pose).reshape(9)
        nm, ns = len(m) // 2, len(sc) // 2
        n = np.zeros(calls, dtype=np.int32)
        mi, si = np.zeros(calls * ns, dtype=np.int32), np.zeros(calls * ns, dtype=np.int32)
        rc = self.lib.tsd_icp_pairs(self.h, _d(m), nm, _d(sc), ns, _d(ps), C.byref(params), calls, n.ctypes.data_as(_ip),
                                    mi.ctypes.data_as(_ip), si.ctypes.data_as(_ip))
        self._check(rc, "tsd_icp_pairs")
        return [(mi[k * ns:k * ns + n[k]].copy(), si[k * ns:k * ns + n[k]].copy()) for k in range(calls)]

    def icp_trace(self, iterations):
        """per iteration: pairs, rms, DistanceFilter threshold before the step, state, Tlast as (c, s, tx, ty)"""
        tr = np.zeros((max(iterations, 1), 8))
        self._check(self.lib.tsd_icp_trace(self.h, _d(tr), iterations), "tsd_icp_trace")
        return tr[:iterations]

    def download_tile_state(self):
        init = np.zeros(self.tiles, dtype=np.uint8)
        iw = np.zeros(self.tiles)
        self._check(self.lib.tsd_download_tile_state(self.h, _u8(init), _d(iw)), "tsd_download_tile_state")
        return init, iw

    def download_tiles(self):
        init = np.zeros(self.tiles, dtype=np.uint8)
        iw = np.zeros(self.tiles)
        tsd = np.zeros((self.tiles, TILE_CELLS))
        w = np.zeros((self.tiles, TILE_CELLS))
        self._check(self.lib.tsd_download_tiles(self.h, _u8(init), _d(iw), _d(tsd), _d(w)), "tsd_download_tiles")
        return init, iw, tsd, w

    def upload_tiles(self, init, iw, tsd, w):
        init = np.ascontiguousarray(init, dtype=np.uint8)
        iw, tsd, w = _f64(iw), _f64(tsd), _f64(w)
        self._check(self.lib.tsd_upload_tiles(self.h, _u8(init), _d(iw), _d(tsd), _d(w)), "tsd_upload_tiles")

    def digest(self) -> dict:
        d = GridDigest()
        self._check(self.lib.tsd_grid_digest(self.h, C.byref(d)), "tsd_grid_digest")
        return d.as_dict()

    def occupancy(self, inflate=False, inflate_factor=2):
        occ = np.zeros(self.cells * self.cells, dtype=np.int8)
        n = C.c_int(0)
        rc = self.lib.tsd_occupancy(self.h, occ.ctypes.data_as(_i8p), int(inflate), inflate_factor, C.byref(n))
        self._check(rc, "tsd_occupancy")
        return occ.reshape(self.cells, self.cells), n.value

    def occupancy_into(self, dev_ptr: int, inflate=False, inflate_factor=2):
        self._check(self.lib.tsd_occupancy_dev(self.h, C.c_void_p(dev_ptr), int(inflate), inflate_factor),
                    "tsd_occupancy_dev")

    def measure_stream(self, n_doubles: int, reps: int = 5):
        """(best, mean) GB/s of the RMW stream kernel over two arrays of ``n_doubles`` (tsd_measure_stream)"""
        best, mean = C.c_double(), C.c_double()
        self._check(self.lib.tsd_measure_stream(self.h, n_doubles, reps, C.byref(best), C.byref(mean)), "tsd_measure_stream")
        return best.value, mean.value

    def calibrate_rmw(self, n_doubles: int, reps: int = 3):
        self._check(self.lib.tsd_calibrate_rmw(self.h, n_doubles, reps), "tsd_calibrate_rmw")

    def profile(self, on=True, kernels="all"):
        self.lib.tsd_profile_select(self.h, kernels.encode())
        self.lib.tsd_profile_enable(self.h, int(on))

    def store_text(self, path):
        """TsdGrid::storeGrid: the reference's text format."""
        self._check(self.lib.tsd_store_grid_text(self.h, str(path).encode()), "tsd_store_grid_text")

    def load_text(self, path):
        """TsdGrid(file): replaces the grid's content by a stored file of the same layout."""
        self._check(self.lib.tsd_load_grid_text(self.h, str(path).encode()), "tsd_load_grid_text")

    def color_image(self, width=None, height=None):
        """TsdGrid::grid2ColorImage: (height, width, 3) uint8."""
        width = self.cells if width is None else width
        height = self.cells if height is None else height
        img = np.zeros((height, width, 3), dtype=np.uint8)
        self._check(self.lib.tsd_color_image(self.h, img.ctypes.data_as(_u8p), width, height), "tsd_color_image")
        return img

    def push_stats_total(self, reset=False):
        st = PushStats()
        n = C.c_int64(0)
        self._check(self.lib.tsd_push_stats_total(self.h, C.byref(st), C.byref(n), int(reset)), "tsd_push_stats_total")
        return st.as_dict(), n.value

    def profile_reset(self):
        self.lib.tsd_profile_reset(self.h)

    def profile_spread(self, kernel: str):
        """(min, max, std) in ms of the timed dispatches of one kernel since the last reset"""
        mn, mx, sd = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        self.lib.tsd_profile_get_spread(self.h, kernel.encode(), C.byref(mn), C.byref(mx), C.byref(sd))
        return mn.value, mx.value, sd.value

    def profile_samples(self, kernel: str, cap: int = 65536) -> np.ndarray:
        """the timed dispatches of one kernel since the last reset, in launch order (ms)"""
        buf = np.zeros(cap, dtype=np.float32)
        n = self.lib.tsd_profile_get_samples(self.h, kernel.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), cap)
        self._check(min(n, 0), "tsd_profile_get_samples")
        return buf[: min(n, cap)].copy()

    def profile_get(self, kernel: str):
        ms = C.c_double(0.0)
        n = C.c_int(0)
        self.lib.tsd_profile_get(self.h, kernel.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value


class TsdSensorDevice:
    """Device-resident ``SensorPolar2D`` + pose bookkeeping of one robot on a grid (fused scan path)."""

    def __init__(self, grid: TsdGridDevice, beams, ang_res, phi_min, max_range, min_range, low_refl):
        self.grid = grid
        self.lib = grid.lib
        self.h = self.lib.tsd_sensor_create(grid.h, beams, ang_res, phi_min, max_range, min_range, low_refl)
        if not self.h:
            raise TsdError("tsd_sensor_create failed")

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsd_sensor_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pose(self, pose, rays_world, rays_local):
        pose, rw, rl = _f64(pose).reshape(9), _f64(rays_world), _f64(rays_local)
        self.grid._check(self.lib.tsd_sensor_set_pose(self.h, _d(pose), _d(rw), _d(rl)), "tsd_sensor_set_pose")

    def scan_ahead(self, ranges, mask, mask_push, params: IcpParams, gates: GateParams, nxt=None, nxt_pre=None) -> ScanResult:
        """tsd_scan_submit (``ranges`` None: the scan staged by the previous call) + tsd_scan_stage of ``nxt`` = (ranges, mask,
        mask_push) while the registration runs (+ tsd_scan_preregister(*nxt_pre) for that staged scan, also ahead of the collect)
        + tsd_scan_collect"""
        if ranges is None:
            rc = self.lib.tsd_scan_submit(self.h, None, None, None, C.byref(params), C.byref(gates))
        else:
            rg, mk = _f64(ranges), np.ascontiguousarray(mask, dtype=np.uint8)
            mp = np.ascontiguousarray(mask_push, dtype=np.uint8) if mask_push is not None else None
            rc = self.lib.tsd_scan_submit(self.h, _d(rg), _u8(mk), _u8(mp) if mp is not None else None, C.byref(params), C.byref(gates))
        self.grid._check(rc, "tsd_scan_submit")
        if nxt is not None:
            rg, mk = _f64(nxt[0]), np.ascontiguousarray(nxt[1], dtype=np.uint8)
            mp = np.ascontiguousarray(nxt[2], dtype=np.uint8) if nxt[2] is not None else None
            self.grid._check(self.lib.tsd_scan_stage(self.h, _d(rg), _u8(mk), _u8(mp) if mp is not None else None), "tsd_scan_stage")
        if nxt_pre is not None:
            self.preregister(*nxt_pre)
        r = ScanResult()
        self.grid._check(self.lib.tsd_scan_collect(self.h, C.byref(r)), "tsd_scan_collect")
        return r

    def preregister(self, scene_xy, mask_s, trials, size_control_set, zrand, phi_max, ang_res, draws_sub, draws_ctrl, draws_trials):
        """tsd_scan_preregister: registration_mode 3's TSD_PDF pre-registration for the NEXT scan of this sensor, on the device
        between its ray cast and its registration"""
        prm = TsdPdfParams(int(trials), int(size_control_set), 0.0, float(zrand), float(phi_max), float(ang_res))
        S, mS = _f64(scene_xy).reshape(-1), np.ascontiguousarray(mask_s, dtype=np.uint8)
        ds, dc, dt = (np.ascontiguousarray(x, dtype=np.int32) for x in (draws_sub, draws_ctrl, draws_trials))
        rc = self.lib.tsd_scan_preregister(self.h, C.byref(prm), _d(S), _u8(mS), ds.ctypes.data_as(_ip), dc.ctypes.data_as(_ip),
                                           dt.ctypes.data_as(_ip))
        self.grid._check(rc, "tsd_scan_preregister")

    def set_async_mapping(self, on=True):
        """tsd_sensor_set_async_mapping: the fused scan's push beside the next registration (the next ray cast one push behind)."""
        self.grid._check(self.lib.tsd_sensor_set_async_mapping(self.h, 1 if on else 0), "tsd_sensor_set_async_mapping")

    def preregistration_result(self) -> dict:
        r = TsdPdfResult()
        self.grid._check(self.lib.tsd_scan_preregistration_result(self.h, C.byref(r)), "tsd_scan_preregistration_result")
        return {"T": np.array(r.T[:]).reshape(3, 3), "prob": r.probability, "idx": r.idx_model, "i": r.idx_scene,
                "candidates": r.candidates, "valid_model": r.valid_model, "valid_scene": r.valid_scene, "control_points": r.control_points}

    def scan(self, ranges, mask, mask_push, params: IcpParams, gates: GateParams) -> ScanResult:
        rg = _f64(ranges)
        mk = np.ascontiguousarray(mask, dtype=np.uint8)
        mp = np.ascontiguousarray(mask_push, dtype=np.uint8) if mask_push is not None else None
        r = ScanResult()
        rc = self.lib.tsd_scan(self.h, _d(rg), _u8(mk), _u8(mp) if mp is not None else None, C.byref(params),
                               C.byref(gates), C.byref(r))
        self.grid._check(rc, "tsd_scan")
        return r


class TsdBatch:
    """One batch slot of the multi-robot path (``tsd_batch_*``): the scans of several sensors on one grid registered by one
    launch of each kernel, their pushes applied in the order of the batch."""

    def __init__(self, grid: TsdGridDevice, max_scans: int):
        self.grid = grid
        self.lib = grid.lib
        self.h = self.lib.tsd_batch_create(grid.h, max_scans)
        if not self.h:
            raise TsdError("tsd_batch_create failed")
        self._keep = None
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsd_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self, sensors, ranges, masks, masks_push, params, gates):
        """``params`` / ``gates``: one IcpParams / GateParams for all scans, or a list with one per scan."""
        n = len(sensors)
        rg = [_f64(r) for r in ranges]
        mk = [np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
        mp = [np.ascontiguousarray(m, dtype=np.uint8) if m is not None else None for m in (masks_push or [None] * n)]
        hs = (C.c_void_p * n)(*[s.h for s in sensors])
        rp = (_dp * n)(*[_d(r) for r in rg])
        mkp = (_u8p * n)(*[_u8(m) for m in mk])
        mpp = (_u8p * n)(*[_u8(m) if m is not None else C.cast(None, _u8p) for m in mp])
        pl = params if isinstance(params, (list, tuple)) else [params] * n
        gl = gates if isinstance(gates, (list, tuple)) else [gates] * n
        pa = (IcpParams * n)(*pl)
        ga = (GateParams * n)(*gl)
        self._keep = (rg, mk, mp)
        self.grid._check(self.lib.tsd_batch_begin(self.h, n, hs, rp, mkp, mpp, pa, ga), "tsd_batch_begin")
        self.n = n

    def push(self):
        self.grid._check(self.lib.tsd_batch_push(self.h), "tsd_batch_push")

    def poll(self) -> bool:
        return self.lib.tsd_batch_poll(self.h) == 1

    def results(self):
        out = (ScanResult * self.n)()
        self.grid._check(self.lib.tsd_batch_results(self.h, out), "tsd_batch_results")
        self._keep = None
        return list(out)
