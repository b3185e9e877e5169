"""ctypes driver of the C++ facade (``lib/libohm_tsd_slam.so``): ``ThreadLocalize`` / ``ThreadMapping``
wired as ``SlamNode::initialize`` wires them (SlamNode.cpp:27-129), fed through ``laserCallBack``.
Used by the tests and by ``bench.py``; the facade itself is C++ (``csrc/host``)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(capi.LIB_DIR, "libohm_tsd_slam.so")

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_ip = C.POINTER(C.c_int)

HOST_ABI = {
    "tsd_node_create": (C.c_void_p, [C.c_char_p]),
    "tsd_node_set_double": (None, [C.c_void_p, C.c_char_p, C.c_double]),
    "tsd_node_set_int": (None, [C.c_void_p, C.c_char_p, C.c_int]),
    "tsd_node_set_bool": (None, [C.c_void_p, C.c_char_p, C.c_int]),
    "tsd_node_set_string": (None, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "tsd_node_initialize": (C.c_int, [C.c_void_p, C.c_int]),
    "tsd_node_declared_parameters": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "tsd_node_set_synchronous": (None, [C.c_void_p, C.c_int]),
    "tsd_node_set_fused": (None, [C.c_void_p, C.c_int]),
    "tsd_node_laser": (C.c_int, [C.c_void_p, C.c_int, _fp, C.c_int, C.c_double, C.c_double, C.c_longlong]),
    "tsd_node_wait_idle": (C.c_int, [C.c_void_p, C.c_int]),
    "tsd_node_processed": (C.c_ulonglong, [C.c_void_p, C.c_int]),
    "tsd_node_report": (None, [C.c_void_p, C.c_int, _dp]),
    "tsd_node_pose_msg": (None, [C.c_void_p, C.c_int, _dp]),
    "tsd_node_pose_topic": (C.c_char_p, [C.c_void_p, C.c_int]),
    "tsd_node_tf_msg": (None, [C.c_void_p, C.c_int, _dp, C.c_char_p, C.c_int]),
    "tsd_node_set_transform": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, _dp, _dp]),
    "tsd_node_laser_ahead": (C.c_int, [C.c_void_p, C.c_int, _fp, C.c_int, C.c_double, C.c_double, C.c_longlong]),
    "tsd_node_play": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(_fp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_longlong, C.c_longlong]),
    "tsd_node_batch_stats": (None, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "tsd_node_grid_ctx": (C.c_void_p, [C.c_void_p]),
    "tsd_node_grid_lock": (None, [C.c_void_p]),
    "tsd_node_grid_unlock": (None, [C.c_void_p]),
    "tsd_node_destroy": (None, [C.c_void_p]),
    "tsd_host_sensor_ingest_f32": (None, [_fp, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _u8p, C.c_int]),
    "tsd_host_sensor_chain": (None, [C.c_int, C.c_double, C.c_double, _dp, _dp, C.c_double, _fp, _dp, _dp, _dp,
                                     _dp, _u8p, _ip]),
    "tsd_host_calc_angle": (C.c_double, [_dp]),
    "tsd_host_is_registration_error": (C.c_int, [_dp, C.c_double, C.c_double]),
    "tsd_host_is_pose_change_significant": (C.c_int, [_dp, _dp]),
    "tsd_host_mat3_inv": (None, [_dp, _dp]),
    "tsd_host_backproject": (C.c_int, [_dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]),
}

_lib = None


def load_library():
    global _lib
    if _lib is None:
        capi.load_library()   # dependency (resolved through rpath as well)
        if not os.path.exists(LIB_PATH):
            raise capi.TsdError(f"{LIB_PATH} not found: run __graft_entry__.build()")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in HOST_ABI.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


REPORT_FIELDS = ("rms", "pairs", "iterations", "icp_state", "valid_model", "valid_scene", "reg_error", "pushed",
                 "no_model", "initialised")


def declared_parameters(params: dict | None = None, name: str = "tsd_slam") -> dict:
    """(type, declared default) of every parameter the facade declares for a node configured with `params` -- SlamNode,
    the ThreadLocalize constructors and ThreadLocalize::init -- without touching a device."""
    lib = load_library()
    h = lib.tsd_node_create(name.encode())
    for k, v in (params or {}).items():
        kb = k.encode()
        if isinstance(v, bool):
            lib.tsd_node_set_bool(h, kb, int(v))
        elif isinstance(v, int):
            lib.tsd_node_set_int(h, kb, v)
        elif isinstance(v, float):
            lib.tsd_node_set_double(h, kb, v)
        else:
            lib.tsd_node_set_string(h, kb, str(v).encode())
    need = lib.tsd_node_declared_parameters(h, None, 0)
    buf = C.create_string_buffer(need)
    lib.tsd_node_declared_parameters(h, buf, need)
    lib.tsd_node_destroy(h)
    out = {}
    for line in buf.value.decode().splitlines():
        k, t, v = line.split("|", 2)
        out[k] = (t, {"bool": lambda x: x == "true", "int": int, "double": float, "string": str}[t](v))
    return out


class SlamNode:
    """``SlamNode`` wiring: parameters (SURVEY Appendix D names), one grid, one mapping thread, N
    localisers.  ``synchronous=True`` runs the event-loop body inside ``laser()`` (strict
    ray-cast -> ICP -> push order); otherwise the reference's threads/queues are used."""

    def __init__(self, params: dict, device: int = 0, synchronous: bool = True, name: str = "tsd_slam",
                 fused: bool = True):
        self.lib = load_library()
        self.h = self.lib.tsd_node_create(name.encode())
        for k, v in params.items():
            kb = k.encode()
            if isinstance(v, bool):
                self.lib.tsd_node_set_bool(self.h, kb, int(v))
            elif isinstance(v, int):
                self.lib.tsd_node_set_int(self.h, kb, v)
            elif isinstance(v, float):
                self.lib.tsd_node_set_double(self.h, kb, v)
            else:
                self.lib.tsd_node_set_string(self.h, kb, str(v).encode())
        self.lib.tsd_node_set_synchronous(self.h, int(synchronous))
        self.lib.tsd_node_set_fused(self.h, int(fused))
        rc = self.lib.tsd_node_initialize(self.h, device)
        if rc != 0:
            self.lib.tsd_node_destroy(self.h)
            self.h = None
            raise capi.TsdError(f"tsd_node_initialize failed ({rc}): no usable GPU; the hot path has no CPU fall-back")
        self._stamp = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsd_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_synchronous(self, on: bool):
        self.lib.tsd_node_set_synchronous(self.h, int(on))

    def laser(self, ranges_f32, angle_min, angle_increment, robot: int = 0, stamp_ns: int | None = None, ahead=None):
        """``ahead``: the scan the NEXT call will deliver (known in a replay): announced to the localiser, which stages it on
        the device while this scan's registration runs."""
        r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
        if stamp_ns is None:
            self._stamp += 25_000_000
            stamp_ns = self._stamp
        if ahead is not None:
            a = np.ascontiguousarray(ahead, dtype=np.float32)
            self.lib.tsd_node_laser_ahead(self.h, robot, a.ctypes.data_as(_fp), a.size, angle_min, angle_increment, stamp_ns + 25_000_000)
        rc = self.lib.tsd_node_laser(self.h, robot, r.ctypes.data_as(_fp), r.size, angle_min, angle_increment, stamp_ns)
        if rc != 0:
            raise capi.TsdError(f"tsd_node_laser failed ({rc})")

    def play(self, scans, first: int, count: int, angle_min, angle_increment):
        """Replay scans[r][first:first+count] of every robot r from one native publisher thread per robot (`rosbag play`):
        ``scans`` = list (one per robot) of C-contiguous float32 arrays [n_scans, beams]."""
        arrs = [np.ascontiguousarray(s, dtype=np.float32) for s in scans]
        ptrs = (_fp * len(arrs))(*[a.ctypes.data_as(_fp) for a in arrs])
        rc = self.lib.tsd_node_play(self.h, len(arrs), ptrs, first, count, arrs[0].shape[1], angle_min, angle_increment,
                                    self._stamp + 25_000_000, 25_000_000)
        self._stamp += 25_000_000 * (first + count)
        if rc != 0:
            raise capi.TsdError(f"tsd_node_play failed ({rc})")

    def batch_stats(self):
        b, s = C.c_ulonglong(0), C.c_ulonglong(0)
        self.lib.tsd_node_batch_stats(self.h, C.byref(b), C.byref(s))
        return int(b.value), int(s.value)

    def wait_idle(self, timeout_ms: int = 10000) -> bool:
        return self.lib.tsd_node_wait_idle(self.h, timeout_ms) == 0

    def processed(self, robot: int = 0) -> int:
        return int(self.lib.tsd_node_processed(self.h, robot))

    def report(self, robot: int = 0) -> dict:
        buf = np.zeros(29)
        self.lib.tsd_node_report(self.h, robot, buf.ctypes.data_as(_dp))
        out = {"pose": buf[:9].reshape(3, 3).copy(), "T": buf[9:18].reshape(3, 3).copy()}
        out["rms"] = float(buf[18])
        for i, k in enumerate(REPORT_FIELDS[1:]):
            out[k] = int(buf[19 + i])
        out["stamp_ns"] = int(buf[28])
        return out

    def pose_msg(self, robot: int = 0) -> dict:
        buf = np.zeros(8)
        self.lib.tsd_node_pose_msg(self.h, robot, buf.ctypes.data_as(_dp))
        return {"position": buf[:3].copy(), "orientation_xyzw": buf[3:7].copy(), "count": int(buf[7]),
                "topic": self.lib.tsd_node_pose_topic(self.h, robot).decode()}

    def tf_msg(self, robot: int = 0) -> dict:
        """the last TransformStamped the robot's tf broadcaster sent (map -> odom, ThreadLocalize.cpp:603-689)"""
        buf = np.zeros(8)
        frames = C.create_string_buffer(256)
        self.lib.tsd_node_tf_msg(self.h, robot, buf.ctypes.data_as(_dp), frames, 256)
        parent, child = frames.value.decode().split("|")
        return {"translation": buf[:3].copy(), "rotation_xyzw": buf[3:7].copy(), "count": int(buf[7]),
                "frame_id": parent, "child_frame_id": child}

    def set_transform(self, parent: str, child: str, xyz, q_xyzw, robot: int = 0):
        """feed the robot's tf buffer (a TransformListener's job under ROS): frame `child` expressed in frame `parent`"""
        t = np.ascontiguousarray(xyz, dtype=np.float64)
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64)
        rc = self.lib.tsd_node_set_transform(self.h, robot, parent.encode(), child.encode(), t.ctypes.data_as(_dp), q.ctypes.data_as(_dp))
        if rc != 0:
            raise ValueError("tsd_node_set_transform refused %s -> %s" % (parent, child))

    def grid(self) -> "GridView":
        return GridView(self.lib.tsd_node_grid_ctx(self.h), self)


class GridView(capi.TsdGridDevice):
    """Non-owning view of the facade's grid context (for dumps / profiling through the tsd_* ABI).  Every device call
    made through it takes the facade's grid mutex (obvious::TsdGrid::mutex()), like the facade's own classes do: the
    node's localise / mapping threads may be enqueueing on the same context."""

    _LOCKED = ("reset", "sync", "free_footprint", "push", "raycast", "icp", "localize", "icp_trace", "download_tile_state",
               "download_tiles", "upload_tiles", "digest", "occupancy", "occupancy_into", "calibrate_rmw", "profile", "store_text",
               "load_text", "color_image", "push_stats_total", "profile_reset", "profile_get", "profile_spread", "profile_samples", "tsdpdf_match")

    def __init__(self, ctx, node=None):  # noqa: D401 - does not call the base constructor on purpose
        self.lib = capi.load_library()
        self.h = ctx
        self._node = node
        self.cells = self.lib.tsd_cells(ctx)
        self.tiles = self.lib.tsd_tiles(ctx)
        self.cell_size = self.lib.tsd_cell_size(ctx)
        self.max_trunc = self.lib.tsd_max_truncation(ctx)
        self.min_x, self.max_x = self.lib.tsd_min_x(ctx), self.lib.tsd_max_x(ctx)
        self.min_y, self.max_y = self.lib.tsd_min_y(ctx), self.lib.tsd_max_y(ctx)

    def __getattribute__(self, name):
        attr = object.__getattribute__(self, name)
        if name in GridView._LOCKED and callable(attr):
            node = object.__getattribute__(self, "_node")
            if node is not None and getattr(node, "h", None):
                def locked(*a, **k):
                    node.lib.tsd_node_grid_lock(node.h)
                    try:
                        return attr(*a, **k)
                    finally:
                        node.lib.tsd_node_grid_unlock(node.h)
                return locked
        return attr

    def close(self):
        self.h = None


def node_params(gc, geo=None, **over) -> dict:
    """Parameter set of the measurement plan: single-laser.yaml values with
    registration_mode 0 and the benchmark's grid size."""
    p = {
        "robot_nbr": 1, "map_size": gc.map_size_log2, "cellsize": float(gc.cell_size),
        "truncation_radius": gc.truncation_radius, "x_offset": 0.0, "y_offset": 0.0,
        "dist_filter_max": 0.4, "dist_filter_min": 0.02, "icp_iterations": 30,
        "reg_trs_max": 1.0, "reg_sin_rot_max": 0.5, "laser_min_range": 0.26, "registration_mode": 0,
        "tsd_slam/local_offset_x": 0.37, "tsd_slam/local_offset_y": -0.21, "tsd_slam/local_offset_yaw": 0.1,
    }
    p.update(over)
    return p
