"""CPU-side checks: the C ABI library loads and exports every declared symbol (no compute calls without a
GPU), the product fails loudly without a device, and the C++ host logic of the facade (sensor model,
pose algebra, gates) equals the oracle bit for bit."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from ohm_tsd_slam_amd import capi, facade, synth
from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def test_header_symbols_are_exported(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "tsd_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tsd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 28
    assert declared == set(capi.ABI.keys()), declared ^ set(capi.ABI.keys())
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in include/tsd_hip.h but not exported"


def test_comm_header_symbols_are_exported():
    """include/tsd_comm.h (the RCCL occupancy merge behind the C ABI, lib/libtsd_comm.so): loads without a GPU, exports
    every declared symbol; no collective is issued here."""
    from ohm_tsd_slam_amd import multigpu
    lib = multigpu.load_comm_library()
    hdr = open(os.path.join(ROOT, "include", "tsd_comm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tsd_comm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(multigpu.COMM_ABI.keys()), declared ^ set(multigpu.COMM_ABI.keys())
    for name in declared:
        assert hasattr(lib, name)
    src = open(os.path.join(ROOT, "ohm_tsd_slam_amd", "csrc", "comm.hip")).read()
    assert "ncclAllReduce(" in src and "ncclInt8" in src and "ncclMax" in src and "hipStreamSynchronize" in src.split("tsd_comm_occupancy_wait")[-1]


def test_struct_layouts_match_header():
    assert C.sizeof(capi.PushStats) == 8 + 8 + 7 * 4 + 4      # padded to 8
    assert C.sizeof(capi.IcpParams) == 8 + 6 * 8 + 9 * 8 + 8  # iterations + estimator share the first 8 bytes; t_init + flag
    assert capi.IcpParams.estimator.offset == 4 and capi.IcpParams.dist_filter_max.offset == 8
    assert C.sizeof(capi.IcpResult) == 9 * 8 + 8 + 6 * 4
    lib = capi.load_library()                                  # (load_library itself refuses a size mismatch)
    for cname, mirror in (("tsd_push_stats", capi.PushStats), ("tsd_icp_params", capi.IcpParams), ("tsd_scan_result", capi.ScanResult),
                          ("tsd_tsdpdf_params", capi.TsdPdfParams), ("tsd_tsdpdf_result", capi.TsdPdfResult),
                          ("tsd_grid_digest_t", capi.GridDigest)):
        assert lib.tsd_abi_sizeof(cname.encode()) == C.sizeof(mirror), cname
    assert lib.tsd_abi_sizeof(b"no_such_struct") == 0


@pytest.mark.skipif(capi.load_library().tsd_device_count() > 0, reason="a GPU is present")
def test_no_gpu_fails_loudly():
    """No silent CPU path: without a device the product refuses to construct."""
    with pytest.raises(capi.TsdError):
        capi.TsdGridDevice(8, 0.05, 0.15)
    with pytest.raises(capi.TsdError):
        facade.SlamNode(facade.node_params(synth.GridConfig(8, 0.05)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ohm_tsd_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "tsd_oracle" not in txt and "libtsd_ref" not in txt, f


def scans():
    gc, geo, scene = synth.CONFIGS["cfg2"]
    world = synth.World(scene, gc)
    r = world.scan(world.start[0], world.start[1], 0.1, geo).copy()
    r[5:9] = 0.0
    r[50:53] = np.nan
    r[100:120] = 45.0
    r[200] = np.inf
    r[300] = 0.3
    return geo, r


@pytest.mark.parametrize("remask", [0, 1, 2])
def test_host_sensor_ingest_equals_oracle(remask):
    H = facade.load_library()
    geo, r = scans()
    data = np.zeros(geo.beams)
    mask = np.zeros(geo.beams, dtype=np.uint8)
    H.tsd_host_sensor_ingest_f32(r.ctypes.data_as(_fp), geo.beams, geo.angle_increment, geo.angle_min, 30.0,
                                 data.ctypes.data_as(_dp), mask.ctypes.data_as(C.POINTER(C.c_uint8)), remask)
    od, om = O.ingest_f32(r, 30.0, geo.angle_increment)
    if remask:
        od, om = O.ingest_f64(od, 30.0, geo.angle_increment)
    assert np.array_equal(data, od) and np.array_equal(mask, om)
    # the quirks: > max_range -> +inf with mask TRUE; NaN -> +inf masked (valid again after re-mask)
    assert np.isinf(data[100]) and mask[100] == 1
    assert np.isinf(data[50]) and mask[50] == (1 if remask else 0)
    assert mask[5] == 0


def test_host_sensor_chain_equals_oracle():
    H = facade.load_library()
    geo, r = scans()
    B = geo.beams
    T1 = synth.pose_matrix(51.57, 50.99, 0.1)
    T2 = synth.pose_matrix(0.031, -0.012, 0.0042)
    pose, rays, rl = np.zeros(9), np.zeros(2 * B), np.zeros(2 * B)
    scene, sm, valid = np.zeros(2 * B), np.zeros(B, dtype=np.uint8), C.c_int(0)
    H.tsd_host_sensor_chain(B, geo.angle_increment, geo.angle_min, T1.ctypes.data_as(_dp), T2.ctypes.data_as(_dp),
                            0.025, r.ctypes.data_as(_fp), pose.ctypes.data_as(_dp), rays.ctypes.data_as(_dp),
                            rl.ctypes.data_as(_dp), scene.ctypes.data_as(_dp),
                            sm.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(valid))
    orl = O.rays_local(B, geo.angle_min, geo.angle_increment)
    orw = O.rays_transform(T1, orl)
    orw = O.rays_rescale(orw, 0.025, 1.0)
    orw = O.rays_transform(T2, orw)
    opose = O.mat3_mul(O.mat3_mul(np.eye(3), T1), T2)
    od, om = O.ingest_f32(r, 30.0, geo.angle_increment)
    oscene, osm, ovalid = O.scene_from_scan(orl, od, om)
    assert np.array_equal(pose.reshape(3, 3), opose)
    assert np.array_equal(rays, orw) and np.array_equal(rl, orl)
    assert valid.value == ovalid and np.array_equal(sm, osm)
    sel = np.repeat(osm.astype(bool), 2)
    assert np.array_equal(scene[sel], oscene[sel])


def test_host_gates_equal_oracle():
    H = facade.load_library()
    L = O.lib()
    rng = np.random.default_rng(11)
    mats = [synth.pose_matrix(rng.normal(0, 0.3), rng.normal(0, 0.3), rng.normal(0, 0.4)) for _ in range(300)]
    mats += [np.eye(3), synth.pose_matrix(0, 0, np.pi), synth.pose_matrix(1.0, 0, 0.0), synth.pose_matrix(0.0, 0.0, -0.2)]
    for i, T in enumerate(mats):
        Tf = np.ascontiguousarray(T).reshape(9)
        assert H.tsd_host_calc_angle(Tf.ctypes.data_as(_dp)) == L.ora_calc_angle(O.d(Tf))
        assert H.tsd_host_is_registration_error(Tf.ctypes.data_as(_dp), 0.25, 0.17) == \
            L.ora_is_registration_error(O.d(Tf), 0.25, 0.17)
        U = np.ascontiguousarray(mats[(i * 7 + 3) % len(mats)]).reshape(9)
        assert H.tsd_host_is_pose_change_significant(Tf.ctypes.data_as(_dp), U.ctypes.data_as(_dp)) == \
            L.ora_is_pose_change_significant(O.d(Tf), O.d(U))
        inv = np.zeros(9)
        H.tsd_host_mat3_inv(Tf.ctypes.data_as(_dp), inv.ctypes.data_as(_dp))
        assert np.array_equal(inv.reshape(3, 3), O.mat3_inv(T))
    # calcAngle quirk: exact identity / pi rotations return 0
    I = np.eye(3).reshape(9)
    assert H.tsd_host_calc_angle(I.ctypes.data_as(_dp)) == 0.0


def test_host_backproject_equals_oracle():
    H = facade.load_library()
    geo = synth.ScanGeometry.utm30lx()
    pose = np.ascontiguousarray(synth.pose_matrix(51.57, 50.99, 0.37))
    pinv = O.mat3_inv(pose).reshape(9)
    rng = np.random.default_rng(5)
    n_valid = 0
    for _ in range(2000):
        x, y = rng.uniform(30, 72, 2)
        a = H.tsd_host_backproject(pose.ctypes.data_as(_dp), x, y, geo.beams, geo.angle_increment, geo.angle_min)
        b = O.lib().ora_backproject(O.d(pinv), x, y, geo.angle_min, geo.angle_increment, geo.beams)
        assert a == b
        n_valid += a >= 0
    assert 1000 < n_valid < 2000      # both in-view and out-of-view (-1 / -2) cases were hit


# ---- parameter surface: names, types and defaults the facade declares, against a table transcribed
# ---- from the reference's declare_parameter calls
def _reference_parameter_table(robot=""):
    """SlamNode.cpp:40-58, ThreadLocalize.cpp:86-129 (constructor) and :424-432 (init); `robot` = _robotName incl. '/'"""
    pi = math.pi
    node = {                                             # SlamNode.cpp:40-58
        "robot_nbr": ("int", 1), "x_off_factor": ("double", 0.5), "y_off_factor": ("double", 0.5),
        "x_offset": ("double", 0.0), "y_offset": ("double", 0.0), "map_size": ("int", 10), "cellsize": ("double", 0.025),
        "truncation_radius": ("int", 3), "occ_grid_time_interval": ("double", 2.0), "tf_map_frame": ("string", "map"),
    }
    ctor = {                                             # ThreadLocalize.cpp:86-129 (constants ThreadLocalize.h:58-70)
        robot + "dist_filter_max": ("double", 1.0), robot + "dist_filter_min": ("double", 0.1),
        robot + "icp_iterations": ("int", 25),
        robot + "tf_laser_frame": ("string", robot + "laser"), robot + "tf_odom_frame": ("string", robot + "odom"),
        robot + "tf_footprint_frame": ("string", robot + "base_footprint"),
        "reg_trs_max": ("double", 0.25), "reg_sin_rot_max": ("double", 0.17), "max_velocity_lin": ("double", 1.5),
        "max_velocity_rot": ("double", 2 * pi), "ude_odom_rescue": ("bool", False), "wait_for_odom_tf": ("double", 1.0),
        "laser_min_range": ("double", 0.0), "trials": ("int", 100), "sizeControlSet": ("int", 140),
        "epsThresh": ("double", 0.15), "zhit": ("double", 0.45), "zphi": ("double", 0.0), "zshort": ("double", 0.25),
        "zmax": ("double", 0.05), "zrand": ("double", 0.25), "percentagePointsInC": ("double", 0.9),
        "rangemax": ("double", 20.0), "sigphi": ("double", pi / 180.0 * 3), "sighit": ("double", 0.2),
        "lamshort": ("double", 0.08), "maxAngleDiff": ("double", 3.0), "maxAnglePenalty": ("double", 0.5),
        robot + "ransac_trials": ("int", 50), robot + "ransac_eps_thresh": ("double", 0.15),
        robot + "ransac_ctrlset_size": ("int", 180), robot + "ransac_phi_max": ("double", 30.0),
        robot + "registration_mode": ("int", 0),
    }
    ns = "tsd_slam/" + robot
    init = {                                             # ThreadLocalize.cpp:424-432
        ns + "local_offset_x": ("double", 0.0), ns + "local_offset_y": ("double", 0.0), ns + "local_offset_yaw": ("double", 0.0),
        ns + "max_range": ("double", 30.0), ns + "min_range": ("double", 0.001), ns + "low_reflectivity_range": ("double", 2.0),
        ns + "footprint_width": ("double", 1.0), ns + "footprint_height": ("double", 1.0), ns + "footprint_x_offset": ("double", 0.28),
    }
    return node, ctor, init


def test_declared_parameters_equal_the_references():
    # the facade's own additions (documented in INTEGRATION.md), everything else must be the reference's
    additions = {"icp_estimator", "tsdpdf_seed", "async_mapping", "pub_tsd_color_map", "object_inflation_factor", "use_object_inflation"}   # (the last three: ThreadGrid.cpp:42-47)
    got = facade.declared_parameters()
    node, ctor, init = _reference_parameter_table("")
    want = {**node, **ctor, **init}
    assert {k for k in got if k.split("/")[-1] not in additions} == set(want), set(got) ^ set(want)
    for k, (t, v) in want.items():
        assert got[k][0] == t, (k, got[k], t)
        assert got[k][1] == v or (t == "double" and abs(got[k][1] - v) <= 1e-15), (k, got[k], v)
    # multi-robot: per-robot prefixes (SlamNode.cpp:101-122), tf frames default to <robot>/laser ... (ThreadLocalize.cpp:90-92)
    got = facade.declared_parameters({"robot_nbr": 2, "robot_0/name": "georg", "robot_1/name": "simon"})
    for r in ("georg/", "simon/"):
        _, ctor, init = _reference_parameter_table(r)
        for k, (t, v) in {**ctor, **init}.items():
            assert k in got and got[k][0] == t and got[k][1] == v, (k, got.get(k), (t, v))
    assert got["georg/tf_laser_frame"] == ("string", "georg/laser")


def test_shipped_single_laser_yaml_loads():
    """config/single-laser.yaml of the reference (values transcribed: registration_mode 3, map_size 10, ...): every key is a
    declared parameter of the right type, so the shipped YAML loads without touching an undeclared parameter."""
    yaml_values = {     # /root/reference/config/single-laser.yaml (ros__parameters), transcribed key by key
        "cellsize": 0.025, "epsThresh": 0.15, "lamshort": 0.08, "laser_min_range": 0.26, "map_size": 10, "maxAngleDiff": 1.5,
        "maxAnglePenalty": 0.5, "max_velocity_lin": 1.0, "max_velocity_rot": 6.283185307179586, "object_inflation_factor": 1,
        "occ_grid_time_interval": 2.0, "percentagePointsInC": 0.6, "pub_tsd_color_map": True, "rangemax": 30.0,
        "reg_sin_rot_max": 0.5, "reg_trs_max": 1.0, "robot_nbr": 1, "sighit": 0.1, "sigphi": 0.05235987755982989,
        "sizeControlSet": 360, "dist_filter_max": 0.4, "dist_filter_min": 0.02, "icp_iterations": 30, "ransac_phi_max": 45.0,
        "registration_mode": 3, "ransac_ctrlset_size": 180, "ransac_eps_thresh": 0.15, "ransac_trials": 50,
        "tf_laser_frame": "laser", "tf_map_frame": "map", "tf_odom_frame": "odom", "trials": 100, "truncation_radius": 3,
        "ude_odom_rescue": False, "use_object_inflation": False, "wait_for_odom_tf": 1.0, "x_off_factor": 0.5, "x_offset": 0.0,
        "y_off_factor": 0.5, "y_offset": 0.0, "zrand": 0.05, "zhit": 0.2, "zphi": 0.2, "zshort": 0.2, "zmax": 0.2,
    }   # (use_sim_time is rclcpp's own parameter)
    got = facade.declared_parameters(yaml_values)
    for k, v in yaml_values.items():
        assert k in got, k
        t = "bool" if isinstance(v, bool) else "int" if isinstance(v, int) else "string" if isinstance(v, str) else "double"
        assert got[k][0] == t, (k, got[k], t)
