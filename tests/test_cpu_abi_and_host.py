"""CPU-side checks: the C ABI library loads and exports every declared symbol (no compute calls without a
GPU), the product fails loudly without a device, and the C++ host logic of the facade (sensor model,
pose algebra, gates) equals the oracle bit for bit."""
import ctypes as C
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


def test_struct_layouts_match_header():
    assert C.sizeof(capi.PushStats) == 8 + 8 + 7 * 4 + 4      # padded to 8
    assert C.sizeof(capi.IcpParams) == 8 + 6 * 8               # iterations + estimator share the first 8 bytes
    assert capi.IcpParams.estimator.offset == 4 and capi.IcpParams.dist_filter_max.offset == 8
    assert C.sizeof(capi.IcpResult) == 9 * 8 + 8 + 6 * 4


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
