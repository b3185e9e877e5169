"""Shared helpers of the parity tests: drive the oracle and the HIP path with identical inputs."""
import math

import numpy as np

from ohm_tsd_slam_amd import synth

MAX_RANGE, MIN_RANGE, LOW_REFL = 30.0, 0.001, 2.0   # ThreadLocalize.cpp:427-429 defaults


def sensor_pose(world, k=0, step_x=0.06, step_yaw=0.01, yaw0=0.1):
    x = world.start[0] + step_x * k
    y = world.start[1]
    return synth.pose_matrix(x, y, yaw0 + step_yaw * k), (x, y, yaw0 + step_yaw * k)


def world_rays(oracle, geo, pose, cell_size):
    """Sensor::transform + getNormalizedRayMap(cellSize) on the host (oracle restatement, inputs only)."""
    rl = oracle.rays_local(geo.beams, geo.angle_min, geo.angle_increment)
    rw = oracle.rays_transform(pose, rl)
    rw = oracle.rays_rescale(rw, cell_size, 1.0)
    return rl, rw


def assert_grids_equal(o_dump, g_dump, tol=1e-5, exact_flags=True):
    oi, oiw, ot, ow = o_dump
    gi, giw, gt, gw = g_dump
    assert np.array_equal(oi, gi), f"tile flags differ at {np.nonzero(oi != gi)[0][:10]}"
    assert np.array_equal(oiw, giw), "initWeight differs"
    sel = oi.astype(bool)
    a, b = ot[sel], gt[sel]
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern of tsd differs"
    m = ~np.isnan(a)
    dt = np.max(np.abs(a[m] - b[m])) if m.any() else 0.0
    dw = np.max(np.abs(ow[sel] - gw[sel])) if sel.any() else 0.0
    assert dt <= tol, f"tsd max abs diff {dt}"
    assert dw <= tol, f"weight max abs diff {dw}"
    return dt, dw


def pose_delta(Ta, Tb):
    d = np.hypot(Ta[0, 2] - Tb[0, 2], Ta[1, 2] - Tb[1, 2])
    a = math.atan2(Ta[1, 0], Ta[0, 0]) - math.atan2(Tb[1, 0], Tb[0, 0])
    a = (a + math.pi) % (2 * math.pi) - math.pi
    return d, abs(a)
