"""Known-answer / property tests of the oracle itself (it is the checker of the GPU path, so it gets
independent checks of its own): a vectorised NumPy re-derivation of one push, analytic ray-cast and ICP
answers, idempotence / invariants of the tile state machine, and dump/load round trips."""
import math

import numpy as np
import pytest

from ohm_tsd_slam_amd import synth
from oracle import pyoracle as O
from tests import helpers as H


def numpy_push_from_empty(gc, geo, pose, data, mask, max_range, low_refl):
    """Independent NumPy restatement of the cell update of ONE push into an empty grid
    (TsdGrid.cpp:237-274, TsdGridPartition.h:170-212) for every cell of the grid, ignoring tile
    classification: returns tsd / weight the cell would get IF its tile is updated."""
    N, cs, maxT = gc.cells, gc.cell_size, max(gc.max_trunc, 2 * gc.cell_size)
    Pi = np.linalg.inv(pose)
    idx = np.arange(N)
    cx = ((idx + 0.5) * cs)[None, :].repeat(N, 0)
    cy = ((idx + 0.5) * cs)[:, None].repeat(N, 1)
    lx = Pi[0, 0] * cx + Pi[0, 1] * cy + Pi[0, 2]
    ly = Pi[1, 0] * cx + Pi[1, 1] * cy + Pi[1, 2]
    phi = np.arctan2(ly, lx)
    lower = geo.angle_min - 0.5 * geo.angle_increment
    upper = geo.angle_min + (geo.beams - 0.5) * geo.angle_increment
    beam = np.round((phi - geo.angle_min) / geo.angle_increment).astype(int)
    vis = (phi > lower) & (phi < upper)
    beam = np.clip(beam, 0, geo.beams - 1)
    dist = np.hypot(cx - pose[0, 2], cy - pose[1, 2])
    r = data[beam]
    ok = vis & (mask[beam] != 0)
    sd = np.where(np.isinf(r), np.where(dist < low_refl, maxT, -np.inf), r - dist)
    upd = ok & (sd >= -maxT)
    tsd = np.where(upd, np.minimum(sd / maxT, 1.0), np.nan)
    # partition weight from the tile centroid
    tx = (idx // 32 * 32 + 16.5) * cs
    dcen = np.hypot(tx[None, :] - pose[0, 2], tx[:, None] - pose[1, 2])
    pw = ((max_range - np.minimum(dcen, max_range)) / max_range) ** 2
    w = np.where(upd, 0.01 * pw, 0.0)
    return tsd, w, upd


@pytest.mark.parametrize("scene,map_log2,cs,geo", [
    ("room", 8, 0.1, synth.ScanGeometry.full_circle_360()),
    ("pillars", 9, 0.1, synth.ScanGeometry.utm30lx()),
])
def test_push_against_numpy_rederivation(scene, map_log2, cs, geo):
    gc = synth.GridConfig(map_log2, cs)
    world = synth.World(scene, gc)
    pose, (x, y, yaw) = H.sensor_pose(world, 0)
    data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    st = g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    init, iw, tsd, w = g.dump()
    e_tsd, e_w, e_upd = numpy_push_from_empty(gc, geo, pose, data, mask, 30.0, 2.0)
    PX = gc.cells // 32
    n_upd = 0
    for p in np.nonzero(init)[0]:
        py, px = divmod(p, PX)
        t = tsd[p].reshape(33, 33)[:32, :32]
        ww = w[p].reshape(33, 33)[:32, :32]
        et = e_tsd[py * 32:(py + 1) * 32, px * 32:(px + 1) * 32]
        ew = e_w[py * 32:(py + 1) * 32, px * 32:(px + 1) * 32]
        assert np.array_equal(np.isnan(t), np.isnan(et))
        m = ~np.isnan(t)
        assert np.allclose(t[m], et[m], rtol=0, atol=1e-12)
        assert np.allclose(ww, ew, rtol=0, atol=1e-15)
        n_upd += int(m.sum())
    assert n_upd == st["cells_updated"] > 1000
    # tiles the classifier skipped or emptied hold no surface: every cell the re-derivation puts within
    # the truncation band of a finite reading lies in an initialised tile
    near = e_upd & (np.abs(e_tsd) < 1.0)
    tiles_of_near = np.unique((np.nonzero(near)[0] // 32) * PX + np.nonzero(near)[1] // 32)
    assert init[tiles_of_near].all()


def test_push_invariants():
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(25):
        pose, (x, y, yaw) = H.sensor_pose(world, k)
        data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
        st = g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
        assert st["tiles_update"] <= st["tiles_range_pass"] <= st["tiles_total"]
        assert st["cells_updated"] <= st["cells_visited"] == 1024 * st["tiles_update"]
    init, iw, tsd, w = g.dump()
    sel = init.astype(bool)
    assert (iw >= 0).all() and (iw <= 32).all()
    assert (w[sel] >= 0).all() and (w[sel] <= 32).all()
    t = tsd[sel]
    assert np.nanmax(t) <= 1.0 and np.nanmin(t) >= -1.0
    assert (w[sel][np.isnan(t)] == 0).all()          # NaN cells always carry weight 0
    # halo consistency where both neighbours are initialised (TsdGrid.cpp:385-424)
    PX = gc.cells // 32
    checked = 0
    for p in np.nonzero(init)[0]:
        py, px = divmod(p, PX)
        if px < PX - 1 and init[p + 1]:
            a = tsd[p].reshape(33, 33)[:32, 32]
            b = tsd[p + 1].reshape(33, 33)[:32, 0]
            assert np.array_equal(a, b, equal_nan=True)
            checked += 1
        if py < PX - 1 and init[p + PX]:
            a = tsd[p].reshape(33, 33)[32, :32]
            b = tsd[p + PX].reshape(33, 33)[0, :32]
            assert np.array_equal(a, b, equal_nan=True)
    assert checked > 10


def test_dump_load_roundtrip():
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    pose, (x, y, yaw) = H.sensor_pose(world, 0)
    data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
    g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    d1 = g.dump()
    g2 = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    g2.load(*d1)
    d2 = g2.dump()
    for a, b in zip(d1, d2):
        assert np.array_equal(a, b, equal_nan=True)
    s1 = g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    s2 = g2.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    assert s1 == s2
    for a, b in zip(g.dump(), g2.dump()):
        assert np.array_equal(a, b, equal_nan=True)


def test_raycast_recovers_walls():
    """After a few pushes of the box room, the ray-cast model points lie on the walls (within a cell)."""
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(3):
        pose, (x, y, yaw) = H.sensor_pose(world, k)
        data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    pose, (x, y, yaw) = H.sensor_pose(world, 1)
    rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
    coords, normals, m, n = g.raycast(pose, rw, 0.001, 30.0)
    assert n > 0.9 * geo.beams
    truth = world.scan(x, y, yaw, geo).astype(float)
    pts = coords.reshape(-1, 2)[m.astype(bool)]
    rng_model = np.hypot(pts[:, 0], pts[:, 1])          # sensor frame => range of the model point
    assert np.max(np.abs(rng_model - truth[m.astype(bool)])) < 1.5 * gc.cell_size
    nn = normals.reshape(-1, 2)[m.astype(bool)]
    assert np.allclose(np.hypot(nn[:, 0], nn[:, 1]), 1.0, atol=1e-9)
    # model points sit on their own beam: angle of the point == beam angle
    ang = np.arctan2(pts[:, 1], pts[:, 0])
    beam_ang = geo.angle_min + np.nonzero(m)[0] * geo.angle_increment
    d = (ang - beam_ang + np.pi) % (2 * np.pi) - np.pi
    assert np.max(np.abs(d)) < 1e-9


@pytest.mark.parametrize("nn_mode", [0, 1])
def test_icp_recovers_known_transform(nn_mode):
    # well separated points (min spacing 0.3 m >> displacement): every nearest neighbour is the true
    # correspondence from the first step, so the closed form must recover the transform exactly
    rng = np.random.default_rng(42)
    pts = []
    while len(pts) < 300:
        p = rng.uniform(2, 14, 2)
        if all(np.hypot(*(p - q)) > 0.3 for q in pts):
            pts.append(p)
    model = np.array(pts)
    th, tx, ty = 0.004, 0.03, -0.02
    R = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
    scene = (model - np.array([tx, ty])) @ R      # scene = R^-1 (model - t)  =>  T maps scene -> model
    pose = np.eye(3)
    r = O.icp(model, scene, pose, 30, 0.4, 0.02, (0, 100, 0, 100), nn_mode=nn_mode, trace=True)
    assert r["iterations"] == 30 and r["state"] == 3 and r["pairs"] == 300
    T = r["T"]
    assert abs(math.atan2(T[1, 0], T[0, 0]) - th) < 1e-12
    assert abs(T[0, 2] - tx) < 1e-11 and abs(T[1, 2] - ty) < 1e-11
    assert r["rms"] < 1e-20
    tr = r["trace"]
    assert (np.diff(tr[:, 2]) <= 0).all() and tr[-1, 2] >= 0.02 ** 2     # threshold schedule monotone, floored
    assert tr[0, 1] > 1e-4 and tr[1, 1] < 1e-20                          # converged after the first step


def test_icp_point_to_line_estimator():
    """PointToLine2DEstimator (PointToLineEstimator2D.cpp:52-157), the reference's other estimator: on two
    perpendicular walls with exact normals the linearised point-to-line step recovers a small rigid motion to
    rounding once the pairs are right, sliding along a wall is free (the closed form keeps a residual there), and
    its "rms" is the mean |n.(s - m)|, not a mean square."""
    xs = np.linspace(-3.0, 3.0, 200)
    model = np.concatenate([np.stack([xs, np.full_like(xs, 2.0)], 1), np.stack([np.full_like(xs, 3.0), 0.6 * xs], 1)])
    normals = np.concatenate([np.tile([0.0, -1.0], (200, 1)), np.tile([-1.0, 0.0], (200, 1))])
    th = 0.02
    R = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
    scene = model @ R.T + np.array([0.03, -0.02])
    pose = np.array([[1.0, 0, 10.0], [0, 1.0, 10.0], [0, 0, 1.0]])
    b = (0.0, 25.6, 0.0, 25.6)
    cf = O.icp(model, scene, pose, 30, 0.4, 0.02, b)
    pl = O.icp(model, scene, pose, 30, 0.4, 0.02, b, model_normals_xy=normals, trace=True)
    assert pl["state"] == 3 and pl["iterations"] == 30 and pl["pairs"] >= 398
    assert abs(math.asin(pl["T"][1, 0]) + th) < 1e-12            # T maps the scene back onto the model
    assert pl["rms"] < 1e-12 and cf["rms"] > 1e-6
    # first step: mean distance to the wall lines of the displaced scene, a few centimetres
    assert 0.01 < pl["trace"][0, 1] < 0.1
    # exact normals and correct pairs: every scene point ends up on its wall line
    moved = scene @ pl["T"][:2, :2].T + pl["T"][:2, 2]
    assert np.abs(moved[:200, 1] - 2.0).max() < 1e-9 and np.abs(moved[200:, 0] - 3.0).max() < 1e-9


def test_icp_brute_force_equals_kdtree():
    gc, geo, scene_name = synth.CONFIGS["cfg1"]
    rng = np.random.default_rng(9)
    model = rng.uniform(3, 9, (700, 2))
    scene = model[rng.integers(0, 700, 500)] + rng.normal(0, 0.04, (500, 2))
    a = O.icp(model, scene, np.eye(3), 30, 0.4, 0.02, (0, 100, 0, 100), nn_mode=0)
    b = O.icp(model, scene, np.eye(3), 30, 0.4, 0.02, (0, 100, 0, 100), nn_mode=1)
    assert a["pairs"] == b["pairs"] and np.array_equal(a["T"], b["T"]) and a["rms"] == b["rms"]


def test_color_image_colours():
    """grid2ColorImage (TsdGrid.cpp:429-488): green channel 255 with r = b = tsd * 255 in front of a surface, red
    (1 + tsd) * 255 behind it, white for tiles that are empty-with-weight, black for unseen space; pixel (w, h)
    shows the cell coord2Cell finds for the accumulated coordinate."""
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    pose = synth.pose_matrix(world.start[0], world.start[1], 0.1)
    far = np.full(geo.beams, 9.0, dtype=np.float32)
    for r32 in (world.scan(world.start[0], world.start[1], 0.1, geo), far):
        data, mask = O.ingest_f32(r32, 30.0, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    img = g.color_image()
    assert img.shape == (gc.cells, gc.cells, 3) and img.dtype == np.uint8
    green = img[..., 1] == 255
    assert green.any() and (img[green][:, 0] == img[green][:, 2]).all()
    red = (img[..., 1] == 0) & (img[..., 0] > 0)
    assert red.any() and (img[red][:, 2] == 0).all()
    assert (img.sum(axis=2) == 0).any()
    small = g.color_image(64, 48)
    assert small.shape == (48, 64, 3) and small.any()


def test_text_grid_file_roundtrip(tmp_path):
    """TsdGrid::storeGrid / TsdGrid(file) (TsdGrid.cpp:548-607, :25-110): header, tile identifiers, 6-digit values,
    halo not stored."""
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    pose = synth.pose_matrix(world.start[0], world.start[1], 0.1)
    far = np.full(geo.beams, 9.0, dtype=np.float32)
    for r32 in (world.scan(world.start[0], world.start[1], 0.1, geo), far):
        data, mask = O.ingest_f32(r32, 30.0, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    path = tmp_path / "a.grid"
    assert g.store_text(path)
    lines = path.read_text().split("\n")
    assert lines[:4] == ["0.1", "5", "8", "0.3"]
    flags, iw, tsd, w = g.dump()
    n_content, n_empty = int(flags.sum()), int(((flags == 0) & (iw > 0)).sum())
    assert len(lines) - 1 == 4 + g.tiles + n_empty + n_content * 2 * 32 * 32
    g2 = O.Grid.load_text(path, gc.cell_size)
    f2, iw2, tsd2, w2 = g2.dump()
    un = flags == 0                                           # (a content tile's _initWeight is not in the file)
    assert np.array_equal(flags, f2) and np.allclose(iw[un], iw2[un], rtol=1e-5)
    a = tsd.reshape(-1, 33, 33)[flags.astype(bool)][:, :32, :32]
    b = tsd2.reshape(-1, 33, 33)[flags.astype(bool)][:, :32, :32]
    m = ~np.isnan(a)
    assert np.array_equal(np.isnan(b), ~m) and np.allclose(a[m], b[m], rtol=1e-5, atol=1e-12)
    assert np.isnan(tsd2.reshape(-1, 33, 33)[flags.astype(bool)][:, 32, :]).all()
    path2 = tmp_path / "b.grid"
    g2.store_text(path2)
    assert path2.read_bytes() == path.read_bytes()            # storing what was loaded reproduces the file
    assert O.Grid.load_text(tmp_path / "missing.grid", gc.cell_size) is None


def test_occupancy_values_and_persistence():
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    content = np.full(gc.cells * gc.cells, -1, dtype=np.int8)
    for k in range(3):
        pose, (x, y, yaw) = H.sensor_pose(world, k)
        data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
        out, n = g.occupancy(content)
    assert set(np.unique(out)) <= {-1, 0, 100} and n > 50
    assert (out == 100).sum() > 50 and (out == 0).sum() > 500
    assert set(np.unique(content)) <= {-1, 0}           # the 100 marks never enter the persistent map


# ------------------------------------------------------------------------------------------------
# independent second derivations (numpy / plain Python from SURVEY Appendix A, written against the canonical 33 x 33
# dump, not against oracle/tsd_oracle.c): ray march and both estimators, as the push already has one above
def _np_bilinear(gc, dump, x, y):
    """TsdGrid::interpolateBilinear + coord2Cell (Appendix A.5) -> (status, tsd); status 0 success, 1 invalid index,
    2 empty partition, 3 nan"""
    init, _, tsd, _ = dump
    cs, N, PX = gc.cell_size, gc.cells, gc.cells // 32
    inv = 1.0 / cs
    xi, yi = math.floor(x * inv), math.floor(y * inv)
    dx, dy = (xi + 0.5) * cs, (yi + 0.5) * cs
    if x < dx:
        xi -= 1; dx -= cs
    if y < dy:
        yi -= 1; dy -= cs
    if xi >= N or xi < 0 or yi >= N or yi < 0:
        return 1, 0.0
    p = (yi // 32) * PX + xi // 32
    if not init[p]:
        return 2, 0.0
    lx, ly = xi % 32, yi % 32
    wx, wy = abs((x - dx) * inv), abs((y - dy) * inv)
    t = tsd[p].reshape(33, 33)
    v = t[ly, lx] * (1. - wy) * (1. - wx) + t[ly + 1, lx] * wy * (1. - wx) + t[ly, lx + 1] * (1. - wy) * wx + t[ly + 1, lx + 1] * wy * wx
    return (3, v) if math.isnan(v) else (0, float(v))


def _np_raycast_beam(gc, dump, tr, ray, min_range, max_range):
    """RayCastPolar2D::rayCastFromCurrentView (Appendix A.5) for one beam -> (hit, cx, cy)"""
    cs, N = gc.cell_size, gc.cells
    maxc = (N + 0.5) * cs
    inside = 0 < tr[0] < maxc and 0 < tr[1] < maxc
    gmin, gmax = (-10e9, 10e9) if inside else (10e9, -10e9)
    lim = (N - 1) * cs
    xmin = (((0.0 if ray[0] > 0 else lim) - tr[0]) / ray[0]) if abs(ray[0]) > 10e-6 else gmin
    ymin = (((0.0 if ray[1] > 0 else lim) - tr[1]) / ray[1]) if abs(ray[1]) > 10e-6 else gmin
    xmax = (((lim if ray[0] > 0 else 0.0) - tr[0]) / ray[0]) if abs(ray[0]) > 10e-6 else gmax
    ymax = (((lim if ray[1] > 0 else 0.0) - tr[1]) / ray[1]) if abs(ray[1]) > 10e-6 else gmax
    imin = max(max(xmin, ymin), 0.0)
    imax = min(xmax, ymax)
    imin = max(imin, min_range / cs)
    imax = min(imax, max_range / cs)
    if imin >= imax:
        return False, 0.0, 0.0
    i = imin
    while i < imax:                                  # coarse skip over empty / outside partitions, 32 cells at a time
        st, _ = _np_bilinear(gc, dump, tr[0] + i * ray[0], tr[1] + i * ray[1])
        if st not in (2, 1):
            break
        imin = i
        i += 32.0
    px, py = tr[0] + imin * ray[0], tr[1] + imin * ray[1]
    st, v = _np_bilinear(gc, dump, px, py)
    prev = v if st == 0 else float("nan")
    i = imin
    while i <= imax:
        px += ray[0]; py += ray[1]                   # repeated addition
        st, v = _np_bilinear(gc, dump, px, py)
        if st != 0:
            prev = float("nan")
        else:
            if prev > 0 and v < 0:
                interp = prev / (prev - v)
                return True, px + ray[0] * (interp - 1.0), py + ray[1] * (interp - 1.0)
            if prev < 0 and v > 0:
                return False, 0.0, 0.0
            prev = v
        i += 1.0
    return False, 0.0, 0.0


def test_raycast_against_python_rederivation():
    gc = synth.GridConfig(8, 0.1)
    geo = synth.ScanGeometry(181, math.radians(-90.0), math.radians(1.0))
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(3):
        pose, (x, y, yaw) = H.sensor_pose(world, 4 * k)
        data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    dump = g.dump()
    pose, _ = H.sensor_pose(world, 5)
    rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
    co, no, mo, cnt = g.raycast(pose, rw, 0.001, 30.0)
    Pi = np.linalg.inv(pose)
    hits = 0
    for b in range(geo.beams):
        hit, cx, cy = _np_raycast_beam(gc, dump, (pose[0, 2], pose[1, 2]), (rw[b], rw[geo.beams + b]), 0.001, 30.0)
        if hit:
            # interpolateNormal (4 look-ups at +-cellSize, all must succeed) decides whether the hit is reported
            vals = [_np_bilinear(gc, dump, cx + dx, cy + dy) for dx, dy in ((gc.cell_size, 0), (-gc.cell_size, 0), (0, gc.cell_size), (0, -gc.cell_size))]
            if any(st != 0 for st, _ in vals):
                hit = False
            else:
                n = np.array([vals[0][1] - vals[1][1], vals[2][1] - vals[3][1]])
                ln = math.sqrt(n[0] * n[0] + n[1] * n[1])
                if abs(ln) > 10e-6:
                    n = n / ln
        assert bool(mo[b]) == hit, f"beam {b}"
        if hit:
            hits += 1
            m = Pi @ np.array([cx, cy, 1.0])
            nn = Pi[:2, :2] @ n
            assert abs(m[0] - co[2 * b]) <= 1e-12 and abs(m[1] - co[2 * b + 1]) <= 1e-12, f"beam {b}"
            assert abs(nn[0] - no[2 * b]) <= 1e-12 and abs(nn[1] - no[2 * b + 1]) <= 1e-12, f"beam {b}"
    assert hits == cnt and hits > 0.8 * geo.beams


def test_estimators_against_numpy_rederivation():
    """Both estimators from the pair list of a step (ora_icp_pairs) by numpy: ClosedFormEstimator2D (Appendix A.6) and
    PointToLine2DEstimator (normal equations by np.linalg.solve), against Tlast of the oracle's first iteration (trace)."""
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()
    world = synth.World("room", gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(3):
        pose, (x, y, yaw) = H.sensor_pose(world, k)
        data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), 30.0, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
    pose, (x, y, yaw) = H.sensor_pose(world, 1)
    rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
    co, no, mo, cnt = g.raycast(pose, rw, 0.001, 30.0)
    data, mask = O.ingest_f32(world.scan(x + 0.05, y - 0.03, yaw + 0.012, geo), 30.0, geo.angle_increment)
    sc, ms, _ = O.scene_from_scan(rl, data, mask)
    M = co.reshape(-1, 2)[mo.astype(bool)]; Nn = no.reshape(-1, 2)[mo.astype(bool)]
    S = sc.reshape(-1, 2)[ms.astype(bool)]
    b = (0.0, g.max_x, 0.0, g.max_x)
    pm, ps, _ = O.icp_pairs(M, S, pose, 30, 0.4, 0.02, b, 0.4 ** 2)
    assert len(pm) > 100
    m, s = M[pm], S[ps]
    # closed form
    cm, csn = m.mean(0), s.mean(0)
    mse = np.mean(np.sum((s - m) ** 2, 1))
    mc, scn = m - cm, s - csn
    th = math.atan2(np.sum(mc[:, 1] * scn[:, 0] - mc[:, 0] * scn[:, 1]), np.sum(mc[:, 0] * scn[:, 0] + mc[:, 1] * scn[:, 1]))
    c, si = math.cos(th), math.sin(th)
    t = cm - np.array([c * csn[0] - si * csn[1], c * csn[1] + si * csn[0]])
    r = O.icp(M, S, pose, 30, 0.4, 0.02, b, trace=True)
    tl = r["trace"][0]
    assert int(tl[0]) == len(pm) and abs(tl[1] - mse) <= 1e-15
    assert np.max(np.abs(tl[4:8] - np.array([c, si, t[0], t[1]]))) <= 1e-13
    # point to line: A x = b over (a_z, n_x, n_y), x = (psi, tx, ty); rms = mean |n.(s - m)|
    n = Nn[pm]
    az = s[:, 0] * n[:, 1] - s[:, 1] * n[:, 0]
    J = np.stack([az, n[:, 0], n[:, 1]], 1)
    resid = np.sum((s - m) * n, 1)
    xsol = np.linalg.solve(J.T @ J, -(J.T @ resid))
    rp = O.icp(M, S, pose, 30, 0.4, 0.02, b, model_normals_xy=Nn, trace=True)
    tlp = rp["trace"][0]
    assert int(tlp[0]) == len(pm) and abs(tlp[1] - np.mean(np.abs(resid))) <= 1e-15
    assert np.max(np.abs(tlp[4:8] - np.array([math.cos(xsol[0]), math.sin(xsol[0]), xsol[1], xsol[2]]))) <= 1e-11


# ------------------------------------------------------------------------------------------------
# row N3 (TSD_PDF pre-registration): an independent derivation written from TSD_PDFMatching.cpp:31-294, RandomMatching.cpp:41-189
# and Matrix::pcaAnalysis (gsl/Matrix.cpp:227-326) -- NOT from oracle/tsd_oracle.c, whose restatement is the most inventive
# one of the oracle: it replaces gsl_linalg_SV_decomp_jacobi of the 2 x 2 scatter matrix by a closed-form eigen-decomposition and
# carries gsl_stats_mean's long-double running mean by hand.  Here the principal axes come from numpy's LAPACK SVD of the centred
# n x 2 window itself (no scatter matrix, no closed form) and the mean from np.longdouble.
def _np_pca_axes(A):
    """Matrix::pcaAnalysis for an n x 2 point set -> axes[2][4] = rows (x0, x1, y0, y1) of the long / the short principal axis"""
    n = len(A)
    cent = np.zeros(2)
    for j in range(2):                         # gsl_stats_mean: running mean in long double, rounded to double at the end
        m = np.longdouble(0.0)
        for i in range(n):
            m += (np.longdouble(A[i, j]) - m) / np.longdouble(i + 1)
        cent[j] = float(m)
    Mc = A - cent
    _, _, Vt = np.linalg.svd(Mc, full_matrices=True)          # rows of Vt = right singular vectors = eigenvectors of Mc^T Mc, descending
    V = Vt.T                                                   # (column signs are free: the normal is oriented afterwards)
    P = V.T @ Mc.T                                             # coordinates in the eigenvectors' system
    for i in range(2):
        mx, mn = P[i].max(), P[i].min()
        align = (mx + mn) / 2.0 if (mx - mn) > 1e-6 else 0.0
        for j in range(2):
            cent[j] += V[j, i] * align
    axes = np.zeros((2, 4))
    for i in range(2):
        ext = P[i].max() - P[i].min()
        for j in range(2):
            e = V[j, i] * ext / 2.0
            axes[i, 2 * j], axes[i, 2 * j + 1] = cent[j] - e, cent[j] + e
    return axes


def _np_calc_normals(X, mask_in, mask_out, sr):
    """RandomMatching::calcNormals (:94-157): PCA over the masked-in points of the window [i - sr, i + sr) -- asymmetric as written"""
    n = len(X)
    N = np.zeros((n, 2))
    mask_out[:sr] = False
    mask_out[n - sr:] = False
    for i in range(sr, n - sr):
        if not mask_in[i]:
            continue
        win = [i + j for j in range(-sr, sr) if mask_in[i + j]]
        if len(win) <= 3:
            mask_out[i] = False
            continue
        ax = _np_pca_axes(X[win])
        x_long, y_long = ax[0, 1] - ax[0, 0], ax[0, 3] - ax[0, 2]
        x_short, y_short = ax[1, 1] - ax[1, 0], ax[1, 3] - ax[1, 2]
        len_long, len_short = x_long * x_long + y_long * y_long, x_short * x_short + y_short * y_short
        if len_short > 1e-6 and len_long / len_short < 4.0:
            mask_out[i] = False
            continue
        ln = math.sqrt(len_short)
        sgn = 1.0 if (X[i, 0] * x_short + X[i, 1] * y_short) < 0.0 else -1.0
        N[i] = sgn * x_short / ln, sgn * y_short / ln
    return N


def np_tsdpdf_match(gc, dump, pose, M, mask_m, S, mask_s, trials, size_control_set, zrand, phi_max, resolution, d_sub, d_ctrl, d_trials):
    """TSD_PDFMatching::match (:31-294) with the three rand() streams as inputs -> dict(T, prob, idx, i, candidates)"""
    n, sr = len(M), 10 // 2
    out = dict(T=np.eye(3), prob=0.0, idx=-1, i=-1, candidates=0)
    if n < 3:
        return out
    m_m = np.array(mask_m, dtype=bool)
    m_mp = m_m.copy()
    NM = _np_calc_normals(M, m_m, m_mp, sr)
    phi_m = np.where(m_mp, np.arctan2(NM[:, 1], NM[:, 0]), -1e6)
    idx_m = [i for i in range(sr, n - sr) if m_mp[i]]
    m_s = np.array(mask_s, dtype=bool)
    m_sp = m_s.copy()
    probability = 180.0 / float(m_sp.sum())
    if probability < 0.99:                                                     # subsampleMask (:176-189): one rand() per beam
        thresh = int(1000.0 - min(max(probability, 0.0), 1.0) * 1000.0 + 0.5)
        m_sp[(np.asarray(d_sub[:n], dtype=np.int64) % 1000) < thresh] = False
    NS = _np_calc_normals(S, m_s, m_sp, sr)                                    # (maskIn = the mask BEFORE the sub-sampling)
    phi_s = np.where(m_sp, np.arctan2(NS[:, 1], NS[:, 0]), -1e6)
    idx_s = [i for i in range(sr, n - sr) if m_sp[i]]
    tmp, control = list(idx_s), []                                            # pickControlSet (:52-76): the r-th REMAINING index
    k = 0
    while len(control) < min(size_control_set, len(idx_s)):
        control.append(tmp.pop(int(d_ctrl[k]) % len(tmp)))
        k += 1
    C = S[control]
    if len(idx_s) < 3 or len(idx_m) < 3:
        return out
    n_trials = min(trials, len(idx_m))
    phi_max = min(phi_max, math.pi * 0.5)
    assert resolution > 1e-6
    span = min(int(math.floor(phi_max / resolution)), n)
    remaining = list(idx_m)
    best = 0.0
    for trial in range(n_trials):
        idx = remaining.pop(int(d_trials[trial]) % len(remaining))
        for i in range(max(idx - span, sr), min(idx + span, n - sr)):
            if not m_sp[i]:
                continue
            phi = phi_m[idx] - phi_s[i]
            if phi > math.pi:
                phi -= 2.0 * math.pi
            elif phi < -math.pi:
                phi += 2.0 * math.pi
            if not abs(phi) < phi_max:
                continue
            out["candidates"] += 1
            c, s = math.cos(phi), math.sin(phi)
            T = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
            T[0, 2] = M[idx, 0] - (T[0, 0] * S[i, 0] + T[0, 1] * S[i, 1])
            T[1, 2] = M[idx, 1] - (T[1, 0] * S[i, 0] + T[1, 1] * S[i, 1])
            TMap = np.asarray(pose) @ T
            prob = 1.0
            for q in C:                                                       # the product in control-set order
                wx = TMap[0, 0] * q[0] + TMap[0, 1] * q[1] + TMap[0, 2]
                wy = TMap[1, 0] * q[0] + TMap[1, 1] * q[1] + TMap[1, 2]
                st, tsd = _np_bilinear(gc, dump, wx, wy)
                prob *= (1.0 - (1.0 - zrand) * abs(tsd)) if st == 0 else zrand
            if prob > best:                                                   # strict: the first best of a serial run
                best = prob
                out.update(T=T, prob=prob, idx=idx, i=i)
    return out


def _np_T_tolerance(M, mask_m, S, mask_s, idx, i, sr=5):
    """How far two CORRECT evaluations of the winner's T may differ.  pcaAnalysis hands an axis back as its two end points
    cent -/+ e and calcNormals forms the normal from their difference, (cent + e) - (cent - e): each component carries an absolute
    rounding error of ~ulp(cent), so the normal's direction is only known to ~ulp(|cent|) / |short axis|.  On a synthetic flat wall
    the short axis of a scene window is the fp32 range rounding (1e-7 .. 1e-6 m at a few metres): 1e-10 .. 1e-9 rad, whoever
    computes it (gsl's Jacobi, the oracle's closed form, LAPACK: they differ in the last bit of V).  T's rotation inherits that
    angle, its translation that angle times |s_i|.  For windows with real extent (1e-3 m) the bound is 1e-12."""
    worst = 0.0
    for X, mask, c in ((M, np.asarray(mask_m, dtype=bool), idx), (S, np.asarray(mask_s, dtype=bool), i)):
        win = [c + j for j in range(-sr, sr) if mask[c + j]]
        ax = _np_pca_axes(X[win])
        short = math.hypot(ax[1, 1] - ax[1, 0], ax[1, 3] - ax[1, 2])
        worst = max(worst, 4.0 * np.spacing(np.abs(X[win]).max()) / short)
    return 1e-12 + worst * max(1.0, float(np.hypot(*S[i])))


def _tsdpdf_case(cfg, seed, n_push, k_scene, trials, size_control_set, phi_max_deg):
    gc, geo, scene = synth.CONFIGS[cfg]
    world = synth.World(scene, gc)
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(n_push):
        pose, (x, y, yaw) = H.sensor_pose(world, k)
        data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), H.MAX_RANGE, geo.angle_increment)
        g.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
    pose, _ = H.sensor_pose(world, n_push - 1)
    rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
    co, no, mo, cnt = g.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    _, (x, y, yaw) = H.sensor_pose(world, k_scene)                            # the scene: a scan from a later pose
    data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), H.MAX_RANGE, geo.angle_increment)
    sc, ms, _ = O.scene_from_scan(rl, data, mask)
    rng = np.random.default_rng(seed)
    draws = tuple(rng.integers(0, 2 ** 31 - 1, m).astype(np.int32) for m in (geo.beams, max(size_control_set, 1), max(trials, 1)))
    args = (trials, size_control_set, 0.05, math.radians(phi_max_deg), geo.angle_increment) + draws
    return gc, g, pose, co, mo, sc, ms, args


@pytest.mark.parametrize("cfg,seed,n_push,k_scene,trials,ctrl,phi", [
    ("cfg1", 1, 4, 5, 100, 140, 30.0),       # the node's defaults (ThreadLocalize.cpp:105-129)
    ("cfg1", 2, 3, 6, 50, 180, 30.0),        # config/single-laser.yaml's ransac_* values
    ("cfg1", 3, 4, 4, 20, 40, 60.0),
    ("cfg2", 4, 3, 4, 30, 140, 30.0),        # 1081 beams: the sub-sampling branch (180 / valid < 0.99)
])
def test_tsdpdf_match_against_numpy_rederivation(cfg, seed, n_push, k_scene, trials, ctrl, phi):
    """ora_tsdpdf_match against the derivation above -- winner (idx, i) and candidate count exact, probability
    <= 1e-9 relative, T within what the reference's own normal formulation determines (_np_T_tolerance: 1e-12 for windows with
    real extent, up to ~1e-9 on flat synthetic walls); also on the committed draws of tests/golden/oracle_tsdpdf.npz (next test)."""
    O.build()
    gc, g, pose, co, mo, sc, ms, args = _tsdpdf_case(cfg, seed, n_push, k_scene, trials, ctrl, phi)
    m = O.tsdpdf_match(g, pose, co, mo, sc, ms, *args)
    ref = np_tsdpdf_match(gc, g.dump(), pose, co.reshape(-1, 2), mo, sc.reshape(-1, 2), ms, *args)
    assert ref["candidates"] > 50 and ref["idx"] >= 0, "the case must have candidates to mean anything"
    assert (m["candidates"], m["idx"], m["i"]) == (ref["candidates"], ref["idx"], ref["i"])
    tol = _np_T_tolerance(co.reshape(-1, 2), mo, sc.reshape(-1, 2), ms, ref["idx"], ref["i"])
    assert tol < 1e-6 and np.max(np.abs(m["T"] - ref["T"])) <= tol
    # (a T that moves the control points by `tol` changes each of the <= ctrl factors by up to tol / maxTruncation, relative)
    assert abs(m["prob"] - ref["prob"]) <= (1e-9 + ctrl * tol / gc.max_trunc) * ref["prob"]
    g.close()


def test_tsdpdf_fixture_against_numpy_rederivation():
    """the committed draws (tests/golden/oracle_tsdpdf.npz): the fixture's winner / counts / T / probability re-derived in numpy"""
    import os
    O.build()
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_tsdpdf.npz"))
    gc = synth.GridConfig(int(f["map_size_log2"]), float(f["cell_size"]))
    assert abs(gc.max_trunc - float(f["max_trunc"])) < 1e-15
    g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    res, phi0 = float(f["angle_increment"]), float(f["angle_min"])
    for k in range(len(f["push_poses"])):
        data, mask = O.ingest_f32(f["push_scans"][k], H.MAX_RANGE, res)
        g.push(f["push_poses"][k], data, mask, res, phi0, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
    ref = np_tsdpdf_match(gc, g.dump(), f["pose"], f["rc_coords"].reshape(-1, 2), f["rc_mask"], f["scene"].reshape(-1, 2), f["scene_mask"],
                          int(f["trials"]), int(f["size_control_set"]), float(f["zrand"]), float(f["phi_max"]), res,
                          f["draws_sub"], f["draws_ctrl"], f["draws_trials"])
    assert [ref["candidates"], ref["idx"], ref["i"]] == list(f["match_counts"])
    tol = _np_T_tolerance(f["rc_coords"].reshape(-1, 2), f["rc_mask"], f["scene"].reshape(-1, 2), f["scene_mask"], ref["idx"], ref["i"])
    assert tol < 1e-6 and np.max(np.abs(ref["T"] - f["match_T"])) <= tol       # (3e-10 here: a 4e-7 m short axis at 4.4 m, see _np_T_tolerance)
    assert abs(ref["prob"] - float(f["match_prob"])) <= (1e-9 + int(f["size_control_set"]) * tol / gc.max_trunc) * float(f["match_prob"])
    g.close()


def test_oracle_against_rederivations_on_random_cases():
    """tools/fuzz_oracle.py: the NumPy / pure-Python re-derivations above (one push for every cell, the ray march with normals) against
    the oracle on RANDOM grids, scenes, scanners (any field of view, any start angle), poses and spoiled scans -- 5 000 such cases
    (65 M cells, 300 k beams) ran clean (profiles/r4_fuzz_parity.txt); a short run here."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_oracle.py"), "60", "4711"], cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 60 cases ok" in p.stdout


# ------------------------------------------------------------------------------------------------
# k_pdf_prepare's pick decoding (csrc/tsdpdf.hip): "pick k erases the r_k-th remaining element" (RandomMatching.cpp:52-80,
# TSD_PDFMatching.cpp:185-199) decoded for all picks at once by merging sorted blocks.  The same index arithmetic, statement by
# statement, against the erase loop itself.
def _decode_picks_like_the_kernel(ranks):
    K = len(ranks)
    src = [(r << 16) | k for k, r in enumerate(ranks)]
    B = 1
    while B < K:
        dst = [None] * K
        for k in range(K):
            base = k & ~(2 * B - 1)
            i = k - base
            nL = min(K - base, B)
            nR = min(K - base - B, B)
            v = src[k]
            if nR <= 0:
                dst[k] = v
                continue
            val = v >> 16
            right = i >= B
            sib = base + (0 if right else B)
            n, thresh = (nL, val + 1) if right else (nR, val - i)
            def in_prefix(m):
                mc = min(m, n - 1)
                return m < n and (src[sib + mc] >> 16) - (mc if right else 0) < thresh
            pos, step = 0, B
            while step >= 2:                      # two steps of the descent per LDS round trip, like the kernel
                h = step >> 1
                p1 = in_prefix(pos + step - 1)
                p2 = in_prefix(pos + step + h - 1) if p1 else in_prefix(pos + h - 1)
                pos += (step if p1 else 0) + (h if p2 else 0)
                step >>= 2
            if step == 1 and in_prefix(pos):
                pos += 1
            at = base + (i - B) + pos if right else base + i + pos
            assert dst[at] is None
            dst[at] = (((val + pos) << 16) | (v & 0xFFFF)) if right else v
        src = dst
        B <<= 1
    out = [0] * K
    for v in src:
        out[v & 0xFFFF] = v >> 16
    return out


def test_pick_decoding_equals_the_erase_loop():
    rng = np.random.default_rng(7)
    cases = [(1, 1), (2, 2), (3, 3), (64, 64), (65, 65), (1081, 140), (1081, 100), (180, 180), (4096, 1024), (4096, 512), (257, 129)]
    cases += [(int(c), int(rng.integers(1, c + 1))) for c in rng.integers(1, 600, size=300)]
    for count, K in cases:
        for mode in range(3):
            if mode == 0:
                ranks = [int(rng.integers(0, count - k)) for k in range(K)]
            elif mode == 1:
                ranks = [0] * K                                   # always the first remaining element
            else:
                ranks = [count - k - 1 for k in range(K)]         # always the last
            remaining = list(range(count))
            want = [remaining.pop(r) for r in ranks]
            assert _decode_picks_like_the_kernel(ranks) == want, (count, K, mode)
