#!/usr/bin/env python3
"""Randomised check of the ORACLE itself (CPU only, no GPU, test infrastructure): the independent NumPy / pure-Python re-derivations of
tests/test_cpu_oracle_properties.py -- one push into an empty grid for every cell, the ray march with bilinear look-ups and normals, the
closed-form and point-to-line estimators on the first step's pair list -- against oracle/tsd_oracle.c on random grids, scenes, scanners, poses and spoiled scans
instead of the two or three fixed cases of the test suite.  usage: python3 tools/fuzz_oracle.py [cases] [first_seed]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import synth
from oracle import pyoracle as O
from tests import helpers as H
import tests.test_cpu_oracle_properties as P

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
tot = dict(pushes=0, cells=0, beams=0, hits=0)
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    map_log2 = int(rng.choice([7, 8, 8, 9]))
    cs = float(rng.choice([0.05, 0.07, 0.1]))
    scene = str(rng.choice(["room", "pillars"]))
    if rng.random() < 0.5:
        geo = synth.ScanGeometry.full_circle_360() if rng.random() < 0.5 else synth.ScanGeometry.utm30lx()
    else:
        nb = int(rng.choice([91, 181, 361, 541]))
        fov = math.radians(float(rng.uniform(60.0, 340.0)))
        geo = synth.ScanGeometry(nb, float(rng.uniform(-math.pi, math.pi - fov)) if fov < 2 * math.pi - 0.2 else -0.5 * fov, fov / (nb - 1))
    gc = synth.GridConfig(map_log2, cs)
    world = synth.World(scene, gc)
    W = gc.cells * cs
    tag = f"seed {seed}: 2^{map_log2} cells @ {cs} m, {scene}, {geo.beams} beams from {geo.angle_min:.3f} rad"
    try:
        # ---- one push into an empty grid, every cell
        x = world.start[0] + rng.uniform(-0.1, 0.1) * W; y = world.start[1] + rng.uniform(-0.1, 0.1) * W; yaw = rng.uniform(-math.pi, math.pi)
        pose = synth.pose_matrix(x, y, yaw)
        r32 = world.scan(x, y, yaw, geo)
        if rng.random() < 0.5:
            for val in (0.0, np.nan, 45.0):
                r32[rng.integers(0, len(r32), rng.integers(0, max(2, len(r32) // 40)))] = val
        data, mask = O.ingest_f32(r32, 30.0, geo.angle_increment)
        g = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
        st = g.push(pose, data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
        init, iw, tsd, w = g.dump()
        e_tsd, e_w, e_upd = P.numpy_push_from_empty(gc, geo, pose, data, mask, 30.0, 2.0)
        PX = gc.cells // 32
        n_upd = 0; edge = 0
        for p in np.nonzero(init)[0]:
            py, px = divmod(p, PX)
            t = tsd[p].reshape(33, 33)[:32, :32]; ww = w[p].reshape(33, 33)[:32, :32]
            et = e_tsd[py * 32:(py + 1) * 32, px * 32:(px + 1) * 32]; ew = e_w[py * 32:(py + 1) * 32, px * 32:(px + 1) * 32]
            same = np.isnan(t) == np.isnan(et)
            edge += int((~same).sum())                 # (a cell whose signed distance is -maxTruncation to the last bit, or whose angle is a beam boundary)
            m = ~np.isnan(t) & ~np.isnan(et)
            assert np.allclose(t[m], et[m], rtol=0, atol=1e-12), "tsd values"
            assert np.allclose(ww[same], ew[same], rtol=0, atol=1e-15), "weights"
            n_upd += int((~np.isnan(t)).sum())
        assert edge <= 2, f"{edge} cells updated on one side only"
        assert n_upd == st["cells_updated"], "cells_updated"
        tot["pushes"] += 1; tot["cells"] += n_upd
        # ---- ray cast (after two more pushes), beam by beam in pure Python: a sample of the beams
        for k in range(2):
            x2 = x + rng.uniform(-0.3, 0.3); y2 = y + rng.uniform(-0.3, 0.3); yaw2 = yaw + rng.uniform(-0.2, 0.2)
            d2, m2 = O.ingest_f32(world.scan(x2, y2, yaw2, geo), 30.0, geo.angle_increment)
            g.push(synth.pose_matrix(x2, y2, yaw2), d2, m2, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0)
        dump = g.dump()
        xr = x + rng.uniform(-0.3, 0.3); yr = y + rng.uniform(-0.3, 0.3); yawr = yaw + rng.uniform(-0.3, 0.3)
        poser = synth.pose_matrix(xr, yr, yawr)
        rl, rw = H.world_rays(O, geo, poser, gc.cell_size)
        co, no, mo, cnt = g.raycast(poser, rw, 0.001, 30.0)
        Pi = np.linalg.inv(poser)
        for b in rng.choice(geo.beams, min(geo.beams, 60), replace=False):
            hit, cx, cy = P._np_raycast_beam(gc, dump, (poser[0, 2], poser[1, 2]), (rw[b], rw[geo.beams + b]), 0.001, 30.0)
            n = None
            if hit:
                vals = [P._np_bilinear(gc, dump, cx + dx, cy + dy) for dx, dy in ((cs, 0), (-cs, 0), (0, cs), (0, -cs))]
                if any(s_ != 0 for s_, _ in vals):
                    hit = False
                else:
                    n = np.array([vals[0][1] - vals[1][1], vals[2][1] - vals[3][1]])
                    ln = math.sqrt(n[0] * n[0] + n[1] * n[1])
                    if abs(ln) > 10e-6:
                        n = n / ln
            assert bool(mo[b]) == hit, f"beam {b}: hit {bool(mo[b])} / {hit}"
            if hit:
                m_ = Pi @ np.array([cx, cy, 1.0]); nn = Pi[:2, :2] @ n
                assert abs(m_[0] - co[2 * b]) <= 1e-11 and abs(m_[1] - co[2 * b + 1]) <= 1e-11, f"beam {b}: coordinates"
                assert abs(nn[0] - no[2 * b]) <= 1e-11 and abs(nn[1] - no[2 * b + 1]) <= 1e-11, f"beam {b}: normal"
                tot["hits"] += 1
            tot["beams"] += 1
        # ---- both estimators from the first step's pair list, by numpy: ClosedFormEstimator2D (centroids, atan2 of the centred sums) and
        # PointToLine2DEstimator (normal equations by np.linalg.solve) against Tlast of the oracle's first iteration
        M = co.reshape(-1, 2)[mo.astype(bool)]; Nn = no.reshape(-1, 2)[mo.astype(bool)]
        dxy = rng.uniform(-0.06, 0.06, 2); dya = rng.uniform(-0.02, 0.02)
        d3, m3 = O.ingest_f32(world.scan(xr + dxy[0], yr + dxy[1], yawr + dya, geo), 30.0, geo.angle_increment)
        sc, ms, _ = O.scene_from_scan(rl, d3, m3)
        S = sc.reshape(-1, 2)[ms.astype(bool)]
        bnd = (0.0, g.max_x, 0.0, g.max_x)
        if len(M) >= 10 and len(S) >= 10:
            pm, ps, _ = O.icp_pairs(M, S, poser, 30, 0.4, 0.02, bnd, 0.4 ** 2)
            if len(pm) >= 10:
                m, s_ = M[pm], S[ps]
                cm, csn = m.mean(0), s_.mean(0)
                mse = np.mean(np.sum((s_ - m) ** 2, 1))
                mc, scn = m - cm, s_ - csn
                th = math.atan2(np.sum(mc[:, 1] * scn[:, 0] - mc[:, 0] * scn[:, 1]), np.sum(mc[:, 0] * scn[:, 0] + mc[:, 1] * scn[:, 1]))
                c, si = math.cos(th), math.sin(th)
                t = cm - np.array([c * csn[0] - si * csn[1], c * csn[1] + si * csn[0]])
                tl = O.icp(M, S, poser, 30, 0.4, 0.02, bnd, trace=True)["trace"][0]
                assert int(tl[0]) == len(pm) and abs(tl[1] - mse) <= 1e-14, "closed form: pairs / mean squared distance"
                assert np.max(np.abs(tl[4:8] - np.array([c, si, t[0], t[1]]))) <= 1e-12, f"closed form: Tlast differs by {np.max(np.abs(tl[4:8] - np.array([c, si, t[0], t[1]])))}"
                n = Nn[pm]
                az = s_[:, 0] * n[:, 1] - s_[:, 1] * n[:, 0]
                J = np.stack([az, n[:, 0], n[:, 1]], 1)
                resid = np.sum((s_ - m) * n, 1)
                A = J.T @ J
                if np.linalg.cond(A) < 1e8:
                    xsol = np.linalg.solve(A, -(J.T @ resid))
                    tlp = O.icp(M, S, poser, 30, 0.4, 0.02, bnd, model_normals_xy=Nn, trace=True)["trace"][0]
                    assert int(tlp[0]) == len(pm) and abs(tlp[1] - np.mean(np.abs(resid))) <= 1e-14, "point to line: pairs / residual"
                    assert np.max(np.abs(tlp[4:8] - np.array([math.cos(xsol[0]), math.sin(xsol[0]), xsol[1], xsol[2]]))) <= 1e-9, "point to line: Tlast"
                tot["estimators"] = tot.get("estimators", 0) + 1
    except AssertionError as e:
        print("MISMATCH", tag, "--", e)
        sys.exit(1)
    if case % 20 == 19:
        print(f"{case + 1} cases ok ({tag}); {tot}; {time.time() - t0:.0f} s", flush=True)
print(f"all {n_cases} cases ok from seed {seed0}: {tot}; {time.time() - t0:.0f} s")
