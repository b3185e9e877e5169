#!/usr/bin/env python3
"""Randomised sweep of tsd_icp on SYNTHETIC point sets (not ray casts) against the oracle's exact-NN registration (test infrastructure:
uses oracle/): models of 3..2048 points -- lattices (exact distance ties everywhere: the lowest original model index must win), circles
and polylines around the sensor, random clouds, duplicated points, given in random ORDER; scenes = a rigidly moved subset + noise +
far outliers + points exactly between two lattice points; a few non-finite scene points; 1..30 iterations; sensor pose and bounds such
that the out-of-bounds filter cuts some of the scene.  pairs / iterations / state exact, T and rms 1e-9.
usage (GPU box): python3 tools/fuzz_icp.py [cases] [first_seed]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests import helpers as H

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_start = time.time()
gc = synth.GridConfig(8, 0.05)
dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
og = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
tot = dict(cases=0, lattice=0, tiny=0, big=0, notmatchable=0)
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["lattice", "circle", "polyline", "cloud"]))
    if kind == "lattice":
        h = float(rng.choice([0.125, 0.25, 0.5]))
        ex = float(rng.uniform(1.0, 4.0))
        xs, ys = np.meshgrid(np.arange(-ex, ex + 1e-9, h), np.arange(-0.7 * ex, 0.7 * ex + 1e-9, h))
        M = np.stack([xs.ravel(), ys.ravel()], axis=1)
        M = M[np.hypot(M[:, 0], M[:, 1]) > 0.3]
    elif kind == "circle":
        nm = int(rng.choice([3, 5, 13, 14, 40, 64, 65, 300, 1081, 2048]))
        a = np.sort(rng.uniform(-math.pi, math.pi, nm))
        r = rng.uniform(1.0, 6.0) * (1.0 + 0.05 * np.sin(5 * a))
        M = np.stack([r * np.cos(a), r * np.sin(a)], axis=1)
    elif kind == "polyline":
        nm = int(rng.integers(3, 1500))
        t = np.sort(rng.uniform(0, 1, nm))
        M = np.stack([-3 + 6 * t, 2.0 + 0.5 * np.sin(7 * t)], axis=1)
    else:
        nm = int(rng.integers(3, 800))
        M = rng.uniform(-4, 4, (nm, 2))
        M = M[np.hypot(M[:, 0], M[:, 1]) > 0.2]
    if len(M) > 2048:
        M = M[rng.choice(len(M), 2048, replace=False)]
    if len(M) < 3:
        continue
    if rng.random() < 0.2:                                  # duplicated model points (exact ties at distance zero apart)
        M = np.concatenate([M, M[rng.integers(0, len(M), max(1, len(M) // 10))]])[:2048]
    M = M[rng.permutation(len(M))]
    # scene: a moved subset + noise + outliers (+ mid-points of a lattice)
    ns = int(rng.integers(3, min(len(M), 1500) + 1))
    sub = M[rng.choice(len(M), ns, replace=(ns > len(M)))]
    th = rng.uniform(-0.05, 0.05); t = rng.uniform(-0.08, 0.08, 2)
    if rng.random() < 0.15:
        th = rng.uniform(-0.4, 0.4); t = rng.uniform(-0.5, 0.5, 2)
    Rm = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
    S = sub @ Rm.T + t
    if kind == "lattice" and rng.random() < 0.7:
        S = sub + np.array([h / 2, 0.0]) * (rng.random() < 0.5) + np.array([0.0, h / 2]) * (rng.random() < 0.5)      # exactly between lattice points
    elif rng.random() < 0.6:
        S = S + rng.normal(0, rng.choice([1e-4, 1e-3, 1e-2]), S.shape)
    if rng.random() < 0.3:
        S = np.concatenate([S, rng.uniform(-8, 8, (int(rng.integers(1, 40)), 2))])
    if rng.random() < 0.1:
        S[rng.integers(0, len(S), 2)] = [np.nan, np.inf][int(rng.integers(0, 2))]
    S = S[rng.permutation(len(S))][:2048]
    iters = int(rng.choice([1, 2, 3, 5, 11, 25, 30, 30]))
    if kind == "lattice":
        # One step only, and the pair LIST itself is compared below: from the second step on a lattice scene lands on exact ties again
        # and again, and the last bits of T (the estimators' own, inside the tolerance) then decide which of two equidistant model
        # points is the neighbour -- the oracle's scene off the lattice by 1e-17, the device's exactly on it: both right, different pairs
        iters = 1
    # the sensor's pose in the map: the out-of-bounds filter works on pose * scene
    px, py, pyaw = rng.uniform(1.0, 11.8), rng.uniform(1.0, 11.8), rng.uniform(-math.pi, math.pi)
    pose = synth.pose_matrix(px, py, pyaw)
    bounds = (0.0, og.max_x, 0.0, og.max_x)
    dmax, dmin = float(rng.choice([0.4, 0.4, 0.2, 1.0])), 0.02
    try:
        ro = O.icp(M, S, pose, iters, dmax, dmin, bounds, nn_mode=0)
        rd = dg.icp(M, S, pose, dg.icp_params(iters, dmax, dmin))
        # (a noise-free scene converges to a residual of EXACTLY 0.0 on one side and 1e-30 on the other -- the estimator's last bits --
        # and `rms <= maxRMS (0.0)` then labels the same final step SUCCESS here and MAXITERATIONS there: same T, same counts)
        # ... or ends the oracle's loop at that step (`rms <= 0.0`: SUCCESS after 3 iterations) while the device's runs on to the last one
        zero_rms = abs(ro["rms"]) < 1e-20 and abs(rd.rms) < 1e-20 and {ro["state"], rd.state} <= {3, 5}
        assert ro["pairs"] == rd.pairs and ((ro["iterations"], ro["state"]) == (rd.iterations, rd.state) or zero_rms), \
            f"oracle {(ro['pairs'], ro['iterations'], ro['state'])} hip {(rd.pairs, rd.iterations, rd.state)}"
        if kind == "lattice":
            pm, ps, _ = O.icp_pairs(M, S, pose, 1, dmax, dmin, bounds, dmax * dmax, nn_mode=0)
            hm, hs = dg.icp_pairs(M, S, pose, dg.icp_params(1, dmax, dmin), 1)[0]
            assert sorted(zip(ps.tolist(), pm.tolist())) == sorted(zip(hs.tolist(), hm.tolist())), "pair lists differ"
        d, a = H.pose_delta(ro["T"], rd.T)
        ok_T = (d <= 1e-9 and a <= 1e-9) or (not np.isfinite(ro["T"]).all() and not np.isfinite(rd.T).all())
        assert ok_T, f"|dT| {d} m {a} rad"
        assert abs(ro["rms"] - rd.rms) <= 1e-9 * max(1.0, abs(ro["rms"])) or (not np.isfinite(ro["rms"]) and not np.isfinite(rd.rms)), f"rms {ro['rms']} / {rd.rms}"
    except AssertionError as e:
        print(f"MISMATCH seed {seed}: {kind}, {len(M)} model / {len(S)} scene points, {iters} iterations, dist_filter_max {dmax} --", e)
        sys.exit(1)
    tot["cases"] += 1; tot["lattice"] += int(kind == "lattice"); tot["tiny"] += int(len(M) < 14); tot["big"] += int(len(M) > 1024)
    tot["notmatchable"] += int(rd.pairs <= 2)
    if case % 200 == 199:
        print(f"{case + 1} cases ok; {tot}; {time.time() - t_start:.0f} s", flush=True)
print(f"all {n_cases} cases ok from seed {seed0}: {tot}; {time.time() - t_start:.0f} s")
