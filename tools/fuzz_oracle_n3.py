#!/usr/bin/env python3
"""Randomised check of the oracle's TSD_PDFMatching::match (row N3, the most inventive restatement in oracle/tsd_oracle.c) against its
independent NumPy derivation (tests/test_cpu_oracle_properties.py: LAPACK SVD of the centred window, long-double means, the sub-sampling,
the control-set pick, the scoring loop) on random draws, trial / control-set sizes, maps and scene poses -- the test suite has four such
cases.  CPU only.  usage: python3 tools/fuzz_oracle_n3.py [cases] [first_seed]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as O
import tests.test_cpu_oracle_properties as P

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
done = skipped = 0
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    cfg = "cfg2" if rng.random() < 0.15 else "cfg1"
    n_push = int(rng.integers(2, 6)); k_scene = n_push - 1 + int(rng.integers(0, 4))
    trials = int(rng.choice([10, 20, 50, 100, 150])); ctrl = int(rng.choice([20, 40, 100, 140, 180])); phi = float(rng.choice([15.0, 30.0, 60.0]))
    gc, g, pose, co, mo, sc, ms, args = P._tsdpdf_case(cfg, seed, n_push, k_scene, trials, ctrl, phi)
    m = O.tsdpdf_match(g, pose, co, mo, sc, ms, *args)
    ref = P.np_tsdpdf_match(gc, g.dump(), pose, co.reshape(-1, 2), mo, sc.reshape(-1, 2), ms, *args)
    tag = f"seed {seed}: {cfg}, {n_push} pushes, scene {k_scene}, {trials} trials, {ctrl} control points, phi_max {phi}"
    try:
        assert (m["candidates"], m["idx"], m["i"]) == (ref["candidates"], ref["idx"], ref["i"]), f"winner / count: oracle {(m['candidates'], m['idx'], m['i'])} numpy {(ref['candidates'], ref['idx'], ref['i'])}"
        if ref["idx"] >= 0:
            tol = P._np_T_tolerance(co.reshape(-1, 2), mo, sc.reshape(-1, 2), ms, ref["idx"], ref["i"])
            assert tol < 1e-6 and np.max(np.abs(m["T"] - ref["T"])) <= tol, f"T differs by {np.max(np.abs(m['T'] - ref['T']))} (tolerance {tol})"
            assert abs(m["prob"] - ref["prob"]) <= (1e-9 + ctrl * tol / gc.max_trunc) * ref["prob"], "probability"
            done += 1
        else:
            skipped += 1
    except AssertionError as e:
        print("MISMATCH", tag, "--", e)
        sys.exit(1)
    finally:
        g.close()
    if case % 20 == 19:
        print(f"{case + 1} cases ok ({tag}); with a winner {done}, without {skipped}; {time.time() - t0:.0f} s", flush=True)
print(f"all {n_cases} cases ok from seed {seed0}: with a winner {done}, without {skipped}; {time.time() - t0:.0f} s")
