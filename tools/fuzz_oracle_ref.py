#!/usr/bin/env python3
"""Randomised check of the oracle's pair chain (exact NN -> DistanceFilter -> ReciprocalFilter over repeated determinePairs calls, row I4)
against the COMPILED REFERENCE (oracle/_ref/libtsd_ref.so: the reference's own PairAssignment.cpp, DistanceFilter.cpp, ReciprocalFilter.cpp
-- build container only, where /root/reference exists): random model / scene clouds, sizes, noise, out-of-bounds boxes, iteration counts
on both sides of the unsigned `icp_iterations - 10` wrap, filter distances, numbers of calls.  tests/test_cpu_oracle_ref.py has seven
fixed cases.  CPU only.  usage: python3 tools/fuzz_oracle_ref.py [cases] [first_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as O
import tests.test_cpu_oracle_ref as T

O.build()
if not O.ref_available():
    print("oracle/_ref is not built (needs /root/reference): nothing to do"); sys.exit(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
pairs_total = 0
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    nm, ns = int(rng.integers(3, 1200)), int(rng.integers(3, 1200))
    # (never 0.0: clouds() draws scene points WITH repetition from the model, and two identical scene points are an exact distance tie
    # for one model point -- which of the two the reference's ReciprocalFilter keeps is whatever its unstable std::sort leaves first
    # (ReciprocalFilter.cpp:16-21,58: the comparator orders by model index and distance only); the oracle and the HIP path keep the lower
    # scene index.  Seed 2 of the first run: the compiled reference kept the later one.  DESIGN 6.)
    noise = float(rng.choice([1e-6, 0.005, 0.05, 0.15, 0.4]))
    model, scene = T.clouds(seed, nm, ns, noise=noise)
    if rng.random() < 0.3:                                   # a lattice-free but clustered model: duplicates of model points
        model = np.concatenate([model, model[rng.integers(0, nm, max(1, nm // 8))]])
    iters = int(rng.choice([30, 25, 12, 11, 10, 9, 3, 1]))
    dmax = float(rng.choice([0.4, 0.4, 1.0, 0.2])); dmin = float(rng.choice([0.02, 0.1, 0.02]))
    if dmin > dmax: dmin = dmax / 4
    lo, hi = sorted(rng.uniform(1.0, 9.0, 2))
    bounds = (float(lo), float(hi) + 1.0, 0.0, 100.0) if rng.random() < 0.6 else (-1e9, 1e9, -1e9, 1e9)
    premask = ~((scene[:, 0] < bounds[0]) | (scene[:, 0] > bounds[1]) | (scene[:, 1] < bounds[2]) | (scene[:, 1] > bounds[3]))
    calls = int(rng.integers(1, 16))
    ref = T.ref_chain_run(model, scene, premask, iters, dmax, dmin, calls=calls)
    for nn_mode in (0, 1):
        ora = T.oracle_chain_run(model, scene, bounds, iters, dmax, dmin, calls=calls, nn_mode=nn_mode)
        for k, ((rm, rs), (om, os_)) in enumerate(zip(ref, ora)):
            if not (np.array_equal(rm, om) and np.array_equal(rs, os_)):
                print(f"MISMATCH seed {seed}: {len(model)} model / {ns} scene points, noise {noise}, iterations {iters}, filter {dmax} / {dmin}, call {k}, nn_mode {nn_mode}: "
                      f"reference {len(rm)} pairs, oracle {len(om)}")
                sys.exit(1)
    pairs_total += sum(len(r[0]) for r in ref)
    if case % 100 == 99:
        print(f"{case + 1} cases ok; {pairs_total} pairs compared; {time.time() - t0:.0f} s", flush=True)
# ---- the text line readers of the grid file format (obcore/base/tools.cpp:190-215, getDoubleLine / getIntLine) on random lines: numbers
# as %g writes them, integers beyond int range, special values, empty lines, garbage, CR LF
import tempfile
rng = np.random.default_rng(seed0)
alphabet = list("0123456789") * 4 + list("+-.eE") * 2 + list(" \tabxnifNAINF,;")
def rand_line():
    u = rng.random()
    if u < 0.35: return "%g" % rng.normal(0, 10.0 ** float(rng.integers(-12, 12)))
    if u < 0.5: return str(int(rng.integers(-2 ** 33, 2 ** 33)))
    if u < 0.55: return str(rng.choice(["nan", "inf", "-inf", "NaN", "INF", "", " ", "\r", "+", "-", ".", "e5", "1e", "1e+", "0x1p3", "1e400", "-1e400", "4.9e-324", "1.7976931348623157e308"]))
    return "".join(rng.choice(alphabet, int(rng.integers(0, 14)))) + ("\r" if rng.random() < 0.1 else "")
n_lines = 0
for it in range(max(1, n_cases // 20)):
    lines = [rand_line().replace("\n", "") for _ in range(200)]
    text = "\n".join(lines) + "\n"
    with tempfile.NamedTemporaryFile("wb", delete=False) as f:
        f.write(text.encode()); path = f.name
    for kind in (0, 1):
        k = np.full(len(lines), kind, dtype=np.int32)
        want, got = np.zeros(len(lines)), np.zeros(len(lines))
        O.ref().ref_text_lines(text.encode(), k.ctypes.data_as(O._ip), len(lines), O.d(want))
        assert O.lib().ora_text_lines(path.encode(), k.ctypes.data_as(O._ip), len(lines), O.d(got)) == 1
        if not np.array_equal(want, got, equal_nan=True):
            bad = [(l, w, g) for l, w, g in zip(lines, want, got) if not (w == g or (w != w and g != g))]
            print("MISMATCH text line reader kind", kind, bad[:5]); sys.exit(1)
    os.unlink(path); n_lines += 2 * len(lines)
print(f"all {n_cases} cases ok from seed {seed0}: {pairs_total} pairs compared, {n_lines} text lines read; {time.time() - t0:.0f} s")
