"""Pins the oracle against the REAL reference where the reference builds in this image.

oracle/_ref/libtsd_ref.so is compiled from /root/reference's own PairAssignment.cpp, DistanceFilter.cpp,
ReciprocalFilter.cpp and mathbase.h (oracle/Makefile target _ref).  These are CPU tests (`-m "not gpu"`): they run in the build
container, where /root/reference exists and the library is built, and are skipped wherever the library is absent.  Nothing in
the `-m gpu` suite loads it (GPUTEST.native_so_loaded lists the product libraries and the oracle only): on the GPU box the
reference's outputs are represented by the committed fixture tests/golden/ref_chain_pairs.npz (generated from this library by
tests/golden/make_ref_chain_fixture.py), which the HIP path meets directly (test_hip_pair_chain_equals_compiled_reference_fixture).
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as O

needs_ref = pytest.mark.skipif(not O.ref_available() and not os.path.isdir("/root/reference/src"),
                               reason="oracle/_ref not built (needs /root/reference)")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def clouds(seed, n_model, n_scene, noise=0.05):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, 2 * np.pi, n_model))
    model = np.stack([5 + 3 * np.cos(t), 5 + 2 * np.sin(t)], axis=1) + rng.normal(0, 0.01, (n_model, 2))
    idx = rng.integers(0, n_model, n_scene)
    scene = model[idx] + rng.normal(0, noise, (n_scene, 2)) + np.array([0.03, -0.02])
    return model, scene


def ref_chain_run(model, scene, premask, iters, dmax, dmin, calls):
    R = O.ref()
    h = R.ref_chain_create(dmax, dmin, iters)
    R.ref_chain_reset(h)
    m, s = O.f64(model).reshape(-1), O.f64(scene).reshape(-1)
    pm = np.zeros(len(scene) + 1, dtype=np.int32)
    ps = np.zeros(len(scene) + 1, dtype=np.int32)
    out = []
    mk = np.ascontiguousarray(premask, dtype=np.uint8)
    for _ in range(calls):
        n = R.ref_chain_pairs(h, O.d(m), len(model), O.d(s), len(scene), O.u8(mk),
                              pm.ctypes.data_as(C.POINTER(C.c_int)), ps.ctypes.data_as(C.POINTER(C.c_int)))
        out.append((pm[:n].copy(), ps[:n].copy()))
    R.ref_chain_destroy(h)
    return out


def oracle_chain_run(model, scene, premask_bounds, iters, dmax, dmin, calls, nn_mode=0):
    thr = dmax * dmax
    out = []
    pose = np.eye(3)
    for _ in range(calls):
        pm, ps, thr = O.icp_pairs(model, scene, pose, iters, dmax, dmin, premask_bounds, thr, nn_mode)
        out.append((pm, ps))
    return out


@needs_ref
@pytest.mark.parametrize("iters", [30, 25, 12, 11, 10, 9, 3])
def test_filter_chain_matches_compiled_reference(iters):
    """DistanceFilter schedule (incl. the unsigned `icp_iterations - 10` wrap),
    ReciprocalFilter and the PairAssignment chain: pair lists identical to the compiled reference over
    repeated determinePairs() calls."""
    O.build()
    model, scene = clouds(7 + iters, 400, 500, noise=0.15)
    bounds = (4.0, 8.5, 0.0, 100.0)   # the OutOfBoundsFilter2D box removes part of the scene
    premask = ~((scene[:, 0] < bounds[0]) | (scene[:, 0] > bounds[1]) | (scene[:, 1] < bounds[2]) | (scene[:, 1] > bounds[3]))
    ref = ref_chain_run(model, scene, premask, iters, 0.4, 0.02, calls=14)
    for nn_mode in (0, 1):
        ora = oracle_chain_run(model, scene, bounds, iters, 0.4, 0.02, calls=14, nn_mode=nn_mode)
        for k, ((rm, rs), (om, os_)) in enumerate(zip(ref, ora)):
            assert np.array_equal(rm, om) and np.array_equal(rs, os_), f"iters={iters} call {k} nn_mode={nn_mode}"
    assert len(ref[0][0]) > 50
    if iters >= 10:
        assert len(ref[-1][0]) < len(ref[0][0])          # the threshold shrinks
    else:
        # icp_iterations < 10: `icp_iterations - 10` wraps as unsigned, the multiplier is ~1 and the
        # threshold is effectively constant
        assert len(ref[-1][0]) == len(ref[0][0])


@needs_ref
def test_multiplier_matches_reference_behaviour():
    """The threshold schedule is observable through which pairs survive: sweep d2 around the expected
    thresholds of the first calls and compare keep/drop decisions with the compiled DistanceFilter."""
    for iters in (30, 11, 10, 5):
        mult = O.lib().ora_distance_filter_multiplier(0.4, 0.02, iters)
        thr = 0.16
        model = np.array([[0.0, 0.0]])
        for call in range(4):
            for eps in (-1e-12, 0.0, 1e-12):
                d = np.sqrt(max(thr + eps, 0.0))
                # 4 well separated model points so that the reciprocal filter keeps every pair
                model = np.array([[0.0, 0.0], [10.0, 0.0], [20.0, 0.0], [30.0, 0.0]])
                scene = model + np.array([d, 0.0])
                ref = ref_chain_run(model, scene, np.ones(4), iters, 0.4, 0.02, calls=call + 1)[-1]
                ora = oracle_chain_run(model, scene, (-1e9, 1e9, -1e9, 1e9), iters, 0.4, 0.02, calls=call + 1)[-1]
                assert np.array_equal(ref[0], ora[0]), (iters, call, eps)
            thr = max(thr * mult, 0.02 * 0.02)


@needs_ref
def test_mathbase_helpers():
    R = O.ref()
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = rng.integers(-5, 1100, 4).astype(np.int32)
        mn, mx = C.c_int(), C.c_int()
        R.ref_minmax4(a.ctypes.data_as(C.POINTER(C.c_int)), C.byref(mn), C.byref(mx))
        # the restated min/max scan (else-if form, mathbase.h:55-64) as used by the oracle
        lo = hi = a[0]
        for v in a[1:]:
            if lo > v:
                lo = v
            elif hi < v:
                hi = v
        assert (mn.value, mx.value) == (lo, hi)
        p, q = rng.normal(0, 30, 2), rng.normal(0, 30, 2)
        assert R.ref_euklid2(O.d(p), O.d(q)) == np.sqrt((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2)
        assert R.ref_dist_sqr2d(O.d(p), O.d(q)) == (q[0] - p[0]) ** 2 + (q[1] - p[1]) ** 2
    assert R.ref_deg2rad(3.0) == 3.0 * np.pi / 180.0
    n = np.array([3e-6, 4e-6])
    R.ref_norm2(O.d(n))
    assert np.array_equal(n, [3e-6, 4e-6])        # len <= 10e-6: left untouched
    n = np.array([3.0, 4.0])
    R.ref_norm2(O.d(n))
    assert np.allclose(n, [0.6, 0.8], rtol=0, atol=1e-16)


def test_golden_chain_fixture():
    """Reference-generated golden vectors (committed): oracle must reproduce them everywhere."""
    path = os.path.join(GOLD, "ref_chain_pairs.npz")
    z = np.load(path)
    for case in range(int(z["n_cases"])):
        model, scene = z[f"model_{case}"], z[f"scene_{case}"]
        iters, calls = int(z[f"iters_{case}"]), int(z[f"calls_{case}"])
        bounds = tuple(z[f"bounds_{case}"])
        ora = oracle_chain_run(model, scene, bounds, iters, 0.4, 0.02, calls)
        for k in range(calls):
            assert np.array_equal(z[f"pm_{case}_{k}"], ora[k][0]), (case, k)
            assert np.array_equal(z[f"ps_{case}_{k}"], ora[k][1]), (case, k)


def test_text_line_readers_match_compiled_reference(tmp_path):
    """getDoubleLine / getIntLine (obcore/base/tools.cpp:190-215, compiled from the reference in oracle/_ref) against the
    oracle's restatement: the readers of TsdGrid's text file constructor (row N2).  Numbers as storeGrid writes them
    (%g), empty lines (-> NaN / 0), garbage, signs, exponents, inf / nan, leading blanks, CR LF."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference in this environment)")
    lines = ["0.025", "5", "12", "0.075", "2", "1", "-1", "nan", "inf", "-inf", "", "1e-05", "3.14159e+10", "  42", "42abc", "abc",
             "7.9", "-0.0", "0x10", "1,5", "\r", "12\r", " ", "+3", "1e400", "4.9e-324", "00012", "2147483647", "-2147483648"]
    text = "\n".join(lines) + "\n"
    path = tmp_path / "lines.txt"
    path.write_bytes(text.encode())
    for kinds in ([0] * len(lines), [1] * len(lines), [i % 2 for i in range(len(lines))]):
        k = np.array(kinds, dtype=np.int32)
        want, got = np.zeros(len(lines)), np.zeros(len(lines))
        O.ref().ref_text_lines(text.encode(), k.ctypes.data_as(O._ip), len(lines), O.d(want))
        assert O.lib().ora_text_lines(str(path).encode(), k.ctypes.data_as(O._ip), len(lines), O.d(got)) == 1
        assert np.array_equal(want, got, equal_nan=True), [(l, w, g) for l, w, g in zip(lines, want, got) if not (w == g or (w != w and g != g))]


@needs_ref
def test_filter_chain_matches_compiled_reference_on_random_clouds():
    """tools/fuzz_oracle_ref.py: the same comparison on random model / scene sizes, noise levels, out-of-bounds boxes, iteration counts on
    both sides of the unsigned wrap, filter distances and numbers of calls (33 000 cases / 34 M pairs ran clean); a short run here."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_oracle_ref.py"), "300", "424242"], cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 300 cases ok" in p.stdout
