#!/usr/bin/env python3
"""Randomised sweep of ASYNCHRONOUS MAPPING (tsd_sensor_set_async_mapping: the push beside the next registration, the next ray cast one
push behind) through the C ABI's staged scan (tsd_scan_submit / _stage / _collect) against that order on the oracle's primitives
(tests/test_gpu_async_mapping.py's two drivers; test infrastructure: uses oracle/).  Random motion and spoiled readings as in
tools/fuzz_slam.py; per scan at random: the next scan staged ahead and then delivered, staged ahead and then REPLACED by another scan
(the staged one is dropped), or not staged; the push stream held back by 0 / 0.5 / 3 ms per push (tsd_debug_stall_push_stream), so the
per-buffer push events are what keeps a staged scan from overwriting buffers a lagging push still reads.  Pose 1e-9 per scan, gates and
pair counts exact, final grid 1e-9.   usage (GPU box): python3 tools/fuzz_async.py [cases] [first_seed]"""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import synth
from oracle import pyoracle as O
from tests import helpers as H
from tests.slam_driver import slam_kwargs
from tests.test_gpu_async_mapping import OracleOnePushBehind, HipFusedAhead

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_start = time.time()
tot = dict(scans=0, pushes=0, staged=0, dropped=0, stalled_cases=0)


def spoil(rng, r32):
    r = r32.copy()
    n = len(r)
    for val in (0.0, np.nan, 45.0, 0.1):
        r[rng.integers(0, n, rng.integers(0, max(2, n // 40)))] = val
    return r


for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    map_log2 = int(rng.choice([9, 10, 10]))
    cs = float(rng.choice([0.03, 0.05, 0.05]))
    scene = str(rng.choice(["room", "pillars"]))
    geo = synth.ScanGeometry.full_circle_360() if rng.random() < 0.4 else synth.ScanGeometry.utm30lx()
    gc = synth.GridConfig(map_log2, cs)
    world = synth.World(scene, gc)
    n = int(rng.integers(8, 26))
    x, y, yaw = world.start[0], world.start[1], 0.1
    scans = []
    for k in range(n):
        r32 = world.scan(x, y, yaw, geo)
        scans.append(spoil(rng, r32) if rng.random() < 0.4 else r32)
        u = rng.random()
        step = rng.uniform(0.0, 0.04) if u < 0.2 else rng.uniform(0.05, 0.12)
        x += step * math.cos(yaw); y += step * math.sin(yaw); yaw += rng.uniform(-0.04, 0.04)
    stall = int(rng.choice([0, 0, 500, 3000]))
    tag = f"seed {seed}: 2^{map_log2} cells @ {cs} m, {scene}, {geo.beams} beams, {n} scans, push stream held back {stall} us"
    kw = slam_kwargs(gc, geo)
    hs = HipFusedAhead(O, **kw)
    oa = OracleOnePushBehind(O, **kw)
    try:
        staged_for_next = None          # the scan staged ahead in the previous call (an index into `scans`, or a decoy array)
        for k in range(n):
            cur = scans[k]
            was_staged = staged_for_next is not None and staged_for_next is cur
            if staged_for_next is not None and not was_staged:
                tot["dropped"] += 1
            nxt = None
            if k >= 1 and k + 1 < n:
                u = rng.random()
                if u < 0.55: nxt = scans[k + 1]
                elif u < 0.75: nxt = spoil(rng, scans[int(rng.integers(0, n))])          # a decoy: staged, never delivered
            rh = hs.process_scan(cur, nxt, staged=was_staged)
            staged_for_next = nxt
            tot["staged"] += int(nxt is not None)
            if k == 0:
                hs.sensor.set_async_mapping(True)
                hs.grid._check(hs.grid.lib.tsd_debug_stall_push_stream(hs.grid.h, stall), "tsd_debug_stall_push_stream")
            ro = oa.process_scan(cur)
            assert (rh["pushed"], rh["reg_error"], rh["pairs"]) == (ro["pushed"], ro["reg_error"], ro["pairs"]), f"scan {k}: hip {rh} oracle {ro}"
            d, a = H.pose_delta(ro["pose"], rh["pose"])
            assert d <= 1e-9 and a <= 1e-9, f"scan {k}: |dpose| {d} m {a} rad"
            tot["scans"] += 1; tot["pushes"] += int(ro["pushed"])
        oa.flush()
        hs.grid.sync()
        H.assert_grids_equal(oa.g.dump(), hs.grid.download_tiles(), 1e-9)
        tot["stalled_cases"] += int(stall > 0)
    except AssertionError as e:
        print("MISMATCH", tag, "--", e)
        sys.exit(1)
    finally:
        hs.grid._check(hs.grid.lib.tsd_debug_stall_push_stream(hs.grid.h, 0), "tsd_debug_stall_push_stream")
    if case % 10 == 9:
        print(f"{case + 1} cases ok ({tag}); {tot}; {time.time() - t_start:.0f} s", flush=True)
print(f"all {n_cases} cases ok from seed {seed0}: {tot}; {time.time() - t_start:.0f} s")
