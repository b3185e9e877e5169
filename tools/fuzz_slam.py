#!/usr/bin/env python3
"""Randomised closed-loop sweep of the C++ facade (ThreadLocalize::laserCallBack -> fused tsd_scan, the path bench.py drives) against
the oracle's SLAM loop (test infrastructure: uses oracle/): random grids / scenes / scan geometries, random motion per scan (steps below
and above the 0.05 m push gate, turns, an occasional jump that trips the registration-error gate), spoiled readings, and the next scan
ANNOUNCED ahead at random -- sometimes the scan that then really comes, sometimes a decoy (staged, then dropped), sometimes nothing.
Per scan: pose 1e-9, pairs / pushed / registration error exact; at the end every cell of the grid.  Short trajectories (the free-running
loop amplifies last-bit differences over hundreds of scans).  Modes: registration_mode 0; "async": asynchronous mapping
against the oracle's primitives one push behind is covered by tests/test_gpu_async_mapping.py, not here.
"mode3": half of the cases run registration_mode 3 (TSD_PDF pre-registration inside the fused scan, config/single-laser.yaml's mode) with
random trials / control-set sizes, both sides fed the same seeded rand() draws (a decoy announcement must not consume a scan's draws).
"async": half of the cases run the facade with `async_mapping: 1` (the push beside the next registration) against the one-push-behind
order on the oracle's primitives (tests/test_gpu_async_mapping.py's OracleOnePushBehind).
usage (GPU box): python3 tools/fuzz_slam.py [cases] [first_seed] [mode3|async]"""
import ctypes as C, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import facade, synth
from oracle import pyoracle as O
from tests import helpers as H
from tests.slam_driver import slam_kwargs

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with_mode3 = len(sys.argv) > 3 and sys.argv[3] == "mode3"
with_async = len(sys.argv) > 3 and sys.argv[3] == "async"
_libc = C.CDLL(None)


def libc_draws(seed, n_sub, n_ctrl, n_trials):
    """what the facade's TSD_PDFMatching draws for `tsdpdf_seed` >= 0: srand(seed + call), then rand() in this order"""
    _libc.srand(C.c_uint(seed))
    return ([_libc.rand() for _ in range(n_sub)], [_libc.rand() for _ in range(n_ctrl)], [_libc.rand() for _ in range(n_trials)])
t_start = time.time()
tot = dict(scans=0, pushes=0, reg_errors=0, decoys=0, announced=0, mode3_cases=0)


def spoil(rng, r32):
    r = r32.copy()
    n = len(r)
    for val in (0.0, np.nan, 45.0, 0.1):
        r[rng.integers(0, n, rng.integers(0, max(2, n // 40)))] = val
    return r


for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    map_log2 = int(rng.choice([9, 10, 10, 11]))
    cs = float(rng.choice([0.03, 0.05, 0.05]))
    scene = str(rng.choice(["room", "pillars"]))
    geo = synth.ScanGeometry.full_circle_360() if rng.random() < 0.4 else synth.ScanGeometry.utm30lx()
    clockwise = False
    if rng.random() < 0.3:
        # any scanner: odd beam counts, narrow and wide fields of view -- and a CLOCKWISE one (negative increment from a positive
        # angle_min: ThreadLocalize.cpp:491-497 reverses the scan and flips the angles)
        nb = int(rng.choice([91, 181, 361, 541, 1000, 1440]))
        fov = math.radians(float(rng.uniform(120.0, 340.0)))
        geo = synth.ScanGeometry(nb, -0.5 * fov, fov / (nb - 1))
        clockwise = bool(rng.random() < 0.4)
    if os.environ.get("FUZZ_GEO") == "utm": geo = synth.ScanGeometry.utm30lx()
    if os.environ.get("FUZZ_GEO") == "360": geo = synth.ScanGeometry.full_circle_360()
    gc = synth.GridConfig(map_log2, cs)
    world = synth.World(scene, gc)
    n = int(rng.integers(8, 22))
    # ground truth: the node's start pose, then random increments in the robot's frame
    x, y, yaw = world.start[0], world.start[1], 0.1
    truth = [(x, y, yaw)]
    for k in range(1, n):
        u = rng.random()
        step = rng.uniform(0.0, 0.04) if u < 0.2 else rng.uniform(0.05, 0.12)          # below / above the push gate
        dyaw = rng.uniform(-0.04, 0.04)
        if u > 0.95 and k > 2:
            step = rng.uniform(1.1, 1.6)                                                  # the registration-error gate (reg_trs_max 1.0)
        x += step * math.cos(yaw); y += step * math.sin(yaw); yaw += dyaw
        truth.append((x, y, yaw))
    scans = []
    for (px, py, pyaw) in truth:
        r32 = world.scan(px, py, pyaw, geo)
        sp = spoil(rng, r32) if rng.random() < 0.4 else r32
        scans.append(r32 if os.environ.get("FUZZ_NO_SPOIL") else sp)
    msg_min, msg_inc = geo.angle_min, geo.angle_increment
    if clockwise:
        # the same physical scanner mounted the other way round reports its beams from +|angle_min| downwards
        scans = [np.ascontiguousarray(sc[::-1]) for sc in scans]
        msg_min, msg_inc = -geo.angle_min, -geo.angle_increment
    tag = f"seed {seed}: 2^{map_log2} cells @ {cs} m, {scene}, {geo.beams} beams{' (clockwise)' if clockwise else ''}, {n} scans"
    mode3 = with_mode3 and rng.random() < 0.5
    params = facade.node_params(gc, geo)
    extra = {}
    if mode3:
        trials = int(rng.choice([30, 60, 100, 200])); ctrl = int(rng.choice([60, 100, 140, 200])); pseed = int(rng.integers(1, 100000))
        params.update({"registration_mode": 3, "trials": trials, "sizeControlSet": ctrl, "zrand": 0.25, "ransac_phi_max": 30.0, "tsdpdf_seed": pseed})
        extra = dict(registration_mode=3, trials=trials, size_control_set=ctrl, zrand=0.25, ransac_phi_max=30.0)
        tag += f", mode 3 ({trials} trials, {ctrl} control points)"
    use_async = with_async and rng.random() < 0.5
    if use_async:
        params["async_mapping"] = 1
        tag += ", asynchronous mapping"
    node = facade.SlamNode(params, device=0, synchronous=True)
    # (sensor_msgs/LaserScan carries angle_min / angle_increment as float32, the facade's scan type likewise -- ros_shim.h:38,
    # ThreadLocalize.cpp:632-642: the oracle's loop gets the same rounded values)
    okw = slam_kwargs(gc, geo, threads=8, angle_min=float(np.float32(msg_min)), angle_increment=float(np.float32(msg_inc)), **extra)
    if use_async:
        from tests.test_gpu_async_mapping import OracleOnePushBehind
        if clockwise:                                    # (the primitives-level driver has no clockwise branch: feed it the flipped scan itself)
            okw.update(angle_min=float(np.float32(-msg_min)), angle_increment=float(np.float32(-msg_inc)))
        oab = OracleOnePushBehind(O, **okw)
        class _Wrap:                                       # the same result fields as O.Slam's
            def process_scan(self, r):
                d = oab.process_scan(r[::-1] if clockwise else r)
                return type("R", (), dict(pose=np.asarray(d["pose"]).reshape(-1), pushed=d["pushed"], reg_error=d["reg_error"], pairs=d["pairs"],
                                         rms=0.0, iterations=0, valid_model=0))()
        osl = _Wrap(); osl.grid = None
    else:
        osl = O.Slam(**okw)
    try:
        for k in range(n):
            u = rng.random()
            ahead = None
            if k + 1 < n and u < 0.6:
                ahead = scans[k + 1]; tot["announced"] += 1
            elif u < 0.8:
                ahead = scans[int(rng.integers(0, n))] if rng.random() < 0.5 else spoil(rng, scans[k]); tot["decoys"] += 1
            if os.environ.get("FUZZ_NO_AHEAD"):
                ahead = None
            node.laser(scans[k], msg_min, msg_inc, ahead=ahead)
            if mode3 and k > 0:
                osl.set_draws(*libc_draws(pseed + (k - 1), geo.beams, ctrl, trials))
            ro = osl.process_scan(scans[k])
            rh = node.report()
            Po = np.array(ro.pose).reshape(3, 3)
            d, a = H.pose_delta(Po, rh["pose"])
            assert d <= 1e-9 and a <= 1e-9, (f"scan {k}: |dpose| {d} m {a} rad (oracle pairs {ro.pairs}, hip {rh.get('pairs')}; rms {ro.rms} / {rh['rms']}; "
                                             f"iterations {ro.iterations} / {rh['iterations']}; valid model {ro.valid_model} / {rh['valid_model']}; T hip {rh['T'].reshape(-1)[:6]})")
            assert int(ro.pushed) == int(rh["pushed"]) and int(ro.reg_error) == int(rh["reg_error"]), \
                f"scan {k}: pushed {ro.pushed} / {rh['pushed']}, reg_error {ro.reg_error} / {rh['reg_error']}"
            if k > 0 and not ro.reg_error:
                assert int(ro.pairs) == int(rh["pairs"]), f"scan {k}: pairs {ro.pairs} / {rh['pairs']}"
            tot["scans"] += 1; tot["pushes"] += int(ro.pushed); tot["reg_errors"] += int(ro.reg_error)
        tot["mode3_cases"] += int(mode3); tot["async_cases"] = tot.get("async_cases", 0) + int(use_async)
        if use_async:
            oab.flush(); node.grid().sync()
            H.assert_grids_equal(oab.g.dump(), node.grid().download_tiles(), 1e-9)
        else:
            H.assert_grids_equal(osl.grid.dump(), node.grid().download_tiles(), 1e-9)      # (free-running: the poses differ by ~1e-14)
    except AssertionError as e:
        print("MISMATCH", tag, "--", e)
        sys.exit(1)
    finally:
        node.close()
    if case % 10 == 9:
        print(f"{case + 1} cases ok ({tag}); {tot}; {time.time() - t_start:.0f} s", flush=True)
print(f"all {n_cases} cases ok from seed {seed0}: {tot}; {time.time() - t_start:.0f} s")
