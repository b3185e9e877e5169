#!/usr/bin/env python3
"""Diagnostic (test infrastructure: uses oracle/): WHERE does a free-running HIP closed loop first take a different
discrete decision than the free-running oracle, and how close to a tie was that decision?

Both loops are ThreadLocalize::eventLoop (ThreadLocalize.cpp:310-409) on the same scans, each feeding its OWN
registration result back through its own pose and grid:
  oracle loop : oracle/tsd_oracle.c primitives (ray cast, Icp::iterate, push)
  HIP loop    : tsd_raycast / tsd_icp / tsd_push of include/tsd_hip.h (the unfused calls, so the inputs of every
                registration are on the host; the fused facade is run beside it and must follow it exactly)
Before any decision flips the two differ by summation order only (1e-15 per registration, fed back).  Per scan the
tool compares the ray cast's hit mask, every iteration's pair count (Icp.cpp:410-512), the gates and the push
statistics.  At the first scan + iteration that differs it
  (1) runs the ORACLE's registration on the HIP loop's inputs: equal to the HIP result => the kernel is right on
      its inputs and the flip comes from the inputs (1e-13 apart), not from the kernel;
  (2) rebuilds that iteration's pair chain in numpy on both sides' inputs (OutOfBoundsFilter2D.cpp:27-37, exact 1-NN,
      DistanceFilter.cpp:32-64, ReciprocalFilter.cpp:32-78), lists the scene points whose fate differs and prints
      the deciding quantity on both sides with its margin.
usage: python tools/first_flip.py [n_scans=340] [cfg=cfg2] [out=gpurun_out/first_flip.txt] [--oracle-f64-angles]

--oracle-f64-angles gives the ORACLE loop the scanner's float64 angle_min / angle_increment instead of the float32 values a
sensor_msgs/LaserScan carries (what round 4's tools/long_run_check.py did by mistake: the two loops then process scanners that
differ by 1.5e-9 rad per beam, drift apart at the 1e-8 level and do flip): it exercises the analysis below and shows what the
1e-2 m "divergence" of profiles/r4_soak_checks.txt was."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import capi, facade, synth
from oracle import pyoracle as O
from tests.slam_driver import PrimitiveLoop as Loop, slam_kwargs

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
F64_ANGLES = "--oracle-f64-angles" in sys.argv
n = int(argv[0]) if len(argv) > 0 else 340
cfg = argv[1] if len(argv) > 1 else "cfg2"
out_path = argv[2] if len(argv) > 2 else "gpurun_out/first_flip.txt"
THREADS = max(1, min(64, os.cpu_count() or 1))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


def scene_at(S, trace, j):
    """the scene as Icp::applyTransformation (Icp.cpp:371-408) left it before iteration j: the traced Tlast of the
    iterations 0..j-1 applied one after the other in the reference's operation order"""
    x, y = S[:, 0].copy(), S[:, 1].copy()
    for i in range(j):
        co, si, dx, dy = trace[i, 4:8]
        nx = (0.0 + x * co) + y * (-si)
        ny = (0.0 + x * si) + y * co
        x, y = nx + dx, ny + dy
    return x, y


def chain(M, x, y, pose, bounds, thr):
    """one determinePairs call in numpy with every deciding quantity kept"""
    wx = ((0.0 + x * pose[0, 0]) + y * pose[0, 1]) + pose[0, 2]
    wy = ((0.0 + x * pose[1, 0]) + y * pose[1, 1]) + pose[1, 2]
    inb = ~((wx < bounds[0]) | (wx > bounds[1]) | (wy < bounds[2]) | (wy > bounds[3]))
    oob_margin = np.minimum.reduce([np.abs(wx - bounds[0]), np.abs(wx - bounds[1]), np.abs(wy - bounds[2]), np.abs(wy - bounds[3])])
    dx = x[:, None] - M[None, :, 0]
    dy = y[:, None] - M[None, :, 1]
    d2 = dx * dx + dy * dy
    order = np.argsort(d2, axis=1, kind="stable")
    nn = order[:, 0]
    d1 = d2[np.arange(len(x)), nn]
    dsec = d2[np.arange(len(x)), order[:, 1]] if M.shape[0] > 1 else np.full(len(x), np.inf)
    passed = inb & (d1 <= thr)
    winner = {}
    for i in np.nonzero(passed)[0]:
        m = int(nn[i])
        if m not in winner or d1[i] < d1[winner[m]]:
            winner[m] = int(i)
    fate = {}
    for i in range(len(x)):
        if not inb[i]:
            fate[i] = ("out of bounds", -1)
        elif not passed[i]:
            fate[i] = ("distance filter", int(nn[i]))
        elif winner[int(nn[i])] != i:
            fate[i] = ("reciprocal loser", int(nn[i]))
        else:
            fate[i] = ("pair", int(nn[i]))
    return dict(inb=inb, oob_margin=oob_margin, nn=nn, d1=d1, dsec=dsec, passed=passed, winner=winner, fate=fate,
                pairs=sum(1 for v in fate.values() if v[0] == "pair"))


def analyse(k, j, lo, lh, ro, rh):
    Mo, So, Po = lo.inputs
    Mh, Sh, Ph = lh.inputs
    kw = lo.kw
    say("inputs of scan %d: model points %d / %d, scene points %d / %d (oracle / HIP)" % (k, len(Mo), len(Mh), len(So), len(Sh)))
    if Mo.shape == Mh.shape:
        say("  max |model_o - model_h| = %.3e m, max |pose_o - pose_h| = %.3e" % (np.max(np.abs(Mo - Mh)), np.max(np.abs(Po - Ph))))
    # (1) the oracle's registration on the HIP loop's inputs
    rx = O.icp(Mh, Sh, Ph, kw["icp_iterations"], kw["dist_filter_max"], kw["dist_filter_min"], lo.bounds, trace=True)
    same = rx["iterations"] == rh["iterations"] and np.array_equal(rx["trace"][:, 0], rh["trace"][:, 0]) and rx["state"] == rh["state"]
    dT = np.max(np.abs(rx["T"] - rh["T"]))
    say("(1) oracle registration ON THE HIP LOOP'S INPUTS vs the HIP result: pair counts of all %d iterations %s, state %s, max |dT| %.3e"
        % (rh["iterations"], "EQUAL" if same else "DIFFER", "equal" if rx["state"] == rh["state"] else "differ", dT))
    if not same:
        say("    oracle-on-HIP-inputs pairs:", rx["trace"][:, 0].astype(int).tolist())
        say("    HIP pairs                 :", rh["trace"][:, 0].astype(int).tolist())
        say("    => the HIP kernel deviates from the oracle ON IDENTICAL INPUTS: a kernel defect, not a rounding flip")
    # (2) the pair chain of iteration j on both sides
    tro, trh = ro["trace"], rh["trace"]
    xo, yo = scene_at(So, tro, j)
    xh, yh = scene_at(Sh, trh, j)
    co = chain(Mo, xo, yo, Po, lo.bounds, tro[j, 2])
    ch = chain(Mh, xh, yh, Ph, lo.bounds, trh[j, 2])
    say("(2) iteration %d: threshold^2 %.17g / %.17g; numpy chain gives %d / %d pairs (traces say %d / %d)"
        % (j, tro[j, 2], trh[j, 2], co["pairs"], ch["pairs"], int(tro[j, 0]), int(trh[j, 0])))
    say("    scene at this iteration: max |scene_o - scene_h| = %.3e m" % max(np.max(np.abs(xo - xh)), np.max(np.abs(yo - yh))))
    worst_rel = 0.0
    nd = 0
    if len(xo) == len(xh) and Mo.shape == Mh.shape:
        for i in range(len(xo)):
            fo, fh = co["fate"][i], ch["fate"][i]
            if fo == fh:
                continue
            nd += 1
            # which decision separates the two fates?
            if co["inb"][i] != ch["inb"][i]:
                what = "OutOfBoundsFilter2D"; a, b = co["oob_margin"][i], ch["oob_margin"][i]; rel = max(a, b) / max(abs(lo.bounds[1]), 1.0)
                detail = "distance to the nearest bound %.3e / %.3e m" % (a, b)
            elif co["nn"][i] != ch["nn"][i]:
                what = "nearest neighbour"
                a = abs(co["dsec"][i] - co["d1"][i]); b = abs(ch["dsec"][i] - ch["d1"][i]); rel = max(a / co["d1"][i], b / ch["d1"][i])
                detail = "d2(best) %.17g d2(second) %.17g | %.17g %.17g" % (co["d1"][i], co["dsec"][i], ch["d1"][i], ch["dsec"][i])
            elif co["passed"][i] != ch["passed"][i]:
                what = "DistanceFilter"
                rel = max(abs(co["d1"][i] - tro[j, 2]) / tro[j, 2], abs(ch["d1"][i] - trh[j, 2]) / trh[j, 2])
                detail = "d2 %.17g vs thr2 %.17g | d2 %.17g vs thr2 %.17g" % (co["d1"][i], tro[j, 2], ch["d1"][i], trh[j, 2])
            else:
                what = "ReciprocalFilter"
                m = int(co["nn"][i]); wo, wh = co["winner"].get(m, -1), ch["winner"].get(m, -1)
                rival = wo if wo != i else wh
                if rival >= 0:
                    a = abs(co["d1"][i] - co["d1"][rival]); b = abs(ch["d1"][i] - ch["d1"][rival])
                    rel = max(a / max(co["d1"][i], 1e-300), b / max(ch["d1"][i], 1e-300))
                    detail = "model %d claimed by scene %d and %d: d2 %.17g vs %.17g | %.17g vs %.17g" % (
                        m, i, rival, co["d1"][i], co["d1"][rival], ch["d1"][i], ch["d1"][rival])
                else:
                    # the rival itself has a different fate (it is listed on its own line): this point only follows it
                    rel = 0.0; detail = "follows another point's flip (model %d)" % m
            worst_rel = max(worst_rel, rel)
            say("    scene point %4d: oracle %-18s (model %4d)  HIP %-18s (model %4d)  decided by %s: %s  => relative margin %.3e"
                % (i, fo[0], fo[1], fh[0], fh[1], what, detail, rel))
    say("    %d scene points with a different fate; the largest relative margin among the deciding quantities: %.3e" % (nd, worst_rel))
    return same, worst_rel, nd


gc, geo, scene = synth.CONFIGS[cfg]
world = synth.World(scene, gc)
poses = synth.trajectory(world, n)
scans = synth.scans_for(world, geo, poses)
geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
kw = slam_kwargs(gc, geo_msg)
og = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
kw_o = slam_kwargs(gc, geo) if F64_ANGLES else kw
lo, lh = Loop(O, kw_o, og, False, THREADS), Loop(O, kw, dg, True)
assert lo.bounds == lh.bounds
node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
say("# tools/first_flip.py %d %s: free-running oracle loop, HIP loop (unfused C ABI) and fused facade on the same scans" % (n, cfg))
if F64_ANGLES:
    say("# --oracle-f64-angles: the oracle loop's scanner has the float64 angles, the HIP loop's the message's float32 ones (round 4's tool error, on purpose)")
flip = None
worst_before = 0.0
worst_facade = 0.0
worst_all = 0.0
for k in range(n):
    ro = lo.scan(scans[k])
    rh = lh.scan(scans[k])
    node.laser(scans[k], geo.angle_min, geo.angle_increment)
    rf = node.report()
    dd = math.hypot(ro["pose"][0, 2] - rh["pose"][0, 2], ro["pose"][1, 2] - rh["pose"][1, 2])
    df = float(np.max(np.abs(np.asarray(rf["pose"]) - rh["pose"])))
    worst_facade = max(worst_facade, df)
    worst_all = max(worst_all, dd)
    if flip is None:
        what = None
        if k > 0:
            if not np.array_equal(ro["hit"], rh["hit"]):
                what = ("ray cast hit mask", -1)
            elif ro["no_model"] != rh["no_model"]:
                what = ("no model", -1)
            elif ro["trace"] is not None and (len(ro["trace"]) != len(rh["trace"]) or not np.array_equal(ro["trace"][:, 0], rh["trace"][:, 0])):
                m = min(len(ro["trace"]), len(rh["trace"]))
                neq = np.nonzero(ro["trace"][:m, 0] != rh["trace"][:m, 0])[0]
                what = ("pair count", int(neq[0]) if len(neq) else m)
            elif (ro["pushed"], ro["reg_error"]) != (rh["pushed"], rh["reg_error"]):
                what = ("gates", -1)
            elif ro["stats"] != rh["stats"]:
                what = ("push statistics", -1)
        if what is None:
            worst_before = max(worst_before, dd)
        else:
            flip = (k, what)
            say("FIRST DIFFERENT DECISION: scan %d, %s%s; |pose_o - pose_h| before this scan's flip was at most %.3e m"
                % (k, what[0], (" at iteration %d" % what[1]) if what[1] >= 0 else "", worst_before))
            if what[0] == "pair count":
                same, worst_rel, nd = analyse(k, what[1], lo, lh, ro, rh)
            elif what[0] == "ray cast hit mask":
                idx = np.nonzero(ro["hit"] != rh["hit"])[0]
                say("    beams whose hit / miss differs:", idx.tolist())
            else:
                say("    oracle:", {a: ro[a] for a in ("pushed", "reg_error", "no_model", "stats")})
                say("    HIP   :", {a: rh[a] for a in ("pushed", "reg_error", "no_model", "stats")})
    if k % 25 == 0 or (flip is not None and k - flip[0] < 3):
        say("scan %4d  |pose_o - pose_h| %.3e m  pairs %s / %s  facade-vs-unfused %.1e" % (k, dd, ro.get("pairs"), rh.get("pairs"), df))
if flip is None:
    say("no discrete decision differs in %d scans; max |pose_o - pose_h| %.3e m" % (n, worst_all))
say("max |pose_o - pose_h| over all %d scans %.3e m; fused facade against the unfused HIP loop: max abs pose difference %.3e" % (n, worst_all, worst_facade))
node.close()
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
with open(out_path, "w") as f:
    f.write("\n".join(lines) + "\n")
