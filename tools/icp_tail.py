#!/usr/bin/env python3
"""Per scan of the bench trajectory: k_icp's dispatch time and the length of each of its steps (wave 0's stamps of the timeline
build).  Run through tools/icp_tail.sh.  Test infrastructure (drives the unfused C ABI through tests/slam_driver.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests.slam_driver import PrimitiveLoop as Loop, slam_kwargs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
FUSED = "--fused" in sys.argv      # the facade's fused scan (what the bench runs: no per-step record, nothing in flight behind the steps)
O.build()
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
poses = synth.trajectory(world, n)
scans = synth.scans_for(world, geo, poses)
geo_msg = synth.ScanGeometry(geo.beams, float(np.float32(geo.angle_min)), float(np.float32(geo.angle_increment)))
kw = slam_kwargs(gc, geo_msg)
if FUSED:
    from ohm_tsd_slam_amd import facade
    node = facade.SlamNode(facade.node_params(gc, geo), device=0, synchronous=True)
    dg = node.grid()
else:
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    lh = Loop(O, kw, dg, True)
rows = []
ws_all, tl_all = {}, {}
for k in range(n):
    dg.profile(True, "icp"); dg.profile_reset()
    if FUSED:
        node.laser(scans[k], geo.angle_min, geo.angle_increment)
        dg.sync()
        out = dict(pairs=-1)
    else:
        out = lh.scan(scans[k])
    ms, cnt = dg.profile_get("icp")
    if not cnt:
        continue
    tr = np.zeros((512, 8)); dg.lib.tsd_icp_trace(dg.h, tr.ctypes.data_as(capi._dp), 512)
    tl = tr.reshape(-1)[256 * 8: 256 * 8 + 30 * 16].reshape(30, 16)
    ph = tr.reshape(-1)[(256 + 192) * 8: (256 + 192) * 8 + 16].copy()
    pw = tr.reshape(-1)[(256 + 194) * 8: (256 + 194) * 8 + 64].reshape(8, 8)[:, :6].copy()
    steps = np.diff(tl[:, 0]).astype(int)
    srch = [i for i in range(30) if tl[i, 13] > tl[i, 4]]
    ws = tr.reshape(-1)[(256 + 128) * 8: (256 + 128) * 8 + 30 * 8].reshape(30, 8)
    M, S, _ = lh.inputs if not FUSED else (np.zeros((0, 2)), np.zeros((0, 2)), None)
    ws_all[k] = ws.copy(); tl_all[k] = tl.copy()
    hist_all = globals().setdefault('hist_all', {}); hist_all[k] = tr.reshape(-1)[(256 + 202) * 8: (256 + 202) * 8 + 16 * 21].copy()
    if FUSED:
        ws = np.zeros_like(ws)
    rows.append(dict(k=k, us=1e3 * ms / cnt, nM=len(M), nS=len(S), pairs=int(out["pairs"]), steps=steps, srch=srch,
                     first=int(tl[0, 0]), ph=ph, pw=pw, whole=[int(tl[i, 14] - tl[i, 13]) for i in srch], win=[int(tl[i, 13] - tl[i, 4]) for i in srch],
                     lists=[int(w[0]) if i in srch else 0 for i, w in enumerate(ws)],
                     rounds=[(int(ws[i][4]), int(ws[i][1])) for i in srch]))
us = np.array([r["us"] for r in rows])
print(f"{len(rows)} registrations: mean {us.mean():.1f} us, p50 {np.median(us):.1f}, p90 {np.percentile(us, 90):.1f}, p99 {np.percentile(us, 99):.1f}, max {us.max():.1f}")
phs = np.array([r["ph"] for r in rows])
print("whole-kernel phases, mean cycles (thread 0): entry -> set-up done %d | wait for the helpers' granules %d | the 30 steps %d | epilogue %d" % tuple(phs.mean(axis=0)[:4]))
print("   set-up in parts: entry -> inputs arrived %d | -> compacted into LDS (2 barriers) %d | -> scene in registers, directions, padding %d | -> slots cleared, 2 barriers %d" % tuple(phs.mean(axis=0)[4:8]))
print("   epilogue in parts: loop end -> trace dump, result %d | gate (atan2, sin) %d | wave 0's rays turned %d | barrier (all rays, pose bookkeeping) %d | record composed, stores issued %d | drained %d | sequence number, end %d" % tuple(phs.mean(axis=0)[8:15]))
pws = np.array([r["pw"] for r in rows]).mean(axis=0)
print("   set-up per wave (cycles since entry; rows = waves 0..7): inputs arrived | compacted (behind 2 barriers) | scene, directions, padding | set-up done")
for w in range(8):
    print("      wave %d: %6d %6d %6d | slots cleared, barrier %6d | rmax known %6d | set-up done %6d" % (w, pws[w][0], pws[w][1], pws[w][2], pws[w][4], pws[w][5], pws[w][3]))
steady = np.array([np.median(r["steps"][18:]) for r in rows])
print(f"steady step (median of steps 18..28): mean {steady.mean():.0f} cycles; sum of steps mean {np.mean([r['steps'].sum() for r in rows]):.0f}")
print(f"search steps per registration: mean {np.mean([len(r['srch']) for r in rows]):.1f}, max {max(len(r['srch']) for r in rows)}")
tot = np.array([r["steps"].sum() for r in rows], dtype=float)
srch_cycles = np.array([sum(r["steps"][i] for i in r["srch"] if i < 29) for r in rows], dtype=float)
print(f"share of the loop spent in search steps: mean {np.mean(srch_cycles / tot):.2f}")
print("mean step length by step index:", np.mean([r["steps"] for r in rows], axis=0).astype(int).tolist())
print("p90 step length by step index: ", np.percentile([r["steps"] for r in rows], 90, axis=0).astype(int).tolist())
rows_late = [(ws_all[r["k"]][i], tl_all[r["k"]][i]) for r in rows for i in r["srch"] if i > 15 and ws_all[r["k"]][i][1] == 1]
if rows_late:
    a = np.array([[w[5], w[6], w[7], w[3] - (w[6] + w[7]) - 0, w[3], t[13] - t[4], t[14] - t[13], t[5] - t[14], t[4] - t[0], t[12] - t[5]] for w, t in rows_late])
    print("search steps after step 15 whose first wave needed ONE window round (%d): mean cycles of" % len(a))
    for name, v in zip(("barrier 1 -> list entry in registers", "-> the round's 15 LDS reads arrived", "-> round evaluated, loop left",
                        "-> neighbours re-evaluated, function returned (incl. entry read)", "whole call incl. entry read", "barrier 1 -> window pass done (barrier)",
                        "-> tier 2 counter read", "-> results read, slots updated, barrier, minima read again (winners known)",
                        "top of step -> barrier 1", "winners known -> end of step"), a.mean(axis=0)):
        print(f"   {name:70s} {v:7.0f}")
late = [(r["lists"][i], r["rounds"][j][1]) for r in rows for j, i in enumerate(r["srch"]) if i > 15]
if late:
    print(f"search steps after step 15: {len(late)} in {len(rows)} registrations; list length mean {np.mean([l for l, _ in late]):.1f}, "
          f"p90 {np.percentile([l for l, _ in late], 90):.0f}; largest window rounds mean {np.mean([m for _, m in late]):.1f}")
hist = tr.reshape(-1)[(256 + 202) * 8: (256 + 202) * 8 + 16 * 21].reshape(16, 3, 7)
print("\nwindow-pass entries of the whole run by step (rows) and number of window rounds (1..7): resolved with a partner | resolved, nothing within the filter distance | unresolved (-> whole-wave search)")
for it in range(16):
    print("   step %2d: %s | %s | %s" % (it, hist[it, 0].astype(int).tolist(), hist[it, 1].astype(int).tolist(), hist[it, 2].astype(int).tolist()))
if 36 in hist_all and 57 in hist_all:
    hd = (hist_all[57] - hist_all[36]).reshape(16, 3, 7)
    print("\nthe same for scans 37..57 alone (the robot turns into space the map does not hold: the registration's tail), per registration:")
    for it in range(16):
        print("   step %2d: %s | %s | %s" % (it, np.round(hd[it, 0] / 21.0, 1).tolist(), np.round(hd[it, 1] / 21.0, 1).tolist(), np.round(hd[it, 2] / 21.0, 1).tolist()))
print("\nthe slowest registrations:")
for r in sorted(rows, key=lambda r: -r["us"])[:14]:
    print(f"scan {r['k']:3d}: {r['us']:6.1f} us, model {r['nM']}, scene {r['nS']}, pairs {r['pairs']}, search steps {r['srch']}")
    print("   steps:", r["steps"].tolist())
    print("   window pass:", r["win"], " whole-wave searches:", r["whole"])
    print("   lists:", r["lists"])
    print("   (entries of the second window pass, largest number of rounds in its first wave):", r["rounds"])
print("\nthe fastest:")
for r in sorted(rows, key=lambda r: r["us"])[:3]:
    print(f"scan {r['k']:3d}: {r['us']:6.1f} us, model {r['nM']}, scene {r['nS']}, pairs {r['pairs']}, search steps {r['srch']}")
    print("   steps:", r["steps"].tolist())
