#!/usr/bin/env python3
"""(diagnostic) one case of tools/fuzz_icp.py in detail: per-iteration traces of both sides and the first pair lists.  usage: dbg_fuzz_icp.py seed"""
import os, sys, runpy
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
seed = int(sys.argv[1])
# run the generator part of fuzz_icp.py for exactly this seed, capturing the inputs
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_icp.py")).read()
src = src.replace("        ro = O.icp(M, S, pose, iters, dmax, dmin, bounds, nn_mode=0)", "        CAP.update(M=M.copy(), S=S.copy(), pose=pose, iters=iters, dmax=dmax, dmin=dmin, bounds=bounds, kind=kind)\n        ro = O.icp(M, S, pose, iters, dmax, dmin, bounds, nn_mode=0)")
CAP = {}
sys.argv = ["fuzz_icp.py", "1", str(seed)]
try:
    exec(compile(src, "fuzz_icp.py", "exec"), {"CAP": CAP, "__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_icp.py")})
except SystemExit:
    pass
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
M, S, pose, iters, dmax, dmin, bounds = CAP["M"], CAP["S"], CAP["pose"], CAP["iters"], CAP["dmax"], CAP["dmin"], CAP["bounds"]
print("kind", CAP["kind"], "model", len(M), "scene", len(S), "iters", iters, "dmax", dmax, "duplicates in model:", len(M) - len(np.unique(M, axis=0)), "in scene:", len(S) - len(np.unique(S, axis=0)))
gc = synth.GridConfig(8, 0.05)
dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
ro = O.icp(M, S, pose, iters, dmax, dmin, bounds, nn_mode=0, trace=True)
rd = dg.icp(M, S, pose, dg.icp_params(iters, dmax, dmin))
tr = dg.icp_trace(iters)
for k in range(min(iters, len(ro["trace"]))):
    print("iter", k, "oracle", np.round(ro["trace"][k], 10).tolist(), "\n        hip   ", np.round(tr[k], 10).tolist())
pm, ps, thr = O.icp_pairs(M, S, pose, iters, dmax, dmin, bounds, dmax * dmax, nn_mode=0)
hp = dg.icp_pairs(M, S, pose, dg.icp_params(iters, dmax, dmin), 1)[0]
so = sorted(zip(ps.tolist(), pm.tolist())); sh = sorted(zip(hp[1].tolist(), hp[0].tolist()))
print("first determinePairs: oracle", len(so), "pairs, hip", len(sh), "pairs; equal:", so == sh)
diff = [(a, b) for a, b in zip(so, sh) if a != b][:10]
print("first differing (scene, model) pairs oracle / hip:", diff)
for (a, b) in diff[:4]:
    s = S[a[0]]; print("  scene", a[0], s, "oracle model", a[1], M[a[1]], "d2", ((s - M[a[1]]) ** 2).sum(), "| hip model", b[1], M[b[1]], "d2", ((S[b[0]] - M[b[1]]) ** 2).sum())
