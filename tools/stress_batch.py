import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi
from tests import helpers as H
import tests.test_gpu_batch as T
O.build()
# longer runs of the two-slot schedule with the pushes enqueued ahead, cfg2 grid, 4 robots
for cfg, n_scans in (("cfg2", 40), ("cfg1", 60)):
    gc, geo, kw, og, dg, robots, scans, sensors, params, gates = T._setup(O, cfg, 4, n_scans)
    slots = [capi.TsdBatch(dg, 2), capi.TsdBatch(dg, 2)]
    groups = [[0, 1], [2, 3]]
    bounds = (dg.min_x, dg.max_x, dg.min_y, dg.max_y)
    worst = 0.0
    for k in range(1, n_scans):
        ing = [rb.ingest(sc[k]) for rb, sc in zip(robots, scans)]
        ros = [rb.localise(og, d_, m_, bounds) for rb, (d_, m_, _) in zip(robots, ing)]
        for rb in robots:
            rb.apply_push(og)
        # alternate which slot begins first
        order = [0, 1]
        for si in order:
            grp = groups[si]
            slots[si].begin([sensors[i] for i in grp], [ing[i][0] for i in grp], [ing[i][1] for i in grp], [ing[i][2] for i in grp], params, gates)
        for si in order:
            slots[si].push()
        for si in order:
            for i, sr in zip(groups[si], slots[si].results()):
                worst = max(worst, T._compare(k, i, ros[i], sr))
        # NOTE: with the order alternating, the oracle's push order must follow: pushes are applied slot by slot in `order`
    try:
        H.assert_grids_equal(og.dump(), dg.download_tiles(), 1e-5)
        print(cfg, n_scans, "scans x 4 robots: OK, worst pose delta %.2e" % worst)
    except AssertionError as e:
        print(cfg, "grid mismatch (expected if the push order differs from the oracle's):", str(e)[:100])
