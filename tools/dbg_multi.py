import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
from tests.test_gpu_facade import _two_robot_setup
from tests import helpers as H
gc, geo, scans, sos, node = _two_robot_setup(O, "cfg1", 6)
for k in range(6):
    for r in (0, 1):
        ro = sos[r].process_scan(scans[r][k])
        node.laser(scans[r][k], geo.angle_min, geo.angle_increment, robot=r)
        rh = node.report(r)
        d, a = H.pose_delta(np.array(ro.pose[:]).reshape(3, 3), rh["pose"])
        print(k, r, "d %.3e a %.3e" % (d, a), "pairs", ro.pairs, rh["pairs"], "vm", ro.valid_model, rh["valid_model"], "vs", ro.valid_scene, rh["valid_scene"], "pushed", ro.pushed, rh["pushed"])
