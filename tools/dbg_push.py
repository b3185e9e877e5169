import numpy as np, sys
sys.path.insert(0, '.')
from oracle import pyoracle as O
from ohm_tsd_slam_amd import capi, synth
from tests import helpers as H
from tests.test_gpu_parity import make_pair, push_both
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
og, dg = make_pair(O, gc)
for k in range(8):
    so, sd = push_both(O, og, dg, world, geo, k)
    oi, oiw, ot, ow = og.dump(); gi, giw, gt, gw = dg.download_tiles()
    sel = oi.astype(bool)
    a, b = ot[sel], gt[sel]; m = ~np.isnan(a)
    dt = np.abs(a[m]-b[m]); dw = np.abs(ow[sel]-gw[sel])
    print(k, "stats eq", so == sd, "tsd maxdiff %.3e n>0: %d of %d" % (dt.max(), (dt>0).sum(), dt.size), "w maxdiff %.3e n>0 %d" % (dw.max(), (dw>0).sum()))
    pose, _ = H.sensor_pose(world, k)
    rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
    co, no, mo, cnt_o = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    cd, nd, md, cnt_d = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
    s2 = np.repeat((mo & md).astype(bool), 2)
    print("   raycast masks equal", np.array_equal(mo, md), cnt_o, "coords maxdiff %.3e normals %.3e" % (np.abs(co[s2]-cd[s2]).max(), np.abs(no[s2]-nd[s2]).max()))
