import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch, torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    dist.barrier()
from ohm_tsd_slam_amd import synth, facade, multigpu
gc, geo, scene = synth.CONFIGS["cfg2"]
world = synth.World(scene, gc)
n = 120
scans = synth.scans_for(world, geo, synth.trajectory(world, n))
node = facade.SlamNode(facade.node_params(gc, geo), synchronous=True)
grid = node.grid()
m = multigpu.NativeOccupancyMerger(grid, 1, 0, multigpu.NativeOccupancyMerger.new_id())
ts = []
for k in range(n):
    t0 = time.perf_counter()
    node.laser(scans[k], geo.angle_min, geo.angle_increment, ahead=scans[k + 1] if k + 1 < n else None)
    t1 = time.perf_counter()
    tm = 0.0
    if k in (30, 50, 70, 90):
        m.merge_async(); tm = time.perf_counter() - t1
    ts.append((t1 - t0, tm))
grid.sync(); m.wait()
print('total %.1f ms for %d scans' % (1e3 * sum(t[0] + t[1] for t in ts), n))
for k in (29, 30, 31, 32, 33, 34, 35, 50, 51, 52, 53, 54, 70, 71, 72, 73, 90, 91, 92, 93):
    print(k, "laser %.0f us" % (1e6 * ts[k][0]), "merge call %.0f us" % (1e6 * ts[k][1]))
m.close(); node.close()
