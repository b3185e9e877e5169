"""Diagnostic build -DTSD_PUSH_VERIFY_INDEX (tools/push_verify_index.sh): every beam index the fp32 estimate of k_push_update DECIDES is
compared with the exact fp64 formulation (SensorPolar2D::backProject) inside the kernel; counts over a trajectory of pushes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from ohm_tsd_slam_amd import capi, synth, facade

lib = facade.load_library()
for cfg, scene, n in (("cfg1", "room", 60), ("cfg2", "pillars", 200), ("cfg2", "room", 60), ("cfg3", "comb", 30), ("cfg3", "pillars", 60)):
    gc, geo, _ = synth.CONFIGS[cfg]
    world = synth.World(scene, gc)
    poses = synth.trajectory(world, n)
    scans = synth.scans_for(world, geo, poses)
    grid = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    for k in range(n):
        data = np.zeros(geo.beams); mask = np.zeros(geo.beams, dtype=np.uint8)
        r = np.ascontiguousarray(scans[k], dtype=np.float32)
        lib.tsd_host_sensor_ingest_f32(r.ctypes.data_as(C.POINTER(C.c_float)), geo.beams, geo.angle_increment, geo.angle_min, 30.0,
                                       data.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), 0)
        grid.push(synth.pose_matrix(*poses[k]), data, mask, geo.angle_increment, geo.angle_min, 30.0, 0.001, 2.0, want_stats=False)
    grid.sync()
    tr = grid.icp_trace(256).reshape(-1)
    wrong, unsure, total, fix_wrong, hard = (int(x) for x in tr[1000:1005].view(np.uint64))
    print(f"{cfg} / {scene}: {n} pushes, {total} cells visited by UPDATE tiles: estimate decided {total - unsure} ({100.0 * (total - unsure) / max(total, 1):.2f} %), "
          f"decided WRONGLY {wrong}; {unsure} cells left to the boundary-side test (fp64 cross product), of which {hard} went on to the fp64 atan2; "
          f"the side test + atan2 decided WRONGLY {fix_wrong}")
    nw = int(tr[1005:1006].view(np.uint64)[0])
    for i in range(min(nw, 40)):
        o = tr[1010 + 8 * i: 1018 + 8 * i]
        PX = gc.cells // 32
        px, py = int(o[0]) % PX, int(o[0]) // PX
        cx, cy = (px * 32 + (int(o[1]) & 31) + 0.5) * gc.cell_size, (py * 32 + (int(o[1]) >> 5) + 0.5) * gc.cell_size
        print(f"   wrong: tile {int(o[0])} cell {int(o[1])} estimate {int(o[2])} exact {int(o[3])} flags(far<<1|interior) {int(o[4]) & 3} th_c {o[5]:.6f} "
              f"cell-sensor distance {np.hypot(cx - o[6], cy - o[7]):.4f} m")
    grid.close()
