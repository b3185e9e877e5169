"""Body of tests/test_gpu_multigpu_plumbing.py::test_native_rccl_merge_one_rank, run as `python -m tests.comm_check` in a
fresh interpreter (no torch mapped): the RCCL occupancy merge of include/tsd_comm.h through its C ABI on one GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    """include/tsd_comm.h on one GPU: a world of one rank -- communicator, extraction kernels and ncclAllReduce(int8, max)
    in stream order, merged map == this rank's own occupancy map, twice (the second extraction waits for the first
    collective through the event).  RCCL refuses two ranks on one GPU, so world sizes > 1 are the driver's multi-GPU run;
    the max-merge semantics are covered on CPU (tests/test_cpu_multigpu.py, gloo, world_size 2)."""
    import numpy as np
    from ohm_tsd_slam_amd import capi, multigpu, synth
    from oracle import pyoracle as O
    from tests import helpers as H
    gc = synth.GridConfig(9, 0.05)
    geo = synth.ScanGeometry.full_circle_360()

    def grid_with_scans(start_k):
        world = synth.World("room", gc)
        g = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
        for k in range(3):
            pose, (x, y, yaw) = H.sensor_pose(world, start_k + 4 * k)
            data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), H.MAX_RANGE, geo.angle_increment)
            g.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)
        return g

    g0 = grid_with_scans(0)
    m = multigpu.NativeOccupancyMerger(g0, 1, 0, multigpu.NativeOccupancyMerger.new_id())
    m.merge_async()
    own, _ = g0.occupancy(False, 2)
    assert np.array_equal(m.merged(), own) and (own == 100).sum() > 50
    world = synth.World("room", gc)
    pose, (x, y, yaw) = H.sensor_pose(world, 20)
    data, mask = O.ingest_f32(world.scan(x, y, yaw, geo), H.MAX_RANGE, geo.angle_increment)
    g0.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL, want_stats=False)
    m.merge_async()
    m.merge_async()
    own2, _ = g0.occupancy(False, 2)
    assert np.array_equal(m.merged(), own2) and (own2 != own).any()
    m.close()
    print("comm_check ok: world of one rank, two merges, merged map == own occupancy map")



if __name__ == "__main__":
    main()
